#!/bin/bash
# A/B of one environment variable inside ONE GPU-box call:
#   tools/ab_env.sh <out-file> <VAR> <command...> -- <value> <value> ...   ("-" = unset)
set -e -o pipefail
out=gpurun_out/$1; var=$2
shift 2
cmd=()
while [ "$1" != "--" ]; do cmd+=("$1"); shift; done
shift
: > "$out"
for v in "$@"; do
  echo "== $var=$v" >> "$out"
  if [ "$v" = "-" ]; then
    timeout -k 10 250 "${cmd[@]}" 2>&1 | grep -v amdgpu.ids >> "$out"
  else
    env "$var=$v" timeout -k 10 250 "${cmd[@]}" 2>&1 | grep -v amdgpu.ids >> "$out"
  fi
done
cat "$out"
