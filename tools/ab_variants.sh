#!/bin/bash
# A/B of library variants inside ONE GPU-box call (boxes differ by a few percent):
#   tools/ab_variants.sh <out-file> <command...> -- <lib> <lib> ...
# runs <command> once per library with RT_HIP_LIB pointing at it.
set -e -o pipefail
out=gpurun_out/$1
shift
cmd=()
while [ "$1" != "--" ]; do cmd+=("$1"); shift; done
shift
: > "$out"
for lib in "$@"; do
  echo "== $lib" >> "$out"
  RT_HIP_LIB=$PWD/ray_tracing_amd/$lib timeout -k 10 250 "${cmd[@]}" 2>&1 | grep -v amdgpu.ids >> "$out"
done
cat "$out"
