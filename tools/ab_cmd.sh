#!/bin/bash
# Runs one command with several builds of the library on one box, twice round.
# Usage: tools/ab_cmd.sh "<command>" "<grep pattern>" lib1 lib2 ...   ("" = the in-tree library)
cmd=$1; pat=$2; shift 2
for round in 1 2; do
  for lib in "$@"; do
    if [ -n "$lib" ]; then export RT_HIP_LIB=$lib; else unset RT_HIP_LIB; fi
    echo "== ${lib:-in-tree}"
    timeout -k 10 200 bash -c "$cmd" 2>&1 | grep -E "$pat"
  done
done
