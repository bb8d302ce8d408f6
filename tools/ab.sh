#!/bin/bash
# Times several builds of the library on one box, one after the other, twice round.
# Usage: tools/ab.sh "<quick_bench args>" lib1 lib2 ...   ("" = the in-tree library)
args=$1; shift
for round in 1 2; do
  for lib in "$@"; do
    if [ -n "$lib" ]; then export RT_HIP_LIB=$lib; else unset RT_HIP_LIB; fi
    echo "== ${lib:-in-tree}"
    timeout -k 10 100 python3 tools/quick_bench.py $args 2>&1 | grep -E "kernel [0-9]" | tail -n 2
  done
done
