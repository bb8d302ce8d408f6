#!/bin/bash
# A/B of library variants on a rank's share of the 1080p frame (N = 1, 2, 4, 8) inside ONE GPU-box
# call, twice round:   tools/ab_share.sh <out-file> <lib> <lib> ...   (names under ray_tracing_amd/)
out=gpurun_out/$1
shift
: > "$out"
for round in 1 2; do
  for lib in "$@"; do
    echo "== $lib" >> "$out"
    RT_HIP_LIB=$PWD/ray_tracing_amd/$lib timeout -k 10 120 python3 tools/rank_share_bench.py 0 100 20 2>&1 | grep -E "^N=" | sed -e 's/rays.sample.*ideal/ideal/' >> "$out" || exit 1
  done
done
cat "$out"
