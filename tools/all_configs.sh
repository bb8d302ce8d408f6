#!/bin/bash
# Kernel time and roofline fraction of every benchmark configuration (and the reference's own
# 68-sphere scene and one rank's share of the 1080p frame at 8 GPUs), one line per launch.
set -e -o pipefail
out=gpurun_out/$1
mkdir -p "$(dirname "$out")"
: > "$out"
run() { timeout -k 10 280 python3 tools/quick_bench.py "$@" >> "$out" 2>&1; }
run 11 1920 1080 100 0 3
run 11 3840 2160 100 0 2
run 11 3840 2160 500 0 1
run 50 1920 1080 64 0 2
run 11 1920 1080 128 0 2 glass
run 11 1920 1080 1024 0 1 glass
run 4 400 225 16 0 3
run 11 1920 136 100 0 3
# the filter kept on the vector ALU (flags = 8) on the three 1080p scenes and the rank share
run 11 1920 1080 100 8 2
run 50 1920 1080 64 8 2
run 11 1920 1080 128 8 2 glass
run 11 1920 136 100 8 3
# the opt-in hierarchy traversal (flags = 4) on the three 1080p scenes
run 11 1920 1080 100 4 2
run 50 1920 1080 64 4 2
run 11 1920 1080 128 4 2 glass
cat "$out"
