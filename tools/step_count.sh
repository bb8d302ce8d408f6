#!/bin/bash
# Instruction counters per step of a lone wave (tools/step_count_probe.py), one counter pass.
# Usage: tools/step_count.sh <out-dir-under-gpurun_out>
set -e -o pipefail
out=gpurun_out/$1
mkdir -p "$out"
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_INST_ANY \
  --output-format csv -d "$out/pmc" -- python3 tools/step_count_probe.py > "$out/pmc.log" 2>&1
python3 tools/step_count_summary.py "$out"
