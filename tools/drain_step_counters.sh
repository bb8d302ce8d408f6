#!/bin/bash
# tools/drain_step_counters.py under the counter passes (one run each), then its table.
set -e -o pipefail
out=gpurun_out/${1:-drain_step_counters}
mkdir -p "$out"
export TMPDIR=/tmp
pmc() {
  name=$1
  shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 tools/drain_step_counters.py > "$out/$name.log" 2>&1
}
pmc a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pmc b SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU
python3 tools/drain_step_counters.py --read "$out" | tee "$out/table.txt"
