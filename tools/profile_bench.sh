#!/bin/bash
# Profiles `bench.py` on the GPU box: the bench line itself, a kernel trace, and the PMC counter
# passes (each counter group in a run of its own; never together with a trace).
# Usage: tools/profile_bench.sh <out-dir-under-gpurun_out> [bench args...]
set -e -o pipefail
out=gpurun_out/$1
shift
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 300 python3 bench.py --steps 5 --warmup 1 "$@" > "$out/bench.json" 2> "$out/bench.err"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- \
  python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > "$out/kt.log" 2>&1
pmc() {
  name=$1
  shift
  timeout -s KILL 300 rocprofv3 --pmc "$@" --output-format csv -d "$out/$name" -- \
    python3 bench.py --steps 2 --warmup 0 --no-cpu-baseline ${BENCH_ARGS} > "$out/$name.log" 2>&1
}
BENCH_ARGS="$*"
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc sqA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_FMA_F32 SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_WAVES
pmc sqB SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
# the matrix pipe (a pass of its own, last: an unknown counter name ends rocprofv3 at once)
pmc sqC SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM || true
echo done
