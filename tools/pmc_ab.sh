#!/bin/bash
# A/B of instruction counters between two builds of the library on one box.
# Usage: tools/pmc_ab.sh <name> <lib path or ""> [quick_bench args...]
set -o pipefail
name=$1; lib=$2; shift 2
out=gpurun_out/pmc_$name
mkdir -p "$out"
export TMPDIR=/tmp
if [ -n "$lib" ]; then export RT_HIP_LIB=$lib; fi
pass() {
  p=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" --output-format csv -d "$out/$p" -- python3 tools/quick_bench.py $ARGS > "$out/$p.log" 2>&1 || return 1
}
ARGS="$*"
pass a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU &&
pass b SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
tot = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "trace_kernel" not in k: continue
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in tot.items():
    disp = [r for r in c]
    print(k[:60])
    # launches: count via dispatch ids
for f in glob.glob(out + "/*/**/*counter_collection.csv", recursive=True):
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]: ids[r["Counter_Name"]].add(r["Dispatch_Id"])
    for cn, s in ids.items(): n[cn] = len(s)
for k, c in tot.items():
    for cn in sorted(c): print("  %-28s %.4e per launch (%d launches)" % (cn, c[cn] / max(n[cn], 1), n[cn]))
PY
