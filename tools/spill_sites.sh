#!/bin/bash
# Where the default kernel (matrix filter, one chunk) reads spilled registers back from scratch
# memory: reloads per phase of the main loop (the `asm` build's phase marks), and by scratch offset.
# Usage: tools/spill_sites.sh     (builds ray_tracing_amd/csrc/asm first)
set -e
make -C ray_tracing_amd/csrc asm > /dev/null 2>&1
S=ray_tracing_amd/csrc/asm/rt_kernels-hip-amdgcn-amd-amdhsa-gfx950.s
K=${1:-_ZN2rt12_GLOBAL__N_112trace_kernelILb0ELb0ELb0ELb1EEEvNS_11TraceParamsE}
start=$(grep -n "^$K:" $S | cut -d: -f1)
end=$(awk -v s=$start 'NR>s && /^\.Lfunc_end/ {print NR; exit}' $S)
sed -n "${start},${end}p" $S > /tmp/k.s
grep -E "^; (ScratchSize|NumVgprs|Occupancy|codeLenInByte)" /tmp/k.s || sed -n "${end},$((end+40))p" $S | grep -E "ScratchSize|NumVgprs:|codeLenInByte"
awk '/RTMARK/ {phase=$3} /scratch_load/ {l[phase]++} /scratch_store/ {st[phase]++}
     END {for (p in l) printf "%-12s reloads %3d\n", p, l[p]; for (p in st) printf "%-12s spill stores %3d\n", (p == "" ? "prologue" : p), st[p]}' /tmp/k.s | sort
grep "scratch_load" /tmp/k.s | grep -o "offset:[0-9]*" | sort | uniq -c | sort -k2 -t: -n | tr '\n' ' '; echo
