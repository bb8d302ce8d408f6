#!/bin/bash
# Re-takes every figure kept under profiles/round3 with the library as built, on the GPU box:
# the four configurations (bench line, kernel trace, PMC passes), a rank's shares, the drain's
# probes and wave log, the CPU port table, all configurations.  Results go to gpurun_out/refresh_*;
# tools/refresh_profiles_collect.sh (run in the build container) condenses them into profiles/round3.
# Before the call, remove gpurun_out/refresh_* in the build container: what an earlier run left there
# would be condensed together with the new results.
# Usage (GPU box): bash tools/refresh_profiles.sh [part]     part: a (profiles), b (the rest); default both
set -o pipefail
export TMPDIR=/tmp
part=${1:-ab}
if [[ $part == *a* ]]; then
  for c in rtiow488_1080p_100spp_d50 rtiow10k_1080p_64spp_d50 rtiow488_2160p_500spp_d50 glass488_1080p_1024spp_d50; do
    bash tools/profile_bench.sh refresh_$c --config $c > gpurun_out/refresh_$c.log 2>&1 || { echo FAILED $c; tail -5 gpurun_out/refresh_$c.log; exit 1; }
    echo ok $c
  done
fi
if [[ $part == *b* ]]; then
  o=gpurun_out/refresh
  (timeout -k 10 120 python3 tools/rank_share_bench.py 0 100 20 2>&1 | grep -E "^N=") > ${o}_rs1.txt &&
  (timeout -k 10 120 python3 tools/rank_share_bench.py 0 100 20 2 2>&1 | grep -E "^N=") > ${o}_rs2.txt &&
  (RT_HIP_LIB=$PWD/ray_tracing_amd/librt_hip_diagdrain.so timeout -k 10 120 python3 tools/wave_log.py 2>&1 | grep -v amdgpu) > ${o}_wave_log.txt &&
  (RT_HIP_LIB=$PWD/ray_tracing_amd/librt_hip_diag.so timeout -k 10 120 python3 tools/drain_probe.py 2>&1 | grep -v amdgpu) > ${o}_drain_probe.txt &&
  (RT_HIP_LIB=$PWD/ray_tracing_amd/librt_hip_diagdrain.so timeout -k 10 120 python3 tools/drain_probe.py 2>&1 | grep -v amdgpu) >> ${o}_drain_probe.txt &&
  (timeout -k 10 120 python3 tools/step_latency_probe.py 2>&1 | grep paths) > ${o}_step_latency.txt &&
  (timeout -k 10 500 python3 tools/cpu_gpu_table.py 20 2>&1 | grep "^|") > ${o}_cpu_gpu_table.txt &&
  (bash tools/all_configs.sh refresh_all_configs.txt > /dev/null 2>&1) || { echo FAILED part b; exit 1; }
  echo ok part b
fi
