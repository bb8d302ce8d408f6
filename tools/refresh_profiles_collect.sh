#!/bin/bash
# Condenses what tools/refresh_profiles.sh left under gpurun_out/ into profiles/round3 (run in the
# build container, in the repository root, at the commit the library was built from).
# Usage: bash tools/refresh_profiles_collect.sh <commit>
set -e
H=$1
o=gpurun_out/refresh
for c in rtiow488_1080p_100spp_d50 rtiow10k_1080p_64spp_d50 rtiow488_2160p_500spp_d50 glass488_1080p_1024spp_d50; do
  python3 tools/summarize_profile.py gpurun_out/refresh_$c profiles/round3/$c
  echo $H > profiles/round3/${c}_profiled_commit.txt
done
(echo "# one rank's share of the 1920x1080 @ 100 spp frame on one MI355X (tools/rank_share_bench.py 0 100 20), library of commit $H"
 echo "# one frame at a time:"; cat ${o}_rs1.txt
 echo "# two frames in flight on alternating streams (tools/rank_share_bench.py 0 100 20 2): wall/step is the figure, and the first"
 echo "# line's carries the first frames' start-up: compare the N = 8 line's with the one-at-a-time frame above"
 cat ${o}_rs2.txt) > profiles/round3/rank_share_1080p.txt
(echo "# tools/cpu_gpu_table.py 20 on the GPU box (16 host threads), library of commit $H"; cat ${o}_cpu_gpu_table.txt) > profiles/round3/cpu_port_vs_gpu_all_configs.md
(echo "# tools/all_configs.sh, library of commit $H, one box"
 grep -E "^N=|^phase|^drain|^replay|^hierarchy" gpurun_out/refresh_all_configs.txt) > profiles/round3/quick_bench_all_configs.txt
(echo "# the drain on a rank's eighth of the 1080p frame, library of commit $H"
 echo "# tools/drain_probe.py with the diag build, then with the diag-drain build"; cat ${o}_drain_probe.txt
 echo "# tools/step_latency_probe.py (production build): a step of a wave alone on the GPU"; cat ${o}_step_latency.txt
 echo "# tools/wave_log.py (diag-drain build, slower than the production build): every wave's time and steps after the frame's work ran out"
 cat ${o}_wave_log.txt) > profiles/round3/drain_1080p_share_of_8.txt
(echo "# registers, scratch and spills of every trace_kernel instantiation as built (tools/kernel_usage.sh), commit $H"
 echo "# <STRICT,MULTI,ACCEL,MFMA>"; bash tools/kernel_usage.sh 2>/dev/null | grep trace_kernel) > profiles/round3/kernel_resource_usage.txt
make -C ray_tracing_amd/csrc asm > /dev/null 2>&1
(echo "# static instruction counts per phase of trace_kernel<false,false,false,true> (tools/phase_instructions.py; build with RT_ASM_MARKS), commit $H"
 python3 tools/phase_instructions.py trace_kernelILb0ELb0ELb0ELb1 | grep -v "^#"
 echo "# (scan: both passes of the main path and the drain's paths - one ray block with four blocks of spheres per trip, four ray blocks in one pass - each with its own column exchange and replay, all between the same marks; refill: with the lending of an item's last sample)") > profiles/round3/phase_instructions_default_kernel.txt
echo collected at $H
