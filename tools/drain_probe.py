"""Developer tool: what the waves do after the frame's work runs out, on one rank's eighth of the
1080p frame.  Load a diagnostic build: RT_HIP_LIB=ray_tracing_amd/librt_hip_diag.so (phase shares,
drain steps) or librt_hip_diagdrain.so (the phases' wave-cycles inside the drain).
Usage: python tools/drain_probe.py [flags]"""
import sys, numpy as np, torch
spp = 100
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import hip, host
hip.init(0)
world = host.create_random_world(1, (-11, 11))
cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, 1920 / 1080)
scene = hip.Scene(world.spheres, world.materials)
import time
# a rank's share of an 8-GPU run: rows dealt in blocks of 8
from ray_tracing_amd import multi_gpu
W, H = 1920, 1080
plan = multi_gpu.RowPlan(W, H, 8)
frame = hip.make_frame(cam.m, W, H, spp, 50, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tiles, offs, strides = plan.tiles_for(0)
out = torch.zeros((plan.rows_per_rank, W, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), stream)
    torch.cuda.synchronize()
st = hip.get_stats()
ph = hip.get_phase_cycles().astype(np.float64)
waves = max(ph[13], 1)
print("kernel %.3f ms, steps/wave %.1f" % (st.kernel_ms, ph[4] / waves))
tot = ph[:4].sum()
if "diagdrain" not in os.environ.get("RT_HIP_LIB", ""):
    print("shares refill %.1f scan %.1f replay %.1f shade %.1f" % tuple(100 * ph[i] / tot for i in range(4)))
print("drain: mean %.0f kcyc  longest %.0f kcyc  longest-lived wave %.0f kcyc" % (ph[5] / waves / 1e3, ph[6] / 1e3, ph[7] / 1e3))
if "diagdrain" not in os.environ.get("RT_HIP_LIB", ""):
    print("drain steps that scanned: mean %.1f max %.0f; tail steps: mean %.1f max %.0f; cycles per tail step %.0f"
          % (ph[8] / waves, ph[10], ph[9] / waves, ph[11], ph[12] / max(ph[9], 1)))
else:
    print("from the first wave's start: first wave out of work at %.3f ms, last at %.3f ms, last wave ends at %.3f ms; "
          "most steps a wave took after it ran out of work: %d" % (ph[8] / 1e5, ph[9] / 1e5, ph[10] / 1e5, ph[11]))
if "diagdrain" in os.environ.get("RT_HIP_LIB", ""):
    print("drain kcycles per wave: refill %.0f  hit %.0f  replay %.0f  shade %.0f"
          % (ph[0] / waves / 1e3, ph[1] / waves / 1e3, ph[2] / waves / 1e3, ph[3] / waves / 1e3))
