"""Developer tool (diag-drain build: RT_HIP_LIB=ray_tracing_amd/librt_hip_diagdrain.so): when every wave
of a rank's eighth of the 1080p frame ran out of work and ended, and the steps it took in between.
Usage: python tools/wave_log.py [flags]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracing_amd import hip, host, multi_gpu  # noqa: E402

hip.init(0)
world = host.create_random_world(1, (-11, 11))
cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, 1920 / 1080)
scene = hip.Scene(world.spheres, world.materials)
W, H = 1920, 1080
plan = multi_gpu.RowPlan(W, H, 8)
frame = hip.make_frame(cam.m, W, H, 100, 50, 1, int(sys.argv[1]) if len(sys.argv) > 1 else 0)
tiles, offs, strides = plan.tiles_for(0)
out = torch.zeros((plan.rows_per_rank, W, 3), dtype=torch.float32, device="cuda:0")
stream = torch.cuda.current_stream().cuda_stream
for rep in range(3):
    hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), stream)
    torch.cuda.synchronize()
st = hip.get_stats()
L = hip.lib()
log = np.zeros((4096, 16), dtype=np.uint64)
n = ctypes.c_uint32(0)
rc = L.rt_debug_wave_log(log.ctypes.data_as(ctypes.c_void_p), 4096, ctypes.byref(n))
assert rc == 0, rc
log = log[: n.value].astype(np.int64)
t0 = log[:, 0].min()
start, dry, end = (log[:, 0] - t0) / 100.0, (log[:, 1] - t0) / 100.0, (log[:, 2] - t0) / 100.0   # us
steps = log[:, 4] + log[:, 5]
print("kernel %.3f ms; %d waves; out of work at %.0f-%.0f us, ended at %.0f-%.0f us"
      % (st.kernel_ms, n.value, dry.min(), dry.max(), end.min(), end.max()))
drain = end - dry
order = np.argsort(-end)
print("drain per wave: mean %.0f us, median %.0f, 90%% %.0f, 99%% %.0f, max %.0f; steps mean %.1f max %d"
      % (drain.mean(), np.median(drain), np.percentile(drain, 90), np.percentile(drain, 99), drain.max(),
         steps.mean(), steps.max()))
print("the 25 waves that ended last:  wave  wg  ended[us]  drain[us]  steps (scan/compact/tail)  us/step   kcycles: refill hit replay shade(tail)")
for w in order[:25]:
    print("   %5d %4d  %8.0f  %8.0f   %3d (%d/%d/%d)   %.2f    %d %d %d %d(%d)"
          % (w, w // 16, end[w], drain[w], steps[w], log[w, 4], log[w, 7], log[w, 5], drain[w] / max(steps[w], 1),
             log[w, 8] // 1000, log[w, 9] // 1000, log[w, 10] // 1000, log[w, 11] // 1000, log[w, 12] // 1000))
# how the drain time relates to the steps
for lo, hi in ((0, 20), (20, 40), (40, 60), (60, 80), (80, 200)):
    m = (steps >= lo) & (steps < hi)
    if m.any():
        print("waves with %3d-%3d steps after the work ran out: %4d, drain %.0f us on average, %.2f us per step"
              % (lo, hi, m.sum(), drain[m].mean(), (drain[m] / np.maximum(steps[m], 1)).mean()))
# waves still running over time
for t in np.arange(dry.max(), end.max(), 100.0):
    print("t = %5.0f us: %4d waves running on %3d CUs" % (t, int((end > t).sum()), len(set((np.nonzero(end > t)[0] // 16).tolist()))))
