"""Developer probe: what a step of a wave that is out of work costs as a function of how many such
waves share a compute unit.  The benchmark scene inside a huge diffuse sphere (every path runs
max_depth segments), frames of P pixels at 100 spp: a wave reserves one pixel (50 items of two
samples, i.e. 50 live paths for 100 steps at max_depth 50) and the frame's work is gone at once,
so every wave of the launch is draining from its first step.  P = 256: one wave per compute unit;
1024: one per SIMD; 4096: four per SIMD.
    python tools/step_load_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ray_tracing_amd import hip, host  # noqa: E402

hip.init(0)
world = host.create_random_world(1, (-11, 11))
s = np.concatenate([world.spheres, world.spheres[:1]])
s[-1]["x"] = s[-1]["y"] = s[-1]["z"] = 0.0
s[-1]["radius"] = 5000.0
scene = hip.Scene(s, world.materials)
stream = torch.cuda.current_stream().cuda_stream
for w, h in ((16, 16), (32, 16), (32, 32), (64, 32), (64, 64), (128, 64)):
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    res = []
    for depth in (50, 25):
        frame = hip.make_frame(cam.m, w, h, 100, depth, 1, 0)
        for _ in range(6):
            hip.render_rows_device_async(scene, frame, 0, h, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        res.append((float(hip.get_launch_ms(5).mean()), st.rays / st.samples, st.slices))
    (t50, r50, sl), (t25, r25, _) = res
    # two samples per item, one after the other: 2 x depth steps
    print("%5d pixels (%d slices): %.3f ms at depth 50 (%.1f rays/sample), %.3f at 25 -> %.2f us per step"
          % (w * h, sl, t50, r50, t25, (t50 - t25) / 50 * 1e3), flush=True)
