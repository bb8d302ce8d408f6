"""Developer probe: how long one step of a wave takes when the GPU is otherwise empty.  The
benchmark scene inside a huge diffuse sphere: no path escapes, every path runs max_depth
segments, so a frame of P paths takes (launch overhead) + max_depth steps of ceil(P / 128) waves.
    python tools/step_latency_probe.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ray_tracing_amd import abi, hip, host  # noqa: E402

hip.init(0)
world = host.create_random_world(1, (-11, 11))
s = np.concatenate([world.spheres, world.spheres[:1]])
s[-1]["x"] = s[-1]["y"] = s[-1]["z"] = 0.0
s[-1]["radius"] = 5000.0
scene = hip.Scene(s, world.materials)
stream = torch.cuda.current_stream().cuda_stream
for w, h, spp in ((8, 8, 2), (8, 8, 1), (4, 2, 1), (64, 32, 2), (256, 128, 2), (512, 256, 4)):
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    res = []
    for depth in (50, 25):
        frame = hip.make_frame(cam.m, w, h, spp, depth, 1, abi.RT_FLAG_SEQUENTIAL_SUM)
        for _ in range(6):
            hip.render_rows_device_async(scene, frame, 0, h, False, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        res.append((float(hip.get_launch_ms(5).mean()), st.rays / st.samples))
    (t50, r50), (t25, r25) = res
    print("%4d paths (%dx%d @ %d): %.3f ms at depth 50 (%.1f rays/sample), %.3f at 25 -> %.2f us per step"
          % (w * h * spp, w, h, spp, t50, r50, t25, (t50 - t25) / 25 * 1e3), flush=True)

# with the diagnostic build (RT_HIP_LIB=.../librt_hip_diag.so): where a lone wave's step goes
ph = hip.get_phase_cycles().astype(float)
if ph[4]:
    for w, h, spp in ((8, 8, 1), (2, 2, 1)):
        cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
        out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
        frame = hip.make_frame(cam.m, w, h, spp, 50, 1, abi.RT_FLAG_SEQUENTIAL_SUM)
        for _ in range(3):
            hip.render_rows_device_async(scene, frame, 0, h, False, out.data_ptr(), stream)
        torch.cuda.synchronize()
        ph = hip.get_phase_cycles().astype(float)
        steps = ph[4]
        print("%d paths: %.0f steps; cycles per step: refill %.0f scan %.0f replay %.0f shade %.0f; tail steps %.0f at %.0f"
              % (w * h * spp, steps, ph[0] / steps, ph[1] / steps, ph[2] / steps, ph[3] / steps, ph[9],
                 ph[12] / max(ph[9], 1)), flush=True)
