"""Developer probe: kernel time of the whole 1080p frame and of one rank's 1/8 share at several
max_depth values.  If the gap between the share and an eighth of the frame were the drain of
deep paths, it would close as max_depth falls."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ray_tracing_amd import hip, host, multi_gpu  # noqa: E402

W, H = 1920, 1080
hip.init(0)
world = host.create_random_world(1, (-11, 11))
cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, W / H)
scene = hip.Scene(world.spheres, world.materials)
stream = torch.cuda.current_stream().cuda_stream
full = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
for depth in (50, 20, 10, 5, 2):
    frame = hip.make_frame(cam.m, W, H, 100, depth, 1, 0)
    res = {}
    for n in (1, 8):
        plan = multi_gpu.RowPlan(W, H, n)
        tiles, offs, strides = plan.tiles_for(0)
        for _ in range(6):
            hip.render_device_async(scene, frame, tiles, offs, strides, True, full.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        res[n] = (float(hip.get_launch_ms(5).mean()), st.rays)
    (t1, r1), (t8, r8) = res[1], res[8]
    print("max_depth %2d: frame %.2f ms, share %.3f ms, share's rays / frame's %.4f -> by work %.3f ms, "
          "gap %.3f ms" % (depth, t1, t8, r8 / r1, t1 * r8 / r1, t8 - t1 * r8 / r1), flush=True)
