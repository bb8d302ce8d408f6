"""Developer probe, to be run under rocprofv3 --pmc (tools/step_count.sh): frames whose paths all
run max_depth segments (the benchmark scene inside a huge diffuse sphere) at depths 50 and 25, for
8 and 64 paths.  The difference of a counter between the two depths, over 25, is what one step of
the one busy wave costs.  Launch order: (8 paths: d50, d25) x 3, then (64 paths: d50, d25) x 3."""
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
for w, h in ((4, 2), (8, 8)):
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    for _ in range(3):
        for depth in (50, 25):
            frame = hip.make_frame(cam.m, w, h, 1, depth, 1, abi.RT_FLAG_SEQUENTIAL_SUM)
            hip.render_rows_device_async(scene, frame, 0, h, False, out.data_ptr(), stream)
            torch.cuda.synchronize()
print("done")
