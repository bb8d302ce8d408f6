"""One rank's 1/8 share with its row blocks issued in different orders (kernel time)."""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from ray_tracing_amd import hip, host, multi_gpu
W, H = 1920, 1080
hip.init(0)
world = host.create_random_world(1, (-11, 11))
cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, W / H)
scene = hip.Scene(world.spheres, world.materials)
frame = hip.make_frame(cam.m, W, H, 100, 50, 1, 0)
plan = multi_gpu.RowPlan(W, H, 8)
tiles, offs, strides = plan.tiles_for(0)
n = len(tiles)
stream = torch.cuda.current_stream().cuda_stream
out = torch.zeros((plan.rows_per_rank, W, 3), dtype=torch.float32, device="cuda:0")
mid = sorted(range(n), key=lambda k: abs(k - 0.45 * n))
orders = {"ascending (as issued now)": list(range(n)), "descending": list(range(n))[::-1],
          "middle out": mid, "middle last": mid[::-1]}
# per-block cost, to see which blocks are the expensive ones
for name, order in orders.items():
    order = np.array(order)
    t, o, s = tiles[order], offs[order], strides[order]
    for _ in range(6):
        hip.render_device_async(scene, frame, t, o, s, True, out.data_ptr(), stream)
    torch.cuda.synchronize()
    ms = hip.get_launch_ms(5)
    print("%-28s kernel %.3f ms (min %.3f)" % (name, ms.mean(), ms.min()), flush=True)
costs = []
for k in range(n):
    for _ in range(2):
        hip.render_device_async(scene, frame, tiles[k:k+1], offs[k:k+1], strides[k:k+1], True, out.data_ptr(), stream)
    torch.cuda.synchronize()
    costs.append((hip.get_stats().rays / hip.get_stats().samples))
print("rays/sample per block, bottom to top:", " ".join("%.2f" % c for c in costs))
