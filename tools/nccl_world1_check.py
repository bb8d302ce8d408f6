"""Plumbing check on a one-GPU box: RCCL (backend "nccl") initialised with world_size 1 in the
same process as librt_hip.so, then the collectives bench.py uses (gather, barrier, all_reduce)
on a frame rendered by the library.  Shows that the two share one HIP runtime and that the
calls are well-formed; says nothing about multi-GPU performance."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from ray_tracing_amd import hip, host, multi_gpu  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
device = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
hip.init(0)
W, H = 320, 180
world = host.create_random_world(1, (-4, 4))
cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, W / H)
scene = hip.Scene(world.spheres, world.materials)
frame = hip.make_frame(cam.m, W, H, 8, 50, 1, 0)
plan = multi_gpu.RowPlan(W, H, 1)
streams = [torch.cuda.Stream(device=device) for _ in range(2)]
frames = []
shardeds = []   # kept until the streams are drained: a buffer freed while its stream still works on it
                # would be handed to the next frame by torch's caching allocator
for k in range(4):
    sharded = multi_gpu.ShardedFrame(plan, device, 0)
    shardeds.append(sharded)
    with torch.cuda.stream(streams[k % 2]):
        raw = streams[k % 2].cuda_stream

        def render_fn(tiles, offs, strides, out):
            hip.render_device_async(scene, frame, tiles, offs, strides, True, out.data_ptr(), raw)

        if len(sharded.tiles):
            render_fn(sharded.tiles, sharded.offsets, sharded.strides, sharded.local)
        # the call bench.py makes, N = 1: gather into the slices of ShardedFrame's one buffer
        dist.gather(sharded.local, sharded.gathered, dst=0)
        frames.append(sharded.stacked.reshape(-1, W, 3).index_select(0, sharded.row_map))
dist.barrier()
torch.cuda.synchronize()
t = torch.tensor([1.5], dtype=torch.float64, device=device)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
ref = torch.zeros((H, W, 3), dtype=torch.float32, device=device)
hip.render_rows_device_async(scene, frame, 0, H, True, ref.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
for f in frames:
    assert torch.equal(f.view(torch.int32), ref.view(torch.int32))
assert float(t.item()) == 1.5
dist.destroy_process_group()
print("nccl world-1 check ok: 4 frames on 2 streams gathered through RCCL equal the direct render")
