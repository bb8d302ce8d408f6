"""The N > 1 path on the CPU: row sharding, gather to rank 0 and reassembly, with the gloo
backend and world sizes 2 and 3.  The renderer is replaced by a function that writes a pattern
into the rows it is asked for, so the test checks exactly the plumbing that surrounds the
kernel on a multi-GPU node."""
import socket

import numpy as np
import pytest

from ray_tracing_amd import multi_gpu


def pattern(y, x, c):
    return (y * 1000 + x) * 4 + c


def _worker(rank, world, port, width, height, out_path):
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    plan = multi_gpu.RowPlan(width, height, world)
    sharded = multi_gpu.ShardedFrame(plan, torch.device("cpu"), rank)
    calls = []

    def fake_render(tiles, offs, strides, out):
        calls.append(len(tiles))
        flat = out.view(-1, 3)
        for t, off, stride in zip(tiles, offs, strides):
            for yy in range(int(t["height"])):
                for xx in range(int(t["width"])):
                    i = int(off) + yy * int(stride) + xx
                    for c in range(3):
                        flat[i, c] = pattern(int(t["y"]) + yy, int(t["x"]) + xx, c)

    for _ in range(2):      # buffers are reusable across frames
        fb = sharded.render_and_gather(fake_render)
    if rank == 0:
        np.save(out_path, fb.numpy())
    else:
        assert fb is None
    dist.barrier()
    dist.destroy_process_group()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,width,height", [(2, 12, 37), (3, 5, 8), (2, 7, 16)])
def test_rows_are_sharded_gathered_and_reassembled(tmp_path, world, width, height):
    import torch.multiprocessing as mp
    out = str(tmp_path / "frame.npy")
    mp.spawn(_worker, args=(world, free_port(), width, height, out), nprocs=world, join=True)
    fb = np.load(out)
    assert fb.shape == (height, width, 3)
    y, x, c = np.meshgrid(np.arange(height), np.arange(width), np.arange(3), indexing="ij")
    assert np.array_equal(fb, pattern(y, x, c).astype(np.float32))


def test_row_plan_covers_every_row_once():
    for world in (1, 2, 4, 8):
        for h in (1, 7, 8, 9, 225, 1080, 2160):
            plan = multi_gpu.RowPlan(16, h, world)
            seen = np.zeros(h, dtype=int)
            for r in range(world):
                tiles, offs, strides = plan.tiles_for(r)
                for t, off in zip(tiles, offs):
                    seen[t["y"]:t["y"] + t["height"]] += 1
                    assert t["x"] == 0 and t["width"] == 16
                    assert off + t["height"] * 16 <= plan.rows_per_rank * 16
                assert plan.rows_of(r) == int(tiles["height"].sum()) if len(tiles) else plan.rows_of(r) == 0
            assert np.all(seen == 1)
            m = plan.frame_row_map()
            assert len(set(m.tolist())) == h
            # balance: no rank has more than one block more than another
            rows = [plan.rows_of(r) for r in range(world)]
            assert max(rows) - min(rows) <= plan.block_rows


def _sample_split_worker(rank, world, port, out_path):
    """Every rank renders the whole (tiny) frame for its share of the samples - here with the
    CPU oracle standing in for the kernel - and rank 0 adds the gathered sums in rank order."""
    import torch
    import torch.distributed as dist
    import oracle_lib as O
    from ray_tracing_amd import host
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank,
                            world_size=world)
    w, h, spp, depth, seed = 20, 12, 7, 50, 5
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = host.create_random_world(1, (-4, 4))
    d = host.Dispatch.new(host.create_job_queue(w, h, 64, 64), w, h, depth, spp, camera, scene)
    tile = np.array([(0, 0, w, h)], dtype=multi_gpu.abi.TILE_DTYPE)

    def accumulate_fn(begin, n, sums):
        flat = sums.view(-1, 3).numpy()
        O.accumulate_tiles(scene.spheres, scene.materials, camera.m, w, h, depth, seed, tile,
                           begin, n, flat)

    fb = multi_gpu.dispatch_work_gpu_sample_split(d, seed=seed, accumulate_fn=accumulate_fn,
                                                  device=torch.device("cpu"))
    if rank == 0:
        np.save(out_path, fb)
    else:
        assert fb is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_samples_are_split_gathered_and_added_in_rank_order(tmp_path, world):
    import torch.multiprocessing as mp
    import oracle_lib as O
    from ray_tracing_amd import host
    assert multi_gpu.sample_ranges(7, 3) == [(0, 2), (2, 4), (4, 7)]
    assert multi_gpu.sample_ranges(2, 4) == [(0, 0), (0, 1), (1, 1), (1, 2)]
    out = str(tmp_path / "frame.npy")
    mp.spawn(_sample_split_worker, args=(world, free_port(), out), nprocs=world, join=True)
    fb = np.load(out)
    w, h, spp, depth, seed = 20, 12, 7, 50, 5
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    scene = host.create_random_world(1, (-4, 4))
    tile = np.array([(0, 0, w, h)], dtype=multi_gpu.abi.TILE_DTYPE)
    total = None
    for begin, end in multi_gpu.sample_ranges(spp, world):
        part = np.zeros((w * h, 3), dtype=np.float32)
        O.accumulate_tiles(scene.spheres, scene.materials, camera.m, w, h, depth, seed, tile,
                           begin, end - begin, part)
        total = part if total is None else total + part
    with np.errstate(invalid="ignore"):
        expect = np.sqrt(total * np.float32(1.0 / spp)).reshape(h, w, 3)
    expect = np.where(expect > np.float32(0.999), np.float32(0.999), expect)
    ok = ~np.isnan(expect)
    assert np.array_equal(np.isnan(expect), np.isnan(fb))
    assert np.allclose(fb[ok], expect[ok], rtol=0, atol=1e-7)
