"""Progressive rendering: several passes adding samples into one buffer of un-normalised sums
(rt_accumulate_device_async, rt_resolve_device_async).

The reference renders a pixel's samples in one loop (src/cpu/job.rs:48-61) and divides by the
sample count afterwards (src/cpu/mod.rs:67); cutting that loop into passes must not change what
is summed.  With RT_FLAG_SEQUENTIAL_SUM the passes carry the running sum on in sample order and
the bar is bit identity with a one-pass render; without it each pass adds its own (sliced) sum,
and the bar is exact additivity plus agreement with the one-pass render to rounding.
"""
import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi, hip

SEQ = abi.RT_FLAG_SEQUENTIAL_SUM
STRICT = abi.RT_FLAG_STRICT_FP


def scene_and_camera(w, h):
    s, m = O.random_world(1, -4, 4)
    return s, m, O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)


def test_oracle_sample_loop_can_be_cut_anywhere():
    w, h = 40, 24
    s, m, cam = scene_and_camera(w, h)
    tiles = O.job_queue(w, h, 16, 16)
    _, _, ctr, ref = O.render(s, m, cam, w, h, 7, 50, 9, mode=O.RO_MODE_ITER_MUL, threads=1,
                              tiles=tiles, want_sums=True)
    for cuts in [(7,), (1, 6), (3, 1, 3), (1, 1, 1, 1, 1, 1, 1)]:
        acc = np.zeros_like(ref)
        begin, rays = 0, 0
        for n in cuts:
            rays += O.accumulate_tiles(s, m, cam, w, h, 50, 9, tiles, begin, n, acc).rays
            begin += n
        assert np.array_equal(acc.view(np.uint32), ref.view(np.uint32))
        assert rays == ctr.rays


def device_tiles(tiles):
    """Concatenated-tile addressing (the JobResult layout) for the device entry points."""
    sizes = tiles["width"].astype(np.uint64) * tiles["height"].astype(np.uint64)
    off = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.uint64)
    return off, tiles["width"].astype(np.uint32), int(sizes.sum())


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [SEQ, SEQ | STRICT])
def test_sequential_passes_are_bit_identical_to_one_render(flags):
    import torch
    hip.init(0)
    w, h, depth, seed = 72, 40, 50, 4
    s, m, cam = scene_and_camera(w, h)
    tiles = O.job_queue(w, h, 32, 32)
    off, stride, npx = device_tiles(tiles)
    _, _, ctr, ref = O.render(s, m, cam, w, h, 9, depth, seed, mode=O.RO_MODE_ITER_MUL, threads=8,
                              tiles=tiles, want_sums=True)
    one = hip.render_tiles(s, m, cam, w, h, 9, depth, seed, tiles, flags=flags)
    assert np.array_equal(one.view(np.uint32), ref.view(np.uint32))
    scene = hip.Scene(s, m)
    acc = torch.zeros((npx, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    begin, rays = 0, 0
    for n in (2, 1, 4, 2):
        frame = hip.make_frame(cam, w, h, n, depth, seed, flags)
        hip.accumulate_device_async(scene, frame, begin, tiles, off, stride, acc.data_ptr(), stream)
        torch.cuda.synchronize()
        rays += hip.get_stats().rays
        begin += n
    got = acc.cpu().numpy()
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert rays == ctr.rays
    # resolve on the device == compose_final_framebuffer on the concatenated sums
    rgb = torch.empty_like(acc)
    hip.resolve_device_async(acc.data_ptr(), npx, 9, rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    expect = hip.compose(tiles, ref, w, h, 9)
    back = hip.compose(tiles, got, w, h, 9)
    assert np.array_equal(back.view(np.uint32), expect.view(np.uint32))
    # the same arithmetic element by element (the tile scatter aside)
    x = got * np.float32(1.0 / np.float32(9))
    with np.errstate(invalid="ignore"):
        elem = np.sqrt(x)
    elem = np.where(elem < 0, np.float32(0), np.where(elem > np.float32(0.999), np.float32(0.999), elem))
    r = rgb.cpu().numpy()
    both = ~(np.isnan(elem) | np.isnan(r))
    assert np.array_equal(np.isnan(elem), np.isnan(r))
    assert np.array_equal(r[both].view(np.uint32), elem[both].astype(np.float32).view(np.uint32))
    scene.close()


@pytest.mark.gpu
def test_host_pointer_passes_are_bit_identical_to_one_render():
    hip.init(0)
    w, h, depth, seed = 48, 27, 50, 6
    s, m, cam = scene_and_camera(w, h)
    tiles = O.job_queue(w, h, 32, 32)
    _, _, _, ref = O.render(s, m, cam, w, h, 7, depth, seed, mode=O.RO_MODE_ITER_MUL, threads=8,
                            tiles=tiles, want_sums=True)
    sums = np.zeros_like(ref)
    begin = 0
    for n in (3, 4):
        hip.accumulate_tiles(s, m, cam, w, h, n, depth, seed, tiles, begin, sums, flags=SEQ)
        begin += n
    assert np.array_equal(sums.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_sliced_passes_add_exactly_and_agree_with_one_render():
    import torch
    hip.init(0)
    w, h, depth, seed = 64, 36, 50, 7
    s, m, cam = scene_and_camera(w, h)
    tiles = O.job_queue(w, h, 64, 64)
    off, stride, npx = device_tiles(tiles)
    scene = hip.Scene(s, m)
    stream = torch.cuda.current_stream().cuda_stream

    def one_pass(into, begin, n):
        frame = hip.make_frame(cam, w, h, n, depth, seed, 0)
        hip.accumulate_device_async(scene, frame, begin, tiles, off, stride, into.data_ptr(), stream)
        torch.cuda.synchronize()
        assert hip.get_stats().slices > 1       # small image: samples are split over lanes
        return into

    a = one_pass(torch.zeros((npx, 3), dtype=torch.float32, device="cuda:0"), 0, 16)
    b = one_pass(torch.zeros((npx, 3), dtype=torch.float32, device="cuda:0"), 16, 24)
    # a pass on an empty buffer is a render of that many samples ...
    first = hip.render_tiles(s, m, cam, w, h, 16, depth, seed, tiles, flags=0)
    assert np.array_equal(a.cpu().numpy().view(np.uint32), first.view(np.uint32))
    # ... and a pass on a filled buffer adds its sum in one f32 addition
    c = one_pass(a.clone(), 16, 24)
    ab = (a + b).cpu().numpy()
    got = c.cpu().numpy()
    same = ~np.isnan(ab)
    assert np.array_equal(np.isnan(ab), np.isnan(got))
    assert np.array_equal(got[same].view(np.uint32), ab[same].view(np.uint32))
    # the same 40 samples as one sequential render, up to the association of the sums
    _, _, _, ref = O.render(s, m, cam, w, h, 40, depth, seed, mode=O.RO_MODE_ITER_MUL, threads=8,
                            tiles=tiles, want_sums=True)
    ok = ~np.isnan(ref)
    assert np.allclose(got[ok], ref[ok], rtol=4e-6, atol=1e-5)
    scene.close()


@pytest.mark.gpu
def test_accumulate_rejects_bad_arguments():
    import torch
    hip.init(0)
    s, m, cam = scene_and_camera(16, 9)
    scene = hip.Scene(s, m)
    tiles = O.job_queue(16, 9, 64, 64)
    off, stride, npx = device_tiles(tiles)
    buf = torch.zeros((npx, 3), dtype=torch.float32, device="cuda:0")
    frame = hip.make_frame(cam, 16, 9, 4, 50, 1, 0)
    with pytest.raises(hip.RtError):
        hip.accumulate_device_async(scene, frame, 0xFFFFFFFE, tiles, off, stride, buf.data_ptr())
    with pytest.raises(hip.RtError):
        hip.accumulate_device_async(scene, frame, 0, tiles, off, stride, 0)
    with pytest.raises(hip.RtError):
        hip.resolve_device_async(buf.data_ptr(), npx, 0, buf.data_ptr())
    scene.close()


@pytest.mark.gpu
def test_host_progressive_dispatch_matches_one_pass_and_honours_a_time_budget():
    hip.init(0)
    from ray_tracing_amd import host
    w, h, depth, seed = 96, 54, 50, 2
    camera = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), 20.0, w / h)
    world = host.create_random_world(1, (-4, 4))
    queue = host.create_job_queue(w, h, 64, 64)
    d = host.Dispatch.new(queue, w, h, depth, 12, camera, world)
    one = host.dispatch_work_gpu(d, 1, seed=seed, flags=SEQ)
    seen = []
    fb, done = host.dispatch_work_gpu_progressive(d, [4, 4, 4], seed=seed, flags=SEQ,
                                                  on_pass=lambda n, sums: seen.append(n))
    assert done == 12 and seen == [4, 8, 12]
    via_host = host.dispatch_work_gpu_progressive_host(d, [5, 7], seed=seed, flags=SEQ)
    ok_h = ~(np.isnan(one) | np.isnan(via_host))
    assert np.array_equal(np.isnan(one), np.isnan(via_host))
    assert np.array_equal(via_host[ok_h].view(np.uint32), one[ok_h].view(np.uint32))
    both = ~(np.isnan(one) | np.isnan(fb))
    assert np.array_equal(np.isnan(one), np.isnan(fb))
    assert np.array_equal(fb[both].view(np.uint32), one[both].view(np.uint32))
    # a budget that is spent after the first pass: one pass runs, the image is resolved with
    # the sample count that was rendered
    fb1, done1 = host.dispatch_work_gpu_progressive(d, [3] * 50, seed=seed, flags=SEQ,
                                                    time_budget_s=0.0)
    assert done1 == 3
    d3 = host.Dispatch.new(queue, w, h, depth, 3, camera, world)
    one3 = host.dispatch_work_gpu(d3, 1, seed=seed, flags=SEQ)
    ok = ~(np.isnan(one3) | np.isnan(fb1))
    assert np.array_equal(fb1[ok].view(np.uint32), one3[ok].view(np.uint32))
