"""Shapes and sizes at the seams of the GPU path: one-pixel images, one sample, sphere counts on
either side of the LDS chunk size, scenes with no or one sphere.  Every case is held to the
oracle's bits (sequential sums) with the matrix filter, the vector filter, the literal arithmetic
and the hierarchy."""
import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi, hip

pytestmark = pytest.mark.gpu

SEQ = abi.RT_FLAG_SEQUENTIAL_SUM
MODES = (SEQ, SEQ | abi.RT_FLAG_SCAN_VALU, SEQ | abi.RT_FLAG_STRICT_FP, SEQ | abi.RT_FLAG_ACCEL)


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


def check(s, m, w, h, spp, depth=50, seed=5, tile=(64, 64)):
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, max(w, 1) / max(h, 1))
    tiles = O.job_queue(w, h, *tile)
    _, _, ctr, ref = O.render(s, m, cam, w, h, spp, depth, seed, mode=O.RO_MODE_ITER_MUL,
                              threads=8, tiles=tiles, want_sums=True)
    for flags in MODES:
        got = hip.render_tiles(s, m, cam, w, h, spp, depth, seed, tiles, flags=flags)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (w, h, spp, flags)
        st = hip.get_stats()
        assert (st.rays, st.samples) == (ctr.rays, ctr.samples), (w, h, spp, flags)
    # sliced samples (the default) reproduce themselves and stay close to the sequential sums
    a = hip.render_tiles(s, m, cam, w, h, spp, depth, seed, tiles, flags=0)
    b = hip.render_tiles(s, m, cam, w, h, spp, depth, seed, tiles, flags=0)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    ok = ~np.isnan(ref)
    assert np.allclose(a[ok], ref[ok], rtol=4e-6, atol=1e-5)


@pytest.mark.parametrize("w,h,spp", [(1, 1, 1), (1, 1, 37), (1, 17, 3), (19, 1, 3), (7, 5, 1),
                                     (65, 33, 2), (64, 64, 1), (3, 2, 200)])
def test_small_and_ragged_frames(w, h, spp):
    s, m = O.random_world(1, -4, 4)
    check(s, m, w, h, spp)


@pytest.mark.parametrize("tile", [(1, 1), (5, 3), (64, 7), (200, 200)])
def test_tile_shapes(tile):
    s, m = O.random_world(1, -4, 4)
    check(s, m, 33, 21, 2, tile=tile)


# 1136 spheres is the vector filter's chunk, 512 rows the matrix filter's (one of the four big
# spheres kept first, the ground, has no row: 513 spheres are 512 rows and a second chunk that
# holds nothing but the listed sphere)
@pytest.mark.parametrize("n", [0, 1, 2, 5, 511, 512, 513, 514, 1024, 1025, 1135, 1136, 1137, 1537,
                               2016, 2272, 2273, 4033])
def test_sphere_counts_around_the_lds_chunk(n):
    s, m = O.random_world(1, -33, 33)        # 4360 spheres to cut from
    keep = np.r_[len(s) - 4:len(s), 0:len(s) - 4][:n]   # the four big ones first
    check(s[np.sort(keep)], m, 24, 14, 2)


def test_max_depth_one_and_two():
    s, m = O.random_world(1, -4, 4)
    for depth in (1, 2):
        check(s, m, 32, 18, 3, depth=depth)


def test_depth_zero_writes_only_the_pixels_its_tiles_cover():
    """Two calls share one buffer at interleaved row blocks (what a rank's blocks look like at
    full-frame offsets).  shoot_ray(ray, 0) is black (src/main.rs:192-194), so a call at
    max_depth 0 sets ITS rows to zero; the rows between them belong to someone else."""
    import torch
    w, h = 32, 16
    s, m = O.random_world(1, -4, 4)
    scene = hip.Scene(s, m)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    out = torch.full((h, w, 3), 5.0, dtype=torch.float32, device="cuda:0")
    tiles = np.array([(0, 0, w, 4), (0, 8, w, 4)], dtype=abi.TILE_DTYPE)
    offs = np.array([0, 8 * w], dtype=np.uint64)
    strides = np.full(2, w, dtype=np.uint32)
    stream = torch.cuda.current_stream().cuda_stream
    for resolve in (False, True):
        out.fill_(5.0)
        for flags in (0, SEQ):
            frame = hip.make_frame(cam, w, h, 4, 0, 1, flags)
            hip.render_device_async(scene, frame, tiles, offs, strides, resolve, out.data_ptr(), stream)
        torch.cuda.synchronize()
        got = out.cpu().numpy()
        assert np.all(got[0:4] == 0.0) and np.all(got[8:12] == 0.0)
        assert np.all(got[4:8] == 5.0) and np.all(got[12:16] == 5.0)
        assert hip.get_stats().samples == 2 * 4 * w * 4
    scene.close()


# Frames so small that the work runs out at once: every wave is draining from its first steps, the
# live paths of a wave go through the one-pass forms of the matrix filter (up to 16 and up to 64
# paths) and items lend their last sample to idle path slots.  Items of one, two and nine samples
# (3, 100 and 500 samples per pixel).  The hierarchy kernel runs one path per lane and lends
# nothing, so its sliced frame checks the lending kernels bit for bit; the oracle checks them all
# through the sequential order.
@pytest.mark.parametrize("w,h,spp", [(64, 36, 3), (48, 27, 100), (16, 9, 500), (5, 3, 100)])
def test_frames_whose_work_runs_out_at_once(w, h, spp):
    s, m = O.random_world(1, -11, 11)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    tiles = O.job_queue(w, h, 64, 64)
    frames = {}
    for name, flags in (("matrix", 0), ("vector", abi.RT_FLAG_SCAN_VALU),
                        ("literal", abi.RT_FLAG_STRICT_FP), ("hierarchy", abi.RT_FLAG_ACCEL)):
        frames[name] = hip.render_tiles(s, m, cam, w, h, spp, 50, 9, tiles, flags=flags)
        st = hip.get_stats()
        frames[name + " rays"] = (st.rays, st.samples, st.slices)
    assert frames["hierarchy rays"][2] > 1      # sliced: the mode in which samples are lent
    for name in ("matrix", "vector", "literal"):
        assert np.array_equal(frames[name].view(np.uint32), frames["hierarchy"].view(np.uint32)), name
        assert frames[name + " rays"] == frames["hierarchy rays"], name
    _, _, ctr, ref = O.render(s, m, cam, w, h, spp, 50, 9, mode=O.RO_MODE_ITER_MUL, threads=8,
                              tiles=tiles, want_sums=True)
    assert frames["matrix rays"][:2] == (ctr.rays, ctr.samples)
    ok = ~np.isnan(ref)
    assert np.allclose(frames["matrix"][ok], ref[ok], rtol=4e-6, atol=2e-5 * max(1, spp // 100))
