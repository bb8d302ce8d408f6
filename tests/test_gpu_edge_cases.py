"""Shapes and sizes at the seams of the GPU path: one-pixel images, one sample, sphere counts on
either side of the LDS chunk size, scenes with no or one sphere.  Every case is held to the
oracle's bits (sequential sums) in the default, literal and hierarchy modes."""
import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi, hip

pytestmark = pytest.mark.gpu

SEQ = abi.RT_FLAG_SEQUENTIAL_SUM
MODES = (SEQ, SEQ | abi.RT_FLAG_STRICT_FP, SEQ | abi.RT_FLAG_ACCEL)


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


@pytest.mark.parametrize("n", [0, 1, 2, 5, 2015, 2016, 2017, 4032, 4033])
def test_sphere_counts_around_the_lds_chunk(n):
    s, m = O.random_world(1, -33, 33)        # 4360 spheres to cut from
    keep = np.r_[len(s) - 4:len(s), 0:len(s) - 4][:n]   # the four big ones first
    check(s[np.sort(keep)], m, 24, 14, 2)


def test_max_depth_one_and_two():
    s, m = O.random_world(1, -4, 4)
    for depth in (1, 2):
        check(s, m, 32, 18, 3, depth=depth)
