"""World::hit's sphere filter on the matrix pipe against the same filter on the vector ALU and
against the literal arithmetic.

A filter only chooses which spheres get the reference's exact Sphere::hit arithmetic
(src/main.rs:128-150); it is correct iff it never drops a sphere that arithmetic would accept.
So every comparison here is bit for bit: per ray (hit, sphere, t, point, normal, scatter
direction, attenuation, draws) and per pixel.  The scenes are chosen to reach every branch of
the matrix form: spheres outside its f16 range (listed for every ray), scenes at very small and
very large scales, rays whose origin is beyond its range (literal loop), candidate lists that
overflow in the middle of a block, more spheres than one LDS chunk, scenes it refuses."""
import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi, hip
from test_gpu_fuzz import adversarial_rays, assert_same, random_scene
from test_gpu_parity import default_camera, make_rays

pytestmark = pytest.mark.gpu

STRICT = abi.RT_FLAG_STRICT_FP
VALU = abi.RT_FLAG_SCAN_VALU
SEQ = abi.RT_FLAG_SEQUENTIAL_SUM


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


def three_ways(s, m, rays, seed=5, matrix=True, listed=None):
    scene = hip.Scene(s, m)
    info = scene.scan_info()
    assert bool(info.matrix_filter) == matrix
    if listed is not None:
        assert info.listed_spheres == listed
    strict = hip.trace_rays(scene, rays, seed=seed, flags=STRICT)
    assert_same(hip.trace_rays(scene, rays, seed=seed, flags=0), strict)
    assert_same(hip.trace_rays(scene, rays, seed=seed, flags=VALU), strict)
    scene.close()
    return strict, info


def test_benchmark_scene_lists_the_ground_and_scales_by_eight():
    s, m = O.random_world(1, -11, 11)
    cam = default_camera()
    rays = make_rays((s, m), cam, 1500, seed=3)
    strict, info = three_ways(s, m, rays, listed=1)
    assert info.scale == 8.0                  # |c| <= 15.6 -> below 128
    assert strict["hit"].sum() > 0.5 * len(rays)


@pytest.mark.parametrize("factor", [2.0 ** -12, 1e-3, 1.0, 37.0, 2.0 ** 14])
def test_scenes_of_every_size(factor):
    """The same adversarial scene (silhouette rays, origins on and inside spheres, directions of
    every length) at scales from 2.4e-4 to 16384: the host's power-of-two scale brings each of
    them to the same place in f16."""
    rs = np.random.default_rng(7)
    s, m = random_scene(rs, 300, (0.5, -1.0, 2.0), 10.0, 0.01, 2.0)
    s = s.copy()
    for f in ("x", "y", "z", "radius"):
        s[f] = (s[f].astype(np.float64) * factor).astype(np.float32)
    rays = adversarial_rays(rs, s, 1200)
    strict, info = three_ways(s, m, rays)
    assert 32.0 < info.scale * np.abs(np.stack([s["x"], s["y"], s["z"]])).max() <= 128.0
    assert strict["hit"].sum() > 0.2 * len(rays)


def test_rays_from_far_outside_and_beyond_the_filters_range():
    """Origins from 10 to 100 000 scene radii away: up to origin_limit they go through the
    matrix filter with a slack that grows with |o|^2, beyond it through the literal loop."""
    rs = np.random.default_rng(11)
    s, m = random_scene(rs, 200, (0, 0, 0), 10.0, 0.2, 2.0)
    scene = hip.Scene(s, m)
    limit = scene.scan_info().origin_limit
    scene.close()
    n = 4000
    c = np.stack([s["x"], s["y"], s["z"]], axis=1).astype(np.float64)
    tgt = c[rs.integers(0, len(s), n)] + rs.normal(size=(n, 3)) * 0.5
    away = rs.normal(size=(n, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    dist = 10.0 ** rs.uniform(2, 6, size=(n, 1))
    org = away * dist
    rays = np.zeros(n, dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = org.astype(np.float32).T
    rays["dx"], rays["dy"], rays["dz"] = (tgt - org).astype(np.float32).T
    assert (dist < 0.5 * limit).sum() > 500 and (dist > 2 * limit).sum() > 500
    strict, _ = three_ways(s, m, rays)
    assert strict["hit"].sum() > 100


def test_rays_from_just_inside_the_origin_limit_do_not_pass_the_padding_rows():
    """300 spheres leave four rows of their last block of 16, and the read-ahead block behind it,
    without a sphere: rows that no ray may pass.  The rays that come closest to passing them are
    those whose threshold has the most slack taken off, i.e. origins at the filter's limit, on a
    line through the middle of the scene (nothing across the ray to add to the threshold)."""
    rs = np.random.default_rng(31)
    s, m = random_scene(rs, 300, (0, 0, 0), 10.0, 0.2, 2.0)
    scene = hip.Scene(s, m)
    limit = scene.scan_info().origin_limit
    scene.close()
    n = 20000
    away = rs.normal(size=(n, 3))
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    org = away * (limit * rs.uniform(0.96, 0.9999, size=(n, 1)))
    tgt = rs.normal(size=(n, 3)) * 3.0
    rays = np.zeros(n, dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = org.astype(np.float32).T
    rays["dx"], rays["dy"], rays["dz"] = (tgt - org).astype(np.float32).T
    strict, _ = three_ways(s, m, rays)
    assert strict["hit"].sum() > 0.5 * n


def test_zero_and_non_finite_rays_take_the_literal_loop():
    s, m = O.random_world(1, -4, 4)
    rays = np.zeros(8, dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = 13, 2, 3
    rays["dx"], rays["dy"], rays["dz"] = -13, -2, -3
    rays["dx"][1] = rays["dy"][1] = rays["dz"][1] = 0.0
    rays["dx"][2] = np.inf
    rays["dy"][3] = np.nan
    rays["ox"][4] = np.nan
    rays["ox"][5] = 3e38
    rays["dx"][6] = 1e-30; rays["dy"][6] = 0.0; rays["dz"][6] = 0.0
    scene = hip.Scene(s, m)
    strict = hip.trace_rays(scene, rays, seed=1, flags=STRICT)
    got = hip.trace_rays(scene, rays, seed=1, flags=0)
    scene.close()
    for f in ("hit", "sphere"):
        assert np.array_equal(got[f], strict[f]), f
    same = (got["t"].view(np.uint32) == strict["t"].view(np.uint32)) | (np.isnan(got["t"]) & np.isnan(strict["t"]))
    assert same.all()


def test_eight_listed_spheres_and_a_full_list():
    """Eight giants stand a factor 4 above the rest: all eight are listed for every ray, which
    overflows a wave's 512-entry list before the scan has started (8 x 64 paths and more)."""
    rs = np.random.default_rng(21)
    s, m = random_scene(rs, 150, (0, 0, 0), 8.0, 0.1, 1.0)
    s = s.copy()
    for k in range(8):
        ang = 2 * np.pi * k / 8
        s[k] = (2000 * np.cos(ang), -1500.0 + 10 * k, 2000 * np.sin(ang), 1400.0 + k, s[k]["material_id"])
    rays = adversarial_rays(rs, s[8:], 1200)
    strict, _ = three_ways(s, m, rays, listed=8)
    assert np.isin(strict["sphere"][strict["hit"] == 1], np.arange(8)).sum() > 50


def test_nine_giants_switch_the_matrix_filter_to_its_widest_scale():
    """More outliers than can be listed: no gap of a factor 4 within the largest eight, so
    nothing is listed and the scale follows the giants (the small spheres then sit in the low
    bits of f16: more false candidates, same answers)."""
    rs = np.random.default_rng(22)
    s, m = random_scene(rs, 150, (0, 0, 0), 8.0, 0.1, 1.0)
    s = s.copy()
    for k in range(9):
        s[k] = (300.0 * (k - 4), -1000.0, 50.0 * k, 990.0 + k, s[k]["material_id"])
    rays = adversarial_rays(rs, s[9:], 1000)
    three_ways(s, m, rays, listed=0)


def test_candidate_list_overflows_in_the_middle_of_a_block():
    """600 nearly concentric shells: every ray through the middle is a candidate for every
    sphere, 16 x 128 entries per block against a list of 512."""
    rs = np.random.default_rng(23)
    n = 600
    mats = O.materials_array([(0.5, 0.5, 0.5, 0, 0.0, 0, 1.5), (0.8, 0.8, 0.8, 1, 0.3, 0, 1.5),
                              (1.0, 1.0, 1.0, 0, 0.0, 1, 1.5)])
    s = O.spheres_array([(rs.normal() * 0.01, rs.normal() * 0.01, rs.normal() * 0.01,
                          1.0 + 0.002 * i, int(rs.integers(0, 3))) for i in range(n)])
    nr = 2000
    org = rs.normal(size=(nr, 3))
    org = org / np.linalg.norm(org, axis=1, keepdims=True) * rs.uniform(0.0, 4.0, size=(nr, 1))
    rays = np.zeros(nr, dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = org.astype(np.float32).T
    rays["dx"], rays["dy"], rays["dz"] = (rs.normal(size=(nr, 3)) * 0.3 - org).astype(np.float32).T
    strict, _ = three_ways(s, mats, rays, listed=0)
    assert strict["hit"].sum() > 0.9 * nr
    # and whole paths through it (both paths of a lane alive, lists shared by 128 paths)
    w, h, spp = 48, 27, 4
    cam = O.camera_new((0.0, 0.5, 6.0), (0, 0, 0), 40.0, w / h)
    a = hip.render(s, mats, cam, w, h, spp, 12, 3, flags=SEQ | STRICT)
    b = hip.render(s, mats, cam, w, h, spp, 12, 3, flags=SEQ)
    assert hip.get_stats().threads_per_workgroup == 1024
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all()


def test_more_spheres_than_one_chunk_of_rows():
    rs = np.random.default_rng(24)
    s, m = random_scene(rs, 1300, (0, 0, 0), 30.0, 0.05, 1.5)    # three chunks of 512 rows
    rays = adversarial_rays(rs, s, 1500)
    three_ways(s, m, rays)


def test_an_empty_world_and_a_single_sphere():
    mats = O.materials_array([(0.5, 0.5, 0.5, 0, 0.0, 0, 1.5)])
    cam = default_camera(16 / 9)
    w, h = 64, 36
    empty = O.spheres_array([])
    scene = hip.Scene(empty, mats)
    assert scene.scan_info().matrix_filter == 0
    scene.close()
    one = O.spheres_array([(0.0, 0.0, 0.0, 1.0, 0)])
    scene = hip.Scene(one, mats)
    assert scene.scan_info().matrix_filter == 1 and scene.scan_info().listed_spheres == 0
    scene.close()
    for s in (empty, one):
        a = hip.render(s, mats, cam, w, h, 4, 8, 2, flags=SEQ | STRICT)
        b = hip.render(s, mats, cam, w, h, 4, 8, 2, flags=SEQ)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.parametrize("glass", [False, True])
def test_frames_equal_the_vector_scans(glass):
    """Whole frames of the benchmark scenes (max_depth 50; the deep paths bounce inside the
    ground sphere, 2000 across, with origins far outside the scene's bulk)."""
    w, h, spp = 320, 180, 16
    s, m = O.random_world(1, -11, 11, 0.1 if glass else 0.8, 0.2 if glass else 0.95, int(glass))
    cam = default_camera(w / h)
    a = hip.render(s, m, cam, w, h, spp, 50, 7, flags=VALU)
    st_a = hip.get_stats()
    b = hip.render(s, m, cam, w, h, spp, 50, 7, flags=0)
    st_b = hip.get_stats()
    assert (st_a.threads_per_workgroup, st_b.threads_per_workgroup) == (256, 1024)
    assert (st_a.rays, st_a.samples) == (st_b.rays, st_b.samples)
    same = (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))
    assert same.all()


def test_a_camera_beyond_the_filters_range_keeps_the_frame_on_the_vector_scan():
    s, m = O.random_world(1, -4, 4)
    scene = hip.Scene(s, m)
    limit = scene.scan_info().origin_limit
    scene.close()
    w, h = 64, 36
    far = O.camera_new((limit, 2.0, 3.0), (0, 0, 0), 0.2, w / h)
    hip.render(s, m, far, w, h, 2, 4, 1, flags=0)
    assert hip.get_stats().threads_per_workgroup == 256
    near = O.camera_new((13.0, 2.0, 3.0), (0, 0, 0), 20.0, w / h)
    hip.render(s, m, near, w, h, 2, 4, 1, flags=0)
    assert hip.get_stats().threads_per_workgroup == 1024
