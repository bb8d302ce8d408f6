"""Adversarial per-ray comparison: scenes and rays chosen to make the discriminant noisy
(silhouette rays, origins on and inside spheres, directions of every length, scenes far from the
origin, radii over five decades).  Three ways to find the closest hit must agree bit for bit -
the literal arithmetic (STRICT_FP), the conservative scan in front of it (default) and the
hierarchy (ACCEL) - and the literal one must equal the CPU oracle."""
import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi, hip

pytestmark = pytest.mark.gpu

STRICT = abi.RT_FLAG_STRICT_FP
ACCEL = abi.RT_FLAG_ACCEL


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


def random_scene(rs, n, centre, spread, r_lo, r_hi):
    mats = []
    for _ in range(12):
        kind = rs.integers(0, 3)
        col = rs.uniform(0.05, 0.95, size=3)
        mats.append((*col, int(kind == 1), float(rs.uniform(0, 1)), int(kind == 2),
                     float(rs.uniform(1.1, 2.0))))
    m = O.materials_array(mats)
    c = np.asarray(centre, dtype=np.float64) + rs.uniform(-spread, spread, size=(n, 3))
    r = np.exp(rs.uniform(np.log(r_lo), np.log(r_hi), size=n))
    s = O.spheres_array([(c[i, 0], c[i, 1], c[i, 2], r[i], int(rs.integers(0, len(mats))))
                         for i in range(n)])
    return s, m


def unit(v):
    return v / np.linalg.norm(v, axis=1, keepdims=True)


def adversarial_rays(rs, s, n_each):
    c_all = np.stack([s["x"], s["y"], s["z"]], axis=1).astype(np.float64)
    r_all = s["radius"].astype(np.float64)
    spread = np.abs(c_all - c_all.mean(axis=0)).max() + r_all.max()
    parts = []

    def pick():
        i = rs.integers(0, len(s), size=n_each)
        return c_all[i], r_all[i]

    # silhouette rays, a hair inside or outside, from near and far
    c, r = pick()
    away = unit(rs.normal(size=(n_each, 3)))
    org = c + away * (r * (1.0 + 10.0 ** rs.uniform(-2, 3, n_each)))[:, None]
    t = rs.normal(size=(n_each, 3))
    t = unit(t - (t * away).sum(axis=1, keepdims=True) * away)
    nudge = 1.0 + rs.choice([-1, 1], size=n_each) * 10.0 ** rs.uniform(-7, -2, n_each)
    parts.append((org, c + t * (r * nudge)[:, None] - org))
    # origins on a surface (as after a bounce), leaving, entering and along the tangent plane
    c, r = pick()
    nrm = unit(rs.normal(size=(n_each, 3)))
    org = c + nrm * r[:, None]
    tang = rs.normal(size=(n_each, 3))
    tang = unit(tang - (tang * nrm).sum(axis=1, keepdims=True) * nrm)
    lift = rs.choice([-1, 1], size=n_each) * 10.0 ** rs.uniform(-7, 0, n_each)
    parts.append((org, tang + nrm * lift[:, None]))
    # origins inside a sphere
    c, r = pick()
    org = c + unit(rs.normal(size=(n_each, 3))) * (r * rs.uniform(0, 0.999, n_each))[:, None]
    parts.append((org, rs.normal(size=(n_each, 3))))
    # from outside the scene towards it
    org = c_all.mean(axis=0) + unit(rs.normal(size=(n_each, 3))) * spread * rs.uniform(2, 50, (n_each, 1))
    tgt = c_all[rs.integers(0, len(s), size=n_each)] + rs.normal(size=(n_each, 3)) * r_all.mean()
    parts.append((org, tgt - org))

    org = np.concatenate([p[0] for p in parts])
    d = np.concatenate([p[1] for p in parts])
    d = unit(d) * 10.0 ** rs.uniform(-3, 3, size=(len(d), 1))    # directions of every length
    rays = np.zeros(len(org), dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = org.astype(np.float32).T
    rays["dx"], rays["dy"], rays["dz"] = d.astype(np.float32).T
    return rays


def assert_same(a, b, nan_equal=False):
    for f in a.dtype.names:
        x, y = a[f], b[f]
        if x.dtype.kind != "f":
            assert np.array_equal(x, y), f
            continue
        same = x.view(np.uint32) == y.view(np.uint32)
        if nan_equal:   # a NaN's sign and payload differ between the CPU and the GPU
            same |= np.isnan(x) & np.isnan(y)
        assert same.all(), (f, int((~same).sum()), int(np.flatnonzero(~same)[0]))


SCENES = [
    # n, centre, spread, r_lo, r_hi, checked against the oracle too
    (300, (0, 0, 0), 10.0, 0.01, 2.0, True),
    (300, (0, 0, 0), 1000.0, 0.001, 100.0, True),
    (200, (1e4, 1e4, -1e4), 50.0, 0.1, 5.0, True),       # f32 spacing of 1e-3 at the centres
    (300, (3, -2, 1), 2.0, 0.2, 1.5, True),              # heavily overlapping spheres
    (2600, (0, 0, 0), 50.0, 0.05, 1.0, False),           # more spheres than one LDS chunk holds
]


@pytest.mark.parametrize("case", range(len(SCENES)))
def test_three_ways_to_the_closest_hit_agree(case):
    n, centre, spread, r_lo, r_hi, with_oracle = SCENES[case]
    rs = np.random.default_rng(100 + case)
    s, m = random_scene(rs, n, centre, spread, r_lo, r_hi)
    rays = adversarial_rays(rs, s, 1500)
    scene = hip.Scene(s, m)
    strict = hip.trace_rays(scene, rays, seed=5, flags=STRICT)
    assert_same(hip.trace_rays(scene, rays, seed=5, flags=0), strict)
    assert_same(hip.trace_rays(scene, rays, seed=5, flags=ACCEL), strict)
    scene.close()
    assert strict["hit"].sum() > 0.2 * len(rays)
    if with_oracle:
        ref = O.bounce(O.make_world(s, m), rays, seed=5, pow5_mode=O.RO_POW5_MUL)
        assert_same(strict, ref, nan_equal=True)


@pytest.mark.parametrize("flags", [0, STRICT, ACCEL])
def test_image_sums_on_a_random_overlapping_scene(flags):
    """Whole paths (max_depth 50) through a scene of random materials and overlapping spheres of
    very different sizes: per-pixel sums in the reference's sample order equal the oracle's."""
    rs = np.random.default_rng(42)
    s, m = random_scene(rs, 150, (0, 0.5, 0), 4.0, 0.05, 3.0)
    w, h, spp = 64, 36, 6
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 30.0, w / h)
    tiles = O.job_queue(w, h, 64, 64)
    _, _, ctr, ref = O.render(s, m, cam, w, h, spp, 50, 9, mode=O.RO_MODE_ITER_MUL, threads=8,
                              tiles=tiles, want_sums=True)
    got = hip.render_tiles(s, m, cam, w, h, spp, 50, 9, tiles,
                           flags=flags | abi.RT_FLAG_SEQUENTIAL_SUM)
    same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
    assert same.all()
    st = hip.get_stats()
    assert st.rays == ctr.rays and st.samples == ctr.samples
    assert ctr.rays > 3 * ctr.samples      # paths do bounce around in there


def test_materials_whose_constants_the_host_prepares():
    """The library forms a dielectric's 1/ir and reflectance's r0, a metal's colour squared and its
    fuzz on the host and stores them in the sphere's record (src/main.rs:199-242 computes them per
    hit).  Materials at the edges of that - both flags set (dielectric wins, src/main.rs:201),
    ir of 1, below 1, zero, negative and huge, metallic_strength outside [0, 1], colours that are
    negative or above 1 (create_random_world draws them from [-0.9, 0.9), src/main.rs:349-358) -
    must still scatter with the oracle's bits, whichever filter runs."""
    rs = np.random.default_rng(77)
    mats = [
        (0.3, 0.6, 0.9, 1, 0.25, 1, 1.5),      # both flags
        (0.3, 0.6, 0.9, 0, 0.0, 1, 1.0),       # ir 1: r0 = 0
        (0.3, 0.6, 0.9, 0, 0.0, 1, 0.66),      # ir below 1
        (0.3, 0.6, 0.9, 0, 0.0, 1, 0.0),       # 1 / ir infinite
        (0.3, 0.6, 0.9, 0, 0.0, 1, -1.5),      # negative ir: 1 + ir negative
        (0.3, 0.6, 0.9, 0, 0.0, 1, -1.0),      # r0's division by zero
        (0.3, 0.6, 0.9, 0, 0.0, 1, 3.0e38),    # 1 + ir rounds, 1 / ir denormal
        (0.9, -0.9, 1.7, 1, 0.0, 0, 1.5),      # metal: fuzz 1, colours squared
        (0.2, 0.4, 0.8, 1, 1.0, 0, 1.5),       # metal: fuzz 0
        (0.2, 0.4, 0.8, 1, 1.75, 0, 1.5),      # metal: fuzz negative
        (0.2, 0.4, 0.8, 1, -0.5, 0, 1.5),      # metal: fuzz above 1
        (-0.9, 0.0, 0.89, 0, 0.3, 0, 1.5),     # diffuse with a negative colour
    ]
    m = O.materials_array(mats)
    n = 240
    c = rs.uniform(-6, 6, size=(n, 3))
    r = rs.uniform(0.2, 1.2, size=n)
    s = O.spheres_array([(c[i, 0], c[i, 1], c[i, 2], r[i], i % len(mats)) for i in range(n)])
    rays = adversarial_rays(rs, s, 2500)
    scene = hip.Scene(s, m)
    strict = hip.trace_rays(scene, rays, seed=11, flags=STRICT)
    assert_same(hip.trace_rays(scene, rays, seed=11, flags=0), strict)
    assert_same(hip.trace_rays(scene, rays, seed=11, flags=ACCEL), strict)
    scene.close()
    ref = O.bounce(O.make_world(s, m), rays, seed=11, pow5_mode=O.RO_POW5_MUL)
    assert_same(strict, ref, nan_equal=True)
    # every material was hit, from outside and from inside
    hit = strict["hit"] == 1
    assert set(np.unique(strict["material"][hit])) == set(range(len(mats)))
    assert (strict["front_face"][hit] == 0).sum() > 100
