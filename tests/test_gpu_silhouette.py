"""Two million rays aimed at silhouettes, through the three ways to run World::hit's search.

A filter in front of the exact Sphere::hit arithmetic (src/main.rs:131-139) is only allowed to
drop a sphere whose discriminant `half_b * half_b - a * c`, as the reference computes it, is
negative.  The rays that can tell a filter with too little slack from a sound one are those whose
discriminant is within a few hundred roundings of zero: here every ray grazes one chosen sphere at
an impact parameter of r (1 + delta), delta a signed multiple (0.1 to 500) of the rounding unit
of the terms the discriminant is made of, so that about half of them hit by the reference's
arithmetic and half miss.  Scenes at scales from 2^-12 to 2^14, origins from just off the surface
out to the matrix filter's limit, directions from 1e-3 to 1e3 long.  The matrix filter (flags 0),
the vector filter (RT_FLAG_SCAN_VALU) and the literal arithmetic (RT_FLAG_STRICT_FP) must give the
same bits for every ray: hit, sphere, t, point, normal, scatter direction, attenuation, draws."""
import numpy as np
import pytest

from ray_tracing_amd import abi, hip
from test_gpu_fuzz import assert_same, random_scene, unit

pytestmark = pytest.mark.gpu

STRICT = abi.RT_FLAG_STRICT_FP
VALU = abi.RT_FLAG_SCAN_VALU
U = 2.0 ** -24


@pytest.fixture(scope="module", autouse=True)
def device():
    hip.init(0)
    yield


def grazing_rays(rs, s, n, limit):
    """n rays, each tangent to one sphere of `s` to within `k` rounding units of the discriminant.
    Returns (rays, index of the sphere each one grazes)."""
    c_all = np.stack([s["x"], s["y"], s["z"]], axis=1).astype(np.float64)
    r_all = s["radius"].astype(np.float64)
    i = rs.integers(0, len(s), size=n)
    c, r = c_all[i], r_all[i]
    # where the ray starts: from a hundredth of a radius off the surface to 0.9 x the origin limit
    # (measured from the World's origin, which is what the filter's limit is about)
    away = unit(rs.normal(size=(n, 3)))
    far = np.maximum(0.9 * limit - np.linalg.norm(c, axis=1) - r, 2.0 * r)
    dist = r * (1.0 + 10.0 ** rs.uniform(-2.0, np.log10(far / r)))
    org = c + away * dist[:, None]
    # the point it passes through: on the tangent plane's circle seen from the origin, i.e. at
    # distance r (1 + delta) from the centre in a direction across `away`
    t = rs.normal(size=(n, 3))
    t = unit(t - (t * away).sum(axis=1, keepdims=True) * away)
    k = 10.0 ** rs.uniform(-1.0, np.log10(500.0), size=n) * rs.choice([-1.0, 1.0], size=n)
    oc2 = dist * dist
    delta = np.clip(k * U * (oc2 + r * r) / (r * r), -0.1, 0.1)
    # a ray from org that passes the centre at distance b = r (1 + delta): aim at the point of the
    # circle of radius b around c that is seen at a right angle from org
    b = r * (1.0 + delta)
    along = b * b / dist                       # foot of the tangent point along `away`
    across = b * np.sqrt(np.maximum(1.0 - (b / dist) ** 2, 0.0))
    tgt = c + away * along[:, None] + t * across[:, None]
    d = unit(tgt - org) * 10.0 ** rs.uniform(-3, 3, size=(n, 1))
    rays = np.zeros(n, dtype=abi.RAY_IN_DTYPE)
    rays["ox"], rays["oy"], rays["oz"] = org.astype(np.float32).T
    rays["dx"], rays["dy"], rays["dz"] = d.astype(np.float32).T
    return rays, i


def reference_discriminant(s, rays, i):
    """half_b * half_b - a * c of src/main.rs:131-135 in f32, operation by operation."""
    f = np.float32
    ocx = rays["ox"] - s["x"][i]; ocy = rays["oy"] - s["y"][i]; ocz = rays["oz"] - s["z"][i]
    dx, dy, dz = rays["dx"], rays["dy"], rays["dz"]
    a = f(f(dx * dx + dy * dy) + dz * dz)
    hb = f(f(ocx * dx + ocy * dy) + ocz * dz)
    c = f(f(f(ocx * ocx + ocy * ocy) + ocz * ocz) - f(s["radius"][i] * s["radius"][i]))
    return f(f(hb * hb) - f(a * c))


@pytest.mark.parametrize("factor", [2.0 ** -12, 1e-3, 1.0, 37.0, 2.0 ** 14])
def test_grazing_rays_at_every_scale(factor):
    n = 420_000                                   # x 5 scales: 2.1e6 grazing (ray, sphere) pairs
    rs = np.random.default_rng(int(np.log2(factor) * 8) + 1000)
    s, m = random_scene(rs, 300, (0.5, -1.0, 2.0), 10.0, 0.01, 2.0)
    s = s.copy()
    for f in ("x", "y", "z", "radius"):
        s[f] = (s[f].astype(np.float64) * factor).astype(np.float32)
    scene = hip.Scene(s, m)
    info = scene.scan_info()
    assert info.matrix_filter == 1
    rays, aimed = grazing_rays(rs, s, n, info.origin_limit)
    with np.errstate(over="ignore", invalid="ignore"):
        disc = reference_discriminant(s, rays, aimed)
    # the batch is what it claims to be: both signs of the reference's discriminant, amply
    frac = float((disc >= 0).mean())
    assert 0.3 < frac < 0.7, frac
    strict = hip.trace_rays(scene, rays, seed=9, flags=STRICT)
    assert_same(hip.trace_rays(scene, rays, seed=9, flags=0), strict)
    assert_same(hip.trace_rays(scene, rays, seed=9, flags=VALU), strict)
    scene.close()
    # the grazed sphere is the closest hit for a good part of those with disc >= 0 (others are
    # hidden behind a nearer sphere), and never for a ray whose discriminant is negative
    on_target = (strict["hit"] == 1) & (strict["sphere"] == aimed)
    assert not (on_target & (disc < 0)).any()
    assert on_target.sum() > 0.1 * n


def test_grazing_rays_on_the_benchmark_scene():
    """The 488-sphere benchmark World (ground listed for every ray, scale 8): grazing rays at its
    small spheres and at the three big ones."""
    import oracle_lib as O
    s, m = O.random_world(1, -11, 11)
    rs = np.random.default_rng(77)
    scene = hip.Scene(s, m)
    info = scene.scan_info()
    rays, aimed = grazing_rays(rs, s, 300_000, info.origin_limit)
    strict = hip.trace_rays(scene, rays, seed=4, flags=STRICT)
    assert_same(hip.trace_rays(scene, rays, seed=4, flags=0), strict)
    assert_same(hip.trace_rays(scene, rays, seed=4, flags=VALU), strict)
    scene.close()
    assert ((strict["hit"] == 1) & (strict["sphere"] == aimed)).sum() > 30_000
