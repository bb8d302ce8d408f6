"""Known-answer tests that pin the CPU oracle to the text of the reference.

The reference ships no tests and no golden vectors, so every expected value here is derived by
hand from the cited lines of /root/reference (Nanoteck137/ray_tracing) and written out as a
literal.  These run on the CPU only.
"""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as O
from ray_tracing_amd import abi

F32 = np.float32


def hit_one(sphere_row, origin, direction, t_min=0.001, t_max=O.FLT_MAX):
    s = O.spheres_array([sphere_row])
    rec = O.Hit()
    r = O.ray(origin, direction)
    ok = O.lib().ro_sphere_hit(abi.ptr(s), C.byref(r), t_min, t_max, C.byref(rec))
    return ok, rec


# ---- Sphere::hit, src/main.rs:128-159 ------------------------------------------------------

def test_sphere_hit_head_on():
    ok, rec = hit_one((0, 0, -3, 1, 7), (0, 0, 0), (0, 0, -1))
    assert ok == 1
    assert rec.t == 2.0
    assert rec.point.tup() == (0.0, 0.0, -2.0)
    assert rec.normal.tup() == (0.0, 0.0, 1.0)
    assert rec.front_face == 1
    assert rec.material_id == 7


def test_sphere_hit_t_scales_with_direction_length():
    # d = (0,0,-2): a=4, half_b=-6... with oc=(0,0,3): half_b = 3*-2 = -6, c=8, disc=36-32=4
    ok, rec = hit_one((0, 0, -3, 1, 0), (0, 0, 0), (0, 0, -2))
    assert ok == 1
    assert rec.t == 1.0
    assert rec.normal.tup() == (0.0, 0.0, 1.0)


def test_sphere_hit_origin_inside_takes_second_root_and_flips_normal():
    ok, rec = hit_one((0, 0, 0, 2, 0), (0, 0, 0), (0, 0, -4))
    assert ok == 1
    assert rec.t == 0.5          # r/|d| : first root -0.5 is below t_min
    assert rec.front_face == 0
    assert rec.normal.tup() == (0.0, 0.0, 1.0)   # outward (0,0,-1) flipped


def test_sphere_hit_miss_and_tangent():
    ok, _ = hit_one((0, 2, -3, 1, 0), (0, 0, 0), (0, 0, -1))    # disc = 9 - (13-1) < 0
    assert ok == 0
    ok, rec = hit_one((0, 1, -3, 1, 0), (0, 0, 0), (0, 0, -1))  # disc = 9 - 9 == 0 is a hit
    assert ok == 1 and rec.t == 3.0


def test_sphere_hit_window_is_closed_on_both_ends():
    # `root < t_min || t_max < root` rejects only strictly outside values
    ok, rec = hit_one((0, 0, -3, 1, 0), (0, 0, 0), (0, 0, -1), t_min=2.0, t_max=O.FLT_MAX)
    assert ok == 1 and rec.t == 2.0
    ok, rec = hit_one((0, 0, -3, 1, 0), (0, 0, 0), (0, 0, -1), t_min=0.001, t_max=2.0)
    assert ok == 1 and rec.t == 2.0
    # first root above t_max -> the second root is larger still -> miss
    ok, _ = hit_one((0, 0, -3, 1, 0), (0, 0, 0), (0, 0, -1), t_min=0.001, t_max=1.5)
    assert ok == 0
    # first root below t_min, second inside the window
    ok, rec = hit_one((0, 0, -3, 1, 0), (0, 0, 0), (0, 0, -1), t_min=2.5, t_max=10.0)
    assert ok == 1 and rec.t == 4.0 and rec.front_face == 0


def test_sphere_behind_the_ray_is_missed():
    ok, _ = hit_one((0, 0, 3, 1, 0), (0, 0, 0), (0, 0, -1))
    assert ok == 0


# ---- World::hit, src/main.rs:175-189 -------------------------------------------------------

def test_world_hit_closest_wins_in_any_storage_order():
    m = O.materials_array([(0.5, 0.5, 0.5, 0, 0, 0, 0)] * 2)
    for rows, expect in (([(0, 0, -3, 1, 0), (0, 0, -6, 1, 1)], 0),
                         ([(0, 0, -6, 1, 0), (0, 0, -3, 1, 1)], 1)):
        w = O.make_world(O.spheres_array(rows), m)
        rec = O.Hit()
        r = O.ray((0, 0, 0), (0, 0, -1))
        assert O.lib().ro_world_hit(C.byref(w), C.byref(r), 0.001, O.FLT_MAX, C.byref(rec)) == 1
        assert rec.t == 2.0 and rec.sphere == expect and rec.material_id == expect


def test_world_hit_equal_t_later_sphere_replaces_earlier():
    # the rejection is the strict `t_max < root` (src/main.rs:144), so a later sphere at
    # exactly the same t is accepted and overwrites the record (src/main.rs:182-185)
    m = O.materials_array([(0.5, 0.5, 0.5, 0, 0, 0, 0)] * 3)
    w = O.make_world(O.spheres_array([(0, 0, -3, 1, 0), (0, 0, -3, 1, 1), (0, 0, -3, 1, 2)]), m)
    rec = O.Hit()
    r = O.ray((0, 0, 0), (0, 0, -1))
    assert O.lib().ro_world_hit(C.byref(w), C.byref(r), 0.001, O.FLT_MAX, C.byref(rec)) == 1
    assert rec.sphere == 2 and rec.material_id == 2


def test_world_hit_empty_world():
    w = O.make_world(np.zeros(0, dtype=abi.SPHERE_DTYPE), np.zeros(0, dtype=abi.MATERIAL_DTYPE))
    rec = O.Hit()
    r = O.ray((0, 0, 0), (0, 0, -1))
    assert O.lib().ro_world_hit(C.byref(w), C.byref(r), 0.001, O.FLT_MAX, C.byref(rec)) == 0


# ---- reflect / refract / reflectance, src/main.rs:162-166, 312-323 --------------------------

def test_reflect():
    assert O.lib().ro_reflect(O.vec3(1, -1, 0), O.vec3(0, 1, 0)).tup() == (1.0, 1.0, 0.0)


def test_refract_normal_incidence_goes_straight():
    out = O.lib().ro_refract(O.vec3(0, -1, 0), O.vec3(0, 1, 0), 1.0 / 1.5).tup()
    assert out == (0.0, -1.0, 0.0)


def test_refract_45_degrees():
    s = float(F32(math.sqrt(0.5)))
    out = O.lib().ro_refract(O.vec3(s, -s, 0), O.vec3(0, 1, 0), float(F32(1.0) / F32(1.5))).tup()
    # perp = ratio*(uv + cos*n) = (0.4714045, 0, 0); par = -sqrt(1 - 0.2222222) n
    assert out[0] == pytest.approx(0.4714045, abs=2e-7)
    assert out[1] == pytest.approx(-0.8819171, abs=2e-7)
    assert out[2] == 0.0


@pytest.mark.parametrize("mode", [O.RO_POW5_LIBM, O.RO_POW5_MUL])
def test_reflectance(mode):
    L = O.lib()
    assert L.ro_reflectance(1.0, 1.5, mode) == pytest.approx(0.04, abs=1e-7)
    assert L.ro_reflectance(1.0, float(F32(1.0) / F32(1.5)), mode) == pytest.approx(0.04, abs=1e-7)
    assert L.ro_reflectance(0.0, 1.5, mode) == 1.0
    assert L.ro_reflectance(0.0, 0.3, mode) == 1.0


def test_pow5_by_multiplication_is_within_2ulp_of_powf():
    L = O.lib()
    xs = np.linspace(0.0, 1.0, 4001, dtype=F32)
    a = np.array([L.ro_reflectance(float(1 - x), 1.5, O.RO_POW5_LIBM) for x in xs], dtype=F32)
    b = np.array([L.ro_reflectance(float(1 - x), 1.5, O.RO_POW5_MUL) for x in xs], dtype=F32)
    ulp = np.spacing(np.maximum(np.abs(a), np.abs(b)))
    assert np.all(np.abs(a - b) <= 2 * ulp)


# ---- sky, src/main.rs:256-262 ---------------------------------------------------------------

@pytest.mark.parametrize("d,expect", [
    ((0, 1, 0), (0.5, 0.7, 1.0)),
    ((0, -1, 0), (1.0, 1.0, 1.0)),
    ((1, 0, 0), (0.75, 0.85, 1.0)),
    ((0, 5, 0), (0.5, 0.7, 1.0)),      # direction is normalised first
])
def test_sky(d, expect):
    r = O.ray((0, 0, 0), d)
    got = O.lib().ro_sky(C.byref(r)).tup()
    assert got == pytest.approx(expect, abs=1e-6)


# ---- camera, src/main.rs:57-84 ---------------------------------------------------------------

def test_camera_centre_ray():
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16.0 / 9.0)
    r = O.lib().ro_camera_get_ray(cam, 0.0, 0.0)
    assert r.origin.tup() == pytest.approx((13.0, 2.0, 3.0), rel=2e-7)
    n = math.sqrt(182.0)
    assert r.dir.tup() == pytest.approx((-13 / n, -2 / n, -3 / n), abs=2e-7)


def test_camera_corner_ray_in_view_space():
    # With position on +z looking at the origin the view rotation is the identity, so the
    # world ray equals the view-space ray normalize(u*aspect*tan(fov/2), v*tan(fov/2), -1).
    aspect = 16.0 / 9.0
    cam = O.camera_new((0, 0, 5), (0, 0, 0), 20.0, aspect)
    r = O.lib().ro_camera_get_ray(cam, 1.0, 1.0)
    tx, ty = aspect * math.tan(math.radians(10)), math.tan(math.radians(10))
    assert (tx, ty) == pytest.approx((0.3134703, 0.1763270), abs=1e-6)
    n = math.sqrt(tx * tx + ty * ty + 1)
    assert r.origin.tup() == pytest.approx((0.0, 0.0, 5.0), abs=1e-6)
    assert r.dir.tup() == pytest.approx((tx / n, ty / n, -1 / n), abs=1e-6)


def test_perspective_and_look_at_matrices():
    L = O.lib()
    m = (C.c_float * 16)()
    L.ro_mat4_perspective_rh(float(F32(math.radians(20))), float(F32(16 / 9)), 0.1, 100.0, m)
    h = 1 / math.tan(math.radians(10))
    r = 100.0 / (0.1 - 100.0)
    cols = np.array(m[:], dtype=np.float64).reshape(4, 4)   # cols[c][r]
    assert cols[0] == pytest.approx([h / (16 / 9), 0, 0, 0], rel=1e-6)
    assert cols[1] == pytest.approx([0, h, 0, 0], rel=1e-6)
    assert cols[2] == pytest.approx([0, 0, r, -1], rel=1e-6)
    assert cols[3] == pytest.approx([0, 0, r * 0.1, 0], rel=1e-6)
    L.ro_mat4_look_at_rh(O.vec3(0, 0, 5), O.vec3(0, 0, 0), O.vec3(0, 1, 0), m)
    cols = np.array(m[:], dtype=np.float64).reshape(4, 4)
    assert cols[0] == pytest.approx([1, 0, 0, 0])
    assert cols[1] == pytest.approx([0, 1, 0, 0])
    assert cols[2] == pytest.approx([0, 0, 1, 0])
    assert cols[3] == pytest.approx([0, 0, -5, 1])


def test_mat4_inverse_round_trip():
    L = O.lib()
    rng = np.random.default_rng(5)
    for _ in range(20):
        a = rng.uniform(-2, 2, size=16).astype(F32)
        inv = np.zeros(16, dtype=F32)
        assert L.ro_mat4_inverse(a.ctypes.data_as(C.POINTER(C.c_float)),
                                 inv.ctypes.data_as(C.POINTER(C.c_float))) == 1
        A = a.reshape(4, 4).T.astype(np.float64)
        I = inv.reshape(4, 4).T.astype(np.float64)
        assert np.allclose(A @ I, np.eye(4), atol=2e-3 * max(1.0, np.abs(I).max()))


def glam_scalar_inverse(a):
    """glam 0.21.2's scalar Mat4::inverse in numpy f32, written from the crate's published
    algorithm independently of the C restatement (call sites src/main.rs:65-66).  m[c][r]."""
    m = a.astype(F32).reshape(4, 4)
    d = lambda p, q, r, s: F32(F32(p * q) - F32(r * s))   # noqa: E731
    c00 = d(m[2][2], m[3][3], m[3][2], m[2][3]); c02 = d(m[1][2], m[3][3], m[3][2], m[1][3])
    c03 = d(m[1][2], m[2][3], m[2][2], m[1][3]); c04 = d(m[2][1], m[3][3], m[3][1], m[2][3])
    c06 = d(m[1][1], m[3][3], m[3][1], m[1][3]); c07 = d(m[1][1], m[2][3], m[2][1], m[1][3])
    c08 = d(m[2][1], m[3][2], m[3][1], m[2][2]); c10 = d(m[1][1], m[3][2], m[3][1], m[1][2])
    c11 = d(m[1][1], m[2][2], m[2][1], m[1][2]); c12 = d(m[2][0], m[3][3], m[3][0], m[2][3])
    c14 = d(m[1][0], m[3][3], m[3][0], m[1][3]); c15 = d(m[1][0], m[2][3], m[2][0], m[1][3])
    c16 = d(m[2][0], m[3][2], m[3][0], m[2][2]); c18 = d(m[1][0], m[3][2], m[3][0], m[1][2])
    c19 = d(m[1][0], m[2][2], m[2][0], m[1][2]); c20 = d(m[2][0], m[3][1], m[3][0], m[2][1])
    c22 = d(m[1][0], m[3][1], m[3][0], m[1][1]); c23 = d(m[1][0], m[2][1], m[2][0], m[1][1])
    fac = [np.array(v, dtype=F32) for v in (
        (c00, c00, c02, c03), (c04, c04, c06, c07), (c08, c08, c10, c11),
        (c12, c12, c14, c15), (c16, c16, c18, c19), (c20, c20, c22, c23))]
    vec = [np.array((m[1][r], m[0][r], m[0][r], m[0][r]), dtype=F32) for r in range(4)]
    sa = np.array((1, -1, 1, -1), dtype=F32)
    inv0 = ((vec[1] * fac[0] - vec[2] * fac[1]) + vec[3] * fac[2]) * sa
    inv1 = ((vec[0] * fac[0] - vec[2] * fac[3]) + vec[3] * fac[4]) * -sa
    inv2 = ((vec[0] * fac[1] - vec[1] * fac[3]) + vec[3] * fac[5]) * sa
    inv3 = ((vec[0] * fac[2] - vec[1] * fac[4]) + vec[2] * fac[5]) * -sa
    cols = np.stack([inv0, inv1, inv2, inv3]).astype(F32)
    dot0 = m[0] * cols[:, 0]
    det = F32(F32(F32(dot0[0] + dot0[1]) + dot0[2]) + dot0[3])
    return (cols * F32(F32(1.0) / det)).astype(F32).reshape(16)


def test_mat4_inverse_is_glam_scalar_operation_for_operation():
    """Bit for bit against the numpy restatement, on matrices whose entries make every rounding
    count, and on the two matrices Camera::new inverts (src/main.rs:59-66)."""
    L = O.lib()
    rng = np.random.default_rng(11)
    mats = [rng.uniform(-2, 2, size=16).astype(F32) for _ in range(200)]
    view = (C.c_float * 16)()
    L.ro_mat4_look_at_rh(O.vec3(13, 2, 3), O.vec3(0, 0, 0), O.vec3(0, 1, 0), view)
    proj = (C.c_float * 16)()
    L.ro_mat4_perspective_rh(float(F32(20.0) * (F32(math.pi) / F32(180.0))), float(F32(16 / 9)),
                             0.1, 100.0, proj)
    mats += [np.array(view[:], dtype=F32), np.array(proj[:], dtype=F32)]
    for a in mats:
        inv = np.zeros(16, dtype=F32)
        assert L.ro_mat4_inverse(a.ctypes.data_as(C.POINTER(C.c_float)),
                                 inv.ctypes.data_as(C.POINTER(C.c_float))) == 1
        want = glam_scalar_inverse(a)
        assert np.array_equal(inv.view(np.uint32), want.view(np.uint32))


# ---- random helpers, src/main.rs:266-286 -----------------------------------------------------

def test_random_in_unit_sphere_rejects_then_accepts():
    rng = O.scripted_rng([0.99, 0.99, 0.99, 0.5, 0.75, 0.25])
    v = O.lib().ro_random_in_unit_sphere_vec3(C.byref(rng)).tup()
    assert v == (0.0, 0.5, -0.5)
    assert rng.draws == 6


def test_random_in_unit_sphere_rejects_length_exactly_one():
    # xi = 1.0 is outside [0,1) but makes |v|^2 == 1 exactly: `>= 1.0` must reject it
    rng = O.scripted_rng([1.0, 0.5, 0.5, 0.5, 0.5, 0.5])
    v = O.lib().ro_random_in_unit_sphere_vec3(C.byref(rng)).tup()
    assert v == (0.0, 0.0, 0.0) and rng.draws == 6


def test_random_unit_vec3_is_normalised():
    rng = O.scripted_rng([0.75, 0.5, 0.5])
    assert O.lib().ro_random_unit_vec3(C.byref(rng)).tup() == (1.0, 0.0, 0.0)


def test_counter_stream_is_24_bit_uniform():
    L = O.lib()
    rng = O.counter_rng(1, 2, 3)
    xs = np.array([L.ro_random_f32(C.byref(rng)) for _ in range(20000)], dtype=np.float64)
    assert xs.min() >= 0.0 and xs.max() < 1.0
    assert np.all(xs * 2 ** 24 == np.round(xs * 2 ** 24))
    assert abs(xs.mean() - 0.5) < 0.01 and abs(xs.var() - 1 / 12) < 0.005
    # different pixels / samples / seeds give different streams
    firsts = set()
    for key in [(1, 2, 3), (1, 2, 4), (1, 3, 3), (2, 2, 3)]:
        r = O.counter_rng(*key)
        firsts.add(tuple(L.ro_random_f32(C.byref(r)) for _ in range(4)))
    assert len(firsts) == 4


def test_counter_stream_matches_python_restatement():
    def mix(x):
        x ^= x >> 16
        x = (x * 0x7FEB352D) & 0xFFFFFFFF
        x ^= x >> 15
        x = (x * 0x846CA68B) & 0xFFFFFFFF
        x ^= x >> 16
        return x
    seed, pixel, sample = 12345, 777, 9
    a = mix(seed ^ 0xA511E9B3)
    b = mix((seed + 0x632BE5AB) & 0xFFFFFFFF)
    state = mix((mix(a ^ pixel) + sample) & 0xFFFFFFFF)
    # a sample's draws: a linear congruential step from the hashed state, its increment `inc`
    inc = (((state >> 11) | (state << 21)) & 0xFFFFFFFF) ^ 0x632BE5AB | 1
    rng = O.counter_rng(seed, pixel, sample)
    x = state
    for _ in range(16):
        x = (x * 747796405 + inc) & 0xFFFFFFFF
        expect = (x >> 8) / 2.0 ** 24
        assert O.lib().ro_random_f32(C.byref(rng)) == expect
    # scene construction (the stream RT_RNG_SCENE_STREAM): the counter form, one hash per draw
    pixel, sample = 0xFFFFFFF1, 0
    state = mix((mix(a ^ pixel) + sample) & 0xFFFFFFFF)
    inc = mix(mix((b + pixel) & 0xFFFFFFFF) ^ sample) | 1
    rng = O.counter_rng(seed, pixel, sample)
    for _ in range(16):
        state = (state + inc) & 0xFFFFFFFF
        expect = (mix(state) >> 8) / 2.0 ** 24
        assert O.lib().ro_random_f32(C.byref(rng)) == expect


# ---- shoot_ray, src/main.rs:191-263 -----------------------------------------------------------

def shoot(world, origin, direction, depth, rng, recursive=True):
    ctr = O.Counters(0, 0)
    r = O.ray(origin, direction)
    if recursive:
        c = O.lib().ro_shoot_ray(C.byref(world), C.byref(r), depth, C.byref(rng), C.byref(ctr))
    else:
        c = O.lib().ro_shoot_ray_iter(C.byref(world), C.byref(r), depth, C.byref(rng),
                                      O.RO_POW5_LIBM, C.byref(ctr))
    return c.tup(), ctr


@pytest.mark.parametrize("recursive", [True, False])
def test_shoot_ray_depth_zero_is_black_and_traces_nothing(recursive):
    s, m = O.simple_world()
    w = O.make_world(s, m)
    c, ctr = shoot(w, (13, 2, 3), (-13, -2, -3), 0, O.scripted_rng([]), recursive)
    assert c == (0.0, 0.0, 0.0) and ctr.rays == 0


@pytest.mark.parametrize("recursive", [True, False])
def test_shoot_ray_empty_world_is_sky(recursive):
    w = O.make_world(np.zeros(0, dtype=abi.SPHERE_DTYPE), np.zeros(0, dtype=abi.MATERIAL_DTYPE))
    c, ctr = shoot(w, (0, 0, 0), (0, 1, 0), 5, O.scripted_rng([]), recursive)
    assert c == pytest.approx((0.5, 0.7, 1.0), abs=1e-6) and ctr.rays == 1


@pytest.mark.parametrize("recursive", [True, False])
def test_metal_applies_albedo_twice(recursive):
    # mirror (strength 1 -> fuzz 0) facing the ray; reflected ray goes straight up to the sky.
    # final = color * (color * sky): src/main.rs:199,236-237,242
    albedo = (0.7, 0.6, 0.5)
    w = O.make_world(O.spheres_array([(0, -1, 0, 1, 0)]),
                     O.materials_array([(*albedo, 1, 1.0, 0, 0)]))
    rng = O.scripted_rng([0.5, 0.5, 0.5])          # random_in_unit_sphere -> (0,0,0), accepted
    c, ctr = shoot(w, (0, 5, 0), (0, -1, 0), 10, rng, recursive)
    sky = (0.5, 0.7, 1.0)
    assert c == pytest.approx(tuple(a * a * s for a, s in zip(albedo, sky)), rel=1e-6)
    assert ctr.rays == 2 and rng.draws == 3


@pytest.mark.parametrize("recursive", [True, False])
def test_metal_draws_before_testing_and_tests_the_unfuzzed_reflection(recursive):
    # reflected.dot(normal) = -(dir.dot(normal)) and the face normal never points along the
    # ray, so the test `> 0.0` (src/main.rs:235) fails only at exact tangency.  Tangent hit:
    # disc == 0, dir.normal == 0 -> front_face false -> reflected == dir, dot == 0 -> black
    # without recursing, yet the three fuzz draws were already consumed (src/main.rs:229-240).
    w = O.make_world(O.spheres_array([(0, 1, -3, 1, 0)]),
                     O.materials_array([(0.7, 0.6, 0.5, 1, 0.5, 0, 0)]))
    rng = O.scripted_rng([0.5, 0.5, 0.5])
    c, ctr = shoot(w, (0, 0, 0), (0, 0, -1), 10, rng, recursive)
    assert c == (0.0, 0.0, 0.0)
    assert ctr.rays == 1 and rng.draws == 3


@pytest.mark.parametrize("recursive", [True, False])
def test_diffuse_multiplies_albedo_once_and_does_not_normalise(recursive):
    w = O.make_world(O.spheres_array([(0, -1, 0, 1, 0)]),
                     O.materials_array([(0.25, 0.5, 0.75, 0, 0, 0, 0)]))
    rng = O.scripted_rng([0.5, 0.75, 0.5])        # unit vector (0,1,0): dir = n + (0,1,0) = (0,2,0)
    c, ctr = shoot(w, (0, 5, 0), (0, -1, 0), 10, rng, recursive)
    assert c == pytest.approx((0.25 * 0.5, 0.5 * 0.7, 0.75 * 1.0), rel=1e-6)
    assert ctr.rays == 2 and rng.draws == 3


@pytest.mark.parametrize("recursive", [True, False])
@pytest.mark.parametrize("xi,reflects", [(0.01, True), (0.5, False)])
def test_dielectric_chooses_by_reflectance_and_ignores_colour(recursive, xi, reflects):
    # normal incidence on glass: reflectance = 0.04.  xi < 0.04 reflects (back up to the sky,
    # 2 rays); otherwise refracts straight through, leaves through the far side (xi again) and
    # reaches the sky below: colour (1,1,1) whatever material.color says (src/main.rs:224-226).
    w = O.make_world(O.spheres_array([(0, -1, 0, 1, 0)]),
                     O.materials_array([(0.1, 0.1, 0.6, 0, 0, 1, 1.5)]))
    rng = O.scripted_rng([xi, 0.5])
    c, ctr = shoot(w, (0, 5, 0), (0, -1, 0), 10, rng, recursive)
    if reflects:
        assert c == pytest.approx((0.5, 0.7, 1.0), abs=1e-6)
        assert ctr.rays == 2 and rng.draws == 1
    else:
        assert c == pytest.approx((1.0, 1.0, 1.0), abs=1e-6)
        assert ctr.rays == 3 and rng.draws == 2


@pytest.mark.parametrize("recursive", [True, False])
def test_total_internal_reflection_draws_nothing(recursive):
    # Inside glass (ratio = ir = 1.5) hitting the wall at a grazing angle: cannot_refract is
    # true and the `||` short-circuits, so no random number is consumed (src/main.rs:213-215).
    w = O.make_world(O.spheres_array([(0, 0, 0, 1, 0)]),
                     O.materials_array([(1, 1, 1, 0, 0, 1, 1.5)]))
    out = np.zeros(1, dtype=abi.RAY_OUT_DTYPE)
    rng = O.scripted_rng([0.999])
    r = O.ray((0.9, 0, 0), (0, 1, 0))
    O.lib().ro_bounce(C.byref(w), C.byref(r), C.byref(rng), O.RO_POW5_LIBM, abi.ptr(out))
    assert out["hit"][0] == 1 and out["kind"][0] == abi.RT_KIND_DIELECTRIC
    assert out["front_face"][0] == 0
    assert out["draws"][0] == 0 and rng.draws == 0
    # reflected direction = d - 2(d.n)n with n the inward normal at the hit point
    n = np.array([out["nx"][0], out["ny"][0], out["nz"][0]], dtype=np.float64)
    d = np.array([0.0, 1.0, 0.0])
    expect = d - 2 * d.dot(n) * n
    got = np.array([out["sx"][0], out["sy"][0], out["sz"][0]], dtype=np.float64)
    assert np.allclose(got, expect, atol=1e-6)


def test_iterative_and_recursive_estimators_agree():
    L = O.lib()
    s, m = O.random_world(3)
    w = O.make_world(s, m)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    rs = np.random.default_rng(0)
    worst = 0.0
    for i in range(3000):
        u, v = rs.uniform(-1, 1, size=2)
        r = L.ro_camera_get_ray(cam, float(u), float(v))
        a_rng, b_rng = O.counter_rng(9, i, 0), O.counter_rng(9, i, 0)
        ca, cb = O.Counters(0, 0), O.Counters(0, 0)
        a = L.ro_shoot_ray(C.byref(w), C.byref(r), 50, C.byref(a_rng), C.byref(ca)).tup()
        b = L.ro_shoot_ray_iter(C.byref(w), C.byref(r), 50, C.byref(b_rng), O.RO_POW5_LIBM,
                                C.byref(cb)).tup()
        assert ca.rays == cb.rays and a_rng.draws == b_rng.draws
        for x, y in zip(a, b):
            scale = max(abs(x), abs(y), 1e-30)
            worst = max(worst, abs(x - y) / scale)
    assert worst < 1e-5     # only the association of the colour products differs


# ---- resolve, src/cpu/mod.rs:51-83 ; u8 cast, src/main.rs:534-536 ----------------------------

def test_compose_scale_sqrt_clamp_and_nan():
    tiles = np.array([(0, 0, 2, 1)], dtype=abi.TILE_DTYPE)
    spp = 4
    sums = np.array([[spp * 0.25, spp * 1.0, spp * 4.0], [-1.0, 0.0, spp * 0.998001]], dtype=F32)
    out = np.zeros((1, 2, 3), dtype=F32)
    O.lib().ro_compose_final_framebuffer(2, 1, spp, abi.ptr(tiles), 1, abi.ptr(sums), abi.ptr(out))
    assert out[0, 0].tolist() == [0.5, F32(0.999), F32(0.999)]
    assert math.isnan(out[0, 1, 0])          # sqrt of a negative sum; clamp keeps NaN
    assert out[0, 1, 1] == 0.0
    assert out[0, 1, 2] == pytest.approx(0.999, abs=1e-6)


def test_f32_as_u8_saturates_and_maps_nan_to_zero():
    L = O.lib()
    assert L.ro_f32_as_u8(float("nan")) == 0
    assert L.ro_f32_as_u8(-3.0) == 0
    assert L.ro_f32_as_u8(254.99) == 254
    assert L.ro_f32_as_u8(255.0) == 255
    assert L.ro_f32_as_u8(1e9) == 255
    assert L.ro_f32_as_u8(0.999 * 255.0) == 254


# ---- tiler, src/cpu/job.rs:76-152 -------------------------------------------------------------

@pytest.mark.parametrize("w,h,n", [(400, 225, 28), (1920, 1080, 510), (3840, 2160, 2040)])
def test_job_queue_counts_and_exact_cover(w, h, n):
    tiles = O.job_queue(w, h, 64, 64)
    assert len(tiles) == n
    cover = np.zeros((h, w), dtype=np.int32)
    for t in tiles:
        cover[t["y"]:t["y"] + t["height"], t["x"]:t["x"] + t["width"]] += 1
    assert cover.min() == 1 and cover.max() == 1


def test_job_queue_order_and_corner():
    tiles = O.job_queue(400, 225, 64, 64)
    # 6x3 full tiles row-major, then the bottom strip (6), the right strip (3), the corner
    assert tuple(tiles[0]) == (0, 0, 64, 64)
    assert tuple(tiles[1]) == (64, 0, 64, 64)
    assert tuple(tiles[6]) == (0, 64, 64, 64)
    assert tuple(tiles[18]) == (0, 192, 64, 33)
    assert tuple(tiles[24]) == (384, 0, 16, 64)
    assert tuple(tiles[27]) == (384, 192, 16, 33)


def test_job_queue_corner_uses_tile_width_for_y():
    # src/cpu/job.rs:140: start_y = num_tiles_y * tile_width (sic)
    tiles = O.job_queue(100, 100, 32, 48)
    assert tuple(tiles[-1]) == (96, 2 * 32, 4, 4)


def test_debug_tile_framebuffer():
    tiles = O.job_queue(400, 225, 64, 64)
    out = np.zeros((225, 400, 3), dtype=F32)
    O.lib().ro_debug_write_job_queue_to_framebuffer(400, 225, abi.ptr(tiles), len(tiles), abi.ptr(out))
    assert out[0, 0].tolist() == [0.0, 0.0, 0.0]
    assert out[10, 70].tolist() == [F32(6) / F32(64), F32(10) / F32(64), 0.0]
    assert out[200, 390].tolist() == [F32(6) / F32(16), F32(8) / F32(33), 0.0]


# ---- scenes, src/main.rs:326-519 ----------------------------------------------------------------

def test_random_world_layout():
    s, m = O.random_world(1)
    assert len(s) == 68 and len(m) == 68
    small = s[:64]
    assert np.all(small["radius"] == F32(0.2)) and np.all(small["y"] == F32(0.2))
    # cell (a,b): x in [a, a+0.9), z in [b, b+0.9), a outer loop, b inner loop
    a = np.repeat(np.arange(-4, 4), 8)
    b = np.tile(np.arange(-4, 4), 8)
    assert np.all((small["x"] >= a) & (small["x"] < a + 0.9 + 1e-6))
    assert np.all((small["z"] >= b) & (small["z"] < b + 0.9 + 1e-6))
    assert np.all(s["material_id"] == np.arange(68))
    assert tuple(s[64]) == (0.0, -1000.0, 0.0, 1000.0, 64)
    assert tuple(s[65]) == (0.0, 1.0, 0.0, 1.0, 65)
    assert tuple(s[66]) == (-4.0, 1.0, 0.0, 1.0, 66)
    assert tuple(s[67]) == (4.0, 1.0, 0.0, 1.0, 67)
    assert tuple(m[64]) == (0.5, 0.5, 0.5, 0, 0.0, 0.0)
    assert m[65]["flags"] == abi.RT_MAT_DIELECTRIC and m[65]["ir"] == F32(1.4)
    assert m[66]["flags"] == 0 and tuple(m[66])[:3] == (F32(0.1), F32(0.1), F32(0.6))
    assert m[67]["flags"] == abi.RT_MAT_METALLIC and m[67]["metallic_strength"] == 1.0
    # diffuse albedo = random_vec3()*0.9 lies in [-0.9, 0.9): negative channels do occur
    diffuse = m[:64][m[:64]["flags"] == 0]
    rgb = np.stack([diffuse["r"], diffuse["g"], diffuse["b"]])
    assert rgb.min() < 0 and rgb.min() >= -0.9 and rgb.max() < 0.9
    glass = m[:64][m[:64]["flags"] == abi.RT_MAT_DIELECTRIC]
    assert np.all(glass["ir"] == F32(1.5)) and np.all(glass["r"] == 1.0)


def test_random_world_scales_to_the_benchmark_sizes():
    s, _ = O.random_world(1, -11, 11)
    assert len(s) == 488
    s, m = O.random_world(1, -11, 11, 0.1, 0.2, 1)
    frac = np.mean(m[:484]["flags"] == abi.RT_MAT_DIELECTRIC)
    assert frac >= 0.75 and np.all(m[485:]["flags"] == abi.RT_MAT_DIELECTRIC)


def test_simple_world():
    s, m = O.simple_world()
    assert tuple(s[0]) == (0.0, -1000.0, 0.0, 1000.0, 0)
    assert tuple(s[1]) == (4.0, 1.0, 0.0, 1.0, 1)
    assert m[1]["flags"] == abi.RT_MAT_METALLIC and m[1]["metallic_strength"] == F32(0.1)


# ---- schedulers, src/cpu/single.rs, src/cpu/multi.rs ------------------------------------------

def test_single_and_multi_threaded_renders_are_identical():
    s, m = O.random_world(1)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    a, _, ca = O.render(s, m, cam, 96, 54, 4, 50, 7, threads=1)
    b, _, cb = O.render(s, m, cam, 96, 54, 4, 50, 7, threads=4)
    assert np.array_equal(a, b, equal_nan=True)
    assert ca.rays == cb.rays and ca.samples == cb.samples == 96 * 54 * 4


def test_empty_world_image_is_the_sky_gradient():
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    s = np.zeros(0, dtype=abi.SPHERE_DTYPE)
    m = np.zeros(0, dtype=abi.MATERIAL_DTYPE)
    fb, _, ctr, sums = O.render(s, m, cam, 64, 36, 8, 50, 1, threads=2, want_sums=True)
    assert ctr.rays == ctr.samples == 64 * 36 * 8
    assert not np.isnan(fb).any()
    # blue channel of the sky is exactly 1 -> resolve gives 0.999 everywhere
    assert np.all(fb[..., 2] == F32(0.999))
    # red = 1 - t/2 with t = (dir.y+1)/2 rises from the top row to the bottom row (y=0 is bottom)
    red = fb[..., 0].mean(axis=1)
    assert np.all(np.diff(red) < 0)


def white_furnace_world():
    """Ground plus three spheres, every one diffuse with albedo exactly 1: a path's throughput
    stays 1, so whatever it returns is a sky colour - blue exactly 1, red and green in [0.5, 1]
    - or black if it runs out of depth.  Energy is neither lost nor gained in the diffuse branch."""
    s = np.zeros(4, dtype=abi.SPHERE_DTYPE)
    m = np.zeros(1, dtype=abi.MATERIAL_DTYPE)
    m[0] = (1.0, 1.0, 1.0, 0, 0.0, 0.0)
    s[0] = (0.0, -1000.0, 0.0, 1000.0, 0)
    s[1] = (0.0, 1.0, 0.0, 1.0, 0)
    s[2] = (-4.0, 1.0, 0.0, 1.0, 0)
    s[3] = (4.0, 1.0, 0.0, 1.0, 0)
    return s, m


def check_white_furnace(sums, spp, rays, samples):
    per_sample = sums / F32(spp)
    assert not np.isnan(sums).any()
    # blue: every finished path adds s*1 + t*1 with s = 1 - t, i.e. 1 to within an ulp
    assert np.all(per_sample[..., 2] <= 1.0 + 1e-6)
    assert np.all(per_sample[..., 0] <= 1.0 + 1e-6) and np.all(per_sample[..., 1] <= 1.0 + 1e-6)
    # at most a few paths in a thousand die at max_depth 50 between the spheres and the ground
    assert per_sample[..., 2].mean() > 0.995
    # red and green between the horizon's white and the zenith's (0.5, 0.7)
    assert np.all(per_sample[..., 0] >= 0.5 * per_sample[..., 2] - 1e-6)
    assert np.all(per_sample[..., 1] >= 0.7 * per_sample[..., 2] - 1e-6)
    assert rays > 2 * samples        # the furnace does bounce


def test_white_furnace_conserves_energy():
    s, m = white_furnace_world()
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    w, h, spp = 64, 36, 32
    tiles = O.job_queue(w, h, 64, 64)
    _, _, ctr, sums = O.render(s, m, cam, w, h, spp, 50, 4, mode=O.RO_MODE_RECURSIVE, threads=8,
                               tiles=tiles, want_sums=True)
    check_white_furnace(sums, spp, ctr.rays, ctr.samples)


def test_bmp_writer(tmp_path):
    fb = np.zeros((2, 3, 3), dtype=F32)
    fb[0, 0] = (1.0, 0.5, 0.0)       # bottom-left pixel of the picture
    fb[1, 2] = (0.2, float("nan"), 2.0)
    path = tmp_path / "t.bmp"
    assert O.lib().ro_write_framebuffer_to_image(str(path).encode(), 3, 2, abi.ptr(fb)) == 0
    raw = path.read_bytes()
    assert raw[:2] == b"BM" and len(raw) == 54 + 2 * 12
    assert int.from_bytes(raw[18:22], "little") == 3 and int.from_bytes(raw[22:26], "little") == 2
    assert int.from_bytes(raw[28:30], "little") == 24
    # first stored row = bottom row = framebuffer row 0; pixel bytes are B,G,R
    assert raw[54:57] == bytes([0, 127, 255])
    second = raw[54 + 12:]
    assert second[6:9] == bytes([255, 0, 51])


def test_sample_streams_are_uniform_in_one_and_three_dimensions():
    """The draws inside a sample come from a linear congruential step (include/rt_rng.h), chosen
    for its cost; what the estimator needs from it is what rand's generator gives the reference
    (src/main.rs:266-282): draws uniform on [0, 1), consecutive draws uncorrelated, consecutive
    triples uniform in the cube (random_in_unit_sphere_vec3 rejects on them).  Checked here over
    the streams of 40 000 (pixel, sample) pairs, 48 draws each, restated in numpy."""
    def mix(x):
        x = x ^ (x >> np.uint64(16))
        x = (x * np.uint64(0x7FEB352D)) & M
        x = x ^ (x >> np.uint64(15))
        x = (x * np.uint64(0x846CA68B)) & M
        return x ^ (x >> np.uint64(16))
    M = np.uint64(0xFFFFFFFF)
    seed = 1
    a = mix(np.array([seed ^ 0xA511E9B3], dtype=np.uint64))[0]
    pixel = np.repeat(np.arange(400, dtype=np.uint64) * np.uint64(5197) + np.uint64(11), 100)
    sample = np.tile(np.arange(100, dtype=np.uint64), 400)
    state = mix((mix(a ^ pixel) + sample) & M)
    inc = ((((state >> np.uint64(11)) | (state << np.uint64(21))) & M) ^ np.uint64(0x632BE5AB)) | np.uint64(1)
    # the first stream against the oracle's own generator
    rng = O.counter_rng(seed, int(pixel[0]), int(sample[0]))
    x = state.copy()
    draws = np.empty((48, len(x)))
    for k in range(48):
        x = (x * np.uint64(747796405) + inc) & M
        draws[k] = (x >> np.uint64(8)).astype(np.float64) / 2.0 ** 24
        assert O.lib().ro_random_f32(C.byref(rng)) == draws[k, 0]
    n = draws.shape[1]
    # every draw position: mean 1/2, variance 1/12 (standard errors 0.0014 and 0.0004)
    assert np.abs(draws.mean(axis=1) - 0.5).max() < 0.007
    assert np.abs(draws.var(axis=1) - 1 / 12).max() < 0.002
    # consecutive draws, and draws three apart (the same coordinate of consecutive triples)
    for lag in (1, 2, 3):
        c = [np.corrcoef(draws[k], draws[k + lag])[0, 1] for k in range(48 - lag)]
        assert np.abs(c).max() < 5.0 / np.sqrt(n)
    # triples (draws 3t, 3t+1, 3t+2) in an 8 x 8 x 8 grid of cells: chi-square with 511 degrees
    # of freedom (mean 511, standard deviation 32) for every t and for all triples together
    cells = (np.floor(draws * 8).astype(np.int64)).reshape(16, 3, n)
    idx = cells[:, 0] * 64 + cells[:, 1] * 8 + cells[:, 2]
    for t in range(16):
        counts = np.bincount(idx[t], minlength=512)
        assert ((counts - n / 512) ** 2 / (n / 512)).sum() < 511 + 5 * 32
    counts = np.bincount(idx.ravel(), minlength=512)
    assert ((counts - 16 * n / 512) ** 2 / (16 * n / 512)).sum() < 511 + 5 * 32
    # the rejection test accepts pi / 6 of the triples
    v = draws.reshape(16, 3, n) * 2 - 1
    accept = ((v ** 2).sum(axis=1) < 1).mean()
    assert abs(accept - np.pi / 6) < 4 * np.sqrt(0.25 / (16 * n))
