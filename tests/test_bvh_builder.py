"""The host-side hierarchy builder (ray_tracing_amd/csrc/rt_bvh.cpp) checked on the CPU: a small
native program links the builder, builds a tree for spheres given on stdin and verifies the
invariants the GPU traversal relies on (every sphere in exactly one leaf or in the always-tested
list, boxes nested and holding their spheres, reported depth, the ball and radius range behind
the per-ray margin)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("bvh") / "bvh_check")
    cmd = [HIPCC, "-O1", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
           os.path.join(ROOT, "tests", "native", "bvh_check.cpp"),
           os.path.join(ROOT, "ray_tracing_amd", "csrc", "rt_bvh.cpp"), "-o", out]
    subprocess.run(cmd, check=True, capture_output=True, timeout=300)
    return out


def run(checker, spheres):
    text = "".join("%r %r %r %r\n" % (float(s[0]), float(s[1]), float(s[2]), float(s[3]))
                   for s in spheres)
    p = subprocess.run([checker], input=text, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stdout + p.stderr
    tag, nodes, depth, n_big = p.stdout.split()
    assert tag == "ok"
    return int(nodes), int(depth), int(n_big)


def rows(s):
    return np.stack([s["x"], s["y"], s["z"], s["radius"]], axis=1)


@pytest.mark.parametrize("grid,expect_big", [((-4, 4), 4), ((-11, 11), 4), ((-50, 50), 4)])
def test_reference_scenes(checker, grid, expect_big):
    s, _ = O.random_world(1, *grid)
    nodes, depth, n_big = run(checker, rows(s))
    assert n_big == expect_big                  # the ground and the three unit spheres
    n_small = len(s) - n_big
    assert depth <= int(np.ceil(np.log2(max(n_small, 2)))) + 1 <= 24
    assert nodes <= 2 * n_small


def test_degenerate_inputs(checker):
    assert run(checker, np.zeros((0, 4))) == (0, 0, 0)
    assert run(checker, [[1, 2, 3, 0.5]]) == (1, 1, 0)
    # forty coincident spheres still split into leaves of at most eight
    nodes, depth, n_big = run(checker, [[0, 0.2, 0, 0.2]] * 40)
    assert n_big == 0 and depth <= 6
    # zero and negative radii, huge coordinates
    rs = np.random.default_rng(3)
    pts = rs.uniform(-1e6, 1e6, size=(200, 3))
    rad = rs.choice([0.0, -0.3, 0.2, 5.0], size=(200, 1))
    run(checker, np.hstack([pts, rad]))


def test_host_prepared_dividers_match_integer_division(tmp_path):
    """The kernel divides work-item numbers by the slice count and the tile width with
    multipliers prepared on the host (rt::make_divider / rt::fast_div); the same inline code,
    compiled for the host, against `/` over edge cases and a sweep."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path / "divider_check")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-x", "hip", "--offload-arch=gfx950",
                    os.path.join(ROOT, "tests", "native", "divider_check.cpp"), "-o", out],
                   check=True, capture_output=True, timeout=300)
    p = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stdout + p.stderr


def test_division_by_a_prepared_reciprocal_is_the_division(tmp_path):
    """The sample loop's `(x + jitter) / image_width as f32` (src/cpu/job.rs:53-54) is computed as
    rt::div_by from the reciprocal prepared on the host; the same inline code, compiled for the
    host, must give the bits of `/`: every float below the extent for two image extents, sweeps
    for the others."""
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path / "div_by_check")
    subprocess.run([HIPCC, "-O2", "-std=c++17", "-ffp-contract=off", "-x", "hip", "--offload-arch=gfx950",
                    os.path.join(ROOT, "tests", "native", "div_by_check.cpp"), "-o", out],
                   check=True, capture_output=True, timeout=300)
    p = subprocess.run([out], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and p.stdout.startswith("ok "), p.stdout + p.stderr
