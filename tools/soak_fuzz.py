"""Developer soak run of tests/test_gpu_fuzz.py's comparison over many random scenes: the literal
arithmetic (STRICT_FP), the conservative scan and the hierarchy must give the same per-ray bits.
Usage: python tools/soak_fuzz.py [n_scenes] [rays_per_kind] [first_seed]
       python tools/soak_fuzz.py images [n_scenes] [first_seed]   whole paths: per-pixel sums of
           small frames (reference sample order, max_depth 50) in the three modes
       python tools/soak_fuzz.py grazing [n_scenes] [rays_per_scene] [first_seed]   rays within a
           few hundred roundings of a silhouette, three filters"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_fuzz as F  # noqa: E402

from ray_tracing_amd import abi, hip  # noqa: E402


def images():
    import oracle_lib as O
    n_scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    hip.init(0)
    seq = abi.RT_FLAG_SEQUENTIAL_SUM
    t0 = time.time()
    rays = 0
    for k in range(n_scenes):
        rs = np.random.default_rng(seed0 + k)
        n = int(rs.choice([3, 40, 150, 600, 2300]))
        r_hi = float(10.0 ** rs.uniform(-0.7, 0.6))
        s, m = F.random_scene(rs, n, (0, 0.5, 0), float(rs.uniform(1.5, 8.0)), r_hi * 0.02, r_hi)
        w, h = int(rs.integers(17, 90)), int(rs.integers(9, 50))
        spp = int(rs.integers(1, 9))
        cam = O.camera_new((13, 2, 3), (0, 0, 0), float(rs.uniform(15, 60)), w / h)
        tiles = O.job_queue(w, h, 64, 64)
        ref = hip.render_tiles(s, m, cam, w, h, spp, 50, k, tiles, flags=seq | abi.RT_FLAG_STRICT_FP)
        ref_rays = hip.get_stats().rays
        if k % 4 == 0:   # and the literal arithmetic against the CPU oracle
            _, _, ctr, want = O.render(s, m, cam, w, h, spp, 50, k, mode=O.RO_MODE_ITER_MUL,
                                       threads=16, tiles=tiles, want_sums=True)
            same = (want.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(want) & np.isnan(ref))
            if not same.all() or ctr.rays != ref_rays:
                print("ORACLE MISMATCH scene seed %d n %d %dx%d spp %d: %d values differ"
                      % (seed0 + k, n, w, h, spp, int((~same).sum())), flush=True)
                sys.exit(1)
        for flags in (0, abi.RT_FLAG_ACCEL):
            got = hip.render_tiles(s, m, cam, w, h, spp, 50, k, tiles, flags=seq | flags)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(got) & np.isnan(ref))
            if not same.all() or hip.get_stats().rays != ref_rays:
                print("MISMATCH scene seed %d flags %d n %d %dx%d spp %d: %d values differ"
                      % (seed0 + k, flags, n, w, h, spp, int((~same).sum())), flush=True)
                sys.exit(1)
        rays += ref_rays
        if k % 20 == 19:
            print("%d scenes, %d rays, %.0f s" % (k + 1, rays, time.time() - t0), flush=True)
    print("all equal: %d scenes, %d rays" % (n_scenes, rays))


def grazing():
    """Rays within a few hundred roundings of a silhouette (tests/test_gpu_silhouette.py's
    generator) over random scenes at random scales: matrix filter, vector filter and literal
    arithmetic must give the same per-ray bits."""
    import test_gpu_silhouette as S
    n_scenes = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    n_rays = int(sys.argv[3]) if len(sys.argv) > 3 else 500000
    seed0 = int(sys.argv[4]) if len(sys.argv) > 4 else 5000
    hip.init(0)
    t0 = time.time()
    rays_total = on_target = 0
    for k in range(n_scenes):
        rs = np.random.default_rng(seed0 + k)
        n = int(rs.choice([7, 60, 300, 488, 1000, 2500]))
        factor = float(2.0 ** rs.uniform(-12, 14))
        s, m = F.random_scene(rs, n, (0.5, -1.0, 2.0), 10.0, 0.01, 2.0)
        s = s.copy()
        for f in ("x", "y", "z", "radius"):
            s[f] = (s[f].astype(np.float64) * factor).astype(np.float32)
        scene = hip.Scene(s, m)
        info = scene.scan_info()
        rays, aimed = S.grazing_rays(rs, s, n_rays, info.origin_limit if info.matrix_filter else 1e4 * factor)
        strict = hip.trace_rays(scene, rays, seed=k, flags=abi.RT_FLAG_STRICT_FP)
        for flags in (0, abi.RT_FLAG_SCAN_VALU):
            got = hip.trace_rays(scene, rays, seed=k, flags=flags)
            try:
                F.assert_same(got, strict)
            except AssertionError as e:
                print("MISMATCH scene seed %d flags %d n %d factor %g: %s" % (seed0 + k, flags, n, factor, e),
                      flush=True)
                raise
        scene.close()
        rays_total += len(rays)
        on_target += int(((strict["hit"] == 1) & (strict["sphere"] == aimed)).sum())
        if k % 5 == 4:
            print("%d scenes, %d grazing rays, %d of them hit the grazed sphere, %.0f s"
                  % (k + 1, rays_total, on_target, time.time() - t0), flush=True)
    print("all equal: %d scenes, %d grazing rays, %d hit the grazed sphere" % (n_scenes, rays_total, on_target))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "images":
        return images()
    if len(sys.argv) > 1 and sys.argv[1] == "grazing":
        return grazing()
    n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    n_each = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
    seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
    hip.init(0)
    t0 = time.time()
    rays_total = hits_total = 0
    for k in range(n_scenes):
        rs = np.random.default_rng(seed0 + k)
        n = int(rs.choice([1, 2, 7, 60, 300, 1000, 2500]))
        scale = float(10.0 ** rs.uniform(-1, 3.5))
        centre = rs.normal(size=3) * scale * float(rs.choice([0, 1, 30]))
        r_lo = scale * 10.0 ** rs.uniform(-4, -1)
        r_hi = max(r_lo * 1.001, scale * 10.0 ** rs.uniform(-1.5, 0.3))
        s, m = F.random_scene(rs, n, centre, scale, r_lo, r_hi)
        rays = F.adversarial_rays(rs, s, n_each)
        scene = hip.Scene(s, m)
        strict = hip.trace_rays(scene, rays, seed=k, flags=abi.RT_FLAG_STRICT_FP)
        for flags in (0, abi.RT_FLAG_ACCEL):
            got = hip.trace_rays(scene, rays, seed=k, flags=flags)
            try:
                F.assert_same(got, strict)
            except AssertionError as e:
                print("MISMATCH scene seed %d flags %d n %d scale %g centre %s r [%g, %g]: %s"
                      % (seed0 + k, flags, n, scale, centre, r_lo, r_hi, e), flush=True)
                raise
        scene.close()
        rays_total += len(rays)
        hits_total += int(strict["hit"].sum())
        if k % 10 == 9:
            print("%d scenes, %d rays, %d hits, %.0f s" % (k + 1, rays_total, hits_total,
                                                           time.time() - t0), flush=True)
    print("all equal: %d scenes, %d rays, %d hits" % (n_scenes, rays_total, hits_total))


if __name__ == "__main__":
    main()
