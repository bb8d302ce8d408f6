"""Developer soak run of tests/test_gpu_fuzz.py's comparison over many random scenes: the literal
arithmetic (STRICT_FP), the conservative scan and the hierarchy must give the same per-ray bits.
Usage: python tools/soak_fuzz.py [n_scenes] [rays_per_kind] [first_seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_fuzz as F  # noqa: E402

from ray_tracing_amd import abi, hip  # noqa: E402


def main():
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
