"""Regenerate the golden fixtures of this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference ships no golden vectors and cannot be run here (no Rust toolchain), so these are
vectors of the CPU oracle, generated only after its hand-derived known-answer tests pass
(tests/test_oracle_known_answers.py).  They freeze the definitions that everything else shares
(the counter RNG, the scene recipe, the camera, the estimator): if any of them drifts, the
oracle and the HIP path would still agree with each other but no longer with these files.

Files (all little-endian numpy .npz, no pickles):
  scene_488.npz     create_random_world(seed 1, grid -11..11): spheres, materials, camera
  rays_488.npz      about 550 rays (250 camera rays and two generations of scattered rays) against that
                    scene and their per-ray records (hit, t, normal, scatter dir, draws) for
                    seed 77, reflectance pow5 by multiplication
  image_48x27.npz   per-pixel un-normalised sums of a 48x27, 4 spp, max_depth 50 render
                    (seed 5) of the reference's own 68-sphere scene, for the recursive/powf
                    estimator and for the loop/pow5-by-multiplication estimator
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

import oracle_lib as O  # noqa: E402
from ray_tracing_amd import abi  # noqa: E402


def make_rays(s, m, cam, n_primary, seed):
    w = O.make_world(s, m)
    rs = np.random.default_rng(seed)
    rays = np.zeros(n_primary, dtype=abi.RAY_IN_DTYPE)
    for i in range(n_primary):
        u, v = rs.uniform(-1, 1, size=2)
        r = O.lib().ro_camera_get_ray(cam, float(u), float(v))
        rays[i] = (*r.origin.tup(), *r.dir.tup())
    gens = [rays]
    for g in range(2):
        out = O.bounce(w, gens[-1], seed + 100 + g)
        sel = out[out["scattered"] == 1]
        nxt = np.zeros(len(sel), dtype=abi.RAY_IN_DTYPE)
        nxt["ox"], nxt["oy"], nxt["oz"] = sel["px"], sel["py"], sel["pz"]
        nxt["dx"], nxt["dy"], nxt["dz"] = sel["sx"], sel["sy"], sel["sz"]
        gens.append(nxt)
    return np.concatenate(gens)


def main():
    s, m = O.random_world(1, -11, 11)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, 16 / 9)
    np.savez(os.path.join(HERE, "scene_488.npz"), spheres=s, materials=m,
             view_inv=np.array(cam.view_inv[:], dtype=np.float32),
             proj_inv=np.array(cam.proj_inv[:], dtype=np.float32))

    rays = make_rays(s, m, cam, 250, seed=3)[:600]
    rec = O.bounce(O.make_world(s, m), rays, seed=77, pow5_mode=O.RO_POW5_MUL)
    np.savez(os.path.join(HERE, "rays_488.npz"), rays=rays, records=rec, seed=np.uint32(77))

    s68, m68 = O.random_world(1)
    w, h, spp, depth, seed = 48, 27, 4, 50, 5
    cam2 = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    tiles = O.job_queue(w, h, 64, 64)
    out = {}
    for name, mode in (("recursive_powf", O.RO_MODE_RECURSIVE), ("loop_mul", O.RO_MODE_ITER_MUL)):
        _, _, ctr, sums = O.render(s68, m68, cam2, w, h, spp, depth, seed, mode=mode, threads=4,
                                   tiles=tiles, want_sums=True)
        out["sums_" + name] = sums
        out["rays_" + name] = np.uint64(ctr.rays)
    np.savez(os.path.join(HERE, "image_48x27.npz"), width=w, height=h, spp=spp, max_depth=depth,
             seed=seed, **out)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
