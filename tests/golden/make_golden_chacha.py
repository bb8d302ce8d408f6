"""Generates tests/golden/chacha_{rtiow488,glass488}_80x45_4096spp.npz and
chacha_glass488_400x225_4096spp.npz.

Each file holds the per-pixel first and second moments (sum and sum of squares of the samples'
radiance, float64 / float32) of the reference's recursive estimator (ro_shoot_ray, powf) over
4096 samples per pixel of an 80x45 frame, with every random number drawn from ChaCha12
generators (oracle/ro_chacha.c: the generator and the u32 -> f32 mapping of rand 0.8.5's
thread_rng, which is what the reference draws from).  Nothing in it comes from include/rt_rng.h,
the counter stream the GPU kernel and the oracle's other entry points use, so a comparison with
these files can see a bias that stream would introduce.

No render of the Rust program exists (no Rust toolchain here), so this is the closest stand-in
for "the reference render": same generator family, same estimator text, restated in C.

    python tests/golden/make_golden_chacha.py        # about a minute on 8 cores
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_lib as O  # noqa: E402

SPP, DEPTH, STREAM_SEED = 4096, 50, 2024
SCENES = {
    # BASELINE config 2's scene, and the dielectric-heavy one of config 5
    "rtiow488": dict(lo=-11, hi=11, diffuse_below=0.8, metal_below=0.95, big_all_dielectric=0),
    "glass488": dict(lo=-11, hi=11, diffuse_below=0.1, metal_below=0.2, big_all_dielectric=1),
}
# (scene, width, height): both scenes small, and the dielectric-heavy one at the 400x225 of the
# reference's own plumbing configuration, the size its fixed-seed accuracy image is checked at
# (about eight minutes on 8 cores; the sums are kept as float32 there to keep the file small)
FILES = (("rtiow488", 80, 45), ("glass488", 80, 45), ("glass488", 400, 225))


def main():
    only = sys.argv[1:]          # e.g. `glass488_400x225` to make just that file
    for name, W, H in FILES:
        if only and "%s_%dx%d" % (name, W, H) not in only:
            continue
        p = SCENES[name]
        s, m = O.random_world(1, p["lo"], p["hi"], p["diffuse_below"], p["metal_below"],
                              p["big_all_dielectric"])
        cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, W / H)
        s1, s2, ctr = O.render_moments(s, m, cam, W, H, SPP, DEPTH, STREAM_SEED,
                                       O.RO_RNG_CHACHA12, threads=os.cpu_count() or 1)
        path = os.path.join(HERE, "chacha_%s_%dx%d_%dspp.npz" % (name, W, H, SPP))
        np.savez_compressed(path, sum=s1 if W * H < 10000 else s1.astype(np.float32),
                            sumsq=s2.astype(np.float32), spp=SPP, depth=DEPTH,
                            width=W, height=H, stream_seed=STREAM_SEED, scene_seed=1,
                            grid=(p["lo"], p["hi"]),
                            thresholds=(p["diffuse_below"], p["metal_below"]),
                            big_all_dielectric=p["big_all_dielectric"],
                            rays=ctr.rays, samples=ctr.samples)
        print(path, "rays/sample %.4f" % (ctr.rays / ctr.samples))


if __name__ == "__main__":
    main()
