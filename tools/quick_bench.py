"""Developer benchmark: time the path-tracing kernel on one configuration and print the
roofline fraction.
Usage: python tools/quick_bench.py [grid_hi] [width] [height] [spp] [flags] [reps] [glass]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ray_tracing_amd import hip, host  # noqa: E402

PEAK_TFLOPS = 157.3


def main():
    hi = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
    glass = len(sys.argv) > 7 and sys.argv[7] == "glass"
    hip.init(0)
    world = (host.create_random_world(1, (-hi, hi), 0.1, 0.2, True) if glass
             else host.create_random_world(1, (-hi, hi)))
    s = world.spheres
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(cam.m, w, h, spp, 50, 1, flags)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for rep in range(reps):
        hip.render_rows_device_async(scene, frame, 0, h, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        ms = st.kernel_ms
        tflops = 17.0 * st.sphere_tests / (ms * 1e-3) * 1e-12
        print("N=%d %dx%d spp=%d flags=%d: kernel %.2f ms  %.1f Msamples/s  rays/sample %.3f  "
              "%.2f Gtests/s  %.2f TFLOP/s = %.1f%% of %.1f  (wg %d x %d thr, slices %d)"
              % (len(s), w, h, spp, flags, ms, st.samples / ms * 1e-3, st.rays / max(st.samples, 1),
                 st.sphere_tests / ms * 1e-6, tflops, 100 * tflops / PEAK_TFLOPS, PEAK_TFLOPS,
                 st.workgroups, st.threads_per_workgroup, st.slices), flush=True)
    ph = hip.get_phase_cycles().astype(np.float64)
    if ph[:4].sum() > 0:
        tot = ph[:4].sum()
        print("phase shares: refill %.1f%%  scan %.1f%%  replay %.1f%%  shade %.1f%%   "
              "cycles/step/wave %.0f (steps %d)" % (100 * ph[0] / tot, 100 * ph[1] / tot,
                                                   100 * ph[2] / tot, 100 * ph[3] / tot,
                                                   tot / max(ph[4], 1), int(ph[4])))
        waves = st.workgroups * st.threads_per_workgroup / 64
        print("drain after the work queue emptied: mean %.0f kcycles per wave, longest %.0f kcycles; "
              "longest-lived wave %.0f kcycles" % (ph[5] / waves / 1e3, ph[6] / 1e3, ph[7] / 1e3))
    if ph[14] + ph[15] > 0 and flags & 4:
        print("hierarchy: %.1f interior nodes opened and %.1f leaf spheres tested per ray"
              % (ph[14] / max(st.rays, 1), ph[15] / max(st.rays, 1)))
    elif ph[14] + ph[15] > 0:
        print("replay: %.2f candidates per ray; the wave's replay loop makes %.2f trips per ray "
              "(%.0f %% of its lanes busy)" % (ph[14] / max(st.rays, 1), ph[15] / max(st.rays, 1),
                                               100.0 * ph[14] / max(ph[15], 1)))
    img = out.cpu().numpy()
    print("mean rgb", np.nanmean(img, axis=(0, 1)), "nan px", int(np.isnan(img).any(axis=2).sum()))


if __name__ == "__main__":
    main()
