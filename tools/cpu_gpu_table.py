"""Every BASELINE configuration on this box: the CPU port of the reference's multithreaded tile
renderer (oracle/, timed spawn to join like src/cpu/multi.rs:146-165, at a reduced sample count
where the full one would take minutes; cost is linear in spp) next to one MI355X (kernel time of
the full configuration, HIP events).  Prints one markdown table row per configuration.
Usage: python tools/cpu_gpu_table.py [seconds of CPU work per configuration, default 20]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import bench  # noqa: E402
import oracle_lib as O  # noqa: E402
from ray_tracing_amd import hip, host  # noqa: E402

CONFIGS = [
    ("1. reference's own scene (68 spheres) 400x225 @ 16 spp", dict(grid=(-4, 4), width=400, height=225, spp=16, depth=50)),
    ("2. 488 spheres 1920x1080 @ 100 spp", bench.CONFIGS["rtiow488_1080p_100spp_d50"]),
    ("3. 488 spheres 3840x2160 @ 500 spp", bench.CONFIGS["rtiow488_2160p_500spp_d50"]),
    ("4. 10 004 spheres 1920x1080 @ 64 spp", bench.CONFIGS["rtiow10k_1080p_64spp_d50"]),
    ("5. dielectric-heavy 488 spheres 1920x1080 @ 1024 spp", bench.CONFIGS["glass488_1080p_1024spp_d50"]),
]


def cpu_leg(cfg, budget_s):
    cores = bench.logical_cores()
    w, h = cfg["width"], cfg["height"]
    lo, hi = cfg["grid"]
    n_spheres = (hi - lo) ** 2 + 4
    per_core_rate = 0.3e6 * 488.0 / n_spheres
    spp = int(max(1, min(cfg["spp"], round(budget_s * per_core_rate * min(cores, 16) / (w * h)))))
    s, m = O.random_world(1, lo, hi, cfg.get("diffuse_below", 0.8), cfg.get("metal_below", 0.95),
                          1 if cfg.get("glass") else 0)
    cam = O.camera_new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    _, secs, ctr = O.render(s, m, cam, w, h, spp, cfg["depth"], 1, mode=O.RO_MODE_RECURSIVE, threads=cores)
    return ctr.samples / secs * 1e-6, cores, spp, secs, ctr.rays / ctr.samples


def gpu_leg(cfg, reps=3):
    w, h, spp = cfg["width"], cfg["height"], cfg["spp"]
    world = host.create_random_world(1, cfg["grid"], cfg.get("diffuse_below", 0.8),
                                     cfg.get("metal_below", 0.95), bool(cfg.get("glass")))
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(cam.m, w, h, spp, cfg["depth"], 1, 0)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    best = None
    for _ in range(reps):
        hip.render_rows_device_async(scene, frame, 0, h, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        best = st.kernel_ms if best is None else min(best, st.kernel_ms)
    scene.close()
    tests = st.rays * len(world.spheres)
    return best, st.samples / best * 1e-3, st.rays / st.samples, 17.0 * tests / (best * 1e-3) * 1e-12


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    hip.init(0)
    print("| config | samples | CPU port (cores; sample) | 1 MI355X (kernel) | GPU / CPU | 17 flop x tests / FP32 vector peak |")
    print("|---|---|---|---|---|---|")
    for name, cfg in CONFIGS:
        t0 = time.time()
        cpu, cores, spp_cpu, secs, rps_cpu = cpu_leg(cfg, budget)
        ms, gpu, rps_gpu, tflops = gpu_leg(cfg)
        total = cfg["width"] * cfg["height"] * cfg["spp"] * 1e-6
        print("| %s | %.2f M | %.3f Msamples/s (%d threads; %d spp in %.1f s, %.3f rays/sample) | "
              "%.1f Msamples/s (%.2f ms, %.3f rays/sample) | %.0fx | %.2f |"
              % (name, total, cpu, cores, spp_cpu, secs, rps_cpu, gpu, ms, rps_gpu, gpu / cpu,
                 tflops / bench.PEAK_FP32_VALU_TFLOPS), flush=True)
        sys.stderr.write("%s done in %.0f s\n" % (name, time.time() - t0))


if __name__ == "__main__":
    main()
