"""Developer soak: whole frames of random size, sample count, seed and scene size through the three
ways to run World::hit (matrix filter, vector filter, hierarchy) and, every fourth, the literal
arithmetic: all must give the same bits and the same ray count.
    python tools/soak_frames.py [n_frames] [seed0]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ray_tracing_amd import abi, hip, host  # noqa: E402


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    hip.init(0)
    stream = torch.cuda.current_stream().cuda_stream
    t0 = time.time()
    rays = 0
    scenes = {}
    for k in range(n_frames):
        rs = np.random.default_rng(seed0 + k)
        grid = int(rs.choice([4, 11, 11, 11, 16, 25]))
        glass = bool(rs.integers(0, 4) == 0)
        key = (grid, glass)
        if key not in scenes:
            w = (host.create_random_world(1, (-grid, grid), 0.1, 0.2, True) if glass
                 else host.create_random_world(1, (-grid, grid)))
            scenes[key] = hip.Scene(w.spheres, w.materials)
        scene = scenes[key]
        W, H = int(rs.integers(64, 1921)), int(rs.integers(36, 1081))
        spp = int(rs.choice([1, 2, 3, 7, 16, 33, 64]))
        if W * H * spp > 1920 * 1080 * 24:
            spp = max(1, 1920 * 1080 * 24 // (W * H))
        depth = int(rs.choice([1, 3, 50, 50, 50]))
        seed = int(rs.integers(0, 2 ** 31))
        cam = host.Camera.new((13.0, 2.0, 3.0), (0.0, 0.0, 0.0), float(rs.uniform(15, 50)), W / H)
        ref = None
        modes = [0, abi.RT_FLAG_SCAN_VALU, abi.RT_FLAG_ACCEL] + ([abi.RT_FLAG_STRICT_FP] if k % 4 == 0 else [])
        for flags in modes:
            out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda:0")
            frame = hip.make_frame(cam.m, W, H, spp, depth, seed, flags=flags)
            hip.render_rows_device_async(scene, frame, 0, H, True, out.data_ptr(), stream)
            torch.cuda.synchronize()
            st = hip.get_stats()
            if ref is None:
                ref, ref_rays = out, st.rays
            elif not torch.equal(out.view(torch.int32), ref.view(torch.int32)) or st.rays != ref_rays:
                bad = int((out.view(torch.int32) != ref.view(torch.int32)).any(dim=2).sum())
                print("MISMATCH frame %d (seed %d): grid %d glass %d %dx%d spp %d depth %d flags %d: %d pixels, "
                      "rays %d vs %d" % (k, seed0 + k, grid, glass, W, H, spp, depth, flags, bad, st.rays, ref_rays),
                      flush=True)
                sys.exit(1)
        rays += ref_rays
        if k % 10 == 9:
            print("%d frames, %.3g rays, %.0f s" % (k + 1, rays, time.time() - t0), flush=True)
    print("all equal: %d frames, %.3g rays" % (n_frames, rays))


if __name__ == "__main__":
    main()
