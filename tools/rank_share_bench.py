"""Developer benchmark: what one rank of an N-GPU run does, measured on one GPU.  For
N in 1, 2, 4, 8 it renders rank r's row blocks of the 1080p frame (the same call bench.py makes
on every rank) and prints kernel time, wall time per step and the scaling that would follow if
every rank took that long.
With frames_in_flight > 1 consecutive frames go to alternating streams, like bench.py does on
several GPUs, so that one frame's drain overlaps the next frame's start.
With follow_up = 1 every frame is followed, on its stream, by a small torch kernel over its
buffer: a stand-in for the gather's kernel, which like it needs a place on the GPU while the
other stream's persistent kernel fills it.
Usage: python tools/rank_share_bench.py [rank] [spp] [reps] [frames_in_flight] [follow_up]"""
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ray_tracing_amd import hip, host, multi_gpu  # noqa: E402


def main():
    pick = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    flight = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    follow_up = len(sys.argv) > 5 and sys.argv[5] == "1"
    W, H = 1920, 1080
    hip.init(0)
    world = host.create_random_world(1, (-11, 11))
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, W / H)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(cam.m, W, H, spp, 50, 1, 0)
    streams = [torch.cuda.Stream() for _ in range(flight)]
    base = None
    for n in (1, 2, 4, 8):
        rank = min(pick, n - 1)
        plan = multi_gpu.RowPlan(W, H, n)
        tiles, offs, strides = plan.tiles_for(rank)
        outs = [torch.zeros((plan.rows_per_rank, W, 3), dtype=torch.float32, device="cuda:0")
                for _ in range(flight)]
        for k in range(flight):
            hip.render_device_async(scene, frame, tiles, offs, strides, True, outs[k].data_ptr(),
                                    streams[k].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        copies = [torch.empty_like(o) for o in outs]
        for k in range(reps):
            hip.render_device_async(scene, frame, tiles, offs, strides, True,
                                    outs[k % flight].data_ptr(), streams[k % flight].cuda_stream)
            if follow_up:
                with torch.cuda.stream(streams[k % flight]):
                    copies[k % flight].copy_(outs[k % flight])
        torch.cuda.synchronize()
        for k in range(1, flight):
            assert torch.equal(outs[0].view(torch.int32), outs[k].view(torch.int32))
        wall = (time.perf_counter() - t0) / reps * 1e3
        ms = hip.get_launch_ms(reps)
        st = hip.get_stats()
        if base is None:
            base = wall
        crc = zlib.crc32(outs[0].cpu().numpy().tobytes())   # the same for every build of the library
        line = ("N=%d rank %d: %3d row blocks, kernel %.2f ms (min %.2f), wall/step %.2f ms, "
                "rays/sample %.3f, ideal %.2f ms, scaling if all ranks alike %.2fx, crc %08x"
                % (n, rank, len(tiles), ms.mean(), ms.min(), wall, st.rays / max(st.samples, 1),
                   base / n, base / wall, crc))
        ph = hip.get_phase_cycles()
        if ph[4]:
            waves = max(1, int(ph[13]))
            line += ("\n      drain share %.1f%%, longest drain %.2f of longest wave (%.0f kcycles); "
                     "per wave: %.1f scanning drain steps (max %d), %.1f tail steps (max %d), "
                     "%.1f kcycles per tail step; all steps per wave %.0f"
                     % (100.0 * ph[5] / max(1, (ph[0] + ph[1] + ph[2] + ph[3])), ph[6] / max(1, ph[7]),
                        ph[7] / 1e3, ph[8] / waves, ph[10], ph[9] / waves, ph[11],
                        ph[12] / max(1, int(ph[9])) / 1e3, ph[4] / waves))
        print(line, flush=True)
    scene.close()


if __name__ == "__main__":
    main()
