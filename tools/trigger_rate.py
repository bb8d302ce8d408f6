"""Developer tool: how often the scan's blocks branch to the candidate-queueing code.
Needs the build of `make -C ray_tracing_amd/csrc diag-triggers`:
    RT_HIP_LIB=ray_tracing_amd/librt_hip_diagtrig.so python tools/trigger_rate.py [grid_hi] [w] [h] [spp]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from ray_tracing_amd import hip, host  # noqa: E402


def main():
    hi = int(sys.argv[1]) if len(sys.argv) > 1 else 11
    w = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
    h = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
    spp = int(sys.argv[4]) if len(sys.argv) > 4 else 100
    hip.init(0)
    world = host.create_random_world(1, (-hi, hi))
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(cam.m, w, h, spp, 50, 1, 0)
    out = torch.empty((h, w, 3), dtype=torch.float32, device="cuda:0")
    hip.render_rows_device_async(scene, frame, 0, h, True, out.data_ptr(),
                                 torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    st = hip.get_stats()
    ph = hip.get_phase_cycles().astype(float)
    blocks, trig, pairs = ph[8], ph[9], ph[10]
    steps = ph[4]
    print("kernel %.2f ms (diagnostic build), %d rays, %.3f rays/sample" % (st.kernel_ms, st.rays, st.rays / st.samples))
    print("scan blocks per wave step %.1f; %.1f %% of blocks branch to the queueing code; "
          "%.2f (sphere, path) pairs handled per branching block; %.2f queued candidates per ray; "
          "replay loop trips per ray %.2f"
          % (blocks / max(steps, 1), 100 * trig / max(blocks, 1), pairs / max(trig, 1),
             ph[14] / max(st.rays, 1), ph[15] / max(st.rays, 1)))
    tot = ph[:4].sum()
    print("of the wave time: candidate code inside the scan %.1f%%, building ray columns %.1f%%"
          % (100 * ph[11] / tot, 100 * ph[12] / tot))
    print("wave time: refill %.1f%% scan %.1f%% replay %.1f%% shade %.1f%%"
          % tuple(100 * ph[i] / tot for i in range(4)))


if __name__ == "__main__":
    main()
