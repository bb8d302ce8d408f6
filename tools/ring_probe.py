"""Developer tool: how the waves' rings of open pixels behave on one frame.
Needs the build of `make -C ray_tracing_amd/csrc diag-ring`:
    RT_HIP_LIB=ray_tracing_amd/librt_hip_diagring.so python tools/ring_probe.py [grid_hi] [w] [h] [spp] [row_end]"""
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
    rows = int(sys.argv[5]) if len(sys.argv) > 5 else h
    hip.init(0)
    world = host.create_random_world(1, (-hi, hi))
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    scene = hip.Scene(world.spheres, world.materials)
    frame = hip.make_frame(cam.m, w, h, spp, 50, 1, 0)
    out = torch.empty((rows, w, 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    for name, call in (("allocating", hip.render_rows_device_async),
                       ("fixed buffer", hip.render_rows_device_noalloc)):
        call(scene, frame, 0, rows, True, out.data_ptr(), stream)
        torch.cuda.synchronize()
        st = hip.get_stats()
        ph = hip.get_phase_cycles().astype(float)
        steps = ph[4]
        print("%s: kernel %.2f ms (diagnostic build), %d wave steps; starved of slots in %.2f %% of "
              "them, %.1f of 128 path slots idle in a starved step (%.2f %% of all slot-steps); "
              "%.0f merges of %.1f pixels" % (name, st.kernel_ms, steps, 100 * ph[8] / steps,
                                             ph[9] / max(ph[8], 1), 100 * ph[9] / (128 * steps),
                                             ph[10], ph[11] / max(ph[10], 1)))


if __name__ == "__main__":
    main()
