"""Developer probe: instructions one step of a lone wave issues, by the number of paths it has.
The benchmark scene inside a huge diffuse sphere (every path runs max_depth segments), one wave's
worth of paths or fewer, frames of depth 50 and 25: the difference over 25 is one step.
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU ... -d DIR -- python3 tools/drain_step_counters.py
    python3 tools/drain_step_counters.py --read DIR"""
import csv
import glob
import os
import sys

CASES = ((4, 2, 1), (4, 4, 1), (8, 4, 1), (8, 8, 1), (8, 8, 2))
LAUNCHES = 4

if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = []
    for f in sorted(glob.glob(os.path.join(sys.argv[2], "**", "*counter_collection.csv"), recursive=True)):
        with open(f) as fh:
            rows += [r for r in csv.DictReader(fh) if "trace_kernel" in r["Kernel_Name"]]
    by = {}
    for r in rows:
        by.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), 0.0)
        by[r["Counter_Name"]][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
    for name, d in sorted(by.items()):
        vals = [d[k] for k in sorted(d)]
        line = []
        for i, (w, h, spp) in enumerate(CASES):
            a = vals[(2 * i) * LAUNCHES:(2 * i + 1) * LAUNCHES]
            b = vals[(2 * i + 1) * LAUNCHES:(2 * i + 2) * LAUNCHES]
            line.append("%d paths: %.0f" % (w * h * spp, (sum(a) / len(a) - sum(b) / len(b)) / 25))
        print("%-28s per step: %s" % (name, "; ".join(line)))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from ray_tracing_amd import abi, hip, host  # noqa: E402

hip.init(0)
world = host.create_random_world(1, (-11, 11))
s = np.concatenate([world.spheres, world.spheres[:1]])
s[-1]["x"] = s[-1]["y"] = s[-1]["z"] = 0.0
s[-1]["radius"] = 5000.0
scene = hip.Scene(s, world.materials)
stream = torch.cuda.current_stream().cuda_stream
for w, h, spp in CASES:
    cam = host.Camera.new((13, 2, 3), (0, 0, 0), 20.0, w / h)
    out = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda:0")
    for depth in (50, 25):
        frame = hip.make_frame(cam.m, w, h, spp, depth, 1, abi.RT_FLAG_SEQUENTIAL_SUM)
        for _ in range(LAUNCHES):
            hip.render_rows_device_async(scene, frame, 0, h, False, out.data_ptr(), stream)
        torch.cuda.synchronize()
print("done", flush=True)
