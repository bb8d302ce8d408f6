"""Reads the counter csv files of rocprofv3 --pmc runs of tools/step_count_probe.py and prints the
counters per step of the busy wave.  Usage: python tools/step_count_summary.py <dir>"""
import csv
import glob
import sys
from collections import defaultdict

per = defaultdict(dict)   # (pass, dispatch id) -> counter -> value
for path in sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)):
    with open(path) as f:
        for r in csv.DictReader(f):
            if "trace_kernel" not in r["Kernel_Name"]:
                continue
            d = per[(path, int(r["Dispatch_Id"]))]
            d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
by_pass = defaultdict(list)
for (path, did) in sorted(per):
    by_pass[path].append(did)
for path, ids in by_pass.items():
    for name, group in (("8 paths", ids[0:6]), ("64 paths", ids[6:12])):
        d50, d25 = group[0::2], group[1::2]
        for c in sorted(per[(path, ids[0])]):
            a = sum(per[(path, i)][c] for i in d50) / len(d50)
            b = sum(per[(path, i)][c] for i in d25) / len(d25)
            print("%s %-28s per step %.0f" % (name, c, (a - b) / 25.0))
