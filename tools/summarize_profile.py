#!/usr/bin/env python3
"""Condense the output of tools/profile_bench.sh into the files kept under profiles/:
kernel_stats.csv (rocprofv3's own per-kernel summary) and pmc_per_launch.csv (each counter
summed over the XCD instances rocprofv3 reports and divided by the number of launches).

    python3 tools/summarize_profile.py gpurun_out/<dir> profiles/<round>/<prefix>
"""
import collections
import csv
import glob
import os
import shutil
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    ks = glob.glob(os.path.join(src, "kt", "**", "*kernel_stats.csv"), recursive=True)
    if ks:
        shutil.copy(ks[0], dst + "_kernel_stats.csv")
    if os.path.exists(os.path.join(src, "bench.json")):
        shutil.copy(os.path.join(src, "bench.json"), dst + "_bench.json")
    rows = []
    for name in ("fetch", "write", "sqA", "sqB", "sqC"):
        files = glob.glob(os.path.join(src, name, "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        total = collections.defaultdict(float)
        launches = collections.defaultdict(set)
        with open(files[0], newline="") as f:
            for r in csv.DictReader(f):
                k = r["Kernel_Name"]
                if "rt::" not in k:
                    continue
                k = k.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
                k = k.replace(", ", ",")
                total[(k, r["Counter_Name"])] += float(r["Counter_Value"])
                launches[k].add(r["Dispatch_Id"])
        for (k, c), v in sorted(total.items()):
            n = len(launches[k])
            rows.append(("pass " + name, k, c, n, "%.6e" % (v / n)))
    with open(dst + "_pmc_per_launch.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["pass", "kernel", "counter", "launches", "value_per_launch"])
        w.writerows(rows)
    print("wrote", dst + "_*")


if __name__ == "__main__":
    main()
