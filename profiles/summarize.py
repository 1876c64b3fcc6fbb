#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace --stats CSVs into short tables.

  summarize.py <kernel_stats.csv> <steps>            per-kernel totals
  summarize.py --trace <kernel_trace.csv> <steps>    per (kernel, grid) = per layer-shape averages
                                                     (one template instantiation serves several layers;
                                                      the grid size separates them)
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:46]


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s  (total kernel time %.2f ms over %d steps = %.2f ms/step)" % (path.split("/")[-1], tot / 1e6, steps, tot / 1e6 / steps))
    print("%-46s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:34]:
        print("%-46s %7s %10.2f %10.1f %6.1f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))


def trace(path, steps):
    agg = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if "igemm" not in r["Kernel_Name"]:
            continue
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]),
               int(r["Grid_Size_Z"]))
        a = agg[key]
        a[0] += 1
        a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    print("# %s: GEMM kernels by (instantiation, grid) over %d steps" % (path.split("/")[-1], steps))
    print("%-46s %9s %4s %4s %7s %10s %10s" % ("kernel", "blocks_x", "gy", "gz", "calls", "avg_us", "ms/step"))
    for key, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
        print("%-46s %9d %4d %4d %7d %10.1f %10.2f" % (key[0], key[1], key[2], key[3], n, us / n, us / 1e3 / steps))


if __name__ == "__main__":
    if sys.argv[1] == "--trace":
        trace(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        stats(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
