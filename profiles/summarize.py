#!/usr/bin/env python3
"""Condense rocprofv3 --kernel-trace CSVs into short tables.

  summarize.py <kernel_stats.csv> <steps>                  per-kernel totals of a --stats file
  summarize.py --trace <kernel_trace.csv> <steps>          per-kernel totals AND per (kernel, grid) = per layer shape,
                                                           restricted to the LAST <steps> training steps (steps are
                                                           delimited by the sgd_kernel dispatch), so the one-off tile
                                                           autotuning and warm-up launches of the first steps are excluded
"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "")[:52]


def stats(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s  (total kernel time %.2f ms over %d steps = %.2f ms/step)" % (path.split("/")[-1], tot / 1e6, steps, tot / 1e6 / steps))
    print("%-52s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:34]:
        print("%-52s %7s %10.2f %10.1f %6.1f" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                 float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))


def trace(path, steps):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "sgd_kernel" in r["Kernel_Name"]]
    t0 = ends[-steps - 1] if len(ends) > steps else 0
    rows = [r for r in rows if int(r["Start_Timestamp"]) >= t0]
    byk, bys = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    pers = defaultdict(list)                       # persistent kernels (one grid for every layer): durations per instantiation
    for r in rows:
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        k = short(r["Kernel_Name"])
        byk[k][0] += 1
        byk[k][1] += us
        if "igemm" in k:
            key = (k, int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]))
            bys[key][0] += 1
            bys[key][1] += us
            if "igemm_k1p" in k:
                pers[k].append(us)
    tot = sum(v[1] for v in byk.values())
    print("# %s: last %d steps, %.2f ms of kernels per step" % (path.split("/")[-1], steps, tot / 1e3 / steps))
    print("%-52s %7s %10s %10s %6s" % ("kernel", "calls", "avg_us", "ms/step", "%"))
    for k, (n, us) in sorted(byk.items(), key=lambda kv: -kv[1][1])[:30]:
        print("%-52s %7d %10.1f %10.2f %6.1f" % (k, n, us / n, us / 1e3 / steps, 100 * us / tot))
    print("\n# GEMM kernels by (instantiation, grid) = per layer shape")
    print("%-52s %9s %4s %7s %10s %10s" % ("kernel", "blocks_x", "gy", "calls", "avg_us", "ms/step"))
    for key, (n, us) in sorted(bys.items(), key=lambda kv: -kv[1][1])[:36]:
        print("%-52s %9d %4d %7d %10.1f %10.2f" % (key[0], key[1], key[2], n, us / n, us / 1e3 / steps))
    if pers:
        # the persistent patch kernel launches min(256, items) blocks whatever the layer: tell the layer shapes apart by duration
        # (a new cluster starts where a launch is > 1.25 x the cluster's shortest)
        print("\n# persistent patch kernels by duration cluster = per layer shape (S1-size launches are the longest cluster)")
        print("%-52s %7s %10s %10s %10s" % ("kernel", "calls", "avg_us", "min_us", "max_us"))
        for k, v in sorted(pers.items()):
            v.sort()
            clusters, cur = [], [v[0]]
            for x in v[1:]:
                if x > 1.25 * cur[0]:
                    clusters.append(cur)
                    cur = []
                cur.append(x)
            clusters.append(cur)
            for c in reversed(clusters):
                print("%-52s %7d %10.1f %10.1f %10.1f" % (k, len(c), sum(c) / len(c), c[0], c[-1]))


if __name__ == "__main__":
    if sys.argv[1] == "--trace":
        trace(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        stats(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
