#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace CSV over the last N steps of a bench run (a step starts at `marker`).

    python tools/trace_summary.py gpurun_out/x/prof/run_kernel_trace.csv --marker cast_b16 --steps 5
"""
import argparse
import csv
import re
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--marker", default="cast_b16", help="substring of the kernel that opens a step")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--top", type=int, default=24)
    ap.add_argument("--grids", default="conv_b16,wgrad_b16", help="comma list: kernels to break down by grid")
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.trace)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if a.marker in r["Kernel_Name"]]
    # the last `steps` occurrences of the marker open the last `steps` steps
    sel = rows[idx[-a.steps]:] if len(idx) >= a.steps else rows
    agg, agg2 = defaultdict(lambda: [0, 0.0]), defaultdict(lambda: [0, 0.0])
    pats = [p for p in a.grids.split(",") if p]
    for r in sel:
        nme = r["Kernel_Name"]
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        k = re.sub(r"\(.*", "", nme).replace("void ", "")
        agg[k][0] += 1
        agg[k][1] += d
        if any(p in nme for p in pats):
            k2 = (k.replace("cstp::", ""), int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
            agg2[k2][0] += 1
            agg2[k2][1] += d
    n = a.steps
    tot = sum(v[1] for v in agg.values())
    span = (int(sel[-1]["End_Timestamp"]) - int(sel[0]["Start_Timestamp"])) / 1e6
    print("# %s: last %d steps: %.2f ms of kernels per step, %.2f ms wall per step, %d launches per step" % (
        a.trace.split("/")[-1], n, tot / n, span / n, len(sel) // n))
    print("%-84s %9s %10s %9s" % ("kernel", "calls/step", "ms/step", "avg_us"))
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:a.top]:
        print("%-84s %9.1f %10.3f %9.1f" % (k[:84], v[0] / n, v[1] / n, v[1] / v[0] * 1e3))
    if agg2:
        print("\n# by (instantiation, grid)")
        for k, v in sorted(agg2.items(), key=lambda kv: -kv[1][1])[:a.top]:
            print("%-30s blocks %6d x %3d x %3d %9.1f %10.3f %9.1f" % (k[0], k[1], k[2], k[3], v[0] / n, v[1] / n, v[1] / v[0] * 1e3))


if __name__ == "__main__":
    main()
