#!/usr/bin/env python3
"""Condense a rocprofv3 --kernel-trace --stats CSV (kernel_stats.csv) into a short per-kernel table."""
import csv
import re
import sys


def main(path, steps):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("# %s  (total kernel time %.2f ms over %d steps = %.2f ms/step)" % (path, tot / 1e6, steps, tot / 1e6 / steps))
    print("%-58s %7s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:32]:
        n = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")[:58]
        print("%-58s %7s %10.2f %10.1f %6.1f" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3,
                                                 100 * float(r["TotalDurationNs"]) / tot))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
