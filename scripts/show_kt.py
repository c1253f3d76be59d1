#!/usr/bin/env python3
"""Prints the per-kernel totals of a rocprofv3 --stats tree: show_kt.py <dir> [divide-by-steps] [rows]"""
import csv
import glob
import sys

d = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
import os
p = max(glob.glob(d + "/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(p)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.3f ms per step (%d kernels names)" % (tot / steps / 1e6, len(rows)))
for r in rows[:top]:
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{name[:100]:100s} {float(r['Calls']) / steps:8.1f}/step {int(r['TotalDurationNs']) / steps / 1e6:8.3f} ms/step  avg {float(r['AverageNs']) / 1e3:8.1f} us")
