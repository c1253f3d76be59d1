#!/usr/bin/env python3
"""Timeline of the last N ms of a rocprofv3 --kernel-trace CSV: start offset, duration, queue, kernel — to see what runs
beside what (show_timeline.py <dir> [window_ms] [min_us])."""
import csv, glob, os, sys
d = sys.argv[1]
win = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
p = max(glob.glob(d + "/**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(p)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = max(int(r["End_Timestamp"]) for r in rows)
t0 = t_end - int(win * 1e6)
qs = {}
busy = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e < t0:
        continue
    q = qs.setdefault(r["Queue_Id"], len(qs))
    busy.append((s, e))
    if (e - s) / 1e3 < min_us:
        continue
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    print(f"{(s - t0) / 1e6:8.3f} ms  +{(e - s) / 1e3:8.1f} us  q{q}  {'    ' * q}{name[:90]}")
busy.sort()
covered, cur_s, cur_e = 0, None, None
for s, e in busy:
    s = max(s, t0)
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            covered += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
if cur_e is not None:
    covered += cur_e - cur_s
print(f"window {win} ms: some kernel running {covered / 1e6:.3f} ms, sum of kernel durations {sum(e - max(s, t0) for s, e in busy) / 1e6:.3f} ms")
