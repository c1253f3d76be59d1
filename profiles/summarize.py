#!/usr/bin/env python3
"""Turns the rocprofv3 CSV trees under gpurun_out/<dir>/ into the small, tracked summaries of this
directory.  Usage: python profiles/summarize.py gpurun_out/prof5 r01

  <tag>_cfg2_kernel_stats.csv   top kernels of `bench.py` (kernel-trace --stats)
  <tag>_cfg2_pmc_hbm.csv        FETCH_SIZE / WRITE_SIZE per spmm dispatch (separate --pmc passes of
                                profiles/pmc_probe.py: 3 calibration launches, then 2 x 3 cfg2 layers)
  pmc_traffic.json              per-launch HBM bytes bench.py reports as roofline.traffic
  <tag>_infonce_kernel_stats.csv, <tag>_infonce_pmc_mfma.csv   profiles/infonce_probe.py
  <tag>_ncl_step_kernel_stats.csv, <tag>_kmeans_kernel_stats.csv  profiles/ncl_step_probe.py, kmeans_probe.py
"""
import collections
import csv
import glob
import json
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern)) or glob.glob(os.path.join(src, pattern.replace("/*/", "/")))
    return hits[0] if hits else None


def stats(pattern, out, top=12):
    path = one(pattern)
    if not path:
        return
    rows = list(csv.DictReader(open(path)))
    with open(os.path.join(here, out), "w") as f:
        w = csv.writer(f)
        cols = ["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"]
        w.writerow(cols)
        for r in rows[:top]:
            w.writerow([r["Name"][:150]] + [r[c] for c in cols[1:]])


stats("kt/*/*_kernel_stats.csv", f"{tag}_cfg2_kernel_stats.csv")
stats("kt_nce/*/*_kernel_stats.csv", f"{tag}_infonce_kernel_stats.csv", top=10)
stats("kt_ncl/*/*_kernel_stats.csv", f"{tag}_ncl_step_kernel_stats.csv", top=16)      # profiles/ncl_step_probe.py
stats("kt_km/*/*_kernel_stats.csv", f"{tag}_kmeans_kernel_stats.csv", top=6)           # profiles/kmeans_probe.py

fetch, write = one("pmc_fetch/*/*_counter_collection.csv"), one("pmc_write/*/*_counter_collection.csv")
if fetch and write:
    rows = []
    for kind, path in (("fetch", fetch), ("write", write)):
        k = 0
        for r in csv.DictReader(open(path)):
            if "spmm_" in r["Kernel_Name"]:
                k += 1
                name = "spmm_parts" if "spmm_parts" in r["Kernel_Name"] else "spmm_long_rows"
                rows.append([kind, k, name, r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]])
    with open(os.path.join(here, f"{tag}_cfg2_pmc_hbm.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "dispatch", "kernel", "grid", "counter", "value_KB"])
        w.writerows(rows)
    f_parts = [float(r[5]) for r in rows if r[0] == "fetch" and r[2] == "spmm_parts"]
    w_parts = [float(r[5]) for r in rows if r[0] == "write" and r[2] == "spmm_parts"]
    cal_f, lay_f, lay_w = f_parts[:3], f_parts[3:], w_parts[3:]
    expected_read_kb = (4e6 * 256 + 4e6 * 8 + 4e6 * 8 + 4e6 / 512 * 32) / 1024
    traffic = {"cfg2": {
        "kernel": "spmm_parts", "fetch_size_kb_avg": sum(lay_f) / len(lay_f), "write_size_kb_avg": sum(lay_w) / len(lay_w),
        "fetch_correction": 2.0,
        "calibration": {"graph": "diagonal N=4M d=64", "expected_read_kb": expected_read_kb,
                        "fetch_size_kb": sum(cal_f) / len(cal_f), "expected_write_kb": 4e6 * 256 / 1024,
                        "write_size_kb": sum(w_parts[:3]) / 3},
        "bytes_per_launch": (2.0 * sum(lay_f) / len(lay_f) + sum(lay_w) / len(lay_w)) * 1024,
        "source": f"profiles/{tag}_cfg2_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                  "profiles/pmc_probe.py)"}}
    json.dump(traffic, open(os.path.join(here, "pmc_traffic.json"), "w"), indent=1)
    print("cfg2 HBM bytes per spmm_parts launch: %.3f GB (calibration ratio %.3f)" %
          (traffic["cfg2"]["bytes_per_launch"] / 1e9, traffic["cfg2"]["calibration"]["fetch_size_kb"] / expected_read_kb))

pmc = one("pmc_nce/*/*_counter_collection.csv")
if pmc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc)):
        k = r["Kernel_Name"]
        if "infonce_fwd" in k or "infonce_bwd" in k:
            name = k.split("::")[-1].split("(")[0]          # infonce_fwd_b3_kernel<64, false, true> ...
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    with open(os.path.join(here, f"{tag}_infonce_pmc_mfma.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "counter", "avg_value", "note"])
        for k, d in agg.items():
            for c, v in d.items():
                w.writerow([k, len(v), c, sum(v) / len(v), ""])
            busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
            util = 100 * busy / ((gui / 8) * 1024)
            w.writerow([k, "", "MfmaUtil_percent", util, "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)"])
            print(k, "MfmaUtil %.1f%%" % util)
bench = os.path.join(src, "bench_kt.json")
if os.path.exists(bench):
    with open(bench) as f, open(os.path.join(here, f"{tag}_bench_cfg2_under_rocprof.json"), "w") as g:
        g.write(f.read())
