#!/usr/bin/env python3
"""Turns the rocprofv3 CSV trees under gpurun_out/<dir>/ into the small, tracked summaries of this
directory.  Usage: python profiles/summarize.py gpurun_out/prof_r02 r02

  <dir>/kt[_cfg4]/                        bench.py under --kernel-trace --stats  -> <tag>_<wl>_kernel_stats.csv
  <dir>/pmc_fetch_<wl>/, pmc_write_<wl>/  profiles/pmc_probe.py --workload <wl> under --pmc FETCH_SIZE / WRITE_SIZE
                                          (separate passes) -> <tag>_<wl>_pmc_hbm.csv + pmc_traffic.json[<wl>]
  <dir>/kt_nce/, pmc_nce/                 profiles/infonce_probe.py -> <tag>_infonce_kernel_stats.csv,
                                          <tag>_infonce_pmc_mfma.csv + pmc_traffic.json["infonce"]
  <dir>/kt_ncl/, kt_km/, kt_rank/, kt_lg/ profiles/ncl_step_probe.py, kmeans_probe.py, rank_probe.py, lightgcn_step_probe.py
  <dir>/kt_bce/, kt_kme/, kt_cfg5/        profiles/bce_probe.py, kmeans_estep_probe.py, bench.py --workload cfg5
  <dir>/kt_gcl/                           profiles/gcl_step_probe.py
pmc_traffic.json entries carry the digest of the kernel sources the probe ran (written by the probe on the
GPU box) and the git commit of the tree they were summarised in; bench.py attaches an entry only when the
digest matches the sources it benchmarks.
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(here)


def one(pattern):
    hits = glob.glob(os.path.join(src, pattern)) or glob.glob(os.path.join(src, pattern.replace("/*/", "/")))
    return max(hits, key=os.path.getmtime) if hits else None          # several runs merged into one tree: the newest


def git_state():
    try:
        sha = subprocess.run(["git", "rev-parse", "HEAD"], cwd=root, capture_output=True, text=True).stdout.strip()
        dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "recommendation_amd", "bench.py"], cwd=root,
                                    capture_output=True, text=True).stdout.strip())
        return sha, dirty
    except OSError:
        return None, None


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
    print("wrote", out)


def stats_by_grid(pattern, bench_json, out):
    """The SpMM rows of bench.py's kernel trace, one row per (kernel, grid size): the bench line carries a cfg2 leg and a
    cfg4 companion leg that launch the SAME kernel template, which `--stats` pools into one row (r02: 204 calls with a
    2x spread).  The grid size (256 threads x ceil(partitions / 4) workgroups) tells the workloads apart; the bench JSON
    names each leg's partition count, and its live HIP-event average is written beside the trace's for comparison."""
    path = one(pattern)
    if not path:
        return
    legs = {}
    try:
        line = json.loads(open(os.path.join(src, bench_json)).read().strip().splitlines()[-1])
        legs[256 * ((line["extra"]["spmm_parts"] + 3) // 4)] = ("cfg2", line["roofline"]["avg_launch_ms"])
        c4 = line["roofline"].get("cfg4_graph")
        if c4 and "spmm_parts" in c4:
            legs[256 * ((c4["spmm_parts"] + 3) // 4)] = ("cfg4", c4["avg_launch_ms"])
    except (OSError, ValueError, KeyError, IndexError):
        pass
    groups = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "spmm_" not in r["Kernel_Name"]:
            continue
        grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        groups[(name, grid)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(here, out), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "grid_threads", "workload", "calls", "avg_ns", "min_ns", "max_ns", "bench_hip_event_avg_launch_ms",
                    "note"])
        for (name, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
            wl, ev = legs.get(grid, ("", ""))
            note = "bench avg_launch_ms spans spmm_parts + spmm_long_rows of one gcr_spmm_csr_f32 launch" if wl and "parts" in name else ""
            w.writerow([name, grid, wl, len(v), round(sum(v) / len(v), 1), min(v), max(v), ev, note])
    print("wrote", out)


stats("kt/*/*_kernel_stats.csv", f"{tag}_bench_kernel_stats.csv", top=16)
stats_by_grid("kt/*/*_kernel_trace.csv", "bench_kt.json", f"{tag}_spmm_by_workload_kernel_stats.csv")
stats("kt_cfg4/*/*_kernel_stats.csv", f"{tag}_cfg4_kernel_stats.csv")
stats("kt_nce/*/*_kernel_stats.csv", f"{tag}_infonce_kernel_stats.csv", top=10)
stats("kt_ncl/*/*_kernel_stats.csv", f"{tag}_ncl_step_kernel_stats.csv", top=16)      # profiles/ncl_step_probe.py
stats("kt_km/*/*_kernel_stats.csv", f"{tag}_kmeans_kernel_stats.csv", top=6)           # profiles/kmeans_probe.py
stats("kt_rank/*/*_kernel_stats.csv", f"{tag}_rank_kernel_stats.csv", top=6)           # profiles/rank_probe.py
stats("kt_lg/*/*_kernel_stats.csv", f"{tag}_lightgcn_step_kernel_stats.csv", top=10)     # profiles/lightgcn_step_probe.py
stats("kt_bce/*/*_kernel_stats.csv", f"{tag}_bce_kernel_stats.csv", top=10)             # profiles/bce_probe.py
stats("kt_kme/*/*_kernel_stats.csv", f"{tag}_kmeans_estep_kernel_stats.csv", top=8)     # profiles/kmeans_estep_probe.py
stats("kt_cfg5/*/*_kernel_stats.csv", f"{tag}_cfg5_kernel_stats.csv", top=14)           # bench.py --workload cfg5
stats("kt_gcl/*/*_kernel_stats.csv", f"{tag}_gcl_step_kernel_stats.csv", top=12)        # profiles/gcl_step_probe.py

traffic_path = os.path.join(here, "pmc_traffic.json")
try:
    traffic = json.load(open(traffic_path))
except (OSError, ValueError):
    traffic = {}
sha, dirty = git_state()

for wl in ("cfg2", "cfg4", "cfg2c_before", "cfg2c_after", "cfg4c_before", "cfg4c_after"):
    fetch, write = one(f"pmc_fetch_{wl}/*/*_counter_collection.csv"), one(f"pmc_write_{wl}/*/*_counter_collection.csv")
    info_path = os.path.join(root, "gpurun_out", f"pmc_probe_{wl}.json")
    if not (fetch and write and os.path.exists(info_path)):
        continue
    info = json.load(open(info_path))
    rows = []
    for kind, path in (("fetch", fetch), ("write", write)):
        k = 0
        for r in csv.DictReader(open(path)):
            if "spmm_" in r["Kernel_Name"]:
                k += 1
                name = "spmm_parts" if "spmm_parts" in r["Kernel_Name"] else "spmm_long_rows"
                rows.append([kind, k, name, r["Grid_Size"], r["Counter_Name"], r["Counter_Value"]])
    with open(os.path.join(here, f"{tag}_{wl}_pmc_hbm.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["pass", "dispatch", "kernel", "grid", "counter", "value_KB"])
        w.writerows(rows)
    f_parts = [float(r[5]) for r in rows if r[0] == "fetch" and r[2] == "spmm_parts"]
    w_parts = [float(r[5]) for r in rows if r[0] == "write" and r[2] == "spmm_parts"]
    f_long = [float(r[5]) for r in rows if r[0] == "fetch" and r[2] == "spmm_long_rows"]
    w_long = [float(r[5]) for r in rows if r[0] == "write" and r[2] == "spmm_long_rows"]
    # dispatch order: 3 calibration launches, [the renumbering's own subspace-iteration launches], then the measured
    # 2 forward passes x K layers — always the LAST 2 K dispatches
    n_meas = 2 * int(info.get("layers", 3))
    cal_f, lay_f, lay_w = f_parts[:3], f_parts[-n_meas:], w_parts[-n_meas:]
    f_long, w_long = f_long[-n_meas:], w_long[-n_meas:]
    nc = info["cal_rows"]
    expected_read_kb = (nc * 256 + nc * 8 + nc * 8 + info["cal_parts"] * 32) / 1024
    cal_ratio = (sum(cal_f) / len(cal_f)) / expected_read_kb
    avg = lambda v: sum(v) / len(v) if v else 0.0        # noqa: E731
    # one gcr_spmm_csr_f32 launch = spmm_parts + spmm_long_rows (the split rows' partial sums)
    bytes_per_launch = (2.0 * (avg(lay_f) + avg(f_long)) + avg(lay_w) + avg(w_long)) * 1024
    traffic[wl] = {
        "kernel": "spmm_parts + spmm_long_rows (one gcr_spmm_csr_f32 launch, Horner layer: reads x0, writes one array)",
        "fetch_size_kb_avg": avg(lay_f), "write_size_kb_avg": avg(lay_w),
        "long_rows_fetch_kb_avg": avg(f_long), "long_rows_write_kb_avg": avg(w_long),
        "fetch_correction": 2.0, "dispatches_averaged": len(lay_f),
        "calibration": {"graph": "diagonal N=%d d=64" % nc, "expected_read_kb": expected_read_kb,
                        "fetch_size_kb": avg(cal_f), "ratio": cal_ratio,
                        "expected_write_kb": nc * 256 / 1024, "write_size_kb": avg(w_parts[:3])},
        "bytes_per_launch": bytes_per_launch,
        "nnz": info["nnz"], "n": info["n"],
        "source_digest": info["source_digest"], "git_sha": sha, "git_dirty_at_summarise": dirty,
        "source": f"profiles/{tag}_{wl}_pmc_hbm.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, "
                  f"profiles/pmc_probe.py --workload {wl})"}
    print("%s fabric bytes per launch: %.3f GB (calibration ratio %.3f, %d dispatches)" %
          (wl, bytes_per_launch / 1e9, cal_ratio, len(lay_f)))

pmc = one("pmc_nce/*/*_counter_collection.csv")
if pmc:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pmc)):
        k = r["Kernel_Name"]
        if "infonce_fwd" in k or "infonce_bwd" in k or "infonce_pipe" in k:
            # "void (anonymous namespace)::infonce_pipe_kernel<(anonymous namespace)::EngH2, 64, 1, false, 0>(float ..."
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split(">(")[0] + ">"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    util = {}
    with open(os.path.join(here, f"{tag}_infonce_pmc_mfma.csv"), "w") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "dispatches", "counter", "avg_value", "note"])
        for k, d in agg.items():
            for c, v in d.items():
                w.writerow([k, len(v), c, sum(v) / len(v), ""])
            if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
                continue
            busy = sum(d["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(d["SQ_VALU_MFMA_BUSY_CYCLES"])
            gui = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"])
            util[k] = 100 * busy / ((gui / 8) * 1024)
            w.writerow([k, "", "MfmaUtil_percent", util[k], "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 XCDs * 1024 SIMDs)"])
            print(k, "MfmaUtil %.1f%%" % util[k])
    info_path = os.path.join(root, "gpurun_out", "pmc_probe_infonce.json")
    if util and os.path.exists(info_path):
        info = json.load(open(info_path))
        # infonce_pipe_kernel<E, D, MODE, EXD, SIDES>: MODE 0 = backward, MODE 1 = forward with the weighted row sum
        is_pipe = lambda k, mode: "infonce_pipe_kernel<" in k and k.split("<")[1].split(",")[2].strip() == str(mode)   # noqa: E731
        fwd = [v for k, v in util.items() if "infonce_fwd_" in k]
        fwdo = [v for k, v in util.items() if "infonce_fwdo_" in k or is_pipe(k, 1)]
        bwd = [v for k, v in util.items() if "infonce_bwd" in k or is_pipe(k, 0)]
        traffic["infonce"] = {
            "fwd_mfma_busy_pct": round(max(fwd), 1) if fwd else None, "bwd_mfma_busy_pct": round(max(bwd), 1) if bwd else None,
            "fwdo_mfma_busy_pct": round(max(fwdo), 1) if fwdo else None,
            "per_kernel": {k: round(v, 1) for k, v in util.items()}, "shape": info.get("shape"),
            "source_digest": info["source_digest"], "git_sha": sha, "git_dirty_at_summarise": dirty,
            "source": f"profiles/{tag}_infonce_pmc_mfma.csv (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, "
                      "profiles/infonce_probe.py)"}

json.dump(traffic, open(traffic_path, "w"), indent=1)
for name in ("bench_kt.json", "bench_cfg4_kt.json"):
    path = os.path.join(src, name)
    if os.path.exists(path):
        with open(path) as f, open(os.path.join(here, f"{tag}_{name.replace('_kt', '_under_rocprof')}"), "w") as g:
            g.write(f.read())
