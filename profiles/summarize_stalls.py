#!/usr/bin/env python3
"""Summarise the per-pass counter CSVs of scripts/gpu_stall_counters.sh into one table.
  usage: python profiles/summarize_stalls.py gpurun_out/<dir> <tag>   ->  profiles/<tag>_infonce_stall_counters.csv"""
import collections
import csv
import glob
import os
import sys

src, tag = sys.argv[1], sys.argv[2]
here = os.path.dirname(os.path.abspath(__file__))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(src, "p*", "*", "*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "infonce_pipe" in k or "infonce_fwd_e" in k:
            name = k.replace("(anonymous namespace)::", "").replace("void ", "").split(">(")[0] + ">"
            agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = os.path.join(here, f"{tag}_infonce_stall_counters.csv")
with open(out, "w") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "quantity", "value", "note"])
    for k, d in sorted(agg.items()):
        v = {c: sum(x) / len(x) for c, x in d.items()}
        for c in sorted(v):
            w.writerow([k, c, "%.6g" % v[c], "average per dispatch"])
        wc, gui = v.get("SQ_WAVE_CYCLES"), v.get("GRBM_GUI_ACTIVE", 0) / 8
        derived = []
        if wc:
            for c, label in (("SQ_WAIT_ANY", "waves parked (s_waitcnt / barrier), share of wave cycles"),
                             ("SQ_WAIT_INST_ANY", "issue-stalled, share of wave cycles"),
                             ("SQ_ACTIVE_INST_ANY", "issuing, share of wave cycles"),
                             ("SQ_ACTIVE_INST_VALU", "issuing VALU, share of wave cycles")):
                if c in v:
                    derived.append((label, v[c] / wc))
        if gui and "SQ_VALU_MFMA_BUSY_CYCLES" in v:
            derived.append(("MFMA busy, share of SIMD cycles (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)",
                            v["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui * 1024)))
            if "SQ_VALU_MFMA_COEXEC_CYCLES" in v:
                derived.append(("VALU co-executing, share of MFMA-busy cycles",
                                v["SQ_VALU_MFMA_COEXEC_CYCLES"] / v["SQ_VALU_MFMA_BUSY_CYCLES"]))
        if "SQ_INSTS_MFMA" in v:
            derived.append(("VALU instructions per MFMA", v.get("SQ_INSTS_VALU", 0) / v["SQ_INSTS_MFMA"]))
            derived.append(("LDS instructions per MFMA", v.get("SQ_INSTS_LDS", 0) / v["SQ_INSTS_MFMA"]))
        if v.get("SQ_LDS_IDX_ACTIVE"):
            derived.append(("LDS bank-conflict cycles, share of LDS-array cycles", v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"]))
            if gui:
                derived.append(("LDS array busy, share of CU cycles", v["SQ_LDS_IDX_ACTIVE"] / (gui * 256)))
        for label, x in derived:
            w.writerow([k, label, "%.3f" % x, "derived"])
            print(k, "|", label, "%.3f" % x)
print("wrote", out)
