#!/usr/bin/env python3
"""Summarises gpurun_out/pmc_<tag>_*/**/*counter_collection.csv (profiles/r03/scripts/collect_pmc.sh) into
profiles/r03/pmc_<tag>.json: per counter the median over the dispatches of the dominant A-SpMV kernel."""
import csv, glob, json, os, statistics, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
rows = {}
kernel = None
for f in glob.glob(os.path.join(root, "gpurun_out", f"pmc_{tag}_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "spmv_window" not in k and "spmv_vs_kernel" not in k:
            continue
        kernel = kernel or k
        rows.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {"tag": tag, "kernel": kernel, "dispatches_per_counter": {k: len(v) for k, v in rows.items()},
       "median": {k: statistics.median(v) for k, v in rows.items()}}
m = out["median"]
if "SQ_WAVE_CYCLES" in m:
    wc = m["SQ_WAVE_CYCLES"]
    out["share_of_wave_cycles"] = {k: m[k] / wc for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY",
                                   "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM") if k in m}
if "SQ_LDS_IDX_ACTIVE" in m and m["SQ_LDS_IDX_ACTIVE"]:
    out["lds_bank_conflict_share"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
if "FETCH_SIZE" in m:
    out["hbm_read_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2   # gfx950: 64 B counted per 128-B request
if "WRITE_SIZE" in m:
    out["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
json.dump(out, open(os.path.join(root, "profiles", "r03", f"pmc_{tag}.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
