#!/usr/bin/env python3
"""Regenerates profiles/spmv_traffic.json (the `traffic` field of bench.py's roofline object) from the
PMC summary profiles/r03/pmc_<tag>.json, which profiles/r03/scripts/summarise_pmc.py writes from the
`rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes of profiles/r03/scripts/collect_pmc.sh:
    gpurun -- 'bash profiles/r03/scripts/collect_pmc.sh vib'        # on the GPU box
    python profiles/r03/scripts/summarise_pmc.py vib && python profiles/r03/scripts/make_traffic_json.py
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request -> x2;
WRITE_SIZE x1 (both re-checked for 1..8-byte-per-lane loads by profiles/r01/v7_fetch_size_calibration.hip)."""
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "vs"     # "vib": the round-1 kernel; "vs": the batch-major kernel
here = os.path.dirname(os.path.abspath(__file__))
p = json.load(open(os.path.join(here, "..", f"pmc_{tag}.json")))
m = p["median"]
read_b = m["FETCH_SIZE"] * 1024 * 2
write_b = m["WRITE_SIZE"] * 1024
out = {
    "n_cells": 74,
    "value_indexed": True,
    "batch_major": 0 if tag == "vib" else 2,          # alfd_matrix_info.batch_major of the run (2 = mesh-brick row blocks)
    "bricks": sys.argv[2] if len(sys.argv) > 2 else "16,4,1",   # bench.py --bricks of the profiled run
    "kernel": p["kernel"],
    "counters": {k: m[k] for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_LDS_IDX_ACTIVE",
                                   "SQ_LDS_BANK_CONFLICT", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES") if k in m},
    "FETCH_SIZE_KB_median": m["FETCH_SIZE"], "WRITE_SIZE_KB_median": m["WRITE_SIZE"],
    "correction": "gfx950: FETCH_SIZE x2 (64 B counted per 128-B request), WRITE_SIZE x1",
    "hbm_read_bytes_per_launch": read_b, "hbm_write_bytes_per_launch": write_b,
    "hbm_bytes_per_launch": read_b + write_b,
    "l2": {k: m[k] for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_sum") if k in m},
    "source": f"profiles/r03/pmc_{tag}.json <- gpurun_out/pmc_{tag}_{{fetch,write,tcc}}/ (profiles/r03/scripts/collect_pmc.sh {tag}: "
              "`rocprofv3 --pmc ... -- python3 bench.py --profile-only-spmv 10`, N = 74^3)",
}
json.dump(out, open(os.path.join(here, "..", "..", "spmv_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
