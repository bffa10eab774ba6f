"""Scratch: the A-SpMV of cfg 5 (215^3 vector Q1 elasticity) in the storage forms of the library."""
import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
_sys.path.insert(0, _ROOT); _sys.path.insert(0, _os.path.join(_ROOT, "tests"))
import sys, os, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fictitious_domain_al_preconditioners_amd import _abi, problems, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
t = time.time(); pb = problems.elasticity3d(n); m = pb.mats["A"]
print("generated", m.nrows, m.nnz, f"{time.time()-t:.1f}s", pb.params, flush=True)
ctx = solver.Context(0)
for tag, bm, blocks in (("vib", 0, None), ("runs", 1, None), ("b16-4-1", 1, (16, 4, 1)), ("b8-4-2", 1, (8, 4, 2)), ("b16-4-2", 1, (16, 4, 2)), ("b32-2-1", 1, (32, 2, 1))):
    t = time.time()
    ctx.set_tunable("batch_major", bm)
    if blocks: ctx.set_row_blocks(_abi.A, *problems.brick_row_blocks(pb.params, blocks))
    else: ctx.set_row_blocks(_abi.A, None, None)
    ctx.set_matrix(_abi.A, m)
    info = ctx.matrix_info(_abi.A)
    ms = min(ctx.bench_spmv_format(_abi.A, 20, True)[0] for _ in range(2))
    nb = ctx.bench_spmv_format(_abi.A, 1, True)[1]
    print(tag, f"{ms:.4f} ms", f"{nb/1e9:.3f} GB", f"{nb/m.nnz:.3f} B/nnz", "fmt", info["batch_major"], "vi", info["value_indexed"], f"(upload {time.time()-t:.0f}s)", flush=True)
