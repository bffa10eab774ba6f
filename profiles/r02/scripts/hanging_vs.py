"""Scratch: the batch-major format on an operator with one condensed layer of hanging nodes (cases.hanging_node_variant)."""
import os as _os, sys as _sys
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
_sys.path.insert(0, _ROOT); _sys.path.insert(0, _os.path.join(_ROOT, "tests"))
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import cases
from fictitious_domain_al_preconditioners_amd import _abi, problems, solver
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
gpu = len(sys.argv) > 2
pb0 = problems.stokes3d_sphere(n, 1)
t = time.time(); pb = cases.hanging_node_variant(pb0); print("condensed in %.1f s, constrained rows %d" % (time.time() - t, pb.n_constrained), flush=True)
for name, p in (("uniform", pb0), ("hanging", pb)):
    m = p.mats["A"]
    lens = np.diff(m.row_ptr)
    for tag, blk in (("runs", None), ("bricks", problems.brick_row_blocks(pb0.params, (16, 4, 1)))):
        i = solver.host_stream_plan(m, blocks=blk)
        print(name, tag, "rows", m.nrows, "nnz", m.nnz, "max len", lens.max(), "ok", i["ok"], "B/nnz %.3f" % (i["stream_bytes"] / m.nnz) if i["ok"] else "", "shared %.3f" % (i["shared_nnz"] / m.nnz) if i["ok"] else "", flush=True)
        if gpu and i["ok"]:
            ctx = solver.Context(0)
            if blk: ctx.set_row_blocks(_abi.A, *blk)
            ctx.set_matrix(_abi.A, m)
            ms = min(ctx.bench_spmv_format(_abi.A, 30, True)[0] for _ in range(2))
            ctx.set_tunable("batch_major", 0)
            ms0 = min(ctx.bench_spmv_format(_abi.A, 30, True)[0] for _ in range(2))
            print("   GPU: batch-major %.4f ms, round-1 kernel %.4f ms, fmt %d" % (ms, ms0, ctx.matrix_info(_abi.A)["batch_major"]), flush=True)
            ctx.close()
