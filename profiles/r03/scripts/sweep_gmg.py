#!/usr/bin/env python3
"""Sweep of the round-3 inner preconditioner settings on one uploaded problem (N from argv):
smoother degree / ratio, interface-patch degree / ratio.  One line per setting."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from fictitious_domain_al_preconditioners_amd import _abi, problems, solver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 74
settings = [tuple(float(v) for v in a.split(",")) for a in sys.argv[2:]] or [(4, 30, 5, 30)]
refine = max(0, int(round(np.log2(n / 64.0))) + 4)
pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine)
levels = problems.tensor_prolongators(pb.params, min_coarse=1024)
cfg = _abi.default_config(_abi.AL_STOKES)
cfg.inner_prec = _abi.PREC_MULTILEVEL
cfg.inner.max_steps = 100
cfg.ml_coarse_direct = 1024
ctx = solver.Context(0)
rb = problems.brick_row_blocks(pb.params, (16, 4, 1))
first = True
for st in settings:
    k, ratio, pd, pr = st[:4]
    cfg.ml_smooth_degree_coarse = int(st[4]) if len(st) > 4 else 0
    cfg.ml_smooth_degree, cfg.ml_smooth_ratio = int(k), ratio
    cfg.ml_patch_degree, cfg.ml_patch_ratio = int(pd), pr
    t0 = time.time()
    if first:
        solver.upload_problem(ctx, pb, cfg, levels, rb)
        rhs = ctx.augment_rhs([pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]])
        first = False
    else:
        ctx.configure(cfg)
        ctx.setup(pb.block_sizes)
    ctx.upload_rhs(rhs)
    ts = time.time() - t0
    try:
        ctx.solve_resident()
        res = ctx.solve_resident()
        print(json.dumps({"n": n, "smooth": [int(k), ratio], "patch": [int(pd), pr], "coarse_smooth": cfg.ml_smooth_degree_coarse, "outer": res.outer_iterations,
                          "inner": res.inner_iterations, "solve_s": round(res.solve_seconds, 4),
                          "it_per_s": round(res.outer_iterations / res.solve_seconds, 3), "setup_s": round(ts, 1)}), flush=True)
    except Exception as e:
        print(json.dumps({"n": n, "smooth": [int(k), ratio], "patch": [int(pd), pr], "error": str(e)}), flush=True)
