#!/usr/bin/env python3
"""cfg 2 (immersed_laplace 3-D, 128^3 + sphere) and cfg 3 (elliptic_interface 2-D 1024^2 / 256^2, modified AL, beta2 = 10 and
1e3) at full size: round-2 settings (Chebyshev sweep / aggregation multigrid) against the round-3 geometric hierarchy
(CSR prolongators + interface patch + explicit coarsest inverse).  One JSON line per run."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from fictitious_domain_al_preconditioners_amd import _abi, problems, solver


def run(tag, pb, cfg, levels, rhs_fn):
    ctx = solver.Context(0)
    t0 = time.time()
    solver.upload_problem(ctx, pb, cfg, levels)
    ts = time.time() - t0
    rhs = rhs_fn(ctx)
    ctx.upload_rhs(rhs)
    try:
        ctx.solve_resident()
        res = ctx.solve_resident()
        out = {"case": tag, "outer": res.outer_iterations, "inner": int(res.inner_iterations), "solve_s": round(res.solve_seconds, 4),
               "setup_s": round(ts, 2), "residual": res.last_residual}
    except Exception as e:
        out = {"case": tag, "error": str(e)[:200]}
    print(json.dumps(out), flush=True)
    ctx.close()


which = sys.argv[1:] or ["cfg2", "cfg3"]
if "cfg2" in which:
    pb = problems.laplace3d_sphere(128, 5)
    base = _abi.default_config(_abi.AL2)
    base.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
    rhs_fn = lambda ctx: ctx.augment_rhs([pb.vecs["f"], pb.vecs["g"]])
    c = _abi.Config.from_buffer_copy(base); c.inner.max_steps = 2000
    run("cfg2 chebyshev(4) sweep (round 2 best)", pb, c, None, rhs_fn)
    c = _abi.bench_multilevel_settings(_abi.Config.from_buffer_copy(base), geometric=True)
    run("cfg2 geometric + patch(20/400) (bench settings)", pb, c, problems.tensor_prolongators(pb.params, min_coarse=1024), rhs_fn)
    c.ml_patch_degree, c.ml_patch_ratio, c.ml_smooth_degree, c.ml_smooth_degree_coarse = 8, 60.0, 2, 3
    run("cfg2 geometric V(2,2)/30 coarse 3 + patch(8/60)", pb, c, problems.tensor_prolongators(pb.params, min_coarse=1024), rhs_fn)
if "cfg3" in which:
    for beta2 in (10.0, 1e3):
        pb = problems.elliptic_interface2d(1024, 256, beta2=beta2)
        base = _abi.default_config(_abi.AL_ELL_MODIFIED)
        base.gamma, base.gamma2 = 10.0, 1e-2
        base.inner = _abi.Control(_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)
        base.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-10)
        rhs_fn = lambda ctx: [pb.vecs["f"].copy(), pb.vecs["f2"].copy(), np.zeros(pb.block_sizes[2])]
        c = _abi.Config.from_buffer_copy(base); c.inner_prec = _abi.PREC_MULTILEVEL
        c.ml_smooth_degree, c.ml_smooth_ratio = 2, 8.0
        run(f"cfg3 beta2={beta2:g} aggregation multigrid (round 2)", pb, c, problems.geometric_aggregates(pb, a=2), rhs_fn)
        c = _abi.bench_multilevel_settings(_abi.Config.from_buffer_copy(base), geometric=True)
        c.inner = base.inner
        run(f"cfg3 beta2={beta2:g} geometric + patch(20/400)", pb, c, problems.tensor_prolongators(pb.params, min_coarse=1024), rhs_fn)
        c.ml_patch_degree = 0
        run(f"cfg3 beta2={beta2:g} geometric, no patch", pb, c, problems.tensor_prolongators(pb.params, min_coarse=1024), rhs_fn)
