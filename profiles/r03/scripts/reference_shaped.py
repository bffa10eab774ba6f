#!/usr/bin/env python3
"""What a reference-shaped operator gets: the Stokes block (0,0) assembled CELL BY CELL (one
numerically integrated cell matrix, contributions summed in Morton order of the cells: synth.h `assembly`), optionally
in a Cuthill-McKee node numbering (stokes_immersed_boundary.cc:533-541), uploaded (a) as handed over, with row blocks
from support points, (b) after the front end's renumbering from support points.  Per variant: storage form, dictionary
and translate-sharing shares, B/nnz, A-SpMV ms, solve s, it/s.   usage: reference_shaped.py N [variants...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from fictitious_domain_al_preconditioners_amd import _abi, problems, solver


def run_variant(n, assembly, numbering, frontend, bricks=(16, 4, 1), solve=True, log=print):
    refine = max(0, int(round(np.log2(n / 64.0))) + 4)
    t0 = time.time()
    pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine, assembly=assembly)
    t_gen = time.time() - t0
    t0 = time.time()
    if numbering == "cuthill_mckee":
        problems.permute_background_nodes(pb, problems.cuthill_mckee_nodes(pb))
    t_num = time.time() - t0
    perm = getattr(pb, "node_permutation", None)
    pts = problems.row_support_points(pb.params, node_permutation=perm)
    t0 = time.time()
    if frontend == "renumber":          # what the adapter does before the upload: lexicographic order of the support points
        n2o = solver.numbering_from_points(pts)
        nc = pb.params["ncomp"]
        assert np.all(n2o.reshape(-1, nc) // nc == (n2o[::nc] // nc)[:, None])      # components of a node stay together
        problems.permute_background_nodes(pb, n2o[::nc] // nc)
        perm = pb.node_permutation
        pts = problems.row_support_points(pb.params, node_permutation=perm)
        blocks = solver.brick_blocks_from_points(pts, bricks)
    elif frontend == "bricks":          # caller's numbering kept, bricks found from the support points
        blocks = solver.brick_blocks_from_points(pts, bricks)
    elif frontend == "bisection":       # caller's numbering kept, coordinate-bisection row blocks
        blocks = solver.row_blocks_from_points(pts, 192)
    else:
        blocks = None
    t_front = time.time() - t0
    cfg = _abi.bench_multilevel_settings(_abi.default_config(_abi.AL_STOKES), geometric=True)
    levels = problems.tensor_prolongators(pb.params, min_coarse=_abi.BENCH_MIN_COARSE, node_permutation=perm)
    ctx = solver.Context(0)
    t0 = time.time()
    solver.upload_problem(ctx, pb, cfg, levels, blocks)
    t_up = time.time() - t0
    info = ctx.matrix_info(_abi.A)
    ms, _ = ctx.bench_spmv(_abi.A, 20)
    out = {"n_cells": n, "assembly": assembly, "numbering": numbering, "front_end": frontend,
           "distinct_values_A": int(np.unique(pb.mats["A"].val).size) if pb.mats["A"].nnz < 4e8 else None,
           "nnz_A": info["nnz"], "storage": ("batch-major" if info["batch_major"] else "value-indexed window" if info["value_indexed"]
                                             else "window 10 B/nnz" if info["windowed"] else "csr"),
           "dictionary_share": info["value_indexed_nnz"] / max(info["nnz"], 1),
           "shared_share": info["shared_nnz"] / max(info["nnz"], 1),
           "bytes_per_nnz": info["streamed_bytes"] / max(info["nnz"], 1), "spmv_A_ms": ms,
           "generate_s": t_gen, "numbering_s": t_num, "front_end_s": t_front, "upload_setup_s": t_up}
    if solve:
        rhs = ctx.augment_rhs([pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]])
        ctx.upload_rhs(rhs)
        ctx.solve_resident()
        res = ctx.solve_resident()
        out.update({"outer": res.outer_iterations, "inner": res.inner_iterations, "solve_s": res.solve_seconds,
                    "it_per_s": res.outer_iterations / res.solve_seconds, "final_residual": res.last_residual})
    ctx.close()
    return out


VARIANTS = {
    "kronecker_lex": ("kronecker", "lexicographic", "bricks"),            # the bench operator
    "cellwise_lex": ("cellwise", "lexicographic", "bricks"),              # cell-wise sums, same numbering
    "cellwise_cm_asis": ("cellwise", "cuthill_mckee", "none"),            # Cuthill-McKee numbering, no hint at all
    "cellwise_cm_bisection": ("cellwise", "cuthill_mckee", "bisection"),  # ... row blocks by coordinate bisection
    "cellwise_cm_bricks": ("cellwise", "cuthill_mckee", "bricks"),        # ... mesh bricks from support points
    "cellwise_cm_renumber": ("cellwise", "cuthill_mckee", "renumber"),    # ... front-end renumbering + bricks
}

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    for name in (sys.argv[2:] or list(VARIANTS)):
        r = run_variant(n, *VARIANTS[name])
        r["variant"] = name
        print(json.dumps(r), flush=True)
