#!/usr/bin/env python3
"""Replays a .alfd dump of a reference run (bench/reference_cmake: ALFD_EXPORT=ON) through libalfd on the GPU and
prints both iteration counts side by side -- the one route by which parity with a real deal.II + Trilinos run can be
pinned (needs a deal.II install, which this repository's pipeline does not have).

    python bench/reference_cmake/replay.py run/stokes_immersed_boundary.alfd [--reference-log run/stokes.log]
                                           [--inner-prec multilevel|chebyshev] [--support-points points.npy]

The inner preconditioner differs from the reference's by construction (Trilinos ML there; here the algebraic aggregation
of alfd_build_aggregates on the dumped A, or geometric transfers when the caller supplies them), so OUTER counts are the
comparable quantity: both inner solvers stop at the same absolute tolerance (parameters_stokes_3d.prm:23-24)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("alfd_file")
    ap.add_argument("--reference-log", help="stdout of the reference run (scrape_deallog.py)")
    ap.add_argument("--inner-prec", choices=["multilevel", "chebyshev"], default="multilevel")
    ap.add_argument("--inner-max", type=int, default=0, help="override the inner CG cap of the dump (0 = keep)")
    ap.add_argument("--support-points", help=".npy with one support point per row of block 0: front-end renumbering + mesh bricks")
    ap.add_argument("--block-size", type=int, default=0, help="components per node of block 0 (default: 3 for Stokes dumps, else 1)")
    args = ap.parse_args()
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import _abi, opfile, solver
    import scrape_deallog

    mats, diags, rhs, x0, cfg = opfile.load(args.alfd_file)
    nblocks = len(rhs)
    bs = args.block_size or (3 if _abi.BT in mats else 1)
    if args.inner_max:
        cfg.inner.max_steps = args.inner_max
    cfg.log_level = 1
    ctx = solver.Context(0)
    perm = None
    if args.support_points:
        pts = np.load(args.support_points)
        perm = solver.numbering_from_points(pts)
        inv = np.empty_like(perm)
        inv[perm] = np.arange(perm.size)
        for slot in (_abi.A, _abi.BT, _abi.CT):
            if slot in mats:
                mats[slot] = solver.permute_csr(mats[slot], row_new_to_old=perm, col_old_to_new=inv if slot == _abi.A else None)
        for slot in (_abi.B, _abi.C_):
            if slot in mats:
                mats[slot] = solver.permute_csr(mats[slot], col_old_to_new=inv)
        rhs[0] = rhs[0][perm]
        if x0 is not None:
            x0[0] = x0[0][perm]
        ctx.set_row_blocks(_abi.A, *solver.brick_blocks_from_points(pts[perm], (16, 4, 1)))
    for slot, m in mats.items():
        ctx.set_matrix(slot, m)
    for slot, d in diags.items():
        ctx.set_diag(slot, d)
    if args.inner_prec == "multilevel" and cfg.variant in (_abi.AL2, _abi.AL_STOKES, _abi.AL_STOKES_DIAG):
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_ratio, cfg.ml_coarse_degree = 4, 256.0, 10
        ctx.configure(cfg)
        levels = ctx.build_aggregates(block_size=bs, threshold=0.02, max_aggregate_nodes=8, min_coarse=3000)
        print(f"algebraic aggregation (ML's threshold 0.02, utilities.h:312): levels {[nc for _, nc in levels]}")
    else:
        cfg.inner_prec = _abi.PREC_CHEBYSHEV
    ctx.configure(cfg)
    ctx.setup([b.size for b in rhs])
    x, res = ctx.solve(rhs, x0)
    info = ctx.matrix_info(_abi.A)
    out = {"gpu": {"outer": res.outer_iterations, "inner": int(res.inner_iterations), "mp": int(res.mp_iterations),
                   "initial_residual": res.initial_residual, "final_residual": res.last_residual,
                   "solve_seconds": res.solve_seconds, "outer_iterations_per_s": res.outer_iterations / res.solve_seconds,
                   "A_storage": ("batch-major" if info["batch_major"] else "value-indexed window" if info["value_indexed"] else
                                 "window 10 B/nnz" if info["windowed"] else "csr"),
                   "A_bytes_per_nnz": info["streamed_bytes"] / max(info["nnz"], 1)},
           "blocks": [int(b.size) for b in rhs]}
    if args.reference_log:
        ref = scrape_deallog.scrape(open(args.reference_log).read())
        out["reference"] = {"outer": ref["outer"]["steps"], "inner": ref["inner"]["steps_total"],
                            "initial_residual": ref["outer"]["initial"], "final_residual": ref["outer"]["final"],
                            "solve_system_s": ref["timers"].get("Solve system", {}).get("wall_s"),
                            "outer_iterations_per_s": ref.get("outer_iterations_per_s")}
        r, g = out["reference"], out["gpu"]
        print(f"{'':24s}{'reference (deal.II + ML)':>28s}{'libalfd (MI355X)':>22s}")
        for key in ("outer", "inner", "initial_residual", "final_residual", "outer_iterations_per_s"):
            print(f"{key:24s}{str(r.get(key)):>28s}{str(g.get(key)):>22s}")
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
