"""Rank body of tests/test_gpu_multiprocess.py (launched by torch.distributed.run, gloo on CPU for the
transport, every rank on GPU 0): the row-partitioned AL-FGMRES solve with one PROCESS per rank, the library's
collectives carried by alfd_comm_init_host.  Rank 0 compares with the oracle's emulation of the partition."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    import cases
    from fictitious_domain_al_preconditioners_amd import _abi, partition, problems, solver
    multilevel = len(sys.argv) > 1 and sys.argv[1] == "multilevel"
    geometric = len(sys.argv) > 1 and sys.argv[1] == "geometric"      # round 3: CSR prolongators + patch + coarsest inverse
    n, ref = 8, 1 if geometric else 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    plan = partition.slab_partition_stokes3d(n, ref, world)
    full = problems.stokes3d_sphere(n, ref)
    levels = None
    if multilevel:
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 2, 8.0
        levels = partition.partitioned_geometric_aggregates(full.params, plan, a=2, min_coarse=100)
    glevels = None
    if geometric:
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.inner.max_steps = 100
        cfg.ml_smooth_degree, cfg.ml_smooth_degree_coarse, cfg.ml_smooth_ratio = 3, 4, 30.0
        cfg.ml_patch_degree, cfg.ml_patch_ratio, cfg.ml_coarse_direct = 6, 40.0, 1024
        glevels = problems.tensor_prolongators(full.params, min_coarse=100)
    pb = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
    ctx = solver.Context(0)
    ctx.comm_init_torch()
    ctx.set_partition(plan.offsets)
    solver.upload_problem(ctx, pb, cfg, partition.local_prolongators(glevels, full.params, plan, rank) if geometric
                          else partition.local_aggregates(levels, rank) if levels else None)
    rhs = ctx.augment_rhs(cases.rhs_of(pb))
    x, res = ctx.solve(rhs)
    hist = ctx.history()
    gathered = [None] * world
    dist.all_gather_object(gathered, dict(x=x, res=res.as_dict(), hist=hist))
    ok = True
    if rank == 0:
        from oracle import oracle
        osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets,
                                          aggregates=glevels if geometric else
                                          [(a, nc, coff) for a, nc, coff, _ in levels] if levels else None)
        rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
        rc, ox, ores, ohist = osys.solve(cfg, orhs)
        ok = rc == 0
        for g in gathered:
            r = g["res"]
            ok &= r["status"] == 0 and (r["outer_iterations"], r["inner_iterations"], r["mp_iterations"]) == \
                (ores.outer_iterations, ores.inner_iterations, ores.mp_iterations)
            ok &= bool(np.array_equal(g["hist"], gathered[0]["hist"]))
            ok &= bool(np.max(np.abs(g["hist"] - ohist) / np.abs(ohist)) <= 1e-10)
        for b in range(3):
            xs = np.concatenate([g["x"][b] for g in gathered])
            ok &= bool(np.allclose(xs, ox[b], rtol=1e-9, atol=1e-10 * max(np.abs(ox[b]).max(), 1e-30)))
        print("MP_RESULT " + json.dumps(dict(ok=bool(ok), world=world, outer=ores.outer_iterations,
                                             inner=ores.inner_iterations)), flush=True)
    ctx.close()
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
