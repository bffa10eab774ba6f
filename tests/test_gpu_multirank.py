"""The row-partitioned multi-rank path executed on ONE GPU: N contexts, one host
thread per rank, exchanging through the in-process rank group
(alfd_local_group) instead of RCCL.  Everything except the literal RCCL calls
runs: per-rank generation, halo plans and id exchange at upload, pack kernels,
halo SpMV, all-gather + rank-ordered reductions, the whole AL-FGMRES solve.
Expected: identical iteration counts and residual history as the oracle's
emulation of the same partition; the assembled solution equals the 1-rank one
to rounding."""
import threading

import numpy as np
import pytest

import cases
from fictitious_domain_al_preconditioners_amd import _abi, partition, problems, solver
from oracle import oracle

pytestmark = pytest.mark.gpu


def _run_ranks(world, n, ref, cfg, levels=None, plan=None):
    plan = plan or partition.slab_partition_stokes3d(n, ref, world)
    group = solver.LocalGroup(world)
    out = [None] * world
    errs = []

    def work(rank):
        try:
            pb = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
            ctx = solver.Context(0)
            ctx.comm_init_local(group.handle, rank)
            ctx.set_partition(plan.offsets)
            solver.upload_problem(ctx, pb, cfg, partition.local_aggregates(levels, rank) if levels else None)
            rhs = ctx.augment_rhs(cases.rhs_of(pb))
            x, res = ctx.solve(rhs)
            sysx = ctx.system_apply(x)
            out[rank] = dict(x=x, res=res.as_dict(), hist=ctx.history(), rhs=rhs, ax=sysx)
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    group.close()
    assert not errs, errs
    assert all(o is not None for o in out)
    return plan, out


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_solve_matches_oracle_emulation(built, world):
    n, ref = 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    plan, out = _run_ranks(world, n, ref, cfg)
    full = problems.stokes3d_sphere(n, ref)
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets)
    rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
    rc, ox, ores, ohist = osys.solve(cfg, orhs)
    assert rc == 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert res["outer_iterations"] == ores.outer_iterations
        assert res["inner_iterations"] == ores.inner_iterations
        assert res["mp_iterations"] == ores.mp_iterations
        assert res["lambda_max"] == ores.lambda_max
        assert np.array_equal(out[r]["hist"], out[0]["hist"])          # every rank sees the same scalars
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    # stitched solution == oracle solution; stitched rhs == augmented global rhs
    for b in range(3):
        xs = np.concatenate([out[r]["x"][b] for r in range(world)])
        assert np.allclose(xs, ox[b], rtol=1e-9, atol=1e-10 * max(np.abs(ox[b]).max(), 1e-30))
        assert np.array_equal(np.concatenate([out[r]["rhs"][b] for r in range(world)]), orhs[b])
    # true residual of the stitched solution
    r2 = sum(float(np.dot(out[r]["rhs"][b] - out[r]["ax"][b], out[r]["rhs"][b] - out[r]["ax"][b]))
             for r in range(world) for b in range(3))
    assert np.sqrt(r2) <= 10 * max(cfg.outer.tol, cfg.outer.reduce * ores.initial_residual)


def test_partitioned_matches_single_rank_to_rounding(built):
    """2-rank vs 1-rank: same algorithm, different dot association -> same counts here, close histories."""
    n, ref = 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    _, out = _run_ranks(2, n, ref, cfg)
    pb = problems.stokes3d_sphere(n, ref)
    ctx = solver.context_from_problem(pb, cfg)
    x, res = ctx.solve(ctx.augment_rhs(cases.rhs_of(pb)))
    h1 = ctx.history()
    ctx.close()
    assert out[0]["res"]["outer_iterations"] == res.outer_iterations
    assert np.allclose(out[0]["hist"], h1, rtol=1e-6)


def test_partitioned_bench_like_solve_matches_single_rank(built):
    """The bench configuration on two ranks at a size where the operators use the LDS-window / batch-major
    formats (halo columns inside the staged windows, multigrid level matrices partitioned or replicated):
    same iteration counts as the single-rank solve with the SAME slab-respecting aggregates, histories equal to
    the rounding of the differently associated dot products."""
    n, ref, world = 28, 1, 2
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 100
    cfg.inner_prec = _abi.PREC_MULTILEVEL
    cfg.ml_smooth_degree, cfg.ml_smooth_ratio, cfg.ml_coarse_degree = 4, 256.0, 10
    full = problems.stokes3d_sphere(n, ref)
    plan = partition.slab_partition_stokes3d(n, ref, world)
    levels = partition.partitioned_geometric_aggregates(full.params, plan, a=2, min_coarse=600)
    plan, out = _run_ranks(world, n, ref, cfg, levels)
    ctx = solver.Context(0)
    ctx.set_row_blocks(_abi.A, *problems.brick_row_blocks(full.params, (16, 4, 1)))
    solver.upload_problem(ctx, full, cfg, [(a, nc) for a, nc, _, _ in levels])
    assert ctx.matrix_info(_abi.A)["batch_major"] == 2
    x, res = ctx.solve(ctx.augment_rhs(cases.rhs_of(full)))
    h1 = ctx.history()
    ctx.close()
    for r in range(world):
        assert out[r]["res"]["status"] == 0
        assert (out[r]["res"]["outer_iterations"], out[r]["res"]["inner_iterations"]) == \
            (res.outer_iterations, res.inner_iterations)
        assert np.allclose(out[r]["hist"], h1, rtol=1e-6)
    xs = np.concatenate([out[r]["x"][0] for r in range(world)])
    assert np.allclose(xs, x[0], rtol=1e-6, atol=1e-8 * np.abs(x[0]).max())


@pytest.mark.parametrize("world", [2, 3])
def test_partitioned_multilevel_solve_matches_oracle_emulation(built, world):
    """ALFD_PREC_MULTILEVEL on the row-partitioned path: slab-respecting aggregates,
    aggregate ids of halo columns fetched through the matrices' halo plans, per-level
    partitions, rank-local transfers.  Must reproduce the oracle's emulation (which builds
    ONE global hierarchy from the same aggregates) -- same counts, history within 1e-10."""
    n, ref = 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    cfg.inner_prec = _abi.PREC_MULTILEVEL
    cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 2, 8.0
    full = problems.stokes3d_sphere(n, ref)
    plan = partition.slab_partition_stokes3d(n, ref, world)
    levels = partition.partitioned_geometric_aggregates(full.params, plan, a=2, min_coarse=100)
    plan, out = _run_ranks(world, n, ref, cfg, levels)
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets,
                                      aggregates=[(a, nc, coff) for a, nc, coff, _ in levels])
    rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
    rc, ox, ores, ohist = osys.solve(cfg, orhs)
    assert rc == 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert (res["outer_iterations"], res["inner_iterations"], res["mp_iterations"]) == \
            (ores.outer_iterations, ores.inner_iterations, ores.mp_iterations)
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    xs = np.concatenate([out[r]["x"][0] for r in range(world)])
    assert np.allclose(xs, ox[0], rtol=1e-9, atol=1e-10 * np.abs(ox[0]).max())
    # fewer inner iterations than the single-level sweep on the same partition
    cfg2 = _abi.default_config(_abi.AL_STOKES)
    cfg2.inner.max_steps = 1000
    _, out2 = _run_ranks(world, n, ref, cfg2)
    assert out[0]["res"]["inner_iterations"] < out2[0]["res"]["inner_iterations"]


def test_partitioned_exact_w_inverse_matches_oracle_emulation(built):
    """`Diagonal mass immersed = false` on the row-partitioned path: the nested CG on the
    (row-partitioned, halo-exchanging) immersed mass matrix inside the operator of the inner CG,
    reductions in rank order -- same counts as the oracle's emulation of the partition."""
    world, n, ref = 2, 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    cfg.w_inverse = _abi.W_MASS_INV_SQUARED
    plan, out = _run_ranks(world, n, ref, cfg)
    full = problems.stokes3d_sphere(n, ref)
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets)
    rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
    assert rc == 0
    rc, ox, ores, ohist = osys.solve(cfg, orhs)
    assert rc == 0 and ores.mass_iterations > 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert (res["outer_iterations"], res["inner_iterations"], res["mp_iterations"], res["mass_iterations"]) == \
            (ores.outer_iterations, ores.inner_iterations, ores.mp_iterations, ores.mass_iterations)
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    for b in range(3):
        assert np.array_equal(np.concatenate([out[r]["rhs"][b] for r in range(world)]), orhs[b])


def test_replicated_coarse_levels_change_nothing_but_the_exchanges(built, monkeypatch):
    """Multi-rank multigrid: the levels below ALFD_ML_REPLICATE unknowns (default 300 k) are replicated on
    every rank after the partitioned build (one all-gather of the restricted residual per V-cycle instead
    of ~40 neighbour exchanges).  Row sums do not depend on the partition and the Chebyshev levels have no
    reductions, so the residual history must be bit-identical to the fully partitioned hierarchy."""
    world, n, ref = 3, 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    cfg.inner_prec = _abi.PREC_MULTILEVEL
    cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 2, 8.0
    full = problems.stokes3d_sphere(n, ref)
    plan = partition.slab_partition_stokes3d(n, ref, world)
    levels = partition.partitioned_geometric_aggregates(full.params, plan, a=2, min_coarse=100)
    _, rep = _run_ranks(world, n, ref, cfg, levels)                  # default: replicated below 300 k
    monkeypatch.setenv("ALFD_ML_REPLICATE", "0")
    _, part = _run_ranks(world, n, ref, cfg, levels)                 # fully partitioned hierarchy
    for r in range(world):
        assert rep[r]["res"]["inner_iterations"] == part[r]["res"]["inner_iterations"]
        assert np.array_equal(rep[r]["hist"], part[r]["hist"])
        for b in range(3):
            assert np.array_equal(rep[r]["x"][b], part[r]["x"][b])


def test_rank_without_multiplier_rows(built):
    """A geometric partition of a localised immersed body leaves some ranks without any multiplier
    row (alfd_set_partition accepts empty ranges).  Such a rank launches nothing on its empty block
    but still enters every collective; counts and history equal the oracle's emulation."""
    world, n, ref = 3, 8, 0
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    plan = partition.slab_partition_stokes3d(n, ref, world)
    nl = int(plan.offsets[-1][-1])
    plan.offsets[-1] = np.array([0, 0, nl // 2, nl], np.int64)      # rank 0 owns no multiplier row
    assert plan.local_sizes(0)[-1] == 0
    plan, out = _run_ranks(world, n, ref, cfg, plan=plan)
    full = problems.stokes3d_sphere(n, ref)
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets)
    rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
    rc, ox, ores, ohist = osys.solve(cfg, orhs)
    assert rc == 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert (res["outer_iterations"], res["inner_iterations"], res["mp_iterations"]) == \
            (ores.outer_iterations, ores.inner_iterations, ores.mp_iterations)
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    assert out[0]["x"][2].size == 0
    for b in range(3):
        xs = np.concatenate([out[r]["x"][b] for r in range(world)])
        assert np.allclose(xs, ox[b], rtol=1e-9, atol=1e-10 * max(np.abs(ox[b]).max(), 1e-30))


@pytest.mark.parametrize("world,overlap", [(2, None), (3, None), (2, "0")])
def test_partitioned_batch_major_spmv_bitwise(built, world, overlap, monkeypatch):
    """The A-SpMV of a partition large enough for the LDS-window / batch-major formats (>= 256 row blocks per
    rank; the solves above are too small for them): halo columns inside the staged x windows, mesh-brick row
    blocks per rank.  Every rank's rows must equal the unpartitioned product bit for bit -- with the halo exchange
    running beside the interior row blocks (the default on this transport) and with ALFD_SPMV_OVERLAP_HALO=0."""
    if overlap is not None:
        monkeypatch.setenv("ALFD_SPMV_OVERLAP_HALO", overlap)     # read when a context is created
    n, ref = 28 + 8 * (world - 2), 0
    plan = partition.slab_partition_stokes3d(n, ref, world)
    full = problems.stokes3d_sphere(n, ref)
    a = full.mats["A"]
    x = np.random.default_rng(3).uniform(-1, 1, a.ncols)
    want, _ = oracle.spmv(a, x, None, mode=0)
    group = solver.LocalGroup(world)
    out, fmt, errs = [None] * world, [None] * world, []
    uoff = plan.offsets[0]

    def work(rank):
        try:
            pb = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
            ctx = solver.Context(0)
            ctx.comm_init_local(group.handle, rank)
            ctx.set_partition(plan.offsets)
            ctx.set_row_blocks(_abi.A, *problems.brick_row_blocks(
                pb.params, (16, 4, 1), node_range=(plan.node_offsets_u[rank], plan.node_offsets_u[rank + 1])))
            ctx.set_matrix(_abi.A, pb.mats["A"])
            fmt[rank] = ctx.matrix_info(_abi.A)
            xl = x[uoff[rank]:uoff[rank + 1]]
            out[rank], _ = ctx.spmv(_abi.A, xl, np.zeros(pb.mats["A"].nrows), mode=0)
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    group.close()
    assert not errs, errs
    assert all(f["windowed"] and f["batch_major"] == 2 for f in fmt), fmt
    # the row blocks without a halo column lead the plan: they run while the exchange is in flight
    assert all(0 < f["batch_major_interior_blocks"] < f["batch_major_blocks"] for f in fmt), fmt
    assert np.array_equal(np.concatenate(out), want)


def _run_interface_ranks(world, plan, make_local, cfg):
    group = solver.LocalGroup(world)
    out = [None] * world
    errs = []

    def work(rank):
        try:
            pb = make_local(plan.generator_ranges(rank))
            ctx = solver.Context(0)
            ctx.comm_init_local(group.handle, rank)
            ctx.set_partition(plan.offsets)
            solver.upload_problem(ctx, pb, cfg)
            x, res = ctx.solve(cases.rhs_of(pb))
            out[rank] = dict(x=x, res=res.as_dict(), hist=ctx.history())
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    group.close()
    assert not errs, errs
    return out


@pytest.mark.parametrize("name,world", [("elliptic_modified", 2), ("elliptic_ideal", 3), ("elliptic_modified_exact_w", 2),
                                        ("elasticity_modified", 3)])
def test_partitioned_interface_problems_match_oracle_emulation(built, name, world):
    """The elliptic-interface / elasticity systems (BASELINE cfg 3 and 5) row-partitioned: background slabs, the
    immersed block and the multiplier block split with the SAME offsets (M maps between them), the 2x2 block CG of the
    ideal variant with its dots over [u | u2], the nested mass solves of the exact W^-1 -- counts equal to the oracle's
    emulation of the partition, history within 1e-10, stitched solution equal."""
    full, cfg = cases.case(name)
    P = full.params
    n_fg_nodes = full.block_sizes[1] // P["ncomp"]
    plan = partition.slab_partition_interface(P["dim"], P["n_cells"], n_fg_nodes, world, ncomp=P["ncomp"])
    if name.startswith("elasticity"):
        make_local = lambda rr: problems.elasticity3d(P["n_cells"], row_ranges=rr)
    else:
        n_fg = int(round(np.sqrt(n_fg_nodes))) - 1
        beta2 = 10.0
        make_local = lambda rr: problems.elliptic_interface2d(P["n_cells"], n_fg, beta2=beta2, row_ranges=rr)
    out = _run_interface_ranks(world, plan, make_local, cfg)
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets)
    rc, ox, ores, ohist = osys.solve(cfg, cases.rhs_of(full))
    assert rc == 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert (res["outer_iterations"], res["inner_iterations"], res["mass_iterations"]) == \
            (ores.outer_iterations, ores.inner_iterations, ores.mass_iterations)
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    for b in range(3):
        xs = np.concatenate([out[r]["x"][b] for r in range(world)])
        assert np.allclose(xs, ox[b], rtol=1e-8, atol=1e-9 * max(np.abs(ox[b]).max(), 1e-30))


@pytest.mark.parametrize("world,patch", [(2, True), (3, True), (2, False), (4, True)])
def test_partitioned_geometric_multigrid_matches_oracle_emulation(built, world, patch):
    """The round-3 inner preconditioner on a row-partitioned context: CSR prolongators (level 0: each rank's rows, the
    levels below whole), fine level partitioned, the coarse hierarchy and the interface patch REPLICATED and built
    partition-independently (remote prolongator / A P rows fetched from their owners, coarse rows formed in global
    order).  Must equal the oracle's emulation of the partition -- which differs from a single-rank run only in the
    rank-ordered level-0 reductions -- in every count and to 1e-10 in the history."""
    n, ref = 8, 1
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner_prec = _abi.PREC_MULTILEVEL
    cfg.inner.max_steps = 100
    cfg.ml_smooth_degree, cfg.ml_smooth_degree_coarse, cfg.ml_smooth_ratio = 3, 4, 30.0
    cfg.ml_coarse_direct = 1024
    if patch:
        cfg.ml_patch_degree, cfg.ml_patch_ratio = 6, 40.0
    full = problems.stokes3d_sphere(n, ref)
    glevels = problems.tensor_prolongators(full.params, min_coarse=100)
    plan = partition.slab_partition_stokes3d(n, ref, world)
    group = solver.LocalGroup(world)
    out, errs = [None] * world, []

    def work(rank):
        try:
            pb = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
            ctx = solver.Context(0)
            ctx.comm_init_local(group.handle, rank)
            ctx.set_partition(plan.offsets)
            solver.upload_problem(ctx, pb, cfg, partition.local_prolongators(glevels, full.params, plan, rank))
            rhs = ctx.augment_rhs(cases.rhs_of(pb))
            x, res = ctx.solve(rhs)
            out[rank] = dict(x=x, res=res.as_dict(), hist=ctx.history())
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=600)
    group.close()
    assert not errs, errs
    osys = oracle.system_from_problem(full, nranks_emulated=world, part_offsets=plan.offsets, aggregates=glevels)
    rc, orhs = osys.augment_rhs(cfg, cases.rhs_of(full))
    rc, ox, ores, ohist = osys.solve(cfg, orhs)
    assert rc == 0
    for r in range(world):
        res = out[r]["res"]
        assert res["status"] == 0
        assert (res["outer_iterations"], res["inner_iterations"], res["mp_iterations"]) == \
            (ores.outer_iterations, ores.inner_iterations, ores.mp_iterations)
        assert np.array_equal(out[r]["hist"], out[0]["hist"])
        assert np.max(np.abs(out[r]["hist"] - ohist) / np.abs(ohist)) <= 1e-10
    for b in range(3):
        xs = np.concatenate([out[r]["x"][b] for r in range(world)])
        assert np.allclose(xs, ox[b], rtol=1e-9, atol=1e-10 * max(np.abs(ox[b]).max(), 1e-30))


def test_partitioned_geometric_bench_like_solve_matches_single_rank(built):
    """bench.py's settings on two ranks at a size where the batch-major / window formats carry halo columns (N = 28,
    mesh bricks per rank): the partition-independent hierarchy gives the SAME iteration counts as the single-rank
    solve, histories equal up to the differently associated level-0 reductions."""
    n, ref, world = 28, 1, 2
    cfg = _abi.bench_multilevel_settings(_abi.default_config(_abi.AL_STOKES), geometric=True)
    full = problems.stokes3d_sphere(n, ref)
    glevels = problems.tensor_prolongators(full.params, min_coarse=_abi.BENCH_MIN_COARSE)
    plan = partition.slab_partition_stokes3d(n, ref, world)
    group = solver.LocalGroup(world)
    out, errs = [None] * world, []

    def work(rank):
        try:
            pb = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
            ctx = solver.Context(0)
            ctx.comm_init_local(group.handle, rank)
            ctx.set_partition(plan.offsets)
            rb = problems.brick_row_blocks(pb.params, (16, 4, 1),
                                           node_range=(int(plan.node_offsets_u[rank]), int(plan.node_offsets_u[rank + 1])))
            solver.upload_problem(ctx, pb, cfg, partition.local_prolongators(glevels, full.params, plan, rank), rb)
            x, res = ctx.solve(ctx.augment_rhs(cases.rhs_of(pb)))
            out[rank] = dict(x=x, res=res.as_dict(), hist=ctx.history(), info=ctx.matrix_info(_abi.A))
            ctx.close()
        except Exception as e:   # noqa: BLE001
            errs.append((rank, repr(e)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=900)
    group.close()
    assert not errs, errs
    ctx = solver.Context(0)
    ctx.set_row_blocks(_abi.A, *problems.brick_row_blocks(full.params, (16, 4, 1)))
    solver.upload_problem(ctx, full, cfg, glevels)
    x, res = ctx.solve(ctx.augment_rhs(cases.rhs_of(full)))
    h1 = ctx.history()
    ctx.close()
    for r in range(world):
        assert out[r]["info"]["batch_major"] == 2
        assert out[r]["res"]["status"] == 0
        assert (out[r]["res"]["outer_iterations"], out[r]["res"]["inner_iterations"]) == (res.outer_iterations, res.inner_iterations)
        assert np.allclose(out[r]["hist"], h1, rtol=1e-6)
    xs = np.concatenate([out[r]["x"][0] for r in range(world)])
    assert np.allclose(xs, x[0], rtol=1e-6, atol=1e-8 * np.abs(x[0]).max())
