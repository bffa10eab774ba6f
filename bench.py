#!/usr/bin/env python3
"""bench.py -- AL-preconditioned FGMRES throughput on MI355X.

One "step" = one full FGMRES solve (to the prm's stop rule) of the synthetic
3-D Stokes-immersed system of BASELINE.json configs[3] (N = 74^3 cells: 10.35 M DoF >= 1e7)
(stokes_immersed_boundary + parameters_stokes_3d.prm),
with every operator and vector already resident in HBM when the timed region
starts.  value = outer FGMRES iterations per second over the K timed solves.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n-cells 74]

N > 1: launched by torch.distributed.run, one rank per GPU; the SAME global
problem is row-partitioned over the ranks ("scaling": "strong"), Krylov inner
products go through RCCL all-gather + ordered sum, SpMV halos through RCCL
send/recv.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-cells", type=int, default=int(os.environ.get("ALFD_BENCH_NCELLS", "74")))
    ap.add_argument("--immersed-refine", type=int, default=-1)
    ap.add_argument("--cheb-degree", type=int, default=4)
    ap.add_argument("--inner-max", type=int, default=100,
                    help="cap of the inner CG: 100 = parameters_stokes_3d.prm:23 (the reference throws beyond it)")
    ap.add_argument("--bricks", default="16,4,1",
                    help="row blocks of the A-SpMV: nodes of an a x b x c patch of the velocity grid (0 = runs of the numbering)")
    ap.add_argument("--general-steps", type=int, default=1,
                    help="extra timed solves with the dictionary-free 10 B/nnz SpMV kernel (0 = skip)")
    ap.add_argument("--inner-prec", choices=["chebyshev", "multilevel"],
                    default=os.environ.get("ALFD_BENCH_PREC", "multilevel"))
    ap.add_argument("--hierarchy", choices=["geometric", "aggregation"], default=os.environ.get("ALFD_BENCH_HIERARCHY", "geometric"),
                    help="multigrid transfers: CSR prolongators (Q2 -> Q1 embedding, then trilinear interpolation; "
                         "alfd_set_prolongator) or piecewise-constant aggregates (round 2; the only one on several ranks)")
    ap.add_argument("--ml-smooth-degree", type=int, default=None, help="default 3 (geometric) / 4 (aggregation)")
    ap.add_argument("--ml-smooth-degree-coarse", type=int, default=None,
                    help="smoother degree on levels >= 1 (default 5 with the geometric hierarchy, else the fine one)")
    ap.add_argument("--ml-smooth-ratio", type=float, default=None, help="default 30 (geometric) / 256 (aggregation)")
    ap.add_argument("--ml-coarse-degree", type=int, default=10)
    ap.add_argument("--patch-degree", type=int, default=20, help="interface-patch Chebyshev degree (geometric hierarchy; 0 = off)")
    ap.add_argument("--patch-ratio", type=float, default=400.0)
    ap.add_argument("--coarse-direct", type=int, default=1024, help="explicit coarsest inverse up to this many unknowns (geometric hierarchy)")
    ap.add_argument("--agg-a", type=int, default=2, help="nodes per aggregate edge (geometric aggregation)")
    ap.add_argument("--min-coarse", type=int, default=4000, help="stop coarsening below this many unknowns")
    ap.add_argument("--comm", choices=["rccl", "host"], default=os.environ.get("ALFD_BENCH_COMM", "rccl"),
                    help="multi-GPU transport: RCCL over xGMI (default) or host buffers through a gloo group "
                         "(alfd_comm_init_host; slower, for boxes where RCCL cannot start)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-only-spmv", type=int, default=0,
                    help="skip the solve; run this many back-to-back A SpMV launches (for rocprofv3)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    import numpy as np
    import torch
    import torch.distributed as dist

    from fictitious_domain_al_preconditioners_amd import _abi, partition, problems, solver

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the solver has no CPU path")
    if os.environ.get("ALFD_BENCH_SINGLE_DEVICE") == "1":
        local_rank = 0          # rehearsal of the N > 1 code path on a one-GPU box (with --comm host): all ranks on cuda:0
    torch.cuda.set_device(local_rank)
    if world > 1:
        # control plane (unique-id broadcast, barriers, max over ranks of the timing) over gloo on CPU tensors; the data
        # path -- halo exchanges and reductions inside libalfd -- is RCCL (or the host transport with --comm host)
        dist.init_process_group("gloo")

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # ---------------------------------------------------------------- problem
    n = args.n_cells
    refine = args.immersed_refine if args.immersed_refine >= 0 else max(0, int(round(np.log2(n / 64.0))) + 4)
    t0 = time.time()
    plan = partition.slab_partition_stokes3d(n, refine, world)
    pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine, row_ranges=plan.generator_ranges(rank))
    gsizes = plan.global_sizes
    ntot = int(sum(gsizes))
    log(f"generated N={n}^3 Taylor-Hood: blocks {gsizes} ({ntot/1e6:.2f} M DoF), local nnz(A) = "
        f"{pb.mats['A'].nnz/1e9:.3f} G in {time.time()-t0:.1f} s")

    cfg = _abi.default_config(_abi.AL_STOKES)  # parameters_stokes_3d.prm:17-24,150-157
    cfg.cheb_degree = args.cheb_degree
    # inner CG cap: the reference's 100 (prm:23) holds with the multigrid inner preconditioner
    # (~32 inner iterations per outer one); the single-level Chebyshev sweep needs --inner-max 2000.
    cfg.inner.max_steps = args.inner_max
    cfg.log_level = int(os.environ.get("ALFD_BENCH_LOG_LEVEL", "0"))
    aggregates = levels = None
    geometric = args.hierarchy == "geometric" and world == 1      # CSR prolongators are single-rank for now
    if args.inner_prec == "multilevel":
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree = args.ml_smooth_degree if args.ml_smooth_degree is not None else (3 if geometric else 4)
        cfg.ml_smooth_degree_coarse = (args.ml_smooth_degree_coarse if args.ml_smooth_degree_coarse is not None
                                       else (5 if geometric else 0))
        cfg.ml_smooth_ratio = args.ml_smooth_ratio if args.ml_smooth_ratio is not None else (30.0 if geometric else 256.0)
        cfg.ml_coarse_degree = args.ml_coarse_degree
        if geometric:
            cfg.ml_patch_degree, cfg.ml_patch_ratio, cfg.ml_coarse_direct = args.patch_degree, args.patch_ratio, args.coarse_direct

    t0 = time.time()
    ctx = solver.Context(local_rank)
    if world > 1:
        if args.comm == "host":
            ctx.comm_init_torch(dist.group.WORLD)
        else:
            uid = [solver.Context.unique_id() if rank == 0 else None]
            dist.broadcast_object_list(uid, src=0)
            ctx.comm_init(rank, world, uid[0])
        ctx.set_partition(plan.offsets)
    if cfg.inner_prec == _abi.PREC_MULTILEVEL and geometric:
        ta = time.time()
        levels = aggregates = problems.tensor_prolongators(pb.params, min_coarse=min(args.min_coarse, max(args.coarse_direct, 100)))
        log(f"prolongators: levels {[lv[1] for lv in levels]} in {time.time()-ta:.1f} s")
    elif cfg.inner_prec == _abi.PREC_MULTILEVEL:
        ta = time.time()
        levels = partition.partitioned_geometric_aggregates(pb.params, plan, a=args.agg_a, min_coarse=args.min_coarse)   # slab-respecting boxes
        aggregates = partition.local_aggregates(levels, rank)
        log(f"aggregates: levels {[lv[1] for lv in levels]} in {time.time()-ta:.1f} s")
    row_blocks = None
    if args.bricks != "0":
        # row blocks of the A-SpMV = bricks of the Q2 grid inside this rank's slab (alfd_set_row_blocks):
        # a third of the x window that 96 consecutive rows of the lexicographic numbering need
        brick = tuple(int(v) for v in args.bricks.split(","))
        row_blocks = problems.brick_row_blocks(
            pb.params, brick, node_range=(int(plan.node_offsets_u[rank]), int(plan.node_offsets_u[rank + 1])))
    solver.upload_problem(ctx, pb, cfg, aggregates, row_blocks)
    rhs = ctx.augment_rhs([pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]])
    ctx.upload_rhs(rhs)
    log(f"uploaded + setup in {time.time()-t0:.1f} s")

    if args.profile_only_spmv > 0:
        ms, nbytes = ctx.bench_spmv(_abi.A, args.profile_only_spmv)
        log(f"A SpMV: {ms:.4f} ms/launch, {nbytes/1e9:.3f} GB algorithmic -> {nbytes/ms/1e6:.1f} GB/s")
        return

    # ------------------------------------------------------------------ solve
    for _ in range(args.warmup):
        res = ctx.solve_resident()
        log(f"warmup solve: outer {res.outer_iterations}, inner {res.inner_iterations}, "
            f"{res.solve_seconds:.2f} s, |r| = {res.last_residual:.3e}")
    ctx.enable_timing(True)  # HIP events around the A-SpMV launches on the solver's stream
    barrier()
    t0 = time.perf_counter()
    outer = inner = 0
    last = None
    for _ in range(args.steps):
        last = ctx.solve_resident()
        outer += last.outer_iterations
        inner += last.inner_iterations
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    tim = ctx.timing()
    ctx.enable_timing(False)
    nnz_A_global = int(pb.mats["A"].nnz)
    if world > 1:
        tn = torch.tensor([nnz_A_global], dtype=torch.int64)
        dist.all_reduce(tn)
        nnz_A_global = int(tn.item())

    info = ctx.matrix_info(_abi.A)
    spmv = tim["spmv_A"]
    avg_ms = spmv["ms"] / max(spmv["launches"], 1)
    csr_bytes = spmv["bytes"] / max(spmv["launches"], 1)      # plain-CSR model of SURVEY 8(d)
    # bytes the kernel has to move per launch in the format it reads (3 B/nnz dictionary-coded
    # stream + descriptors + x + y, or 10 B/nnz for the general kernel): the physical roofline
    fmt_bytes = info["streamed_bytes"] if info["windowed"] else csr_bytes
    achieved = fmt_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if os.path.exists(tpath):
        try:
            t = json.load(open(tpath))
            if (t.get("n_cells") == n and world == 1 and bool(t.get("value_indexed")) == bool(info["value_indexed"])
                    and int(t.get("batch_major", 0)) == int(info["batch_major"])):
                traffic = t.get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # what else is loaded besides HBM (PMC counters of the same kernel, profiles/r02/pmc_vs.json): with 0.9 B/nnz the
    # A-SpMV is bound by the vector ALU and the LDS gather rate, not by HBM -- reported next to the HBM figures
    other_limits = None
    ppath = os.path.join(ROOT, "profiles", "r02", "pmc_vs.json")
    if traffic and os.path.exists(ppath) and avg_ms > 0:
        try:
            pm = json.load(open(ppath))["median"]
            cycles = avg_ms * 1e-3 * 2.4e9                      # 2.4 GHz engine clock
            other_limits = {
                "source": "profiles/r02/pmc_vs.json (rocprofv3 --pmc, same kernel and matrix)",
                "valu_issue_frac": pm["SQ_INSTS_VALU"] * 4.0 / (1024 * cycles),      # 4 cycles per wave64 VALU op, 1024 SIMDs
                "lds_busy_frac": pm["SQ_LDS_IDX_ACTIVE"] / (256 * cycles),           # LDS pipe cycles per CU
                "lds_gather_GBps": (info["nnz"] * 8.0 * 1.15) / (avg_ms * 1e-3) / 1e9,  # 8-byte window gather per entry + dictionary gathers
                "lds_peak_GBps": 256 * 128 * 2.4,                                     # 128 B/clk/CU
                "vmem_load_instructions": pm["SQ_INSTS_VMEM_RD"], "valu_instructions": pm["SQ_INSTS_VALU"],
            }
        except Exception:
            other_limits = None
    # ---- the same solve with the dictionary-free kernel (what a matrix with unrelated values gets)
    general = None
    if world == 1 and info["value_indexed"] and args.general_steps > 0:
        ctx.set_tunable("value_index", 0)
        ctx.solve_resident()                                   # warm-up of this leg
        ctx.enable_timing(True)
        barrier()
        g0 = time.perf_counter()
        g_outer = 0
        for _ in range(args.general_steps):
            g_outer += ctx.solve_resident().outer_iterations
        barrier()
        gdt = time.perf_counter() - g0
        gt = ctx.timing()["spmv_A"]
        ctx.enable_timing(False)
        ctx.set_tunable("value_index", 1)
        g_ms = gt["ms"] / max(gt["launches"], 1)
        g_bytes = ctx.bench_spmv_format(_abi.A, 2, value_index=False)[1]
        general = {"kernel": "spmv_window_kernel<2,8,0,0> (8-byte values + 16-bit window columns, 10 B/nnz)",
                   "steps": args.general_steps, "ms_per_step": 1e3 * gdt / args.general_steps,
                   "value": g_outer / gdt, "unit": "iterations/s", "avg_launch_ms": g_ms,
                   "bytes_per_launch": g_bytes, "achieved": g_bytes / (g_ms * 1e-3) / 1e9,
                   "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}

    # ---- and with translate sharing off: every row stored (3.1 B/nnz), what a matrix whose values repeat but whose
    # rows are no translates of one another (unstructured mesh) gets from the batch-major kernel
    unshared = None
    if world == 1 and info["batch_major"] and args.general_steps > 0:
        ctx.set_tunable("batch_major_share", 0)
        solver.upload_problem(ctx, pb, cfg, aggregates, row_blocks)
        ctx.upload_rhs(rhs)
        ctx.solve_resident()
        ctx.enable_timing(True)
        barrier()
        u0 = time.perf_counter()
        ures = ctx.solve_resident()
        barrier()
        udt = time.perf_counter() - u0
        ut = ctx.timing()["spmv_A"]
        ctx.enable_timing(False)
        uinfo = ctx.matrix_info(_abi.A)
        u_ms = ut["ms"] / max(ut["launches"], 1)
        unshared = {"what": "batch-major kernel with every row stored (no translate sharing)", "ms_per_step": udt * 1e3,
                    "value": ures.outer_iterations / udt, "unit": "iterations/s", "avg_launch_ms": u_ms,
                    "bytes_per_launch": uinfo["streamed_bytes"], "bytes_per_nnz": uinfo["streamed_bytes"] / max(uinfo["nnz"], 1),
                    "achieved": uinfo["streamed_bytes"] / (u_ms * 1e-3) / 1e9 if u_ms > 0 else 0.0,
                    "frac": uinfo["streamed_bytes"] / (u_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if u_ms > 0 else 0.0}
        ctx.set_tunable("batch_major_share", 1)
    out = {
        "metric": "FGMRES iterations/sec to 1e-8 residual, 3D Stokes-immersed (AL-preconditioned)",
        "value": outer / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / max(args.steps, 1),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"stokes_immersed_boundary 3D Taylor-Hood Q2/Q1 N={n}^3 + cubed-sphere R=0.1 refine "
                        f"{refine}, IBStokesAL, parameters_stokes_3d.prm solver settings",
            "dofs": ntot, "blocks": [int(g) for g in gsizes],
            "nnz_A": nnz_A_global,
            "outer_iterations_per_solve": outer / max(args.steps, 1),
            "inner_iterations_per_solve": inner / max(args.steps, 1),
            "dof_iterations_per_s": ntot * outer / dt,
            "final_residual": last.last_residual, "initial_residual": last.initial_residual,
            "inner_prec": (f"chebyshev({cfg.cheb_degree})-jacobi" if cfg.inner_prec == _abi.PREC_CHEBYSHEV else
                           f"geometric multigrid (Q2->Q1 embedding + trilinear prolongators, Galerkin), V-cycle with "
                           f"chebyshev({cfg.ml_smooth_degree})/{cfg.ml_smooth_ratio:g} smoothing (degree {cfg.ml_smooth_degree_coarse or cfg.ml_smooth_degree} below the fine level), interface-patch "
                           f"chebyshev({cfg.ml_patch_degree})/{cfg.ml_patch_ratio:g} corrections, "
                           f"{'explicit inverse' if cfg.ml_coarse_direct >= levels[-1][1] else f'chebyshev({cfg.ml_coarse_degree})'} "
                           f"on the coarsest level, levels {[lv[1] for lv in levels]}" if geometric else
                           f"aggregation-multigrid V-cycle, chebyshev({cfg.ml_smooth_degree}) smoothing, "
                           f"chebyshev({cfg.ml_coarse_degree}) coarsest solve, "
                           f"levels {[lv[1] for lv in levels]}"),
            "inner_max_steps": cfg.inner.max_steps,
            "restart": cfg.restart, "partition": f"row-slabs x{world}",
            "transport": ("rccl" if args.comm == "rccl" else "host buffers over gloo") if world > 1 else None,
        },
        "roofline": {
            "bound": "hbm",
            "kernel": ("spmv_vs_kernel<0,0,4> (A, batch-major dictionary-coded stream, "
                       + ("mesh-brick row blocks " + args.bricks.replace(",", "x") if info["batch_major"] == 2 else "row runs") + ")"
                       if info["batch_major"] else
                       "spmv_window_vib_kernel<0,0,4> (A, LDS-windowed CSR, dictionary-coded values, class-batched rows)"
                       if info["value_indexed"] else "spmv_window_kernel<2,8,0,0> (A, LDS-windowed CSR)"),
            # achieved = bytes one launch has to move in the storage format the kernel reads (DESIGN.md
            # section 5: what a perfect cache would still fetch) / HIP-event launch time; frac <= 1 by
            # construction.  csr_equivalent_* restates the same time against the plain-CSR byte model of
            # SURVEY 8(d) (12 B/nnz): > peak when the format is smaller than CSR, a speed-up, not a bandwidth.
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
            "avg_launch_ms": avg_ms, "bytes_per_launch": fmt_bytes,
            "bytes_per_nnz": fmt_bytes / max(info["nnz"], 1),
            "csr_equivalent_bytes_per_launch": csr_bytes,
            "csr_equivalent_GBps": csr_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
            "value_indexed_nnz_share": info["value_indexed_nnz"] / max(info["nnz"], 1),
            "launches": spmv["launches"],
            "time_share_spmv_A": spmv["ms"] * 1e-3 / dt,
            "other_limits": other_limits,
        },
        # the whole solve again with the general-matrix SpMV kernel (no value dictionary anywhere):
        # the figure a matrix WITHOUT repeating entry values would get
        "general_matrix_leg": general,
        "no_translate_sharing_leg": unshared,
    }

    # ----------------------------------------------------------- CPU baseline
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(pb, cfg, rhs, [(lv[0], lv[1]) for lv in levels] if levels else None,
                                           outer / max(args.steps, 1), inner / max(args.steps, 1))
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


def host_cpu_share():
    """CPUs this process may really use: the cgroup quota / cpuset of the container, not the
    host's core count (a GPU box hands one GPU's share of a 256-thread host to the job)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max":
            n = min(n, max(1, int(float(q) / float(per))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def cpu_baseline(pb, cfg, rhs, aggregates, outer_its, inner_its):
    """The oracle (CPU port of the same algorithm and inner preconditioner, sequential row sums
    like deal.II's vmult) timed on the host cores of this box on the SAME full-size operators.
    A complete CPU solve would take minutes, so the sample is bounded:
      * all cores: the oracle's FGMRES run for its first TWO complete outer iterations (inner CG
        to its real stop rule, system operator, orthogonalisation): seconds per inner iteration,
        everything else included;
      * 1 thread (the reference is single-threaded, MPI_InitFinalize(argc, argv, 1)): one
        preconditioner application cut to two inner iterations.
    GPU and oracle perform the SAME iterations (counts are bit-identical, tests/), so
    value = outer iterations of the solve / (seconds per inner iteration x its inner iterations)."""
    from fictitious_domain_al_preconditioners_amd import _abi
    from oracle import oracle

    share = host_cpu_share()
    if share == (os.cpu_count() or 1) and share > 32:
        share = 16      # no quota visible on a big shared host: one GPU's share of the box is 16 CPUs
    want = int(os.environ.get("ALFD_CPU_THREADS", "0")) or share
    cores = oracle.set_threads(want)
    oracle.set_row_order(1)  # plain sequential row sums, as deal.II's vmult does
    osys = oracle.system_from_problem(pb, aggregates=aggregates)
    c = _abi.Config.from_buffer_copy(cfg)
    c.outer.max_steps = 2                  # stop after two complete outer iterations
    t0 = time.time()
    rc, _, res, _ = osys.solve(c, rhs)
    t_total = time.time() - t0
    if rc not in (0, _abi.E_NO_CONVERGENCE_OUTER) or res.inner_iterations < 1:
        raise RuntimeError(f"cpu_baseline: oracle run failed (rc={rc})")
    per_inner = res.solve_seconds / res.inner_iterations
    # 1 thread
    c1 = _abi.Config.from_buffer_copy(cfg)
    h = osys.open(c1)
    oracle.set_threads(1)
    t0 = time.time()
    rc1, _, r1 = osys.handle_precond_apply(h, rhs, _abi.Control(_abi.CTRL_FIXED_ITERS, 2, 0.0, 0.0))
    t_one = time.time() - t0
    oracle.set_threads(cores)
    osys.close_handle(h)
    oracle.set_row_order(0)
    if rc1 != 0 or r1.inner_iterations != 2:
        raise RuntimeError(f"cpu_baseline: 1-thread sample failed (rc={rc1})")
    per_inner_1 = t_one / 2
    return {
        "value": outer_its / (per_inner * inner_its), "unit": "iterations/s", "cores": cores,
        "host_cpu_share": share, "host_logical_cpus": os.cpu_count(), "kind": "port",
        "sample": f"the oracle's FGMRES on the full-size operators with {cores} threads, first 2 complete outer "
                  f"iterations: {res.inner_iterations} inner + {res.mp_iterations} pressure-mass CG iterations in "
                  f"{res.solve_seconds:.1f} s (setup {t_total - res.solve_seconds:.1f} s untimed) = {per_inner:.2f} s per "
                  f"inner iteration, all overheads included; projected to the solve's {outer_its:g} outer / "
                  f"{inner_its:g} inner iterations (identical counts on GPU and oracle)",
        "seconds_per_inner_iteration": per_inner,
        "one_thread": {
            "value": outer_its / (per_inner_1 * inner_its), "unit": "iterations/s", "cores": 1,
            "sample": f"one preconditioner application cut to 2 inner iterations on 1 thread: {t_one:.1f} s = "
                      f"{per_inner_1:.2f} s per inner iteration, projected the same way; the reference itself runs "
                      f"on 1 thread (MPI_InitFinalize(argc, argv, 1))",
            "seconds_per_inner_iteration": per_inner_1,
        },
    }


if __name__ == "__main__":
    main()
