#!/usr/bin/env python3
"""bench.py -- AL-preconditioned FGMRES throughput on MI355X.

One "step" = one full FGMRES solve (to the prm's stop rule) of the synthetic
3-D Stokes-immersed system of BASELINE.json configs[3] (N = 74^3 cells: 10.35 M DoF >= 1e7)
(stokes_immersed_boundary + parameters_stokes_3d.prm),
with every operator and vector already resident in HBM when the timed region
starts.  value = outer FGMRES iterations per second over the K timed solves.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--n-cells 74]

N > 1: one rank per GPU.  Launched by torch.distributed.run (the driver's contract), or directly --
bench.py then starts its own N ranks as child processes before anything touches a GPU.  The SAME global
problem is row-partitioned over the ranks ("scaling": "strong"), Krylov inner products go through RCCL
all-gather + ordered sum, SpMV halos through RCCL send/recv.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def self_launch(n_gpus):
    """`python bench.py --gpus N` without a launcher: start N ranks as CHILD processes of
    torch.distributed.run (never a re-exec: this process has not touched the GPU and does not afterwards),
    relay rank 0's JSON line, exit non-zero if any rank failed."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    log(f"--gpus {n_gpus} without a launcher: starting {n_gpus} ranks through torch.distributed.run on port {port}")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        out = out.rstrip("\n")
        if out.startswith("{") and '"metric"' in out:
            line = out
        elif out:
            print(out, file=sys.stderr, flush=True)
    rc = proc.wait()
    if line:
        print(line, flush=True)
    raise SystemExit(rc if rc != 0 else (0 if line else 1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--n-cells", type=int, default=int(os.environ.get("ALFD_BENCH_NCELLS", "74")))
    ap.add_argument("--immersed-refine", type=int, default=-1)
    ap.add_argument("--cheb-degree", type=int, default=4)
    ap.add_argument("--inner-max", type=int, default=100,
                    help="cap of the inner CG: 100 = parameters_stokes_3d.prm:23 (the reference throws beyond it)")
    ap.add_argument("--bricks", default="16,4,1",
                    help="row blocks of the A-SpMV: nodes of an a x b x c patch of the velocity grid (0 = runs of the numbering)")
    ap.add_argument("--general-steps", type=int, default=1,
                    help="extra timed solves with the dictionary-free 10 B/nnz SpMV kernel and with translate sharing off (0 = skip)")
    ap.add_argument("--inner-prec", choices=["chebyshev", "multilevel"],
                    default=os.environ.get("ALFD_BENCH_PREC", "multilevel"))
    ap.add_argument("--hierarchy", choices=["geometric", "aggregation"], default=os.environ.get("ALFD_BENCH_HIERARCHY", "geometric"),
                    help="multigrid transfers: CSR prolongators (Q2 -> Q1 embedding, then trilinear interpolation; "
                         "alfd_set_prolongator) or piecewise-constant aggregates (round 2)")
    ap.add_argument("--ml-smooth-degree", type=int, default=None, help="default 3 (geometric) / 4 (aggregation)")
    ap.add_argument("--ml-smooth-degree-coarse", type=int, default=None,
                    help="smoother degree on levels >= 1 (default 5 with the geometric hierarchy, else the fine one)")
    ap.add_argument("--ml-smooth-ratio", type=float, default=None, help="default 40 (geometric) / 256 (aggregation)")
    ap.add_argument("--ml-coarse-degree", type=int, default=None)
    ap.add_argument("--patch-degree", type=int, default=None, help="interface-patch Chebyshev degree (geometric hierarchy; 0 = off; default 15)")
    ap.add_argument("--patch-ratio", type=float, default=None, help="default 200")
    ap.add_argument("--coarse-direct", type=int, default=None, help="explicit coarsest inverse up to this many unknowns (default 1024)")
    ap.add_argument("--agg-a", type=int, default=2, help="nodes per aggregate edge (aggregation hierarchy)")
    ap.add_argument("--min-coarse", type=int, default=None,
                    help="stop coarsening below this many unknowns (default 1024 geometric / 4000 aggregation)")
    ap.add_argument("--comm", choices=["rccl", "host"], default=os.environ.get("ALFD_BENCH_COMM", "rccl"),
                    help="multi-GPU transport: RCCL over xGMI (default) or host buffers through a gloo group "
                         "(alfd_comm_init_host; slower, for boxes where RCCL cannot start)")
    ap.add_argument("--reference-shaped-n", type=int, default=int(os.environ.get("ALFD_BENCH_REFSHAPE_N", "64")),
                    help="cells per direction of the reference-shaped leg (cell-wise assembled block (0,0) in a Cuthill-McKee "
                         "numbering, as handed over and after the front end's renumbering); 0 = skip")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--profile-only-spmv", type=int, default=0,
                    help="skip the solve; run this many back-to-back A SpMV launches (for rocprofv3)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args.gpus)          # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

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
    setup = {}
    t0 = time.time()
    plan = partition.slab_partition_stokes3d(n, refine, world)
    pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine, row_ranges=plan.generator_ranges(rank))
    gsizes = plan.global_sizes
    ntot = int(sum(gsizes))
    setup["generate_operators_s"] = time.time() - t0        # FE assembly in the reference: outside its "Solve system" timer
    log(f"generated N={n}^3 Taylor-Hood: blocks {gsizes} ({ntot/1e6:.2f} M DoF), local nnz(A) = "
        f"{pb.mats['A'].nnz/1e9:.3f} G in {setup['generate_operators_s']:.1f} s")

    cfg = _abi.default_config(_abi.AL_STOKES)  # parameters_stokes_3d.prm:17-24,150-157
    cfg.cheb_degree = args.cheb_degree
    cfg.log_level = int(os.environ.get("ALFD_BENCH_LOG_LEVEL", "0"))
    aggregates = levels = None
    geometric = args.hierarchy == "geometric"      # several ranks: fine level partitioned, coarse levels + patch replicated
    if args.inner_prec == "multilevel":
        _abi.bench_multilevel_settings(cfg, geometric)
        for field, val in (("ml_smooth_degree", args.ml_smooth_degree), ("ml_smooth_degree_coarse", args.ml_smooth_degree_coarse),
                           ("ml_smooth_ratio", args.ml_smooth_ratio), ("ml_coarse_degree", args.ml_coarse_degree),
                           ("ml_patch_degree", args.patch_degree if geometric else None),
                           ("ml_patch_ratio", args.patch_ratio if geometric else None),
                           ("ml_coarse_direct", args.coarse_direct if geometric else None)):
            if val is not None:
                setattr(cfg, field, val)
    # inner CG cap: the reference's 100 (prm:23) holds with the multigrid inner preconditioners;
    # the single-level Chebyshev sweep needs --inner-max 2000.
    cfg.inner.max_steps = args.inner_max
    min_coarse = args.min_coarse if args.min_coarse is not None else (_abi.BENCH_MIN_COARSE if geometric else 4000)

    rccl_fallback = False
    t0 = time.time()
    ctx = solver.Context(local_rank)
    if world > 1:
        if args.comm == "host":
            ctx.comm_init_torch(dist.group.WORLD)
        else:
            # RCCL inside the library.  If it cannot be brought up on ANY rank (the library links the system's librccl, torch
            # ships its own copy; this path has not run on hardware yet), every rank falls back to the host transport
            # rather than leaving the line empty; config.transport says which one ran.
            ok = True
            try:
                uid = [solver.Context.unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                ctx.comm_init(rank, world, uid[0])
            except Exception as e:   # noqa: BLE001
                ok = False
                log(f"RCCL initialisation failed on rank {rank}: {e!r}")
            flags = [None] * world
            dist.all_gather_object(flags, ok)
            if not all(flags):
                ctx.close()
                ctx = solver.Context(local_rank)
                ctx.comm_init_torch(dist.group.WORLD)
                args.comm = "host"
                rccl_fallback = True
        ctx.set_partition(plan.offsets)
    ta = time.time()
    overlapped = {}
    if cfg.inner_prec == _abi.PREC_MULTILEVEL and geometric:
        # the transfer operators are built while the library plans and uploads the operators (solver.upload_problem takes
        # a callable): a deal.II caller has them from MGTransfer before the solve starts
        def build_transfers():
            tt = time.time()
            lv = problems.tensor_prolongators(pb.params, min_coarse=min_coarse)
            out = partition.local_prolongators(lv, pb.params, plan, rank) if world > 1 else lv
            overlapped["levels"] = lv
            overlapped["seconds"] = time.time() - tt
            return out
        aggregates = build_transfers
    elif cfg.inner_prec == _abi.PREC_MULTILEVEL:
        levels = partition.partitioned_geometric_aggregates(pb.params, plan, a=args.agg_a, min_coarse=min_coarse)   # slab-respecting boxes
        aggregates = partition.local_aggregates(levels, rank)
        log(f"aggregates: levels {[lv[1] for lv in levels]} in {time.time()-ta:.1f} s")
    setup["transfers_s"] = time.time() - ta                 # a deal.II caller takes these from MGTransfer / the mesh
    row_blocks = None
    if args.bricks != "0":
        # row blocks of the A-SpMV = bricks of the Q2 grid inside this rank's slab (alfd_set_row_blocks):
        # a third of the x window that 96 consecutive rows of the lexicographic numbering need
        brick = tuple(int(v) for v in args.bricks.split(","))
        row_blocks = problems.brick_row_blocks(
            pb.params, brick, node_range=(int(plan.node_offsets_u[rank]), int(plan.node_offsets_u[rank + 1])))
    tu = time.time()
    solver.upload_problem(ctx, pb, cfg, aggregates, row_blocks)
    rhs = ctx.augment_rhs([pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]])
    ctx.upload_rhs(rhs)
    torch.cuda.synchronize()
    # what the reference's "Solve system" timer holds besides the Krylov loop (AMG setup + factorisations,
    # stokes_immersed_boundary.cc:827): format planning + upload of the operators, diag / lambda_max, the multigrid hierarchy
    setup["library_s"] = time.time() - tu
    setup["library_phases_s"] = ctx.setup_seconds()
    if overlapped:
        levels = overlapped["levels"]
        setup["transfers_overlapped_s"] = overlapped["seconds"]     # ran beside the upload: inside library_s, not added
        log(f"prolongators: levels {[lv[1] for lv in levels]} in {overlapped['seconds']:.1f} s (beside the upload)")
    setup_s = setup["library_s"] + setup["transfers_s"]
    log(f"uploaded + setup in {time.time()-t0:.1f} s (library {setup['library_s']:.1f} s: "
        + ", ".join(f"{k} {v:.1f}" for k, v in setup["library_phases_s"].items()) + ")")

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
    solve_s = dt / max(args.steps, 1)
    outer_per = outer / max(args.steps, 1)
    inner_per = inner / max(args.steps, 1)

    # ---- the whole solve against the HBM roofline: one more solve with HIP events around EVERY kernel class;
    # bytes from that run, time from the uninstrumented timed solves above
    whole = None
    if world == 1:
        ctx.enable_timing(2)
        ctx.solve_resident()
        tall = ctx.timing()
        ctx.enable_timing(False)
        alg = sum(v["bytes"] for v in tall.values())
        fmt = sum(v["format_bytes"] for v in tall.values())
        kms = sum(v["ms"] for v in tall.values())
        whole = {
            "what": "sum over all kernel launches of one solve / (solve time x 8 TB/s); algorithmic = SURVEY 8(d) byte model "
                    "(plain CSR, 12 B/nnz), format = the bytes of the storage formats the kernels read",
            "solve_s": solve_s, "algorithmic_bytes": alg, "format_bytes": fmt,
            "algorithmic_GBps": alg / solve_s / 1e9, "algorithmic_frac_of_peak": alg / solve_s / 1e9 / HBM_PEAK_GBS,
            "format_GBps": fmt / solve_s / 1e9, "format_frac_of_peak": fmt / solve_s / 1e9 / HBM_PEAK_GBS,
            "kernel_time_share_of_solve": kms * 1e-3 / solve_s,
            "classes": {k: {"launches": v["launches"], "ms": v["ms"], "algorithmic_bytes": v["bytes"],
                            "format_bytes": v["format_bytes"]} for k, v in tall.items()},
        }

    info = ctx.matrix_info(_abi.A)
    spmv = tim["spmv_A"]
    avg_ms = spmv["ms"] / max(spmv["launches"], 1)
    csr_bytes = spmv["bytes"] / max(spmv["launches"], 1)      # plain-CSR model of SURVEY 8(d)
    # bytes the kernel has to move per launch in the format it reads (dictionary-coded
    # stream + descriptors + x + y, or 10 B/nnz for the general kernel): the physical roofline
    fmt_bytes = info["streamed_bytes"] if info["windowed"] else csr_bytes
    achieved = fmt_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
    # HBM traffic and the other loaded units come from PMC passes of an EARLIER profiled run of the same kernel on the
    # same matrix (separate rocprofv3 --pmc passes cannot run inside this process); keyed on size, format and row blocks
    traffic = traffic_source = None
    other_limits = None
    pmc = load_pmc(n, info, args.bricks, world)
    if pmc and avg_ms > 0:
        traffic = pmc["hbm_bytes_per_launch"]
        traffic_source = f"from_file: {pmc['source']} (PMC passes of an earlier run of this kernel on this matrix)"
        cycles = avg_ms * 1e-3 * 2.4e9                      # 2.4 GHz engine clock
        c = pmc["counters"]
        other_limits = {
            "source": traffic_source,
            "valu_issue_frac": c["SQ_INSTS_VALU"] * 4.0 / (1024 * cycles),      # 4 cycles per wave64 VALU op, 1024 SIMDs
            "lds_busy_frac": c["SQ_LDS_IDX_ACTIVE"] / (256 * cycles),           # LDS pipe cycles per CU
            "lds_bank_conflict_share": c.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(c["SQ_LDS_IDX_ACTIVE"], 1.0),
            "vmem_load_instructions": c["SQ_INSTS_VMEM_RD"], "valu_instructions": c["SQ_INSTS_VALU"],
        }
    bound = "hbm"
    if info["batch_major"]:
        # 0.8 B/nnz: the batch-major kernel is bound by the vector ALU and the LDS gather rate together, not by HBM
        bound = "valu+lds (the kernel streams 0.8 B/nnz; frac is what is left of the HBM roofline)"

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
                   "frac": g_bytes / (g_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "bound": "hbm"}

    # ---- and with translate sharing off: every row stored (3.1 B/nnz), what a matrix whose values repeat but whose
    # rows are no translates of one another (unstructured mesh) gets from the batch-major kernel
    unshared = None
    if world == 1 and info["batch_major"] and args.general_steps > 0:
        ctx.set_tunable("batch_major_share", 0)
        ctx.set_matrix(_abi.A, pb.mats["A"])                    # A is re-planned; setup rebuilds the hierarchy
        ctx.setup(pb.block_sizes)
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

    def prec_name():
        if cfg.inner_prec == _abi.PREC_CHEBYSHEV:
            return f"chebyshev({cfg.cheb_degree})-jacobi"
        lv = [x[1] for x in levels]
        if geometric:
            coarsest = ("explicit inverse" if cfg.ml_coarse_direct >= lv[-1] else f"chebyshev({cfg.ml_coarse_degree})")
            return (f"geometric multigrid (Q2->Q1 embedding + trilinear prolongators, Galerkin), V-cycle with "
                    f"chebyshev({cfg.ml_smooth_degree})/{cfg.ml_smooth_ratio:g} smoothing (degree "
                    f"{cfg.ml_smooth_degree_coarse or cfg.ml_smooth_degree} below the fine level), interface-patch "
                    f"chebyshev({cfg.ml_patch_degree})/{cfg.ml_patch_ratio:g} corrections, {coarsest} on the coarsest level, levels {lv}")
        return (f"aggregation-multigrid V-cycle, chebyshev({cfg.ml_smooth_degree}) smoothing, "
                f"chebyshev({cfg.ml_coarse_degree}) coarsest solve, levels {lv}")

    out = {
        "metric": "FGMRES iterations/sec to 1e-8 residual, 3D Stokes-immersed (AL-preconditioned)",
        "value": outer / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * solve_s,
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
            "outer_iterations_per_solve": outer_per,
            "inner_iterations_per_solve": inner_per,
            "inner_per_outer": inner_per / max(outer_per, 1),
            "dof_iterations_per_s": ntot * outer / dt,
            "final_residual": last.last_residual, "initial_residual": last.initial_residual,
            "inner_prec": prec_name(),
            "inner_max_steps": cfg.inner.max_steps,
            "restart": cfg.restart, "partition": f"row-slabs x{world}",
            "transport": ("rccl" if args.comm == "rccl" else
                          "host buffers over gloo" + (" (RCCL could not be initialised)" if rccl_fallback else "")) if world > 1 else None,
        },
        # setup in the record: the reference solves ONCE per run and its "Solve system" timer includes AMG setup and
        # factorisations (stokes_immersed_boundary.cc:827); value above is the Krylov loop with resident operators
        "setup_s": setup_s,
        "setup": setup,
        "end_to_end": {"what": "one solve including the library setup and the transfer operators (operators generated, "
                               "nothing resident): outer iterations / (setup_s + solve_s)",
                       "seconds": setup_s + solve_s, "value": outer_per / (setup_s + solve_s), "unit": "iterations/s"},
        "roofline": {
            "bound": bound,
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
            "traffic": traffic, "traffic_source": traffic_source,
            "traffic_frac": (traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and avg_ms > 0) else None,
            "avg_launch_ms": avg_ms, "bytes_per_launch": fmt_bytes,
            "bytes_per_nnz": fmt_bytes / max(info["nnz"], 1),
            "csr_equivalent_bytes_per_launch": csr_bytes,
            "csr_equivalent_GBps": csr_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0,
            "value_indexed_nnz_share": info["value_indexed_nnz"] / max(info["nnz"], 1),
            "launches": spmv["launches"],
            "launches_per_solve": spmv["launches"] / max(args.steps, 1),
            "time_share_spmv_A": spmv["ms"] * 1e-3 / dt,
            "valu_issue_frac": other_limits["valu_issue_frac"] if other_limits else None,
            "lds_busy_frac": other_limits["lds_busy_frac"] if other_limits else None,
            "other_limits": other_limits,
            "whole_solve": whole,
        },
        # the whole solve again with the general-matrix SpMV kernel (no value dictionary anywhere):
        # the figure a matrix WITHOUT repeating entry values would get
        "general_matrix_leg": general,
        "no_translate_sharing_leg": unshared,
    }
    if world == 1 and args.reference_shaped_n > 0 and args.inner_prec == "multilevel" and geometric:
        out["reference_shaped_leg"] = reference_shaped_leg(args.reference_shaped_n, cfg, args.bricks)

    # ----------------------------------------------------------- CPU baseline + parity at the size the bench runs
    rc_exit = 0
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        base, prefix = cpu_baseline(pb, cfg, rhs, [(lv[0], lv[1]) for lv in levels] if levels else None,
                                    outer_per, inner_per)
        out["cpu_baseline"] = base
        # GPU run of the same two outer iterations (stop rules changed through alfd_set_controls, setup kept)
        full = _abi.Control(cfg.outer.kind, cfg.outer.max_steps, cfg.outer.tol, cfg.outer.reduce)
        ctx.set_controls(outer=_abi.Control(cfg.outer.kind, 2, cfg.outer.tol, cfg.outer.reduce))
        g2 = ctx.solve_resident(raise_on_failure=False)
        h2 = ctx.history()
        ctx.set_controls(outer=full)
        ho = prefix["history"]
        m = min(len(h2), len(ho))
        dev = float(np.max(np.abs(h2[:m] - ho[:m]) / np.abs(ho[:m]))) if m else float("inf")
        ok = (m == len(ho) == len(h2) and g2.inner_iterations == prefix["inner"] and g2.mp_iterations == prefix["mp"]
              and dev <= 1e-10)
        out["parity_prefix"] = {
            "what": "the first 2 outer FGMRES iterations at the bench size, GPU against the CPU oracle on the same operators: "
                    "residuals checked after every outer step, inner CG and pressure-mass CG iteration counts",
            "outer": 2, "inner": int(g2.inner_iterations), "inner_oracle": int(prefix["inner"]),
            "mp": int(g2.mp_iterations), "mp_oracle": int(prefix["mp"]),
            "history_gpu": [float(v) for v in h2], "history_oracle": [float(v) for v in ho],
            "max_rel_dev": dev, "tolerance": 1e-10, "ok": bool(ok),
        }
        if not ok:
            log(f"PARITY FAILURE at the bench size: {out['parity_prefix']}")
            rc_exit = 3
    if rank == 0:
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()
    if rc_exit:
        raise SystemExit(rc_exit)


def reference_shaped_leg(n, cfg, bricks):
    """What an operator shaped like the reference's gets.  The reference assembles cell by cell
    (stokes_immersed_boundary.cc:668-760) on hyper_cube meshes of 2^k cells per direction and numbers its DoFs with
    Cuthill-McKee, then block-wise (:533-541).  Here: block (0,0) from ONE numerically integrated cell matrix whose
    contributions are summed in Morton order of the cells (mathematically equal entries then differ in their last
    bits: 720 instead of 285 distinct values), nodes renumbered by Cuthill-McKee; solved with the bench's settings
    (a) exactly as handed over, no hint, (b) after the front end's renumbering from support points
    (alfd_host_numbering_from_points + alfd_host_brick_blocks_from_points, what dealii_adapter.hpp / solver.py do
    before the upload).  The same-size Kronecker / lexicographic operator of the headline is the yardstick."""
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import _abi, problems, solver

    refine = max(0, int(round(np.log2(n / 64.0))) + 4)
    brick = tuple(int(v) for v in bricks.split(",")) if bricks != "0" else (16, 4, 1)
    c = _abi.Config.from_buffer_copy(cfg)
    c.log_level = 0

    def measure(pb, blocks, what):
        perm = getattr(pb, "node_permutation", None)
        levels = problems.tensor_prolongators(pb.params, min_coarse=_abi.BENCH_MIN_COARSE, node_permutation=perm)
        ctx = solver.Context(0)
        t0 = time.time()
        solver.upload_problem(ctx, pb, c, levels, blocks)
        t_up = time.time() - t0
        info = ctx.matrix_info(_abi.A)
        rhs = ctx.augment_rhs([pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]])
        ctx.upload_rhs(rhs)
        ctx.solve_resident()
        ctx.enable_timing(True)
        res = ctx.solve_resident()
        t = ctx.timing()["spmv_A"]
        ctx.enable_timing(False)
        ctx.close()
        return {"what": what,
                "storage": ("batch-major" + (" (10-bit codes)" if info["batch_major_wide"] else "") if info["batch_major"]
                            else "value-indexed window" if info["value_indexed"] else "window 10 B/nnz" if info["windowed"] else "csr"),
                "shared_share": info["shared_nnz"] / max(info["nnz"], 1),
                "bytes_per_nnz": info["streamed_bytes"] / max(info["nnz"], 1),
                "spmv_A_ms": t["ms"] / max(t["launches"], 1), "upload_setup_s": t_up,
                "outer": res.outer_iterations, "inner": res.inner_iterations, "solve_s": res.solve_seconds,
                "value": res.outer_iterations / res.solve_seconds, "unit": "iterations/s"}

    t0 = time.time()
    legs = {}
    pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine)
    legs["kronecker_lexicographic"] = measure(pb, problems.brick_row_blocks(pb.params, brick),
                                              "the headline's operator at this size (closed-form rows, node-major lexicographic, mesh bricks)")
    del pb
    pb = problems.stokes3d_sphere(n_cells=n, immersed_refine=refine, assembly="cellwise")
    problems.permute_background_nodes(pb, problems.cuthill_mckee_nodes(pb))
    legs["cellwise_cuthill_mckee_as_handed_over"] = measure(pb, None, "cell-wise sums, Cuthill-McKee numbering, uploaded as is (no hint)")
    tf = time.time()
    pts = problems.row_support_points(pb.params, node_permutation=pb.node_permutation)
    n2o = solver.numbering_from_points(pts)
    nc = pb.params["ncomp"]
    problems.permute_background_nodes(pb, n2o[::nc] // nc)
    pts = problems.row_support_points(pb.params, node_permutation=pb.node_permutation)
    blocks = solver.brick_blocks_from_points(pts, brick)
    t_front = time.time() - tf
    legs["cellwise_cuthill_mckee_front_end_renumbered"] = measure(
        pb, blocks, "the same operator after the front end's renumbering from support points + mesh bricks from the points")
    legs["cellwise_cuthill_mckee_front_end_renumbered"]["front_end_s"] = t_front
    base = legs["kronecker_lexicographic"]["value"]
    for v in legs.values():
        v["fraction_of_kronecker_lexicographic"] = v["value"] / base
    log(f"reference-shaped leg (N={n}) in {time.time()-t0:.1f} s: " + ", ".join(f"{k} {v['value']:.2f} it/s" for k, v in legs.items()))
    return {"n_cells": n, "dofs_velocity": int(3 * (2 * n + 1) ** 3), "legs": legs}


def load_pmc(n_cells, info, bricks, world):
    """PMC summary of the A-SpMV kernel from an earlier profiled run (profiles/spmv_traffic.json, written by
    profiles/r03/scripts/make_traffic_json.py), used only when it was taken on this size, format and row blocks."""
    path = os.path.join(ROOT, "profiles", "spmv_traffic.json")
    if world != 1 or not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))
        if (t.get("n_cells") == n_cells and bool(t.get("value_indexed")) == bool(info["value_indexed"])
                and int(t.get("batch_major", 0)) == int(info["batch_major"]) and t.get("bricks", bricks) == bricks
                and "counters" in t):
            return t
    except Exception:
        pass
    return None


def host_cpu_share():
    """(CPUs this process may really use, how that number was found): the cgroup quota of the container if one is
    set, else the scheduler affinity mask -- never the host's core count (a GPU box hands one GPU's share of a
    256-thread host to the job)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    source = "affinity"
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if q != "max" and int(float(q) / float(per)) < n:
            n, source = max(1, int(float(q) / float(per))), "cgroup"
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())     # cgroup v1
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and q // per < n:
                n, source = max(1, q // per), "cgroup"
        except Exception:
            pass
    return n, source


def cpu_baseline(pb, cfg, rhs, aggregates, outer_its, inner_its):
    """The oracle (CPU port of the same algorithm and inner preconditioner, sequential row sums
    like deal.II's vmult) timed on the host cores of this box on the SAME full-size operators.
    A complete CPU solve would take minutes, so the sample is bounded:
      * all cores: the oracle's FGMRES run for its first TWO complete outer iterations (inner CG
        to its real stop rule, system operator, orthogonalisation): seconds per inner iteration,
        everything else included;
      * 1 thread (the reference is single-threaded, MPI_InitFinalize(argc, argv, 1)): one
        preconditioner application cut to two inner iterations.
    The same two outer iterations in canonical arithmetic (a second oracle run, row order 0) are the
    parity check at the bench size: returned as `prefix`, compared by the caller with the GPU's."""
    from fictitious_domain_al_preconditioners_amd import _abi
    from oracle import oracle

    share, source = host_cpu_share()
    if os.environ.get("ALFD_CPU_THREADS"):
        share, source = int(os.environ["ALFD_CPU_THREADS"]), "ALFD_CPU_THREADS"
    cores = oracle.set_threads(share)
    osys = oracle.system_from_problem(pb, aggregates=aggregates)
    c = _abi.Config.from_buffer_copy(cfg)
    c.outer.max_steps = 2                  # stop after two complete outer iterations
    t0 = time.time()
    h = osys.open(c)                       # setup once: diagonals, lambda_max, hierarchy, patch, coarsest inverse
    t_setup = time.time() - t0
    log(f"cpu_baseline: oracle setup {t_setup:.1f} s on {cores} threads")
    # (1) parity: canonical arithmetic
    oracle.set_row_order(0)
    rc, _, pres, phist = osys.handle_solve(h, rhs)
    if rc not in (0, _abi.E_NO_CONVERGENCE_OUTER):
        raise RuntimeError(f"cpu_baseline: oracle parity run failed (rc={rc})")
    prefix = {"history": phist, "inner": pres.inner_iterations, "mp": pres.mp_iterations}
    log(f"cpu_baseline: parity run {pres.solve_seconds:.1f} s")
    # (2) timing: plain sequential row sums, as deal.II's vmult does
    oracle.set_row_order(1)
    rc, _, res, _ = osys.handle_solve(h, rhs)
    if rc not in (0, _abi.E_NO_CONVERGENCE_OUTER) or res.inner_iterations < 1:
        raise RuntimeError(f"cpu_baseline: oracle run failed (rc={rc})")
    per_inner = res.solve_seconds / res.inner_iterations
    # 1 thread
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
        "host_cpu_share": share, "cpu_share_source": source, "host_logical_cpus": os.cpu_count(), "kind": "port",
        "sample": f"the oracle's FGMRES on the full-size operators with {cores} threads, first 2 complete outer "
                  f"iterations: {res.inner_iterations} inner + {res.mp_iterations} pressure-mass CG iterations in "
                  f"{res.solve_seconds:.1f} s (setup {t_setup:.1f} s untimed) = {per_inner:.2f} s per "
                  f"inner iteration, all overheads included; projected to the solve's {outer_its:g} outer / "
                  f"{inner_its:g} inner iterations (the same two iterations, in canonical arithmetic, are "
                  f"compared with the GPU's in parity_prefix)",
        "seconds_per_inner_iteration": per_inner, "oracle_setup_s": t_setup,
        "one_thread": {
            "value": outer_its / (per_inner_1 * inner_its), "unit": "iterations/s", "cores": 1,
            "sample": f"one preconditioner application cut to 2 inner iterations on 1 thread: {t_one:.1f} s = "
                      f"{per_inner_1:.2f} s per inner iteration, projected the same way; the reference itself runs "
                      f"on 1 thread (MPI_InitFinalize(argc, argv, 1))",
            "seconds_per_inner_iteration": per_inner_1,
        },
    }, prefix


if __name__ == "__main__":
    main()
