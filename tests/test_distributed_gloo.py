"""world_size-2 gloo test of the N > 1 path on CPU ranks.

Each rank generates only its row slab, builds the halo plan of every operator
with the PRODUCT's host routine (alfd_host_halo_plan -- the one
alfd_set_matrix runs before uploading), derives the send lists with the same
protocol the library runs (counts all-gather + id exchange) and exchanges the halo
values -- all through the product's host transport (hostcomm.torch_callbacks, the
callbacks alfd_comm_init_host installs; RCCL carries the same calls on GPUs) -- and
applies the oracle's canonical SpMV to [owned | halo].  The result must equal, bit for bit, the rows of the
single-process SpMV; the rank-ordered sum of local dots must equal the
oracle's emulated 2-rank dot."""
import os
import socket

import numpy as np
import pytest

WORLD = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _exchange(cbs, rank, world, halo_globals, recv_off, col_offsets, x_local):
    """Halo exchange with the library's protocol (upload_matrix / halo_exchange in csrc/alfd.hip), carried by
    the PRODUCT's host transport (hostcomm.torch_callbacks -- the callbacks alfd_comm_init_host installs):
    (1) all-gather of the per-owner request counts, (2) owners receive the wanted global ids,
    (3) owners send x[id - own_offset]."""
    import ctypes as C
    ag, a2a = cbs

    def ptr(a):
        return a.ctypes.data_as(C.c_void_p)

    def offs(a):
        return a.ctypes.data_as(C.POINTER(C.c_int64))

    recv_off = np.ascontiguousarray(recv_off, np.int64)
    want_cnt = np.diff(recv_off).astype(np.int32)
    all_cnt = np.zeros(world * world, np.int32)
    assert ag(None, ptr(want_cnt), ptr(all_cnt), want_cnt.nbytes) == 0
    send_off = np.zeros(world + 1, np.int64)
    for p in range(world):
        send_off[p + 1] = send_off[p] + all_cnt[p * world + rank]        # what p wants from me
    ids = np.ascontiguousarray(halo_globals, np.int32)
    wanted = np.zeros(max(int(send_off[-1]), 1), np.int32)
    assert a2a(None, ptr(ids), offs(recv_off), ptr(wanted), offs(send_off), 4) == 0
    c0 = int(col_offsets[rank])
    vals = np.ascontiguousarray(x_local[wanted[:int(send_off[-1])] - c0], np.float64)
    if vals.size == 0:
        vals = np.zeros(1)
    halo = np.zeros(max(len(halo_globals), 1))
    assert a2a(None, ptr(vals), offs(send_off), ptr(halo), offs(recv_off), 8) == 0
    return halo[:len(halo_globals)]


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        sys.path.insert(0, root)
        from fictitious_domain_al_preconditioners_amd import _abi, hostcomm, partition, problems, solver
        from oracle import oracle
        r_, w_, ag, a2a = hostcomm.torch_callbacks()
        assert (r_, w_) == (rank, world)
        n, ref = 6, 1
        plan = partition.slab_partition_stokes3d(n, ref, world)
        loc = problems.stokes3d_sphere(n, ref, row_ranges=plan.generator_ranges(rank))
        full = problems.stokes3d_sphere(n, ref)          # the single-process reference
        rng = np.random.default_rng(11)
        xg = [rng.uniform(-1, 1, s) for s in full.block_sizes]     # same on both ranks
        colblock = {"A": 0, "Bt": 1, "B": 0, "Ct": 2, "C": 0, "Mp": 1}
        rowblock = {"A": 0, "Bt": 0, "B": 1, "Ct": 0, "C": 2, "Mp": 1}
        ok = True
        for name in colblock:
            m = loc.mats[name]
            offs = plan.offsets[colblock[name]]
            col_local, halo_globals, recv_off = solver.host_halo_plan(m.col, offs, rank)
            x_local = xg[colblock[name]][int(offs[rank]):int(offs[rank + 1])]
            halo = _exchange((ag, a2a), rank, world, halo_globals, recv_off, offs, x_local)
            x_ext = np.concatenate([x_local, halo])
            mloc = problems.Csr(m.nrows, x_ext.size, m.row_ptr, col_local, m.val)
            # lanes must follow the GLOBAL matrix' rule here; the library applies the
            # rule to the local rows, as does the oracle emulation of that rank
            y_loc, lanes = oracle.spmv(mloc, x_ext)
            y_ref, _ = oracle.spmv(full.mats[name], xg[colblock[name]], lanes=lanes)
            ro = plan.offsets[rowblock[name]]
            ok &= bool(np.array_equal(y_loc, y_ref[int(ro[rank]):int(ro[rank + 1])]))
            # every halo id is off-rank and owned by the rank the plan says
            for p in range(world):
                ids = halo_globals[recv_off[p]:recv_off[p + 1]]
                ok &= bool(np.all((ids >= offs[p]) & (ids < offs[p + 1]))) and (p != rank or ids.size == 0)
        # rank-ordered dot: local canonical dot of the padded local block vector
        pad = lambda k: (k + 4095) // 4096 * 4096
        rhs = [full.vecs["f"], full.vecs["rhs_p"], full.vecs["g"]]
        lv = np.zeros(sum(pad(s) for s in loc.block_sizes))
        o = 0
        for b, s in enumerate(loc.block_sizes):
            g0 = int(plan.offsets[b][rank])
            lv[o:o + s] = rhs[b][g0:g0 + s]
            o += pad(s)
        mine = torch.tensor([oracle.dot(lv, lv)], dtype=torch.float64)
        allv = [torch.zeros(1, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(allv, mine)
        total = float(allv[0])
        for p in range(1, world):
            total = total + float(allv[p])
        cfg = _abi.default_config(_abi.AL_STOKES)
        cfg.inner.max_steps = 1000
        cfg.outer = _abi.Control(_abi.CTRL_ABS, 1000, 1e300, 0.0)     # stop at step 0: history[0] = |b|
        rc, _, res, hist = oracle.system_from_problem(full, nranks_emulated=world).solve(cfg, rhs)
        ok &= rc == 0 and float(np.sqrt(total)) == float(hist[0])
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_two_rank_halo_spmv_and_ordered_dot(built):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, q)) for r in range(WORLD)]
    for p in procs:
        p.start()
    results = [q.get(timeout=300) for _ in range(WORLD)]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
