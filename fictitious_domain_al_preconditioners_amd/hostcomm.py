"""torch.distributed (gloo / any CPU backend) transport for alfd_comm_init_host.

The library's multi-rank path needs two collectives (include/alfd/alfd.h): an all-gather of a
few bytes per reduction and a personalised neighbour exchange per halo.  With RCCL they run on
device buffers over xGMI; here they are handed over as HOST buffers and travel through whatever
process group the launcher has -- what an MPI program such as the reference's deal.II drivers
would plug in, and the vehicle of the two-process GPU test (tests/test_gpu_multiprocess.py).
"""
import ctypes as C

AG_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
A2A_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.c_void_p, C.POINTER(C.c_int64),
                     C.c_size_t)


def torch_callbacks(group=None):
    """-> (rank, world, allgather trampoline, alltoallv trampoline)."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)

    def view(ptr, nbytes):
        if nbytes <= 0:
            return torch.empty(0, dtype=torch.uint8)
        return torch.frombuffer((C.c_char * nbytes).from_address(ptr), dtype=torch.uint8)

    def allgather(_user, send, recv, nbytes):
        try:
            pieces = [torch.empty(nbytes, dtype=torch.uint8) for _ in range(world)]
            dist.all_gather(pieces, view(send, nbytes).clone(), group=group)
            out = view(recv, nbytes * world)
            for p in range(world):
                out[p * nbytes:(p + 1) * nbytes] = pieces[p]
            return 0
        except Exception as e:     # noqa: BLE001 -- an exception must not unwind through the C frame
            print("alfd host all-gather failed:", repr(e), flush=True)
            return 1

    def alltoallv(_user, send, soff, recv, roff, es):
        try:
            sb, rb = view(send, soff[world] * es), view(recv, roff[world] * es)
            ops, landing = [], {}
            for p in range(world):
                ns, nr = (soff[p + 1] - soff[p]) * es, (roff[p + 1] - roff[p]) * es
                if p == rank:
                    if nr:
                        rb[roff[p] * es:roff[p] * es + nr] = sb[soff[p] * es:soff[p] * es + ns]
                    continue
                if ns:
                    ops.append(dist.P2POp(dist.isend, sb[soff[p] * es:soff[p] * es + ns].clone(), p, group=group))
                if nr:
                    landing[p] = torch.empty(nr, dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.irecv, landing[p], p, group=group))
            if ops:
                for req in dist.batch_isend_irecv(ops):
                    req.wait()
            for p, buf in landing.items():
                rb[roff[p] * es:roff[p] * es + buf.numel()] = buf
            return 0
        except Exception as e:     # noqa: BLE001
            print("alfd host all-to-all failed:", repr(e), flush=True)
            return 1

    return rank, world, AG_FN(allgather), A2A_FN(alltoallv)
