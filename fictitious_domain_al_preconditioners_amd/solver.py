"""Host-side mirror of the reference's solver interface over the C ABI.

Everything here goes through ``lib/libalfd.so`` (include/alfd/alfd.h); there is
no Python or CPU fallback: if the HIP library is missing or no GPU is present
the calls raise.  The class names and ``vmult(dst, src)`` / ``solve(A, x, b, P)``
signatures follow the reference (augmented_lagrangian_preconditioner.h:14-110,
stokes_immersed_boundary.cc:1067-1074) so that tests read like the call sites.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libalfd.so")
_lib = None

# every symbol include/alfd/alfd.h declares
ABI_SYMBOLS = [
    "alfd_abi_version", "alfd_strerror", "alfd_last_error", "alfd_create", "alfd_destroy",
    "alfd_comm_unique_id", "alfd_comm_init", "alfd_set_partition", "alfd_set_matrix", "alfd_set_diag",
    "alfd_configure", "alfd_default_config", "alfd_setup", "alfd_precond_apply", "alfd_system_apply",
    "alfd_augment_rhs", "alfd_solve", "alfd_upload_rhs", "alfd_solve_resident", "alfd_download_solution",
    "alfd_get_history", "alfd_spmv", "alfd_dot", "alfd_matrix_lanes", "alfd_bench_spmv",
    "alfd_enable_timing", "alfd_get_timing", "alfd_host_halo_plan", "alfd_local_group_create",
    "alfd_local_group_destroy", "alfd_comm_init_local", "alfd_set_aggregates",
    "alfd_set_aggregate_partition", "alfd_get_matrix_info", "alfd_bench_spmv_format",
    "alfd_host_window_plan", "alfd_set_tunable", "alfd_build_aggregates", "alfd_get_aggregates",
    "alfd_host_aggregate_level", "alfd_comm_init_host",
    "alfd_get_device_memory", "alfd_set_row_blocks", "alfd_host_stream_plan",
    "alfd_host_row_blocks_from_points", "alfd_host_stream_plan_short", "alfd_set_prolongator", "alfd_set_controls", "alfd_get_timing_streamed", "alfd_get_setup_seconds",
    "alfd_host_numbering_from_points", "alfd_host_brick_blocks_from_points", "alfd_host_permute_csr",
]


class AlfdError(RuntimeError):
    """Non-zero status from the C ABI.  NoConvergence mirrors
    dealii::SolverControl::NoConvergence (stokes_immersed_boundary.cc:1233-1254)."""

    def __init__(self, status, message):
        super().__init__(f"alfd status {status}: {message}")
        self.status = status


class NoConvergence(AlfdError):
    pass


def load_library():
    """dlopen libalfd.so; raises ImportError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: the HIP extension is required (no CPU fallback). "
                          "Build it with `python -c 'import __graft_entry__ as g; g.build()'`.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    PP = C.POINTER(C.c_void_p)
    sig = {
        "alfd_abi_version": (C.c_int, []),
        "alfd_strerror": (C.c_char_p, [C.c_int]),
        "alfd_last_error": (C.c_char_p, [vp]),
        "alfd_create": (C.c_int, [C.POINTER(vp), C.c_int]),
        "alfd_destroy": (C.c_int, [vp]),
        "alfd_comm_unique_id": (C.c_int, [vp, C.c_size_t]),
        "alfd_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp, C.c_size_t]),
        "alfd_set_partition": (C.c_int, [vp, C.c_int, PP]),
        "alfd_set_matrix": (C.c_int, [vp, C.c_int, i64, i64, vp, vp, vp]),
        "alfd_set_diag": (C.c_int, [vp, C.c_int, i64, vp]),
        "alfd_configure": (C.c_int, [vp, C.POINTER(_abi.Config)]),
        "alfd_default_config": (None, [C.POINTER(_abi.Config), C.c_int]),
        "alfd_setup": (C.c_int, [vp]),
        "alfd_precond_apply": (C.c_int, [vp, PP, PP, C.POINTER(_abi.Result)]),
        "alfd_system_apply": (C.c_int, [vp, PP, PP]),
        "alfd_augment_rhs": (C.c_int, [vp, PP]),
        "alfd_solve": (C.c_int, [vp, PP, PP, C.POINTER(_abi.Result)]),
        "alfd_upload_rhs": (C.c_int, [vp, PP, PP]),
        "alfd_solve_resident": (C.c_int, [vp, C.POINTER(_abi.Result)]),
        "alfd_download_solution": (C.c_int, [vp, PP]),
        "alfd_get_history": (C.c_int, [vp, vp, i32, C.POINTER(i32)]),
        "alfd_spmv": (C.c_int, [vp, C.c_int, vp, vp, C.c_int, dbl]),
        "alfd_dot": (C.c_int, [vp, i64, vp, vp, C.POINTER(dbl)]),
        "alfd_matrix_lanes": (C.c_int, [vp, C.c_int, C.POINTER(i32)]),
        "alfd_bench_spmv": (C.c_int, [vp, C.c_int, i32, C.POINTER(dbl), C.POINTER(dbl)]),
        "alfd_enable_timing": (C.c_int, [vp, C.c_int]),
        "alfd_get_timing": (C.c_int, [vp, vp, vp, vp]),
        "alfd_host_halo_plan": (C.c_int, [i64, vp, vp, C.c_int, C.c_int, vp, vp, i64, C.POINTER(i64), vp]),
        "alfd_local_group_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "alfd_local_group_destroy": (C.c_int, [vp]),
        "alfd_comm_init_local": (C.c_int, [vp, vp, C.c_int]),
        "alfd_set_aggregates": (C.c_int, [vp, C.c_int, i64, vp, vp, i64]),
        "alfd_set_aggregate_partition": (C.c_int, [vp, C.c_int, vp]),
        "alfd_set_prolongator": (C.c_int, [vp, C.c_int, i64, i64, vp, vp, vp]),
        "alfd_set_controls": (C.c_int, [vp, vp, vp, vp]),
        "alfd_get_timing_streamed": (C.c_int, [vp, vp]),
        "alfd_get_setup_seconds": (C.c_int, [vp, vp]),
        "alfd_get_matrix_info": (C.c_int, [vp, C.c_int, C.POINTER(_abi.MatrixInfo)]),
        "alfd_bench_spmv_format": (C.c_int, [vp, C.c_int, i32, C.c_int, C.POINTER(dbl), C.POINTER(dbl)]),
        "alfd_host_window_plan": (C.c_int, [i64, vp, vp, vp, i32, i32, C.POINTER(_abi.WindowPlanInfo)]),
        "alfd_set_tunable": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "alfd_build_aggregates": (C.c_int, [vp, i32, dbl, i32, i64, i32, C.POINTER(i32)]),
        "alfd_get_aggregates": (C.c_int, [vp, C.c_int, vp, i64, C.POINTER(i64), C.POINTER(i64)]),
        "alfd_host_aggregate_level": (C.c_int, [i64, vp, vp, vp, i32, dbl, i32, vp, C.POINTER(i64)]),
        "alfd_comm_init_host": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
        "alfd_get_device_memory": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "alfd_set_row_blocks": (C.c_int, [vp, C.c_int, i64, vp, vp]),
        "alfd_host_stream_plan": (C.c_int, [i64, vp, vp, vp, C.c_int32, i64, vp, vp, vp]),
        "alfd_host_row_blocks_from_points": (C.c_int, [i64, C.c_int32, vp, C.c_int32, vp, vp, vp]),
        "alfd_host_stream_plan_short": (C.c_int, [i64, vp, vp, vp, C.c_int32, vp]),
        "alfd_host_numbering_from_points": (C.c_int, [i64, C.c_int32, vp, vp]),
        "alfd_host_brick_blocks_from_points": (C.c_int, [i64, C.c_int32, vp, vp, C.c_int32, vp, vp, vp]),
        "alfd_host_permute_csr": (C.c_int, [i64, vp, vp, vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def _blocks(arrs):
    a = (C.c_void_p * len(arrs))()
    for i, x in enumerate(arrs):
        a[i] = x.ctypes.data
    return a


class Context:
    """One alfd_ctx_t: one GPU, HBM-resident operators."""

    def __init__(self, device_id=0):
        self._lib = load_library()
        self._h = C.c_void_p()
        rc = self._lib.alfd_create(C.byref(self._h), device_id)
        if rc != _abi.OK:
            self._h = None
            raise AlfdError(rc, "alfd_create failed: " + self._lib.alfd_strerror(rc).decode() +
                            " (a HIP device is required; there is no CPU path)")
        self.block_sizes = None
        self.cfg = None

    def close(self):
        if getattr(self, "_h", None):
            self._lib.alfd_destroy(self._h)
            self._h = None

    __del__ = close

    def _ck(self, rc):
        if rc != _abi.OK:
            msg = self._lib.alfd_last_error(self._h).decode() or self._lib.alfd_strerror(rc).decode()
            cls = NoConvergence if rc in (_abi.E_NO_CONVERGENCE_OUTER, _abi.E_NO_CONVERGENCE_INNER) else AlfdError
            raise cls(rc, msg)

    # -- multi-GPU
    @staticmethod
    def unique_id() -> bytes:
        buf = C.create_string_buffer(_abi.UNIQUE_ID_BYTES)
        rc = load_library().alfd_comm_unique_id(buf, _abi.UNIQUE_ID_BYTES)
        if rc != _abi.OK:
            raise AlfdError(rc, "alfd_comm_unique_id failed")
        return buf.raw

    def comm_init(self, rank, nranks, unique_id: bytes):
        self._ck(self._lib.alfd_comm_init(self._h, rank, nranks, unique_id, len(unique_id)))

    def comm_init_local(self, group, rank):
        self._ck(self._lib.alfd_comm_init_local(self._h, group, rank))

    def comm_init_torch(self, group=None):
        """Multi-rank through torch.distributed HOST collectives (gloo or any CPU backend): the
        library hands its all-gathers and neighbour exchanges to hostcomm's callbacks as host
        buffers (alfd_comm_init_host).  One process per rank; library calls are collective."""
        from . import hostcomm
        rank, world, ag, a2a = hostcomm.torch_callbacks(group)
        self._host_cbs = (ag, a2a)                               # keep the trampolines alive
        self._ck(self._lib.alfd_comm_init_host(self._h, rank, world, C.cast(ag, C.c_void_p), C.cast(a2a, C.c_void_p),
                                               None))

    def set_partition(self, offsets):
        arrs = [np.ascontiguousarray(o, np.int64) for o in offsets]
        self._ck(self._lib.alfd_set_partition(self._h, len(arrs), _blocks(arrs)))

    # -- upload
    def set_matrix(self, slot, m):
        rp = np.ascontiguousarray(m.row_ptr, np.int64)
        col = np.ascontiguousarray(m.col, np.int32)
        val = np.ascontiguousarray(m.val, np.float64)
        self._ck(self._lib.alfd_set_matrix(self._h, slot, m.nrows, m.ncols, rp.ctypes.data, col.ctypes.data,
                                           val.ctypes.data))

    def set_diag(self, slot, d):
        d = np.ascontiguousarray(d, np.float64)
        self._ck(self._lib.alfd_set_diag(self._h, slot, d.size, d.ctypes.data))

    def set_aggregates(self, level, agg, n_coarse, weight=None):
        agg = np.ascontiguousarray(agg, np.int32)
        w = None if weight is None else np.ascontiguousarray(weight, np.float64)
        self._ck(self._lib.alfd_set_aggregates(self._h, level, agg.size, agg.ctypes.data,
                                               None if w is None else w.ctypes.data, int(n_coarse)))

    def set_prolongator(self, level, P):
        """CSR prolongator (problems.Csr, n_fine x n_coarse) of a multigrid level: alfd_set_prolongator."""
        self._ck(self._lib.alfd_set_prolongator(self._h, level, P.nrows, P.ncols, P.row_ptr.ctypes.data,
                                                P.col.ctypes.data, P.val.ctypes.data))

    def build_aggregates(self, block_size=1, threshold=0.02, max_aggregate_nodes=8, min_coarse=600, max_levels=7):
        """Algebraic aggregation from the uploaded A alone (alfd_build_aggregates); returns
        [(agg, n_coarse), ...] -- the list Context.set_aggregates / the oracle take."""
        nlev = C.c_int32(0)
        self._ck(self._lib.alfd_build_aggregates(self._h, block_size, threshold, max_aggregate_nodes, min_coarse,
                                                 max_levels, C.byref(nlev)))
        out = []
        for level in range(nlev.value):
            nf, nc = C.c_int64(0), C.c_int64(0)
            self._ck(self._lib.alfd_get_aggregates(self._h, level, None, 0, C.byref(nf), C.byref(nc)))
            agg = np.empty(nf.value, np.int32)
            self._ck(self._lib.alfd_get_aggregates(self._h, level, agg.ctypes.data, agg.size, C.byref(nf), C.byref(nc)))
            out.append((agg, int(nc.value)))
        return out

    def set_aggregate_partition(self, level, coarse_offsets):
        o = np.ascontiguousarray(coarse_offsets, np.int64)
        self._ck(self._lib.alfd_set_aggregate_partition(self._h, level, o.ctypes.data))

    def configure(self, cfg: _abi.Config):
        self.cfg = cfg
        self._ck(self._lib.alfd_configure(self._h, C.byref(cfg)))

    def set_controls(self, outer=None, inner=None, mp_inner=None):
        """New stop rules for the next solves, keeping the setup (alfd_set_controls)."""
        ref = lambda c: None if c is None else C.byref(c)
        self._ck(self._lib.alfd_set_controls(self._h, ref(outer), ref(inner), ref(mp_inner)))
        for name, c in (("outer", outer), ("inner", inner), ("mp_inner", mp_inner)):
            if c is not None:
                setattr(self.cfg, name, c)

    def setup(self, block_sizes):
        self._ck(self._lib.alfd_setup(self._h))
        self.block_sizes = [int(b) for b in block_sizes]

    # -- hot path
    def _in(self, blocks):
        out = [np.ascontiguousarray(b, np.float64) for b in blocks]
        if [b.size for b in out] != self.block_sizes:
            raise ValueError(f"block sizes {[b.size for b in out]} != {self.block_sizes}")
        return out

    def precond_apply(self, src):
        src = self._in(src)
        dst = [np.zeros(n) for n in self.block_sizes]
        res = _abi.Result()
        self._ck(self._lib.alfd_precond_apply(self._h, _blocks(src), _blocks(dst), C.byref(res)))
        return dst, res

    def system_apply(self, src):
        src = self._in(src)
        dst = [np.zeros(n) for n in self.block_sizes]
        self._ck(self._lib.alfd_system_apply(self._h, _blocks(src), _blocks(dst)))
        return dst

    def augment_rhs(self, rhs):
        rhs = [b.copy() for b in self._in(rhs)]
        self._ck(self._lib.alfd_augment_rhs(self._h, _blocks(rhs)))
        return rhs

    def solve(self, rhs, x0=None, raise_on_failure=True):
        rhs = self._in(rhs)
        x = [np.zeros(n) for n in self.block_sizes] if x0 is None else [b.copy() for b in self._in(x0)]
        res = _abi.Result()
        rc = self._lib.alfd_solve(self._h, _blocks(rhs), _blocks(x), C.byref(res))
        if rc != _abi.OK and raise_on_failure:
            self._ck(rc)
        return x, res

    def upload_rhs(self, rhs, x0=None):
        rhs = self._in(rhs)
        x0_arrs = self._in(x0) if x0 is not None else None   # kept alive across the call (may be copies)
        x0b = _blocks(x0_arrs) if x0_arrs is not None else None
        self._ck(self._lib.alfd_upload_rhs(self._h, _blocks(rhs), x0b))
        del x0_arrs

    def solve_resident(self, raise_on_failure=True):
        res = _abi.Result()
        rc = self._lib.alfd_solve_resident(self._h, C.byref(res))
        if rc != _abi.OK and raise_on_failure:
            self._ck(rc)
        return res

    def download_solution(self):
        x = [np.zeros(n) for n in self.block_sizes]
        self._ck(self._lib.alfd_download_solution(self._h, _blocks(x)))
        return x

    def history(self):
        cnt = C.c_int32(0)
        self._lib.alfd_get_history(self._h, None, 0, C.byref(cnt))
        out = np.zeros(max(cnt.value, 1))
        self._lib.alfd_get_history(self._h, out.ctypes.data, cnt.value, C.byref(cnt))
        return out[:cnt.value]

    # -- primitives
    def spmv(self, slot, x, y=None, mode=0, alpha=1.0):
        x = np.ascontiguousarray(x, np.float64)
        lanes = C.c_int32()
        self._ck(self._lib.alfd_matrix_lanes(self._h, slot, C.byref(lanes)))
        if y is None:
            raise ValueError("pass y (its length is the row count)")
        y = np.ascontiguousarray(y, np.float64).copy()
        self._ck(self._lib.alfd_spmv(self._h, slot, x.ctypes.data, y.ctypes.data, mode, alpha))
        return y, lanes.value

    def dot(self, x, y):
        x = np.ascontiguousarray(x, np.float64)
        y = np.ascontiguousarray(y, np.float64)
        out = C.c_double()
        self._ck(self._lib.alfd_dot(self._h, x.size, x.ctypes.data, y.ctypes.data, C.byref(out)))
        return out.value

    def bench_spmv(self, slot, reps=20):
        ms, nbytes = C.c_double(), C.c_double()
        self._ck(self._lib.alfd_bench_spmv(self._h, slot, reps, C.byref(ms), C.byref(nbytes)))
        return ms.value, nbytes.value

    def bench_spmv_format(self, slot, reps=20, value_index=True):
        """(ms per launch, bytes streamed by format) with the value-indexed kernel on or off."""
        ms, nbytes = C.c_double(), C.c_double()
        self._ck(self._lib.alfd_bench_spmv_format(self._h, slot, reps, int(value_index), C.byref(ms),
                                                  C.byref(nbytes)))
        return ms.value, nbytes.value

    def matrix_info(self, slot):
        info = _abi.MatrixInfo()
        self._ck(self._lib.alfd_get_matrix_info(self._h, slot, C.byref(info)))
        return {k: getattr(info, k) for k, _ in info._fields_ if k != "reserved"}

    def set_tunable(self, name, value):
        """Run-time switch (alfd_set_tunable), e.g. ("value_index", 0): general-matrix SpMV kernel."""
        self._ck(self._lib.alfd_set_tunable(self._h, name.encode(), int(value)))

    def set_row_blocks(self, slot, block_ptr, rows):
        """Row-block hint of the batch-major SpMV format (alfd_set_row_blocks); None removes it."""
        if block_ptr is None:
            self._ck(self._lib.alfd_set_row_blocks(self._h, slot, 0, None, None))
            return
        bp = np.ascontiguousarray(block_ptr, np.int64)
        rw = np.ascontiguousarray(rows, np.int32)
        self._ck(self._lib.alfd_set_row_blocks(self._h, slot, bp.size - 1, bp.ctypes.data, rw.ctypes.data))

    def device_memory(self):
        """(free, total) bytes of the context's GPU."""
        f, t = C.c_int64(0), C.c_int64(0)
        self._ck(self._lib.alfd_get_device_memory(self._h, C.byref(f), C.byref(t)))
        return f.value, t.value

    def enable_timing(self, on=True):
        self._ck(self._lib.alfd_enable_timing(self._h, int(on)))

    def timing(self):
        ms = np.zeros(_abi.T_NCLASSES)
        n = np.zeros(_abi.T_NCLASSES, np.int64)
        b = np.zeros(_abi.T_NCLASSES)
        self._ck(self._lib.alfd_get_timing(self._h, ms.ctypes.data, n.ctypes.data, b.ctypes.data))
        fb = np.zeros(_abi.T_NCLASSES)
        self._ck(self._lib.alfd_get_timing_streamed(self._h, fb.ctypes.data))
        names = ["spmv_A", "spmv_other", "dot", "vec"]
        return {k: dict(ms=float(ms[i]), launches=int(n[i]), bytes=float(b[i]), format_bytes=float(fb[i]))
                for i, k in enumerate(names)}

    def setup_seconds(self):
        """Wall seconds of the last uploads + setup by phase (alfd_get_setup_seconds)."""
        s = np.zeros(8)
        self._ck(self._lib.alfd_get_setup_seconds(self._h, s.ctypes.data))
        names = ["upload", "diag_lambda", "ml_fetch", "ml_galerkin", "ml_upload", "ml_lambda", "ml_patch", "ml_coarse"]
        return {k: float(s[i]) for i, k in enumerate(names)}


class LocalGroup:
    """In-process rank group (alfd_local_group): N contexts driven by N threads."""

    def __init__(self, nranks):
        self._lib = load_library()
        self.handle = C.c_void_p()
        rc = self._lib.alfd_local_group_create(nranks, C.byref(self.handle))
        if rc != _abi.OK:
            raise AlfdError(rc, "alfd_local_group_create failed")
        self.nranks = nranks

    def close(self):
        if self.handle:
            self._lib.alfd_local_group_destroy(self.handle)
            self.handle = None


def host_halo_plan(col, col_offsets, rank):
    """Host-only halo plan (no GPU): returns (col_local, halo_globals, recv_off)."""
    lib = load_library()
    col = np.ascontiguousarray(col, np.int32)
    offs = np.ascontiguousarray(col_offsets, np.int64)
    nranks = offs.size - 1
    col_local = np.empty_like(col)
    halo = np.empty(max(col.size, 1), np.int32)
    n_halo = C.c_int64()
    recv_off = np.zeros(nranks + 1, np.int64)
    rc = lib.alfd_host_halo_plan(col.size, col.ctypes.data, offs.ctypes.data, nranks, rank,
                                 col_local.ctypes.data, halo.ctypes.data, halo.size, C.byref(n_halo),
                                 recv_off.ctypes.data)
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_halo_plan failed")
    return col_local, halo[:n_halo.value].copy(), recv_off


def host_aggregate_level(m, block_size=1, threshold=0.02, max_aggregate_nodes=8):
    """Host-only: one level of the library's algebraic aggregation on a problems.Csr; (agg, n_coarse)."""
    lib = load_library()
    rp = np.ascontiguousarray(m.row_ptr, np.int64)
    col = np.ascontiguousarray(m.col, np.int32)
    val = np.ascontiguousarray(m.val, np.float64)
    agg = np.empty(m.nrows, np.int32)
    nc = C.c_int64(0)
    rc = lib.alfd_host_aggregate_level(m.nrows, rp.ctypes.data, col.ctypes.data, val.ctypes.data, block_size,
                                       threshold, max_aggregate_nodes, agg.ctypes.data, C.byref(nc))
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_aggregate_level failed")
    return agg, int(nc.value)


def host_window_plan(m, lanes=64, value_index=True):
    """Host-only plan of the LDS-window / value-indexed storage of a problems.Csr (no GPU);
    returns the alfd_window_plan_info fields, incl. decode_mismatches (0 for a correct plan)."""
    lib = load_library()
    rp = np.ascontiguousarray(m.row_ptr, np.int64)
    col = np.ascontiguousarray(m.col, np.int32)
    val = np.ascontiguousarray(m.val, np.float64)
    info = _abi.WindowPlanInfo()
    rc = lib.alfd_host_window_plan(m.nrows, rp.ctypes.data, col.ctypes.data, val.ctypes.data, lanes,
                                   int(value_index), C.byref(info))
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_window_plan failed")
    return {k: getattr(info, k) for k, _ in info._fields_}


def host_stream_plan_short(m, lanes):
    """Host-only plan + decode of the short-row batch-major form (alfd_host_stream_plan_short; lanes = 8, 16, 32)."""
    lib = load_library()
    rp = np.ascontiguousarray(m.row_ptr, np.int64)
    col = np.ascontiguousarray(m.col, np.int32)
    val = np.ascontiguousarray(m.val, np.float64)
    info = _abi.StreamPlanInfo()
    rc = lib.alfd_host_stream_plan_short(m.nrows, rp.ctypes.data, col.ctypes.data, val.ctypes.data, lanes, C.byref(info))
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_stream_plan_short failed")
    return {k: getattr(info, k) for k, _ in info._fields_}


def numbering_from_points(points):
    """new_to_old permutation that puts unknowns into the lexicographic order of their support points
    (alfd_host_numbering_from_points); the components of a node stay together."""
    pts = np.ascontiguousarray(points, np.float64)
    out = np.empty(pts.shape[0], np.int64)
    rc = load_library().alfd_host_numbering_from_points(pts.shape[0], pts.shape[1], pts.ctypes.data, out.ctypes.data)
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_numbering_from_points")
    return out


def brick_blocks_from_points(points, brick=(16, 4, 1), max_rows=250):
    """(block_ptr, rows) for Context.set_row_blocks: mesh bricks found from support points alone
    (alfd_host_brick_blocks_from_points)."""
    pts = np.ascontiguousarray(points, np.float64)
    n, dim = pts.shape
    b = np.ascontiguousarray(list(brick)[:dim], np.int32)
    bp = np.empty(n + 1, np.int64)
    rows = np.empty(n, np.int32)
    nb = C.c_int64(0)
    rc = load_library().alfd_host_brick_blocks_from_points(n, dim, pts.ctypes.data, b.ctypes.data, max_rows, C.byref(nb),
                                                           bp.ctypes.data, rows.ctypes.data)
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_brick_blocks_from_points")
    return bp[:nb.value + 1].copy(), rows


def permute_csr(m, row_new_to_old=None, col_old_to_new=None):
    """problems.Csr permuted on the host (alfd_host_permute_csr): rows taken in the order row_new_to_old, column j
    renamed col_old_to_new[j], rows re-sorted."""
    from .problems import Csr
    rp = np.empty(m.nrows + 1, np.int64)
    col = np.empty(m.nnz, np.int32)
    val = np.empty(m.nnz, np.float64)
    r = None if row_new_to_old is None else np.ascontiguousarray(row_new_to_old, np.int64)
    c = None if col_old_to_new is None else np.ascontiguousarray(col_old_to_new, np.int64)
    rc = load_library().alfd_host_permute_csr(m.nrows, m.row_ptr.ctypes.data, m.col.ctypes.data, m.val.ctypes.data,
                                              None if r is None else r.ctypes.data, None if c is None else c.ctypes.data,
                                              rp.ctypes.data, col.ctypes.data, val.ctypes.data)
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_permute_csr")
    return Csr(m.nrows, m.ncols, rp, col, val)


def row_blocks_from_points(points, max_rows=192):
    """(block_ptr, rows) for Context.set_row_blocks from one support point per matrix row
    (alfd_host_row_blocks_from_points: recursive coordinate bisection, no grid metadata needed)."""
    lib = load_library()
    pts = np.ascontiguousarray(points, np.float64)
    n, dim = pts.shape
    ptr = np.zeros(n + 1, np.int64)
    rows = np.zeros(n, np.int32)
    nb = C.c_int64(0)
    rc = lib.alfd_host_row_blocks_from_points(n, dim, pts.ctypes.data, max_rows, C.byref(nb), ptr.ctypes.data,
                                              rows.ctypes.data)
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_row_blocks_from_points failed")
    return ptr[:nb.value + 1].copy(), rows


def host_stream_plan(m, row_block=96, blocks=None):
    """Host-only plan + decode of the batch-major format (alfd_host_stream_plan);
    blocks = (block_ptr, rows) as for Context.set_row_blocks, or None for runs of row_block rows."""
    lib = load_library()
    rp = np.ascontiguousarray(m.row_ptr, np.int64)
    col = np.ascontiguousarray(m.col, np.int32)
    val = np.ascontiguousarray(m.val, np.float64)
    info = _abi.StreamPlanInfo()
    if blocks is None:
        rc = lib.alfd_host_stream_plan(m.nrows, rp.ctypes.data, col.ctypes.data, val.ctypes.data, row_block,
                                       0, None, None, C.byref(info))
    else:
        bp = np.ascontiguousarray(blocks[0], np.int64)
        rw = np.ascontiguousarray(blocks[1], np.int32)
        rc = lib.alfd_host_stream_plan(m.nrows, rp.ctypes.data, col.ctypes.data, val.ctypes.data, row_block,
                                       bp.size - 1, bp.ctypes.data, rw.ctypes.data, C.byref(info))
    if rc != _abi.OK:
        raise AlfdError(rc, "alfd_host_stream_plan failed")
    return {k: getattr(info, k) for k, _ in info._fields_}


def upload_problem(ctx: Context, pb, cfg: _abi.Config, aggregates=None, row_blocks=None) -> Context:
    """Upload a problems.SyntheticProblem (whole, or this rank's rows) with the
    reference's diagonal choices: W^-1 = 1/M_ii^2 (stokes...:976-978), lumped
    pressure mass (stokes...:946-954).  aggregates: [(agg, n_coarse), ...] for
    ALFD_PREC_MULTILEVEL (problems.geometric_aggregates), or [(Csr P, n_coarse), ...]
    (problems.tensor_prolongators), or a zero-argument callable returning either (evaluated on a helper
    thread while the operators are uploaded).  row_blocks: (block_ptr, rows)
    for the SpMV on A (Context.set_row_blocks, e.g. problems.brick_row_blocks)."""
    if row_blocks is not None:
        ctx.set_row_blocks(_abi.A, *row_blocks)

    def set_hierarchy(levels):
        for level, entry in enumerate(levels or []):
            agg, nc = entry[0], entry[1]
            if hasattr(agg, "row_ptr"):                       # a CSR prolongator (problems.tensor_prolongators)
                ctx.set_prolongator(level, agg)
                if len(entry) > 2 and entry[2] is not None:   # partitioned context: coarse offsets by rank (partition.local_prolongators)
                    ctx.set_aggregate_partition(level, entry[2])
                continue
            ctx.set_aggregates(level, agg, nc)
            if len(entry) > 2 and entry[2] is not None:      # (agg_local, n_coarse_global, coarse_offsets)
                ctx.set_aggregate_partition(level, entry[2])

    # aggregates may be a zero-argument callable: the transfer operators are then built on a helper thread while this one
    # uploads the operators (the library's format planning runs outside the interpreter lock), and handed over before setup
    pending = None
    if callable(aggregates):
        import threading
        box = {}

        def work():
            try:
                box["levels"] = aggregates()
            except BaseException as e:   # noqa: BLE001  (re-raised on the caller's thread)
                box["error"] = e
        pending = threading.Thread(target=work)
        pending.start()
    else:
        set_hierarchy(aggregates)

    def finish_hierarchy():
        if pending is not None:
            pending.join()
            if "error" in box:
                raise box["error"]
            set_hierarchy(box["levels"])

    ctx.set_matrix(_abi.A, pb.mats["A"])
    # C before CT (and B before BT): an explicitly uploaded transpose is left alone, otherwise alfd_set_matrix(CT) would
    # first derive C on the host (a single-threaded transpose + upload) only to see it replaced by the next call
    ctx.set_matrix(_abi.C_, pb.mats["C"])
    ctx.set_matrix(_abi.CT, pb.mats["Ct"])
    if cfg.variant == _abi.RATIONAL:
        # rational branch (immersed_laplace.cc:585-631): K, Ct, immersed stiffness and mass
        ctx.set_matrix(_abi.M, pb.mats["M"])
        ctx.set_matrix(_abi.KIMM, pb.mats["K"])
        finish_hierarchy()
        ctx.configure(cfg)
        ctx.setup(pb.block_sizes)
        return ctx
    if "A2" in pb.mats:
        # elliptic interface: W^-1 = 1/(M^2)_ii (utilities.h:348-374, elliptic_interface.cc:726)
        ctx.set_matrix(_abi.A2, pb.mats["A2"])
        ctx.set_matrix(_abi.M, pb.mats["M"])
        ctx.set_diag(_abi.INVW, pb.inv_w_diag_of_mass_squared())
        finish_hierarchy()
        ctx.configure(cfg)
        ctx.setup(pb.block_sizes)
        return ctx
    if cfg.w_inverse != _abi.W_DIAGONAL:
        ctx.set_matrix(_abi.M, pb.mats["M"])     # exact W^-1: CG on the immersed mass matrix
    ctx.set_diag(_abi.INVW, pb.inv_w_diag_squared())
    if "B" in pb.mats:
        ctx.set_matrix(_abi.B, pb.mats["B"])
        ctx.set_matrix(_abi.BT, pb.mats["Bt"])
        ctx.set_matrix(_abi.MP, pb.mats["Mp"])
        ctx.set_diag(_abi.MP_LUMPED_INV, pb.mp_lumped_inv())
    finish_hierarchy()
    ctx.configure(cfg)
    ctx.setup(pb.block_sizes)
    return ctx


def context_from_problem(pb, cfg: _abi.Config, device_id=0, aggregates=None, row_blocks=None) -> Context:
    return upload_problem(Context(device_id), pb, cfg, aggregates, row_blocks)


# ---------------------------------------------------------------------------
# Reference-shaped front-ends.  A BlockVector is a list of numpy arrays.
class _ALPreconditionerBase:
    variant = None

    def __init__(self, ctx: Context):
        if ctx.cfg is None or ctx.cfg.variant != self.variant:
            raise ValueError("context is configured for a different preconditioner variant")
        self.ctx = ctx
        self.last_result = None

    def vmult(self, dst, src):
        """void vmult(BlockVector<double>& v, const BlockVector<double>& u) const"""
        out, self.last_result = self.ctx.precond_apply(src)
        for d, o in zip(dst, out):
            d[...] = o


class BlockPreconditionerAugmentedLagrangian(_ALPreconditionerBase):
    """augmented_lagrangian_preconditioner.h:14-42."""
    variant = _abi.AL2


class BlockPreconditionerAugmentedLagrangianStokes(_ALPreconditionerBase):
    """augmented_lagrangian_preconditioner.h:44-79."""
    variant = _abi.AL_STOKES


class BlockPreconditionerAugmentedLagrangianDiagonal(_ALPreconditionerBase):
    """augmented_lagrangian_preconditioner.h:81-110."""
    variant = _abi.AL_STOKES_DIAG


class BlockTriangularALPreconditioner(_ALPreconditionerBase):
    """EllipticInterfacePreconditioners::BlockTriangularALPreconditioner,
    augmented_lagrangian_preconditioner.h:115-164."""
    variant = _abi.AL_ELL_IDEAL


class BlockTriangularALPreconditionerModified(_ALPreconditionerBase):
    """EllipticInterfacePreconditioners::BlockTriangularALPreconditionerModified,
    augmented_lagrangian_preconditioner.h:168-238."""
    variant = _abi.AL_ELL_MODIFIED


class RationalPreconditioner(_ALPreconditionerBase):
    """rational_preconditioner.h:12-99."""
    variant = _abi.RATIONAL


class SystemOperator:
    """The block_operator AA (stokes_immersed_boundary.cc:1000-1003)."""

    def __init__(self, ctx: Context):
        self.ctx = ctx

    def vmult(self, dst, src):
        out = self.ctx.system_apply(src)
        for d, o in zip(dst, out):
            d[...] = o


class SolverMinRes:
    """SolverMinRes<BlockVector<double>> (immersed_laplace.cc:629-631, stokes...:1057-1064):
    the context must be configured with outer_solver = OUTER_MINRES."""

    def __init__(self, ctx: Context):
        if ctx.cfg is None or ctx.cfg.outer_solver != _abi.OUTER_MINRES:
            raise ValueError("context is not configured for MinRes")
        self.ctx = ctx
        self.last_result = None

    def solve(self, A, x, b, P):
        if A.ctx is not self.ctx or P.ctx is not self.ctx:
            raise ValueError("operator, preconditioner and solver must share one context")
        sol, self.last_result = self.ctx.solve(b, x0=x)
        for d, o in zip(x, sol):
            d[...] = o

    def last_step(self):
        return self.last_result.outer_iterations


class SolverFGMRES:
    """SolverFGMRES<BlockVector<double>> (stokes_immersed_boundary.cc:1067): the
    control lives in the context's alfd_config.outer; solve() runs wholly on the GPU."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.last_result = None

    def solve(self, A: SystemOperator, x, b, P: _ALPreconditionerBase):
        if A.ctx is not self.ctx or P.ctx is not self.ctx:
            raise ValueError("operator, preconditioner and solver must share one context")
        sol, self.last_result = self.ctx.solve(b, x0=x)
        for d, o in zip(x, sol):
            d[...] = o

    def last_step(self):
        return self.last_result.outer_iterations
