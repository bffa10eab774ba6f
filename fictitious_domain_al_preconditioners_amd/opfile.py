"""Operator dump / load wire format.

A deal.II user runs the reference once with the exporter of
``include/alfd/dealii_export.hpp`` next to ``solve()`` and gets one ``.alfd``
file holding the real operators (hanging nodes, local refinement, true coupling
quadrature), the right-hand side and the solver knobs; the file is replayed
here without deal.II (``python -m fictitious_domain_al_preconditioners_amd.opfile
file.alfd``).  The reference only has an ad-hoc text dump for matrices up to
1000 rows (utilities.h:84-109, ``row+1 col+1 value``).

Layout (little endian), all integers int64 unless noted:
    magic  "ALFDOPS1" (8 bytes)
    n_records
    record*:  kind(int64: 1 matrix, 2 diag, 3 vector, 4 config)  id  payload
      matrix: id = alfd_matrix_slot; nrows ncols nnz; row_ptr[nrows+1] int64;
              col[nnz] int32 (padded to 8 bytes); val[nnz] float64
      diag:   id = alfd_diag_slot;   n; d[n] float64
      vector: id = 100 + block for rhs blocks, 200 + block for initial guess; n; v[n] float64
      config: id = sizeof(alfd_config); raw bytes padded to 8
"""
from __future__ import annotations

import struct
import sys

import numpy as np

from . import _abi
from .problems import Csr

MAGIC = b"ALFDOPS1"
K_MATRIX, K_DIAG, K_VECTOR, K_CONFIG = 1, 2, 3, 4


def save(path, mats: dict, diags: dict, rhs, cfg: _abi.Config, x0=None):
    """mats: {slot: Csr}; diags: {slot: array}; rhs / x0: list of block arrays."""
    recs = []
    for slot, m in mats.items():
        col = np.ascontiguousarray(m.col, np.int32)
        pad = b"\0" * ((-col.nbytes) % 8)
        recs.append(struct.pack("<qqqqq", K_MATRIX, slot, m.nrows, m.ncols, m.nnz)
                    + np.ascontiguousarray(m.row_ptr, np.int64).tobytes() + col.tobytes() + pad
                    + np.ascontiguousarray(m.val, np.float64).tobytes())
    for slot, d in diags.items():
        d = np.ascontiguousarray(d, np.float64)
        recs.append(struct.pack("<qqq", K_DIAG, slot, d.size) + d.tobytes())
    for base, blocks in ((100, rhs), (200, x0 or [])):
        for b, v in enumerate(blocks):
            v = np.ascontiguousarray(v, np.float64)
            recs.append(struct.pack("<qqq", K_VECTOR, base + b, v.size) + v.tobytes())
    raw = bytes(cfg)
    recs.append(struct.pack("<qq", K_CONFIG, len(raw)) + raw + b"\0" * ((-len(raw)) % 8))
    with open(path, "wb") as f:
        f.write(MAGIC + struct.pack("<q", len(recs)))
        for r in recs:
            f.write(r)


def load(path):
    """-> (mats {slot: Csr}, diags {slot: array}, rhs blocks, x0 blocks or None, Config)"""
    buf = memoryview(np.fromfile(path, dtype=np.uint8))
    if bytes(buf[:8]) != MAGIC:
        raise ValueError("not an ALFDOPS1 file")
    pos = 8
    (nrec,) = struct.unpack_from("<q", buf, pos)
    pos += 8
    mats, diags, rhs, x0, cfg = {}, {}, {}, {}, None

    def take(dtype, count):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=pos)
        pos += a.nbytes + ((-a.nbytes) % 8)
        return a

    for _ in range(nrec):
        kind, ident = struct.unpack_from("<qq", buf, pos)
        pos += 16
        if kind == K_MATRIX:
            nrows, ncols, nnz = struct.unpack_from("<qqq", buf, pos)
            pos += 24
            rp = take(np.int64, nrows + 1)
            col = take(np.int32, nnz)
            val = take(np.float64, nnz)
            mats[ident] = Csr(nrows, ncols, rp, col, val)
        elif kind in (K_DIAG, K_VECTOR):
            (n,) = struct.unpack_from("<q", buf, pos)
            pos += 8
            v = take(np.float64, n)
            if kind == K_DIAG:
                diags[ident] = v
            elif ident >= 200:
                x0[ident - 200] = v
            else:
                rhs[ident - 100] = v
        elif kind == K_CONFIG:
            raw = bytes(take(np.uint8, ident))
            if ident != len(bytes(_abi.Config())):
                raise ValueError(f"alfd_config size {ident} in file, {len(bytes(_abi.Config()))} here (ABI mismatch)")
            cfg = _abi.Config.from_buffer_copy(raw)
        else:
            raise ValueError(f"unknown record kind {kind}")
    return (mats, diags, [rhs[b] for b in sorted(rhs)], [x0[b] for b in sorted(x0)] or None, cfg)


def replay(path, device_id=0):
    """Upload a dumped system and solve it on the GPU; returns (x blocks, Result, history)."""
    from . import solver
    mats, diags, rhs, x0, cfg = load(path)
    ctx = solver.Context(device_id)
    for slot, m in mats.items():
        ctx.set_matrix(slot, m)
    for slot, d in diags.items():
        ctx.set_diag(slot, d)
    ctx.configure(cfg)
    ctx.setup([b.size for b in rhs])
    x, res = ctx.solve(rhs, x0=x0, raise_on_failure=False)
    hist = ctx.history()
    ctx.close()
    return x, res, hist


if __name__ == "__main__":
    _, res, hist = replay(sys.argv[1])
    print(f"status {res.status} outer {res.outer_iterations} inner {res.inner_iterations} "
          f"residual {res.last_residual:.6e}")
