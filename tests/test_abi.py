"""The C-ABI library loads and exports every symbol include/alfd/*.h declares;
no compute calls (no GPU here)."""
import ctypes as C
import os
import re

import pytest

from fictitious_domain_al_preconditioners_amd import _abi, solver

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _built(built):
    return built


def _declared_symbols():
    txt = open(os.path.join(ROOT, "include", "alfd", "alfd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(alfd_[a-z_0-9]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    lib = solver.load_library()
    declared = _declared_symbols()
    assert len(declared) >= 25
    missing = [s for s in declared if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(solver.ABI_SYMBOLS) == declared     # the Python mirror binds all of them


def test_version_strerror_and_default_config_mirror():
    lib = solver.load_library()
    assert lib.alfd_abi_version() == 12
    assert b"NoConvergence" in lib.alfd_strerror(_abi.E_NO_CONVERGENCE_INNER)
    assert lib.alfd_strerror(_abi.OK) == b"ok"
    for variant in (_abi.AL2, _abi.AL_STOKES, _abi.AL_ELL_MODIFIED, _abi.RATIONAL):
        c = _abi.Config()
        lib.alfd_default_config(C.byref(c), variant)
        py = _abi.default_config(variant)
        assert bytes(c) == bytes(py)                  # same layout, same defaults
    # reference defaults: restart 30 (50 for elliptic_interface.cc:863), inner 100 / 1e-2 abs
    c = _abi.default_config(_abi.AL_STOKES)
    assert (c.restart, c.inner.max_steps, c.inner.tol, c.inner.kind) == (30, 100, 1e-2, _abi.CTRL_ABS)
    assert _abi.default_config(_abi.AL_ELL_MODIFIED).restart == 50
    assert C.sizeof(_abi.Config) == 264 and C.sizeof(_abi.Result) == 80
    assert C.sizeof(_abi.MatrixInfo) == 120 and C.sizeof(_abi.WindowPlanInfo) == 88


def test_argument_validation_without_gpu():
    lib = solver.load_library()
    assert lib.alfd_destroy(None) == _abi.E_INVALID
    assert lib.alfd_setup(None) == _abi.E_INVALID
    assert lib.alfd_comm_unique_id(None, 0) == _abi.E_INVALID
    assert lib.alfd_last_error(None) == b"null context"


def test_no_cpu_fallback():
    """Without a HIP device the product path fails loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(solver.AlfdError) as e:
        solver.Context(0)
    assert e.value.status == _abi.E_HIP
    assert "no CPU path" in str(e.value)


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "fictitious_domain_al_preconditioners_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                # no import, dlopen, include or path of anything under oracle/
                hits = re.findall(r"(import\s+oracle|from\s+oracle|liboracle|oracle/|orc_[a-z_]+\s*\()", txt)
                assert not hits, (os.path.join(dirpath, f), hits)


def test_host_stream_plan_decodes_back():
    """The batch-major value-indexed storage (kernels_vs.hpp) planned on the host decodes back to the
    CSR it was made from -- rows through the batch descriptors, lane-major chunks, window segments,
    dictionaries -- for runs of the numbering, mesh bricks, ragged caller blocks; blocks with more than
    512 distinct values are halved; a matrix with unrelated values is refused.  No GPU involved."""
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import problems
    pb = problems.generate(dim=3, degree=2, ncomp=3, n_cells=6, stokes=False, grad_div=True,
                           gamma_grad_div=10.0, radius=0.1, immersed_refine=0)
    a = pb.mats["A"]
    runs = solver.host_stream_plan(a)
    assert runs["ok"] and runs["decode_mismatches"] == 0 and runs["rows_covered"] == a.nrows
    assert runs["blocks"] >= -(-a.nrows // 96) and runs["max_rows"] <= 96
    assert 0.75 * a.nnz <= runs["stream_bytes"] < 3.6 * a.nnz          # 3 B/nnz plain, 1 B/nnz in template-shared batches
    assert 0 < runs["shared_nnz"] <= a.nnz
    bricks = solver.host_stream_plan(a, blocks=problems.brick_row_blocks(pb.params, (8, 4, 2)))
    assert bricks["ok"] and bricks["decode_mismatches"] == 0 and bricks["rows_covered"] == a.nrows
    assert bricks["max_window"] < runs["max_window"]                # what the bricks are for (a third on large grids)
    rng = np.random.default_rng(7)
    perm = rng.permutation(a.nrows)
    perm = perm[np.argsort(perm // 40, kind="stable")]               # shuffled inside groups of 40 rows
    ptr = np.unique(np.r_[0, np.cumsum(rng.integers(1, 60, a.nrows // 20)), a.nrows])
    ptr = ptr[ptr <= a.nrows]
    ragged = solver.host_stream_plan(a, blocks=(ptr, perm))
    assert ragged["ok"] and ragged["decode_mismatches"] == 0 and ragged["rows_covered"] == a.nrows
    # every value scaled by one of 16 factors: some blocks exceed 512 distinct values (with 9-bit codes alone they were
    # halved) -- re-planned with 10-bit codes / 11-bit window columns (VsFmt<1>: up to 1024 values per block), none is cut
    v = np.array(a.val) * (1.0 + 0.0625 * rng.integers(0, 16, a.nnz))
    wide = solver.host_stream_plan(problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), v))
    assert wide["ok"] and wide["decode_mismatches"] == 0 and wide["rows_covered"] == a.nrows
    assert wide["blocks"] == runs["blocks"] and wide["dictionary_entries"] > 10 * runs["dictionary_entries"]
    # one of 64 factors: beyond 1024 values per block too, blocks are halved until they fit
    v = np.array(a.val) * (1.0 + 0.015625 * rng.integers(0, 64, a.nnz))
    many = problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), v)
    info = solver.host_stream_plan(many)
    assert info["ok"] and info["decode_mismatches"] == 0 and info["rows_covered"] == a.nrows
    assert info["blocks"] > runs["blocks"] and info["dictionary_entries"] <= 1024 * info["blocks"]
    assert info["shared_nnz"] < runs["shared_nnz"]                       # randomly scaled rows are no translates
    # unrelated values: not representable (the library keeps the other formats)
    rnd = problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), rng.uniform(-1, 1, a.nnz))
    assert not solver.host_stream_plan(rnd)["ok"]


def test_host_stream_plan_splits_blocks_with_wide_windows():
    """An operator whose rows reach far -- the divergence block B of a Taylor-Hood pair: 96 consecutive pressure rows
    touch far more than 4096 velocity columns -- keeps the batch-major form: blocks are split until their x window
    fits (sampled split factor, exact halving as the fall-back), 16 translate rows per shared batch; the plan decodes
    back to the CSR it was made from.  No GPU involved."""
    from fictitious_domain_al_preconditioners_amd import problems
    pb = problems.stokes3d_sphere(n_cells=12, immersed_refine=1)
    b = pb.mats["B"]
    info = solver.host_stream_plan(b)
    assert info["ok"] and info["decode_mismatches"] == 0 and info["rows_covered"] == b.nrows, info
    assert info["max_window"] <= 2048 and info["max_rows"] < 96            # split for the window, to 8 workgroups per CU
    assert info["shared_nnz"] > 0.6 * b.nnz and info["stream_bytes"] < 3.0 * b.nnz   # a 12^3 grid is mostly boundary
    a = pb.mats["A"]
    bricks = solver.host_stream_plan(a, blocks=problems.brick_row_blocks(pb.params, (16, 4, 1)))
    assert bricks["ok"] and bricks["decode_mismatches"] == 0 and bricks["rows_covered"] == a.nrows
    assert bricks["shared_nnz"] > 0.8 * a.nnz      # 16-row batches included (twelve row types x 16 translates per interior brick)


def test_host_stream_plan_short_rows_decodes_back():
    """The short-row batch-major form (spmv_vss_kernel: one stored template row per batch of translate rows)
    planned on the host decodes back to the CSR, for L = 32 / 16 / 8 lanes per row; nearly every entry of a
    stencil operator sits in a shared batch; rows longer than 4 L are refused."""
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import problems
    for gen, lanes in ((dict(dim=3, degree=1, ncomp=1, n_cells=24), 32), (dict(dim=2, degree=2, ncomp=1, n_cells=60), 16),
                       (dict(dim=2, degree=1, ncomp=1, n_cells=120), 8)):
        a = problems.generate(radius=0.1, **gen).mats["A"]
        info = solver.host_stream_plan_short(a, lanes)
        assert info["ok"] and info["decode_mismatches"] == 0 and info["rows_covered"] == a.nrows, (lanes, info)
        assert info["shared_nnz"] > 0.9 * a.nnz and info["stream_bytes"] < 2.5 * a.nnz
    rng = np.random.default_rng(2)
    v = np.array(a.val) * (1.0 + 0.25 * rng.integers(0, 3, a.nnz))          # rows no longer translates
    info = solver.host_stream_plan_short(problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), v), 8)
    assert info["ok"] and info["decode_mismatches"] == 0 and info["shared_nnz"] < 0.5 * a.nnz
    import scipy.sparse as sp
    wide = problems.Csr.from_scipy(sp.random(400, 400, density=0.2, random_state=1, format="csr"))   # ~80 per row
    assert not solver.host_stream_plan_short(wide, 8)["ok"]


def test_row_blocks_from_support_points():
    """alfd_host_row_blocks_from_points: a partition of the rows into spatially compact blocks of at most
    max_rows rows, usable as alfd_set_row_blocks input without grid metadata (window well below that of
    runs of the numbering)."""
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import problems
    pb = problems.generate(dim=3, degree=2, ncomp=3, n_cells=10, stokes=False, grad_div=True,
                           gamma_grad_div=10.0, radius=0.1, immersed_refine=0)
    a = pb.mats["A"]
    pts = problems.row_support_points(pb.params)
    assert pts.shape == (a.nrows, 3)
    ptr, rows = solver.row_blocks_from_points(pts, 192)
    assert ptr[0] == 0 and ptr[-1] == a.nrows and np.all(np.diff(ptr) > 0) and np.max(np.diff(ptr)) <= 192
    assert np.array_equal(np.sort(rows), np.arange(a.nrows))
    for b in (0, len(ptr) // 2, len(ptr) - 2):                       # rows of a block are close in space
        ext = np.ptp(pts[rows[ptr[b]:ptr[b + 1]]], axis=0)
        assert np.all(ext <= 0.55), ext
    rcb = solver.host_stream_plan(a, blocks=(ptr, rows))
    runs = solver.host_stream_plan(a)
    assert rcb["ok"] and rcb["decode_mismatches"] == 0 and rcb["rows_covered"] == a.nrows
    assert rcb["max_window"] < runs["max_window"]
    # 2-D points, tiny input, one block
    p2 = np.stack([np.arange(7.0), np.zeros(7)], axis=1)
    ptr, rows = solver.row_blocks_from_points(p2, 250)
    assert list(ptr) == [0, 7] and sorted(rows) == list(range(7))


def test_host_window_plan_decodes_back():
    """The LDS-window / value-indexed storage planned on the host (what alfd_set_matrix uploads)
    decodes back to the CSR it was made from: window columns, dictionary values (8-bit, 16-bit,
    raw blocks) and the class-sorted row batches -- no GPU involved."""
    import numpy as np
    from fictitious_domain_al_preconditioners_amd import problems
    pb = problems.generate(dim=3, degree=2, ncomp=3, n_cells=6, stokes=False, grad_div=True,
                           gamma_grad_div=10.0, radius=0.1, immersed_refine=0)
    a = pb.mats["A"]                                         # 6591 rows, long rows, few distinct values
    info = solver.host_window_plan(a, 64, True)
    assert info["windowed"] and info["value_indexed"] and info["decode_mismatches"] == 0
    assert info["row_block"] == 96 and info["blocks"] == -(-a.nrows // 96)
    assert info["value_indexed_nnz"] == a.nnz and info["value_wide_nnz"] == 0 and info["batches"] > 0
    assert info["dictionary_entries"] <= 256 * info["blocks"]
    rng = np.random.default_rng(3)
    # 16-bit codes: every value scaled by one of 9 factors
    v = np.array(a.val) * (1.0 + 0.125 * rng.integers(0, 9, a.nnz))
    wide = problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), v)
    info = solver.host_window_plan(wide, 64, True)
    assert info["value_indexed"] and info["value_wide_nnz"] > 0 and info["decode_mismatches"] == 0
    # raw blocks in the first third of the rows
    v = np.array(a.val)
    cut = int(a.row_ptr[a.nrows // 3])
    v[:cut] = rng.uniform(-1, 1, cut)
    raw = problems.Csr(a.nrows, a.ncols, np.array(a.row_ptr), np.array(a.col), v)
    info = solver.host_window_plan(raw, 64, True)
    assert info["value_indexed"] and 0 < info["value_indexed_blocks"] < info["blocks"]
    assert info["decode_mismatches"] == 0
    # no dictionary requested / random values: plain window format
    info = solver.host_window_plan(a, 64, False)
    assert info["windowed"] and not info["value_indexed"] and info["decode_mismatches"] == 0
    # short rows (27-point stencil, 32 lanes): window format with larger row blocks
    q1 = problems.generate(dim=3, degree=1, ncomp=1, n_cells=20, radius=0.1).mats["A"]
    info = solver.host_window_plan(q1, 32, True)
    assert info["windowed"] and not info["value_indexed"] and info["row_block"] == 384
    assert info["decode_mismatches"] == 0
    # ragged rows, empty rows and far-away columns: some blocks fall back, the rest still decode
    import scipy.sparse as sp
    n = 20000
    cnt = rng.integers(0, 200, n)
    rows = np.repeat(np.arange(n), cnt)
    cols = np.clip(rows + rng.integers(-300, 301, rows.size), 0, n - 1)
    far = rng.integers(0, rows.size, 50)
    cols[far] = rng.integers(0, n, 50)
    m = sp.csr_matrix((rng.integers(1, 6, rows.size).astype(float), (rows, cols)), shape=(n, n))
    m.sum_duplicates()
    info = solver.host_window_plan(problems.Csr.from_scipy(m), 64, True)
    assert info["decode_mismatches"] == 0 and info["blocks"] == -(-n // 96)
