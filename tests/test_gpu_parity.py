"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle.

Bar: the canonical arithmetic makes the kernels
bit-reproducible, so primitives are compared bit for bit; whole solves must
give identical iteration counts and residual histories within 1e-10 relative
(the tolerance north_star states)."""
import numpy as np
import pytest

from fictitious_domain_al_preconditioners_amd import _abi, problems, solver
from oracle import oracle

pytestmark = pytest.mark.gpu
HIST_RTOL = 1e-10


def _rng_vec(n, seed):
    return np.random.default_rng(seed).uniform(-1.0, 1.0, n)


@pytest.fixture(scope="module")
def stokes_small(built):
    pb = problems.stokes3d_sphere(n_cells=8, immersed_refine=0)
    cfg = _abi.default_config(_abi.AL_STOKES)
    cfg.inner.max_steps = 1000
    return pb, cfg, solver.context_from_problem(pb, cfg), oracle.system_from_problem(pb)


@pytest.fixture(scope="module")
def laplace_small(built):
    pb = problems.laplace2d_circle(64, 4)
    cfg = _abi.default_config(_abi.AL2)
    cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
    cfg.inner.max_steps = 1000
    return pb, cfg, solver.context_from_problem(pb, cfg), oracle.system_from_problem(pb)


@pytest.mark.parametrize("name", ["A", "Bt", "B", "Ct", "C", "Mp"])
@pytest.mark.parametrize("mode", [0, 1])
def test_spmv_bitwise(stokes_small, name, mode):
    pb, cfg, ctx, _ = stokes_small
    m = pb.mats[name]
    x = _rng_vec(m.ncols, 1)
    y0 = _rng_vec(m.nrows, 2)
    got, lanes = ctx.spmv(_abi.SLOT_BY_NAME[name], x, y0, mode=mode, alpha=-0.75)
    ref, olanes = oracle.spmv(m, x, y0 if mode else None, mode=mode, alpha=-0.75)
    assert lanes == olanes
    assert np.array_equal(got, ref)
    # and the oracle itself against SciPy (independent summation order)
    sp = m.to_scipy() @ x
    exp = sp if mode == 0 else y0 - 0.75 * sp
    assert np.allclose(ref, exp, rtol=1e-12, atol=1e-12 * np.abs(exp).max())


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 100003, 1 << 20])
def test_dot_bitwise(stokes_small, n):
    _, _, ctx, _ = stokes_small
    x, y = _rng_vec(n, 3), _rng_vec(n, 4)
    got = ctx.dot(x, y)
    assert got == oracle.dot(x, y)
    assert abs(got - float(np.dot(x, y))) <= 1e-12 * max(1.0, np.abs(x * y).sum())


@pytest.mark.parametrize("fix", ["stokes_small", "laplace_small"])
def test_system_and_rhs_bitwise(fix, request):
    pb, cfg, ctx, osys = request.getfixturevalue(fix)
    src = [_rng_vec(n, 10 + i) for i, n in enumerate(pb.block_sizes)]
    got = ctx.system_apply(src)
    rc, ref = osys.system_apply(cfg, src)
    assert rc == 0
    for g, r in zip(got, ref):
        assert np.array_equal(g, r)
    rhs = [pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]] if len(pb.block_sizes) == 3 else [pb.vecs["f"], pb.vecs["g"]]
    rc, oref = osys.augment_rhs(cfg, rhs)
    for g, r in zip(ctx.augment_rhs(rhs), oref):
        assert np.array_equal(g, r)


@pytest.mark.parametrize("fix", ["stokes_small", "laplace_small"])
def test_precond_vmult_parity(fix, request):
    pb, cfg, ctx, osys = request.getfixturevalue(fix)
    src = [_rng_vec(n, 20 + i) for i, n in enumerate(pb.block_sizes)]
    got, res = ctx.precond_apply(src)
    rc, ref, ores = osys.precond_apply(cfg, src)
    assert rc == 0
    assert res.inner_iterations == ores.inner_iterations
    assert res.mp_iterations == ores.mp_iterations
    for g, r in zip(got, ref):
        assert np.allclose(g, r, rtol=1e-10, atol=1e-10 * np.abs(r).max())


@pytest.mark.parametrize("fix", ["stokes_small", "laplace_small"])
def test_solve_iteration_counts_and_history(fix, request):
    pb, cfg, ctx, osys = request.getfixturevalue(fix)
    rhs = [pb.vecs["f"], pb.vecs["rhs_p"], pb.vecs["g"]] if len(pb.block_sizes) == 3 else [pb.vecs["f"], pb.vecs["g"]]
    rhs = ctx.augment_rhs(rhs)
    x, res = ctx.solve(rhs)
    hist = ctx.history()
    rc, ox, ores, ohist = osys.solve(cfg, rhs)
    assert rc == 0 and res.status == 0
    assert res.outer_iterations == ores.outer_iterations
    assert res.inner_iterations == ores.inner_iterations
    assert res.mp_iterations == ores.mp_iterations
    assert len(hist) == len(ohist)
    assert np.max(np.abs(hist - ohist) / np.abs(ohist)) <= HIST_RTOL
    for g, r in zip(x, ox):
        assert np.allclose(g, r, rtol=1e-9, atol=1e-10 * max(np.abs(r).max(), 1e-30))
    # the GPU solution really solves the system (independent of the oracle)
    ax = ctx.system_apply(x)
    r = np.concatenate([a - b for a, b in zip(rhs, ax)])
    assert np.linalg.norm(r) <= 10 * max(cfg.outer.tol, cfg.outer.reduce * res.initial_residual)
