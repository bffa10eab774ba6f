"""Shared small test cases (seeded, sizes the oracle finishes in seconds)."""
import numpy as np

from fictitious_domain_al_preconditioners_amd import _abi, problems


def rhs_of(pb):
    if "A2" in pb.mats:   # elliptic_interface: last row of the rhs is 0 (elliptic_interface.cc:903)
        return [pb.vecs["f"].copy(), pb.vecs["f2"].copy(), np.zeros(pb.block_sizes[2])]
    if "B" in pb.mats:
        return [pb.vecs["f"].copy(), pb.vecs["rhs_p"].copy(), pb.vecs["g"].copy()]
    return [pb.vecs["f"].copy(), pb.vecs["g"].copy()]


def rng_blocks(pb, seed):
    rng = np.random.default_rng(seed)
    return [rng.uniform(-1.0, 1.0, n) for n in pb.block_sizes]


def case(name):
    """name -> (problem, config).  Solver knobs follow the reference prms; the
    inner cap is raised because the ML-AMG preconditioner is replaced by the
    Chebyshev/Jacobi sweep."""
    if name == "laplace2d_circle":          # BASELINE cfg 1 (immersed_laplace 2-D, circle, Q1)
        pb = problems.laplace2d_circle(64, 4)
        cfg = _abi.default_config(_abi.AL2)
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)  # Circle_parameters_f0_g1.prm:40-42
    elif name == "laplace2d_jacobi":
        pb = problems.laplace2d_circle(32, 3)
        cfg = _abi.default_config(_abi.AL2)
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        cfg.inner_prec = _abi.PREC_JACOBI
        cfg.orthogonalization = _abi.ORTH_MGS
    elif name == "laplace3d_sphere":        # BASELINE cfg 2, scaled down
        pb = problems.laplace3d_sphere(16, 1)
        cfg = _abi.default_config(_abi.AL2)
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        cfg.orthogonalization = _abi.ORTH_CGS
    elif name == "stokes2d_circle":         # what the reference binary is compiled for (dim 1 in 2)
        pb = problems.stokes2d_circle(16, 3)
        cfg = _abi.default_config(_abi.AL_STOKES)
    elif name == "stokes3d_sphere":         # BASELINE cfg 4 (north star), scaled down
        pb = problems.stokes3d_sphere(8, 0)
        cfg = _abi.default_config(_abi.AL_STOKES)
    elif name == "stokes3d_restart":        # forces FGMRES restarts and the identity inner preconditioner
        pb = problems.stokes3d_sphere(6, 0)
        cfg = _abi.default_config(_abi.AL_STOKES)
        cfg.restart = 3
        cfg.inner_prec = _abi.PREC_IDENTITY
        cfg.inner.max_steps = 5000
    elif name in ("elliptic_modified", "elliptic_ideal", "elliptic_modified_jump1e3"):
        # BASELINE cfg 3: elliptic_interface 2-D, parameters_modified.prm / parameters_ideal.prm
        beta2 = 1e3 if name.endswith("1e3") else 10.0      # prm:5 says 10, BASELINE.json says 1e3
        pb = problems.elliptic_interface2d(64, 16, beta2=beta2)
        ideal = name == "elliptic_ideal"
        cfg = _abi.default_config(_abi.AL_ELL_IDEAL if ideal else _abi.AL_ELL_MODIFIED)
        cfg.gamma, cfg.gamma2 = 10.0, (10.0 if ideal else 1e-2)            # prm:48-49
        cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)  # prm:59-66
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-10)   # prm:76-83
    elif name == "rational_minres":
        # immersed_laplace "rational" branch: MinRes + RationalPreconditioner (immersed_laplace.cc:585-631)
        pb = problems.laplace2d_circle(32, 3)
        cfg = _abi.default_config(_abi.RATIONAL)
        cfg.rho_bound = pb.rho_bound()
        cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 5000, 1e-13, 1e-12)   # K_inv: UMFPACK in the reference
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)   # Schur solver control
    elif name == "stokes_minres_diag":
        # stokes...:1056-1064: MinRes + BlockPreconditionerAugmentedLagrangianDiagonal
        pb = problems.stokes3d_sphere(6, 0)
        cfg = _abi.default_config(_abi.AL_STOKES_DIAG)
        cfg.outer_solver = _abi.OUTER_MINRES
        cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 5000, 1e-12, 1e-10)   # MinRes wants a linear SPD preconditioner
        cfg.mp_inner = _abi.Control(_abi.CTRL_REDUCTION, 500, 1e-13, 1e-11)
    elif name == "laplace2d_operator_form":
        # immersed_laplace "Use operator version = true" + "Use diagonal inverse = true"
        # (immersed_laplace.cc:653-705, 855-858): AL term assembled into A, gamma = 10/h, W^-1 = 1/M_ii
        pb = problems.laplace2d_circle(64, 4, surface_mass=True)
        a_op, gamma_h, inv_w = problems.operator_form(pb)
        pb.mats = dict(pb.mats, A=a_op)
        pb.inv_w_override = inv_w
        cfg = _abi.default_config(_abi.AL2)
        cfg.gamma, cfg.aug_assembled = gamma_h, 1
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
    elif name in ("laplace2d_exact_w", "stokes2d_exact_w", "laplace2d_operator_form_exact_w"):
        # `Use diagonal inverse = false`: W^-1 = (M^-1)^2, or M^-1 in operator
        # form (immersed_laplace.cc:859-877, stokes...:979-985; UMFPACK there, CG on M here).  The inner
        # preconditioner keeps the diagonal weight, as the reference's AMG does.
        if name.startswith("stokes"):
            pb = problems.stokes2d_circle(16, 3)            # parameters_stokes.prm:22
            cfg = _abi.default_config(_abi.AL_STOKES)
            cfg.w_inverse = _abi.W_MASS_INV_SQUARED
        elif name == "laplace2d_exact_w":
            pb = problems.laplace2d_circle(32, 3)           # Circle_parameters_f0_g1.prm:11-14
            cfg = _abi.default_config(_abi.AL2)
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
            cfg.w_inverse = _abi.W_MASS_INV_SQUARED
        else:
            pb = problems.laplace2d_circle(32, 3, surface_mass=True)
            a_op, gamma_h, inv_w = problems.operator_form(pb)
            pb.mats = dict(pb.mats, A=a_op)
            pb.inv_w_override = inv_w
            cfg = _abi.default_config(_abi.AL2)
            cfg.gamma, cfg.aug_assembled = gamma_h, 1
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
            cfg.w_inverse = _abi.W_MASS_INV
    elif name in ("elliptic_modified_exact_w", "elliptic_ideal_exact_w"):
        # elliptic_interface with `Use diagonal inverse = false`: invW = M^-1 M^-1 (elliptic...:733-737;
        # parameters_ideal.prm:38)
        pb = problems.elliptic_interface2d(32, 8)
        ideal = name.startswith("elliptic_ideal")
        cfg = _abi.default_config(_abi.AL_ELL_IDEAL if ideal else _abi.AL_ELL_MODIFIED)
        cfg.gamma, cfg.gamma2 = 10.0, (10.0 if ideal else 1e-2)
        cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-10)
        cfg.w_inverse = _abi.W_MASS_INV_SQUARED
    elif name in ("stokes3d_multilevel", "laplace3d_multilevel", "elliptic_modified_multilevel"):
        # aggregation-multigrid inner preconditioner (ML in the reference)
        if name.startswith("stokes"):
            pb = problems.stokes3d_sphere(8, 0)
            cfg = _abi.default_config(_abi.AL_STOKES)
        elif name.startswith("laplace"):
            pb = problems.laplace3d_sphere(16, 1)
            cfg = _abi.default_config(_abi.AL2)
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        else:
            pb = problems.elliptic_interface2d(64, 16)
            cfg = _abi.default_config(_abi.AL_ELL_MODIFIED)
            cfg.gamma, cfg.gamma2 = 10.0, 1e-2
            cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-10)
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 2, 8.0
    elif name in ("elasticity_modified", "elasticity_modified_multilevel"):
        # BASELINE cfg 5: elliptic_interface 3-D elasticity, parameters_elliptic_interface/elasticity.prm,
        # scaled down: modified AL (prm:43), gamma 10 / 1e-2 (prm:49-50), exact W^-1 = (M^-1)^2
        # (`Use diagonal inverse = false`, prm:39), inner ReductionControl abs 1e-2 (prm:60-66), outer
        # reduction 1e-6 (prm:77-84)
        ml = name.endswith("multilevel")
        pb = problems.elasticity3d(16 if ml else 10)
        cfg = _abi.default_config(_abi.AL_ELL_MODIFIED)
        cfg.gamma, cfg.gamma2 = 10.0, 1e-2
        cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 10000, 1e-2, 1e-20)
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-6)
        if ml:      # diagonal W^-1 = 1/(M^2)_ii + the aggregation multigrid on the 3-component background
            cfg.inner_prec = _abi.PREC_MULTILEVEL
            cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 2, 8.0
        else:
            cfg.w_inverse = _abi.W_MASS_INV_SQUARED
    elif name == "laplace2d_operator_form_gmg_patch":
        # operator form (AL term assembled into A, immersed_laplace.cc:653-705) + the round-3 multigrid: the patch
        # operator is A[S,S] alone (aug_assembled), S still the rows the coupling matrix touches
        pb = problems.laplace2d_circle(64, 4, surface_mass=True)
        a_op, gamma_h, inv_w = problems.operator_form(pb)
        pb.mats = dict(pb.mats, A=a_op)
        pb.inv_w_override = inv_w
        cfg = _abi.default_config(_abi.AL2)
        cfg.gamma, cfg.aug_assembled = gamma_h, 1
        cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_degree_coarse, cfg.ml_smooth_ratio = 2, 3, 20.0
        cfg.ml_patch_degree, cfg.ml_patch_ratio, cfg.ml_coarse_direct = 6, 50.0, 1024
        cfg.inner.max_steps = 100
        return pb, cfg
    elif name in ("stokes3d_gmg_patch", "stokes3d_gmg", "laplace3d_gmg_patch", "elliptic_modified_gmg_patch"):
        # round 3: geometric multigrid through CSR prolongators (alfd_set_prolongator: Q2 -> Q1 embedding,
        # then (bi/tri)linear interpolation), the interface-patch corrections around the V-cycle and the
        # explicit inverse on the coarsest level -- the counterpart of ML's smoothed prolongators + KLU
        # (utilities.h:304-317).  "stokes3d_gmg_patch" carries bench.py's settings.
        if name.startswith("stokes"):
            pb = problems.stokes3d_sphere(8, 1)
            cfg = _abi.default_config(_abi.AL_STOKES)
        elif name.startswith("laplace"):
            pb = problems.laplace3d_sphere(16, 1)
            cfg = _abi.default_config(_abi.AL2)
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-12)
        else:
            pb = problems.elliptic_interface2d(64, 16)
            cfg = _abi.default_config(_abi.AL_ELL_MODIFIED)
            cfg.gamma, cfg.gamma2 = 10.0, 1e-2
            cfg.inner = _abi.Control(_abi.CTRL_REDUCTION, 100000, 1e-2, 1e-20)
            cfg.outer = _abi.Control(_abi.CTRL_REDUCTION, 1000, 1e-10, 1e-10)
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_ratio = 4, 30.0
        if name == "stokes3d_gmg":              # no patch, Chebyshev sweep on the coarsest level
            cfg.ml_coarse_degree, cfg.ml_coarse_ratio, cfg.ml_coarse_direct = 12, 100.0, -1
        else:
            cfg.ml_patch_degree, cfg.ml_patch_ratio, cfg.ml_coarse_direct = 5, 30.0, 1024
        cfg.inner.max_steps = 100
        return pb, cfg
    elif name == "stokes3d_fgmres95":
        # deal.II <= 9.5 SolverFGMRES loop (alfd_fgmres_flavour): MGS, delayed least-squares check; restart 5
        # forces several cycles so that the per-cycle counting rule (m - 1 steps per m applications) shows
        pb = problems.stokes3d_sphere(6, 0)
        cfg = _abi.default_config(_abi.AL_STOKES)
        cfg.fgmres_flavour = _abi.FGMRES_DEALII_95
        cfg.restart = 5
    elif name == "stokes3d_bench_settings":
        # exactly bench.py's solver settings (multigrid: Chebyshev(4) over [lmax/256, lmax], Chebyshev(10)
        # coarsest solve, geometric aggregates a = 2 / min_coarse 4000, inner cap 100 = prm:23) at small N
        pb = problems.stokes3d_sphere(8, 1)
        cfg = _abi.default_config(_abi.AL_STOKES)
        cfg.inner_prec = _abi.PREC_MULTILEVEL
        cfg.ml_smooth_degree, cfg.ml_smooth_ratio, cfg.ml_coarse_degree = 4, 256.0, 10
        cfg.inner.max_steps = 100
        return pb, cfg
    else:
        raise KeyError(name)
    cfg.inner.max_steps = max(cfg.inner.max_steps, 1000)
    return pb, cfg


def aggregates_of(pb, cfg):
    """Aggregates handed to both the library and the oracle for ALFD_PREC_MULTILEVEL."""
    if cfg.inner_prec != _abi.PREC_MULTILEVEL:
        return None
    if cfg.ml_coarse_direct != 0:       # the *_gmg* cases: CSR prolongators of the tensor grid
        return problems.tensor_prolongators(pb.params, min_coarse=100)
    if cfg.ml_coarse_degree == 10:      # stokes3d_bench_settings: bench.py's --agg-a 2 --min-coarse 4000
        return problems.geometric_aggregates(pb, a=2, min_coarse=4000)
    return problems.geometric_aggregates(pb, a=2, min_coarse=100)


ALL_CASES = ["laplace2d_circle", "laplace2d_jacobi", "laplace3d_sphere", "stokes2d_circle", "stokes3d_sphere",
             "stokes3d_restart", "elliptic_modified", "elliptic_ideal", "elliptic_modified_jump1e3",
             "rational_minres", "stokes_minres_diag", "stokes3d_multilevel", "laplace3d_multilevel",
             "elliptic_modified_multilevel", "laplace2d_operator_form", "laplace2d_exact_w", "stokes2d_exact_w",
             "laplace2d_operator_form_exact_w", "elliptic_modified_exact_w", "elliptic_ideal_exact_w",
             "elasticity_modified", "elasticity_modified_multilevel", "stokes3d_bench_settings",
             "stokes3d_fgmres95", "stokes3d_gmg_patch", "stokes3d_gmg", "laplace3d_gmg_patch",
             "elliptic_modified_gmg_patch", "laplace2d_operator_form_gmg_patch"]


def oracle_system(pb, cfg):
    from oracle import oracle
    if cfg.variant == _abi.RATIONAL:
        return oracle.rational_system_from_problem(pb)
    return oracle.system_from_problem(pb, aggregates=aggregates_of(pb, cfg))


def prepared_rhs(osys, pb, cfg):
    """The right-hand side the reference hands to the Krylov solver: augmented for the
    AL variants of immersed_laplace / stokes (stokes...:1012-1018), plain otherwise."""
    rhs = rhs_of(pb)
    if cfg.variant in (_abi.AL2, _abi.AL_STOKES, _abi.AL_STOKES_DIAG):
        rc, rhs = osys.augment_rhs(cfg, rhs)
        assert rc == 0
    return rhs


def hanging_node_variant(pb, plane=None):
    """What deal.II's AffineConstraints::condense leaves behind for one layer of hanging nodes
    (stokes_immersed_boundary.cc:468-482 refines locally around the immersed body): every second node
    of one x-plane of the Taylor-Hood velocity grid is constrained to the mean of its two y-neighbours,
    u_h = (u_a + u_b) / 2.  With T = the interpolation from the unconstrained dofs, the condensed
    operators are T^T A T (+ identity on the constrained rows), T^T Bt, B T, T^T Ct, C T, T^T f.
    The result has NO tensor-grid structure to offer: rows next to the plane carry new values and
    constrained rows hold a lone diagonal -- the input for the algebraic aggregator."""
    import scipy.sparse as sp
    P = pb.params
    assert P["dim"] == 3 and P["degree"] == 2 and P["ncomp"] == 3
    n1 = 2 * P["n_cells"] + 1
    plane = n1 // 2 if plane is None else plane
    n = pb.mats["A"].nrows
    rows, cols, vals = [], [], []
    constrained = np.zeros(n, bool)
    for k in range(2, n1 - 2):
        for j in range(3, n1 - 3, 2):                     # odd interior y-index: hanging
            h = (k * n1 + j) * n1 + plane
            a, b = h - n1, h + n1
            for c in range(3):
                constrained[3 * h + c] = True
                rows += [3 * h + c, 3 * h + c]
                cols += [3 * a + c, 3 * b + c]
                vals += [0.5, 0.5]
    free = np.nonzero(~constrained)[0]
    T = sp.csr_matrix((np.concatenate([np.ones(free.size), vals]),
                       (np.concatenate([free, rows]), np.concatenate([free, cols]))), shape=(n, n))
    Dc = sp.diags(constrained.astype(float))
    mats = dict(pb.mats)
    A = (T.T @ pb.mats["A"].to_scipy() @ T + Dc).tocsr()
    A.eliminate_zeros()
    mats["A"] = problems.Csr.from_scipy(A)
    for name in ("Bt", "Ct"):
        m = (T.T @ pb.mats[name].to_scipy()).tocsr()
        mats[name] = problems.Csr.from_scipy(m)
    mats["B"], mats["C"] = mats["Bt"].transpose(), mats["Ct"].transpose()
    vecs = dict(pb.vecs)
    vecs["f"] = T.T @ pb.vecs["f"]
    out = problems.SyntheticProblem(params=dict(pb.params), mats=mats, vecs=vecs)
    out.n_constrained = int(constrained.sum())
    return out
