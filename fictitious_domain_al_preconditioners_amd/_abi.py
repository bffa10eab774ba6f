"""ctypes mirror of include/alfd/alfd.h (structs and enums only)."""
import ctypes as C

ALFD_MAX_BLOCKS = 3

# enum alfd_status
OK, E_INVALID, E_HIP, E_NO_CONVERGENCE_OUTER, E_NO_CONVERGENCE_INNER, E_BREAKDOWN, E_NOT_SETUP, \
    E_COMM, E_UNSUPPORTED = range(9)
# enum alfd_matrix_slot
A, BT, B, CT, C_, M, MP, A2, KIMM = range(9)
NSLOTS = 9
SLOT_BY_NAME = {"A": A, "Bt": BT, "B": B, "Ct": CT, "C": C_, "M": M, "Mp": MP, "A2": A2, "K": KIMM}
# enum alfd_diag_slot
INVW, MP_LUMPED_INV = 0, 1
NDIAGS = 2
# enum alfd_variant
AL2, AL_STOKES, AL_STOKES_DIAG, AL_ELL_IDEAL, AL_ELL_MODIFIED, RATIONAL = range(6)
# enum alfd_control_kind
CTRL_ABS, CTRL_REDUCTION, CTRL_FIXED_ITERS = range(3)
# enum alfd_inner_prec
PREC_IDENTITY, PREC_JACOBI, PREC_CHEBYSHEV, PREC_MULTILEVEL = range(4)
# enum alfd_orthogonalization
ORTH_MGS, ORTH_CGS, ORTH_CGS2 = range(3)
# enum alfd_outer_solver
OUTER_FGMRES, OUTER_MINRES = range(2)
# enum alfd_fgmres_flavour
FGMRES_DEALII_96, FGMRES_DEALII_95 = range(2)
# enum alfd_w_inverse
W_DIAGONAL, W_MASS_INV_SQUARED, W_MASS_INV = range(3)
# enum alfd_inner_failure_policy
INNER_THROW, INNER_ACCEPT = range(2)
# enum alfd_timing_class
T_SPMV_A, T_SPMV_OTHER, T_DOT, T_VEC = range(4)
T_NCLASSES = 4
UNIQUE_ID_BYTES = 128


class Control(C.Structure):
    _fields_ = [("kind", C.c_int32), ("max_steps", C.c_int32), ("tol", C.c_double), ("reduce", C.c_double)]

    def __init__(self, kind=CTRL_ABS, max_steps=100, tol=1e-10, reduce=0.0):
        super().__init__(kind, max_steps, tol, reduce)


class Config(C.Structure):
    _fields_ = [
        ("variant", C.c_int32), ("restart", C.c_int32), ("orthogonalization", C.c_int32),
        ("grad_div_in_A", C.c_int32),
        ("gamma", C.c_double), ("gamma_grad_div", C.c_double), ("gamma2", C.c_double),
        ("outer", Control), ("inner", Control), ("mp_inner", Control),
        ("inner_prec", C.c_int32), ("cheb_degree", C.c_int32), ("cheb_power_its", C.c_int32),
        ("on_inner_failure", C.c_int32),
        ("cheb_eig_ratio", C.c_double), ("cheb_safety", C.c_double),
        ("log_level", C.c_int32), ("outer_solver", C.c_int32),
        ("rho_bound", C.c_double), ("rational", Control),
        ("ml_smooth_degree", C.c_int32), ("ml_coarse_degree", C.c_int32),
        ("ml_smooth_ratio", C.c_double), ("ml_coarse_ratio", C.c_double),
        ("aug_assembled", C.c_int32), ("w_inverse", C.c_int32),
        ("mass", Control),
        ("fgmres_flavour", C.c_int32), ("ml_smooth_degree_coarse", C.c_int32),
        ("ml_patch_degree", C.c_int32), ("ml_coarse_direct", C.c_int32), ("ml_patch_ratio", C.c_double),
    ]


class Result(C.Structure):
    _fields_ = [
        ("status", C.c_int32), ("outer_iterations", C.c_int32),
        ("initial_residual", C.c_double), ("last_residual", C.c_double),
        ("inner_iterations", C.c_int64), ("mp_iterations", C.c_int64),
        ("inner_failures", C.c_int32), ("precond_applications", C.c_int32),
        ("solve_seconds", C.c_double), ("lambda_max", C.c_double),
        ("rational_iterations", C.c_int64), ("mass_iterations", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class MatrixInfo(C.Structure):
    _fields_ = [
        ("lanes", C.c_int32), ("windowed", C.c_int32), ("value_indexed", C.c_int32), ("batch_major", C.c_int32),
        ("nnz", C.c_int64), ("window_blocks", C.c_int64), ("window_fallback_blocks", C.c_int64),
        ("value_indexed_blocks", C.c_int64), ("value_indexed_nnz", C.c_int64),
        ("dictionary_entries", C.c_int64), ("value_wide_nnz", C.c_int64),
        ("algorithmic_bytes", C.c_double), ("streamed_bytes", C.c_double),
        ("shared_nnz", C.c_int64), ("batch_major_blocks", C.c_int64), ("batch_major_wide", C.c_int64),
        ("batch_major_interior_blocks", C.c_int64),
    ]


class WindowPlanInfo(C.Structure):
    _fields_ = [
        ("windowed", C.c_int32), ("value_indexed", C.c_int32), ("row_block", C.c_int32), ("max_window", C.c_int32),
        ("blocks", C.c_int64), ("fallback_blocks", C.c_int64), ("segments", C.c_int64),
        ("value_indexed_blocks", C.c_int64), ("value_indexed_nnz", C.c_int64), ("value_wide_nnz", C.c_int64),
        ("dictionary_entries", C.c_int64), ("batches", C.c_int64), ("decode_mismatches", C.c_int64),
    ]


class StreamPlanInfo(C.Structure):
    _fields_ = [
        ("ok", C.c_int32), ("max_window", C.c_int32), ("max_rows", C.c_int32), ("max_batches", C.c_int32),
        ("blocks", C.c_int64), ("batches", C.c_int64), ("segments", C.c_int64), ("dictionary_entries", C.c_int64),
        ("stream_bytes", C.c_int64), ("decode_mismatches", C.c_int64), ("rows_covered", C.c_int64),
        ("shared_nnz", C.c_int64),
    ]


def default_config(variant=AL_STOKES) -> Config:
    """Same defaults as alfd_default_config(): the reference's solver knobs
    (parameters_stokes_3d.prm:17-24,150-157; immersed_laplace.cc:907; elliptic...:863)."""
    c = Config()
    c.variant = variant
    c.restart = 50 if variant in (AL_ELL_IDEAL, AL_ELL_MODIFIED) else 30
    c.orthogonalization = ORTH_CGS2
    c.grad_div_in_A = 1
    c.gamma, c.gamma_grad_div, c.gamma2 = 10.0, 10.0, 1e-2
    c.outer = Control(CTRL_REDUCTION, 1000, 1e-8, 1e-12)
    c.inner = Control(CTRL_ABS, 100, 1e-2, 0.0)
    c.mp_inner = Control(CTRL_ABS, 100, 1e-6, 0.0)
    c.inner_prec = PREC_CHEBYSHEV
    c.cheb_degree = 4
    c.cheb_power_its = 20
    c.on_inner_failure = INNER_THROW
    c.cheb_eig_ratio = 30.0
    c.cheb_safety = 1.2
    c.log_level = 0
    c.outer_solver = OUTER_MINRES if variant == RATIONAL else OUTER_FGMRES
    c.rho_bound = 0.0
    c.rational = Control(CTRL_ABS, 2000, 1e-14, 0.0)      # rational_preconditioner.h:34
    c.ml_smooth_degree, c.ml_coarse_degree = 3, 40
    c.ml_smooth_ratio, c.ml_coarse_ratio = 4.0, 400.0
    c.w_inverse = W_DIAGONAL
    c.mass = Control(CTRL_REDUCTION, 1000, 1e-30, 1e-14)
    c.ml_patch_degree, c.ml_coarse_direct, c.ml_patch_ratio = 0, 0, 30.0
    return c


def bench_multilevel_settings(cfg: Config, geometric: bool = True) -> Config:
    """The multigrid settings of bench.py's default run, in one place so that the full-size property test and the
    mid-size oracle parity case cannot drift from the benchmark: geometric hierarchy (CSR prolongators) with
    Chebyshev(3) / 40 smoothing on the fine level and degree 5 below, interface patch Chebyshev(15) / 200, explicit
    coarsest inverse up to 1024 unknowns; inner CG cap 100 as in parameters_stokes_3d.prm:23.  geometric = False:
    the round-2 aggregation hierarchy (Chebyshev(4) / 256, Chebyshev(10) coarsest sweep), still used on several ranks."""
    cfg.inner_prec = PREC_MULTILEVEL
    cfg.inner.max_steps = 100
    cfg.ml_coarse_degree = 10
    if geometric:
        cfg.ml_smooth_degree, cfg.ml_smooth_degree_coarse, cfg.ml_smooth_ratio = 3, 5, 40.0
        cfg.ml_patch_degree, cfg.ml_patch_ratio, cfg.ml_coarse_direct = 15, 200.0, 1024
    else:
        cfg.ml_smooth_degree, cfg.ml_smooth_degree_coarse, cfg.ml_smooth_ratio = 4, 0, 256.0
        cfg.ml_patch_degree, cfg.ml_coarse_direct = 0, 0
    return cfg


BENCH_MIN_COARSE = 1024     # tensor_prolongators(min_coarse=...) of the bench hierarchy
