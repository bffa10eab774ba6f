/* alfd.h -- C ABI of the MI355X-native augmented-Lagrangian FGMRES solver.
 *
 * This is the drop-in boundary.  The reference has no FFI:
 * its "operator API" is deal.II's duck-typed concept
 *     void vmult(BlockVector<double>& dst, const BlockVector<double>& src) const
 * implemented by the preconditioner classes of
 * augmented_lagrangian_preconditioner.h:28,62,95,130,186 and
 * rational_preconditioner.h:29, and consumed by
 *     SolverFGMRES<BlockVector<double>>::solve(AA, x, b, P)
 * (immersed_laplace.cc:943-944, stokes_immersed_boundary.cc:1073-1074,
 * elliptic_interface.cc:905-906, 947-948).  Each entry point below says which
 * of those it replaces.  include/alfd/dealii_adapter.hpp wraps this ABI back
 * into deal.II-shaped C++ classes; INTEGRATION.md shows the call-site diff.
 *
 * Conventions: every call returns an int status (never throws across the
 * ABI); all pointers at the ABI are HOST pointers -- device residency is
 * internal; a context is single-threaded (like the reference, which runs
 * MPI_InitFinalize(argc, argv, 1)); several contexts may coexist.
 * All floating point data is fp64; column indices are int32, row starts int64
 * (deal.II: unsigned int columns, std::size_t rowstart).
 */
#ifndef ALFD_H
#define ALFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ALFD_ABI_VERSION 12
#define ALFD_MAX_BLOCKS 3

/* ------------------------------------------------------------------ status */
enum alfd_status {
  ALFD_OK = 0,
  ALFD_E_INVALID = 1,              /* bad argument / inconsistent sizes */
  ALFD_E_HIP = 2,                  /* a HIP runtime call failed */
  ALFD_E_NO_CONVERGENCE_OUTER = 3, /* SolverControl::NoConvergence from FGMRES */
  ALFD_E_NO_CONVERGENCE_INNER = 4, /* ... from an inner CG (stokes...:1020-1024) */
  ALFD_E_BREAKDOWN = 5,            /* NaN / non-positive curvature */
  ALFD_E_NOT_SETUP = 6,
  ALFD_E_COMM = 7,                 /* RCCL failure */
  ALFD_E_UNSUPPORTED = 8
};

/* ---------------------------------------------------------------- operators
 * Matrix slots (square brackets: which reference object it is).
 *   A    [stokes_matrix.block(0,0) stokes...:923 | stiffness_matrix immersed_laplace.cc:638
 *         | stiffness_matrix_bg elliptic...:680]
 *   BT   [stokes_matrix.block(0,1) stokes...:924]   B [block(1,0) :925]
 *   CT   [coupling_matrix, n_u x n_lambda, stokes...:926, immersed_laplace.cc:640]
 *   C    optional (single rank): when it has not been uploaded explicitly, every
 *        alfd_set_matrix(ALFD_CT) also stores the transpose as C (the reference applies
 *        transpose_operator(Ct), i.e. SparseMatrix::Tvmult -- stokes...:927), so a
 *        re-upload of CT with new values keeps C consistent.  Same for BT -> B.
 *        Multi-rank: the C and B rows of a rank must be uploaded explicitly.
 *   M    [mass_matrix_immersed stokes...:928 | mass_matrix_fg elliptic...:682]
 *   MP   [preconditioner_matrix.block(1,1) stokes...:929]
 *   A2   [stiffness_matrix_fg elliptic...:681]
 *   KIMM [embedded_stiffness_matrix immersed_laplace.cc:639; rational_preconditioner.h:15]
 */
enum alfd_matrix_slot {
  ALFD_A = 0,
  ALFD_BT = 1,
  ALFD_B = 2,
  ALFD_CT = 3,
  ALFD_C = 4,
  ALFD_M = 5,
  ALFD_MP = 6,
  ALFD_A2 = 7,
  ALFD_KIMM = 8,
  ALFD_NSLOTS = 9
};

/* Diagonal operators.
 *   INVW            W^-1 as a vector: 1/M_ii^2 (stokes...:976-978,
 *                   immersed_laplace.cc:866-869), 1/M_ii (operator form, :856-858),
 *                   1/(M^2)_ii (utilities.h:348-374, elliptic...:726)
 *   MP_LUMPED_INV   1/(Mp 1)_i, the preconditioner of the pressure-mass CG
 *                   (stokes...:946-957)
 */
enum alfd_diag_slot { ALFD_INVW = 0, ALFD_MP_LUMPED_INV = 1, ALFD_NDIAGS = 2 };

/* Which preconditioner class / system operator pair is configured. */
enum alfd_variant {
  ALFD_AL2 = 0,            /* BlockPreconditionerAugmentedLagrangian, ...preconditioner.h:14-42;
                              AA = [[Aug,Ct],[C,0]] immersed_laplace.cc:891-892 */
  ALFD_AL_STOKES = 1,      /* ...Stokes, :44-79; AA 3x3 stokes...:1000-1003 */
  ALFD_AL_STOKES_DIAG = 2, /* ...Diagonal, :81-110 (SPD, for MinRes) */
  ALFD_AL_ELL_IDEAL = 3,   /* BlockTriangularALPreconditioner, :115-164 */
  ALFD_AL_ELL_MODIFIED = 4,/* ...Modified, :168-238; system elliptic...:816-819 */
  ALFD_RATIONAL = 5        /* RationalPreconditioner, rational_preconditioner.h:12-99 */
};

/* deal.II stop rules [EXT], SURVEY.md 8(a)-12.
 *   ABS          SolverControl:        success if r <= tol; failure if k >= max_steps or NaN
 *   REDUCTION    ReductionControl:     also success if r < reduce * r0
 *   FIXED_ITERS  IterationNumberControl: success if k >= max_steps (or r <= tol)
 */
enum alfd_control_kind { ALFD_CTRL_ABS = 0, ALFD_CTRL_REDUCTION = 1, ALFD_CTRL_FIXED_ITERS = 2 };

typedef struct alfd_control {
  int32_t kind;
  int32_t max_steps;
  double tol;
  double reduce;
} alfd_control;

/* Preconditioner of the inner CG on the augmented block.  The reference uses
 * Trilinos ML (stokes...:1027-1045); north_star replaces it by a Jacobi /
 * Chebyshev sweep. */
enum alfd_inner_prec {
  ALFD_PREC_IDENTITY = 0,
  ALFD_PREC_JACOBI = 1,
  ALFD_PREC_CHEBYSHEV = 2,
  /* Aggregation multigrid V-cycle with Chebyshev smoothing on every level (the
   * GPU counterpart of the ML smoothed-aggregation AMG the reference initialises
   * in utilities.h:304-317).  Needs alfd_set_aggregates(); applies to the
   * augmented (1,1) block, other inner operators fall back to CHEBYSHEV. */
  ALFD_PREC_MULTILEVEL = 3
};

/* Arnoldi orthogonalisation in FGMRES [EXT]: deal.II <= 9.5 modified
 * Gram-Schmidt; >= 9.6 classical Gram-Schmidt variants. */
enum alfd_orthogonalization { ALFD_ORTH_MGS = 0, ALFD_ORTH_CGS = 1, ALFD_ORTH_CGS2 = 2 };

/* Which deal.II release's SolverFGMRES loop is followed [EXT] (CMakeLists.txt:6 asks for 9.6.0, README.md:45
 * for 9.7): they differ in what last_step() counts.
 *   DEALII_96  >= 9.6: Arnoldi process with Givens rotations; the residual is checked and the counter
 *              incremented after EVERY Arnoldi step; the true residual of a restart is re-checked with the
 *              same counter.  Orthogonalisation: alfd_config::orthogonalization.
 *   DEALII_95  <= 9.5: modified Gram-Schmidt by add_and_dot; within a cycle the projected least-squares
 *              problem is solved with the FIRST j columns after the (j+1)-th Arnoldi vector was built
 *              (no check, no increment at j = 0), so a cycle of m preconditioner applications counts
 *              m - 1 steps and uses m - 1 of its m search directions.  Needs restart >= 2. */
enum alfd_fgmres_flavour { ALFD_FGMRES_DEALII_96 = 0, ALFD_FGMRES_DEALII_95 = 1 };

/* Outer Krylov method: SolverFGMRES (stokes...:1067) or SolverMinRes
 * (stokes...:1057-1064 with the diagonal SPD preconditioner; immersed_laplace.cc:629-631
 * with the rational preconditioner).  MinRes needs a symmetric system and an SPD
 * preconditioner, i.e. ALFD_AL_STOKES_DIAG or ALFD_RATIONAL. */
enum alfd_outer_solver { ALFD_OUTER_FGMRES = 0, ALFD_OUTER_MINRES = 1 };

/* What to do when an inner CG hits max_steps: the reference throws
 * SolverControl::NoConvergence (THROW); ACCEPT keeps the last iterate. */
enum alfd_inner_failure_policy { ALFD_INNER_THROW = 0, ALFD_INNER_ACCEPT = 1 };

/* The weight W^-1 of the AL term gamma Ct W^-1 C.  DIAGONAL: the caller's vector
 * (ALFD_INVW: 1/M_ii^2, or 1/M_ii in operator form) -- `Use diagonal inverse = true`.
 * MASS_INV_SQUARED / MASS_INV: the exact (M^-1)^2 / M^-1 of the reference's
 * `Use diagonal inverse = false` branch (immersed_laplace.cc:866-877, stokes...:979-985,
 * elliptic_interface.cc:713-737; UMFPACK there): every application runs Jacobi-preconditioned CG on the immersed mass
 * matrix (slot ALFD_M) to alfd_config::mass.  The inner preconditioner (Jacobi /
 * Chebyshev / multilevel) keeps using the diagonal weight, as the reference builds its
 * AMG from the diagonal form in either case (utilities.h:218-331). */
enum alfd_w_inverse { ALFD_W_DIAGONAL = 0, ALFD_W_MASS_INV_SQUARED = 1, ALFD_W_MASS_INV = 2 };

typedef struct alfd_config {
  int32_t variant;            /* enum alfd_variant */
  int32_t restart;            /* FGMRES max_basis_size: 30 default, 50 elliptic...:863 */
  int32_t orthogonalization;  /* enum alfd_orthogonalization */
  int32_t grad_div_in_A;      /* 1: A already holds gamma_gd (div,div) (stokes...:991-993) */
  double gamma;               /* AL parameter (gamma_1 for elliptic) */
  double gamma_grad_div;      /* stokes...:987 */
  double gamma2;              /* gamma_2, elliptic...:751-752 */
  alfd_control outer;         /* outer_solver_control, stokes...:385-389 */
  alfd_control inner;         /* control_lagrangian, stokes...:1020-1023 */
  alfd_control mp_inner;      /* control_mass(100, 1e-6), stokes...:934 */
  int32_t inner_prec;         /* enum alfd_inner_prec */
  int32_t cheb_degree;        /* polynomial degree k >= 1 */
  int32_t cheb_power_its;     /* power iterations for lambda_max(D^-1 Aug) */
  int32_t on_inner_failure;   /* enum alfd_inner_failure_policy */
  double cheb_eig_ratio;      /* lambda_min = lambda_max / ratio */
  double cheb_safety;         /* lambda_max *= safety (deal.II uses 1.2) */
  int32_t log_level;          /* 0 silent; 1 result lines; 2 per-iteration "Check" lines */
  int32_t outer_solver;       /* enum alfd_outer_solver */
  /* ALFD_RATIONAL only (rational_preconditioner.h): */
  double rho_bound;           /* ||A_Gamma||_inf / min_i M_ii, immersed_laplace.cc:609-614 */
  alfd_control rational;      /* SolverControl(2000, 1e-14) of the 21 immersed solves, :34 */
  /* ALFD_PREC_MULTILEVEL only (ML: smoother_sweeps = 2, utilities.h:312): */
  int32_t ml_smooth_degree;   /* Chebyshev degree of the pre-/post-smoother on every level */
  int32_t ml_coarse_degree;   /* Chebyshev degree that stands in for the coarsest-level solve */
  double ml_smooth_ratio;     /* smoother targets [lambda_max/ratio, lambda_max] */
  double ml_coarse_ratio;     /* same for the coarsest level */
  /* "operator form" (immersed_laplace.cc:653-705, 880-882; `Use operator version = true`):
   * the caller has assembled the AL term into A (gamma/h * int_Gamma phi_i phi_j), so
   * Aug = A and only the preconditioner and the rhs augmentation use gamma and invW. */
  int32_t aug_assembled;
  int32_t w_inverse;          /* enum alfd_w_inverse */
  alfd_control mass;          /* CG on M for the exact W^-1: ReductionControl(1000, 1e-30, 1e-14) */
  int32_t fgmres_flavour;     /* enum alfd_fgmres_flavour */
  int32_t ml_smooth_degree_coarse; /* > 0: smoother degree on levels >= 1 (ml_smooth_degree then applies to level 0
                                      only: the fine operator is the expensive one, the coarse ones are nearly free) */
  /* ALFD_PREC_MULTILEVEL, round 3 (ML in the reference: smoothed prolongators, 2 smoother sweeps, KLU on
   * the coarsest level -- utilities.h:304-317):
   * ml_patch_degree > 0 wraps the V-cycle symmetrically into two corrections on the INTERFACE PATCH, the
   * rows S of the (1,1) block the coupling matrix touches (non-empty rows of Ct): z1 = E q(Aug_SS) E^T r,
   * z2 = z1 + V(r - Aug z1), z = z2 + E q(Aug_SS) E^T (r - Aug z2), q = Chebyshev polynomial of that degree
   * in D^-1 Aug_SS over [lambda_max / ml_patch_ratio, lambda_max].  The penalty gamma Ct W^-1 C is what a
   * multigrid cycle on the background mesh cannot resolve; it lives on S only (a few 10^4 rows at 10^7 DoF).
   * ml_coarse_direct > 0: when the coarsest level has at most that many unknowns its operator is inverted
   * explicitly (dense Cholesky at setup, one dense product per cycle) instead of the Chebyshev sweep. */
  int32_t ml_patch_degree;
  int32_t ml_coarse_direct;
  double ml_patch_ratio;
} alfd_config;

typedef struct alfd_result {
  int32_t status;             /* enum alfd_status of the solve */
  int32_t outer_iterations;   /* SolverControl::last_step(), stokes...:1087 */
  double initial_residual;
  double last_residual;       /* SolverControl::last_value() */
  int64_t inner_iterations;   /* total CG iterations on the augmented block(s) */
  int64_t mp_iterations;      /* total CG iterations on Mp */
  int32_t inner_failures;     /* inner solves that hit max_steps (ACCEPT policy) */
  int32_t precond_applications;
  double solve_seconds;       /* wall time inside alfd_solve, device-synchronised */
  double lambda_max;          /* Chebyshev: estimated lambda_max(D^-1 Aug) incl. safety */
  int64_t rational_iterations;/* total CG iterations of the 21 immersed solves (ALFD_RATIONAL) */
  int64_t mass_iterations;    /* total CG iterations on M (exact W^-1 modes) */
} alfd_result;

typedef struct alfd_ctx *alfd_ctx_t;

/* ---------------------------------------------------------------- lifecycle */
int alfd_abi_version(void);
const char *alfd_strerror(int status);
/* Last error message of a context (HIP error strings etc.). */
const char *alfd_last_error(alfd_ctx_t ctx);

/* One context = one GPU = one rank.  device_id is the HIP ordinal. */
int alfd_create(alfd_ctx_t *ctx, int device_id);
int alfd_destroy(alfd_ctx_t ctx);

/* ------------------------------------------------------- multi-GPU (RCCL)
 * One process per GPU.  Rank 0 obtains an id with alfd_comm_unique_id(), the
 * host launcher broadcasts the bytes (bench.py uses torch.distributed), every
 * rank calls alfd_comm_init().  Without it the context is single-rank.
 * Row partition: every block b of the block vectors is split in contiguous
 * row ranges; offsets[b] has nranks+1 entries (global prefix).  Must be set
 * before matrices are uploaded.  Matrices are then uploaded as LOCAL rows
 * with GLOBAL column indices. */
#define ALFD_UNIQUE_ID_BYTES 128
int alfd_comm_unique_id(void *id_out, size_t bytes);
int alfd_comm_init(alfd_ctx_t ctx, int rank, int nranks, const void *id, size_t bytes);
int alfd_set_partition(alfd_ctx_t ctx, int nblocks, const int64_t *const *offsets /*[nblocks][nranks+1]*/);

/* In-process rank group: N contexts of ONE process (one host thread per rank,
 * typically all on one GPU) exchange through device-to-device copies and a host
 * barrier instead of RCCL.  Test vehicle for the multi-rank path on a single-GPU
 * box; collective calls (alfd_set_matrix, alfd_setup, alfd_solve, ...) must
 * then be issued by all ranks concurrently, one thread each. */
typedef struct alfd_local_group alfd_local_group;
int alfd_local_group_create(int nranks, alfd_local_group **group);
int alfd_local_group_destroy(alfd_local_group *group);
int alfd_comm_init_local(alfd_ctx_t ctx, alfd_local_group *group, int rank);

/* Host-transport rank group: the collectives of the row-partitioned path (one all-gather of a few
 * scalars per reduction, one personalised neighbour exchange per halo) are handed to the caller as
 * HOST buffers -- the library copies device -> host, calls back, copies host -> device.  For
 * launchers whose ranks talk through MPI or gloo instead of RCCL (a deal.II program is an MPI
 * program), and the vehicle of the two-process GPU test.  Both callbacks return 0 on success.
 *   allgather: every rank contributes `bytes` bytes; recv holds nranks * bytes in rank order.
 *   alltoallv: rank r sends send[send_off[p] .. send_off[p+1]) (elements of elem_size bytes) to p and
 *              receives recv[recv_off[p] .. recv_off[p+1]) from p. */
typedef int (*alfd_host_allgather_fn)(void *user, const void *send, void *recv, size_t bytes);
typedef int (*alfd_host_alltoallv_fn)(void *user, const void *send, const int64_t *send_off, void *recv,
                                      const int64_t *recv_off, size_t elem_size);
int alfd_comm_init_host(alfd_ctx_t ctx, int rank, int nranks, alfd_host_allgather_fn allgather,
                        alfd_host_alltoallv_fn alltoallv, void *user);

/* Host-only halo plan of one row-partitioned matrix (no GPU, no communication):
 * rewrites the GLOBAL column indices of this rank's rows into the local index
 * space [owned columns | halo entries], lists the halo's global ids (sorted,
 * hence grouped by owning rank) and the receive prefix per owner
 * (recv_off[nranks+1]).  alfd_set_matrix() runs the same routine internally;
 * it is exported so that the partition logic can be tested on CPU ranks. */
int alfd_host_halo_plan(int64_t nnz, const int32_t *col, const int64_t *col_offsets /*[nranks+1]*/,
                        int nranks, int rank, int32_t *col_local /*[nnz]*/,
                        int32_t *halo_globals /*[halo_capacity] or NULL*/, int64_t halo_capacity,
                        int64_t *n_halo, int64_t *recv_off /*[nranks+1]*/);

/* ------------------------------------------------------------------- upload
 * Replaces: linear_operator(SparseMatrix) captures, stokes...:923-929.
 * Caller keeps ownership of host arrays; the library copies to HBM.
 * Rows in CSR order, columns ascending within a row (the adapter converts
 * deal.II's diagonal-first order).  nrows = local rows, ncols = GLOBAL columns. */
int alfd_set_matrix(alfd_ctx_t ctx, int slot, int64_t nrows, int64_t ncols, const int64_t *row_ptr,
                    const int32_t *col, const double *val);
/* Replaces: DiagonalMatrix<Vector<double>> (stokes...:954, 980). n = local length. */
int alfd_set_diag(alfd_ctx_t ctx, int slot, int64_t n, const double *d);
#define ALFD_MAX_LEVELS 8
/* Aggregates of the multilevel inner preconditioner (replaces ML's aggregation,
 * utilities.h:304-317): level l maps the n_fine unknowns of level l (level 0 = block 0,
 * the velocity / background space) onto n_coarse unknowns of level l+1.  agg[i] in
 * [0, n_coarse) or -1 (not represented on the coarse level, e.g. Dirichlet rows);
 * weight[i] (NULL = 1) is the prolongation entry P_{i,agg[i]} -- pass the near-nullspace
 * vector for non-constant modes.  Coarse operators are the Galerkin products
 * P^T (A + gamma Ct invW C) P, kept factored as (P^T A P) + gamma (C P)^T invW (C P). */
int alfd_set_aggregates(alfd_ctx_t ctx, int level, int64_t n_fine, const int32_t *agg, const double *weight,
                        int64_t n_coarse);
/* General prolongator of level l as a CSR matrix P (n_fine x n_coarse, columns ascending per row):
 * geometric multigrid transfers (e.g. the embedding of Q1 into Q2 on the same mesh followed by trilinear
 * interpolation between nested or non-nested grids -- what deal.II's MGTransfer / FETools interpolation
 * matrices hold), or a smoothed-aggregation prolongator as ML builds it (utilities.h:304-317).  Replaces
 * the aggregates of that level; rows without entries (Dirichlet / constrained unknowns) are not
 * represented on the coarse level.  Coarse operators are the Galerkin products, formed in two steps,
 * (A P) then P^T (A P), each output entry a sequential fma chain in CSR order.
 * Partitioned contexts: level 0 takes the rows of P this rank owns (n_fine = its rows, GLOBAL coarse columns)
 * together with alfd_set_aggregate_partition(ctx, 0, coarse offsets by rank) -- the coarse unknowns whose fine
 * support lies in a rank's rows + halo of A; the prolongators of the levels below are handed over whole on every
 * rank.  The fine level stays partitioned, levels >= 1 (and the interface patch) are replicated; the hierarchy
 * does not depend on the partition.  Aggregates and prolongators cannot be mixed there. */
int alfd_set_prolongator(alfd_ctx_t ctx, int level, int64_t n_fine, int64_t n_coarse, const int64_t *row_ptr,
                         const int32_t *col, const double *val);
/* Multi-rank: agg[] holds this rank's unknowns of level l and GLOBAL coarse ids; the coarse
 * unknowns of each level are numbered rank-major, coarse_offsets[nranks+1] gives the rank
 * ranges.  An aggregate must not span two ranks. */
int alfd_set_aggregate_partition(alfd_ctx_t ctx, int level, const int64_t *coarse_offsets);
/* Algebraic aggregation for callers without grid information (a matrix replayed from an .alfd
 * file, a locally refined mesh with hanging-node rows): builds the aggregates of every level from
 * the uploaded slot A alone and stores them as alfd_set_aggregates would -- the counterpart of
 * ML's uncoupled aggregation (utilities.h:304-317: aggregation_threshold 0.02, one constant mode
 * per component).  Unknowns are node-major with block_size components per node; a node pair is
 * strongly coupled if max |a_ij| >= threshold * sqrt(d_I d_J); rows holding only their diagonal
 * (Dirichlet / constrained rows) are left out; aggregates hold at most max_aggregate_nodes nodes.
 * Coarsening stops at <= min_coarse unknowns or max_levels.  Deterministic (natural order).
 * Single rank.  alfd_get_aggregates returns a level (agg may be NULL to query the sizes), e.g. to
 * hand the same aggregates to another solver instance. */
int alfd_build_aggregates(alfd_ctx_t ctx, int32_t block_size, double threshold, int32_t max_aggregate_nodes,
                          int64_t min_coarse, int32_t max_levels, int32_t *levels_out);
int alfd_get_aggregates(alfd_ctx_t ctx, int level, int32_t *agg, int64_t capacity, int64_t *n_fine,
                        int64_t *n_coarse);
/* Host-only (no device, no context): ONE level of the same aggregation on a CSR matrix; agg[nrows]. */
int alfd_host_aggregate_level(int64_t nrows, const int64_t *row_ptr, const int32_t *col, const double *val,
                              int32_t block_size, double threshold, int32_t max_aggregate_nodes, int32_t *agg,
                              int64_t *n_coarse);
int alfd_configure(alfd_ctx_t ctx, const alfd_config *cfg);
/* New stop rules for the following solves WITHOUT a new alfd_setup (alfd_configure invalidates the setup): the
 * reference's SolverControl objects are plain members that a caller may change between two solve() calls
 * (outer_solver_control stokes...:282, control_lagrangian :1020-1023, control_mass :934).  NULL keeps a rule. */
int alfd_set_controls(alfd_ctx_t ctx, const alfd_control *outer, const alfd_control *inner, const alfd_control *mp_inner);
void alfd_default_config(alfd_config *cfg, int variant);
/* Builds transposes, sparse-row views, diag(Aug), lambda_max, halo plans
 * (replaces the setup in stokes...:1027-1045 / utilities.h:112-331). */
int alfd_setup(alfd_ctx_t ctx);

/* --------------------------------------------------------------- hot path
 * Depth 1. Replaces <Preconditioner>::vmult(dst, src)
 * (augmented_lagrangian_preconditioner.h:28-34, 62-70, 95-103, 130-156, 185-229).
 * src/dst: one host pointer per block (local rows). */
int alfd_precond_apply(alfd_ctx_t ctx, const double *const *src_blocks, double *const *dst_blocks,
                       alfd_result *res);
/* Replaces AA.vmult(dst, src): the block system operator (stokes...:1000-1003). */
int alfd_system_apply(alfd_ctx_t ctx, const double *const *src_blocks, double *const *dst_blocks);
/* Replaces the rhs augmentation f += gamma Ct invW g (stokes...:1012-1018,
 * immersed_laplace.cc:900-905): rhs_blocks[0] is updated in place from rhs_blocks[last]. */
int alfd_augment_rhs(alfd_ctx_t ctx, double *const *rhs_blocks);
/* Depth 2 (the measured path). Replaces
 * SolverFGMRES<BlockVector<double>>::solve(AA, x, b, P) (stokes...:1067-1074).
 * x_blocks: in = initial guess, out = solution. */
int alfd_solve(alfd_ctx_t ctx, const double *const *rhs_blocks, double *const *x_blocks,
               alfd_result *res);
/* (alfd_precond_apply, alfd_system_apply and alfd_augment_rhs stage their host vectors in buffers of
 * their own: a right-hand side / initial guess uploaded with alfd_upload_rhs stays intact, so the
 * depth-1 calls may be mixed with alfd_solve_resident.) */
/* Same solve with the vectors already resident in HBM (no PCIe in the timed
 * region): alfd_upload_rhs() then alfd_solve_resident() any number of times,
 * alfd_download_solution() at the end. */
int alfd_upload_rhs(alfd_ctx_t ctx, const double *const *rhs_blocks, const double *const *x0_blocks);
int alfd_solve_resident(alfd_ctx_t ctx, alfd_result *res);
int alfd_download_solution(alfd_ctx_t ctx, double *const *x_blocks);
/* Residual history of the last solve: out[k] = residual checked at step k. */
int alfd_get_history(alfd_ctx_t ctx, double *out, int32_t capacity, int32_t *count);

/* ------------------------------------------------------------- primitives
 * The kernels under the solver, callable on host data for parity tests
 * (SURVEY.md 8(a) a13/a14).  y = A x (mode 0) or y += alpha A x (mode 1).  On a partitioned context
 * alfd_spmv is collective: x holds this rank's owned columns of the slot's column block, y its rows. */
int alfd_spmv(alfd_ctx_t ctx, int slot, const double *x, double *y, int mode, double alpha);
int alfd_dot(alfd_ctx_t ctx, int64_t n, const double *x, const double *y, double *result);
/* Lanes per row the canonical SpMV order uses for this slot (after setup). */
int alfd_matrix_lanes(alfd_ctx_t ctx, int slot, int32_t *lanes);
/* Benchmark hook: run `reps` back-to-back y = A x launches of `slot` on resident
 * device data and return the mean kernel time in ms measured with HIP events on
 * the library's stream, plus the algorithmic bytes of one launch. */
int alfd_bench_spmv(alfd_ctx_t ctx, int slot, int32_t reps, double *ms_per_launch,
                    double *algorithmic_bytes);
/* Device storage format chosen for a matrix slot at upload:
 * lanes per row, LDS-window blocks, and -- for matrices whose row blocks repeat
 * <= 256 distinct entry values, as finite-element matrices on the reference's
 * uniformly refined hyper_cube grids do (stokes_immersed_boundary.cc:355-372) --
 * dictionary-coded values.  streamed_bytes is what one SpMV launch of the kernel
 * in use moves by format; algorithmic_bytes is the plain-CSR figure of SURVEY 8(d). */
typedef struct alfd_matrix_info {
  int32_t lanes, windowed, value_indexed;
  int32_t batch_major;   /* 0: no; 1: batch-major format on runs of the numbering; 2: on the caller's row blocks */
  int64_t nnz, window_blocks, window_fallback_blocks;
  int64_t value_indexed_blocks, value_indexed_nnz, dictionary_entries, value_wide_nnz;
  double algorithmic_bytes, streamed_bytes;
  int64_t shared_nnz;    /* batch-major forms: entries of rows stored as translates of a template row */
  int64_t batch_major_blocks;
  int64_t batch_major_wide;  /* 1: 10-bit dictionary codes / 11-bit window columns (blocks with > 512 distinct values) */
  int64_t batch_major_interior_blocks; /* partitioned contexts: leading row blocks that read no halo column -- launched
                                        * before the halo exchange is started (the rest after it has arrived); else 0 */
} alfd_matrix_info;
int alfd_get_matrix_info(alfd_ctx_t ctx, int slot, alfd_matrix_info *out);
/* Host-only (no device, no context): plans the LDS-window / value-indexed storage of
 * a CSR matrix exactly as alfd_set_matrix would (default tunables), decodes the plan
 * back -- window columns through the segment table, values through the block
 * dictionaries, every row through the class-sorted batch descriptors -- and reports
 * the number of entries / rows that do not reproduce the input (must be 0).
 * lanes: the canonical lanes-per-row of the matrix (alfd_matrix_lanes). */
typedef struct alfd_window_plan_info {
  int32_t windowed, value_indexed, row_block, max_window;
  int64_t blocks, fallback_blocks, segments;
  int64_t value_indexed_blocks, value_indexed_nnz, value_wide_nnz, dictionary_entries;
  int64_t batches, decode_mismatches;
} alfd_window_plan_info;
int alfd_host_window_plan(int64_t nrows, const int64_t *row_ptr, const int32_t *col, const double *val,
                          int32_t lanes, int32_t want_value_index, alfd_window_plan_info *out);
/* Host-only: plans the batch-major format (tunable "batch_major") of a CSR matrix as
 * alfd_set_matrix would -- row blocks = runs of `row_block` rows, or the caller's blocks as in
 * alfd_set_row_blocks when n_blocks > 0 -- and decodes it back (rows through the batch
 * descriptors, columns through the window segments, values through the dictionaries). */
typedef struct alfd_stream_plan_info {
  int32_t ok, max_window, max_rows, max_batches;
  int64_t blocks, batches, segments, dictionary_entries, stream_bytes;
  int64_t decode_mismatches, rows_covered;
  int64_t shared_nnz;   /* entries of rows stored as translates of a template row (one stored row per batch) */
} alfd_stream_plan_info;
int alfd_host_stream_plan(int64_t nrows, const int64_t *row_ptr, const int32_t *col, const double *val,
                          int32_t row_block, int64_t n_blocks, const int64_t *block_ptr, const int32_t *rows,
                          alfd_stream_plan_info *out);
/* The same for a short-row matrix (lanes = 8, 16 or 32, see alfd_matrix_lanes): the batch-major form of
 * spmv_vss_kernel -- one stored template row per batch of translate rows -- planned on runs of the numbering and
 * decoded back.  stream_bytes includes the batch descriptors. */
int alfd_host_stream_plan_short(int64_t nrows, const int64_t *row_ptr, const int32_t *col, const double *val,
                                int32_t lanes, alfd_stream_plan_info *out);
/* alfd_bench_spmv with the value-indexed kernel switched on (1) or off (0: the same
 * matrix through the 10 B/nnz window kernel); streamed_bytes as in alfd_matrix_info. */
int alfd_bench_spmv_format(alfd_ctx_t ctx, int slot, int32_t reps, int use_value_index,
                           double *ms_per_launch, double *streamed_bytes);
/* Row-block hint for the batch-major SpMV format of a long-row matrix (tunable "batch_major" = 1):
 * a partition of the rows of `slot` into blocks of at most 250 rows -- rows[block_ptr[b] .. block_ptr[b+1])
 * -- whose columns are close together, e.g. bricks of the mesh (all components of the nodes of a
 * 4 x 4 x 4 patch).  A block stages ONE window of x in LDS, so the fewer distinct columns a block
 * touches the better; the result of the SpMV does not depend on the blocks (each row keeps the
 * canonical summation order).  Takes effect at the next alfd_set_matrix of that slot; n_blocks = 0
 * removes the hint (blocks are then runs of the row numbering). */
int alfd_set_row_blocks(alfd_ctx_t ctx, int slot, int64_t n_blocks, const int64_t *block_ptr, const int32_t *rows);

/* Host-only helper for alfd_set_row_blocks when the caller has no grid metadata: recursive coordinate
 * bisection of one support point per matrix row (deal.II: DoFTools::map_dofs_to_support_points) into
 * blocks of at most max_rows (<= 250) rows.  block_ptr_out needs room for nrows + 1 entries, rows_out
 * for nrows; *n_blocks_out receives the number of blocks. */
int alfd_host_row_blocks_from_points(int64_t nrows, int32_t dim, const double *points, int32_t max_rows,
                                     int64_t *n_blocks_out, int64_t *block_ptr_out, int32_t *rows_out);

/* Host-only helpers for callers whose DoF numbering is not the one the SpMV formats like (a deal.II program numbers
 * with Cuthill-McKee and then block-wise, stokes_immersed_boundary.cc:533-541; the batch-major form wants the rows of a
 * mesh brick to read a compact set of columns).  From one support point per unknown of a block
 * (DoFTools::map_dofs_to_support_points):
 *   alfd_host_numbering_from_points: new_to_old[nrows] = the unknowns in lexicographic order of their points (last
 *     coordinate slowest), unknowns with the same point -- the components of a node -- kept together in their order.
 *     The front end (include/alfd/dealii_adapter.hpp, solver.py) permutes the operators and vectors with it BEFORE the
 *     upload and the solution back after the download: results live in the permuted numbering, the library itself
 *     never sees the caller's one.
 *   alfd_host_brick_blocks_from_points: row blocks for alfd_set_row_blocks = bricks of brick[0] x brick[1] x brick[2]
 *     grid nodes, a node's grid index along an axis being the rank of its coordinate among the distinct coordinates
 *     of that axis (exact on tensor grids, consistent on locally refined ones); blocks above max_rows rows are split.
 *   alfd_host_permute_csr: out = in with rows taken in the order row_new_to_old (NULL: unchanged) and column j renamed
 *     col_old_to_new[j] (NULL: unchanged), the entries of a row re-sorted by the new column.  out_row_ptr[nrows + 1],
 *     out_col / out_val [nnz]. */
int alfd_host_numbering_from_points(int64_t nrows, int32_t dim, const double *points, int64_t *new_to_old);
int alfd_host_brick_blocks_from_points(int64_t nrows, int32_t dim, const double *points, const int32_t *brick,
                                       int32_t max_rows, int64_t *n_blocks_out, int64_t *block_ptr_out, int32_t *rows_out);
int alfd_host_permute_csr(int64_t nrows, const int64_t *row_ptr, const int32_t *col, const double *val,
                          const int64_t *row_new_to_old, const int64_t *col_old_to_new, int64_t *out_row_ptr,
                          int32_t *out_col, double *out_val);

/* Free / total bytes of the context's device (hipMemGetInfo): leak checks, capacity planning. */
int alfd_get_device_memory(alfd_ctx_t ctx, int64_t *free_bytes, int64_t *total_bytes);
/* Run-time switches of a context (measurement and A/B comparison; results never change):
 *   "value_index"  1 (default): matrices whose row blocks were dictionary-coded at upload use the
 *                  3 B/nnz kernel; 0: every windowed matrix goes through the general 10 B/nnz kernel
 *                  (8-byte values + 16-bit window columns), as a matrix with unrelated values would.
 *   "batch_major"  1 (default): long-row matrices with repeating values, and short-row (8 / 16 / 32 lanes per
 *                  row) matrices whose rows are mostly translates of one another, use the batch-major forms of
 *                  csrc/kernels_vs.hpp (decided at alfd_set_matrix; also switches the kernel at launch); 0: the
 *                  round-1 window formats.  "batch_major_rows" (4..250, default 96): rows per block of the long-row
 *                  form when no alfd_set_row_blocks hint is given; "batch_major_waves" (2, 4, 8): waves per workgroup;
 *                  "batch_major_xcd" (0/1): XCD-contiguous block order (measured slower); "batch_major_share" (0/1):
 *                  store rows that are translates of one another once (0: every row stored, ~3.1 B/nnz -- what a
 *                  matrix with repeating values but no translate structure gets; at the next alfd_set_matrix);
 *                  "batch_major_wide" (0/1, default 1): blocks with more than 512 distinct values are re-planned with
 *                  10-bit codes / 11-bit window columns instead of being halved (cell-wise assembled matrices).
 * Returns ALFD_E_INVALID for an unknown name. */
int alfd_set_tunable(alfd_ctx_t ctx, const char *name, int value);
/* Kernel-class timing of the last solve, accumulated with HIP events when
 * alfd_enable_timing(ctx, 1) was called before: class ids in alfd_timing_class. */
enum alfd_timing_class {
  ALFD_T_SPMV_A = 0,
  ALFD_T_SPMV_OTHER = 1,
  ALFD_T_DOT = 2,
  ALFD_T_VEC = 3,
  ALFD_T_NCLASSES = 4
};
int alfd_enable_timing(alfd_ctx_t ctx, int on); /* 0 off, 1 = A-SpMV launches only, 2 = all classes */
int alfd_get_timing(alfd_ctx_t ctx, double *ms /*[ALFD_T_NCLASSES]*/, int64_t *launches /*[..]*/,
                    double *algorithmic_bytes /*[..]*/);
/* The same launches priced by the bytes of the storage format each kernel reads (alfd_matrix_info::streamed_bytes for
 * the SpMV classes; equal to the algorithmic bytes for vector kernels): what a perfect cache would still move. */
int alfd_get_timing_streamed(alfd_ctx_t ctx, double *streamed_bytes /*[ALFD_T_NCLASSES]*/);
/* Wall seconds of the last uploads + alfd_setup by phase (host clock, device-synchronised at the phase ends): what
 * the reference's "Solve system" timer also contains (AMG setup, factorisations: stokes...:827, immersed_laplace.cc:504). */
enum alfd_setup_phase {
  ALFD_SETUP_UPLOAD = 0,        /* alfd_set_matrix calls since the last alfd_setup: format planning + copies to HBM */
  ALFD_SETUP_DIAG_LAMBDA = 1,   /* diag(Aug), lambda_max of the inner operators */
  ALFD_SETUP_ML_FETCH = 2,      /* multilevel: host copies of the level-0 operators */
  ALFD_SETUP_ML_GALERKIN = 3,   /* Galerkin products of all levels */
  ALFD_SETUP_ML_UPLOAD = 4,     /* level operators, transfers: format planning + copies */
  ALFD_SETUP_ML_LAMBDA = 5,     /* per-level diagonals and lambda_max */
  ALFD_SETUP_ML_PATCH = 6,      /* interface-patch operators + lambda_max */
  ALFD_SETUP_ML_COARSE = 7,     /* explicit coarsest inverse */
  ALFD_SETUP_NPHASES = 8
};
int alfd_get_setup_seconds(alfd_ctx_t ctx, double *seconds /*[ALFD_SETUP_NPHASES]*/);

#ifdef __cplusplus
}
#endif
#endif /* ALFD_H */
