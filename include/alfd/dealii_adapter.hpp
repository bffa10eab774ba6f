// dealii_adapter.hpp -- header-only C++17 adapter that presents the C ABI of
// alfd.h as deal.II-shaped classes, so that the reference's solve() functions
// (immersed_laplace.cc:636-949, stokes_immersed_boundary.cc:918-1079) keep
// their call-site shape:
//
//     alfd::dealii_adapter::System sys(/*device*/ 0);
//     sys.set_matrix(ALFD_A,  stokes_matrix.block(0, 0));   // linear_operator(...) captures
//     sys.set_matrix(ALFD_BT, stokes_matrix.block(0, 1));
//     sys.set_matrix(ALFD_CT, coupling_matrix);
//     sys.set_matrix(ALFD_MP, preconditioner_matrix.block(1, 1));
//     sys.set_diag(ALFD_INVW, inverse_squares);             // DiagonalMatrix<Vector<double>>
//     sys.set_diag(ALFD_MP_LUMPED_INV, pressure_diagonal_inv);
//     // `Diagonal mass immersed = false` (stokes...:979-985): cfg.w_inverse = ALFD_W_MASS_INV_SQUARED and
//     // sys.set_matrix(ALFD_M, mass_matrix) -- the UMFPACK solves become CG on M inside the library
//     sys.configure(cfg); sys.setup();
//     BlockPreconditionerAugmentedLagrangianStokes P(sys);  // vmult(v, u) const
//     SolverFGMRES<BlockVector<double>> solver(sys);        // solve(AA, x, b, P)
//     solver.solve(sys.system_operator(), solution_block, system_rhs_block, P);
//     outer_solver_control.last_step()  ->  solver.last_step()
// The other call sites map the same way: elliptic_interface.cc:900-906 / 940-948 ->
// EllipticInterfacePreconditioners::BlockTriangularALPreconditioner{,Modified} + SolverFGMRES;
// immersed_laplace.cc:625-631 -> RationalPreconditioner + SolverMinRes (tests/adapter/adapter_demo.cpp
// runs all three against the golden iteration counts).
//
// It is templated on the matrix / vector types and needs only the members the
// reference uses: SparseMatrix: m(), n(), n_nonzero_elements(), begin(row) /
// end(row) iterators with column() and value(); Vector: size(), begin();
// BlockVector: n_blocks(), block(i).  tests/test_adapter.py compiles it against
// a 60-line mock of those classes (deal.II itself is not installed here).
//
// Errors: a non-zero status from the ABI becomes a C++ exception --
// alfd::dealii_adapter::NoConvergence (mirror of dealii::SolverControl::
// NoConvergence, caught by the reference's main(): stokes...:1233-1254) or
// alfd::dealii_adapter::Error (mirror of ExcMessage).
#ifndef ALFD_DEALII_ADAPTER_HPP
#define ALFD_DEALII_ADAPTER_HPP

#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "alfd/alfd.h"

namespace alfd {
namespace dealii_adapter {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string &what) : std::runtime_error(what), status(s) {}
};
// dealii::SolverControl::NoConvergence carries last_step / last_residual.
struct NoConvergence : Error {
  unsigned int last_step;
  double last_residual;
  NoConvergence(int s, const std::string &what, unsigned int step, double res)
      : Error(s, what), last_step(step), last_residual(res) {}
};

class System {
 public:
  explicit System(int device_id = 0) {
    const int rc = alfd_create(&ctx_, device_id);
    if (rc != ALFD_OK) throw Error(rc, std::string("alfd_create: ") + alfd_strerror(rc));
  }
  ~System() {
    if (ctx_) alfd_destroy(ctx_);
  }
  System(const System &) = delete;
  System &operator=(const System &) = delete;

  // deal.II stores the diagonal first in each row of a square matrix; the ABI
  // wants ascending columns.  Copies once into CSR, uploads, frees.
  template <class SparseMatrixType>
  void set_matrix(int slot, const SparseMatrixType &M) {
    const int64_t nrows = (int64_t)M.m();
    std::vector<int64_t> rp(nrows + 1, 0);
    std::vector<int32_t> col;
    std::vector<double> val;
    col.reserve(M.n_nonzero_elements());
    val.reserve(M.n_nonzero_elements());
    // front-end renumbering of the block-0 space (set_numbering_from_support_points): rows of A / BT / CT are taken
    // in the new order, columns of A / B / C renamed -- the library only ever sees the new numbering
    const bool prow = !new_to_old_.empty() && (slot == ALFD_A || slot == ALFD_BT || slot == ALFD_CT);
    const bool pcol = !new_to_old_.empty() && (slot == ALFD_A || slot == ALFD_B || slot == ALFD_C);
    if ((prow && nrows != (int64_t)new_to_old_.size()) || (pcol && (int64_t)M.n() != (int64_t)new_to_old_.size()))
      throw Error(ALFD_E_INVALID, "set_matrix: the renumbering does not match this operator's block-0 size");
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t r = 0; r < nrows; ++r) {
      row.clear();
      const int64_t src = prow ? new_to_old_[r] : r;
      for (auto it = M.begin(src); it != M.end(src); ++it)
        row.emplace_back(pcol ? (int32_t)old_to_new_[it->column()] : (int32_t)it->column(), it->value());
      std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      for (const auto &e : row) {
        col.push_back(e.first);
        val.push_back(e.second);
      }
      rp[r + 1] = (int64_t)col.size();
    }
    check(alfd_set_matrix(ctx_, slot, nrows, (int64_t)M.n(), rp.data(), col.data(), val.data()));
  }

  // Front-end renumbering of the block-0 unknowns (velocity / background space) from one support point per unknown
  // (DoFTools::map_dofs_to_support_points restricted to that block).  The reference numbers its DoFs with
  // Cuthill-McKee, then block-wise (stokes_immersed_boundary.cc:533-541); the batch-major SpMV wants the rows of a
  // mesh brick to read a compact window of x.  This call (BEFORE any set_matrix / set_prolongator) fixes a
  // permutation -- lexicographic order of the points, components of a node together
  // (alfd_host_numbering_from_points) -- and the brick row blocks of A (alfd_host_brick_blocks_from_points);
  // set_matrix / set_prolongator then permute what they upload, and every vector that crosses the ABI
  // (vmult, solve, augment_rhs) is permuted on the way in and back on the way out: the caller keeps its numbering.
  template <class PointContainer>
  void set_numbering_from_support_points(const PointContainer &points, int dim, const int32_t brick[3] = nullptr) {
    const int64_t n = (int64_t)points.size();
    std::vector<double> xyz((size_t)n * dim), sorted((size_t)n * dim);
    int64_t i = 0;
    for (const auto &p : points) {
      for (int d = 0; d < dim; ++d) xyz[(size_t)i * dim + d] = p[d];
      ++i;
    }
    new_to_old_.assign((size_t)n, 0);
    old_to_new_.assign((size_t)n, 0);
    check(alfd_host_numbering_from_points(n, dim, xyz.data(), new_to_old_.data()));
    for (int64_t k = 0; k < n; ++k) {
      old_to_new_[new_to_old_[k]] = k;
      for (int d = 0; d < dim; ++d) sorted[(size_t)k * dim + d] = xyz[(size_t)new_to_old_[k] * dim + d];
    }
    static const int32_t default_brick[3] = {16, 4, 1};
    std::vector<int64_t> ptr((size_t)n + 1);
    std::vector<int32_t> rows((size_t)n);
    int64_t nb = 0;
    check(alfd_host_brick_blocks_from_points(n, dim, sorted.data(), brick ? brick : default_brick, 250, &nb, ptr.data(),
                                             rows.data()));
    check(alfd_set_row_blocks(ctx_, ALFD_A, nb, ptr.data(), rows.data()));
    in0_.assign((size_t)n, 0.0);
    out0_.assign((size_t)n, 0.0);
  }
  bool renumbered() const { return !new_to_old_.empty(); }

  // CSR prolongator of a multigrid level (alfd_set_prolongator), e.g. from MGTransferPrebuilt or
  // FETools::get_interpolation_matrix; the rows of level 0 follow the block-0 renumbering.
  template <class SparseMatrixType>
  void set_prolongator(int level, const SparseMatrixType &P) {
    const int64_t nrows = (int64_t)P.m();
    const bool prow = level == 0 && !new_to_old_.empty();
    std::vector<int64_t> rp(nrows + 1, 0);
    std::vector<int32_t> col;
    std::vector<double> val;
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t r = 0; r < nrows; ++r) {
      row.clear();
      const int64_t src = prow ? new_to_old_[r] : r;
      for (auto it = P.begin(src); it != P.end(src); ++it) row.emplace_back((int32_t)it->column(), it->value());
      std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      for (const auto &e : row) {
        col.push_back(e.first);
        val.push_back(e.second);
      }
      rp[r + 1] = (int64_t)col.size();
    }
    check(alfd_set_prolongator(ctx_, level, nrows, (int64_t)P.n(), rp.data(), col.data(), val.data()));
  }

  template <class VectorType>
  void set_diag(int slot, const VectorType &d) {
    check(alfd_set_diag(ctx_, slot, (int64_t)d.size(), &*d.begin()));
  }

  // Row blocks of the SpMV on `slot` (alfd_set_row_blocks) from one support point per row, e.g.
  //   std::map<types::global_dof_index, Point<dim>> sp;  DoFTools::map_dofs_to_support_points(mapping, dh, sp);
  // restricted to the velocity block.  Call before set_matrix(slot, ...).  A block stages one window of x
  // in LDS: spatially compact blocks (recursive coordinate bisection here) keep that window a third of
  // what a run of consecutive DoF indices needs.  Does not change any result.
  template <class PointContainer>
  void set_row_blocks_from_support_points(int slot, const PointContainer &points, int dim, int max_rows = 192) {
    const int64_t n = (int64_t)points.size();
    std::vector<double> xyz((size_t)n * dim);
    int64_t i = 0;
    for (const auto &p : points) {
      for (int d = 0; d < dim; ++d) xyz[(size_t)i * dim + d] = p[d];
      ++i;
    }
    std::vector<int64_t> ptr((size_t)n + 1);
    std::vector<int32_t> rows((size_t)n);
    int64_t nb = 0;
    check(alfd_host_row_blocks_from_points(n, dim, xyz.data(), max_rows, &nb, ptr.data(), rows.data()));
    check(alfd_set_row_blocks(ctx_, slot, nb, ptr.data(), rows.data()));
  }

  // Aggregates of the multilevel inner preconditioner (ALFD_PREC_MULTILEVEL; replaces the ML
  // aggregation of utilities.h:304-317).  weights may be empty (constant modes).
  void set_aggregates(int level, const std::vector<int32_t> &agg, int64_t n_coarse,
                      const std::vector<double> &weights = {}) {
    check(alfd_set_aggregates(ctx_, level, (int64_t)agg.size(), agg.data(), weights.empty() ? nullptr : weights.data(),
                              n_coarse));
  }
  // `Use diagonal inverse = false` / `Diagonal mass immersed = false` (immersed_laplace.cc:859-877,
  // stokes...:979-985, elliptic_interface.cc:713-737): the exact W^-1 = (M^-1)^2 (mode
  // ALFD_W_MASS_INV_SQUARED) or M^-1 (ALFD_W_MASS_INV) from the immersed mass matrix; applied to the
  // config passed to configure() afterwards.
  template <class SparseMatrixType>
  void set_w_inverse(int mode, const SparseMatrixType &mass_matrix) {
    set_matrix(ALFD_M, mass_matrix);
    w_inverse_ = mode;
  }
  // multi-GPU: one System per rank (alfd.h: alfd_comm_init / alfd_set_partition), before any set_matrix
  void comm_init(int rank, int nranks, const void *unique_id, size_t bytes) {
    check(alfd_comm_init(ctx_, rank, nranks, unique_id, bytes));
  }
  void set_partition(const std::vector<std::vector<int64_t>> &offsets) {
    std::vector<const int64_t *> p;
    for (const auto &o : offsets) p.push_back(o.data());
    check(alfd_set_partition(ctx_, (int)p.size(), p.data()));
  }

  void configure(const alfd_config &cfg_in) {
    alfd_config cfg = cfg_in;
    if (w_inverse_ >= 0) cfg.w_inverse = w_inverse_;
    cfg_ = cfg;
    check(alfd_configure(ctx_, &cfg));
  }
  void setup() { check(alfd_setup(ctx_)); }
  const alfd_config &config() const { return cfg_; }
  alfd_ctx_t handle() const { return ctx_; }

  // f += gamma Ct invW g  (stokes...:1012-1018)
  template <class BlockVectorType>
  void augment_rhs(BlockVectorType &rhs) {
    std::vector<double *> p = out_ptrs(rhs, /*load=*/true);
    check(alfd_augment_rhs(ctx_, p.data()));
    finish_out(rhs);
  }

  // The block_operator AA (stokes...:1000-1003) as an object with vmult().
  class SystemOperator {
   public:
    explicit SystemOperator(System &s) : sys_(&s) {}
    template <class BlockVectorType>
    void vmult(BlockVectorType &dst, const BlockVectorType &src) const {
      std::vector<const double *> s = sys_->in_ptrs(src);
      std::vector<double *> d = sys_->out_ptrs(dst, false);
      sys_->check(alfd_system_apply(sys_->ctx_, s.data(), d.data()));
      sys_->finish_out(dst);
    }
    System *system() const { return sys_; }

   private:
    System *sys_;
  };
  SystemOperator system_operator() { return SystemOperator(*this); }

  void check(int rc, unsigned int step = 0, double res = 0) const {
    if (rc == ALFD_OK) return;
    const std::string msg = std::string(alfd_strerror(rc)) + ": " + alfd_last_error(ctx_);
    if (rc == ALFD_E_NO_CONVERGENCE_OUTER || rc == ALFD_E_NO_CONVERGENCE_INNER)
      throw NoConvergence(rc, msg, step, res);
    throw Error(rc, msg);
  }

  // block pointers for the ABI; with a front-end renumbering block 0 goes through staging copies in the new order
  template <class BlockVectorType>
  std::vector<const double *> in_ptrs(const BlockVectorType &v) {
    std::vector<const double *> p = cptrs(v);
    if (!new_to_old_.empty()) {
      for (size_t k = 0; k < new_to_old_.size(); ++k) in0_[k] = p[0][new_to_old_[k]];
      p[0] = in0_.data();
    }
    return p;
  }
  template <class BlockVectorType>
  std::vector<double *> out_ptrs(BlockVectorType &v, bool load) {
    std::vector<double *> p = ptrs(v);
    if (!new_to_old_.empty()) {
      if (load)
        for (size_t k = 0; k < new_to_old_.size(); ++k) out0_[k] = p[0][new_to_old_[k]];
      p[0] = out0_.data();
    }
    return p;
  }
  template <class BlockVectorType>
  void finish_out(BlockVectorType &v) {
    if (new_to_old_.empty()) return;
    double *b0 = &*v.block(0).begin();
    for (size_t k = 0; k < new_to_old_.size(); ++k) b0[new_to_old_[k]] = out0_[k];
  }

  template <class BlockVectorType>
  static std::vector<double *> ptrs(BlockVectorType &v) {
    std::vector<double *> p(v.n_blocks());
    for (unsigned int b = 0; b < v.n_blocks(); ++b) p[b] = &*v.block(b).begin();
    return p;
  }
  template <class BlockVectorType>
  static std::vector<const double *> cptrs(const BlockVectorType &v) {
    std::vector<const double *> p(v.n_blocks());
    for (unsigned int b = 0; b < v.n_blocks(); ++b) p[b] = &*v.block(b).begin();
    return p;
  }

 private:
  alfd_ctx_t ctx_ = nullptr;
  alfd_config cfg_{};
  int w_inverse_ = -1;
  std::vector<int64_t> new_to_old_, old_to_new_;   // front-end renumbering of block 0 (empty: none)
  std::vector<double> in0_, out0_;                  // staging of block-0 vectors in the library's numbering
  friend class SystemOperator;
};

// Common base of the preconditioner classes: depth-1 drop-in, deal.II keeps
// its own SolverFGMRES and calls vmult() once per outer iteration.
template <int Variant>
class ALPreconditioner {
 public:
  explicit ALPreconditioner(System &s) : sys_(&s) {
    if (s.config().variant != Variant) throw Error(ALFD_E_INVALID, "context configured for another variant");
  }
  // void vmult(BlockVector<double>& v, const BlockVector<double>& u) const
  template <class BlockVectorType>
  void vmult(BlockVectorType &v, const BlockVectorType &u) const {
    std::vector<const double *> s = sys_->in_ptrs(u);
    std::vector<double *> d = sys_->out_ptrs(v, false);
    alfd_result res{};
    sys_->check(alfd_precond_apply(sys_->handle(), s.data(), d.data(), &res));
    sys_->finish_out(v);
    last_ = res;
  }
  const alfd_result &last_result() const { return last_; }
  System *system() const { return sys_; }

 private:
  System *sys_;
  mutable alfd_result last_{};
};

// augmented_lagrangian_preconditioner.h:14-42, :44-79, :81-110
using BlockPreconditionerAugmentedLagrangian = ALPreconditioner<ALFD_AL2>;
using BlockPreconditionerAugmentedLagrangianStokes = ALPreconditioner<ALFD_AL_STOKES>;
using BlockPreconditionerAugmentedLagrangianDiagonal = ALPreconditioner<ALFD_AL_STOKES_DIAG>;

// rational_preconditioner.h:12-99 (used with SolverMinRes, immersed_laplace.cc:625-631)
using RationalPreconditioner = ALPreconditioner<ALFD_RATIONAL>;
// augmented_lagrangian_preconditioner.h:115-164 and :168-238 live in this namespace in the reference
namespace EllipticInterfacePreconditioners {
using BlockTriangularALPreconditioner = ALPreconditioner<ALFD_AL_ELL_IDEAL>;
using BlockTriangularALPreconditionerModified = ALPreconditioner<ALFD_AL_ELL_MODIFIED>;
}  // namespace EllipticInterfacePreconditioners

// Depth-2 drop-ins for SolverFGMRES<BlockVector<double>> (stokes...:1067-1074,
// elliptic_interface.cc:862-906) and SolverMinRes<BlockVector<double>> (immersed_laplace.cc:629-631,
// stokes...:1057-1064): the whole solve runs on the GPU; the stop rule is alfd_config::outer and the
// Krylov method alfd_config::outer_solver, which must match the class used.
template <class BlockVectorType, int OuterSolver>
class GpuKrylovSolver {
 public:
  explicit GpuKrylovSolver(System &s) : sys_(&s) {
    if (s.config().outer_solver != OuterSolver)
      throw Error(ALFD_E_INVALID, "alfd_config::outer_solver does not match this solver class");
  }
  template <class MatrixType, class PreconditionerType>
  void solve(const MatrixType &A, BlockVectorType &x, const BlockVectorType &b, const PreconditionerType &P) {
    if (A.system() != sys_ || P.system() != sys_)
      throw Error(ALFD_E_INVALID, "operator, preconditioner and solver must share one System");
    std::vector<const double *> rhs = sys_->in_ptrs(b);
    std::vector<double *> sol = sys_->out_ptrs(x, /*load=*/true);   // x carries the initial guess
    const int rc = alfd_solve(sys_->handle(), rhs.data(), sol.data(), &last_);
    sys_->finish_out(x);
    sys_->check(rc, (unsigned int)last_.outer_iterations, last_.last_residual);
  }
  unsigned int last_step() const { return (unsigned int)last_.outer_iterations; }   // SolverControl::last_step()
  double last_value() const { return last_.last_residual; }                        // SolverControl::last_value()
  const alfd_result &last_result() const { return last_; }

 private:
  System *sys_;
  alfd_result last_{};
};
template <class BlockVectorType>
using SolverFGMRES = GpuKrylovSolver<BlockVectorType, ALFD_OUTER_FGMRES>;
template <class BlockVectorType>
using SolverMinRes = GpuKrylovSolver<BlockVectorType, ALFD_OUTER_MINRES>;

}  // namespace dealii_adapter
}  // namespace alfd

#endif  // ALFD_DEALII_ADAPTER_HPP
