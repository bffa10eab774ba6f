// Drives include/alfd/dealii_adapter.hpp the way immersed_laplace.cc:636-949
// drives deal.II: operators from (mock) SparseMatrix objects, W^-1 = 1/M_ii^2,
// rhs augmentation, BlockPreconditionerAugmentedLagrangian + SolverFGMRES.
// Prints "outer=<n> inner=<n>"; exit code 3 if no GPU context can be created.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>

#include "alfd/dealii_adapter.hpp"
#include "alfd/dealii_export.hpp"
#include "mock_dealii.hpp"

extern "C" {
struct alfd_synth_params {
  int32_t dim, degree, ncomp, n_cells;
  double lo, hi;
  int32_t stokes, grad_div;
  double gamma_grad_div, beta;
  double center[3];
  double radius;
  int32_t immersed_refine, coupling_nq;
  double body_force[3];
  double embedded_value[3];
  int64_t u_node0, u_node1, p_node0, p_node1, l0, l1;
};
void *alfd_synth_generate(const alfd_synth_params *, char *, int);
void alfd_synth_free(void *);
int alfd_synth_matrix(void *, const char *, int64_t *, int64_t *, int64_t *, const int64_t **, const int32_t **,
                      const double **);
int alfd_synth_vector(void *, const char *, int64_t *, const double **);
}

static mock::SparseMatrix load(void *h, const char *name) {
  int64_t m, n, nnz;
  const int64_t *rp;
  const int32_t *col;
  const double *val;
  if (alfd_synth_matrix(h, name, &m, &n, &nnz, &rp, &col, &val) != 0) throw std::runtime_error(name);
  return mock::SparseMatrix((size_t)m, (size_t)n, (const long *)rp, col, val);
}

int main(int argc, char **argv) {
  using namespace alfd::dealii_adapter;
  alfd_synth_params sp;
  std::memset(&sp, 0, sizeof(sp));
  sp.dim = 2, sp.degree = 1, sp.ncomp = 1, sp.n_cells = 64, sp.lo = 0, sp.hi = 1, sp.beta = 1;
  sp.center[0] = sp.center[1] = 0.4, sp.radius = 0.2, sp.immersed_refine = 4, sp.coupling_nq = 3;
  sp.embedded_value[0] = 1.0;
  sp.u_node0 = sp.u_node1 = sp.p_node0 = sp.p_node1 = sp.l0 = sp.l1 = -1;
  char err[256];
  void *h = alfd_synth_generate(&sp, err, 256);
  if (!h) return std::fprintf(stderr, "generator: %s\n", err), 2;
  try {
    mock::SparseMatrix stiffness_matrix = load(h, "A"), coupling_matrix = load(h, "Ct"),
                       mass_matrix = load(h, "M");
    const size_t n_u = stiffness_matrix.m(), n_l = mass_matrix.m();
    mock::Vector inv_diagonal(n_l);   // immersed_laplace.cc:866-869
    for (size_t i = 0; i < n_l; ++i)
      inv_diagonal[i] = 1. / (mass_matrix.diag_element(i) * mass_matrix.diag_element(i));

    if (argc > 2 && std::string(argv[1]) == "export") {
      // dump the operators the way a deal.II user would (no GPU needed)
      alfd_config cfg;
      alfd_default_config(&cfg, ALFD_AL2);
      cfg.outer = {ALFD_CTRL_REDUCTION, 1000, 1e-10, 1e-12};
      cfg.inner.max_steps = 1000;
      mock::BlockVector rhs({n_u, n_l});
      int64_t ng;
      const double *gg;
      alfd_synth_vector(h, "g", &ng, &gg);
      for (size_t i = 0; i < n_l; ++i) rhs.block(1)[i] = gg[i];
      alfd::dealii_export::Writer w(argv[2]);
      w.matrix(ALFD_A, stiffness_matrix);
      w.matrix(ALFD_CT, coupling_matrix);
      w.diag(ALFD_INVW, inv_diagonal);
      w.rhs(rhs);
      w.config(cfg);
      w.close();
      alfd_synth_free(h);
      return 0;
    }
    System sys(0);
    sys.set_matrix(ALFD_A, stiffness_matrix);
    sys.set_matrix(ALFD_CT, coupling_matrix);     // C = transpose_operator(Ct) is derived
    sys.set_diag(ALFD_INVW, inv_diagonal);
    alfd_config cfg;
    alfd_default_config(&cfg, ALFD_AL2);
    cfg.gamma = 10;                                // immersed_laplace.cc:647
    cfg.outer = {ALFD_CTRL_REDUCTION, 1000, 1e-10, 1e-12};
    cfg.inner.max_steps = 1000;
    sys.configure(cfg);
    sys.setup();

    mock::BlockVector solution_block({n_u, n_l}), system_rhs_block({n_u, n_l});
    int64_t n;
    const double *g;
    alfd_synth_vector(h, "g", &n, &g);
    for (size_t i = 0; i < n_l; ++i) system_rhs_block.block(1)[i] = g[i];
    sys.augment_rhs(system_rhs_block);             // immersed_laplace.cc:900-905

    auto AA = sys.system_operator();
    BlockPreconditionerAugmentedLagrangian augmented_lagrangian_preconditioner(sys);
    SolverFGMRES<mock::BlockVector> solver_fgmres(sys);
    solver_fgmres.solve(AA, solution_block, system_rhs_block, augmented_lagrangian_preconditioner);
    std::printf("outer=%u inner=%lld residual=%.6e\n", solver_fgmres.last_step(),
                (long long)solver_fgmres.last_result().inner_iterations, solver_fgmres.last_value());
    // depth-1 use: one preconditioner application through vmult()
    mock::BlockVector v({n_u, n_l});
    augmented_lagrangian_preconditioner.vmult(v, system_rhs_block);
    std::printf("vmult inner=%lld\n", (long long)augmented_lagrangian_preconditioner.last_result().inner_iterations);
  } catch (const NoConvergence &e) {
    std::fprintf(stderr, "NoConvergence at step %u: %s\n", e.last_step, e.what());
    alfd_synth_free(h);
    return 1;
  } catch (const Error &e) {
    std::fprintf(stderr, "alfd error %d: %s\n", e.status, e.what());
    alfd_synth_free(h);
    return e.status == ALFD_E_HIP ? 3 : 1;
  }
  alfd_synth_free(h);
  return 0;
}
