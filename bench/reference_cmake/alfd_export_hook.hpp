// alfd_export_hook.hpp -- what the inserted hook statements of inject_export.cmake expand to.  They sit inside the
// reference's solve() right in front of the FGMRES solver object, where every operator of the block system is in scope
// under the reference's own names (stokes_immersed_boundary.cc:923-1018, immersed_laplace.cc:638-905): the blocks are
// written with include/alfd/dealii_export.hpp, the diagonal weights recomputed exactly as the reference computes them
// (1 / M_ii^2, :976-978; lumped 1 / (Mp 1)_i, :946-951), the stop rules copied from the SolverControl objects.
#ifndef ALFD_EXPORT_HOOK_HPP
#define ALFD_EXPORT_HOOK_HPP

#include "alfd/dealii_export.hpp"

namespace alfd_export_hook {
template <class ControlType>
inline alfd_control reduction_control(const ControlType &c) {
  alfd_control out;
  out.kind = ALFD_CTRL_REDUCTION;
  out.max_steps = (int32_t)c.max_steps();
  out.tol = c.tolerance();
  out.reduce = c.reduction();
  return out;
}
}  // namespace alfd_export_hook

// stokes_immersed_boundary.cc, branch "IBStokesAL", in front of :1067
#define ALFD_EXPORT_STOKES_HOOK(path)                                                                              \
  do {                                                                                                             \
    alfd::dealii_export::Writer alfd_w(path);                                                                      \
    alfd_w.matrix(ALFD_A, stokes_matrix.block(0, 0));                                                              \
    alfd_w.matrix(ALFD_BT, stokes_matrix.block(0, 1));                                                             \
    alfd_w.matrix(ALFD_B, stokes_matrix.block(1, 0));                                                              \
    alfd_w.matrix(ALFD_CT, coupling_matrix);                                                                       \
    alfd_w.matrix(ALFD_M, mass_matrix_immersed);                                                                   \
    alfd_w.matrix(ALFD_MP, preconditioner_matrix.block(1, 1));                                                     \
    alfd_w.diag(ALFD_INVW, inverse_squares);                                                                       \
    {                                                                                                              \
      Vector<double> alfd_ones(preconditioner_matrix.block(1, 1).m()), alfd_lumped(alfd_ones.size());             \
      alfd_ones = 1.;                                                                                              \
      preconditioner_matrix.block(1, 1).vmult(alfd_lumped, alfd_ones);                                             \
      for (double &alfd_x : alfd_lumped) alfd_x = 1. / alfd_x;                                                     \
      alfd_w.diag(ALFD_MP_LUMPED_INV, alfd_lumped);                                                                \
    }                                                                                                              \
    alfd_w.rhs(system_rhs_block);       /* already augmented (:1012-1018) */                                       \
    alfd_config alfd_cfg;                                                                                          \
    alfd_default_config(&alfd_cfg, ALFD_AL_STOKES);                                                                \
    alfd_cfg.gamma = gamma;                                                                                        \
    alfd_cfg.gamma_grad_div = gamma_grad_div;                                                                      \
    alfd_cfg.grad_div_in_A = augmented_lagrangian_control.grad_div_stabilization ? 1 : 0;                         \
    alfd_cfg.w_inverse = augmented_lagrangian_control.inverse_diag_square ? ALFD_W_DIAGONAL : ALFD_W_MASS_INV_SQUARED; \
    alfd_cfg.outer = alfd_export_hook::reduction_control(outer_solver_control);                                    \
    alfd_cfg.inner.kind = ALFD_CTRL_ABS;                                                                           \
    alfd_cfg.inner.max_steps = (int32_t)control_lagrangian.max_steps();                                            \
    alfd_cfg.inner.tol = control_lagrangian.tolerance();                                                           \
    alfd_cfg.mp_inner.kind = ALFD_CTRL_ABS;                                                                        \
    alfd_cfg.mp_inner.max_steps = (int32_t)control_mass.max_steps();                                               \
    alfd_cfg.mp_inner.tol = control_mass.tolerance();                                                              \
    alfd_w.config(alfd_cfg);                                                                                       \
    alfd_w.close();                                                                                                \
  } while (0)

// immersed_laplace.cc, branch "augmented", in front of :917
#define ALFD_EXPORT_LAPLACE_HOOK(path)                                                                             \
  do {                                                                                                             \
    alfd::dealii_export::Writer alfd_w(path);                                                                      \
    alfd_w.matrix(ALFD_A, stiffness_matrix);                                                                       \
    alfd_w.matrix(ALFD_CT, coupling_matrix);                                                                       \
    alfd_w.matrix(ALFD_M, mass_matrix);                                                                            \
    {                                                                                                              \
      Vector<double> alfd_inv(mass_matrix.m());                                                                    \
      for (types::global_dof_index alfd_i = 0; alfd_i < mass_matrix.m(); ++alfd_i)                                 \
        alfd_inv(alfd_i) = 1. / (mass_matrix.diag_element(alfd_i) * mass_matrix.diag_element(alfd_i));             \
      alfd_w.diag(ALFD_INVW, alfd_inv);                                                                            \
    }                                                                                                              \
    alfd_w.rhs(system_rhs_block);                                                                                  \
    alfd_config alfd_cfg;                                                                                          \
    alfd_default_config(&alfd_cfg, ALFD_AL2);                                                                      \
    alfd_cfg.gamma = gamma;                                                                                        \
    alfd_cfg.outer = alfd_export_hook::reduction_control(schur_solver_control);                                    \
    alfd_w.config(alfd_cfg);                                                                                       \
    alfd_w.close();                                                                                                \
  } while (0)

#endif  // ALFD_EXPORT_HOOK_HPP
