/* synth.h -- C interface of libalfd_synth.so, the synthetic fictitious-domain operator
 * generator (input tooling: it stands in for the deal.II assembly the reference does in
 * immersed_laplace.cc:278-496, stokes_immersed_boundary.cc:410-820, elliptic_interface.cc:450-670;
 * no solver arithmetic).  Python binds it in problems.py, the C++ adapter demo includes this header. */
#ifndef ALFD_SYNTH_H
#define ALFD_SYNTH_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct alfd_synth_params {
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
  int32_t immersed_kind, imm_cells;
  double imm_lo, imm_hi, beta2;
  int32_t want_surface_mass;
  /* 0: rows written in closed form from Kronecker products of 1-D element matrices (default);
   * 1: CELL-WISE assembly as deal.II does it (stokes_immersed_boundary.cc:668-760): ONE cell matrix integrated
   *    numerically (Gauss quadrature, Jacobian from the vertex coordinates of a cell), every global entry the
   *    floating-point sum of its cells' contributions in the order a refinement tree visits the cells
   *    (Morton / z-order of the cell coordinates) -- mathematically equal entries then differ in their last bits
   *    wherever the visiting order of the cells around a node differs.  Block (0,0) only (scalar / Stokes). */
  int32_t assembly;
  int32_t elasticity, pad2_;
  double lame_lambda, lame_mu, lame2_lambda, lame2_mu;
  double box_lo[3], box_hi[3];
  int32_t box_cells[3], pad3_;
} alfd_synth_params;

/* NULL on failure (message in err). */
void *alfd_synth_generate(const alfd_synth_params *sp, char *err, int errlen);
void alfd_synth_free(void *h);
/* Renumbers the NODES of the background / velocity space of a generated (unpartitioned) problem: node new_to_old[k]
 * becomes node k, all components of a node stay together.  A is permuted symmetrically, Bt / Ct by rows, B / C by
 * columns (entries of a row re-sorted by the new column index), f with the rows.  What a caller's DoF renumbering
 * (Cuthill-McKee, stokes_immersed_boundary.cc:533-541) does to the operators.  0 on success. */
int alfd_synth_permute_nodes(void *h, const int64_t *new_to_old, int64_t n_nodes);
/* 0 and dims/pointers if the matrix / vector exists, else -1; pointers stay valid until alfd_synth_free. */
int alfd_synth_matrix(void *h, const char *name, int64_t *nrows, int64_t *ncols, int64_t *nnz,
                      const int64_t **row_ptr, const int32_t **col, const double **val);
int alfd_synth_vector(void *h, const char *name, int64_t *n, const double **data);
void alfd_synth_transpose(int64_t nrows, int64_t ncols, const int64_t *row_ptr, const int32_t *col,
                          const double *val, int64_t *t_row_ptr, int32_t *t_col, double *t_val);

#ifdef __cplusplus
}
#endif
#endif
