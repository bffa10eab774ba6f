// Synthetic fictitious-domain operator generator (host, OpenMP).
//
// Produces the CSR blocks that the reference obtains from deal.II FE assembly
// (reference: immersed_laplace.cc:278-496, stokes_immersed_boundary.cc:410-820,
// elliptic_interface.cc:450-670 -- OUT OF SCOPE code that only *produces inputs*
// for the hot path; see SURVEY.md section 8(d) for the concrete instances).
// deal.II is absent here, so we build structurally faithful operators on
// tensor-product grids of a box: every FE matrix on such a grid is a sum of
// Kronecker products of 1-D element matrices, which lets each CSR row be
// written in closed form, in parallel, with no global assembly step.
//
//   background: Q_p (p = 1,2) on N^dim cells of [lo,hi]^dim, scalar or vector
//               (node-major interleaved components -- SURVEY.md 8(e)),
//               homogeneous Dirichlet rows replaced by identity and Dirichlet
//               columns eliminated (what AffineConstraints does).
//   Stokes:     A = (grad u, grad v) + gamma_gd (div u, div v)
//               (stokes_immersed_boundary.cc:725-732), B = -(div u, q) with q in
//               Q_{p-1}, Mp = pressure mass.
//   elasticity: vector Q_1, A = lambda (div u, div v) + 2 mu (eps(u), eps(v))
//               (ElasticityUtilities::assemble_elasticity, utilities.h:377-427); the immersed
//               body is a 3-D box of Q1 hexahedra (elasticity.prm:55-57) carrying the jump
//               operator A2 with (lambda_2 - lambda_1, mu_2 - mu_1) and a VOLUME coupling.
//   immersed:   closed circle (P1 segments) in 2-D, cubed-sphere (Q1 quads) in
//               3-D; Ct_{(j,b),(k,b)} = int_Gamma phi_j chi_k by Gauss
//               quadrature on the immersed cells (the non-matching coupling of
//               stokes_immersed_boundary.cc:650-660), M = immersed mass,
//               K = immersed stiffness (for rational_preconditioner.h).
//
// This file is product-side input tooling: bench.py, the tests and the oracle
// all consume its output. It contains no solver arithmetic.

#include "synth.h"

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

struct Csr {
  int64_t nrows = 0, ncols = 0;
  std::vector<int64_t> row_ptr;
  std::vector<int32_t> col;
  std::vector<double> val;
  int64_t nnz() const { return row_ptr.empty() ? 0 : row_ptr.back(); }
};

struct Triplet {
  int64_t r;
  int64_t c;
  double v;
};

// Sum duplicates; columns ascending in each row.
Csr csr_from_triplets(int64_t nrows, int64_t ncols, std::vector<Triplet> &t) {
  std::sort(t.begin(), t.end(), [](const Triplet &a, const Triplet &b) {
    return a.r != b.r ? a.r < b.r : a.c < b.c;
  });
  Csr m;
  m.nrows = nrows;
  m.ncols = ncols;
  m.row_ptr.assign(nrows + 1, 0);
  for (size_t i = 0; i < t.size();) {
    size_t j = i;
    double s = 0;
    while (j < t.size() && t[j].r == t[i].r && t[j].c == t[i].c) s += t[j++].v;
    m.col.push_back((int32_t)t[i].c);
    m.val.push_back(s);
    m.row_ptr[t[i].r + 1]++;
    i = j;
  }
  for (int64_t r = 0; r < nrows; ++r) m.row_ptr[r + 1] += m.row_ptr[r];
  return m;
}

Csr csr_transpose(const Csr &a) {
  Csr t;
  t.nrows = a.ncols;
  t.ncols = a.nrows;
  t.row_ptr.assign(t.nrows + 1, 0);
  const int64_t nnz = a.nnz();
  t.col.resize(nnz);
  t.val.resize(nnz);
  for (int64_t k = 0; k < nnz; ++k) t.row_ptr[a.col[k] + 1]++;
  for (int64_t r = 0; r < t.nrows; ++r) t.row_ptr[r + 1] += t.row_ptr[r];
  std::vector<int64_t> cur(t.row_ptr.begin(), t.row_ptr.end() - 1);
  for (int64_t r = 0; r < a.nrows; ++r)
    for (int64_t k = a.row_ptr[r]; k < a.row_ptr[r + 1]; ++k) {
      const int64_t p = cur[a.col[k]]++;
      t.col[p] = (int32_t)r;
      t.val[p] = a.val[k];
    }
  return t;
}

// rows [r0, r1) of a (column indices unchanged)
Csr csr_slice_rows(const Csr &a, int64_t r0, int64_t r1) {
  Csr s;
  s.nrows = r1 - r0;
  s.ncols = a.ncols;
  s.row_ptr.resize(s.nrows + 1);
  const int64_t k0 = a.row_ptr[r0];
  for (int64_t r = r0; r <= r1; ++r) s.row_ptr[r - r0] = a.row_ptr[r] - k0;
  s.col.assign(a.col.begin() + k0, a.col.begin() + a.row_ptr[r1]);
  s.val.assign(a.val.begin() + k0, a.val.begin() + a.row_ptr[r1]);
  return s;
}

// ---------------------------------------------------------------- 1-D pieces
struct Gauss {
  std::vector<double> x, w;  // on [0,1]
  explicit Gauss(int n) {
    static const double X[6][5] = {{},
                                   {0.0},
                                   {-0.5773502691896257, 0.5773502691896257},
                                   {-0.7745966692414834, 0.0, 0.7745966692414834},
                                   {-0.8611363115940526, -0.3399810435848563, 0.3399810435848563,
                                    0.8611363115940526},
                                   {-0.9061798459386640, -0.5384693101056831, 0.0, 0.5384693101056831,
                                    0.9061798459386640}};
    static const double W[6][5] = {{},
                                   {2.0},
                                   {1.0, 1.0},
                                   {0.5555555555555556, 0.8888888888888888, 0.5555555555555556},
                                   {0.3478548451374538, 0.6521451548625461, 0.6521451548625461,
                                    0.3478548451374538},
                                   {0.2369268850561891, 0.4786286704993665, 0.5688888888888889,
                                    0.4786286704993665, 0.2369268850561891}};
    for (int i = 0; i < n; ++i) {
      x.push_back(0.5 * (X[n][i] + 1.0));
      w.push_back(0.5 * W[n][i]);
    }
  }
};

// Lagrange shape functions of degree p (1 or 2) on [0,1], nodes equispaced and
// numbered left to right; d = their derivatives.
inline void shape1d(int p, double xi, double *v, double *d) {
  if (p == 1) {
    v[0] = 1 - xi;
    v[1] = xi;
    d[0] = -1;
    d[1] = 1;
  } else {
    v[0] = 2 * (xi - 0.5) * (xi - 1);
    v[1] = 4 * xi * (1 - xi);
    v[2] = 2 * xi * (xi - 0.5);
    d[0] = 4 * xi - 3;
    d[1] = 4 - 8 * xi;
    d[2] = 4 * xi - 1;
  }
}

// Banded 1-D matrix between a Q_pr row space and a Q_pc column space on the
// same N cells: entry (i,j) stored at band[i*W + (j - i*pc/pr... )]; we simply
// store a dense (n_r x W) window starting at column first(i).
struct Band1D {
  int pr, pc, N, nr, nc, W;
  std::vector<double> a;     // nr * W
  std::vector<int> first;    // first column of row i's window
  std::vector<int> count;    // window length (structural couplings)
  double at(int i, int j) const { return a[(size_t)i * W + (j - first[i])]; }
};

// kind: 0 mass (u v), 1 stiffness (u' v'), 2 G (row' col): int phi_i' phi_j
Band1D band1d(int pr, int pc, int N, double h, int kind) {
  Band1D b;
  b.pr = pr;
  b.pc = pc;
  b.N = N;
  b.nr = pr * N + 1;
  b.nc = pc * N + 1;
  b.W = 2 * pc + 1;
  b.a.assign((size_t)b.nr * b.W, 0.0);
  b.first.resize(b.nr);
  b.count.resize(b.nr);
  for (int i = 0; i < b.nr; ++i) {
    int c_lo, c_hi;  // cells touching row node i
    if (i % pr == 0) {
      c_lo = std::max(0, i / pr - 1);
      c_hi = std::min(N - 1, i / pr);
    } else {
      c_lo = c_hi = i / pr;
    }
    b.first[i] = c_lo * pc;
    b.count[i] = (c_hi + 1) * pc - c_lo * pc + 1;
  }
  Gauss q(4);
  double vr[3], dr[3], vc[3], dc[3];
  for (int c = 0; c < N; ++c)
    for (size_t g = 0; g < q.x.size(); ++g) {
      shape1d(pr, q.x[g], vr, dr);
      shape1d(pc, q.x[g], vc, dc);
      for (int li = 0; li <= pr; ++li)
        for (int lj = 0; lj <= pc; ++lj) {
          const int i = c * pr + li, j = c * pc + lj;
          double e;
          if (kind == 0)
            e = vr[li] * vc[lj] * h;
          else if (kind == 1)
            e = dr[li] * dc[lj] / h;
          else
            e = dr[li] * vc[lj];  // (1/h) * h
          b.a[(size_t)i * b.W + (j - b.first[i])] += e * q.w[g];
        }
    }
  return b;
}

// ------------------------------------------------------------ the generator
struct Params {
  int dim = 2, degree = 1, ncomp = 1, n_cells = 8;
  double lo = 0, hi = 1;
  int stokes = 0, grad_div = 0;
  double gamma_grad_div = 0;
  double beta = 1;
  double center[3] = {0.5, 0.5, 0.5};
  double radius = 0.2;
  int immersed_refine = 3;
  int coupling_nq = 3;
  double body_force[3] = {0, 0, 0};
  double embedded_value[3] = {1, 0, 0};
  // immersed_kind 1: the immersed domain is a BOX meshed with imm_cells^dim Q1
  // cells (elliptic_interface: Omega_2 = [-0.14, 0.47]^2, parameters_modified.prm:53-56);
  // the coupling is then a VOLUME integral over Omega_2 and A2 = beta2 * stiffness.
  int immersed_kind = 0;
  double imm_lo = 0.25, imm_hi = 0.75;
  int imm_cells = 8;
  double beta2 = 0.0;
  // also emit G_ij = int_Gamma phi_i phi_j on the background space (the particle-assembled
  // AL term of the "operator form", immersed_laplace.cc:659-705)
  int want_surface_mass = 0;
  int assembly = 0;  // 1: cell-wise sums of one numerically integrated cell matrix in Morton order (synth.h)
  // linear elasticity (BASELINE cfg 5): background Lame parameters; lame2_* = the JUMP
  // (immersed minus background) that A2 carries, as beta_2 - beta_1 does in the scalar case
  // (elliptic_interface.cc:648-663).  immersed_kind 2: 3-D box [box_lo, box_hi] meshed with
  // box_cells[0] x box_cells[1] x box_cells[2] trilinear cells.
  int elasticity = 0;
  double lame_lambda = 2.0, lame_mu = 1.0, lame2_lambda = 18.0, lame2_mu = 9.0;
  double box_lo[3] = {-0.65, -0.3, -0.4}, box_hi[3] = {0.65, 0.3, 0.4};
  int box_cells[3] = {2, 2, 2};
  // row ranges of this process (multi-GPU row partition); -1 = everything.
  // u/p ranges are in NODES (z-slabs of the lexicographic numbering), l in dofs.
  int64_t u_node0 = -1, u_node1 = -1, p_node0 = -1, p_node1 = -1, l0 = -1, l1 = -1;
};

struct Problem {
  Params p;
  std::map<std::string, Csr> mats;
  std::map<std::string, std::vector<double>> vecs;
  std::string err;
};

struct Grid {
  int dim, p, n1, N;
  double lo, h;
  int64_t nnodes;
  bool boundary(const int *idx) const {
    for (int a = 0; a < dim; ++a)
      if (idx[a] == 0 || idx[a] == n1 - 1) return true;
    return false;
  }
  int64_t node(const int *idx) const {
    int64_t r = 0;
    for (int a = dim - 1; a >= 0; --a) r = r * n1 + idx[a];
    return r;
  }
  void split(int64_t n, int *idx) const {
    for (int a = 0; a < dim; ++a) {
      idx[a] = (int)(n % n1);
      n /= n1;
    }
  }
};

// ---- cell-wise assembly (Params::assembly == 1) ------------------------------------------------
// The local matrix of ONE cell, integrated as an FE library does: tensor Gauss quadrature with p + 2 points per
// direction, Jacobian (diagonal) from the vertex coordinates of cell 0, shape gradients J^-T grad_hat, the
// quadrature sum accumulated point by point.  Index: ((local node) * nc + a) x ((local node) * nc + b),
// local node = (l2 * (p+1) + l1) * (p+1) + l0.
struct CellMatrix {
  int nl, nc;
  std::vector<double> k;
  double at(int li, int a, int lj, int b) const { return k[((size_t)li * nc + a) * (nl * nc) + (size_t)lj * nc + b]; }
};
CellMatrix cell_matrix(const Params &P, const Grid &g) {
  const int dim = g.dim, p = g.p, n1l = p + 1, nc = P.ncomp;
  CellMatrix C;
  C.nl = dim == 3 ? n1l * n1l * n1l : n1l * n1l;
  C.nc = nc;
  C.k.assign((size_t)C.nl * nc * C.nl * nc, 0.0);
  // vertex coordinates of cell 0 along each axis -> Jacobian entries (as a mapping would compute them)
  const double x0 = g.lo, x1 = g.lo + g.h;
  const double jac = x1 - x0, jinv = 1.0 / jac;
  Gauss q(p + 2);
  const int nq = (int)q.x.size();
  const double ggd = (P.stokes && P.grad_div) ? P.gamma_grad_div : 0.0;
  std::vector<double> grad((size_t)C.nl * 3);
  double v[3][3], d[3][3];
  for (int q2 = 0; q2 < (dim == 3 ? nq : 1); ++q2)
    for (int q1 = 0; q1 < nq; ++q1)
      for (int q0 = 0; q0 < nq; ++q0) {
        shape1d(p, q.x[q0], v[0], d[0]);
        shape1d(p, q.x[q1], v[1], d[1]);
        if (dim == 3) shape1d(p, q.x[q2], v[2], d[2]);
        double jxw = q.w[q0] * q.w[q1] * (dim == 3 ? q.w[q2] : 1.0);
        for (int a = 0; a < dim; ++a) jxw *= jac;
        for (int l = 0; l < C.nl; ++l) {
          const int l0 = l % n1l, l1 = (l / n1l) % n1l, l2 = l / (n1l * n1l);
          const double s0 = v[0][l0], s1 = v[1][l1], s2 = dim == 3 ? v[2][l2] : 1.0;
          grad[l * 3 + 0] = jinv * (d[0][l0] * s1 * s2);
          grad[l * 3 + 1] = jinv * (s0 * d[1][l1] * s2);
          grad[l * 3 + 2] = dim == 3 ? jinv * (s0 * s1 * d[2][l2]) : 0.0;
        }
        for (int li = 0; li < C.nl; ++li)
          for (int lj = 0; lj < C.nl; ++lj) {
            double gg = 0.0;
            for (int a = 0; a < dim; ++a) gg += grad[li * 3 + a] * grad[lj * 3 + a];
            for (int a = 0; a < nc; ++a)
              for (int b = 0; b < nc; ++b) {
                double e = (a == b) ? P.beta * gg : 0.0;
                if (nc > 1 && ggd != 0.0) e += ggd * (grad[li * 3 + a] * grad[lj * 3 + b]);   // (div u, div v)
                C.k[((size_t)li * nc + a) * (C.nl * nc) + (size_t)lj * nc + b] += e * jxw;
              }
          }
      }
  return C;
}
inline uint64_t morton3(uint32_t x, uint32_t y, uint32_t z) {
  auto spread = [](uint64_t v) {
    v &= 0x1fffff;
    v = (v | v << 32) & 0x1f00000000ffffull;
    v = (v | v << 16) & 0x1f0000ff0000ffull;
    v = (v | v << 8) & 0x100f00f00f00f00full;
    v = (v | v << 4) & 0x10c30c30c30c30c3ull;
    v = (v | v << 2) & 0x1249249249249249ull;
    return v;
  };
  return spread(x) | spread(y) << 1 | spread(z) << 2;
}

// Velocity / background block. For ncomp == 1: beta * stiffness. For
// ncomp == dim: vector Laplace (+ gamma_gd * grad-div).
void build_A(const Params &P, const Grid &g, Csr &A, int64_t node0, int64_t node1) {
  const int dim = g.dim, nc = P.ncomp;
  const Band1D M = band1d(g.p, g.p, g.N, g.h, 0), K = band1d(g.p, g.p, g.N, g.h, 1),
               G = band1d(g.p, g.p, g.N, g.h, 2);
  const int64_t nrows = (node1 - node0) * nc;
  A.nrows = nrows;
  A.ncols = g.nnodes * nc;
  A.row_ptr.assign(nrows + 1, 0);
  // pass 1: counts
#pragma omp parallel for schedule(static)
  for (int64_t n = node0; n < node1; ++n) {
    int idx[3] = {0, 0, 0};
    g.split(n, idx);
    int64_t cnt;
    if (g.boundary(idx)) {
      cnt = 1;
    } else {
      cnt = 0;
      int j[3] = {0, 0, 0};
      const int f0 = M.first[idx[0]], c0 = M.count[idx[0]];
      const int f1 = M.first[idx[1]], c1 = M.count[idx[1]];
      const int f2 = dim == 3 ? M.first[idx[2]] : 0, c2 = dim == 3 ? M.count[idx[2]] : 1;
      for (int k2 = 0; k2 < c2; ++k2)
        for (int k1 = 0; k1 < c1; ++k1)
          for (int k0 = 0; k0 < c0; ++k0) {
            j[0] = f0 + k0;
            j[1] = f1 + k1;
            j[2] = f2 + k2;
            if (!g.boundary(j)) cnt += nc;
          }
    }
    for (int a = 0; a < nc; ++a) A.row_ptr[(n - node0) * nc + a + 1] = cnt;
  }
  for (int64_t r = 0; r < nrows; ++r) A.row_ptr[r + 1] += A.row_ptr[r];
  A.col.resize(A.row_ptr[nrows]);
  A.val.resize(A.row_ptr[nrows]);
  const double ggd = (P.stokes && P.grad_div) ? P.gamma_grad_div : 0.0;
  const bool cellwise = P.assembly == 1 && !P.elasticity;
  CellMatrix KC;
  if (cellwise) KC = cell_matrix(P, g);
  // pass 2: fill
#pragma omp parallel for schedule(static)
  for (int64_t n = node0; n < node1; ++n) {
    int idx[3] = {0, 0, 0};
    g.split(n, idx);
    if (g.boundary(idx)) {
      for (int a = 0; a < nc; ++a) {
        const int64_t p0 = A.row_ptr[(n - node0) * nc + a];
        A.col[p0] = (int32_t)(n * nc + a);
        A.val[p0] = 1.0;
      }
      continue;
    }
    int64_t pos[3];
    for (int a = 0; a < nc; ++a) pos[a] = A.row_ptr[(n - node0) * nc + a];
    int j[3] = {0, 0, 0};
    const int f0 = M.first[idx[0]], c0 = M.count[idx[0]];
    const int f1 = M.first[idx[1]], c1 = M.count[idx[1]];
    const int f2 = dim == 3 ? M.first[idx[2]] : 0, c2 = dim == 3 ? M.count[idx[2]] : 1;
    for (int k2 = 0; k2 < c2; ++k2)
      for (int k1 = 0; k1 < c1; ++k1)
        for (int k0 = 0; k0 < c0; ++k0) {
          j[0] = f0 + k0;
          j[1] = f1 + k1;
          j[2] = f2 + k2;
          if (g.boundary(j)) continue;
          if (cellwise) {
            // cells holding both nodes, visited in Morton order; global entry = running sum of their contributions
            int clo[3] = {0, 0, 0}, chi[3] = {0, 0, 0};
            for (int a = 0; a < dim; ++a) {
              const int ilo = idx[a] % g.p == 0 ? std::max(0, idx[a] / g.p - 1) : idx[a] / g.p;
              const int ihi = idx[a] % g.p == 0 ? std::min(g.N - 1, idx[a] / g.p) : idx[a] / g.p;
              const int jlo = j[a] % g.p == 0 ? std::max(0, j[a] / g.p - 1) : j[a] / g.p;
              const int jhi = j[a] % g.p == 0 ? std::min(g.N - 1, j[a] / g.p) : j[a] / g.p;
              clo[a] = std::max(ilo, jlo);
              chi[a] = std::min(ihi, jhi);
            }
            struct CellRef {
              uint64_t key;
              int c[3];
            } cells[8];
            int ncell = 0;
            for (int c2 = clo[2]; c2 <= chi[2]; ++c2)
              for (int c1 = clo[1]; c1 <= chi[1]; ++c1)
                for (int c0 = clo[0]; c0 <= chi[0]; ++c0) {
                  cells[ncell].key = morton3((uint32_t)c0, (uint32_t)c1, (uint32_t)c2);
                  cells[ncell].c[0] = c0, cells[ncell].c[1] = c1, cells[ncell].c[2] = c2;
                  ++ncell;
                }
            std::sort(cells, cells + ncell, [](const CellRef &x, const CellRef &y) { return x.key < y.key; });
            const int64_t jn = g.node(j);
            const int n1l = g.p + 1;
            double acc[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int ci = 0; ci < ncell; ++ci) {
              int li = 0, lj = 0;
              for (int a = dim - 1; a >= 0; --a) {
                li = li * n1l + (idx[a] - cells[ci].c[a] * g.p);
                lj = lj * n1l + (j[a] - cells[ci].c[a] * g.p);
              }
              for (int a = 0; a < nc; ++a)
                for (int b = 0; b < nc; ++b) acc[a][b] += KC.at(li, a, lj, b);
            }
            for (int a = 0; a < nc; ++a)
              for (int b = 0; b < nc; ++b) {
                A.col[pos[a]] = (int32_t)(jn * nc + b);
                A.val[pos[a]++] = acc[a][b];
              }
            continue;
          }
          // per axis: M, K, G[i][j], G[j][i]
          double m[3] = {1, 1, 1}, k[3] = {0, 0, 0}, gij[3] = {0, 0, 0}, gji[3] = {0, 0, 0};
          for (int a = 0; a < dim; ++a) {
            m[a] = M.at(idx[a], j[a]);
            k[a] = K.at(idx[a], j[a]);
            gij[a] = G.at(idx[a], j[a]);
            gji[a] = G.at(j[a], idx[a]);
          }
          if (dim == 2) m[2] = 1.0;
          const double lap = k[0] * m[1] * m[2] + m[0] * k[1] * m[2] +
                             (dim == 3 ? m[0] * m[1] * k[2] : 0.0);
          const int64_t jn = g.node(j);
          if (nc == 1) {
            A.col[pos[0]] = (int32_t)jn;
            A.val[pos[0]++] = P.beta * lap;
          } else if (P.elasticity) {
            // lambda int d_a phi_i d_b phi_j + mu (delta_ab grad phi_i . grad phi_j + int d_b phi_i d_a phi_j)
            for (int a = 0; a < nc; ++a)
              for (int b = 0; b < nc; ++b) {
                double tab = 1.0, tba = 1.0;
                for (int c = 0; c < dim; ++c) {
                  if (a == b) {
                    tab *= (c == a) ? k[c] : m[c];
                  } else {
                    tab *= (c == a) ? gij[c] : (c == b) ? gji[c] : m[c];
                    tba *= (c == b) ? gij[c] : (c == a) ? gji[c] : m[c];
                  }
                }
                if (a == b) tba = tab;
                const double v = P.lame_lambda * tab + P.lame_mu * ((a == b ? lap : 0.0) + tba);
                A.col[pos[a]] = (int32_t)(jn * nc + b);
                A.val[pos[a]++] = v;
              }
          } else {
            for (int a = 0; a < nc; ++a)
              for (int b = 0; b < nc; ++b) {
                double v = (a == b) ? P.beta * lap : 0.0;
                if (ggd != 0.0) {
                  double t;
                  if (a == b) {
                    t = 1.0;
                    for (int c = 0; c < dim; ++c) t *= (c == a) ? k[c] : m[c];
                  } else {
                    // int d_a phi_i d_b phi_j
                    t = 1.0;
                    for (int c = 0; c < dim; ++c)
                      t *= (c == a) ? gij[c] : (c == b) ? gji[c] : m[c];
                  }
                  v += ggd * t;
                }
                A.col[pos[a]] = (int32_t)(jn * nc + b);
                A.val[pos[a]++] = v;
              }
          }
        }
  }
}

// B = -(div u, q), rows = Q_{p-1} pressure nodes, cols = velocity dofs.
// Mp = pressure mass (no constraints on pressure).
void build_B_Mp(const Grid &gu, Csr &B, Csr &Mp, int64_t &n_p, int64_t pn0, int64_t pn1,
                bool want_mp) {
  const int dim = gu.dim, pp = gu.p - 1, N = gu.N;
  Grid gp = gu;
  gp.p = pp;
  gp.n1 = pp * N + 1;
  gp.nnodes = 1;
  for (int a = 0; a < dim; ++a) gp.nnodes *= gp.n1;
  n_p = gp.nnodes;
  if (pn0 < 0) {
    pn0 = 0;
    pn1 = n_p;
  }
  const int64_t nloc = pn1 - pn0;
  const Band1D MX = band1d(pp, gu.p, N, gu.h, 0);  // psi_i phi_j
  // GX[i][j] = int psi_i phi_j'  -> use kind 2 with roles swapped: build
  // (row = Q_p velocity)' x (col = pressure) and read transposed.
  const Band1D GT = band1d(gu.p, pp, N, gu.h, 2);  // GT[j][i] = int phi_j' psi_i
  const Band1D MPP = band1d(pp, pp, N, gu.h, 0);
  // ---- B
  B.nrows = nloc;
  B.ncols = gu.nnodes * dim;
  B.row_ptr.assign(nloc + 1, 0);
  auto for_row = [&](int64_t n, auto &&emit) {
    int idx[3] = {0, 0, 0};
    gp.split(n, idx);
    int j[3] = {0, 0, 0};
    const int f0 = MX.first[idx[0]], c0 = MX.count[idx[0]];
    const int f1 = MX.first[idx[1]], c1 = MX.count[idx[1]];
    const int f2 = dim == 3 ? MX.first[idx[2]] : 0, c2 = dim == 3 ? MX.count[idx[2]] : 1;
    for (int k2 = 0; k2 < c2; ++k2)
      for (int k1 = 0; k1 < c1; ++k1)
        for (int k0 = 0; k0 < c0; ++k0) {
          j[0] = f0 + k0;
          j[1] = f1 + k1;
          j[2] = f2 + k2;
          if (gu.boundary(j)) continue;
          emit(idx, j);
        }
  };
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nloc; ++n) {
    int64_t cnt = 0;
    for_row(n + pn0, [&](const int *, const int *) { cnt += dim; });
    B.row_ptr[n + 1] = cnt;
  }
  for (int64_t r = 0; r < nloc; ++r) B.row_ptr[r + 1] += B.row_ptr[r];
  B.col.resize(B.row_ptr[nloc]);
  B.val.resize(B.row_ptr[nloc]);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nloc; ++n) {
    int64_t pos = B.row_ptr[n];
    for_row(n + pn0, [&](const int *idx, const int *j) {
      const int64_t jn = gu.node(j);
      for (int b = 0; b < dim; ++b) {
        double t = 1.0;
        for (int c = 0; c < dim; ++c) {
          if (c == b) {
            // int psi_i phi_j' : stored in GT at (j, i) window
            const int off = idx[c] - GT.first[j[c]];
            t *= (off >= 0 && off < GT.count[j[c]]) ? GT.a[(size_t)j[c] * GT.W + off] : 0.0;
          } else {
            t *= MX.at(idx[c], j[c]);
          }
        }
        B.col[pos] = (int32_t)(jn * dim + b);
        B.val[pos++] = -t;
      }
    });
  }
  // ---- Mp
  if (!want_mp) return;
  Mp.nrows = nloc;
  Mp.ncols = n_p;
  Mp.row_ptr.assign(nloc + 1, 0);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nloc; ++n) {
    int idx[3] = {0, 0, 0};
    gp.split(n + pn0, idx);
    int64_t cnt = 1;
    for (int a = 0; a < dim; ++a) cnt *= MPP.count[idx[a]];
    Mp.row_ptr[n + 1] = cnt;
  }
  for (int64_t r = 0; r < nloc; ++r) Mp.row_ptr[r + 1] += Mp.row_ptr[r];
  Mp.col.resize(Mp.row_ptr[nloc]);
  Mp.val.resize(Mp.row_ptr[nloc]);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nloc; ++n) {
    int idx[3] = {0, 0, 0};
    gp.split(n + pn0, idx);
    int64_t pos = Mp.row_ptr[n];
    int j[3] = {0, 0, 0};
    const int f0 = MPP.first[idx[0]], c0 = MPP.count[idx[0]];
    const int f1 = MPP.first[idx[1]], c1 = MPP.count[idx[1]];
    const int f2 = dim == 3 ? MPP.first[idx[2]] : 0, c2 = dim == 3 ? MPP.count[idx[2]] : 1;
    for (int k2 = 0; k2 < c2; ++k2)
      for (int k1 = 0; k1 < c1; ++k1)
        for (int k0 = 0; k0 < c0; ++k0) {
          j[0] = f0 + k0;
          j[1] = f1 + k1;
          j[2] = f2 + k2;
          double t = MPP.at(idx[0], j[0]) * MPP.at(idx[1], j[1]);
          if (dim == 3) t *= MPP.at(idx[2], j[2]);
          Mp.col[pos] = (int32_t)gp.node(j);
          Mp.val[pos++] = t;
        }
  }
}

// Immersed mesh: nodes (3 coords each, z = 0 in 2-D) and cells (2 or 4 nodes).
struct Immersed {
  int cell_nodes;  // 2 (segment) or 4 (quad)
  std::vector<double> xyz;
  std::vector<int> cells;
  int64_t nnodes() const { return (int64_t)xyz.size() / 3; }
  int64_t ncells() const { return (int64_t)cells.size() / cell_nodes; }
};

Immersed make_circle(const Params &P) {
  Immersed im;
  im.cell_nodes = 2;
  // 4 * 2^refine segments (GridGenerator::hyper_sphere<1,2> refined), or exactly imm_cells segments
  const int n = (P.immersed_kind == 0 && P.imm_cells > 0) ? P.imm_cells : 4 << P.immersed_refine;
  for (int i = 0; i < n; ++i) {
    const double th = 2.0 * M_PI * i / n;
    im.xyz.push_back(P.center[0] + P.radius * std::cos(th));
    im.xyz.push_back(P.center[1] + P.radius * std::sin(th));
    im.xyz.push_back(0.0);
  }
  for (int i = 0; i < n; ++i) {
    im.cells.push_back(i);
    im.cells.push_back((i + 1) % n);
  }
  return im;
}

// Cubed sphere: each cube face split m x m (m = 2^refine), vertices projected
// radially (what SphericalManifold refinement of GridGenerator::hyper_sphere
// produces up to point distribution).
Immersed make_cubed_sphere(const Params &P) {
  Immersed im;
  im.cell_nodes = 4;
  const int m = 1 << P.immersed_refine;
  std::map<std::array<int, 3>, int> ids;
  auto node = [&](int i, int j, int k) {
    std::array<int, 3> key = {i, j, k};
    auto it = ids.find(key);
    if (it != ids.end()) return it->second;
    const int id = (int)ids.size();
    ids[key] = id;
    double x = 2.0 * i / m - 1, y = 2.0 * j / m - 1, z = 2.0 * k / m - 1;
    const double r = std::sqrt(x * x + y * y + z * z);
    im.xyz.push_back(P.center[0] + P.radius * x / r);
    im.xyz.push_back(P.center[1] + P.radius * y / r);
    im.xyz.push_back(P.center[2] + P.radius * z / r);
    return id;
  };
  for (int face = 0; face < 6; ++face) {
    const int axis = face / 2, side = (face % 2) * m;
    for (int a = 0; a < m; ++a)
      for (int b = 0; b < m; ++b) {
        int q[4];
        for (int v = 0; v < 4; ++v) {
          const int da = (v == 1 || v == 2) ? 1 : 0, db = (v >= 2) ? 1 : 0;
          int c[3];
          c[axis] = side;
          c[(axis + 1) % 3] = a + da;
          c[(axis + 2) % 3] = b + db;
          q[v] = node(c[0], c[1], c[2]);
        }
        // counter-clockwise ordering v0,v1,v2,v3 around the quad
        for (int v = 0; v < 4; ++v) im.cells.push_back(q[v]);
      }
  }
  return im;
}

// 2-D box [lo,hi]^2 meshed with m x m bilinear cells (nodes lexicographic, x fastest).
Immersed make_box_2d(const Params &P) {
  Immersed im;
  im.cell_nodes = 4;
  const int m = P.imm_cells;
  const double h = (P.imm_hi - P.imm_lo) / m;
  for (int j = 0; j <= m; ++j)
    for (int i = 0; i <= m; ++i) {
      im.xyz.push_back(P.imm_lo + i * h);
      im.xyz.push_back(P.imm_lo + j * h);
      im.xyz.push_back(0.0);
    }
  for (int j = 0; j < m; ++j)
    for (int i = 0; i < m; ++i) {
      const int n0 = j * (m + 1) + i;
      im.cells.push_back(n0);
      im.cells.push_back(n0 + 1);
      im.cells.push_back(n0 + m + 2);
      im.cells.push_back(n0 + m + 1);
    }
  return im;
}

// One quadrature point on an immersed cell.
struct QPoint {
  const int *cn;      // cell node ids
  double sh[4];       // immersed shape values
  double gr[4][2];    // reference derivatives (d/ds, d/dt); segments: d/darclength in [0]
  double minv[3];     // inverse surface metric (quads): G^{-1} = [[0],[1]],[[1],[2]]
  double x[3];        // physical point
  double JxW;
};

template <class F>
void immersed_quadrature(const Immersed &im, int nq, F &&f) {
  Gauss q(nq);
  QPoint qp;
  for (int64_t c = 0; c < im.ncells(); ++c) {
    const int *cn = &im.cells[c * im.cell_nodes];
    qp.cn = cn;
    if (im.cell_nodes == 2) {
      const double *a = &im.xyz[3 * cn[0]], *b = &im.xyz[3 * cn[1]];
      const double len = std::sqrt((b[0] - a[0]) * (b[0] - a[0]) + (b[1] - a[1]) * (b[1] - a[1]));
      for (size_t g = 0; g < q.x.size(); ++g) {
        const double s = q.x[g];
        qp.sh[0] = 1 - s;
        qp.sh[1] = s;
        qp.sh[2] = qp.sh[3] = 0;
        for (int v = 0; v < 4; ++v) qp.gr[v][0] = qp.gr[v][1] = 0;
        qp.gr[0][0] = -1 / len;
        qp.gr[1][0] = 1 / len;
        qp.minv[0] = 1;
        qp.minv[1] = 0;
        qp.minv[2] = 0;
        qp.x[0] = a[0] + s * (b[0] - a[0]);
        qp.x[1] = a[1] + s * (b[1] - a[1]);
        qp.x[2] = 0;
        qp.JxW = q.w[g] * len;
        f(qp);
      }
    } else {
      const double *X[4] = {&im.xyz[3 * cn[0]], &im.xyz[3 * cn[1]], &im.xyz[3 * cn[2]],
                            &im.xyz[3 * cn[3]]};
      for (size_t g0 = 0; g0 < q.x.size(); ++g0)
        for (size_t g1 = 0; g1 < q.x.size(); ++g1) {
          const double s = q.x[g0], t = q.x[g1];
          // bilinear map, corner order (0,0),(1,0),(1,1),(0,1)
          const double sh[4] = {(1 - s) * (1 - t), s * (1 - t), s * t, (1 - s) * t};
          const double ds[4] = {-(1 - t), (1 - t), t, -t}, dt[4] = {-(1 - s), -s, s, (1 - s)};
          double xs[3] = {0, 0, 0}, xt[3] = {0, 0, 0};
          qp.x[0] = qp.x[1] = qp.x[2] = 0;
          for (int v = 0; v < 4; ++v)
            for (int d = 0; d < 3; ++d) {
              qp.x[d] += sh[v] * X[v][d];
              xs[d] += ds[v] * X[v][d];
              xt[d] += dt[v] * X[v][d];
            }
          const double E = xs[0] * xs[0] + xs[1] * xs[1] + xs[2] * xs[2];
          const double Fm = xs[0] * xt[0] + xs[1] * xt[1] + xs[2] * xt[2];
          const double Gm = xt[0] * xt[0] + xt[1] * xt[1] + xt[2] * xt[2];
          const double det = E * Gm - Fm * Fm;
          for (int v = 0; v < 4; ++v) {
            qp.sh[v] = sh[v];
            qp.gr[v][0] = ds[v];
            qp.gr[v][1] = dt[v];
          }
          qp.minv[0] = Gm / det;
          qp.minv[1] = -Fm / det;
          qp.minv[2] = E / det;
          qp.JxW = q.w[g0] * q.w[g1] * std::sqrt(det);
          f(qp);
        }
    }
  }
}

void build_immersed(const Params &P, const Grid &g, Problem &pb) {
  const int dim = g.dim, nc = P.ncomp;
  Immersed im = P.immersed_kind == 1 ? make_box_2d(P) : (dim == 2) ? make_circle(P) : make_cubed_sphere(P);
  const int64_t nl = im.nnodes();
  std::vector<Triplet> tc, tm, tk, tg;
  std::vector<std::pair<int64_t, double>> qnodes;
  std::vector<double> gint(nl, 0.0);  // int chi_k
  const int p = g.p;
  immersed_quadrature(im, P.coupling_nq, [&](const QPoint &qp) {
    const int *cn = qp.cn;
    const double *sh = qp.sh, *x = qp.x;
    const double JxW = qp.JxW;
    // immersed mass / stiffness
    for (int a = 0; a < im.cell_nodes; ++a) {
      gint[cn[a]] += sh[a] * JxW;
      for (int b = 0; b < im.cell_nodes; ++b) {
        tm.push_back({cn[a], cn[b], sh[a] * sh[b] * JxW});
        const double gg =
            qp.gr[a][0] * (qp.minv[0] * qp.gr[b][0] + qp.minv[1] * qp.gr[b][1]) +
            qp.gr[a][1] * (qp.minv[1] * qp.gr[b][0] + qp.minv[2] * qp.gr[b][1]);
        tk.push_back({cn[a], cn[b], gg * JxW});
      }
    }
    // locate background cell, evaluate tensor shape functions
    int cell[3] = {0, 0, 0};
    double v1[3][3], d1[3][3];
    for (int a = 0; a < dim; ++a) {
      double s = (x[a] - g.lo) / g.h;
      int c = (int)std::floor(s);
      c = std::max(0, std::min(g.N - 1, c));
      cell[a] = c;
      shape1d(p, s - c, v1[a], d1[a]);
    }
    int l[3] = {0, 0, 0}, j[3] = {0, 0, 0};
    qnodes.clear();
    for (l[2] = 0; l[2] <= (dim == 3 ? p : 0); ++l[2])
      for (l[1] = 0; l[1] <= p; ++l[1])
        for (l[0] = 0; l[0] <= p; ++l[0]) {
          double phi = v1[0][l[0]] * v1[1][l[1]];
          if (dim == 3) phi *= v1[2][l[2]];
          for (int a = 0; a < dim; ++a) j[a] = cell[a] * p + l[a];
          if (g.boundary(j)) continue;
          const int64_t jn = g.node(j);
          for (int a = 0; a < im.cell_nodes; ++a) tc.push_back({cn[a], jn, phi * sh[a] * JxW});
          if (P.want_surface_mass) qnodes.emplace_back(jn, phi);
        }
    if (P.want_surface_mass)
      for (const auto &na : qnodes)
        for (const auto &nb : qnodes) tg.push_back({na.first, nb.first, na.second * nb.second * JxW});
  });
  Csr Cs = csr_from_triplets(nl, g.nnodes, tc);  // scalar C
  Csr Ms = csr_from_triplets(nl, nl, tm);
  Csr Ks = csr_from_triplets(nl, nl, tk);
  // expand to nc interleaved components
  auto expand = [&](const Csr &s, int64_t ncols_scalar) {
    Csr e;
    e.nrows = s.nrows * nc;
    e.ncols = ncols_scalar * nc;
    e.row_ptr.assign(e.nrows + 1, 0);
    for (int64_t r = 0; r < s.nrows; ++r)
      for (int b = 0; b < nc; ++b)
        e.row_ptr[r * nc + b + 1] = s.row_ptr[r + 1] - s.row_ptr[r];
    for (int64_t r = 0; r < e.nrows; ++r) e.row_ptr[r + 1] += e.row_ptr[r];
    e.col.resize(e.row_ptr[e.nrows]);
    e.val.resize(e.row_ptr[e.nrows]);
    for (int64_t r = 0; r < s.nrows; ++r)
      for (int b = 0; b < nc; ++b) {
        int64_t pos = e.row_ptr[r * nc + b];
        for (int64_t k = s.row_ptr[r]; k < s.row_ptr[r + 1]; ++k) {
          e.col[pos] = (int32_t)((int64_t)s.col[k] * nc + b);
          e.val[pos++] = s.val[k];
        }
      }
    return e;
  };
  Csr C = expand(Cs, g.nnodes);
  Csr Ct = csr_transpose(C);
  Csr Mx = expand(Ms, nl), Kx = expand(Ks, nl);
  std::vector<double> gv(nl * nc);
  for (int64_t k = 0; k < nl; ++k)
    for (int b = 0; b < nc; ++b) gv[k * nc + b] = P.embedded_value[b] * gint[k];
  pb.vecs["n_lambda_global"] = {(double)(nl * nc)};
  if (P.want_surface_mass) {
    Csr Gs = csr_from_triplets(g.nnodes, g.nnodes, tg);
    pb.mats["G"] = expand(Gs, g.nnodes);
  }
  if (P.immersed_kind == 1) {
    // elliptic_interface: A2 = (beta_2 - beta_1) (grad, grad) on Omega_2 (elliptic...:681),
    // f2 = int (f_2 - f) chi_k with f_2 - f = 1 (parameters_modified.prm:25-29)
    Csr A2 = Kx;
    for (auto &v : A2.val) v *= P.beta2;
    pb.mats["A2"] = std::move(A2);
    std::vector<double> f2(nl * nc);
    for (int64_t k = 0; k < nl; ++k)
      for (int b = 0; b < nc; ++b) f2[k * nc + b] = gint[k];
    pb.vecs["f2"] = std::move(f2);
  }
  if (P.u_node0 >= 0) {
    pb.mats["Ct"] = csr_slice_rows(Ct, P.u_node0 * nc, P.u_node1 * nc);
    pb.mats["C"] = csr_slice_rows(C, P.l0, P.l1);
    pb.mats["M"] = csr_slice_rows(Mx, P.l0, P.l1);
    pb.mats["K"] = csr_slice_rows(Kx, P.l0, P.l1);
    gv = std::vector<double>(gv.begin() + P.l0, gv.begin() + P.l1);
    if (pb.mats.count("A2")) {   // elliptic interface: the immersed block shares the multiplier's row range
      pb.mats["A2"] = csr_slice_rows(pb.mats["A2"], P.l0, P.l1);
      std::vector<double> &f2 = pb.vecs["f2"];
      f2 = std::vector<double>(f2.begin() + P.l0, f2.begin() + P.l1);
    }
  } else {
    pb.mats["Ct"] = std::move(Ct);
    pb.mats["C"] = std::move(C);
    pb.mats["M"] = std::move(Mx);
    pb.mats["K"] = std::move(Kx);
  }
  pb.vecs["g"] = std::move(gv);
  pb.vecs["immersed_xyz"] = im.xyz;
}

// immersed_kind 2: the immersed body is an axis-aligned 3-D box meshed with trilinear cells
// (elasticity.prm:56-57: hyper_rectangle).  Everything on it is a tensor-product operator, so
// M, K and the elasticity jump operator A2 are written row by row from 1-D element matrices
// (no Dirichlet rows on the immersed space); the volume coupling C_kj = int_{Omega_2} chi_k phi_j
// is accumulated per immersed node over its adjacent cells (Gauss coupling_nq^3 per cell,
// elliptic_interface.cc:572).  No triplet lists: the full-size instance has 4e5 cells.
void build_immersed_box3d(const Params &P, const Grid &g, Problem &pb) {
  const int nc = P.ncomp;
  int n1[3];
  double h[3];
  Band1D M1[3], K1[3], G1[3];
  for (int a = 0; a < 3; ++a) {
    n1[a] = P.box_cells[a] + 1;
    h[a] = (P.box_hi[a] - P.box_lo[a]) / P.box_cells[a];
    M1[a] = band1d(1, 1, P.box_cells[a], h[a], 0);
    K1[a] = band1d(1, 1, P.box_cells[a], h[a], 1);
    G1[a] = band1d(1, 1, P.box_cells[a], h[a], 2);
  }
  const int64_t nl = (int64_t)n1[0] * n1[1] * n1[2];
  auto split = [&](int64_t n, int *idx) {
    idx[0] = (int)(n % n1[0]);
    idx[1] = (int)((n / n1[0]) % n1[1]);
    idx[2] = (int)(n / ((int64_t)n1[0] * n1[1]));
  };
  auto node = [&](const int *idx) { return ((int64_t)idx[2] * n1[1] + idx[1]) * n1[0] + idx[0]; };
  // ---- M (expanded to nc components), K (scalar stiffness, expanded) and A2
  Csr Mx, Kx, A2;
  for (Csr *m : {&Mx, &Kx, &A2}) {
    m->nrows = m->ncols = nl * nc;
    m->row_ptr.assign(nl * nc + 1, 0);
  }
  const bool el = P.elasticity != 0;
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nl; ++n) {
    int idx[3];
    split(n, idx);
    const int64_t cnt = (int64_t)M1[0].count[idx[0]] * M1[1].count[idx[1]] * M1[2].count[idx[2]];
    for (int a = 0; a < nc; ++a) {
      Mx.row_ptr[n * nc + a + 1] = cnt;
      Kx.row_ptr[n * nc + a + 1] = cnt;
      A2.row_ptr[n * nc + a + 1] = el ? cnt * nc : cnt;
    }
  }
  for (Csr *m : {&Mx, &Kx, &A2}) {
    for (int64_t r = 0; r < nl * nc; ++r) m->row_ptr[r + 1] += m->row_ptr[r];
    m->col.resize(m->row_ptr[nl * nc]);
    m->val.resize(m->row_ptr[nl * nc]);
  }
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nl; ++n) {
    int idx[3], j[3];
    split(n, idx);
    int64_t pm[3], pk[3], pa[3];
    for (int a = 0; a < nc; ++a) {
      pm[a] = Mx.row_ptr[n * nc + a];
      pk[a] = Kx.row_ptr[n * nc + a];
      pa[a] = A2.row_ptr[n * nc + a];
    }
    for (int k2 = 0; k2 < M1[2].count[idx[2]]; ++k2)
      for (int k1 = 0; k1 < M1[1].count[idx[1]]; ++k1)
        for (int k0 = 0; k0 < M1[0].count[idx[0]]; ++k0) {
          j[0] = M1[0].first[idx[0]] + k0;
          j[1] = M1[1].first[idx[1]] + k1;
          j[2] = M1[2].first[idx[2]] + k2;
          double m[3], k[3], gij[3], gji[3];
          for (int a = 0; a < 3; ++a) {
            m[a] = M1[a].at(idx[a], j[a]);
            k[a] = K1[a].at(idx[a], j[a]);
            gij[a] = G1[a].at(idx[a], j[a]);
            gji[a] = G1[a].at(j[a], idx[a]);
          }
          const double mass = m[0] * m[1] * m[2];
          const double lap = k[0] * m[1] * m[2] + m[0] * k[1] * m[2] + m[0] * m[1] * k[2];
          const int64_t jn = node(j);
          for (int a = 0; a < nc; ++a) {
            Mx.col[pm[a]] = (int32_t)(jn * nc + a);
            Mx.val[pm[a]++] = mass;
            Kx.col[pk[a]] = (int32_t)(jn * nc + a);
            Kx.val[pk[a]++] = lap;
            if (!el) {
              A2.col[pa[a]] = (int32_t)(jn * nc + a);
              A2.val[pa[a]++] = P.beta2 * lap;
              continue;
            }
            for (int b = 0; b < nc; ++b) {
              double tab = 1.0, tba = 1.0;
              for (int c = 0; c < 3; ++c) {
                if (a == b) {
                  tab *= (c == a) ? k[c] : m[c];
                } else {
                  tab *= (c == a) ? gij[c] : (c == b) ? gji[c] : m[c];
                  tba *= (c == b) ? gij[c] : (c == a) ? gji[c] : m[c];
                }
              }
              if (a == b) tba = tab;
              A2.col[pa[a]] = (int32_t)(jn * nc + b);
              A2.val[pa[a]++] = P.lame2_lambda * tab + P.lame2_mu * ((a == b ? lap : 0.0) + tba);
            }
          }
        }
  }
  // ---- scalar coupling rows, one immersed node at a time
  Gauss q(P.coupling_nq);
  const int p = g.p;
  std::vector<std::vector<std::pair<int64_t, double>>> rows(nl);
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t n = 0; n < nl; ++n) {
    int idx[3];
    split(n, idx);
    std::vector<std::pair<int64_t, double>> acc;
    for (int c2 = std::max(0, idx[2] - 1); c2 <= std::min(P.box_cells[2] - 1, idx[2]); ++c2)
      for (int c1 = std::max(0, idx[1] - 1); c1 <= std::min(P.box_cells[1] - 1, idx[1]); ++c1)
        for (int c0 = std::max(0, idx[0] - 1); c0 <= std::min(P.box_cells[0] - 1, idx[0]); ++c0) {
          const int cell[3] = {c0, c1, c2};
          for (size_t g0 = 0; g0 < q.x.size(); ++g0)
            for (size_t g1 = 0; g1 < q.x.size(); ++g1)
              for (size_t g2 = 0; g2 < q.x.size(); ++g2) {
                const double t[3] = {q.x[g0], q.x[g1], q.x[g2]};
                double chi = 1.0, x[3];
                for (int a = 0; a < 3; ++a) {
                  chi *= (idx[a] == cell[a]) ? 1 - t[a] : t[a];  // node sits at the low / high end of the cell
                  x[a] = P.box_lo[a] + (cell[a] + t[a]) * h[a];
                }
                const double w = chi * q.w[g0] * q.w[g1] * q.w[g2] * h[0] * h[1] * h[2];
                int bc[3];
                double v1[3][3], d1[3][3];
                for (int a = 0; a < 3; ++a) {
                  const double sx = (x[a] - g.lo) / g.h;
                  bc[a] = std::max(0, std::min(g.N - 1, (int)std::floor(sx)));
                  shape1d(p, sx - bc[a], v1[a], d1[a]);
                }
                int l[3], jb[3];
                for (l[2] = 0; l[2] <= p; ++l[2])
                  for (l[1] = 0; l[1] <= p; ++l[1])
                    for (l[0] = 0; l[0] <= p; ++l[0]) {
                      for (int a = 0; a < 3; ++a) jb[a] = bc[a] * p + l[a];
                      if (g.boundary(jb)) continue;
                      acc.emplace_back(g.node(jb), v1[0][l[0]] * v1[1][l[1]] * v1[2][l[2]] * w);
                    }
              }
        }
    std::sort(acc.begin(), acc.end(), [](const auto &u, const auto &v) { return u.first < v.first; });
    auto &row = rows[n];
    for (size_t i = 0; i < acc.size();) {
      size_t e = i;
      double sum = 0;
      while (e < acc.size() && acc[e].first == acc[i].first) sum += acc[e++].second;
      row.emplace_back(acc[i].first, sum);
      i = e;
    }
  }
  Csr C;
  C.nrows = nl * nc;
  C.ncols = g.nnodes * nc;
  C.row_ptr.assign(C.nrows + 1, 0);
  for (int64_t n = 0; n < nl; ++n)
    for (int b = 0; b < nc; ++b) C.row_ptr[n * nc + b + 1] = C.row_ptr[n * nc + b] + (int64_t)rows[n].size();
  C.col.resize(C.row_ptr[C.nrows]);
  C.val.resize(C.row_ptr[C.nrows]);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nl; ++n)
    for (int b = 0; b < nc; ++b) {
      int64_t pos = C.row_ptr[n * nc + b];
      for (const auto &e : rows[n]) {
        C.col[pos] = (int32_t)(e.first * nc + b);
        C.val[pos++] = e.second;
      }
    }
  rows.clear();
  // g = embedded_value * int chi_k, f2 = (f_2 - f) int chi_k with f_2 - f = 1 (elasticity.prm:19-23);
  // int chi_k is the row sum of the scalar mass matrix
  std::vector<double> gv(nl * nc), f2(nl * nc), xyz(nl * 3);
#pragma omp parallel for schedule(static)
  for (int64_t n = 0; n < nl; ++n) {
    int idx[3];
    split(n, idx);
    double gi = 0.0;
    for (int64_t k = Mx.row_ptr[n * nc]; k < Mx.row_ptr[n * nc + 1]; ++k) gi += Mx.val[k];
    for (int b = 0; b < nc; ++b) {
      gv[n * nc + b] = P.embedded_value[b] * gi;
      f2[n * nc + b] = gi;
    }
    for (int a = 0; a < 3; ++a) xyz[n * 3 + a] = P.box_lo[a] + idx[a] * h[a];
  }
  pb.vecs["n_lambda_global"] = {(double)(nl * nc)};
  Csr Ct = csr_transpose(C);
  if (P.u_node0 >= 0) {   // this rank's rows only (column indices stay global)
    pb.mats["Ct"] = csr_slice_rows(Ct, P.u_node0 * nc, P.u_node1 * nc);
    pb.mats["C"] = csr_slice_rows(C, P.l0, P.l1);
    pb.mats["M"] = csr_slice_rows(Mx, P.l0, P.l1);
    pb.mats["K"] = csr_slice_rows(Kx, P.l0, P.l1);
    pb.mats["A2"] = csr_slice_rows(A2, P.l0, P.l1);
    pb.vecs["g"] = std::vector<double>(gv.begin() + P.l0, gv.begin() + P.l1);
    pb.vecs["f2"] = std::vector<double>(f2.begin() + P.l0, f2.begin() + P.l1);
  } else {
    pb.mats["Ct"] = std::move(Ct);
    pb.mats["C"] = std::move(C);
    pb.mats["M"] = std::move(Mx);
    pb.mats["K"] = std::move(Kx);
    pb.mats["A2"] = std::move(A2);
    pb.vecs["g"] = std::move(gv);
    pb.vecs["f2"] = std::move(f2);
  }
  pb.vecs["immersed_xyz"] = std::move(xyz);
}

void build_rhs(const Params &P, const Grid &g, Problem &pb) {
  const int dim = g.dim, nc = P.ncomp;
  const Band1D M = band1d(g.p, g.p, g.N, g.h, 0);
  std::vector<double> one(g.n1, 0.0);  // (M 1)_i
  for (int i = 0; i < g.n1; ++i)
    for (int k = 0; k < M.count[i]; ++k) one[i] += M.a[(size_t)i * M.W + k];
  const int64_t n0 = P.u_node0 >= 0 ? P.u_node0 : 0, n1 = P.u_node0 >= 0 ? P.u_node1 : g.nnodes;
  std::vector<double> f((n1 - n0) * nc, 0.0);
#pragma omp parallel for schedule(static)
  for (int64_t n = n0; n < n1; ++n) {
    int idx[3] = {0, 0, 0};
    g.split(n, idx);
    if (g.boundary(idx)) continue;
    double w = one[idx[0]] * one[idx[1]];
    if (dim == 3) w *= one[idx[2]];
    for (int a = 0; a < nc; ++a) f[(n - n0) * nc + a] = P.body_force[a] * w;
  }
  pb.vecs["f"] = std::move(f);
}

bool generate(Problem &pb) {
  const Params &P = pb.p;
  if (P.dim != 2 && P.dim != 3) return pb.err = "dim must be 2 or 3", false;
  if (P.degree != 1 && P.degree != 2) return pb.err = "degree must be 1 or 2", false;
  if (P.ncomp != 1 && P.ncomp != P.dim) return pb.err = "ncomp must be 1 or dim", false;
  if (P.stokes && (P.degree != 2 || P.ncomp != P.dim))
    return pb.err = "stokes needs degree 2 and ncomp == dim", false;
  if (P.n_cells < 2) return pb.err = "n_cells must be >= 2", false;
  if (P.coupling_nq < 1 || P.coupling_nq > 5) return pb.err = "coupling_nq in 1..5", false;
  if (P.immersed_kind == 1 && (P.dim != 2 || P.ncomp != 1 || P.imm_cells < 1 || !(P.imm_hi > P.imm_lo)))
    return pb.err = "box-immersed mode needs dim 2, ncomp 1, imm_cells >= 1, imm_hi > imm_lo", false;
  if (P.immersed_kind == 2) {
    if (P.dim != 3) return pb.err = "3-D box-immersed mode needs dim 3", false;
    for (int a = 0; a < 3; ++a)
      if (P.box_cells[a] < 1 || !(P.box_hi[a] > P.box_lo[a])) return pb.err = "bad immersed box", false;
    int64_t nlx = P.ncomp;
    for (int a = 0; a < 3; ++a) nlx *= P.box_cells[a] + 1;
    if (nlx > 2147483647LL) return pb.err = "immersed space has more than 2^31-1 dofs", false;
  }
  if (P.elasticity && (P.ncomp != P.dim || P.dim != 3 || P.stokes || P.degree != 1))
    return pb.err = "elasticity needs dim 3, degree 1, ncomp 3, stokes off", false;

  Grid g;
  g.dim = P.dim;
  g.p = P.degree;
  g.N = P.n_cells;
  g.n1 = P.degree * P.n_cells + 1;
  g.lo = P.lo;
  g.h = (P.hi - P.lo) / P.n_cells;
  g.nnodes = 1;
  for (int a = 0; a < g.dim; ++a) g.nnodes *= g.n1;
  if (g.nnodes * P.ncomp > 2147483647LL) return pb.err = "more than 2^31-1 columns", false;
  const bool part = P.u_node0 >= 0;
  if (part) {
    if (P.u_node1 < P.u_node0 || P.u_node1 > g.nnodes || P.l0 < 0 || P.l1 < P.l0)
      return pb.err = "bad row ranges", false;
  }
  build_A(P, g, pb.mats["A"], part ? P.u_node0 : 0, part ? P.u_node1 : g.nnodes);
  if (P.stokes) {
    int64_t n_p = 0;
    Csr B, Mp;
    build_B_Mp(g, B, Mp, n_p, part ? P.p_node0 : -1, part ? P.p_node1 : -1, true);
    if (part) {
      // Bt rows of my velocity slab: transpose the B rows of every pressure
      // node that can couple to it (pressure z-planes touching the slab).
      const int64_t plane_u = (int64_t)g.n1 * g.n1, np1d = g.N + 1, plane_p = np1d * np1d;
      const int64_t zu0 = P.u_node0 / plane_u, zu1 = (P.u_node1 + plane_u - 1) / plane_u;  // [zu0, zu1)
      int64_t zp0 = std::max<int64_t>(0, zu0 / 2 - 1), zp1 = std::min<int64_t>(np1d, (zu1 + 1) / 2 + 1);
      int64_t stride_p = plane_p;
      if (g.dim == 2) {  // 2-D: the slabs are grid rows
        const int64_t yu0 = P.u_node0 / g.n1, yu1 = (P.u_node1 + g.n1 - 1) / g.n1;
        zp0 = std::max<int64_t>(0, yu0 / 2 - 1);
        zp1 = std::min<int64_t>(np1d, (yu1 + 1) / 2 + 1);
        stride_p = np1d;
      }
      Csr Bext, dummy;
      int64_t np_all = 0;
      build_B_Mp(g, Bext, dummy, np_all, zp0 * stride_p, zp1 * stride_p, false);
      Csr Bt = csr_transpose(Bext);
      for (auto &c : Bt.col) c += (int32_t)(zp0 * stride_p);
      Bt.ncols = np_all;
      pb.mats["Bt"] = csr_slice_rows(Bt, P.u_node0 * P.ncomp, P.u_node1 * P.ncomp);
      pb.vecs["rhs_p"] = std::vector<double>(P.p_node1 - P.p_node0, 0.0);
    } else {
      pb.mats["Bt"] = csr_transpose(B);
      pb.vecs["rhs_p"] = std::vector<double>(n_p, 0.0);
    }
    pb.mats["B"] = std::move(B);
    pb.mats["Mp"] = std::move(Mp);
    pb.vecs["n_p_global"] = {(double)n_p};
  }
  if (P.immersed_kind == 2)
    build_immersed_box3d(P, g, pb);
  else
    build_immersed(P, g, pb);
  build_rhs(P, g, pb);
  return true;
}

}  // namespace

// ------------------------------------------------------------------- C API
extern "C" {


void *alfd_synth_generate(const alfd_synth_params *sp, char *err, int errlen) {
  Problem *pb = new Problem;
  Params &P = pb->p;
  P.dim = sp->dim;
  P.degree = sp->degree;
  P.ncomp = sp->ncomp;
  P.n_cells = sp->n_cells;
  P.lo = sp->lo;
  P.hi = sp->hi;
  P.stokes = sp->stokes;
  P.grad_div = sp->grad_div;
  P.gamma_grad_div = sp->gamma_grad_div;
  P.beta = sp->beta;
  for (int i = 0; i < 3; ++i) {
    P.center[i] = sp->center[i];
    P.body_force[i] = sp->body_force[i];
    P.embedded_value[i] = sp->embedded_value[i];
  }
  P.radius = sp->radius;
  P.immersed_refine = sp->immersed_refine;
  P.coupling_nq = sp->coupling_nq;
  P.u_node0 = sp->u_node0;
  P.u_node1 = sp->u_node1;
  P.p_node0 = sp->p_node0;
  P.p_node1 = sp->p_node1;
  P.l0 = sp->l0;
  P.l1 = sp->l1;
  P.immersed_kind = sp->immersed_kind;
  P.imm_cells = sp->imm_cells;
  P.imm_lo = sp->imm_lo;
  P.imm_hi = sp->imm_hi;
  P.beta2 = sp->beta2;
  P.want_surface_mass = sp->want_surface_mass;
  P.assembly = sp->assembly;
  P.elasticity = sp->elasticity;
  P.lame_lambda = sp->lame_lambda;
  P.lame_mu = sp->lame_mu;
  P.lame2_lambda = sp->lame2_lambda;
  P.lame2_mu = sp->lame2_mu;
  for (int i = 0; i < 3; ++i) {
    P.box_lo[i] = sp->box_lo[i];
    P.box_hi[i] = sp->box_hi[i];
    P.box_cells[i] = sp->box_cells[i];
  }
  if (!generate(*pb)) {
    if (err && errlen > 0) std::snprintf(err, errlen, "%s", pb->err.c_str());
    delete pb;
    return nullptr;
  }
  return pb;
}

void alfd_synth_free(void *h) { delete static_cast<Problem *>(h); }

int alfd_synth_permute_nodes(void *h, const int64_t *new_to_old, int64_t n_nodes) {
  Problem *pb = static_cast<Problem *>(h);
  const Params &P = pb->p;
  if (P.u_node0 >= 0) return -1;                       // unpartitioned problems only
  auto itA = pb->mats.find("A");
  if (itA == pb->mats.end()) return -1;
  const int nc = P.ncomp;
  if (itA->second.nrows != n_nodes * nc) return -1;
  const int64_t n = n_nodes * nc;
  std::vector<int64_t> old_to_new_node(n_nodes, -1);
  for (int64_t k = 0; k < n_nodes; ++k) {
    if (new_to_old[k] < 0 || new_to_old[k] >= n_nodes || old_to_new_node[new_to_old[k]] >= 0) return -2;
    old_to_new_node[new_to_old[k]] = k;
  }
  auto new_of = [&](int64_t dof) { return old_to_new_node[dof / nc] * nc + dof % nc; };
  auto old_of = [&](int64_t dof) { return new_to_old[dof / nc] * nc + dof % nc; };
  // rows: out row r = in row old_of(r); cols mapped (and re-sorted) when map_cols
  auto permute = [&](Csr &m, bool map_rows, bool map_cols) {
    Csr o;
    o.nrows = m.nrows;
    o.ncols = m.ncols;
    o.row_ptr.assign(m.nrows + 1, 0);
    for (int64_t r = 0; r < m.nrows; ++r) {
      const int64_t src = map_rows ? old_of(r) : r;
      o.row_ptr[r + 1] = m.row_ptr[src + 1] - m.row_ptr[src];
    }
    for (int64_t r = 0; r < m.nrows; ++r) o.row_ptr[r + 1] += o.row_ptr[r];
    o.col.resize(m.nnz());
    o.val.resize(m.nnz());
#pragma omp parallel
    {
      std::vector<std::pair<int32_t, double>> row;
#pragma omp for schedule(static)
      for (int64_t r = 0; r < m.nrows; ++r) {
        const int64_t src = map_rows ? old_of(r) : r;
        const int64_t k0 = m.row_ptr[src], len = m.row_ptr[src + 1] - k0, o0 = o.row_ptr[r];
        if (!map_cols) {
          for (int64_t k = 0; k < len; ++k) o.col[o0 + k] = m.col[k0 + k], o.val[o0 + k] = m.val[k0 + k];
          continue;
        }
        row.resize(len);
        for (int64_t k = 0; k < len; ++k) row[k] = {(int32_t)new_of(m.col[k0 + k]), m.val[k0 + k]};
        std::sort(row.begin(), row.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
        for (int64_t k = 0; k < len; ++k) o.col[o0 + k] = row[k].first, o.val[o0 + k] = row[k].second;
      }
    }
    m = std::move(o);
  };
  permute(itA->second, true, true);
  for (const char *name : {"Bt", "Ct"})
    if (pb->mats.count(name)) permute(pb->mats[name], true, false);
  for (const char *name : {"B", "C"})
    if (pb->mats.count(name)) permute(pb->mats[name], false, true);
  if (pb->mats.count("G")) permute(pb->mats["G"], true, true);
  if (pb->vecs.count("f") && (int64_t)pb->vecs["f"].size() == n) {
    std::vector<double> f(n);
    for (int64_t r = 0; r < n; ++r) f[r] = pb->vecs["f"][old_of(r)];
    pb->vecs["f"] = f;
  }
  return 0;
}

// Returns 0 and fills dims/pointers if the matrix exists, else -1. Pointers
// stay valid until alfd_synth_free.
int alfd_synth_matrix(void *h, const char *name, int64_t *nrows, int64_t *ncols, int64_t *nnz,
                      const int64_t **row_ptr, const int32_t **col, const double **val) {
  Problem *pb = static_cast<Problem *>(h);
  auto it = pb->mats.find(name);
  if (it == pb->mats.end()) return -1;
  const Csr &m = it->second;
  *nrows = m.nrows;
  *ncols = m.ncols;
  *nnz = m.nnz();
  *row_ptr = m.row_ptr.data();
  *col = m.col.data();
  *val = m.val.data();
  return 0;
}

int alfd_synth_vector(void *h, const char *name, int64_t *n, const double **data) {
  Problem *pb = static_cast<Problem *>(h);
  auto it = pb->vecs.find(name);
  if (it == pb->vecs.end()) return -1;
  *n = (int64_t)it->second.size();
  *data = it->second.data();
  return 0;
}

// Generic CSR transpose into caller-provided arrays (t_row_ptr: ncols+1,
// t_col/t_val: nnz). Columns ascending in each output row.
void alfd_synth_transpose(int64_t nrows, int64_t ncols, const int64_t *row_ptr, const int32_t *col,
                          const double *val, int64_t *t_row_ptr, int32_t *t_col, double *t_val) {
  const int64_t nnz = row_ptr[nrows];
  for (int64_t r = 0; r <= ncols; ++r) t_row_ptr[r] = 0;
  for (int64_t k = 0; k < nnz; ++k) t_row_ptr[col[k] + 1]++;
  for (int64_t r = 0; r < ncols; ++r) t_row_ptr[r + 1] += t_row_ptr[r];
  std::vector<int64_t> cur(t_row_ptr, t_row_ptr + ncols);
  for (int64_t r = 0; r < nrows; ++r)
    for (int64_t k = row_ptr[r]; k < row_ptr[r + 1]; ++k) {
      const int64_t p = cur[col[k]]++;
      t_col[p] = (int32_t)r;
      t_val[p] = val[k];
    }
}

}  // extern "C"
