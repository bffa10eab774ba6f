// dealii_export.hpp -- header-only exporter of the operators of one solve() into the
// ".alfd" wire format read by fictitious_domain_al_preconditioners_amd/opfile.py
// (SURVEY.md 8(f) rank 2).  Drop next to the reference's solve(), e.g. in
// stokes_immersed_boundary.cc after line 1018:
//
//     alfd::dealii_export::Writer w("stokes.alfd");
//     w.matrix(ALFD_A,  stokes_matrix.block(0, 0));
//     w.matrix(ALFD_BT, stokes_matrix.block(0, 1));
//     w.matrix(ALFD_B,  stokes_matrix.block(1, 0));
//     w.matrix(ALFD_CT, coupling_matrix);
//     w.matrix(ALFD_MP, preconditioner_matrix.block(1, 1));
//     w.diag(ALFD_INVW, inverse_squares);
//     w.diag(ALFD_MP_LUMPED_INV, pressure_diagonal_inv);
//     w.rhs(system_rhs_block);            // after the augmentation, or before + let the GPU augment
//     w.config(cfg);
//     w.close();
//
// It replaces the reference's ad-hoc text dump (utilities.h:84-109, matrices up to
// 1000 rows only).  Needs from the matrix type: m(), n(), n_nonzero_elements(),
// begin(r)/end(r) with column()/value(); from vectors: size(), begin().
#ifndef ALFD_DEALII_EXPORT_HPP
#define ALFD_DEALII_EXPORT_HPP

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "alfd/alfd.h"

namespace alfd {
namespace dealii_export {

class Writer {
 public:
  explicit Writer(const std::string &path) : f_(std::fopen(path.c_str(), "wb")) {
    if (!f_) throw std::runtime_error("cannot open " + path);
    std::fwrite("ALFDOPS1", 1, 8, f_);
    put(0);  // record count, patched in close()
  }
  ~Writer() {
    if (f_) close();
  }
  template <class SparseMatrixType>
  void matrix(int slot, const SparseMatrixType &M) {
    const int64_t nrows = (int64_t)M.m();
    std::vector<int64_t> rp(nrows + 1, 0);
    std::vector<int32_t> col;
    std::vector<double> val;
    std::vector<std::pair<int32_t, double>> row;
    for (int64_t r = 0; r < nrows; ++r) {
      row.clear();
      for (auto it = M.begin(r); it != M.end(r); ++it) row.emplace_back((int32_t)it->column(), it->value());
      std::sort(row.begin(), row.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
      for (const auto &e : row) col.push_back(e.first), val.push_back(e.second);
      rp[r + 1] = (int64_t)col.size();
    }
    put(1), put(slot), put(nrows), put((int64_t)M.n()), put((int64_t)col.size());
    std::fwrite(rp.data(), 8, rp.size(), f_);
    std::fwrite(col.data(), 4, col.size(), f_);
    if (col.size() % 2) {
      const int32_t z = 0;
      std::fwrite(&z, 4, 1, f_);
    }
    std::fwrite(val.data(), 8, val.size(), f_);
    ++n_;
  }
  template <class VectorType>
  void diag(int slot, const VectorType &d) {
    put(2), put(slot), put((int64_t)d.size());
    std::fwrite(&*d.begin(), 8, d.size(), f_);
    ++n_;
  }
  template <class BlockVectorType>
  void rhs(const BlockVectorType &b) { blocks(100, b); }
  template <class BlockVectorType>
  void initial_guess(const BlockVectorType &x) { blocks(200, x); }
  void config(const alfd_config &c) {
    put(4), put((int64_t)sizeof(c));
    std::fwrite(&c, 1, sizeof(c), f_);
    const char z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::fwrite(z, 1, (8 - sizeof(c) % 8) % 8, f_);
    ++n_;
  }
  void close() {
    std::fseek(f_, 8, SEEK_SET);
    put(n_);
    std::fclose(f_);
    f_ = nullptr;
  }

 private:
  template <class BlockVectorType>
  void blocks(int base, const BlockVectorType &v) {
    for (unsigned int b = 0; b < v.n_blocks(); ++b) {
      put(3), put(base + (int)b), put((int64_t)v.block(b).size());
      std::fwrite(&*v.block(b).begin(), 8, v.block(b).size(), f_);
      ++n_;
    }
  }
  void put(int64_t v) { std::fwrite(&v, 8, 1, f_); }
  std::FILE *f_;
  int64_t n_ = 0;
};

}  // namespace dealii_export
}  // namespace alfd
#endif
