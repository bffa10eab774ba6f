// Minimal stand-ins for the deal.II classes the adapter touches
// (dealii::Vector, BlockVector, SparseMatrix with DIAGONAL-FIRST row storage).
// Test scaffolding only: deal.II is not installed in this environment.
#pragma once
#include <cstddef>
#include <vector>

namespace mock {

class Vector {
 public:
  Vector() = default;
  explicit Vector(std::size_t n) : v_(n, 0.0) {}
  void reinit(std::size_t n) { v_.assign(n, 0.0); }
  std::size_t size() const { return v_.size(); }
  double *begin() { return v_.data(); }
  const double *begin() const { return v_.data(); }
  double &operator[](std::size_t i) { return v_[i]; }
  double operator[](std::size_t i) const { return v_[i]; }

 private:
  std::vector<double> v_;
};

class BlockVector {
 public:
  explicit BlockVector(const std::vector<std::size_t> &sizes) {
    for (auto s : sizes) b_.emplace_back(s);
  }
  unsigned int n_blocks() const { return (unsigned int)b_.size(); }
  Vector &block(unsigned int i) { return b_[i]; }
  const Vector &block(unsigned int i) const { return b_[i]; }

 private:
  std::vector<Vector> b_;
};

// CSR with the diagonal entry stored first in each row of a square matrix,
// the other columns ascending -- deal.II's SparsityPattern convention.
class SparseMatrix {
 public:
  struct Entry {
    unsigned int c;
    double v;
    unsigned int column() const { return c; }
    double value() const { return v; }
  };
  struct It {
    const Entry *p;
    const Entry *operator->() const { return p; }
    It &operator++() {
      ++p;
      return *this;
    }
    bool operator!=(const It &o) const { return p != o.p; }
  };
  SparseMatrix(std::size_t m, std::size_t n, const long *rp, const int *col, const double *val) : m_(m), n_(n) {
    rp_.assign(rp, rp + m + 1);
    e_.reserve(rp[m]);
    for (std::size_t r = 0; r < m; ++r) {
      const std::size_t start = e_.size();
      for (long k = rp[r]; k < rp[r + 1]; ++k) e_.push_back({(unsigned int)col[k], val[k]});
      if (m == n)
        for (std::size_t k = start; k < e_.size(); ++k)
          if (e_[k].c == r) {   // rotate the diagonal to the front
            Entry d = e_[k];
            for (std::size_t q = k; q > start; --q) e_[q] = e_[q - 1];
            e_[start] = d;
            break;
          }
    }
  }
  std::size_t m() const { return m_; }
  std::size_t n() const { return n_; }
  std::size_t n_nonzero_elements() const { return e_.size(); }
  It begin(std::size_t r) const { return It{e_.data() + rp_[r]}; }
  It end(std::size_t r) const { return It{e_.data() + rp_[r + 1]}; }
  double diag_element(std::size_t r) const { return e_[rp_[r]].v; }

 private:
  std::size_t m_, n_;
  std::vector<long> rp_;
  std::vector<Entry> e_;
};

}  // namespace mock
