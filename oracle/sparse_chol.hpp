// TEST INFRASTRUCTURE — CPU oracle (see vdo_oracle.h).
// Up-looking sparse Cholesky (A = L L^T) on the upper triangle of a CSC matrix.
// Restates the role of g2o::LinearSolverCSparse (dependencies/g2o/g2o/solvers/
// linear_solver_csparse.h:108-144,248-310), which calls CSparse cs_schol/cs_chol;
// CSparse itself is a system library absent from /root/reference, so the
// published algorithm (elimination tree + row-subtree reach, T. Davis, "Direct
// Methods for Sparse Linear Systems", ch. 4) is restated here.  The fill-reducing
// ordering is supplied by the caller (points before poses) instead of block-AMD;
// the solution is ordering-independent up to rounding.  Fails (returns false) on a
// non-positive pivot exactly like cs_chol, which the LM loop treats as a rejected step.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

namespace vdo_oracle {

class SparseChol {
 public:
  // Ap/Ai: CSC pattern; entries with row > col are ignored.
  void analyze(int n, const std::vector<int64_t>& Ap, const std::vector<int>& Ai) {
    n_ = n;
    parent_.assign(n, -1);
    std::vector<int> anc(n, -1);
    for (int k = 0; k < n; ++k)
      for (int64_t p = Ap[k]; p < Ap[k + 1]; ++p) {
        int i = Ai[p];
        while (i != -1 && i < k) {
          int nxt = anc[i];
          anc[i] = k;
          if (nxt == -1) parent_[i] = k;
          i = nxt;
        }
      }
    // column counts of L by walking every row subtree once
    std::vector<int64_t> cnt(n, 1);  // diagonal
    mark_.assign(n, -1);
    stack_.resize(n);
    tmp_.resize(n);
    for (int k = 0; k < n; ++k) {
      int top = reach(k, Ap, Ai);
      for (int t = top; t < n; ++t) cnt[stack_[t]]++;
    }
    Lp_.assign(n + 1, 0);
    for (int k = 0; k < n; ++k) Lp_[k + 1] = Lp_[k] + cnt[k];
    Li_.resize(Lp_[n]);
    Lx_.resize(Lp_[n]);
    x_.assign(n, 0.0);
    fill_.resize(n);
  }
  int64_t nnzL() const { return Lp_.empty() ? 0 : Lp_[n_]; }

  bool factor(const std::vector<int64_t>& Ap, const std::vector<int>& Ai, const std::vector<double>& Ax) {
    const int n = n_;
    for (int k = 0; k < n; ++k) fill_[k] = Lp_[k];
    std::fill(mark_.begin(), mark_.end(), -1);
    for (int k = 0; k < n; ++k) {
      int top = reach(k, Ap, Ai);
      x_[k] = 0;
      for (int64_t p = Ap[k]; p < Ap[k + 1]; ++p)
        if (Ai[p] <= k) x_[Ai[p]] = Ax[p];
      double d = x_[k];
      x_[k] = 0;
      for (int t = top; t < n; ++t) {
        int i = stack_[t];
        double lki = x_[i] / Lx_[Lp_[i]];
        x_[i] = 0;
        for (int64_t p = Lp_[i] + 1; p < fill_[i]; ++p) x_[Li_[p]] -= Lx_[p] * lki;
        d -= lki * lki;
        int64_t q = fill_[i]++;
        Li_[q] = k;
        Lx_[q] = lki;
      }
      if (!(d > 0)) return false;
      int64_t q = fill_[k]++;
      Li_[q] = k;
      Lx_[q] = std::sqrt(d);
    }
    return true;
  }

  // x := A^-1 x
  void solve(double* x) const {
    const int n = n_;
    for (int j = 0; j < n; ++j) {
      x[j] /= Lx_[Lp_[j]];
      for (int64_t p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) x[Li_[p]] -= Lx_[p] * x[j];
    }
    for (int j = n - 1; j >= 0; --j) {
      for (int64_t p = Lp_[j] + 1; p < Lp_[j + 1]; ++p) x[j] -= Lx_[p] * x[Li_[p]];
      x[j] /= Lx_[Lp_[j]];
    }
  }

 private:
  // nonzero pattern of row k of L in topological order, returned in stack_[top..n-1]
  int reach(int k, const std::vector<int64_t>& Ap, const std::vector<int>& Ai) {
    int top = n_;
    mark_[k] = k;
    for (int64_t p = Ap[k]; p < Ap[k + 1]; ++p) {
      int i = Ai[p];
      if (i > k) continue;
      int len = 0;
      for (; mark_[i] != k; i = parent_[i]) {
        tmp_[len++] = i;
        mark_[i] = k;
      }
      while (len > 0) stack_[--top] = tmp_[--len];
    }
    return top;
  }
  int n_ = 0;
  std::vector<int> parent_, mark_, stack_, tmp_;
  std::vector<int64_t> Lp_, fill_;
  std::vector<int> Li_;
  std::vector<double> Lx_, x_;
};

}  // namespace vdo_oracle
