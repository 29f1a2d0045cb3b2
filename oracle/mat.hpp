// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Minimal dense column-major FP64 matrix type for the CPU restatement of the
// idocp hot path.  It stands in for the tiny subset of Eigen the reference uses
// on this path (MatrixXd/VectorXd, blocks, noalias GEMM/GEMV, LLT; call sites
// listed in SURVEY.md section 8c).  Deliberately simple: clarity over speed.
#ifndef ORACLE_MAT_HPP_
#define ORACLE_MAT_HPP_

#include <cassert>
#include <cmath>
#include <cstring>
#include <vector>

#include "flops.hpp"      // FLOP_REGION(...); `make liboracle_flops.so`: the scalar below becomes a counting double

// Scalar of the restatement.  The default build is FP64 like the reference; `make liboracle_hp.so` builds the SAME sources with
// -DORACLE_REAL="long double" (x87 extended precision, 64-bit significand): the high-precision REFEREE the parity tests use to
// decide, where GPU and FP64 oracle disagree by more than 1e-10 on an ill-conditioned stage, which of the two is nearer the truth.
#ifdef ORACLE_COUNT_FLOPS
#define ORACLE_REAL oracle::Counted
#endif
#ifndef ORACLE_REAL
#define ORACLE_REAL double
#endif

namespace oracle {

typedef ORACLE_REAL real;

struct Mat {
  int r = 0, c = 0;
  std::vector<real> d;
  Mat() {}
  Mat(int rows, int cols) : r(rows), c(cols), d((size_t)rows * cols, 0.0) {}
  explicit Mat(int n) : r(n), c(1), d((size_t)n, 0.0) {}
  static Mat Identity(int n) { Mat m(n, n); for (int i = 0; i < n; ++i) m(i, i) = 1; return m; }
  real& operator()(int i, int j) { assert(i >= 0 && i < r && j >= 0 && j < c); return d[(size_t)j * r + i]; }
  real operator()(int i, int j) const { assert(i >= 0 && i < r && j >= 0 && j < c); return d[(size_t)j * r + i]; }
  real& operator[](int i) { assert(i >= 0 && i < (int)d.size()); return d[i]; }
  real operator[](int i) const { assert(i >= 0 && i < (int)d.size()); return d[i]; }
  int size() const { return (int)d.size(); }
  void setZero() { std::fill(d.begin(), d.end(), 0.0); }
  void resize(int rows, int cols) { r = rows; c = cols; d.assign((size_t)rows * cols, 0.0); }
  Mat block(int i0, int j0, int nr, int nc) const {
    Mat b(nr, nc);
    for (int j = 0; j < nc; ++j) for (int i = 0; i < nr; ++i) b(i, j) = (*this)(i0 + i, j0 + j);
    return b;
  }
  void setBlock(int i0, int j0, const Mat& b) {
    for (int j = 0; j < b.c; ++j) for (int i = 0; i < b.r; ++i) (*this)(i0 + i, j0 + j) = b(i, j);
  }
  void addBlock(int i0, int j0, const Mat& b, real alpha = 1.0) {
    for (int j = 0; j < b.c; ++j) for (int i = 0; i < b.r; ++i) (*this)(i0 + i, j0 + j) += alpha * b(i, j);
  }
  Mat segment(int i0, int n) const { return block(i0, 0, n, 1); }
  void setSegment(int i0, const Mat& v) { setBlock(i0, 0, v); }
  Mat t() const { Mat m(c, r); for (int j = 0; j < c; ++j) for (int i = 0; i < r; ++i) m(j, i) = (*this)(i, j); return m; }
  real squaredNorm() const { real s = 0; for (real x : d) s += x * x; return s; }
  real lpNorm1() const { real s = 0; for (real x : d) s += std::fabs(x); return s; }
  bool hasNaN() const { for (real x : d) if (std::isnan(x)) return true; return false; }
};

inline Mat operator*(const Mat& A, const Mat& B) {
  assert(A.c == B.r);
  Mat C(A.r, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int k = 0; k < A.c; ++k) {
      const real b = B(k, j);
      for (int i = 0; i < A.r; ++i) C(i, j) += A(i, k) * b;
    }
  return C;
}
inline Mat operator*(real s, const Mat& A) { Mat C = A; for (real& x : C.d) x *= s; return C; }
inline Mat operator+(const Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); Mat C = A; for (int i = 0; i < C.size(); ++i) C.d[i] += B.d[i]; return C; }
inline Mat operator-(const Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); Mat C = A; for (int i = 0; i < C.size(); ++i) C.d[i] -= B.d[i]; return C; }
inline Mat operator-(const Mat& A) { Mat C = A; for (real& x : C.d) x = -x; return C; }
inline Mat& operator+=(Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.d[i] += B.d[i]; return A; }
inline Mat& operator-=(Mat& A, const Mat& B) { assert(A.r == B.r && A.c == B.c); for (int i = 0; i < A.size(); ++i) A.d[i] -= B.d[i]; return A; }

// Cholesky A = L L^T (Eigen::LLT stand-in).  Returns false if not positive definite.
struct LLT {
  Mat L; bool ok = false;
  bool compute(const Mat& A) {
    const int n = A.r; L = Mat(n, n); ok = true;
    for (int j = 0; j < n; ++j) {
      real s = A(j, j);
      for (int k = 0; k < j; ++k) s -= L(j, k) * L(j, k);
      if (!(s > 0)) { ok = false; return false; }
      const real ljj = std::sqrt(s); L(j, j) = ljj;
      for (int i = j + 1; i < n; ++i) {
        real t = A(i, j);
        for (int k = 0; k < j; ++k) t -= L(i, k) * L(j, k);
        L(i, j) = t / ljj;
      }
    }
    return true;
  }
  Mat solve(const Mat& B) const {
    const int n = L.r; Mat X = B;
    for (int c = 0; c < X.c; ++c) {
      for (int i = 0; i < n; ++i) { real t = X(i, c); for (int k = 0; k < i; ++k) t -= L(i, k) * X(k, c); X(i, c) = t / L(i, i); }
      for (int i = n - 1; i >= 0; --i) { real t = X(i, c); for (int k = i + 1; k < n; ++k) t -= L(k, i) * X(k, c); X(i, c) = t / L(i, i); }
    }
    return X;
  }
};

}  // namespace oracle
#endif  // ORACLE_MAT_HPP_
