// Workgroup-cooperative dense FP64 helpers on LDS-resident blocks.
//
// The contact path's contractions are 12..36 wide -- too small for a library
// GEMM, and FP64 MFMA has no rate advantage over the FP64 vector pipe on CDNA4
// -- so a workgroup (one or four wavefronts) sweeps the output elements
// cooperatively: consecutive threads take consecutive ROWS of a column-major
// output, so the A operand is read conflict-free and the B operand is a
// broadcast within each column group.
#ifndef IDOCP_DEV_DENSE_HPP_
#define IDOCP_DEV_DENSE_HPP_

#include <hip/hip_runtime.h>

namespace idocp_dev {

// element (i, j) of a strided matrix view
struct MatView {
  double* p; int rs, cs;
  __device__ __forceinline__ double& operator()(int i, int j) const { return p[i * rs + j * cs]; }
};
__device__ __forceinline__ MatView colMajor(double* p, int ld) { return MatView{p, 1, ld}; }
__device__ __forceinline__ MatView transposed(MatView a) { return MatView{a.p, a.cs, a.rs}; }
__device__ __forceinline__ MatView sub(MatView a, int i0, int j0) { return MatView{a.p + i0 * a.rs + j0 * a.cs, a.rs, a.cs}; }

// C (m x n) = beta * C + alpha * A (m x k) * B (k x n); beta in {0, 1}
__device__ __forceinline__ void mm(MatView C, MatView A, MatView Bm, int m, int n, int k, double alpha, bool accumulate,
                                   int tid, int nthreads) {
  for (int e = tid; e < m * n; e += nthreads) {
    const int j = e / m, i = e - j * m;
    double acc = 0.0;
    for (int p = 0; p < k; ++p) acc += A(i, p) * Bm(p, j);
    if (accumulate) C(i, j) += alpha * acc; else C(i, j) = alpha * acc;
  }
}

// y (m) = beta * y + alpha * A (m x k) x (k)
__device__ __forceinline__ void mv(double* y, MatView A, const double* x, int m, int k, double alpha, bool accumulate, int tid,
                                   int nthreads) {
  for (int i = tid; i < m; i += nthreads) {
    double acc = 0.0;
    for (int p = 0; p < k; ++p) acc += A(i, p) * x[p];
    if (accumulate) y[i] += alpha * acc; else y[i] = alpha * acc;
  }
}

// In-place Cholesky A = L L^T of an n x n column-major block (lower triangle
// holds L on exit).  Right-looking, one barrier pair per column.  Returns via
// *ok (shared) whether every pivot was positive.
__device__ __forceinline__ void choleskyInPlace(double* A, int ld, int n, int tid, int nthreads, int* ok) {
  for (int j = 0; j < n; ++j) {
    __syncthreads();
    const double d = A[j + j * ld];
    if (tid == 0 && !(d > 0.0)) *ok = 0;
    const double ljj = sqrt(d);
    __syncthreads();
    for (int i = j + tid; i < n; i += nthreads) A[i + j * ld] = (i == j) ? ljj : A[i + j * ld] / ljj;
    __syncthreads();
    const int rem = n - j - 1;
    for (int e = tid; e < rem * rem; e += nthreads) {
      const int c = e / rem, r = e - c * rem;
      if (r >= c) A[(j + 1 + r) + (j + 1 + c) * ld] -= A[(j + 1 + r) + j * ld] * A[(j + 1 + c) + j * ld];
    }
  }
  __syncthreads();
}

// In-place inverse of an SPD n x n column-major block by Gauss-Jordan
// elimination without pivoting (stable for SPD: every pivot is a positive Schur
// complement).  n steps of fully parallel O(n^2) work and two barriers each --
// no triangular solves with only n busy threads, which is what makes it the
// better fit for a 256-thread workgroup than Cholesky + substitution.
__device__ __forceinline__ void spdInverseInPlace(double* A, int ld, int n, int tid, int nthreads, int* ok) {
  for (int k = 0; k < n; ++k) {
    __syncthreads();
    const double p = A[k + k * ld];
    if (tid == 0 && !(p > 0.0)) *ok = 0;
    const double ip = 1.0 / p;
    double upd[4];                         // n*n <= 4 * nthreads for every block on this path
    int cnt = 0;
    for (int e = tid; e < n * n; e += nthreads, ++cnt) {
      const int j = e / n, i = e - j * n;
      const double aik = A[i + k * ld], akj = A[k + j * ld], aij = A[i + j * ld];
      double v;
      if (i == k && j == k) v = ip;
      else if (i == k) v = akj * ip;
      else if (j == k) v = -aik * ip;
      else v = aij - aik * akj * ip;
      upd[cnt & 3] = v;
    }
    __syncthreads();
    cnt = 0;
    for (int e = tid; e < n * n; e += nthreads, ++cnt) { const int j = e / n, i = e - j * n; A[i + j * ld] = upd[cnt & 3]; }
  }
  __syncthreads();
}

// Solve L L^T X = Bm for nrhs columns, one thread per right-hand side (in place).
__device__ __forceinline__ void choleskySolve(const double* Lm, int ld, int n, double* X, int ldx, int nrhs, int tid, int nthreads) {
  for (int c = tid; c < nrhs; c += nthreads) {
    double* x = X + c * ldx;
    for (int i = 0; i < n; ++i) {
      double t = x[i];
      for (int p = 0; p < i; ++p) t -= Lm[i + p * ld] * x[p];
      x[i] = t / Lm[i + i * ld];
    }
    for (int i = n - 1; i >= 0; --i) {
      double t = x[i];
      for (int p = i + 1; p < n; ++p) t -= Lm[p + i * ld] * x[p];
      x[i] = t / Lm[i + i * ld];
    }
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_DENSE_HPP_
