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

#include <type_traits>

namespace idocp_dev {

// element (i, j) of a strided matrix view
struct MatView {
  double* p; int rs, cs;
  __device__ __forceinline__ double& operator()(int i, int j) const { return p[i * rs + j * cs]; }
};
__device__ __forceinline__ MatView colMajor(double* p, int ld) { return MatView{p, 1, ld}; }
__device__ __forceinline__ MatView transposed(MatView a) { return MatView{a.p, a.cs, a.rs}; }
__device__ __forceinline__ MatView sub(MatView a, int i0, int j0) { return MatView{a.p + i0 * a.rs + j0 * a.cs, a.rs, a.cs}; }

// C (m x n) = beta * C + alpha * A (m x k) * B (k x n); beta in {0, 1}.
// The inner product is unrolled by 6 (or 3) with independent accumulators so that
// several LDS reads are in flight per lane: with a rolled loop every FMA waits for
// two dependent ds_read latencies and the kernel is latency-bound.  Every inner
// dimension on this path (12, 18, 18 + 3 n_contacts, 36) is a multiple of 3.
template <int U>
__device__ __forceinline__ double dotStrided(const double* a, int as, const double* b, int bs, int k) {
  double acc[U];
#pragma unroll
  for (int u = 0; u < U; ++u) acc[u] = 0.0;
  for (int p = 0; p < k; p += U) {
#pragma unroll
    for (int u = 0; u < U; ++u) acc[u] += a[(p + u) * as] * b[(p + u) * bs];
  }
  double r = acc[0];
#pragma unroll
  for (int u = 1; u < U; ++u) r += acc[u];
  return r;
}
__device__ __forceinline__ double dotAny(const double* a, int as, const double* b, int bs, int k) {
  if (k % 6 == 0) return dotStrided<6>(a, as, b, bs, k);
  if (k % 3 == 0) return dotStrided<3>(a, as, b, bs, k);
  return dotStrided<1>(a, as, b, bs, k);
}
__device__ __forceinline__ void mm(MatView C, MatView A, MatView Bm, int m, int n, int k, double alpha, bool accumulate,
                                   int tid, int nthreads) {
  for (int e = tid; e < m * n; e += nthreads) {
    const int j = e / m, i = e - j * m;
    const double acc = dotAny(A.p + i * A.rs, A.cs, Bm.p + j * Bm.cs, Bm.rs, k);
    if (accumulate) C(i, j) += alpha * acc; else C(i, j) = alpha * acc;
  }
}

// y (m) = beta * y + alpha * A (m x k) x (k)
__device__ __forceinline__ void mv(double* y, MatView A, const double* x, int m, int k, double alpha, bool accumulate, int tid,
                                   int nthreads) {
  for (int i = tid; i < m; i += nthreads) {
    const double acc = dotAny(A.p + i * A.rs, A.cs, x, 1, k);
    if (accumulate) y[i] += alpha * acc; else y[i] = alpha * acc;
  }
}

// Inverse of an SPD n x n column-major block by Gauss-Jordan elimination without
// pivoting (stable for SPD: every pivot is a positive Schur complement).  n steps
// of fully parallel O(n^2) work with ONE barrier each: the sweep ping-pongs
// between A and the scratch block W (same ld), and every thread keeps the (i, j)
// of its (at most two) elements in registers, so a step is three LDS reads, one
// reciprocal and one LDS write per element.  The inverse ends up in A.
// Requires n * n <= 2 * nthreads.
__device__ __forceinline__ void spdInverse(double* A, double* W, int ld, int n, int tid, int nthreads, int* ok) {
  const int e0 = tid, e1 = tid + nthreads;
  const bool has0 = e0 < n * n, has1 = e1 < n * n;
  const int j0 = has0 ? e0 / n : 0, i0 = has0 ? e0 - j0 * n : 0;
  const int j1 = has1 ? e1 / n : 0, i1 = has1 ? e1 - j1 * n : 0;
  double* src = A;
  double* dst = W;
  __syncthreads();
  for (int k = 0; k < n; ++k) {
    const double p = src[k + k * ld];
    if (tid == 0 && !(p > 0.0)) *ok = 0;
    const double ip = 1.0 / p;
    if (has0) {
      const double aik = src[i0 + k * ld], akj = src[k + j0 * ld], aij = src[i0 + j0 * ld];
      dst[i0 + j0 * ld] = (i0 == k) ? ((j0 == k) ? ip : akj * ip) : ((j0 == k) ? -aik * ip : aij - aik * akj * ip);
    }
    if (has1) {
      const double aik = src[i1 + k * ld], akj = src[k + j1 * ld], aij = src[i1 + j1 * ld];
      dst[i1 + j1 * ld] = (i1 == k) ? ((j1 == k) ? ip : akj * ip) : ((j1 == k) ? -aik * ip : aij - aik * akj * ip);
    }
    __syncthreads();
    double* t = src; src = dst; dst = t;
  }
  if (src != A) {            // odd n: the result sits in W
    if (has0) A[i0 + j0 * ld] = src[i0 + j0 * ld];
    if (has1) A[i1 + j1 * ld] = src[i1 + j1 * ld];
    __syncthreads();
  }
}

// Same elimination executed by ONE wavefront (lanes 0..63) with no workgroup
// barrier: LDS operations of a single wavefront are processed in program order,
// so a step only needs a compiler-level fence between the writes of step k and
// the reads of step k+1.  Each step costs one LDS round trip + one reciprocal
// (~300 cycles) instead of a 4-wave barrier round (~900 cycles).  n * n <= 64 * ME.
template <int ME>
__device__ __forceinline__ void spdInverseWave(double* A, double* W, int ld, int n, int lane, int* ok) {
  int ii[ME], jj[ME];
#pragma unroll
  for (int t = 0; t < ME; ++t) { const int e = lane + 64 * t; const int j = e / n; jj[t] = j; ii[t] = e - j * n; }
  double* src = A;
  double* dst = W;
  for (int k = 0; k < n; ++k) {
    const double p = src[k + k * ld];
    if (lane == 0 && !(p > 0.0)) *ok = 0;
    const double ip = 1.0 / p;
#pragma unroll
    for (int t = 0; t < ME; ++t) {
      if (lane + 64 * t < n * n) {
        const int i = ii[t], j = jj[t];
        const double aik = src[i + k * ld], akj = src[k + j * ld], aij = src[i + j * ld];
        dst[i + j * ld] = (i == k) ? ((j == k) ? ip : akj * ip) : ((j == k) ? -aik * ip : aij - aik * akj * ip);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    double* tmp = src; src = dst; dst = tmp;
  }
  if (src != A) {
#pragma unroll
    for (int t = 0; t < ME; ++t) if (lane + 64 * t < n * n) A[ii[t] + jj[t] * ld] = src[ii[t] + jj[t] * ld];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
}

// The same elimination with the matrix in REGISTERS of one wavefront: lane r keeps row r (n <= N <= 64 rows, N columns),
// the pivot row travels through v_readlane (scalar registers), nothing touches LDS between the load and the store, so a
// step is the reciprocal of the pivot plus N independent multiply-adds per lane.  Same update formula, same results as
// spdInverse.  In place on the column-major block A (ld); every lane of the wavefront must call it.
__device__ __forceinline__ double readLaneF64(double v, int lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}
// 1 / p for a positive, normal p: hardware estimate + two Newton steps (the error of the estimate, 2^-26 or so, is squared
// twice; the result is within an ulp of the IEEE quotient) -- a third of the dependent instructions of the division
// sequence, which sits on the critical path of every pivot step.
__device__ __forceinline__ double recipNewton(double p) {
  double x = __builtin_amdgcn_rcp(p);
  double e = __builtin_fma(-p, x, 1.0);
  x = __builtin_fma(x, e, x);
  e = __builtin_fma(-p, x, 1.0);
  return __builtin_fma(x, e, x);
}
template <int N>
__device__ __forceinline__ void spdInverseRows(double* A, int ld, int n, int lane, int* ok) {
  double a[N];
  const bool on = lane < n;
#pragma unroll
  for (int j = 0; j < N; ++j) a[j] = (on && j < n) ? A[lane + ld * j] : 0.0;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < n) {
      double prow[N];
#pragma unroll
      for (int j = 0; j < N; ++j) prow[j] = readLaneF64(a[j], k);
      const double p = prow[k];
      if (lane == 0 && !(p > 0.0)) *ok = 0;
      const double ip = recipNewton(p);
      const double aik = a[k];
      const bool isk = lane == k;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j == k) continue;
        const double akj = prow[j];
        a[j] = isk ? akj * ip : a[j] - aik * akj * ip;
      }
      a[k] = isk ? ip : -aik * ip;
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) if (on && j < n) A[lane + ld * j] = a[j];
}

// 1 / sqrt(p) and sqrt(p) for a positive, normal p: hardware estimate + Newton steps (same idea as recipNewton)
__device__ __forceinline__ void rsqrtNewton(double p, double& rs, double& s) {
  double x = __builtin_amdgcn_rsq(p);
  double e = __builtin_fma(-p * x, x, 1.0);
  x = __builtin_fma(0.5 * x, e, x);
  e = __builtin_fma(-p * x, x, 1.0);
  x = __builtin_fma(0.5 * x, e, x);
  double sq = p * x;
  const double r = __builtin_fma(-sq, sq, p);
  sq = __builtin_fma(r, 0.5 * x, sq);
  rs = x; s = sq;
}

// Cholesky + solve in one pass, everything in registers: EVERY lane of the wavefront keeps a right-hand side x of its own and leaves
// with x <- A^-1 x; lane l also keeps row (l mod 16) of the SPD N x N block A (N <= 16; read from LDS, never written back), i.e. each
// row of 16 lanes holds its own copy of the matrix and factorises it (redundant, but free in SIMT) so that the entries of column k
// of L reach every lane as a DPP row broadcast (row_newbcast: lane k of the reader's own row of 16) -- VGPR to VGPR.  (Round 2 first
// used v_readlane: the column entries then live in scalar registers, and in the Riccati kernel, whose scalar file is full of
// pointers and strides, every one of them was spilled to a VGPR lane and reloaded -- two thirds of the instructions of a step.)
// The entries of column k serve both the trailing update and the forward substitution; the backward substitution reads L back out
// of the row registers the same way.  No LDS traffic, no wait between the steps.
// n <= N: the block is padded with the identity (the caller keeps x[j] = 0 for j >= n).
template <int LANE>
__device__ __forceinline__ double rowBcast(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + LANE, 0xF, 0xF, true);      // row_newbcast:LANE (every lane has a valid source: "old" is never used)
  hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + LANE, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// (n is a constant after unrolling: the switch folds to one DPP move pair)
__device__ __forceinline__ double rowBcastN(double v, int n) {
  switch (n) {
    case 0: return rowBcast<0>(v);   case 1: return rowBcast<1>(v);   case 2: return rowBcast<2>(v);   case 3: return rowBcast<3>(v);
    case 4: return rowBcast<4>(v);   case 5: return rowBcast<5>(v);   case 6: return rowBcast<6>(v);   case 7: return rowBcast<7>(v);
    case 8: return rowBcast<8>(v);   case 9: return rowBcast<9>(v);   case 10: return rowBcast<10>(v); case 11: return rowBcast<11>(v);
    case 12: return rowBcast<12>(v); case 13: return rowBcast<13>(v); case 14: return rowBcast<14>(v); default: return rowBcast<15>(v);
  }
}
// Round 5: the same solve in the square-root-free form A = L D L^T (unit lower L) with the broadcast FUSED into the multiply-add.  One
// wavefront per SIMD pays four cycles for every instruction it issues, so the phase costs what it issues: the round-4 form took
// two v_mov_b32_dpp + two v_fma_f64 per (pivot, row) pair, a v_cndmask pair and a second Newton chain (sqrt next to 1 / sqrt) per pivot and
// three instructions per pair of the backward substitution -- about 700 in all for N = 12.  gfx950 has the 64-bit DPP forms
// `v_mov_b64_dpp` and `v_fmac_f64_dpp` (row_newbcast only), so `a -= bcast_c(u) * y` is ONE instruction; the pivot needs 1 / p only; the
// diagonal needs no select (lane k's own entry is the pivot) and the backward substitution no scaling: 9 + 2 (N - 1 - k) instructions per
// pivot and one per pair on the way back, about 310 for N = 12.  Same pivots, same update formulas up to the scaling by D (each product
// u_r u_c / p is rounded once more or less than (u_r / sqrt p)(u_c / sqrt p)): as backward-stable as Eigen's LLT (tests/test_hybrid_gpu.py).
// The instructions are inline assembly (the compiler does not fold a 64-bit DPP move into the multiply-add), which the hazard recogniser
// does not look into: a DPP operand must not have been written by the two instructions before.  Every DPP operand below is either an entry
// of `a` that the reciprocal chain of the pivot separates from its last update, or -- the pivot broadcast itself -- preceded by s_nop 1.
template <int LANE>
__device__ __forceinline__ void fmacNegRowBcast(double& acc, double u, double y) {       // acc -= u(lane LANE of the row of 16) * y
  asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(u), "v"(y), "n"(LANE));
}
template <int LANE, bool SPACED = false>      // SPACED: the caller has put two instructions between the last write of v and this read
__device__ __forceinline__ double rowBcastGuarded(double v) {
  double r;
  if constexpr (SPACED) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(LANE));
  else asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(LANE));
  return r;
}
__device__ __forceinline__ void fmacNegRowBcastN(double& acc, double u, double y, int n) {      // (n is a constant after unrolling)
  switch (n) {
    case 0: fmacNegRowBcast<0>(acc, u, y); break;   case 1: fmacNegRowBcast<1>(acc, u, y); break;   case 2: fmacNegRowBcast<2>(acc, u, y); break;
    case 3: fmacNegRowBcast<3>(acc, u, y); break;   case 4: fmacNegRowBcast<4>(acc, u, y); break;   case 5: fmacNegRowBcast<5>(acc, u, y); break;
    case 6: fmacNegRowBcast<6>(acc, u, y); break;   case 7: fmacNegRowBcast<7>(acc, u, y); break;   case 8: fmacNegRowBcast<8>(acc, u, y); break;
    case 9: fmacNegRowBcast<9>(acc, u, y); break;   case 10: fmacNegRowBcast<10>(acc, u, y); break; case 11: fmacNegRowBcast<11>(acc, u, y); break;
    case 12: fmacNegRowBcast<12>(acc, u, y); break; case 13: fmacNegRowBcast<13>(acc, u, y); break; case 14: fmacNegRowBcast<14>(acc, u, y); break;
    default: fmacNegRowBcast<15>(acc, u, y); break;
  }
}
__device__ __forceinline__ double rowBcastGuardedN(double v, int n) {
  switch (n) {
    case 0: return rowBcastGuarded<0>(v);   case 1: return rowBcastGuarded<1>(v);   case 2: return rowBcastGuarded<2>(v);   case 3: return rowBcastGuarded<3>(v);
    case 4: return rowBcastGuarded<4>(v);   case 5: return rowBcastGuarded<5>(v);   case 6: return rowBcastGuarded<6>(v);   case 7: return rowBcastGuarded<7>(v);
    case 8: return rowBcastGuarded<8>(v);   case 9: return rowBcastGuarded<9>(v);   case 10: return rowBcastGuarded<10>(v); case 11: return rowBcastGuarded<11>(v);
    case 12: return rowBcastGuarded<12>(v); case 13: return rowBcastGuarded<13>(v); case 14: return rowBcastGuarded<14>(v); default: return rowBcastGuarded<15>(v);
  }
}
#ifndef IDOCP_CHOLESKY_LLT
// The pivots are a dependent chain -- broadcast, reciprocal estimate, two Newton steps, scaling, the update of the next pivot: about eight
// FP64 latencies per step, more than the step's 2 (N - 1 - k) independent updates take to issue.  The loop is therefore written one pivot
// AHEAD: the update of the next pivot's own entry goes first, its broadcast and reciprocal chain start at once and the remaining updates of
// step k are spread between the links of that chain.  The last Newton step is folded into the scaling (a x1 + (a x1) e1 instead of
// a (x1 + x1 e1): one link less).
// (compile-time loop: the pivot and column numbers are template arguments of the DPP instructions)
template <int I, int E, typename F>
__device__ __forceinline__ void staticFor(F&& f) {
  if constexpr (I < E) { f(std::integral_constant<int, I>{}); staticFor<I + 1, E>(f); }
}
template <int N>
__device__ __forceinline__ void choleskySolveRows(const double* A, int ld, int lane, int* ok, double (&x)[N], int n = N) {
  static_assert(N <= 16, "one matrix row per lane of a DPP row");
  const int row = lane & 15;
  double a[N];
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; ++j) { const double v = A[(row < n ? row : n - 1) + ld * j]; a[j] = (row < n && j < n) ? v : ((j == row) ? 1.0 : 0.0); }      // (every lane reads -- from a row that exists --, then selects: no mask around the N loads)
  // reciprocal of the pivot as (x1, e1): 1 / p = x1 + x1 e1 up to the square of e1
  double p = rowBcastGuarded<0>(a[0]);
  double x1 = __builtin_amdgcn_rcp(p);
  { const double e0 = __builtin_fma(-p, x1, 1.0); x1 = __builtin_fma(x1, e0, x1); }
  double e1 = __builtin_fma(-p, x1, 1.0);
  staticFor<0, N>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    bad = bad || !(p > 0.0);             // (one store at the end: a branch per step would split the steps into separate blocks)
    const double t = a[k] * x1, tz = x[k] * x1;
    const double lr = __builtin_fma(t, e1, t);         // L(row, k) = A(row, k) / d_k
    const double z = __builtin_fma(tz, e1, tz);        // entry k of D^-1 L^-1 b
    const double u = a[k];
    a[k] = lr;
    x[k] = z;
    if constexpr (k + 1 < N) {
      fmacNegRowBcast<k + 1>(a[k + 1], u, lr);         // the next pivot's entry first ...
      fmacNegRowBcast<k + 1>(x[k + 1], u, z);
      if constexpr (k + 2 < N) fmacNegRowBcast<k + 2>(a[k + 2], u, lr);
      p = rowBcastGuarded<k + 1, (k + 2 < N)>(a[k + 1]);      // ... and its chain under the rest of this step's updates (two updates behind a[k + 1]'s: the DPP read is clear)
      asm volatile("v_rcp_f64_e32 %0, %1" : "=v"(x1) : "v"(p));
      // the updates left (column k + 2: x only) are dealt out in four parts between the links of the chain
      constexpr int c0 = k + 2, left = N - c0 > 0 ? N - c0 : 0, q4 = (left + 3) / 4;
      auto upd = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        if constexpr (c > c0) fmacNegRowBcast<c>(a[c], u, lr);
        fmacNegRowBcast<c>(x[c], u, z);
      };
      constexpr int c1 = c0 + q4 < N ? c0 + q4 : N, c2 = c0 + 2 * q4 < N ? c0 + 2 * q4 : N, c3 = c0 + 3 * q4 < N ? c0 + 3 * q4 : N;
      // (the links are volatile assembly like the updates: the compiler knows no latency of inline assembly and pulls the chain together
      // again otherwise.  s_nop 0: the reciprocal estimate is a transcendental, whose result the next instruction must not read.)
      double e0;
      staticFor<c0, c1>(upd);
      asm volatile("s_nop 0\n\tv_fma_f64 %0, -%1, %2, 1.0" : "=v"(e0) : "v"(p), "v"(x1));
      staticFor<c1, c2>(upd);
      asm volatile("v_fmac_f64_e32 %0, %0, %1" : "+v"(x1) : "v"(e0));
      staticFor<c2, c3>(upd);
      asm volatile("v_fma_f64 %0, -%1, %2, 1.0" : "=v"(e1) : "v"(p), "v"(x1));
      staticFor<c3, N>(upd);
    }
  });
  staticFor<0, N - 1>([&](auto ic) {
    constexpr int i = N - 1 - decltype(ic)::value;
    staticFor<0, i>([&](auto rc) {
      constexpr int r = i - 1 - decltype(rc)::value;
      fmacNegRowBcast<i>(x[r], a[r], x[i]);      // x_r -= L(i, r) x_i: row i of L lives in lane i; x_(i-1), the next multiplier, first
    });
  });
  if (bad && lane == 0) *ok = 0;
}
#else
template <int N>
__device__ __forceinline__ void choleskySolveRows(const double* A, int ld, int lane, int* ok, double (&x)[N], int n = N) {
  static_assert(N <= 16, "one matrix row per lane of a DPP row");
  const int row = lane & 15;
  double a[N], dinv = 1.0;      // dinv: 1 / L_kk, kept by the lanes of row k
  bool bad = false;
#pragma unroll
  for (int j = 0; j < N; ++j) { const double v = A[(row < n ? row : n - 1) + ld * j]; a[j] = (row < n && j < n) ? v : ((j == row) ? 1.0 : 0.0); }      // (every lane reads -- from a row that exists --, then selects: no mask around the N loads)
#pragma unroll
  for (int k = 0; k < N; ++k) {
    const double p = rowBcastN(a[k], k);
    bad = bad || !(p > 0.0);             // (one store at the end: a branch per step would split the steps into separate blocks)
    double is, sq;
    rsqrtNewton(p, is, sq);
    dinv = (row == k) ? is : dinv;
    const double lrk = (row == k) ? sq : a[k] * is;
    a[k] = lrk;
    x[k] *= is;
#pragma unroll
    for (int c = k + 1; c < N; ++c) {
      const double lck = rowBcastN(lrk, c);
      a[c] -= lrk * lck;
      x[c] -= lck * x[k];
    }
    // The right-hand-side updates are off the critical path (the pivots), and the compiler postpones them -- ALL of them, keeping
    // the 66 broadcast values of all steps alive (200 VGPRs instead of 74).  An empty asm that "redefines" x[c] pins each update to its step.
#pragma unroll
    for (int c = k + 1; c < N; ++c) asm volatile("" : "+v"(x[c]));
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
#pragma unroll
    for (int r = 0; r < i; ++r) asm volatile("" : "+v"(a[r]));      // likewise: the broadcasts of L for step i are not hoisted above it
    x[i] *= rowBcastN(dinv, i);
#pragma unroll
    for (int r = 0; r < i; ++r) x[r] -= rowBcastN(a[r], i) * x[i];
#pragma unroll
    for (int r = 0; r < i; ++r) asm volatile("" : "+v"(x[r]));
  }
  if (bad && lane == 0) *ok = 0;
}


#endif

// spdInverseRows for N <= 16 with the pivot row travelling as DPP row broadcasts (VGPR to VGPR, lane k of the row of 16 the matrix
// lives in) instead of v_readlane: no scalar registers, no VALU-reads-SGPR wait states on the critical path of every pivot.
template <int N>
__device__ __forceinline__ void spdInverseRowsDpp(double* A, int ld, int n, int lane, int* ok) {
  static_assert(N <= 16, "one matrix row per lane of a DPP row");
  double a[N];
  const bool on = lane < n;
  // identity padding: rows n .. N - 1 of the matrix, and -- in the lanes of the other DPP rows (16 .. 63), which only ride along -- a unit
  // row each, so that every row of 16 lanes eliminates a non-singular matrix and no lane manufactures inf / NaN (their results are
  // discarded, but NaNs in idle lanes would mask real ones under debug tooling or in a wave-wide reduction of `bad`)
  const int prow_id = lane & 15;
#pragma unroll
  for (int j = 0; j < N; ++j) a[j] = (on && j < n) ? A[lane + ld * j] : ((j == prow_id) ? 1.0 : 0.0);
  bool bad = false;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    if (k < n) {
      double prow[N];
#pragma unroll
      for (int j = 0; j < N; ++j) prow[j] = rowBcastN(a[j], k);
      const double p = prow[k];
      bad = bad || !(p > 0.0);
      const double ip = recipNewton(p);
      const double aik = a[k];
      const bool isk = lane == k;
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j == k) continue;
        const double akj = prow[j];
        a[j] = isk ? akj * ip : a[j] - aik * akj * ip;      // (same order of operations as spdInverseRows)
      }
      a[k] = isk ? ip : -aik * ip;
    }
  }
#pragma unroll
  for (int j = 0; j < N; ++j) if (on && j < n) A[lane + ld * j] = a[j];
  if (bad && lane == 0) *ok = 0;
}

// ordering point between LDS writes and reads of ONE wavefront (no workgroup barrier: its LDS operations execute in order)
__device__ __forceinline__ void waveLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
// The forward half of choleskySolveRows in the L L^T form (the callers want W = L^-1 itself: A^-1 = W^T W), with the broadcast fused into the
// multiply-add like there: pivot k costs a 64-bit DPP move, the reciprocal square root (estimate + two Newton steps), two scalings and
// 2 (NP - 1 - k) v_fmac_f64_dpp -- the round-4 form took two 32-bit DPP moves and two v_fma_f64 per (pivot, row) pair, a second Newton chain for
// sqrt(p) and a select pair per pivot (lane k's own entry is p / sqrt(p) like everybody's).  x: the lane's right-hand side (a unit vector for
// the inverse).  The DPP operand L(row, k) is written by the first instruction of the scaling block; the second scaling and the s_nop are
// the two wait states a DPP read of a fresh VALU result needs (inline assembly is invisible to the hazard recogniser).
// Like the solve above the loop runs one pivot AHEAD: the next pivot's own entry is updated first, its broadcast and reciprocal-square-root
// chain (estimate, two Newton steps: six links) start at once, and the other updates of the step are dealt out between the links.  Everything
// is volatile assembly, i.e. in program order (the compiler knows no latency of inline assembly and pulls the chain together otherwise).
// arrays of N; the pivot block is NP x NP (what lies beyond is identity padding that is never touched); n <= NP pivots are walked
// (GUARD: n is a run-time number, the steps beyond it are skipped by a uniform branch).
struct RsqrtChain {       // 1 / sqrt(p): x after link 5
  double p, x, t, h, e;
  // GUARD: the estimate is a transcendental whose result the next instruction must not read -- link 0 right behind it needs a wait state
  template <int J, bool GUARD = false> __device__ __forceinline__ void link() {
    if constexpr ((J == 0 || J == 3) && GUARD) asm volatile("s_nop 0\n\tv_mul_f64 %0, -%2, %3\n\tv_mul_f64 %1, 0.5, %3" : "=&v"(t), "=&v"(h) : "v"(p), "v"(x));
    else if constexpr (J == 0 || J == 3) asm volatile("v_mul_f64 %0, -%2, %3\n\tv_mul_f64 %1, 0.5, %3" : "=&v"(t), "=&v"(h) : "v"(p), "v"(x));
    else if constexpr (J == 1 || J == 4) asm volatile("v_fma_f64 %0, %1, %2, 1.0" : "=v"(e) : "v"(t), "v"(x));
    else if constexpr (J == 2 || J == 5) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x) : "v"(h), "v"(e));
  }
};
template <int N, int NP, bool GUARD>
__device__ __forceinline__ void cholForwardFused(double (&a)[N], double (&x)[N], int& bad, int n) {
  RsqrtChain ch;
  ch.p = rowBcastGuarded<0>(a[0]);
  asm volatile("v_rsq_f64_e32 %0, %1" : "=v"(ch.x) : "v"(ch.p));
  staticFor<0, 6>([&](auto jc) { ch.template link<decltype(jc)::value, (decltype(jc)::value == 0)>(); });
  staticFor<0, NP>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
    if (!GUARD || k < n) {
      bad |= !(ch.p > 0.0);
      asm volatile("" : "+v"(bad));      // (evaluated here: postponed to the end, the comparison keeps every pivot alive -- in scratch)
      double lr;
      // L(row, k) and y_k; the DPP reads of L(row, k) below need two wait states behind its write: the second scaling and the s_nop
      asm volatile("v_mul_f64 %0, %2, %3\n\tv_mul_f64 %1, %1, %3\n\ts_nop 0" : "=&v"(lr), "+v"(x[k]) : "v"(a[k]), "v"(ch.x));
      a[k] = lr;
      auto upd = [&](auto cc) {
        constexpr int c = decltype(cc)::value;
        fmacNegRowBcast<c>(a[c], lr, lr);        // A(row, c) -= L(c, k) L(row, k)
        fmacNegRowBcast<c>(x[c], lr, x[k]);      // b_c -= L(c, k) y_k
      };
      if constexpr (k + 1 < NP) {
        upd(std::integral_constant<int, k + 1>{});
        if constexpr (k + 2 < NP) fmacNegRowBcast<k + 2>(a[k + 2], lr, lr);
        ch.p = rowBcastGuarded<k + 1, (k + 2 < NP)>(a[k + 1]);
        asm volatile("v_rsq_f64_e32 %0, %1" : "=v"(ch.x) : "v"(ch.p));
        constexpr int c0 = k + 2, cnt = NP - c0 > 0 ? NP - c0 : 0, q = cnt >= 6 ? (cnt + 5) / 6 : 1;      // a link behind every q columns
        staticFor<c0, NP>([&](auto cc) {
          constexpr int c = decltype(cc)::value;
          if constexpr (c > c0) fmacNegRowBcast<c>(a[c], lr, lr);
          fmacNegRowBcast<c>(x[c], lr, x[k]);
          if constexpr ((c - c0 + 1) % q == 0 && (c - c0 + 1) / q <= 6) ch.template link<(c - c0 + 1) / q - 1>();
        });
        staticFor<(cnt / q < 6 ? cnt / q : 6), 6>([&](auto jc) { ch.template link<decltype(jc)::value, (decltype(jc)::value == 0)>(); });      // (link 0 here: nothing lay between it and the estimate)
      }
    }
  });
}

// Inverse of an SPD n x n block (n <= N <= 16, in place in LDS, column-major, leading dimension ld) by ONE wavefront as A^-1 = W^T W with
// W = L^-1, A = L L^T.  Lane l < n holds row l of A and the right-hand side e_l; the pivots walk the rows with DPP row broadcasts as in
// choleskySolveRows, forward substitution only (x = column l of L^-1); the columns meet in the scratch Wb (N x N doubles: Wb[k + ldw l] =
// W(k, l)) and the lanes form the n^2 products.  What this buys over the Gauss-Jordan form (spdInverseRowsDpp) is instruction count -- a pivot
// step broadcasts ONE column entry per remaining row instead of the whole pivot row: ~160 instead of ~270 instructions for n = 6, and on a
// SIMD shared by four wavefronts the chain of such a phase costs what it issues (round 4: 1.44 us for six Gauss-Jordan pivots).
template <int N>
__device__ __forceinline__ void spdInverseCholDpp(double* A, int ld, int n, int lane, int* ok, double* Wb, int ldw) {
  static_assert(N <= 16, "one matrix row per lane of a DPP row");
  const int row = lane & 15;
  double a[N], x[N];
  int bad = 0;
#pragma unroll
  for (int j = 0; j < N; ++j) { const double v = A[(row < n ? row : n - 1) + ld * j]; a[j] = (row < n && j < n) ? v : ((j == row) ? 1.0 : 0.0); x[j] = (j == row) ? 1.0 : 0.0; }
  cholForwardFused<N, N, true>(a, x, bad, n);      // (pivots beyond n: the identity padding, nothing to do)
  if (lane < n) {
#pragma unroll
    for (int k = 0; k < N; ++k) Wb[k + ldw * lane] = x[k];
  }
  waveLdsSync();
  for (int e = lane; e < n * n; e += 64) {
    const int j = e / n, i = e - j * n;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < N; ++k) acc = __builtin_fma(Wb[k + ldw * i], Wb[k + ldw * j], acc);      // (W(k, l) = 0 for k < l and beyond n: the padding is the identity)
    A[i + ld * j] = acc;
  }
  if (bad && lane == 0) *ok = 0;
}

// Barrier of a workgroup of several wavefronts for exchanges through LDS: the wavefront's LDS accesses are complete before and visible
// after, but -- unlike __syncthreads() -- the global loads and stores in flight are not waited for (no s_waitcnt vmcnt(0)).  Only where no
// data passes from one wavefront to another through global memory.
__device__ __forceinline__ void blockLdsSync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Inverse of the joint-space inertia matrix of a floating base with NL independent legs of LJ joints, by ONE wavefront, in
// place (column-major, leading dimension ld, n = NB + NL LJ; only the UPPER triangle is read, like the reference, which mirrors
// the upper triangle of dRNEA/da, robot.hxx:496-499).  The matrix is a block arrow
//     M = [ A  B ; B^T  D ],   A: NB x NB base block,  D = blockdiag(D_1 .. D_NL),  B = [B_1 .. B_NL]
// (different legs only couple through the base), so instead of n dependent pivot steps on the whole matrix:
//     D_i^-1 (NL lanes, LJ pivots each, all at once),  E = B D^-1,  S = A - E B^T,  S^-1 (NB pivots in the registers of NB lanes),
//     M^-1 = [ S^-1   -S^-1 E ; -E^T S^-1   D^-1 + E^T S^-1 E ]
// -- NB + LJ = 9 dependent pivot steps instead of 18 and a quarter of the multiply-adds; pinocchio's sparse Cholesky
// (Robot::computeMJtJinv, robot.hxx:576-615) exploits the same tree structure.  E is kept in the (redundant) lower-left block
// while it is needed.  Every lane of the wavefront must call this.
// progress / readers (LDS words, optional): other wavefronts of the workgroup may use the intermediate results -- *progress becomes 1 when
// D^-1 (diagonal leg blocks) and E (transposed, lower-left block) are in place, 2 when S^-1 stands in the base block, 3 when the top-right
// block -S^-1 E does; the function overwrites D^-1 and E only after *readers >= 1 (ldsFlagSet / ldsFlagWait below).
__device__ __forceinline__ void ldsFlagSet(int* flag, int value, int lane) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) __hip_atomic_store(flag, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void ldsFlagWait(int* flag, int value, int lane) {
  if (lane == 0) while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(2);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  __builtin_amdgcn_wave_barrier();
}
template <int NB, int NL, int LJ>
__device__ __forceinline__ void blockArrowInverse(double* M, int ld, int lane, int* ok, long long* prof = nullptr, int* progress = nullptr,
                                                  int* readers = nullptr, const double* Min = nullptr) {
  // Min: where the matrix stands if not in M itself (same leading dimension): steps (i) - (iii) read it from there, everything is written to M,
  // and Min is left untouched
  if (!Min) Min = M;
#define BA_STAMP(k) do { if (prof) prof[k] = wall_clock64(); } while (0)
  constexpr int NJ = NL * LJ;
  // (i) D_i^-1 in place (full symmetric blocks)
  if (lane < NL) {
    const int o = NB + lane * LJ;
    double d[LJ][LJ];
#pragma unroll
    for (int r = 0; r < LJ; ++r)
#pragma unroll
      for (int c = 0; c < LJ; ++c) d[r][c] = Min[(o + (r < c ? r : c)) + ld * (o + (r < c ? c : r))];
#pragma unroll
    for (int k = 0; k < LJ; ++k) {
      const double p = d[k][k];
      if (!(p > 0.0)) *ok = 0;
      const double ip = recipNewton(p);
#pragma unroll
      for (int r = 0; r < LJ; ++r)
#pragma unroll
        for (int c = 0; c < LJ; ++c) {
          if (r == k || c == k) continue;
          d[r][c] -= d[r][k] * d[k][c] * ip;
        }
#pragma unroll
      for (int c = 0; c < LJ; ++c) if (c != k) { d[k][c] *= ip; }
#pragma unroll
      for (int r = 0; r < LJ; ++r) if (r != k) { d[r][k] *= -ip; }
      d[k][k] = ip;
    }
#pragma unroll
    for (int r = 0; r < LJ; ++r)
#pragma unroll
      for (int c = 0; c < LJ; ++c) M[(o + r) + ld * (o + c)] = d[r][c];
  }
  waveLdsSync();
  BA_STAMP(0);
  // (ii) E = B D^-1 (NB x NJ), stored transposed in the lower-left block: M[NB + c, r] = E[r, c]
  for (int e = lane; e < NB * NJ; e += 64) {
    const int c = e / NB, r = e - c * NB, o = NB + (c / LJ) * LJ;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < LJ; ++m) acc += Min[r + ld * (o + m)] * M[(o + m) + ld * (NB + c)];
    M[(NB + c) + ld * r] = acc;
  }
  waveLdsSync();
  if (progress) ldsFlagSet(progress, 1, lane);
  BA_STAMP(1);
  // (iii) S = A - E B^T: one entry per lane (NB NB lanes, NJ-term dots) written over the A block, then row r
  // into the registers of lane r; S^-1 by Gauss-Jordan, the pivot row as DPP row broadcasts.  (Round 2: lane r formed its whole row
  // itself -- 2 NB NJ serialised LDS reads on NB lanes -- and fetched pivot rows with v_readlane.)
  {
    static_assert(NB * NB <= 64 && NB <= 16 && NJ % 2 == 0, "block-arrow Schur complement on one wavefront");
    double sval = 0.0;
    const int sc = lane / NB, sr = lane - sc * NB;
    if (lane < NB * NB) {
      double acc = Min[(sr < sc ? sr : sc) + ld * (sr < sc ? sc : sr)];
#pragma unroll
      for (int m = 0; m < NJ; ++m) acc -= M[(NB + m) + ld * sr] * Min[sc + ld * (NB + m)];      // (the operands are all fetched up front)
      sval = acc;
    }
    waveLdsSync();                               // every lane has read A (and B) before A is overwritten
    if (lane < NB * NB) M[sr + ld * sc] = sval;
    waveLdsSync();
    BA_STAMP(2);
    if constexpr (NB <= 8 && NB <= NJ) spdInverseCholDpp<NB>(M, ld, NB, lane, ok, M + ld * NB, ld);      // (scratch: the B block, dead from here)
    else spdInverseRowsDpp<NB>(M, ld, NB, lane, ok);
  }
  waveLdsSync();
  if (progress) ldsFlagSet(progress, 2, lane);
  BA_STAMP(3);
  // (iv) top-right = -S^-1 E (B is dead)
  for (int e = lane; e < NB * NJ; e += 64) {
    const int c = e / NB, r = e - c * NB;
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < NB; ++m) acc += M[r + ld * m] * M[(NB + c) + ld * m];
    M[r + ld * (NB + c)] = -acc;
  }
  waveLdsSync();
  if (progress) ldsFlagSet(progress, 3, lane);
  if (readers) ldsFlagWait(readers, 1, lane);
  BA_STAMP(4);
  // (v) bottom-right = D^-1 - E^T (top-right)
  for (int e = lane; e < NJ * NJ; e += 64) {
    const int c = e / NJ, i = e - c * NJ;
    double acc = (i / LJ == c / LJ) ? M[(NB + i) + ld * (NB + c)] : 0.0;
#pragma unroll
    for (int m = 0; m < NB; ++m) acc -= M[(NB + i) + ld * m] * M[m + ld * (NB + c)];
    M[(NB + i) + ld * (NB + c)] = acc;
  }
  waveLdsSync();
  // (vi) lower-left = (top-right)^T
  for (int e = lane; e < NB * NJ; e += 64) {
    const int c = e / NB, r = e - c * NB;
    M[(NB + c) + ld * r] = M[r + ld * (NB + c)];
  }
  waveLdsSync();
  BA_STAMP(5);
#undef BA_STAMP
}

// The same register-resident elimination for G independent n = N matrices at once: lanes [g N, (g + 1) N) of the wavefront
// hold the rows of matrix g (column-major, leading dimension N, `gstride` doubles apart), the pivot rows travel through
// ds_bpermute (__shfl) instead of v_readlane, so one pass of N dependent pivot steps serves all G matrices.  G N <= 64;
// every lane of the wavefront must call it; ok[g] is cleared when matrix g is not positive definite.
template <int N, int G>
__device__ __forceinline__ void spdInverseRowsGrouped(double* A, int gstride, int lane, int* ok) {
  static_assert(N * G <= 64, "the groups must fit in one wavefront");
  const int g0 = lane / N;
  const bool on = g0 < G;
  const int g = on ? g0 : 0, r = on ? lane - g0 * N : 0, src0 = g * N;
  double* Ag = A + g * gstride;
  double a[N];
#pragma unroll
  for (int j = 0; j < N; ++j) a[j] = Ag[r + N * j];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double prow[N];
#pragma unroll
    for (int j = 0; j < N; ++j) prow[j] = __shfl(a[j], src0 + k);
    const double p = prow[k];
    if (on && r == k && !(p > 0.0)) ok[g] = 0;
    const double ip = recipNewton(p);
    const bool isk = r == k;
    // a[j] = isk ? akj ip : a[j] - (aik akj) ip  without a select per entry: the row factor x and what is kept of the old entry are chosen once per step, and
    // fma(-(x akj), ip, keep a[j]) is, lane by lane, the value of either branch with the SAME roundings (x akj and keep a[j] are exact on the pivot lane): two
    // v_cndmask_b32 and a multiply fewer per entry on a chain that is issue-bound with one wavefront per SIMD, bit for bit the results of spdInverseRows
    const double aik = a[k];
    const double x = isk ? -1.0 : aik, keep = isk ? 0.0 : 1.0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      if (j == k) continue;
      a[j] = __builtin_fma(-(x * prow[j]), ip, keep * a[j]);
    }
    a[k] = isk ? ip : -aik * ip;
  }
#pragma unroll
  for (int j = 0; j < N; ++j) if (on) Ag[r + N * j] = a[j];
}

// ---- 16 x 16 output tiles on the matrix cores ----
// Tiles of C = X^T Y (X: k x m, Y: k x n, both contiguous along the contraction index) with v_mfma_f64_16x16x4_f64.  FP64 MFMA has
// the rate of the FP64 vector pipe on CDNA4; what it buys here is operand traffic: a lane reads ONE double of each operand per 16
// multiply-adds it contributes to (2 x 2 register tiles on the vector pipe read one per multiply-add), and the stage kernels'
// products were bound by LDS bandwidth, not by FP64 issue.
// Lane map of the instruction (lane = 16 g + li): A[i = li][k = g], B[k = g][j = li], D[i = g + 4 reg][j = li], reg = 0..3.
// Here A is taken from Y (i = column of C) and B from X (j = row of C), so that the 16 lanes li run along a COLUMN of C, i.e.
// along contiguous addresses of a column-major result.  Every lane reads two consecutive k's (16 B) per step and feeds them to
// two instructions (the sum over k does not care which k's share an instruction).  xr / yc = number of valid rows / columns of
// the tile (lanes beyond re-read the last valid one; their results are never stored), k is masked against KMAX's padding.
typedef double mfma_d4 __attribute__((ext_vector_type(4)));
// Two tiles at once: the operand reads of both are issued before the first multiply, and the two accumulation chains alternate on
// the matrix core (a chain of dependent 16-pass instructions alone leaves it idle between them).
template <int KMAX>
__device__ __forceinline__ void mfmaTilePairTN(const double* X0, int xr0, const double* Y0, int yc0, const double* X1, int xr1, const double* Y1,
                                               int yc1, int ldx, int ldy, int k, int lane, mfma_d4& acc0, mfma_d4& acc1) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int KS = (KMAX + 7) / 8;
  const int li = lane & 15, g = lane >> 4;
  const double* xp0 = X0 + ldx * (li < xr0 ? li : xr0 - 1) + 2 * g;
  const double* yp0 = Y0 + ldy * (li < yc0 ? li : yc0 - 1) + 2 * g;
  const double* xp1 = X1 + ldx * (li < xr1 ? li : xr1 - 1) + 2 * g;
  const double* yp1 = Y1 + ldy * (li < yc1 ? li : yc1 - 1) + 2 * g;
  d2 xa0[KS], ya0[KS], xa1[KS], ya1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    xa0[s] = *reinterpret_cast<const d2*>(xp0 + 8 * s); ya0[s] = *reinterpret_cast<const d2*>(yp0 + 8 * s);
    xa1[s] = *reinterpret_cast<const d2*>(xp1 + 8 * s); ya1[s] = *reinterpret_cast<const d2*>(yp1 + 8 * s);
  }
  acc0 = mfma_d4{0.0, 0.0, 0.0, 0.0};
  acc1 = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 8 * s + 2 * g;
    const bool v0 = kk < k, v1 = kk + 1 < k;
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0 ? ya0[s].x : 0.0, v0 ? xa0[s].x : 0.0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v0 ? ya1[s].x : 0.0, v0 ? xa1[s].x : 0.0, acc1, 0, 0, 0);
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1 ? ya0[s].y : 0.0, v1 ? xa0[s].y : 0.0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(v1 ? ya1[s].y : 0.0, v1 ? xa1[s].y : 0.0, acc1, 0, 0, 0);
  }
}

// The same two tiles on v_mfma_f64_4x4x4_4b_f64 (round 5).  That form of the instruction reads its operands in the lane map above but only
// multiplies the four diagonal 4 x 4 blocks -- D_b[i][j] = sum_k A[4 b + i][k] B[k][4 b + j] lands at lane 16 i + 4 b + j -- at 8 ns per
// instruction instead of 59 (256 against 1024 multiply-adds: 1.85 x the rate, DESIGN_HISTORY 4.0a).  With B = sixteen rows of the tile as before
// (X, lane li) and A = FOUR columns c0 + 4 q .. + 3 of it copied into all four blocks (Y column 4 q + (li & 3): an LDS broadcast read), one
// instruction yields C[r0 + li][c0 + 4 q + g] at lane 16 g + li: register q of the accumulator layout of the tile.  A tile is thus up to four
// independent one-register accumulations, and only the q's whose columns EXIST are computed: the stage blocks are 36, 18 and 12 wide, so of
// the 14 tiles of phase H five are a quarter full or less (84 instructions of 59 ns become 216 of 8 ns).  Same operands per k-step as above.
template <int KMAX>
__device__ __forceinline__ void mfmaTilePairTN4(const double* X0, int xr0, const double* Y0, int yc0, const double* X1, int xr1, const double* Y1,
                                                int yc1, int ldx, int ldy, int k, int lane, mfma_d4& acc0, mfma_d4& acc1) {
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int KS = (KMAX + 7) / 8;
#ifdef IDOCP_K5_TILES_HYBRID
  if (yc0 >= 16 && yc1 >= 16) { mfmaTilePairTN<KMAX>(X0, xr0, Y0, yc0, X1, xr1, Y1, yc1, ldx, ldy, k, lane, acc0, acc1); return; }
#endif
  const int li = lane & 15, g = lane >> 4, l4 = lane & 3;
  const double* xp0 = X0 + ldx * (li < xr0 ? li : xr0 - 1) + 2 * g;
  const double* xp1 = X1 + ldx * (li < xr1 ? li : xr1 - 1) + 2 * g;
  d2 xa0[KS], xa1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) { xa0[s] = *reinterpret_cast<const d2*>(xp0 + 8 * s); xa1[s] = *reinterpret_cast<const d2*>(xp1 + 8 * s); }
  acc0 = mfma_d4{0.0, 0.0, 0.0, 0.0};
  acc1 = mfma_d4{0.0, 0.0, 0.0, 0.0};
  bool v0[KS], v1[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const int kk = 8 * s + 2 * g;
    v0[s] = kk < k; v1[s] = kk + 1 < k;
    if (!v0[s]) xa0[s].x = 0.0;
    if (!v1[s]) xa0[s].y = 0.0;
    if (!v0[s]) xa1[s].x = 0.0;
    if (!v1[s]) xa1[s].y = 0.0;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const bool on0 = 4 * q < yc0, on1 = 4 * q < yc1;      // (uniform: the tile has columns there)
    if (!on0 && !on1) continue;
    const int cq0 = 4 * q + l4 < yc0 ? 4 * q + l4 : (yc0 > 0 ? yc0 - 1 : 0), cq1 = 4 * q + l4 < yc1 ? 4 * q + l4 : (yc1 > 0 ? yc1 - 1 : 0);
    const double* yp0 = Y0 + ldy * cq0 + 2 * g;
    const double* yp1 = Y1 + ldy * cq1 + 2 * g;
    d2 ya0[KS], ya1[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (on0) ya0[s] = *reinterpret_cast<const d2*>(yp0 + 8 * s);
      if (on1) ya1[s] = *reinterpret_cast<const d2*>(yp1 + 8 * s);
    }
    double a0 = 0.0, a1 = 0.0;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (on0) a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(v0[s] ? ya0[s].x : 0.0, xa0[s].x, a0, 0, 0, 0);
      if (on1) a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v0[s] ? ya1[s].x : 0.0, xa1[s].x, a1, 0, 0, 0);
      if (on0) a0 = __builtin_amdgcn_mfma_f64_4x4x4f64(v1[s] ? ya0[s].y : 0.0, xa0[s].y, a0, 0, 0, 0);
      if (on1) a1 = __builtin_amdgcn_mfma_f64_4x4x4f64(v1[s] ? ya1[s].y : 0.0, xa1[s].y, a1, 0, 0, 0);
    }
    acc0[q] = a0; acc1[q] = a1;
  }
}

// store(r, c, value) for the elements of a tile at (r0, c0) that lie inside m x n
template <typename Store>
__device__ __forceinline__ void mfmaTileStore(const mfma_d4& acc, int r0, int c0, int m, int n, int lane, Store store) {
  const int r = r0 + (lane & 15), g = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int c = c0 + g + 4 * q;
    if (r < m && c < n) store(r, c, acc[q]);
  }
}
// ---- 3 x 3 register tiles ----
// K9b / K9g keep every matrix it factorises or multiplies as 3 x 3 tiles in registers, thread (bi, bj) owning rows 3 bi .. 3 bi + 2 and
// columns 3 bj .. 3 bj + 2: an inner-product step is 6 LDS reads for 9 multiply-adds, a Gauss-Jordan pivot step shares only the
// pivot row and column through LDS.

// acc += A(3 bi + r, m) B(m, 3 bj + c), m = 0 .. K - 1 in ascending order; a(r, m), b(m, c) fetch the operands
template <int K, typename FA, typename FB>
__device__ __forceinline__ void tileMM(double (&acc)[3][3], FA a, FB b) {
#pragma unroll 4
  for (int m = 0; m < K; ++m) {
    double av[3], bv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) av[r] = a(r, m);
#pragma unroll
    for (int c = 0; c < 3; ++c) bv[c] = b(m, c);
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) acc[r][c] += av[r] * bv[c];
  }
}

// Inverse of an SPD N x N matrix held as 3 x 3 tiles by the threads with active = true (N / 3 x N / 3 of them), by Gauss-Jordan
// without pivoting.  Per pivot k the owners of column k and of row k publish them in pv (two ping-pong buffers of 2 N + 1
// doubles, the last one the reciprocal of the pivot), one barrier, and every thread updates its tile.  Every thread of the
// workgroup must call this (barriers).
template <int N>
__device__ __forceinline__ void gaussJordanTiles(double (&a)[3][3], const bool active, const int bi, const int bj, double* pv, int* ok) {
  constexpr int NB = N / 3, PV = 2 * N + 2;
  static_assert(N % 3 == 0, "tile size");
  for (int kb = 0; kb < NB; ++kb) {
#pragma unroll
    for (int kr = 0; kr < 3; ++kr) {
      const int k = 3 * kb + kr;
      double* col = pv + ((k & 1) ? PV : 0);
      double* row = col + N;
      if (active && bj == kb) {
#pragma unroll
        for (int r = 0; r < 3; ++r) col[3 * bi + r] = a[r][kr];
        if (bi == kb) {
          const double p = a[kr][kr];
          if (!(p > 0.0)) *ok = 0;
          row[N] = 1.0 / p;
        }
      }
      if (active && bi == kb) {
#pragma unroll
        for (int c = 0; c < 3; ++c) row[3 * bj + c] = a[kr][c];
      }
      __syncthreads();
      if (active) {
        const double ip = row[N];
        double ci[3], rj[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) ci[r] = col[3 * bi + r];
#pragma unroll
        for (int c = 0; c < 3; ++c) rj[c] = row[3 * bj + c];
        const bool rowk = bi == kb, colk = bj == kb;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const bool ik = (r == kr) && rowk, jk = (c == kr) && colk;
            a[r][c] = ik ? (jk ? ip : rj[c] * ip) : (jk ? -ci[r] * ip : a[r][c] - ci[r] * rj[c] * ip);
          }
      }
    }
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_DENSE_HPP_
