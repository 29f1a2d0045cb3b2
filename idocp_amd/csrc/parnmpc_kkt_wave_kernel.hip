// K9w: the KKT inverse of a regular ParNMPC stage on ONE wavefront, register-resident, on the matrix cores (round 4).
//
// Replaces (like K9b, parnmpc_kernels.hip, which stays as the fallback IDOCP_K9_WAVE=0)
//   SplitKKTMatrixInverter::invert            (include/idocp/ocp/split_kkt_matrix_inverter.hxx:44-108)
//   SplitBackwardCorrection::coarseUpdate     (include/idocp/ocp/split_backward_correction.hxx:30-58)
// for the stages without switching rows / impulses (those: K9g, parnmpc_event_kernels.hip).
//
//   KKT = [0 F; F^T Q],  Q (48 x 48, order u q v) = Qss + aux_mat_next,  F (36 x 48) = [0 Fqq Fqv; Fvu Fvq Fvv]
//   KKT^-1 = [-S^-1, S^-1 F Q^-1; . , Q^-1 - Q^-1 F^T S^-1 F Q^-1],  S = F Q^-1 F^T
//
// The reference factorises Q and S (Eigen::LLT) and solves; so does this kernel, in the form that suits a wavefront: with Q = L L^T,
// W = L^-1 (lower triangular), S = M M^T, N = M^-1 every block of the inverse is a product of W, N, F -- and every product is taken in the
// one form the matrix core offers without moving data: a 16 x 16 tile in the ACCUMULATOR layout of v_mfma_f64_16x16x4_f64 (lane = 16 g + li
// holds rows 4 q + g, q = 0 .. 3, of column li) is, register by register, the A operand of X^T and the B operand of X, so that
// P(X, Y) = X^T Y costs four instructions per tile and nothing else.  With Y = W F^T:
//   S = P(Y, Y),  V = Y N^T = P(Y^T, N^T),  U^T = P(V, W),  TR = S^-1 F Q^-1 = P(N, U^T),  TL = -P(N, N),  BR = P(W, W) - P(U^T, U^T)
// (transposed copies of tiles go through a 2 kB LDS scratch: four writes, four reads).  The factorisations are blocked by 16: a diagonal
// block is factorised AND inverted in one pass over its 16 pivots with one matrix row and one right-hand side (a unit vector) per lane and
// the column entries travelling as DPP row broadcasts (the scheme of choleskySolveRows, dev_dense.hpp); panels, trailing updates and the
// off-diagonal blocks of W are P-products again.  S has 36 rows: its third block is 4 x 4 padded with the identity, and column 36 of
// the padding carries the vectors -- F^T gets r2 = [lu; lx] as a 37th column, so that W r2 and F Q^-1 r2 fall out of Y and S.
// The coarse direction KKT^-1 [r1; r2] is four matrix-vector products that contract over the ROWS of accumulator-layout tiles.
// One wavefront per stage, no workgroup barrier; against K9b's 256 threads with 84 barriers and ~10^4 vector instructions per thread.
#include <hip/hip_runtime.h>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

typedef mfma_d4 wtile;

template <typename D>
struct KktWaveSmem {
  using L = OcpLayout<D>;
  static constexpr int NX = D::NX, NU = D::NU, NQ_ = NU + NX;
  static_assert(NQ_ == 48 && NX == 36, "three 16 x 16 blocks for (u, q, v); S = 16 + 16 + 4");
  static constexpr int LDT = 18;                                    // leading dimension of the tile scratch (rows 16 bytes aligned, 16 lanes on 16 bank pairs)
  static constexpr int RECN = ((L::K_FX + NX + 1) / 2) * 2;         // the staged part of the kkt record
  static constexpr int REC = 0, AUXM = REC + RECN, CPAD = AUXM + NX * NX,      // CPAD: 0, -1, dt, 1 (structural entries of F read like data)
                       TS = CPAD + 4, RED = TS + 16 * LDT, DIR = RED + 4 * 48, TOTAL = DIR + NX + NQ_ + 4;
  static_assert(RECN % 2 == 0 && AUXM % 2 == 0 && TS % 2 == 0, "16-byte pieces");
};

// acc += X^T Y over STEPS k-steps of four rows (X, Y in accumulator layout)
template <int STEPS = 4>
__device__ __forceinline__ void pAcc(wtile& acc, const wtile& X, const wtile& Y) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(X[s], Y[s], acc, 0, 0, 0);
}
template <int STEPS = 4>
__device__ __forceinline__ void pSub(wtile& acc, const wtile& X, const wtile& Y) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-X[s], Y[s], acc, 0, 0, 0);
}
__device__ __forceinline__ wtile wzero() { return wtile{0.0, 0.0, 0.0, 0.0}; }

// register 4 q + g of sixteen per-lane values -> accumulator-layout register q (g = the lane's row of 16)
__device__ __forceinline__ wtile pickByGroup(const double (&v)[16], int g) {
  wtile t;
  const bool g1 = g & 1, g2 = g & 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double a = g1 ? v[4 * q + 1] : v[4 * q], b = g1 ? v[4 * q + 3] : v[4 * q + 2];
    t[q] = g2 ? b : a;
  }
  return t;
}

// One diagonal block: lane li of every row of 16 lanes holds row li of the SPD 16 x 16 block in a[]; returns W = L^-1 (A = L L^T) in
// accumulator layout: the lane's right-hand side is the unit vector e_li, forward substitution only (x = column li of L^-1).
__device__ __forceinline__ wtile cholInvPass16(double (&a)[16], int lane, bool& bad) {
  const int li = lane & 15, g = lane >> 4;
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = (k == li) ? 1.0 : 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const double p = rowBcastN(a[k], k);
    bad = bad || !(p > 0.0);
    double is, sq;
    rsqrtNewton(p, is, sq);
    const double lrk = (li == k) ? sq : a[k] * is;
    x[k] *= is;
#pragma unroll
    for (int c = k + 1; c < 16; ++c) {
      const double lck = rowBcastN(lrk, c);
      a[c] -= lrk * lck;
      x[c] -= lck * x[k];
    }
#pragma unroll
    for (int c = k + 1; c < 16; ++c) asm volatile("" : "+v"(x[c]));      // (pins the updates to their step, see choleskySolveRows)
  }
  return pickByGroup(x, g);
}

// index of the tile (i, j), i <= j, of a symmetric / upper block-triangular 3 x 3 arrangement; a lower block (i, k), i >= k, of W is kept at U3(k, i)
__host__ __device__ constexpr int U3(int i, int j) { return i == 0 ? j : (i == 1 ? 2 + j : 5); }

template <typename D>
__global__ __launch_bounds__(64, 1) void parnmpc_kkt_inverse_wave_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = KktWaveSmem<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NQ = S::NQ_, NK = L::NK, LDT = S::LDT;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = P->M;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const long unit = blockIdx.x;
  const long b = unit / (M - 1);
  const int pos = (int)(unit - b * (M - 1));
  const OcpNode* __restrict__ nd = B.nodes + pos;
  if (parnmpcShape<L>(*nd).general) return;
  const bool last = P->has_terminal && (pos == M - 2);
  const double dt = nd->dt;
  const long rec = b * P->NS + nd->slot;
  const double* __restrict__ kk = B.kkt + rec * L::KKT;
  const double* __restrict__ aux = B.aux + (b * P->NS + nd->next) * L::AUX;
  double* __restrict__ ki = B.kinv + rec * L::KINV;
#ifdef IDOCP_K9_STAMPS      // (diagnostic build: per-phase clock stamps of one wavefront in the middle of the launch)
  const bool stamp = B.prof && blockIdx.x == gridDim.x / 2 + 7 && threadIdx.x == 0;
#define KSTAMP(k) do { if (stamp) B.prof[k] = wall_clock64(); } while (0)
#else
#define KSTAMP(k) do { } while (0)
#endif
  KSTAMP(0);

  // ---- the record and aux_mat of the next stage -> LDS, 16 bytes per lane and load ----
  {
    constexpr int NR = S::RECN / 2, NA = NX * NX / 2, TR_ = (NR + 63) / 64, TA = (NA + 63) / 64;
    double2 rv[TR_], av[TA];
#pragma unroll
    for (int t = 0; t < TR_; ++t) { const int e = lane + 64 * t; rv[t] = reinterpret_cast<const double2*>(kk)[e < NR ? e : NR - 1]; }
#pragma unroll
    for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; av[t] = last ? double2{0.0, 0.0} : reinterpret_cast<const double2*>(aux)[e < NA ? e : NA - 1]; }
#pragma unroll
    for (int t = 0; t < TR_; ++t) { const int e = lane + 64 * t; if (e < NR) reinterpret_cast<double2*>(&sm[S::REC])[e] = rv[t]; }
#pragma unroll
    for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; if (e < NA) reinterpret_cast<double2*>(&sm[S::AUXM])[e] = av[t]; }
    if (lane == 0) { sm[S::CPAD] = 0.0; sm[S::CPAD + 1] = -1.0; sm[S::CPAD + 2] = dt; sm[S::CPAD + 3] = 1.0; }
  }
  waveLdsSync();
  KSTAMP(1);

  // ---- tile helpers (LDS scratch TS) ----
  auto tileToLds = [&](const wtile& t) {
#pragma unroll
    for (int q = 0; q < 4; ++q) sm[S::TS + (4 * q + g) * LDT + li] = t[q];
  };
  auto transposeTile = [&](const wtile& t) -> wtile {
    tileToLds(t);
    waveLdsSync();
    wtile r;
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] = sm[S::TS + li * LDT + 4 * q + g];
    waveLdsSync();
    return r;
  };
  bool bad = false;
  // A (six upper tiles of an SPD 48 x 48 matrix, destroyed) -> W = L^-1 (six lower tiles, W(i, k) at U3(k, i)) and G = W^T (upper, G(k, i) at U3(k, i))
  auto cholInv3 = [&](wtile (&A)[6], wtile (&W)[6], wtile (&G)[6]) {
    wtile R01, R02, R12;
    // block 0
    {
      double a[16];
      tileToLds(A[U3(0, 0)]);
      waveLdsSync();
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = sm[S::TS + li * LDT + k];
      waveLdsSync();
      W[U3(0, 0)] = cholInvPass16(a, lane, bad);
      G[U3(0, 0)] = transposeTile(W[U3(0, 0)]);
    }
    R01 = wzero(); pAcc(R01, G[U3(0, 0)], A[U3(0, 1)]);          // R_0j = W_00 Q_0j  (= L_j0^T)
    R02 = wzero(); pAcc(R02, G[U3(0, 0)], A[U3(0, 2)]);
    pSub(A[U3(1, 1)], R01, R01);
    pSub(A[U3(1, 2)], R01, R02);
    pSub(A[U3(2, 2)], R02, R02);
    // block 1
    {
      double a[16];
      tileToLds(A[U3(1, 1)]);
      waveLdsSync();
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = sm[S::TS + li * LDT + k];
      waveLdsSync();
      W[U3(1, 1)] = cholInvPass16(a, lane, bad);
      G[U3(1, 1)] = transposeTile(W[U3(1, 1)]);
    }
    R12 = wzero(); pAcc(R12, G[U3(1, 1)], A[U3(1, 2)]);
    pSub(A[U3(2, 2)], R12, R12);
    // block 2
    {
      double a[16];
      tileToLds(A[U3(2, 2)]);
      waveLdsSync();
#pragma unroll
      for (int k = 0; k < 16; ++k) a[k] = sm[S::TS + li * LDT + k];
      waveLdsSync();
      W[U3(2, 2)] = cholInvPass16(a, lane, bad);
      G[U3(2, 2)] = transposeTile(W[U3(2, 2)]);
    }
    // off-diagonal blocks of W = L^-1:  W_ik = -W_ii sum_{k <= m < i} L_im W_mk,  L_im = R_mi^T
    wtile T = wzero();
    pAcc(T, R01, W[U3(0, 0)]);
    W[U3(0, 1)] = wzero(); pSub(W[U3(0, 1)], G[U3(1, 1)], T);      // W_10
    T = wzero();
    pAcc(T, R12, W[U3(1, 1)]);
    W[U3(1, 2)] = wzero(); pSub(W[U3(1, 2)], G[U3(2, 2)], T);      // W_21
    T = wzero();
    pAcc(T, R02, W[U3(0, 0)]);
    pAcc(T, R12, W[U3(0, 1)]);
    W[U3(0, 2)] = wzero(); pSub(W[U3(0, 2)], G[U3(2, 2)], T);      // W_20
    G[U3(0, 1)] = transposeTile(W[U3(0, 1)]);
    G[U3(1, 2)] = transposeTile(W[U3(1, 2)]);
    G[U3(0, 2)] = transposeTile(W[U3(0, 2)]);
  };

  // ---- Q (order u, q, v; SplitBackwardCorrection::coarseUpdate: Qxx += aux_mat_next): the six upper tiles ----
  // entry (r, c) of Q, any order: LDS address in the staged record + the address of its aux_mat term (CPAD[0] = 0 where there is none)
  auto qEntry = [&](int r, int c) -> double {
    const int lo = r < c ? r : c, hi = r < c ? c : r;
    int ad, ax = S::CPAD;
    if (hi < NU) ad = S::REC + L::K_QUU + lo + NU * hi;
    else if (lo < NU) ad = S::REC + L::K_QXU + (hi - NU) + NX * lo;
    else {
      const int rr = lo - NU, cc = hi - NU;
      ad = S::REC + L::K_QXX + cc * (cc + 1) / 2 + rr;
      ax = S::AUXM + rr + NX * cc;
    }
    return sm[ad] + sm[ax];
  };
  wtile Wq[6], Gq[6];
  {
    wtile A[6];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) A[U3(it, jt)][q] = qEntry(16 * it + 4 * q + g, 16 * jt + li);
    KSTAMP(2);
    cholInv3(A, Wq, Gq);
  }
  KSTAMP(3);

  // ---- F^T (48 x 48: columns 0 .. 35 the rows of F = [0 Fqq Fqv; Fvu Fvq Fvv], backward Euler: Fqq = -I, Fqv = dt I outside the base
  //      blocks; column 36 = r2 = [lu; lx]; the rest zero) ----
  auto ftEntry = [&](int z, int e) -> double {      // F^T(z, e) = F(e, z)
    int ad = S::CPAD;                                // 0
    if (e < NV) {
      if (z >= NU && z < NU + NV) { const int cq = z - NU; ad = (e < 6 && cq < 6) ? S::REC + L::K_FQQ + e + 6 * cq : ((e >= 6 && e == cq) ? S::CPAD + 1 : S::CPAD); }
      else if (z >= NU + NV) { const int cv = z - NU - NV; ad = (e < 6 && cv < 6) ? S::REC + L::K_FQV + e + 6 * cv : ((e >= 6 && e == cv) ? S::CPAD + 2 : S::CPAD); }
    } else if (e < NX) {
      const int rv = e - NV;
      if (z < NU) ad = S::REC + L::K_FVU + rv + NV * z;
      else if (z < NU + NV) ad = S::REC + L::K_FVQ + rv + NV * (z - NU);
      else ad = S::REC + L::K_FVV + rv + NV * (z - NU - NV);
    } else if (e == NX) {
      ad = z < NU ? S::REC + L::K_LU + z : S::REC + L::K_LX + (z - NU);
    }
    return sm[ad];
  };
  // ---- Y = W F^T = P(G, F^T) and its transposed tiles ----
  wtile Y[3][3], Yt[3][3];
  {
    wtile Ft[3][3];
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) Ft[kt][jt][q] = ftEntry(16 * kt + 4 * q + g, 16 * jt + li);
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) {
        Y[it][jt] = wzero();
#pragma unroll
        for (int kt = 0; kt <= it; ++kt) pAcc(Y[it][jt], Gq[U3(kt, it)], Ft[kt][jt]);
      }
  }
  KSTAMP(4);
  // y2 = W r2 = column 36 of Y, as a vector over the rows (every lane of a row of 16 holds the entries of its rows 4 q + g)
  double y2v[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) y2v[it][q] = rowBcast<4>(Y[it][2][q]);
  // ---- S = P(Y, Y) (upper tiles); column 36 = F Q^-1 r2 ----
  wtile Sm[6];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = it; jt < 3; ++jt) {
      Sm[U3(it, jt)] = wzero();
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) pAcc(Sm[U3(it, jt)], Y[kt][it], Y[kt][jt]);
    }
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) Yt[jt][it] = transposeTile(Y[it][jt]);
  // t1 = r1 - F Q^-1 r2 over the rows (zero beyond row 35)
  double t1v[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * it + 4 * q + g;
      const double s36 = rowBcast<4>(Sm[U3(it, 2)][q]);
      t1v[it][q] = row < NX ? sm[S::REC + L::K_FX + (row < NX ? row : 0)] - s36 : 0.0;
    }
  // the padding of S: rows / columns 36 .. 47 = identity
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * it + 4 * q + g, col = 32 + li;
      if (col >= NX || row >= NX) Sm[U3(it, 2)][q] = (row == col) ? 1.0 : 0.0;
    }
  wtile Ns[6], Hs[6];
  KSTAMP(5);
  cholInv3(Sm, Ns, Hs);
  KSTAMP(6);

  // ---- V = Y N^T = P(Y^T, H) (48 x 36): contraction over the 36 rows of H (tile row 2: one k-step) ----
  wtile V[3][3];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      V[it][jt] = wzero();
#pragma unroll
      for (int kt = 0; kt <= jt; ++kt) {
        if (kt < 2) pAcc<4>(V[it][jt], Yt[kt][it], Hs[U3(kt, jt)]);
        else pAcc<1>(V[it][jt], Yt[kt][it], Hs[U3(kt, jt)]);
      }
    }
  // ---- U^T = P(V, W) (36 x 48): contraction over 48 ----
  wtile Ut[3][3];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      Ut[it][jt] = wzero();
#pragma unroll
      for (int kt = jt; kt < 3; ++kt) pAcc(Ut[it][jt], V[kt][it], Wq[U3(jt, kt)]);
    }
  // ---- TR = S^-1 F Q^-1 = P(N, U^T) (36 x 48): contraction over 36 ----
  wtile TRm[3][3];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
      TRm[it][jt] = wzero();
#pragma unroll
      for (int kt = it; kt < 3; ++kt) {
        if (kt < 2) pAcc<4>(TRm[it][jt], Ns[U3(it, kt)], Ut[kt][jt]);
        else pAcc<1>(TRm[it][jt], Ns[U3(it, kt)], Ut[kt][jt]);
      }
    }

  KSTAMP(7);
  // ---- stores.  A tile Z whose entry (row, col) belongs at kinv row r0 + col, kinv column c0 + row of a column block (ld NK): 16 lanes of a
  //      row of the wavefront write 128 consecutive bytes ----
  auto storeTile = [&](const wtile& Z, double* __restrict__ blk, int r0, int nrow, int c0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 4 * q + g;
      if (li < nrow && c >= 0 && c < NX) blk[r0 + li + NK * c] = Z[q];
    }
  };
  double* __restrict__ C0 = ki + L::I_C0;
  double* __restrict__ C1 = ki + L::I_C1;
  // C0 rows 36 .. 83: TR^T, i.e. entry (36 + rho, kappa) = TR(kappa, rho): tile (a, b) of TR as it is
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) storeTile(TRm[a][bb], C0, NX + 16 * bb, 16, 16 * a);
  // C1 rows 0 .. 35: TR(:, NU:), entry (rho, kappa) = TR(rho, NU + kappa): the transposed tiles
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const wtile Z = transposeTile(TRm[bb][a]);
      storeTile(Z, C1, 16 * bb, NX - 16 * bb, 16 * a - NU);
    }
  KSTAMP(8);
  // ---- the coarse direction (split_backward_correction.hxx:30-47): top = -S^-1 t1, bottom = W^T y2 + TR^T t1.  A product M^T x
  //      contracts over the rows of accumulator-layout tiles: per lane the terms of its rows, then the four rows of 16 lanes are summed
  //      through LDS ----
  auto reduceRows = [&](const double (&p)[3], double* out) {       // out[0 .. 47] = sum over the rows of lanes
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) sm[S::RED + 48 * g + 16 * jt + li] = p[jt];
    waveLdsSync();
    if (lane < 48) out[lane] = (sm[S::RED + lane] + sm[S::RED + 48 + lane]) + (sm[S::RED + 96 + lane] + sm[S::RED + 144 + lane]);
    waveLdsSync();
  };
  double* dir = &sm[S::DIR];                 // dlmd dgmm | du dq dv
  {
    // z = N t1 = H^T t1
    double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
      for (int kt = 0; kt <= jt; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) p[jt] += Hs[U3(kt, jt)][q] * t1v[kt][q];
    double* zz = &sm[S::DIR + NX];           // (scratch: the bottom of dir is written last)
    reduceRows(p, zz);
    double zv[3][4];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const int row = 16 * it + 4 * q + g; zv[it][q] = row < NX ? zz[row] : 0.0; }
    waveLdsSync();
    // top = -N^T z
    double pt[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
      for (int kt = jt; kt < 3; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) pt[jt] -= Ns[U3(jt, kt)][q] * zv[kt][q];
    double* tmp = &sm[S::DIR + NX];
    reduceRows(pt, tmp);
    if (lane < NX) dir[lane] = tmp[lane];
    waveLdsSync();
    // bottom = W^T y2 + TR^T t1
    double pb[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) {
#pragma unroll
      for (int kt = jt; kt < 3; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) pb[jt] += Wq[U3(jt, kt)][q] * y2v[kt][q];
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) pb[jt] += TRm[kt][jt][q] * t1v[kt][q];
    }
    reduceRows(pb, &sm[S::DIR + NX]);
  }

  KSTAMP(9);
  // ---- TL = -S^-1 = -P(N, N) (36 x 36; C0 rows 0 .. 35, symmetric: lower tiles are the transposed upper ones) ----
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = it; jt < 3; ++jt) {
      wtile Z = wzero();
#pragma unroll
      for (int kt = jt; kt < 3; ++kt) {
        if (kt < 2) pSub<4>(Z, Ns[U3(it, kt)], Ns[U3(jt, kt)]);
        else pSub<1>(Z, Ns[U3(it, kt)], Ns[U3(jt, kt)]);
      }
      // entry (rho, kappa) = TL(rho, kappa) = TL(kappa, rho): the tile (a, b) = (it, jt) goes to rows 16 b + col, columns 16 a + row
      storeTile(Z, C0, 16 * jt, NX - 16 * jt, 16 * it);
      if (jt > it) { const wtile Zt = transposeTile(Z); storeTile(Zt, C0, 16 * it, 16, 16 * jt); }
    }
  // ---- BR(:, NU:) = (W^T W - U U^T)(:, NU:) (48 x 36; C1 rows 36 .. 83): entry (36 + rho, kappa) = BR(NU + kappa, rho) ----
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int jt = it; jt < 3; ++jt) {
      wtile Z = wzero();
#pragma unroll
      for (int kt = jt; kt < 3; ++kt) pAcc(Z, Wq[U3(it, kt)], Wq[U3(jt, kt)]);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        if (kt < 2) pSub<4>(Z, Ut[kt][it], Ut[kt][jt]);
        else pSub<1>(Z, Ut[kt][it], Ut[kt][jt]);
      }
      storeTile(Z, C1, NX + 16 * jt, 16, 16 * it - NU);
      if (jt > it) { const wtile Zt = transposeTile(Z); storeTile(Zt, C1, NX + 16 * it, 16, 16 * jt - NU); }
    }

  KSTAMP(10);
  // ---- s_new = s - direction (split_backward_correction.hxx:49-58) ----
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ sn = B.snew + rec * L::SNEW;
  if (lane < NV) {
    sn[L::N_LMD + lane] = s[L::S_LMD + lane] - dir[lane];
    sn[L::N_GMM + lane] = s[L::S_GMM + lane] - dir[NV + lane];
    sn[L::N_V + lane] = s[L::S_V + lane] - dir[NX + NU + NV + lane];
    if (lane >= 6) sn[L::N_Q + lane + 1] = s[L::S_Q + lane + 1] - dir[NX + NU + lane];
  }
  if (lane >= 32 && lane < 32 + NU) sn[L::N_U + lane - 32] = s[L::S_U + lane - 32] - dir[NX + lane - 32];
  if (lane == 63) {
    double qn[7];
    lieIntegrateBase(s + L::S_Q, dir + NX + NU, -1.0, qn);
    for (int k = 0; k < 7; ++k) sn[L::N_Q + k] = qn[k];
  }
  KSTAMP(11);
#undef KSTAMP
  if (__builtin_amdgcn_ballot_w64(bad) != 0 && lane == 0 && B.status[b] == 0) B.status[b] = 1000 + pos;
}

template <typename D>
void OcpLaunch<D>::parnmpcInverseWave(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  const size_t smem = KktWaveSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)parnmpc_kkt_inverse_wave_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  hipLaunchKernelGGL((parnmpc_kkt_inverse_wave_kernel<D>), dim3((unsigned)(batch * (M - 1))), dim3(64), smem, st, B);
}

template void OcpLaunch<LeggedDims<4, 3>>::parnmpcInverseWave(const OcpBuffers&, long, int, hipStream_t);

}  // namespace idocp_dev
