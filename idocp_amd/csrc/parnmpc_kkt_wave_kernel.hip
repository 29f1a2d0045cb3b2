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
// P(X, Y) = X^T Y costs four instructions per tile and nothing else.  With Y = W F^T = P(W^T, F^T):
//   S = P(Y, Y),  Q^-1 = P(W, W),  FQ = F Q^-1 = P(Y, W),  Z1 = N FQ = P(N^T, FQ),
//   TR = S^-1 F Q^-1 = P(N, Z1),  TR^T = P(Z1, N),  TL = -S^-1 = -P(N, N),  BR = Q^-1 - P(Z1, Z1)
// (transposed copies of tiles -- W^T, N^T -- go through a 2 kB LDS scratch: four writes, four reads).  The factorisations are blocked by 16: a diagonal
// block is factorised AND inverted in one pass over its 16 pivots with one matrix row and one right-hand side (a unit vector) per lane and
// the column entries travelling as DPP row broadcasts (the scheme of choleskySolveRows, dev_dense.hpp); panels, trailing updates and the
// off-diagonal blocks of W are P-products again.  S has 36 rows: its third block is 4 x 4 padded with the identity, and column 36 of
// the padding carries the vectors -- F^T gets r2 = [lu; lx] as a 37th column, so that W r2 and F Q^-1 r2 fall out of Y and S.
// The coarse direction KKT^-1 [r1; r2] is four matrix-vector products that contract over the ROWS of accumulator-layout tiles.
// One wavefront per stage, no workgroup barrier, TWO wavefronts per SIMD (<= 256 registers, 19.5 kB LDS: what does not fit waits in LDS --
// FQ during the factorisation of S -- or in the stage's dead lin record -- Q^-1 until BR is formed); every global access is a 16-byte piece
// of a 1 kB run (loads requested at once at the top, output staged in LDS by blocks of 16 columns).  Against K9b's 256 threads with 84
// barriers and ~10^4 vector instructions per thread: 4.45 -> 2.2 ms (with the list of what the compiler had to be talked out of).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

typedef mfma_d4 wtile;
typedef double v2d __attribute__((ext_vector_type(2)));      // (a native vector, not the double2 struct: arrays of it stay in registers)

template <typename D>
struct KktWaveSmem {
  using L = OcpLayout<D>;
  static constexpr int NX = D::NX, NU = D::NU, NQ_ = NU + NX;
  static_assert(NQ_ == 48 && NX == 36, "three 16 x 16 blocks for (u, q, v); S = 16 + 16 + 4");
  static constexpr int LDT = 18;                                    // leading dimension of the tile scratch (rows 16 bytes aligned, 16 lanes on 16 bank pairs)
  static constexpr int QPART = L::K_QUU + NU * NU;                  // Qxx (packed) | Qxu | Quu: the head of the kkt record
  static constexpr int FPART = L::K_FX + NX - QPART;                // Fqq Fqv Fvq Fvv Fvu | lx lu | Fx: its tail
  // BUF, by lifetime: aux_mat (NX x NX) -> the Q part of the record -> its F part -> FQ = F Q^-1 parked during the factorisation of S (six
  // full tiles, 4 doubles per lane, and the three tiles of its rows 32 .. 35, one double per lane) -> the staging area of the output
  // (84 rows x 16 columns of a column block)
  static constexpr int NBUF = 6 * 256 + 3 * 64;
  static_assert(L::K_QXX == 0 && L::K_QXU < L::K_QUU && L::K_FQQ == QPART && QPART % 2 == 0 && FPART % 2 == 0, "record order");
  static_assert(NX * NX <= NBUF && QPART <= NBUF && FPART <= NBUF && L::NK * 16 <= NBUF, "everything that passes through BUF");
  static constexpr int BUF = 0, CPAD = BUF + NBUF,      // CPAD: 0, -1, dt, 1 (structural entries of F^T read like data)
                       TS = CPAD + 4, RED = TS + 16 * LDT, Y2 = RED + 4 * 48, T1 = Y2 + 48, ZZ = T1 + 48, R1 = ZZ + 48, DIR = R1 + NX,
                       DUMP = DIR + NX + NQ_ + 4,      // where the entries of an output tile that lie outside its block are written (no branch around the store)
                       TOTAL = DUMP + 2;
  static_assert(TS % 2 == 0, "16-byte pieces");
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
// The same products where X has FOUR columns (a column tile 32 .. 35 of a 36-wide matrix): the result has four rows, register 0 of the
// accumulator tile.  v_mfma_f64_4x4x4_4b_f64 reads its operands in the lane map of the 16 x 16 x 4 form but only multiplies the four
// diagonal 4 x 4 blocks, D[4 b + i][4 b + j] at lane 16 i + 4 b + j -- register 0's place for rows 0 .. 3 -- in 8 ns instead of 46 - 59
// (DESIGN 4.0a).  X's first four columns are copied into all four blocks first (two DPP row shifts under a bank mask).
__device__ __forceinline__ double quadToRow(double v) {      // lanes 4 b + i of every row of 16  <-  lane i of that row
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xF, 0x2, false);      // row_shr:4 into bank 1
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xF, 0x2, false);
  lo = __builtin_amdgcn_update_dpp(lo, lo, 0x118, 0xF, 0xC, false);      // row_shr:8 into banks 2, 3
  hi = __builtin_amdgcn_update_dpp(hi, hi, 0x118, 0xF, 0xC, false);
  return __hiloint2double(hi, lo);
}
template <int STEPS = 4>
__device__ __forceinline__ void pAcc4(wtile& acc, const wtile& X, const wtile& Y) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(quadToRow(X[s]), Y[s], acc[0], 0, 0, 0);
}
template <int STEPS = 4>
__device__ __forceinline__ void pSub4(wtile& acc, const wtile& X, const wtile& Y) {
#pragma unroll
  for (int s = 0; s < STEPS; ++s) acc[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(-quadToRow(X[s]), Y[s], acc[0], 0, 0, 0);
}
__device__ __forceinline__ wtile wzero() { return wtile{0.0, 0.0, 0.0, 0.0}; }

// register 4 q + g of sixteen per-lane values -> accumulator-layout register q (g = the lane's row of 16)
__device__ __forceinline__ wtile pickByGroup(const double (&v)[16], int g) {
  wtile t;
  const bool g1 = g & 1, g2 = g & 2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    // (the four candidates pass through an empty asm: a select between two array elements would otherwise be folded into ONE load with a
    //  selected address, i.e. a dynamically indexed array, i.e. the whole array in scratch memory)
    double v0 = v[4 * q], v1 = v[4 * q + 1], v2 = v[4 * q + 2], v3 = v[4 * q + 3];
    asm volatile("" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3));
    const double a = g1 ? v1 : v0, b = g1 ? v3 : v2;
    t[q] = g2 ? b : a;
  }
  return t;
}

// One diagonal block: lane li of every row of 16 lanes holds row li of the SPD 16 x 16 block in a[]; returns W = L^-1 (A = L L^T) in
// accumulator layout: the lane's right-hand side is the unit vector e_li, forward substitution only (x = column li of L^-1).
// NP < 16: the block is [A 0; 0 I] with A NP x NP -- only its NP pivots are walked.
template <int NP = 16>
__device__ __forceinline__ wtile cholInvPass16(double (&a)[16], int lane, int& bad) {
  const int li = lane & 15, g = lane >> 4;
  double x[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) x[k] = (k == li) ? 1.0 : 0.0;
  // (the rows / columns NP .. 15 of a padded block are the identity and never touched: the updates stop at NP)
  cholForwardFused<16, NP, false>(a, x, bad, NP);
  return pickByGroup(x, g);
}

// index of the tile (i, j), i <= j, of a symmetric / upper block-triangular 3 x 3 arrangement; a lower block (i, k), i >= k, of W is kept at U3(k, i)
__host__ __device__ constexpr int U3(int i, int j) { return i == 0 ? j : (i == 1 ? 2 + j : 5); }

template <typename D>
__global__ __launch_bounds__(64, 2) void parnmpc_kkt_inverse_wave_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = KktWaveSmem<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NK = L::NK, LDT = S::LDT;
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const OcpProblem* __restrict__ P = B.prob;
  // (uniform values are made scalar by hand: the kernel stores to global memory, so the problem block and the node table are read with
  //  vector loads, and addresses derived from them would live in vector registers)
  const int M = B.M;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const int unit = blockIdx.x;
  const int bi = unit / (M - 1);
  const int pos = unit - bi * (M - 1);
  const long b = bi;
  const OcpNode* __restrict__ nd = B.nodes + pos;
  if (__builtin_amdgcn_readfirstlane((int)parnmpcShape<L>(*nd).general)) return;
  const bool last = __builtin_amdgcn_readfirstlane(P->has_terminal) && (pos == M - 2);
  const double dt = nd->dt;
  const int NS = B.NS;
  const long rec = b * NS + __builtin_amdgcn_readfirstlane(nd->slot);
  auto uniformPtr = [](const double* p) -> const double* {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const double*>(((unsigned long long)hi << 32) | lo);
  };
  const double* __restrict__ kk = uniformPtr(B.kkt + rec * L::KKT);
  const double* __restrict__ aux = uniformPtr(B.aux + (b * NS + __builtin_amdgcn_readfirstlane(nd->next)) * L::AUX);
  double* __restrict__ ki = const_cast<double*>(uniformPtr(B.kinv + rec * L::KINV));
  // six raw tiles of Q^-1 wait here for the end of the kernel (the lin record of the stage: written by K5a, read by K5b<BWD>, dead by now)
  static_assert(L::LIN >= 6 * 256, "six tiles of scratch per stage");
  double* __restrict__ scr = const_cast<double*>(uniformPtr(B.lin + rec * L::LIN));
#ifdef IDOCP_K9_STAMPS      // (diagnostic build: per-phase clock stamps of one wavefront in the middle of the launch)
  const bool stamp = B.prof && blockIdx.x == gridDim.x / 2 + 7 && threadIdx.x == 0;
#define KSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp) B.prof[k] = wall_clock64(); } while (0)
#else
#define KSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); } while (0)      // (the phases are not to be interleaved: registers)
#endif
  KSTAMP(0);

  // ---- everything the stage reads, requested at once, 16 bytes per lane and load: aux_mat of the next stage, the Q part and the F part
  //      of the kkt record ----
  constexpr int NA2 = NX * NX / 2, NQ2 = S::QPART / 2, NF2 = S::FPART / 2, TA = (NA2 + 63) / 64, TQ = (NQ2 + 63) / 64, TF = (NF2 + 63) / 64;
  v2d rvA[TA], rvQ[TQ], rvF[TF];
#pragma unroll
  for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; rvA[t] = reinterpret_cast<const v2d*>(aux)[e < NA2 ? e : NA2 - 1]; }
#pragma unroll
  for (int t = 0; t < TQ; ++t) { const int e = lane + 64 * t; rvQ[t] = reinterpret_cast<const v2d*>(kk)[e < NQ2 ? e : NQ2 - 1]; }
#pragma unroll
  for (int t = 0; t < TF; ++t) { const int e = lane + 64 * t; rvF[t] = reinterpret_cast<const v2d*>(kk + S::QPART)[e < NF2 ? e : NF2 - 1]; }
  if (lane == 0) { sm[S::CPAD] = 0.0; sm[S::CPAD + 1] = -1.0; sm[S::CPAD + 2] = dt; sm[S::CPAD + 3] = 1.0; }

  // ---- tile helpers ----
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
  auto rowsOfTile = [&](const wtile& t, double (&a)[16]) {      // lane li of every row of 16 lanes <- row li of the (symmetric) tile
    tileToLds(t);
    waveLdsSync();
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = sm[S::TS + li * LDT + k];
    waveLdsSync();
  };
  int bad = 0;
  // A (six upper tiles of an SPD 48 x 48 matrix, destroyed) -> W = L^-1 (six lower tiles, W(i, k) at U3(k, i)); np2: pivots of the last block
  auto cholInv3 = [&](wtile (&A)[6], wtile (&W)[6], auto np2) {
    wtile R01, R02, R12, Gd;
    {
      double a[16];
      rowsOfTile(A[U3(0, 0)], a);
      W[U3(0, 0)] = cholInvPass16(a, lane, bad);
    }
    Gd = transposeTile(W[U3(0, 0)]);
    R01 = wzero(); pAcc(R01, Gd, A[U3(0, 1)]);          // R_0j = W_00 Q_0j  (= L_j0^T)
    R02 = wzero(); pAcc(R02, Gd, A[U3(0, 2)]);
    pSub(A[U3(1, 1)], R01, R01);
    pSub(A[U3(1, 2)], R01, R02);
    constexpr bool QUAD = decltype(np2)::value <= 4;      // the last block has four rows / columns: its products as 4 x 4 x 4 blocks
    if constexpr (QUAD) pSub4(A[U3(2, 2)], R02, R02); else pSub(A[U3(2, 2)], R02, R02);
    {
      double a[16];
      rowsOfTile(A[U3(1, 1)], a);
      W[U3(1, 1)] = cholInvPass16(a, lane, bad);
    }
    Gd = transposeTile(W[U3(1, 1)]);
    R12 = wzero(); pAcc(R12, Gd, A[U3(1, 2)]);
    if constexpr (QUAD) pSub4(A[U3(2, 2)], R12, R12); else pSub(A[U3(2, 2)], R12, R12);
    // W_10 = -W_11 L_10 W_00,  L_im = R_mi^T
    {
      wtile T = wzero();
      pAcc(T, R01, W[U3(0, 0)]);
      W[U3(0, 1)] = wzero(); pSub(W[U3(0, 1)], Gd, T);
    }
    {
      double a[16];
      rowsOfTile(A[U3(2, 2)], a);
      W[U3(2, 2)] = cholInvPass16<decltype(np2)::value>(a, lane, bad);
    }
    Gd = transposeTile(W[U3(2, 2)]);
    {
      wtile T = wzero();
      if constexpr (QUAD) {      // (T has four rows; of W_22^T only its 4 x 4 block meets them)
        pAcc4(T, R12, W[U3(1, 1)]);
        W[U3(1, 2)] = wzero(); pSub4<1>(W[U3(1, 2)], Gd, T);      // W_21
        T = wzero();
        pAcc4(T, R02, W[U3(0, 0)]);
        pAcc4(T, R12, W[U3(0, 1)]);
        W[U3(0, 2)] = wzero(); pSub4<1>(W[U3(0, 2)], Gd, T);      // W_20
      } else {
      pAcc(T, R12, W[U3(1, 1)]);
      W[U3(1, 2)] = wzero(); pSub(W[U3(1, 2)], Gd, T);      // W_21
      T = wzero();
      pAcc(T, R02, W[U3(0, 0)]);
      pAcc(T, R12, W[U3(0, 1)]);
      W[U3(0, 2)] = wzero(); pSub(W[U3(0, 2)], Gd, T);      // W_20
      }
    }
  };
  // parking place of tiles: 4 doubles per lane and slot
  auto parkTile = [&](int slot, const wtile& t) {
    reinterpret_cast<v2d*>(&sm[S::BUF + 256 * slot + 4 * lane])[0] = v2d{t[0], t[1]};
    reinterpret_cast<v2d*>(&sm[S::BUF + 256 * slot + 4 * lane])[1] = v2d{t[2], t[3]};
  };
  auto parkedTile = [&](int slot) -> wtile {
    int o = 4 * lane;
    asm volatile("" : "+v"(o));      // (every visit is a read of its own: merged into one early read the tiles would live in registers -- in scratch -- after all)
    const v2d lo = reinterpret_cast<const v2d*>(&sm[S::BUF + 256 * slot + o])[0], hi = reinterpret_cast<const v2d*>(&sm[S::BUF + 256 * slot + o])[1];
    return wtile{lo.x, lo.y, hi.x, hi.y};
  };

  // ---- Q (order u, q, v; SplitBackwardCorrection::coarseUpdate: Qxx += aux_mat_next): the six upper tiles, gathered from LDS -- first the
  //      aux_mat terms, then the record's ----
  wtile Wq[6];
  {
    wtile A[6];
#pragma unroll
    for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; if (e < NA2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = last ? v2d{0.0, 0.0} : rvA[t]; }
    waveLdsSync();
    KSTAMP(1);
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * it + 4 * q + g, c = 16 * jt + li;
          const bool xx = r >= NU && c >= NU;
          const int lo = r < c ? r : c, hi = r < c ? c : r;
          A[U3(it, jt)][q] = sm[xx ? S::BUF + (lo - NU) + NX * (hi - NU) : S::CPAD];
        }
    waveLdsSync();
#pragma unroll
    for (int t = 0; t < TQ; ++t) { const int e = lane + 64 * t; if (e < NQ2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = rvQ[t]; }
    waveLdsSync();
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * it + 4 * q + g, c = 16 * jt + li;
          const int lo = r < c ? r : c, hi = r < c ? c : r;
          int ad;
          if (hi < NU) ad = L::K_QUU + lo + NU * hi;
          else if (lo < NU) ad = L::K_QXU + (hi - NU) + NX * lo;
          else { const int rr = lo - NU, cc = hi - NU; ad = L::K_QXX + cc * (cc + 1) / 2 + rr; }
          A[U3(it, jt)][q] += sm[S::BUF + ad];
        }
    waveLdsSync();
    KSTAMP(2);
    cholInv3(A, Wq, std::integral_constant<int, 16>{});
  }
  // the F part of the record takes the place of the Q part; r1 = [Fq; Fv] is copied aside (BUF is reused before t1 is formed)
#pragma unroll
  for (int t = 0; t < TF; ++t) { const int e = lane + 64 * t; if (e < NF2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = rvF[t]; }
  waveLdsSync();
  if (lane < NX) sm[S::R1 + lane] = sm[S::BUF + (L::K_FX - S::QPART) + lane];
  KSTAMP(3);

  // ---- the output: a column block of C0 / C1 (84 rows x up to 16 columns, ld NK) is put together in BUF and leaves with 16-byte stores,
  //      64 lanes on 1 kB of consecutive memory.  A tile Z whose entry (row, col) belongs at kinv row r0 + col of the block's column
  //      c0 + row: ----
  auto stageTile = [&](const wtile& Z, int r0, int nrow, int c0, int ncol) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (every lane stores: a branch around each of the 126 stores of a stage cost five instructions of mask bookkeeping apiece)
      const int c = c0 + 4 * q + g;
      sm[(li < nrow && c >= 0 && c < ncol) ? S::BUF + (r0 + li) + NK * c : S::DUMP] = Z[q];
    }
  };
  auto flushBlock = [&](int first_col, int ncol) {
    waveLdsSync();
    const int n2 = NK * ncol / 2;
    v2d* __restrict__ dst = reinterpret_cast<v2d*>(ki + (long)NK * first_col);
#pragma unroll
    for (int t = 0; t < (NK * 16 / 2 + 63) / 64; ++t) {
      const int e = lane + 64 * t;
      if (64 * t + 63 < n2 || e < n2) dst[e] = reinterpret_cast<const v2d*>(&sm[S::BUF])[e];      // (only the last piece of a block is partial)
    }
    waveLdsSync();
  };
  double pt[3] = {0.0, 0.0, 0.0}, pb[3] = {0.0, 0.0, 0.0};       // top = -N^T z, bottom = W^T y2 + TR^T t1 (split_backward_correction.hxx:30-47)

  // ---- Y = W F^T = P(G, F^T), G(kt, it) = W(it, kt)^T on the fly.  F^T (48 x 48: columns 0 .. 35 the rows of F = [0 Fqq Fqv; Fvu Fvq Fvv],
  //      backward Euler: Fqq = -I, Fqv = dt I outside the base blocks; column 36 = r2 = [lu; lx]; the rest zero): one block row (three tiles)
  //      at a time from the staged F part, structural entries from CPAD ----
  auto ftRow = [&](int kt, wtile (&Fr)[3]) {
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int z = 16 * kt + 4 * q + g, e = 16 * jt + li;
        int ad = S::CPAD;
        if (e < NV) {
          if (z >= NU && z < NU + NV) { const int cq = z - NU; ad = (e < 6 && cq < 6) ? S::BUF - S::QPART + L::K_FQQ + e + 6 * cq : ((e >= 6 && e == cq) ? S::CPAD + 1 : S::CPAD); }
          else if (z >= NU + NV) { const int cv = z - NU - NV; ad = (e < 6 && cv < 6) ? S::BUF - S::QPART + L::K_FQV + e + 6 * cv : ((e >= 6 && e == cv) ? S::CPAD + 2 : S::CPAD); }
        } else if (e < NX) {
          const int rv_ = e - NV;
          if (z < NU) ad = S::BUF - S::QPART + L::K_FVU + rv_ + NV * z;
          else if (z < NU + NV) ad = S::BUF - S::QPART + L::K_FVQ + rv_ + NV * (z - NU);
          else ad = S::BUF - S::QPART + L::K_FVV + rv_ + NV * (z - NU - NV);
        } else if (e == NX) {
          ad = z < NU ? S::BUF - S::QPART + L::K_LU + z : S::BUF - S::QPART + L::K_LX + (z - NU);
        }
        Fr[jt][q] = sm[ad];
      }
  };
  wtile Sm[6];
  {
    wtile Y[3][3];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) Y[it][jt] = wzero();
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      wtile Fr[3];
      ftRow(kt, Fr);
#pragma unroll
      for (int it = kt; it < 3; ++it) {
        const wtile Gt = transposeTile(Wq[U3(kt, it)]);
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) pAcc(Y[it][jt], Gt, Fr[jt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    KSTAMP(4);
    // y2 = W r2 = column 36 of Y
    if (li == 4) {
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) sm[S::Y2 + 16 * it + 4 * q + g] = Y[it][2][q];
    }
    // bottom of the coarse direction, first part: W^T y2 (per lane the terms of its rows; summed over the rows of lanes at the end)
    waveLdsSync();
    {
      double y2v[3][4];
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) y2v[it][q] = sm[S::Y2 + 16 * it + 4 * q + g];
#pragma unroll
      for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int kt = jt; kt < 3; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) pb[jt] += Wq[U3(jt, kt)][q] * y2v[kt][q];
      // (evaluated HERE: the sums are needed at the very end, and the compiler would sink the whole computation -- with y2 and the six tiles of
      //  W alive, in scratch -- down to there)
      asm volatile("" : "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // Q^-1 = W^T W: the six upper tiles wait in global scratch for the end of the kernel (BR = Q^-1 - Z1^T Z1); 32 bytes per lane, the
    // wavefront on 2 kB of consecutive memory
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt) {
        wtile Z = wzero();
#pragma unroll
        for (int kt = jt; kt < 3; ++kt) pAcc(Z, Wq[U3(it, kt)], Wq[U3(jt, kt)]);
        reinterpret_cast<v2d*>(scr + 256 * U3(it, jt))[2 * lane] = v2d{Z[0], Z[1]};
        reinterpret_cast<v2d*>(scr + 256 * U3(it, jt))[2 * lane + 1] = v2d{Z[2], Z[3]};
        __builtin_amdgcn_sched_barrier(0);
      }
    // FQ = F Q^-1 = Y^T W = P(Y, W) (36 x 48), parked in LDS during the factorisation of S (of its rows 32 .. 47 only 32 .. 35 exist: one
    // register per lane): slot 3 it + jt
    waveLdsSync();
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) {
        wtile Z = wzero();
#pragma unroll
        for (int kt = jt; kt < 3; ++kt) { if (it == 2) pAcc4(Z, Y[kt][it], Wq[U3(jt, kt)]); else pAcc(Z, Y[kt][it], Wq[U3(jt, kt)]); }
        if (it < 2) parkTile(3 * it + jt, Z);
        else sm[S::BUF + 6 * 256 + 64 * jt + lane] = Z[0];
        __builtin_amdgcn_sched_barrier(0);
      }
    // S = P(Y, Y) (upper tiles); column 36 = F Q^-1 r2
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt) {
        Sm[U3(it, jt)] = wzero();
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) { if (it == 2) pAcc4(Sm[U3(it, jt)], Y[kt][it], Y[kt][jt]); else pAcc(Sm[U3(it, jt)], Y[kt][it], Y[kt][jt]); }
      }
  }
  // t1 = r1 - F Q^-1 r2 (zero beyond row 35), from the lanes that hold column 36 of S
  if (li == 4) {
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * it + 4 * q + g;
        sm[S::T1 + row] = row < NX ? sm[S::R1 + (row < NX ? row : 0)] - Sm[U3(it, 2)][q] : 0.0;
      }
  }
  // the padding of S: rows / columns 36 .. 47 = identity
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * it + 4 * q + g, col = 32 + li;
      if (col >= NX || row >= NX) Sm[U3(it, 2)][q] = (row == col) ? 1.0 : 0.0;
    }
  wtile Ns[6];
  KSTAMP(5);
  cholInv3(Sm, Ns, std::integral_constant<int, NX - 32>{});
  KSTAMP(6);
  waveLdsSync();
  double t1v[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) t1v[it][q] = sm[S::T1 + 16 * it + 4 * q + g];

  // ---- Z1 = N FQ = P(H, FQ) (36 x 48), H(kt, it) = N(it, kt)^T on the fly; z = N t1 = H^T t1 on the way (a product M^T x contracts over the
  //      rows of accumulator-layout tiles: per lane the terms of its rows, then the four rows of 16 lanes are summed through LDS) ----
  auto reduceRows = [&](const double (&p)[3], double* out) {       // out[0 .. 47] = sum over the rows of lanes
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) sm[S::RED + 48 * g + 16 * jt + li] = p[jt];
    waveLdsSync();
    if (lane < 48) out[lane] = (sm[S::RED + lane] + sm[S::RED + 48 + lane]) + (sm[S::RED + 96 + lane] + sm[S::RED + 144 + lane]);
    waveLdsSync();
  };
  wtile Z1[3][3];
  {
    double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int it = 0; it < 3; ++it) {
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) Z1[it][jt] = wzero();
#pragma unroll
      for (int kt = 0; kt <= it; ++kt) {
        const wtile Ht = transposeTile(Ns[U3(kt, it)]);
#pragma unroll
        for (int q = 0; q < 4; ++q) p[it] += Ht[q] * t1v[kt][q];
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) {
          if (it == 2) {      // (four rows of Z1)
            if (kt < 2) pAcc4<4>(Z1[it][jt], Ht, parkedTile(3 * kt + jt));
            else pAcc4<1>(Z1[it][jt], Ht, wtile{sm[S::BUF + 6 * 256 + 64 * jt + lane], 0.0, 0.0, 0.0});
          } else if (kt < 2) pAcc<4>(Z1[it][jt], Ht, parkedTile(3 * kt + jt));
          else pAcc<1>(Z1[it][jt], Ht, wtile{sm[S::BUF + 6 * 256 + 64 * jt + lane], 0.0, 0.0, 0.0});
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    reduceRows(p, &sm[S::ZZ]);
  }
  double zv[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int row = 16 * it + 4 * q + g; zv[it][q] = row < NX ? sm[S::ZZ + row] : 0.0; }
  // top of the coarse direction: -N^T z
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kt = jt; kt < 3; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) pt[jt] -= Ns[U3(jt, kt)][q] * zv[kt][q];
  // bottom, second part: TR^T t1 = Z1^T N t1 = Z1^T z
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) pb[jt] += Z1[kt][jt][q] * zv[kt][q];
  // ---- the coarse direction and s_new = s - direction (split_backward_correction.hxx:30-58), in front of the output: its loads and stores
  //      do not queue behind 48 kB of stores ----
  {
    double* dir = &sm[S::DIR];                 // dlmd dgmm | du dq dv
    reduceRows(pt, &sm[S::ZZ]);
    if (lane < NX) dir[lane] = sm[S::ZZ + lane];
    reduceRows(pb, dir + NX);
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
  }
  // Q^-1: the six raw tiles come back from the scratch while C0 is put together
  wtile Qi[6];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (stored by these very lanes, never in this CU's L1 before: the stores only have to be complete)
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const v2d lo = reinterpret_cast<const v2d*>(scr + 256 * t)[2 * lane], hi = reinterpret_cast<const v2d*>(scr + 256 * t)[2 * lane + 1];
    Qi[t] = wtile{lo.x, lo.y, hi.x, hi.y};
  }
  KSTAMP(7);

  // ---- C0 = [TL; TR^T], TL = -S^-1 = -N^T N, TR = S^-1 F Q^-1 = N^T Z1, one block of columns a (rows a of S) at a time: kinv entry
  //      (rho, kappa) = TL(kappa, rho) for rho < 36, TR(kappa, rho - 36) below -- the tiles (a, b) as they are ----
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int ncol = a < 2 ? 16 : NX - 32;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      wtile Z = wzero();
#pragma unroll
      for (int kt = (a > bb ? a : bb); kt < 3; ++kt) {
        if (kt < 2) pSub<4>(Z, Ns[U3(a, kt)], Ns[U3(bb, kt)]);
        else pSub<1>(Z, Ns[U3(a, kt)], Ns[U3(bb, kt)]);
      }
      stageTile(Z, 16 * bb, NX - 16 * bb, 0, ncol);
      wtile R = wzero();
#pragma unroll
      for (int kt = a; kt < 3; ++kt) {
        if (kt < 2) pAcc<4>(R, Ns[U3(a, kt)], Z1[kt][bb]);
        else pAcc<1>(R, Ns[U3(a, kt)], Z1[kt][bb]);
      }
      stageTile(R, NX + 16 * bb, 16, 0, ncol);
      __builtin_amdgcn_sched_barrier(0);
    }
    flushBlock(L::I_C0 / NK + 16 * a, ncol);
  }
  KSTAMP(8);
  // ---- C1 = [TR(:, NU:); BR(:, NU:)], BR = Q^-1 - Z1^T Z1, one block of columns a (columns 16 a .. of (u, q, v), of which 12 .. exist) at a
  //      time: kinv entry (rho, kappa) = TR(rho, NU + kappa) for rho < 36 -- the tile (a, b) of TR^T = Z1^T N --, BR(NU + kappa, rho - 36) below ----
  static_assert(L::I_C0 % NK == 0 && L::I_C1 % NK == 0, "column blocks of the record");
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int c0 = a == 0 ? -NU : 0;                         // the block's first column is max(0, 16 a - NU) of C1
    const int first = a == 0 ? 0 : 16 * a - NU, ncol = a == 0 ? 16 - NU : 16;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      wtile R = wzero();                                     // TR^T(a, b) = sum_kt Z1(kt, a)^T N(kt, b), kt >= b
#pragma unroll
      for (int kt = bb; kt < 3; ++kt) {
        if (kt < 2) pAcc<4>(R, Z1[kt][a], Ns[U3(bb, kt)]);
        else pAcc<1>(R, Z1[kt][a], Ns[U3(bb, kt)]);
      }
      stageTile(R, 16 * bb, NX - 16 * bb, c0, ncol);
      // BR(a, b): Q^-1(a, b) from the scratch (the transposed tile below the diagonal)
      wtile Z = Qi[a <= bb ? U3(a, bb) : U3(bb, a)];
      if (a > bb) Z = transposeTile(Z);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        if (kt < 2) pSub<4>(Z, Z1[kt][a], Z1[kt][bb]);
        else pSub<1>(Z, Z1[kt][a], Z1[kt][bb]);
      }
      stageTile(Z, NX + 16 * bb, 16, c0, ncol);
      __builtin_amdgcn_sched_barrier(0);
    }
    flushBlock(L::I_C1 / NK + first, ncol);
  }
  KSTAMP(10);
  KSTAMP(11);
#undef KSTAMP
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && lane == 0 && B.status[b] == 0) B.status[b] = 1000 + pos;
}

// K9w on the event stages (aux stages with switching rows, impulse stages): everything padded to 48 + 48 like K9g's register tiles; S may have
// all of its 48 rows, so J Q^-1 is parked as nine full tiles, and the constraint Jacobian travels next to the F part
template <typename D>
struct KktWaveSmemG {
  using L = OcpLayout<D>;
  static constexpr int NX = D::NX, NU = D::NU, NF = D::NF, NQ_ = NU + NX;
  static_assert(NQ_ == 48 && NX + NF == 48, "three 16 x 16 blocks for (w, q, v) and for (lmd, gmm, xi)");
  static constexpr int LDT = 18;
  static constexpr int QPART = L::K_QUU + NU * NU, FPART = L::K_FX + NX - QPART;
  static constexpr int NBUF = 9 * 256;
  static constexpr int PHIX = ((FPART + 15) / 16) * 16;      // (offset within BUF of the constraint Jacobian, NF x NX, ld NF)
  static_assert(L::K_QXX == 0 && L::K_QXU < L::K_QUU && L::K_FQQ == QPART && QPART % 2 == 0 && FPART % 2 == 0, "record order");
  static_assert(NX * NX <= NBUF && QPART <= NBUF && PHIX + NF * NX <= NBUF && L::NKG * 16 <= NBUF, "everything that passes through BUF");
  static constexpr int BUF = 0, CPAD = BUF + NBUF, TS = CPAD + 4, RED = TS + 16 * LDT, Y2 = RED + 4 * 48, T1 = Y2 + 48, ZZ = T1 + 48, R1 = ZZ + 48,
                       DIR = R1 + 48, DUMP = DIR + 48 + 48 + 4, TOTAL = DUMP + 2;
  static_assert(TS % 2 == 0, "16-byte pieces");
};

template <typename D>
__global__ __launch_bounds__(64, 2) void parnmpc_kkt_inverse_wave_general_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = KktWaveSmemG<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF, NC = D::NC, NK = L::NKG, LDT = S::LDT;      // (NK: the leading dimension of the kinv record of an event stage)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  const OcpProblem* __restrict__ P = B.prob;
  // (uniform values are made scalar by hand: the kernel stores to global memory, so the problem block and the node table are read with
  //  vector loads, and addresses derived from them would live in vector registers)
  const int M = B.M;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const int pos = __builtin_amdgcn_readfirstlane(B.general_pos[blockIdx.y]);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  // the shape of the stage's KKT matrix: ni extra constraint rows (switching constraint of an aux stage / contact-velocity constraint of an
  // impulse stage), nw inputs (u, or the packed impulse forces f); everything is PADDED to 48 + 48 with the identity (K9g's scheme)
  const ParnmpcShape sh = parnmpcShape<L>(*nd);
  const int ni = __builtin_amdgcn_readfirstlane(sh.ni), nw = __builtin_amdgcn_readfirstlane(sh.nw), nr = NX + ni;
  const bool impulse = __builtin_amdgcn_readfirstlane((int)sh.impulse) != 0;
  const bool last = __builtin_amdgcn_readfirstlane(P->has_terminal) && (pos == M - 2);
  const double dt = nd->dt;
  const int NS = B.NS;
  const long rec = b * NS + __builtin_amdgcn_readfirstlane(nd->slot);
  auto uniformPtr = [](const double* p) -> const double* {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const double*>(((unsigned long long)hi << 32) | lo);
  };
  const double* __restrict__ kk = uniformPtr(B.kkt + rec * L::KKT);
  const double* __restrict__ aux = uniformPtr(B.aux + (b * NS + __builtin_amdgcn_readfirstlane(nd->next)) * L::AUX);
  double* __restrict__ ki = const_cast<double*>(uniformPtr(B.kinv + rec * L::KINV));
  // six raw tiles of Q^-1 wait here for the end of the kernel (the lin record of the stage: written by K5a, read by K5b<BWD>, dead by now)
  static_assert(L::LIN >= 6 * 256, "six tiles of scratch per stage");
  double* __restrict__ scr = const_cast<double*>(uniformPtr(B.lin + rec * L::LIN));      // (an impulse stage's lin record was read by K9i: dead as well)
  const double* __restrict__ Wc = uniformPtr(B.swc + rec * L::SWC);
#ifdef IDOCP_K9_STAMPS      // (diagnostic build: per-phase clock stamps of one wavefront in the middle of the launch)
  const bool stamp = B.prof && blockIdx.x == gridDim.x / 2 && blockIdx.y == 0 && threadIdx.x == 0;
#define KSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); if (stamp) B.prof[k] = wall_clock64(); } while (0)
#else
#define KSTAMP(k) do { __builtin_amdgcn_sched_barrier(0); } while (0)      // (the phases are not to be interleaved: registers)
#endif
  KSTAMP(0);

  // ---- everything the stage reads, requested at once, 16 bytes per lane and load: aux_mat of the next stage, the Q part and the F part
  //      of the kkt record ----
  constexpr int NA2 = NX * NX / 2, NQ2 = S::QPART / 2, NF2 = S::FPART / 2, TA = (NA2 + 63) / 64, TQ = (NQ2 + 63) / 64, TF = (NF2 + 63) / 64;
  static_assert(L::W_PHIX % 2 == 0 && (NF * NX) % 2 == 0, "16-byte pieces of the constraint Jacobian");
  constexpr int NP2 = NF * NX / 2, TP = (NP2 + 63) / 64;
  v2d rvA[TA], rvQ[TQ], rvF[TF], rvP[TP];
#pragma unroll
  for (int t = 0; t < TP; ++t) { const int e = lane + 64 * t; rvP[t] = reinterpret_cast<const v2d*>(Wc + L::W_PHIX)[e < NP2 ? e : NP2 - 1]; }
  const double wp_r = Wc[L::W_P + (lane < NF ? lane : 0)];
#pragma unroll
  for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; rvA[t] = reinterpret_cast<const v2d*>(aux)[e < NA2 ? e : NA2 - 1]; }
#pragma unroll
  for (int t = 0; t < TQ; ++t) { const int e = lane + 64 * t; rvQ[t] = reinterpret_cast<const v2d*>(kk)[e < NQ2 ? e : NQ2 - 1]; }
#pragma unroll
  for (int t = 0; t < TF; ++t) { const int e = lane + 64 * t; rvF[t] = reinterpret_cast<const v2d*>(kk + S::QPART)[e < NF2 ? e : NF2 - 1]; }
  if (lane == 0) { sm[S::CPAD] = 0.0; sm[S::CPAD + 1] = -1.0; sm[S::CPAD + 2] = dt; sm[S::CPAD + 3] = 1.0; }

  // ---- tile helpers ----
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
  auto rowsOfTile = [&](const wtile& t, double (&a)[16]) {      // lane li of every row of 16 lanes <- row li of the (symmetric) tile
    tileToLds(t);
    waveLdsSync();
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = sm[S::TS + li * LDT + k];
    waveLdsSync();
  };
  int bad = 0;
  // A (six upper tiles of an SPD 48 x 48 matrix, destroyed) -> W = L^-1 (six lower tiles, W(i, k) at U3(k, i)); np2: pivots of the last block
  auto cholInv3 = [&](wtile (&A)[6], wtile (&W)[6], auto np2) {
    wtile R01, R02, R12, Gd;
    {
      double a[16];
      rowsOfTile(A[U3(0, 0)], a);
      W[U3(0, 0)] = cholInvPass16(a, lane, bad);
    }
    Gd = transposeTile(W[U3(0, 0)]);
    R01 = wzero(); pAcc(R01, Gd, A[U3(0, 1)]);          // R_0j = W_00 Q_0j  (= L_j0^T)
    R02 = wzero(); pAcc(R02, Gd, A[U3(0, 2)]);
    pSub(A[U3(1, 1)], R01, R01);
    pSub(A[U3(1, 2)], R01, R02);
    pSub(A[U3(2, 2)], R02, R02);
    {
      double a[16];
      rowsOfTile(A[U3(1, 1)], a);
      W[U3(1, 1)] = cholInvPass16(a, lane, bad);
    }
    Gd = transposeTile(W[U3(1, 1)]);
    R12 = wzero(); pAcc(R12, Gd, A[U3(1, 2)]);
    pSub(A[U3(2, 2)], R12, R12);
    // W_10 = -W_11 L_10 W_00,  L_im = R_mi^T
    {
      wtile T = wzero();
      pAcc(T, R01, W[U3(0, 0)]);
      W[U3(0, 1)] = wzero(); pSub(W[U3(0, 1)], Gd, T);
    }
    {
      double a[16];
      rowsOfTile(A[U3(2, 2)], a);
      W[U3(2, 2)] = cholInvPass16<decltype(np2)::value>(a, lane, bad);
    }
    Gd = transposeTile(W[U3(2, 2)]);
    {
      wtile T = wzero();
      pAcc(T, R12, W[U3(1, 1)]);
      W[U3(1, 2)] = wzero(); pSub(W[U3(1, 2)], Gd, T);      // W_21
      T = wzero();
      pAcc(T, R02, W[U3(0, 0)]);
      pAcc(T, R12, W[U3(0, 1)]);
      W[U3(0, 2)] = wzero(); pSub(W[U3(0, 2)], Gd, T);      // W_20
    }
  };
  // parking place of tiles: 4 doubles per lane and slot
  auto parkTile = [&](int slot, const wtile& t) {
    reinterpret_cast<v2d*>(&sm[S::BUF + 256 * slot + 4 * lane])[0] = v2d{t[0], t[1]};
    reinterpret_cast<v2d*>(&sm[S::BUF + 256 * slot + 4 * lane])[1] = v2d{t[2], t[3]};
  };
  auto parkedTile = [&](int slot) -> wtile {
    int o = 4 * lane;
    asm volatile("" : "+v"(o));      // (every visit is a read of its own: merged into one early read the tiles would live in registers -- in scratch -- after all)
    const v2d lo = reinterpret_cast<const v2d*>(&sm[S::BUF + 256 * slot + o])[0], hi = reinterpret_cast<const v2d*>(&sm[S::BUF + 256 * slot + o])[1];
    return wtile{lo.x, lo.y, hi.x, hi.y};
  };

  // ---- Q (order u, q, v; SplitBackwardCorrection::coarseUpdate: Qxx += aux_mat_next): the six upper tiles, gathered from LDS -- first the
  //      aux_mat terms, then the record's ----
  wtile Wq[6];
  {
    wtile A[6];
#pragma unroll
    for (int t = 0; t < TA; ++t) { const int e = lane + 64 * t; if (e < NA2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = last ? v2d{0.0, 0.0} : rvA[t]; }
    waveLdsSync();
    KSTAMP(1);
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * it + 4 * q + g, c = 16 * jt + li;
          const bool xx = r >= NU && c >= NU;
          const int lo = r < c ? r : c, hi = r < c ? c : r;
          A[U3(it, jt)][q] = sm[xx ? S::BUF + (lo - NU) + NX * (hi - NU) : S::CPAD];
        }
    waveLdsSync();
#pragma unroll
    for (int t = 0; t < TQ; ++t) { const int e = lane + 64 * t; if (e < NQ2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = rvQ[t]; }
    waveLdsSync();
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * it + 4 * q + g, c = 16 * jt + li;
          const int lo = r < c ? r : c, hi = r < c ? c : r;
          int ad;      // (w = u or f padded to NU entries: identity on the padding; CPAD holds 0, -1, dt, 1)
          if (hi < NU) ad = (hi < nw) ? S::BUF + L::K_QUU + lo + NU * hi : (lo == hi ? S::CPAD + 3 : S::CPAD);
          else if (lo < NU) ad = lo < nw ? S::BUF + L::K_QXU + (hi - NU) + NX * lo : S::CPAD;
          else { const int rr = lo - NU, cc = hi - NU; ad = S::BUF + L::K_QXX + cc * (cc + 1) / 2 + rr; }
          A[U3(it, jt)][q] += sm[ad];
        }
    waveLdsSync();
    KSTAMP(2);
    cholInv3(A, Wq, std::integral_constant<int, 16>{});
  }
  // the F part of the record takes the place of the Q part; r1 = [Fq; Fv] is copied aside (BUF is reused before t1 is formed)
#pragma unroll
  for (int t = 0; t < TF; ++t) { const int e = lane + 64 * t; if (e < NF2) reinterpret_cast<v2d*>(&sm[S::BUF])[e] = rvF[t]; }
  waveLdsSync();
#pragma unroll
  for (int t = 0; t < TP; ++t) { const int e = lane + 64 * t; if (e < NP2) reinterpret_cast<v2d*>(&sm[S::BUF + S::PHIX])[e] = rvP[t]; }
  // r1 = [Fq; Fv; P or V residual (ni rows); 0], copied aside (BUF is reused before t1 is formed)
  if (lane < NX) sm[S::R1 + lane] = sm[S::BUF + (L::K_FX - S::QPART) + lane];
  if (lane < NF) sm[S::R1 + NX + lane] = lane < ni ? wp_r : 0.0;
  waveLdsSync();
  KSTAMP(3);

  // ---- the output: a column block of C0 / C1 (up to 96 rows x up to 16 columns, ld NK = NKG; rows: lmd gmm | xi or mu (ni) | the nw true rows of w | q v) is put together in BUF and leaves with 16-byte stores,
  //      64 lanes on 1 kB of consecutive memory.  A tile Z whose entry (row, col) belongs at kinv row r0 + col of the block's column
  //      c0 + row: ----
  auto stageTile = [&](const wtile& Z, int r0, int nrow, int c0, int ncol) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      // (every lane stores: a branch around each of the 126 stores of a stage cost five instructions of mask bookkeeping apiece)
      const int c = c0 + 4 * q + g;
      sm[(li < nrow && c >= 0 && c < ncol) ? S::BUF + (r0 + li) + NK * c : S::DUMP] = Z[q];
    }
  };
  // the same for a tile whose columns are VARIABLES 16 bz + li of the padded order (w padded to NU, q, v): kinv row nr + the true index
  // (the padding rows of w are dropped)
  auto stageVarTile = [&](const wtile& Z, int bz, int c0, int ncol) {
    const int z = 16 * bz + li;
    const int vrow = z < NU ? (z < nw ? nr + z : -1) : nr + nw + (z - NU);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c = c0 + 4 * q + g;
      if (vrow >= 0 && c >= 0 && c < ncol) sm[S::BUF + vrow + NK * c] = Z[q];
    }
  };
  auto flushBlock = [&](int first_col, int ncol) {
    waveLdsSync();
    const int n2 = NK * ncol / 2;
    v2d* __restrict__ dst = reinterpret_cast<v2d*>(ki + (long)NK * first_col);
#pragma unroll
    for (int t = 0; t < (NK * 16 / 2 + 63) / 64; ++t) {
      const int e = lane + 64 * t;
      if (64 * t + 63 < n2 || e < n2) dst[e] = reinterpret_cast<const v2d*>(&sm[S::BUF])[e];      // (only the last piece of a block is partial)
    }
    waveLdsSync();
  };
  auto reduceRows = [&](const double (&p)[3], double* out) {       // out[0 .. 47] = sum over the rows of lanes
#pragma unroll
    for (int jt = 0; jt < 3; ++jt) sm[S::RED + 48 * g + 16 * jt + li] = p[jt];
    waveLdsSync();
    if (lane < 48) out[lane] = (sm[S::RED + lane] + sm[S::RED + 48 + lane]) + (sm[S::RED + 96 + lane] + sm[S::RED + 144 + lane]);
    waveLdsSync();
  };
  double pt[3] = {0.0, 0.0, 0.0}, pb[3] = {0.0, 0.0, 0.0};       // top = -N^T z, bottom = W^T y2 + TR^T t1 (split_backward_correction.hxx:30-47)

  // ---- Y = W F^T = P(G, F^T), G(kt, it) = W(it, kt)^T on the fly.  F^T (48 x 48: columns 0 .. 35 the rows of F = [0 Fqq Fqv; Fvu Fvq Fvv],
  //      backward Euler: Fqq = -I, Fqv = dt I outside the base blocks; column 36 = r2 = [lu; lx]; the rest zero): one block row (three tiles)
  //      at a time from the staged F part, structural entries from CPAD ----
  auto ftRow = [&](int kt, wtile (&Fr)[3]) {
#pragma unroll
    for (int jt = 0; jt < 3; ++jt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int z = 16 * kt + 4 * q + g, e = 16 * jt + li;
        int ad = S::CPAD;
        if (e < NV) {
          if (z >= NU && z < NU + NV) { const int cq = z - NU; ad = (e < 6 && cq < 6) ? S::BUF - S::QPART + L::K_FQQ + e + 6 * cq : ((e >= 6 && e == cq) ? S::CPAD + 1 : S::CPAD); }
          else if (z >= NU + NV && !impulse) { const int cv = z - NU - NV; ad = (e < 6 && cv < 6) ? S::BUF - S::QPART + L::K_FQV + e + 6 * cv : ((e >= 6 && e == cv) ? S::CPAD + 2 : S::CPAD); }
        } else if (e < NX) {
          const int rv_ = e - NV;
          if (z < NU) ad = z < nw ? S::BUF - S::QPART + L::K_FVU + rv_ + NV * z : S::CPAD;
          else if (z < NU + NV) ad = S::BUF - S::QPART + L::K_FVQ + rv_ + NV * (z - NU);
          else ad = S::BUF - S::QPART + L::K_FVV + rv_ + NV * (z - NU - NV);
        } else if (e < nr && z >= NU) {
          ad = S::BUF + S::PHIX + (e - NX) + NF * (z - NU);      // the rows of the switching / contact-velocity constraint
        }
        Fr[jt][q] = sm[ad];
      }
  };
  wtile Sm[6];
  {
    wtile Y[3][3];
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) Y[it][jt] = wzero();
    // (no free column for the vectors here -- S may have all of its 48 rows --: y2 = W r2 and F Q^-1 r2 = Y^T y2 are contractions over the rows
    //  of tiles, r2 = [lw (0 on the padding); lx])
    double py[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      wtile Fr[3];
      ftRow(kt, Fr);
      double r2v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int z = 16 * kt + 4 * q + g;
        r2v[q] = sm[z < NU ? (z < nw ? S::BUF - S::QPART + L::K_LU + z : S::CPAD) : S::BUF - S::QPART + L::K_LX + (z - NU)];
      }
#pragma unroll
      for (int it = kt; it < 3; ++it) {
        const wtile Gt = transposeTile(Wq[U3(kt, it)]);
#pragma unroll
        for (int q = 0; q < 4; ++q) py[it] += Gt[q] * r2v[q];
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) pAcc(Y[it][jt], Gt, Fr[jt]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    KSTAMP(4);
    reduceRows(py, &sm[S::Y2]);
    // bottom of the coarse direction, first part: W^T y2 (per lane the terms of its rows; summed over the rows of lanes at the end)
    {
      double y2v[3][4];
#pragma unroll
      for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int q = 0; q < 4; ++q) y2v[it][q] = sm[S::Y2 + 16 * it + 4 * q + g];
#pragma unroll
      for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int kt = jt; kt < 3; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) pb[jt] += Wq[U3(jt, kt)][q] * y2v[kt][q];
      // J Q^-1 r2 = Y^T y2, and t1 = r1 - J Q^-1 r2 (zero on the padding rows)
      double ps[3] = {0.0, 0.0, 0.0};
#pragma unroll
      for (int jt = 0; jt < 3; ++jt)
#pragma unroll
        for (int kt = 0; kt < 3; ++kt)
#pragma unroll
          for (int q = 0; q < 4; ++q) ps[jt] += Y[kt][jt][q] * y2v[kt][q];
      reduceRows(ps, &sm[S::T1]);
      if (lane < 48) sm[S::T1 + lane] = lane < nr ? sm[S::R1 + lane] - sm[S::T1 + lane] : 0.0;
      // (evaluated HERE: the sums are needed at the very end, and the compiler would sink the whole computation -- with y2 and the six tiles of
      //  W alive, in scratch -- down to there)
      asm volatile("" : "+v"(pb[0]), "+v"(pb[1]), "+v"(pb[2]));
    }
    __builtin_amdgcn_sched_barrier(0);
    // Q^-1 = W^T W: the six upper tiles wait in global scratch for the end of the kernel (BR = Q^-1 - Z1^T Z1); 32 bytes per lane, the
    // wavefront on 2 kB of consecutive memory
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt) {
        wtile Z = wzero();
#pragma unroll
        for (int kt = jt; kt < 3; ++kt) pAcc(Z, Wq[U3(it, kt)], Wq[U3(jt, kt)]);
        reinterpret_cast<v2d*>(scr + 256 * U3(it, jt))[2 * lane] = v2d{Z[0], Z[1]};
        reinterpret_cast<v2d*>(scr + 256 * U3(it, jt))[2 * lane + 1] = v2d{Z[2], Z[3]};
        __builtin_amdgcn_sched_barrier(0);
      }
    // JQ = J Q^-1 = Y^T W = P(Y, W) (48 x 48), parked in LDS during the factorisation of S: slot 3 it + jt
    waveLdsSync();
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) {
        wtile Z = wzero();
#pragma unroll
        for (int kt = jt; kt < 3; ++kt) pAcc(Z, Y[kt][it], Wq[U3(jt, kt)]);
        parkTile(3 * it + jt, Z);
        __builtin_amdgcn_sched_barrier(0);
      }
    // S = P(Y, Y) (upper tiles); column 36 = F Q^-1 r2
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
      for (int jt = it; jt < 3; ++jt) {
        Sm[U3(it, jt)] = wzero();
#pragma unroll
        for (int kt = 0; kt < 3; ++kt) pAcc(Sm[U3(it, jt)], Y[kt][it], Y[kt][jt]);
      }
  }
  // the padding of S: rows / columns nr .. 47 = identity (exact zeros there already, J has no such rows: only the diagonal is set)
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 16 * it + 4 * q + g, col = 32 + li;
      if (col >= nr || row >= nr) Sm[U3(it, 2)][q] = (row == col) ? 1.0 : 0.0;
    }
  wtile Ns[6];
  KSTAMP(5);
  cholInv3(Sm, Ns, std::integral_constant<int, 16>{});
  KSTAMP(6);
  waveLdsSync();
  double t1v[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) t1v[it][q] = sm[S::T1 + 16 * it + 4 * q + g];

  // ---- Z1 = N FQ = P(H, FQ) (36 x 48), H(kt, it) = N(it, kt)^T on the fly; z = N t1 = H^T t1 on the way (a product M^T x contracts over the
  //      rows of accumulator-layout tiles: per lane the terms of its rows, then the four rows of 16 lanes are summed through LDS) ----
  wtile Z1[3][3];
  {
    double p[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int it = 0; it < 3; ++it) {
#pragma unroll
      for (int jt = 0; jt < 3; ++jt) Z1[it][jt] = wzero();
#pragma unroll
      for (int kt = 0; kt <= it; ++kt) {
        const wtile Ht = transposeTile(Ns[U3(kt, it)]);
#pragma unroll
        for (int q = 0; q < 4; ++q) p[it] += Ht[q] * t1v[kt][q];
#pragma unroll
        for (int jt = 0; jt < 3; ++jt) {
          pAcc<4>(Z1[it][jt], Ht, parkedTile(3 * kt + jt));
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    reduceRows(p, &sm[S::ZZ]);
  }
  double zv[3][4];
#pragma unroll
  for (int it = 0; it < 3; ++it)
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int row = 16 * it + 4 * q + g; zv[it][q] = row < nr ? sm[S::ZZ + row] : 0.0; }
  // top of the coarse direction: -N^T z
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kt = jt; kt < 3; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) pt[jt] -= Ns[U3(jt, kt)][q] * zv[kt][q];
  // bottom, second part: TR^T t1 = Z1^T N t1 = Z1^T z
#pragma unroll
  for (int jt = 0; jt < 3; ++jt)
#pragma unroll
    for (int kt = 0; kt < 3; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) pb[jt] += Z1[kt][jt][q] * zv[kt][q];
  // ---- the coarse direction and s_new = s - direction (split_backward_correction.hxx:30-58), in front of the output: its loads and stores
  //      do not queue behind 48 kB of stores ----
  {
    double* dir = &sm[S::DIR];                 // padded: [dlmd dgmm | dxi or dmu (NF)] | [dw (NU) | dq dv]
    reduceRows(pt, dir);
    reduceRows(pb, dir + 48);
    const double* dw = dir + 48;
    const double* __restrict__ s = B.sol + rec * L::SOL;
    double* __restrict__ sn = B.snew + rec * L::SNEW;
    // s_new = s - direction (split_backward_correction.hxx:49-63, impulse_split_backward_correction.hxx:43-55)
    if (lane < NV) {
      sn[L::N_LMD + lane] = s[L::S_LMD + lane] - dir[lane];
      sn[L::N_GMM + lane] = s[L::S_GMM + lane] - dir[NV + lane];
      sn[L::N_V + lane] = s[L::S_V + lane] - dw[NU + NV + lane];
      if (lane >= 6) sn[L::N_Q + lane + 1] = s[L::S_Q + lane + 1] - dw[NU + lane];
    }
    if (!impulse) {
      if (lane >= 32 && lane < 32 + NU) sn[L::N_U + lane - 32] = s[L::S_U + lane - 32] - dw[lane - 32];
      if (lane >= 48 && lane < 48 + ni) sn[L::N_XI + lane - 48] = s[L::S_XI + lane - 48] - dir[NX + lane - 48];
    } else if (lane >= 32 && lane < 32 + NC && nd->active[lane - 32]) {
      const int c = lane - 32, row = nd->row_of[c];
      for (int k = 0; k < 3; ++k) {
        sn[L::N_U + row + k] = s[L::S_F + 3 * c + k] - dw[row + k];             // f, packed rows
        sn[L::N_XI + row + k] = s[L::S_MU + 3 * c + k] - dir[NX + row + k];     // mu, packed rows
      }
    }
    if (lane == 63) {
      double qn[7];
      lieIntegrateBase(s + L::S_Q, dw + NU, -1.0, qn);
      for (int k = 0; k < 7; ++k) sn[L::N_Q + k] = qn[k];
    }
  }
  // Q^-1: the six raw tiles come back from the scratch while C0 is put together
  wtile Qi[6];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // (stored by these very lanes, never in this CU's L1 before: the stores only have to be complete)
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const v2d lo = reinterpret_cast<const v2d*>(scr + 256 * t)[2 * lane], hi = reinterpret_cast<const v2d*>(scr + 256 * t)[2 * lane + 1];
    Qi[t] = wtile{lo.x, lo.y, hi.x, hi.y};
  }
  KSTAMP(7);

  // ---- C0 = [TL; TR^T], TL = -S^-1 = -N^T N, TR = S^-1 F Q^-1 = N^T Z1, one block of columns a (rows a of S) at a time: kinv entry
  //      (rho, kappa) = TL(kappa, rho) for rho < 36, TR(kappa, rho - 36) below -- the tiles (a, b) as they are ----
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int ncol = a < 2 ? 16 : NX - 32;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      wtile Z = wzero();
#pragma unroll
      for (int kt = (a > bb ? a : bb); kt < 3; ++kt) {
        pSub<4>(Z, Ns[U3(a, kt)], Ns[U3(bb, kt)]);
      }
      stageTile(Z, 16 * bb, nr - 16 * bb, 0, ncol);       // rows 0 .. nr - 1: TL(:, 0 : NX)
      wtile R = wzero();
#pragma unroll
      for (int kt = a; kt < 3; ++kt) {
        pAcc<4>(R, Ns[U3(a, kt)], Z1[kt][bb]);
      }
      stageVarTile(R, bb, 0, ncol);
      __builtin_amdgcn_sched_barrier(0);
    }
    flushBlock(L::I_C0 / NK + 16 * a, ncol);
  }
  KSTAMP(8);
  // ---- C1 = [TR(:, NU:); BR(:, NU:)], BR = Q^-1 - Z1^T Z1, one block of columns a (columns 16 a .. of (u, q, v), of which 12 .. exist) at a
  //      time: kinv entry (rho, kappa) = TR(rho, NU + kappa) for rho < 36 -- the tile (a, b) of TR^T = Z1^T N --, BR(NU + kappa, rho - 36) below ----
  static_assert(L::I_C0 % NK == 0 && L::I_C1G % NK == 0, "column blocks of the record");
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int c0 = a == 0 ? -NU : 0;                         // the block's first column is max(0, 16 a - NU) of C1
    const int first = a == 0 ? 0 : 16 * a - NU, ncol = a == 0 ? 16 - NU : 16;
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      wtile R = wzero();                                     // TR^T(a, b) = sum_kt Z1(kt, a)^T N(kt, b), kt >= b
#pragma unroll
      for (int kt = bb; kt < 3; ++kt) {
        pAcc<4>(R, Z1[kt][a], Ns[U3(bb, kt)]);
      }
      stageTile(R, 16 * bb, nr - 16 * bb, c0, ncol);
      // BR(a, b): Q^-1(a, b) from the scratch (the transposed tile below the diagonal)
      wtile Z = Qi[a <= bb ? U3(a, bb) : U3(bb, a)];
      if (a > bb) Z = transposeTile(Z);
#pragma unroll
      for (int kt = 0; kt < 3; ++kt) {
        pSub<4>(Z, Z1[kt][a], Z1[kt][bb]);
      }
      stageVarTile(Z, bb, c0, ncol);
      __builtin_amdgcn_sched_barrier(0);
    }
    flushBlock(L::I_C1G / NK + first, ncol);
  }
  KSTAMP(10);
  KSTAMP(11);
#undef KSTAMP
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0 && lane == 0 && B.status[b] == 0) B.status[b] = 1000 + pos;
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

template <typename D>
void OcpLaunch<D>::parnmpcEventInverseWave(const OcpBuffers& B, long batch, int n_general, hipStream_t st) {
  if (n_general <= 0) return;
  const size_t smem = KktWaveSmemG<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)parnmpc_kkt_inverse_wave_general_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  hipLaunchKernelGGL((parnmpc_kkt_inverse_wave_general_kernel<D>), dim3((unsigned)batch, (unsigned)n_general), dim3(64), smem, st, B);
}
template void OcpLaunch<LeggedDims<4, 3>>::parnmpcEventInverseWave(const OcpBuffers&, long, int, hipStream_t);



}  // namespace idocp_dev
