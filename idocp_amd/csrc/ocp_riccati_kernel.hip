// S3 / S4 -- backward and forward Riccati sweeps of the contact path.
//
// Replaces RiccatiRecursionSolver::backwardRiccatiRecursion / computeInitialStateDirection /
// forwardRiccatiRecursion for a horizon without discrete events (src/ocp/riccati_recursion_solver.cpp:48-162)
// and the per-stage factorizers they call:
//   BackwardRiccatiRecursionFactorizer::factorizeKKTMatrix / factorizeRiccatiFactorization
//                              (include/idocp/ocp/backward_riccati_recursion_factorizer.hxx:44-161)
//   SplitRiccatiFactorizer::backwardRiccatiRecursion / forwardRiccatiRecursion
//                              (include/idocp/ocp/split_riccati_factorizer.hxx:36-52, 103-128)
// The dynamics are x+ = A x + B u + Fx with A = [Fqq Fqv; Fvq Fvv], B = [0; Fvu]; only the leading
// 6x6 blocks of Fqq / Fqv differ from I / dt I (floating base) and are the only ones stored.
//
// S3: one 256-thread workgroup per OCP instance walks the horizon backwards with
// P_{i+1} and the stage's LQR blocks resident in LDS (~52 kB -> three instances per CU).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <type_traits>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

typedef double rd2 __attribute__((ext_vector_type(2)));      // 16-byte pair

template <typename D>
struct RiccatiSmem {
  static constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF;
  static constexpr int LDW = NX + 2;                                         // 38: rows of W start 12 banks apart, 16 lanes hit 16 bank groups
  static constexpr int PQQ = 0, PQV = PQQ + NV * NV, PVV = PQV + NV * NV, SQ = PVV + NV * NV, SV = SQ + NV,
                       STAGE = SV + NV + 4,                                  // copy of the kkt record without Qxx
                       STAGE_LEN = OcpLayout<D>::KKT - OcpLayout<D>::K_QXU,
                       // W = [A^T P; B^T P] (NX + NU rows, NX columns) stored row by row: W(i, c) at WT + c + LDW i
                       WT = STAGE + STAGE_LEN, GK = WT + LDW * NX,           // Quu K reuses the rows of B^T P
                       KM = WT + LDW * (NX + NU), KV = KM + NU * NX,
                       GW = KV + 16, SQN = GW + NU * NU, SVN = SQN + NV, INVD = SVN + NV + 4, TOTAL = INVD + NU + 4;
  static_assert(NU * NX <= NU * LDW, "GK aliases B^T P");
  // extra blocks of the HYBRID instantiation (stages that carry a switching constraint: Schur-complement step of
  // SplitRiccatiFactorizer::backwardRiccatiRecursion, split_riccati_factorizer.hxx:43-101).  They alias the rows of A^T P, dead once
  // F, H, G and the k-independent part of the s recursion are done, so the HYBRID instantiation needs no extra LDS; M aliases
  // B^T P / GK (dead once DtM is formed, before GK = Quu K is written).  Phix, Phiu, P are read from the swc record (L2) where needed.
  static constexpr int DG = WT, SS = DG + NF * NU, DTM = SS + NF * NF, SDG = DTM + NU * NX, MV = SDG + NF * NU, SCORR = MV + NF,
                       MMX = GK;
  static_assert(SCORR + NX <= GK && NF * NX <= NU * LDW, "hybrid aliases");
  static_assert(TOTAL * 8 + 64 <= 40960, "four instances per CU");
};

// tile (ib, jb) number t of the upper block triangle of a 3 x 3 tiling: (0,0) (0,1) (0,2) (1,1) (1,2) (2,2)
__device__ __forceinline__ void upperTile3(int t, int& ib, int& jb) {
  if (t < 3) { ib = 0; jb = t; } else if (t < 5) { ib = 1; jb = t - 2; } else { ib = 2; jb = 2; }
}

// ---- the four tiled products of a stage.  A wavefront owns the tiles [TB, TE) of a phase (compile-time, so that every operand
// set and accumulator is a named register): it reads the operand sets its tiles share ONCE, runs the k-steps of all its tiles
// interleaved (independent accumulators: the matrix core issues back to back) and finishes with the epilogues. ----
constexpr int tileBegin(int ntiles, int w, int nw) { return (ntiles * w + nw - 1) / nw; }
constexpr bool tilesUseMod3(int tb, int te, int v) { for (int t = tb; t < te; ++t) if (t % 3 == v) return true; return false; }
constexpr bool tilesUseDiv3(int tb, int te, int v) { for (int t = tb; t < te; ++t) if (t / 3 == v) return true; return false; }
constexpr int upperIb(int t) { return t < 3 ? 0 : (t < 5 ? 1 : 2); }
constexpr int upperJb(int t) { return t < 3 ? t : (t < 5 ? t - 2 : 2); }
constexpr bool tilesUseUpperIb(int tb, int te, int v) { for (int t = tb; t < te; ++t) if (upperIb(t) == v) return true; return false; }
constexpr bool tilesUseUpperJb(int tb, int te, int v) { for (int t = tb; t < te; ++t) if (upperJb(t) == v) return true; return false; }

template <int K>
__device__ __forceinline__ void mfmaLoadOp(double (&o)[(K + 3) / 4], const double* p, int stride, int lane) {
  const int g = lane >> 4;
#pragma unroll
  for (int s = 0; s < (K + 3) / 4; ++s) {
    const int k = 4 * s + g;
    if (4 * s + 3 < K) o[s] = p[stride * k];
    else { const double v = p[stride * (k < K ? k : K - 1)]; o[s] = k < K ? v : 0.0; }
  }
}

// phase 1: W = [A^T P; B^T P], tile t = 3 ib + cb covers rows 16 ib .. of W and columns 16 cb .. of P
// (mid() runs between the k-steps and the epilogues: the place where the operand registers are free again -- the caller issues
// the global loads of phase 2 there)
// WIDE (one wavefront per instance, 512 registers): the part of an epilogue that does not depend on the accumulators is computed
// in the same scheduling region as the k-steps, i.e. while the matrix core works; otherwise one epilogue at a time behind them.
template <typename D, bool WIDE, int TB, int TE, typename Mid>
__device__ __forceinline__ void riccatiPhase1(const double* Pqq, const double* Pqv, const double* Pvv, const double* Fall, const double* Fqq6,
                                              const double* Fqv6, double dt, double* Wt, int lane, Mid mid) {
  constexpr int NV = D::NV, NX = D::NX, KS = (NV + 3) / 4, LDW = RiccatiSmem<D>::LDW, NTL = TE - TB;
  if constexpr (NTL > 0) {
    const int li = lane & 15, g = lane >> 4;
    double xo[3][KS], yo[3][KS];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      if (tilesUseMod3(TB, TE, v)) {
        const int c = 16 * v + li, cc = c < NX ? c : NX - 1;
        mfmaLoadOp<NV>(xo[v], cc < NV ? Pqv + cc : Pvv + NV * (cc - NV), cc < NV ? NV : 1, lane);      // P(v, c): column c of Pvq = row c of Pqv
      }
      if (tilesUseDiv3(TB, TE, v)) mfmaLoadOp<NV>(yo[v], Fall + NV * (16 * v + li), 1, lane);
    }
    // the structured part of tile j, register q: rows outside the 6 x 6 base blocks of Fqq / Fqv are one term (P(q,:) or dt P(q,:)),
    // the others a 6-term dot; which (tile, register) pairs meet a base block is known at compile time
    auto structured = [&](int j, int q) -> double {
      const int cb = (TB + j) % 3, ib = (TB + j) / 3;
      const int c = 16 * cb + li, cc = c < NX ? c : NX - 1;
      const double* pq = cc < NV ? Pqq + NV * cc : Pqv + NV * (cc - NV);          // P(q, c)
      const int lo = 16 * ib + 4 * q, r = lo + g;                                // rows lo .. lo + 3 of W over the four lane groups
      if (lo >= NX) return 0.0;
      const int lo6 = lo < NV ? lo : lo - NV;                                   // (a group of four never straddles NV = 18: 16 | 20)
      const bool isv = r >= NV;
      const int rr6 = isv ? r - NV : r;
      const bool any6 = lo6 < 6 || (lo < NV && lo + 3 >= NV), all6 = lo6 + 3 < 6 && !(lo < NV && lo + 3 >= NV);
      double dot6 = 0.0;
      if (any6) {
        const double* F6 = (isv ? Fqv6 : Fqq6) + 6 * (rr6 < 6 ? rr6 : 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) dot6 += F6[m] * pq[m];
      }
      if (all6) return dot6;
      const double direct = (isv ? dt : 1.0) * pq[rr6 < NV ? rr6 : 0];
      return (any6 && rr6 < 6) ? dot6 : direct;
    };
    constexpr bool EARLY = false;      // (measured with WIDE: nine tiles' worth of early terms push the factorisation's registers into AGPRs)
    double extra[EARLY ? NTL : 1][4];
    if constexpr (EARLY) {
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) extra[j][q] = structured(j, q);
    }
    mfma_d4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(yo[(TB + j) / 3][s], xo[(TB + j) % 3][s], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    mid();
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      __builtin_amdgcn_sched_barrier(0);      // one epilogue at a time: hoisted together their LDS reads overflow the registers
      const int cb = (TB + j) % 3, ib = (TB + j) / 3;
      const int c = 16 * cb + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ib + 4 * q + g;
        double sx;
        if constexpr (EARLY) sx = extra[j][q]; else sx = structured(j, q);
        if (c < NX) Wt[c + LDW * r] = acc[j][q] + sx;
      }
    }
  } else {
    mid();
  }
}

// phase 2: [F H; . G] on the upper block triangle, tile t -> (upperIb, upperJb); qxx[j][q]: Qxx entry of local tile j, register q
template <typename D, bool WIDE, int TB, int TE, int T2W>
__device__ __forceinline__ void riccatiPhase2(const double* Wt, const double* Fall, const double* Fqq6, const double* Fqv6, double dt,
                                              const double (&qxx)[T2W][4], double* Pqq, double* Pqv, double* Pvv, double* Qxu, double* Quu,
                                              int lane) {
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, KS = (NV + 3) / 4, LDW = RiccatiSmem<D>::LDW, NTL = TE - TB;
  static_assert(NTL <= T2W, "Qxx registers");
  if constexpr (NTL > 0) {
    const int li = lane & 15, g = lane >> 4;
    double xo[3][KS], yo[3][KS];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      if (tilesUseUpperIb(TB, TE, v)) mfmaLoadOp<NV>(xo[v], Wt + LDW * (16 * v + li) + NV, 1, lane);
      if (tilesUseUpperJb(TB, TE, v)) mfmaLoadOp<NV>(yo[v], Fall + NV * (16 * v + li), 1, lane);
    }
    // the structured part of tile j, register q (see riccatiPhase1): W(:, q) [Fqq Fqv 0]
    auto structured = [&](int j, int q) -> double {
      const int ib = upperIb(TB + j), jb = upperJb(TB + j);
      const double* wr = Wt + LDW * (16 * ib + li);
      const int lo = 16 * jb + 4 * q, c = lo + g;                               // columns lo .. lo + 3 over the four lane groups
      if (lo >= NX) return 0.0;
      const bool isv = c >= NV;
      const int c6 = isv ? c - NV : c, lo6 = lo < NV ? lo : lo - NV;
      const bool any6 = lo6 < 6 || (lo < NV && lo + 3 >= NV), all6 = lo6 + 3 < 6 && !(lo < NV && lo + 3 >= NV);
      double dot6 = 0.0;
      if (any6) {
        const double* F6 = (isv ? Fqv6 : Fqq6) + 6 * (c6 < 6 ? c6 : 0);
#pragma unroll
        for (int m = 0; m < 6; ++m) dot6 += wr[m] * F6[m];
      }
      if (all6) return dot6;
      const double direct = (isv ? dt : 1.0) * wr[c6 < NV ? c6 : 0];
      return (any6 && c6 < 6) ? dot6 : direct;
    };
    double extra[WIDE ? NTL : 1][4];
    if constexpr (WIDE) {
#pragma unroll
      for (int j = 0; j < NTL; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) extra[j][q] = structured(j, q) + qxx[j][q];
    }
    mfma_d4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(yo[upperJb(TB + j)][s], xo[upperIb(TB + j)][s], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      if constexpr (!WIDE) __builtin_amdgcn_sched_barrier(0);
      const int ib = upperIb(TB + j), jb = upperJb(TB + j);
      const int r = 16 * ib + li;                                             // row of [F H; . G]
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int lo = 16 * jb + 4 * q, c = lo + g;
        double val;
        if constexpr (WIDE) val = acc[j][q] + extra[j][q]; else val = acc[j][q] + qxx[j][q] + structured(j, q);
        // Only the entries on and above the diagonal are used and mirrored (also inside the diagonal tiles): P stays EXACTLY
        // symmetric.  An antisymmetric rounding residue would not be damped by the feedback term and grows with the open-loop
        // dynamics from stage to stage (measured: 2.9 x per stage) -- the reason for the reference's (P + P^T) / 2.
        // Pqq, Pqv, Pvv are consecutive NV x NV blocks: entry (r, c), r <= c, lives in block [r >= NV] + [c >= NV]; its mirror image
        // exists inside the two diagonal blocks (for Pqv the "mirror" address is the entry itself).
        if (lo + 3 >= NX) {
          if (c >= NX) { if (r < NX) Qxu[r + NX * (c - NX)] += val; else Quu[(r - NX) + NU * (c - NX)] += val; }
        }
        if (lo < NX && r <= c && c < NX) {
          const int rb = r >= NV, cb = c >= NV, rr = r - NV * rb, cc = c - NV * cb;
          double* blk = Pqq + NV * NV * (rb + cb);
          blk[rr + NV * cc] = val;
          blk[(rb == cb) ? cc + NV * rr : rr + NV * cc] = val;
        }
      }
    }
  }
}

// phase 4: GK = Quu K, tile t covers the columns 16 t .. of K
template <typename D, bool WIDE, int TB, int TE>
__device__ __forceinline__ void riccatiPhase4(const double* Quu, const double* KM, double* GK, int lane) {
  constexpr int NX = D::NX, NU = D::NU, KS = (NU + 3) / 4, NTL = TE - TB;
  if constexpr (NTL > 0) {
    const int li = lane & 15, g = lane >> 4;
    double xo[KS], yo[NTL][KS];
    mfmaLoadOp<NU>(xo, Quu + NU * (li < NU ? li : NU - 1), 1, lane);           // Quu symmetric
#pragma unroll
    for (int j = 0; j < NTL; ++j) { const int c = 16 * (TB + j) + li; mfmaLoadOp<NU>(yo[j], KM + NU * (c < NX ? c : NX - 1), 1, lane); }
    mfma_d4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(yo[j][s], xo[s], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 16 * (TB + j) + g + 4 * q;
        if (li < NU && c < NX) GK[li + NU * c] = acc[j][q];
      }
  }
}

// phase 5: P = F - K^T Y on the upper block triangle, Y (NU x NX) = GK.  Where K comes straight out of the solve G K = -H^T (every
// stage without a switching constraint), GK IS -H^T up to the residual of a backward-stable solve (|G K + H^T| ~ eps |G| |K|, not
// amplified by the condition of G), so the product Quu K (phase 4, one more barrier) is skipped and Y = -Qxu^T is read in place:
// entry (j, c) of Y at Y + ycs c + yks j, sign = +1 for GK, -1 for Qxu^T.
template <typename D, bool WIDE, int TB, int TE>
__device__ __forceinline__ void riccatiPhase5(const double* KM, const double* Y, int ycs, int yks, double sign, double* Pqq, double* Pqv,
                                              double* Pvv, int lane) {
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, KS = (NU + 3) / 4, NTL = TE - TB;
  if constexpr (NTL > 0) {
    const int li = lane & 15, g = lane >> 4;
    double xo[3][KS], yo[3][KS];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      const int e = 16 * v + li, ec = e < NX ? e : NX - 1;
      if (tilesUseUpperIb(TB, TE, v)) mfmaLoadOp<NU>(xo[v], KM + NU * ec, 1, lane);
      if (tilesUseUpperJb(TB, TE, v)) mfmaLoadOp<NU>(yo[v], Y + ycs * ec, yks, lane);
    }
    mfma_d4 acc[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j) acc[j] = mfma_d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(yo[upperJb(TB + j)][s], xo[upperIb(TB + j)][s], acc[j], 0, 0, 0);
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      if constexpr (!WIDE) __builtin_amdgcn_sched_barrier(0);
      const int r = 16 * upperIb(TB + j) + li;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 16 * upperJb(TB + j) + g + 4 * q;
        if (r <= c && c < NX) {
          const int rb = r >= NV, cb = c >= NV, rr = r - NV * rb, cc = c - NV * cb;
          double* blk = Pqq + NV * NV * (rb + cb);
          const double val = blk[rr + NV * cc] - sign * acc[j][q];               // the mirror image holds the same value: store, don't update
          blk[rr + NV * cc] = val;
          blk[(rb == cb) ? cc + NV * rr : rr + NV * cc] = val;
        }
      }
    }
  }
}

// the wavefront's share of a phase: FN<D, tile range of wave w>(args)
#define RICCATI_TILES_CASE8(FN, NTILES, W, ...)                                                                                        \
  case W: if constexpr (tileBegin(NTILES, W, 8) < tileBegin(NTILES, W + 1, 8)) FN<D, false, tileBegin(NTILES, W, 8), tileBegin(NTILES, W + 1, 8)>(__VA_ARGS__); break;
#define RICCATI_TILES(FN, NTILES, ...)                                                                                                 \
  do {                                                                                                                                 \
    if constexpr (NW == 1) { FN<D, true, 0, NTILES>(__VA_ARGS__); }                                                                          \
    else if constexpr (NW == 2) {                                                                                                      \
      if (wave == 0) FN<D, false, tileBegin(NTILES, 0, 2), tileBegin(NTILES, 1, 2)>(__VA_ARGS__);                                            \
      else FN<D, false, tileBegin(NTILES, 1, 2), tileBegin(NTILES, 2, 2)>(__VA_ARGS__);                                                      \
    } else if constexpr (NW == 8) {                                                                                                    \
      switch (wave) {                                                                                                                  \
        RICCATI_TILES_CASE8(FN, NTILES, 0, __VA_ARGS__) RICCATI_TILES_CASE8(FN, NTILES, 1, __VA_ARGS__)                                \
        RICCATI_TILES_CASE8(FN, NTILES, 2, __VA_ARGS__) RICCATI_TILES_CASE8(FN, NTILES, 3, __VA_ARGS__)                                \
        RICCATI_TILES_CASE8(FN, NTILES, 4, __VA_ARGS__) RICCATI_TILES_CASE8(FN, NTILES, 5, __VA_ARGS__)                                \
        RICCATI_TILES_CASE8(FN, NTILES, 6, __VA_ARGS__) RICCATI_TILES_CASE8(FN, NTILES, 7, __VA_ARGS__)                                \
        default: break;                                                                                                                \
      }                                                                                                                                \
    } else {                                                                                                                           \
      if (wave == 0) FN<D, false, tileBegin(NTILES, 0, 4), tileBegin(NTILES, 1, 4)>(__VA_ARGS__);                                            \
      else if (wave == 1) FN<D, false, tileBegin(NTILES, 1, 4), tileBegin(NTILES, 2, 4)>(__VA_ARGS__);                                       \
      else if (wave == 2) FN<D, false, tileBegin(NTILES, 2, 4), tileBegin(NTILES, 3, 4)>(__VA_ARGS__);                                       \
      else FN<D, false, tileBegin(NTILES, 3, 4), tileBegin(NTILES, 4, 4)>(__VA_ARGS__);                                                      \
    }                                                                                                                                  \
  } while (0)

// Barrier of the workgroup.  With ONE wavefront (the batch-wide launch) the exchange through LDS needs no barrier: the wavefront's DS
// instructions execute in program order, and all the compiler has to be told is not to move LDS accesses across this point.
// __syncthreads() would also wait for every global access in flight (s_waitcnt vmcnt(0)): the prefetch of the next stage's record and
// the stores of this stage's would then complete at the first barrier after they were issued instead of under the stage's arithmetic.
template <int NW> __device__ __forceinline__ void blockSync() {
  if constexpr (NW == 1) { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }
  else blockLdsSync();      // (no s_waitcnt vmcnt(0): nothing passes between wavefronts through global memory)
}

template <typename D, int NT, bool HYBRID>
__global__ __launch_bounds__(NT, (NT == 64 || NT == 512) ? 1 : 2) void ocp_riccati_backward_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = RiccatiSmem<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NN = NV * NV, NF = D::NF, LDW = S::LDW, NW = NT / 64;
  static_assert(NX + NU == 48 && NX <= 48, "3 x 3 tiles of 16");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const OcpNode* __restrict__ nodes = B.nodes;
  const int tid = threadIdx.x;
  constexpr int nt = NT;
  const int wave = tid >> 6, lane = tid & 63, li = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  const long base = b * B.NS;                    // first record of this instance
  double* Pqq = &sm[S::PQQ];
  double* Pqv = &sm[S::PQV];
  double* Pvv = &sm[S::PVV];
  double* Wt = &sm[S::WT];
  double* st = &sm[S::STAGE];     // st[k - K_QXU] holds kkt[k] for k >= K_QXU
  constexpr int KO = L::K_QXU, SL = S::STAGE_LEN;
  if (tid == 0) s_ok = 1;
  // P, s stored in single precision (opt-in, OcpProblem::ric_fp32): every value of the cost-to-go is rounded to FP32 when it is
  // written (the arithmetic of a stage stays FP64), which is exactly what an FP32 ric record / FP32 copy in LDS would hold
  const bool p32 = P->ric_fp32 != 0;
  auto st32 = [&](double x) { return p32 ? (double)(float)x : x; };
  // terminal stage (riccati_recursion_solver.cpp:53-56)
  {
    const double* __restrict__ kk = B.kkt + (base + nodes[M - 1].slot) * L::KKT;
    double* __restrict__ rr = B.ric + (base + nodes[M - 1].slot) * L::RIC;
    for (int e = tid; e < NN; e += nt) {
      const int c = e / NV, r = e - c * NV;
      const double pqq = st32(kk[L::K_QXX + L::xsym(r, c)]), pvv = st32(kk[L::K_QXX + L::xsym(NV + r, NV + c)]);
      Pqq[e] = pqq; Pqv[e] = 0.0; Pvv[e] = pvv;
      rr[L::R_PQV + e] = 0.0;
      if (r <= c) { rr[L::R_PQQ + L::psym(r, c)] = pqq; rr[L::R_PVV + L::psym(r, c)] = pvv; }
    }
    if (tid < NV) {
      const double sq = st32(-kk[L::K_LX + tid]), sv = st32(-kk[L::K_LX + NV + tid]);
      sm[S::SQ + tid] = sq; sm[S::SV + tid] = sv;
      rr[L::R_SQ + tid] = sq; rr[L::R_SV + tid] = sv;
    }
  }
  {
    const double* __restrict__ kk = B.kkt + (base + nodes[M - 2].slot) * L::KKT;
    for (int e = tid; e < SL; e += nt) st[e] = kk[KO + e];
  }
  blockSync<NW>();
  // walk the chain backwards (riccati_recursion_solver.cpp:48-107): impulse stages are ordinary steps with
  // Fqv = 0 (dtq = 0), Fvu = 0, Qxu = 0, Quu = I written by K5b, which makes K = 0, k = 0 and P = F
  for (int i = M - 2; i >= 0; --i) {
    const OcpNode* __restrict__ nd = nodes + i;
    const double dt = nd->dtq;
    const long rec = base + nd->slot;
    const int dimi = HYBRID ? nd->sw_dimi : 0;
    const bool stamp = tid == 0 && b == (gridDim.x > 7 ? 7 : 0) && i == M / 2 && B.prof != nullptr;
#define RSTAMP(k) do { if (stamp) B.prof[16 + k] = wall_clock64(); } while (0)
    RSTAMP(0);
    // The matrix products of a stage run as 16 x 16 tiles on the matrix cores (riccatiPhase1 .. 5 above), a tile per
    // wavefront and round, with the structured part of the dynamics (A = [Fqq Fqv; Fvq Fvv], Fqq = diag(Fqq6, I), Fqv = diag(Fqv6, dt I))
    // added in the epilogue of the tile:
    //   phase 1 (9 tiles):  W = [A^T P; B^T P] = [Fvq Fvv Fvu]^T P(v, :)  +  [Fqq Fqv 0]^T P(q, :)
    //   phase 2 (6 tiles):  [F H; . G] = [Qxx Qxu; . Quu] + W(:, v) [Fvq Fvv Fvu]  +  W(:, q) [Fqq Fqv 0], upper block triangle, mirrored
    //   phase 4 (3 tiles):  GK = Quu K            phase 5 (6 tiles):  P = F - K^T GK, upper block triangle, mirrored
    // P and F are symmetric; computing the tiles above the diagonal and mirroring them replaces the reference's P = (P + P^T) / 2
    // (backward_riccati_recursion_factorizer.hxx:133-135).
    constexpr int PF = (SL / 2 + NT - 1) / NT, T1 = 9, T2 = 6, T2W = (T2 + NW - 1) / NW;      // the next record travels as 16-byte pairs
    static_assert(SL % 2 == 0 && KO % 2 == 0 && S::STAGE % 2 == 0 && L::KKT % 2 == 0, "16-byte loads of the stage record");
    rd2 pre[PF];
    double qxx[T2W][4];
    auto loadQxx = [&]() {
      // Qxx of THIS stage is consumed once per element in phase 2: straight to the registers of the lane that adds it
      const double* __restrict__ kc = B.kkt + rec * L::KKT;
      const int t2b = tileBegin(T2, wave, NW), t2e = tileBegin(T2, wave + 1, NW);      // this wavefront's tiles of phase 2
#pragma unroll
      for (int jj = 0; jj < T2W; ++jj) {
        const int t = t2b + jj;
        int ib, jb;
        upperTile3(t < t2e ? t : 0, ib, jb);
        const int r = 16 * ib + li;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = 16 * jb + g + 4 * q;
          qxx[jj][q] = (t < t2e && r <= c && c < NX) ? kc[L::K_QXX + L::xsym(r, c)] : 0.0;
        }
      }
    };
    double* Qxu = st + (L::K_QXU - KO);
    double* Quu = st + (L::K_QUU - KO);
    const double* Fqq6 = st + (L::K_FQQ - KO);
    const double* Fqv6 = st + (L::K_FQV - KO);
    const double* Fall = st + (L::K_FVQ - KO);       // [Fvq Fvv Fvu]: NV rows, NX + NU columns
    static_assert(L::K_FVV == L::K_FVQ + NV * NV && L::K_FVU == L::K_FVV + NV * NV, "[Fvq Fvv Fvu] contiguous");
    const double* Fvq = st + (L::K_FVQ - KO);
    const double* Fvv = st + (L::K_FVV - KO);
    const double* Fvu = st + (L::K_FVU - KO);
    const double* lx = st + (L::K_LX - KO);
    double* lu = st + (L::K_LU - KO);
    const double* Fx = st + (L::K_FX - KO);
    RSTAMP(1);
    // ---- phase 1: A^T P and B^T P (backward_riccati_recursion_factorizer.hxx:48-78) ----
    RICCATI_TILES(riccatiPhase1, T1, Pqq, Pqv, Pvv, Fall, Fqq6, Fqv6, dt, Wt, lane, loadQxx);
    blockSync<NW>();
    RSTAMP(2);
    // ---- phase 2: F, H, G (:79-113); F overwrites P_{i+1}, which is dead from here ----
    RICCATI_TILES(riccatiPhase2, T2, Wt, Fall, Fqq6, Fqv6, dt, qxx, Pqq, Pqv, Pvv, Qxu, Quu, lane);
    // The two vector terms run behind the tile jobs of different wavefronts (NT >= 128): lu on the last NU threads,
    // the k-independent part of the s recursion (:141-160) on the first 2 NV.  Neither reads anything the tile jobs write
    // (P_{i+1} is only read through W here).
    RSTAMP(11);
    if (tid >= NT - 64 && lane < 4 * NU) {
      // lu += (B^T P) Fx - Fvu^T sv: four lanes per row (10 | 8 | 10 | 8 and 5 | 5 | 4 | 4 terms), 16-byte LDS reads
      const int j = lane >> 2, part = lane & 3;
      const int c0 = (part >> 1) * (NX / 2) + (part & 1) * 10, nc2 = (part & 1) ? 4 : 5;      // first column, pairs of columns
      static_assert(NX == 36 && LDW % 2 == 0, "column split of the lu rows");
      const rd2* wj = reinterpret_cast<const rd2*>(Wt + LDW * (NX + j) + c0);                 // row j of B^T P
      const rd2* fx = reinterpret_cast<const rd2*>(Fx + c0);
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 5; ++c) if (c < nc2) { const rd2 w = wj[c], f = fx[c]; acc += w.x * f.x + w.y * f.y; }
#pragma unroll
      for (int t = 0; t < (NV + 3) / 4; ++t) { const int m = part + 4 * t; if (m < NV) acc -= Fvu[m + NV * j] * sm[S::SV + m]; }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (part == 0) lu[j] += acc;
    }
    // k-independent part of the s recursion (:141-160): one lane per row of x; it needs W and the old s only, so with more than one
    // wavefront it runs on the second one BEHIND the barrier, next to the factorisation on the first
    auto sPart1 = [&](int row) {
      const bool isv = row >= NV;
      const int r = isv ? row - NV : row;
      const double* F6 = isv ? Fqv6 : Fqq6;
      const rd2* Fm = reinterpret_cast<const rd2*>((isv ? Fvv : Fvq) + NV * r);
      const rd2* wr = reinterpret_cast<const rd2*>(Wt + LDW * row);                            // row of A^T P
      const rd2* svp = reinterpret_cast<const rd2*>(&sm[S::SV]);
      const rd2* fx = reinterpret_cast<const rd2*>(Fx);
      static_assert(NV % 2 == 0 && S::SV % 2 == 0 && (L::K_FX - KO) % 2 == 0 && (L::K_FVQ - KO) % 2 == 0, "16-byte reads of the vector terms");
      double acc;
      if (r < 6) {
        acc = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += F6[m + 6 * r] * sm[S::SQ + m];
      } else {
        acc = isv ? dt * sm[S::SQ + r] : sm[S::SQ + r];
      }
      double a2 = 0.0, a3 = 0.0;
#pragma unroll
      for (int m = 0; m < NV / 2; ++m) { const rd2 f = Fm[m], sv2 = svp[m]; acc += f.x * sv2.x + f.y * sv2.y; }
#pragma unroll
      for (int c = 0; c < NX / 2; c += 2) {
        const rd2 w0 = wr[c], f0 = fx[c], w1 = wr[c + 1], f1 = fx[c + 1];
        a2 += w0.x * f0.x + w0.y * f0.y;
        a3 += w1.x * f1.x + w1.y * f1.y;
      }
      sm[(isv ? S::SVN : S::SQN) + r] = acc - (a2 + a3) - lx[row];
    };
    if (NW == 1 && tid < NX) sPart1(tid);
    RSTAMP(12);
    blockSync<NW>();
    RSTAMP(3);
    // software pipeline: the record of stage i was staged into LDS at the end of the previous iteration; the global loads of
    // stage i - 1 are issued here (factorisation, phases 4 and 5 ahead of them) and parked in registers
    if (i > 0) {
      const rd2* __restrict__ kn = reinterpret_cast<const rd2*>(B.kkt + (base + nodes[i - 1].slot) * L::KKT + KO);
#pragma unroll
      for (int t = 0; t < PF; ++t) { const int e = tid + NT * t; pre[t] = kn[e < SL / 2 ? e : SL / 2 - 1]; }
    }
    RSTAMP(9);
    // Quu = L L^T and the solves K = -Quu^-1 Qxu^T, k = -Quu^-1 lu in the registers of one wavefront, one right-hand side per lane
    // (Eigen::LLT compute + solve, split_riccati_factorizer.hxx:43-46).  Round 1 multiplied with an explicit Gauss-Jordan
    // inverse: on the stage behind a switching constraint G = Quu + B^T P B has a condition number of 1e8 and P = F - K^T G K came
    // out 6e-9 off (5e-12 with the solves; long double referee, tests/test_hybrid_gpu.py).
    if (NW > 1 && tid >= 64 && tid < 64 + NX) sPart1(tid - 64);
    const bool constrained = HYBRID && dimi > 0;
    if (tid < 64) {
      double x[NU];
      if (!constrained) {
#pragma unroll
        for (int m = 0; m < NU; ++m) x[m] = (tid < NX) ? Qxu[tid + NX * m] : lu[m];
      } else {
#pragma unroll
        for (int m = 0; m < NU; ++m) x[m] = (m == tid) ? 1.0 : 0.0;      // Ginv = llt.solve(I) (:60): the Schur-complement step works with the explicit inverse, like the reference
      }
      choleskySolveRows<NU>(Quu, NU, tid, &s_ok, x);
      if (!constrained) {
        if (tid <= NX) {
#pragma unroll
          for (int m = 0; m < NU; ++m) { if (tid < NX) sm[S::KM + m + NU * tid] = -x[m]; else sm[S::KV + m] = -x[m]; }
        }
      } else if (tid < NU) {
#pragma unroll
        for (int m = 0; m < NU; ++m) sm[S::GW + m + NU * tid] = x[m];
      }
    }
    blockSync<NW>();
    RSTAMP(10);
    if (HYBRID && dimi > 0) {
      // ---- Schur complement w.r.t. the switching constraint Phix dx + Phiu du + P = 0 (split_riccati_factorizer.hxx:56-70) ----
      const double* __restrict__ W = B.swc + rec * L::SWC;
      const double* __restrict__ Phiu = W + L::W_PHIU;
      for (int e = tid; e < dimi * NU; e += nt) {                // DGinv = Phiu Ginv
        const int c = e / dimi, j = e - c * dimi;
        double acc = 0.0;
        for (int m = 0; m < NU; ++m) acc += Phiu[j + NF * m] * sm[S::GW + m + NU * c];
        sm[S::DG + j + NF * c] = acc;
      }
      blockSync<NW>();
      for (int e = tid; e < dimi * dimi; e += nt) {              // S = DGinv Phiu^T
        const int c = e / dimi, j = e - c * dimi;
        double acc = 0.0;
        for (int m = 0; m < NU; ++m) acc += sm[S::DG + j + NF * m] * Phiu[c + NF * m];
        sm[S::SS + j + NF * c] = acc;
      }
      blockSync<NW>();
      // S = L L^T and S^-1 [DGinv, Phix, P] by triangular solves like the reference's llt_s_.solve (:64-66, 71-74): Cholesky and the
      // 49 right-hand sides in the registers of one wavefront (lane = column).  Round 1 formed S^-1 by unpivoted Gauss-Jordan and
      // multiplied: on a stage a few milliseconds in front of a touch-down S is ill-conditioned and the explicit inverse added its
      // share to a direction 1e-6 off (now 1e-10, as far as the FP64 oracle itself is from the long double build, tests/test_hybrid_gpu.py).
      if (tid < 64) {
        static_assert(NU + NX + 1 <= 64, "one lane per right-hand side");
        const double* __restrict__ Wc = B.swc + rec * L::SWC;
        double x[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          double val = 0.0;
          if (j < dimi) val = tid < NU ? sm[S::DG + j + NF * tid] : (tid < NU + NX ? Wc[L::W_PHIX + j + NF * (tid - NU)] : (tid == NU + NX ? Wc[L::W_P + j] : 0.0));
          x[j] = val;
        }
        choleskySolveRows<NF>(&sm[S::SS], NF, tid, &s_ok, x, dimi);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          if (j < dimi) {
            if (tid < NU) sm[S::SDG + j + NF * tid] = x[j];
            else if (tid < NU + NX) sm[S::MMX + j + NF * (tid - NU)] = x[j];      // S^-1 Phix; - SinvDGinv Qxu^T follows below
            else if (tid == NU + NX) sm[S::MV + j] = x[j];
          }
        }
      }
      blockSync<NW>();
      for (int e = tid; e < NU * NU; e += nt) {                  // Ginv -= SinvDGinv^T DGinv
        const int c = e / NU, r = e - c * NU;
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * r] * sm[S::DG + l + NF * c];
        sm[S::GW + e] -= acc;
      }
      blockSync<NW>();
    }
    if (constrained) {
      // K = -Ginv Qxu^T - SinvDGinv^T Phix, k = -Ginv lu - SinvDGinv^T P with the updated Ginv (:67-70)
      const double* __restrict__ Phix = B.swc + rec * L::SWC + L::W_PHIX;
      const double* __restrict__ Pv = B.swc + rec * L::SWC + L::W_P;
      for (int e = tid; e < NU * NX; e += nt) {
        const int c = e / NU, j = e - c * NU;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * Qxu[c + NX * m];
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * j] * Phix[l + NF * c];
        sm[S::KM + e] = -acc;
      }
      if (tid >= NT - NU) {
        const int j = tid - (NT - NU);
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * lu[m];
        for (int l = 0; l < dimi; ++l) acc += sm[S::SDG + l + NF * j] * Pv[l];
        sm[S::KV + j] = -acc;
      }
    }
    if (HYBRID && dimi > 0) {
      // multiplier policy dxi = M dx + m (:71-74): M = S^-1 Phix - SinvDGinv Qxu^T, m = S^-1 P - SinvDGinv lu (S^-1 Phix, S^-1 P: see above)
      double* __restrict__ W = B.swc + rec * L::SWC;
      for (int e = tid; e < dimi * NX; e += nt) {
        const int c = e / dimi, l = e - c * dimi;
        double acc = sm[S::MMX + l + NF * c];
        for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * Qxu[c + NX * m];
        sm[S::MMX + l + NF * c] = acc;
        W[L::W_M + l + NF * c] = acc;
      }
      if (tid < dimi) {
        const int l = tid;
        double acc = sm[S::MV + l];
        for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * lu[m];
        sm[S::MV + l] = acc;
        W[L::W_m + l] = acc;
      }
    }
    blockSync<NW>();
    if (HYBRID && dimi > 0) {
      const double* __restrict__ W = B.swc + rec * L::SWC;
      for (int e = tid; e < NU * NX; e += nt) {                  // DtM = Phiu^T M (:88)
        const int c = e / NU, m = e - c * NU;
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += W[L::W_PHIU + l + NF * m] * sm[S::MMX + l + NF * c];
        sm[S::DTM + m + NU * c] = acc;
      }
      if (tid < NX) {                                            // Phix^T m (:98-99), consumed at the write-back
        double acc = 0.0;
        for (int l = 0; l < dimi; ++l) acc += W[L::W_PHIX + l + NF * tid] * sm[S::MV + l];
        sm[S::SCORR + tid] = acc;
      }
      blockSync<NW>();                                           // M (aliasing GK) is dead from here
    }
    RSTAMP(4);
    // ---- phase 4: GK = Quu K (backward_riccati_recursion_factorizer.hxx:128) -- only where K is not the plain solution of
    //      G K = -H^T (the stages with a switching constraint); see riccatiPhase5 ----
    if (constrained) {
      RICCATI_TILES(riccatiPhase4, 3, Quu, &sm[S::KM], &sm[S::GK], lane);
      blockSync<NW>();
    }
    RSTAMP(5);
    // s recursion, part 2: - Qxu k (next to the tiles below: it reads Qxu and k, they read K and Qxu / GK)
    if (tid >= NT - NV) {
      const int r = tid - (NT - NV);
      double sq = sm[S::SQN + r], sv = sm[S::SVN + r];
#pragma unroll
      for (int j = 0; j < NU; ++j) { sq -= Qxu[r + NX * j] * sm[S::KV + j]; sv -= Qxu[(NV + r) + NX * j] * sm[S::KV + j]; }
      sm[S::SQN + r] = sq; sm[S::SVN + r] = sv;
    }
    // ---- phase 5: P = F - K^T G K (:122-131) ----
    {
      const double* Y = constrained ? &sm[S::GK] : Qxu;
      const int ycs = constrained ? NU : 1, yks = constrained ? 1 : NX;
      const double sign = constrained ? 1.0 : -1.0;
      RICCATI_TILES(riccatiPhase5, T2, &sm[S::KM], Y, ycs, yks, sign, Pqq, Pqv, Pvv, lane);
    }
    blockSync<NW>();
    RSTAMP(6);
    RSTAMP(7);
    double* __restrict__ rr = B.ric + rec * L::RIC;
    double* __restrict__ gg = B.gain + rec * L::GAIN;
    static_assert(S::PQV % 2 == 0 && L::R_PQV % 2 == 0 && NN % 2 == 0 && L::RIC % 2 == 0, "Pqv is one contiguous copy");
    if (HYBRID && dimi > 0) {
      for (int e = tid; e < NN; e += nt) {
        // P -= K^T D^T M + (K^T D^T M)^T, block by block (split_riccati_factorizer.hxx:88-97)
        const int c = e / NV, r = e - c * NV;
        double aqq = 0.0, aqv = 0.0, avv = 0.0;
        for (int j = 0; j < NU; ++j) {
          const double kqr = sm[S::KM + j + NU * r], kqc = sm[S::KM + j + NU * c], kvr = sm[S::KM + j + NU * (NV + r)], kvc = sm[S::KM + j + NU * (NV + c)];
          const double dqr = sm[S::DTM + j + NU * r], dqc = sm[S::DTM + j + NU * c], dvr = sm[S::DTM + j + NU * (NV + r)], dvc = sm[S::DTM + j + NU * (NV + c)];
          aqq += kqr * dqc + kqc * dqr;
          aqv += kqr * dvc + kvc * dqr;
          avv += kvr * dvc + kvc * dvr;
        }
        const double pqq = st32(Pqq[e] - aqq), pqv = st32(Pqv[e] - aqv), pvv = st32(Pvv[e] - avv);
        Pqq[e] = pqq; Pqv[e] = pqv; Pvv[e] = pvv;
        rr[L::R_PQV + e] = pqv;
        if (r <= c) { rr[L::R_PQQ + L::psym(r, c)] = pqq; rr[L::R_PVV + L::psym(r, c)] = pvv; }
      }
    } else {
      if (p32) { for (int e = tid; e < 3 * NN; e += nt) Pqq[e] = (double)(float)Pqq[e]; blockSync<NW>(); }      // (uniform branch)
      for (int e = tid; e < NN / 2; e += nt) reinterpret_cast<rd2*>(rr + L::R_PQV)[e] = reinterpret_cast<const rd2*>(Pqv)[e];
      for (int e = tid; e < NN; e += nt) {               // the upper triangles of Pqq, Pvv
        const int c = e / NV, r = e - c * NV;
        if (r <= c) { rr[L::R_PQQ + L::psym(r, c)] = Pqq[e]; rr[L::R_PVV + L::psym(r, c)] = Pvv[e]; }
      }
    }
    if (tid < NV) {
      double sq = sm[S::SQN + tid], sv = sm[S::SVN + tid];
      if (HYBRID && dimi > 0) { sq -= sm[S::SCORR + tid]; sv -= sm[S::SCORR + NV + tid]; }     // Phix^T m (:98-99)
      sq = st32(sq); sv = st32(sv);
      sm[S::SQ + tid] = sq; sm[S::SV + tid] = sv;
      rr[L::R_SQ + tid] = sq; rr[L::R_SV + tid] = sv;
    }
    static_assert(S::KM % 2 == 0 && L::G_K == 0 && (NU * NX) % 2 == 0 && L::GAIN % 2 == 0, "K is one contiguous copy");
    for (int e = tid; e < NU * NX / 2; e += nt) reinterpret_cast<rd2*>(gg)[e] = reinterpret_cast<const rd2*>(&sm[S::KM])[e];
    if (tid < NU) gg[L::G_k + tid] = sm[S::KV + tid];
    // nothing above reads the stage copy any more: stage i - 1 moves from the registers into LDS
    if (i > 0) {
#pragma unroll
      for (int t = 0; t < PF; ++t) { const int e = tid + NT * t; if (e < SL / 2) reinterpret_cast<rd2*>(st)[e] = pre[t]; }
    }
    blockSync<NW>();
    RSTAMP(8);
#undef RSTAMP
  }
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1;
}

// ---------------------------------------------------------------------------------------------------------------------------------
// S3, round 4: the backward sweep with every matrix in the REGISTERS of one wavefront.
//
// v_mfma_f64_16x16x4_f64 leaves a 16 x 16 tile T in the accumulator layout "lane (g, li) holds T[4 q + g][li] in register q".  Register
// s of such a tile is, as it stands, the operand of k-step s of another instruction -- as B operand it is T[4 s + g][li] = B[k = g][j = li],
// as A operand it is A[i = li][k = g] = T[4 s + g][li], the transpose -- so for two tiles X, Y in that layout
//     D = X^T Y     (contraction over the ROWS of both)
// costs no data movement at all.  The backward Riccati step is three such products once P is kept symmetric:
//     Wt = P C          = P^T C         C = [A B]: x+ = C z + Fx,  z = [x; u]  (NX + NU = 48 = 3 tiles)
//     Qh = Q + C^T Wt                   [F H; H^T G], Q = [Qxx Qxu; . Quu]
//     P  = F + K^T H^T                  K = -G^-1 H^T  (G K = -H^T up to the residual of the backward-stable solve, see riccatiPhase5)
// The round-2 kernel staged P, the stage record, W and K in LDS (40.7 kB, 497 LDS instructions per stage and instance) and one
// wavefront per SIMD had nothing to hide their latency behind: 15.6 us per stage under load, 8.2 us alone.  Here LDS only carries the
// 12 rows of [H^T G] to the solve, K back from it, the vector terms and the mirror image of P.
//
// SLOTS.  z is permuted so that the rows of C that are NOT dense fall on whole registers of a tile: the twelve leg joints obey
// q+ = q + dt v exactly (Fqq = diag(Fqq6, I), Fqv = diag(Fqv6, dt I)), i.e. row qj_r of C is e(qj_r) + dt e(vj_r):
//     slot  0..11 qj   12..15 qb0..3 | 16..27 vj   28..31 vb0..3 | 32 33 qb4 qb5   34 35 vb4 vb5   36..47 u
// With qj at 0..11 and vj at 16..27 the structured rows become lane-local adds between tiles 0 and 1 (same lane, same register),
// and the dense rows of C -- 6 base configuration rows + 18 velocity rows -- are exactly six k-steps: (block 0, s = 3), (1, 0..3), (2, 0).
// x+ uses slots 0..35 of the same table.
// ---------------------------------------------------------------------------------------------------------------------------------
template <typename D>
struct RiccatiSlots {
  static_assert(D::NV == 18 && D::NU == 12, "slot table of a 6 + 12 dof floating base");
  __host__ __device__ static constexpr int nat(int s) {           // slot -> natural index of z = [q v u]
    return s < 12 ? 6 + s : s < 16 ? s - 12 : s < 28 ? 24 + (s - 16) : s < 32 ? 18 + (s - 28) : s < 34 ? 4 + (s - 32) : s < 36 ? 22 + (s - 34) : s;
  }
  __host__ __device__ static constexpr int slot(int n) {          // natural -> slot
    return n < 4 ? 12 + n : n < 6 ? 32 + (n - 4) : n < 18 ? n - 6 : n < 22 ? 28 + (n - 18) : n < 24 ? 34 + (n - 22) : n < 36 ? 16 + (n - 24) : n;
  }
};

template <typename D>
struct RiccatiRegSmem {
  static constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF;
  static constexpr int LDH = 48;        // row stride of HT / KT: rows 32 dwords apart mod 64, the two lane groups of a read hit disjoint banks
  static constexpr int LDP = 49;        // row stride of PM: transposed reads walk 16 rows through 16 bank pairs
  static constexpr int HT = 0,                    // [NU][LDH]  rows u of Qh: H^T | G, slot columns
                       KT = HT + NU * LDH,        // [NU][LDH]  K, slot columns
                       RED = KT + NU * LDH,       // [3][64]    partial sums of the column contractions
                       VF = RED + 192, VT = VF + 48, LUH = VT + 48, KV = LUH + 16,      // Fx and t by slot, lu_hat, k
                       VS = KV + 16,              // [48]       s of the stage by slot (on its way to the ric record)
                       PM = VS + 48,              // [48][LDP]  P of the stage, master entries (slot order)
                       PLAIN = PM + 48 * LDP;
  // scratch of a stage with a switching constraint, natural layouts of the round-2 code; it lives between the solve and the mirror,
  // when PM is dead, and extends behind it
  static constexpr int QXU = PM, QUU = QXU + NX * NU, LU = QUU + NU * NU, KM = LU + 16, KVN = KM + NU * NX, GW = KVN + 16,
                       DG = GW + NU * NU, SS = DG + NF * NU, DTM = SS + NF * NF, SDG = DTM + NU * NX, MV = SDG + NF * NU,
                       SCORR = MV + 16, MMX = SCORR + 48,
                       GK = MMX,                  // Quu K takes the place of M (dead once DtM = Phiu^T M is formed)
                       WL = MMX + NF * NX,        // P | Phix | Phiu of the swc record (one trip to memory instead of one per step of the algebra)
                       WL_P = WL, WL_PHIX = WL + NF, WL_PHIU = WL_PHIX + NF * NX, HYB = WL_PHIU + NF * NU;
  static_assert(NF * NX == NU * NX, "GK aliases MMX");
  static constexpr int NTBL = OcpLayout<D>::R_SV + NV;             // doubles of Pqq | Pqv | Pvv | sq | sv in the ric record
  static constexpr size_t BYTES = (HYB > PLAIN ? HYB : PLAIN) * sizeof(double);
  static_assert(BYTES + 64 <= 40960, "four instances per CU");
};

template <typename D, bool HYBRID>
__global__ __launch_bounds__(64, 1) void ocp_riccati_backward_reg_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = RiccatiRegSmem<D>;
  using SL = RiccatiSlots<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF, LDH = S::LDH, LDP = S::LDP;
  static_assert(NX + NU == 48 && L::R_PQV == L::R_PQQ + L::P_TRI && L::R_PVV == L::R_PQV + NV * NV && L::R_SQ == L::R_PVV + L::P_TRI &&
                L::R_SV == L::R_SQ + NV && L::R_PQQ == 0, "ric record: Pqq | Pqv | Pvv | sq | sv");
  static_assert(L::K_FVV == L::K_FVQ + NV * NV && L::K_FVU == L::K_FVV + NV * NV, "[Fvq Fvv Fvu] contiguous");
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const OcpNode* __restrict__ nodes = B.nodes;
  const int lane = threadIdx.x, li = lane & 15, g = lane >> 4;
  const long b = blockIdx.x;
  // (values that are the same in every lane are made scalar by hand: the node table and the problem block are read with vector loads --
  // the kernel also stores to global memory, so the compiler will not use the scalar cache -- and everything derived from a vector
  // load, the record addresses first of all, would live in VGPR pairs)
  const long base = b * B.NS;
  const bool p32 = __builtin_amdgcn_readfirstlane(P->ric_fp32) != 0;
  auto st32 = [&](double x) { return p32 ? (double)(float)x : x; };
  auto slotOf = [&](int pos) { return (long)__builtin_amdgcn_readfirstlane(nodes[pos].slot); };
  if (lane == 0) s_ok = 1;
  // dense k-steps of C: (row block, k-step); the six tiles that are computed of a symmetric 3 x 3 tiling: the others are mirror images
  constexpr int DC[6] = {0, 1, 1, 1, 1, 2}, DS[6] = {3, 0, 1, 2, 3, 0};
  constexpr int TA[6] = {2, 2, 2, 0, 0, 1}, TB[6] = {0, 1, 2, 0, 1, 1};       // the rows of [H^T G] first: the solve waits for them
  // ---- loop-invariant addresses inside a kkt record ----
  // A structural zero of C is read from the first pad double of the record (records are allocated zeroed and nothing writes their
  // padding): a select behind the load would need a VALU pass over every loaded value, and with the register file as full as it is
  // here the compiler then serialises the loads (load, wait, select, move to an AGPR -- measured: 18 x the latency of HBM per stage)
  constexpr int ZERO_AT = L::K_FX + NX;
  static_assert(ZERO_AT < L::KKT, "a pad double behind Fx");
  int cOff[6][3];      // dense rows of C in accumulator layout
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int rs = 16 * DC[d] + 4 * DS[d] + g;                 // x+ slot of this lane's row
    const int rn = SL::nat(rs);                                // natural row of x+ = [q+ v+]
#pragma unroll
    for (int bb = 0; bb < 3; ++bb) {
      const int cn = SL::nat(16 * bb + li);
      int off;
      if (rn < NV) off = cn < 6 ? L::K_FQQ + rn + 6 * cn : ((cn >= NV && cn < NV + 6) ? L::K_FQV + rn + 6 * (cn - NV) : ZERO_AT);      // base rows of [Fqq Fqv 0]
      else off = L::K_FVQ + (rn - NV) + NV * cn;
      cOff[d][bb] = off;
    }
  }
  int qOff[6][4];      // [Qxx Qxu; . Quu] in accumulator layout
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ni = SL::nat(16 * TA[t] + 4 * q + g), nj = SL::nat(16 * TB[t] + li);
      qOff[t][q] = (ni < NX && nj < NX) ? L::K_QXX + L::xsym(ni, nj)
                 : (ni < NX ? L::K_QXU + ni + NX * (nj - NX) : (nj < NX ? L::K_QXU + nj + NX * (ni - NX) : L::K_QUU + (ni - NX) + NU * (nj - NX)));
    }
  // The 54 gather addresses of a lane are loop invariants that the register allocator spilled in the HYBRID instantiation (and a
  // reload from scratch behind the stage's stores drains the whole memory queue): they travel as 16-bit BYTE offsets, two to a register
  unsigned cqPk[21];
  {
    static_assert(8 * (L::KKT) < 65536 && 8 * (S::PM + 48 * LDP) < 65536, "16-bit byte offsets");
#pragma unroll
    for (int e = 0; e < 21; ++e) {
      const int e0 = 2 * e, e1 = 2 * e + 1;
      const int o0 = e0 < 18 ? cOff[e0 / 3][e0 % 3] : qOff[(e0 - 18) / 4][(e0 - 18) % 4];
      const int o1 = e1 < 18 ? cOff[e1 / 3][e1 % 3] : qOff[(e1 - 18) / 4][(e1 - 18) % 4];
      cqPk[e] = (unsigned)(8 * o0) | ((unsigned)(8 * o1) << 16);
    }
  }
  const int myNat = SL::nat(lane < 48 ? lane : 47);
  const int lzOff = myNat < NX ? L::K_LX + myNat : L::K_LU + (myNat - NX);
  // PM offset of the doubles of Pqq | Pqv | Pvv (packed upper triangles, ric record order) this lane copies to the ric record -- pairs
  // e = lane + 64 t of consecutive doubles --: the orientation of the entry that a master tile wrote
  constexpr int N2 = S::NTBL / 2, NP = (N2 + 63) / 64;
  static_assert(S::NTBL % 2 == 0 && L::R_SQ % 2 == 0, "pairs");
  int pmOff[NP][2];
#pragma unroll
  for (int t = 0; t < NP; ++t)
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      // (pairs beyond the record repeat its last pair: every lane stores, without a branch around the store -- see writeP)
      const int e2 = lane + 64 * t < N2 ? lane + 64 * t : N2 - 1;
      const int e = 2 * e2 + hh;
      int off;
      if (e >= L::R_SQ) {
        off = S::VS + SL::slot(e - L::R_SQ);                       // sq | sv: natural index -> slot
      } else {
        int r = 0, c = 0;
        if (e < L::R_PQV) { int cc = 0; while ((cc + 1) * (cc + 2) / 2 <= e && cc < NV - 1) ++cc; c = cc; r = e - cc * (cc + 1) / 2; if (r > c) r = c; }
        else if (e < L::R_PVV) { const int k = e - L::R_PQV; c = k / NV; r = k - c * NV; c += NV; }
        else { const int k = e - L::R_PVV; int cc = 0; while ((cc + 1) * (cc + 2) / 2 <= k && cc < NV - 1) ++cc; c = cc; r = k - cc * (cc + 1) / 2; if (r > c) r = c; r += NV; c += NV; }
        int i = SL::slot(r), j = SL::slot(c);
        const int bi = i >> 4, bj = j >> 4;
        const bool keep = bi == bj ? i <= j : ((bi == 0 && bj == 1) || bi == 2);
        if (!keep) { const int tt = i; i = j; j = tt; }
        off = S::PM + LDP * i + j;
      }
      pmOff[t][hh] = off;
    }
  unsigned pmPk[NP];
#pragma unroll
  for (int t = 0; t < NP; ++t) pmPk[t] = (unsigned)(8 * pmOff[t][0]) | ((unsigned)(8 * pmOff[t][1]) << 16);
  // the gain record K | k (natural order, K column-major NU x NX) as pairs of consecutive doubles read from KT (slot columns) / KV
  constexpr int G2 = (NU * NX + NU) / 2, NG = (G2 + 63) / 64;
  static_assert(L::G_K == 0 && L::G_k == NU * NX && NU % 2 == 0 && 2 * G2 <= L::GAIN, "gain record: K | k");
  unsigned gnPk[NG];
#pragma unroll
  for (int t = 0; t < NG; ++t) {
    const int e2 = lane + 64 * t < G2 ? lane + 64 * t : G2 - 1, e = 2 * e2;
    const int o0 = e < NU * NX ? S::KT + LDH * (e % NU) + SL::slot(e / NU) : S::KV + (e - NU * NX);
    const int o1 = e < NU * NX ? o0 + LDH : o0 + 1;
    gnPk[t] = (unsigned)(8 * o0) | ((unsigned)(8 * o1) << 16);
  }
  auto ldsAt = [&](const double* b0, unsigned byteoff) { return *reinterpret_cast<const double*>(reinterpret_cast<const char*>(b0) + byteoff); };
  typedef mfma_d4 tile_t;
  tile_t Pt[3][3];       // P (slot order): [row block][column block]
  double s_reg = 0.0;    // s of slot `lane`
  // store the six master tiles into PM
  auto tilesToPM = [&](const tile_t (&T)[6]) {
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) sm[S::PM + LDP * (16 * TA[t] + 4 * q + g) + 16 * TB[t] + li] = T[t][q];
  };
  // P of the stage from PM to its ric record (16-byte stores of consecutive lanes), in two halves: the LDS reads, and the global stores --
  // issued BEHIND the staging of the next record: s_waitcnt vmcnt counts loads and stores in one queue, so a wait for the prefetched
  // record in front of which these stores had just been issued would wait for their acknowledgement too
  rd2 pv[NP];
  auto readP = [&]() {
#pragma unroll
    for (int t = 0; t < NP; ++t) {
      unsigned pk = pmPk[t];
      asm volatile("" : "+v"(pk));        // (opaque: the unpacked addresses are loop invariants too, and would be hoisted and spilled)
      pv[t].x = ldsAt(sm, pk & 0xffffu); pv[t].y = ldsAt(sm, pk >> 16);
    }
  };
  // (no branch around a store: lanes beyond the record repeat its last pair.  Behind a store inside a conditional block the compiler
  // cannot count the stores that are in flight, and its next wait for a LOAD then waits for every store too)
  auto writeP = [&](double* __restrict__ rr) {
#pragma unroll
    for (int t = 0; t < NP; ++t) { const int e2 = lane + 64 * t < N2 ? lane + 64 * t : N2 - 1; reinterpret_cast<rd2*>(rr)[e2] = pv[t]; }
  };
  auto writeGain = [&](double* __restrict__ gg) {
    rd2 gv[NG];
#pragma unroll
    for (int t = 0; t < NG; ++t) {
      unsigned pk = gnPk[t];
      asm volatile("" : "+v"(pk));
      gv[t].x = ldsAt(sm, pk & 0xffffu); gv[t].y = ldsAt(sm, pk >> 16);
    }
#pragma unroll
    for (int t = 0; t < NG; ++t) { const int e2 = lane + 64 * t < G2 ? lane + 64 * t : G2 - 1; reinterpret_cast<rd2*>(gg)[e2] = gv[t]; }
  };
  // rebuild the nine tiles of P from the six masters (registers) and their mirror images (PM); entries outside the x slots are zero
  auto mirrorP = [&](const tile_t (&T)[6]) {
    // T order: (2,0) (2,1) (2,2) (0,0) (0,1) (1,1)
    const bool c4 = li < 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool r0 = (q == 0);
      Pt[2][0][q] = r0 ? T[0][q] : 0.0;
      Pt[2][1][q] = r0 ? T[1][q] : 0.0;
      Pt[0][1][q] = T[4][q];
      Pt[1][0][q] = sm[S::PM + LDP * li + 16 + 4 * q + g];                       // (16 + 4q + g, li)  <-  (li, 16 + 4q + g) of tile (0,1)
      const double m02 = sm[S::PM + LDP * (32 + (c4 ? li : 0)) + 4 * q + g];     // (4q + g, 32 + li)  <-  (32 + li, 4q + g) of tile (2,0)
      const double m12 = sm[S::PM + LDP * (32 + (c4 ? li : 0)) + 16 + 4 * q + g];
      Pt[0][2][q] = c4 ? m02 : 0.0;
      Pt[1][2][q] = c4 ? m12 : 0.0;
      const int ir = 4 * q + g;                                                  // row inside a diagonal tile
      const double d0 = sm[S::PM + LDP * li + ir], d1 = sm[S::PM + LDP * (16 + li) + 16 + ir], d2 = sm[S::PM + LDP * (32 + li) + 32 + ir];
      Pt[0][0][q] = ir <= li ? T[3][q] : d0;
      Pt[1][1][q] = ir <= li ? T[5][q] : d1;
      Pt[2][2][q] = (r0 && c4) ? (ir <= li ? T[2][q] : d2) : 0.0;
    }
  };
  // ---- terminal stage (riccati_recursion_solver.cpp:53-56): P = blockdiag(Qxx_qq, Qxx_vv), s = -lx ----
  {
    const double* __restrict__ kk = B.kkt + (base + slotOf(M - 1)) * L::KKT;
    double* __restrict__ rr = B.ric + (base + slotOf(M - 1)) * L::RIC;
    tile_t T[6];
#pragma unroll
    for (int t = 0; t < 6; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ni = SL::nat(16 * TA[t] + 4 * q + g), nj = SL::nat(16 * TB[t] + li);
        const bool in = ni < NX && nj < NX && ((ni < NV) == (nj < NV));
        const double v = kk[in ? L::K_QXX + L::xsym(ni, nj) : 0];
        T[t][q] = in ? st32(v) : 0.0;
      }
    if (lane < NX) s_reg = st32(-kk[L::K_LX + myNat]);
    if (lane < 48) sm[S::VS + lane] = s_reg;
    tilesToPM(T);
    waveLdsSync();
    mirrorP(T);
    readP();
    writeP(rr);
    waveLdsSync();
  }
  // ---- registers of the stage: dense rows of C, Q, lz, Fx ----
  // The record travels as 16-byte pieces of consecutive lanes (prefetch: issued in the middle of the previous stage, parked in
  // registers), is laid down in LDS where PM was (dead by then) and gathered from there into the accumulator layout.  (Gathering
  // straight from global memory -- 44 loads of 64 scattered 8-byte addresses each -- kept the wavefront in the address path for 3 us
  // per stage alone and 6 us under load.)
  constexpr int SL2 = (ZERO_AT + 2) / 2, PF = (SL2 + 63) / 64;
  static_assert(ZERO_AT % 2 == 0 && 2 * SL2 <= L::KKT && 2 * SL2 <= 48 * LDP && S::PM % 2 == 0 && L::KKT % 2 == 0, "the staged record fits the record and the PM block");
  rd2 pre[PF];
  auto prefetch = [&](const double* __restrict__ kk) {
    const rd2* __restrict__ k2 = reinterpret_cast<const rd2*>(kk);
#pragma unroll
    for (int t = 0; t < PF; ++t) { const int e = lane + 64 * t; pre[t] = k2[e < SL2 ? e : SL2 - 1]; }
  };
  double Cd[6][3], lz = 0.0, fx = 0.0;
  tile_t Qc[6];
  auto stageIn = [&]() {
#pragma unroll
    for (int t = 0; t < PF; ++t) { const int e = lane + 64 * t; if (64 * t + 63 < SL2 || e < SL2) reinterpret_cast<rd2*>(&sm[S::PM])[e] = pre[t]; }      // (only the last piece is partial: the others store without a mask)
    waveLdsSync();
    const double* kk = &sm[S::PM];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      unsigned pk = cqPk[e];
      asm volatile("" : "+v"(pk));      // (opaque: the unpacked addresses are loop invariants too, and would be hoisted and spilled)
      const int e0 = 2 * e, e1 = 2 * e + 1;
      Cd[e0 / 3][e0 % 3] = ldsAt(kk, pk & 0xffffu);
      Cd[e1 / 3][e1 % 3] = ldsAt(kk, pk >> 16);
    }
    lz = kk[lzOff];
    fx = kk[L::K_FX + (myNat < NX ? myNat : 0)];
    waveLdsSync();
  };
  // Q of the stage is gathered when phase 2 is about to need it (the staged record stays where it is until the end of the stage: PM and
  // the scratch of a constrained stage are written behind phase 2): 48 registers less while phase 1 holds P, C and Wt
  auto gatherQ = [&]() {
    const double* kk = &sm[S::PM];
#pragma unroll
    for (int e = 9; e < 21; ++e) {
      unsigned pk = cqPk[e];
      asm volatile("" : "+v"(pk));
      const int e0 = 2 * e - 18, e1 = 2 * e - 17;
      Qc[e0 / 4][e0 % 4] = ldsAt(kk, pk & 0xffffu);
      Qc[e1 / 4][e1 % 4] = ldsAt(kk, pk >> 16);
    }
  };
  if (M > 1) { prefetch(B.kkt + (base + slotOf(M - 2)) * L::KKT); stageIn(); }
  // the node fields of a stage are fetched one stage ahead (a dependent L2 round trip in front of the record's loads otherwise)
  int n_slot = M > 1 ? nodes[M - 2].slot : 0, n_dimi = (HYBRID && M > 1) ? nodes[M - 2].sw_dimi : 0;
  double n_dt = M > 1 ? nodes[M - 2].dtq : 0.0;
  for (int i = M - 2; i >= 0; --i) {
    const double dt = n_dt;
    const long rec = base + __builtin_amdgcn_readfirstlane(n_slot);
    const int dimi = HYBRID ? __builtin_amdgcn_readfirstlane(n_dimi) : 0;
    if (i > 0) { n_slot = nodes[i - 1].slot; n_dt = nodes[i - 1].dtq; if (HYBRID) n_dimi = nodes[i - 1].sw_dimi; }
    const bool constrained = HYBRID && dimi > 0;
#ifdef IDOCP_S3_STAMPS      // (diagnostic build: a store inside a conditional block costs the memory-queue bookkeeping of the whole loop)
    const bool stamp = lane == 0 && b == (gridDim.x > 7 ? 7 : 0) && i == M / 2 && B.prof != nullptr;
#define RSTAMP(k) do { if (stamp) B.prof[16 + k] = wall_clock64(); } while (0)
    const bool cstamp = lane == 0 && b == (gridDim.x > 7 ? 7 : 0) && constrained && B.prof != nullptr;      // the constrained stages (the last one walked wins)
#define CSTAMP(k) do { if (cstamp) B.prof[32 + k] = wall_clock64(); } while (0)
#else
#define RSTAMP(k) do { } while (0)
#define CSTAMP(k) do { } while (0)
#endif
    RSTAMP(0);
    double* __restrict__ rr = B.ric + rec * L::RIC;
    double* __restrict__ gg = B.gain + rec * L::GAIN;
    // ---- vector head: t = P Fx - s, lz_hat = [lx; lu] + C^T t (backward_riccati_recursion_factorizer.hxx:141-160 and the lu term) ----
    // y[j] = sum_i X[i][j] v[i] over tiles in accumulator layout: every lane multiplies its registers with the v of their rows and the
    // four lane groups add up through LDS
    // The head is a chain of four LDS round trips with a handful of multiply-adds each; phase 1 -- Wt = P C (:48-78): nine tiles, six
    // dense k-steps each -- needs nothing from it, so its k-steps are issued IN BETWEEN: the matrix core works through nine queued
    // instructions (576 cycles) while the wavefront waits for LDS.
    tile_t Wt[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) Wt[a][bb] = tile_t{0.0, 0.0, 0.0, 0.0};
    // The third row tile of Wt has FOUR rows (slots 32 .. 35 of x; the slots behind them are u, and P has no such columns): on the 16 x 16 x 4
    // instruction that is 18 of the 54 products of the phase for a quarter of a tile each.  v_mfma_f64_4x4x4_4b_f64 takes its operands in the
    // SAME lane map (A: lane 16 k + i16, B: lane 16 k + j16) but forms only the four diagonal 4 x 4 blocks, D[4 b + i][4 b + j] at lane
    // 16 i + 4 b + j -- which is where register 0 of an accumulator-layout tile keeps rows 0 .. 3 -- at 8 instead of 59 ns.  So
    // the operand's first four columns are copied into all four blocks (two DPP row shifts under a bank mask) and the product of block-row 0
    // with the sixteen columns of a C tile drops straight into Wt[2][bb][0].  IDOCP_S3_MFMA4=0 at build time keeps the old form.
#ifndef IDOCP_S3_MFMA4
#define IDOCP_S3_MFMA4 1
#endif
    auto quadToRow = [&](double v) -> double {      // lanes 4 b + i of every row of 16  <-  lane i of that row
      int lo = __double2loint(v), hi = __double2hiint(v);
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0x114, 0xF, 0x2, false);      // row_shr:4 into bank 1
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0x114, 0xF, 0x2, false);
      lo = __builtin_amdgcn_update_dpp(lo, lo, 0x118, 0xF, 0xC, false);      // row_shr:8 into banks 2, 3
      hi = __builtin_amdgcn_update_dpp(hi, hi, 0x118, 0xF, 0xC, false);
      return __hiloint2double(hi, lo);
    };
    auto phase1 = [&](auto dtag) {
      constexpr int d = decltype(dtag)::value;
#pragma unroll
      for (int a = 0; a < (IDOCP_S3_MFMA4 ? 2 : 3); ++a)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) Wt[a][bb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[DC[d]][a][DS[d]], Cd[d][bb], Wt[a][bb], 0, 0, 0);
      if (IDOCP_S3_MFMA4) {
        const double p4 = quadToRow(Pt[DC[d]][2][DS[d]]);
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) Wt[2][bb][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(p4, Cd[d][bb], Wt[2][bb][0], 0, 0, 0);
      }
    };
    if (lane < 48) sm[S::VF + lane] = lane < NX ? fx : 0.0;
    waveLdsSync();
    double lzh = 0.0;
    {
      double part[3] = {0.0, 0.0, 0.0};
      double fr[9];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < (a < 2 ? 4 : 1); ++q) fr[4 * a + q] = sm[S::VF + 16 * a + 4 * q + g];
      phase1(std::integral_constant<int, 0>{});
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < (a < 2 ? 4 : 1); ++q)
#pragma unroll
          for (int bb = 0; bb < 3; ++bb) part[bb] += Pt[a][bb][q] * fr[4 * a + q];
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) sm[S::RED + 64 * bb + lane] = part[bb];
      waveLdsSync();
      const int rb = S::RED + 64 * (lane < 48 ? (lane >> 4) : 0) + li;
      const double r0 = sm[rb], r1 = sm[rb + 16], r2 = sm[rb + 32], r3 = sm[rb + 48];
      phase1(std::integral_constant<int, 1>{});
      const double y = (r0 + r1) + (r2 + r3);
      const double t = lane < NX ? y - s_reg : 0.0;
      waveLdsSync();
      if (lane < 48) sm[S::VT + lane] = t;
      waveLdsSync();
      double tv[6];
#pragma unroll
      for (int d = 0; d < 6; ++d) tv[d] = sm[S::VT + 16 * DC[d] + 4 * DS[d] + g];
      // the structured rows: row qj_r of C is e(slot r) + dt e(slot 16 + r)
      const double ts = sm[S::VT + (lane < 12 ? lane : (lane >= 16 && lane < 28 ? lane - 16 : 0))];
      phase1(std::integral_constant<int, 2>{});
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) part[bb] = 0.0;
#pragma unroll
      for (int d = 0; d < 6; ++d)
#pragma unroll
        for (int bb = 0; bb < 3; ++bb) part[bb] += Cd[d][bb] * tv[d];
#pragma unroll
      for (int bb = 0; bb < 3; ++bb) sm[S::RED + 64 * bb + lane] = part[bb];
      waveLdsSync();
      const double u0 = sm[rb], u1 = sm[rb + 16], u2 = sm[rb + 32], u3 = sm[rb + 48];
      phase1(std::integral_constant<int, 3>{});
      const double y2 = (u0 + u1) + (u2 + u3);
      lzh = lz + y2 + (lane < 12 ? ts : (lane >= 16 && lane < 28 ? dt * ts : 0.0));
    }
    RSTAMP(1);
    phase1(std::integral_constant<int, 4>{});
    phase1(std::integral_constant<int, 5>{});
    gatherQ();
    {
      const bool m12 = li < 12;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < (a < 2 ? 4 : 1); ++q) {
          const double pv = m12 ? Pt[a][0][q] : 0.0;
          Wt[a][0][q] += pv;
          Wt[a][1][q] += dt * pv;
        }
    }
    RSTAMP(2);
    // ---- phase 2: Qh = Q + C^T Wt (:79-113) on the six master tiles.  The rows of [H^T G] -- tiles (2, b) -- first: they go to the solve
    //      through LDS while the matrix core works on the three tiles of F ----
    tile_t Qh[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) Qh[t] = Qc[t];
#pragma unroll
    for (int d = 0; d < 6; ++d)
#pragma unroll
      for (int t = 0; t < 3; ++t) Qh[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cd[d][TA[t]], Wt[DC[d]][TB[t]][DS[d]], Qh[t], 0, 0, 0);
#pragma unroll
    for (int d = 0; d < 2; ++d)
#pragma unroll
      for (int t = 3; t < 6; ++t) Qh[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cd[d][TA[t]], Wt[DC[d]][TB[t]][DS[d]], Qh[t], 0, 0, 0);
    // ---- the rows of [H^T G] and lu_hat go to the solve ----
#pragma unroll
    for (int bb = 0; bb < 3; ++bb)
#pragma unroll
      for (int q = 1; q < 4; ++q) sm[S::HT + LDH * (4 * (q - 1) + g) + 16 * bb + li] = Qh[bb][q];
    if (lane >= NX && lane < 48) sm[S::LUH + lane - NX] = lzh;
#pragma unroll
    for (int d = 2; d < 6; ++d)
#pragma unroll
      for (int t = 3; t < 6; ++t) Qh[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Cd[d][TA[t]], Wt[DC[d]][TB[t]][DS[d]], Qh[t], 0, 0, 0);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      Qh[3][q] += Wt[0][0][q];                 // rows qj of tile (0, 0), (0, 1)
      Qh[4][q] += Wt[0][1][q];
      Qh[5][q] += dt * Wt[0][1][q];            // rows vj of tile (1, 1)
    }
    RSTAMP(3);
    waveLdsSync();
    // the record of the next stage of the walk: its loads are in flight during the solve and phase 5
    if (i > 0) prefetch(B.kkt + (base + __builtin_amdgcn_readfirstlane(n_slot)) * L::KKT);
    RSTAMP(9);
    double Kr[3][3];      // K as the operand of phase 5: [column block][k-step - 1], NEGATED on a constrained stage (see below)
    tile_t Pn[6];
    if (!constrained) {
      // Quu = L L^T and K = -Quu^-1 Qxu^T, k = -Quu^-1 lu in registers, one right-hand side per lane (Eigen::LLT compute + solve,
      // split_riccati_factorizer.hxx:43-46)
      double x[NU], h[NU];
      int xo = lane < NX ? S::HT + lane : S::LUH, xs = lane < NX ? LDH : 1;      // column `lane` of H^T, or lu_hat
      asm volatile("" : "+v"(xo), "+v"(xs));                                   // (twelve hoisted addresses otherwise)
#pragma unroll
      for (int m = 0; m < NU; ++m) { x[m] = sm[xo + xs * m]; h[m] = x[m]; }
      choleskySolveRows<NU>(&sm[S::HT + NX], LDH, lane, &s_ok, x);         // G is symmetric: element (row, j) read as (j, row)
      {
        const int kc = lane < 48 ? lane : 47;              // (columns 36 .. 47 of KT are never read: no branch needed)
#pragma unroll
        for (int m = 0; m < NU; ++m) sm[S::KT + LDH * m + kc] = -x[m];
      }
      if (lane == NX) {
#pragma unroll
        for (int m = 0; m < NU; ++m) sm[S::KV + m] = -x[m];
      }
      waveLdsSync();
      RSTAMP(10);
      writeGain(gg);                                       // K, k: gathered from KT / KV into the natural order of the record
      // s = -lx_hat - H k (:141-160)
      double hk = 0.0;
#pragma unroll
      for (int m = 0; m < NU; ++m) hk += h[m] * sm[S::KV + m];
      s_reg = lane < NX ? st32(-lzh - hk) : 0.0;
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) { const int c = 16 * a + li; const double v = sm[S::KT + LDH * (4 * s + g) + (c < NX ? c : 0)]; Kr[a][s] = c < NX ? v : 0.0; }
      RSTAMP(4);
      // ---- phase 5: P = F - K^T G K = F + K^T H^T (:122-131; G K = -H^T, see riccatiPhase5) ----
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        Pn[t] = Qh[t];
        if (TA[t] == 2) { Pn[t][1] = 0.0; Pn[t][2] = 0.0; Pn[t][3] = 0.0; }
      }
      // (tiles 0 .. 2 are the row tile with the four rows 32 .. 35 of x: 4 x 4 x 4 products into register 0, as in phase 1)
#pragma unroll
      for (int s = 0; s < 3; ++s) {
#pragma unroll
        for (int t = (IDOCP_S3_MFMA4 ? 3 : 0); t < 6; ++t) Pn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[TA[t]][s], Qh[TB[t]][s + 1], Pn[t], 0, 0, 0);
        if (IDOCP_S3_MFMA4) {
          const double k4 = quadToRow(Kr[2][s]);
#pragma unroll
          for (int t = 0; t < 3; ++t) Pn[t][0] = __builtin_amdgcn_mfma_f64_4x4x4f64(k4, Qh[TB[t]][s + 1], Pn[t][0], 0, 0, 0);
        }
      }
    } else {
      // ---- stage with a switching constraint (split_riccati_factorizer.hxx:56-101): the Schur-complement step of the round-2 kernel
      //      on natural-layout copies in LDS ----
      CSTAMP(0);
      double* Qxu = &sm[S::QXU];
      double* Quu = &sm[S::QUU];
      double* lu = &sm[S::LU];
      // the constraint's P | Phix and Phiu (two contiguous pieces of the swc record) on their way to LDS: in flight during the first
      // factorisation
      const double* __restrict__ W = B.swc + rec * L::SWC;
      constexpr int W2A = (L::W_PHIX + NF * NX) / 2, W2B = NF * NU / 2, NWL = (W2A + W2B + 63) / 64;
      static_assert(L::W_P == 0 && L::W_PHIX == NF && (L::W_PHIX + NF * NX) % 2 == 0 && L::W_PHIU % 2 == 0 && S::WL % 2 == 0 && L::SWC % 2 == 0, "16-byte pieces of the swc record");
      rd2 wl[NWL];
#pragma unroll
      for (int t = 0; t < NWL; ++t) {
        const int e = lane + 64 * t;
        wl[t] = reinterpret_cast<const rd2*>(W)[e < W2A ? e : (e < W2A + W2B ? L::W_PHIU / 2 + (e - W2A) : 0)];
      }
#pragma unroll
      for (int bb = 0; bb < 3; ++bb)
#pragma unroll
        for (int q = 1; q < 4; ++q) {
          const int u = 4 * (q - 1) + g, j = 16 * bb + li;
          if (j < NX) Qxu[SL::nat(j) + NX * u] = Qh[bb][q]; else Quu[u + NU * (j - NX)] = Qh[bb][q];
        }
      if (lane >= NX && lane < 48) lu[lane - NX] = lzh;
      waveLdsSync();
      {
        double x[NU];
#pragma unroll
        for (int m = 0; m < NU; ++m) x[m] = (m == lane) ? 1.0 : 0.0;      // Ginv = llt.solve(I) (:60)
        choleskySolveRows<NU>(Quu, NU, lane, &s_ok, x);
        if (lane < NU) {
#pragma unroll
          for (int m = 0; m < NU; ++m) sm[S::GW + m + NU * lane] = x[m];
        }
      }
#pragma unroll
      for (int t = 0; t < NWL; ++t) { const int e = lane + 64 * t; if (e < W2A + W2B) reinterpret_cast<rd2*>(&sm[S::WL])[e] = wl[t]; }
      waveLdsSync();
      CSTAMP(1);
      // The algebra of the switching rows loops over dimi, a run-time number: three per foot that touches down.  The gaits of the configs bring two
      // feet down at a time, so the block exists a second time with dimi = 6 at compile time (the divisions by it become shifts, the
      // six-term loops unroll and their LDS reads are in flight together): same operations in the same order, bitwise the same results.
      auto switchingRows = [&](auto ditag) {
        constexpr int DI = decltype(ditag)::value;
        const int dimi_ = DI > 0 ? DI : dimi;
        const double* Phiu = &sm[S::WL_PHIU];
        const double* Phix = &sm[S::WL_PHIX];
        const double* Pv = &sm[S::WL_P];
        for (int e = lane; e < dimi_ * NU; e += 64) {                // DGinv = Phiu Ginv
          const int c = e / dimi_, j = e - c * dimi_;
          double acc = 0.0;
  #pragma unroll
          for (int m = 0; m < NU; ++m) acc += Phiu[j + NF * m] * sm[S::GW + m + NU * c];
          sm[S::DG + j + NF * c] = acc;
        }
        waveLdsSync();
        for (int e = lane; e < dimi_ * dimi_; e += 64) {              // S = DGinv Phiu^T
          const int c = e / dimi_, j = e - c * dimi_;
          double acc = 0.0;
  #pragma unroll
          for (int m = 0; m < NU; ++m) acc += sm[S::DG + j + NF * m] * Phiu[c + NF * m];
          sm[S::SS + j + NF * c] = acc;
        }
        waveLdsSync();
        CSTAMP(2);
        // S = L L^T and S^-1 [DGinv, Phix, P] by triangular solves (llt_s_.solve, :64-66, 71-74), lane = column.  dimi is 3 per foot that
        // touches down: the trotting and running gaits bring two feet down at a time, so the 6-row instantiation serves them
        static_assert(NU + NX + 1 <= 64, "one lane per right-hand side");
        auto sSolve = [&](auto ntag) {
          constexpr int NR = decltype(ntag)::value;
          double x[NR];
  #pragma unroll
          for (int j = 0; j < NR; ++j) {
            double val = 0.0;
            if (j < dimi_) val = lane < NU ? sm[S::DG + j + NF * lane] : (lane < NU + NX ? Phix[j + NF * (lane - NU)] : (lane == NU + NX ? Pv[j] : 0.0));
            x[j] = val;
          }
          choleskySolveRows<NR>(&sm[S::SS], NF, lane, &s_ok, x, dimi_);
  #pragma unroll
          for (int j = 0; j < NR; ++j) {
            if (j < dimi_) {
              if (lane < NU) sm[S::SDG + j + NF * lane] = x[j];
              else if (lane < NU + NX) sm[S::MMX + j + NF * (lane - NU)] = x[j];
              else if (lane == NU + NX) sm[S::MV + j] = x[j];
            }
          }
        };
        if (dimi_ <= 6) sSolve(std::integral_constant<int, 6>{}); else sSolve(std::integral_constant<int, NF>{});
        waveLdsSync();
        CSTAMP(3);
        for (int e = lane; e < NU * NU; e += 64) {                  // Ginv -= SinvDGinv^T DGinv
          const int c = e / NU, r = e - c * NU;
          double acc = 0.0;
          for (int l = 0; l < dimi_; ++l) acc += sm[S::SDG + l + NF * r] * sm[S::DG + l + NF * c];
          sm[S::GW + e] -= acc;
        }
        waveLdsSync();
        {
          // K = -Ginv Qxu^T - SinvDGinv^T Phix, k = -Ginv lu - SinvDGinv^T P with the updated Ginv (:67-70)
          for (int e = lane; e < NU * NX; e += 64) {
            const int c = e / NU, j = e - c * NU;
            double acc = 0.0;
  #pragma unroll
            for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * Qxu[c + NX * m];
            for (int l = 0; l < dimi_; ++l) acc += sm[S::SDG + l + NF * j] * Phix[l + NF * c];
            sm[S::KM + e] = -acc;
          }
          if (lane >= 64 - NU) {
            const int j = lane - (64 - NU);
            double acc = 0.0;
  #pragma unroll
            for (int m = 0; m < NU; ++m) acc += sm[S::GW + j + NU * m] * lu[m];
            for (int l = 0; l < dimi_; ++l) acc += sm[S::SDG + l + NF * j] * Pv[l];
            sm[S::KVN + j] = -acc;
          }
        }
        CSTAMP(4);
        {
          // multiplier policy dxi = M dx + m (:71-74): M = S^-1 Phix - SinvDGinv Qxu^T, m = S^-1 P - SinvDGinv lu
          double* __restrict__ Ww = B.swc + rec * L::SWC;
          for (int e = lane; e < dimi_ * NX; e += 64) {
            const int c = e / dimi_, l = e - c * dimi_;
            double acc = sm[S::MMX + l + NF * c];
  #pragma unroll
            for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * Qxu[c + NX * m];
            sm[S::MMX + l + NF * c] = acc;
            Ww[L::W_M + l + NF * c] = acc;
          }
          if (lane < dimi_) {
            const int l = lane;
            double acc = sm[S::MV + l];
            for (int m = 0; m < NU; ++m) acc -= sm[S::SDG + l + NF * m] * lu[m];
            sm[S::MV + l] = acc;
            Ww[L::W_m + l] = acc;
          }
        }
        waveLdsSync();
        CSTAMP(5);
        for (int e = lane; e < NU * NX; e += 64) {                  // DtM = Phiu^T M (:88)
          const int c = e / NU, m = e - c * NU;
          double acc = 0.0;
          for (int l = 0; l < dimi_; ++l) acc += Phiu[l + NF * m] * sm[S::MMX + l + NF * c];
          sm[S::DTM + m + NU * c] = acc;
        }
        if (lane < NX) {                                            // Phix^T m (:98-99)
          double acc = 0.0;
          for (int l = 0; l < dimi_; ++l) acc += Phix[l + NF * lane] * sm[S::MV + l];
          sm[S::SCORR + lane] = acc;
        }
        waveLdsSync();
      };
      if (dimi == 6) switchingRows(std::integral_constant<int, 6>{}); else switchingRows(std::integral_constant<int, 0>{});
      CSTAMP(6);
      riccatiPhase4<D, true, 0, 3>(Quu, &sm[S::KM], &sm[S::GK], lane);     // GK = Quu K (backward_riccati_recursion_factorizer.hxx:128)
      waveLdsSync();
      RSTAMP(10);
      // gain record, s = -lx_hat - Qxu k - Phix^T m
      for (int e = lane; e < NU * NX / 2; e += 64) reinterpret_cast<rd2*>(gg)[e] = reinterpret_cast<const rd2*>(&sm[S::KM])[e];
      if (lane < NU) gg[L::G_k + lane] = sm[S::KVN + lane];
      static_assert(S::KM % 2 == 0 && L::G_K == 0 && (NU * NX) % 2 == 0, "K is one contiguous copy");
      if (lane < NX) {
        double hk = 0.0;
#pragma unroll
        for (int m = 0; m < NU; ++m) hk += Qxu[myNat + NX * m] * sm[S::KVN + m];
        s_reg = st32(-lzh - hk - sm[S::SCORR + myNat]);
      }
      RSTAMP(4);
      CSTAMP(7);
      // ---- phase 5: P = F - K^T (G K) - K^T DtM - DtM^T K (:122-131, split_riccati_factorizer.hxx:88-97): [K; DtM]^T [G K + DtM; K] ----
      double Gr[3][3], Dr[3][3];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int s = 0; s < 3; ++s) {
          const int c = 16 * a + li, cn = NU * SL::nat(c < NX ? c : 0) + 4 * s + g;
          const double kv = sm[S::KM + cn], gv = sm[S::GK + cn], dv = sm[S::DTM + cn];
          Kr[a][s] = c < NX ? -kv : 0.0;
          Gr[a][s] = c < NX ? gv + dv : 0.0;
          Dr[a][s] = c < NX ? dv : 0.0;
        }
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        Pn[t] = Qh[t];
        if (TA[t] == 2) { Pn[t][1] = 0.0; Pn[t][2] = 0.0; Pn[t][3] = 0.0; }
      }
#pragma unroll
      for (int s = 0; s < 3; ++s)
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          Pn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Kr[TA[t]][s], Gr[TB[t]][s], Pn[t], 0, 0, 0);      // - K^T (G K + DtM)
          Pn[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(Dr[TA[t]][s], Kr[TB[t]][s], Pn[t], 0, 0, 0);      // - DtM^T K
        }
    }
    RSTAMP(5);
    CSTAMP(8);
    if (p32) {
#pragma unroll
      for (int t = 0; t < 6; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) Pn[t][q] = (double)(float)Pn[t][q];
    }
    // ---- P stays EXACTLY symmetric: master entries (upper triangle in slot order) and their mirror images through LDS.  This replaces
    //      the reference's P = (P + P^T) / 2 (:133-135); an antisymmetric rounding residue grows 2.9 x per stage otherwise. ----
    waveLdsSync();                  // (every read of the constrained stage's scratch, which PM overlays, is done)
    if (lane < 48) sm[S::VS + lane] = s_reg;
    tilesToPM(Pn);
    waveLdsSync();
    mirrorP(Pn);
    readP();
    waveLdsSync();
    RSTAMP(7);
    if (i > 0) stageIn();             // the next record: registers -> LDS (over PM) -> accumulator layout
    writeP(rr);
    RSTAMP(8);
    CSTAMP(15);
#undef RSTAMP
#undef CSTAMP
  }
  if (lane == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1;
}

// S4: forward sweep, one wavefront per instance.  The sweep is a chain of small matrix-vector products whose operands (the gain
// K, k and the dynamics blocks of the stage, 11.7 kB) do not depend on the state: the records of stage i + 1 are fetched with
// 16-byte loads into registers (two stages ahead) while stage i is computed out of LDS, so that the chain never waits for HBM.
template <typename D>
__global__ __launch_bounds__(64) void ocp_riccati_forward_kernel(OcpBuffers B, const double* __restrict__ q0,
                                                                const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NU = D::NU;
  typedef double d2 __attribute__((ext_vector_type(2)));
  // the part of the kkt record the sweep reads: Fqq6 Fqv6 Fvq Fvv Fvu (lx lu) Fx, contiguous
  constexpr int FO = L::K_FQQ, FL = L::K_FX + NX - L::K_FQQ, GL = L::GAIN;
  static_assert(FO % 2 == 0 && FL % 2 == 0 && GL % 2 == 0 && L::KKT % 2 == 0, "16-byte loads");
  constexpr int F2 = FL / 2, G2 = GL / 2, NF2 = (F2 + 63) / 64, NG2 = (G2 + 63) / 64;
  __shared__ __attribute__((aligned(16))) double fb[FL], gb[GL];
  __shared__ double dx[NX], du[NU], dxn[NX];
  // the node table as the sweep needs it (slot, time step), in LDS up front: a stage's loads are requested two stages ahead, and an address
  // that itself waits for a load from the node table shortens that distance by a memory latency (round 4; chains beyond the table read it
  // from memory as before)
  constexpr int MAXM = 1024;
  __shared__ int s_slot[MAXM];
  __shared__ double s_dtq[MAXM];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const OcpNode* __restrict__ nodes = B.nodes;
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const long base = b * B.NS;
  const bool tab = M <= MAXM;
  if (tab) {
    for (int i = lane; i < M; i += 64) { s_slot[i] = nodes[i].slot; s_dtq[i] = nodes[i].dtq; }
    waveLdsSync();
  }
  auto slotOf = [&](int i) -> int { return tab ? s_slot[i] : nodes[i].slot; };
  d2 frA[NF2], grA[NG2], frB[NF2], grB[NG2];        // two stages in flight
  auto fetch = [&](int i, d2 (&fr)[NF2], d2 (&gr)[NG2]) {
    const long rec = base + slotOf(i);
    const d2* __restrict__ fp = reinterpret_cast<const d2*>(B.kkt + rec * L::KKT + FO);
    const d2* __restrict__ gp = reinterpret_cast<const d2*>(B.gain + rec * L::GAIN);
#pragma unroll
    for (int t = 0; t < NF2; ++t) { const int e = lane + 64 * t; fr[t] = fp[e < F2 ? e : F2 - 1]; }
#pragma unroll
    for (int t = 0; t < NG2; ++t) { const int e = lane + 64 * t; gr[t] = gp[e < G2 ? e : G2 - 1]; }
  };
  if (M > 1) fetch(0, frA, grA);
  if (M > 2) fetch(1, frB, grB);
  const double* __restrict__ s0 = B.sol + (base + slotOf(0)) * L::SOL;
  // RiccatiRecursionSolver::computeInitialStateDirection (riccati_recursion_solver.cpp:110-126)
  if (lane == 0) {
    double R[9], p[3], d6[6];
    lieRelative(s0 + L::S_Q, q0 + b * NQ, R, p);         // q (-) s[0].q
    lieLog6(R, p, d6);
    const double* __restrict__ Fi = B.exp + (base + slotOf(0)) * L::EXP + L::E_FQQPI;
    for (int r = 0; r < 6; ++r) { double acc = 0.0; for (int m = 0; m < 6; ++m) acc += Fi[r + 6 * m] * d6[m]; dx[r] = -acc; }
  }
  if (lane >= 6 && lane < NV) dx[lane] = q0[b * NQ + lane + 1] - s0[L::S_Q + lane + 1];
  if (lane < NV) dx[NV + lane] = v0[b * NV + lane] - s0[L::S_V + lane];
  const double* Fqq6 = fb + (L::K_FQQ - FO);
  const double* Fqv6 = fb + (L::K_FQV - FO);
  const double* Fvq = fb + (L::K_FVQ - FO);
  const double* Fvv = fb + (L::K_FVV - FO);
  const double* Fvu = fb + (L::K_FVU - FO);
  const double* Fx = fb + (L::K_FX - FO);
  auto step = [&](int i, d2 (&fr)[NF2], d2 (&gr)[NG2]) {       // forwardRiccatiRecursion along the chain (riccati_recursion_solver.cpp:129-162)
    const long rec = base + slotOf(i);
    const double dt = tab ? s_dtq[i] : nodes[i].dtq;
    double* __restrict__ dd = B.dir + rec * L::DIR;
#pragma unroll
    for (int t = 0; t < NF2; ++t) { const int e = lane + 64 * t; if (e < F2) reinterpret_cast<d2*>(fb)[e] = fr[t]; }
#pragma unroll
    for (int t = 0; t < NG2; ++t) { const int e = lane + 64 * t; if (e < G2) reinterpret_cast<d2*>(gb)[e] = gr[t]; }
    waveLdsSync();
    if (i + 2 < M - 1) fetch(i + 2, fr, gr);
    // du = K dx + k: four lanes per row, nine terms each
    {
      const int j = lane >> 2, part = lane & 3;
      double acc = 0.0;
      if (lane < 4 * NU) {
#pragma unroll
        for (int t = 0; t < NX / 4; ++t) { const int c = (NX / 4) * part + t; acc += gb[L::G_K + j + NU * c] * dx[c]; }
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      if (lane < 4 * NU && part == 0) { acc += gb[L::G_k + j]; du[j] = acc; dd[L::D_U + j] = acc; }
    }
    if (lane < NV) { dd[L::D_Q + lane] = dx[lane]; dd[L::D_V + lane] = dx[NV + lane]; }
    waveLdsSync();
    // dx+ = A dx + B du + Fx: two lanes per velocity row (Fvq dq + half of Fvu du | Fvv dv + the other half), one per configuration row
    {
      double acc = 0.0;
      if (lane < 2 * NV) {
        const int r = lane >> 1, half = lane & 1;
        const double* Fm = half ? Fvv : Fvq;
        const double* xs = half ? dx + NV : dx;
#pragma unroll
        for (int c = 0; c < NV; ++c) acc += Fm[r + NV * c] * xs[c];
#pragma unroll
        for (int j = 0; j < NU / 2; ++j) acc += Fvu[r + NV * ((NU / 2) * half + j)] * du[(NU / 2) * half + j];
      }
      acc += __shfl_xor(acc, 1);
      if (lane < 2 * NV && (lane & 1) == 0) dxn[NV + (lane >> 1)] = acc + Fx[NV + (lane >> 1)];
      if (lane >= 2 * NV && lane < 3 * NV) {
        const int r = lane - 2 * NV;
        double dq = Fx[r];
        if (r < 6) {
#pragma unroll
          for (int m = 0; m < 6; ++m) dq += Fqq6[r + 6 * m] * dx[m] + Fqv6[r + 6 * m] * dx[NV + m];
        } else {
          dq += dx[r] + dt * dx[NV + r];
        }
        dxn[r] = dq;
      }
    }
    waveLdsSync();
    if (lane < NX) dx[lane] = dxn[lane];
    waveLdsSync();
  };
  for (int i = 0; i < M - 1; i += 2) {
    step(i, frA, grA);
    if (i + 1 < M - 1) step(i + 1, frB, grB);
  }
  double* __restrict__ dd = B.dir + (base + slotOf(M - 1)) * L::DIR;
  if (lane < NV) { dd[L::D_Q + lane] = dx[lane]; dd[L::D_V + lane] = dx[NV + lane]; }
}

template <typename D>
void OcpLaunch<D>::riccatiBackward(const OcpBuffers& B, long batch, int M, bool hybrid, hipStream_t st, bool wide) {
  (void)M;
  const size_t smem = RiccatiSmem<D>::TOTAL * sizeof(double), smem_h = smem;      // the Schur blocks alias dead ones
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 256, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 128, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 128, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 256, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 512, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_h);
    (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_kernel<D, 512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  static const int nt_env = getenv("IDOCP_RICCATI_NT") ? atoi(getenv("IDOCP_RICCATI_NT")) : 0;
  // round 4: the register-resident sweep (one wavefront per instance at every batch size); IDOCP_RICCATI_REG=0 selects the round-2 kernels
  static const int reg_env = getenv("IDOCP_RICCATI_REG") ? atoi(getenv("IDOCP_RICCATI_REG")) : 1;
  // Latency mode (a handful of instances, idocp_ocp_set_riccati_sweep): EIGHT wavefronts per instance on the LDS-staged kernel -- the tile
  // products of a phase run side by side on the four matrix cores of a CU instead of one behind the other on one (batch 1: 0.79 against
  // 0.855 ms per sweep; from 64 instances on the register-resident sweep wins: 0.884 against 0.859 ms)
  if (nt_env == 512 || (nt_env == 0 && wide)) {
    if (hybrid) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 512, true>), dim3((unsigned)batch), dim3(512), smem_h, st, B);
    else hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 512, false>), dim3((unsigned)batch), dim3(512), smem, st, B);
    return;
  }
  if (reg_env && nt_env == 0) {
    const size_t rs = RiccatiRegSmem<D>::BYTES;
    static bool reg_configured = false;
    if (!reg_configured) {
      (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_reg_kernel<D, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs);
      (void)hipFuncSetAttribute((const void*)ocp_riccati_backward_reg_kernel<D, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)rs);
      reg_configured = true;
    }
    if (hybrid) hipLaunchKernelGGL((ocp_riccati_backward_reg_kernel<D, true>), dim3((unsigned)batch), dim3(64), rs, st, B);
    else hipLaunchKernelGGL((ocp_riccati_backward_reg_kernel<D, false>), dim3((unsigned)batch), dim3(64), rs, st, B);
    return;
  }
  // A chain with switching constraints takes the HYBRID instantiation (same LDS footprint).
  if (hybrid) {
    // measured on the trotting / running chains: four instances per CU (batch 1024) are fastest with ONE wavefront each (512 registers,
    // nothing spilled, no wavefront waits for another: 2.02 vs 2.15 ms), two per CU with two wavefronts each (4.08 vs 4.43 ms)
    if (nt_env == 64 || (nt_env == 0 && batch >= 1024)) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 64, true>), dim3((unsigned)batch), dim3(64), smem_h, st, B);
    else if (nt_env == 256 || (nt_env == 0 && batch < 512)) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 256, true>), dim3((unsigned)batch), dim3(256), smem_h, st, B);
    else hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, true>), dim3((unsigned)batch), dim3(128), smem_h, st, B);
    return;
  }
  // Throughput mode (many instances): two wavefronts per instance, four instances per CU (39.8 kB
  // of LDS each) -- measured best at batch 1024 (2.48 ms vs 3.22 ms with one and 2.96 ms with four
  // wavefronts).  Latency mode (few instances): four wavefronts per instance.
  // IDOCP_RICCATI_NT={64,128,256} overrides the choice (tuning aid).
  if (nt_env == 128) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, false>), dim3((unsigned)batch), dim3(128), smem, st, B);
  else if (nt_env == 256) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 256, false>), dim3((unsigned)batch), dim3(256), smem, st, B);
  else if (nt_env == 64) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 64, false>), dim3((unsigned)batch), dim3(64), smem, st, B);
  else if (batch >= 512) hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 128, false>), dim3((unsigned)batch), dim3(128), smem, st, B);
  else hipLaunchKernelGGL((ocp_riccati_backward_kernel<D, 256, false>), dim3((unsigned)batch), dim3(256), smem, st, B);
}
template <typename D>
void OcpLaunch<D>::riccatiForward(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, hipStream_t st) {
  (void)M;
  hipLaunchKernelGGL((ocp_riccati_forward_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B, q0, v0);
}

template void OcpLaunch<LeggedDims<4, 3>>::riccatiBackward(const OcpBuffers&, long, int, bool, hipStream_t, bool);
template void OcpLaunch<LeggedDims<4, 3>>::riccatiForward(const OcpBuffers&, long, int, const double*, const double*, hipStream_t);

}  // namespace idocp_dev
