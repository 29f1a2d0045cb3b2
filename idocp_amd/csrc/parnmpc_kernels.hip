// ParNMPC: per-stage KKT inverse, coarse update and the correction sweeps.
//
// Replaces, for horizons without discrete events,
//   SplitKKTMatrixInverter::invert            (include/idocp/ocp/split_kkt_matrix_inverter.hxx:44-80)
//   SplitBackwardCorrection::coarseUpdate ... forwardCorrectionParallel, computeDirection
//                                             (include/idocp/ocp/split_backward_correction.hxx:30-155)
//   BackwardCorrectionSolver                  (src/ocp/backward_correction_solver.cpp:54-490)
// The stage linearisation itself is K5a + K5b<BWD> (ocp_rnea_kernel.hip, ocp_condense_kernel.hip).
//
// K9b  parnmpc_kkt_inverse_kernel   one 256-thread workgroup per stage: Qss^-1 (48 x 48) and (F Qss^-1 F^T)^-1 (36 x 36) by
//                                   Gauss-Jordan on 3 x 3 register tiles, the two column blocks of the KKT inverse the sweeps
//                                   need, the coarse direction and the coarse iterate s_new
// S5   parnmpc_backward_serial      one wavefront per instance walks the stages backwards (lmd, gmm corrections)
// K10a parnmpc_backward_parallel    one wavefront per stage (u, q, v corrections)
// S6   parnmpc_forward_serial       one wavefront per instance walks forwards (q, v corrections)
// K10b parnmpc_forward_parallel     one wavefront per stage (lmd, gmm, u corrections), aux_mat, Newton direction s_new - s
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <type_traits>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

template <typename D>
struct KktInvSmem {
  static constexpr int NX = D::NX, NU = D::NU, NQ_ = NU + NX;       // NQ_ = dim of (u, q, v)
  // Two matrix regions reused by lifetime (36 kB in all, four workgroups per CU):
  //   A : Q^-1 (NQ x NQ) -> FQ = F Q^-1 (NX x NQ) -> BR[:, NU:] (NQ x NX)
  //   B : F (NX x NQ)    -> S^-1 (NX x NX)        -> TR = S^-1 FQ (NX x NQ)
  // then the pivot rows / columns of the Gauss-Jordan steps and the vectors.
  static constexpr int A = 0, B = A + NQ_ * NQ_, PV = B + NX * NQ_,
                       R1 = PV + 2 * (2 * NQ_ + 2), R2 = R1 + NX, T1 = R2 + NQ_, W = T1 + NQ_, DIR = W + NQ_, TOTAL = DIR + NX + NQ_ + 4;
  static_assert(NX % 3 == 0 && NQ_ % 3 == 0 && NU % 3 == 0, "3 x 3 tiles");
  static_assert((NQ_ / 3) * (NQ_ / 3) <= 256, "one tile per thread");
  static_assert((NQ_ / 3) * (NX / 3) + NQ_ <= 256 && (NQ_ / 3) * (NX / 3) + NX <= 256, "side jobs next to the NX x NQ tiles");
};

template <typename D>
__global__ __launch_bounds__(256) void parnmpc_kkt_inverse_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = KktInvSmem<D>;
  constexpr int NV = D::NV, NQc = D::NQ, NX = D::NX, NU = D::NU, NQ = S::NQ_, NK = L::NK;
  constexpr int TQ = NQ / 3, TX = NX / 3, TU = NU / 3;        // tiles per side of Q (16), of S (12), of the u block (4)
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int tid = threadIdx.x, nt = 256;
  const long unit = blockIdx.x;
  const long b = unit / (M - 1);
  const int pos = (int)(unit - b * (M - 1));
  const OcpNode* __restrict__ nd = B.nodes + pos;
  if (parnmpcShape<L>(*nd).general) return;        // aux stages with switching rows and impulse stages: parnmpc_event_kernels.hip
  const bool last = P->has_terminal && (pos == M - 2);
  const double dt = nd->dt;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ kk = B.kkt + rec * L::KKT;
  const double* __restrict__ aux = B.aux + (b * B.NS + nd->next) * L::AUX;
  double* __restrict__ ki = B.kinv + rec * L::KINV;
  if (tid == 0) s_ok = 1;
  // ---- Qss in the order (u, q, v) (SplitBackwardCorrection::coarseUpdate: Qxx += aux_mat_next, Qvq = Qqv^T, Qux = Qxu^T):
  //      straight into the register tiles ----
  double qinv[3][3];
  const int qi = tid % TQ, qj = tid / TQ;
  const bool q_on = tid < TQ * TQ;
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) {
      const int r = 3 * qi + tr, c = 3 * qj + tc;
      double v = 0.0;
      if (q_on) {
        if (r < NU && c < NU) v = kk[L::K_QUU + r + NU * c];
        else if (r < NU) v = kk[L::K_QXU + (c - NU) + NX * r];
        else if (c < NU) v = kk[L::K_QXU + (r - NU) + NX * c];
        else {
          int rr = r - NU, cc = c - NU;
          const double ax = last ? 0.0 : aux[rr + NX * cc];
          if (rr > cc) { const int t = rr; rr = cc; cc = t; }      // K5 writes the triangle on and above the diagonal (Qxx symmetric, Qvq = Qqv^T)
          v = kk[L::K_QXX + L::xsym(rr, cc)] + ax;
        }
      }
      qinv[tr][tc] = v;
    }
  // ---- F = [0 Fqq Fqv; Fvu Fvq Fvv] (NX x NQ): backward Euler has Fqq = -I, Fqv = dt I outside the base blocks ----
  for (int e = tid; e < NX * NQ; e += nt) {
    const int c = e / NX, r = e - c * NX;
    double v = 0.0;
    if (r < NV) {
      if (c >= NU && c < NU + NV) { const int cq = c - NU; v = (r < 6 && cq < 6) ? kk[L::K_FQQ + r + 6 * cq] : ((r >= 6 && r == cq) ? -1.0 : 0.0); }
      else if (c >= NU + NV) { const int cv = c - NU - NV; v = (r < 6 && cv < 6) ? kk[L::K_FQV + r + 6 * cv] : ((r >= 6 && r == cv) ? dt : 0.0); }
    } else {
      const int rv = r - NV;
      if (c < NU) v = kk[L::K_FVU + rv + NV * c];
      else if (c < NU + NV) v = kk[L::K_FVQ + rv + NV * (c - NU)];
      else v = kk[L::K_FVV + rv + NV * (c - NU - NV)];
    }
    sm[S::B + e] = v;
  }
  // residual (split_kkt_residual.hxx): r1 = [Fq; Fv], r2 = [lu; lq; lv]
  if (tid < NX) sm[S::R1 + tid] = kk[L::K_FX + tid];
  if (tid >= 64 && tid < 64 + NU) sm[S::R2 + tid - 64] = kk[L::K_LU + tid - 64];
  if (tid >= 128 && tid < 128 + NX) sm[S::R2 + NU + tid - 128] = kk[L::K_LX + tid - 128];
  blockLdsSync();                                   // s_ok
  // ---- Q^-1 (llt_Q_.solve(I), split_kkt_matrix_inverter.hxx:55-58); the tiles stay in registers for BR ----
  gaussJordanTiles<NQ>(qinv, q_on, qi, qj, &sm[S::PV], &s_ok);
  if (q_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * qi + tr + NQ * (3 * qj + tc)] = qinv[tr][tc];
  }
  blockLdsSync();
  // ---- FQ = F Q^-1 (multiplyF, :60): TX x TQ tiles ; w = Q^-1 r2 ----
  const int fi = tid % TX, fj = tid / TX;            // tile of an NX x NQ matrix
  const bool f_on = tid < TX * TQ;
  double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (f_on) {
    tileMM<NQ>(acc, [&](int r, int m) { return sm[S::B + 3 * fi + r + NX * m]; },
               [&](int m, int c) { return sm[S::A + m + NQ * (3 * fj + c)]; });
  } else if (tid < TX * TQ + NQ) {
    const int r = tid - TX * TQ;
    double w = 0.0;
    for (int m = 0; m < NQ; ++m) w += sm[S::A + r + NQ * m] * sm[S::R2 + m];
    sm[S::W + r] = w;
  }
  blockLdsSync();                                   // Q^-1 is dead in LDS: FQ takes its place
  if (f_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * fi + tr + NX * (3 * fj + tc)] = acc[tr][tc];
  }
  blockLdsSync();
  // ---- S = F FQ^T (:61) into register tiles, S^-1 (:62-66) ----
  const bool s_on = tid < TX * TX;                   // (fi, fj) is then a tile of an NX x NX matrix as well
  double sinv[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (s_on)
    tileMM<NQ>(sinv, [&](int r, int m) { return sm[S::B + 3 * fi + r + NX * m]; },
               [&](int m, int c) { return sm[S::A + 3 * fj + c + NX * m]; });
  gaussJordanTiles<NX>(sinv, s_on, fi, fj, &sm[S::PV], &s_ok);      // its first barrier also ends the reads of F
  if (s_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::B + 3 * fi + tr + NX * (3 * fj + tc)] = sinv[tr][tc];
  }
  blockLdsSync();
  // ---- TR = S^-1 FQ  (= - topLeft * Jac_Qinv, :67-69) ; t1 = r1 - FQ r2 ; what else reads S^-1: TL = -S^-1 (the top
  //      of C0) and the first half of the top of the coarse direction, -S^-1 r1 ----
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) acc[tr][tc] = 0.0;
  if (f_on) {
    tileMM<NX>(acc, [&](int r, int m) { return sm[S::B + 3 * fi + r + NX * m]; },
               [&](int m, int c) { return sm[S::A + m + NX * (3 * fj + c)]; });
  } else if (tid < TX * TQ + NX) {
    const int r = tid - TX * TQ;
    double t = sm[S::R1 + r];
    for (int m = 0; m < NQ; ++m) t -= sm[S::A + r + NX * m] * sm[S::R2 + m];
    sm[S::T1 + r] = t;
    double d = 0.0;
    for (int m = 0; m < NX; ++m) d -= sm[S::B + r + NX * m] * sm[S::R1 + m];
    sm[S::DIR + r] = d;
  }
  for (int e = tid; e < NX * NX; e += nt) {
    const int c = e / NX, r = e - c * NX;
    ki[L::I_C0 + r + NK * c] = -sm[S::B + e];
  }
  blockLdsSync();                                   // S^-1 is dead: TR takes its place
  if (f_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::B + 3 * fi + tr + NX * (3 * fj + tc)] = acc[tr][tc];
  }
  blockLdsSync();
  // ---- BR[:, NU:] = Q^-1[:, NU:] - TR^T FQ[:, NU:] (:70-78) by the threads that hold those tiles of Q^-1 ;
  //      coarse direction = KKT_inv * [r1; r2]: top = -S^-1 r1 + TR r2 ; bottom = Q^-1 r2 + TR^T (r1 - FQ r2) ----
  const bool b_on = q_on && qj >= TU;
  if (b_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) acc[tr][tc] = 0.0;
    tileMM<NX>(acc, [&](int r, int m) { return sm[S::B + m + NX * (3 * qi + r)]; },
               [&](int m, int c) { return sm[S::A + m + NX * (3 * qj + c)]; });
  } else if (tid < NX) {
    double d = sm[S::DIR + tid];
    for (int m = 0; m < NQ; ++m) d += sm[S::B + tid + NX * m] * sm[S::R2 + m];
    sm[S::DIR + tid] = d;
  }
  if (tid >= 256 - NQ) {                             // after its tile: the bottom of the direction
    const int r = tid - (256 - NQ);
    double d = sm[S::W + r];
    for (int m = 0; m < NX; ++m) d += sm[S::B + m + NX * r] * sm[S::T1 + m];
    sm[S::DIR + NX + r] = d;
  }
  blockLdsSync();                                   // FQ is dead: BR[:, NU:] takes its place (NQ x NX)
  if (b_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * qi + tr + NQ * (3 * (qj - TU) + tc)] = qinv[tr][tc] - acc[tr][tc];
  }
  blockLdsSync();
  // ---- the column blocks of the inverse: C0 = [TL; TR^T] with TL = -S^-1 (above) ; C1 = [TR[:, NU:]; BR[:, NU:]] ----
  for (int e = tid; e < NK * NX; e += nt) {
    const int c = e / NK, r = e - c * NK;
    if (r < NX) {
      ki[L::I_C1 + e] = sm[S::B + r + NX * (NU + c)];
    } else {
      const int rq = r - NX;
      ki[L::I_C0 + e] = sm[S::B + c + NX * rq];
      ki[L::I_C1 + e] = sm[S::A + rq + NQ * c];
    }
  }
  // ---- s_new = s - direction (split_backward_correction.hxx:49-58) ----
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ sn = B.snew + rec * L::SNEW;
  const double* dir = &sm[S::DIR];          // dlmd dgmm | du dq dv
  if (tid < NV) {
    sn[L::N_LMD + tid] = s[L::S_LMD + tid] - dir[tid];
    sn[L::N_GMM + tid] = s[L::S_GMM + tid] - dir[NV + tid];
    sn[L::N_V + tid] = s[L::S_V + tid] - dir[NX + NU + NV + tid];
    if (tid >= 6) sn[L::N_Q + tid + 1] = s[L::S_Q + tid + 1] - dir[NX + NU + tid];
  }
  if (tid >= 64 && tid < 64 + NU) sn[L::N_U + tid - 64] = s[L::S_U + tid - 64] - dir[NX + tid - 64];
  if (tid == 128) {
    double qn[7];
    lieIntegrateBase(s + L::S_Q, dir + NX + NU, -1.0, qn);
    for (int k = 0; k < 7; ++k) sn[L::N_Q + k] = qn[k];
  }
  (void)NQc;
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1000 + pos;
}

// y[0:rows] = A[r0 : r0 + rows, 0 : NX] x  for a column block A (ld NK) in global memory; one lane per row
template <int NX>
__device__ __forceinline__ double blockRowDot(const double* __restrict__ A, int ld, int row, const double* x) {
  double acc = 0.0;
  for (int m = 0; m < NX; ++m) acc += A[row + ld * m] * x[m];
  return acc;
}

// S5: BackwardCorrectionSolver::backwardCorrectionSerial (backward_correction_solver.cpp:253-287)
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_backward_serial_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NX = D::NX;
  __shared__ double x[2][NX];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x, ln = lane < NX ? lane : 0;
  const long base = (long)blockIdx.x * B.NS;
  // a shard that does not end the horizon also corrects its last stage, against the imported first stage of its
  // right neighbour (held in the placeholder records)
  const int i_first = P->has_terminal ? M - 3 : M - 2;
  if (i_first < 0) return;
  // Lane r carries the corrected (lmd, gmm)[r] of the stage after in a register; the rows of the next stage's block
  // of the KKT inverse and its s / s_new entries are fetched while the current stage is multiplied.
  auto loadRows = [&](int i, double (&row)[NX]) {
    const ParnmpcShape sh = parnmpcShape<L>(B.nodes[i]);
    const double* __restrict__ A = B.kinv + (base + B.nodes[i].slot) * L::KINV + sh.c1 + ln;
    const int ld = sh.ld;
#pragma unroll
    for (int m = 0; m < NX; ++m) row[m] = A[ld * m];
  };
  double rowa[NX], rowb[NX];
  double cur, s_next, s_i, sn_i;
  {
    const long recn = base + B.nodes[i_first + 1].slot, rec = base + B.nodes[i_first].slot;
    cur = B.snew[recn * L::SNEW + L::N_LMD + ln];          // lmd then gmm: contiguous in both records
    s_next = B.sol[recn * L::SOL + L::S_LMD + ln];
    s_i = B.sol[rec * L::SOL + L::S_LMD + ln];
    sn_i = B.snew[rec * L::SNEW + L::N_LMD + ln];
    loadRows(i_first, rowa);
  }
  auto stage = [&](int i, const double (&row)[NX], double (&row_next)[NX], double* xb) {
    const long rec = base + B.nodes[i].slot;
    const double xl = cur - s_next;
    if (lane < NX) { xb[lane] = xl; B.xres[rec * L::XRES + lane] = xl; }
    const double s_here = s_i, sn_here = sn_i;
    if (i > 0) {
      const long recp = base + B.nodes[i - 1].slot;
      s_i = B.sol[recp * L::SOL + L::S_LMD + ln];
      sn_i = B.snew[recp * L::SNEW + L::N_LMD + ln];
      loadRows(i - 1, row_next);
    }
    waveLdsSync();
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < NX; ++m) acc += row[m] * xb[m];
    cur = sn_here - acc;
    s_next = s_here;
    if (lane < NX) B.snew[rec * L::SNEW + L::N_LMD + lane] = cur;
  };
  for (int i = i_first; i >= 0; i -= 2) {
    stage(i, rowa, rowb, x[0]);
    if (i >= 1) stage(i - 1, rowb, rowa, x[1]);
  }
}

#ifndef IDOCP_S5_BUFFERS
#define IDOCP_S5_BUFFERS 3
#endif
template <int N, int K = 0, typename F>
__device__ __forceinline__ void forEachConstS5(F f) {
  if constexpr (K < N) { f(std::integral_constant<int, K>{}); forEachConstS5<N, K + 1>(f); }
}
// S5, round 4: the same sweep with the rows of the KKT-inverse blocks requested TWO stages ahead.  A stage of the sweep is 36 fused multiply-adds
// behind an LDS exchange (0.3 us); what the one-stage-ahead version above waits for is memory: a stage's loads are issued while the
// stage in front of it is multiplied, i.e. one stage-time before they are needed, and a stage-time is shorter than the latency -- so the
// stage-time BECOMES the latency (1.3 us), and behind it sits a second, dependent latency: the node table entry that tells where the rows
// are.  Here the node table is turned into row offsets once, up front, in LDS (one lane per chain position), and three row buffers rotate
// (IDOCP_S5_BUFFERS; four to six measured: 0.26 ms like three -- what is left, 1.0 us per stage, is the ~250 instructions of a stage on
// the one wavefront a CU has of this kernel: 36 loads with their addresses, 36 LDS reads, 36 dependent multiply-adds).
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_backward_serial2_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NX = D::NX, MAXM = 1024;
  __shared__ double x[3][NX];
  __shared__ long s_off[MAXM];
  __shared__ int s_ld[MAXM], s_slot[MAXM];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x, ln = lane < NX ? lane : 0;
  const long base = (long)blockIdx.x * B.NS;
  const int i_first = __builtin_amdgcn_readfirstlane(P->has_terminal) ? M - 3 : M - 2;
  if (i_first < 0) return;
  for (int i = lane; i < M; i += 64) {
    const OcpNode nd = B.nodes[i];
    const ParnmpcShape sh = parnmpcShape<L>(nd);
    s_slot[i] = nd.slot; s_off[i] = (long)nd.slot * L::KINV + sh.c1; s_ld[i] = sh.ld;
  }
  waveLdsSync();
  const double* __restrict__ kinv0 = B.kinv + base * L::KINV + ln;
  auto loadRows = [&](int i, double (&row)[NX]) {
    const double* __restrict__ A = kinv0 + s_off[i];
    const int ld = s_ld[i];
#pragma unroll
    for (int m = 0; m < NX; ++m) row[m] = A[ld * m];
  };
  auto loadS = [&](int i, double& s_, double& sn_) {
    const long rec = base + s_slot[i];
    s_ = B.sol[rec * L::SOL + L::S_LMD + ln];            // lmd then gmm: contiguous in both records
    sn_ = B.snew[rec * L::SNEW + L::N_LMD + ln];
  };
  constexpr int NB = IDOCP_S5_BUFFERS;                  // row buffers: a stage's rows are requested NB - 1 stages ahead
  double r[NB][NX];
  double sv[NB], snv[NB];
  double cur, s_next;
  {
    const long recn = base + s_slot[i_first + 1];
    cur = B.snew[recn * L::SNEW + L::N_LMD + ln];
    s_next = B.sol[recn * L::SOL + L::S_LMD + ln];
#pragma unroll
    for (int k = 0; k < NB - 1; ++k) if (i_first - k >= 0) { loadS(i_first - k, sv[k], snv[k]); loadRows(i_first - k, r[k]); }
  }
  // one stage: multiplies with its rows (requested NB - 1 stages ago), requests the rows of stage i - (NB - 1) into the buffer that is free
  auto stage = [&](int i, auto use_, auto far_) {
    constexpr int use = decltype(use_)::value, far = decltype(far_)::value;
    double* xb = x[use % 3];
    const long rec = base + s_slot[i];
    const double xl = cur - s_next;
    if (lane < NX) { xb[lane] = xl; B.xres[rec * L::XRES + lane] = xl; }
    if (i >= NB - 1) { loadS(i - (NB - 1), sv[far], snv[far]); loadRows(i - (NB - 1), r[far]); }
    waveLdsSync();
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < NX; ++m) acc += r[use][m] * xb[m];      // (the order of the one-stage-ahead kernel: bit-identical results)
    cur = snv[use] - acc;
    s_next = sv[use];
    if (lane < NX) B.snew[rec * L::SNEW + L::N_LMD + lane] = cur;
  };
  for (int i = i_first; i >= 0; i -= NB) {
    forEachConstS5<NB>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if (i - k >= 0) stage(i - k, std::integral_constant<int, k>{}, std::integral_constant<int, (k + NB - 1) % NB>{});
    });
  }
}

// K10a: backwardCorrectionParallel (:288-318; split_backward_correction.hxx:96-108): stages 0 .. N-2
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_backward_parallel_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX;
  __shared__ double x[NX], dz[L::NKG - NX];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const int per = P->has_terminal ? M - 2 : M - 1;
  const long b = unit / per;
  const int pos = (int)(unit - b * per);
  const long rec = b * B.NS + B.nodes[pos].slot;
  const ParnmpcShape sh = parnmpcShape<L>(B.nodes[pos]);
  const int ni = sh.ni, nw = sh.nw;
  if (lane < NX) x[lane] = B.xres[rec * L::XRES + lane];
  waveLdsSync();
  if (lane < sh.nk - NX) dz[lane] = blockRowDot<NX>(B.kinv + rec * L::KINV + sh.c1, sh.ld, NX + lane, x);     // (dxi | dmu, du | df, dq, dv)
  waveLdsSync();
  double* __restrict__ sn = B.snew + rec * L::SNEW;
  if (lane < ni) sn[L::N_XI + lane] -= dz[lane];
  if (lane < nw) sn[L::N_U + lane] -= dz[ni + lane];
  if (lane < NV) {
    sn[L::N_V + lane] -= dz[ni + nw + NV + lane];
    if (lane >= 6) sn[L::N_Q + lane + 1] -= dz[ni + nw + lane];
  }
  if (lane == 32) {
    double qn[7];
    lieIntegrateBase(sn + L::N_Q, dz + ni + nw, -1.0, qn);
    for (int k = 0; k < 7; ++k) sn[L::N_Q + k] = qn[k];
  }
}

// S6: forwardCorrectionSerial (:319-352; split_backward_correction.hxx:109-120)
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_forward_serial_kernel(OcpBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX;
  constexpr int NQ = D::NQ, NS_ = NQ + NV;
  // cur = the corrected (q, v) of the stage before; sp / sn = s and the coarse s_new (q, v) of the stages i - 1 / i
  __shared__ double x[NX], dx[NX], cur[NS_], spL[NS_], snL[NS_];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long base = (long)blockIdx.x * B.NS;
  const long b = blockIdx.x;
  // a shard with a left neighbour also corrects its first stage: previous state = the imported (q, v) of the
  // neighbour's last stage (q0, v0), its corrected value = fwd_prev
  const int i0 = P->has_prev ? 0 : 1;
  {
    const double *cq, *cv;
    if (i0 == 0) { cq = B.fwd_prev + b * NS_; cv = cq + NQ; }
    else { const long recp = base + B.nodes[0].slot; cq = B.snew + recp * L::SNEW + L::N_Q; cv = B.snew + recp * L::SNEW + L::N_V; }
    if (lane < NQ) cur[lane] = cq[lane];
    if (lane < NV) cur[NQ + lane] = cv[lane];
  }
  for (int i = i0; i <= M - 2; ++i) {
    const long rec = base + B.nodes[i].slot;
    const double *spq, *spv;
    if (i == 0) { spq = q0 + b * NQ; spv = v0 + b * NV; }
    else { const long recp = base + B.nodes[i - 1].slot; spq = B.sol + recp * L::SOL + L::S_Q; spv = B.sol + recp * L::SOL + L::S_V; }
    double* __restrict__ sn = B.snew + rec * L::SNEW;
    // every load of the stage is issued here, ahead of the SE(3) log that only needs cur and s_prev: the rows of the
    // KKT-inverse block arrive while lane 0 works
    const double sp_q = lane < NQ ? spq[lane] : 0.0, sp_v = lane < NV ? spv[lane] : 0.0;
    const double sn_q = lane < NQ ? sn[L::N_Q + lane] : 0.0, sn_v = lane < NV ? sn[L::N_V + lane] : 0.0;
    double arow[NX];
    {
      const ParnmpcShape sh = parnmpcShape<L>(B.nodes[i]);
      const double* __restrict__ A = B.kinv + rec * L::KINV + L::I_C0 + (sh.nk - NX) + (lane < NX ? lane : 0);
      const int ld = sh.ld;
#pragma unroll
      for (int m = 0; m < NX; ++m) arow[m] = A[ld * m];
    }
    if (lane < NQ) { spL[lane] = sp_q; snL[lane] = sn_q; }
    if (lane < NV) { spL[NQ + lane] = sp_v; snL[NQ + lane] = sn_v; }
    waveLdsSync();
    if (lane == 0) {
      double R[9], p[3], d6[6];
      lieRelative(spL, cur, R, p);          // s_new_prev.q (-) s_prev.q
      lieLog6(R, p, d6);
      for (int k = 0; k < 6; ++k) x[k] = d6[k];
    }
    if (lane >= 6 && lane < NV) x[lane] = cur[lane + 1] - spL[lane + 1];
    if (lane < NV) x[NV + lane] = cur[NQ + lane] - spL[NQ + lane];
    waveLdsSync();
    if (lane < NX) {
      B.xres[rec * L::XRES + lane] = x[lane];
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < NX; ++m) acc += arow[m] * x[m];
      dx[lane] = acc;
    }
    waveLdsSync();
    if (lane < NV) {
      const double nv = snL[NQ + lane] - dx[NV + lane];
      sn[L::N_V + lane] = nv; cur[NQ + lane] = nv;
      if (lane >= 6) { const double nq = snL[lane + 1] - dx[lane]; sn[L::N_Q + lane + 1] = nq; cur[lane + 1] = nq; }
    }
    if (lane == 32) {
      double qn[7];
      lieIntegrateBase(snL, dx, -1.0, qn);
      for (int k = 0; k < 7; ++k) { sn[L::N_Q + k] = qn[k]; cur[k] = qn[k]; }
    }
    waveLdsSync();
  }
}

// S6, round 4: the same sweep with the stage's loads -- the 36 rows of its block of the KKT inverse, s of the stage before, the coarse s_new
// of the stage -- requested TWO stages ahead (as in S5 above; the node table as offsets in LDS).  In the kernel above they are issued at the
// top of the stage and hide behind the SE(3) log (1.1 us), but their latency is 2 us: the stage waits for them.  Same arithmetic, same order.
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_forward_serial2_kernel(OcpBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, MAXM = 1024;
  constexpr int NQ = D::NQ, NS_ = NQ + NV;
  __shared__ double x[NX], dx[NX], cur[NS_], spL[NS_], snL[NS_];
  __shared__ long s_off[MAXM];
  __shared__ int s_ld[MAXM], s_slot[MAXM];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const long base = b * B.NS;
  const int i0 = __builtin_amdgcn_readfirstlane(P->has_prev) ? 0 : 1;
  for (int i = lane; i < M; i += 64) {
    const OcpNode nd = B.nodes[i];
    const ParnmpcShape sh = parnmpcShape<L>(nd);
    s_slot[i] = nd.slot; s_off[i] = (long)nd.slot * L::KINV + L::I_C0 + (sh.nk - NX); s_ld[i] = sh.ld;
  }
  {
    const double *cq, *cv;
    if (i0 == 0) { cq = B.fwd_prev + b * NS_; cv = cq + NQ; }
    else { const long recp = base + B.nodes[0].slot; cq = B.snew + recp * L::SNEW + L::N_Q; cv = B.snew + recp * L::SNEW + L::N_V; }
    if (lane < NQ) cur[lane] = cq[lane];
    if (lane < NV) cur[NQ + lane] = cv[lane];
  }
  waveLdsSync();
  const int lq = lane < NQ ? lane : 0, lvv = lane < NV ? lane : 0, lx = lane < NX ? lane : 0;
  const double* __restrict__ kinv0 = B.kinv + base * L::KINV + lx;
  // everything stage i reads from memory: s of the stage before (the measured state in front of stage 0), its own coarse s_new, its rows
  auto fetch = [&](int i, double (&row)[NX], double& sp_q, double& sp_v, double& sn_q, double& sn_v) {
    const long rec = base + s_slot[i];
    const double *spq, *spv;
    if (i == 0) { spq = q0 + b * NQ; spv = v0 + b * NV; }
    else { const long recp = base + s_slot[i - 1]; spq = B.sol + recp * L::SOL + L::S_Q; spv = B.sol + recp * L::SOL + L::S_V; }
    sp_q = spq[lq]; sp_v = spv[lvv];
    sn_q = B.snew[rec * L::SNEW + L::N_Q + lq]; sn_v = B.snew[rec * L::SNEW + L::N_V + lvv];
    const double* __restrict__ A = kinv0 + s_off[i];
    const int ld = s_ld[i];
#pragma unroll
    for (int m = 0; m < NX; ++m) row[m] = A[ld * m];
  };
  constexpr int NB = 3;
  double r[NB][NX], spq_[NB], spv_[NB], snq_[NB], snv_[NB];
#pragma unroll
  for (int k = 0; k < NB - 1; ++k) if (i0 + k <= M - 2) fetch(i0 + k, r[k], spq_[k], spv_[k], snq_[k], snv_[k]);
  auto stage = [&](int i, auto use_, auto far_) {
    constexpr int use = decltype(use_)::value, far = decltype(far_)::value;
    const long rec = base + s_slot[i];
    double* __restrict__ sn = B.snew + rec * L::SNEW;
    if (i + NB - 1 <= M - 2) fetch(i + NB - 1, r[far], spq_[far], spv_[far], snq_[far], snv_[far]);
    if (lane < NQ) { spL[lane] = spq_[use]; snL[lane] = snq_[use]; }
    if (lane < NV) { spL[NQ + lane] = spv_[use]; snL[NQ + lane] = snv_[use]; }
    waveLdsSync();
    if (lane == 0) {
      double R[9], p[3], d6[6];
      lieRelative(spL, cur, R, p);          // s_new_prev.q (-) s_prev.q
      lieLog6(R, p, d6);
      for (int k = 0; k < 6; ++k) x[k] = d6[k];
    }
    if (lane >= 6 && lane < NV) x[lane] = cur[lane + 1] - spL[lane + 1];
    if (lane < NV) x[NV + lane] = cur[NQ + lane] - spL[NQ + lane];
    waveLdsSync();
    if (lane < NX) {
      B.xres[rec * L::XRES + lane] = x[lane];
      double acc = 0.0;
#pragma unroll
      for (int m = 0; m < NX; ++m) acc += r[use][m] * x[m];
      dx[lane] = acc;
    }
    waveLdsSync();
    if (lane < NV) {
      const double nv = snL[NQ + lane] - dx[NV + lane];
      sn[L::N_V + lane] = nv; cur[NQ + lane] = nv;
      if (lane >= 6) { const double nq = snL[lane + 1] - dx[lane]; sn[L::N_Q + lane + 1] = nq; cur[lane + 1] = nq; }
    }
    if (lane == 32) {
      double qn[7];
      lieIntegrateBase(snL, dx, -1.0, qn);
      for (int k = 0; k < 7; ++k) { sn[L::N_Q + k] = qn[k]; cur[k] = qn[k]; }
    }
    waveLdsSync();
  };
  for (int i = i0; i <= M - 2; i += NB) {
    forEachConstS5<NB>([&](auto k_) {
      constexpr int k = decltype(k_)::value;
      if (i + k <= M - 2) stage(i + k, std::integral_constant<int, k>{}, std::integral_constant<int, (k + NB - 1) % NB>{});
    });
  }
}

// K10b: forwardCorrectionParallel (:353-470; split_backward_correction.hxx:121-155): lmd / gmm / u corrections (stages > 0),
// aux_mat = - KKT_inv.topLeftCorner(nx, nx), and the Newton direction d = s_new - s written into the dir record
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_forward_parallel_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU;
  __shared__ double x[NX], dh[L::NKG - NX];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const long b = unit / (M - 1);
  const int pos = (int)(unit - b * (M - 1));
  const long rec = b * B.NS + B.nodes[pos].slot;
  const ParnmpcShape sh = parnmpcShape<L>(B.nodes[pos]);
  const int ni = sh.ni, nw = sh.nw, ld = sh.ld;
  const double* __restrict__ ki = B.kinv + rec * L::KINV;
  double* __restrict__ sn = B.snew + rec * L::SNEW;
  if (pos > 0 || P->has_prev) {
    if (lane < NX) x[lane] = B.xres[rec * L::XRES + lane];
    waveLdsSync();
    if (lane < sh.nk - NX) dh[lane] = blockRowDot<NX>(ki + L::I_C0, ld, lane, x);        // (dlmd, dgmm, dxi | dmu, du | df)
    waveLdsSync();
    if (lane < NX) sn[L::N_LMD + lane] -= dh[lane];
    if (lane < ni) sn[L::N_XI + lane] -= dh[NX + lane];
    if (lane < nw) sn[L::N_U + lane] -= dh[NX + ni + lane];
  }
  double* __restrict__ aux = B.aux + rec * L::AUX;
  for (int e = lane; e < NX * NX; e += 64) { const int c = e / NX, r = e - c * NX; aux[e] = -ki[L::I_C0 + r + ld * c]; }
  waveLdsSync();
  // computeDirection (split_backward_correction.hxx:141-154)
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ dd = B.dir + rec * L::DIR;
  if (lane < NV) {
    dd[L::D_LMD + lane] = sn[L::N_LMD + lane] - s[L::S_LMD + lane];
    dd[L::D_GMM + lane] = sn[L::N_GMM + lane] - s[L::S_GMM + lane];
    dd[L::D_V + lane] = sn[L::N_V + lane] - s[L::S_V + lane];
    if (lane >= 6) dd[L::D_Q + lane] = sn[L::N_Q + lane + 1] - s[L::S_Q + lane + 1];
  }
  const OcpNode* __restrict__ nd = B.nodes + pos;
  if (!sh.impulse) {
    if (lane < NU) dd[L::D_U + lane] = sn[L::N_U + lane] - s[L::S_U + lane];
    if (lane < ni) dd[L::D_XI + lane] = sn[L::N_XI + lane] - s[L::S_XI + lane];             // aux stage: dxi (:147-150)
  } else if (lane < D::NC && nd->active[lane]) {
    // impulse stage (impulse_split_backward_correction.hxx:113-125): dmu, df, packed rows -> contact slots
    const int row = nd->row_of[lane];
    for (int k = 0; k < 3; ++k) {
      const double df = sn[L::N_U + row + k] - s[L::S_F + 3 * lane + k];
      dd[L::D_F + 3 * lane + k] = df;
      dd[L::D_U + row + k] = df;                       // packed rows: what K6 / K7 multiply with Fvf / Qdvf
      dd[L::D_MU + 3 * lane + k] = sn[L::N_XI + row + k] - s[L::S_MU + 3 * lane + k];
    }
  }
  if (lane == 32) {
    double R[9], p[3], d6[6];
    lieRelative(s + L::S_Q, sn + L::N_Q, R, p);            // s_new.q (-) s.q
    lieLog6(R, p, d6);
    for (int k = 0; k < 6; ++k) dd[L::D_Q + k] = d6[k];
  }
}

// BackwardCorrectionSolver::initAuxMat (:54-92): every aux_mat = the terminal cost Hessian at the last stage
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_init_aux_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NQ = D::NQ;
  __shared__ double Jq[36];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long b = blockIdx.x;
  const long base = b * B.NS;
  if (lane == 0) {
    double R[9], p[3];
    lieRelative(B.q_ref + (long)(M - 2) * NQ, B.sol + (base + B.nodes[M - 2].slot) * L::SOL + L::S_Q, R, p);
    lieJlog6(R, p, Jq);
  }
  waveLdsSync();
  for (int pos = 0; pos < M - 1; ++pos) {
    double* __restrict__ aux = B.aux + (base + B.nodes[pos].slot) * L::AUX;
    for (int e = lane; e < NX * NX; e += 64) {
      const int c = e / NX, r = e - c * NX;
      double v = 0.0;
      if (r < 6 && c < 6) { for (int m = 0; m < 6; ++m) v += Jq[m + 6 * r] * P->qf_weight[m] * Jq[m + 6 * c]; }
      else if (r == c) v = r < NV ? P->qf_weight[r] : P->vf_weight[r - NV];
      aux[e] = v;
    }
  }
}

// Pack / unpack the halo of a shard (idocp_amd/parnmpc_dist.py): buf[batch][size]
//   0 state_last  : export (q, v) of the last stage            | import -> q0, v0 (the state in front of the first stage)
//   1 costate_first: export (lmd, gmm, q) of the first stage   | import -> placeholder sol record
//   2 aux_first   : export aux_mat of the first stage          | import -> placeholder aux record
//   3 bwd_first   : export corrected (lmd, gmm) of stage 0     | import -> placeholder snew record
//   4 fwd_last    : export corrected (q, v) of the last stage  | import -> fwd_prev
//   5 aux_all     : export aux_mat of stage 0                  | import -> every aux record (initBackwardCorrection)
template <typename D>
__global__ void parnmpc_halo_kernel(OcpBuffers B, int kind, int do_import, double* __restrict__ buf, double* __restrict__ q0,
                                    double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long b = blockIdx.x;
  const long base = b * B.NS;
  const long first = base + B.nodes[0].slot, lastr = base + B.nodes[M - 2].slot, ph = base + B.nodes[M - 1].slot;
  const int size = (kind == 0 || kind == 4) ? NQ + NV : (kind == 1 ? 2 * NV + NQ : (kind == 3 ? 2 * NV : NX * NX));
  double* __restrict__ x = buf + b * size;
  for (int e = threadIdx.x; e < size; e += blockDim.x) {
    if (kind == 0) {
      if (do_import) { if (e < NQ) q0[b * NQ + e] = x[e]; else v0[b * NV + e - NQ] = x[e]; }
      else x[e] = e < NQ ? B.sol[lastr * L::SOL + L::S_Q + e] : B.sol[lastr * L::SOL + L::S_V + e - NQ];
    } else if (kind == 1) {
      const int off = e < 2 * NV ? L::S_LMD + e : L::S_Q + e - 2 * NV;
      if (do_import) B.sol[ph * L::SOL + off] = x[e]; else x[e] = B.sol[first * L::SOL + off];
    } else if (kind == 2) {
      if (do_import) B.aux[ph * L::AUX + e] = x[e]; else x[e] = B.aux[first * L::AUX + e];
    } else if (kind == 3) {
      if (do_import) B.snew[ph * L::SNEW + L::N_LMD + e] = x[e]; else x[e] = B.snew[first * L::SNEW + L::N_LMD + e];
    } else if (kind == 4) {
      if (do_import) B.fwd_prev[b * (NQ + NV) + e] = x[e];
      else x[e] = e < NQ ? B.snew[lastr * L::SNEW + L::N_Q + e] : B.snew[lastr * L::SNEW + L::N_V + e - NQ];
    } else {
      if (do_import) { for (int pos = 0; pos < M; ++pos) B.aux[(base + B.nodes[pos].slot) * L::AUX + e] = x[e]; }
      else x[e] = B.aux[first * L::AUX + e];
    }
  }
}

template <typename D>
void OcpLaunch<D>::parnmpcHalo(const OcpBuffers& B, long batch, int kind, bool do_import, double* buf, double* q0, double* v0, hipStream_t st) {
  hipLaunchKernelGGL((parnmpc_halo_kernel<D>), dim3((unsigned)batch), dim3(128), 0, st, B, kind, do_import ? 1 : 0, buf, q0, v0);
}

template <typename D>
void OcpLaunch<D>::parnmpcInverse(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  // K9w (one wavefront per stage on the matrix cores, parnmpc_kkt_wave_kernel.hip) unless IDOCP_K9_WAVE=0 asks for the round-1 kernel below
  static const bool wave = [] { const char* e = getenv("IDOCP_K9_WAVE"); return !(e && e[0] == '0'); }();
  if (wave) { parnmpcInverseWave(B, batch, M, st); return; }
  const size_t smem = KktInvSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)parnmpc_kkt_inverse_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  hipLaunchKernelGGL((parnmpc_kkt_inverse_kernel<D>), dim3((unsigned)(batch * (M - 1))), dim3(256), smem, st, B);
}
template <typename D>
void OcpLaunch<D>::parnmpcPhase(int phase, const OcpBuffers& B, long batch, int M, bool has_terminal, const double* q0, const double* v0,
                                hipStream_t st) {
  const int nbp = has_terminal ? M - 2 : M - 1;      // stages of the backward parallel correction
  switch (phase) {
    case 0: {
      // S5 with the rows requested two stages ahead (round 4) unless the chain is longer than its LDS tables or IDOCP_S5_DEPTH=1 asks for the old one
      static const bool deep = [] { const char* e = getenv("IDOCP_S5_DEPTH"); return !(e && e[0] == '1'); }();
      if (deep && M <= 1024) hipLaunchKernelGGL((parnmpc_backward_serial2_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B);
      else hipLaunchKernelGGL((parnmpc_backward_serial_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B);
      break;
    }
    case 1: if (nbp > 0) hipLaunchKernelGGL((parnmpc_backward_parallel_kernel<D>), dim3((unsigned)(batch * nbp)), dim3(64), 0, st, B); break;
    case 2: {
      static const bool deep = [] { const char* e = getenv("IDOCP_S6_DEPTH"); return !(e && e[0] == '1'); }();
      if (deep && M <= 1024) hipLaunchKernelGGL((parnmpc_forward_serial2_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B, q0, v0);
      else hipLaunchKernelGGL((parnmpc_forward_serial_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B, q0, v0);
      break;
    }
    case 3: hipLaunchKernelGGL((parnmpc_forward_parallel_kernel<D>), dim3((unsigned)(batch * (M - 1))), dim3(64), 0, st, B); break;
    default: hipLaunchKernelGGL((parnmpc_init_aux_kernel<D>), dim3((unsigned)batch), dim3(64), 0, st, B); break;
  }
}

template void OcpLaunch<LeggedDims<4, 3>>::parnmpcInverse(const OcpBuffers&, long, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::parnmpcHalo(const OcpBuffers&, long, int, bool, double*, double*, double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::parnmpcPhase(int, const OcpBuffers&, long, int, bool, const double*, const double*, hipStream_t);

}  // namespace idocp_dev
