// ParNMPC: the event stages of a horizon with discrete events (ParNMPCDiscretizer chain, see ocp_capi.hip).
//
//   lift stage      an ordinary backward-Euler stage with its own time step: K5a / K5b<BWD> / K9b as they are
//   aux stage       ordinary stage + the switching constraint P(q) = 0 of the impulse that follows (switching_constraint.hxx:
//                   8-21): K5s produces P, Pq; K5b<BWD> adds Pq^T xi; its KKT matrix has dimi more constraint rows
//                   (SplitKKTMatrixInverter::invert with Pq, split_kkt_matrix_inverter.hxx:110-166)  -> K9g below
//   impulse stage   ImpulseSplitParNMPC::linearizeOCP (impulse_split_parnmpc.hxx:33-60): impulse cost + impulse friction cone,
//                   linearizeImpulseBackwardEuler / condenseImpulseBackwardEuler (impulse_state_equation.hxx:59-111),
//                   ImpulseDynamicsBackwardEuler::linearizeImpulseDynamics / condenseImpulseDynamics
//                   (impulse_dynamics_backward_euler.hxx:20-97)                                  -> K9i below
//                   and its KKT matrix in (lmd, gmm, mu | f, q, v) (ImpulseSplitKKTMatrixInverter::invert,
//                   impulse_split_kkt_matrix_inverter.hxx:34-80)                                  -> K9g below
//
// K9i writes the impulse stage into the SAME records as a regular stage, re-using their blocks
//   kkt:  K_QXX = Qxx, K_QXU = [Qqf; 0] (NX x ni), K_QUU = Qff (ni x ni), K_FQQ = Fqq6, K_FVQ = Fvq, K_FVV = -I,
//         K_FVU = Fvf (NV x ni), K_LX = (lq, lv), K_LU = lf, K_FX = (Fq, Fv)
//   swc:  W_P = V (contact-velocity residual), W_PHIX = [Vq Vv]
//   exp:  what K6 / K7 need for computeCondensed{Primal,Dual}Direction of the impulse stage
// so that K9g treats aux and impulse stages alike:  J = [F ; C], C = the W_PHIX rows on the (q, v) columns,
// Q over (w, q, v) with w = u (aux) or f (impulse).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

namespace {


}  // namespace

// ---------------------------------------------------------------------------------------------------- K9i ----
// MERIT (with RESIDUAL): the line search's evaluation of a trial iterate on the impulse stage -- ImpulseSplitParNMPC::stageCost and
// constraintViolation (impulse_split_parnmpc.hxx:147-194): B.sol is the trial iterate, the lin / lie records have been rebuilt on it.
template <typename D, bool RESIDUAL, bool MERIT = false>
__global__ __launch_bounds__(256) void parnmpc_impulse_condense_kernel(OcpBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NF = D::NF, NVF = D::NVF, NU = D::NU, NC = D::NC;
  __shared__ double Mi[NV * NV], Fvq[NV * NV], Fvf[NV * NF], Vq[NF * NV], Vv[NF * NV], dIq[NV * NV];
  __shared__ double lq[NV], lv[NV], ldv[NV], lf[NF], Fq[NV], Fv[NV], ImD[NV], Vr[NF], MiI[NV], qdd[NV], mu_p[NF], f_p[NF], err[256];
  __shared__ double Qff[NF * NF], Jq6[36], Fqq6[36], FqqI[36], FqqP[36], Fq6[6];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int tid = threadIdx.x;
  const int n_imp = gridDim.y;
  (void)n_imp;
  const long b = blockIdx.x;
  const int pos = B.impulse_pos[blockIdx.y];
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const int ni = nd->dimf;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ sn = B.sol + (b * B.NS + nd->next) * L::SOL;
  const double* __restrict__ sp = nd->prev >= 0 ? B.sol + (b * B.NS + nd->prev) * L::SOL : nullptr;
  const double* __restrict__ lin = B.lin + rec * L::LIN;
  const double* __restrict__ zz = B.lie + rec * L::LIE;
  const double* __restrict__ qref = B.q_ref + (long)pos * NQ;
  const double* __restrict__ slack = B.slack + rec * L::CON;
  const double* __restrict__ dual = B.dual + rec * L::CON;
  const bool cone = P->use_impulse_friction_cone != 0;
  const double vref_on = nd->vref_on;
  if (tid == 0) s_ok = 1;
  // ---- load: dImD/dq, M, Vq, Vv, ImD, V (lin record of the impulse pass of K5a), the Lie-group blocks ----
  for (int e = tid; e < NV * NV; e += 256) { const int c = e / NV, r = e - c * NV; dIq[e] = lin[L::L_DIDC + r + NVF * c]; Mi[e] = lin[L::L_M + e]; }
  for (int e = tid; e < ni * NV; e += 256) {
    const int c = e / ni, j = e - c * ni;
    Vq[j + NF * c] = lin[L::L_DIDC + (NV + j) + NVF * c];
    Vv[j + NF * c] = lin[L::L_DIDC + (NV + j) + NVF * (NV + c)];
  }
  if (tid < NV) ImD[tid] = lin[L::L_IDC + tid];
  if (tid < ni) Vr[tid] = lin[L::L_IDC + NV + tid];
  if (tid < 36) { Jq6[tid] = zz[L::Z_JQ + tid]; FqqP[tid] = zz[L::Z_FQQP + tid]; FqqI[tid] = zz[L::Z_FQQI + tid]; Fqq6[tid] = zz[L::Z_FQQ + tid]; }
  if (tid < 6) { Fq6[tid] = zz[L::Z_FQ6 + tid]; qdd[tid] = zz[L::Z_QDIFF + tid]; }
  if (tid < NC && nd->active[tid]) for (int x = 0; x < 3; ++x) { mu_p[nd->row_of[tid] + x] = s[L::S_MU + 3 * tid + x]; f_p[nd->row_of[tid] + x] = s[L::S_F + 3 * tid + x]; }
  for (int e = tid; e < NF * NF; e += 256) Qff[e] = 0.0;
  blockLdsSync();
  // ---- impulse cost, state equation, multipliers of the impulse dynamics and of the velocity constraint: gradients ----
  if (tid < NV) {
    const int r = tid;
    const double q_r = s[L::S_Q + (r < 6 ? r : r + 1)];
    (void)q_r;
    const double vr = s[L::S_V + r], dvr = s[L::S_A + r];
    const double lmd = s[L::S_LMD + r], gmm = s[L::S_GMM + r], lmdn = sn[L::S_LMD + r], gmmn = sn[L::S_GMM + r];
    const double vpr = sp ? sp[L::S_V + r] : v0[b * NV + r];
    double a_q, a_v, a_dv;
    if (r < 6) {
      a_q = 0.0;
      for (int m2 = 0; m2 < 6; ++m2) a_q += Jq6[m2 + 6 * r] * P->qi_weight[m2] * qdd[m2];
      double t1 = 0.0;
      for (int m2 = 0; m2 < 6; ++m2) t1 += Fqq6[m2 + 6 * r] * sn[L::S_LMD + m2] + FqqP[m2 + 6 * r] * s[L::S_LMD + m2];
      a_q += t1;
      Fq[r] = Fq6[r];
    } else {
      const double qpr = sp ? sp[L::S_Q + r + 1] : q0[b * NQ + r + 1];
      a_q = P->qi_weight[r] * (s[L::S_Q + r + 1] - qref[r + 1]) + lmdn - lmd;
      Fq[r] = qpr - s[L::S_Q + r + 1];
    }
    a_v = P->vi_weight[r] * (vr - vref_on * P->v_ref[r]) - gmm + gmmn;
    a_dv = P->dvi_weight[r] * dvr + gmm;
    Fv[r] = vpr - vr + dvr;
    // + dImD/dq^T beta, + M^T beta, + Vq^T mu, + Vv^T mu
    double dq = 0.0, dm = 0.0;
    for (int m = 0; m < NV; ++m) { dq += dIq[m + NV * r] * s[L::S_BETA + m]; dm += Mi[m + NV * r] * s[L::S_BETA + m]; }
    a_q += dq; a_dv += dm;
    if (B.ext) a_q += B.ext[rec * L::EXT + L::X_LQ + r];      // task-space cost with its impulse weights (ocp_ext_kernel.hip)
    double vq = 0.0, vv = 0.0;
    for (int j = 0; j < ni; ++j) { vq += Vq[j + NF * r] * mu_p[j]; vv += Vv[j + NF * r] * mu_p[j]; }
    a_q += vq; a_v += vv;
    lq[r] = a_q; lv[r] = a_v; ldv[r] = a_dv;
  }
  // lf of the contacts that touch down: impulse force cost + cone + (- Vv beta)
  double e_ipm = 0.0, m_cost = 0.0, m_viol = 0.0;
  if (tid >= 64 && tid < 64 + NC && nd->active[tid - 64]) {
    const int c = tid - 64, row = nd->row_of[c];
    // (Linearized)ImpulseFrictionCone, row by row (coneRow, ocp_device.hpp)
    const int ck = P->impulse_cone_kind, nr = coneRows(ck);
    const double fc[3] = {s[L::S_F + 3 * c], s[L::S_F + 3 * c + 1], s[L::S_F + 3 * c + 2]};
    double rr[5], ddv[5], Jr[5][3];
    for (int r = 0; r < 5; ++r) {
      if (r >= nr) { rr[r] = 0.0; ddv[r] = 0.0; Jr[r][0] = Jr[r][1] = Jr[r][2] = 0.0; continue; }
      const int idx = L::C_FRIC + 5 * c + r;
      const double g = coneRow(ck, P->mu, r, fc, Jr[r]);
      const double sl = slack[idx], du = dual[idx];
      const double res = g + sl, duality = sl * du - P->barrier;
      if (RESIDUAL) { rr[r] = du; ddv[r] = 0.0; e_ipm += res * res + duality * duality; if (MERIT && cone) m_viol += fabs(res); }
      else { rr[r] = du + (du * res - duality) / sl; ddv[r] = du / sl; }
    }
    for (int x = 0; x < 3; ++x) {
      double a = P->fi_weight[c][x] * (s[L::S_F + 3 * c + x] - P->fi_ref[c][x]);
      if (MERIT) m_cost += 0.5 * a * (s[L::S_F + 3 * c + x] - P->fi_ref[c][x]);      // ImpulseForceCost::computeImpulseCost
      if (cone) for (int r = 0; r < 5; ++r) a += Jr[r][x] * rr[r];
      double vb = 0.0;
      for (int m = 0; m < NV; ++m) vb += Vv[(row + x) + NF * m] * s[L::S_BETA + m];
      lf[row + x] = a - vb;
      if (!RESIDUAL) {
        for (int y = 0; y < 3; ++y) {
          double h = (x == y) ? P->fi_weight[c][x] : 0.0;
          if (cone) for (int r = 0; r < 5; ++r) h += Jr[r][x] * ddv[r] * Jr[r][y];
          Qff[(row + x) + NF * (row + y)] = h;
        }
      }
    }
    if (!cone) e_ipm = 0.0;
  }
  blockLdsSync();
  if (MERIT) {
    if (tid < NV) {
      const int r = tid;
      const double qd = r < 6 ? qdd[r] : s[L::S_Q + r + 1] - qref[r + 1];
      const double dvr = s[L::S_V + r] - vref_on * P->v_ref[r], ddv = s[L::S_A + r];
      m_cost += 0.5 * (P->qi_weight[r] * qd * qd + P->vi_weight[r] * dvr * dvr + P->dvi_weight[r] * ddv * ddv);
      m_viol += fabs(Fq[r]) + fabs(Fv[r]) + fabs(ImD[r]);
    }
    if (tid < ni) m_viol += fabs(Vr[tid]);
    if (tid == 0 && B.ext) m_cost += B.ext[rec * L::EXT + L::X_COST];
    err[tid] = m_cost;
    blockLdsSync();
    if (tid == 0) { double acc = 0.0; for (int t = 0; t < 256; ++t) acc += err[t]; B.merit_stage[rec * 4] = acc; }
    blockLdsSync();
    err[tid] = m_viol;
    blockLdsSync();
    if (tid == 0) { double acc = 0.0; for (int t = 0; t < 256; ++t) acc += err[t]; B.merit_stage[rec * 4 + 1] = acc; }
    return;
  }
  if (RESIDUAL) {
    // ImpulseSplitParNMPC::squaredNormKKTResidual (impulse_split_parnmpc.hxx:114-124)
    double e = e_ipm;
    if (tid < NV) e += lq[tid] * lq[tid] + lv[tid] * lv[tid] + ldv[tid] * ldv[tid] + Fq[tid] * Fq[tid] + Fv[tid] * Fv[tid] + ImD[tid] * ImD[tid];
    if (tid < ni) e += lf[tid] * lf[tid] + Vr[tid] * Vr[tid];
    err[tid] = e;
    blockLdsSync();
    if (tid == 0) { double acc = 0.0; for (int t = 0; t < 256; ++t) acc += err[t]; B.err_stage[rec] = acc; }
    return;
  }
  double* __restrict__ kk = B.kkt + rec * L::KKT;
  double* __restrict__ ee = B.exp + rec * L::EXP;
  double* __restrict__ W = B.swc + rec * L::SWC;
  // ---- condenseImpulseBackwardEuler: Fqq = Fqq_inv Fqq(q_prev, q), Fq.head(6) = Fqq_inv Fq.head(6) ----
  if (tid < 36) {
    const int c = tid / 6, r = tid - 6 * c;
    double acc = 0.0;
    for (int m = 0; m < 6; ++m) acc += FqqI[r + 6 * m] * FqqP[m + 6 * c];
    kk[L::K_FQQ + tid] = acc;
    kk[L::K_FQV + tid] = 0.0;
    ee[L::E_FQQPI + tid] = FqqI[tid];
  } else if (tid >= 64 && tid < 70) {
    const int r = tid - 64;
    double acc = 0.0;
    for (int m = 0; m < 6; ++m) acc += FqqI[r + 6 * m] * Fq6[m];
    Fq[r] = acc;
  }
  // ---- Minv (Robot::computeMinv), Fvq = -Minv dImD/dq, Fvf = Minv Vv^T, Minv ImD ----
  blockLdsSync();
  if (tid < 64) spdInverseRows<NV>(Mi, NV, NV, tid, &s_ok);
  blockLdsSync();
  for (int e = tid; e < NV * NV; e += 256) {
    const int c = e / NV, r = e - c * NV;
    double acc = 0.0;
    for (int m = 0; m < NV; ++m) acc += Mi[r + NV * m] * dIq[m + NV * c];
    Fvq[e] = -acc;
  }
  for (int e = tid; e < NV * ni; e += 256) {
    const int j = e / NV, r = e - j * NV;
    double acc = 0.0;
    for (int m = 0; m < NV; ++m) acc += Mi[r + NV * m] * Vv[j + NF * m];
    Fvf[r + NV * j] = acc;
  }
  if (tid >= 192 && tid < 192 + NV) {
    const int r = tid - 192;
    double acc = 0.0;
    for (int m = 0; m < NV; ++m) acc += Mi[r + NV * m] * ImD[m];
    MiI[r] = acc;
  }
  blockLdsSync();
  if (tid < NV) ldv[tid] -= P->dvi_weight[tid] * MiI[tid];            // data.ldv = ldv - Qdvdv Minv ImD
  blockLdsSync();
  // ---- condensed blocks (Qdvq = Qdvdv Fvq, Qdvf = Qdvdv Fvf) ----
  // Qxx = [Qqq 0; 0 Qvv]
  for (int e = tid; e < NX * NX; e += 256) {
    const int c = e / NX, r = e - c * NX;
    double v = 0.0;
    if (r < NV && c < NV) {
      if (r < 6 && c < 6) { for (int m2 = 0; m2 < 6; ++m2) v += Jq6[m2 + 6 * r] * P->qi_weight[m2] * Jq6[m2 + 6 * c]; }
      else if (r == c) v = P->qi_weight[r];
      double acc = 0.0;
      for (int m = 0; m < NV; ++m) acc += Fvq[m + NV * r] * (P->dvi_weight[m] * Fvq[m + NV * c]);
      v += acc;
      if (B.ext) {                                   // JJ^T diag(w_i) JJ of the task-space cost (the rows ocp_ext_kernel left)
        const double* __restrict__ xx = B.ext + rec * L::EXT;
        for (int k = 0; k < 6 * L::NT; ++k) v += xx[L::X_TW + k] * xx[L::X_TJ + k * NV + r] * xx[L::X_TJ + k * NV + c];      // (unused component slots: zero weights)
      }
    } else if (r == c) {
      v = P->vi_weight[r - NV];
    }
    if (r <= c) kk[L::K_QXX + L::xsym(r, c)] = v;
  }
  // Qxf = [Qqf; 0] in the place of Qxu, Qff in the place of Quu, Fvf in the place of Fvu
  for (int e = tid; e < NX * NU; e += 256) {
    const int j = e / NX, r = e - j * NX;
    double v = 0.0;
    if (r < NV && j < ni) for (int m = 0; m < NV; ++m) v += Fvq[m + NV * r] * (P->dvi_weight[m] * Fvf[m + NV * j]);
    kk[L::K_QXU + e] = v;
  }
  for (int e = tid; e < NU * NU; e += 256) {
    const int c = e / NU, r = e - c * NU;
    double v = 0.0;
    if (r < ni && c < ni) {
      v = Qff[r + NF * c];
      double acc = 0.0;
      for (int m = 0; m < NV; ++m) acc += Fvf[m + NV * r] * (P->dvi_weight[m] * Fvf[m + NV * c]);
      v += acc;
    } else if (r == c) v = 1.0;
    kk[L::K_QUU + e] = v;
  }
  for (int e = tid; e < NV * NV; e += 256) {
    const int c = e / NV, r = e - c * NV;
    kk[L::K_FVQ + e] = Fvq[e];
    kk[L::K_FVV + e] = (r == c) ? -1.0 : 0.0;
    if (r >= c) ee[L::E_MJ + r * (r + 1) / 2 + c] = Mi[e];      // Minv is symmetric: the exp record keeps the lower triangle (ocp_device.hpp)
  }
  for (int e = tid; e < NV * NU; e += 256) {
    const int j = e / NV;
    kk[L::K_FVU + e] = j < ni ? Fvf[e] : 0.0;
  }
  // gradients, residuals
  if (tid < NV) {
    const int r = tid;
    double aq = 0.0;
    for (int m = 0; m < NV; ++m) aq += Fvq[m + NV * r] * ldv[m];
    kk[L::K_LX + r] = lq[r] + aq;
    kk[L::K_LX + NV + r] = lv[r];
    kk[L::K_FX + r] = Fq[r];
    kk[L::K_FX + NV + r] = Fv[r] - MiI[r];
    ee[L::E_MJIDC + r] = MiI[r];
    ee[L::E_LAF + r] = ldv[r];
    ee[L::E_QAA + r] = P->dvi_weight[r];             // Qdvdv
  }
  if (tid >= 64 && tid < 64 + NU) {
    const int j = tid - 64;
    double v = 0.0;
    if (j < ni) { v = lf[j]; double af = 0.0; for (int m = 0; m < NV; ++m) af += Fvf[m + NV * j] * ldv[m]; v += af; }
    kk[L::K_LU + j] = v;
  }
  // the contact-velocity constraint rows
  if (tid >= 128 && tid < 128 + NF) W[L::W_P + tid - 128] = (tid - 128 < ni) ? Vr[tid - 128] : 0.0;
  for (int e = tid; e < NF * NX; e += 256) {
    const int c = e / NF, j = e - c * NF;
    W[L::W_PHIX + e] = j < ni ? (c < NV ? Vq[j + NF * c] : Vv[j + NF * (c - NV)]) : 0.0;
  }
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 2000 + pos;
}

// ---------------------------------------------------------------------------------------------------- K9g ----
// KKT inverse + coarse update of an aux stage (switching rows) or an impulse stage, on the 3 x 3 register tiles of K9b.
// The stage's matrices are PADDED to fixed sizes so that every loop bound is a compile-time constant:
//   variables  (w, q, v) with w = u or f padded to NU entries (identity on the padding)      -> Q is NQ x NQ, NQ = NU + NX = 48
//   constraints [F (NX rows); C (ni rows); zero rows up to NR = NX + NF = 48]; S = J Q^-1 J^T gets 1 on the padded diagonal
// Padded rows / columns of the inverse are never written out; the kinv record holds the true layout
//   lmd gmm | xi or mu (ni) | u or f (nw) | q v     with leading dimension NKG.
template <typename D>
struct KktInvEventSmem {
  static constexpr int NX = D::NX, NU = D::NU, NF = D::NF, NQ_ = NU + NX, NR = NX + NF;
  static_assert(NQ_ == NR && NR % 3 == 0 && (NR / 3) * (NR / 3) <= 256, "square 48 x 48 tiles, one per thread");
  //   A : Q^-1 -> JQ = J Q^-1 -> BR[:, NU:] (NQ x NX) ;  B : J -> S^-1 -> TR = S^-1 JQ
  static constexpr int A = 0, B = A + NQ_ * NQ_, PV = B + NR * NQ_,
                       R1 = PV + 2 * (2 * NQ_ + 2), R2 = R1 + NR, T1 = R2 + NQ_, W = T1 + NR, DIR = W + NQ_, TOTAL = DIR + NR + NQ_ + 4;
};

template <typename D>
__global__ __launch_bounds__(256) void parnmpc_kkt_inverse_general_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  using S = KktInvEventSmem<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF, NC = D::NC, NQ = S::NQ_, NR = S::NR;
  constexpr int TQ = NQ / 3;                       // 16 tiles per side of every matrix
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int tid = threadIdx.x, nt = 256;
  const long b = blockIdx.x;
  const int pos = B.general_pos[blockIdx.y];
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const ParnmpcShape sh = parnmpcShape<L>(*nd);
  const int ni = sh.ni, nw = sh.nw, nr = NX + ni, nK = sh.nk;
  const bool impulse = sh.impulse;
  const bool last = P->has_terminal && (pos == M - 2);
  const double dt = nd->dt;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ kk = B.kkt + rec * L::KKT;
  const double* __restrict__ Wc = B.swc + rec * L::SWC;
  const double* __restrict__ aux = B.aux + (b * B.NS + nd->next) * L::AUX;
  double* __restrict__ ki = B.kinv + rec * L::KINV;
  if (tid == 0) s_ok = 1;
  // ---- Q over (w, q, v), padded, straight into the register tiles ----
  double qinv[3][3];
  const int ti = tid % TQ, tj = tid / TQ;          // tile (ti, tj) of a 48 x 48 matrix
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) {
      const int r = 3 * ti + tr, c = 3 * tj + tc;
      double v;
      if (r < NU && c < NU) v = (r < nw && c < nw) ? kk[L::K_QUU + r + NU * c] : (r == c ? 1.0 : 0.0);
      else if (r < NU) v = r < nw ? kk[L::K_QXU + (c - NU) + NX * r] : 0.0;
      else if (c < NU) v = c < nw ? kk[L::K_QXU + (r - NU) + NX * c] : 0.0;
      else {
        int rr = r - NU, cc = c - NU;
        const double ax = last ? 0.0 : aux[rr + NX * cc];
        if (rr > cc) { const int t = rr; rr = cc; cc = t; }      // K5 writes the triangle on and above the diagonal (Qxx symmetric, Qvq = Qqv^T)
        v = kk[L::K_QXX + L::xsym(rr, cc)] + ax;
      }
      qinv[tr][tc] = v;
    }
  // ---- J = [0 Fqq Fqv; Fvw Fvq Fvv; 0 Cq Cv; 0] (NR x NQ) -> region B ----
  for (int e = tid; e < NR * NQ; e += nt) {
    const int c = e / NR, r = e - c * NR;
    double v = 0.0;
    if (r < NV) {
      if (c >= NU && c < NU + NV) { const int cq = c - NU; v = (r < 6 && cq < 6) ? kk[L::K_FQQ + r + 6 * cq] : ((r >= 6 && r == cq) ? -1.0 : 0.0); }
      else if (c >= NU + NV && !impulse) { const int cv = c - NU - NV; v = (r < 6 && cv < 6) ? kk[L::K_FQV + r + 6 * cv] : ((r >= 6 && r == cv) ? dt : 0.0); }
    } else if (r < NX) {
      const int rv = r - NV;
      if (c < NU) v = c < nw ? kk[L::K_FVU + rv + NV * c] : 0.0;
      else if (c < NU + NV) v = kk[L::K_FVQ + rv + NV * (c - NU)];
      else v = kk[L::K_FVV + rv + NV * (c - NU - NV)];
    } else if (r < nr && c >= NU) {
      v = Wc[L::W_PHIX + (r - NX) + NF * (c - NU)];
    }
    sm[S::B + e] = v;
  }
  if (tid < NR) sm[S::R1 + tid] = tid < NX ? kk[L::K_FX + tid] : (tid < nr ? Wc[L::W_P + tid - NX] : 0.0);
  if (tid >= 64 && tid < 64 + NQ) { const int r = tid - 64; sm[S::R2 + r] = r < NU ? (r < nw ? kk[L::K_LU + r] : 0.0) : kk[L::K_LX + r - NU]; }
  blockLdsSync();
  // ---- Q^-1 ----
  gaussJordanTiles<NQ>(qinv, true, ti, tj, &sm[S::PV], &s_ok);
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * ti + tr + NQ * (3 * tj + tc)] = qinv[tr][tc];
  blockLdsSync();
  // ---- JQ = J Q^-1 (all 256 tiles), then w = Q^-1 r2 ----
  double acc[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  tileMM<NQ>(acc, [&](int r, int m) { return sm[S::B + 3 * ti + r + NR * m]; },
             [&](int m, int c) { return sm[S::A + m + NQ * (3 * tj + c)]; });
  if (tid < NQ) {
    double w = 0.0;
    for (int m = 0; m < NQ; ++m) w += sm[S::A + tid + NQ * m] * sm[S::R2 + m];
    sm[S::W + tid] = w;
  }
  blockLdsSync();                                   // Q^-1 is dead in LDS: JQ takes its place
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * ti + tr + NR * (3 * tj + tc)] = acc[tr][tc];
  blockLdsSync();
  // ---- S = J JQ^T (+ 1 on the padded diagonal) into register tiles, S^-1 ----
  double sinv[3][3] = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  tileMM<NQ>(sinv, [&](int r, int m) { return sm[S::B + 3 * ti + r + NR * m]; },
             [&](int m, int c) { return sm[S::A + 3 * tj + c + NR * m]; });
  if (ti == tj) {
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) if (3 * ti + t3 >= nr) sinv[t3][t3] = 1.0;
  }
  gaussJordanTiles<NR>(sinv, true, ti, tj, &sm[S::PV], &s_ok);      // its first barrier also ends the reads of J
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) sm[S::B + 3 * ti + tr + NR * (3 * tj + tc)] = sinv[tr][tc];
  blockLdsSync();
  // ---- TR = S^-1 JQ ; t1 = r1 - JQ r2 ; what else reads S^-1: TL = -S^-1[:, 0:NX] and -S^-1 r1 ----
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) acc[tr][tc] = 0.0;
  tileMM<NR>(acc, [&](int r, int m) { return sm[S::B + 3 * ti + r + NR * m]; },
             [&](int m, int c) { return sm[S::A + m + NR * (3 * tj + c)]; });
  if (tid < NR) {
    const int r = tid;
    double t = sm[S::R1 + r];
    for (int m = 0; m < NQ; ++m) t -= sm[S::A + r + NR * m] * sm[S::R2 + m];
    sm[S::T1 + r] = t;
    double d = 0.0;
    for (int m = 0; m < NR; ++m) d -= sm[S::B + r + NR * m] * sm[S::R1 + m];
    sm[S::DIR + r] = d;
  }
  for (int e = tid; e < nr * NX; e += nt) {
    const int c = e / nr, r = e - c * nr;
    ki[L::I_C0 + r + L::NKG * c] = -sm[S::B + r + NR * c];
  }
  blockLdsSync();                                   // S^-1 is dead: TR takes its place
#pragma unroll
  for (int tr = 0; tr < 3; ++tr)
#pragma unroll
    for (int tc = 0; tc < 3; ++tc) sm[S::B + 3 * ti + tr + NR * (3 * tj + tc)] = acc[tr][tc];
  blockLdsSync();
  // ---- BR[:, NU:] = Q^-1[:, NU:] - TR^T JQ[:, NU:] by the threads that hold those tiles of Q^-1 ; coarse direction ----
  const bool b_on = tj >= NU / 3;
  if (b_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) acc[tr][tc] = 0.0;
    tileMM<NR>(acc, [&](int r, int m) { return sm[S::B + m + NR * (3 * ti + r)]; },
               [&](int m, int c) { return sm[S::A + m + NR * (3 * tj + c)]; });
  } else if (tid < NR) {
    double d = sm[S::DIR + tid];
    for (int m = 0; m < NQ; ++m) d += sm[S::B + tid + NR * m] * sm[S::R2 + m];
    sm[S::DIR + tid] = d;
  }
  if (tid >= 256 - NQ) {
    const int r = tid - (256 - NQ);
    double d = sm[S::W + r];
    for (int m = 0; m < NR; ++m) d += sm[S::B + m + NR * r] * sm[S::T1 + m];
    sm[S::DIR + NR + r] = d;
  }
  blockLdsSync();                                   // JQ is dead: BR[:, NU:] takes its place (NQ x NX)
  if (b_on) {
#pragma unroll
    for (int tr = 0; tr < 3; ++tr)
#pragma unroll
      for (int tc = 0; tc < 3; ++tc) sm[S::A + 3 * ti + tr + NQ * (3 * (tj - NU / 3) + tc)] = qinv[tr][tc] - acc[tr][tc];
  }
  blockLdsSync();
  // ---- column blocks of the inverse in the true row layout: C0 = KKT_inv[:, 0:NX] (top part written above),
  //      C1 = KKT_inv[:, nK-NX : nK] ----
  for (int e = tid; e < nK * NX; e += nt) {
    const int c = e / nK, r = e - c * nK;
    if (r < nr) {
      ki[L::I_C1G + r + L::NKG * c] = sm[S::B + r + NR * (NU + c)];
    } else {
      const int rt = r - nr;                                   // true variable row: w (nw), then q, v
      const int rq = rt < nw ? rt : NU + (rt - nw);            // padded variable row
      ki[L::I_C0 + r + L::NKG * c] = sm[S::B + c + NR * rq];
      ki[L::I_C1G + r + L::NKG * c] = sm[S::A + rq + NQ * c];
    }
  }
  // ---- s_new = s - direction (split_backward_correction.hxx:49-63, impulse_split_backward_correction.hxx:43-55) ----
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ sn = B.snew + rec * L::SNEW;
  const double* dir = &sm[S::DIR];          // padded: [lmd gmm | extra (NF) ] | [w (NU) | dq dv]
  const double* dw = dir + NR;
  if (tid < NV) {
    sn[L::N_LMD + tid] = s[L::S_LMD + tid] - dir[tid];
    sn[L::N_GMM + tid] = s[L::S_GMM + tid] - dir[NV + tid];
    sn[L::N_V + tid] = s[L::S_V + tid] - dw[NU + NV + tid];
    if (tid >= 6) sn[L::N_Q + tid + 1] = s[L::S_Q + tid + 1] - dw[NU + tid];
  }
  if (!impulse) {
    if (tid >= 64 && tid < 64 + NU) sn[L::N_U + tid - 64] = s[L::S_U + tid - 64] - dw[tid - 64];
    if (tid >= 96 && tid < 96 + ni) sn[L::N_XI + tid - 96] = s[L::S_XI + tid - 96] - dir[NX + tid - 96];
  } else if (tid >= 64 && tid < 64 + NC && nd->active[tid - 64]) {
    const int c = tid - 64, row = nd->row_of[c];
    for (int k = 0; k < 3; ++k) {
      sn[L::N_U + row + k] = s[L::S_F + 3 * c + k] - dw[row + k];             // f, packed rows
      sn[L::N_XI + row + k] = s[L::S_MU + 3 * c + k] - dir[NX + row + k];     // mu, packed rows
    }
  }
  if (tid == 128) {
    double qn[7];
    lieIntegrateBase(s + L::S_Q, dw + NU, -1.0, qn);
    for (int k = 0; k < 7; ++k) sn[L::N_Q + k] = qn[k];
  }
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1000 + pos;
}

template <typename D>
void OcpLaunch<D>::parnmpcImpulseCondense(const OcpBuffers& B, long batch, int n_impulse, bool residual, const double* q0, const double* v0,
                                          hipStream_t st) {
  if (n_impulse <= 0) return;
  if (residual) hipLaunchKernelGGL((parnmpc_impulse_condense_kernel<D, true>), dim3((unsigned)batch, (unsigned)n_impulse), dim3(256), 0, st, B, q0, v0);
  else hipLaunchKernelGGL((parnmpc_impulse_condense_kernel<D, false>), dim3((unsigned)batch, (unsigned)n_impulse), dim3(256), 0, st, B, q0, v0);
}

// line search: stage cost and l1 violation of the impulse stages of the trial iterate Btry.sol points at (after K5a's impulse pass and
// the Lie-group kernel on Btry; overwrites what the stage kernel left in merit_stage for these slots)
template <typename D>
void OcpLaunch<D>::parnmpcImpulseMerit(const OcpBuffers& Btry, long batch, int n_impulse, const double* q0, const double* v0, hipStream_t st) {
  if (n_impulse <= 0) return;
  hipLaunchKernelGGL((parnmpc_impulse_condense_kernel<D, true, true>), dim3((unsigned)batch, (unsigned)n_impulse), dim3(256), 0, st, Btry, q0, v0);
}

template <typename D>
void OcpLaunch<D>::parnmpcEventInverse(const OcpBuffers& B, long batch, int n_general, hipStream_t st) {
  if (n_general <= 0) return;
  // K9w's general instantiation (one wavefront per stage, parnmpc_kkt_wave_kernel.hip) unless IDOCP_K9G_WAVE=0 / IDOCP_K9_WAVE=0 ask for K9g below
  static const bool wave = [] { const char* e = getenv("IDOCP_K9G_WAVE"); const char* e2 = getenv("IDOCP_K9_WAVE"); return !((e && e[0] == '0') || (e2 && e2[0] == '0')); }();
  if (wave) { parnmpcEventInverseWave(B, batch, n_general, st); return; }
  const size_t smem = KktInvEventSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)parnmpc_kkt_inverse_general_kernel<D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  hipLaunchKernelGGL((parnmpc_kkt_inverse_general_kernel<D>), dim3((unsigned)batch, (unsigned)n_general), dim3(256), smem, st, B);
}

template void OcpLaunch<LeggedDims<4, 3>>::parnmpcImpulseCondense(const OcpBuffers&, long, int, bool, const double*, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::parnmpcImpulseMerit(const OcpBuffers&, long, int, const double*, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::parnmpcEventInverse(const OcpBuffers&, long, int, hipStream_t);

}  // namespace idocp_dev
