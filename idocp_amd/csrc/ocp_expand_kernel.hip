// K6 / K7 and small helpers of the contact path.
//
// K6  ocp_expand_primal_kernel   RiccatiRecursionSolver::computeDirection (src/ocp/riccati_recursion_solver.cpp:165-251):
//       costate direction (split_riccati_factorizer.hxx:131-139), ContactDynamics::computeCondensedPrimalDirection
//       (contact_dynamics.hxx:161-168), slack/dual directions and fraction-to-boundary step sizes
//       (joint_*_limit.cpp, linearized_friction_cone.cpp:155-182, pdipm.hxx:52-81)
// K7  ocp_expand_dual_integrate_kernel   OCPLinearizer::integrateSolution (src/ocp/ocp_linearizer.cpp:140-221):
//       ContactDynamics::computeCondensedDualDirection (contact_dynamics.hxx:171-190), costate correction
//       (state_equation.hxx:96-108), SplitSolution::integrate (split_solution.hxx:215-240), slack/dual update
// One wavefront per stage.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dev_lie.hpp"
#include "dev_dense.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

__device__ __forceinline__ double ocpLimit2(const OcpProblem* __restrict__ P, int comp, int r) {
  switch (comp) {
    case 0: return P->q_min[r];
    case 1: return P->q_max[r];
    case 2: return -P->v_max[r];
    case 3: return P->v_max[r];
    case 4: return -P->u_max[r];
    case 8: return P->a_min[r];
    case 9: return P->a_max[r];
    default: return P->u_max[r];
  }
}
// which IPM rows exist on a stage (constraints_data.hpp:18-42); see ocpRowValid in ocp_condense_kernel.hip
__device__ __forceinline__ bool ocpRowValid2(const OcpProblem* __restrict__ P, int comp, int level, bool impulse) {
  if (impulse) return comp == 6 && P->use_impulse_friction_cone != 0;
  if (comp < 2) return P->use_q_limits && level >= 2;
  if (comp < 4) return P->use_v_limits && level >= 1;
  if (comp < 6) return P->use_u_limits != 0;
  if (comp == 8) return P->use_a_lower != 0;
  if (comp == 9) return P->use_a_upper != 0;
  return comp == 6 && P->use_friction_cone != 0;
}
__device__ __forceinline__ double f2b(double rate, double x, double dx, double cur) {
  const double f = -rate * (x / dx);
  return (f > 0.0 && f < 1.0 && f < cur) ? f : cur;
}

// One IPM row (joint limit or friction cone) of stage i: value of the
// constrained function g(x) and its directional derivative dg.
template <typename D>
__device__ __forceinline__ bool ipmRow(const OcpProblem* __restrict__ P, const OcpNode* __restrict__ nd, int row, const double* __restrict__ s,
                                       const double* dq, const double* dv, const double* du, const double* df_slot, const double* da,
                                       const double* __restrict__ xx, double* g, double* dg) {
  using L = OcpLayout<D>;
  constexpr int NU = D::NU;
  if (row >= L::C_CD) {
    // ContactDistance of a contact that is not active (contact_distance.cpp:105-146): g = - z, dg = - J_c dq, both from the ext record
    const int c = row - L::C_CD;
    if (!xx || !P->use_contact_distance || nd->kind == 1 || nd->level < 2 || nd->active[c]) return false;
    *g = -xx[L::X_Z + c];
    double acc = 0.0;
    for (int t = 0; t < D::NV; ++t) acc += xx[L::X_CDJ + c * D::NV + t] * dq[t];
    *dg = -acc;
    return true;
  }
  if (row >= L::C_ACC) {
    // JointAccelerationLowerLimit / UpperLimit on a.tail(dimu) (joint_acceleration_{lower,upper}_limit.cpp:78-93)
    const int c = 8 + (row - L::C_ACC) / NU, j = (row - L::C_ACC) % NU;
    if (!ocpRowValid2(P, c, nd->level, nd->kind == 1)) return false;
    const double sgn = (c & 1) ? 1.0 : -1.0;
    *g = sgn * (s[L::S_A + 6 + j] - ocpLimit2(P, c, j));
    *dg = sgn * da[6 + j];
    return true;
  }
  if (row < L::C_FRIC) {
    const int c = row / NU, j = row - c * NU;
    if (!ocpRowValid2(P, c, nd->level, nd->kind == 1)) return false;
    const double sgn = (c & 1) ? 1.0 : -1.0;
    const double x = (c < 2) ? s[L::S_Q + 7 + j] : ((c < 4) ? s[L::S_V + 6 + j] : s[L::S_U + j]);
    const double dx = (c < 2) ? dq[6 + j] : ((c < 4) ? dv[6 + j] : du[j]);
    *g = sgn * (x - ocpLimit2(P, c, j));
    *dg = sgn * dx;
    return true;
  }
  const int fr = row - L::C_FRIC, c = fr / 5, r = fr - 5 * c;
  const int ck = nd->kind == 1 ? P->impulse_cone_kind : P->cone_kind;
  if (r >= coneRows(ck) || !ocpRowValid2(P, 6, nd->level, nd->kind == 1) || !nd->active[c]) return false;
  double J[3];
  const double f[3] = {s[L::S_F + 3 * c], s[L::S_F + 3 * c + 1], s[L::S_F + 3 * c + 2]};
  *g = coneRow(ck, P->mu, r, f, J);              // (FrictionCone: J is the reference's data.r[i], the gradient at the linearisation point)
  *dg = J[0] * df_slot[3 * c] + J[1] * df_slot[3 * c + 1] + J[2] * df_slot[3 * c + 2];
  return true;
}
// a cone row of a contact that is NOT active on this stage: dslack = ddual = 1, so that it never limits the step
// (linearized_friction_cone.cpp:162-163, friction_cone.cpp:155-156)
// (likewise the ContactDistance row of a contact that IS active, contact_distance.cpp:113-114)
template <typename D>
__device__ __forceinline__ bool ipmIdleConeRow(const OcpProblem* __restrict__ P, const OcpNode* __restrict__ nd, int row) {
  using L = OcpLayout<D>;
  if (row >= L::C_CD) return P->use_contact_distance && nd->kind != 1 && nd->level >= 2 && nd->active[row - L::C_CD];
  if (row < L::C_FRIC || row >= L::C_ACC || !ocpRowValid2(P, 6, nd->level, nd->kind == 1)) return false;
  const int r = (row - L::C_FRIC) % 5;
  return r < coneRows(nd->kind == 1 ? P->impulse_cone_kind : P->cone_kind);
}

// 16-byte loads of n2 double2's into registers (all in flight at once), and their way into LDS
typedef double ex_d2 __attribute__((ext_vector_type(2)));
template <int N2>
__device__ __forceinline__ void wideLoad(ex_d2 (&r)[(N2 + 63) / 64], const double* __restrict__ src, int lane) {
  const ex_d2* __restrict__ p = reinterpret_cast<const ex_d2*>(src);
#pragma unroll
  for (int t = 0; t < (N2 + 63) / 64; ++t) { const int e = lane + 64 * t; r[t] = p[e < N2 ? e : N2 - 1]; }
}
// The same for a record of which only the first n2 pieces carry data (lanes beyond re-read the last one: same cache line, no traffic)
template <int N2>
__device__ __forceinline__ void wideLoadN(ex_d2 (&r)[(N2 + 63) / 64], const double* __restrict__ src, int lane, int n2) {
  const ex_d2* __restrict__ p = reinterpret_cast<const ex_d2*>(src);
#pragma unroll
  for (int t = 0; t < (N2 + 63) / 64; ++t) { const int e = lane + 64 * t; r[t] = p[e < n2 ? e : n2 - 1]; }
}
// A column-major block with leading dimension 2 LD2 of which only the first 2 h rows of a column carry data (h pieces per column, NC columns):
// piece e of the compact numbering is row pair e mod h of column e / h.  inv = ceil(65536 / h) turns the division into a multiplication (exact
// for e < 65536 / h).
template <int N2, int LD2>
__device__ __forceinline__ void wideLoadRows(ex_d2 (&r)[(N2 + 63) / 64], const double* __restrict__ src, int lane, int h, int inv, int n2) {
  const ex_d2* __restrict__ p = reinterpret_cast<const ex_d2*>(src);
#pragma unroll
  for (int t = 0; t < (N2 + 63) / 64; ++t) {
    int e = lane + 64 * t;
    e = e < n2 ? e : n2 - 1;
    const int c = (e * inv) >> 16;
    r[t] = p[c * LD2 + (e - c * h)];
  }
}
template <int N2, int LD2>
__device__ __forceinline__ void wideStoreLdsRows(double* dst, const ex_d2 (&r)[(N2 + 63) / 64], int lane, int h, int inv, int n2) {
#pragma unroll
  for (int t = 0; t < (N2 + 63) / 64; ++t) {
    const int e = lane + 64 * t;
    const int c = (e * inv) >> 16;
    if (e < n2) reinterpret_cast<ex_d2*>(dst)[c * LD2 + (e - c * h)] = r[t];
  }
}
template <int N2>
__device__ __forceinline__ void wideStoreLds(double* dst, const ex_d2 (&r)[(N2 + 63) / 64], int lane) {
#pragma unroll
  for (int t = 0; t < (N2 + 63) / 64; ++t) { const int e = lane + 64 * t; if (e < N2) reinterpret_cast<ex_d2*>(dst)[e] = r[t]; }
}

// K6.  One wavefront per stage.  The stage's matrices (P, MJtJinv_dIDCdqv and the torque columns of MJtJinv, 19 kB) are fetched
// with 16-byte loads issued back to back at the top of the kernel and staged through LDS; the matrix-vector products then read
// LDS with one row per lane.  (Round 1 read the columns straight from global memory: 30 of 64 lanes, 8 bytes each, per load.)
template <typename D>
__global__ __launch_bounds__(64) void ocp_expand_primal_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF, NVF = D::NVF, NC = D::NC;
  constexpr int RL = L::R_SV + NV, MJDL = NVF * NX, MJUL = L::MJ_TRI;      // MJtJinv: the packed lower triangle (its torque columns are read out of it)
  static_assert(RL % 2 == 0 && MJDL % 2 == 0 && MJUL % 2 == 0 && L::E_MJD % 2 == 0 && L::EXP % 2 == 0 && L::RIC % 2 == 0, "16-byte loads");
  // P and MJtJinv_dIDCdqv take turns in ONE buffer (both wait in registers; 13 instead of 21 kB per wavefront: eleven instead of
  // seven stages in flight per CU)
  constexpr int PML = RL > MJDL ? RL : MJDL;
  __shared__ __attribute__((aligned(16))) double pm[PML], mju[MJUL], sr[L::SOL];
  double* pb = pm;
  double* mjd = pm;
  __shared__ double dx[NX], du[NU], dfs[NF], das[NV];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const bool terminal = (pos == M - 1);
  const long rec = b * B.NS + nd->slot;
  double* __restrict__ dd = B.dir + rec * L::DIR;
  if (P->backward_euler && terminal) return;        // ParNMPC: placeholder stage
  const bool costate = !P->backward_euler;          // ParNMPC: dlmd, dgmm come from the backward correction (K10b)
  const bool bimp = P->backward_euler && nd->kind == 1;      // ParNMPC impulse stage: df, dmu come from the backward correction
  const bool expand = !terminal && !bimp;
  ex_d2 pw[(RL / 2 + 63) / 64], mw[(MJDL / 2 + 63) / 64], uw[(MJUL / 2 + 63) / 64];
  if (costate) wideLoad<RL / 2>(pw, B.ric + rec * L::RIC, lane);
  // only the rows the stage has are fetched: NV + dimf of the NVF rows of MJtJinv_dIDCdqv, the first (NV + dimf)(NV + dimf + 1) / 2 entries of
  // the triangle of MJtJinv -- 3 kB less per stage with half of the feet in contact (round 4: the whole padded blocks)
  const int rows_ld = NV + nd->dimf, h_ld = (rows_ld + 1) >> 1, inv_ld = 65535 / h_ld + 1, nmjd2 = h_ld * NX, ntri2 = (rows_ld * (rows_ld + 1) / 2 + 1) >> 1;
  static_assert(NVF % 2 == 0 && (NVF / 2) * NX == MJDL / 2, "pieces per column of MJtJinv_dIDCdqv");
  if (expand) {
    wideLoadRows<MJDL / 2, NVF / 2>(mw, B.exp + rec * L::EXP + L::E_MJD, lane, h_ld, inv_ld, nmjd2);
    wideLoadN<MJUL / 2>(uw, B.exp + rec * L::EXP + L::E_MJ, lane, ntri2);
  }
  // the small operands of the second half (solution, slack / dual rows of this lane, MJtJinv_IDC) travel with the matrices
  static_assert(L::SOL % 2 == 0 && L::NCON <= 128, "two IPM rows per lane");
  ex_d2 sw[(L::SOL / 2 + 63) / 64];
  double sl_r[2] = {1.0, 1.0}, dl_r[2] = {1.0, 1.0}, mjidc_r = 0.0;
  if (!terminal) {
    wideLoad<L::SOL / 2>(sw, B.sol + rec * L::SOL, lane);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = lane + 64 * t;
      if (row < L::NCON) { sl_r[t] = B.slack[rec * L::CON + row]; dl_r[t] = B.dual[rec * L::CON + row]; }
    }
    if (lane < NVF) mjidc_r = B.exp[rec * L::EXP + L::E_MJIDC + lane];
  }
  if (lane < NV) { dx[lane] = dd[L::D_Q + lane]; dx[NV + lane] = dd[L::D_V + lane]; }
  if (lane < NU && !terminal) du[lane] = dd[L::D_U + lane];
  if (lane < NF) dfs[lane] = 0.0;
  if (costate) wideStoreLds<RL / 2>(pb, pw, lane);
  if (expand) wideStoreLds<MJUL / 2>(mju, uw, lane);
  if (!terminal) wideStoreLds<L::SOL / 2>(sr, sw, lane);
  waveLdsSync();
  if (lane < NX && costate) {
    // costate direction (split_riccati_factorizer.hxx:131-139): [dlmd; dgmm] = P dx - s, one row per lane
    const bool isv = lane >= NV;
    const int r = isv ? lane - NV : lane;
    double acc = -pb[(isv ? L::R_SV : L::R_SQ) + r];
    if (!isv) {
#pragma unroll
      for (int c = 0; c < NV; ++c) acc += pb[L::R_PQQ + L::psym(r, c)] * dx[c] + pb[L::R_PQV + r + NV * c] * dx[NV + c];
    } else {
#pragma unroll
      for (int c = 0; c < NV; ++c) acc += pb[L::R_PQV + c + NV * r] * dx[c] + pb[L::R_PVV + L::psym(r, c)] * dx[NV + c];
    }
    dd[(isv ? L::D_GMM : L::D_LMD) + r] = acc;
  }
  if (terminal) return;
  waveLdsSync();                                  // P has been read: the buffer takes MJtJinv_dIDCdqv
  if (expand) wideStoreLdsRows<MJDL / 2, NVF / 2>(mjd, mw, lane, h_ld, inv_ld, nmjd2);
  waveLdsSync();
  const long su = rec;
  const double* s = sr;
  const int dimf = nd->dimf, dimvf = NV + dimf;
  // SplitRiccatiFactorizer::computeLagrangeMultiplierDirection (split_riccati_factorizer.hxx:139-145): dxi = M dx + m
  if (bimp) {
    // ImpulseDynamicsBackwardEuler::computeCondensedPrimalDirection (impulse_dynamics_backward_euler.hxx:98-104):
    // ddv = - Minv ImD + Fvq dq + Fvf df   (K9i left Fvq / Fvf in the kkt record, du holds df in packed rows)
    const double* __restrict__ kk = B.kkt + rec * L::KKT;
    if (lane < NF) dfs[lane] = dd[L::D_F + lane];
    if (lane < NV) {
      double acc = -mjidc_r;
      double t1 = 0.0;
      for (int c = 0; c < NV; ++c) t1 += kk[L::K_FVQ + lane + NV * c] * dx[c];
      acc += t1;
      double t2 = 0.0;
      for (int j = 0; j < dimf; ++j) t2 += kk[L::K_FVU + lane + NV * j] * du[j];
      acc += t2;
      dd[L::D_A + lane] = acc; das[lane] = acc;
    }
  }
  if (!P->backward_euler && nd->sw_dimi > 0 && lane >= 32 && lane < 32 + nd->sw_dimi) {
    const int l = lane - 32;
    const double* __restrict__ W = B.swc + rec * L::SWC;
    double acc = W[L::W_m + l];
    for (int c = 0; c < NX; ++c) acc += W[L::W_M + l + NF * c] * dx[c];
    dd[L::D_XI + l] = acc;
  }
  if (!bimp && lane < dimvf) {
    const int r = lane;
    double acc = -mjidc_r, tt = 0.0, ww = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) tt += mjd[r + NVF * c] * dx[c];
#pragma unroll
    for (int j = 0; j < NU; ++j) { const int c = 6 + j; ww += (r >= c ? mju[r * (r + 1) / 2 + c] : mju[c * (c + 1) / 2 + r]) * du[j]; }      // MJtJinv(r, 6 + j)
    acc += ww - tt;
    dd[L::D_T + r] = tt; dd[L::D_W + r] = ww;        // for the dual expansion (K7)
    if (r < NV) { dd[L::D_A + r] = acc; das[r] = acc; }
    else {
      // d.df() *= -1; packed active row -> contact slot
      const int pr = r - NV;
      for (int c = 0; c < NC; ++c) if (nd->active[c] && pr >= nd->row_of[c] && pr < nd->row_of[c] + 3) {
        const int slot = 3 * c + (pr - nd->row_of[c]);
        dfs[slot] = -acc; dd[L::D_F + slot] = -acc;
      }
    }
  }
  waveLdsSync();
  double ps = 1.0, ds = 1.0;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int row = lane + 64 * t;
    double g, dg;
    if (row >= L::NCON || !ipmRow<D>(P, nd, row, s, dx, dx + NV, du, dfs, das, B.ext ? B.ext + rec * L::EXT : nullptr, &g, &dg)) continue;
    const double sl = sl_r[t], dl = dl_r[t];
    const double res = g + sl, duality = sl * dl - P->barrier;
    const double dslack = -dg - res;
    const double ddual = -(dl * dslack + duality) / sl;
    ps = f2b(P->fraction_rate, sl, dslack, ps);
    ds = f2b(P->fraction_rate, dl, ddual, ds);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { ps = fmin(ps, __shfl_xor(ps, off)); ds = fmin(ds, __shfl_xor(ds, off)); }
  if (lane == 0) { B.step_stage[su * 2] = ps; B.step_stage[su * 2 + 1] = ds; }
}

// ------------------------------------------------------------------------------------------------------------------------------
// S4 + K6 in ONE kernel (round 5): the forward Riccati sweep that expands as it walks.
//
// RiccatiRecursionSolver::computeInitialStateDirection + forwardRiccatiRecursion (src/ocp/riccati_recursion_solver.cpp:110-162) and
// computeDirection (:165-251) -- the costate P dx - s (split_riccati_factorizer.hxx:131-139), dxi = M dx + m (:139-145),
// ContactDynamics::computeCondensedPrimalDirection (contact_dynamics.hxx:161-168), slack / dual directions and the fraction-to-boundary
// candidates -- stage by stage along the chain of ONE instance, one workgroup (two wavefronts, below) per instance.  Rounds 1 - 4 ran S4 (a serial sweep that read
// the gain and the F blocks of the kkt record, 11.2 kB per stage) and then K6 (one wavefront per stage: P, MJtJinv [dIDC], MJtJinv, the
// solution and IPM rows, 22 kB per stage) back to back over the same stages.  What S4 needs of a stage's dynamics is what K6 computes anyway:
//     t = MJtJinv_dIDCdqv dx,   w = MJtJinv[:, u] du,   d[a; f] = w - t - MJtJinv_IDC            (contact_dynamics.hxx:161-168)
//     dv+ = Fvq dq + Fvv dv + Fvu du + Fv = dv + dt (w - t)[0:nv] + Fv                          (Fvq = -dt MJD[0:nv, q], Fvv = I - dt MJD[0:nv, v],
//                                                                                                 Fvu = dt MJtJinv[0:nv, u]: contact_dynamics.hxx:150-156)
// so the walk reads every record of a stage ONCE (gain, exp, ric, sol, slack / dual, and Fqq6, Fqv6, Fx of the kkt record: 26 kB instead
// of 33) and the 6.9 kB of Fvq / Fvv / Fvu are not read at all.  The records of stages i + 1 and i + 2 are in flight (16-byte loads into two
// register sets, S4's scheme) while stage i is computed out of LDS; the step sizes of the instance are reduced on the way
// (ocp_reduce_steps_kernel's job: the minimum over the chain's stages).
// TWO wavefronts per instance.  One wavefront per instance (the first form of this kernel: 0.95 ms, then 0.86 with the row metadata decoded
// up front and P unpacked to a full matrix; S4 + K6 took 0.97) is a chain of ~2 000 dependent instructions per stage with nothing to hide a
// single latency behind -- one wavefront per SIMD at batch 1024 --: 7.2 us per stage.  Only a part of that is the recursion proper.  Wavefront 0
// keeps  du = K dx + k,  t, w, d[a; f]  and  dx+  (gain, MJtJinv [dIDC], MJtJinv, Fqq6 / Fqv6 / Fx); wavefront 1 follows ONE STAGE BEHIND with
// what only consumes the direction: the costate P dx - s, dxi, the slack / dual directions and the fraction-to-boundary candidates (P, s,
// solution, slack / dual rows).  The direction of a stage travels through a ring of three LDS vectors (stage i in slot i mod 3: wavefront 0
// fills dx of slot i + 1 while wavefront 1 still reads slot i - 1), one LDS barrier per stage: 0.69 ms, 4.7 TB/s.
// Everything the walk needs besides the stage records lives in LDS or in registers BEFORE the first store of the kernel: the chain (slot, time
// steps, contact status: two words and two doubles per node), the rows' limits, the problem's switches.  A load from global memory inside the
// walk -- a node field, a limit indexed by the lane -- would make a wavefront wait for it with s_waitcnt vmcnt(0), i.e. for the records of the
// NEXT two stages as well, which are in flight by then (vector memory returns in order): 12 us per stage (1.45 ms) in the very first form.
template <typename D>
__global__ __launch_bounds__(128, 1) void ocp_forward_expand_kernel(OcpBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NU = D::NU, NF = D::NF, NVF = D::NVF, NC = D::NC;
  constexpr int GL = L::GAIN, RL = L::R_SV + NV, MJDL = NVF * NX, MJUL = L::MJ_TRI;
  static_assert(GL % 2 == 0 && RL % 2 == 0 && MJDL % 2 == 0 && MJUL % 2 == 0 && L::SOL % 2 == 0 && L::GAIN % 2 == 0 && L::RIC % 2 == 0 && L::EXP % 2 == 0 &&
                L::E_MJD % 2 == 0 && L::KKT % 2 == 0 && L::K_FQQ % 2 == 0 && L::K_FX % 2 == 0 && NVF % 2 == 0, "16-byte loads");
  static_assert(L::K_FQV == L::K_FQQ + 36 && L::NCON <= 128 && NX <= 36 && 36 + NX / 2 <= 64, "Fqq6 | Fqv6 and Fx on one load; two IPM rows per lane");
  static_assert(NC <= 4 && NF <= 15 && NVF + NV <= 64 && L::R_PQQ == 0 && L::R_SV == L::R_SQ + NV, "node word: four contacts; lane maps");
  // wavefront 1 reaches the cone and contact-distance rows only as row = lane + 64 (the joint-limit / acceleration rows through both t): every
  // such row must lie in [64, 128)
  static_assert(L::C_FRIC >= 64 && L::C_CD >= 64 && L::NCON <= 128, "cone / contact-distance rows are mapped to lane + 64 only");
  constexpr int NG2 = (GL / 2 + 63) / 64, NP2 = (RL / 2 + 63) / 64, NM2 = (MJDL / 2 + 63) / 64, NT2 = (MJUL / 2 + 63) / 64, NS2 = (L::SOL / 2 + 63) / 64;
  constexpr int MAXM = OcpForwardExpandMaxChain;
  constexpr int PF = NX * NX, PSV = PF, PPAD = PF + NX;
  constexpr int DV = NX + NU + NV + NF + 2;             // one slot of the ring: dx | du | da | df (contact slots)
  __shared__ __attribute__((aligned(16))) double gb[GL], pfull[PF + NX + 2], mjd[MJDL], mju[MJUL], sr[L::SOL], fb[72 + NX];
  __shared__ double ring[3][DV];
  __shared__ int n_slot[MAXM], n_word[MAXM];
  __shared__ double n_dt[MAXM], n_dtq[MAXM];
  const int M = B.M;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const long b = blockIdx.x;
  const long base = b * B.NS;
  int use_cone, use_icone, use_cd, ck_stage, ck_imp;
  double cmu, barrier, frate;
  int r_xoff[2], r_doff[2], r_thr[2];
  double r_sgn[2], r_bound[2];
  {
    const OcpProblem* __restrict__ P = B.prob;
    use_cone = P->use_friction_cone; use_icone = P->use_impulse_friction_cone;
    use_cd = P->use_contact_distance; ck_stage = P->cone_kind; ck_imp = P->impulse_cone_kind;
    cmu = P->mu; barrier = P->barrier; frate = P->fraction_rate;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = lane + 64 * t;
      r_xoff[t] = 0; r_doff[t] = 0; r_thr[t] = 4; r_sgn[t] = 1.0; r_bound[t] = 0.0;
      if (row < L::C_FRIC) {
        const int c = row / NU, j = row - c * NU;
        r_sgn[t] = (c & 1) ? 1.0 : -1.0;
        r_xoff[t] = (c < 2) ? L::S_Q + 7 + j : ((c < 4) ? L::S_V + 6 + j : L::S_U + j);
        r_doff[t] = (c < 2) ? 6 + j : ((c < 4) ? NV + 6 + j : NX + j);
        r_bound[t] = ocpLimit2(P, c, j);
        r_thr[t] = (c < 2) ? (P->use_q_limits ? 2 : 4) : ((c < 4) ? (P->use_v_limits ? 1 : 4) : (P->use_u_limits ? 0 : 4));
      } else if (row >= L::C_ACC && row < L::C_CD) {
        const int c = 8 + (row - L::C_ACC) / NU, j = (row - L::C_ACC) % NU;
        r_sgn[t] = (c & 1) ? 1.0 : -1.0;
        r_xoff[t] = L::S_A + 6 + j; r_doff[t] = NX + NU + 6 + j;
        r_bound[t] = ocpLimit2(P, c, j);
        r_thr[t] = (c == 8 ? P->use_a_lower : P->use_a_upper) ? 0 : 4;
      }
    }
    const OcpNode* __restrict__ nodes = B.nodes;
    for (int i = tid; i < M; i += 128) {
      const OcpNode* __restrict__ nd = nodes + i;
      int w = (nd->kind & 7) | ((nd->level < 3 ? (nd->level < 0 ? 0 : nd->level) : 3) << 3) | (nd->dimf << 5) | (nd->sw_dimi << 9) | ((nd->has_u ? 1 : 0) << 13);
#pragma unroll
      for (int c = 0; c < NC; ++c) if (nd->active[c]) w |= (1 << (14 + c)) | ((nd->row_of[c] / 3) << (18 + 2 * c));
      n_slot[i] = nd->slot; n_word[i] = w; n_dt[i] = nd->dt; n_dtq[i] = nd->dtq;
    }
  }
  // RiccatiRecursionSolver::computeInitialStateDirection (riccati_recursion_solver.cpp:110-126) -> slot 0 of the ring
  if (wave == 0) {
    double* dx = ring[0];
    const int slot0 = B.nodes[0].slot;
    const double* __restrict__ s0 = B.sol + (base + slot0) * L::SOL;
    if (lane == 0) {
      double R[9], p[3], d6[6];
      lieRelative(s0 + L::S_Q, q0 + b * NQ, R, p);         // q (-) s[0].q
      lieLog6(R, p, d6);
      const double* __restrict__ Fi = B.exp + (base + slot0) * L::EXP + L::E_FQQPI;
      for (int r = 0; r < 6; ++r) { double acc = 0.0; for (int m = 0; m < 6; ++m) acc += Fi[r + 6 * m] * d6[m]; dx[r] = -acc; }
    }
    if (lane >= 6 && lane < NV) dx[lane] = q0[b * NQ + lane + 1] - s0[L::S_Q + lane + 1];
    if (lane < NV) dx[NV + lane] = v0[b * NV + lane] - s0[L::S_V + lane];
  }
  blockLdsSync();
  if (wave == 0) {
    // =============== wavefront 0: the recursion ===============
    struct Regs { ex_d2 g[NG2], m[NM2], u[NT2], f; double mjidc; };
    Regs RA, RB;
    // lanes 0 .. NVF - 1: row `lane` of the (a, f) block; lanes 32 + (0 .. NV - 1): row of dq+ (NVF <= 32 keeps them apart when all feet stand)
    const int af_row = lane < NVF ? lane : NVF - 1;
    int tri_off[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) { const int c = 6 + j; tri_off[j] = af_row >= c ? af_row * (af_row + 1) / 2 + c : c * (c + 1) / 2 + af_row; }
    auto fetch = [&](int i, Regs& R) {
      const long rec = base + n_slot[i];
      wideLoad<GL / 2>(R.g, B.gain + rec * L::GAIN, lane);
      const int rows_ld = NV + ((n_word[i] >> 5) & 15), h_ld = (rows_ld + 1) >> 1, inv_ld = 65535 / h_ld + 1;
      wideLoadRows<MJDL / 2, NVF / 2>(R.m, B.exp + rec * L::EXP + L::E_MJD, lane, h_ld, inv_ld, h_ld * NX);
      wideLoadN<MJUL / 2>(R.u, B.exp + rec * L::EXP + L::E_MJ, lane, (rows_ld * (rows_ld + 1) / 2 + 1) >> 1);
      {
        const ex_d2* __restrict__ kq = reinterpret_cast<const ex_d2*>(B.kkt + rec * L::KKT + L::K_FQQ);
        const ex_d2* __restrict__ kx = reinterpret_cast<const ex_d2*>(B.kkt + rec * L::KKT + L::K_FX);
        R.f = lane < 36 ? kq[lane] : kx[lane - 36 < NX / 2 ? lane - 36 : NX / 2 - 1];
      }
      R.mjidc = B.exp[rec * L::EXP + L::E_MJIDC + af_row];
    };
    if (M > 1) fetch(0, RA);
    if (M > 2) fetch(1, RB);
    auto step = [&](int i, Regs& R) {
      double* const dv = ring[i % 3];
      double* const dx = dv; double* const du = dv + NX; double* const das = dv + NX + NU; double* const dfs = dv + NX + NU + NV;
      double* const dxn = ring[(i + 1) % 3];
      const int word = n_word[i];
      const long rec = base + n_slot[i];
      const int dimf = (word >> 5) & 15, amask = (word >> 14) & 15, rows3 = (word >> 18) & 255;
      const double sdt = n_dt[i], sdtq = n_dtq[i];
      double* __restrict__ dd = B.dir + rec * L::DIR;
      const int dimvf = NV + dimf;
      const int h_ld = (dimvf + 1) >> 1, inv_ld = 65535 / h_ld + 1;
      wideStoreLds<GL / 2>(gb, R.g, lane);
      wideStoreLdsRows<MJDL / 2, NVF / 2>(mjd, R.m, lane, h_ld, inv_ld, h_ld * NX);
      wideStoreLds<MJUL / 2>(mju, R.u, lane);
      if (lane < 36 + NX / 2) reinterpret_cast<ex_d2*>(fb)[lane] = R.f;
      const double mjidc_r = R.mjidc;
      waveLdsSync();
      if (i + 2 < M - 1) fetch(i + 2, R);
      // du = K dx + k: four lanes per row
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
      // t = MJtJinv_dIDCdqv dx: one row per lane
      double tt = 0.0;
      if (lane < dimvf) {
#pragma unroll
        for (int c = 0; c < NX; ++c) tt += mjd[lane + NVF * c] * dx[c];
      }
      if (lane < NF) dfs[lane] = 0.0;
      if (lane >= 40 && lane < 40 + NV) { const int r = lane - 40; dd[L::D_Q + r] = dx[r]; dd[L::D_V + r] = dx[NV + r]; }
      waveLdsSync();                                             // du stands
      // w = MJtJinv[:, u] du;  d[a; f] = w - t - MJtJinv_IDC (contact_dynamics.hxx:161-168; d.df() *= -1);  dv+ = dv + dt (w - t) + Fv
      if (lane < dimvf) {
        const int r = lane;
        double ww = 0.0;
#pragma unroll
        for (int j = 0; j < NU; ++j) ww += mju[tri_off[j]] * du[j];      // MJtJinv(r, 6 + j)
        const double acc_af = -mjidc_r + (ww - tt);
        dd[L::D_T + r] = tt; dd[L::D_W + r] = ww;                // for the dual expansion (K7)
        if (r < NV) {
          dd[L::D_A + r] = acc_af; das[r] = acc_af;
          dxn[NV + r] = dx[NV + r] + sdt * (ww - tt) + fb[72 + NV + r];
        } else {
          const int pr = r - NV;
#pragma unroll
          for (int c = 0; c < NC; ++c) {
            const int row_of = 3 * ((rows3 >> (2 * c)) & 3);
            if (((amask >> c) & 1) && pr >= row_of && pr < row_of + 3) {
              const int slot = 3 * c + (pr - row_of);
              dfs[slot] = -acc_af; dd[L::D_F + slot] = -acc_af;
            }
          }
        }
      } else if (lane >= 32 && lane < 32 + NV) {
        // dq+ = Fqq dq + Fqv dv + Fq: the base block from the kkt record, identity / dt I on the joints (state_equation.hxx:52-61)
        const int r = lane - 32;
        double dq = fb[72 + r];
        if (r < 6) {
#pragma unroll
          for (int m = 0; m < 6; ++m) dq += fb[r + 6 * m] * dx[m] + fb[36 + r + 6 * m] * dx[NV + m];
        } else {
          dq += dx[r] + sdtq * dx[NV + r];
        }
        dxn[r] = dq;
      }
    };
    static_assert(NVF <= 32, "the dq+ rows sit on the lanes behind the (a, f) rows");
    for (int i = 0; i <= M; i += 2) {
      if (i < M - 1) step(i, RA);
      blockLdsSync();
      if (i + 1 <= M) {
        if (i + 1 < M - 1) step(i + 1, RB);
        blockLdsSync();
      }
    }
    // the terminal stage's dq, dv
    {
      const double* dx = ring[(M - 1) % 3];
      double* __restrict__ dd = B.dir + (base + n_slot[M - 1]) * L::DIR;
      if (lane < NV) { dd[L::D_Q + lane] = dx[lane]; dd[L::D_V + lane] = dx[NV + lane]; }
    }
  } else {
    // =============== wavefront 1: one stage behind -- costate, dxi, slack / dual directions, step sizes ===============
    struct Regs { ex_d2 p[NP2], s[NS2]; double sl[2], dl[2]; };
    Regs RA, RB;
    int p_a[NP2][2], p_b[NP2][2];
#pragma unroll
    for (int t = 0; t < NP2; ++t) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * (lane + 64 * t) + h;
        int a = PPAD + h, bb = PPAD + h;                                 // padding entries of the record, pieces beyond it
        if (k < L::R_PQV || (k >= L::R_PVV && k < L::R_SQ)) {
          const bool vv = k >= L::R_PVV;
          const int kk = vv ? k - L::R_PVV : k;
          int c = 0;
          while ((c + 1) * (c + 2) / 2 <= kk) ++c;                       // entry (r, c), r <= c, at c (c + 1) / 2 + r
          const int r = kk - c * (c + 1) / 2;
          if (c < NV) { const int o = vv ? NV : 0; a = (o + r) + NX * (o + c); bb = (o + c) + NX * (o + r); }
        } else if (k >= L::R_PQV && k < L::R_PVV) {
          const int kk = k - L::R_PQV, c = kk / NV, r = kk - c * NV;     // Pqv(r, c)
          a = r + NX * (NV + c); bb = (NV + c) + NX * r;
        } else if (k >= L::R_SQ && k < RL) {
          a = bb = PSV + (k - L::R_SQ);
        }
        p_a[t][h] = a; p_b[t][h] = bb;
      }
    }
    auto fetch = [&](int i, Regs& R) {
      const long rec = base + n_slot[i];
      wideLoad<RL / 2>(R.p, B.ric + rec * L::RIC, lane);
      wideLoad<L::SOL / 2>(R.s, B.sol + rec * L::SOL, lane);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const int row = lane + 64 * t < L::NCON ? lane + 64 * t : L::NCON - 1;
        R.sl[t] = B.slack[rec * L::CON + row]; R.dl[t] = B.dual[rec * L::CON + row];
      }
    };
    fetch(0, RA);
    if (M > 1) fetch(1, RB);
    double ps_min = 1.0, ds_min = 1.0;
    auto step = [&](int i, Regs& R) {
      const double* const dvec = ring[i % 3];
      const double* const dx = dvec; const double* const dfs = dvec + NX + NU + NV;
      const int word = n_word[i];
      const long rec = base + n_slot[i];
      const int kind = word & 7, level = (word >> 3) & 3, sw_dimi = (word >> 9) & 15, amask = (word >> 14) & 15;
      double* __restrict__ dd = B.dir + rec * L::DIR;
      const bool terminal = (i == M - 1);
#pragma unroll
      for (int t = 0; t < NP2; ++t) {
        pfull[p_a[t][0]] = R.p[t].x; pfull[p_b[t][0]] = R.p[t].x;
        pfull[p_a[t][1]] = R.p[t].y; pfull[p_b[t][1]] = R.p[t].y;
      }
      wideStoreLds<L::SOL / 2>(sr, R.s, lane);
      const double sl0 = R.sl[0], sl1 = R.sl[1], dl0 = R.dl[0], dl1 = R.dl[1];
      waveLdsSync();
      if (i + 2 < M) fetch(i + 2, R);
      // costate [dlmd; dgmm] = P dx - s (split_riccati_factorizer.hxx:131-139): one row per lane
      if (lane < NX) {
        double acc = 0.0;
#pragma unroll
        for (int c = 0; c < NX; ++c) acc += pfull[lane + NX * c] * dx[c];
        dd[lane] = acc - pfull[PSV + lane];
      }
      static_assert(L::D_LMD == 0 && L::D_GMM == NV, "dlmd | dgmm lead the direction record");
      if (terminal) return;
      // SplitRiccatiFactorizer::computeLagrangeMultiplierDirection (split_riccati_factorizer.hxx:139-145): dxi = M dx + m
      if (sw_dimi > 0 && lane >= 40 && lane < 40 + sw_dimi) {
        const int l = lane - 40;
        const double* __restrict__ W = B.swc + rec * L::SWC;
        double acc = W[L::W_m + l];
        for (int c = 0; c < NX; ++c) acc += W[L::W_M + l + NF * c] * dx[c];
        dd[L::D_XI + l] = acc;
      }
      // slack / dual directions, fraction-to-boundary candidates (pdipm.hxx:52-81)
      const bool impulse = kind == 1;
      double ps = 1.0, ds = 1.0;
      auto candidate = [&](double g, double dg, double sl, double dl) {
        const double res = g + sl, duality = sl * dl - barrier;
        const double dslack = -dg - res;
        const double ddual = -(dl * dslack + duality) / sl;
        ps = f2b(frate, sl, dslack, ps);
        ds = f2b(frate, dl, ddual, ds);
      };
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (!impulse && level >= r_thr[t]) candidate(r_sgn[t] * (sr[r_xoff[t]] - r_bound[t]), r_sgn[t] * dvec[r_doff[t]], t ? sl1 : sl0, t ? dl1 : dl0);
      }
      {
        const int row = lane + 64;
        if (row >= L::C_FRIC && row < L::C_ACC) {
          const int fr = row - L::C_FRIC, c = fr / 5, r = fr - 5 * c;
          const int ck = impulse ? ck_imp : ck_stage;
          if (r < coneRows(ck) && (impulse ? use_icone != 0 : use_cone != 0) && ((amask >> c) & 1)) {
            double J[3];
            const double f[3] = {sr[L::S_F + 3 * c], sr[L::S_F + 3 * c + 1], sr[L::S_F + 3 * c + 2]};
            const double g = coneRow(ck, cmu, r, f, J);
            candidate(g, J[0] * dfs[3 * c] + J[1] * dfs[3 * c + 1] + J[2] * dfs[3 * c + 2], sl1, dl1);
          }
        } else if (row >= L::C_CD && row < L::NCON) {
          const int c = row - L::C_CD;
          if (B.ext != nullptr && use_cd && !impulse && level >= 2 && !((amask >> c) & 1)) {
            const double* __restrict__ xx = B.ext + rec * L::EXT;
            double acc = 0.0;
            for (int q = 0; q < NV; ++q) acc += xx[L::X_CDJ + c * NV + q] * dx[q];
            candidate(-xx[L::X_Z + c], -acc, sl1, dl1);
          }
        }
      }
#pragma unroll
      for (int off = 32; off >= 1; off >>= 1) { ps = fmin(ps, __shfl_xor(ps, off)); ds = fmin(ds, __shfl_xor(ds, off)); }
      if (lane == 0) { B.step_stage[rec * 2] = ps; B.step_stage[rec * 2 + 1] = ds; }
      ps_min = fmin(ps_min, ps); ds_min = fmin(ds_min, ds);
    };
    // iteration i of wavefront 0 is stage i; this wavefront takes stage i - 1 next to it
    for (int i = 0; i <= M; i += 2) {
      if (i >= 1) step(i - 1, RB);
      blockLdsSync();
      if (i + 1 <= M) {
        step(i, RA);
        blockLdsSync();
      }
    }
    if (lane == 0) { B.step[b * 2] = ps_min; B.step[b * 2 + 1] = ds_min; }
  }
}

__global__ __launch_bounds__(64) void ocp_reduce_steps_kernel(OcpBuffers B) {
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long b = blockIdx.x;
  double ps = 1.0, ds = 1.0;
  for (int i = threadIdx.x; i < M - 1; i += 64) {
    const long rec = b * B.NS + B.nodes[i].slot;
    ps = fmin(ps, B.step_stage[rec * 2]); ds = fmin(ds, B.step_stage[rec * 2 + 1]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { ps = fmin(ps, __shfl_xor(ps, off)); ds = fmin(ds, __shfl_xor(ds, off)); }
  if (threadIdx.x == 0) { B.step[b * 2] = ps; B.step[b * 2 + 1] = ds; }
}

__global__ __launch_bounds__(64) void ocp_kkt_error_kernel(OcpBuffers B, double* __restrict__ squared_out) {
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long b = blockIdx.x;
  double e = 0.0;
  for (int i = threadIdx.x; i < M; i += 64) e += B.err_stage[b * B.NS + B.nodes[i].slot];
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
  if (threadIdx.x == 0) { if (squared_out) squared_out[b] = e; else B.err[b] = sqrt(e); }
}

// K7.  One wavefront per stage.  EVERY global read of the stage (MJtJinv and the small blocks behind it in the exp record: Qaa, Qff,
// MJtJinv_IDC, laf, lu_passive, Quu_passive, Qxu_passive, Fqq_prev_inv, contiguous; the direction and solution records, slack and
// dual, dgmm of the next stage) is issued at the top with 16-byte loads and staged through LDS: one trip to memory per stage
// instead of a chain of dependent ones.  The kernel is bound by the bytes its resident wavefronts keep in flight (adding 4 kB of LDS
// per wavefront, i.e. seven instead of nine of them per CU, took it from 0.65 to 0.75 ms), so LDS is kept small: MJtJinv, which is
// symmetric, travels and is staged as its lower triangle (3.7 instead of 7.2 kB), and slack / dual stay in the registers of the lane
// that owns the row.
template <typename D>
__global__ __launch_bounds__(64) void ocp_expand_dual_integrate_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF, NVF = D::NVF, NC = D::NC;
  constexpr int MJL = L::MJ_TRI, TO = L::E_QAA, TL = L::E_FQQPI + 36 - L::E_QAA;
  static_assert(MJL % 2 == 0 && TO % 2 == 0 && TL % 2 == 0 && L::E_MJ == 0 && L::DIR % 2 == 0 && L::SOL % 2 == 0 && L::CON % 2 == 0, "16-byte loads");
  __shared__ __attribute__((aligned(16))) double mjt[MJL], tl[TL], dr[L::DIR], sr[L::SOL];
  auto mj = [&](int r, int c) -> double { return r >= c ? mjt[r * (r + 1) / 2 + c] : mjt[c * (c + 1) / 2 + r]; };      // MJtJinv(r, c)
  static_assert(L::NCON <= 128, "two IPM rows per lane");
  __shared__ double dgn[NV], laf[NVF + 2], dbm[NVF + 2], nup[6], dmu[NF];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const bool stage = (pos < M - 1);                 // not the terminal stage
  const bool bwd = P->backward_euler != 0;          // ParNMPC: own dgmm, + Fqq_inv^T in the costate correction
  if (bwd && !stage) return;
  const double dt = nd->dt;                         // 1 on impulse stages
  const long rec = b * B.NS + nd->slot;
  const double ap = B.step[b * 2], ad = B.step[b * 2 + 1];
  double* __restrict__ dd = B.dir + rec * L::DIR;
  double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ slack = B.slack + rec * L::CON;
  double* __restrict__ dual = B.dual + rec * L::CON;
  ex_d2 mw[(MJL / 2 + 63) / 64], tw[(TL / 2 + 63) / 64], dw[(L::DIR / 2 + 63) / 64], sw[(L::SOL / 2 + 63) / 64];
  double sl_r[2] = {1.0, 1.0}, dl_r[2] = {1.0, 1.0};
  const int rows_ld = (bwd && nd->kind == 1) ? NV : NV + nd->dimf;      // the rows of the triangle of MJtJinv this stage has (see dimvf below)
  if (stage) wideLoadN<MJL / 2>(mw, B.exp + rec * L::EXP, lane, (rows_ld * (rows_ld + 1) / 2 + 1) >> 1);
  wideLoad<TL / 2>(tw, B.exp + rec * L::EXP + TO, lane);
  wideLoad<L::DIR / 2>(dw, dd, lane);
  wideLoad<L::SOL / 2>(sw, s, lane);
  double dgn_r = 0.0;
  if (stage) {
#pragma unroll
    for (int t = 0; t < 2; ++t) { const int row = lane + 64 * t; if (row < L::NCON) { sl_r[t] = slack[row]; dl_r[t] = dual[row]; } }
    // dgmm of the next stage of the chain (backward Euler: of this stage)
    if (lane < NV) dgn_r = B.dir[(b * B.NS + (bwd ? nd->slot : nd->next)) * L::DIR + L::D_GMM + lane];
  }
  const bool bimp = bwd && nd->kind == 1;           // ParNMPC impulse stage (K9i filled the exp record): only the dv rows, dmu from K10b
  const int dimf = nd->dimf, dimvf = bimp ? NV : NV + dimf;
  if (stage) {
    wideStoreLds<MJL / 2>(mjt, mw, lane);                // the packed lower triangle, as it lies in the exp record
    if (lane < NV) dgn[lane] = dgn_r;
  }
  wideStoreLds<TL / 2>(tl, tw, lane);
  wideStoreLds<L::DIR / 2>(dr, dw, lane);
  wideStoreLds<L::SOL / 2>(sr, sw, lane);
  waveLdsSync();
  const double* dx = dr + L::D_Q;                   // dq, dv are contiguous in the record
  static_assert(L::D_V == L::D_Q + NV, "dx = [dq; dv]");
  const double* du = dr + L::D_U;
  const double* dfs = dr + L::D_F;
  if (lane < NF) dmu[lane] = dr[L::D_MU + lane];     // ParNMPC impulse stages: dmu comes from the backward correction (K10b); overwritten below otherwise
  if (stage) {
    // ---- ContactDynamics::computeCondensedDualDirection ----
    if (lane < dimvf) {
      const int r = lane;
      double acc = tl[L::E_LAF - TO + r];
      if (bimp) {
        // ImpulseDynamicsBackwardEuler::computeCondensedDualDirection (impulse_dynamics_backward_euler.hxx:105-111):
        // ldv += Qdvdv (Fvq dq + Fvf df) + dgmm, with Fvq dq + Fvf df = ddv + Minv ImD
        acc += tl[L::E_QAA - TO + r] * (dr[L::D_A + r] + tl[L::E_MJIDC - TO + r]);
      } else if (r < NV) {
        // Qafqv dx + Qafu du with Qafqv = -diag(Qaa) MJD, Qafu = diag(Qaa) MJ[:, u] (contact_dynamics.hxx:112-123)
        const double qaa = tl[L::E_QAA - TO + r];
        acc += -qaa * dr[L::D_T + r];
        acc += (nd->has_u ? qaa : 0.0) * dr[L::D_W + r];
      } else {
        double a1 = 0.0, a2 = 0.0;
        for (int p = 0; p < dimf; ++p) { const double qff = tl[L::E_QFF - TO + (r - NV) + NF * p]; a1 += qff * dr[L::D_T + NV + p]; a2 += qff * dr[L::D_W + NV + p]; }
        acc += -a1;
        acc += nd->has_u ? a2 : 0.0;
      }
      if (r < NV) acc += dt * dgn[r];
      laf[r] = acc;
    }
    {
      // dnu_passive: six rows of NU + NX + NV = 66 terms; eight lanes per row (six lanes walking 66 terms each was as long as the
      // 30 x 30 product below, with 58 lanes idle)
      constexpr int NT66 = NU + NX + NV;
      const int r = lane >> 3, part = lane & 7;
      double acc = 0.0;
      if (r < 6) {
#pragma unroll
        for (int t0 = 0; t0 < (NT66 + 7) / 8; ++t0) {
          const int t = part + 8 * t0;
          if (t < NU) acc += tl[L::E_QUUP - TO + r + 6 * t] * du[t];
          else if (t < NU + NX) acc += tl[L::E_QXUP - TO + (t - NU) + NX * r] * dx[t - NU];
          else if (t < NT66) acc += dt * mj(r, t - NU - NX) * dgn[t - NU - NX];
        }
      }
      acc += __shfl_xor(acc, 1);
      acc += __shfl_xor(acc, 2);
      acc += __shfl_xor(acc, 4);
      if (r < 6 && part == 0) {
        acc += tl[L::E_LUP - TO + r];
        const double v = nd->has_u ? -acc / dt : 0.0;
        nup[r] = v; dd[L::D_NUP + r] = v;
      }
    }
    waveLdsSync();
    if (lane < dimvf) {
      const int r = lane;
      double acc = 0.0;
      for (int p = 0; p < dimvf; ++p) acc += mj(r, p) * laf[p];
      const double v = -acc / dt;
      dbm[r] = v;
      if (r < NV) dd[L::D_BETA + r] = v;
      else {
        const int pr = r - NV;
        for (int c = 0; c < NC; ++c) if (nd->active[c] && pr >= nd->row_of[c] && pr < nd->row_of[c] + 3) { const int slot = 3 * c + (pr - nd->row_of[c]); dmu[slot] = v; dd[L::D_MU + slot] = v; }
      }
    }
  }
  // ---- correctCostateDirectionForwardEuler: dlmd.head(6) = -Fqq_prev_inv^T dlmd.head(6) ----
  double dl_corr = 0.0;
  if (lane < 6) {
    double acc = 0.0;
#pragma unroll
    for (int m = 0; m < 6; ++m) acc += tl[L::E_FQQPI - TO + m + 6 * lane] * dr[L::D_LMD + m];
    dl_corr = bwd ? acc : -acc;                     // state_equation.hxx:96-108 / 172-181
    dd[L::D_LMD + lane] = dl_corr;
  }
  // ---- IPM slack / dual update (needs the pre-update primal variables) ----
  if (stage) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int row = lane + 64 * t;
      if (row >= L::NCON) continue;
      double g, dg;
      const bool valid = ipmRow<D>(P, nd, row, sr, dx, dx + NV, du, dfs, dr + L::D_A, B.ext ? B.ext + rec * L::EXT : nullptr, &g, &dg);
      const double sl = sl_r[t], dl = dl_r[t];
      double dslack, ddual;
      if (valid) {
        const double res = g + sl, duality = sl * dl - P->barrier;
        dslack = -dg - res;
        ddual = -(dl * dslack + duality) / sl;
      } else if (ipmIdleConeRow<D>(P, nd, row)) {
        dslack = 1.0; ddual = 1.0;       // rows of inactive contacts (linearized_friction_cone.cpp:162-163)
      } else {
        continue;
      }
      slack[row] = sl + ap * dslack;
      dual[row] = dl + ad * ddual;
    }
  }
  waveLdsSync();
  // ---- SplitSolution::integrate ----
  if (lane < NV) {
    const int r = lane;
    s[L::S_LMD + r] = sr[L::S_LMD + r] + ap * (r < 6 ? dl_corr : dr[L::D_LMD + r]);
    s[L::S_GMM + r] = sr[L::S_GMM + r] + ap * dr[L::D_GMM + r];
    s[L::S_V + r] = sr[L::S_V + r] + ap * dx[NV + r];
    if (r >= 6) s[L::S_Q + r + 1] = sr[L::S_Q + r + 1] + ap * dx[r];
    if (stage) {
      s[L::S_A + r] = sr[L::S_A + r] + ap * dr[L::D_A + r];
      s[L::S_BETA + r] = sr[L::S_BETA + r] + ap * dbm[r];
    }
  }
  // (the base pose q (+) ap dq -- an SE(3) exponential -- is NOT done here: one lane walking its ~400 instructions while 63 wait is a tenth
  //  of this kernel's time; ocp_integrate_base_kernel below does it with one LANE per stage.  It reads the base pose and dq, which this
  //  kernel does not touch.)
  if (stage) {
    if (nd->has_u) {
      if (lane >= 40 && lane < 40 + NU) s[L::S_U + lane - 40] = sr[L::S_U + lane - 40] + ap * du[lane - 40];
      if (lane >= 52 && lane < 58) s[L::S_NUP + lane - 52] = sr[L::S_NUP + lane - 52] + ap * nup[lane - 52];
    }
    if (lane < NF) {
      const int c = lane / 3;
      if (nd->active[c]) { s[L::S_F + lane] = sr[L::S_F + lane] + ap * dfs[lane]; s[L::S_MU + lane] = sr[L::S_MU + lane] + ap * dmu[lane]; }
      if (lane < nd->sw_dimi) s[L::S_XI + lane] = sr[L::S_XI + lane] + ap * dr[L::D_XI + lane];       // split_solution.hxx:235-238
    }
  }
}

// K7b: the base pose of SplitSolution::integrate (robot.integrateConfiguration, split_solution.hxx:212-239) for every stage of every instance,
// one lane per stage: q_base <- q_base (+) ap dq_base.  Runs behind K7 on the same stream (either order would do: K7 leaves the base pose and
// dq alone).  Round 4; until then lane 32 of every K7 wavefront did this while the other 63 lanes waited (0.06 of K7's 0.56 ms).
template <typename D>
__global__ __launch_bounds__(64) void ocp_integrate_base_kernel(OcpBuffers B, long total) {
  using L = OcpLayout<D>;
  const long unit = (long)blockIdx.x * 64 + threadIdx.x;
  if (unit >= total) return;
  const int M = B.M;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  if (B.prob->backward_euler != 0 && pos == M - 1) return;      // ParNMPC: the placeholder stage is not integrated (K7 returns there too)
  const long rec = b * B.NS + B.nodes[pos].slot;
  double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ dd = B.dir + rec * L::DIR;
  const double ap = B.step[b * 2];
  double q[7], dq[6], qn[7];
#pragma unroll
  for (int k = 0; k < 7; ++k) q[k] = s[L::S_Q + k];
#pragma unroll
  for (int k = 0; k < 6; ++k) dq[k] = dd[L::D_Q + k];
  lieIntegrateBase(q, dq, ap, qn);
#pragma unroll
  for (int k = 0; k < 7; ++k) s[L::S_Q + k] = qn[k];
}

template <typename D>
__global__ __launch_bounds__(64) void ocp_trial_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NU = D::NU, NF = D::NF;
  __shared__ double dx[NX], du[NU], dfs[NF];
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const bool stage = (pos < M - 1);
  const long rec = b * B.NS + nd->slot;
  const double a = B.ls_alpha[b];
  const double* __restrict__ dd = B.dir + rec * L::DIR;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ st = B.sol_try + rec * L::SOL;
  if (lane < NV) { dx[lane] = dd[L::D_Q + lane]; dx[NV + lane] = dd[L::D_V + lane]; }
  if (stage) {
    if (lane < NU) du[lane] = dd[L::D_U + lane];
    if (lane < NF) dfs[lane] = dd[L::D_F + lane];
  }
  for (int e = lane; e < L::SOL; e += 64) st[e] = s[e];
  waveLdsSync();
  if (lane < NV) {
    const int r = lane;
    st[L::S_V + r] = s[L::S_V + r] + a * dx[NV + r];
    if (r >= 6) st[L::S_Q + r + 1] = s[L::S_Q + r + 1] + a * dx[r];
    if (stage) st[L::S_A + r] = s[L::S_A + r] + a * dd[L::D_A + r];
  }
  if (lane == 32) {
    double qn[7];
    lieIntegrateBase(s + L::S_Q, dx, a, qn);
    for (int k = 0; k < 7; ++k) st[L::S_Q + k] = qn[k];
  }
  double bar = 0.0;
  if (stage) {
    if (nd->has_u && lane >= 40 && lane < 40 + NU) st[L::S_U + lane - 40] = s[L::S_U + lane - 40] + a * du[lane - 40];
    if (lane < NF && nd->active[lane / 3]) st[L::S_F + lane] = s[L::S_F + lane] + a * dfs[lane];
    const double* __restrict__ slack = B.slack + rec * L::CON;
    for (int row = lane; row < L::NCON; row += 64) {
      double g, dg, dslack;
      if (ipmRow<D>(P, nd, row, s, dx, dx + NV, du, dfs, dd + L::D_A, B.ext ? B.ext + rec * L::EXT : nullptr, &g, &dg)) dslack = -dg - (g + slack[row]);
      else if (ipmIdleConeRow<D>(P, nd, row)) dslack = 1.0;
      else continue;
      bar -= log(slack[row] + a * dslack);
    }
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) bar += __shfl_xor(bar, off);
  if (lane == 0) B.merit_stage[rec * 4 + 2] = stage ? nd->dt * P->barrier * bar : 0.0;
}

// totals of the chain: merit[b] = (sum of stage costs + barrier costs, sum of violations)
__global__ __launch_bounds__(64) void ocp_merit_reduce_kernel(OcpBuffers B) {
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long b = blockIdx.x;
  double c = 0.0, v = 0.0;
  for (int i = threadIdx.x; i < M; i += 64) {
    const double* __restrict__ ms = B.merit_stage + (b * B.NS + B.nodes[i].slot) * 4;
    c += ms[0] + ms[2]; v += ms[1];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off); v += __shfl_xor(v, off); }
  if (threadIdx.x == 0) { B.merit[b * 2] = c; B.merit[b * 2 + 1] = v; }
}

// SplitOCP::initConstraints -> setSlackAndDual -> pdipm::SetSlackAndDualPositive (split_ocp.hxx:50-55; pdipm.hxx:13-23)
template <typename D>
__global__ __launch_bounds__(64) void ocp_init_constraints_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NU = D::NU;
  // Every SLOT is initialised, whether or not the current chain uses it (OCPLinearizer::initConstraints,
  // ocp_linearizer.cpp:40-70, covers all N grid stages and the event stages): a later re-discretisation may bring a
  // slot into the chain.  Kind and gating level follow from the slot index alone.
  const OcpProblem* __restrict__ P = B.prob;
  const int NS = B.NS, N = P->N, E = P->E;
  const long su = blockIdx.x;                       // over batch * NS
  const int slot = (int)(su % NS);
  const bool impulse = (slot > N && slot <= N + E);
  const int i = (slot <= N) ? slot + (P->backward_euler ? 1 + P->stage_offset : 0) : (impulse ? -1 : 0);      // grid stage index (ParNMPC: + 1); 0 on aux / lift stages
  const double* __restrict__ s = B.sol + su * L::SOL;
  for (int row = threadIdx.x; row < L::NCON; row += 64) {
    double sl = 1.0, dl = 0.0;
    bool valid;
    double g = 0.0;
    if (row >= L::C_CD) {
      if (B.ext) continue;                              // ContactDistance rows: ocp_ext_init_kernel (needs the kinematics of the feet)
      valid = false;
    } else if (row >= L::C_ACC) {
      const int c = 8 + (row - L::C_ACC) / NU, j = (row - L::C_ACC) % NU;
      valid = ocpRowValid2(P, c, i, impulse);
      if (valid) g = ((c & 1) ? 1.0 : -1.0) * (s[L::S_A + 6 + j] - ocpLimit2(P, c, j));      // joint_acceleration_{lower,upper}_limit.cpp:50-54
    } else if (row < L::C_FRIC) {
      const int c = row / NU, j = row - c * NU;
      valid = ocpRowValid2(P, c, i, impulse);
      if (valid) {
        const double sgn = (c & 1) ? 1.0 : -1.0;
        const double x = (c < 2) ? s[L::S_Q + 7 + j] : ((c < 4) ? s[L::S_V + 6 + j] : s[L::S_U + j]);
        g = sgn * (x - ocpLimit2(P, c, j));
      }
    } else {
      // all contacts, active or not (linearized_friction_cone.cpp:96-104)
      const int fr = row - L::C_FRIC, c = fr / 5, r = fr - 5 * c;
      const int ck = impulse ? P->impulse_cone_kind : P->cone_kind;
      valid = ocpRowValid2(P, 6, i, impulse) && r < coneRows(ck);
      if (valid) { double J[3]; const double f[3] = {s[L::S_F + 3 * c], s[L::S_F + 3 * c + 1], s[L::S_F + 3 * c + 2]}; g = coneRow(ck, P->mu, r, f, J); }
    }
    if (valid) {
      sl = -g;
      sl = slackPositive(sl, P->barrier);      // pdipm.hxx:17-20
      dl = P->barrier / sl;
    }
    B.slack[su * L::CON + row] = sl;
    B.dual[su * L::CON + row] = dl;
  }
}

// setSolution: write `repeat` copies of value[dim] into one field of every stage record
__global__ void ocp_fill_field_kernel(double* __restrict__ sol, int stride, int offset, int dim, long nrec_per_inst, long batch,
                                      const double* __restrict__ value, int per_instance, int repeat) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long width = (long)dim * repeat;
  const long total = batch * nrec_per_inst * width;
  if (idx >= total) return;
  const int e = (int)(idx % width);
  const long rec = idx / width;
  const long b = rec / nrec_per_inst;
  sol[rec * stride + offset + e] = value[(per_instance ? b * dim : 0) + (e % dim)];
}

// warm start: values[nstages][dim] -> field `offset` of the records of slots 0 .. nstages-1 of every instance
// (nodes != nullptr: values[nstages][dim] in CHAIN order, entry p goes to the slot of chain position p)
__global__ void ocp_fill_stages_kernel(double* __restrict__ rec, int stride, int offset, int dim, long NS, int nstages, long batch,
                                       const double* __restrict__ values, const OcpNode* __restrict__ nodes) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long per = (long)nstages * dim;
  if (idx >= batch * per) return;
  const long b = idx / per;
  const int rem = (int)(idx - b * per), i = rem / dim, e = rem - i * dim;
  rec[(b * NS + (nodes ? nodes[i].slot : i)) * stride + offset + e] = values[rem];
}
void ocpFillStages(double* rec, int stride, int offset, int dim, long NS, int nstages, long batch, const double* values, hipStream_t st,
                   const OcpNode* nodes) {
  const long total = batch * nstages * dim;
  hipLaunchKernelGGL(ocp_fill_stages_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rec, stride, offset, dim, NS, nstages, batch, values, nodes);
}

template <typename D>
void OcpLaunch<D>::expandPrimal(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  hipLaunchKernelGGL((ocp_expand_primal_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B);
  hipLaunchKernelGGL(ocp_reduce_steps_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
}
// S4 + K6 + the step-size reduction of an OCPSolver iteration in one launch (the caller decides when: ocp_capi.hip, fusedForward)
template <typename D>
void OcpLaunch<D>::forwardExpand(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, hipStream_t st) {
  (void)M;
  hipLaunchKernelGGL((ocp_forward_expand_kernel<D>), dim3((unsigned)batch), dim3(128), 0, st, B, q0, v0);
}
template <typename D>
void OcpLaunch<D>::expandDualIntegrate(const OcpBuffers& B, long batch, int M, hipStream_t st, hipStream_t st_base) {
  hipLaunchKernelGGL((ocp_expand_dual_integrate_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B);
  const long total = batch * M;                      // K7b: the base poses, one lane per stage
  hipLaunchKernelGGL((ocp_integrate_base_kernel<D>), dim3((unsigned)((total + 63) / 64)), dim3(64), 0, st_base ? st_base : st, B, total);
}
template <typename D>
void OcpLaunch<D>::trialIterate(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  hipLaunchKernelGGL((ocp_trial_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B);
}
template <typename D>
void OcpLaunch<D>::meritReduce(const OcpBuffers& B, long batch, hipStream_t st) {
  hipLaunchKernelGGL(ocp_merit_reduce_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
}
template <typename D>
void OcpLaunch<D>::initConstraints(const OcpBuffers& B, long batch, int NS, hipStream_t st) {
  hipLaunchKernelGGL((ocp_init_constraints_kernel<D>), dim3((unsigned)(batch * NS)), dim3(64), 0, st, B);
}

template <typename D>
void OcpLaunch<D>::single(int kernel_id, const OcpBuffers& B, long batch, int M, hipStream_t st) {
  if (kernel_id == 4) hipLaunchKernelGGL((ocp_expand_primal_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B);
  else if (kernel_id == 5) hipLaunchKernelGGL(ocp_reduce_steps_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
  else expandDualIntegrate(B, batch, M, st);
}

// squared_out != nullptr: the SUM of the squared stage residuals goes there (horizon shards add theirs up before the root)
void ocpKktErrorReduce(const OcpBuffers& B, long batch, hipStream_t st, double* squared_out) {
  hipLaunchKernelGGL(ocp_kkt_error_kernel, dim3((unsigned)batch), dim3(64), 0, st, B, squared_out);
}

void ocpFillField(double* sol, int stride, int offset, int dim, long nrec_per_inst, long batch, const double* value,
                  int per_instance, int repeat, hipStream_t st) {
  const long total = batch * nrec_per_inst * dim * repeat;
  hipLaunchKernelGGL(ocp_fill_field_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sol, stride, offset, dim,
                     nrec_per_inst, batch, value, per_instance, repeat);
}

template void OcpLaunch<LeggedDims<4, 3>>::expandPrimal(const OcpBuffers&, long, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::expandDualIntegrate(const OcpBuffers&, long, int, hipStream_t, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::forwardExpand(const OcpBuffers&, long, int, const double*, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::trialIterate(const OcpBuffers&, long, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::meritReduce(const OcpBuffers&, long, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::initConstraints(const OcpBuffers&, long, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::single(int, const OcpBuffers&, long, int, hipStream_t);

}  // namespace idocp_dev
