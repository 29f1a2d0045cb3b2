// HIP kernels of the fixed-base (UnOCPSolver) hot path for gfx950 / MI355X.
//
// Kernel inventory (SURVEY.md section 2.3 numbering; reference loops they replace):
//   K1  un_linearize_kernel<NV,0>  SplitUnOCP::linearizeOCP for every stage       (src/unocp/unocp_solver.cpp:78-94)
//   S1  un_riccati_backward_kernel UnRiccatiRecursion::backwardRiccatiRecursion   (src/unocp/unriccati_recursion.cpp:39-58)
//   S2  un_riccati_forward_kernel  ... ::forwardRiccatiRecursion                  (src/unocp/unriccati_recursion.cpp:60-65)
//   K2  un_expand_kernel           costate/condensed/slack-dual directions + steps (src/unocp/unocp_solver.cpp:103-115)
//   K3  un_integrate_kernel        updatePrimal / updateDual                      (src/unocp/unocp_solver.cpp:121-133)
//   K4  un_linearize_kernel<NV,1>  computeKKTResidual + squaredNormKKTResidual     (src/unocp/unocp_solver.cpp:184-225)
//
// Work decomposition: wavefronts are 64 lanes wide but the
// per-stage blocks are 7x7, so a wavefront is split into LANE GROUPS that each
// own one stage (K1: 3*NV lanes = one tangent seed per lane; K2/K3: 8 lanes = one
// row per lane) or one OCP instance (S1/S2: 8 lanes = one block column per lane).
// Workgroups are a single wavefront, so the barriers below are wave-local.
#include <hip/hip_runtime.h>

#include "dev_dense.hpp"
#include "dev_rnea_analytic.hpp"
#include "unocp_launch.hpp"

namespace idocp_dev {

// LDS exchange between the lanes of the (only) wavefront of a workgroup: its DS instructions execute in program order, so all that is
// needed is that the compiler keeps that order.  (__syncthreads() also drains the wavefront's outstanding GLOBAL stores and loads --
// s_waitcnt vmcnt(0) --: in un_linearize_kernel, which stores records in every round, a third of the wavefront's time.)
#define WAVE_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// One inequality row of the primal-dual interior point method
// (include/idocp/constraints/pdipm.hxx:13-87; row formulas e.g.
// src/constraints/joint_torques_upper_limit.cpp:48-87).  sgn = -1 lower, +1 upper:
//   g(x) = sgn (x - lim) <= 0,  residual = g + slack,  duality = slack dual - barrier
struct IpmRow {
  double residual, duality;
};
__device__ __forceinline__ IpmRow ipmResidual(double sgn, double x, double lim, double slack, double dual, double barrier) {
  IpmRow r;
  r.residual = sgn * (x - lim) + slack;
  r.duality = slack * dual - barrier;
  return r;
}

// pdipm::FractionToBoundary for one row (pdipm.hxx:52-73)
__device__ __forceinline__ double fractionToBoundary(double rate, double x, double dx, double cur) {
  const double f = -rate * (x / dx);
  return (f > 0.0 && f < 1.0 && f < cur) ? f : cur;
}

// Limit of IPM row `comp` (0..5 = q_lo,q_hi,v_lo,v_hi,u_lo,u_hi), joint k, and its sign.
__device__ __forceinline__ double limitOf(const UnProblem* __restrict__ P, int comp, int k) {
  switch (comp) {
    case 0: return P->q_min[k];
    case 1: return P->q_max[k];
    case 2: return -P->v_max[k];
    case 3: return P->v_max[k];
    case 6: return P->a_min[k];       // JointAccelerationLowerLimit / UpperLimit carry their own bounds
    case 7: return P->a_max[k];
    case 4: return -P->u_max[k];
    default: return P->u_max[k];
  }
}
// ConstraintsData(time_stage) level gating (constraints_data.hpp:18-42) + enabled components
__device__ __forceinline__ bool rowValid(const UnProblem* __restrict__ P, int comp, int stage) {
  if (P->backward_euler) stage += 1 + P->stage_offset;     // UnParNMPC creates stage i with time step i + 1 (unparnmpc_solver.cpp:55-66)
  if (comp < 2) return P->use_q_limits && stage >= 2;
  if (comp < 4) return P->use_v_limits && stage >= 1;
  if (comp == 6) return P->use_a_lower != 0;       // acceleration level: every stage
  if (comp == 7) return P->use_a_upper != 0;
  return P->use_u_limits != 0;
}

// --------------------------------------------------------------------- K1 ----
// MODE 0: linearize + condense -> kkt, dyn records.  MODE 1: KKT residual ->
// err_stage.  Lane (kind, k) of a stage group differentiates ID w.r.t. q_k
// (kind 0), v_k (kind 1) or a_k (kind 2).
// BWD: the backward-Euler stage of UnParNMPC (SplitUnParNMPC / TerminalUnParNMPC::linearizeOCP, split_unparnmpc.hxx:69-102,
// terminal_unparnmpc.hxx:69-102; stateequation::linearizeBackwardEuler[Terminal], state_equation.hxx:111-167): the state
// equation couples to the PREVIOUS stage (the measured state q0, v0 for stage 0), the last stage carries the terminal cost.
// TASK: the cost carries a TaskSpace3DCost / TaskSpace6DCost (dev_task.hpp); lane (0, k) adds dt JJ^T W diff to lq[k] and
// column k of dt JJ^T W JJ to Qqq.
// ZAX: every joint axis of the chain is +z (iiwa14): the compile-time variant of the rigid-body sweep (dev_rbd.hpp).
// -DK1_PROF: cycle counts per section of un_linearize_kernel<., 0> (sampled wavefronts), read with idocp_debug_k1_prof -- a developer aid
#ifdef K1_PROF
__device__ unsigned long long g_k1_prof[16];
#define K1_T(i) do { if (MODE == 0) { const unsigned long long t_now = __builtin_readcyclecounter(); t_acc[i] += t_now - t_prev; t_prev = t_now; } } while (0)
#else
#define K1_T(i) do { } while (0)
#endif
#ifndef K1_ROUNDS
#define K1_ROUNDS 3
#endif
#ifndef K1_WAVES
#define K1_WAVES 2
#endif
template <int NV, int MODE, bool BWD = false, bool TASK = false, bool ZAX = false>
__global__ __launch_bounds__(64, K1_WAVES) void un_linearize_kernel(UnBuffers B, const double* __restrict__ q0 = nullptr,
                                                          const double* __restrict__ v0 = nullptr) {
  using L = UnLayout<NV>;
  constexpr int LPS = 3 * NV;        // lanes per stage (phase B and the rest of the kernel)
  constexpr int SPW = 64 / LPS;      // stages per round
  constexpr int ROUNDS = K1_ROUNDS;
  constexpr int SPA = SPW * ROUNDS;  // stages per wavefront: phase A of the analytic recursion has a lane per (stage, joint)
  static_assert(SPA * NV <= 64, "whole rounds");
  constexpr int BLK = RneaBlock::LEN;
  // s_dyn: the dyn records of the round's stages as they go to memory -- dID/d(q | v | a) [kind][c * NV + r], which the condensation reads
  // back, then ID, lu, diag(Quu).  s_po: the blocks of phase A, then (the rows are assembled by then) the kkt records of the round.
  // Both records leave the wavefront as 16-byte pieces of consecutive lanes: written element by element from the lanes that compute
  // them, a store instruction touched ~60 different cache lines, and the stores were a third of the kernel's time.
  // (the chain's constants and the sines / cosines are read by phase A only: they share the storage of s_dyn)
  struct PhaseALds { ChainConsts<NV> model; double cs[SPA][NV][2]; };
  union DynOrPhaseA { double dyn[SPW][L::DYN]; PhaseALds a; };
  __shared__ __attribute__((aligned(16))) DynOrPhaseA s_da;
  union PubOut { double pub[SPW][NV][BLK]; double out[SPW][L::KKT]; };
  __shared__ __attribute__((aligned(16))) PubOut s_po;
  static_assert(L::KKT % 2 == 0 && L::DYN % 2 == 0 && L::D_DV == NV * NV && L::D_DA == 2 * NV * NV && L::D_LU == L::D_ID + NV && L::D_QUU == L::D_LU + NV, "record images");
  // the solution records of the wavefront's stages (and the part of each successor's record the stage reads: lmd, gmm, q, v -- its
  // first 4 NV numbers, contiguous with the stage's own record) arrive with 16-byte loads issued back to back at the top of the kernel:
  // round 2 read them field by field where they were needed, and the wavefronts spent 57 % of their cycles in s_waitcnt (SQ_WAIT_ANY,
  // profiles/r03_iiwa14_pmc_sq.txt)
  constexpr int IN_SN = L::SOL, IN_LEN = L::SOL + 4 * NV;
  static_assert(L::S_LMD == 0 && L::S_GMM == NV && L::S_Q == 2 * NV && L::S_V == 3 * NV, "the successor's fields read here are a prefix of its record");
  __shared__ __attribute__((aligned(16))) double s_in[SPA][IN_LEN];
  __shared__ long s_inst[SPA];
  __shared__ int s_stage[SPA];
  __shared__ double s_tJ[TASK ? SPW : 1][6][NV];     // TASK: the columns JJ[:, k] of the stage group
  __shared__ double s_task[TASK ? SPA : 1][TASK ? 54 : 1];      // TASK: the frame's world placement (12), diff (6), Jlog6 (36), per stage
  __shared__ double s_err[SPW][LPS];
  __shared__ double s_tau[SPA][NV];
  __shared__ double s_lu[SPW][NV], s_quu[SPW][NV];      // torque-level rows of the stage group: lu and diag(Quu)
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x;
  const long total = (long)P->batch * N;
  const long unit0 = (long)blockIdx.x * SPA;
#ifdef K1_PROF
  unsigned long long t_prev = __builtin_readcyclecounter();
  unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  {
    typedef double in_d2 __attribute__((ext_vector_type(2)));
    static_assert(IN_LEN % 2 == 0 && IN_LEN / 2 <= 64, "one 16-byte piece per lane and stage");
    in_d2 rs[SPA];
    // (instance, stage) of the wavefront's units: one 64-bit division, then counted up (unit = instance * N + stage)
    long bb = unit0 / N;
    int ii = (int)(unit0 - bb * N);
    long uu = unit0;
#pragma unroll
    for (int gg = 0; gg < SPA; ++gg) {
      const double* sg = B.sol + (bb * (N + 1) + ii) * L::SOL;
      rs[gg] = reinterpret_cast<const in_d2*>(sg)[lane < IN_LEN / 2 ? lane : 0];
      if (lane == 0) { s_inst[gg] = bb; s_stage[gg] = ii; }
      if (uu + 1 < total) { ++uu; if (++ii == N) { ii = 0; ++bb; } }      // (past the end: the last unit again, masked out below)
    }
    s_da.a.model.load(B.model, lane, 64);
#pragma unroll
    for (int gg = 0; gg < SPA; ++gg)
      if (lane < IN_LEN / 2) reinterpret_cast<in_d2*>(&s_in[gg][0])[lane] = rs[gg];
    WAVE_SYNC();
  }
  K1_T(0);

  // ---- inverse dynamics and its derivatives, phase A: lane (stage sA of the wavefront, joint jA) ----
  // (round 3: ONE analytic evaluation per stage, dev_rnea_analytic.hpp, instead of a forward-mode sweep per lane -- rneaChain<Dual>, kept
  //  in dev_rbd.hpp for the kernels that want a single tangent; its seed-independent part once per (stage, joint), not once per lane)
  const int sA0 = lane / NV;
  const int sA = sA0 < SPA ? sA0 : SPA - 1;
  const int jA = sA0 < SPA ? lane - sA0 * NV : 0;
  if (sA0 < SPA) {
    double sj, cj;
    sincos(s_in[sA][L::S_Q + jA], &sj, &cj);
    s_da.a.cs[sA][jA][0] = cj; s_da.a.cs[sA][jA][1] = sj;
  }
  WAVE_SYNC();
  double blk[BLK];
  double kinA[TASK ? 12 : 1];         // TASK: world placement (R, p) of the lane's joint
  rneaDerivPhaseA<NV, ZAX>(&s_da.a.model, &s_da.a.cs[sA][0][0], &s_in[sA][L::S_V], &s_in[sA][L::S_A], jA, blk, TASK ? kinA : nullptr);
  if (TASK) {
    // The frame's world placement, diff = log6(M_ref^-1 M_frame) and Jlog6 of a stage are evaluated ONCE, by the phase-A lane of the
    // frame's joint -- the nine stages of the wavefront side by side -- out of the walk of the analytic recursion.  (Round 2 walked the
    // chain a second time in every lane: taskSpaceColumn, dev_task.hpp -- still what the terminal and the line-search kernels use.)
    const TaskCost& tc = P->task;
    if (sA0 < SPA && jA == tc.joint) {
      const double* __restrict__ ref = B.task_ref + 12 * s_stage[sA];
      double fR[9], fp[3], e[3];
      lieMatmul3(kinA, tc.R, fR);
      lieMatvec3(kinA, tc.p, fp);
#pragma unroll
      for (int r = 0; r < 3; ++r) { fp[r] += kinA[9 + r]; e[r] = fp[r] - ref[9 + r]; }
      double* o = s_task[sA];
#pragma unroll
      for (int r = 0; r < 9; ++r) o[r] = fR[r];
#pragma unroll
      for (int r = 0; r < 3; ++r) o[9 + r] = fp[r];
      if (tc.dim == 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { o[12 + r] = e[r]; o[15 + r] = 0.0; }
      } else {
        double Rd[9], pd[3], df[6], J[36];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
          for (int c = 0; c < 3; ++c) Rd[3 * r + c] = ref[r] * fR[c] + ref[3 + r] * fR[3 + c] + ref[6 + r] * fR[6 + c];
          pd[r] = ref[r] * e[0] + ref[3 + r] * e[1] + ref[6 + r] * e[2];
        }
        lieLog6Jlog6(Rd, pd, df, J);
#pragma unroll
        for (int r = 0; r < 6; ++r) o[12 + r] = df[r];
#pragma unroll
        for (int r = 0; r < 36; ++r) o[18 + r] = J[r];
      }
    }
  }
  K1_T(1);

  // ---- phase B, SPW stages at a time: lane (stage g of the round, kind, k) assembles row k of its matrix d tau / d (q | v | a) ----
  // The rows of all rounds are assembled before the rest of the kernel runs: a lane then carries 7 numbers per round still to come
  // instead of the 55 of its phase-A block (which did not fit beside the rest of the kernel: the block's spilled part came back from
  // scratch memory behind the previous round's record stores).
  const int g0 = lane / LPS;
  const int g = g0 < SPW ? g0 : SPW - 1;
  const int seed = lane - g0 * LPS;
  const int kind = (g0 < SPW) ? seed / NV : 0;
  const int k = (g0 < SPW) ? seed - kind * NV : 0;
  if (sA0 < SPA) s_tau[sA][jA] = blk[RneaBlock::TAU];
  // slack and dual of the IPM rows the lane evaluates in each round (two rows of its seed kind): requested here, wanted far below
  // (not beside a task-space cost: its registers are taken)
  constexpr bool ROWS_AHEAD = !TASK;
  double sl_all[ROWS_AHEAD ? ROUNDS : 1][2], du_all[ROWS_AHEAD ? ROUNDS : 1][2];
#pragma unroll
  for (int rho = 0; rho < (ROWS_AHEAD ? ROUNDS : 0); ++rho) {
    long u = unit0 + rho * SPW + g; if (u >= total) u = total - 1;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      sl_all[rho][cc] = B.slack[u * L::CON + (2 * kind + cc) * NV + k];
      du_all[rho][cc] = B.dual[u * L::CON + (2 * kind + cc) * NV + k];
    }
  }
  double rows[ROUNDS][NV];
  double kinS[TASK ? ROUNDS : 1][6];       // TASK: the motion subspace of joint k, per round
#pragma unroll
  for (int rho = 0; rho < ROUNDS; ++rho) {
    WAVE_SYNC();                       // (the previous round has read the table)
    if (sA0 < SPA && sA0 / SPW == rho) {
      double* o = &s_po.pub[sA0 - rho * SPW][jA][0];
#pragma unroll
      for (int e = 0; e < BLK; ++e) o[e] = blk[e];
    }
    WAVE_SYNC();
    rneaDerivPhaseB<NV>(&s_po.pub[g][0][0], BLK, kind, k, rows[rho]);
    if (TASK) {
#pragma unroll
      for (int e = 0; e < 6; ++e) kinS[rho][e] = s_po.pub[g][k][RneaBlock::S + e];
    }
  }
  K1_T(2);

  // ---- the rest of the kernel, SPW stages at a time ----
#pragma unroll 1
  for (int rho = 0; rho < ROUNDS; ++rho) {
  WAVE_SYNC();                       // (the previous round has read the LDS tables written below)
  long unit = unit0 + rho * SPW + g;
  const bool active = (g0 < SPW) && (unit < total);
  if (unit >= total) unit = total - 1;
  const long b = s_inst[rho * SPW + g];
  const int i = s_stage[rho * SPW + g];
  const double* __restrict__ s_g = B.sol + (b * (N + 1) + i) * L::SOL;
  const double* s = &s_in[rho * SPW + g][0];
  const double* sn = &s_in[rho * SPW + g][IN_SN];
  // the IPM rows this lane evaluates: components (0, 1) / (2, 3) / (4, 5) of joint k on the q / v / a seed lane
  double sl_own[2], du_own[2];
  if (ROWS_AHEAD) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) { sl_own[cc] = sl_all[0][cc]; du_own[cc] = du_all[0][cc]; }
#pragma unroll
    for (int r = 0; r + 1 < (ROWS_AHEAD ? ROUNDS : 1); ++r) {
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) { sl_all[r][cc] = sl_all[r + 1][cc]; du_all[r][cc] = du_all[r + 1][cc]; }
    }
  } else {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      sl_own[cc] = B.slack[unit * L::CON + (2 * kind + cc) * NV + k];
      du_own[cc] = B.dual[unit * L::CON + (2 * kind + cc) * NV + k];
    }
  }
  // the rows of the round into the image of the dyn record (element (r, c) of a matrix at c * NV + r): the lanes read columns below
  if (g0 < SPW) {
#pragma unroll
    for (int c = 0; c < NV; ++c) s_da.dyn[g][kind * NV * NV + c * NV + k] = rows[0][c];
  }
#pragma unroll
  for (int r = 0; r + 1 < ROUNDS; ++r) {
#pragma unroll
    for (int c = 0; c < NV; ++c) rows[r][c] = rows[r + 1][c];
  }
  WAVE_SYNC();
  K1_T(3);
  double tau_d[NV], ID[NV];
#pragma unroll
  for (int r = 0; r < NV; ++r) { tau_d[r] = s_da.dyn[g][kind * NV * NV + k * NV + r]; ID[r] = s_tau[rho * SPW + g][r] - s[L::S_U + r]; }
  double kin[TASK ? 18 : 1];        // TASK: the motion subspace S = (p x w, w) of joint k at [12, 18), from the walk
  if (TASK) {
#pragma unroll
    for (int e = 0; e < 6; ++e) kin[12 + e] = kinS[0][e];
#pragma unroll
    for (int r = 0; r + 1 < ROUNDS; ++r) {
#pragma unroll
      for (int e = 0; e < 6; ++e) kinS[r][e] = kinS[r + 1][e];
    }
  }

  // ---- task-space cost: gradient element k and the weighted column dt W JJ[:, k] (kept by the kind-0 lanes) ----
  double task_g = 0.0, task_wc[6];
  if (TASK) {
    // every lane turns its own motion subspace into its column JJ[:, k] = Jlog6 (R_f^T (S.l + S.a x p_f), R_f^T S.a)  (3D: the
    // world-frame linear part) with the frame terms of its stage (s_task, evaluated behind phase A)
    double tdiff[6], tcol[6];
    {
      const TaskCost& tc = P->task;
      const double* o = s_task[rho * SPW + g];
#pragma unroll
      for (int r = 0; r < 6; ++r) tdiff[r] = o[12 + r];
      // world-frame velocity of the frame origin per unit rate of joint k: S.l + S.a x p_f; zero past the frame's joint
      const bool moves = k <= tc.joint;
      const double Sl[3] = {kin[12], kin[13], kin[14]}, Sa[3] = {kin[15], kin[16], kin[17]};
      const double lw[3] = {Sl[0] + Sa[1] * o[11] - Sa[2] * o[10], Sl[1] + Sa[2] * o[9] - Sa[0] * o[11], Sl[2] + Sa[0] * o[10] - Sa[1] * o[9]};
      if (tc.dim == 3) {
#pragma unroll
        for (int r = 0; r < 3; ++r) { tcol[r] = moves ? lw[r] : 0.0; tcol[3 + r] = 0.0; }
      } else {
        double tw[6];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          tw[r] = o[r] * lw[0] + o[3 + r] * lw[1] + o[6 + r] * lw[2];
          tw[3 + r] = o[r] * Sa[0] + o[3 + r] * Sa[1] + o[6 + r] * Sa[2];
        }
        const double* J = o + 18;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          tcol[r] = moves ? J[r] * tw[0] + J[r + 6] * tw[1] + J[r + 12] * tw[2] + J[r + 18] * tw[3] + J[r + 24] * tw[4] + J[r + 30] * tw[5] : 0.0;
          tcol[3 + r] = moves ? J[3 + r + 18] * tw[3] + J[3 + r + 24] * tw[4] + J[3 + r + 30] * tw[5] : 0.0;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double wc = P->task.weight[c] * tcol[c];
      task_g += wc * tdiff[c];
      task_wc[c] = dt * wc;
      if (MODE == 0 && g0 < SPW && kind == 0) s_tJ[g][c][k] = tcol[c];
    }
    if (MODE == 0) WAVE_SYNC();
  }

  // ---- torque-level rows: needed by every lane (lu_condensed couples all rows) ----
  // UnconstrainedDynamics::linearize/condense (unconstrained_dynamics.hxx:55-94)
  // (each row is evaluated ONCE, by the a-seed lane of its joint, and shared through LDS: with every lane evaluating all NV rows the
  //  28 FP64 divisions of the IPM terms were a third of the kernel's vector instructions)
  double lu_c[NV], quu[NV], lu_mine = 0.0;
  double e_con = 0.0;   // MODE 1: squared IPM residuals owned by this lane
  if (g0 < SPW && kind == 2) {
    const int r = k;
    const double u = s[L::S_U + r];
    double lu = dt * P->u_weight[r] * (u - P->u_ref[r]) - dt * s[L::S_BETA + r];
    double h = dt * P->u_weight[r];
    if (P->use_u_limits) {
#pragma unroll
      for (int c = 4; c < 6; ++c) {
        const double sgn = (c == 4) ? -1.0 : 1.0;
        const double sl = sl_own[c - 4], du = du_own[c - 4];
        const IpmRow row = ipmResidual(sgn, u, limitOf(P, c, r), sl, du, P->barrier);
        lu += sgn * dt * du;
        if (MODE == 0) {
          const double isl = recipNewton(sl);      // (slacks are positive: estimate + two Newton steps instead of two division sequences)
          lu += sgn * dt * (du * row.residual - row.duality) * isl;
          h += dt * du * isl;
        } else {
          e_con += row.residual * row.residual + row.duality * row.duality;
        }
      }
    }
    s_lu[g][r] = lu; s_quu[g][r] = h;
  }
  WAVE_SYNC();
#pragma unroll
  for (int r = 0; r < NV; ++r) { quu[r] = s_quu[g][r]; lu_c[r] = s_lu[g][r] + quu[r] * ID[r]; }
  lu_mine = s_lu[g][k];
  K1_T(4);

  // ---- this lane's own gradient element: lq[k] / lv[k] / la[k] ----
  // cost (configuration_space_cost.cpp:292-310), dual residual, state equation
  // (state_equation.hxx:12-37), inverse-dynamics multiplier (unconstrained_dynamics.hxx:55-65)
  double l, h, F = 0.0;
  {
    const double qk = s[L::S_Q + k], vk = s[L::S_V + k], ak = s[L::S_A + k];
    const double lmd = s[L::S_LMD + k], gmm = s[L::S_GMM + k];
    const double lmdn = sn[L::S_LMD + k], gmmn = sn[L::S_GMM + k];
    double x, w, ref;
    const bool term = BWD && P->has_terminal && (i == N - 1);
    if (kind == 0) {
      x = qk; w = P->q_weight[k]; ref = P->q_ref[k];
      if (BWD) {
        l = (term ? 0.0 : lmdn) - lmd;
        F = (i > 0 ? (s_g - L::SOL)[L::S_Q + k] : q0[b * NV + k]) - qk + dt * vk;
      } else {
        l = lmdn - lmd;
        F = qk - sn[L::S_Q + k] + dt * vk;
      }
    } else if (kind == 1) {
      x = vk; w = P->v_weight[k]; ref = P->v_ref[k];
      if (BWD) {
        l = dt * lmd - gmm + (term ? 0.0 : gmmn);
        F = (i > 0 ? (s_g - L::SOL)[L::S_V + k] : v0[b * NV + k]) - vk + dt * ak;
      } else {
        l = dt * lmdn + gmmn - gmm;
        F = vk + dt * ak - sn[L::S_V + k];
      }
    } else {
      x = ak; w = P->a_weight[k]; ref = 0.0;
      l = BWD ? dt * gmm : dt * gmmn;
    }
    l += dt * w * (x - ref);
    h = dt * w;
    if (TASK && kind == 0) l += dt * task_g;     // TaskSpace*Cost::computeStageCostDerivatives
    if (TASK && kind == 0 && B.task_xs) l += B.task_xs[((long)b * N + i) * L::TASK + L::T_G + k];      // ... of the task_extra components (un_task_terminal_kernel<.., STAGES>)
    if (term && kind < 2) {          // computeTerminalCostDerivatives / Hessian (configuration_space_cost.cpp:313-329, 368-380)
      const double wf = (kind == 0) ? P->qf_weight[k] : P->vf_weight[k];
      l += wf * (x - ref);
      h += wf;
    }
    {      // own-row limits: position / velocity rows on the q / v seed lanes, acceleration rows (components 6, 7) on the a seed lanes
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const int c = kind < 2 ? 2 * kind + cc : 6 + cc;
        if (rowValid(P, c, i)) {
          const double sgn = (cc == 0) ? -1.0 : 1.0;
          const double sl = c < 6 ? sl_own[cc] : B.slack_a[unit * 2 * NV + (c - 6) * NV + k], du = c < 6 ? du_own[cc] : B.dual_a[unit * 2 * NV + (c - 6) * NV + k];
          const IpmRow row = ipmResidual(sgn, x, limitOf(P, c, k), sl, du, P->barrier);
          l += sgn * dt * du;
          if (MODE == 0) {
            const double isl = recipNewton(sl);
            l += sgn * dt * (du * row.residual - row.duality) * isl;
            h += dt * du * isl;
          } else {
            e_con += row.residual * row.residual + row.duality * row.duality;
          }
        }
      }
    }
    double dotb = 0.0;
#pragma unroll
    for (int r = 0; r < NV; ++r) dotb += tau_d[r] * s[L::S_BETA + r];
    l += dt * dotb;
  }
  K1_T(5);

  if (MODE == 1) {
    // SplitUnOCP::squaredNormKKTResidual (split_unocp.hxx:164-174)
    double e = l * l + dt * dt * e_con;
    if (kind == 0) e += F * F + dt * dt * ID[k] * ID[k];
    if (kind == 1) e += F * F + lu_mine * lu_mine;
    if (g0 < SPW) s_err[g][seed] = active ? e : 0.0;
    WAVE_SYNC();
    if (active && seed == 0) {
      double sum = 0.0;
#pragma unroll
      for (int j = 0; j < LPS; ++j) sum += s_err[g][j];
      B.err_stage[b * (N + 1) + i] = sum;
    }
    continue;
  }

  // ---- condensation: l += dID^T lu_c ; Q = dID^T diag(Quu) dID + diag ----
  double dotc = 0.0;
#pragma unroll
  for (int r = 0; r < NV; ++r) dotc += tau_d[r] * lu_c[r];
  const double l_c = l + dotc;

  double* kk = &s_po.out[g][0];        // (LDS images of the records)
  double* dy = &s_da.dyn[g][0];
  double dcol[NV];
#pragma unroll
  for (int r = 0; r < NV; ++r) dcol[r] = quu[r] * tau_d[r];
  // block (kind1, kind2): Q_{kind1,kind2}[:, k2] for this lane's (kind2, k2) = (kind, k)
#pragma unroll 1
  for (int k1kind = 0; k1kind < 3; ++k1kind) {
    // destination offset inside the kkt record, -1 if the block is not needed
    int dst;
    if (k1kind == 0) dst = (kind == 0) ? L::K_QQQ : ((kind == 1) ? L::K_QQV : -1);
    else if (k1kind == 1) dst = (kind == 1) ? L::K_QVV : -1;
    else dst = (kind == 0) ? L::K_QAQ : ((kind == 1) ? L::K_QAV : L::K_QAA);
    const double* A = &s_da.dyn[g][k1kind * NV * NV];
#pragma unroll
    for (int k1 = 0; k1 < NV; ++k1) {
      double acc = (k1kind == kind && k1 == k) ? h : 0.0;
#pragma unroll
      for (int r = 0; r < NV; ++r) acc += A[k1 * NV + r] * dcol[r];
      if (TASK && k1kind == 0 && kind == 0) {     // TaskSpace*Cost::computeStageCostHessian: Qqq += dt JJ^T W JJ
#pragma unroll
        for (int c = 0; c < 6; ++c) acc += s_tJ[g][c][k1] * task_wc[c];
        if (B.task_xs) acc += B.task_xs[((long)b * N + i) * L::TASK + L::T_H + k * NV + k1];      // the task_extra components
      }
      if (g0 < SPW && dst >= 0) {
        if (k1kind != kind) kk[dst + k * NV + k1] = acc;
        else if (k1 <= k) kk[dst + k * (k + 1) / 2 + k1] = acc;       // diagonal block: its upper triangle
      }
    }
  }
  if (g0 < SPW) {
    if (kind == 0) { kk[L::K_LQ + k] = l_c; kk[L::K_FQ + k] = F; dy[L::D_ID + k] = ID[k]; }
    else if (kind == 1) { kk[L::K_LV + k] = l_c; kk[L::K_FV + k] = F; dy[L::D_LU + k] = lu_mine; }
    else { kk[L::K_LA + k] = l_c; dy[L::D_QUU + k] = quu[k]; }
  }
  WAVE_SYNC();
  K1_T(6);
  {      // the round's units are consecutive: so are their records
    typedef double out_d2 __attribute__((ext_vector_type(2)));
    const long ubase = unit0 + rho * SPW;
    const long left = total - ubase;
    const int nst = left >= SPW ? SPW : (left > 0 ? (int)left : 0);
    out_d2* __restrict__ gk = reinterpret_cast<out_d2*>(B.kkt + ubase * L::KKT);
    out_d2* __restrict__ gd = reinterpret_cast<out_d2*>(B.dyn + ubase * L::DYN);
    const out_d2* lk = reinterpret_cast<const out_d2*>(&s_po.out[0][0]);
    const out_d2* ld = reinterpret_cast<const out_d2*>(&s_da.dyn[0][0]);
#pragma unroll
    for (int p0 = 0; p0 < SPW * L::KKT / 2; p0 += 64) { const int p = p0 + lane; if (p < nst * (L::KKT / 2)) gk[p] = lk[p]; }
#pragma unroll
    for (int p0 = 0; p0 < SPW * L::DYN / 2; p0 += 64) { const int p = p0 + lane; if (p < nst * (L::DYN / 2)) gd[p] = ld[p]; }
  }
  K1_T(7);
  }  // rounds
#ifdef K1_PROF
  if (MODE == 0 && threadIdx.x == 0 && blockIdx.x % 61 == 0) for (int e = 0; e < 8; ++e) atomicAdd(&g_k1_prof[e], t_acc[e]);
#endif
}

// ------------------------------------------------------- terminal task cost ----
// TaskSpace*Cost::computeTerminalCost / computeTerminalCostDerivatives / computeTerminalCostHessian at stage N of every
// instance (task_space_6d_cost.cpp; terminal_ocp.hxx:50-66, 118-144) -> B.task_term.  8 lanes per instance, lane k = joint k.
// TRIAL: at the line-search trial point q_N + alpha dq_N (cost only is read).
// STAGES = false: the terminal stage N of every instance, ALL components with their terminal weights -> B.task_term (8 lanes per instance).
// STAGES = true (round 6): the stages 0 .. N - 1 of every instance, the task_extra components only, stage weights times dt -> B.task_xs
// (8 lanes per (instance, stage)); the first component's stage terms are formed inside K1.
template <int NV, bool TRIAL, bool STAGES = false>
__global__ __launch_bounds__(64) void un_task_terminal_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  static_assert(NV <= 8, "one joint per lane of an 8-lane group");
  __shared__ double s_cs[8][NV][2];
  __shared__ double s_J[8][6][NV];
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int k0 = lane & 7;
  const int k = k0 < NV ? k0 : NV - 1;
  const long nunits = STAGES ? (long)P->batch * N : (long)P->batch;
  long unit = (long)blockIdx.x * 8 + g;
  const bool active = (unit < nunits) && (k0 < NV);
  if (unit >= nunits) unit = nunits - 1;
  const long inst = STAGES ? unit / N : unit;
  const int stage = STAGES ? (int)(unit - inst * N) : N;
  const double* __restrict__ sN = B.sol + (inst * (N + 1) + stage) * L::SOL;
  double qk = sN[L::S_Q + k];
  if (TRIAL) qk += B.ls_alpha[inst] * (B.dir + (inst * (N + 1) + stage) * L::SOL)[L::S_Q + k];
  double sj, cj;
  sincos(qk, &sj, &cj);
  s_cs[g][k][0] = cj; s_cs[g][k][1] = sj;
  WAVE_SYNC();
  double cost = 0.0, gk = 0.0, hk[NV];
#pragma unroll
  for (int r = 0; r < NV; ++r) hk[r] = 0.0;
  const double scale = STAGES ? P->dt : 1.0;
  for (int tcomp = STAGES ? 1 : 0; tcomp < P->task_n; ++tcomp) {
    const TaskCost& tc = tcomp == 0 ? P->task : P->task_extra[tcomp - 1];
    const double* __restrict__ ref = tcomp == 0 ? B.task_ref + 12 * stage : tc.ref;
    double diff[6], col[6], wc[6];
    taskSpaceColumn<NV>(B.model, tc, &s_cs[g][0][0], ref, k, diff, col);
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      const double w = scale * (STAGES ? tc.weight[c] : tc.weightf[c]);
      cost += 0.5 * w * diff[c] * diff[c];
      wc[c] = w * col[c];
      gk += wc[c] * diff[c];
      s_J[g][c][k] = col[c];
    }
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < NV; ++r) {
#pragma unroll
      for (int c = 0; c < 6; ++c) hk[r] += s_J[g][c][r] * wc[c];
    }
    WAVE_SYNC();
  }
  if (!active) return;
  double* __restrict__ out = STAGES ? B.task_xs + unit * L::TASK : B.task_term + inst * L::TASK;
  if (k == 0) out[L::T_COST] = cost;
  out[L::T_G + k] = gk;
#pragma unroll
  for (int r = 0; r < NV; ++r) out[L::T_H + k * NV + r] = hk[r];
}

// --------------------------------------------------------------------- S1 ----
// Backward Riccati sweep.  8 lanes per OCP instance, lane c owns column c of
// every NV x NV block (registers); whole blocks that other lanes need (rows of
// P, Qaa for the Cholesky, K for K^T G K) go through the group's LDS slab.
template <int NV>
__global__ __launch_bounds__(64) void un_riccati_backward_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  static_assert(NV <= 8, "one block column per lane of an 8-lane group");
  constexpr int NN = NV * NV;
  __shared__ double sh[8][6 * NN + 8];
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int c0 = lane & 7;
  const int c = c0 < NV ? c0 : NV - 1;
  long inst = (long)blockIdx.x * 8 + g;
  const bool active = (inst < P->batch) && (c0 < NV);
  if (inst >= P->batch) inst = P->batch - 1;
  double* pqq = sh[g];
  double* pqv = pqq + NN;
  double* pvv = pqv + NN;
  double* sQaa = pvv + NN;
  double* sKq = sQaa + NN;
  double* sKv = sKq + NN;
  double* sla = sKv + NN;

  // terminal stage: UnRiccatiRecursion::backwardRiccatiRecursionTerminal
  // (unriccati_recursion.cpp:39-47) on TerminalOCP::linearizeOCP (terminal_ocp.hxx:50-66)
  double Pqq[NV], Pqv[NV], Pvv[NV], sq, sv;
  {
    const double* __restrict__ sN = B.sol + (inst * (N + 1) + N) * L::SOL;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      Pqq[r] = (r == c) ? P->qf_weight[c] : 0.0;
      Pqv[r] = 0.0;
      Pvv[r] = (r == c) ? P->vf_weight[c] : 0.0;
    }
    sq = -(P->qf_weight[c] * (sN[L::S_Q + c] - P->q_ref[c]) - sN[L::S_LMD + c]);
    sv = -(P->vf_weight[c] * (sN[L::S_V + c] - P->v_ref[c]) - sN[L::S_GMM + c]);
    if (P->task.dim) {       // terminal TaskSpace*Cost: dense Pqq (un_task_terminal_kernel ran before this sweep)
      const double* __restrict__ tt = B.task_term + inst * L::TASK;
#pragma unroll
      for (int r = 0; r < NV; ++r) Pqq[r] += tt[L::T_H + c * NV + r];
      sq -= tt[L::T_G + c];
    }
    double* __restrict__ rr = B.ric + (inst * (N + 1) + N) * L::RIC;
    if (active) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        rr[L::R_PQV + c * NV + r] = Pqv[r];
        if (r <= c) { rr[L::R_PQQ + L::sym(r, c)] = Pqq[r]; rr[L::R_PVV + L::sym(r, c)] = Pvv[r]; }
        pqq[c * NV + r] = Pqq[r]; pqv[c * NV + r] = Pqv[r]; pvv[c * NV + r] = Pvv[r];
      }
      rr[L::R_SQ + c] = sq; rr[L::R_SV + c] = sv;
    }
  }
  WAVE_SYNC();

  for (int i = N - 1; i >= 0; --i) {
    const double* __restrict__ kk = B.kkt + (inst * N + i) * L::KKT;
    double Qaa[NV], Qaq[NV], Qav[NV], Qqq[NV], Qqv[NV], Qvv[NV], Fq[NV], Fv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      Qaa[r] = kk[L::K_QAA + L::sym(r, c)]; Qaq[r] = kk[L::K_QAQ + c * NV + r]; Qav[r] = kk[L::K_QAV + c * NV + r];
      Qqq[r] = kk[L::K_QQQ + L::sym(r, c)]; Qqv[r] = kk[L::K_QQV + c * NV + r]; Qvv[r] = kk[L::K_QVV + L::sym(r, c)];
      Fq[r] = kk[L::K_FQ + r]; Fv[r] = kk[L::K_FV + r];
    }
    double la = kk[L::K_LA + c];
    const double lq = kk[L::K_LQ + c], lv = kk[L::K_LV + c];

    // BackwardUnRiccatiRecursionFactorizer::factorizeKKTMatrix
    // (backward_unriccati_recursion_factorizer.hxx:29-54), column c of each block
    double PqqFq = 0.0, PqvFv = 0.0, PqvtFq = 0.0, PvvFv = 0.0;
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      const double PqqRow = pqq[r * NV + c];   // Pqq(c, r)
      const double PqvRow = pqv[r * NV + c];   // Pqv(c, r) = Pqv^T(r, c)
      const double PvvRow = pvv[r * NV + c];   // Pvv(c, r) = Pvv^T(r, c)
      Qqq[r] += Pqq[r];
      Qqv[r] += dt * Pqq[r] + Pqv[r];
      Qvv[r] += dt * dt * Pqq[r] + dt * Pqv[r] + dt * PqvRow + Pvv[r];
      Qaq[r] += dt * PqvRow;
      Qav[r] += dt * dt * PqvRow + dt * PvvRow;
      Qaa[r] += dt * dt * Pvv[r];
      PqqFq += PqqRow * Fq[r];
      PqvFv += PqvRow * Fv[r];
      PqvtFq += Pqv[r] * Fq[r];
      PvvFv += PvvRow * Fv[r];
    }
    la += dt * PqvtFq + dt * PvvFv - dt * sv;
    // factorizeRiccatiFactorization, vector part that needs P_{i+1}
    // (backward_unriccati_recursion_factorizer.hxx:78-86; note sv += dt*sq before lq is subtracted)
    double sq_new = sq - PqqFq - PqvFv;
    double sv_new = sv + dt * sq_new - PqvtFq - PvvFv;
    sq_new -= lq;
    sv_new -= lv;

    WAVE_SYNC();          // all lanes finished reading rows of P_{i+1}
    if (active) {
#pragma unroll
      for (int r = 0; r < NV; ++r) sQaa[c * NV + r] = Qaa[r];
      sla[c] = la;
    }
    WAVE_SYNC();

    // SplitUnRiccatiFactorizer::backwardRiccatiRecursion (split_unriccati_factorizer.hxx:30-46):
    // Eigen::LLT(Qaa) done redundantly by every lane, then each lane solves its own columns.
    // (sqrt and 1 / sqrt of a pivot from the hardware estimate + Newton steps, dev_dense.hpp: the library sqrt and the division sequence were
    // 216 of the 1 580 instructions of a stage, all of them on the chain of the pivots)
    double Lm[NV][NV], Linv[NV];
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      double d = sQaa[j * NV + j];
#pragma unroll
      for (int m = 0; m < j; ++m) d -= Lm[j][m] * Lm[j][m];
      ok = ok && (d > 0.0);
      double ljj, inv;
      rsqrtNewton(d > 0.0 ? d : 1.0, inv, ljj);      // (a failed pivot is reported through `ok`; the lanes go on with finite numbers)
      Lm[j][j] = ljj;
      Linv[j] = inv;
#pragma unroll
      for (int r = j + 1; r < NV; ++r) {
        double t = sQaa[j * NV + r];
#pragma unroll
        for (int m = 0; m < j; ++m) t -= Lm[r][m] * Lm[j][m];
        Lm[r][j] = t * inv;
      }
    }
    if (!ok && active && c == 0 && B.status[inst] == 0) B.status[inst] = 1 + i;
    double Kq[NV], Kv[NV], kv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) { Kq[r] = -Qaq[r]; Kv[r] = -Qav[r]; kv[r] = -sla[r]; }
#pragma unroll
    for (int r = 0; r < NV; ++r) {           // forward substitution  L y = b
#pragma unroll
      for (int m = 0; m < r; ++m) { Kq[r] -= Lm[r][m] * Kq[m]; Kv[r] -= Lm[r][m] * Kv[m]; kv[r] -= Lm[r][m] * kv[m]; }
      const double inv = Linv[r];
      Kq[r] *= inv; Kv[r] *= inv; kv[r] *= inv;
    }
#pragma unroll
    for (int r = NV - 1; r >= 0; --r) {      // backward substitution  L^T x = y
#pragma unroll
      for (int m = r + 1; m < NV; ++m) { Kq[r] -= Lm[m][r] * Kq[m]; Kv[r] -= Lm[m][r] * Kv[m]; kv[r] -= Lm[m][r] * kv[m]; }
      const double inv = Linv[r];
      Kq[r] *= inv; Kv[r] *= inv; kv[r] *= inv;
    }
    // GK = Qaa K (backward_unriccati_recursion_factorizer.hxx:68)
    double GKq[NV], GKv[NV];
#pragma unroll
    for (int r = 0; r < NV; ++r) { GKq[r] = 0.0; GKv[r] = 0.0; }
#pragma unroll
    for (int m = 0; m < NV; ++m) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const double a = sQaa[m * NV + r];
        GKq[r] += a * Kq[m];
        GKv[r] += a * Kv[m];
      }
    }
    if (active) {
#pragma unroll
      for (int r = 0; r < NV; ++r) { sKq[c * NV + r] = Kq[r]; sKv[c * NV + r] = Kv[r]; }
    }
    WAVE_SYNC();
    // P = F - K^T G K  (:69-74), column c
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      double a = 0.0, b2 = 0.0, d2 = 0.0;
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        const double kq = sKq[j * NV + r], kvv = sKv[j * NV + r];
        a += kq * GKq[r];
        b2 += kq * GKv[r];
        d2 += kvv * GKv[r];
      }
      Pqq[j] = Qqq[j] - a;
      Pqv[j] = Qqv[j] - b2;
      Pvv[j] = Qvv[j] - d2;
    }
    double qaqk = 0.0, qavk = 0.0;
#pragma unroll
    for (int r = 0; r < NV; ++r) { qaqk += Qaq[r] * kv[r]; qavk += Qav[r] * kv[r]; }
    sq = sq_new - qaqk;                      // (:87-88)
    sv = sv_new - qavk;
    // preserve the symmetry of Pqq, Pvv (:76-77): needs the transposes
    if (active) {
#pragma unroll
      for (int r = 0; r < NV; ++r) { pqq[c * NV + r] = Pqq[r]; pqv[c * NV + r] = Pqv[r]; pvv[c * NV + r] = Pvv[r]; }
    }
    WAVE_SYNC();
#pragma unroll
    for (int r = 0; r < NV; ++r) {
      Pqq[r] = 0.5 * (Pqq[r] + pqq[r * NV + c]);
      Pvv[r] = 0.5 * (Pvv[r] + pvv[r * NV + c]);
    }
    WAVE_SYNC();
    double* __restrict__ rr = B.ric + (inst * (N + 1) + i) * L::RIC;
    double* __restrict__ gg = B.gain + (inst * N + i) * L::GAIN;
    if (active) {
#pragma unroll
      for (int r = 0; r < NV; ++r) {
        pqq[c * NV + r] = Pqq[r]; pvv[c * NV + r] = Pvv[r];
        rr[L::R_PQV + c * NV + r] = Pqv[r];
        if (r <= c) { rr[L::R_PQQ + L::sym(r, c)] = Pqq[r]; rr[L::R_PVV + L::sym(r, c)] = Pvv[r]; }
        gg[L::G_K + c * NV + r] = Kq[r]; gg[L::G_K + NN + c * NV + r] = Kv[r];
      }
      rr[L::R_SQ + c] = sq; rr[L::R_SV + c] = sv;
      gg[L::G_k + c] = kv[c];
    }
    WAVE_SYNC();
  }
}

// --------------------------------------------------------------------- S2 ----
// Forward Riccati sweep: 8 lanes per instance, lane r owns row r.
// d[0] (unocp_solver.cpp:100-101) then split_unriccati_factorizer.hxx:49-57.
template <int NV>
__global__ __launch_bounds__(64) void un_riccati_forward_kernel(UnBuffers B, const double* __restrict__ q0,
                                                               const double* __restrict__ v0) {
  using L = UnLayout<NV>;
  constexpr int NN = NV * NV;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int r0 = lane & 7;
  const int r = r0 < NV ? r0 : NV - 1;
  long inst = (long)blockIdx.x * 8 + g;
  const bool active = (inst < P->batch) && (r0 < NV);
  if (inst >= P->batch) inst = P->batch - 1;
  const double* __restrict__ s0 = B.sol + inst * (N + 1) * L::SOL;
  double dq = q0[inst * NV + r] - s0[L::S_Q + r];
  double dv = v0[inst * NV + r] - s0[L::S_V + r];
  // the operands of a stage (row r of K, k, Fq, Fv: 2 NV + 3 doubles per lane) do not depend on the sweep: they are fetched one stage
  // ahead, so that the chain dq -> da -> dq_next never waits for memory
  double Kq[NV], Kv[NV], kr, fq, fv;
  auto fetch = [&](int i, double (&kq)[NV], double (&kv)[NV], double& k0, double& f0, double& f1) {
    const double* __restrict__ gg = B.gain + (inst * N + i) * L::GAIN;
    const double* __restrict__ kk = B.kkt + (inst * N + i) * L::KKT;
#pragma unroll
    for (int c = 0; c < NV; ++c) { kq[c] = gg[L::G_K + c * NV + r]; kv[c] = gg[L::G_K + NN + c * NV + r]; }
    k0 = gg[L::G_k + r]; f0 = kk[L::K_FQ + r]; f1 = kk[L::K_FV + r];
  };
  fetch(0, Kq, Kv, kr, fq, fv);
  for (int i = 0; i < N; ++i) {
    double Kqn[NV], Kvn[NV], krn, fqn, fvn;
    fetch(i + 1 < N ? i + 1 : i, Kqn, Kvn, krn, fqn, fvn);
    double da = kr;
#pragma unroll
    for (int c = 0; c < NV; ++c) {
      const double dqc = __shfl(dq, (g << 3) + c);
      const double dvc = __shfl(dv, (g << 3) + c);
      da += Kq[c] * dqc + Kv[c] * dvc;
    }
    double* __restrict__ dd = B.dir + (inst * (N + 1) + i) * L::SOL;
    if (active) { dd[L::S_Q + r] = dq; dd[L::S_V + r] = dv; dd[L::S_A + r] = da; }
    const double dqn = fq + dq + dt * dv;
    const double dvn = fv + dv + dt * da;
    dq = dqn; dv = dvn;
#pragma unroll
    for (int c = 0; c < NV; ++c) { Kq[c] = Kqn[c]; Kv[c] = Kvn[c]; }
    kr = krn; fq = fqn; fv = fvn;
  }
  double* __restrict__ dd = B.dir + (inst * (N + 1) + N) * L::SOL;
  if (active) { dd[L::S_Q + r] = dq; dd[L::S_V + r] = dv; }
}

// --------------------------------------------------------------------- K2 ----
// Direction expansion for every stage incl. the terminal one: 8 lanes per stage,
// lane r owns row r.  Costate direction (split_unriccati_factorizer.hxx:60-68),
// condensed direction (unconstrained_dynamics.hxx:97-106), slack/dual directions
// (joint_*_limit.cpp computeSlackAndDualDirection; pdipm.hxx:76-81) and the
// fraction-to-boundary step sizes of the stage (pdipm.hxx:52-73).
template <int NV>
__global__ __launch_bounds__(64) void un_expand_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int r0 = lane & 7;
  const int r = r0 < NV ? r0 : NV - 1;
  const long total = (long)P->batch * (N + 1);
  long unit = (long)blockIdx.x * 8 + g;
  const bool active = (unit < total) && (r0 < NV);
  if (unit >= total) unit = total - 1;
  const long b = unit / (N + 1);
  const int i = (int)(unit - b * (N + 1));
  double* __restrict__ dd = B.dir + unit * L::SOL;
  const double* __restrict__ rr = B.ric + unit * L::RIC;
  double dq[NV], dv[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) { dq[c] = dd[L::S_Q + c]; dv[c] = dd[L::S_V + c]; }
  double dlmd = -rr[L::R_SQ + r], dgmm = -rr[L::R_SV + r];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    dlmd += rr[L::R_PQQ + L::sym(r, c)] * dq[c] + rr[L::R_PQV + c * NV + r] * dv[c];
    dgmm += rr[L::R_PQV + r * NV + c] * dq[c] + rr[L::R_PVV + L::sym(r, c)] * dv[c];
  }
  if (active) { dd[L::S_LMD + r] = dlmd; dd[L::S_GMM + r] = dgmm; }
  if (i == N) return;          // uniform within the 8-lane group; other groups continue

  const long su = b * N + i;
  const double* __restrict__ dy = B.dyn + su * L::DYN;
  const double* __restrict__ s = B.sol + unit * L::SOL;
  double du = dy[L::D_ID + r];
#pragma unroll
  for (int c = 0; c < NV; ++c) {
    du += dy[L::D_DQ + c * NV + r] * dq[c] + dy[L::D_DV + c * NV + r] * dv[c] + dy[L::D_DA + c * NV + r] * dd[L::S_A + c];
  }
  const double dbeta = (dy[L::D_LU + r] + dy[L::D_QUU + r] * du) / dt;
  if (active) { dd[L::S_U + r] = du; dd[L::S_BETA + r] = dbeta; }

  const double* __restrict__ slack = B.slack + su * L::CON;
  const double* __restrict__ dual = B.dual + su * L::CON;
  double ps = 1.0, ds = 1.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {      // 6, 7: joint acceleration limits (rows in slack_a / dual_a, joint_acceleration_{lower,upper}_limit.cpp:78-93)
    if (!rowValid(P, c, i)) continue;
    const double sgn = (c & 1) ? 1.0 : -1.0;
    const double x = (c < 2) ? s[L::S_Q + r] : ((c < 4) ? s[L::S_V + r] : (c < 6 ? s[L::S_U + r] : s[L::S_A + r]));
    const double dx = (c < 2) ? dq[r] : ((c < 4) ? dv[r] : (c < 6 ? du : dd[L::S_A + r]));
    const double sl = c < 6 ? slack[c * NV + r] : B.slack_a[su * 2 * NV + (c - 6) * NV + r], dl = c < 6 ? dual[c * NV + r] : B.dual_a[su * 2 * NV + (c - 6) * NV + r];
    const IpmRow row = ipmResidual(sgn, x, limitOf(P, c, r), sl, dl, P->barrier);
    const double dslack = -sgn * dx - row.residual;
    const double ddual = -(dl * dslack + row.duality) * recipNewton(sl);
    ps = fractionToBoundary(P->fraction_rate, sl, dslack, ps);
    ds = fractionToBoundary(P->fraction_rate, dl, ddual, ds);
  }
  if (!active) { ps = 1.0; ds = 1.0; }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    ps = fmin(ps, __shfl_xor(ps, off));
    ds = fmin(ds, __shfl_xor(ds, off));
  }
  if (active && r0 == 0) { B.step_stage[su * 2] = ps; B.step_stage[su * 2 + 1] = ds; }
}

// min over the stages of one instance (unocp_solver.cpp:114-115): one wavefront per instance
__global__ __launch_bounds__(64) void un_reduce_steps_kernel(UnBuffers B) {
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const long b = blockIdx.x;
  double ps = 1.0, ds = 1.0;
  for (int i = threadIdx.x; i < N; i += 64) {
    ps = fmin(ps, B.step_stage[(b * N + i) * 2]);
    ds = fmin(ds, B.step_stage[(b * N + i) * 2 + 1]);
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    ps = fmin(ps, __shfl_xor(ps, off));
    ds = fmin(ds, __shfl_xor(ds, off));
  }
  if (threadIdx.x == 0) { B.step[b * 2] = ps; B.step[b * 2 + 1] = ds; }
}

// --------------------------------------------------------------------- K3 ----
// updatePrimal / updateDual (split_unocp.hxx:123-138; split_solution.hxx:215-240;
// terminal_ocp.hxx:100-110).  dslack / ddual are recomputed from the pre-update
// state instead of being round-tripped through HBM.
template <int NV>
__global__ __launch_bounds__(64) void un_integrate_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int r0 = lane & 7;
  const int r = r0 < NV ? r0 : NV - 1;
  const long total = (long)P->batch * (N + 1);
  long unit = (long)blockIdx.x * 8 + g;
  const bool active = (unit < total) && (r0 < NV);
  if (unit >= total) unit = total - 1;
  const long b = unit / (N + 1);
  const int i = (int)(unit - b * (N + 1));
  const double ap = B.step[b * 2], ad = B.step[b * 2 + 1];
  double* __restrict__ s = B.sol + unit * L::SOL;
  const double* __restrict__ dd = B.dir + unit * L::SOL;
  const double q = s[L::S_Q + r], v = s[L::S_V + r], u = s[L::S_U + r];
  const double dq = dd[L::S_Q + r], dv = dd[L::S_V + r], du = dd[L::S_U + r];
  if (active) {
    s[L::S_LMD + r] += ap * dd[L::S_LMD + r];
    s[L::S_GMM + r] += ap * dd[L::S_GMM + r];
    s[L::S_Q + r] = q + ap * dq;
    s[L::S_V + r] = v + ap * dv;
  }
  if (i == N) return;
  const double a_old = s[L::S_A + r], da_dir = dd[L::S_A + r];
  if (active) {
    s[L::S_A + r] = a_old + ap * da_dir;
    s[L::S_U + r] = u + ap * du;
    s[L::S_BETA + r] += ap * dd[L::S_BETA + r];
  }
  const long su = b * N + i;
  double* __restrict__ slack = B.slack + su * L::CON;
  double* __restrict__ dual = B.dual + su * L::CON;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    if (!rowValid(P, c, i)) continue;
    const double sgn = (c & 1) ? 1.0 : -1.0;
    const double x = (c < 2) ? q : ((c < 4) ? v : (c < 6 ? u : a_old));
    const double dx = (c < 2) ? dq : ((c < 4) ? dv : (c < 6 ? du : da_dir));
    double* slp = c < 6 ? slack + c * NV + r : B.slack_a + su * 2 * NV + (c - 6) * NV + r;
    double* dlp = c < 6 ? dual + c * NV + r : B.dual_a + su * 2 * NV + (c - 6) * NV + r;
    const double sl = *slp, dl = *dlp;
    const IpmRow row = ipmResidual(sgn, x, limitOf(P, c, r), sl, dl, P->barrier);
    const double dslack = -sgn * dx - row.residual;
    const double ddual = -(dl * dslack + row.duality) * recipNewton(sl);
    if (active) { *slp = sl + ap * dslack; *dlp = dl + ad * ddual; }
  }
}

// SplitUnOCP::initConstraints -> pdipm::SetSlackAndDualPositive (split_unocp.hxx:61-66; pdipm.hxx:13-23)
template <int NV>
__global__ __launch_bounds__(64) void un_init_constraints_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x;
  const int g = lane >> 3;
  const int r = lane & 7;
  const long total = (long)P->batch * N;
  const long su = (long)blockIdx.x * 8 + g;
  if (su >= total || r >= NV) return;
  const long b = su / N;
  const int i = (int)(su - b * N);
  const double* __restrict__ s = B.sol + (b * (N + 1) + i) * L::SOL;
  for (int c = 0; c < 8; ++c) {
    if (c >= 6 && !B.slack_a) continue;
    double sl = 1.0, dl = 0.0;
    if (rowValid(P, c, i)) {
      const double sgn = (c & 1) ? 1.0 : -1.0;
      const double x = (c < 2) ? s[L::S_Q + r] : ((c < 4) ? s[L::S_V + r] : (c < 6 ? s[L::S_U + r] : s[L::S_A + r]));
      sl = -sgn * (x - limitOf(P, c, r));
      sl = slackPositive(sl, P->barrier);      // pdipm.hxx:17-20
      dl = P->barrier / sl;
    }
    if (c < 6) { B.slack[su * L::CON + c * NV + r] = sl; B.dual[su * L::CON + c * NV + r] = dl; }
    else { B.slack_a[su * 2 * NV + (c - 6) * NV + r] = sl; B.dual_a[su * 2 * NV + (c - 6) * NV + r] = dl; }
  }
}

// KKTError(): sum the stage errors + the terminal stage's lx (unocp_solver.cpp:190-202;
// terminal_ocp.hxx:118-144).  One wavefront per instance.
template <int NV>
__global__ __launch_bounds__(64) void un_kkt_error_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const long b = blockIdx.x;
  double e = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) e += B.err_stage[b * (N + 1) + i];
  if (threadIdx.x < NV && !P->backward_euler) {     // UnParNMPC: the last STAGE carries the terminal cost
    const int r = threadIdx.x;
    const double* __restrict__ sN = B.sol + (b * (N + 1) + N) * L::SOL;
    double lq = P->qf_weight[r] * (sN[L::S_Q + r] - P->q_ref[r]) - sN[L::S_LMD + r];
    const double lv = P->vf_weight[r] * (sN[L::S_V + r] - P->v_ref[r]) - sN[L::S_GMM + r];
    if (P->task.dim) lq += B.task_term[b * L::TASK + L::T_G + r];
    e += lq * lq + lv * lv;
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) e += __shfl_xor(e, off);
  if (threadIdx.x == 0) B.err[b] = sqrt(e);
}

// setSolution(name, value): write one field of every stage record (unocp_solver.cpp:157-181).
// per_instance = 0: value[dim] broadcast; 1: value[batch][dim].
__global__ void un_fill_field_kernel(double* __restrict__ sol, int stride, int offset, int dim, long nrec_per_inst,
                                     long batch, const double* __restrict__ value, int per_instance) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = batch * nrec_per_inst * dim;
  if (idx >= total) return;
  const int e = (int)(idx % dim);
  const long rec = idx / dim;
  const long b = rec / nrec_per_inst;
  sol[rec * stride + offset + e] = value[(per_instance ? b * dim : 0) + e];
}

// Stand-alone inverse dynamics + derivatives for n samples (parity tests of the rigid-body layer against the reference's golden
// vectors): the two-phase analytic recursion exactly as K1 runs it -- phase A with a lane per (sample, joint), nine samples per
// wavefront, then the rows of three samples at a time (dev_rnea_analytic.hpp).
template <int NV, bool ZAX>
__global__ __launch_bounds__(64, 2) void rnea_derivatives_kernel(const DevModel* __restrict__ model, int n,
                                                                const double* __restrict__ q, const double* __restrict__ v,
                                                                const double* __restrict__ a, double* __restrict__ tau,
                                                                double* __restrict__ dq, double* __restrict__ dv,
                                                                double* __restrict__ da) {
  constexpr int LPS = 3 * NV, SPW = 64 / LPS, SPA = 64 / NV, ROUNDS = SPA / SPW, BLK = RneaBlock::LEN;
  static_assert(SPA == SPW * ROUNDS, "whole rounds");
  __shared__ ChainConsts<NV> s_model;
  __shared__ double s_cs[SPA][NV][2], s_v[SPA][NV], s_a[SPA][NV];
  __shared__ double s_pub[SPW][NV][BLK];
  const int lane = threadIdx.x;
  const long smp0 = (long)blockIdx.x * SPA;
  const int sA0 = lane / NV;
  const int sA = sA0 < SPA ? sA0 : SPA - 1;
  const int jA = sA0 < SPA ? lane - sA0 * NV : 0;
  const long smpA = smp0 + sA < n ? smp0 + sA : n - 1;
  s_model.load(model, lane, 64);
  if (sA0 < SPA) {
    double sj, cj;
    sincos(q[smpA * NV + jA], &sj, &cj);
    s_cs[sA][jA][0] = cj; s_cs[sA][jA][1] = sj;
    s_v[sA][jA] = v[smpA * NV + jA]; s_a[sA][jA] = a[smpA * NV + jA];
  }
  WAVE_SYNC();
  double blk[BLK];
  rneaDerivPhaseA<NV, ZAX>(&s_model, &s_cs[sA][0][0], &s_v[sA][0], &s_a[sA][0], jA, blk, nullptr);
  if (sA0 < SPA && smp0 + sA0 < n) tau[(smp0 + sA0) * NV + jA] = blk[RneaBlock::TAU];
  const int g0 = lane / LPS;
  const int g = g0 < SPW ? g0 : SPW - 1;
  const int seed = lane - g0 * LPS;
  const int kind = (g0 < SPW) ? seed / NV : 0;
  const int k = (g0 < SPW) ? seed - kind * NV : 0;
  double* out = (kind == 0) ? dq : ((kind == 1) ? dv : da);
#pragma unroll
  for (int rho = 0; rho < ROUNDS; ++rho) {
    WAVE_SYNC();
    if (sA0 < SPA && sA0 / SPW == rho) {
      double* o = &s_pub[sA0 - rho * SPW][jA][0];
#pragma unroll
      for (int e = 0; e < BLK; ++e) o[e] = blk[e];
    }
    WAVE_SYNC();
    double row[NV];
    rneaDerivPhaseB<NV>(&s_pub[g][0][0], BLK, kind, k, row);
    const long smp = smp0 + rho * SPW + g;
    if (g0 < SPW && smp < n) {
#pragma unroll
      for (int c = 0; c < NV; ++c) out[smp * NV * NV + c * NV + k] = row[c];      // element (k, c) of the column-major matrix
    }
  }
}


// ================================================================ UnParNMPC ====
// UnParNMPCSolver::updateSolution (src/unocp/unparnmpc_solver.cpp:74-103) = UnBackwardCorrection::coarseUpdate +
// backwardCorrection (src/unocp/unbackward_correction.cpp:67-132):
//   K1b  un_linearize_kernel<NV, 0, true>       SplitUnParNMPC / TerminalUnParNMPC::linearizeOCP          (per stage)
//   K9u  unparnmpc_coarse_update_kernel         SplitUnKKTMatrixInverter::invert + coarseUpdate            (per stage)
//   S5u  unparnmpc_backward_serial_kernel       backwardCorrectionSerial, stages N-2 .. 0                  (per instance)
//   K10u unparnmpc_backward_parallel_kernel     backwardCorrectionParallel                                 (per stage)
//   S6u  unparnmpc_forward_serial_kernel        forwardCorrectionSerial, stages 1 .. N-1                   (per instance)
//   K11u unparnmpc_expand_kernel                forwardCorrectionParallel + aux_mat + computeDirection +
//                                               computeCondensedDirection + slack / dual directions + steps (per stage)
//   K3   un_integrate_kernel                    updatePrimal / updateDual
// Stage i of instance b uses record b * (N + 1) + i of the (N + 1)-record arrays (record N stays unused and zero).

// K9u: one wavefront per stage, three stages per workgroup.  KKT matrix [[0 F]; [F^T Q]] in the ordering
// (lmd, gmm | a, q, v) with F = [0 -I dt I; dt I 0 -I] (split_unkkt_matrix_inverter.hxx:37-80):
//   Q^-1, FQ = F Q^-1, S = FQ F^T, TL = -S^-1, TR = S^-1 FQ, BR = Q^-1 - FQ^T TR;
// d = K^-1 [Fq Fv la lq lv], s_new = s - d (split_unbackward_correction.hxx:50-64).  The two inverses are chains of
// dependent pivot steps that use 21 (14) lanes each: wavefront 0 runs them for the three stages of the workgroup at once in
// its registers (spdInverseRowsGrouped), everything else is done by each wavefront for its own stage through LDS.
#ifdef K1_PROF
#define K9_T0() unsigned long long t_prev = __builtin_readcyclecounter(); unsigned long long t_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define K9_T(i) do { const unsigned long long t_now = __builtin_readcyclecounter(); t_acc[i] += t_now - t_prev; t_prev = t_now; if (i == 7 && threadIdx.x == 0 && blockIdx.x % 61 == 0) for (int e = 0; e < 8; ++e) atomicAdd(&g_k1_prof[8 + e], t_acc[e]); } while (0)
#else
#define K9_T0() do { } while (0)
#define K9_T(i) do { } while (0)
#endif
#ifndef K9U_SPB
#define K9U_SPB 3          // stages (= wavefronts) per workgroup of the coarse update
#endif
template <int NV>
__global__ __launch_bounds__(64 * K9U_SPB) void unparnmpc_coarse_update_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX, NQ = L::NQ3, NK = 5 * NV, SPB = K9U_SPB;
  __shared__ __attribute__((aligned(16))) double sQa[SPB][NQ * NQ], sFQa[SPB][NX * NQ], sSa[SPB][NX * NX], sTRa[SPB][NX * NQ], sresa[SPB][NK], st1a[SPB][NX], sda[SPB][NK];
  __shared__ int s_ok[SPB];
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long total = (long)P->batch * N;
  long unit = (long)blockIdx.x * SPB + w;
  const bool valid = unit < total;
  if (!valid) unit = total - 1;
  double* sQ = sQa[w]; double* sFQ = sFQa[w]; double* sS = sSa[w]; double* sTR = sTRa[w];
  double* sres = sresa[w]; double* st1 = st1a[w]; double* sd = sda[w];
  const long b = unit / N;
  const int i = (int)(unit - b * N);
  const long rec = b * (N + 1) + i;
  const double* __restrict__ kk = B.kkt + unit * L::KKT;
  const double* __restrict__ aux = B.aux + (rec + 1) * L::AUX;       // aux_mat of the NEXT stage (none behind the last one)
  if (lane == 0) s_ok[w] = 1;
  K9_T0();
  // Q in the ordering (a, q, v) from the six upper blocks K1b stores, + aux_mat_next on the (q, v) block
  // (unrolled: the loads of all passes are in flight together)
#pragma unroll
  for (int e0 = 0; e0 < NQ * NQ; e0 += 64) {
    const int e1 = e0 + lane;
    const int e = e1 < NQ * NQ ? e1 : NQ * NQ - 1;      // (the lanes past the end redo the last element)
    const int c = e / NQ, r = e - c * NQ;
    const int bi = r / NV, ri = r - bi * NV, bj = c / NV, cj = c - bj * NV;
    const int lo = bi < bj ? bi : bj, hi = bi < bj ? bj : bi;
    const int rr = bi <= bj ? ri : cj, cc = bi <= bj ? cj : ri;
    const int off = lo == 0 ? (hi == 0 ? L::K_QAA : (hi == 1 ? L::K_QAQ : L::K_QAV)) : (lo == 1 ? (hi == 1 ? L::K_QQQ : L::K_QQV) : L::K_QVV);
    const double val0 = lo == hi ? kk[off + L::sym(ri, cj)] : kk[off + cc * NV + rr];      // diagonal blocks: packed upper triangle
    double val = val0;
    if ((i < N - 1 || !P->has_terminal) && bi >= 1 && bj >= 1) val += aux[(c - NV) * NX + (r - NV)];
    sQ[e] = val;
  }
  if (lane < NK) sres[lane] = kk[L::K_FQ + lane];                     // [Fq Fv la lq lv] are contiguous in the kkt record
  __syncthreads();
  K9_T(0);
  if (w == 0) spdInverseRowsGrouped<NQ, SPB>(&sQa[0][0], NQ * NQ, lane, s_ok);
  __syncthreads();
  K9_T(1);
  for (int e = lane; e < NX * NQ; e += 64) {
    const int c = e / NX, r = e - c * NX;
    sFQ[e] = r < NV ? -sQ[(NV + r) + NQ * c] + dt * sQ[(2 * NV + r) + NQ * c] : dt * sQ[(r - NV) + NQ * c] - sQ[(NV + r) + NQ * c];
  }
  __syncthreads();
  for (int e = lane; e < NX * NX; e += 64) {
    const int c = e / NX, r = e - c * NX;
    sS[e] = c < NV ? -sFQ[r + NX * (NV + c)] + dt * sFQ[r + NX * (2 * NV + c)] : dt * sFQ[r + NX * (c - NV)] - sFQ[r + NX * (NV + c)];
  }
  __syncthreads();
  K9_T(2);
  if (w == 0) spdInverseRowsGrouped<NX, SPB>(&sSa[0][0], NX * NX, lane, s_ok);
  __syncthreads();
  K9_T(3);
  // TR = S^-1 FQ (NX x NQ).  A lane forms a 2 x 3 tile of it: per step of the contraction one 16-byte read of S^-1 (two rows of a column) and three
  // reads of FQ feed six multiply-adds -- 4 LDS instructions where one output per lane takes 12; every entry still sums k = 0 .. NX - 1 in that order.
  // (The LDS pipe of a CU is active 68 % of this kernel's time: profiles/experiments/r06_k1_occupancy.md.)
  static_assert(NX % 2 == 0 && NQ % 3 == 0 && (NX / 2) * (NQ / 3) <= 64 && (NX * NQ) % 2 == 0 && (NX * NX) % 2 == 0, "2 x 3 tiles of TR, 3 x 2 tiles of BR on one wavefront");
  typedef double kd2 __attribute__((ext_vector_type(2)));
  if (lane < (NX / 2) * (NQ / 3)) {
    const int tc = lane / (NX / 2), tr = lane - tc * (NX / 2), r0 = 2 * tr, c0 = 3 * tc;
    double a00 = 0.0, a01 = 0.0, a02 = 0.0, a10 = 0.0, a11 = 0.0, a12 = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) {
      const kd2 sv = *reinterpret_cast<const kd2*>(&sS[r0 + NX * k]);
      const double b0 = sFQ[k + NX * c0], b1 = sFQ[k + NX * (c0 + 1)], b2 = sFQ[k + NX * (c0 + 2)];
      a00 += sv.x * b0; a01 += sv.x * b1; a02 += sv.x * b2;
      a10 += sv.y * b0; a11 += sv.y * b1; a12 += sv.y * b2;
    }
    *reinterpret_cast<kd2*>(&sTR[r0 + NX * c0]) = kd2{a00, a10};
    *reinterpret_cast<kd2*>(&sTR[r0 + NX * (c0 + 1)]) = kd2{a01, a11};
    *reinterpret_cast<kd2*>(&sTR[r0 + NX * (c0 + 2)]) = kd2{a02, a12};
  }
  __syncthreads();
  K9_T(4);
  double* __restrict__ ki = B.kinv + unit * L::KINV;
  if (valid) {
    for (int e = lane; e < NX * NX; e += 64) ki[L::I_TL + e] = -sS[e];
    for (int e = lane; e < NX * NQ; e += 64) ki[L::I_TR + e] = sTR[e];
  }
  // the (q, v) columns of BR = Q^-1 - FQ^T TR (NQ x NX): 3 x 2 tiles, both operands contiguous along the contraction -- two steps per 16-byte read, five
  // reads for twelve multiply-adds
  if (lane < (NQ / 3) * (NX / 2)) {
    const int tc = lane / (NQ / 3), tr = lane - tc * (NQ / 3), r0 = 3 * tr, c0 = 2 * tc;
    double acc[3][2];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[i][j] = sQ[(r0 + i) + NQ * (NV + c0 + j)];
#pragma unroll
    for (int k = 0; k < NX; k += 2) {
      kd2 fa[3], tb[2];
#pragma unroll
      for (int i = 0; i < 3; ++i) fa[i] = *reinterpret_cast<const kd2*>(&sFQ[k + NX * (r0 + i)]);
#pragma unroll
      for (int j = 0; j < 2; ++j) tb[j] = *reinterpret_cast<const kd2*>(&sTR[k + NX * (NV + c0 + j)]);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) { acc[i][j] -= fa[i].x * tb[j].x; acc[i][j] -= fa[i].y * tb[j].y; }
    }
    if (valid)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) ki[L::I_BRC + (r0 + i) + NQ * (c0 + j)] = acc[i][j];
  }
  K9_T(5);
  // d = K^-1 res:  t1 = TR l,  d_top = -S^-1 Fx + t1,  d_bot = TR^T Fx + Q^-1 l - FQ^T t1
  if (lane < NX) {
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc += sTR[lane + NX * c] * sres[NX + c];
    st1[lane] = acc;
  }
  __syncthreads();
  if (lane < NX) {
    double acc = st1[lane];
#pragma unroll
    for (int k = 0; k < NX; ++k) acc -= sS[lane + NX * k] * sres[k];
    sd[lane] = acc;
  } else if (lane < NX + NQ) {
    const int r = lane - NX;
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < NX; ++k) acc += sTR[k + NX * r] * sres[k] - sFQ[k + NX * r] * st1[k];
#pragma unroll
    for (int c = 0; c < NQ; ++c) acc += sQ[r + NQ * c] * sres[NX + c];
    sd[lane] = acc;
  }
  __syncthreads();
  K9_T(6);
  // s_new = s - d in the fields (lmd, gmm | a, q, v)
  if (valid && lane < NK) {
    const int f = lane < NX ? L::S_LMD + lane : (lane < NX + NV ? L::S_A + (lane - NX) : L::S_Q + (lane - NX - NV));
    B.snew[rec * L::SOL + f] = B.sol[rec * L::SOL + f] - sd[lane];
  }
  if (valid && lane == 0 && !s_ok[w]) atomicMax(&B.status[b], 1 + i);
  K9_T(7);
}

// S5u: 16 lanes per instance, lane r < NX owns row r of the costate pair (lmd, gmm).  Stage i:
//   x_res = s_new[i+1].(lmd, gmm) - s[i+1].(lmd, gmm);  s_new[i].(lmd, gmm) -= K^-1(0.., 3nv..) x_res   (TR's (q, v) columns)
// (split_unbackward_correction.hxx:72-81).  The loads of stage i - 1 are issued before the dependent chain of stage i.
template <int NV>
__global__ __launch_bounds__(64) void unparnmpc_backward_serial_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x, g = lane >> 4, r0 = lane & 15, r = r0 < NX ? r0 : NX - 1;
  long b = (long)blockIdx.x * 4 + g;
  const bool active = b < P->batch && r0 < NX;
  if (b >= P->batch) b = P->batch - 1;
  const int base = lane & ~15;
  const double* __restrict__ sol = B.sol + b * (N + 1) * L::SOL;
  double* __restrict__ snew = B.snew + b * (N + 1) * L::SOL;
  double* __restrict__ xres = B.xres + b * (N + 1) * L::XRES;
  const double* __restrict__ kinv = B.kinv + b * N * L::KINV;
  // the sweep starts behind the terminal stage; on a shard that ends earlier it starts from the right neighbour's first
  // stage, whose (lmd, gmm) and corrected (lmd, gmm) sit in record N
  const int i_first = P->has_terminal ? N - 2 : N - 1;
  if (i_first < 0) return;
  double cur = snew[(i_first + 1) * L::SOL + r];           // (lmd, gmm) are the first NX entries of a record
  double m[NX], mn[NX];
  double s_next = sol[(i_first + 1) * L::SOL + r], own = snew[i_first * L::SOL + r];
#pragma unroll
  for (int c = 0; c < NX; ++c) m[c] = kinv[(long)i_first * L::KINV + L::I_TR + (NV + c) * NX + r];
  for (int i = i_first; i >= 0; --i) {
    double s_next_n = 0.0, own_n = 0.0;
    if (i > 0) {
      s_next_n = sol[i * L::SOL + r]; own_n = snew[(i - 1) * L::SOL + r];
#pragma unroll
      for (int c = 0; c < NX; ++c) mn[c] = kinv[(long)(i - 1) * L::KINV + L::I_TR + (NV + c) * NX + r];
    }
    const double x = cur - s_next;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) acc += m[c] * __shfl(x, base + c);
    cur = own - acc;
    if (active) { xres[i * L::XRES + r] = x; snew[i * L::SOL + r] = cur; }
    s_next = s_next_n; own = own_n;
#pragma unroll
    for (int c = 0; c < NX; ++c) m[c] = mn[c];
  }
}

// K10u: s_new[i].(a, q, v) -= K^-1(2nv.., 3nv..) x_res for i <= N - 2 (split_unbackward_correction.hxx:84-92): 32 lanes per
// stage, lane r < 3 NV owns one row.
template <int NV>
__global__ __launch_bounds__(64) void unparnmpc_backward_parallel_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX, NQ = L::NQ3;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x, g = lane >> 5, r = lane & 31;
  const int nc = P->has_terminal ? N - 1 : N;             // stages that have a successor
  const long total = (long)P->batch * nc;
  const long unit = (long)blockIdx.x * 2 + g;
  if (unit >= total || r >= NQ) return;
  const long b = unit / nc;
  const int i = (int)(unit - b * nc);
  const long rec = b * (N + 1) + i;
  const double* __restrict__ x = B.xres + rec * L::XRES;
  const double* __restrict__ m = B.kinv + (b * N + i) * L::KINV + L::I_BRC;
  double acc = 0.0;
#pragma unroll
  for (int c = 0; c < NX; ++c) acc += m[r + NQ * c] * x[c];
  const int f = r < NV ? L::S_A + r : L::S_Q + (r - NV);
  B.snew[rec * L::SOL + f] -= acc;
}

// S6u: 16 lanes per instance, lane r < NX owns row r of the state (q, v).  Stage i >= 1:
//   x_res = s_new[i-1].(q, v) - s[i-1].(q, v);  s_new[i].(q, v) -= K^-1(3nv.., 0..) x_res   (= TR(:, q v)^T)
// (split_unbackward_correction.hxx:95-104)
template <int NV>
__global__ __launch_bounds__(64) void unparnmpc_forward_serial_kernel(UnBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const int lane = threadIdx.x, g = lane >> 4, r0 = lane & 15, r = r0 < NX ? r0 : NX - 1;
  long b = (long)blockIdx.x * 4 + g;
  const bool active = b < P->batch && r0 < NX;
  if (b >= P->batch) b = P->batch - 1;
  const int base = lane & ~15;
  const double* __restrict__ sol = B.sol + b * (N + 1) * L::SOL;
  double* __restrict__ snew = B.snew + b * (N + 1) * L::SOL;
  double* __restrict__ xres = B.xres + b * (N + 1) * L::XRES;
  const double* __restrict__ kinv = B.kinv + b * N * L::KINV;
  // the sweep starts at stage 1; on a shard with a left neighbour it starts at stage 0 from that neighbour's last stage
  // (its state is the shard's "measured" state q0 / v0, its corrected state B.xprev)
  const int i_first = P->has_prev ? 0 : 1;
  if (i_first >= N) return;
  double cur, s_prev;
  if (P->has_prev) { cur = B.xprev[b * NX + r]; s_prev = r < NV ? q0[b * NV + r] : v0[b * NV + r - NV]; }
  else { cur = snew[L::S_Q + r]; s_prev = sol[L::S_Q + r]; }           // (q, v) are contiguous in a record
  double m[NX], mn[NX];
  double own = snew[i_first * L::SOL + L::S_Q + r];
#pragma unroll
  for (int c = 0; c < NX; ++c) m[c] = kinv[(long)i_first * L::KINV + L::I_TR + (NV + r) * NX + c];
  for (int i = i_first; i < N; ++i) {
    double s_prev_n = 0.0, own_n = 0.0;
    if (i + 1 < N) {
      s_prev_n = sol[i * L::SOL + L::S_Q + r]; own_n = snew[(i + 1) * L::SOL + L::S_Q + r];
#pragma unroll
      for (int c = 0; c < NX; ++c) mn[c] = kinv[(long)(i + 1) * L::KINV + L::I_TR + (NV + r) * NX + c];
    }
    const double x = cur - s_prev;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) acc += m[c] * __shfl(x, base + c);
    cur = own - acc;
    if (active) { xres[i * L::XRES + NX + r] = x; snew[i * L::SOL + L::S_Q + r] = cur; }
    s_prev = s_prev_n; own = own_n;
#pragma unroll
    for (int c = 0; c < NX; ++c) m[c] = mn[c];
  }
}

// K11u: the last parallel loop of UnBackwardCorrection::backwardCorrection (unbackward_correction.cpp:114-131), 8 lanes per
// stage, lane r owns row r: forwardCorrectionParallel (i > 0: (lmd, gmm, a) -= K^-1(0.., 0..) x_res; aux_mat = S^-1),
// computeDirection d = s_new - s, the condensed direction (unconstrained_dynamics.hxx:97-106), the slack / dual directions
// and the fraction-to-boundary step sizes of the stage.
template <int NV>
__global__ __launch_bounds__(64) void unparnmpc_expand_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX;
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x, g = lane >> 3, r0 = lane & 7, r = r0 < NV ? r0 : NV - 1;
  const long total = (long)P->batch * N;
  long su = (long)blockIdx.x * 8 + g;
  const bool active = (su < total) && (r0 < NV);
  if (su >= total) su = total - 1;
  const long b = su / N;
  const int i = (int)(su - b * N);
  const long rec = b * (N + 1) + i;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  double* __restrict__ sn = B.snew + rec * L::SOL;
  double* __restrict__ dd = B.dir + rec * L::SOL;
  const double* __restrict__ ki = B.kinv + su * L::KINV;
  double n_lmd = sn[L::S_LMD + r], n_gmm = sn[L::S_GMM + r], n_a = sn[L::S_A + r];
  if (i > 0 || P->has_prev) {
    const double* __restrict__ x = B.xres + rec * L::XRES + NX;
    double c_lmd = 0.0, c_gmm = 0.0, c_a = 0.0;
#pragma unroll
    for (int c = 0; c < NX; ++c) {
      const double xc = x[c];
      c_lmd += ki[L::I_TL + r + NX * c] * xc;
      c_gmm += ki[L::I_TL + NV + r + NX * c] * xc;
      c_a += ki[L::I_TR + c + NX * r] * xc;                 // K^-1(2nv + r, c) = TR(c, r)
    }
    n_lmd -= c_lmd; n_gmm -= c_gmm; n_a -= c_a;
    if (active) { sn[L::S_LMD + r] = n_lmd; sn[L::S_GMM + r] = n_gmm; sn[L::S_A + r] = n_a; }
    double* __restrict__ aux = B.aux + rec * L::AUX;        // aux_mat[i] = -K^-1.topLeft = S^-1
    if (su < total) for (int e = r0; e < NX * NX; e += 8) aux[e] = -ki[L::I_TL + e];
  }
  double dq[NV], dv[NV], da[NV];
#pragma unroll
  for (int c = 0; c < NV; ++c) { dq[c] = sn[L::S_Q + c] - s[L::S_Q + c]; dv[c] = sn[L::S_V + c] - s[L::S_V + c]; }
  const double da_r = n_a - s[L::S_A + r];
#pragma unroll
  for (int c = 0; c < NV; ++c) da[c] = __shfl(da_r, (lane & ~7) + c);
  const double* __restrict__ dy = B.dyn + su * L::DYN;
  double du = dy[L::D_ID + r];
#pragma unroll
  for (int c = 0; c < NV; ++c) du += dy[L::D_DQ + c * NV + r] * dq[c] + dy[L::D_DV + c * NV + r] * dv[c] + dy[L::D_DA + c * NV + r] * da[c];
  const double dbeta = (dy[L::D_LU + r] + dy[L::D_QUU + r] * du) / dt;
  if (active) {
    dd[L::S_LMD + r] = n_lmd - s[L::S_LMD + r]; dd[L::S_GMM + r] = n_gmm - s[L::S_GMM + r];
    dd[L::S_Q + r] = dq[r]; dd[L::S_V + r] = dv[r]; dd[L::S_A + r] = da_r; dd[L::S_U + r] = du; dd[L::S_BETA + r] = dbeta;
  }
  const double* __restrict__ slack = B.slack + su * L::CON;
  const double* __restrict__ dual = B.dual + su * L::CON;
  double ps = 1.0, ds = 1.0;
#pragma unroll
  for (int c = 0; c < 8; ++c) {      // 6, 7: joint acceleration limits (rows in slack_a / dual_a, joint_acceleration_{lower,upper}_limit.cpp:78-93)
    if (!rowValid(P, c, i)) continue;
    const double sgn = (c & 1) ? 1.0 : -1.0;
    const double x = (c < 2) ? s[L::S_Q + r] : ((c < 4) ? s[L::S_V + r] : (c < 6 ? s[L::S_U + r] : s[L::S_A + r]));
    const double dx = (c < 2) ? dq[r] : ((c < 4) ? dv[r] : (c < 6 ? du : da_r));
    const double sl = c < 6 ? slack[c * NV + r] : B.slack_a[su * 2 * NV + (c - 6) * NV + r], dl = c < 6 ? dual[c * NV + r] : B.dual_a[su * 2 * NV + (c - 6) * NV + r];
    const IpmRow row = ipmResidual(sgn, x, limitOf(P, c, r), sl, dl, P->barrier);
    const double dslack = -sgn * dx - row.residual;
    const double ddual = -(dl * dslack + row.duality) * recipNewton(sl);
    ps = fractionToBoundary(P->fraction_rate, sl, dslack, ps);
    ds = fractionToBoundary(P->fraction_rate, dl, ddual, ds);
  }
  if (!active) { ps = 1.0; ds = 1.0; }
#pragma unroll
  for (int off = 4; off >= 1; off >>= 1) {
    ps = fmin(ps, __shfl_xor(ps, off));
    ds = fmin(ds, __shfl_xor(ds, off));
  }
  if (active && r0 == 0) { B.step_stage[su * 2] = ps; B.step_stage[su * 2 + 1] = ds; }
}

// UnBackwardCorrection::initAuxMat (unbackward_correction.cpp:55-64): the terminal cost Hessian on every stage
template <int NV>
__global__ void unparnmpc_init_aux_kernel(UnBuffers B) {
  using L = UnLayout<NV>;
  constexpr int NX = L::NX;
  const UnProblem* __restrict__ P = B.prob;
  const long total = (long)P->batch * (P->N + 1) * NX * NX;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long rec = idx / (NX * NX);
  const int e = (int)(idx - rec * NX * NX), c = e / NX, r = e - c * NX;
  B.aux[rec * L::AUX + e] = (r == c) ? (r < NV ? P->qf_weight[r] : P->vf_weight[r - NV]) : 0.0;
}


// ============================================================= line search ====
// UnLineSearch::computeCostAndViolation (src/line_search/unline_search.cpp:55-121) at the trial iterate s + alpha d
// (computeTrySolution, unline_search.hpp:125-133; alpha = ls_alpha[instance], 0 = the iterate itself): per stage
//   cost      = stage cost (configuration_space_cost.cpp:241-256) + dt * barrier(slack + alpha dslack) (pdipm.hxx:84-87)
//               [+ terminal cost: of stage N for UnOCP, of the last stage for UnParNMPC]
//   violation = |state-equation residual|_1 + dt |ID - u|_1 + dt |g(x_try) + slack|_1
// (split_unocp.hxx:177-217, split_unparnmpc.hxx:179-227).  Same lane mapping as K1 (the inverse dynamics of the trial point
// needs the chain recursion; its tangent outputs are discarded); lane (kind, k) adds the terms of q_k / v_k / (a_k, u_k).
template <int NV, bool BWD>
__global__ __launch_bounds__(64) void un_line_search_kernel(UnBuffers B, const double* __restrict__ q0, const double* __restrict__ v0) {
  using L = UnLayout<NV>;
  constexpr int LPS = 3 * NV;
  constexpr int SPW = 64 / LPS;
  __shared__ double s_x[SPW][4][NV];          // trial q, v, a, u
  __shared__ double s_cs[SPW][NV][2];
  __shared__ double s_tau[SPW][NV];
  __shared__ double s_dcol[SPW][LPS][NV];     // discarded tangent columns
  __shared__ double s_sum[SPW][LPS][2];
  __shared__ double s_dummy[NV];
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const double dt = P->dt;
  const int lane = threadIdx.x;
  const int g0 = lane / LPS;
  const int g = g0 < SPW ? g0 : SPW - 1;
  const int seed = lane - g0 * LPS;
  const int kind = (g0 < SPW) ? seed / NV : 0;
  const int k = (g0 < SPW) ? seed - kind * NV : 0;
  const long total = (long)P->batch * N;
  long unit = (long)blockIdx.x * SPW + g;
  const bool active = (g0 < SPW) && (unit < total);
  if (unit >= total) unit = total - 1;
  const long b = unit / N;
  const int i = (int)(unit - b * N);
  const long rec = b * (N + 1) + i;
  const double al = B.ls_alpha[b];
  const double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ dd = B.dir + rec * L::SOL;
  const double* __restrict__ slack = B.slack + unit * L::CON;
  if (g0 < SPW && seed < NV) {
    const double qt = s[L::S_Q + seed] + al * dd[L::S_Q + seed];
    s_x[g][0][seed] = qt;
    s_x[g][1][seed] = s[L::S_V + seed] + al * dd[L::S_V + seed];
    s_x[g][2][seed] = s[L::S_A + seed] + al * dd[L::S_A + seed];
    s_x[g][3][seed] = s[L::S_U + seed] + al * dd[L::S_U + seed];
    double sj, cj;
    sincos(qt, &sj, &cj);
    s_cs[g][seed][0] = cj; s_cs[g][seed][1] = sj;
  }
  WAVE_SYNC();
  rneaChain<NV>(B.model, &s_cs[g][0][0], &s_x[g][1][0], &s_x[g][2][0], kind, k, (g0 < SPW) && seed == 0, &s_tau[g][0],
                (g0 < SPW) ? &s_dcol[g][seed][0] : &s_dummy[0]);
  WAVE_SYNC();
  double cost = 0.0, viol = 0.0;
  const double qt = s_x[g][0][k], vt = s_x[g][1][k], at = s_x[g][2][k], ut = s_x[g][3][k];
  const bool term = BWD ? (P->has_terminal && i == N - 1) : false;
  if (!BWD && P->task.dim) {       // TaskSpace*Cost::computeStageCost at the trial configuration (one lane of the group adds it)
    double tdiff[6], tcol[6];
    taskSpaceColumn<NV>(B.model, P->task, &s_cs[g][0][0], B.task_ref + 12 * i, k, tdiff, tcol);
    if (seed == 0) {
#pragma unroll
      for (int c = 0; c < 6; ++c) cost += 0.5 * dt * P->task.weight[c] * tdiff[c] * tdiff[c];
      if (B.task_xs && k == 0) cost += B.task_xs[((long)b * N + i) * L::TASK + L::T_COST];      // task_extra components at the trial point (un_task_terminal_kernel<.., true, true> ran first)
    }
  }
  // the neighbour of the state equation at ITS trial point (the measured state is fixed)
  double qo, vo;
  if (BWD) {
    if (i > 0) { qo = (s - L::SOL)[L::S_Q + k] + al * (dd - L::SOL)[L::S_Q + k]; vo = (s - L::SOL)[L::S_V + k] + al * (dd - L::SOL)[L::S_V + k]; }
    else { qo = q0[b * NV + k]; vo = v0[b * NV + k]; }
  } else {
    qo = (s + L::SOL)[L::S_Q + k] + al * (dd + L::SOL)[L::S_Q + k]; vo = (s + L::SOL)[L::S_V + k] + al * (dd + L::SOL)[L::S_V + k];
  }
  double x_cur, x_try, dx;
  if (kind == 0) {
    x_cur = s[L::S_Q + k]; x_try = qt; dx = dd[L::S_Q + k];
    cost += 0.5 * dt * P->q_weight[k] * (qt - P->q_ref[k]) * (qt - P->q_ref[k]);
    if (term) cost += 0.5 * P->qf_weight[k] * (qt - P->q_ref[k]) * (qt - P->q_ref[k]);
    if (!BWD && i == N - 1) cost += 0.5 * P->qf_weight[k] * (qo - P->q_ref[k]) * (qo - P->q_ref[k]);     // TerminalOCP::terminalCost of stage N
    viol += fabs(BWD ? qo - qt + dt * vt : qt - qo + dt * vt);
  } else if (kind == 1) {
    x_cur = s[L::S_V + k]; x_try = vt; dx = dd[L::S_V + k];
    cost += 0.5 * dt * P->v_weight[k] * (vt - P->v_ref[k]) * (vt - P->v_ref[k]);
    if (term) cost += 0.5 * P->vf_weight[k] * (vt - P->v_ref[k]) * (vt - P->v_ref[k]);
    if (!BWD && i == N - 1) cost += 0.5 * P->vf_weight[k] * (vo - P->v_ref[k]) * (vo - P->v_ref[k]);
    viol += fabs(BWD ? vo - vt + dt * at : vt + dt * at - vo);
    viol += dt * fabs(s_tau[g][k] - ut);
  } else {
    x_cur = s[L::S_U + k]; x_try = ut; dx = dd[L::S_U + k];
    cost += 0.5 * dt * (P->a_weight[k] * at * at + P->u_weight[k] * (ut - P->u_ref[k]) * (ut - P->u_ref[k]));
  }
#pragma unroll
  for (int cc = 0; cc < 2; ++cc) {
    const int c = 2 * kind + cc;
    if (!rowValid(P, c, i)) continue;
    const double sgn = (cc == 0) ? -1.0 : 1.0;
    const double lim = limitOf(P, c, k), sl = slack[c * NV + k];
    const double dslack = -sgn * dx - (sgn * (x_cur - lim) + sl);
    cost -= dt * P->barrier * log(sl + al * dslack);
    viol += dt * fabs(sgn * (x_try - lim) + sl);
  }
  if (kind == 2) {      // joint acceleration limits (components 6, 7) on the a seed lanes
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int c = 6 + cc;
      if (!rowValid(P, c, i)) continue;
      const double sgn = (cc == 0) ? -1.0 : 1.0;
      const double lim = limitOf(P, c, k), sl = B.slack_a[unit * 2 * NV + cc * NV + k];
      const double dslack = -sgn * dd[L::S_A + k] - (sgn * (s[L::S_A + k] - lim) + sl);
      cost -= dt * P->barrier * log(sl + al * dslack);
      viol += dt * fabs(sgn * (at - lim) + sl);
    }
  }
  if (g0 < SPW) { s_sum[g][seed][0] = active ? cost : 0.0; s_sum[g][seed][1] = active ? viol : 0.0; }
  WAVE_SYNC();
  if (active && seed == 0) {
    double c0 = 0.0, v1 = 0.0;
#pragma unroll
    for (int j = 0; j < LPS; ++j) { c0 += s_sum[g][j][0]; v1 += s_sum[g][j][1]; }
    B.ls_stage[rec * 2] = c0; B.ls_stage[rec * 2 + 1] = v1;
  }
}

// totalCosts / totalViolations (unline_search.hpp:143-149): one wavefront per instance
__global__ __launch_bounds__(64) void un_line_search_reduce_kernel(UnBuffers B) {
  const UnProblem* __restrict__ P = B.prob;
  const int N = P->N;
  const long b = blockIdx.x;
  double c = 0.0, v = 0.0;
  for (int i = threadIdx.x; i < N; i += 64) { c += B.ls_stage[(b * (N + 1) + i) * 2]; v += B.ls_stage[(b * (N + 1) + i) * 2 + 1]; }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) { c += __shfl_xor(c, off); v += __shfl_xor(v, off); }
  if (threadIdx.x == 0) {
    if (B.task) c += B.task_term[b * B.task_stride];      // terminal TaskSpace*Cost at the trial point (un_task_terminal_kernel<TRIAL>)
    B.ls_out[b * 2] = c; B.ls_out[b * 2 + 1] = v;
  }
}

// Halo exchange of a horizon shard: dst[b][dst_off + e] = src[b][src_off + e], e < n, with per-instance strides
__global__ void un_strided_copy_kernel(double* __restrict__ dst, long dst_stride, long dst_off, const double* __restrict__ src,
                                       long src_stride, long src_off, int n, long batch) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= batch * n) return;
  const long b = idx / n;
  const int e = (int)(idx - b * n);
  dst[b * dst_stride + dst_off + e] = src[b * src_stride + src_off + e];
}
// squared KKT error of the local stages (summed over the ranks by the caller)
__global__ void un_square_kernel(double* __restrict__ out, const double* __restrict__ err, long batch) {
  const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (b < batch) out[b] = err[b] * err[b];
}

// ------------------------------------------------------------ launchers ----
template <int NV>
void UnLaunch<NV>::linearize(const UnBuffers& B, long batch, int N, hipStream_t st) {
    constexpr int SPA = (64 / (3 * NV)) * K1_ROUNDS;       // stages per wavefront of un_linearize_kernel
    const long units = batch * N;
    const dim3 grid((unsigned)((units + SPA - 1) / SPA));
    const double* none = nullptr;
    if (B.task) {
      if (B.task_xs) hipLaunchKernelGGL((un_task_terminal_kernel<NV, false, true>), dim3((unsigned)((units + 7) / 8)), dim3(64), 0, st, B);      // stage terms of the task_extra components
      if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 0, false, true, true>), grid, dim3(64), 0, st, B, none, none);
      else hipLaunchKernelGGL((un_linearize_kernel<NV, 0, false, true>), grid, dim3(64), 0, st, B, none, none);
    } else {
      if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 0, false, false, true>), grid, dim3(64), 0, st, B, none, none);
      else hipLaunchKernelGGL((un_linearize_kernel<NV, 0>), grid, dim3(64), 0, st, B, none, none);
    }
  }
template <int NV>
void UnLaunch<NV>::residual(const UnBuffers& B, long batch, int N, hipStream_t st) {
    constexpr int SPA = (64 / (3 * NV)) * K1_ROUNDS;       // stages per wavefront of un_linearize_kernel
    const long units = batch * N;
    const dim3 grid((unsigned)((units + SPA - 1) / SPA));
    const double* none = nullptr;
    if (B.task) {
      if (B.task_xs) hipLaunchKernelGGL((un_task_terminal_kernel<NV, false, true>), dim3((unsigned)((units + 7) / 8)), dim3(64), 0, st, B);
      if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 1, false, true, true>), grid, dim3(64), 0, st, B, none, none);
      else hipLaunchKernelGGL((un_linearize_kernel<NV, 1, false, true>), grid, dim3(64), 0, st, B, none, none);
      hipLaunchKernelGGL((un_task_terminal_kernel<NV, false>), dim3((unsigned)((batch + 7) / 8)), dim3(64), 0, st, B);
    } else {
      if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 1, false, false, true>), grid, dim3(64), 0, st, B, none, none);
      else hipLaunchKernelGGL((un_linearize_kernel<NV, 1>), grid, dim3(64), 0, st, B, none, none);
    }
    hipLaunchKernelGGL((un_kkt_error_kernel<NV>), dim3((unsigned)batch), dim3(64), 0, st, B);
  }
template <int NV>
void UnLaunch<NV>::riccati(const UnBuffers& B, long batch, int /*N*/, const double* q0, const double* v0, hipStream_t st) {
    const unsigned blocks = (unsigned)((batch + 7) / 8);
    if (B.task) hipLaunchKernelGGL((un_task_terminal_kernel<NV, false>), dim3(blocks), dim3(64), 0, st, B);
    hipLaunchKernelGGL((un_riccati_backward_kernel<NV>), dim3(blocks), dim3(64), 0, st, B);
    hipLaunchKernelGGL((un_riccati_forward_kernel<NV>), dim3(blocks), dim3(64), 0, st, B, q0, v0);
  }
template <int NV>
void UnLaunch<NV>::expand(const UnBuffers& B, long batch, int N, hipStream_t st) {
    const long units = batch * (N + 1);
    hipLaunchKernelGGL((un_expand_kernel<NV>), dim3((unsigned)((units + 7) / 8)), dim3(64), 0, st, B);
    hipLaunchKernelGGL(un_reduce_steps_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
  }
template <int NV>
void UnLaunch<NV>::integrate(const UnBuffers& B, long batch, int N, hipStream_t st) {
    const long units = batch * (N + 1);
    hipLaunchKernelGGL((un_integrate_kernel<NV>), dim3((unsigned)((units + 7) / 8)), dim3(64), 0, st, B);
  }
template <int NV>
void UnLaunch<NV>::initConstraints(const UnBuffers& B, long batch, int N, hipStream_t st) {
    const long units = batch * N;
    hipLaunchKernelGGL((un_init_constraints_kernel<NV>), dim3((unsigned)((units + 7) / 8)), dim3(64), 0, st, B);
  }
template <int NV>
void UnLaunch<NV>::rneaDerivatives(const DevModel* m, int n, const double* q, const double* v, const double* a, double* tau,
                              double* dq, double* dv, double* da, bool zaxes, hipStream_t st) {
    constexpr int SPA = 64 / NV;
    const dim3 grid((unsigned)((n + SPA - 1) / SPA));
    if (zaxes) hipLaunchKernelGGL((rnea_derivatives_kernel<NV, true>), grid, dim3(64), 0, st, m, n, q, v, a, tau, dq, dv, da);
    else hipLaunchKernelGGL((rnea_derivatives_kernel<NV, false>), grid, dim3(64), 0, st, m, n, q, v, a, tau, dq, dv, da);
  }

template <int NV>
void UnLaunch<NV>::single(int kernel_id, const UnBuffers& B, long batch, int N, const double* q0, const double* v0,
                          hipStream_t st) {
  const unsigned inst_blocks = (unsigned)((batch + 7) / 8);
  const unsigned stage_blocks = (unsigned)((batch * (N + 1) + 7) / 8);
  switch (kernel_id) {
    case 0: linearize(B, batch, N, st); break;
    case 1:
      if (B.task) hipLaunchKernelGGL((un_task_terminal_kernel<NV, false>), dim3(inst_blocks), dim3(64), 0, st, B);
      hipLaunchKernelGGL((un_riccati_backward_kernel<NV>), dim3(inst_blocks), dim3(64), 0, st, B);
      break;
    case 2: hipLaunchKernelGGL((un_riccati_forward_kernel<NV>), dim3(inst_blocks), dim3(64), 0, st, B, q0, v0); break;
    case 3: hipLaunchKernelGGL((un_expand_kernel<NV>), dim3(stage_blocks), dim3(64), 0, st, B); break;
    case 4: hipLaunchKernelGGL(un_reduce_steps_kernel, dim3((unsigned)batch), dim3(64), 0, st, B); break;
    default: hipLaunchKernelGGL((un_integrate_kernel<NV>), dim3(stage_blocks), dim3(64), 0, st, B); break;
  }
}

// UnParNMPC: phase 0 linearize, 1 coarse update, 2 backward serial, 3 backward parallel, 4 forward serial,
// 5 forward parallel + direction + step sizes, 6 integrate
template <int NV>
void UnLaunch<NV>::parnmpcPhase(int phase, const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st) {
  const unsigned inst_blocks = (unsigned)((batch + 3) / 4);
  switch (phase) {
    case 0:
      if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 0, true, false, true>), dim3((unsigned)((batch * N + (64 / (3 * NV)) * K1_ROUNDS - 1) / ((64 / (3 * NV)) * K1_ROUNDS))), dim3(64), 0, st, B, q0, v0);
      else hipLaunchKernelGGL((un_linearize_kernel<NV, 0, true>), dim3((unsigned)((batch * N + (64 / (3 * NV)) * K1_ROUNDS - 1) / ((64 / (3 * NV)) * K1_ROUNDS))), dim3(64), 0, st, B, q0, v0);
      break;
    case 1: hipLaunchKernelGGL((unparnmpc_coarse_update_kernel<NV>), dim3((unsigned)((batch * N + K9U_SPB - 1) / K9U_SPB)), dim3(64 * K9U_SPB), 0, st, B); break;
    case 2: hipLaunchKernelGGL((unparnmpc_backward_serial_kernel<NV>), dim3(inst_blocks), dim3(64), 0, st, B); break;
    case 3: hipLaunchKernelGGL((unparnmpc_backward_parallel_kernel<NV>), dim3((unsigned)((batch * N + 1) / 2)), dim3(64), 0, st, B); break;
    case 4: hipLaunchKernelGGL((unparnmpc_forward_serial_kernel<NV>), dim3(inst_blocks), dim3(64), 0, st, B, q0, v0); break;
    case 5:
      hipLaunchKernelGGL((unparnmpc_expand_kernel<NV>), dim3((unsigned)((batch * N + 7) / 8)), dim3(64), 0, st, B);
      hipLaunchKernelGGL(un_reduce_steps_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
      break;
    default: hipLaunchKernelGGL((un_integrate_kernel<NV>), dim3((unsigned)((batch * (N + 1) + 7) / 8)), dim3(64), 0, st, B); break;
  }
}
template <int NV>
void UnLaunch<NV>::parnmpcResidual(const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st) {
  if (B.zaxes) hipLaunchKernelGGL((un_linearize_kernel<NV, 1, true, false, true>), dim3((unsigned)((batch * N + (64 / (3 * NV)) * K1_ROUNDS - 1) / ((64 / (3 * NV)) * K1_ROUNDS))), dim3(64), 0, st, B, q0, v0);
  else hipLaunchKernelGGL((un_linearize_kernel<NV, 1, true>), dim3((unsigned)((batch * N + (64 / (3 * NV)) * K1_ROUNDS - 1) / ((64 / (3 * NV)) * K1_ROUNDS))), dim3(64), 0, st, B, q0, v0);
  hipLaunchKernelGGL((un_kkt_error_kernel<NV>), dim3((unsigned)batch), dim3(64), 0, st, B);
}
template <int NV>
void UnLaunch<NV>::parnmpcInitAux(const UnBuffers& B, long batch, int N, hipStream_t st) {
  const long total = batch * (N + 1) * 4 * NV * NV;
  hipLaunchKernelGGL((unparnmpc_init_aux_kernel<NV>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, B);
}

template <int NV>
void UnLaunch<NV>::lineSearchEval(const UnBuffers& B, long batch, int N, bool bwd, const double* q0, const double* v0, hipStream_t st) {
  constexpr int SPW = 64 / (3 * NV);
  const unsigned blocks = (unsigned)((batch * N + SPW - 1) / SPW);
  if (B.task_xs && !bwd) hipLaunchKernelGGL((un_task_terminal_kernel<NV, true, true>), dim3((unsigned)((batch * N + 7) / 8)), dim3(64), 0, st, B);      // task_extra stage costs at the trial point
  if (bwd) hipLaunchKernelGGL((un_line_search_kernel<NV, true>), dim3(blocks), dim3(64), 0, st, B, q0, v0);
  else hipLaunchKernelGGL((un_line_search_kernel<NV, false>), dim3(blocks), dim3(64), 0, st, B, q0, v0);
  if (B.task && !bwd) hipLaunchKernelGGL((un_task_terminal_kernel<NV, true>), dim3((unsigned)((batch + 7) / 8)), dim3(64), 0, st, B);
  hipLaunchKernelGGL(un_line_search_reduce_kernel, dim3((unsigned)batch), dim3(64), 0, st, B);
}

void stridedCopy(double* dst, long dst_stride, long dst_off, const double* src, long src_stride, long src_off, int n, long batch, hipStream_t st) {
  hipLaunchKernelGGL(un_strided_copy_kernel, dim3((unsigned)((batch * n + 255) / 256)), dim3(256), 0, st, dst, dst_stride, dst_off, src, src_stride,
                     src_off, n, batch);
}
void squareInto(double* out, const double* err, long batch, hipStream_t st) {
  hipLaunchKernelGGL(un_square_kernel, dim3((unsigned)((batch + 255) / 256)), dim3(256), 0, st, out, err, batch);
}

template struct UnLaunch<7>;

void fillField(double* sol, int stride, int offset, int dim, long nrec_per_inst, long batch, const double* value,
               int per_instance, hipStream_t st) {
  const long total = batch * nrec_per_inst * dim;
  hipLaunchKernelGGL(un_fill_field_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, sol, stride, offset, dim,
                     nrec_per_inst, batch, value, per_instance);
}

}  // namespace idocp_dev

#ifdef K1_PROF
extern "C" int idocp_debug_k1_prof(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(idocp_dev::g_k1_prof), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(idocp_dev::g_k1_prof), z, sizeof(z)) != hipSuccess) return 2; }
  return 0;
}
#endif
