// K5b -- per-stage KKT assembly and contact-dynamics condensation (and, with
// RESIDUAL = true, K8: the stage's squared KKT residual).
//
// Replaces SplitOCP::linearizeOCP after the rigid-body part (include/idocp/ocp/split_ocp.hxx:58-91):
//   cost derivatives / Hessian      (src/cost/configuration_space_cost.cpp:292-365,
//                                    trotting_configuration_space_cost.cpp:269-343, contact_force_cost.cpp:153-194)
//   IPM terms                       (src/constraints/joint_*_limit.cpp, linearized_friction_cone.cpp:107-152)
//   state equation, floating base   (include/idocp/ocp/state_equation.hxx:12-63)
//   multipliers of ID and C         (include/idocp/ocp/contact_dynamics.hxx:48-81)
//   ContactDynamics::condenseContactDynamics incl. Robot::computeMJtJinv
//                                   (contact_dynamics.hxx:105-158; include/idocp/robot/robot.hxx:576-615)
// and TerminalOCP::linearizeOCP (include/idocp/ocp/terminal_ocp.hxx:50-66) for the last stage.
//
// One 256-thread workgroup per stage; every block of the stage lives in LDS
// (~35 - 39 kB after aliasing, four workgroups per CU).  Reads the stage's solution / slack / dual / lie records and the nominal
// rigid-body record of ocp_nominal_kernel (impulse stages: the lin record of ocp_rnea_kernel<D, true> instead), runs the tangent
// items of dev_rnea_tangent.hpp, writes the kkt record (LQR stage for the Riccati sweep) and the exp record (expansion cache).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "dev_dense.hpp"
#include "dev_lie.hpp"
#include "dev_rnea_tangent.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

// the tile products of K5 on the 4 x 4 x 4 form of the FP64 matrix instruction (dev_dense.hpp, mfmaTilePairTN4); -DIDOCP_K5_TILES16 brings
// the 16 x 16 x 4 form of rounds 2 - 4 back
#ifdef IDOCP_K5_TILES16
#define K5_TILE_PAIR mfmaTilePairTN
#else
#define K5_TILE_PAIR mfmaTilePairTN4
#endif

__device__ __forceinline__ double ocpLimit(const OcpProblem* __restrict__ P, int comp, int r) {
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
// which IPM rows exist on a stage (constraints_data.hpp:18-42): `level` is the time step the constraint data was created
// with (grid stage; 0 on aux / lift stages); impulse stages only carry the impulse friction cone
__device__ __forceinline__ bool ocpRowValid(const OcpProblem* __restrict__ P, int comp, int level, bool impulse) {
  if (impulse) return comp == 6 && P->use_impulse_friction_cone != 0;
  if (comp < 2) return P->use_q_limits && level >= 2;
  if (comp < 4) return P->use_v_limits && level >= 1;
  if (comp < 6) return P->use_u_limits != 0;
  if (comp == 8) return P->use_a_lower != 0;
  if (comp == 9) return P->use_a_upper != 0;
  return comp == 6 && P->use_friction_cone != 0;
}

// SFP: number of contact rows the LDS blocks are laid out for (= leading dimension of J, Qff, BL, SM; NV + SFP for the
// (a, f)-sized blocks).  D::NF in general; the instantiations with a compile-time contact count use that count, which
// takes the footprint from 38.7 kB (12 rows) to 35.0 kB (6 rows; the RNEA scratch that lives in MJ .. MJD pads the narrow layout).
template <typename D, int SFP = D::NF>
struct CondenseSmem {
  using L = OcpLayout<D>;
  // NVF: leading dimension of the (a, f)-sized blocks.  A multiple of 8 doubles (24, the half-contact layout) would put the columns
  // that the 2 x 2 products read side by side (stride 2 NVF) on two bank groups only -- SQ_LDS_BANK_CONFLICT was 46 % of the LDS
  // cycles of that class against 25 % with NVF = 30 -- so it is padded by two rows.
  static constexpr int NV = D::NV, NX = D::NX, NF = SFP, NVF = D::NV + SFP + (((D::NV + SFP) % 8 == 0) ? 2 : 0), NU = D::NU;
  static constexpr bool SEP_MINV = SFP <= 6;
  // matrices.  Aliases (lifetimes in the kernel body):
  //   MINV  = MM            the mass matrix is inverted in place (the wide layouts; the narrow ones: a block of its own, see below)
  //   QAFQV = DIDC          dIDCdqv is dead once MJD = MJtJinv * dIDCdqv is formed
  //   QAFU  = MM .. JM      M^-1 and J are dead once MJtJinv is assembled
  //   BL, SM  share the block that holds the solution / slack / dual copies during phase C
  //   ERR   = MJ            (RESIDUAL variant only, which never forms MJtJinv)
  // The condensed Hessian blocks are never staged in LDS: phase H writes them to the kkt record.
  // (dIDCdqv and MJD carry one more column: [ID; C] itself, so that MJtJinv [ID; C] falls out of the tile product MJtJinv dIDCdqv)
  static constexpr int DIDC = 0, MM = DIDC + NVF * (NX + 1), JM = MM + NV * NV,
                       IDC = JM + ((NF * NV > NVF * NV - NV * NV) ? NF * NV : NVF * NV - NV * NV),      // (Qafu_full, NVF x NV, later takes the M .. J blocks)
                       MJ = IDC + 32,
                       MJD = MJ + NVF * NVF,
                       // MJ .. MJD also hold the scratch of the RNEA sweeps (dead before MJtJinv is assembled); the narrow layouts are padded for it
                       QFF = (MJD + NVF * (NX + 1) > MJ + RneaScratch<D>::TOTAL) ? MJD + NVF * (NX + 1) : MJ + RneaScratch<D>::TOTAL, TMP = QFF + NF * NF,
                       QAFQV = DIDC, QAFU = MM, ERR = MJ;
  static constexpr int SOLS = TMP, SOLN = SOLS + L::SOL, SLK = SOLN + L::SOL, DUL = SLK + L::CON, TMP_EARLY = DUL + L::CON - TMP;
  static constexpr int BL = TMP, SM = BL + NF * NV, TMP_LATE = SM + NF * NF - TMP;
  static constexpr int VEC = TMP + (TMP_EARLY > TMP_LATE ? TMP_EARLY : TMP_LATE);
  static_assert(NVF * NV <= IDC - MM, "Qafu_full must fit in the M / J blocks");
  static_assert(512 <= NVF * NVF, "ERR (two accumulators per thread in the MERIT variant) aliases MJ");
  // vectors
  static constexpr int LQ = VEC, LV = LQ + NV, LA = LV + NV, LF = LA + NV, LU = LF + NF, LUP = LU + NU, FQ = LUP + 6, FV = FQ + NV,
                       LAF = FV + NV, TLA = LAF + 32, QAA = TLA + 32,      // TLA: t = M^T beta + J^T mu (wave 1 -> stage 3)
                       BM = QAA + NV,
                       // the Lie-group terms of the base in the order of the lie record (OcpLayout Z_*)
                       LIEB = BM + 32, JQ = LIEB + L::Z_JQ, QDIFF = LIEB + L::Z_QDIFF, FQQ = LIEB + L::Z_FQQ, FQ6 = LIEB + L::Z_FQ6,
                       FQQI = LIEB + L::Z_FQQI, FQQP = LIEB + L::Z_FQQP, FQQPI = LIEB + L::Z_FQQPI,
                       HQD = LIEB + 196, HVD = HQD + NV, HUD = HVD + NV, QB6 = HUD + NU,       // diagonal / base-block Hessian terms before condensing
                       // the narrow layouts have room for M^-1 in a block of its own: M then survives the inversion, and t = M^T beta + J^T mu
                       // moves from the head of wave 1's chain to wave 3
                       MINV = SEP_MINV ? QB6 + 36 + 2 : MM,
                       // scratch of the contact Schur complement's inverse (W = L^-1): the wide layouts have no room for it behind G and Jl D^-1
                       WSM = QB6 + 36 + 2 + (SEP_MINV ? NV * NV : 0),
                       TOTAL = WSM + (SEP_MINV ? 0 : NF * NF);
  static_assert(SOLS % 2 == 0 && LIEB % 2 == 0 && IDC % 2 == 0 && MJ % 2 == 0, "16-byte pieces");
};

// BWD = true: the backward-Euler stage of ParNMPC (SplitParNMPC / TerminalParNMPC::linearizeOCP,
// include/idocp/ocp/split_parnmpc.hxx:50-84, terminal_parnmpc.hxx:50-82; state_equation.hxx:96-170;
// ContactDynamics::condenseContactDynamics(..., is_forward_euler = false)).  The lie record then holds, in the same
// places: FQQ = dSubtract_dPlus(q, q_next), FQQP = dSubtract_dMinus(q_prev, q), FQQI = dSubtract_dPlus(q_prev, q)^-1,
// FQ6 = (q_prev (-) q).head(6) (parnmpc_lie_kernel).  The chain ends with an unused placeholder stage; the last real
// stage (position M - 2) carries the terminal cost.
// MERIT = true (with RESIDUAL): the filter line search's evaluation of a trial iterate (src/line_search/line_search.cpp:63-196):
// per stage SplitOCP::stageCost without the barrier term (split_ocp.hxx:270-289; the barrier part comes from ocp_trial_kernel) and
// SplitOCP::constraintViolation (:292-346), |Fx|_1 + dt |[ID - u; C]|_1 + dt |g + slack|_1 + |P|_1; only the NOMINAL rigid-body
// sweeps run (no tangent items).  Launched on a copy of the buffers whose sol points at the trial iterate.
// XYY: the joint axes of the legs are known at compile time (OcpBuffers::leg_axes_xyy, dev_rnea_tangent.hpp JointFrame).
// Barrier of the workgroup for exchanges through LDS: the LDS accesses of the wavefront are complete (s_waitcnt lgkmcnt(0)) before and
// visible after; the global stores and loads in flight are NOT waited for, as __syncthreads() would (s_waitcnt vmcnt(0): the kkt / exp
// entries a phase has just stored would have to reach memory before the next phase may start).  Nothing in the kernel hands data from
// one wavefront to another through global memory.
__device__ __forceinline__ void blockLdsBarrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

template <typename D, bool RESIDUAL, int DIMF, bool BWD = false, bool MERIT = false, bool XYY = false, bool EVENTS = false>
__global__ __launch_bounds__(256, 4) void ocp_condense_kernel(OcpBuffers B, const double* __restrict__ q0, const double* __restrict__ v0 = nullptr,
                                                              const int* __restrict__ plist = nullptr, int nlist = 0) {
  using L = OcpLayout<D>;
  using S = CondenseSmem<D, (DIMF > 0) ? DIMF : (DIMF == 0 ? D::NF / 2 : D::NF)>;      // (flight stages, DIMF = 0: the narrow layout, nothing of it used by contacts)
  constexpr int NV = D::NV, NQ = D::NQ, NX = D::NX, NF = D::NF, NVF = D::NVF, NU = D::NU, NC = D::NC;
  constexpr int SF = (DIMF > 0) ? DIMF : (DIMF == 0 ? NF / 2 : NF), SVF = S::NVF, RVF = NV + SF;       // LDS leading dimensions (CondenseSmem<D, SF>) and rows; NF / NVF: the HBM records'
  extern __shared__ __attribute__((aligned(16))) double sm[];
  __shared__ int s_ok, s_c1, s_ba, s_gt, s_m;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  int tid = threadIdx.x;
  // the roles of the wave-specialised front half (wave 1 carries the long M^-1 chain) rotate with the workgroup, so that the workgroups
  // resident on a CU do not all put the same role on the same SIMD (2.745 against 2.77 ms; everything below derives from tid)
  tid = (((tid >> 6) + (int)(blockIdx.x & 3)) & 3) * 64 + (tid & 63);
  constexpr int nt = 256;
  const int per = plist ? nlist : M;              // units: batch * M, one stage of the chain each (or batch * nlist: the chain positions of one
                                                  // stage class, launchCondenseMixed)
  // (The body is a lambda on purpose: as the kernel's own top-level block the same code spilled 4 - 26 registers in the tangent
  //  items; a persistent variant -- a workgroup walking over several units, the next unit's records fetched by LDS DMA during the
  //  back half -- was measured as well: loop-invariant hoisting out of the unit loop brings the spills back and the stage-0 wait
  //  then also covers the previous unit's output stores: 4.16 instead of 3.44 ms, not kept.)
  auto stage = [&](const long unit) {
  const long b = unit / per;
  const int pos = plist ? plist[unit - b * per] : (int)(unit - b * per);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const bool terminal = (pos == M - 1);
  if (BWD && terminal) {                            // placeholder stage (no cost, no constraint: the line search sums zeros)
    if (MERIT && threadIdx.x < 2) B.merit_stage[(b * B.NS + nd->slot) * 4 + threadIdx.x] = 0.0;
    return;
  }
  const bool last = BWD && P->has_terminal && (pos == M - 2);   // ParNMPC: the stage that carries the terminal cost
  // DIMF >= 0 is only launched on stages without an impulse or a switching constraint (launchCondense: event-free chains;
  // launchCondenseMixed: those stages of a chain with events, grouped by their number of contact rows)
  // EVENTS: a compile-time contact count on stages that DO carry an impulse or a switching constraint (the event stages of a trot all have
  // half of the feet: as members of the general class, with its run-time count, they cost 1.5 x a plain stage)
  constexpr bool PLAIN = (DIMF >= 0) && !EVENTS;
  const bool impulse = PLAIN ? false : (nd->kind == 1);
  if (BWD && impulse) return;                       // ParNMPC: the backward-Euler impulse stage is K9i (parnmpc_event_kernels.hip)
  const int sw_dimi = PLAIN ? 0 : nd->sw_dimi;
  int c_act[NC], c_row[NC];                        // contact status of the stage (uniform; fetched once, up front)
#pragma unroll
  for (int c = 0; c < NC; ++c) { c_act[c] = nd->active[c]; c_row[c] = nd->row_of[c]; }
  const int i = nd->level;                          // constraint gating level
  const double dt = nd->dt;                         // scaling of cost / constraints / dynamics multipliers (1 on impulse stages)
  const double dtq = nd->dtq;                       // q+ = q (+) dtq v (0 on impulse stages)
  // DIMF >= 0: the number of active contact rows is a compile-time constant (every loop bound and
  // index division below folds); DIMF < 0: read it from the stage's node.
  const int dimf = (DIMF >= 0) ? DIMF : nd->dimf, dimvf = NV + dimf;
  const int dimfd = (DIMF == 0) ? 1 : dimf;          // divisor of the index splits (the flight class, DIMF = 0, never runs those loops)
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ s_g = B.sol + rec * L::SOL;
  const double* __restrict__ sn_g = B.sol + (b * B.NS + (terminal ? nd->slot : nd->next)) * L::SOL;
  const double* s = &sm[S::SOLS];                   // LDS copies of this stage's and the next stage's solution records
  const double* sn = &sm[S::SOLN];                  // (only valid for non-terminal stages)
  const double* q = s + L::S_Q;
  const double* __restrict__ qref = B.q_ref + (long)pos * NQ;
  const long su = rec;
#ifdef IDOCP_K5_STAMPS      // diagnostic build (IDOCP_EXTRA_HIPCC_FLAGS=-DIDOCP_K5_STAMPS): clock stamps at the phase boundaries of one workgroup
  const bool stamp = (!RESIDUAL) && tid == 0 && unit == (long)(gridDim.x / 2 + 7) && B.prof != nullptr && DIMF == B.prof_dimf;
#define STAMP(k) do { if (stamp) B.prof[k] = wall_clock64(); } while (0)
  // per-wavefront stamps of the wave-specialised stage: slot 32 + 8 wave + k
  const bool stampw = (!RESIDUAL) && (tid & 63) == 0 && unit == (long)(gridDim.x / 2 + 7) && B.prof != nullptr && DIMF == B.prof_dimf;
#define STAMPW(k) do { if (stampw) B.prof[32 + 8 * (tid >> 6) + (k)] = wall_clock64(); } while (0)
#else
#define STAMP(k) do { } while (0)
#define STAMPW(k) do { } while (0)
#endif
  STAMP(0);
  double* kk = B.kkt + rec * L::KKT;
  double* ee = B.exp + rec * L::EXP;
  // cost weights of this stage: the impulse stage has its own (trotting_configuration_space_cost.cpp:308-376)
  const double* __restrict__ w_q = impulse ? P->qi_weight : P->q_weight;
  const double* __restrict__ w_v = impulse ? P->vi_weight : P->v_weight;
  const double* __restrict__ w_a = impulse ? P->dvi_weight : P->a_weight;

  // ======================================================================================================================
  // Front half, WAVE-SPECIALISED.  Round 1 ran K5a -> lin record -> phases A .. E of this kernel one after the other, nineteen
  // workgroup barriers in all; round 2 produced the rigid-body terms in place (dev_rnea_tangent.hpp) with the NOMINAL sweeps on five
  // lanes of two wavefronts (5 us of the 28 us a stage spent here).  Round 3: the nominal sweeps are a kernel of their own
  // (ocp_nominal_kernel: one lane per stage and leg) whose record stage 0 copies into the scratch, so the tangent items start at once:
  //   stage 0  all      fetch the small records and the nominal record (registers), clear the output blocks, model constants -> LDS
  //   stage 1  wave 0   the 60 q- and v-seed tangent items
  //            wave 1   the 36 a-seed items (M, J) and then, without waiting for the q / v items, Robot::computeMJtJinv:
  //                     block-arrow M^-1, J M^-1, (J M^-1 J^T)^-1
  //            wave 2/3 C1: every term of the gradients / Hessian diagonals that does not need the rigid-body derivatives
  //                     (wave 1 waits for them through an LDS flag before it reads their vectors / reuses their input block)
  //   stage 3  all      base rows + position term of the q / v columns, assembly of MJtJinv
  //   (then F: MJtJinv [dIDCdqv, IDC] together with C2, the multiplier terms l += dt [dID; dC]^T [beta; mu])
  // ======================================================================================================================
  using RS = RneaScratch<D>;
  using RI = RneaItems<D>;
  static_assert(S::QFF - S::MJ >= RS::TOTAL, "the RNEA scratch lives in MJ .. MJD (dead until stage 3)");
  static_assert(S::MJ % 2 == 0 && S::IDC % 2 == 0, "16-byte pieces of the nominal record");
  static_assert(RI::NQV <= 64 && RI::NA <= 64, "one wavefront per item list");
  constexpr int NPRE = (2 * L::SOL + 2 * L::CON + nt - 1) / nt;
  static_assert(S::SOLN == S::SOLS + L::SOL && S::SLK == S::SOLN + L::SOL && S::DUL == S::SLK + L::CON, "the fetched records are contiguous in LDS");
  const int wave = tid >> 6, lane = tid & 63;
  double* const sc = &sm[S::MJ];
  RneaOut out;
  out.didc = &sm[S::DIDC]; out.ldd = SVF; out.mm = &sm[S::MM]; out.jm = &sm[S::JM]; out.ldj = SF; out.idc = &sm[S::IDC];
  // ---- stage 0 ----
  {
    // (every address below is a valid record of this instance, so the loads are unconditional: no branches, all in flight at once;
    //  on the terminal stage sn_g = s_g and the slack / dual copies are simply not used)
    double pre[NPRE], prez[7];
    RneaNominalCopy<D, nt> nomc;
    const bool has_nom = !terminal;      // (impulse stages have nominal records of their own kind, ocp_nominal_kernel<.., IMP>)
    if (has_nom) nomc.fetch(B.nom + rec * L::NOM, tid);
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      const int e = tid + nt * t;
      const double* src = e < L::SOL ? s_g + e : (e < 2 * L::SOL ? sn_g + (e - L::SOL) : (e < 2 * L::SOL + L::CON ? B.slack + su * L::CON + (e - 2 * L::SOL)
                                                                         : B.dual + su * L::CON + (e < 2 * L::SOL + 2 * L::CON ? e - 2 * L::SOL - L::CON : 0)));
      pre[t] = *src;
    }
    {
      const double* __restrict__ zz = B.lie + rec * L::LIE;
      if (tid < 36) { prez[0] = zz[L::Z_JQ + tid]; prez[1] = zz[L::Z_FQQ + tid]; prez[2] = zz[L::Z_FQQI + tid]; prez[3] = zz[L::Z_FQQP + tid]; prez[4] = zz[L::Z_FQQPI + tid]; }
      if (tid >= 64 && tid < 70) { prez[5] = zz[L::Z_QDIFF + tid - 64]; prez[6] = zz[L::Z_FQ6 + tid - 64]; }
    }
    STAMP(3);
    if (tid == 0) { s_ok = 1; s_c1 = 0; s_ba = 0; s_gt = 0; s_m = 0; }
    if (!terminal) {
      for (int e = tid; e < S::IDC - S::DIDC; e += nt) sm[S::DIDC + e] = 0.0;      // rows a seed does not reach, inactive contacts
      rneaSetup<D>(B.model, P, nd, tid, sc);
      nomc.store(tid, sc, &sm[S::IDC]);
    }
    for (int e = tid; e < SF * SF; e += nt) sm[S::QFF + e] = 0.0;
#pragma unroll
    for (int t = 0; t < NPRE; ++t) {
      const int e = tid + nt * t;
      if (e < 2 * L::SOL + 2 * L::CON) sm[S::SOLS + e] = pre[t];
    }
    if (tid < 36) { sm[S::JQ + tid] = prez[0]; sm[S::FQQ + tid] = prez[1]; sm[S::FQQI + tid] = prez[2]; sm[S::FQQP + tid] = prez[3]; sm[S::FQQPI + tid] = prez[4]; }
    if (tid >= 64 && tid < 70) { sm[S::QDIFF + tid - 64] = prez[5]; sm[S::FQ6 + tid - 64] = prez[6]; }
    STAMP(4);
  }
  blockLdsBarrier();
  STAMP(1);

  const double vref_on = nd->vref_on;          // TimeVaryingConfigurationSpaceCost::v_ref(t): zero outside the window
  const double v_ref0 = vref_on * P->v_ref[0];
  if (terminal) {
    // ---- TerminalOCP::linearizeOCP ----
    if (tid < NV) {
      const int r = tid;
      double lq, lv;
      if (r < 6) {
        lq = 0.0;
        for (int m2 = 0; m2 < 6; ++m2) lq += sm[S::JQ + m2 + 6 * r] * P->qf_weight[m2] * sm[S::QDIFF + m2];
        double t1 = 0.0;
        for (int m2 = 0; m2 < 6; ++m2) t1 += sm[S::FQQP + m2 + 6 * r] * s[L::S_LMD + m2];
        lq += t1;
      } else {
        lq = P->qf_weight[r] * (q[r + 1] - qref[r + 1]) - s[L::S_LMD + r];
      }
      lv = P->vf_weight[r] * (s[L::S_V + r] - (r == 0 ? v_ref0 : vref_on * P->v_ref[r])) - s[L::S_GMM + r];
      if (B.ext) lq += B.ext[rec * L::EXT + L::X_LQ + r];      // task-space cost with its terminal weights (ocp_ext_kernel.hip)
      if (MERIT) {      // TerminalOCP::terminalCost (terminal_ocp.hxx:81-87)
        const double qd = r < 6 ? sm[S::QDIFF + r] : q[r + 1] - qref[r + 1], dvr = s[L::S_V + r] - (r == 0 ? v_ref0 : vref_on * P->v_ref[r]);
        sm[S::ERR + tid] = 0.5 * (P->qf_weight[r] * qd * qd + P->vf_weight[r] * dvr * dvr);
      }
      else if (RESIDUAL) { sm[S::ERR + tid] = lq * lq + lv * lv; }
      else { kk[L::K_LX + r] = lq; kk[L::K_LX + NV + r] = lv; }
    } else if (RESIDUAL) {
      sm[S::ERR + tid] = 0.0;
    }
    if (RESIDUAL) {
      blockLdsBarrier();
      if (tid == 0) {
        double e = 0.0; for (int t = 0; t < nt; ++t) e += sm[S::ERR + t];
        if (MERIT) { B.merit_stage[rec * 4] = e + (B.ext ? B.ext[rec * L::EXT + L::X_COST] : 0.0); B.merit_stage[rec * 4 + 1] = 0.0; } else B.err_stage[rec] = e;
      }
      return;
    }
    for (int e = tid; e < NX * NX; e += nt) {
      const int c = e / NX, r = e - NX * c;
      double val = 0.0;
      if (r < 6 && c < 6) { for (int m2 = 0; m2 < 6; ++m2) val += sm[S::JQ + m2 + 6 * r] * P->qf_weight[m2] * sm[S::JQ + m2 + 6 * c]; }
      else if (r == c) val = (r < NV) ? P->qf_weight[r] : P->vf_weight[r - NV];
      if (r <= c) kk[L::K_QXX + L::xsym(r, c)] = val;
    }
    if (tid < 36) ee[L::E_FQQPI + tid] = sm[S::FQQPI + tid];
    return;
  }

  // ---- stage 1: tangent items (wave 0: q, v seeds; wave 1: a seeds) and Robot::computeMJtJinv (wave 1) (robot.hxx:576-615), next to
  // C1 (waves 2, 3) ----
  // M^-1 and (J M^-1 J^T)^-1 by Gauss-Jordan on the SPD blocks (the reference uses pinocchio's sparse Cholesky + Eigen::LLT; same
  // inverses up to rounding).  BL, SM live in the block that holds the solution / slack / dual copies C1 reads: wave 1 waits for C1
  // (s_c1: one count per C1 wavefront) before it reads C1's vectors and before it writes there.
  const double gz = B.model->gravity[2];
  const double bwv = 2.0 / P->baumgarte_time_step, bwp = 1.0 / (P->baumgarte_time_step * P->baumgarte_time_step);
  const double* slack = &sm[S::SLK];
  const double* dual = &sm[S::DUL];
  double err_local = 0.0, err_ipm = 0.0;     // RESIDUAL: plain squared residuals / IPM residuals (weighted by dt^2 below), per thread
  double merit_cost = 0.0, merit_viol = 0.0; // MERIT: this thread's share of the stage cost / l1 constraint violation
  // C2 for the acceleration rows: t = M^T beta + J^T mu (beta, mu from the LDS copy of the solution record; la += dt t follows in stage 3).
  // Needs M as the a-seed items left it: by wave 1 in front of the in-place inversion, or by wave 3 where M^-1 has a block of its own.
  auto accelerationMultiplierTerm = [&](int lane_) {
    if (lane_ < NV) {
      double acc = dotAny(&sm[S::MM + NV * lane_], 1, s + L::S_BETA, 1, NV);
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        if (!c_act[c]) continue;
        const double* jc = &sm[S::JM + c_row[c] + SF * lane_];
        acc += jc[0] * s[L::S_MU + 3 * c] + jc[1] * s[L::S_MU + 3 * c + 1] + jc[2] * s[L::S_MU + 3 * c + 2];
      }
      sm[S::TLA + lane_] = acc;
    }
  };
  if (wave == 0) {
    if (!MERIT && lane < RI::NQV) {
      if (!PLAIN && impulse) rneaTangentItem<D, XYY, true>(0.0, bwv, RI::qv(lane), sc, out);      // impulse stage: dynamics at zero velocity, contact velocity
      else rneaTangentItem<D, XYY, false>(gz, bwv, RI::qv(lane), sc, out);
    }
    STAMPW(0);
  } else if (wave == 1) {
    if (!MERIT) {
      {
        if (lane < RI::NA) rneaTangentItemA<D, XYY>(RI::a(lane), sc, out);
        waveLdsSync();
        rneaAssembleA<D>(lane, sc, out);
        waveLdsSync();
#ifdef IDOCP_K5_REP_AITEMS      // timing experiment: the a items a second time (idempotent) -- what they cost on wave 1's chain
        asm volatile("" ::: "memory");
        if (lane < RI::NA) rneaTangentItemA<D, XYY>(RI::a(lane), sc, out);
        waveLdsSync();
        rneaAssembleA<D>(lane, sc, out);
        waveLdsSync();
#endif
      }
      STAMPW(0);
      if (!RESIDUAL) {
        // C2 for the acceleration rows, while M is still M: t = M^T beta + J^T mu (beta, mu from the LDS copy of the solution record;
        // la += dt t follows in stage 3 -- nothing here waits for C1)
        if constexpr (S::SEP_MINV) ldsFlagSet(&s_m, 1, lane);      // M and J stand (and stay): wave 3 forms t
        else { accelerationMultiplierTerm(lane); waveLdsSync(); }
        STAMPW(4);
        // The contact Schur complement does not wait for M^-1: with M^-1 in its block-arrow form, J = [Jb Jl] and G = Jb - Jl E^T,
        //   J M^-1 = [G S^-1 | G (-S^-1 E) + Jl D^-1],   J M^-1 J^T = (G S^-1) G^T + (Jl D^-1) Jl^T,
        // so wave 2 forms G and Jl D^-1 as soon as E and D^-1 exist, the product and its inverse as soon as S^-1 does, and wave 3 the right part
        // of BL once the top-right block does (below, behind C1) -- next to the rest of this inverse instead of behind it (round 4: 2.4 of
        // the 9.1 us of this wavefront's chain)
#ifdef IDOCP_K5_STAMPS
        blockArrowInverse<6, D::NL, D::LJ>(&sm[S::MINV], NV, lane, &s_ok, stampw ? B.prof + 16 : nullptr, &s_ba, &s_gt, &sm[S::MM]);
#else
        blockArrowInverse<6, D::NL, D::LJ>(&sm[S::MINV], NV, lane, &s_ok, nullptr, &s_ba, &s_gt, &sm[S::MM]);
#endif
        STAMPW(1);
        STAMPW(3);
      }
    }
  } else if (tid < 128 + NV) {
    // ---- C1, one row of (q, v, a) per thread: cost, state equation, joint limits, switching-constraint multipliers ----
    const int r = tid - 128;
    const double vr = s[L::S_V + r], ar = s[L::S_A + r];
    const double lmd = s[L::S_LMD + r], gmm = s[L::S_GMM + r], lmdn = sn[L::S_LMD + r], gmmn = sn[L::S_GMM + r];
    double lq, lv, la, hq = 0.0, hv, ha;
    // backward Euler: the predecessor's state (the measured state in front of the first stage)
    const double* __restrict__ sp_g = (BWD && nd->prev >= 0) ? B.sol + (b * B.NS + nd->prev) * L::SOL : nullptr;
    const double qpr = !BWD ? 0.0 : (sp_g ? sp_g[L::S_Q + r + 1] : q0[b * NQ + r + 1]);
    const double vpr = !BWD ? 0.0 : (sp_g ? sp_g[L::S_V + r] : v0[b * NV + r]);
    double fq;
    if (r < 6) {
      lq = 0.0;
      for (int m2 = 0; m2 < 6; ++m2) lq += sm[S::JQ + m2 + 6 * r] * w_q[m2] * sm[S::QDIFF + m2];
      lq *= dt;
      double t1 = 0.0;
      for (int m2 = 0; m2 < 6; ++m2) t1 += sm[S::FQQ + m2 + 6 * r] * sn[L::S_LMD + m2] + sm[S::FQQP + m2 + 6 * r] * s[L::S_LMD + m2];
      lq += t1;
      fq = sm[S::FQ6 + r] + dtq * vr;
    } else {
      lq = dt * w_q[r] * (q[r + 1] - qref[r + 1]) + lmdn - lmd;
      hq = dt * w_q[r];
      fq = BWD ? (qpr - q[r + 1] + dtq * vr) : (q[r + 1] - sn[L::S_Q + r + 1] + dtq * vr);
    }
    sm[S::FQ + r] = fq;
    lv = dt * w_v[r] * (vr - (r == 0 ? v_ref0 : vref_on * P->v_ref[r])) + (BWD ? dtq * lmd : dtq * lmdn) + gmmn - gmm;
    la = dt * w_a[r] * ar + dt * (BWD ? gmm : gmmn);
    hv = dt * w_v[r];
    ha = dt * w_a[r];
    const double fv = BWD ? (vpr - vr + dt * ar) : (vr + dt * ar - sn[L::S_V + r]);
    sm[S::FV + r] = fv;
    if (MERIT) {      // (Trotting / TimeVarying)ConfigurationSpaceCost::computeStageCost / computeImpulseCost; |Fq|_1 + |Fv|_1
      const double qd = r < 6 ? sm[S::QDIFF + r] : q[r + 1] - qref[r + 1], dvr = vr - (r == 0 ? v_ref0 : vref_on * P->v_ref[r]);
      merit_cost += 0.5 * dt * (w_q[r] * qd * qd + w_v[r] * dvr * dvr + w_a[r] * ar * ar);
      merit_viol += fabs(fq) + fabs(fv);
    }
    if (last) {
      // TerminalParNMPC: + terminal cost (computeTerminalCostDerivatives / Hessian) on the last stage
      if (r < 6) {
        double tq = 0.0;
        for (int m2 = 0; m2 < 6; ++m2) tq += sm[S::JQ + m2 + 6 * r] * P->qf_weight[m2] * sm[S::QDIFF + m2];
        lq += tq;
      } else {
        lq += P->qf_weight[r] * (q[r + 1] - qref[r + 1]);
        hq += P->qf_weight[r];
      }
      lv += P->vf_weight[r] * (vr - (r == 0 ? v_ref0 : vref_on * P->v_ref[r]));
      hv += P->vf_weight[r];
    }
    // joint position / velocity limits act on the actuated joints (tail(dimu))
    if (r >= 6) {
      const int j = r - 6;
      for (int c = 0; c < 4; ++c) {
        if (!ocpRowValid(P, c, i, impulse)) continue;
        const double sgn = (c & 1) ? 1.0 : -1.0;
        const double x = (c < 2) ? q[r + 1] : vr;
        const double sl = slack[c * NU + j], du = dual[c * NU + j];
        const double res = sgn * (x - ocpLimit(P, c, j)) + sl, duality = sl * du - P->barrier;
        double g = sgn * dt * du, h = 0.0;
        if (MERIT) merit_viol += dt * fabs(res);
        if (RESIDUAL) err_ipm += res * res + duality * duality;
        else { const double isl = recipNewton(sl); g += sgn * dt * (du * res - duality) * isl; h = dt * du * isl; }
        if (c < 2) { lq += g; hq += h; } else { lv += g; hv += h; }
      }
      // JointAccelerationLowerLimit / UpperLimit (joint_acceleration_{lower,upper}_limit.cpp:57-75): la, Qaa diagonal
      if (P->use_a_lower | P->use_a_upper) {
        for (int c = 8; c < 10; ++c) {
          if (!ocpRowValid(P, c, i, impulse)) continue;
          const double sgn = (c & 1) ? 1.0 : -1.0;
          const int row = L::C_ACC + (c - 8) * NU + j;
          const double sl = slack[row], du = dual[row];
          const double res = sgn * (ar - ocpLimit(P, c, j)) + sl, duality = sl * du - P->barrier;
          double g = sgn * dt * du, h = 0.0;
          if (MERIT) merit_viol += dt * fabs(res);
          if (RESIDUAL) err_ipm += res * res + duality * duality;
          else { const double isl = recipNewton(sl); g += sgn * dt * (du * res - duality) * isl; h = dt * du * isl; }
          la += g; ha += h;
        }
      }
    }
    // ForwardSwitchingConstraint::linearizeSwitchingConstraint (forward_switching_constraint.hxx:49-51): + Phi^T xi
    if (sw_dimi > 0) {
      const double* __restrict__ W = B.swc + rec * L::SWC;
      for (int j = 0; j < sw_dimi; ++j) {
        const double xi = s[L::S_XI + j];
        lq += W[L::W_PHIX + j + NF * r] * xi; lv += W[L::W_PHIX + j + NF * (NV + r)] * xi; la += W[L::W_PHIA + j + NF * r] * xi;
      }
    }
    // the multipliers of [ID; C] are added in C2; bm = [beta ; mu_stack] for them
    if (B.ext) lq += B.ext[rec * L::EXT + L::X_LQ + r];      // terms with frame Jacobians of their own (ocp_ext_kernel.hip; ContactDistance)
    sm[S::LQ + r] = lq; sm[S::LV + r] = lv; sm[S::LA + r] = la; sm[S::QAA + r] = ha;
    sm[S::BM + r] = s[L::S_BETA + r];
    if (!RESIDUAL) { sm[S::HQD + r] = hq; sm[S::HVD + r] = hv; }
    else err_local += fq * fq + fv * fv;
  } else if (tid >= 128 + 32 && tid < 128 + 32 + NU) {
    // torque rows: lu, Quu diagonal.  An impulse stage has no torques: lu = 0 and a unit Quu make the
    // Riccati step below return K = 0, k = 0 and P = F, i.e. ImpulseSplitRiccatiFactorizer's recursion.
    const int j = tid - 128 - 32;
    const double u = s[L::S_U + j];
    double lu = dt * P->u_weight[j] * (u - P->u_ref[j]) - dt * s[L::S_BETA + 6 + j];
    double h = dt * P->u_weight[j];
    if (impulse) { lu = 0.0; h = 1.0; }
    if (MERIT && !impulse) merit_cost += 0.5 * dt * P->u_weight[j] * (u - P->u_ref[j]) * (u - P->u_ref[j]);
    for (int c = 4; c < 6; ++c) {
      if (!ocpRowValid(P, c, i, impulse)) continue;
      const double sgn = (c & 1) ? 1.0 : -1.0;
      const double sl = slack[c * NU + j], du = dual[c * NU + j];
      const double res = sgn * (u - ocpLimit(P, c, j)) + sl, duality = sl * du - P->barrier;
      if (MERIT) merit_viol += dt * fabs(res);
      lu += sgn * dt * du;
      if (RESIDUAL) err_ipm += res * res + duality * duality;
      else { const double isl = recipNewton(sl); lu += sgn * dt * (du * res - duality) * isl; h += dt * du * isl; }
    }
    sm[S::LU + j] = lu;
    if (!RESIDUAL) sm[S::HUD + j] = h;
    else err_local += lu * lu;
  } else if (tid >= 128 + 48 && tid < 128 + 48 + 6) {
    // passive (floating-base) rows: lu_passive = dt nu_passive - dt beta.head(6)
    const int r = tid - 128 - 48;
    const double lup = impulse ? 0.0 : dt * s[L::S_NUP + r] - dt * s[L::S_BETA + r];
    sm[S::LUP + r] = lup;
    if (RESIDUAL) err_local += lup * lup;
  } else if (tid >= 192 && tid < 192 + NC) {
    // contact-force rows: ContactForceCost + LinearizedFrictionCone (the - dt J beta term follows in C2)
    const int c = tid - 192;
    if (nd->active[c]) {
      const int row = nd->row_of[c];
      double lf[3], f[3];
      for (int x = 0; x < 3; ++x) {
        f[x] = s[L::S_F + 3 * c + x];
        const double wf = impulse ? P->fi_weight[c][x] : P->f_weight[c][x], rf = impulse ? P->fi_ref[c][x] : P->f_ref[c][x];
        lf[x] = dt * wf * (f[x] - rf);
        if (MERIT) merit_cost += 0.5 * dt * wf * (f[x] - rf) * (f[x] - rf);      // ContactForceCost::computeStageCost / computeImpulseCost
        if (!RESIDUAL) sm[S::QFF + (row + x) + SF * (row + x)] = dt * wf;
        sm[S::BM + NV + row + x] = s[L::S_MU + 3 * c + x];
      }
      if (ocpRowValid(P, 6, i, impulse)) {
        // (Linearized)(Impulse)FrictionCone: augmentDualResidual + condenseSlackAndDual, row by row (coneRow, ocp_device.hpp)
        const int ck = impulse ? P->impulse_cone_kind : P->cone_kind, nr = coneRows(ck);
        double dd[5], Jr[5][3];
        for (int r = 0; r < 5; ++r) {
          if (r >= nr) { dd[r] = 0.0; Jr[r][0] = Jr[r][1] = Jr[r][2] = 0.0; continue; }
          const int idx = L::C_FRIC + 5 * c + r;
          const double sl = slack[idx], du = dual[idx];
          const double g = coneRow(ck, P->mu, r, f, Jr[r]);
          const double res = g + sl, duality = sl * du - P->barrier;
          double coef = du;
          const double isl = recipNewton(sl);
          if (MERIT) merit_viol += dt * fabs(res);
          if (RESIDUAL) err_ipm += res * res + duality * duality;
          else coef += (du * res - duality) * isl;
          dd[r] = du * isl;
          for (int x = 0; x < 3; ++x) lf[x] += dt * Jr[r][x] * coef;
        }
        if (!RESIDUAL) {
          for (int x = 0; x < 3; ++x) for (int y = 0; y < 3; ++y) {
            double acc = 0.0;
            for (int r = 0; r < 5; ++r) acc += Jr[r][x] * dd[r] * Jr[r][y];
            sm[S::QFF + (row + x) + SF * (row + y)] += dt * acc;
          }
        }
      }
      for (int x = 0; x < 3; ++x) sm[S::LF + row + x] = lf[x];
    }
  } else if (!RESIDUAL && tid >= 192 + 8 && tid < 192 + 8 + 36) {
    // cost Hessian of the base block dt Jq^T W Jq (6 x 6), and condenseForwardEuler (state_equation.hxx:40-63)
    const int e = tid - 192 - 8, c = e / 6, r = e - 6 * c;
    double acc = 0.0;
    for (int m2 = 0; m2 < 6; ++m2) acc += sm[S::JQ + m2 + 6 * r] * w_q[m2] * sm[S::JQ + m2 + 6 * c];
    double accf = 0.0;
    if (last) for (int m2 = 0; m2 < 6; ++m2) accf += sm[S::JQ + m2 + 6 * r] * P->qf_weight[m2] * sm[S::JQ + m2 + 6 * c];
    sm[S::QB6 + e] = dt * acc + accf;
    double fqq = 0.0;
    for (int m2 = 0; m2 < 6; ++m2) fqq += sm[S::FQQI + r + 6 * m2] * sm[(BWD ? S::FQQP : S::FQQ) + m2 + 6 * c];
    kk[L::K_FQQ + e] = BWD ? fqq : -fqq;                              // Fqq = -Fqq_inv * Fqq  (backward Euler: + Fqq_inv * Fqq)
    kk[L::K_FQV + e] = (BWD ? dtq : -dtq) * sm[S::FQQI + e];          // Fqv = -dt Fqq_inv (0 on impulse stages; backward: + dt Fqq_inv)
    ee[L::E_FQQPI + e] = sm[(BWD ? S::FQQI : S::FQQPI) + e];          // what the costate correction needs
  }
  if (wave >= 2) STAMPW(0);
  if (wave >= 2 && !MERIT && !RESIDUAL) {
    if constexpr (S::SEP_MINV) {
      if (wave == 3) { ldsFlagWait(&s_m, 1, lane); accelerationMultiplierTerm(lane); }      // (reads the solution copy: before this wavefront releases that block)
    }
    // C1 of this wavefront is in LDS: tell wave 1
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __hip_atomic_fetch_add(&s_c1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // ---- the contact Schur complement next to wave 1's inverse (see there).  G and T = Jl D^-1 live behind BL and SM in the block C1 read. ----
    constexpr int NJ = D::NL * D::LJ, LJ = D::LJ;
    constexpr int GG = S::SM + SF * SF, TT = GG + SF * 6;
    static_assert(TT + SF * NJ <= S::VEC, "G and Jl D^-1 fit behind BL, SM");
    ldsFlagWait(&s_c1, 2, lane);                                   // both C1 wavefronts are done with that block
    ldsFlagWait(&s_ba, 1, lane);                                   // D^-1 (leg blocks) and E (lower-left block, transposed) stand
    if (wave == 2) {
      for (int e = lane; e < dimf * 6; e += 64) {                  // G = Jb - Jl E^T
        const int c = e / dimfd, r = e - c * dimfd;
        double acc = sm[S::JM + r + SF * c];
#pragma unroll
        for (int m = 0; m < NJ; ++m) acc -= sm[S::JM + r + SF * (6 + m)] * sm[S::MINV + (6 + m) + NV * c];
        sm[GG + r + SF * c] = acc;
      }
      for (int e = lane; e < dimf * NJ; e += 64) {                 // T = Jl D^-1 (D^-1: full symmetric LJ x LJ blocks)
        const int c = e / dimfd, r = e - c * dimfd, o = 6 + (c / LJ) * LJ;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < LJ; ++m) acc += sm[S::JM + r + SF * (o + m)] * sm[S::MINV + (o + m) + NV * (6 + c)];
        sm[TT + r + SF * c] = acc;
      }
      waveLdsSync();
      ldsFlagSet(&s_gt, 1, lane);                                  // wave 1 may overwrite D^-1 and E; wave 3 may read G, T
      ldsFlagWait(&s_ba, 2, lane);                                 // S^-1 stands in the base block
      for (int e = lane; e < dimf * 6; e += 64) {                  // BL[:, 0:6] = G S^-1
        const int c = e / dimfd, r = e - c * dimfd;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += sm[GG + r + SF * m] * sm[S::MINV + m + NV * c];
        sm[S::BL + r + SF * c] = acc;
      }
      waveLdsSync();
      for (int e = lane; e < dimf * dimf; e += 64) {               // SM = (G S^-1) G^T + T Jl^T
        const int c = e / dimfd, r = e - c * dimfd;
        double acc = 0.0;
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += sm[S::BL + r + SF * m] * sm[GG + c + SF * m];
#pragma unroll
        for (int m = 0; m < NJ; ++m) acc += sm[TT + r + SF * m] * sm[S::JM + c + SF * (6 + m)];
        sm[S::SM + r + SF * c] = acc;
      }
      waveLdsSync();
      if (dimf > 0) {                                                                 // SM = (J Minv J^T)^-1
        constexpr int WW = (TT + SF * NJ + SF * SF <= S::VEC) ? TT + SF * NJ : S::WSM;
        static_assert(WW != S::WSM || !S::SEP_MINV, "the narrow layouts keep W behind G and Jl D^-1");
        spdInverseCholDpp<SF>(&sm[S::SM], SF, dimf, lane, &s_ok, &sm[WW], SF);
      }
    } else {
      ldsFlagWait(&s_gt, 1, lane);
      ldsFlagWait(&s_ba, 3, lane);                                 // the top-right block -S^-1 E stands
      for (int e = lane; e < dimf * NJ; e += 64) {                 // BL[:, 6:] = G (-S^-1 E) + T
        const int c = e / dimfd, r = e - c * dimfd;
        double acc = sm[TT + r + SF * c];
#pragma unroll
        for (int m = 0; m < 6; ++m) acc += sm[GG + r + SF * m] * sm[S::MINV + m + NV * (6 + c)];
        sm[S::BL + r + SF * (6 + c)] = acc;
      }
    }
    STAMPW(3);
  }
  blockLdsBarrier();
  STAMP(2);
  if (MERIT) {
    // nominal [ID - u; C] (contact_dynamics.hxx:202-217; impulse stages: [ImD; C] from the lin record), the switching-constraint
    // residual P (forward_switching_constraint.hxx:27-47, from ocp_switch_kernel on the trial iterate), and the stage's totals
    rneaAssembleNominal<D>(tid, sc, out);
    blockLdsBarrier();
    if (!impulse && tid >= 6 && tid < NV && nd->has_u) sm[S::IDC + tid] -= s_g[L::S_U + tid - 6];
    blockLdsBarrier();
    if (tid < dimvf) merit_viol += dt * fabs(sm[S::IDC + tid]);
    if (sw_dimi > 0 && tid >= 200 && tid < 200 + sw_dimi) merit_viol += fabs(B.swc[rec * L::SWC + L::W_P + tid - 200]);
    blockLdsBarrier();                                                   // ERR aliases the scratch
    sm[S::ERR + tid] = merit_cost; sm[S::ERR + nt + tid] = merit_viol;
    blockLdsBarrier();
    if (tid < 2) {
      double e = 0.0;
      for (int t = 0; t < nt; ++t) e += sm[S::ERR + nt * tid + t];
      if (B.ext) e += B.ext[rec * L::EXT + (tid == 1 ? L::X_VIOL : L::X_COST)];
      B.merit_stage[rec * 4 + tid] = e;
    }
    return;
  }
  STAMP(5);
  // Everything below indexes by `tid` and reads the problem / node constants; making `tid` opaque and ordering memory here keeps the
  // compiler from hoisting those index computations and loads ABOVE the sweeps and carrying them through in registers (the q / v
  // items need the whole 128-VGPR budget).
  asm volatile("" : "+v"(tid) :: "memory");
  if (RESIDUAL) {
    // ---- C2 + SplitOCP::squaredNormKKTResidual (split_ocp.hxx:251-267); IPM residuals weighted by dt^2 (:264) ----
    rneaAssembleQV<D>(bwp, tid, nt, sc, out);
    blockLdsBarrier();
    if (!impulse && tid >= 6 && tid < NV && nd->has_u) sm[S::IDC + tid] -= s_g[L::S_U + tid - 6];      // ID - u on the actuated rows (contact_dynamics.hxx:88)
    blockLdsBarrier();
    if (tid >= 128 && tid < 128 + NV) {
      const int r = tid - 128;
      const double dq = dotAny(&sm[S::DIDC + SVF * r], 1, &sm[S::BM], 1, dimvf);
      const double dv = dotAny(&sm[S::DIDC + SVF * (NV + r)], 1, &sm[S::BM], 1, dimvf);
      const double da = dotAny(&sm[S::MM + NV * r], 1, &sm[S::BM], 1, NV) + dotAny(&sm[S::JM + SF * r], 1, &sm[S::BM + NV], 1, dimf);
      const double lq = sm[S::LQ + r] + dt * dq, lv = sm[S::LV + r] + dt * dv, la = sm[S::LA + r] + dt * da;
      const double idr = sm[S::IDC + r];
      err_local += lq * lq + lv * lv + la * la + dt * dt * idr * idr;
    } else if (tid >= 192 && tid < 192 + NC) {
      const int c = tid - 192;
      if (nd->active[c]) {
        const int row = nd->row_of[c];
        for (int x = 0; x < 3; ++x) {
          double jb = 0.0;
          for (int col = 0; col < NV; ++col) jb += sm[S::JM + (row + x) + SF * col] * sm[S::BM + col];
          const double lf = sm[S::LF + row + x] - dt * jb;
          const double cr = sm[S::IDC + NV + row + x];
          err_local += lf * lf + dt * dt * cr * cr;
        }
      }
    }
    if (sw_dimi > 0 && tid >= 200 && tid < 200 + sw_dimi) {
      const double pr = B.swc[rec * L::SWC + L::W_P + tid - 200];
      err_local += pr * pr;
    }
    blockLdsBarrier();                                                   // ERR aliases the scratch
    sm[S::ERR + tid] = err_local + (BWD ? 1.0 : dt * dt) * err_ipm;      // split_parnmpc.hxx:263 does not weight by dt^2
    blockLdsBarrier();
    if (tid == 0) {
      double e = 0.0;
      for (int t = 0; t < nt; ++t) e += sm[S::ERR + t];
      if (B.ext) e += (BWD ? 1.0 : dt * dt) * B.ext[rec * L::EXT + L::X_ERR];
      B.err_stage[rec] = e;
    }
    return;
  }
  // ---- stage 3: finish the q / v columns; assemble MJtJinv = [Minv - TR BL, TR; TR^T, -SM], TR = BL^T SM ----
  // EARLY_MJ: where MJtJinv only covers the JOINT records of the scratch (the narrow layouts; dead since the items are done) its TR
  // and -SM blocks are written NEXT to the assembly of the q / v columns, which reads the foot / base / base-force records behind
  // them: one barrier and a phase less.  The wide layouts keep the order assemble | TR, -SM | TL.
  constexpr bool EARLY_MJ = (SVF * SVF <= RS::FEET);
  auto assembleTR = [&]() {
    for (int e = tid; e < dimf * NV; e += nt) {
      const int c = e / NV, r = e - c * NV;                              // TR(r, c) = sum_p BL(p, r) SM(p, c)
      const double tr = dotAny(&sm[S::BL + SF * r], 1, &sm[S::SM + SF * c], 1, dimf);
      sm[S::MJ + r + SVF * (NV + c)] = tr;
      sm[S::MJ + (NV + c) + SVF * r] = tr;
    }
    for (int e = tid; e < dimf * dimf; e += nt) { const int c = e / dimfd, r = e - c * dimfd; sm[S::MJ + (NV + r) + SVF * (NV + c)] = -sm[S::SM + r + SF * c]; }
  };
  STAMPW(7);
  rneaAssembleQV<D>(bwp, tid, nt, sc, out);                           // reads the scratch behind the joint records
  STAMPW(5);
  if (tid >= 160 && tid < 160 + NV) sm[S::LA + tid - 160] += dt * sm[S::TLA + tid - 160];      // C2, acceleration rows (t of wave 1, stage 1)
  if (tid >= 128 && tid < 128 + 6) {
    // condenseForwardEuler: Fq.head(6) <- -+ Fqq_inv Fq.head(6)
    const int r = tid - 128;
    double acc = 0.0;
    for (int m2 = 0; m2 < 6; ++m2) acc += sm[S::FQQI + r + 6 * m2] * sm[S::FQ + m2];
    sm[S::FQ6 + r] = BWD ? acc : -acc;
  }
  if constexpr (EARLY_MJ) assembleTR();
  STAMPW(6);
  blockLdsBarrier();
  if (tid < dimvf) {
    // ID - u on the actuated rows (contact_dynamics.hxx:88); [ID; C] also becomes column NX of dIDCdqv: MJtJinv [ID; C] then falls out
    // of the tile product below (round 2: a 24-term dot product per row on 24 threads, 1 us)
    double val = sm[S::IDC + tid];
    if (!impulse && tid >= 6 && tid < NV && nd->has_u) { val -= s_g[L::S_U + tid - 6]; sm[S::IDC + tid] = val; }
    sm[S::DIDC + tid + SVF * NX] = val;
  }
  if (tid >= 64 && tid < 64 + 6) sm[S::FQ + tid - 64] = sm[S::FQ6 + tid - 64];
  if constexpr (!EARLY_MJ) {
    assembleTR();
    blockLdsBarrier();
  }
  STAMP(6);
  // TL = Minv - TR BL
  for (int e = tid; e < NV * NV; e += nt) {
    const int c = e / NV, r = e - c * NV;
    sm[S::MJ + r + SVF * c] = sm[S::MINV + e] - dotAny(&sm[S::MJ + r + SVF * NV], SVF, &sm[S::BL + SF * c], 1, dimf);
  }
  // ---- C2: multipliers of [ID; C]:  l += dt [dID;dC]^T [beta; mu]; contact rows: - dt J beta ----
  // four lanes per dot product (a quarter of the terms each, summed over the quad): round 2 had 18 + 4 threads walk 24-term products
  {
    const int part = tid & 3;
    double acc = 0.0;
    int dst = -1;
    double sgn = dt;
    if (tid < 4 * NX) {
      const int g = tid >> 2;                                           // column g of [dID; dC] / d(q, v)
      const int q4 = (dimvf + 3) >> 2, k0 = part * q4, k1 = (k0 + q4 < dimvf) ? k0 + q4 : dimvf;
      const double* col = &sm[S::DIDC + SVF * g];
      for (int k = k0; k < k1; ++k) acc += col[k] * sm[S::BM + k];
      dst = (g < NV) ? S::LQ + g : S::LV + g - NV;      // (the acceleration rows were completed by wave 1 before it inverted M)
    } else if (tid >= 160 && tid < 160 + 4 * SF) {
      const int r = (tid - 160) >> 2;                                   // packed contact row r: - dt (J beta)_r
      if (r < dimf) {
        constexpr int q4 = (NV + 3) / 4;
        const int k0 = part * q4, k1 = (k0 + q4 < NV) ? k0 + q4 : NV;
        for (int col = k0; col < k1; ++col) acc += sm[S::JM + r + SF * col] * sm[S::BM + col];
        dst = S::LF + r; sgn = -dt;
      }
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0 && dst >= 0) sm[dst] += sgn * acc;
  }
  blockLdsBarrier();

  const double hu = (PLAIN || nd->has_u) ? 1.0 : 0.0;          // impulse stages have no torque variables: Qafu = 0, Fvu = 0
  STAMP(7);
  // ---- F/G. MJtJinv * [dIDCdqv, IDC], Qafqv, Qafu_full, laf (contact_dynamics.hxx:112-128) ----
  // (MJ symmetric: MJ * dIDC = MJ^T dIDC, both operands contiguous along the contraction index; 16 x 16 tiles on the matrix cores)
  {
    // tile pairs (rows 0..15 | 16..) x (columns 16 jb ..): the pair shares its dIDC operand; wavefront jb takes pair jb
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int TX = (NX + 15) / 16;
    static_assert(RVF <= 32, "two row tiles");
    for (int jb = wave; jb < TX; jb += nt >> 6) {
      mfma_d4 a0, a1;
      const int r1 = dimvf > 16 ? dimvf - 16 : 1;
      K5_TILE_PAIR<RVF>(&sm[S::MJ], dimvf < 16 ? dimvf : 16, &sm[S::DIDC + SVF * 16 * jb], NX + 1 - 16 * jb, &sm[S::MJ + SVF * 16], r1,
                          &sm[S::DIDC + SVF * 16 * jb], NX + 1 - 16 * jb, SVF, SVF, dimvf, lane, a0, a1);
      auto put = [&](int r, int c, double v) { sm[S::MJD + r + SVF * c] = v; };
      mfmaTileStore(a0, 0, 16 * jb, dimvf, NX + 1, lane, put);
      mfmaTileStore(a1, 16, 16 * jb, dimvf, NX + 1, lane, put);
    }
  }
  STAMP(11);
  const double* const mjidc = &sm[S::MJD + SVF * NX];      // MJtJinv [ID; C]: column NX of the product above
  // Qafu_full = Qaf MJ.leftCols(nv) needs only MJ: it runs next to the product above (QAFU aliases M, J, dead since the last barrier).
  // Acceleration rows (Qaa diagonal) and contact rows (Qff dense) as loops of their own: no divergence, constant trip counts.
#pragma unroll
  for (int t = 0; t < (NV * NV + nt - 1) / nt; ++t) {
    const int e = tid + nt * t;
    if (e < NV * NV) { const int c = e / NV, r = e - c * NV; sm[S::QAFU + r + SVF * c] = hu * sm[S::QAA + r] * sm[S::MJ + r + SVF * c]; }
  }
  for (int e = tid; e < dimf * NV; e += nt) {
    const int c = e / dimfd, r = e - c * dimfd;
    sm[S::QAFU + NV + r + SVF * c] = hu * dotAny(&sm[S::QFF + r], SF, &sm[S::MJ + NV + SVF * c], 1, dimf);
  }
  STAMP(12);
  blockLdsBarrier();
  STAMP(13);
  // Qafqv = -Qaf MJD (QAFQV aliases dIDC, which the product above was still reading)
#pragma unroll
  for (int t = 0; t < (NV * NX + nt - 1) / nt; ++t) {
    const int e = tid + nt * t;
    if (e < NV * NX) { const int c = e / NV, r = e - c * NV; sm[S::QAFQV + r + SVF * c] = -sm[S::QAA + r] * sm[S::MJD + r + SVF * c]; }
  }
  for (int e = tid; e < dimf * NX; e += nt) {
    const int c = e / dimfd, r = e - c * dimfd;
    sm[S::QAFQV + NV + r + SVF * c] = -dotAny(&sm[S::QFF + r], SF, &sm[S::MJD + NV + SVF * c], 1, dimf);
  }
  if (tid < dimvf) {
    const int r = tid;
    double val;
    if (r < NV) val = sm[S::LA + r] - sm[S::QAA + r] * mjidc[r];
    else { double acc = 0.0; for (int p = 0; p < dimf; ++p) acc += sm[S::QFF + (r - NV) + SF * p] * mjidc[NV + p]; val = -sm[S::LF + r - NV] - acc; }
    sm[S::LAF + r] = val;
  }
  blockLdsBarrier();
  STAMP(8);
  // ---- H. condensed Hessian / gradients / dynamics (contact_dynamics.hxx:129-157) ----
  // Qxx = (cost + IPM terms) - MJD^T Qafqv ; Qxu_full = -MJD^T Qafu_full ; Quu_full = diag + MJ.topRows^T Qafu_full.
  // The products go straight to the kkt / exp records.
  {
    // One job list for the three products.  Qxx is symmetric and its consumers (S3, K9b) read the block triangle on and above the
    // diagonal: 6 tiles of Qxx, 6 of Qxu_full, 2 of Quu_full, taken two at a time (neighbours in the list share an operand) by
    // the four wavefronts.  Job j:  0..5 Qxx (0,0) (0,1) (0,2) (1,1) (1,2) (2,2);  6..11 Qxu_full (ib, jb) = (j - 6) / 2, (j - 6) % 2;  12, 13 Quu_full ib
    const int wave = tid >> 6, lane = tid & 63;
    constexpr int NJOB = 14;
    static_assert(NX <= 48 && NV <= 32 && NU <= 16, "tile counts of the job list");
    auto operands = [&](int j, const double*& X, int& xr, const double*& Y, int& yc, int& r0, int& c0) {
      if (j < 6) {
        const int ib = j < 3 ? 0 : (j < 5 ? 1 : 2), jb = j < 3 ? j : (j < 5 ? j - 2 : 2);
        X = &sm[S::MJD + SVF * 16 * ib]; xr = NX - 16 * ib; Y = &sm[S::QAFQV + SVF * 16 * jb]; yc = NX - 16 * jb; r0 = 16 * ib; c0 = 16 * jb;
      } else if (j < 12) {
        const int ib = (j - 6) >> 1, jb = (j - 6) & 1;
        X = &sm[S::MJD + SVF * 16 * ib]; xr = NX - 16 * ib; Y = &sm[S::QAFU + SVF * 16 * jb]; yc = NV - 16 * jb; r0 = 16 * ib; c0 = 16 * jb;
      } else {
        const int ib = j - 12;
        X = &sm[S::MJ + SVF * 16 * ib]; xr = NV - 16 * ib; Y = &sm[S::QAFU + SVF * 6]; yc = NU; r0 = 16 * ib; c0 = 0;      // MJ symmetric
      }
    };
    // The tiles leave branch-free: every LDS operand of the "base" terms is fetched from a clamped (always valid) address up front, the
    // term is selected, and the only conditional is the store itself.  (Round 3 had `if (r < 6 && c < 6) base = sm[..]; else if (r == c) ..`
    // per element: four separate blocks with an LDS wait each, 1.0 us for a pair of Qxx tiles against 0.7 us for their products.)
    auto finish = [&](int j, const mfma_d4& acc, int r0, int c0) {
      const int r = r0 + (lane & 15), g4 = lane >> 4;                   // (mfmaTileStore's map: row = lane & 15, column = g + 4 q)
      if (j < 6) {
        double b6[4];
        const double dg = sm[S::HQD + (r < NX ? r : NX - 1)];           // HQD | HVD are contiguous: entry r of the diagonal
        static_assert(S::HVD == S::HQD + NV, "one diagonal vector");
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = c0 + g4 + 4 * q; b6[q] = sm[S::QB6 + (r < 6 ? r : 5) + 6 * (c < 6 ? c : 5)]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + g4 + 4 * q;
          const double base = (r < 6 && c < 6) ? b6[q] : (r == c ? dg : 0.0);
          if (r <= c && c < NX) kk[L::K_QXX + L::xsym(r, c)] = base - acc[q];      // (the tiles on the diagonal carry both triangles)
        }
      } else if (j < 12) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + g4 + 4 * q;
          double* dst = c < 6 ? ee + (L::E_QXUP + r + NX * c) : kk + (L::K_QXU + r + NX * (c - 6));      // passive columns of Qxu_full | Qxu
          if (r < NX && c < NV) *dst = -acc[q];
        }
      } else {
        double hd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int c = c0 + g4 + 4 * q; hd[q] = sm[S::HUD + (c < NU ? c : NU - 1)]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int c = c0 + g4 + 4 * q;
          double* dst = r < 6 ? ee + (L::E_QUUP + r + 6 * c) : kk + (L::K_QUU + (r - 6) + NU * c);      // Quu_passive_topRight | Quu
          const double v = acc[q] + ((r >= 6 && r - 6 == c) ? hd[q] : 0.0);
          if (r < NV && c < NU) *dst = v;
        }
      }
    };
    // The job numbers of a wavefront are made compile-time constants (a switch over the wavefront, the pair body instantiated per job): as
    // run-time values they cost a chain of selects for the operand pointers and extents and a three-way branch in front of every store
    auto pair = [&](auto jc) {
      constexpr int j = decltype(jc)::value;
      const double *X0, *Y0, *X1, *Y1;
      int xr0, yc0, r00, c00, xr1, yc1, r01, c01;
      operands(j, X0, xr0, Y0, yc0, r00, c00);
      operands(j + 1, X1, xr1, Y1, yc1, r01, c01);
      mfma_d4 a0, a1;
      K5_TILE_PAIR<RVF>(X0, xr0, Y0, yc0, X1, xr1, Y1, yc1, SVF, SVF, dimvf, lane, a0, a1);
      finish(j, a0, r00, c00);
      finish(j + 1, a1, r01, c01);
    };
    static_assert(nt == 256 && NJOB == 14, "four wavefronts: jobs 2 w, 2 w + 1 and 2 w + 8, 2 w + 9");
    switch (__builtin_amdgcn_readfirstlane(wave)) {
      case 0: pair(std::integral_constant<int, 0>{}); pair(std::integral_constant<int, 8>{}); break;
      case 1: pair(std::integral_constant<int, 2>{}); pair(std::integral_constant<int, 10>{}); break;
      case 2: pair(std::integral_constant<int, 4>{}); pair(std::integral_constant<int, 12>{}); break;
      default: pair(std::integral_constant<int, 6>{}); break;
    }
  }
  STAMP(14);
  {
    // lx -= MJD^T laf ; [lu_passive; lu] += MJ.topRows(NV) laf ; Fv -= dt MJIDC -- four lanes per dot product, like C2
    const int part = tid & 3, g = tid >> 2;
    const int q4 = (dimvf + 3) >> 2, k0 = part * q4, k1 = (k0 + q4 < dimvf) ? k0 + q4 : dimvf;
    double acc = 0.0;
    if (g < NX) { const double* col = &sm[S::MJD + SVF * g]; for (int k = k0; k < k1; ++k) acc += col[k] * sm[S::LAF + k]; }
    else if (g < NX + NV) { const double* row = &sm[S::MJ + (g - NX)]; for (int k = k0; k < k1; ++k) acc += row[SVF * k] * sm[S::LAF + k]; }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (part == 0) {
      if (g < NV) sm[S::LQ + g] -= acc;
      else if (g < NX) sm[S::LV + g - NV] -= acc;
      else if (g < NX + NV) {
        const int r = g - NX;
        if (r < 6) sm[S::LUP + r] += hu * acc; else sm[S::LU + r - 6] += hu * acc;
        sm[S::FV + r] -= dt * mjidc[r];
      }
    }
  }
  static_assert(4 * (NX + NV) <= nt, "a quad per row of the vector updates");
  blockLdsBarrier();

  STAMP(9);
  // ---- I. write the kkt and exp records ----
  typedef double wd2 __attribute__((ext_vector_type(2)));      // 16-byte pieces: two consecutive rows of a column
  static_assert(NV % 2 == 0 && SVF % 2 == 0 && S::MJD % 2 == 0 && L::K_FVQ % 2 == 0 && L::K_FVV % 2 == 0 && L::KKT % 2 == 0, "16-byte stores of Fvq / Fvv");
  for (int e = tid; e < NV * NV / 2; e += nt) {
    const int c = e / (NV / 2), r = 2 * (e - c * (NV / 2));
    const wd2 mq = *reinterpret_cast<const wd2*>(&sm[S::MJD + r + SVF * c]), mvv = *reinterpret_cast<const wd2*>(&sm[S::MJD + r + SVF * (NV + c)]);
    const double one = BWD ? -1.0 : 1.0;
    wd2 fq, fv;
    fq.x = -dt * mq.x; fq.y = -dt * mq.y;
    fv.x = -dt * mvv.x + (r == c ? one : 0.0); fv.y = -dt * mvv.y + (r + 1 == c ? one : 0.0);
    *reinterpret_cast<wd2*>(&kk[L::K_FVQ + r + NV * c]) = fq;
    *reinterpret_cast<wd2*>(&kk[L::K_FVV + r + NV * c]) = fv;
  }
  for (int e = tid; e < NV * NU; e += nt) { const int c = e / NV, r = e - c * NV; kk[L::K_FVU + e] = hu * dt * sm[S::MJ + r + SVF * (6 + c)]; }
  if (tid < NV) {
    kk[L::K_LX + tid] = sm[S::LQ + tid]; kk[L::K_LX + NV + tid] = sm[S::LV + tid];
    kk[L::K_FX + tid] = sm[S::FQ + tid]; kk[L::K_FX + NV + tid] = sm[S::FV + tid];
  }
  if (tid < NU) kk[L::K_LU + tid] = sm[S::LU + tid];
  // MJtJinv: the lower triangle.  Rows p and RVF - 1 - p together have RVF + 1 entries: thread t takes entry t mod (RVF + 1) of pair
  // t / (RVF + 1) -- a division by a constant and a select (round 3 took the row of element t from a square root: twice the instructions),
  // consecutive threads on consecutive addresses within a row
  for (int t = tid; t < RVF * (RVF + 1) / 2; t += nt) {
    const int pr = t / (RVF + 1), idx = t - pr * (RVF + 1);
    const bool lo = idx <= pr;
    const int r = lo ? pr : RVF - 1 - pr, c = lo ? idx : idx - pr - 1;
    ee[L::E_MJ + r * (r + 1) / 2 + c] = sm[S::MJ + r + SVF * c];
  }
  static_assert(RVF % 2 == 0 && NVF % 2 == 0 && L::E_MJD % 2 == 0 && L::EXP % 2 == 0, "16-byte stores of MJtJinv_dIDCdqv");
  for (int e = tid; e < (RVF / 2) * NX; e += nt) {
    const int c = e / (RVF / 2), r = 2 * (e - c * (RVF / 2));
    *reinterpret_cast<wd2*>(&ee[L::E_MJD + r + NVF * c]) = *reinterpret_cast<const wd2*>(&sm[S::MJD + r + SVF * c]);
  }
  if (tid < NV) ee[L::E_QAA + tid] = sm[S::QAA + tid];
  for (int e = tid; e < SF * SF; e += nt) { const int c = e / SF, r = e - c * SF; ee[L::E_QFF + r + NF * c] = sm[S::QFF + e]; }
  if (tid < RVF) { ee[L::E_MJIDC + tid] = mjidc[tid]; ee[L::E_LAF + tid] = sm[S::LAF + tid]; }
  if (tid < 6) ee[L::E_LUP + tid] = sm[S::LUP + tid];
  // ---- ContactDynamics::condenseSwitchingConstraint (contact_dynamics.hxx:193-199) ----
  if (sw_dimi > 0) {
    const int dimi = sw_dimi;
    double* __restrict__ W = B.swc + rec * L::SWC;
    for (int e = tid; e < dimi * NX; e += nt) {           // Phix -= Phia MJtJinv_dIDCdqv.topRows(nv)
      const int c = e / dimi, j = e - c * dimi;
      double acc = 0.0;
      for (int m2 = 0; m2 < NV; ++m2) acc += W[L::W_PHIA + j + NF * m2] * sm[S::MJD + m2 + SVF * c];
      W[L::W_PHIX + j + NF * c] -= acc;
    }
    for (int e = tid; e < dimi * NU; e += nt) {           // Phiu = Phia MJtJinv.block(0, 6, nv, nu)
      const int c = e / dimi, j = e - c * dimi;
      double acc = 0.0;
      for (int m2 = 0; m2 < NV; ++m2) acc += W[L::W_PHIA + j + NF * m2] * sm[S::MJ + m2 + SVF * (6 + c)];
      W[L::W_PHIU + j + NF * c] = acc;
    }
    if (tid < dimi) {                                     // P -= Phia MJtJinv_IDC.head(nv)
      double acc = 0.0;
      for (int m2 = 0; m2 < NV; ++m2) acc += W[L::W_PHIA + tid + NF * m2] * mjidc[m2];
      W[L::W_P + tid] -= acc;
    }
  }
  STAMP(10);
  if (tid == 0 && !s_ok && B.status[b] == 0) B.status[b] = 1 + pos;
  };      // stage
  stage((long)blockIdx.x);
#undef STAMP
#undef STAMPW
}

// (The Lie-group terms of the forward-Euler stages are tasks of ocp_nominal_kernel since round 3.)

// Lie-group terms of the backward-Euler stage (state_equation.hxx:96-170): same record, same places as the forward-Euler tasks of ocp_nominal_kernel
//   task 0: qdiff = q (-) q_ref and Jq
//   task 1: FQQ = dSubtractdConfigurationPlus(q, q_next)                      (coupling with the next stage's lmd)
//   task 2: FQ6 = (q_prev (-) q).head(6), FQQP = dSubtractdConfigurationMinus(q_prev, q),
//           FQQI = dSubtractdConfigurationPlus(q_prev, q)^-1
template <typename D>
__global__ __launch_bounds__(64) void parnmpc_lie_kernel(OcpBuffers B, const double* __restrict__ q0) {
  using L = OcpLayout<D>;
  constexpr int NQ = D::NQ;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long unit = (long)blockIdx.x * 64 + threadIdx.x;
  if (unit >= (long)P->batch * (M - 1)) return;
  const long b = unit / (M - 1);
  const int pos = (int)(unit - b * (M - 1));
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const int task = blockIdx.y;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ q = s + L::S_Q;
  double* __restrict__ zz = B.lie + rec * L::LIE;
  double R[9], p[3], Ja[36], Jb[36], d6[6];
  if (task == 0) {
    lieRelative(B.q_ref + (long)pos * NQ, q, R, p);
    lieLog6(R, p, d6);
    lieJlog6(R, p, Ja);
    for (int k = 0; k < 36; ++k) zz[L::Z_JQ + k] = Ja[k];
    for (int k = 0; k < 6; ++k) zz[L::Z_QDIFF + k] = d6[k];
  } else if (task == 1) {
    lieRelative(B.sol + (b * B.NS + nd->next) * L::SOL + L::S_Q, q, R, p);      // q (-) q_next; ARG of q: Jlog6
    lieJlog6(R, p, Ja);
    for (int k = 0; k < 36; ++k) zz[L::Z_FQQ + k] = Ja[k];
  } else {
    const double* __restrict__ q_prev = (nd->prev < 0) ? (q0 + b * NQ) : (B.sol + (b * B.NS + nd->prev) * L::SOL + L::S_Q);
    lieRelative(q, q_prev, R, p);                                                 // q_prev (-) q
    lieLog6(R, p, d6);
    for (int k = 0; k < 6; ++k) zz[L::Z_FQ6 + k] = d6[k];
    lieJlog6(R, p, Ja);                                                           // d/d q_prev  (dSubtract_dPlus)
    lieDDiffArg0(R, p, Ja, Jb);                                                   // d/d q       (dSubtract_dMinus)
    for (int k = 0; k < 36; ++k) zz[L::Z_FQQP + k] = Jb[k];
    lieBlockInverse(Ja, Jb);
    for (int k = 0; k < 36; ++k) zz[L::Z_FQQI + k] = Jb[k];
  }
}

// M = chain length; dimf >= 0: every non-terminal stage of the chain is a regular stage with `dimf` active contact rows
// (enables the compile-time instantiation), -1: mixed chain.
template <typename D>
// part: 0 = everything, 1 = the nominal sweeps and the rows of the external terms only, 2 = the K5 launches only (bench.py brackets them apart)
static void launchCondense(const OcpBuffers& B, long batch, int M, int dimf, const double* q0, hipStream_t st, bool residual, int part = 0) {
  const size_t smem = CondenseSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, true, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  const unsigned blocks = (unsigned)(batch * M);
  if (part != 2) {
    OcpLaunch<D>::nominal(B, batch, M, st, q0);      // (+ the Lie-group tasks)
    OcpLaunch<D>::extRows(B, batch, M, residual, st);
  }
  if (part == 1) return;
  if (residual) hipLaunchKernelGGL((ocp_condense_kernel<D, true, -1>), dim3(blocks), dim3(256), smem, st, B, q0);
  else if (dimf == D::NF) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF, false, false, true>), dim3(blocks), dim3(256), smem, st, B, q0); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF>), dim3(blocks), dim3(256), smem, st, B, q0); }
  else { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1, false, false, true>), dim3(blocks), dim3(256), smem, st, B, q0); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1>), dim3(blocks), dim3(256), smem, st, B, q0); }
  if (!residual) OcpLaunch<D>::extHessian(B, batch, M, st);
}

template <typename D>
void OcpLaunch<D>::condense(const OcpBuffers& B, long batch, int M, int dimf, const double* q0, hipStream_t st, int part) {
  launchCondense<D>(B, batch, M, dimf, q0, st, false, part);
}
// Chains with discrete events: the stages are grouped on the host by what the kernel can fold at compile time --
// class 0: all feet in contact, class 1: half of them (trot, pace, bound), both without impulse / switching constraint;
// class 2: everything else (impulse stages, stages carrying a switching constraint, other contact counts, the terminal
// stage) on the general instantiation.  B.cond_pos holds the chain positions class by class, n[c] their counts.
template <typename D>
void OcpLaunch<D>::condenseMixed(const OcpBuffers& B, long batch, int M, const int n[5], const double* q0, hipStream_t st, int part, hipStream_t st_imp) {
  const size_t smem = CondenseSmem<D>::TOTAL * sizeof(double), smem_half = CondenseSmem<D, D::NF / 2>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF / 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF / 2, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF / 2, false, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, 0, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, D::NF / 2, false, false, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_half);
    configured = true;
  }
  const unsigned blocks = (unsigned)(batch * M);
  const double* none = nullptr;
  if (part != 2) {
    OcpLaunch<D>::nominal(B, batch, M, st, q0, st_imp);      // (+ the Lie-group tasks)
    OcpLaunch<D>::extRows(B, batch, M, false, st);
  }
  if (part == 1) return;
  if (part == 5) { OcpLaunch<D>::extHessian(B, batch, M, st); return; }
  // the largest class first; the launches are independent (every stage writes its own records)
  const bool big = part != 4, rest = part != 3;
  if (big && n[1] > 0) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF / 2, false, false, true>), dim3((unsigned)(batch * n[1])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0], n[1]); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF / 2>), dim3((unsigned)(batch * n[1])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0], n[1]); }
  if (rest && n[0] > 0) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF, false, false, true>), dim3((unsigned)(batch * n[0])), dim3(256), smem, st, B, q0, none, B.cond_pos, n[0]); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF>), dim3((unsigned)(batch * n[0])), dim3(256), smem, st, B, q0, none, B.cond_pos, n[0]); }
  // class 4: flight stages (no contact rows at all; the narrow LDS layout, every contact loop folded away)
  if (rest && n[4] > 0) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, 0, false, false, true>), dim3((unsigned)(batch * n[4])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0] + n[1] + n[2] + n[3], n[4]); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, 0>), dim3((unsigned)(batch * n[4])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0] + n[1] + n[2] + n[3], n[4]); }
  // class 3: event stages (impulse / switching constraint) with half of the feet
  if (rest && n[3] > 0) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF / 2, false, false, true, true>), dim3((unsigned)(batch * n[3])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0] + n[1] + n[2], n[3]); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, D::NF / 2, false, false, false, true>), dim3((unsigned)(batch * n[3])), dim3(256), smem_half, st, B, q0, none, B.cond_pos + n[0] + n[1] + n[2], n[3]); }
  if (rest && n[2] > 0) { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1, false, false, true>), dim3((unsigned)(batch * n[2])), dim3(256), smem, st, B, q0, none, B.cond_pos + n[0] + n[1], n[2]); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1>), dim3((unsigned)(batch * n[2])), dim3(256), smem, st, B, q0, none, B.cond_pos + n[0] + n[1], n[2]); }
  if (part == 0 || part == 2) OcpLaunch<D>::extHessian(B, batch, M, st);
}
// Line search: cost and l1 violation of every stage of the chain for the iterate Btry.sol points at (Btry.nodes: the chain with the
// reference's pairing of the successors).  The caller has run the impulse RNEA and the switching kernel on Btry.
template <typename D>
void OcpLaunch<D>::merit(const OcpBuffers& Btry, long batch, int M, const double* q0, hipStream_t st) {
  const size_t smem = CondenseSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, true, -1, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  const unsigned blocks = (unsigned)(batch * M);
  OcpLaunch<D>::nominal(Btry, batch, M, st, q0);
  OcpLaunch<D>::extRows(Btry, batch, M, true, st);
  hipLaunchKernelGGL((ocp_condense_kernel<D, true, -1, false, true>), dim3(blocks), dim3(256), smem, st, Btry, q0);
}
// The same for ParNMPC (backward-Euler stages; event-free horizons): Split / TerminalParNMPC::stageCost and constraintViolation
// (split_parnmpc.hxx:269-311, terminal_parnmpc.hxx:188-229) -- the terminal cost is NOT part of the reference's merit.
template <typename D>
void OcpLaunch<D>::meritBackwardEuler(const OcpBuffers& Btry, long batch, int M, const double* q0, const double* v0, hipStream_t st) {
  const size_t smem = CondenseSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, true, -1, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  const unsigned stages = (unsigned)(batch * (M - 1));
  hipLaunchKernelGGL((parnmpc_lie_kernel<D>), dim3((stages + 63) / 64, 3), dim3(64), 0, st, Btry, q0);
  OcpLaunch<D>::nominal(Btry, batch, M, st);
  OcpLaunch<D>::extRows(Btry, batch, M, true, st);
  hipLaunchKernelGGL((ocp_condense_kernel<D, true, -1, true, true>), dim3((unsigned)(batch * M)), dim3(256), smem, st, Btry, q0, v0);
}
template <typename D>
void OcpLaunch<D>::residual(const OcpBuffers& B, long batch, int M, const double* q0, hipStream_t st) {
  launchCondense<D>(B, batch, M, -1, q0, st, true);
}
// ParNMPC: backward-Euler stages 0..M-2 (the chain's last entry is a placeholder)
template <typename D>
void OcpLaunch<D>::condenseBackwardEuler(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, bool residual,
                                         hipStream_t st) {
  const size_t smem = CondenseSmem<D>::TOTAL * sizeof(double);
  static bool configured = false;
  if (!configured) {
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, false, -1, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute((const void*)ocp_condense_kernel<D, true, -1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    configured = true;
  }
  const unsigned stages = (unsigned)(batch * (M - 1));
  hipLaunchKernelGGL((parnmpc_lie_kernel<D>), dim3((stages + 63) / 64, 3), dim3(64), 0, st, B, q0);
  OcpLaunch<D>::nominal(B, batch, M, st);
  OcpLaunch<D>::extRows(B, batch, M, residual, st);
  if (residual) hipLaunchKernelGGL((ocp_condense_kernel<D, true, -1, true>), dim3((unsigned)(batch * M)), dim3(256), smem, st, B, q0, v0);
  else { if (B.leg_axes_xyy) hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1, true, false, true>), dim3((unsigned)(batch * M)), dim3(256), smem, st, B, q0, v0); else hipLaunchKernelGGL((ocp_condense_kernel<D, false, -1, true>), dim3((unsigned)(batch * M)), dim3(256), smem, st, B, q0, v0); }
  if (!residual) OcpLaunch<D>::extHessian(B, batch, M, st);
}

template void OcpLaunch<LeggedDims<4, 3>>::condense(const OcpBuffers&, long, int, int, const double*, hipStream_t, int);
template void OcpLaunch<LeggedDims<4, 3>>::condenseMixed(const OcpBuffers&, long, int, const int*, const double*, hipStream_t, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::merit(const OcpBuffers&, long, int, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::meritBackwardEuler(const OcpBuffers&, long, int, const double*, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::residual(const OcpBuffers&, long, int, const double*, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::condenseBackwardEuler(const OcpBuffers&, long, int, const double*, const double*, bool, hipStream_t);

}  // namespace idocp_dev
