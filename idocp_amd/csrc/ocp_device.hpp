// Device data layout + parameter block of the contact-capable (OCPSolver) path.
//
// Same conventions as unocp_device.hpp: horizon-batched arrays of fixed-stride
// FP64 records, strides rounded to 16 doubles, column-major matrices.  Robot
// shape is a compile-time trait: a free-flyer base with NL serial legs of LJ
// revolute joints and one point contact at the tip of each leg (ANYmal: 4 x 3).
// Blocks that depend on the number of active contacts use the leading dimension
// of the maximum (NVF = NV + 3 NC) and keep the active rows packed at the top,
// like the reference's *_full_ buffers (contact_dynamics_data.hxx:8-29).
#ifndef IDOCP_OCP_DEVICE_HPP_
#define IDOCP_OCP_DEVICE_HPP_

#include "unocp_device.hpp"

namespace idocp_dev {

template <int NL_, int LJ_>
struct LeggedDims {
  static constexpr int NL = NL_, LJ = LJ_;
  static constexpr int NJ = 1 + NL * LJ, NV = 6 + NL * LJ, NQ = NV + 1, NU = NL * LJ, NC = NL, NF = 3 * NC;
  static constexpr int NX = 2 * NV, NVF = NV + NF;
};

template <typename D>
struct OcpLayout {
  static constexpr int NV = D::NV, NQ = D::NQ, NU = D::NU, NF = D::NF, NX = D::NX, NVF = D::NVF, NC = D::NC;
  // solution (split_solution.hxx:10-31): lmd gmm q v a u beta f mu nu_passive xi  (a = dv on impulse stages)
  static constexpr int S_LMD = 0, S_GMM = NV, S_Q = 2 * NV, S_V = S_Q + NQ, S_A = S_V + NV, S_U = S_A + NV, S_BETA = S_U + NU,
                       S_F = S_BETA + NV, S_MU = S_F + NF, S_NUP = S_MU + NF, S_XI = S_NUP + 6;
  static constexpr int SOL = roundUp16(S_XI + NF);
  // direction (split_direction.hxx:8-23)
  static constexpr int D_LMD = 0, D_GMM = NV, D_Q = 2 * NV, D_V = 3 * NV, D_A = 4 * NV, D_U = 5 * NV, D_BETA = D_U + NU,
                       D_F = D_BETA + NV, D_MU = D_F + NF, D_NUP = D_MU + NF, D_XI = D_NUP + 6,
                       D_T = D_XI + NF, D_W = D_T + NVF;       // scratch of the expansion: MJD dx and MJ[:, u] du (K6 -> K7)
  static constexpr int DIR = roundUp16(D_W + NVF);
  // IPM rows: 6 joint-limit components x NU (position / velocity / torque, lower then upper), 5 friction-cone rows per contact, then the
  // joint-acceleration limits (components 8 = lower, 9 = upper; component 6 is the cone, 7 is not used: odd components are upper bounds)
  static constexpr int C_FRIC = 6 * NU, C_ACC = C_FRIC + 5 * NC, C_CD = C_ACC + 2 * NU, NCON = C_CD + NC;      // C_CD: ContactDistance, one row per contact
  static constexpr int CON = roundUp16(NCON);
  // linearisation record written by the tangent-RNEA kernel: [dID;dC]/d(q,v) (NVF x NX, ld NVF),
  // dID/da = M (NV x NV), dC/da = J (NF x NV, ld NF), [ID; C]
  static constexpr int L_DIDC = 0, L_M = NVF * NX, L_J = L_M + NV * NV, L_IDC = L_J + NF * NV;
  static constexpr int LIN = roundUp16(L_IDC + NVF);
  // condensed LQR stage for the Riccati sweep
  // Qxx is symmetric and travels as its UPPER triangle, column by column: entry (r, c), r <= c, at K_QXX + c (c + 1) / 2 + r
  // (round 3: 5 of the record's 23 kB were the mirror image of the rest -- every reader used the upper triangle already)
  static constexpr int X_TRI = (NX * (NX + 1) / 2 + 1) / 2 * 2;
  __host__ __device__ static constexpr int xsym(int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; }
  static constexpr int K_QXX = 0, K_QXU = X_TRI, K_QUU = K_QXU + NX * NU, K_FQQ = K_QUU + NU * NU, K_FQV = K_FQQ + 36,
                       K_FVQ = K_FQV + 36, K_FVV = K_FVQ + NV * NV, K_FVU = K_FVV + NV * NV, K_LX = K_FVU + NV * NU,
                       K_LU = K_LX + NX, K_FX = K_LU + NU;
  static constexpr int KKT = roundUp16(K_FX + NX);
  // expansion cache (ContactDynamicsData members + passive blocks + Fqq_prev_inv)
  // (Qafqv = -[Qaa; Qff] MJD and Qafu = [Qaa; Qff] MJ[:, u] are NOT stored: K6 leaves t = MJD dx and w = MJ[:, u] du in the
  //  dir record and K7 applies diag(Qaa) / Qff to them -- 11.5 kB less HBM traffic per stage, once written and once read)
  // MJtJinv is symmetric: its lower triangle, row by row -- element (r, c), r >= c, at E_MJ + r (r + 1) / 2 + c (3.7 instead of 7.2 kB
  // per stage, written once by K5 and read by K6 and K7)
  static constexpr int MJ_TRI = (NVF * (NVF + 1) / 2 + 1) / 2 * 2;
  static constexpr int E_MJ = 0, E_MJD = MJ_TRI, E_QAA = E_MJD + NVF * NX, E_QFF = E_QAA + NV,
                       E_MJIDC = E_QFF + NF * NF, E_LAF = E_MJIDC + NVF, E_LUP = E_LAF + NVF, E_QUUP = E_LUP + 6,
                       E_QXUP = E_QUUP + 6 * NU, E_FQQPI = E_QXUP + NX * 6;
  static constexpr int EXP = roundUp16(E_FQQPI + 36);
  // Riccati factorisation of a stage: Pqq and Pvv are symmetric and travel as their UPPER triangle, column by column -- entry (r, c),
  // r <= c, at c (c + 1) / 2 + r (round 3: 2.4 of the record's 8.1 kB were the mirror image of the rest; S3 writes it, K6 reads it)
  static constexpr int P_TRI = (NV * (NV + 1) / 2 + 1) / 2 * 2;
  __host__ __device__ static constexpr int psym(int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; }
  static constexpr int R_PQQ = 0, R_PQV = P_TRI, R_PVV = R_PQV + NV * NV, R_SQ = R_PVV + P_TRI, R_SV = R_SQ + NV;
  static constexpr int RIC = roundUp16(R_SV + NV);
  // Lie-group terms of the floating base, produced by the small pre-kernel (6x6 blocks column-major)
  static constexpr int Z_JQ = 0, Z_QDIFF = 36, Z_FQQ = 44, Z_FQ6 = 80, Z_FQQI = 88, Z_FQQP = 124, Z_FQQPI = 160;
  static constexpr int LIE = roundUp16(196);
  // nominal rigid-body record written by ocp_nominal_kernel (one lane per stage and leg) and copied into the LDS scratch of the
  // condensation kernel (dev_rnea_tangent.hpp, RneaScratch: the records keep their dynamic fields first, in this order):
  //   per leg joint (NJL = NL LJ of them) NJ_DYN = 42: R (9, row-major), wc, vc, bwc, blc, zc, vJ, w, hl, hn, Fl, Fn
  //   per foot 18: R_world,foot Rc (9), fv, fw, pose part of the Baumgarte residual
  //   base 16: z, v, w, hl, hn (+ 1 pad);  (NL + 1) x 6 nominal base force: own, then per leg;  nominal [ID; C] (NVF)
  static constexpr int NJL = NV - 6, NJ_DYN = 42, NF_DYN = 18, NB_DYN = 16;
  static constexpr int O_JOINT = 0, O_FEET = NJL * NJ_DYN, O_BASE = O_FEET + NC * NF_DYN, O_BN = O_BASE + NB_DYN,
                       O_IDC = O_BN + (NC + 1) * 6;
  static constexpr int NOM = roundUp16(O_IDC + NVF);
  static_assert(NVF % 2 == 0, "the nominal record is copied in 16-byte pieces");
  // ext record (ocp_ext_kernel.hip; allocated only when a term of it is in use): the ContactDistance rows J_c (NC x NV, row-major), the
  // heights z_c, the gradient term to be added to lq, the Hessian weights, the stage's share of the KKT error / line-search violation
  // then the task-space cost (TaskSpace3DCost / TaskSpace6DCost): the rows JJ (6 x NV, row-major), their weights (time step included), the cost
  static constexpr int X_CDJ = 0, X_Z = NC * NV, X_LQ = X_Z + NC, X_W = X_LQ + NV, X_ERR = X_W + NC, X_VIOL = X_ERR + 1;
  // (NT component slots: idocp_cost_t's task_* block and its task_extra components; unused slots carry zero weights)
  static constexpr int NT = 1 + IDOCP_MAX_EXTRA_TASKS;
  static constexpr int X_TJ = X_VIOL + 1, X_TW = X_TJ + NT * 6 * NV, X_COST = X_TW + NT * 6;
  static constexpr int EXT = roundUp16(X_COST + 1);
  static constexpr int G_K = 0, G_k = NU * NX;
  static constexpr int GAIN = roundUp16(G_k + NU);
  // switching-constraint record of a stage two steps ahead of an impulse (SplitStateConstraintJacobian +
  // SplitConstrainedRiccatiFactorization): P (NF), Phix (NF x NX, ld NF), Phia (NF x NV), Phiu (NF x NU),
  // then the multiplier policy dxi = M dx + m written by the Riccati sweep: M (NF x NX), m (NF)
  static constexpr int W_P = 0, W_PHIX = NF, W_PHIA = W_PHIX + NF * NX, W_PHIU = W_PHIA + NF * NV, W_M = W_PHIU + NF * NU,
                       W_m = W_M + NF * NX;
  static constexpr int SWC = roundUp16(W_m + NF);
  // ---- ParNMPC (backward correction) ----
  // coarse / corrected iterate of a stage (SplitBackwardCorrection's s_new): lmd gmm u q v
  // (event stages: N_U holds f on an impulse stage; N_XI the switching multiplier xi of an aux stage / mu of an impulse stage)
  static constexpr int N_LMD = 0, N_GMM = NV, N_U = 2 * NV, N_Q = N_U + NU, N_V = N_Q + NQ, N_XI = N_V + NV;
  static constexpr int SNEW = roundUp16(N_XI + NF);
  // the columns of the stage's KKT-matrix inverse the correction sweeps need (split_backward_correction.hxx:84-140),
  // column-major with leading dimension NK = 2 NX + NU (rows: lmd gmm | u q v):
  //   C0 = KKT_inv[:, 0 : NX]  (auxMat, forward corrections),  C1 = KKT_inv[:, NK - NX : NK]  (backward corrections)
  static constexpr int NK = 2 * NX + NU, I_C0 = 0, I_C1 = NK * NX;
  // event stages (aux with the switching-constraint rows, impulse): rows  lmd gmm | xi or mu (ni) | u or f (nw) | q v,
  // nK = 2 NX + ni + nw <= NKG; same two column blocks with leading dimension NKG, C1 at I_C1G
  static constexpr int NKG = 2 * NX + NU + NF, I_C1G = NKG * NX;
  static constexpr int KINV = roundUp16(2 * NKG * NX);
  static constexpr int AUX = roundUp16(NX * NX);
  static constexpr int XRES = roundUp16(NX);
};

// One stage of the CHAIN (time order): stage, [impulse, aux | lift], stage, ..., terminal.  Every stage owns a fixed
// storage SLOT (grid stage i -> i, impulse k -> N+1+k, aux k -> N+1+E+k, lift k -> N+1+2E+k, like the separate arrays
// of the reference's hybrid_container.hpp:60-168); the chain is rebuilt by the host-side discretiser
// (OCPDiscretizer, ocp_discretizer.hxx:65-374) and says who the neighbours are.
struct OcpNode {
  int slot, next, prev;     // prev = -1: the predecessor is the initial state
  int kind;                 // 0 stage, 1 impulse, 2 aux, 3 lift, 4 terminal
  int level;                // time step for the constraint gating (constraints_data.hpp:18-42): stage index, 0 aux / lift
  int has_u;                // 0 on impulse stages (no torque variables)
  int dimf, active[IDOCP_MAX_CONTACTS], row_of[IDOCP_MAX_CONTACTS];   // contact (or impulse) status of this stage
  double dt;                // scaling of cost / constraint / dynamics multipliers: the time step, 1 on impulse stages
  double dtq;               // q+ = q (+) dtq v: the time step, 0 on impulse stages
  double vref_on;           // 1, or 0 where a time-varying cost switches its velocity reference off (stage time outside its window)
  double contact_point[IDOCP_MAX_CONTACTS][3];
  int sw_dimi, sw_active[IDOCP_MAX_CONTACTS], sw_row[IDOCP_MAX_CONTACTS];     // switching constraint carried by this stage
  double sw_dt1, sw_dt2, sw_point[IDOCP_MAX_CONTACTS][3];
};

// ParNMPC: shape of a stage's KKT matrix (SplitBackwardCorrection / ImpulseSplitBackwardCorrection::dimKKT_)
struct ParnmpcShape {
  bool general, impulse;   // general = impulse stage, or aux stage with switching-constraint rows
  int ni, nw, nk, ld, c1;  // extra constraint rows, size of the u / f block, dimKKT, leading dimension and offset of C1 in the kinv record
};
template <typename L>
__host__ __device__ inline ParnmpcShape parnmpcShape(const OcpNode& nd) {
  ParnmpcShape s;
  s.impulse = nd.kind == 1;
  const bool aux = nd.kind == 2 && nd.sw_dimi > 0;
  s.general = s.impulse || aux;
  s.ni = s.impulse ? nd.dimf : (aux ? nd.sw_dimi : 0);
  s.nw = s.impulse ? nd.dimf : L::NU;
  s.nk = 2 * L::NX + s.ni + s.nw;
  s.ld = s.general ? L::NKG : L::NK;
  s.c1 = s.general ? L::I_C1G : L::I_C1;
  return s;
}

// The friction-cone component of ONE contact with force f: value g_r(f) (<= 0 inside the cone) and gradient J_r(f) of row r.
//   kind 0  LinearizedFrictionCone, 5 rows:  g = Jc f, Jc = [0 0 -1; 1 0 -m; -1 0 -m; 0 1 -m; 0 -1 -m], m = mu / sqrt(2)
//           (linearized_friction_cone.cpp:25-29, linearized_friction_cone.hpp:72-84)
//   kind 1  FrictionCone, 2 rows:  g0 = -fz,  g1 = fx^2 + fy^2 - mu^2 fz^2  (friction_cone.hpp:70-80); J1 = (2 fx, 2 fy, -2 mu^2 fz) is the
//           reference's data.r[i] (friction_cone.cpp:100-118) and the Hessian term is the Gauss-Newton one, J^T diag(dual / slack) J
//           (:121-146): every formula of the component is the same for the two kinds.
// The IPM records keep five slots per contact; rows >= coneRows(kind) do not exist (never read, never updated).
__host__ __device__ inline int coneRows(int kind) { return kind == 1 ? 2 : 5; }
__host__ __device__ inline double coneRow(int kind, double mu, int r, const double* f, double* J) {
  if (kind == 1) {
    if (r == 0) { J[0] = 0.0; J[1] = 0.0; J[2] = -1.0; return -f[2]; }
    J[0] = 2.0 * f[0]; J[1] = 2.0 * f[1]; J[2] = -2.0 * mu * mu * f[2];
    return f[0] * f[0] + f[1] * f[1] - mu * mu * f[2] * f[2];
  }
  const double m2 = mu * 0.70710678118654752440;
  J[0] = (r == 1) ? 1.0 : (r == 2 ? -1.0 : 0.0);
  J[1] = (r == 3) ? 1.0 : (r == 4 ? -1.0 : 0.0);
  J[2] = (r == 0) ? -1.0 : -m2;
  return J[0] * f[0] + J[1] * f[1] + J[2] * f[2];
}

struct OcpProblem;
// first IPM row of a joint-limit component (0 .. 5, 8, 9)
template <typename L>
__host__ __device__ inline int ipmCompRow(int comp) { return comp < 6 ? comp * L::NU : L::C_ACC + (comp - 8) * L::NU; }

struct OcpProblem {
  int N, batch;            // N = grid intervals (N_ideal)
  int M, NS, E;            // chain length of the current discretisation; storage slots per instance; max events
  int stage_offset;             // ParNMPC horizon sharding: global index of this shard's first stage
  int has_terminal, has_prev;   // ParNMPC horizon sharding: this shard ends with the terminal stage / has a left neighbour
  int backward_euler;      // 1: ParNMPC stages (stage i lives at t + (i+1) dt, constraint level i + 1, own dgmm in the dual expansion)
  double T, dt;            // dt = T / N: Baumgarte time step and the time step of the regular stages
  double v_ref[IDOCP_MAX_NV], u_ref[IDOCP_MAX_NV];
  double q_weight[IDOCP_MAX_NV], v_weight[IDOCP_MAX_NV], a_weight[IDOCP_MAX_NV], u_weight[IDOCP_MAX_NV];
  double qf_weight[IDOCP_MAX_NV], vf_weight[IDOCP_MAX_NV];
  double f_weight[IDOCP_MAX_CONTACTS][3], f_ref[IDOCP_MAX_CONTACTS][3];
  double q_min[IDOCP_MAX_NV], q_max[IDOCP_MAX_NV], v_max[IDOCP_MAX_NV], u_max[IDOCP_MAX_NV];
  double qi_weight[IDOCP_MAX_NV], vi_weight[IDOCP_MAX_NV], dvi_weight[IDOCP_MAX_NV];         // impulse stages
  double fi_weight[IDOCP_MAX_CONTACTS][3], fi_ref[IDOCP_MAX_CONTACTS][3];
  int use_q_limits, use_v_limits, use_u_limits, use_friction_cone, use_impulse_friction_cone;
  int use_contact_distance;             // ContactDistance rows (ext record)
  // TaskSpace3DCost / TaskSpace6DCost on one frame (ext record): parent joint (0 = the floating base), placement in it, weights of the
  // stages / the terminal stage / the impulse stages, constant reference (rotation row-major, position)
  int task_dim, task_joint;
  double task_R[9], task_p[3], task_weight[6], task_weightf[6], task_weighti[6], task_ref[12];
  int task_n;                // number of task-space components: 0, or 1 + idocp_cost_t::task_extra_count
  struct TaskExtra { int dim, joint; double R[9], p[3], weight[6], weightf[6], weighti[6], ref[12]; } task_extra[IDOCP_MAX_EXTRA_TASKS];
  int use_a_lower, use_a_upper;         // JointAccelerationLowerLimit / UpperLimit (acceleration level: every stage with torques)
  double a_min[IDOCP_MAX_NV], a_max[IDOCP_MAX_NV];
  int cone_kind, impulse_cone_kind;     // 0: Linearized(Impulse)FrictionCone (5 rows per contact), 1: (Impulse)FrictionCone (2 rows); coneRow below
  double mu, barrier, fraction_rate;
  double contact_R[IDOCP_MAX_CONTACTS][9], contact_p[IDOCP_MAX_CONTACTS][3];   // frame placement in the tip joint
  double baumgarte_time_step;
  int ric_fp32;            // 1: the cost-to-go P, s is STORED in single precision (rounded after every stage of the backward sweep; BASELINE configs[4]'s tolerance study)
};

struct OcpBuffers {
  const DevModel* model;
  const OcpProblem* prob;
  const OcpNode* nodes;  // [M] the chain
  int M, NS;              // host-side copies of prob->M, prob->NS (set with every discretisation): kernel ARGUMENTS, i.e. scalar loads from the
                          // kernarg segment -- every per-stage kernel needs them to find its stage, and P->M is a global load it would wait for
                          // before it can even request its node (three dependent latencies to the first record byte instead of two)
  const int* impulse_pos; // chain positions of the impulse stages
  int n_impulse_fe;       // host-side: number of impulse stages of a FORWARD-EULER chain (OCPSolver): nominal records + tangent items; 0 under ParNMPC (K5a / K9i)
  const int* general_pos; // ParNMPC: chain positions of the stages with a general KKT shape (aux with switching rows, impulse)
  const int* switch_pos;  // chain positions of the stages that carry a switching constraint (K5s runs on these only)
  int n_switch;           // host-side: their number
  const int* cond_pos;    // chain positions grouped by stage class of K5b (OcpLaunch::condenseMixed)
  int leg_axes_xyy;       // host-side: every leg is (joint about +x, about +y, about +y) with identity placement rotations, and the
                          // contact frames are not rotated against their joints (ANYmal):
                          // selects the instantiations of K5 whose rigid-body sweeps know the joint axes at compile time
  const double* q_ref;   // [M][NQ] reference configuration of every stage of the chain (time-varying cost)
  const double* task_refs;   // [M][12] reference pose of the task-space cost at every stage of the chain (rotation row-major, position): the
                             // constant cost.task_ref, or the poses of idocp_ocp_set_task_refs (TimeVaryingTaskSpace3DCost / 6DCost)
  // per-stage arrays, indexed [instance][slot] (NS slots per instance)
  double* sol;           // [batch][NS][SOL]
  double* dir;           // [batch][NS][DIR]
  double* slack;         // [batch][NS][CON]
  double* dual;          // [batch][NS][CON]
  double* lin;           // [batch][NS][LIN]
  double* lie;           // [batch][NS][LIE]
  double* nom;           // [batch][NS][NOM]   nominal rigid-body record (ocp_nominal_kernel -> K5 / K8 / merit)
  double* ext;           // [batch][NS][EXT]   nullptr unless a term of ocp_ext_kernel.hip is in use (ContactDistance)
  double* kkt;           // [batch][NS][KKT]   (terminal record holds Qxx and lx only)
  double* exp;           // [batch][NS][EXP]   (terminal record holds Fqq_prev_inv only)
  double* ric;           // [batch][NS][RIC]
  double* gain;          // [batch][NS][GAIN]
  double* swc;           // [batch][NS][SWC]   (only stages that carry a switching constraint)
  // ParNMPC only
  double* snew;          // [batch][NS][SNEW]
  double* kinv;          // [batch][NS][KINV]
  double* aux;           // [batch][NS][AUX]   aux_mat of every stage (BackwardCorrectionSolver::aux_mat_)
  double* xres;          // [batch][NS][XRES]
  double* fwd_prev;      // [batch][NQ + NV]   corrected (q, v) of the left neighbour's last stage (sharded horizon)
  double* step_stage;    // [batch][NS][2]
  double* step;          // [batch][2]
  double* err_stage;     // [batch][NS]
  double* err;           // [batch]
  int* status;           // [batch]
  long long* prof;       // [64] diagnostic: wall-clock stamps of one workgroup of the condensation kernel
  int prof_dimf;         // which instantiation of the condensation kernel stamps (its DIMF; IDOCP_PROF_DIMF, default: all feet)
  // filter line search (src/line_search/line_search.cpp): trial iterate s (+) alpha d, per-stage (cost, violation, barrier) and their sums
  double* sol_try;       // [batch][NS][SOL]
  double* merit_stage;   // [batch][NS][4]   cost, l1 violation, dt * barrier cost, -
  double* merit;         // [batch][2]
  double* ls_alpha;      // [batch]
  const OcpNode* nodes_ls;   // the chain with the reference's pairing of the stages in front of an event (line_search.cpp:80-113)
};

}  // namespace idocp_dev
#endif  // IDOCP_OCP_DEVICE_HPP_
