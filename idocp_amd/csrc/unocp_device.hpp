// Device data layout + kernel parameter block of the fixed-base (UnOCP) path.
//
// HBM layout.  Every per-stage quantity of every OCP
// instance lives in ONE horizon-batched array of fixed-stride records,
//     record(b, i) = base + (b * (N+1) + i) * STRIDE        (FP64)
// with STRIDE rounded up to 16 doubles (128 B = one L2 line) so that the lane
// group that owns a stage reads/writes its record with whole-line, coalesced
// accesses.  Matrices inside a record are column-major like the reference's
// Eigen blocks.
#ifndef IDOCP_UNOCP_DEVICE_HPP_
#define IDOCP_UNOCP_DEVICE_HPP_

#include "dev_rbd.hpp"
#include "dev_task.hpp"

namespace idocp_dev {

__host__ __device__ constexpr int roundUp16(int n) { return (n + 15) / 16 * 16; }

// pdipm::SetSlackAndDualPositive for one entry (include/idocp/constraints/pdipm.hxx:13-23: `while (slack < barrier) slack += barrier`).
// The reference's loop is followed addition by addition as long as it is short (the same roundings: the result is bit-identical in
// every case a test can reach); a slack that is still below the barrier after 1024 additions -- a constraint violated by more than
// a thousand barriers, e.g. a foot a metre under the ground with barrier 1e-4 -- is lifted in one step, so that no lane spins through
// millions of serial iterations and none leaves with a negative slack (a negative slack would make the dual negative and the
// barrier cost NaN without an error status).
__host__ __device__ inline double slackPositive(double sl, double barrier) {
  for (int it = 0; it < 1024 && sl < barrier; ++it) sl += barrier;
  if (sl < barrier) {
    sl += ceil((barrier - sl) / barrier) * barrier;
    while (sl < barrier) sl += barrier;          // (rounding of the product: at most one more)
  }
  return sl;
}
// records of the fixed-base path: even length (16-byte accesses stay aligned), no padding to 128 bytes -- the kernels that stream them walk
// consecutive records, and with seven joints the padding was 7.5 % of the bytes of a step (solution record: 49 -> 64 doubles)
__host__ __device__ constexpr int roundUp2(int n) { return (n + 1) / 2 * 2; }

template <int NV>
struct UnLayout {
  // solution record  (SplitSolution, include/idocp/ocp/split_solution.hxx:10-31)
  static constexpr int S_LMD = 0, S_GMM = NV, S_Q = 2 * NV, S_V = 3 * NV, S_A = 4 * NV, S_U = 5 * NV, S_BETA = 6 * NV;
  static constexpr int SOL = roundUp2(7 * NV);
  // direction record (SplitDirection, split_direction.hxx:8-23): same field order with d-prefix
  // IPM rows: [q_lower, q_upper, v_lower, v_upper, u_lower, u_upper] x NV
  static constexpr int NC = 6 * NV;
  static constexpr int CON = roundUp2(NC);
  // condensed stage KKT (SplitUnKKTMatrix / SplitUnKKTResidual, split_unkkt_matrix.hxx:31-147,
  // split_unkkt_residual.hxx:29-103); only the blocks the Riccati step reads
  // The symmetric blocks (Qaa, Qqq, Qvv here; Pqq, Pvv below) travel as their UPPER triangle, column by column: entry (r, c), r <= c,
  // at c (c + 1) / 2 + r  (round 3: 105 of the 492 doubles of the kkt + ric records were the mirror image of another 105)
  static constexpr int NS = NV * (NV + 1) / 2;
  __host__ __device__ static constexpr int sym(int r, int c) { return r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c; }
  static constexpr int K_QAA = 0, K_QAQ = NS, K_QAV = K_QAQ + NV * NV, K_QQQ = K_QAV + NV * NV, K_QQV = K_QQQ + NS,
                       K_QVV = K_QQV + NV * NV, K_FQ = K_QVV + NS, K_FV = K_FQ + NV, K_LA = K_FV + NV, K_LQ = K_LA + NV,
                       K_LV = K_LQ + NV;
  static constexpr int KKT = roundUp2(K_LV + NV);
  // inverse-dynamics cache needed by the expansion (UnconstrainedDynamics members,
  // unconstrained_dynamics.hpp): dID/dq, dID/dv, dID/da, ID, lu, diag(Quu)
  static constexpr int D_DQ = 0, D_DV = NV * NV, D_DA = 2 * NV * NV, D_ID = 3 * NV * NV, D_LU = D_ID + NV, D_QUU = D_LU + NV;
  static constexpr int DYN = roundUp2(3 * NV * NV + 3 * NV);
  // Riccati factorization (SplitRiccatiFactorization, split_riccati_factorization.hpp:15-134)
  static constexpr int R_PQQ = 0, R_PQV = NS, R_PVV = R_PQV + NV * NV, R_SQ = R_PVV + NS, R_SV = R_SQ + NV;
  static constexpr int RIC = roundUp2(R_SV + NV);
  // LQR policy (lqr_state_feedback_policy.hpp:11-28): K (NV x 2NV col-major), k
  static constexpr int G_K = 0, G_k = 2 * NV * NV;
  static constexpr int GAIN = roundUp2(2 * NV * NV + NV);
  // UnParNMPC: the blocks of the per-stage KKT inverse the correction sweeps read (KKT ordering lmd, gmm | a, q, v;
  // split_unbackward_correction.hxx:72-104): TL = -S^-1 (NX x NX), TR = S^-1 F Q^-1 (NX x 3NV), BRC = the (q, v) columns
  // of the bottom-right block (3NV x NX); all column-major
  static constexpr int NX = 2 * NV, NQ3 = 3 * NV;
  static constexpr int I_TL = 0, I_TR = NX * NX, I_BRC = I_TR + NX * NQ3;
  static constexpr int KINV = roundUp16(I_BRC + NQ3 * NX);
  static constexpr int AUX = roundUp16(NX * NX);            // aux_mat of a stage (unbackward_correction.hpp)
  static constexpr int XRES = roundUp16(2 * NX);            // x_res of the backward [0, NX) and of the forward [NX, 2NX) sweep
  // terminal task-space terms of an instance (TaskSpace*Cost::computeTerminalCost / Derivatives / Hessian): cost, gradient, Hessian
  static constexpr int T_COST = 0, T_G = 1, T_H = 1 + NV;
  static constexpr int TASK = roundUp16(1 + NV + NV * NV);
};

// Problem constants (cost, limits, IPM parameters), uniform across the grid.
struct UnProblem {
  int N, batch;
  double T, dt;
  double q_ref[IDOCP_MAX_NV], v_ref[IDOCP_MAX_NV], u_ref[IDOCP_MAX_NV];
  double q_weight[IDOCP_MAX_NV], v_weight[IDOCP_MAX_NV], a_weight[IDOCP_MAX_NV], u_weight[IDOCP_MAX_NV];
  double qf_weight[IDOCP_MAX_NV], vf_weight[IDOCP_MAX_NV];
  double q_min[IDOCP_MAX_NV], q_max[IDOCP_MAX_NV], v_max[IDOCP_MAX_NV], u_max[IDOCP_MAX_NV];
  double a_min[IDOCP_MAX_NV], a_max[IDOCP_MAX_NV];      // JointAccelerationLowerLimit / UpperLimit (components 6 / 7; rows in slack_a / dual_a)
  int use_a_lower, use_a_upper;
  int use_q_limits, use_v_limits, use_u_limits;
  double barrier, fraction_rate;
  int backward_euler;     // UnParNMPC: stage i sits at t + (i + 1) dt, constraint time step i + 1, the last stage is terminal
  // UnParNMPC horizon shard (idocp_unparnmpc_create_shard): local stage i is stage stage_offset + i of the horizon;
  // has_terminal: the shard ends with the terminal stage (else record N holds the right neighbour's first stage: lmd, gmm,
  // aux, corrected lmd, gmm); has_prev: (q0, v0) is the left neighbour's last stage and B.xprev its corrected (q, v)
  int stage_offset, has_terminal, has_prev;
  TaskCost task;          // TaskSpace3DCost / TaskSpace6DCost (UnOCP only); task.dim = 0: none
  int task_n;             // number of task-space components: 0, or 1 + idocp_cost_t::task_extra_count
  TaskCost task_extra[IDOCP_MAX_EXTRA_TASKS];      // the further components (constant references)
};

// All device pointers of one handle.
struct UnBuffers {
  const DevModel* model;
  const UnProblem* prob;
  double* sol;       // [batch][N+1][SOL]
  double* dir;       // [batch][N+1][SOL]
  double* slack;     // [batch][N][CON]
  double* dual;      // [batch][N][CON]
  double* slack_a;   // [batch][N][2 NV]   rows of the acceleration limits (components 6, 7); nullptr unless one of them is in use
  double* dual_a;    // [batch][N][2 NV]
  double* kkt;       // [batch][N][KKT]
  double* dyn;       // [batch][N][DYN]
  double* ric;       // [batch][N+1][RIC]
  double* gain;      // [batch][N][GAIN]
  double* step_stage;  // [batch][N][2]  (primal, dual) fraction-to-boundary per stage
  double* step;        // [batch][2]
  double* err_stage;   // [batch][N+1]   squared KKT residual per stage
  double* err;         // [batch]
  int* status;         // [batch]  0 ok, 1+stage on a failed Cholesky
  // filter line search (UnLineSearch): trial step per instance, (cost, violation) per stage and per instance
  double* ls_alpha;    // [batch]
  double* ls_stage;    // [batch][N+1][2]
  double* ls_out;      // [batch][2]
  // UnParNMPC only (null otherwise)
  double* kinv;        // [batch][N][KINV]
  double* snew;        // [batch][N+1][SOL]  coarse / corrected iterate s_new (lmd, gmm, q, v, a)
  double* aux;         // [batch][N+1][AUX]
  double* xres;        // [batch][N+1][XRES]
  double* xprev;       // [batch][2 NV]  shard with has_prev: corrected (q, v) of the left neighbour's last stage
  int zaxes;             // host-side: every joint axis of the chain is +z (selects the compile-time variant of the rigid-body sweep)
  // task-space cost (null / 0 without one)
  int task;              // host-side copy of prob->task.dim != 0: selects the kernel instantiations
  int task_stride;       // UnLayout<NV>::TASK for the kernels that are not templated on NV
  double* task_ref;      // [N+1][12]  reference pose of every stage: rotation (row-major) + position, shared by the batch
  double* task_term;     // [batch][TASK]  terminal cost, gradient, Gauss-Newton Hessian (un_task_terminal_kernel), all components
  double* task_xs;       // [batch][N][TASK]  STAGE cost, gradient, Hessian of the task_extra components, time step included (un_task_terms_kernel;
                         // null without such components): K1 / K4 add them to lq and Qqq, the line search to its cost
};

}  // namespace idocp_dev
#endif  // IDOCP_UNOCP_DEVICE_HPP_
