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
  // solution (split_solution.hxx:10-31): lmd gmm q v a u beta f mu nu_passive
  static constexpr int S_LMD = 0, S_GMM = NV, S_Q = 2 * NV, S_V = S_Q + NQ, S_A = S_V + NV, S_U = S_A + NV, S_BETA = S_U + NU,
                       S_F = S_BETA + NV, S_MU = S_F + NF, S_NUP = S_MU + NF;
  static constexpr int SOL = roundUp16(S_NUP + 6);
  // direction (split_direction.hxx:8-23)
  static constexpr int D_LMD = 0, D_GMM = NV, D_Q = 2 * NV, D_V = 3 * NV, D_A = 4 * NV, D_U = 5 * NV, D_BETA = D_U + NU,
                       D_F = D_BETA + NV, D_MU = D_F + NF, D_NUP = D_MU + NF;
  static constexpr int DIR = roundUp16(D_NUP + 6);
  // IPM rows: 6 joint-limit components x NU, then 5 friction-cone rows per contact
  static constexpr int C_FRIC = 6 * NU, NCON = 6 * NU + 5 * NC;
  static constexpr int CON = roundUp16(NCON);
  // linearisation record written by the tangent-RNEA kernel: [dID;dC]/d(q,v) (NVF x NX, ld NVF),
  // dID/da = M (NV x NV), dC/da = J (NF x NV, ld NF), [ID; C]
  static constexpr int L_DIDC = 0, L_M = NVF * NX, L_J = L_M + NV * NV, L_IDC = L_J + NF * NV;
  static constexpr int LIN = roundUp16(L_IDC + NVF);
  // condensed LQR stage for the Riccati sweep
  static constexpr int K_QXX = 0, K_QXU = NX * NX, K_QUU = K_QXU + NX * NU, K_FQQ = K_QUU + NU * NU, K_FQV = K_FQQ + 36,
                       K_FVQ = K_FQV + 36, K_FVV = K_FVQ + NV * NV, K_FVU = K_FVV + NV * NV, K_LX = K_FVU + NV * NU,
                       K_LU = K_LX + NX, K_FX = K_LU + NU;
  static constexpr int KKT = roundUp16(K_FX + NX);
  // expansion cache (ContactDynamicsData members + passive blocks + Fqq_prev_inv)
  static constexpr int E_MJ = 0, E_MJD = NVF * NVF, E_QAFQV = E_MJD + NVF * NX, E_QAFU = E_QAFQV + NVF * NX,
                       E_MJIDC = E_QAFU + NVF * NU, E_LAF = E_MJIDC + NVF, E_LUP = E_LAF + NVF, E_QUUP = E_LUP + 6,
                       E_QXUP = E_QUUP + 6 * NU, E_FQQPI = E_QXUP + NX * 6;
  static constexpr int EXP = roundUp16(E_FQQPI + 36);
  static constexpr int R_PQQ = 0, R_PQV = NV * NV, R_PVV = 2 * NV * NV, R_SQ = 3 * NV * NV, R_SV = R_SQ + NV;
  static constexpr int RIC = roundUp16(R_SV + NV);
  // Lie-group terms of the floating base, produced by the small pre-kernel (6x6 blocks column-major)
  static constexpr int Z_JQ = 0, Z_QDIFF = 36, Z_FQQ = 44, Z_FQ6 = 80, Z_FQQI = 88, Z_FQQP = 124, Z_FQQPI = 160;
  static constexpr int LIE = roundUp16(196);
  static constexpr int G_K = 0, G_k = NU * NX;
  static constexpr int GAIN = roundUp16(G_k + NU);
};

struct OcpProblem {
  int N, batch;
  double T, dt;
  double v_ref[IDOCP_MAX_NV], u_ref[IDOCP_MAX_NV];
  double q_weight[IDOCP_MAX_NV], v_weight[IDOCP_MAX_NV], a_weight[IDOCP_MAX_NV], u_weight[IDOCP_MAX_NV];
  double qf_weight[IDOCP_MAX_NV], vf_weight[IDOCP_MAX_NV];
  double f_weight[IDOCP_MAX_CONTACTS][3], f_ref[IDOCP_MAX_CONTACTS][3];
  double q_min[IDOCP_MAX_NV], q_max[IDOCP_MAX_NV], v_max[IDOCP_MAX_NV], u_max[IDOCP_MAX_NV];
  int use_q_limits, use_v_limits, use_u_limits, use_friction_cone;
  double mu, barrier, fraction_rate;
  // contact status of the horizon (uniform): active flags, packed row of each contact (-1 inactive), dimf
  int active[IDOCP_MAX_CONTACTS], row_of[IDOCP_MAX_CONTACTS], dimf;
  double contact_point[IDOCP_MAX_CONTACTS][3];     // world
  double contact_R[IDOCP_MAX_CONTACTS][9], contact_p[IDOCP_MAX_CONTACTS][3];   // frame placement in the tip joint
  double baumgarte_time_step;
};

struct OcpBuffers {
  const DevModel* model;
  const OcpProblem* prob;
  const double* q_ref;   // [N+1][NQ] reference configuration of every stage (time-varying cost)
  double* sol;           // [batch][N+1][SOL]
  double* dir;           // [batch][N+1][DIR]
  double* slack;         // [batch][N][CON]
  double* dual;          // [batch][N][CON]
  double* lin;           // [batch][N][LIN]
  double* lie;           // [batch][N+1][LIE]
  double* kkt;           // [batch][N+1][KKT]   (terminal record holds Qxx and lx only)
  double* exp;           // [batch][N+1][EXP]   (terminal record holds Fqq_prev_inv only)
  double* ric;           // [batch][N+1][RIC]
  double* gain;          // [batch][N][GAIN]
  double* step_stage;    // [batch][N][2]
  double* step;          // [batch][2]
  double* err_stage;     // [batch][N+1]
  double* err;           // [batch]
  int* status;           // [batch]
  long long* prof;       // [64] diagnostic: wall-clock stamps of one workgroup of the condensation kernel
};

}  // namespace idocp_dev
#endif  // IDOCP_OCP_DEVICE_HPP_
