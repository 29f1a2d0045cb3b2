// Terms of a stage that need frame Jacobians of their own, outside the condensation kernel: they exist only when the problem
// asks for them (OcpBuffers::ext != nullptr), so the hot path carries two uniform branches and nothing else.
//
//   ContactDistance (src/constraints/contact_distance.cpp): the frames of the contacts that are NOT active on a stage stay above
//   z = 0 -- one IPM row per contact (rows of active contacts idle).  With J_c = row 2 of the LOCAL frame Jacobian of contact c
//   (Robot::getFrameJacobian, contact_distance.cpp:73, the reference's choice) and z_c the world height of the frame:
//     residual = - z_c + slack,   lq -= dt (dual + (dual residual - duality) / slack) J_c^T,   Qqq += dt (dual / slack) J_c^T J_c,
//     dslack = J_c dq - residual.
//
// K_x1  ocp_ext_kernel          before the condensation kernel: kinematics of the feet, the rows J_c, the gradient term (added to lq
//                               by the condensation kernel's row threads), the Hessian weights, the stage's share of the KKT error /
//                               of the line search's violation
// K_x2  ocp_ext_hessian_kernel  after it: the rank-one terms on the Qqq block of the kkt record
// The expansion kernels (K6 / K7 / trial iterate) read z_c and J_c from the ext record through ipmRow (ocp_expand_kernel.hip).
#include <hip/hip_runtime.h>

#include "dev_lie.hpp"
#include "dev_rbd.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

namespace {

struct Rot { double m[9]; };      // row-major
__device__ __forceinline__ void matmul(const double* A, const double* B, double* C) {
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void matvec(const double* A, const double* x, double* y) {
#pragma unroll
  for (int i = 0; i < 3; ++i) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1]; c[1] = a[2] * b[0] - a[0] * b[2]; c[2] = a[0] * b[1] - a[1] * b[0];
}

}  // namespace

// World placement of a frame at configuration q (Robot::updateFrameKinematics + framePosition / frameRotation, robot.hxx:166-188) --
// parent joint jf of the model (0: the floating base, 1 + leg LJ + j: joint j of a leg), placement (Rf, pf) in the joint frame -- and,
// for the velocity index `dof` (< 0: none), the world-frame motion the unit velocity of that degree of freedom gives the frame: angular
// part w, linear part v at the frame origin (zero if the frame does not move with it).
template <typename D>
__device__ inline void frameKinematics(const DevModel* __restrict__ m, const double* __restrict__ q, int jf, const double* __restrict__ Rf,
                                       const double* __restrict__ pf, int dof, double* pF, double* RF, double* w, double* v) {
  constexpr int LJ = D::LJ;
  const int leg = jf > 0 ? (jf - 1) / LJ : 0, njoints = jf > 0 ? (jf - 1) % LJ + 1 : 0;
  const double qx = q[3], qy = q[4], qz = q[5], qw = q[6];
  double Rw[9];
  Rw[0] = 1 - 2 * (qy * qy + qz * qz); Rw[1] = 2 * (qx * qy - qz * qw);     Rw[2] = 2 * (qx * qz + qy * qw);
  Rw[3] = 2 * (qx * qy + qz * qw);     Rw[4] = 1 - 2 * (qx * qx + qz * qz); Rw[5] = 2 * (qy * qz - qx * qw);
  Rw[6] = 2 * (qx * qz - qy * qw);     Rw[7] = 2 * (qy * qz + qx * qw);     Rw[8] = 1 - 2 * (qx * qx + qy * qy);
  double pw[3] = {q[0], q[1], q[2]};
  double wdof[3] = {0, 0, 0}, odof[3] = {0, 0, 0};      // world axis and origin of the lane's own degree of freedom
  bool lin = false, moves = false;
  if (dof >= 0 && dof < 3) { lin = true; moves = true; wdof[0] = Rw[dof]; wdof[1] = Rw[3 + dof]; wdof[2] = Rw[6 + dof]; }       // R_wb e_dof
  else if (dof >= 3 && dof < 6) { moves = true; wdof[0] = Rw[dof - 3]; wdof[1] = Rw[3 + dof - 3]; wdof[2] = Rw[6 + dof - 3]; odof[0] = pw[0]; odof[1] = pw[1]; odof[2] = pw[2]; }
  for (int j = 0; j < njoints; ++j) {
    const int ji = 1 + leg * LJ + j, d = 6 + leg * LJ + j;
    double s, c;
    sincos(q[d + 1], &s, &c);
    Mat3<double> Rm;
    revoluteRotation<double>(m->R[ji], m->axis[ji], c, s, Rm);
    double t[3];
    matvec(Rw, m->p[ji], t);
    pw[0] += t[0]; pw[1] += t[1]; pw[2] += t[2];
    double Rn[9];
    matmul(Rw, Rm.m, Rn);
#pragma unroll
    for (int e = 0; e < 9; ++e) Rw[e] = Rn[e];
    if (d == dof) { moves = true; matvec(Rw, m->axis[ji], wdof); odof[0] = pw[0]; odof[1] = pw[1]; odof[2] = pw[2]; }      // (the axis is invariant under its own rotation)
  }
  double t[3];
  matvec(Rw, pf, t);
  pF[0] = pw[0] + t[0]; pF[1] = pw[1] + t[1]; pF[2] = pw[2] + t[2];
  matmul(Rw, Rf, RF);
  if (!moves) { w[0] = w[1] = w[2] = 0.0; v[0] = v[1] = v[2] = 0.0; return; }
  if (lin) { w[0] = w[1] = w[2] = 0.0; v[0] = wdof[0]; v[1] = wdof[1]; v[2] = wdof[2]; return; }
  w[0] = wdof[0]; w[1] = wdof[1]; w[2] = wdof[2];
  const double r[3] = {pF[0] - odof[0], pF[1] - odof[1], pF[2] - odof[2]};
  cross3(wdof, r, v);
}
// the contact frame of leg `leg` (the tip joint of the leg)
template <typename D>
__device__ inline void contactFrameKinematics(const DevModel* __restrict__ m, const OcpProblem* __restrict__ P, const double* __restrict__ q,
                                              int leg, int dof, double* pF, double* RF, double* w, double* v) {
  frameKinematics<D>(m, q, 1 + leg * D::LJ + D::LJ - 1, P->contact_R[leg], P->contact_p[leg], dof, pF, RF, w, v);
}

template <typename D>
__global__ __launch_bounds__(64) void ocp_ext_kernel(OcpBuffers B, int residual) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NC = D::NC;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const int lane = threadIdx.x;
  const long unit = blockIdx.x;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const long rec = b * B.NS + nd->slot;
  double* __restrict__ xx = B.ext + rec * L::EXT;
  const bool bwd = P->backward_euler != 0;
  if (bwd && pos == M - 1) {                        // ParNMPC: placeholder behind the last stage
    for (int e = lane; e < L::EXT; e += 64) xx[e] = 0.0;
    return;
  }
  const double* __restrict__ q = B.sol + rec * L::SOL + L::S_Q;
  const int dof = lane < NV ? lane : -1;
  const double dt = nd->dt;                         // (1 on impulse stages)
  double lq = 0.0, err = 0.0, viol = 0.0;
  // ---- ContactDistance ----
  const bool cd = P->use_contact_distance && nd->kind != 1 && nd->kind != 4 && nd->level >= 2;
  const double* __restrict__ slack = B.slack + rec * L::CON + L::C_CD;
  const double* __restrict__ dual = B.dual + rec * L::CON + L::C_CD;
  for (int c = 0; c < NC; ++c) {
    double e = 0.0, z = 0.0, h = 0.0;
    if (cd) {
      double pF[3], RF[9], w[3], v[3];
      contactFrameKinematics<D>(B.model, P, q, c, dof, pF, RF, w, v);
      z = pF[2];
      if (!nd->active[c]) {
        e = RF[2] * v[0] + RF[5] * v[1] + RF[8] * v[2];      // (R_wF^T v)_z: row 2 of the LOCAL frame Jacobian
        const double sl = slack[c], du = dual[c];
        const double res = -z + sl, duality = sl * du - P->barrier;
        double g = -dt * du;                                                   // augmentDualResidual (contact_distance.cpp:68-78)
        if (!residual) { g -= dt * (du * res - duality) / sl; h = dt * du / sl; }      // condenseSlackAndDual (:81-102)
        err += res * res + duality * duality;
        viol += dt * fabs(res);
        lq += g * e;
      }
    }
    if (lane < NV) xx[L::X_CDJ + c * NV + lane] = e;
    if (lane == 0) { xx[L::X_Z + c] = z; xx[L::X_W + c] = h; }
  }
  // ---- TaskSpace3DCost / TaskSpace6DCost (task_space_3d_cost.cpp:60-157, task_space_6d_cost.cpp:68-178) ----
  double cost = 0.0;
  // one pass per component (CostFunction sums its components): the task_* block of the cost, then its task_extra components (constant references)
  for (int tcomp = 0; tcomp < L::NT; ++tcomp) {
    if (tcomp >= P->task_n) {
      for (int e = lane; e < 6 * NV; e += 64) xx[L::X_TJ + tcomp * 6 * NV + e] = 0.0;
      if (lane < 6) xx[L::X_TW + tcomp * 6 + lane] = 0.0;
      continue;
    }
    const int t_dim = tcomp == 0 ? P->task_dim : P->task_extra[tcomp - 1].dim;
    const int t_joint = tcomp == 0 ? P->task_joint : P->task_extra[tcomp - 1].joint;
    const double* __restrict__ t_R = tcomp == 0 ? P->task_R : P->task_extra[tcomp - 1].R;
    const double* __restrict__ t_p = tcomp == 0 ? P->task_p : P->task_extra[tcomp - 1].p;
    const double* __restrict__ t_w = tcomp == 0 ? P->task_weight : P->task_extra[tcomp - 1].weight;
    const double* __restrict__ t_wf = tcomp == 0 ? P->task_weightf : P->task_extra[tcomp - 1].weightf;
    const double* __restrict__ t_wi = tcomp == 0 ? P->task_weighti : P->task_extra[tcomp - 1].weighti;
    double pF[3], RF[9], w[3], v[3], diff[6], col[6];
    frameKinematics<D>(B.model, q, t_joint, t_R, t_p, dof, pF, RF, w, v);
    const double* __restrict__ ref = tcomp == 0 ? B.task_refs + (long)pos * 12 : P->task_extra[tcomp - 1].ref;      // (time_varying_task_space_{3d,6d}_cost.cpp: the reference at the stage's own time)
    const double ev[3] = {pF[0] - ref[9], pF[1] - ref[10], pF[2] - ref[11]};
    if (t_dim == 3) {
      // diff = p - p_ref ; J_3d = R_frame J_lin,LOCAL = the world-frame linear column
      for (int r = 0; r < 3; ++r) { diff[r] = ev[r]; col[r] = v[r]; diff[3 + r] = 0.0; col[3 + r] = 0.0; }
    } else {
      // diff = log6(M_ref^-1 M_frame), JJ = Jlog6(M_ref^-1 M_frame) J_frame,LOCAL
      double Rd[9], pd[3], tw[6], J[36];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) Rd[3 * r + c] = ref[r] * RF[c] + ref[3 + r] * RF[3 + c] + ref[6 + r] * RF[6 + c];
        pd[r] = ref[r] * ev[0] + ref[3 + r] * ev[1] + ref[6 + r] * ev[2];
        tw[r] = RF[r] * v[0] + RF[3 + r] * v[1] + RF[6 + r] * v[2];
        tw[3 + r] = RF[r] * w[0] + RF[3 + r] * w[1] + RF[6 + r] * w[2];
      }
      lieLog6Jlog6(Rd, pd, diff, J);
      for (int r = 0; r < 6; ++r) {
        double acc = 0.0;
        for (int c = 0; c < 6; ++c) acc += J[r + 6 * c] * tw[c];
        col[r] = acc;
      }
    }
    // weights: stage dt w; impulse stage w_i; terminal stage w_f; ParNMPC's last stage carries stage AND terminal cost (terminal_parnmpc.hxx),
    // its terminal part is not in the line search's merit (line_search.cpp:228-237)
    const bool last = bwd && P->has_terminal && pos == M - 2;
    for (int k = 0; k < 6; ++k) {
      const double wraw = nd->kind == 1 ? t_wi[k] : (nd->kind == 4 ? t_wf[k] : dt * t_w[k]);
      const double wm = (t_dim == 3 && k >= 3) ? 0.0 : wraw;
      const double wk = wm + ((last && !(t_dim == 3 && k >= 3)) ? t_wf[k] : 0.0);
      lq += wk * diff[k] * col[k];
      cost += 0.5 * wm * diff[k] * diff[k];
      if (lane < NV) xx[L::X_TJ + (tcomp * 6 + k) * NV + lane] = col[k];
      if (lane == 0) xx[L::X_TW + tcomp * 6 + k] = residual ? 0.0 : wk;
    }
  }
  if (lane < NV) xx[L::X_LQ + lane] = lq;
  if (lane == 0) { xx[L::X_ERR] = err; xx[L::X_VIOL] = viol; xx[L::X_COST] = cost; }
}

// Qqq += sum_c w_c J_c^T J_c + JJ^T diag(w) JJ on the kkt record the condensation kernel has just written (the terminal record included)
template <typename D>
__global__ __launch_bounds__(64) void ocp_ext_hessian_kernel(OcpBuffers B) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NX = D::NX, NC = D::NC;
  const OcpProblem* __restrict__ P = B.prob;
  const int M = B.M;
  const long unit = blockIdx.x;
  const long b = unit / M;
  const int pos = (int)(unit - b * M);
  const OcpNode* __restrict__ nd = B.nodes + pos;
  if (P->backward_euler && pos == M - 1) return;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ xx = B.ext + rec * L::EXT;
  double* __restrict__ kk = B.kkt + rec * L::KKT;
  constexpr int NTW = 6 * L::NT;
  double w[NC + NTW];
  bool any = false;
#pragma unroll
  for (int c = 0; c < NC + NTW; ++c) { w[c] = c < NC ? xx[L::X_W + c] : xx[L::X_TW + c - NC]; any = any || w[c] != 0.0; }
  if (!any) return;
  const int ntw = 6 * P->task_n;
  for (int e = threadIdx.x; e < NV * NV; e += 64) {
    const int c2 = e / NV, r = e - c2 * NV;
    double acc = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) acc += w[c] * xx[L::X_CDJ + c * NV + r] * xx[L::X_CDJ + c * NV + c2];
    for (int k = 0; k < ntw; ++k) acc += w[NC + k] * xx[L::X_TJ + k * NV + r] * xx[L::X_TJ + k * NV + c2];
    if (r <= c2) kk[L::K_QXX + L::xsym(r, c2)] += acc;
  }
}

// heights of the contact frames of one configuration (ContactDistance::setSlackAndDual, contact_distance.cpp:58-65)
template <typename D>
__device__ void contactFrameHeights(const DevModel* __restrict__ m, const OcpProblem* __restrict__ P, const double* __restrict__ q, double* z) {
  for (int c = 0; c < D::NC; ++c) {
    double pF[3], RF[9], w[3], v[3];
    contactFrameKinematics<D>(m, P, q, c, -1, pF, RF, w, v);
    z[c] = pF[2];
  }
}
template <typename D>
__global__ __launch_bounds__(64) void ocp_ext_init_kernel(OcpBuffers B, long nrec) {
  // slack / dual of the ContactDistance rows of EVERY slot (the other rows: ocp_init_constraints_kernel)
  using L = OcpLayout<D>;
  const OcpProblem* __restrict__ P = B.prob;
  const long su = (long)blockIdx.x * 64 + threadIdx.x;
  if (su >= nrec) return;
  const int NS = B.NS, N = P->N, E = P->E;
  const int slot = (int)(su % NS);
  const bool impulse = (slot > N && slot <= N + E);
  const int level = (slot <= N) ? slot + (P->backward_euler ? 1 + P->stage_offset : 0) : (impulse ? -1 : 0);
  const bool valid = P->use_contact_distance && !impulse && level >= 2;
  double z[D::NC];
  if (valid) contactFrameHeights<D>(B.model, P, B.sol + su * L::SOL + L::S_Q, z);
  for (int c = 0; c < D::NC; ++c) {
    double sl = 1.0, dl = 0.0;
    if (valid) {
      sl = z[c];
      sl = slackPositive(sl, P->barrier);      // pdipm.hxx:17-20
      dl = P->barrier / sl;
    }
    B.slack[su * L::CON + L::C_CD + c] = sl;
    B.dual[su * L::CON + L::C_CD + c] = dl;
  }
}

template <typename D>
void OcpLaunch<D>::extRows(const OcpBuffers& B, long batch, int M, bool residual, hipStream_t st) {
  if (!B.ext) return;
  hipLaunchKernelGGL((ocp_ext_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B, residual ? 1 : 0);
}
template <typename D>
void OcpLaunch<D>::extHessian(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  if (!B.ext) return;
  hipLaunchKernelGGL((ocp_ext_hessian_kernel<D>), dim3((unsigned)(batch * M)), dim3(64), 0, st, B);
}
template <typename D>
void OcpLaunch<D>::extInit(const OcpBuffers& B, long batch, int NS, hipStream_t st) {
  if (!B.ext) return;
  const long nrec = batch * NS;
  hipLaunchKernelGGL((ocp_ext_init_kernel<D>), dim3((unsigned)((nrec + 63) / 64)), dim3(64), 0, st, B, nrec);
}

template void OcpLaunch<LeggedDims<4, 3>>::extRows(const OcpBuffers&, long, int, bool, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::extHessian(const OcpBuffers&, long, int, hipStream_t);
template void OcpLaunch<LeggedDims<4, 3>>::extInit(const OcpBuffers&, long, int, hipStream_t);

}  // namespace idocp_dev
