// Task-space cost terms of a fixed-base serial chain on the device
// (TaskSpace3DCost / TaskSpace6DCost and their TimeVarying variants:
// src/cost/task_space_3d_cost.cpp:60-157, src/cost/task_space_6d_cost.cpp,
// src/cost/time_varying_task_space_6d_cost.cpp:53-192 of the reference).
//
// The reference evaluates  diff = log6(M_ref^-1 M_frame(q))  (3D: p_frame - p_ref)
// and JJ = Jlog6(M_ref^-1 M_frame) J_frame,LOCAL  (3D: R_frame J_lin) through
// pinocchio's frame kinematics, then lq += dt JJ^T W diff, Qqq += dt JJ^T W JJ.
// Here ONE LANE PER JOINT produces its own column JJ[:, k] in registers: the
// forward kinematics of the chain is cheap enough (7 rotations) to be repeated by
// every lane, and the lane only has to remember the axis and origin of ITS joint
// on the way; the columns meet in LDS for the Gauss-Newton product.
#ifndef IDOCP_DEV_TASK_HPP_
#define IDOCP_DEV_TASK_HPP_

#include "dev_lie.hpp"
#include "dev_rbd.hpp"

namespace idocp_dev {

// Frame + weights of the task cost (idocp_cost_t::task_*), part of UnProblem.
struct TaskCost {
  int dim;             // 0 none, 3, 6
  int joint;           // parent joint of the frame
  double R[9], p[3];   // placement of the frame in the joint frame (R row-major)
  double weight[6], weightf[6];
  double ref[12];      // constant reference of a task_extra component (the first component's references live in UnBuffers::task_ref, per stage)
};

// diff[6] and, for joint k, col[6] = JJ[:, k].  cs = {cos q_i, sin q_i}; ref = rotation (row-major) + position.
// dim 3 leaves entries 3..5 zero, so that the callers always sum six weighted terms.
template <int NJ, typename Model>
__device__ __forceinline__ void taskSpaceColumn(const Model* m, const TaskCost& tc, const double* cs,
                                                const double* __restrict__ ref, int k, double* diff, double* col) {
  double oR[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, op[3] = {0, 0, 0};
  double ak[3] = {0, 0, 0}, ok[3] = {0, 0, 0};
#pragma unroll 1
  for (int i = 0; i < NJ; ++i) {
    if (i > tc.joint) break;
    Mat3<double> Ri;
    revoluteRotation<double>(m->R[i], m->axis[i], cs[2 * i], cs[2 * i + 1], Ri);
    double t[3];
    lieMatvec3(oR, m->p[i], t);
    op[0] += t[0]; op[1] += t[1]; op[2] += t[2];
    lieMatmul3(oR, Ri.m, oR);
    if (i == k) {                    // joint k: axis in world coordinates and the joint origin
      lieMatvec3(oR, m->axis[i], ak);
      ok[0] = op[0]; ok[1] = op[1]; ok[2] = op[2];
    }
  }
  double fR[9], fp[3];
  lieMatmul3(oR, tc.R, fR);
  lieMatvec3(oR, tc.p, fp);
  fp[0] += op[0]; fp[1] += op[1]; fp[2] += op[2];
  // world-frame velocity of the frame origin per unit joint rate: a_k x (p_frame - o_k); zero past the frame's joint
  const double d[3] = {fp[0] - ok[0], fp[1] - ok[1], fp[2] - ok[2]};
  const double lw[3] = {ak[1] * d[2] - ak[2] * d[1], ak[2] * d[0] - ak[0] * d[2], ak[0] * d[1] - ak[1] * d[0]};
  const double e[3] = {fp[0] - ref[9], fp[1] - ref[10], fp[2] - ref[11]};
  if (tc.dim == 3) {
    // diff = p - p_ref ; JJ = R_frame J_lin,LOCAL = the world-frame linear column
#pragma unroll
    for (int r = 0; r < 3; ++r) { diff[r] = e[r]; col[r] = lw[r]; diff[3 + r] = 0.0; col[3 + r] = 0.0; }
    return;
  }
  // diff_SE3 = M_ref^-1 M_frame
  double Rd[9], pd[3], tw[6];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int c = 0; c < 3; ++c) Rd[3 * r + c] = ref[r] * fR[c] + ref[3 + r] * fR[3 + c] + ref[6 + r] * fR[6 + c];
    pd[r] = ref[r] * e[0] + ref[3 + r] * e[1] + ref[6 + r] * e[2];
    // LOCAL twist column of joint k: (R_frame^T lin_world, R_frame^T a_k)
    tw[r] = fR[r] * lw[0] + fR[3 + r] * lw[1] + fR[6 + r] * lw[2];
    tw[3 + r] = fR[r] * ak[0] + fR[3 + r] * ak[1] + fR[6 + r] * ak[2];
  }
  double J[36];
  lieLog6Jlog6(Rd, pd, diff, J);
  // Jlog6 = [[A, B], [0, A]]: the lower-left block is structurally zero
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    col[r] = J[r] * tw[0] + J[r + 6] * tw[1] + J[r + 12] * tw[2] + J[r + 18] * tw[3] + J[r + 24] * tw[4] + J[r + 30] * tw[5];
    col[3 + r] = J[3 + r + 18] * tw[3] + J[3 + r + 24] * tw[4] + J[3 + r + 30] * tw[5];
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_TASK_HPP_
