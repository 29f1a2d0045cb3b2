// K5s -- terms of the switching constraint carried by the stage two steps ahead of an impulse.
//
// Replaces ForwardSwitchingConstraint::linearizeSwitchingConstraint / computeSwitchingConstraintResidual
// (include/idocp/ocp/forward_switching_constraint.hxx:27-66) and the rigid-body calls behind it:
//   Robot::integrateConfiguration, updateKinematics(q), computeContactResidual, computeContactDerivative,
//   dIntegratedConfiguration, dIntegratedVelocity (robot.hxx:23-95, 322-356; point_contact.hxx:176-200).
//
//   dq = (dt1 + dt2) v + dt1 dt2 a ;  q' = q (+) dq ;  P = p_foot(q') - contact_point          (impulse-active feet)
//   Phiq = Pq dInt_dq ; Phiv = (dt1 + dt2) Pq dInt_dv ; Phia = dt1 dt2 Pq dInt_dv
//
// One wavefront per stage (stages without a switching constraint return at once).  Lane k < NV carries
// the tangent d/d(q'_k) through the forward kinematics of the tree, so column k of Pq = R_world J_lin
// comes out of one pass; dIntegrate only mixes the six base columns (Ad(exp6(dq)^-1) and Jexp6(dq)).
// Output: swc record = P, Phix = [Phiq Phiv], Phia (K5b then condenses them, S3 consumes them).
#include <hip/hip_runtime.h>

#include "dev_lie.hpp"
#include "dev_rbd.hpp"
#include "dev_dense.hpp"
#include "ocp_device.hpp"
#include "ocp_launch.hpp"

namespace idocp_dev {

template <typename D>
__global__ __launch_bounds__(64) void ocp_switch_kernel(OcpBuffers B, int nsw) {
  using L = OcpLayout<D>;
  constexpr int NV = D::NV, NQ = D::NQ, NL = D::NL, LJ = D::LJ, NF = D::NF, NU = D::NU;
  typedef Dual T;
  const OcpProblem* __restrict__ P = B.prob;
  // (a workgroup per instance and stage WITH a switching constraint: launched over every stage of the chain, the 110 of 119 workgroups
  //  per instance that returned at once still cost their dispatch -- 0.10 ms for a kernel with 0.03 ms of work)
  const long unit = blockIdx.x;
  const long b = unit / nsw;
  const int pos = B.switch_pos[(int)(unit - b * nsw)];
  const OcpNode* __restrict__ nd = B.nodes + pos;
  const int dimi = nd->sw_dimi;
  if (dimi == 0) return;
  __shared__ double s_dq[NV], s_q2[NQ], s_pq[NF][NV + 1], s_cs[NU][2], s_A0[36], s_Je[36], s_P[NF];
  const DevModel* __restrict__ m = B.model;
  const int lane = threadIdx.x;
  const long rec = b * B.NS + nd->slot;
  const double* __restrict__ s = B.sol + rec * L::SOL;
  const double* __restrict__ q = s + L::S_Q;
  const double dt1 = nd->sw_dt1, dt2 = nd->sw_dt2;
  if (lane < NV) s_dq[lane] = (dt1 + dt2) * s[L::S_V + lane] + (dt1 * dt2) * s[L::S_A + lane];
  for (int e = lane; e < NF * (NV + 1); e += 64) (&s_pq[0][0])[e] = 0.0;
  waveLdsSync();
  if (lane == 0) lieIntegrateBase(q, s_dq, 1.0, s_q2);
  if (lane >= 6 && lane < NV) s_q2[lane + 1] = q[lane + 1] + s_dq[lane];
  if (lane == 32) {
    double R[9], p[3];
    lieExp6(s_dq, R, p);
    lieDIntegrateArg0(R, p, s_A0);                 // dIntegrate_dq = Ad(exp6(dq)^-1)
  }
  if (lane == 33) {
    double R[9], p[3], Jl[36];
    lieExp6(s_dq, R, p);
    lieJlog6(R, p, Jl);
    lieBlockInverse(Jl, s_Je);                     // dIntegrate_dv = Jexp6(dq) = Jlog6(exp6(dq))^-1
  }
  waveLdsSync();
  if (lane < NU) {
    double sj, cj;
    sincos(s_q2[7 + lane], &sj, &cj);
    s_cs[lane][0] = cj; s_cs[lane][1] = sj;
  }
  waveLdsSync();
  if (lane < NV) {
    const int k = lane;
    double Rn[9];
    lieQuatToR(s_q2 + 3, Rn);
    const double el[3] = {k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, k == 2 ? 1.0 : 0.0};
    const double ea[3] = {k == 3 ? 1.0 : 0.0, k == 4 ? 1.0 : 0.0, k == 5 ? 1.0 : 0.0};
    Mat3<T> Rb;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      Rb.m[3 * r + 0] = T(Rn[3 * r + 0], Rn[3 * r + 1] * ea[2] - Rn[3 * r + 2] * ea[1]);
      Rb.m[3 * r + 1] = T(Rn[3 * r + 1], Rn[3 * r + 2] * ea[0] - Rn[3 * r + 0] * ea[2]);
      Rb.m[3 * r + 2] = T(Rn[3 * r + 2], Rn[3 * r + 0] * ea[1] - Rn[3 * r + 1] * ea[0]);
    }
    const Vec3<T> pb = mk<T>(T(s_q2[0], Rn[0] * el[0] + Rn[1] * el[1] + Rn[2] * el[2]), T(s_q2[1], Rn[3] * el[0] + Rn[4] * el[1] + Rn[5] * el[2]),
                             T(s_q2[2], Rn[6] * el[0] + Rn[7] * el[1] + Rn[8] * el[2]));
#pragma unroll 1
    for (int leg = 0; leg < NL; ++leg) {
      if (!nd->sw_active[leg]) continue;
      Mat3<T> Rw = Rb;
      Vec3<T> pw = pb;
#pragma unroll 1
      for (int j = 0; j < LJ; ++j) {
        const int ji = 1 + leg * LJ + j, dof = 6 + leg * LJ + j, ci = leg * LJ + j;
        const bool mine = (k == dof);
        const T cqi(s_cs[ci][0], mine ? -s_cs[ci][1] : 0.0);
        const T sqi(s_cs[ci][1], mine ? s_cs[ci][0] : 0.0);
        Mat3<T> R;
        revoluteRotation<T>(m->R[ji], m->axis[ji], cqi, sqi, R);
        const double* p = m->p[ji];
        pw = pw + mul(Rw, mk<T>(T(p[0]), T(p[1]), T(p[2])));
        Mat3<T> Rn2;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int c = 0; c < 3; ++c) Rn2.m[3 * r + c] = Rw.m[3 * r] * R.m[c] + Rw.m[3 * r + 1] * R.m[3 + c] + Rw.m[3 * r + 2] * R.m[6 + c];
        Rw = Rn2;
      }
      const double* pc = P->contact_p[leg];
      const Vec3<T> pf = pw + mul(Rw, mk<T>(T(pc[0]), T(pc[1]), T(pc[2])));
      const int row = nd->sw_row[leg];
      s_pq[row][k] = pf.x.d; s_pq[row + 1][k] = pf.y.d; s_pq[row + 2][k] = pf.z.d;
      if (k == 0) {
        s_P[row] = pf.x.v - nd->sw_point[leg][0]; s_P[row + 1] = pf.y.v - nd->sw_point[leg][1]; s_P[row + 2] = pf.z.v - nd->sw_point[leg][2];
      }
    }
  }
  waveLdsSync();
  double* __restrict__ W = B.swc + rec * L::SWC;
  if (lane < dimi) W[L::W_P + lane] = s_P[lane];
  for (int e = lane; e < dimi * NV; e += 64) {
    const int c = e / dimi, j = e - c * dimi;
    double pq, pj;
    if (c < 6) {
      pq = 0.0; pj = 0.0;
      for (int m2 = 0; m2 < 6; ++m2) { pq += s_pq[j][m2] * s_A0[m2 + 6 * c]; pj += s_pq[j][m2] * s_Je[m2 + 6 * c]; }
    } else {
      pq = s_pq[j][c]; pj = s_pq[j][c];
    }
    W[L::W_PHIX + j + NF * c] = pq;
    W[L::W_PHIX + j + NF * (NV + c)] = (dt1 + dt2) * pj;
    W[L::W_PHIA + j + NF * c] = (dt1 * dt2) * pj;
  }
}

template <typename D>
void OcpLaunch<D>::switching(const OcpBuffers& B, long batch, int M, hipStream_t st) {
  if (B.n_switch <= 0) return;
  hipLaunchKernelGGL((ocp_switch_kernel<D>), dim3((unsigned)(batch * B.n_switch)), dim3(64), 0, st, B, B.n_switch);
}

template void OcpLaunch<LeggedDims<4, 3>>::switching(const OcpBuffers&, long, int, hipStream_t);

}  // namespace idocp_dev
