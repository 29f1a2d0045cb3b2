// Inverse dynamics + contact (Baumgarte) constraint and ALL their derivatives for one stage of a legged robot:
// ONE nominal Newton-Euler sweep per leg, TANGENT-ONLY sweeps per (seed, leg) item.
//
// Replaces the rigid-body part of ContactDynamics::linearizeContactDynamics (include/idocp/ocp/contact_dynamics.hxx:48-102):
//   robot.updateKinematics(q,v,a); robot.setContactForces; robot.RNEA; robot.RNEADerivatives;
//   robot.computeBaumgarteResidual / computeBaumgarteDerivatives
// (include/idocp/robot/robot.hxx:193-203,237-279,408-500; point_contact.hxx:15-144).
//
// Round 1 carried a dual number (value, tangent) per lane through the sweep (ocp_rnea_kernel.hip): every lane recomputed the
// SAME nominal values next to its own tangent -- a third to a half of the FP64 work and half of the 256 VGPRs.  Here the nominal
// sweep runs once per leg (lane = leg) and leaves a per-joint record in LDS (rotation, motion, momenta, accumulated forces, the
// model constants of the joint); an item lane then propagates only its tangent, reading the nominal operands as LDS broadcasts:
//   d(R^T a) = R^T da - dq (u x R^T a),   d(R f) = R (df + dq u x f)        (R = P Rot(u, q): dR/dq = R skew(u))
// The world pose is not differentiated at all: the only place it enters is the position term of the Baumgarte residual, and
//   d p_foot(world) / dq_k = R_world,foot (d v_foot / d qdot_k)             (position Jacobian = velocity Jacobian in the local
// tangent), i.e. (1 / D^2) R_wf Rc J is added to dC/dq after the sweep, with J = dC/da the frame Jacobian the a-seeds emit.
// The gravity field acceleration a_gf = a - R_w^T g needs the tangent of one row of the world rotation only (dz, 3 doubles).
// Live tangent state: 15 doubles out, 21 in (was ~130): the sweep fits the 128-VGPR budget of the condensation kernel it is fused into.
//
// The Baumgarte derivative keeps THE REFERENCE'S OWN FORMULA (point_contact.hxx:117-143 adds skew(v_lin) d(omega); an exact
// derivative would subtract it) so that the Newton direction matches the reference, not just the mathematics.
#ifndef IDOCP_DEV_RNEA_TANGENT_HPP_
#define IDOCP_DEV_RNEA_TANGENT_HPP_

#include <hip/hip_runtime.h>

#include "dev_rbd.hpp"
#include "ocp_device.hpp"

namespace idocp_dev {

namespace rt {
// plain 3-vector helpers on registers
struct V3 { double x, y, z; };
__device__ __forceinline__ V3 v3(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void st3(double* p, V3 a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator*(double s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// R row-major 3 x 3
__device__ __forceinline__ V3 mul(const double* R, V3 a) {
  return v3(R[0] * a.x + R[1] * a.y + R[2] * a.z, R[3] * a.x + R[4] * a.y + R[5] * a.z, R[6] * a.x + R[7] * a.y + R[8] * a.z);
}
__device__ __forceinline__ V3 mulT(const double* R, V3 a) {
  return v3(R[0] * a.x + R[3] * a.y + R[6] * a.z, R[1] * a.x + R[4] * a.y + R[7] * a.z, R[2] * a.x + R[5] * a.y + R[8] * a.z);
}
// Y (v, w) with the inertia about the joint origin: f = m v - mc x w ; n = Io w + mc x v   (inertiaMul of dev_rbd.hpp)
__device__ __forceinline__ void inertia(double mass, V3 mc, const double* I, V3 v, V3 w, V3& f, V3& n) {
  f = mass * v - cross(mc, w);
  const V3 mcxv = cross(mc, v);
  n = v3(I[0] * w.x + I[1] * w.y + I[2] * w.z + mcxv.x, I[1] * w.x + I[3] * w.y + I[4] * w.z + mcxv.y, I[2] * w.x + I[4] * w.y + I[5] * w.z + mcxv.z);
}
// The transform of one revolute joint as the sweeps use it: R = P Rot(u, q) and the axis u.
// AX = -1: general (the 3 x 3 record and the axis vector from LDS).  AX = 0 / 1: the placement rotation P is the identity and the axis is
// +x / +y (every leg joint of ANYmal: HAA about x, HFE and KFE about y), so R^T a, R a, u x a, u . a and a x (k u) lose their zero
// terms: two numbers (cos, sin) instead of twelve come from LDS and a rotation is four multiply-adds instead of nine.
template <int AX> struct JointFrame;
template <> struct JointFrame<-1> {
  const double* R; V3 u;
  __device__ __forceinline__ explicit JointFrame(const double* jr, int jR, int jU) : R(jr + jR), u(ld3(jr + jU)) {}
  __device__ __forceinline__ V3 mulT(V3 a) const { return rt::mulT(R, a); }
  __device__ __forceinline__ V3 mul(V3 a) const { return rt::mul(R, a); }
  __device__ __forceinline__ V3 crossU(V3 a) const { return cross(u, a); }             // u x a
  __device__ __forceinline__ V3 crossUk(V3 a, double k) const { return k * cross(u, a); }      // k (u x a)
  __device__ __forceinline__ double dotU(V3 a) const { return dot(u, a); }
  __device__ __forceinline__ V3 timesU(double k) const { return k * u; }
  __device__ __forceinline__ V3 crossKU(V3 a, double k) const { return cross(a, k * u); }      // a x (k u)
};
template <> struct JointFrame<0> {      // R = [1 0 0; 0 c -s; 0 s c]
  double c, s;
  __device__ __forceinline__ explicit JointFrame(const double* jr, int jR, int) : c(jr[jR + 4]), s(jr[jR + 7]) {}
  __device__ __forceinline__ V3 mulT(V3 a) const { return v3(a.x, c * a.y + s * a.z, c * a.z - s * a.y); }
  __device__ __forceinline__ V3 mul(V3 a) const { return v3(a.x, c * a.y - s * a.z, s * a.y + c * a.z); }
  __device__ __forceinline__ V3 crossU(V3 a) const { return v3(0.0, -a.z, a.y); }
  __device__ __forceinline__ V3 crossUk(V3 a, double k) const { return v3(0.0, -(a.z * k), a.y * k); }
  __device__ __forceinline__ double dotU(V3 a) const { return a.x; }
  __device__ __forceinline__ V3 timesU(double k) const { return v3(k, 0.0, 0.0); }
  __device__ __forceinline__ V3 crossKU(V3 a, double k) const { return v3(0.0, a.z * k, -(a.y * k)); }
};
template <> struct JointFrame<1> {      // R = [c 0 s; 0 1 0; -s 0 c]
  double c, s;
  __device__ __forceinline__ explicit JointFrame(const double* jr, int jR, int) : c(jr[jR]), s(jr[jR + 2]) {}
  __device__ __forceinline__ V3 mulT(V3 a) const { return v3(c * a.x - s * a.z, a.y, s * a.x + c * a.z); }
  __device__ __forceinline__ V3 mul(V3 a) const { return v3(c * a.x + s * a.z, a.y, c * a.z - s * a.x); }
  __device__ __forceinline__ V3 crossU(V3 a) const { return v3(a.z, 0.0, -a.x); }
  __device__ __forceinline__ V3 crossUk(V3 a, double k) const { return v3(a.z * k, 0.0, -(a.x * k)); }
  __device__ __forceinline__ double dotU(V3 a) const { return a.y; }
  __device__ __forceinline__ V3 timesU(double k) const { return v3(0.0, k, 0.0); }
  __device__ __forceinline__ V3 crossKU(V3 a, double k) const { return v3(-(a.z * k), 0.0, a.x * k); }
};
template <int AX> struct AxisTag { static constexpr int value = AX; };
// Walks the joints of a leg outward (FWD) or inward with the frame type of each joint: XYY = the ANYmal pattern (joint 0 about x,
// the others about y), else the general frame for every joint.  The loops stay rolled (see dev_rbd.hpp on why).
template <bool XYY, bool FWD, int LJ, typename Body>
__device__ __forceinline__ void forEachLegJoint(Body body) {
  if constexpr (XYY) {
    if constexpr (FWD) {
      body(AxisTag<0>{}, 0);
#pragma unroll 1
      for (int j = 1; j < LJ; ++j) body(AxisTag<1>{}, j);
    } else {
#pragma unroll 1
      for (int j = LJ - 1; j >= 1; --j) body(AxisTag<1>{}, j);
      body(AxisTag<0>{}, 0);
    }
  } else {
    if constexpr (FWD) {
#pragma unroll 1
      for (int j = 0; j < LJ; ++j) body(AxisTag<-1>{}, j);
    } else {
#pragma unroll 1
      for (int j = LJ - 1; j >= 0; --j) body(AxisTag<-1>{}, j);
    }
  }
}
// keeps the scheduler (and the load hoisting in front of it) from moving anything across: the item sweeps are written phase by phase
// so that the operands of a phase are fetched only when the previous phase's are dead -- left alone, every LDS load of the item is
// issued at its top and two hundred registers spill
__device__ __forceinline__ void phaseFence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
// ... and ties a value to its place in that order: instruction selection works on the whole straight-line item as ONE block and is free
// to sink the arithmetic below every fence (it did: all loads first, all arithmetic last); a value that passes through an (empty) asm
// has to exist at that point
__device__ __forceinline__ void pin(V3& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z)); }
__device__ __forceinline__ void pin(double& a) { asm volatile("" : "+v"(a)); }
// The same walk fully unrolled, the joint index a compile-time constant (body(axis tag, integral constant)): per-joint values can
// then live in arrays that stay in registers.
template <int J> struct JointIdx { static constexpr int value = J; };
template <bool XYY, int J, typename Body>
__device__ __forceinline__ void callLegJoint(Body& body) {
  phaseFence();      // one joint step at a time
  if constexpr (XYY) { if constexpr (J == 0) body(AxisTag<0>{}, JointIdx<J>{}); else body(AxisTag<1>{}, JointIdx<J>{}); }
  else body(AxisTag<-1>{}, JointIdx<J>{});
}
template <bool XYY, bool FWD, int LJ, typename Body>
__device__ __forceinline__ void forEachLegJointUnrolled(Body body) {
  static_assert(LJ <= 4, "unrolled by hand");
  if constexpr (FWD) {
    if constexpr (LJ > 0) callLegJoint<XYY, 0>(body);
    if constexpr (LJ > 1) callLegJoint<XYY, 1>(body);
    if constexpr (LJ > 2) callLegJoint<XYY, 2>(body);
    if constexpr (LJ > 3) callLegJoint<XYY, 3>(body);
  } else {
    if constexpr (LJ > 3) callLegJoint<XYY, 3>(body);
    if constexpr (LJ > 2) callLegJoint<XYY, 2>(body);
    if constexpr (LJ > 1) callLegJoint<XYY, 1>(body);
    if constexpr (LJ > 0) callLegJoint<XYY, 0>(body);
  }
}
template <bool XYY, int J> using LegFrame = JointFrame<XYY ? (J == 0 ? 0 : 1) : -1>;
// fn(JointIdx<J>), fn(JointIdx<J - 1>), ..., fn(JointIdx<0>)
template <int J, typename Fn>
__device__ __forceinline__ void downFrom(Fn& fn) {
  fn(JointIdx<J>{});
  if constexpr (J > 0) downFrom<J - 1>(fn);
}
}  // namespace rt

// LDS scratch of one stage (doubles).  The records keep their DYNAMIC fields first, in the order of the nominal record in HBM
// (OcpLayout: O_JOINT .. O_IDC, written by ocp_nominal_kernel), so that stage 0 of the condensation kernel copies that record in
// 16-byte pieces; the model constants behind them are filled from the model.
template <typename D>
struct RneaScratch {
  using L = OcpLayout<D>;
  static constexpr int NL = D::NL, LJ = D::LJ, NV = D::NV, NQ = D::NQ, NF = D::NF, NJL = NL * LJ;
  // JOINT record, one per leg joint: rotation R = P Rot(u, q) (row-major), the child-frame motion before the joint's own velocity
  // is added (wc, vc, bwc, blc; zc = R_w,child^T e_z), the joint velocity vJ = u qd, the body's angular velocity w (its linear
  // velocity is vc), momenta hl, hn, the force accumulated up to and including this body (Fl, Fn) | axis, placement translation and
  // the inertia constants of the body
  static constexpr int J_R = 0, J_WC = 9, J_VC = 12, J_BWC = 15, J_BLC = 18, J_ZC = 21, J_VJ = 24, J_W = 27, J_HL = 30, J_HN = 33,
                       J_FL = 36, J_FN = 39, J_U = 42, J_P = 45, J_MASS = 48, J_MC = 49, J_IO = 52,
                       JREC = 58;     // the records of the four legs (3 JREC apart = 348 dwords) start in different LDS banks: the item lanes
                                      // of a wavefront read the SAME field of four different legs at once
  static_assert(J_U == L::NJ_DYN, "dynamic part of the joint record = the nominal record's");
  // FOOT record, one per leg: R_world,foot Rc (row-major), local frame velocity fv / angular velocity fw (nominal), the pose-dependent
  // part of the Baumgarte residual | the frame placement (Rc, pc) in the tip joint, contact flag and first packed row
  static constexpr int F_RWC = 0, F_FV = 9, F_FW = 12, F_CP = 15, F_RC = 18, F_PC = 27, F_ACT = 30, F_ROW = 31, FREC = 34;
  static_assert(F_RC == L::NF_DYN, "dynamic part of the foot record");
  // BASE record: z = R_w^T e_z, v, w, momenta hl, hn of the base body (+ 1 pad) | its inertia constants
  static constexpr int B_Z = 0, B_V = 3, B_W = 6, B_HL = 9, B_HN = 12, B_MASS = 16, B_MC = 17, B_IO = 20, BREC = 26;
  static_assert(B_MASS == L::NB_DYN, "dynamic part of the base record");
  static constexpr int IPL = 18 + 3 * LJ, NITEMS = NL * IPL;
  static constexpr int JOINTS = 0, FEET = JOINTS + NJL * JREC, BASE = FEET + NL * FREC,
                       BT = BASE + BREC,                       // [NITEMS][6] tangent of the force each item's leg transmits to the base
                       BOWN = BT + NITEMS * 6,                 // [18][6] tangent of the base's own inertial force, per base seed
                       BN = BOWN + 18 * 6,                     // [NL + 1][6] nominal base force: own, then per leg
                       TOTAL = BN + (NL + 1) * 6 + 2;
  static_assert(JREC % 2 == 0 && FEET % 2 == 0 && BASE % 2 == 0 && BN % 2 == 0, "16-byte pieces");
  // the nominal record in 16-byte pieces: joints, feet, base, BN, [ID; C]
  static constexpr int C_J = NJL * (L::NJ_DYN / 2), C_F = C_J + NL * (L::NF_DYN / 2), C_B = C_F + L::NB_DYN / 2, C_BN = C_B + (NL + 1) * 3,
                       C_ALL = C_BN + L::NVF / 2;
  static_assert(2 * C_ALL == L::O_IDC + L::NVF, "piece count of the nominal record");
  // piece c -> destination (doubles) in the scratch; pieces >= C_BN go to the [ID; C] vector of the caller (index 2 (c - C_BN))
  __device__ static __forceinline__ int pieceDst(int c) {
    if (c < C_J) { const int j = c / (L::NJ_DYN / 2); return JOINTS + j * JREC + 2 * (c - j * (L::NJ_DYN / 2)); }
    if (c < C_F) { const int e = c - C_J, l = e / (L::NF_DYN / 2); return FEET + l * FREC + 2 * (e - l * (L::NF_DYN / 2)); }
    if (c < C_B) return BASE + 2 * (c - C_F);
    return BN + 2 * (c - C_B);
  }
};

// Where the derivative columns go: [dID; dC] / d(q, v) as one (NV + contact rows) x 2 NV block (leading dimension ldd), dID/da = M
// (NV x NV), dC/da = J (leading dimension ldj), and the nominal [ID; C].  The LDS blocks of the condensation kernel have this shape.
struct RneaOut {
  double* didc; int ldd;     // rows: NV dynamics rows, then the packed contact rows
  double* mm;                // NV x NV
  double* jm; int ldj;
  double* idc;
};

// ---- setup, all threads: the model constants the item sweeps read go to LDS once (the sweeps then touch no global memory); the
// dynamic fields of the same records arrive from the nominal record (rneaNominalFetch / rneaNominalStore) ----
template <typename D>
__device__ __forceinline__ void rneaSetup(const DevModel* __restrict__ m, const OcpProblem* __restrict__ P, const OcpNode* __restrict__ nd,
                                          int tid, double* sc) {
  using S = RneaScratch<D>;
  using namespace rt;
  constexpr int NL = D::NL, LJ = D::LJ, NJL = NL * LJ;
  if (tid >= 128 && tid < 128 + NJL) {
    const int t = tid - 128, ji = 1 + t;
    double* jr = sc + S::JOINTS + t * S::JREC;
    st3(jr + S::J_U, ld3(m->axis[ji])); st3(jr + S::J_P, ld3(m->p[ji]));
    jr[S::J_MASS] = m->mass[ji]; st3(jr + S::J_MC, ld3(m->mc[ji]));
#pragma unroll
    for (int e = 0; e < 6; ++e) jr[S::J_IO + e] = m->Io[ji][e];
  } else if (tid >= 192 && tid < 192 + NL) {
    const int leg = tid - 192;
    double* fr = sc + S::FEET + leg * S::FREC;
#pragma unroll
    for (int e = 0; e < 9; ++e) fr[S::F_RC + e] = P->contact_R[leg][e];
    st3(fr + S::F_PC, ld3(P->contact_p[leg]));
    fr[S::F_ACT] = nd->active[leg] ? 1.0 : 0.0;
    fr[S::F_ROW] = (double)nd->row_of[leg];
  } else if (tid == 192 + NL) {
    double* br = sc + S::BASE;
    br[S::B_MASS] = m->mass[0]; st3(br + S::B_MC, ld3(m->mc[0]));
#pragma unroll
    for (int e = 0; e < 6; ++e) br[S::B_IO + e] = m->Io[0][e];
  }
}

// The nominal record of the stage (HBM, ocp_nominal_kernel) -> registers (issued with the other loads of stage 0) -> LDS scratch.
template <typename D, int NT>
struct RneaNominalCopy {
  using S = RneaScratch<D>;
  typedef double d2 __attribute__((ext_vector_type(2)));
  static constexpr int PER = (S::C_ALL + NT - 1) / NT;
  d2 r[PER];
  __device__ __forceinline__ void fetch(const double* __restrict__ nom, int tid) {
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int c = tid + NT * t;
      r[t] = *reinterpret_cast<const d2*>(nom + 2 * (c < S::C_ALL ? c : 0));
    }
  }
  __device__ __forceinline__ void store(int tid, double* sc, double* idc) const {
#pragma unroll
    for (int t = 0; t < PER; ++t) {
      const int c = tid + NT * t;
      if (c < S::C_BN) *reinterpret_cast<d2*>(sc + S::pieceDst(c)) = r[t];
      else if (c < S::C_ALL) *reinterpret_cast<d2*>(idc + 2 * (c - S::C_BN)) = r[t];
    }
  }
};

// ---- one tangent item = (seed, leg), round 3: NO inward sweep.  The force tangent of a body is formed on the way out, as soon as that
// body's motion tangent exists, and is walked back to the base at once (through the joints passed so far: j + 1 transforms for body j),
// leaving its share in the tau tangents of those joints and in the base force.  Live state: the motion tangent (30 registers), the tau
// tangents (LJ) and the base force (6 doubles).  Round 2 carried no per-body state either but UNDID the kinematic recursion on an inward
// sweep (dev_rbd.hpp's trick for chains): the whole inertia / momentum algebra sat on the inward dependency chain, a fifth more
// arithmetic, and 13 - 27 registers spilled.  Same columns as rneaTangentItemUndo up to the order of the sums. ----
// IMPULSE: the item of an impulse stage (ImpulseDynamicsForwardEuler::linearizeImpulseDynamics, impulse_dynamics_forward_euler.hxx:18-58;
// robot.hxx:283-320, 505-541).  The dynamics ImD = rnea_impulse(q, dv) are linearised at ZERO velocity without gravity, the contact
// constraint is the LOCAL linear velocity of the frame at v + dv (point_contact.hxx:145-175): the nominal record carries the
// velocity-level fields of the (v + dv) sweep next to the acceleration-level and force fields of the (0, dv) sweep (ocp_nominal_kernel),
// and the item drops everything that couples the two levels -- velocity products in the acceleration recursion and in the body forces --
// and emits d(frame velocity) as its contact rows.  A v seed then yields the frame Jacobian (dC/dv = J) and zero dynamics rows.
template <typename D, bool XYY = false, bool IMPULSE = false>
__device__ __forceinline__ void rneaTangentItem(double gz, double wv, int item, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  using namespace rt;
  constexpr int LJ = D::LJ, NV = D::NV;
  const int leg = item / S::IPL, j0 = item - leg * S::IPL;
  const bool base_seed = j0 < 18;
  const int kind = base_seed ? j0 / 6 : (j0 - 18) / LJ;                           // 0: q, 1: v, 2: a
  const int k = base_seed ? j0 - 6 * kind : 6 + leg * LJ + (j0 - 18 - LJ * kind);   // velocity index of the seed
  double* __restrict__ colp = (kind < 2) ? out.didc + (long)out.ldd * (kind * NV + k) : out.mm + NV * k;     // dynamics rows of this column
  double* __restrict__ colc = (kind < 2) ? colp + NV : out.jm + (long)out.ldj * k;                            // its contact rows
  const double* br = sc + S::BASE;
  auto e3 = [](int i) { return v3(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0); };
  const V3 zero = v3(0, 0, 0);
  // ---- base: tangents of (z, v, w, a_gf linear, a angular) ----
  V3 dz = (kind == 0 && k >= 3 && k < 6) ? cross(ld3(br + S::B_Z), e3(k - 3)) : zero;      // d(R_w^T e_z) for R_w <- R_w exp(e_ang)
  V3 dv = (kind == 1 && k < 3) ? e3(k) : zero, dw = (kind == 1 && k >= 3 && k < 6) ? e3(k - 3) : zero;
  V3 dbl = ((kind == 2 && k < 3) ? e3(k) : zero) - gz * dz, dbw = (kind == 2 && k >= 3 && k < 6) ? e3(k - 3) : zero;
  if (leg == 0 && base_seed) {
    // the base's own inertial force: once per base seed
    V3 dhl, dhn, df, dn;
    const V3 mc = ld3(br + S::B_MC);
    inertia(br[S::B_MASS], mc, br + S::B_IO, dbl, dbw, df, dn);
    double* o = sc + S::BOWN + 6 * j0;
    if constexpr (IMPULSE) {
      st3(o, df); st3(o + 3, dn);
    } else {
      const V3 w0 = ld3(br + S::B_W), v0 = ld3(br + S::B_V), hl0 = ld3(br + S::B_HL), hn0 = ld3(br + S::B_HN);
      inertia(br[S::B_MASS], mc, br + S::B_IO, dv, dw, dhl, dhn);
      st3(o, df + cross(dw, hl0) + cross(w0, dhl));
      st3(o + 3, dn + cross(dw, hn0) + cross(w0, dhn) + cross(dv, hl0) + cross(v0, dhl));
    }
  }
  double tau[LJ];
#pragma unroll
  for (int j = 0; j < LJ; ++j) tau[j] = 0.0;
  V3 baseFl = zero, baseFn = zero;
  const double* jleg = sc + S::JOINTS + leg * LJ * S::JREC;
  forEachLegJointUnrolled<XYY, true, LJ>([&](auto tag, auto jc) {
    constexpr int AX = decltype(tag)::value, j = decltype(jc)::value;
    const int dof = 6 + leg * LJ + j;
    const double* jr = jleg + j * S::JREC;
    const bool mine = (k == dof);
    const double sq = (mine && kind == 0) ? 1.0 : 0.0, sv = (mine && kind == 1) ? 1.0 : 0.0, sa = (mine && kind == 2) ? 1.0 : 0.0;
    const JointFrame<AX> F(jr, S::J_R, S::J_U);
    // ---- phase 1: the motion tangent of this body ----
    {
      const V3 p = ld3(jr + S::J_P);
      const V3 wj = ld3(jr + S::J_W), vj = ld3(jr + S::J_VC);
      const V3 dwc = F.mulT(dw) - F.crossUk(ld3(jr + S::J_WC), sq);
      const V3 dvc = F.mulT(dv + cross(dw, p)) - F.crossUk(vj, sq);
      const V3 dbwc = F.mulT(dbw) - F.crossUk(ld3(jr + S::J_BWC), sq);
      const V3 dblc = F.mulT(dbl + cross(dbw, p)) - F.crossUk(ld3(jr + S::J_BLC), sq);
      dz = F.mulT(dz) - F.crossUk(ld3(jr + S::J_ZC), sq);
      dw = dwc + F.timesU(sv); dv = dvc;
      if constexpr (IMPULSE) {
        dbw = dbwc + F.timesU(sa); dbl = dblc;          // (the acceleration level sees no velocity)
      } else if constexpr (AX < 0) {
        const V3 vJ = ld3(jr + S::J_VJ), dvJ = F.timesU(sv);
        dbw = dbwc + F.timesU(sa) + cross(dw, vJ) + cross(wj, dvJ);
        dbl = dblc + cross(dv, vJ) + cross(vj, dvJ);
      } else {
        const double qd = jr[S::J_VJ + (AX < 0 ? 0 : AX)];      // S qd = qd u (coordinate axis: one component)
        dbw = dbwc + F.timesU(sa) + F.crossKU(dw, qd) + F.crossKU(wj, sv);
        dbl = dblc + F.crossKU(dv, qd) + F.crossKU(vj, sv);
      }
    }
    pin(dw); pin(dv); pin(dbw); pin(dbl); pin(dz);
    phaseFence();
    // ---- phase 2: d(I a + w x I v) of this body: inertia products first (their operands are dead before the momenta are fetched) ----
    V3 Fl, Fn;
    {
      if constexpr (IMPULSE) {
        inertia(jr[S::J_MASS], ld3(jr + S::J_MC), jr + S::J_IO, dbl, dbw, Fl, Fn);
      } else {
        V3 dhl, dhn;
        {
          const V3 mc = ld3(jr + S::J_MC);
          inertia(jr[S::J_MASS], mc, jr + S::J_IO, dv, dw, dhl, dhn);
          inertia(jr[S::J_MASS], mc, jr + S::J_IO, dbl, dbw, Fl, Fn);
        }
        pin(dhl); pin(dhn); pin(Fl); pin(Fn);
        phaseFence();
        const V3 wj = ld3(jr + S::J_W), vj = ld3(jr + S::J_VC), hl = ld3(jr + S::J_HL), hn = ld3(jr + S::J_HN);
        Fl = Fl + cross(dw, hl) + cross(wj, dhl);
        Fn = Fn + cross(dw, hn) + cross(wj, dhn) + cross(dv, hl) + cross(vj, dhl);
      }
    }
    pin(Fl); pin(Fn);
    phaseFence();
    // ---- phase 3: walk the force back to the base through joints j .. 0 ----
    auto hop = [&](auto jjc) {
      constexpr int jj = decltype(jjc)::value;
      const double* jq = jleg + jj * S::JREC;
      const LegFrame<XYY, jj> G(jq, S::J_R, S::J_U);
      tau[jj] += G.dotU(Fn);
      if constexpr (jj == j) {      // the joint's own transform turns with a q seed: + d(R)/dq applied to the nominal accumulated force
        Fl = Fl + G.crossUk(ld3(jq + S::J_FL), sq);
        Fn = Fn + G.crossUk(ld3(jq + S::J_FN), sq);
      }
      const V3 Rf = G.mul(Fl);
      Fn = G.mul(Fn) + cross(ld3(jq + S::J_P), Rf);
      Fl = Rf;
    };
    downFrom<j>(hop);
    baseFl = baseFl + Fl; baseFn = baseFn + Fn;
    pin(baseFl); pin(baseFn);
#pragma unroll
    for (int t = 0; t <= j; ++t) pin(tau[t]);
  });
  phaseFence();
  // ---- contact frame at the foot: Baumgarte derivative column (point_contact.hxx:117-143) without its position term ----
  const double* fr = sc + S::FEET + leg * S::FREC;
  if (fr[S::F_ACT] != 0.0) {
    const double* Rc = fr + S::F_RC;
    const V3 pc = ld3(fr + S::F_PC);
    const V3 dal = dbl + gz * dz;
    // (XYY also promises an identity rotation of the contact frame in its joint, like ANYmal's feet: Rc^T a = a)
    const V3 dfv = XYY ? dv + cross(dw, pc) : mulT(Rc, dv + cross(dw, pc));
    if constexpr (IMPULSE) {
      st3(colc + (int)fr[S::F_ROW], dfv);            // contact-velocity constraint: d(local linear velocity of the frame)
    } else {
      const V3 dfw = XYY ? dw : mulT(Rc, dw), dfa = XYY ? dal + cross(dbw, pc) : mulT(Rc, dal + cross(dbw, pc));
      const V3 dc = dfa + cross(ld3(fr + S::F_FW), dfv) + cross(ld3(fr + S::F_FV), dfw) + wv * dfv;
      st3(colc + (int)fr[S::F_ROW], dc);
    }
  }
#pragma unroll
  for (int j = 0; j < LJ; ++j) colp[6 + leg * LJ + j] = tau[j];
  double* o = sc + S::BT + 6 * item;
  st3(o, baseFl); st3(o + 3, baseFn);
}

// ---- the round-2 form of the item (no per-body state, the kinematic steps undone on the way in); kept for comparison
// (IDOCP_ITEM_UNDO at build time selects it) ----
template <typename D, bool XYY = false>
__device__ __forceinline__ void rneaTangentItemUndo(double gz, double wv, int item, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  using namespace rt;
  constexpr int LJ = D::LJ, NV = D::NV;
  const int leg = item / S::IPL, j0 = item - leg * S::IPL;
  const bool base_seed = j0 < 18;
  const int kind = base_seed ? j0 / 6 : (j0 - 18) / LJ;                           // 0: q, 1: v, 2: a
  const int k = base_seed ? j0 - 6 * kind : 6 + leg * LJ + (j0 - 18 - LJ * kind);   // velocity index of the seed
  double* __restrict__ colp = (kind < 2) ? out.didc + (long)out.ldd * (kind * NV + k) : out.mm + NV * k;     // dynamics rows of this column
  double* __restrict__ colc = (kind < 2) ? colp + NV : out.jm + (long)out.ldj * k;                            // its contact rows
  const double* br = sc + S::BASE;
  auto e3 = [](int i) { return v3(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0); };
  const V3 zero = v3(0, 0, 0);
  // ---- base: tangents of (z, v, w, a_gf linear, a angular) ----
  V3 dz = (kind == 0 && k >= 3 && k < 6) ? cross(ld3(br + S::B_Z), e3(k - 3)) : zero;      // d(R_w^T e_z) for R_w <- R_w exp(e_ang)
  V3 dv = (kind == 1 && k < 3) ? e3(k) : zero, dw = (kind == 1 && k >= 3 && k < 6) ? e3(k - 3) : zero;
  V3 dbl = ((kind == 2 && k < 3) ? e3(k) : zero) - gz * dz, dbw = (kind == 2 && k >= 3 && k < 6) ? e3(k - 3) : zero;
  if (leg == 0 && base_seed) {
    // the base's own inertial force: once per base seed
    V3 dhl, dhn, df, dn;
    const V3 mc = ld3(br + S::B_MC), w0 = ld3(br + S::B_W), v0 = ld3(br + S::B_V), hl0 = ld3(br + S::B_HL), hn0 = ld3(br + S::B_HN);
    inertia(br[S::B_MASS], mc, br + S::B_IO, dv, dw, dhl, dhn);
    inertia(br[S::B_MASS], mc, br + S::B_IO, dbl, dbw, df, dn);
    double* o = sc + S::BOWN + 6 * j0;
    st3(o, df + cross(dw, hl0) + cross(w0, dhl));
    st3(o + 3, dn + cross(dw, hn0) + cross(w0, dhn) + cross(dv, hl0) + cross(v0, dhl));
  }
  // ---- outward along the leg ----
  forEachLegJoint<XYY, true, LJ>([&](auto tag, int j) {
    constexpr int AX = decltype(tag)::value;
    const int dof = 6 + leg * LJ + j;
    const double* jr = sc + S::JOINTS + (leg * LJ + j) * S::JREC;
    const bool mine = (k == dof);
    const double sq = (mine && kind == 0) ? 1.0 : 0.0, sv = (mine && kind == 1) ? 1.0 : 0.0, sa = (mine && kind == 2) ? 1.0 : 0.0;
    const JointFrame<AX> F(jr, S::J_R, S::J_U);
    const V3 p = ld3(jr + S::J_P);
    const double qd = AX < 0 ? 0.0 : jr[S::J_VJ + (AX < 0 ? 0 : AX)];      // S qd = qd u (coordinate axis: one component)
    const V3 dwc = F.mulT(dw) - F.crossUk(ld3(jr + S::J_WC), sq);
    const V3 dvc = F.mulT(dv + cross(dw, p)) - F.crossUk(ld3(jr + S::J_VC), sq);
    const V3 dbwc = F.mulT(dbw) - F.crossUk(ld3(jr + S::J_BWC), sq);
    const V3 dblc = F.mulT(dbl + cross(dbw, p)) - F.crossUk(ld3(jr + S::J_BLC), sq);
    dz = F.mulT(dz) - F.crossUk(ld3(jr + S::J_ZC), sq);
    dw = dwc + F.timesU(sv); dv = dvc;
    if constexpr (AX < 0) {
      const V3 vJ = ld3(jr + S::J_VJ), dvJ = F.timesU(sv);
      dbw = dbwc + F.timesU(sa) + cross(dw, vJ) + cross(ld3(jr + S::J_W), dvJ);
      dbl = dblc + cross(dv, vJ) + cross(ld3(jr + S::J_VC), dvJ);
    } else {
      dbw = dbwc + F.timesU(sa) + F.crossKU(dw, qd) + F.crossKU(ld3(jr + S::J_W), sv);
      dbl = dblc + F.crossKU(dv, qd) + F.crossKU(ld3(jr + S::J_VC), sv);
    }
  });
  // ---- contact frame at the foot: Baumgarte derivative column (point_contact.hxx:117-143) without its position term ----
  const double* fr = sc + S::FEET + leg * S::FREC;
  if (fr[S::F_ACT] != 0.0) {
    const double* Rc = fr + S::F_RC;
    const V3 pc = ld3(fr + S::F_PC);
    const V3 dal = dbl + gz * dz;
    // (XYY also promises an identity rotation of the contact frame in its joint, like ANYmal's feet: Rc^T a = a)
    const V3 dfv = XYY ? dv + cross(dw, pc) : mulT(Rc, dv + cross(dw, pc)), dfw = XYY ? dw : mulT(Rc, dw),
             dfa = XYY ? dal + cross(dbw, pc) : mulT(Rc, dal + cross(dbw, pc));
    const V3 dc = dfa + cross(ld3(fr + S::F_FW), dfv) + cross(ld3(fr + S::F_FV), dfw) + wv * dfv;
    st3(colc + (int)fr[S::F_ROW], dc);
  }
  // ---- inward sweep: force tangents, tau tangents, undo the kinematic steps ----
  V3 dFl = zero, dFn = zero;
  forEachLegJoint<XYY, false, LJ>([&](auto tag, int j) {
    constexpr int AX = decltype(tag)::value;
    const int dof = 6 + leg * LJ + j;
    const double* jr = sc + S::JOINTS + (leg * LJ + j) * S::JREC;
    const bool mine = (k == dof);
    const double sq = (mine && kind == 0) ? 1.0 : 0.0, sv = (mine && kind == 1) ? 1.0 : 0.0, sa = (mine && kind == 2) ? 1.0 : 0.0;
    const JointFrame<AX> F(jr, S::J_R, S::J_U);
    const V3 p = ld3(jr + S::J_P), mc = ld3(jr + S::J_MC);
    const V3 wj = ld3(jr + S::J_W), vj = ld3(jr + S::J_VC), hl = ld3(jr + S::J_HL), hn = ld3(jr + S::J_HN);
    V3 dhl, dhn, df, dn;
    inertia(jr[S::J_MASS], mc, jr + S::J_IO, dv, dw, dhl, dhn);
    inertia(jr[S::J_MASS], mc, jr + S::J_IO, dbl, dbw, df, dn);
    dFl = dFl + df + cross(dw, hl) + cross(wj, dhl);
    dFn = dFn + dn + cross(dw, hn) + cross(wj, dhn) + cross(dv, hl) + cross(vj, dhl);
    colp[dof] = F.dotU(dFn);
    const V3 dRf = F.mul(dFl + F.crossUk(ld3(jr + S::J_FL), sq));
    dFn = F.mul(dFn + F.crossUk(ld3(jr + S::J_FN), sq)) + cross(p, dRf);
    dFl = dRf;
    if (j > 0) {
      V3 dbwc, dblc;
      if constexpr (AX < 0) {
        const V3 vJ = ld3(jr + S::J_VJ), dvJ = F.timesU(sv);
        dbwc = dbw - F.timesU(sa) - cross(dw, vJ) - cross(wj, dvJ);
        dblc = dbl - cross(dv, vJ) - cross(vj, dvJ);
      } else {
        const double qd = jr[S::J_VJ + (AX < 0 ? 0 : AX)];
        dbwc = dbw - F.timesU(sa) - F.crossKU(dw, qd) - F.crossKU(wj, sv);
        dblc = dbl - F.crossKU(dv, qd) - F.crossKU(vj, sv);
      }
      const V3 dwc = dw - F.timesU(sv);
      dw = F.mul(dwc + F.crossUk(ld3(jr + S::J_WC), sq));
      dv = F.mul(dv + F.crossUk(ld3(jr + S::J_VC), sq)) - cross(dw, p);
      dbw = F.mul(dbwc + F.crossUk(ld3(jr + S::J_BWC), sq));
      dbl = F.mul(dblc + F.crossUk(ld3(jr + S::J_BLC), sq)) - cross(dbw, p);
    }
  });
  double* o = sc + S::BT + 6 * item;
  st3(o, dFl); st3(o + 3, dFn);
}

// ---- the acceleration seeds on their own: with dv = dw = dz = 0 only (dbl, dbw) travel, a third of the full item's work.
// Their columns are dID/da = M (the joint-space inertia matrix) and dC/da = J (the contact Jacobian) -- all that
// Robot::computeMJtJinv needs, so the wavefront that runs these items goes straight on to the inverses while another
// wavefront is still busy with the q and v seeds. ----
template <typename D, bool XYY = false>
__device__ __forceinline__ void rneaTangentItemA(int item, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  using namespace rt;
  constexpr int LJ = D::LJ, NV = D::NV;
  const int leg = item / S::IPL, j0 = item - leg * S::IPL;
  const bool base_seed = j0 < 18;
  const int k = base_seed ? j0 - 12 : 6 + leg * LJ + (j0 - 18 - 2 * LJ);      // velocity index of the seed (kind = 2)
  double* __restrict__ colp = out.mm + NV * k;
  double* __restrict__ colc = out.jm + (long)out.ldj * k;
  const double* br = sc + S::BASE;
  auto e3 = [](int i) { return v3(i == 0 ? 1.0 : 0.0, i == 1 ? 1.0 : 0.0, i == 2 ? 1.0 : 0.0); };
  const V3 zero = v3(0, 0, 0);
  V3 dbl = (k < 3) ? e3(k) : zero, dbw = (k >= 3 && k < 6) ? e3(k - 3) : zero;
  if (leg == 0 && base_seed) {
    V3 df, dn;
    inertia(br[S::B_MASS], ld3(br + S::B_MC), br + S::B_IO, dbl, dbw, df, dn);
    double* o = sc + S::BOWN + 6 * j0;
    st3(o, df); st3(o + 3, dn);
  }
  forEachLegJoint<XYY, true, LJ>([&](auto tag, int j) {
    constexpr int AX = decltype(tag)::value;
    const int dof = 6 + leg * LJ + j;
    const double* jr = sc + S::JOINTS + (leg * LJ + j) * S::JREC;
    const JointFrame<AX> F(jr, S::J_R, S::J_U);
    const double sa = (k == dof) ? 1.0 : 0.0;
    const V3 t = dbl + cross(dbw, ld3(jr + S::J_P));
    dbw = F.mulT(dbw) + F.timesU(sa);
    dbl = F.mulT(t);
  });
  const double* fr = sc + S::FEET + leg * S::FREC;
  if (fr[S::F_ACT] != 0.0) {                                     // column of the frame Jacobian
    const V3 t = dbl + cross(dbw, ld3(fr + S::F_PC));
    st3(colc + (int)fr[S::F_ROW], XYY ? t : mulT(fr + S::F_RC, t));
  }
  V3 dFl = zero, dFn = zero;
  forEachLegJoint<XYY, false, LJ>([&](auto tag, int j) {
    constexpr int AX = decltype(tag)::value;
    const int dof = 6 + leg * LJ + j;
    const double* jr = sc + S::JOINTS + (leg * LJ + j) * S::JREC;
    const JointFrame<AX> F(jr, S::J_R, S::J_U);
    const V3 p = ld3(jr + S::J_P);
    V3 df, dn;
    inertia(jr[S::J_MASS], ld3(jr + S::J_MC), jr + S::J_IO, dbl, dbw, df, dn);
    dFl = dFl + df; dFn = dFn + dn;
    colp[dof] = F.dotU(dFn);
    const V3 dRf = F.mul(dFl);
    dFn = F.mul(dFn) + cross(p, dRf);
    dFl = dRf;
    if (j > 0) {
      const double sa = (k == dof) ? 1.0 : 0.0;
      dbw = F.mul(dbw - F.timesU(sa));
      dbl = F.mul(dbl) - cross(dbw, p);
    }
  });
  double* o = sc + S::BT + 6 * item;
  st3(o, dFl); st3(o + 3, dFn);
}

// item lists (positions in the (leg, j0) numbering of BT: j0 < 18 base seeds by kind, then the leg's joint seeds by kind):
//   q / v seeds: per leg 3 base orientation seeds (a base POSITION seed changes nothing but the position term, which rneaAssembleQV
//   adds), 6 base velocity seeds, LJ joint angle and LJ joint velocity seeds -- (9 + 2 LJ) NL items, 60 for a quadruped: one wavefront
//   a seeds: per leg 6 base + LJ joint seeds -- (6 + LJ) NL items
template <typename D> struct RneaItems {
  static constexpr int LJ = D::LJ, NL = D::NL, QV_PER_LEG = 9 + 2 * LJ, NQV = NL * QV_PER_LEG, A_PER_LEG = 6 + LJ, NA = NL * A_PER_LEG;
  using S = RneaScratch<D>;
  __device__ static __forceinline__ int qv(int idx) {
    const int leg = idx / QV_PER_LEG, e = idx - leg * QV_PER_LEG;
    const int j0 = e < 3 ? 3 + e : (e < 9 ? 6 + (e - 3) : (e < 9 + LJ ? 18 + (e - 9) : 18 + LJ + (e - 9 - LJ)));
    return leg * S::IPL + j0;
  }
  __device__ static __forceinline__ int a(int idx) {
    const int leg = idx / A_PER_LEG, e = idx - leg * A_PER_LEG;
    return leg * S::IPL + (e < 6 ? 12 + e : 18 + 2 * LJ + (e - 6));
  }
};

// ---- after the a items, by the SAME wavefront (lane = 0 .. 63, no workgroup barrier): base rows of M ----
//   tau[0:6] = total spatial force on the base (S = identity): own term + legs, in leg order
template <typename D>
__device__ __forceinline__ void rneaAssembleA(int lane, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  constexpr int NL = D::NL, LJ = D::LJ, NV = D::NV;
  for (int e = lane; e < NV * 6; e += 64) {
    const int k = e / 6, r = e - 6 * k;
    double acc;
    if (k < 6) {
      acc = sc[S::BOWN + 6 * (12 + k) + r];
      for (int leg = 0; leg < NL; ++leg) acc += sc[S::BT + 6 * (leg * S::IPL + 12 + k) + r];
    } else {
      const int leg = (k - 6) / LJ;
      acc = sc[S::BT + 6 * (leg * S::IPL + 18 + 2 * LJ + (k - 6 - leg * LJ)) + r];
    }
    out.mm[r + NV * k] = acc;
  }
}

// ---- nominal values only (the line search evaluates residuals, no derivatives): base rows of ID and the pose part of C ----
template <typename D>
__device__ __forceinline__ void rneaAssembleNominal(int tid, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  constexpr int NL = D::NL, NV = D::NV, NC = D::NC;
  if (tid < 6) {
    double acc = sc[S::BN + tid];
    for (int leg = 0; leg < NL; ++leg) acc += sc[S::BN + 6 * (1 + leg) + tid];
    out.idc[tid] = acc;
  }
  if (tid >= 64 && tid < 64 + 3 * NC) {
    const int c = (tid - 64) / 3, x = tid - 64 - 3 * c;
    const double* fr = sc + S::FEET + c * S::FREC;
    if (fr[S::F_ACT] != 0.0) out.idc[NV + (int)fr[S::F_ROW] + x] += fr[S::F_CP + x];
  }
}

// ---- after the q / v items, all threads: base rows of the q and v columns, nominal base rows, and the pose-dependent terms ----
//   dC/dq[contact rows of leg c, :] += (1 / D^2) (R_wf Rc) J[contact rows of leg c, :]
template <typename D>
__device__ __forceinline__ void rneaAssembleQV(double wp, int tid, int nt, double* sc, const RneaOut& out) {
  using S = RneaScratch<D>;
  constexpr int NL = D::NL, LJ = D::LJ, NV = D::NV, NC = D::NC;
  for (int e = tid; e < 2 * NV * 6; e += nt) {
    const int c = e / 6, r = e - 6 * c, kind = c / NV, k = c - kind * NV;
    if (kind == 0 && k < 3) continue;                      // base position seeds: zero columns (zero-filled by the caller)
    double acc;
    if (k < 6) {
      acc = sc[S::BOWN + 6 * (kind * 6 + k) + r];
      for (int leg = 0; leg < NL; ++leg) acc += sc[S::BT + 6 * (leg * S::IPL + kind * 6 + k) + r];
    } else {
      const int leg = (k - 6) / LJ;
      acc = sc[S::BT + 6 * (leg * S::IPL + 18 + LJ * kind + (k - 6 - leg * LJ)) + r];
    }
    out.didc[r + (long)out.ldd * c] = acc;
  }
  if (tid < 6) {
    double acc = sc[S::BN + tid];
    for (int leg = 0; leg < NL; ++leg) acc += sc[S::BN + 6 * (1 + leg) + tid];
    out.idc[tid] = acc;
  }
  // nominal residual: + the pose-dependent part left by rneaNominalPose
  if (tid >= 64 && tid < 64 + 3 * NC) {
    const int c = (tid - 64) / 3, x = tid - 64 - 3 * c;
    const double* fr = sc + S::FEET + c * S::FREC;
    if (fr[S::F_ACT] != 0.0) out.idc[NV + (int)fr[S::F_ROW] + x] += fr[S::F_CP + x];
  }
  for (int e = tid; e < NC * 3 * NV; e += nt) {
    const int k = e / (3 * NC), rem = e - k * 3 * NC, c = rem / 3, x = rem - 3 * c;
    const double* fr = sc + S::FEET + c * S::FREC;
    if (fr[S::F_ACT] == 0.0) continue;
    const int row = (int)fr[S::F_ROW];
    const double* rwc = fr + S::F_RWC + 3 * x;
    const double* jc = out.jm + row + (long)out.ldj * k;
    out.didc[NV + row + x + (long)out.ldd * k] += wp * (rwc[0] * jc[0] + rwc[1] * jc[1] + rwc[2] * jc[2]);
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_RNEA_TANGENT_HPP_
