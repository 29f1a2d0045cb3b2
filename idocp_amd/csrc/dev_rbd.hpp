// Device-side rigid-body arithmetic for gfx950.
//
// Design: the derivatives of inverse dynamics that the
// reference obtains from pinocchio::computeRNEADerivatives
// (include/idocp/robot/robot.hxx:466-500) are produced here by forward-mode
// differentiation of a body-frame recursive Newton-Euler sweep, ONE TANGENT SEED
// PER LANE: lane (kind, k) carries d/dq_k, d/dv_k or d/da_k through the whole
// recursion in registers.  All lanes of a wavefront execute the identical
// instruction stream (no divergence, no LDS traffic, no cross-lane exchange),
// which suits the 64-wide CDNA4 wavefront far better than the irregular,
// 6-wide spatial-algebra steps of the analytic world-frame algorithm the CPU
// reference runs.  The result is the same mathematical derivative (exact, not
// finite differences); the CPU oracle uses the analytic algorithm, so the two
// are independent derivations checked against each other.
#ifndef IDOCP_DEV_RBD_HPP_
#define IDOCP_DEV_RBD_HPP_

#include <hip/hip_runtime.h>

#include "idocp_hip.h"

namespace idocp_dev {

// Model constants in device memory, read through uniform (scalar) loads.
// Inertia is stored about the JOINT-FRAME ORIGIN:  Y = [[m 1, -[mc]x],[[mc]x, Io]].
struct DevModel {
  int njoints, nq, nv, nu, has_floating_base;
  int parent[IDOCP_MAX_JOINTS];
  int jtype[IDOCP_MAX_JOINTS];
  int idx_q[IDOCP_MAX_JOINTS];
  int idx_v[IDOCP_MAX_JOINTS];
  double axis[IDOCP_MAX_JOINTS][3];
  double R[IDOCP_MAX_JOINTS][9];
  double p[IDOCP_MAX_JOINTS][3];
  double mass[IDOCP_MAX_JOINTS];
  double mc[IDOCP_MAX_JOINTS][3];
  double Io[IDOCP_MAX_JOINTS][6];   // xx, xy, xz, yy, yz, zz
  double gravity[3];
};

// ------------------------------------------------------------------ dual ----
struct Dual {
  double v, d;
  __device__ __forceinline__ Dual() {}
  __device__ __forceinline__ Dual(double v_) : v(v_), d(0.0) {}
  __device__ __forceinline__ Dual(double v_, double d_) : v(v_), d(d_) {}
};
__device__ __forceinline__ Dual operator+(Dual a, Dual b) { return Dual(a.v + b.v, a.d + b.d); }
__device__ __forceinline__ Dual operator-(Dual a, Dual b) { return Dual(a.v - b.v, a.d - b.d); }
__device__ __forceinline__ Dual operator-(Dual a) { return Dual(-a.v, -a.d); }
__device__ __forceinline__ Dual operator*(Dual a, Dual b) { return Dual(a.v * b.v, a.v * b.d + a.d * b.v); }
__device__ __forceinline__ Dual operator*(double a, Dual b) { return Dual(a * b.v, a * b.d); }
__device__ __forceinline__ Dual operator*(Dual a, double b) { return Dual(a.v * b, a.d * b); }
__device__ __forceinline__ Dual operator+(Dual a, double b) { return Dual(a.v + b, a.d); }
__device__ __forceinline__ Dual operator+(double a, Dual b) { return Dual(a + b.v, b.d); }
__device__ __forceinline__ Dual operator-(Dual a, double b) { return Dual(a.v - b, a.d); }
__device__ __forceinline__ Dual operator-(double a, Dual b) { return Dual(a - b.v, -b.d); }
__device__ __forceinline__ double value(double a) { return a; }
__device__ __forceinline__ double value(Dual a) { return a.v; }
__device__ __forceinline__ double tangent(double) { return 0.0; }
__device__ __forceinline__ double tangent(Dual a) { return a.d; }

template <typename T> struct Vec3 { T x, y, z; };
template <typename T> __device__ __forceinline__ Vec3<T> mk(T x, T y, T z) { Vec3<T> r; r.x = x; r.y = y; r.z = z; return r; }
template <typename T> __device__ __forceinline__ Vec3<T> operator+(Vec3<T> a, Vec3<T> b) { return mk<T>(a.x + b.x, a.y + b.y, a.z + b.z); }
template <typename T> __device__ __forceinline__ Vec3<T> operator-(Vec3<T> a, Vec3<T> b) { return mk<T>(a.x - b.x, a.y - b.y, a.z - b.z); }
template <typename T> __device__ __forceinline__ Vec3<T> operator*(T s, Vec3<T> a) { return mk<T>(s * a.x, s * a.y, s * a.z); }
template <typename T> __device__ __forceinline__ Vec3<T> cross(Vec3<T> a, Vec3<T> b) {
  return mk<T>(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
template <typename T> __device__ __forceinline__ T dot(Vec3<T> a, Vec3<T> b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// constant (double) vector ops against T vectors
template <typename T> __device__ __forceinline__ Vec3<T> crossC(const double* c, Vec3<T> b) {   // c x b
  return mk<T>(c[1] * b.z - c[2] * b.y, c[2] * b.x - c[0] * b.z, c[0] * b.y - c[1] * b.x);
}
template <typename T> __device__ __forceinline__ Vec3<T> crossVC(Vec3<T> a, const double* c) {  // a x c
  return mk<T>(a.y * c[2] - a.z * c[1], a.z * c[0] - a.x * c[2], a.x * c[1] - a.y * c[0]);
}

template <typename T> struct Mat3 { T m[9]; };   // row-major
template <typename T> __device__ __forceinline__ Vec3<T> mul(const Mat3<T>& R, Vec3<T> a) {
  return mk<T>(R.m[0] * a.x + R.m[1] * a.y + R.m[2] * a.z, R.m[3] * a.x + R.m[4] * a.y + R.m[5] * a.z,
               R.m[6] * a.x + R.m[7] * a.y + R.m[8] * a.z);
}
template <typename T> __device__ __forceinline__ Vec3<T> mulT(const Mat3<T>& R, Vec3<T> a) {
  return mk<T>(R.m[0] * a.x + R.m[3] * a.y + R.m[6] * a.z, R.m[1] * a.x + R.m[4] * a.y + R.m[7] * a.z,
               R.m[2] * a.x + R.m[5] * a.y + R.m[8] * a.z);
}

// liMi rotation of a revolute joint: R = P * Rot(axis, q), from cached cos/sin.
template <typename T, bool ZAX = false>
__device__ __forceinline__ void revoluteRotation(const double* __restrict__ P, const double* __restrict__ u, T c, T s, Mat3<T>& R) {
  // joint axis = +z (every joint of the iiwa14 chain; ZAX: known at compile time, else a uniform branch on scalar loads): Rot(z, q) only
  // mixes the first two columns of P -- 12 multiply-adds instead of ~70
  if (ZAX || (u[0] == 0.0 && u[1] == 0.0 && u[2] == 1.0)) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      R.m[3 * i] = P[3 * i] * c + P[3 * i + 1] * s;
      R.m[3 * i + 1] = P[3 * i + 1] * c - P[3 * i] * s;
      R.m[3 * i + 2] = T(P[3 * i + 2]);
    }
    return;
  }
  const T t = 1.0 - c;
  T J[9];
  J[0] = c + t * (u[0] * u[0]);        J[1] = t * (u[0] * u[1]) - s * u[2]; J[2] = t * (u[0] * u[2]) + s * u[1];
  J[3] = t * (u[1] * u[0]) + s * u[2]; J[4] = c + t * (u[1] * u[1]);        J[5] = t * (u[1] * u[2]) - s * u[0];
  J[6] = t * (u[2] * u[0]) - s * u[1]; J[7] = t * (u[2] * u[1]) + s * u[0]; J[8] = c + t * (u[2] * u[2]);
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) R.m[3 * i + j] = P[3 * i] * J[j] + P[3 * i + 1] * J[3 + j] + P[3 * i + 2] * J[6 + j];
}

// Y * (v, w) with Y about the joint origin: f = m v - mc x w ; n = Io w + mc x v
template <typename T>
__device__ __forceinline__ void inertiaMul(const DevModel* __restrict__ m, int i, Vec3<T> v, Vec3<T> w, Vec3<T>& f, Vec3<T>& n) {
  const double* mc = m->mc[i];
  const double* I = m->Io[i];
  const Vec3<T> mcxw = crossC<T>(mc, w);
  const Vec3<T> mcxv = crossC<T>(mc, v);
  f = mk<T>(m->mass[i] * v.x - mcxw.x, m->mass[i] * v.y - mcxw.y, m->mass[i] * v.z - mcxw.z);
  n = mk<T>(I[0] * w.x + I[1] * w.y + I[2] * w.z + mcxv.x, I[1] * w.x + I[3] * w.y + I[4] * w.z + mcxv.y,
            I[2] * w.x + I[4] * w.y + I[5] * w.z + mcxv.z);
}

// Inverse dynamics of a fixed-base SERIAL CHAIN of NJ revolute joints,
// tau = ID(q, v, a), with the scalar type T = double or Dual.  The joint-angle
// tangent enters through (cq, sq) = cos/sin as Duals built by the caller.
//
// Same recursion as pinocchio::rnea (call site robot.hxx:454; Featherstone RBDA
// table 5.1), arranged so that NO per-joint state is kept between the sweeps:
// the outward sweep carries only the current body's spatial velocity and
// acceleration to the tip; the inward sweep accumulates the force, emits
// tau_i = S_i . f_i and UNDOES the kinematic recursion step (v_{i-1} =
// X_i^{-1}(v_i - S_i qd_i), likewise for a) to recover the parent's motion.
// The live state is ~40 doubles per lane instead of ~130, which is what lets
// the one-tangent-per-lane scheme run without spilling; the inverse transform
// is exact up to rounding (orthonormal rotations), costing ~1e-16 relative.
//
// Inputs are NOMINAL values only -- cs[i] = {cos q_i, sin q_i} (an LDS table the
// stage group shares), qd, qdd -- and the tangent seed (kind, k): kind 0/1/2 =
// d/dq_k, d/dv_k, d/da_k.  The seed's unit tangents are rebuilt at the point of
// use instead of being carried in registers.  Outputs go straight to memory
// (LDS in the kernels): tau_d[i] = d tau_i / d seed, and the nominal tau_v[i]
// from the lane with write_nominal set.
//
// The joint loops are deliberately NOT unrolled: unrolled, the scheduler hoists
// every joint's rotation and model constants to the top of a 20k-instruction
// block (450+ VGPRs, 160 KB of code, far beyond the instruction cache); rolled,
// the body is ~1.5k instructions and the live state is the recursion state.
// ZAX: every joint axis is +z (checked by the host, unocp_capi.hip): S qd = (0, 0, qd), so the joint-velocity products and the
// projection tau_i = S_i . f_i lose their zero terms at compile time.
template <int NJ, bool ZAX = false>
__device__ __forceinline__ void rneaChain(const DevModel* m, const double* cs, const double* qdn,
                                          const double* qddn, int kind, int k, bool write_nominal,
                                          double* tau_v, double* tau_d) {
  typedef Dual T;
  Vec3<T> v = mk<T>(0.0, 0.0, 0.0), w = v, bw = v;
  Vec3<T> bl = mk<T>(-m->gravity[0], -m->gravity[1], -m->gravity[2]);
#pragma unroll 1
  for (int i = 0; i < NJ; ++i) {
    const bool mine = (k == i);
    const T cqi(cs[2 * i], (mine && kind == 0) ? -cs[2 * i + 1] : 0.0);
    const T sqi(cs[2 * i + 1], (mine && kind == 0) ? cs[2 * i] : 0.0);
    const T qdi(qdn[i], (mine && kind == 1) ? 1.0 : 0.0);
    const T qddi(qddn[i], (mine && kind == 2) ? 1.0 : 0.0);
    Mat3<T> R;
    revoluteRotation<T, ZAX>(m->R[i], m->axis[i], cqi, sqi, R);
    const double* p = m->p[i];
    const double* u = m->axis[i];
    // motion transform into the child frame: w = R^T w_p ; v = R^T (v_p + w_p x p)
    const Vec3<T> wc = mulT(R, w);
    const Vec3<T> vc = mulT(R, v + crossVC<T>(w, p));
    const Vec3<T> bwc = mulT(R, bw);
    const Vec3<T> blc = mulT(R, bl + crossVC<T>(bw, p));
    if (ZAX) {
      // S qd = (0, 0, qd), x (S qd) = (y qd, -x qd, 0)
      w = wc; w.z = w.z + qdi;
      v = vc;
      bw = mk<T>(bwc.x + w.y * qdi, bwc.y - w.x * qdi, bwc.z + qddi);
      bl = mk<T>(blc.x + v.y * qdi, blc.y - v.x * qdi, blc.z);
    } else {
      const Vec3<T> vJ = mk<T>(u[0] * qdi, u[1] * qdi, u[2] * qdi);
      w = wc + vJ;
      v = vc;
      // a_i = X a_p + S qdd + v_i x (S qd)
      bw = bwc + mk<T>(u[0] * qddi, u[1] * qddi, u[2] * qddi) + cross(w, vJ);
      bl = blc + cross(v, vJ);
    }
  }
  Vec3<T> Fl = mk<T>(0.0, 0.0, 0.0), Fn = Fl;
#pragma unroll 1
  for (int i = NJ - 1; i >= 0; --i) {
    const double* u = m->axis[i];
    Vec3<T> hl, hn, f, n;
    inertiaMul<T>(m, i, v, w, hl, hn);
    inertiaMul<T>(m, i, bl, bw, f, n);
    Fl = Fl + f + cross(w, hl);
    Fn = Fn + n + cross(w, hn) + cross(v, hl);
    const T ti = ZAX ? Fn.z : u[0] * Fn.x + u[1] * Fn.y + u[2] * Fn.z;
    tau_d[i] = ti.d;
    if (write_nominal) tau_v[i] = ti.v;
    if (i > 0) {
      const bool mine = (k == i);
      const T cqi(cs[2 * i], (mine && kind == 0) ? -cs[2 * i + 1] : 0.0);
      const T sqi(cs[2 * i + 1], (mine && kind == 0) ? cs[2 * i] : 0.0);
      const T qdi(qdn[i], (mine && kind == 1) ? 1.0 : 0.0);
      const T qddi(qddn[i], (mine && kind == 2) ? 1.0 : 0.0);
      Mat3<T> R;
      revoluteRotation<T, ZAX>(m->R[i], m->axis[i], cqi, sqi, R);
      const double* p = m->p[i];
      // force into the parent frame: f_p = R f ; n_p = R n + p x (R f)
      const Vec3<T> Rf = mul(R, Fl);
      Fn = mul(R, Fn) + crossC<T>(p, Rf);
      Fl = Rf;
      // undo the kinematic step, then map the motion back to the parent frame
      Vec3<T> bwc, blc, wc;
      if (ZAX) {
        bwc = mk<T>(bw.x - w.y * qdi, bw.y + w.x * qdi, bw.z - qddi);
        blc = mk<T>(bl.x - v.y * qdi, bl.y + v.x * qdi, bl.z);
        wc = w; wc.z = wc.z - qdi;
      } else {
        const Vec3<T> vJ = mk<T>(u[0] * qdi, u[1] * qdi, u[2] * qdi);
        bwc = bw - mk<T>(u[0] * qddi, u[1] * qddi, u[2] * qddi) - cross(w, vJ);
        blc = bl - cross(v, vJ);
        wc = w - vJ;
      }
      w = mul(R, wc);
      v = mul(R, v) - crossVC<T>(w, p);
      bw = mul(R, bwc);
      bl = mul(R, blc) - crossVC<T>(bw, p);
    }
  }
}

// Serial chain: joint i hangs off joint i-1 (iiwa14).
template <int N> struct ChainTopo {
  static constexpr int NJ = N;
  __host__ __device__ static constexpr int parent(int i) { return i - 1; }
};

}  // namespace idocp_dev
#endif  // IDOCP_DEV_RBD_HPP_
