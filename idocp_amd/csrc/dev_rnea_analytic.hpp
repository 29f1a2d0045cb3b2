// Analytic derivatives of the inverse dynamics of a fixed-base SERIAL CHAIN, tau = ID(q, v, a): dtau/dq, dtau/dv, dtau/da = M,
// by the world-frame recursion of pinocchio::computeRNEADerivatives (call site robot.hxx:466-500; Carpentier & Mansard, RSS 2018)
// that the oracle restates (oracle/rbd.cpp, Robot::forwardPass / RNEADerivatives).  Round 1 / 2 differentiated the RNEA by forward
// mode, one tangent per lane (rneaChain<Dual>, dev_rbd.hpp): 3 NJ full sweeps of dual numbers per stage.  Here the 3 NJ lanes of a
// stage group share ONE evaluation:
//
//   lane (kind, k) walks the chain to joint k (every lane of the wavefront walks in lockstep: no exchange needed), keeping the world
//   placement of joint k, its motion subspace S_k = (p x w, w), the spatial velocity / acceleration of body k and of its parent;
//   it forms body k's spatial inertia about the world origin Y_k = (m, h, J), its momentum Y_k v_k = (hf, hn), its force and the
//   "B" matrix of the algorithm, B_k = crf(v) Y - Y crm(v) - [[0, S(hf)], [S(hf), S(hn)]], which collapses to
//   [[0, -2 S(hf)], [0, Sym - S(hn)]] with Sym = W J + (W J)^T - (h vl^T + vl h^T) + 2 (vl . h) I  (W = S(w)): 28 numbers per body;
//   composites (sums over the bodies j >= k of the chain) come out of an LDS table; then
//     dFda_k = Yc S_k,  dFdv_k = Bc S_k + Yc dAdv_k,  dFdq_k = Yc dAdq_k + Bc dVdq_k  (+ S_k x* fc_k for the rows above),
//   and lane (kind, r) assembles ROW r of its matrix from three published vectors per column:
//     X(r, c) = S_r . P1_c                      (c >= r)
//             = (Yc_r S_r) . P2_c + (Bc_r^T S_r) . P3_c   (c <  r)
//     kind 0 (d/dq): P1 = dFdq (corrected), P2 = dAdq, P3 = dVdq;  kind 1 (d/dv): P1 = dFdv, P2 = dAdv, P3 = S;
//     kind 2 (d/da): P1 = dFda, P2 = S, P3 = 0  (symmetry of M).
// ~2.1 k instructions per lane instead of ~6.5 k.
#ifndef IDOCP_DEV_RNEA_ANALYTIC_HPP_
#define IDOCP_DEV_RNEA_ANALYTIC_HPP_

#include "dev_rbd.hpp"

namespace idocp_dev {

namespace ra {
typedef Vec3<double> V;
__device__ __forceinline__ V v3(double x, double y, double z) { return mk<double>(x, y, z); }
__device__ __forceinline__ V ldv(const double* p) { return v3(p[0], p[1], p[2]); }
__device__ __forceinline__ void stv(double* p, V a) { p[0] = a.x; p[1] = a.y; p[2] = a.z; }
struct Mot { V l, a; };                  // spatial motion (linear, angular) or force (force, torque), about the world origin
__device__ __forceinline__ Mot operator+(Mot x, Mot y) { Mot r; r.l = x.l + y.l; r.a = x.a + y.a; return r; }
__device__ __forceinline__ Mot operator-(Mot x, Mot y) { Mot r; r.l = x.l - y.l; r.a = x.a - y.a; return r; }
__device__ __forceinline__ Mot scale(double s, Mot x) { Mot r; r.l = s * x.l; r.a = s * x.a; return r; }
__device__ __forceinline__ double dot6(Mot x, Mot y) { return dot(x.l, y.l) + dot(x.a, y.a); }
// motion x motion: (v, w) x (v2, w2) = (w x v2 + v x w2, w x w2)
__device__ __forceinline__ Mot crossMM(Mot a, Mot b) { Mot r; r.l = cross(a.a, b.l) + cross(a.l, b.a); r.a = cross(a.a, b.a); return r; }
// motion x* force: (v, w) x* (f, n) = (w x f, w x n + v x f)
__device__ __forceinline__ Mot crossMF(Mot m, Mot f) { Mot r; r.l = cross(m.a, f.l); r.a = cross(m.a, f.a) + cross(m.l, f.l); return r; }
__device__ __forceinline__ V symMul(const double* S, V x) {      // S = (xx, xy, xz, yy, yz, zz)
  return v3(S[0] * x.x + S[1] * x.y + S[2] * x.z, S[1] * x.x + S[3] * x.y + S[4] * x.z, S[2] * x.x + S[4] * x.y + S[5] * x.z);
}
// composite record of a body (28 doubles): m, h(3), J(6), hf(3), hn(3), Sym(6), f(6)
constexpr int C_M = 0, C_H = 1, C_J = 4, C_HF = 10, C_HN = 13, C_SYM = 16, C_F = 22, CREC = 28;
// Y x, B x, B^T x with the composite record c
__device__ __forceinline__ Mot mulY(const double* c, Mot x) {
  const V h = ldv(c + C_H);
  Mot r; r.l = c[C_M] * x.l + cross(x.a, h); r.a = cross(h, x.l) + symMul(c + C_J, x.a); return r;
}
__device__ __forceinline__ Mot mulB(const double* c, Mot x) {
  const V hf = ldv(c + C_HF), hn = ldv(c + C_HN);
  Mot r; r.l = (-2.0) * cross(hf, x.a); r.a = symMul(c + C_SYM, x.a) - cross(hn, x.a); return r;
}
__device__ __forceinline__ Mot mulBt(const double* c, Mot x) {
  const V hf = ldv(c + C_HF), hn = ldv(c + C_HN);
  Mot r; r.l = v3(0.0, 0.0, 0.0); r.a = 2.0 * cross(hf, x.l) + symMul(c + C_SYM, x.a) + cross(hn, x.a); return r;
}
__device__ __forceinline__ void stm(double* p, Mot x) { stv(p, x.l); stv(p + 3, x.a); }
__device__ __forceinline__ Mot ldm(const double* p) { Mot r; r.l = ldv(p); r.a = ldv(p + 3); return r; }
}  // namespace ra

// LDS of one stage group: the bodies' records, then (same storage: a barrier lies between) the three published vectors per (kind, joint)
template <int NJ> struct RneaAnalyticLds {
  union {
    double comp[NJ][ra::CREC];
    double pub[NJ][42];      // per joint: S, dAdq, dVdq, dAdv, dFdq (corrected), dFdv, dFda -- each written by the lane of the kind that owns it
  };
};
// The constants of the chain the recursion reads, compact (a workgroup keeps one copy in LDS: from the DevModel in global memory every
// joint of the walk waited for its own scalar loads)
template <int NJ> struct ChainConsts {
  double R[NJ][9], axis[NJ][3], p[NJ][3], mc[NJ][3], Io[NJ][6], mass[NJ], gravity[3];
  __device__ void load(const DevModel* __restrict__ m, int tid, int nt) {
    for (int e = tid; e < NJ * 9; e += nt) R[e / 9][e % 9] = m->R[e / 9][e % 9];
    for (int e = tid; e < NJ * 3; e += nt) { axis[e / 3][e % 3] = m->axis[e / 3][e % 3]; p[e / 3][e % 3] = m->p[e / 3][e % 3]; mc[e / 3][e % 3] = m->mc[e / 3][e % 3]; }
    for (int e = tid; e < NJ * 6; e += nt) Io[e / 6][e % 6] = m->Io[e / 6][e % 6];
    for (int e = tid; e < NJ; e += nt) mass[e] = m->mass[e];
    if (tid < 3) gravity[tid] = m->gravity[tid];
  }
};

// Lane (kind, k) of a stage group; `sync` is a barrier of the workgroup (every lane of the wavefront calls the function).
// cs[2 j] = cos q_j, cs[2 j + 1] = sin q_j; qd, qdd: nominal joint velocity / acceleration.
// Outputs (LDS): dID[kind][c * NJ + r] = d tau_r / d (q | v | a)_c (the layout of the forward-mode path), tau[r] from the kind-0 lanes.
template <int NJ, bool ZAX, typename Model, typename Sync>
__device__ __forceinline__ void rneaDerivativesChain(const Model* m, const double* cs, const double* qd,
                                                     const double* qdd, int kind, int k, bool valid,
                                                     RneaAnalyticLds<NJ>& L, double* dID0, double* dID1, double* dID2, double* tau,
                                                     Sync sync, double* kin_out = nullptr) {
  using namespace ra;
  // ---- walk to joint k: world placement, motion subspace, velocity / acceleration of body k and of its parent ----
  double Rw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V pw = v3(0.0, 0.0, 0.0);
  Mot vel; vel.l = v3(0, 0, 0); vel.a = v3(0, 0, 0);
  Mot acc; acc.l = v3(-m->gravity[0], -m->gravity[1], -m->gravity[2]); acc.a = v3(0, 0, 0);
  Mot S, vk = vel, ak = acc;
  double Rk[9];
  V pk = pw;
  S.l = v3(0, 0, 0); S.a = v3(0, 0, 0);
#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
    Mat3<double> Rm;
    revoluteRotation<double, ZAX>(m->R[j], m->axis[j], cs[2 * j], cs[2 * j + 1], Rm);
    const double* pj = m->p[j];
    pw = pw + v3(Rw[0] * pj[0] + Rw[1] * pj[1] + Rw[2] * pj[2], Rw[3] * pj[0] + Rw[4] * pj[1] + Rw[5] * pj[2], Rw[6] * pj[0] + Rw[7] * pj[1] + Rw[8] * pj[2]);
    double Rn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rn[3 * r + c] = Rw[3 * r] * Rm.m[c] + Rw[3 * r + 1] * Rm.m[3 + c] + Rw[3 * r + 2] * Rm.m[6 + c];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rw[e] = Rn[e];
    const double* u = m->axis[j];
    Mot Sj;
    Sj.a = ZAX ? v3(Rw[2], Rw[5], Rw[8]) : v3(Rw[0] * u[0] + Rw[1] * u[1] + Rw[2] * u[2], Rw[3] * u[0] + Rw[4] * u[1] + Rw[5] * u[2], Rw[6] * u[0] + Rw[7] * u[1] + Rw[8] * u[2]);
    Sj.l = cross(pw, Sj.a);
    const Mot vJ = scale(qd[j], Sj);
    const Mot vn = vel + vJ;
    const Mot an = acc + scale(qdd[j], Sj) + crossMM(vn, vJ);
    if (j == k) {
      S = Sj; vk = vn; ak = an; pk = pw;
#pragma unroll
      for (int e = 0; e < 9; ++e) Rk[e] = Rw[e];
    }
    vel = vn; acc = an;
  }
  // (for a task-space cost on the chain: the world placement of joint k and its motion subspace, 18 doubles)
  if (kin_out) {
#pragma unroll
    for (int e = 0; e < 9; ++e) kin_out[e] = Rk[e];
    stv(kin_out + 9, pk); stm(kin_out + 12, S);
  }
  // ---- body k: spatial inertia about the world origin, momentum, force, B ----
  double rec[CREC];
  {
    const double mass = m->mass[k];
    const double* mc = m->mc[k];
    const double* Io = m->Io[k];
    const V hc = v3(Rk[0] * mc[0] + Rk[1] * mc[1] + Rk[2] * mc[2], Rk[3] * mc[0] + Rk[4] * mc[1] + Rk[5] * mc[2], Rk[6] * mc[0] + Rk[7] * mc[1] + Rk[8] * mc[2]);
    const V h = hc + mass * pk;
    // R Io R^T
    double RI[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      RI[3 * r] = Rk[3 * r] * Io[0] + Rk[3 * r + 1] * Io[1] + Rk[3 * r + 2] * Io[2];
      RI[3 * r + 1] = Rk[3 * r] * Io[1] + Rk[3 * r + 1] * Io[3] + Rk[3 * r + 2] * Io[4];
      RI[3 * r + 2] = Rk[3 * r] * Io[2] + Rk[3 * r + 1] * Io[4] + Rk[3 * r + 2] * Io[5];
    }
    auto rirt = [&](int r, int c) { return RI[3 * r] * Rk[3 * c] + RI[3 * r + 1] * Rk[3 * c + 1] + RI[3 * r + 2] * Rk[3 * c + 2]; };
    // J = R Io R^T - m p p^T + m |p|^2 I - hc p^T - p hc^T + 2 (p . hc) I     (parallel axis from the joint origin to the world origin)
    const double pp = dot(pk, pk), phc = dot(pk, hc);
    const double dg = mass * pp + 2.0 * phc;
    const double P[3] = {pk.x, pk.y, pk.z}, H[3] = {hc.x, hc.y, hc.z};
    auto jel = [&](int r, int c) { return rirt(r, c) - mass * P[r] * P[c] - H[r] * P[c] - P[r] * H[c] + (r == c ? dg : 0.0); };
    double J[6] = {jel(0, 0), jel(0, 1), jel(0, 2), jel(1, 1), jel(1, 2), jel(2, 2)};
    rec[C_M] = mass; stv(rec + C_H, h);
#pragma unroll
    for (int e = 0; e < 6; ++e) rec[C_J + e] = J[e];
    const Mot oh = mulY(rec, vk);
    stv(rec + C_HF, oh.l); stv(rec + C_HN, oh.a);
    // Sym = W J + (W J)^T - (h vl^T + vl h^T) + 2 (vl . h) I
    const V w = vk.a, vl = vk.l;
    const V Jc0 = v3(J[0], J[1], J[2]), Jc1 = v3(J[1], J[3], J[4]), Jc2 = v3(J[2], J[4], J[5]);      // columns (= rows) of J
    const V WJ0 = cross(w, Jc0), WJ1 = cross(w, Jc1), WJ2 = cross(w, Jc2);                            // columns of W J
    const double vh = dot(vl, h);
    rec[C_SYM + 0] = 2.0 * WJ0.x - 2.0 * h.x * vl.x + 2.0 * vh;
    rec[C_SYM + 1] = WJ1.x + WJ0.y - (h.x * vl.y + vl.x * h.y);
    rec[C_SYM + 2] = WJ2.x + WJ0.z - (h.x * vl.z + vl.x * h.z);
    rec[C_SYM + 3] = 2.0 * WJ1.y - 2.0 * h.y * vl.y + 2.0 * vh;
    rec[C_SYM + 4] = WJ2.y + WJ1.z - (h.y * vl.z + vl.y * h.z);
    rec[C_SYM + 5] = 2.0 * WJ2.z - 2.0 * h.z * vl.z + 2.0 * vh;
    const Mot f = mulY(rec, ak) + crossMF(vk, oh);
    stm(rec + C_F, f);
  }
  if (valid && kind == 0) {
#pragma unroll
    for (int e = 0; e < CREC; ++e) L.comp[k][e] = rec[e];
  }
  sync();
  // ---- composites: the bodies j >= k ----
#pragma unroll 1
  for (int j = k + 1; j < NJ; ++j) {
#pragma unroll
    for (int e = 0; e < CREC; ++e) rec[e] += L.comp[j][e];
  }
  sync();                                         // (the published vectors take the records' storage)
  // ---- columns of joint k ----
  // (the parent's motion is the body's minus the joint's own share: v_k = v_p + S qd, a_k = a_p + S qdd + v_k x S qd)
  const Mot vJk = scale(qd[k], S);
  const Mot vpar = vk - vJk;
  const Mot apar = ak - scale(qdd[k], S) - crossMM(vk, vJk);
  const Mot dV = crossMM(vpar, S);
  const Mot dA = crossMM(apar, S) + crossMM(vpar, dV);
  const Mot dAv = crossMM(vk, S) + dV;
  const Mot YS = mulY(rec, S);
  const Mot BtS = mulBt(rec, S);
  const Mot fc = ldm(rec + C_F);
  Mot P1, P2, P3, own;
  if (kind == 0) {
    own = mulY(rec, dA) + mulB(rec, dV);          // dFdq, as joint k's own row sees it
    P1 = own + crossMF(S, fc);                    // ... and as the rows above it do
    P2 = dA; P3 = dV;
  } else if (kind == 1) {
    own = mulB(rec, S) + mulY(rec, dAv);
    P1 = own; P2 = dAv; P3 = S;
  } else {
    own = YS;
    P1 = YS; P2 = S; P3.l = v3(0, 0, 0); P3.a = v3(0, 0, 0);
  }
  // offsets of (P1, P2, P3) of a kind inside the joint's published block
  const int o1 = kind == 0 ? 24 : (kind == 1 ? 30 : 36), o2 = kind == 0 ? 6 : (kind == 1 ? 18 : 0), o3 = kind == 0 ? 12 : 0;
  if (valid) {
    double* o = L.pub[k];
    stm(o + o1, P1);
    if (kind < 2) stm(o + o2, P2);
    if (kind == 0) stm(o + o3, P3);             // (S: written by the kind-1 lane as its P3 at offset 0 -- see o3 -- and read by kind 2 as P2)
    if (kind == 1) stm(o, S);
    if (kind == 0) tau[k] = dot6(S, fc);
  }
  sync();
  // ---- row k of this lane's matrix ----
  double* dID = kind == 0 ? dID0 : (kind == 1 ? dID1 : dID2);
#pragma unroll 1
  for (int c = 0; c < NJ; ++c) {
    const double* o = L.pub[c];
    double val;
    if (c == k) val = dot6(S, own);
    else if (c > k) val = dot6(S, ldm(o + o1));
    else val = dot6(YS, ldm(o + o2)) + (kind < 2 ? dot6(BtS, ldm(o + o3)) : 0.0);
    if (valid) dID[c * NJ + k] = val;
  }
}

// ---- the same recursion in two phases (un_linearize_kernel) ------------------------------------------------------------------------
// The walk, the body record, the composites and the columns dF/d(q, v, a) of a joint do not depend on the seed kind: in the
// single-phase form above the q / v / a lanes of a joint each ran them.  Phase A runs them once per (stage, joint) -- seven lanes
// per stage, nine stages per wavefront, the composites by a suffix scan over the seven lanes of a stage (shuffles) -- and leaves a block
// of per-joint vectors in the lane's registers; phase B, three stages at a time with a lane per (stage, kind, joint), assembles the rows
// out of the blocks of its three stages, which phase A's lanes have just put into LDS.
struct RneaBlock {
  // S, dA/dq, dV/dq, dA/dv, dF/dq (+ S x* fc: as the rows above see it), dF/dv, dF/da = Yc S, dF/dq (own row), Bc^T S, tau
  static constexpr int S = 0, DA = 6, DV = 12, DAV = 18, FQC = 24, FV = 30, FA = 36, FQ = 42, BTS = 48, TAU = 54;
  static constexpr int LEN = 55;      // (odd: the lanes of a stage then hit distinct LDS banks)
};

// kin_out (or null): world placement (R, p) of joint k, 12 doubles -- for a task-space cost on the chain
template <int NJ, bool ZAX, typename Model>
__device__ __forceinline__ void rneaDerivPhaseA(const Model* m, const double* cs, const double* qd, const double* qdd, int k, double* blk, double* kin_out) {
  using namespace ra;
  double Rw[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  V pw = v3(0.0, 0.0, 0.0);
  Mot vel; vel.l = v3(0, 0, 0); vel.a = v3(0, 0, 0);
  Mot acc; acc.l = v3(-m->gravity[0], -m->gravity[1], -m->gravity[2]); acc.a = v3(0, 0, 0);
  Mot S, vk = vel, ak = acc;
  double Rk[9];
  V pk = pw;
  S.l = v3(0, 0, 0); S.a = v3(0, 0, 0);
#pragma unroll 1
  for (int j = 0; j < NJ; ++j) {
    Mat3<double> Rm;
    revoluteRotation<double, ZAX>(m->R[j], m->axis[j], cs[2 * j], cs[2 * j + 1], Rm);
    const double* pj = m->p[j];
    pw = pw + v3(Rw[0] * pj[0] + Rw[1] * pj[1] + Rw[2] * pj[2], Rw[3] * pj[0] + Rw[4] * pj[1] + Rw[5] * pj[2], Rw[6] * pj[0] + Rw[7] * pj[1] + Rw[8] * pj[2]);
    double Rn[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int c = 0; c < 3; ++c) Rn[3 * r + c] = Rw[3 * r] * Rm.m[c] + Rw[3 * r + 1] * Rm.m[3 + c] + Rw[3 * r + 2] * Rm.m[6 + c];
#pragma unroll
    for (int e = 0; e < 9; ++e) Rw[e] = Rn[e];
    const double* u = m->axis[j];
    Mot Sj;
    Sj.a = ZAX ? v3(Rw[2], Rw[5], Rw[8]) : v3(Rw[0] * u[0] + Rw[1] * u[1] + Rw[2] * u[2], Rw[3] * u[0] + Rw[4] * u[1] + Rw[5] * u[2], Rw[6] * u[0] + Rw[7] * u[1] + Rw[8] * u[2]);
    Sj.l = cross(pw, Sj.a);
    const Mot vJ = scale(qd[j], Sj);
    const Mot vn = vel + vJ;
    const Mot an = acc + scale(qdd[j], Sj) + crossMM(vn, vJ);
    if (j == k) {
      S = Sj; vk = vn; ak = an; pk = pw;
#pragma unroll
      for (int e = 0; e < 9; ++e) Rk[e] = Rw[e];
    }
    vel = vn; acc = an;
  }
  if (kin_out) {
#pragma unroll
    for (int e = 0; e < 9; ++e) kin_out[e] = Rk[e];
    stv(kin_out + 9, pk);
  }
  double rec[CREC];
  {
    const double mass = m->mass[k];
    const double* mc = m->mc[k];
    const double* Io = m->Io[k];
    const V hc = v3(Rk[0] * mc[0] + Rk[1] * mc[1] + Rk[2] * mc[2], Rk[3] * mc[0] + Rk[4] * mc[1] + Rk[5] * mc[2], Rk[6] * mc[0] + Rk[7] * mc[1] + Rk[8] * mc[2]);
    const V h = hc + mass * pk;
    double RI[9];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      RI[3 * r] = Rk[3 * r] * Io[0] + Rk[3 * r + 1] * Io[1] + Rk[3 * r + 2] * Io[2];
      RI[3 * r + 1] = Rk[3 * r] * Io[1] + Rk[3 * r + 1] * Io[3] + Rk[3 * r + 2] * Io[4];
      RI[3 * r + 2] = Rk[3 * r] * Io[2] + Rk[3 * r + 1] * Io[4] + Rk[3 * r + 2] * Io[5];
    }
    auto rirt = [&](int r, int c) { return RI[3 * r] * Rk[3 * c] + RI[3 * r + 1] * Rk[3 * c + 1] + RI[3 * r + 2] * Rk[3 * c + 2]; };
    const double pp = dot(pk, pk), phc = dot(pk, hc);
    const double dg = mass * pp + 2.0 * phc;
    const double P[3] = {pk.x, pk.y, pk.z}, H[3] = {hc.x, hc.y, hc.z};
    auto jel = [&](int r, int c) { return rirt(r, c) - mass * P[r] * P[c] - H[r] * P[c] - P[r] * H[c] + (r == c ? dg : 0.0); };
    double J[6] = {jel(0, 0), jel(0, 1), jel(0, 2), jel(1, 1), jel(1, 2), jel(2, 2)};
    rec[C_M] = mass; stv(rec + C_H, h);
#pragma unroll
    for (int e = 0; e < 6; ++e) rec[C_J + e] = J[e];
    const Mot oh = mulY(rec, vk);
    stv(rec + C_HF, oh.l); stv(rec + C_HN, oh.a);
    const V w = vk.a, vl = vk.l;
    const V Jc0 = v3(J[0], J[1], J[2]), Jc1 = v3(J[1], J[3], J[4]), Jc2 = v3(J[2], J[4], J[5]);
    const V WJ0 = cross(w, Jc0), WJ1 = cross(w, Jc1), WJ2 = cross(w, Jc2);
    const double vh = dot(vl, h);
    rec[C_SYM + 0] = 2.0 * WJ0.x - 2.0 * h.x * vl.x + 2.0 * vh;
    rec[C_SYM + 1] = WJ1.x + WJ0.y - (h.x * vl.y + vl.x * h.y);
    rec[C_SYM + 2] = WJ2.x + WJ0.z - (h.x * vl.z + vl.x * h.z);
    rec[C_SYM + 3] = 2.0 * WJ1.y - 2.0 * h.y * vl.y + 2.0 * vh;
    rec[C_SYM + 4] = WJ2.y + WJ1.z - (h.y * vl.z + vl.y * h.z);
    rec[C_SYM + 5] = 2.0 * WJ2.z - 2.0 * h.z * vl.z + 2.0 * vh;
    const Mot f = mulY(rec, ak) + crossMF(vk, oh);
    stm(rec + C_F, f);
  }
  // composites over the bodies j >= k of the stage: suffix scan over its NJ consecutive lanes (the lanes of the next stage are masked out)
  static_assert(NJ <= 8, "three doubling steps");
#pragma unroll
  for (int off = 1; off < 8; off <<= 1) {
    const bool take = k + off < NJ;
#pragma unroll
    for (int e = 0; e < CREC; ++e) {
      const double t = __shfl_down(rec[e], off);
      rec[e] += take ? t : 0.0;
    }
  }
  const Mot vJk = scale(qd[k], S);
  const Mot vpar = vk - vJk;
  const Mot apar = ak - scale(qdd[k], S) - crossMM(vk, vJk);
  const Mot dV = crossMM(vpar, S);
  const Mot dA = crossMM(apar, S) + crossMM(vpar, dV);
  const Mot dAv = crossMM(vk, S) + dV;
  const Mot YS = mulY(rec, S);
  const Mot fc = ldm(rec + C_F);
  const Mot Fq = mulY(rec, dA) + mulB(rec, dV);
  stm(blk + RneaBlock::S, S); stm(blk + RneaBlock::DA, dA); stm(blk + RneaBlock::DV, dV); stm(blk + RneaBlock::DAV, dAv);
  stm(blk + RneaBlock::FQC, Fq + crossMF(S, fc)); stm(blk + RneaBlock::FV, mulB(rec, S) + mulY(rec, dAv)); stm(blk + RneaBlock::FA, YS);
  stm(blk + RneaBlock::FQ, Fq); stm(blk + RneaBlock::BTS, mulBt(rec, S));
  blk[RneaBlock::TAU] = dot6(S, fc);
}

// Row r of the kind's matrix out of the blocks of the stage's joints (LDS, `stride` doubles apart): row[c] = d tau_r / d (q | v | a)_c
template <int NJ>
__device__ __forceinline__ void rneaDerivPhaseB(const double* pub, int stride, int kind, int r, double* row) {
  using namespace ra;
  const double* me = pub + r * stride;
  const Mot S = ldm(me + RneaBlock::S), YS = ldm(me + RneaBlock::FA), BtS = ldm(me + RneaBlock::BTS);
  const Mot own = ldm(me + (kind == 0 ? RneaBlock::FQ : (kind == 1 ? RneaBlock::FV : RneaBlock::FA)));
  const int o1 = kind == 0 ? RneaBlock::FQC : (kind == 1 ? RneaBlock::FV : RneaBlock::FA);
  const int o2 = kind == 0 ? RneaBlock::DA : (kind == 1 ? RneaBlock::DAV : RneaBlock::S);
  const int o3 = kind == 0 ? RneaBlock::DV : RneaBlock::S;
#pragma unroll
  for (int c = 0; c < NJ; ++c) {
    const double* o = pub + c * stride;
    double val;
    if (c == r) val = dot6(S, own);
    else if (c > r) val = dot6(S, ldm(o + o1));
    else val = dot6(YS, ldm(o + o2)) + (kind < 2 ? dot6(BtS, ldm(o + o3)) : 0.0);
    row[c] = val;
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_RNEA_ANALYTIC_HPP_
