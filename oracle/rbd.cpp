// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md and rbd.hpp).
#include "rbd.hpp"

#include <cmath>
#include <cstring>

namespace oracle {

// ------------------------------------------------------------ helpers ----
static inline void cross3(const real* a, const real* b, real* c) {
  const real x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static inline void matvec3(const real* R, const real* x, real* y) {
  const real a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  const real b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  const real c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static inline void matmul3(const real* A, const real* B, real* C) {
  real T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  std::memcpy(C, T, sizeof(T));
}
// motion x motion:  (v,w) x (v2,w2) = (w x v2 + v x w2, w x w2)
static inline void crossMM(const real* a, const real* b, real* o) {
  real t1[3], t2[3], t3[3];
  cross3(a + 3, b, t1); cross3(a, b + 3, t2); cross3(a + 3, b + 3, t3);
  for (int k = 0; k < 3; ++k) { o[k] = t1[k] + t2[k]; o[3 + k] = t3[k]; }
}
// motion x* force:  (v,w) x* (f,n) = (w x f, w x n + v x f)
static inline void crossMF(const real* m, const real* f, real* o) {
  real t1[3], t2[3], t3[3];
  cross3(m + 3, f, t1); cross3(m + 3, f + 3, t2); cross3(m, f, t3);
  for (int k = 0; k < 3; ++k) { o[k] = t1[k]; o[3 + k] = t2[k] + t3[k]; }
}
static inline void skew(const real* v, real* S) {   // row-major 3x3
  S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static Mat crm(const real* v) {        // 6x6: crm(v) m = v x m
  Mat X(6, 6); real Sv[9], Sw[9]; skew(v, Sv); skew(v + 3, Sw);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    X(i, j) = Sw[3 * i + j]; X(i, 3 + j) = Sv[3 * i + j]; X(3 + i, 3 + j) = Sw[3 * i + j];
  }
  return X;
}
static Mat crf(const real* v) {        // 6x6: crf(v) f = v x* f
  Mat X(6, 6); real Sv[9], Sw[9]; skew(v, Sv); skew(v + 3, Sw);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    X(i, j) = Sw[3 * i + j]; X(3 + i, j) = Sv[3 * i + j]; X(3 + i, 3 + j) = Sw[3 * i + j];
  }
  return X;
}
static inline real dot6(const real* a, const real* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

static void rotAxis(const real* u, real q, real* R) {
  const real c = std::cos(q), s = std::sin(q), t = 1 - c;
  R[0] = c + t * u[0] * u[0];        R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
  R[3] = t * u[1] * u[0] + s * u[2]; R[4] = c + t * u[1] * u[1];        R[5] = t * u[1] * u[2] - s * u[0];
  R[6] = t * u[2] * u[0] - s * u[1]; R[7] = t * u[2] * u[1] + s * u[0]; R[8] = c + t * u[2] * u[2];
}
static void quatToR(const real* qt, real* R) {  // xyzw
  const real x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// ---------------------------------------------------------------- Robot ----
Robot::Robot(const RModel& model) : m_(model) {
  const int n = m_.njoints, nv = m_.nv;
  fjoint_.assign(n, std::vector<real>(6, 0.0));
  in_subtree_.assign(n, std::vector<bool>(n, false));
  for (int j = 0; j < n; ++j) for (int a = j; a >= 0; a = m_.parent[a]) in_subtree_[a][j] = true;
  oMi_.resize(n);
  S_ = Mat(6, nv); dVdq_ = Mat(6, nv); dAdq_ = Mat(6, nv); dAdv_ = Mat(6, nv);
  ov_.assign(n, Mat(6)); oa_.assign(n, Mat(6)); of_.assign(n, Mat(6));
  oY_.assign(n, Mat(6, 6)); oB_.assign(n, Mat(6, 6));
  dof_joint_.assign(nv, 0);
  for (int i = 0; i < n; ++i) {
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    for (int k = 0; k < ndof; ++k) dof_joint_[m_.idx_v[i] + k] = i;
  }
}

void Robot::setContactForces(const std::vector<bool>& active, const std::vector<Mat>& f) {
  FLOP_REGION(R_RNEA);
  // PointContact::computeJointForceFromContactForce (point_contact.hxx:15-20):
  // fjoint[parent] = jXf.act(Force(f, 0))
  for (auto& fj : fjoint_) std::fill(fj.begin(), fj.end(), 0.0);
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    real fl[3], n[3];
    matvec3(m_.contact_R[c], f[c].d.data(), fl);
    cross3(m_.contact_p[c], fl, n);
    auto& fj = fjoint_[m_.contact_joint[c]];
    for (int k = 0; k < 3; ++k) { fj[k] = fl[k]; fj[3 + k] = n[k]; }
  }
}

void Robot::jointPlacement(int i, const Mat& q, SE3& M) const {
  real Rj[9], pj[3] = {0, 0, 0};
  if (m_.jtype[i] == IDOCP_JOINT_REVOLUTE) {
    rotAxis(m_.axis[i], q[m_.idx_q[i]], Rj);
  } else {
    const int iq = m_.idx_q[i];
    quatToR(&q.d[iq + 3], Rj);
    pj[0] = q[iq]; pj[1] = q[iq + 1]; pj[2] = q[iq + 2];
  }
  matmul3(m_.plc_R[i], Rj, M.R);
  real t[3]; matvec3(m_.plc_R[i], pj, t);
  for (int k = 0; k < 3; ++k) M.p[k] = m_.plc_p[i][k] + t[k];
}

// World-frame forward pass (ComputeRNEADerivativesForwardStep of pinocchio's
// rnea-derivatives): oMi, J columns, ov, oa_gf, oYcrb, doYcrb, of, and the
// column sets dVdq, dAdq, dAdv.
void Robot::forwardPass(const Mat& q, const Mat& v, const Mat& a, bool gravity) {
  const int n = m_.njoints;
  real a0[6] = {0, 0, 0, 0, 0, 0};
  if (gravity) for (int k = 0; k < 3; ++k) a0[k] = -m_.gravity[k];
  for (int i = 0; i < n; ++i) {
    const int pa = m_.parent[i];
    SE3 li; jointPlacement(i, q, li);
    if (pa >= 0) {
      matmul3(oMi_[pa].R, li.R, oMi_[i].R);
      real t[3]; matvec3(oMi_[pa].R, li.p, t);
      for (int k = 0; k < 3; ++k) oMi_[i].p[k] = oMi_[pa].p[k] + t[k];
    } else {
      oMi_[i] = li;
    }
    const real* R = oMi_[i].R; const real* p = oMi_[i].p;
    const int iv = m_.idx_v[i];
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    // J columns = oMi.act(S)
    if (ndof == 1) {
      real w[3], l[3]; matvec3(R, m_.axis[i], w); cross3(p, w, l);
      for (int k = 0; k < 3; ++k) { S_(k, iv) = l[k]; S_(3 + k, iv) = w[k]; }
    } else {
      for (int c = 0; c < 3; ++c) {
        real e[3] = {0, 0, 0}; e[c] = 1; real Re[3], l[3];
        matvec3(R, e, Re); cross3(p, Re, l);
        for (int k = 0; k < 3; ++k) {
          S_(k, iv + c) = Re[k]; S_(3 + k, iv + c) = 0;
          S_(k, iv + 3 + c) = l[k]; S_(3 + k, iv + 3 + c) = Re[k];
        }
      }
    }
    real vJ[6] = {0, 0, 0, 0, 0, 0}, aJ[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ndof; ++c) for (int k = 0; k < 6; ++k) {
      vJ[k] += S_(k, iv + c) * v[iv + c]; aJ[k] += S_(k, iv + c) * a[iv + c];
    }
    const real* ovp = pa >= 0 ? ov_[pa].d.data() : nullptr;
    const real* oap = pa >= 0 ? oa_[pa].d.data() : a0;
    for (int k = 0; k < 6; ++k) ov_[i][k] = (ovp ? ovp[k] : real(0.0)) + vJ[k];
    real vxvJ[6]; crossMM(ov_[i].d.data(), vJ, vxvJ);
    for (int k = 0; k < 6; ++k) oa_[i][k] = oap[k] + aJ[k] + vxvJ[k];
    // oYcrb = oMi.act(inertia) as a 6x6 matrix about the world origin
    real c[3], Rc[3]; matvec3(R, m_.com[i], Rc);
    for (int k = 0; k < 3; ++k) c[k] = p[k] + Rc[k];
    real RI[9], Rt[9], Iw[9];
    matmul3(R, m_.inertia[i], RI);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
    matmul3(RI, Rt, Iw);
    const real mass = m_.mass[i];
    real Sc[9]; skew(c, Sc);
    Mat& Y = oY_[i]; Y.setZero();
    for (int r = 0; r < 3; ++r) {
      Y(r, r) = mass;
      for (int s = 0; s < 3; ++s) {
        Y(r, 3 + s) = -mass * Sc[3 * r + s];
        Y(3 + r, s) = mass * Sc[3 * r + s];
        real scsc = 0; for (int k = 0; k < 3; ++k) scsc += Sc[3 * r + k] * Sc[3 * k + s];
        Y(3 + r, 3 + s) = Iw[3 * r + s] - mass * scsc;
      }
    }
    Mat oh = Y * ov_[i];
    Mat Ya = Y * oa_[i];
    real vxh[6]; crossMF(ov_[i].d.data(), oh.d.data(), vxh);
    // external force: of -= oMi.act(fext)
    real fw[3], nw[3], pxf[3];
    matvec3(R, fjoint_[i].data(), fw); matvec3(R, fjoint_[i].data() + 3, nw); cross3(p, fw, pxf);
    for (int k = 0; k < 3; ++k) {
      of_[i][k] = Ya[k] + vxh[k] - fw[k];
      of_[i][3 + k] = Ya[3 + k] + vxh[3 + k] - (nw[k] + pxf[k]);
    }
    // doYcrb = oYcrb.variation(ov) + forceCrossMatrix(oh)
    Mat B = crf(ov_[i].d.data()) * Y - Y * crm(ov_[i].d.data());
    real Sf[9], Sn[9]; skew(oh.d.data(), Sf); skew(oh.d.data() + 3, Sn);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
      B(r, 3 + s) -= Sf[3 * r + s]; B(3 + r, s) -= Sf[3 * r + s]; B(3 + r, 3 + s) -= Sn[3 * r + s];
    }
    oB_[i] = B;
    for (int cidx = 0; cidx < ndof; ++cidx) {
      const int col = iv + cidx;
      real Sk[6], dJ[6], dV[6] = {0, 0, 0, 0, 0, 0}, dA[6], t[6];
      for (int k = 0; k < 6; ++k) Sk[k] = S_(k, col);
      crossMM(ov_[i].d.data(), Sk, dJ);
      crossMM(oap, Sk, dA);
      if (pa >= 0) {
        crossMM(ovp, Sk, dV);
        crossMM(ovp, dV, t);
        for (int k = 0; k < 6; ++k) dA[k] += t[k];
      }
      for (int k = 0; k < 6; ++k) { dVdq_(k, col) = dV[k]; dAdq_(k, col) = dA[k]; dAdv_(k, col) = dJ[k] + dV[k]; }
    }
  }
}

void Robot::RNEA(const Mat& q, const Mat& v, const Mat& a, Mat& tau, bool gravity) {
  FLOP_REGION(R_RNEA);
  forwardPass(q, v, a, gravity);
  const int n = m_.njoints;
  tau = Mat(m_.nv);
  for (int i = n - 1; i >= 0; --i) {
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    for (int c = 0; c < ndof; ++c) tau[m_.idx_v[i] + c] = dot6(&S_.d[6 * (m_.idx_v[i] + c)], of_[i].d.data());
    if (m_.parent[i] >= 0) of_[m_.parent[i]] += of_[i];
  }
}

void Robot::RNEADerivatives(const Mat& q, const Mat& v, const Mat& a, Mat& dq, Mat& dv, Mat& da,
                            bool gravity) {
  FLOP_REGION(R_RNEA_DERIV);
  forwardPass(q, v, a, gravity);
  const int n = m_.njoints, nv = m_.nv;
  dq = Mat(nv, nv); dv = Mat(nv, nv); da = Mat(nv, nv);
  Mat dFda(6, nv), dFdv(6, nv), dFdq(6, nv);
  // ComputeRNEADerivativesBackwardStep
  for (int i = n - 1; i >= 0; --i) {
    const int pa = m_.parent[i], iv = m_.idx_v[i];
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    const Mat& Y = oY_[i]; const Mat& B = oB_[i];
    for (int c = 0; c < ndof; ++c) {
      const int col = iv + c;
      Mat Sk = S_.block(0, col, 6, 1);
      dFda.setBlock(0, col, Y * Sk);
      dFdv.setBlock(0, col, B * Sk + Y * dAdv_.block(0, col, 6, 1));
      Mat fq = Y * dAdq_.block(0, col, 6, 1);
      if (pa >= 0) fq += B * dVdq_.block(0, col, 6, 1);
      dFdq.setBlock(0, col, fq);
    }
    // rows of joint i x columns of subtree(i)
    for (int r = iv; r < iv + ndof; ++r)
      for (int col = 0; col < nv; ++col) {
        if (!in_subtree_[i][dof_joint_[col]]) continue;
        da(r, col) = dot6(&S_.d[6 * r], &dFda.d[6 * col]);
        dv(r, col) = dot6(&S_.d[6 * r], &dFdv.d[6 * col]);
        dq(r, col) = dot6(&S_.d[6 * r], &dFdq.d[6 * col]);
      }
    for (int c = 0; c < ndof; ++c) {
      real t[6]; crossMF(&S_.d[6 * (iv + c)], of_[i].d.data(), t);
      for (int k = 0; k < 6; ++k) dFdq(k, iv + c) += t[k];
    }
    if (pa >= 0) {
      for (int r = iv; r < iv + ndof; ++r) {
        Mat Sr = S_.block(0, r, 6, 1);
        Mat YS = Y * Sr;            // (S_r^T Y)^T, Y symmetric
        Mat BtS = B.t() * Sr;       // (S_r^T B)^T
        for (int col = 0; col < nv; ++col) {
          const int jc = dof_joint_[col];
          if (jc == i || !in_subtree_[jc][i]) continue;   // strict ancestors only
          dq(r, col) = dot6(YS.d.data(), &dAdq_.d[6 * col]) + dot6(BtS.d.data(), &dVdq_.d[6 * col]);
          dv(r, col) = dot6(YS.d.data(), &dAdv_.d[6 * col]) + dot6(BtS.d.data(), &S_.d[6 * col]);
        }
      }
      oY_[pa] += oY_[i]; oB_[pa] += oB_[i]; of_[pa] += of_[i];
    }
  }
  // Robot::RNEADerivatives symmetrises dtau/da (robot.hxx:496-499)
  for (int r = 0; r < nv; ++r) for (int c = 0; c < r; ++c) da(r, c) = da(c, r);
}

}  // namespace oracle

// ======================================================================= contact
namespace oracle {

static void actInvMotion(const real* R, const real* p, const real* m, real* out) {
  // SE3::actInv on a motion: lin = R^T (v - p x w), ang = R^T w
  real pxw[3]; cross3(p, m + 3, pxw);
  real t[3] = {m[0] - pxw[0], m[1] - pxw[1], m[2] - pxw[2]};
  for (int i = 0; i < 3; ++i) {
    out[i] = R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2];
    out[3 + i] = R[i] * m[3] + R[3 + i] * m[4] + R[6 + i] * m[5];
  }
}

void Robot::updateKinematics(const Mat& q, const Mat& v, const Mat& a) {
  FLOP_REGION(R_KINEMATICS);
  const int n = m_.njoints, nv = m_.nv;
  if ((int)kMi_.size() != n) { kMi_.resize(n); kS_ = Mat(6, nv); kv_.assign(n, Mat(6)); ka_.assign(n, Mat(6)); }
  for (int i = 0; i < n; ++i) {
    const int pa = m_.parent[i];
    SE3 li; jointPlacement(i, q, li);
    if (pa >= 0) {
      matmul3(kMi_[pa].R, li.R, kMi_[i].R);
      real t[3]; matvec3(kMi_[pa].R, li.p, t);
      for (int k = 0; k < 3; ++k) kMi_[i].p[k] = kMi_[pa].p[k] + t[k];
    } else {
      kMi_[i] = li;
    }
    const real* R = kMi_[i].R; const real* p = kMi_[i].p;
    const int iv = m_.idx_v[i];
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    if (ndof == 1) {
      real w[3], l[3]; matvec3(R, m_.axis[i], w); cross3(p, w, l);
      for (int k = 0; k < 3; ++k) { kS_(k, iv) = l[k]; kS_(3 + k, iv) = w[k]; }
    } else {
      for (int c = 0; c < 3; ++c) {
        real e[3] = {0, 0, 0}; e[c] = 1; real Re[3], l[3];
        matvec3(R, e, Re); cross3(p, Re, l);
        for (int k = 0; k < 3; ++k) {
          kS_(k, iv + c) = Re[k]; kS_(3 + k, iv + c) = 0;
          kS_(k, iv + 3 + c) = l[k]; kS_(3 + k, iv + 3 + c) = Re[k];
        }
      }
    }
    real vJ[6] = {0, 0, 0, 0, 0, 0}, aJ[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ndof; ++c) for (int k = 0; k < 6; ++k) {
      vJ[k] += kS_(k, iv + c) * v[iv + c]; aJ[k] += kS_(k, iv + c) * a[iv + c];
    }
    for (int k = 0; k < 6; ++k) kv_[i][k] = (pa >= 0 ? kv_[pa][k] : real(0.0)) + vJ[k];
    real vxvJ[6]; crossMM(kv_[i].d.data(), vJ, vxvJ);
    for (int k = 0; k < 6; ++k) ka_[i][k] = (pa >= 0 ? ka_[pa][k] : real(0.0)) + aJ[k] + vxvJ[k];
  }
}

static void framePlacement(const RModel& m, const std::vector<SE3>& kMi, int c, real* R, real* p) {
  const int j = m.contact_joint[c];
  matmul3(kMi[j].R, m.contact_R[c], R);
  real t[3]; matvec3(kMi[j].R, m.contact_p[c], t);
  for (int k = 0; k < 3; ++k) p[k] = kMi[j].p[k] + t[k];
}

void Robot::contactFrame(int c, real* p_world, real* R_world, real* v_local, real* a_local) const {
  real R[9], p[3];
  framePlacement(m_, kMi_, c, R, p);
  const int j = m_.contact_joint[c];
  if (p_world) std::memcpy(p_world, p, sizeof(p));
  if (R_world) std::memcpy(R_world, R, sizeof(R));
  if (v_local) actInvMotion(R, p, kv_[j].d.data(), v_local);
  if (a_local) actInvMotion(R, p, ka_[j].d.data(), a_local);
}

// Derivatives of the frame's LOCAL spatial velocity / acceleration.  With
// V-, A- the world-frame velocity / acceleration of the parent of column j's
// joint, S_j the world-frame column, ov_i/oa_i those of the frame's joint:
//   dv/dq_j = fXo (V- x S_j)
//   da/dq_j = fXo (A- x S_j + (V- x S_j) x (ov_i - V-))
//   da/dv_j = fXo (ov_J(j) x S_j + V- x S_j - ov_i x S_j)
//   da/da_j = fXo S_j                       (= LOCAL frame Jacobian)
// for j in the support of the frame, 0 otherwise (Carpentier & Mansard 2018,
// as implemented by pinocchio::getFrame{Velocity,Acceleration}Derivatives).
void Robot::frameDerivatives(int c, Mat& vdq, Mat& adq, Mat& adv, Mat& ada) const {
  const int nv = m_.nv, ji = m_.contact_joint[c];
  vdq = Mat(6, nv); adq = Mat(6, nv); adv = Mat(6, nv); ada = Mat(6, nv);
  real R[9], p[3];
  framePlacement(m_, kMi_, c, R, p);
  const real zero[6] = {0, 0, 0, 0, 0, 0};
  for (int col = 0; col < nv; ++col) {
    const int jc = dof_joint_[col];
    if (!in_subtree_[jc][ji]) continue;
    const int pa = m_.parent[jc];
    const real* Vm = pa >= 0 ? kv_[pa].d.data() : zero;
    const real* Am = pa >= 0 ? ka_[pa].d.data() : zero;
    const real* Sj = &kS_.d[6 * col];
    const real* ovi = kv_[ji].d.data();
    real VxS[6], AxS[6], d[6], t[6], JxS[6], ixS[6], w[6];
    crossMM(Vm, Sj, VxS); crossMM(Am, Sj, AxS);
    for (int k = 0; k < 6; ++k) d[k] = ovi[k] - Vm[k];
    crossMM(VxS, d, t);
    crossMM(kv_[jc].d.data(), Sj, JxS); crossMM(ovi, Sj, ixS);
    actInvMotion(R, p, VxS, &vdq.d[6 * col]);
    for (int k = 0; k < 6; ++k) w[k] = AxS[k] + t[k];
    actInvMotion(R, p, w, &adq.d[6 * col]);
    for (int k = 0; k < 6; ++k) w[k] = JxS[k] + VxS[k] - ixS[k];
    actInvMotion(R, p, w, &adv.d[6 * col]);
    actInvMotion(R, p, Sj, &ada.d[6 * col]);
  }
}

void Robot::computeBaumgarteResidual(const std::vector<bool>& active, real time_step,
                                     const std::vector<Mat>& contact_points, Mat& C) const {
  FLOP_REGION(R_BAUMGARTE);
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  C = Mat(3 * na);
  const real wv = 2 / time_step, wp = 1 / (time_step * time_step);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    real p[3], v[6], a[6], wxv[3];
    contactFrame(c, p, nullptr, v, a);
    cross3(v + 3, v, wxv);                       // classical acceleration = a_lin + w x v_lin
    for (int k = 0; k < 3; ++k)
      C[row + k] = (a[k] + wxv[k]) + wv * v[k] + wp * (p[k] - contact_points[c][k]);
    row += 3;
  }
}

void Robot::computeBaumgarteDerivatives(const std::vector<bool>& active, real time_step, Mat& dCdq, Mat& dCdv,
                                        Mat& dCda) const {
  FLOP_REGION(R_BAUMGARTE);
  const int nv = m_.nv;
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  dCdq = Mat(3 * na, nv); dCdv = Mat(3 * na, nv); dCda = Mat(3 * na, nv);
  const real wv = 2 / time_step, wp = 1 / (time_step * time_step);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    Mat vdq, adq, adv, J;
    frameDerivatives(c, vdq, adq, adv, J);
    real R[9], v[6], Sl[9], Sw[9];
    contactFrame(c, nullptr, R, v, nullptr);
    skew(v, Sl); skew(v + 3, Sw);
    // point_contact.hxx:117-143, term by term (note the reference ADDS v_linear_skew * d(omega))
    for (int col = 0; col < nv; ++col) {
      for (int r = 0; r < 3; ++r) {
        real dq = adq(r, col), dv = adv(r, col);
        for (int k = 0; k < 3; ++k) {
          dq += Sw[3 * r + k] * vdq(k, col) + Sl[3 * r + k] * vdq(3 + k, col);
          dv += Sw[3 * r + k] * J(k, col) + Sl[3 * r + k] * J(3 + k, col);
        }
        dq += wv * vdq(r, col);
        dv += wv * J(r, col);
        real RJ = 0; for (int k = 0; k < 3; ++k) RJ += R[3 * r + k] * J(k, col);
        dq += wp * RJ;
        dCdq(row + r, col) = dq; dCdv(row + r, col) = dv; dCda(row + r, col) = J(r, col);
      }
    }
    row += 3;
  }
}

void Robot::computeImpulseVelocityResidual(const std::vector<bool>& active, Mat& C) const {
  FLOP_REGION(R_BAUMGARTE);
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  C = Mat(3 * na);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    real v[6];
    contactFrame(c, nullptr, nullptr, v, nullptr);
    for (int k = 0; k < 3; ++k) C[row + k] = v[k];
    row += 3;
  }
}

void Robot::computeImpulseVelocityDerivatives(const std::vector<bool>& active, Mat& dCdq, Mat& dCdv) const {
  FLOP_REGION(R_BAUMGARTE);
  const int nv = m_.nv;
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  dCdq = Mat(3 * na, nv); dCdv = Mat(3 * na, nv);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    Mat vdq, adq, adv, J;
    frameDerivatives(c, vdq, adq, adv, J);
    for (int col = 0; col < nv; ++col) for (int r = 0; r < 3; ++r) { dCdq(row + r, col) = vdq(r, col); dCdv(row + r, col) = J(r, col); }
    row += 3;
  }
}

void Robot::computeContactResidual(const std::vector<bool>& active, const std::vector<Mat>& contact_points, Mat& P) const {
  FLOP_REGION(R_BAUMGARTE);
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  P = Mat(3 * na);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    real p[3];
    contactFrame(c, p, nullptr, nullptr, nullptr);
    for (int k = 0; k < 3; ++k) P[row + k] = p[k] - contact_points[c][k];
    row += 3;
  }
}

void Robot::computeContactDerivative(const std::vector<bool>& active, Mat& Pq) const {
  FLOP_REGION(R_BAUMGARTE);
  const int nv = m_.nv;
  int na = 0; for (int c = 0; c < m_.ncontacts; ++c) if (active[c]) ++na;
  Pq = Mat(3 * na, nv);
  int row = 0;
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    Mat vdq, adq, adv, J;
    frameDerivatives(c, vdq, adq, adv, J);
    real R[9];
    contactFrame(c, nullptr, R, nullptr, nullptr);
    for (int col = 0; col < nv; ++col)
      for (int r = 0; r < 3; ++r) { real acc = 0; for (int k = 0; k < 3; ++k) acc += R[3 * r + k] * J(k, col); Pq(row + r, col) = acc; }
    row += 3;
  }
}

static void log3(const real* R, real* w, real* theta);
static void VmatInv(const real* w, real* Vi);
static void Jlog6(const real* R, const real* p, Mat& J);
void Robot::taskSpaceTerms(int dim, const real* ref, const real* w, const Mat& q, real& cost, Mat& grad, Mat& hess) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const int nv = m_.nv;
  const Mat zero(nv);
  updateKinematics(q, zero, zero);
  real R[9], p[3];
  contactFrame(0, p, R, nullptr, nullptr);             // Robot::framePlacement (robot.hxx:166-178)
  Mat vdq, adq, adv, J;
  frameDerivatives(0, vdq, adq, adv, J);                // J = getFrameJacobian(LOCAL) (robot.hxx:181-188), rows: linear, angular
  const int m = dim == 6 ? 6 : 3;
  Mat JJ(m, nv), diff(m);
  if (dim == 6) {
    // diff_SE3 = SE3_ref^-1 * oMf ; diff_6d = log6(diff_SE3) = [linear; angular] ; J_66 = Jlog6(diff_SE3)
    real Rd[9], pd[3], t[3] = {p[0] - ref[9], p[1] - ref[10], p[2] - ref[11]};
    for (int r = 0; r < 3; ++r) {
      for (int c = 0; c < 3; ++c) { real a = 0; for (int k = 0; k < 3; ++k) a += ref[3 * k + r] * R[3 * k + c]; Rd[3 * r + c] = a; }
      pd[r] = ref[r] * t[0] + ref[3 + r] * t[1] + ref[6 + r] * t[2];
    }
    real wv[3], th, Vi[9], vl[3];
    log3(Rd, wv, &th); VmatInv(wv, Vi); matvec3(Vi, pd, vl);
    for (int k = 0; k < 3; ++k) { diff[k] = vl[k]; diff[3 + k] = wv[k]; }
    Mat J66; Jlog6(Rd, pd, J66);
    JJ = J66 * J;
  } else {
    for (int k = 0; k < 3; ++k) diff[k] = p[k] - ref[9 + k];
    for (int col = 0; col < nv; ++col)
      for (int r = 0; r < 3; ++r) { real a = 0; for (int k = 0; k < 3; ++k) a += R[3 * r + k] * J(k, col); JJ(r, col) = a; }
  }
  cost = 0;
  for (int r = 0; r < m; ++r) cost += 0.5 * w[r] * diff[r] * diff[r];
  grad = Mat(nv); hess = Mat(nv, nv);
  for (int c = 0; c < nv; ++c) {
    real g = 0;
    for (int r = 0; r < m; ++r) g += JJ(r, c) * w[r] * diff[r];
    grad[c] = g;
    for (int c2 = 0; c2 < nv; ++c2) { real h = 0; for (int r = 0; r < m; ++r) h += JJ(r, c) * w[r] * JJ(r, c2); hess(c, c2) = h; }
  }
}

void Robot::computeMJtJinv(const Mat& M, const Mat& J, Mat& out) {
  FLOP_REGION(R_MJTJINV);
  // robot.hxx:576-615.  pinocchio's sparse U D U^T factorisation of M is replaced
  // by a dense Cholesky (same solution); the block algebra follows the reference.
  const int nv = M.r, nf = J.r;
  LLT lltM; lltM.compute(M);
  Mat Minv = lltM.solve(Mat::Identity(nv));
  out = Mat(nv + nf, nv + nf);
  if (nf == 0) { out.setBlock(0, 0, Minv); return; }
  Mat JMinvJt = J * Minv * J.t();
  LLT llt; llt.compute(JMinvJt);
  Mat bottomRight = llt.solve(-1.0 * Mat::Identity(nf));          // -(J Minv Jt)^-1
  Mat bottomLeft = J * Minv;
  Mat topRight = bottomLeft.t() * (-bottomRight);
  Mat topLeft = Minv - topRight * bottomLeft;
  out.setBlock(0, 0, topLeft); out.setBlock(0, nv, topRight);
  out.setBlock(nv, 0, topRight.t()); out.setBlock(nv, nv, bottomRight);
}

// ------------------------------------------------------------- Lie group ----
static void exp3(const real* w, real* R) {
  const real t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  real a, b;                      // R = I + a K + b K^2
  if (t < 1e-8) { a = 1 - t2 / 6; b = 0.5 - t2 / 24; } else { a = std::sin(t) / t; b = (1 - std::cos(t)) / t2; }
  real K[9]; skew(w, K);
  real K2[9]; matmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) R[i] = a * K[i] + b * K2[i];
  R[0] += 1; R[4] += 1; R[8] += 1;
}
static void log3(const real* R, real* w, real* theta) {
  real c = (R[0] + R[4] + R[8] - 1) / 2; c = c > 1 ? real(1) : (c < -1 ? real(-1) : c);
  const real t = std::acos(c);
  const real ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const real s = t < 1e-8 ? 0.5 + t * t / 12 : t / (2 * std::sin(t));
  for (int k = 0; k < 3; ++k) w[k] = s * ax[k];
  *theta = t;
}
static void Vmat(const real* w, real* V) {   // exp6: t = V(w) v
  const real t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  real b, c;
  if (t < 1e-8) { b = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; } else { b = (1 - std::cos(t)) / t2; c = (t - std::sin(t)) / (t2 * t); }
  real K[9]; skew(w, K);
  real K2[9]; matmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) V[i] = b * K[i] + c * K2[i];
  V[0] += 1; V[4] += 1; V[8] += 1;
}
static void VmatInv(const real* w, real* Vi) {   // log6: v = V(w)^-1 t
  const real t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = std::sqrt(t2);
  real beta;
  if (t < 1e-4) beta = 1.0 / 12 + t2 / 720; else beta = 1 / t2 - std::sin(t) / (2 * t * (1 - std::cos(t)));
  real K[9]; skew(w, K);
  real K2[9]; matmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * K[i] + beta * K2[i];
  Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
}
static void RtoQuat(const real* R, real* q) {   // xyzw
  const real tr = R[0] + R[4] + R[8];
  if (tr > 0) { const real s = std::sqrt(tr + 1) * 2; q[3] = s / 4; q[0] = (R[7] - R[5]) / s; q[1] = (R[2] - R[6]) / s; q[2] = (R[3] - R[1]) / s; }
  else if (R[0] > R[4] && R[0] > R[8]) { const real s = std::sqrt(1 + R[0] - R[4] - R[8]) * 2; q[3] = (R[7] - R[5]) / s; q[0] = s / 4; q[1] = (R[1] + R[3]) / s; q[2] = (R[2] + R[6]) / s; }
  else if (R[4] > R[8]) { const real s = std::sqrt(1 + R[4] - R[0] - R[8]) * 2; q[3] = (R[2] - R[6]) / s; q[0] = (R[1] + R[3]) / s; q[1] = s / 4; q[2] = (R[5] + R[7]) / s; }
  else { const real s = std::sqrt(1 + R[8] - R[0] - R[4]) * 2; q[3] = (R[3] - R[1]) / s; q[0] = (R[2] + R[6]) / s; q[1] = (R[5] + R[7]) / s; q[2] = s / 4; }
}

// pinocchio::integrate (SpecialEuclideanOperation<3>: q (+) v = q * exp6(v))
void Robot::integrateConfiguration(const Mat& q, const Mat& v, real length, Mat& q_out) const {
  FLOP_REGION(R_LIE);
  Mat out = q;
  for (int i = 0; i < m_.njoints; ++i) {
    const int iq = m_.idx_q[i], iv = m_.idx_v[i];
    if (m_.jtype[i] == IDOCP_JOINT_REVOLUTE) { out[iq] = q[iq] + length * v[iv]; continue; }
    real R[9], w[3], vl[3], V[9], E[9], t[3], Rt[3], Rn[9], qt[4];
    quatToR(&q.d[iq + 3], R);
    for (int k = 0; k < 3; ++k) { vl[k] = length * v[iv + k]; w[k] = length * v[iv + 3 + k]; }
    Vmat(w, V); matvec3(V, vl, t); matvec3(R, t, Rt);
    exp3(w, E); matmul3(R, E, Rn); RtoQuat(Rn, qt);
    // keep the quaternion on the same hemisphere as the input
    real dotq = 0; for (int k = 0; k < 4; ++k) dotq += qt[k] * q[iq + 3 + k];
    const real sgn = dotq < 0 ? -1.0 : 1.0;
    real nrm = 0; for (int k = 0; k < 4; ++k) nrm += qt[k] * qt[k];
    nrm = std::sqrt(nrm);
    for (int k = 0; k < 3; ++k) out[iq + k] = q[iq + k] + Rt[k];
    for (int k = 0; k < 4; ++k) out[iq + 3 + k] = sgn * qt[k] / nrm;
  }
  q_out = out;
}

// M = M_minus^-1 M_plus of the free-flyer (R, p)
static void relativePlacement(const real* qm, const real* qp, real* R, real* p) {
  real Rm[9], Rp[9], Rmt[9], d[3];
  quatToR(qm + 3, Rm); quatToR(qp + 3, Rp);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rmt[3 * r + s] = Rm[3 * s + r];
  matmul3(Rmt, Rp, R);
  for (int k = 0; k < 3; ++k) d[k] = qp[k] - qm[k];
  matvec3(Rmt, d, p);
}

// pinocchio::difference(q_minus, q_plus) = log6(q_minus^-1 q_plus)
void Robot::subtractConfiguration(const Mat& q_plus, const Mat& q_minus, Mat& diff) const {
  FLOP_REGION(R_LIE);
  diff = Mat(m_.nv);
  for (int i = 0; i < m_.njoints; ++i) {
    const int iq = m_.idx_q[i], iv = m_.idx_v[i];
    if (m_.jtype[i] == IDOCP_JOINT_REVOLUTE) { diff[iv] = q_plus[iq] - q_minus[iq]; continue; }
    real R[9], p[3], w[3], th, Vi[9], vl[3];
    relativePlacement(&q_minus.d[iq], &q_plus.d[iq], R, p);
    log3(R, w, &th); VmatInv(w, Vi); matvec3(Vi, p, vl);
    for (int k = 0; k < 3; ++k) { diff[iv + k] = vl[k]; diff[iv + 3 + k] = w[k]; }
  }
}

// Jlog6 of pinocchio's explog.hpp: derivative of log6(M exp6(d)) w.r.t. d.
static void Jlog6(const real* R, const real* p, Mat& J) {
  real w[3], t;
  log3(R, w, &t);
  const real t2 = t * t;
  real alpha, diag;
  if (t < 1e-4) { alpha = 1.0 / 12 + t2 / 720; diag = 0.5 * (2 - t2 / 6); }
  else { const real st = std::sin(t), ct = std::cos(t), st_1mct = st / (1 - ct); alpha = 1 / t2 - st_1mct / (2 * t); diag = 0.5 * t * st_1mct; }
  real A[9];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) A[3 * r + s] = alpha * w[r] * w[s];
  A[0] += diag; A[4] += diag; A[8] += diag;
  real Kw[9]; skew(w, Kw);
  for (int k = 0; k < 9; ++k) A[k] += 0.5 * Kw[k];
  real beta, bdot;
  if (t < 1e-4) { beta = 1.0 / 12 + t2 / 720; bdot = 1.0 / 360; }
  else {
    const real tinv = 1 / t, t2inv = tinv * tinv, st = std::sin(t), ct = std::cos(t), inv22ct = 1 / (2 * (1 - ct));
    beta = t2inv - st * tinv * inv22ct;
    bdot = -2 * t2inv * t2inv + (1 + st * tinv) * t2inv * inv22ct;
  }
  const real wTp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  real v3[3];
  for (int k = 0; k < 3; ++k) v3[k] = (bdot * wTp) * w[k] - (t2 * bdot + 2 * beta) * p[k];
  real Cm[9];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Cm[3 * r + s] = v3[r] * w[s] + beta * w[r] * p[s];
  Cm[0] += wTp * beta; Cm[4] += wTp * beta; Cm[8] += wTp * beta;
  real Kp[9]; skew(p, Kp);
  for (int k = 0; k < 9; ++k) Cm[k] += 0.5 * Kp[k];
  real B[9]; matmul3(Cm, A, B);
  J = Mat(6, 6);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) { J(r, s) = A[3 * r + s]; J(3 + r, 3 + s) = A[3 * r + s]; J(r, 3 + s) = B[3 * r + s]; }
}

// pinocchio::dDifference(q_minus, q_plus, ARG1)
void Robot::dSubtractdConfigurationPlus(const Mat& q_plus, const Mat& q_minus, Mat& J) const {
  FLOP_REGION(R_LIE);
  J = Mat::Identity(m_.nv);
  for (int i = 0; i < m_.njoints; ++i) {
    if (m_.jtype[i] != IDOCP_JOINT_FREEFLYER) continue;
    real R[9], p[3]; Mat J6;
    relativePlacement(&q_minus.d[m_.idx_q[i]], &q_plus.d[m_.idx_q[i]], R, p);
    Jlog6(R, p, J6);
    J.setBlock(m_.idx_v[i], m_.idx_v[i], J6);
  }
}

// pinocchio::dDifference(q_minus, q_plus, ARG0) = -Jlog6(M) Ad(M^-1),  M = q_minus^-1 q_plus
void Robot::dSubtractdConfigurationMinus(const Mat& q_plus, const Mat& q_minus, Mat& J) const {
  FLOP_REGION(R_LIE);
  J = -1.0 * Mat::Identity(m_.nv);
  for (int i = 0; i < m_.njoints; ++i) {
    if (m_.jtype[i] != IDOCP_JOINT_FREEFLYER) continue;
    real R[9], p[3]; Mat J6;
    relativePlacement(&q_minus.d[m_.idx_q[i]], &q_plus.d[m_.idx_q[i]], R, p);
    Jlog6(R, p, J6);
    // action matrix of M^-1 = (R^T, -R^T p): [[R^T, [-R^T p]x R^T],[0, R^T]]
    real Rt[9], mp[3], K[9], KRt[9];
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
    matvec3(Rt, p, mp); for (int k = 0; k < 3; ++k) mp[k] = -mp[k];
    skew(mp, K); matmul3(K, Rt, KRt);
    Mat Ad(6, 6);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) { Ad(r, s) = Rt[3 * r + s]; Ad(3 + r, 3 + s) = Rt[3 * r + s]; Ad(r, 3 + s) = KRt[3 * r + s]; }
    J.setBlock(m_.idx_v[i], m_.idx_v[i], -1.0 * (J6 * Ad));
  }
}

// exp6(v) of the free-flyer tangent v = (lin, ang): (R, p) = (exp3(w), V(w) lin)
static void exp6(const real* v6, real* R, real* p) {
  real V[9];
  exp3(v6 + 3, R); Vmat(v6 + 3, V); matvec3(V, v6, p);
}

// pinocchio::dIntegrate(q, v, ARG0): SpecialEuclideanOperation<3>::dIntegrate_dq_impl = exp6(v).toActionMatrixInverse()
void Robot::dIntegratedConfiguration(const Mat& /*q*/, const Mat& v, Mat& J) const {
  FLOP_REGION(R_LIE);
  J = Mat::Identity(m_.nv);
  for (int i = 0; i < m_.njoints; ++i) {
    if (m_.jtype[i] != IDOCP_JOINT_FREEFLYER) continue;
    real R[9], p[3], Rt[9], mp[3], K[9], KRt[9];
    exp6(&v.d[m_.idx_v[i]], R, p);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
    matvec3(Rt, p, mp); for (int k = 0; k < 3; ++k) mp[k] = -mp[k];
    skew(mp, K); matmul3(K, Rt, KRt);
    Mat Ad(6, 6);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) { Ad(r, s) = Rt[3 * r + s]; Ad(3 + r, 3 + s) = Rt[3 * r + s]; Ad(r, 3 + s) = KRt[3 * r + s]; }
    J.setBlock(m_.idx_v[i], m_.idx_v[i], Ad);
  }
}

// pinocchio::dIntegrate(q, v, ARG1): dIntegrate_dv_impl = Jexp6(v) = Jlog6(exp6(v))^-1
void Robot::dIntegratedVelocity(const Mat& /*q*/, const Mat& v, Mat& J) const {
  FLOP_REGION(R_LIE);
  J = Mat::Identity(m_.nv);
  for (int i = 0; i < m_.njoints; ++i) {
    if (m_.jtype[i] != IDOCP_JOINT_FREEFLYER) continue;
    real R[9], p[3];
    exp6(&v.d[m_.idx_v[i]], R, p);
    Mat Jl, Je;
    Jlog6(R, p, Jl);
    dSubtractdConfigurationInverse(Jl, Je);       // Jlog6 is block upper-triangular [A B; 0 A]
    J.setBlock(m_.idx_v[i], m_.idx_v[i], Je);
  }
}

// Robot::dSubtractdConfigurationInverse (robot.hxx:151-163): block-triangular 6x6 inverse
void Robot::dSubtractdConfigurationInverse(const Mat& J, Mat& Jinv) {
  FLOP_REGION(R_LIE);
  if (J.r == 0) { Jinv = Mat(0, 0); return; }      // fixed base: no passive rows
  auto inv3 = [](const Mat& A) {
    Mat I(3, 3);
    const real det = A(0, 0) * (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) - A(0, 1) * (A(1, 0) * A(2, 2) - A(1, 2) * A(2, 0)) +
                       A(0, 2) * (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0));
    I(0, 0) = (A(1, 1) * A(2, 2) - A(1, 2) * A(2, 1)) / det; I(0, 1) = (A(0, 2) * A(2, 1) - A(0, 1) * A(2, 2)) / det; I(0, 2) = (A(0, 1) * A(1, 2) - A(0, 2) * A(1, 1)) / det;
    I(1, 0) = (A(1, 2) * A(2, 0) - A(1, 0) * A(2, 2)) / det; I(1, 1) = (A(0, 0) * A(2, 2) - A(0, 2) * A(2, 0)) / det; I(1, 2) = (A(0, 2) * A(1, 0) - A(0, 0) * A(1, 2)) / det;
    I(2, 0) = (A(1, 0) * A(2, 1) - A(1, 1) * A(2, 0)) / det; I(2, 1) = (A(0, 1) * A(2, 0) - A(0, 0) * A(2, 1)) / det; I(2, 2) = (A(0, 0) * A(1, 1) - A(0, 1) * A(1, 0)) / det;
    return I;
  };
  Jinv = Mat(6, 6);
  Mat TLi = inv3(J.block(0, 0, 3, 3)), BRi = inv3(J.block(3, 3, 3, 3));
  Jinv.setBlock(0, 0, TLi); Jinv.setBlock(3, 3, BRi);
  Jinv.setBlock(0, 3, -1.0 * (TLi * (J.block(0, 3, 3, 3) * BRi)));
}

}  // namespace oracle
