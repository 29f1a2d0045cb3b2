// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md and rbd.hpp).
#include "rbd.hpp"

#include <cmath>
#include <cstring>

namespace oracle {

// ------------------------------------------------------------ helpers ----
static inline void cross3(const double* a, const double* b, double* c) {
  const double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
  c[0] = x; c[1] = y; c[2] = z;
}
static inline void matvec3(const double* R, const double* x, double* y) {
  const double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2];
  const double b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2];
  const double c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
static inline void matmul3(const double* A, const double* B, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j)
    T[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
  std::memcpy(C, T, sizeof(T));
}
// motion x motion:  (v,w) x (v2,w2) = (w x v2 + v x w2, w x w2)
static inline void crossMM(const double* a, const double* b, double* o) {
  double t1[3], t2[3], t3[3];
  cross3(a + 3, b, t1); cross3(a, b + 3, t2); cross3(a + 3, b + 3, t3);
  for (int k = 0; k < 3; ++k) { o[k] = t1[k] + t2[k]; o[3 + k] = t3[k]; }
}
// motion x* force:  (v,w) x* (f,n) = (w x f, w x n + v x f)
static inline void crossMF(const double* m, const double* f, double* o) {
  double t1[3], t2[3], t3[3];
  cross3(m + 3, f, t1); cross3(m + 3, f + 3, t2); cross3(m, f, t3);
  for (int k = 0; k < 3; ++k) { o[k] = t1[k]; o[3 + k] = t2[k] + t3[k]; }
}
static inline void skew(const double* v, double* S) {   // row-major 3x3
  S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
static Mat crm(const double* v) {        // 6x6: crm(v) m = v x m
  Mat X(6, 6); double Sv[9], Sw[9]; skew(v, Sv); skew(v + 3, Sw);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    X(i, j) = Sw[3 * i + j]; X(i, 3 + j) = Sv[3 * i + j]; X(3 + i, 3 + j) = Sw[3 * i + j];
  }
  return X;
}
static Mat crf(const double* v) {        // 6x6: crf(v) f = v x* f
  Mat X(6, 6); double Sv[9], Sw[9]; skew(v, Sv); skew(v + 3, Sw);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
    X(i, j) = Sw[3 * i + j]; X(3 + i, j) = Sv[3 * i + j]; X(3 + i, 3 + j) = Sw[3 * i + j];
  }
  return X;
}
static inline double dot6(const double* a, const double* b) {
  return a[0] * b[0] + a[1] * b[1] + a[2] * b[2] + a[3] * b[3] + a[4] * b[4] + a[5] * b[5];
}

static void rotAxis(const double* u, double q, double* R) {
  const double c = std::cos(q), s = std::sin(q), t = 1 - c;
  R[0] = c + t * u[0] * u[0];        R[1] = t * u[0] * u[1] - s * u[2]; R[2] = t * u[0] * u[2] + s * u[1];
  R[3] = t * u[1] * u[0] + s * u[2]; R[4] = c + t * u[1] * u[1];        R[5] = t * u[1] * u[2] - s * u[0];
  R[6] = t * u[2] * u[0] - s * u[1]; R[7] = t * u[2] * u[1] + s * u[0]; R[8] = c + t * u[2] * u[2];
}
static void quatToR(const double* qt, double* R) {  // xyzw
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}

// ---------------------------------------------------------------- Robot ----
Robot::Robot(const idocp_model_t& model) : m_(model) {
  const int n = m_.njoints, nv = m_.nv;
  fjoint_.assign(n, std::vector<double>(6, 0.0));
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
  // PointContact::computeJointForceFromContactForce (point_contact.hxx:15-20):
  // fjoint[parent] = jXf.act(Force(f, 0))
  for (auto& fj : fjoint_) std::fill(fj.begin(), fj.end(), 0.0);
  for (int c = 0; c < m_.ncontacts; ++c) {
    if (!active[c]) continue;
    double fl[3], n[3];
    matvec3(m_.contact_R[c], f[c].d.data(), fl);
    cross3(m_.contact_p[c], fl, n);
    auto& fj = fjoint_[m_.contact_joint[c]];
    for (int k = 0; k < 3; ++k) { fj[k] = fl[k]; fj[3 + k] = n[k]; }
  }
}

void Robot::jointPlacement(int i, const Mat& q, SE3& M) const {
  double Rj[9], pj[3] = {0, 0, 0};
  if (m_.jtype[i] == IDOCP_JOINT_REVOLUTE) {
    rotAxis(m_.axis[i], q[m_.idx_q[i]], Rj);
  } else {
    const int iq = m_.idx_q[i];
    quatToR(&q.d[iq + 3], Rj);
    pj[0] = q[iq]; pj[1] = q[iq + 1]; pj[2] = q[iq + 2];
  }
  matmul3(m_.plc_R[i], Rj, M.R);
  double t[3]; matvec3(m_.plc_R[i], pj, t);
  for (int k = 0; k < 3; ++k) M.p[k] = m_.plc_p[i][k] + t[k];
}

// World-frame forward pass (ComputeRNEADerivativesForwardStep of pinocchio's
// rnea-derivatives): oMi, J columns, ov, oa_gf, oYcrb, doYcrb, of, and the
// column sets dVdq, dAdq, dAdv.
void Robot::forwardPass(const Mat& q, const Mat& v, const Mat& a, bool gravity) {
  const int n = m_.njoints;
  double a0[6] = {0, 0, 0, 0, 0, 0};
  if (gravity) for (int k = 0; k < 3; ++k) a0[k] = -m_.gravity[k];
  for (int i = 0; i < n; ++i) {
    const int pa = m_.parent[i];
    SE3 li; jointPlacement(i, q, li);
    if (pa >= 0) {
      matmul3(oMi_[pa].R, li.R, oMi_[i].R);
      double t[3]; matvec3(oMi_[pa].R, li.p, t);
      for (int k = 0; k < 3; ++k) oMi_[i].p[k] = oMi_[pa].p[k] + t[k];
    } else {
      oMi_[i] = li;
    }
    const double* R = oMi_[i].R; const double* p = oMi_[i].p;
    const int iv = m_.idx_v[i];
    const int ndof = m_.jtype[i] == IDOCP_JOINT_FREEFLYER ? 6 : 1;
    // J columns = oMi.act(S)
    if (ndof == 1) {
      double w[3], l[3]; matvec3(R, m_.axis[i], w); cross3(p, w, l);
      for (int k = 0; k < 3; ++k) { S_(k, iv) = l[k]; S_(3 + k, iv) = w[k]; }
    } else {
      for (int c = 0; c < 3; ++c) {
        double e[3] = {0, 0, 0}; e[c] = 1; double Re[3], l[3];
        matvec3(R, e, Re); cross3(p, Re, l);
        for (int k = 0; k < 3; ++k) {
          S_(k, iv + c) = Re[k]; S_(3 + k, iv + c) = 0;
          S_(k, iv + 3 + c) = l[k]; S_(3 + k, iv + 3 + c) = Re[k];
        }
      }
    }
    double vJ[6] = {0, 0, 0, 0, 0, 0}, aJ[6] = {0, 0, 0, 0, 0, 0};
    for (int c = 0; c < ndof; ++c) for (int k = 0; k < 6; ++k) {
      vJ[k] += S_(k, iv + c) * v[iv + c]; aJ[k] += S_(k, iv + c) * a[iv + c];
    }
    const double* ovp = pa >= 0 ? ov_[pa].d.data() : nullptr;
    const double* oap = pa >= 0 ? oa_[pa].d.data() : a0;
    for (int k = 0; k < 6; ++k) ov_[i][k] = (ovp ? ovp[k] : 0.0) + vJ[k];
    double vxvJ[6]; crossMM(ov_[i].d.data(), vJ, vxvJ);
    for (int k = 0; k < 6; ++k) oa_[i][k] = oap[k] + aJ[k] + vxvJ[k];
    // oYcrb = oMi.act(inertia) as a 6x6 matrix about the world origin
    double c[3], Rc[3]; matvec3(R, m_.com[i], Rc);
    for (int k = 0; k < 3; ++k) c[k] = p[k] + Rc[k];
    double RI[9], Rt[9], Iw[9];
    matmul3(R, m_.inertia[i], RI);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
    matmul3(RI, Rt, Iw);
    const double mass = m_.mass[i];
    double Sc[9]; skew(c, Sc);
    Mat& Y = oY_[i]; Y.setZero();
    for (int r = 0; r < 3; ++r) {
      Y(r, r) = mass;
      for (int s = 0; s < 3; ++s) {
        Y(r, 3 + s) = -mass * Sc[3 * r + s];
        Y(3 + r, s) = mass * Sc[3 * r + s];
        double scsc = 0; for (int k = 0; k < 3; ++k) scsc += Sc[3 * r + k] * Sc[3 * k + s];
        Y(3 + r, 3 + s) = Iw[3 * r + s] - mass * scsc;
      }
    }
    Mat oh = Y * ov_[i];
    Mat Ya = Y * oa_[i];
    double vxh[6]; crossMF(ov_[i].d.data(), oh.d.data(), vxh);
    // external force: of -= oMi.act(fext)
    double fw[3], nw[3], pxf[3];
    matvec3(R, fjoint_[i].data(), fw); matvec3(R, fjoint_[i].data() + 3, nw); cross3(p, fw, pxf);
    for (int k = 0; k < 3; ++k) {
      of_[i][k] = Ya[k] + vxh[k] - fw[k];
      of_[i][3 + k] = Ya[3 + k] + vxh[3 + k] - (nw[k] + pxf[k]);
    }
    // doYcrb = oYcrb.variation(ov) + forceCrossMatrix(oh)
    Mat B = crf(ov_[i].d.data()) * Y - Y * crm(ov_[i].d.data());
    double Sf[9], Sn[9]; skew(oh.d.data(), Sf); skew(oh.d.data() + 3, Sn);
    for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
      B(r, 3 + s) -= Sf[3 * r + s]; B(3 + r, s) -= Sf[3 * r + s]; B(3 + r, 3 + s) -= Sn[3 * r + s];
    }
    oB_[i] = B;
    for (int cidx = 0; cidx < ndof; ++cidx) {
      const int col = iv + cidx;
      double Sk[6], dJ[6], dV[6] = {0, 0, 0, 0, 0, 0}, dA[6], t[6];
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
      double t[6]; crossMF(&S_.d[6 * (iv + c)], of_[i].d.data(), t);
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
