// SE(3) Lie-group pieces of the floating base (host + device; on the device single-lane scalar
// code; executed by one lane per task inside the condensation / integration
// kernels).  Same closed forms as pinocchio's explog.hpp (log3, Jlog3, Jlog6,
// exp6), which the reference reaches through pinocchio::difference /
// dDifference / integrate (include/idocp/robot/robot.hxx:23-163).
#ifndef IDOCP_DEV_LIE_HPP_
#define IDOCP_DEV_LIE_HPP_

#include <hip/hip_runtime.h>

namespace idocp_dev {

__host__ __device__ __forceinline__ void lieQuatToR(const double* qt, double* R) {
  const double x = qt[0], y = qt[1], z = qt[2], w = qt[3];
  R[0] = 1 - 2 * (y * y + z * z); R[1] = 2 * (x * y - z * w);     R[2] = 2 * (x * z + y * w);
  R[3] = 2 * (x * y + z * w);     R[4] = 1 - 2 * (x * x + z * z); R[5] = 2 * (y * z - x * w);
  R[6] = 2 * (x * z - y * w);     R[7] = 2 * (y * z + x * w);     R[8] = 1 - 2 * (x * x + y * y);
}
__host__ __device__ __forceinline__ void lieMatmul3(const double* A, const double* Bm, double* C) {
  double T[9];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) T[3 * i + j] = A[3 * i] * Bm[j] + A[3 * i + 1] * Bm[3 + j] + A[3 * i + 2] * Bm[6 + j];
  for (int i = 0; i < 9; ++i) C[i] = T[i];
}
__host__ __device__ __forceinline__ void lieMatvec3(const double* R, const double* x, double* y) {
  const double a = R[0] * x[0] + R[1] * x[1] + R[2] * x[2], b = R[3] * x[0] + R[4] * x[1] + R[5] * x[2],
               c = R[6] * x[0] + R[7] * x[1] + R[8] * x[2];
  y[0] = a; y[1] = b; y[2] = c;
}
__host__ __device__ __forceinline__ void lieSkew(const double* v, double* S) {
  S[0] = 0; S[1] = -v[2]; S[2] = v[1]; S[3] = v[2]; S[4] = 0; S[5] = -v[0]; S[6] = -v[1]; S[7] = v[0]; S[8] = 0;
}
__host__ __device__ __forceinline__ void lieLog3(const double* R, double* w, double* theta) {
  double c = (R[0] + R[4] + R[8] - 1) / 2; c = c > 1 ? 1 : (c < -1 ? -1 : c);
  const double t = acos(c);
  const double ax[3] = {R[7] - R[5], R[2] - R[6], R[3] - R[1]};
  const double s = t < 1e-8 ? 0.5 + t * t / 12 : t / (2 * sin(t));
  for (int k = 0; k < 3; ++k) w[k] = s * ax[k];
  *theta = t;
}
// M = M_minus^-1 M_plus of two free-flyer configurations (xyz + quat xyzw)
__host__ __device__ __forceinline__ void lieRelative(const double* qm, const double* qp, double* R, double* p) {
  double Rm[9], Rp[9], Rmt[9], d[3];
  lieQuatToR(qm + 3, Rm); lieQuatToR(qp + 3, Rp);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rmt[3 * r + s] = Rm[3 * s + r];
  lieMatmul3(Rmt, Rp, R);
  for (int k = 0; k < 3; ++k) d[k] = qp[k] - qm[k];
  lieMatvec3(Rmt, d, p);
}
// log6(M) = (V(w)^-1 p, w)
__host__ __device__ __forceinline__ void lieLog6(const double* R, const double* p, double* out) {
  double w[3], t;
  lieLog3(R, w, &t);
  const double t2 = t * t;
  const double beta = t < 1e-4 ? 1.0 / 12 + t2 / 720 : 1 / t2 - sin(t) / (2 * t * (1 - cos(t)));
  double K[9], K2[9], Vi[9];
  lieSkew(w, K); lieMatmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * K[i] + beta * K2[i];
  Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
  lieMatvec3(Vi, p, out);
  out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
}
// Jlog6(M): 6x6, written column-major with leading dimension 6
__host__ __device__ __forceinline__ void lieJlog6(const double* R, const double* p, double* J) {
  double w[3], t;
  lieLog3(R, w, &t);
  const double t2 = t * t;
  double alpha, diag;
  if (t < 1e-4) { alpha = 1.0 / 12 + t2 / 720; diag = 0.5 * (2 - t2 / 6); }
  else { const double st = sin(t), ct = cos(t), q = st / (1 - ct); alpha = 1 / t2 - q / (2 * t); diag = 0.5 * t * q; }
  double A[9], Kw[9];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) A[3 * r + s] = alpha * w[r] * w[s];
  A[0] += diag; A[4] += diag; A[8] += diag;
  lieSkew(w, Kw);
  for (int k = 0; k < 9; ++k) A[k] += 0.5 * Kw[k];
  double beta, bdot;
  if (t < 1e-4) { beta = 1.0 / 12 + t2 / 720; bdot = 1.0 / 360; }
  else {
    const double tinv = 1 / t, t2inv = tinv * tinv, st = sin(t), ct = cos(t), i22 = 1 / (2 * (1 - ct));
    beta = t2inv - st * tinv * i22;
    bdot = -2 * t2inv * t2inv + (1 + st * tinv) * t2inv * i22;
  }
  const double wTp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double v3[3], Cm[9], Kp[9], Bm[9];
  for (int k = 0; k < 3; ++k) v3[k] = (bdot * wTp) * w[k] - (t2 * bdot + 2 * beta) * p[k];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Cm[3 * r + s] = v3[r] * w[s] + beta * w[r] * p[s];
  Cm[0] += wTp * beta; Cm[4] += wTp * beta; Cm[8] += wTp * beta;
  lieSkew(p, Kp);
  for (int k = 0; k < 9; ++k) Cm[k] += 0.5 * Kp[k];
  lieMatmul3(Cm, A, Bm);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
    J[r + 6 * s] = A[3 * r + s]; J[3 + r + 6 * (3 + s)] = A[3 * r + s]; J[r + 6 * (3 + s)] = Bm[3 * r + s]; J[3 + r + 6 * s] = 0.0;
  }
}
// log6(M) and Jlog6(M) together: the angle, sin, cos and the Taylor switches are evaluated once (same closed forms as
// lieLog6 / lieJlog6 above; used where both are needed, dev_task.hpp)
__host__ __device__ __forceinline__ void lieLog6Jlog6(const double* R, const double* p, double* out, double* J) {
  double w[3], t;
  lieLog3(R, w, &t);
  const double t2 = t * t;
  double alpha, diag, beta, bdot;
  if (t < 1e-4) { alpha = beta = 1.0 / 12 + t2 / 720; diag = 0.5 * (2 - t2 / 6); bdot = 1.0 / 360; }
  else {
    const double st = sin(t), ct = cos(t), tinv = 1 / t, t2inv = tinv * tinv, i22 = 1 / (2 * (1 - ct)), q = st / (1 - ct);
    alpha = 1 / t2 - q / (2 * t); diag = 0.5 * t * q;
    beta = t2inv - st * tinv * i22;
    bdot = -2 * t2inv * t2inv + (1 + st * tinv) * t2inv * i22;
  }
  // log6: beta of lieLog6 is 1/t^2 - sin t / (2 t (1 - cos t)) = the beta above
  double K[9], K2[9];
  lieSkew(w, K); lieMatmul3(K, K, K2);
  {
    double Vi[9];
    for (int i = 0; i < 9; ++i) Vi[i] = -0.5 * K[i] + beta * K2[i];
    Vi[0] += 1; Vi[4] += 1; Vi[8] += 1;
    lieMatvec3(Vi, p, out);
    out[3] = w[0]; out[4] = w[1]; out[5] = w[2];
  }
  double A[9];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) A[3 * r + s] = alpha * w[r] * w[s];
  A[0] += diag; A[4] += diag; A[8] += diag;
  for (int k = 0; k < 9; ++k) A[k] += 0.5 * K[k];
  const double wTp = w[0] * p[0] + w[1] * p[1] + w[2] * p[2];
  double v3[3], Cm[9], Kp[9], Bm[9];
  for (int k = 0; k < 3; ++k) v3[k] = (bdot * wTp) * w[k] - (t2 * bdot + 2 * beta) * p[k];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Cm[3 * r + s] = v3[r] * w[s] + beta * w[r] * p[s];
  Cm[0] += wTp * beta; Cm[4] += wTp * beta; Cm[8] += wTp * beta;
  lieSkew(p, Kp);
  for (int k = 0; k < 9; ++k) Cm[k] += 0.5 * Kp[k];
  lieMatmul3(Cm, A, Bm);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
    J[r + 6 * s] = A[3 * r + s]; J[3 + r + 6 * (3 + s)] = A[3 * r + s]; J[r + 6 * (3 + s)] = Bm[3 * r + s]; J[3 + r + 6 * s] = 0.0;
  }
}
// dDifference ARG0 = -Jlog6(M) Ad(M^-1); J1 = Jlog6(M) given (col-major 6x6) -> J0 (col-major 6x6)
__host__ __device__ __forceinline__ void lieDDiffArg0(const double* R, const double* p, const double* J1, double* J0) {
  double Rt[9], mp[3], K[9], KRt[9], Ad[36];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
  lieMatvec3(Rt, p, mp);
  for (int k = 0; k < 3; ++k) mp[k] = -mp[k];
  lieSkew(mp, K); lieMatmul3(K, Rt, KRt);
  for (int k = 0; k < 36; ++k) Ad[k] = 0.0;
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) { Ad[r + 6 * s] = Rt[3 * r + s]; Ad[3 + r + 6 * (3 + s)] = Rt[3 * r + s]; Ad[r + 6 * (3 + s)] = KRt[3 * r + s]; }
  for (int c = 0; c < 6; ++c) for (int r = 0; r < 6; ++r) {
    double acc = 0.0;
    for (int k = 0; k < 6; ++k) acc += J1[r + 6 * k] * Ad[k + 6 * c];
    J0[r + 6 * c] = -acc;
  }
}
// Robot::dSubtractdConfigurationInverse (robot.hxx:151-163): inverse of [[A, B],[0, D]] (col-major 6x6 in / out)
__host__ __device__ __forceinline__ void lieBlockInverse(const double* J, double* Ji) {
  auto inv3 = [](const double* A /*col-major ld 6*/, double* I /*row-major 3x3*/) {
    const double a00 = A[0], a01 = A[6], a02 = A[12], a10 = A[1], a11 = A[7], a12 = A[13], a20 = A[2], a21 = A[8], a22 = A[14];
    const double det = a00 * (a11 * a22 - a12 * a21) - a01 * (a10 * a22 - a12 * a20) + a02 * (a10 * a21 - a11 * a20);
    I[0] = (a11 * a22 - a12 * a21) / det; I[1] = (a02 * a21 - a01 * a22) / det; I[2] = (a01 * a12 - a02 * a11) / det;
    I[3] = (a12 * a20 - a10 * a22) / det; I[4] = (a00 * a22 - a02 * a20) / det; I[5] = (a02 * a10 - a00 * a12) / det;
    I[6] = (a10 * a21 - a11 * a20) / det; I[7] = (a01 * a20 - a00 * a21) / det; I[8] = (a00 * a11 - a01 * a10) / det;
  };
  double TL[9], BR[9], Bm[9], T1[9], T2[9];
  inv3(J, TL); inv3(J + 3 + 18, BR);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Bm[3 * r + s] = J[r + 6 * (3 + s)];
  lieMatmul3(Bm, BR, T1); lieMatmul3(TL, T1, T2);
  for (int k = 0; k < 36; ++k) Ji[k] = 0.0;
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) { Ji[r + 6 * s] = TL[3 * r + s]; Ji[3 + r + 6 * (3 + s)] = BR[3 * r + s]; Ji[r + 6 * (3 + s)] = -T2[3 * r + s]; }
}
// q (+) length * v for the free-flyer part (pinocchio::integrate): p' = p + R V(w) v_lin ; quat' = quat(R exp3(w))
__host__ __device__ __forceinline__ void lieIntegrateBase(const double* q, const double* vin, double length, double* qout) {
  double R[9], w[3], vl[3];
  lieQuatToR(q + 3, R);
  for (int k = 0; k < 3; ++k) { vl[k] = length * vin[k]; w[k] = length * vin[3 + k]; }
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  double a, b, c;
  if (t < 1e-8) { a = 1 - t2 / 6; b = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
  else { a = sin(t) / t; b = (1 - cos(t)) / t2; c = (t - sin(t)) / (t2 * t); }
  double K[9], K2[9], V[9], E[9], tv[3], Rt[3], Rn[9];
  lieSkew(w, K); lieMatmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) { V[i] = b * K[i] + c * K2[i]; E[i] = a * K[i] + b * K2[i]; }
  V[0] += 1; V[4] += 1; V[8] += 1; E[0] += 1; E[4] += 1; E[8] += 1;
  lieMatvec3(V, vl, tv); lieMatvec3(R, tv, Rt);
  lieMatmul3(R, E, Rn);
  double qt[4];
  const double tr = Rn[0] + Rn[4] + Rn[8];
  if (tr > 0) { const double s = sqrt(tr + 1) * 2; qt[3] = s / 4; qt[0] = (Rn[7] - Rn[5]) / s; qt[1] = (Rn[2] - Rn[6]) / s; qt[2] = (Rn[3] - Rn[1]) / s; }
  else if (Rn[0] > Rn[4] && Rn[0] > Rn[8]) { const double s = sqrt(1 + Rn[0] - Rn[4] - Rn[8]) * 2; qt[3] = (Rn[7] - Rn[5]) / s; qt[0] = s / 4; qt[1] = (Rn[1] + Rn[3]) / s; qt[2] = (Rn[2] + Rn[6]) / s; }
  else if (Rn[4] > Rn[8]) { const double s = sqrt(1 + Rn[4] - Rn[0] - Rn[8]) * 2; qt[3] = (Rn[2] - Rn[6]) / s; qt[0] = (Rn[1] + Rn[3]) / s; qt[1] = s / 4; qt[2] = (Rn[5] + Rn[7]) / s; }
  else { const double s = sqrt(1 + Rn[8] - Rn[0] - Rn[4]) * 2; qt[3] = (Rn[3] - Rn[1]) / s; qt[0] = (Rn[2] + Rn[6]) / s; qt[1] = (Rn[5] + Rn[7]) / s; qt[2] = s / 4; }
  double dotq = 0, nrm = 0;
  for (int k = 0; k < 4; ++k) { dotq += qt[k] * q[3 + k]; nrm += qt[k] * qt[k]; }
  const double sgn = (dotq < 0 ? -1.0 : 1.0) / sqrt(nrm);
  for (int k = 0; k < 3; ++k) qout[k] = q[k] + Rt[k];
  for (int k = 0; k < 4; ++k) qout[3 + k] = sgn * qt[k];
}

// exp6 of a free-flyer tangent v = (lin, ang): (R, p) = (exp3(w), V(w) lin)   (pinocchio::exp6)
__host__ __device__ __forceinline__ void lieExp6(const double* vin, double* R, double* p) {
  const double* w = vin + 3;
  const double t2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2], t = sqrt(t2);
  double a, b, c;
  if (t < 1e-8) { a = 1 - t2 / 6; b = 0.5 - t2 / 24; c = 1.0 / 6 - t2 / 120; }
  else { a = sin(t) / t; b = (1 - cos(t)) / t2; c = (t - sin(t)) / (t2 * t); }
  double K[9], K2[9], V[9];
  lieSkew(w, K); lieMatmul3(K, K, K2);
  for (int i = 0; i < 9; ++i) { V[i] = b * K[i] + c * K2[i]; R[i] = a * K[i] + b * K2[i]; }
  V[0] += 1; V[4] += 1; V[8] += 1; R[0] += 1; R[4] += 1; R[8] += 1;
  lieMatvec3(V, vin, p);
}

// pinocchio::dIntegrate(q, v, ARG0) for the free-flyer: action matrix of exp6(v)^-1 (6x6 column-major)
__host__ __device__ __forceinline__ void lieDIntegrateArg0(const double* R, const double* p, double* A) {
  double Rt[9], mp[3], K[9], KRt[9];
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) Rt[3 * r + s] = R[3 * s + r];
  lieMatvec3(Rt, p, mp);
  for (int k = 0; k < 3; ++k) mp[k] = -mp[k];
  lieSkew(mp, K); lieMatmul3(K, Rt, KRt);
  for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) {
    A[r + 6 * s] = Rt[3 * r + s]; A[(3 + r) + 6 * (3 + s)] = Rt[3 * r + s]; A[r + 6 * (3 + s)] = KRt[3 * r + s]; A[(3 + r) + 6 * s] = 0.0;
  }
}

}  // namespace idocp_dev
#endif  // IDOCP_DEV_LIE_HPP_
