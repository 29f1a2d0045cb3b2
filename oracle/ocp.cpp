// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md and ocp.hpp).
#include "ocp.hpp"

#include <chrono>
#include <cmath>
#include <stdexcept>

namespace oracle {

static const int kP = 6;   // kDimFloatingBase

SplitSolutionC::SplitSolutionC(const Robot& r)
    : lmd(r.dimv()), gmm(r.dimv()), q(r.dimq()), v(r.dimv()), a(r.dimv()), u(r.dimu()), beta(r.dimv()), nu_passive(6),
      f(r.maxPointContacts(), Mat(3)), mu(r.maxPointContacts(), Mat(3)) {
  if (r.hasFloatingBase()) q[6] = 1.0;
}
Mat SplitSolutionC::f_stack(const ContactStatus& cs) const {
  Mat o(cs.dimf()); int k = 0;
  for (size_t c = 0; c < f.size(); ++c) if (cs.active[c]) { o.setSegment(k, f[c]); k += 3; }
  return o;
}
Mat SplitSolutionC::mu_stack(const ContactStatus& cs) const {
  Mat o(cs.dimf()); int k = 0;
  for (size_t c = 0; c < mu.size(); ++c) if (cs.active[c]) { o.setSegment(k, mu[c]); k += 3; }
  return o;
}
SplitDirectionC::SplitDirectionC(const Robot& r)
    : dlmd(r.dimv()), dgmm(r.dimv()), du(r.dimu()), dq(r.dimv()), dv(r.dimv()), daf(r.dimv()), dbetamu(r.dimv()), dnu_passive(6) {}
SplitKKTMatrixC::SplitKKTMatrixC(int nv_, int nu_)
    : nv(nv_), nu(nu_), Qxx(2 * nv_, 2 * nv_), Qxu_full(2 * nv_, nv_), Quu_full(nv_, nv_), Qaa_diag(nv_), Qff(0, 0),
      Fqq6(6, 6), Fqv6(6, 6), Fvq(nv_, nv_), Fvv(nv_, nv_), Fvu(nv_, nu_), Fqq_prev6(6, 6), Fqq_inv(6, 6), Fqq_prev_inv(6, 6) {}
SplitKKTResidualC::SplitKKTResidualC(int nv, int nu) : Fq(nv), Fv(nv), lq(nv), lv(nv), la(nv), lf(0), lu(nu), lu_passive(6), Fq_prev(6) {}

OCPSolver::OCPSolver(const idocp_model_t& model, const idocp_cost_t& cost_, const idocp_constraints_t& constraints, double T, int N)
    : robot(model), cost(cost_), cons(constraints), s(N + 1, SplitSolutionC(robot)), d(N + 1, SplitDirectionC(robot)),
      kkt_matrix(N + 1, SplitKKTMatrixC(model.nv, model.nu)), kkt_residual(N + 1, SplitKKTResidualC(model.nv, model.nu)),
      cd(N), ipm(N), riccati(N + 1, RiccatiC(model.nv)), K(N, Mat(model.nu, 2 * model.nv)), k(N, Mat(model.nu)),
      N_(N), nv_(model.nv), nu_(model.nu), nc_(model.ncontacts), T_(T), dt_(T / N) {
  if (T <= 0) throw std::out_of_range("invalid value: T must be positive!");
  if (N <= 0) throw std::out_of_range("invalid value: N must be positive!");
  if (!robot.hasFloatingBase()) throw std::logic_error("OCPSolver oracle: floating-base robots only");
  contact_status.active.assign(nc_, false);
  contact_status.points.assign(nc_, Mat(3));
}

void OCPSolver::setContactStatusUniformly(const std::vector<int>& active, const double* pts) {
  for (int c = 0; c < nc_; ++c) {
    contact_status.active[c] = active[c] != 0;
    for (int k2 = 0; k2 < 3; ++k2) contact_status.points[c][k2] = pts[3 * c + k2];
  }
}

void OCPSolver::setSolution(const std::string& name, const Mat& value) {
  for (auto& e : s) {
    if (name == "q") e.q = value;
    else if (name == "v") e.v = value;
    else if (name == "a") e.a = value;
    else if (name == "u") e.u = value;
    else if (name == "f") { for (auto& f : e.f) f = value; }
    else throw std::invalid_argument("invalid arugment: name must be q, v, a, f, or u!");
  }
}

// ------------------------------------------------------------ constraints ----
bool OCPSolver::componentEnabled(int c) const {
  if (c < 2) return cons.joint_position_limits != 0;
  if (c < 4) return cons.joint_velocity_limits != 0;
  if (c < 6) return cons.joint_torque_limits != 0;
  return cons.linearized_friction_cone != 0;
}
bool OCPSolver::componentValid(int c, int stage) const {     // constraints_data.hpp:18-42
  if (!componentEnabled(c)) return false;
  if (c < 2) return stage >= 2;
  if (c < 4) return stage >= 1;
  return true;
}
int OCPSolver::componentDim(int c) const { return c < 6 ? nu_ : 5 * nc_; }
int OCPSolver::dimc() const { int n = 0; for (int c = 0; c < 7; ++c) if (componentEnabled(c)) n += componentDim(c); return n; }

static double limitOf(const idocp_model_t& m, int c, int k2) {
  switch (c) {
    case 0: return m.q_min[k2];
    case 1: return m.q_max[k2];
    case 2: return -m.v_max[k2];
    case 3: return m.v_max[k2];
    case 4: return -m.u_max[k2];
    default: return m.u_max[k2];
  }
}
// value of the limited variable (joint part)
static double limitedVar(const SplitSolutionC& s, int c, int k2, int nv, int nu) {
  if (c < 2) return s.q[s.q.size() - nu + k2];
  if (c < 4) return s.v[nv - nu + k2];
  return s.u[k2];
}
// LinearizedFrictionCone::frictionConeResidual (linearized_friction_cone.hpp:72-84)
static void frictionConeResidual(double mu, const Mat& f, double* res) {
  const double m2 = mu / std::sqrt(2.0);
  res[0] = -f[2]; res[1] = f[0] - m2 * f[2]; res[2] = -f[0] - m2 * f[2]; res[3] = f[1] - m2 * f[2]; res[4] = -f[1] - m2 * f[2];
}
static void frictionJac(double mu, double J[5][3]) {        // linearized_friction_cone.cpp:25-29
  const double m2 = mu / std::sqrt(2.0);
  const double Jc[5][3] = {{0, 0, -1}, {1, 0, -m2}, {-1, 0, -m2}, {0, 1, -m2}, {0, -1, -m2}};
  for (int r = 0; r < 5; ++r) for (int c = 0; c < 3; ++c) J[r][c] = Jc[r][c];
}

// OCPLinearizer::initConstraints (ocp_linearizer.cpp:40-70) -> SplitOCP::initConstraints
void OCPSolver::initConstraints(double /*t*/) {
  for (int i = 0; i < N_; ++i) {
    ipm[i].clear();
    for (int c = 0; c < 7; ++c) {
      IpmData data(componentDim(c));
      if (componentValid(c, i)) {
        if (c < 6) {
          const double sgn = (c & 1) ? 1.0 : -1.0;
          for (int r = 0; r < nu_; ++r) data.slack[r] = -sgn * (limitedVar(s[i], c, r, nv_, nu_) - limitOf(robot.model(), c, r));
        } else {
          for (int cc = 0; cc < nc_; ++cc) {      // all contacts, active or not (linearized_friction_cone.cpp:96-104)
            double res[5]; frictionConeResidual(cons.mu, s[i].f[cc], res);
            for (int r = 0; r < 5; ++r) data.slack[5 * cc + r] = -res[r];
          }
        }
        for (int r = 0; r < data.slack.size(); ++r) {
          while (data.slack[r] < cons.barrier) data.slack[r] += cons.barrier;
          data.dual[r] = cons.barrier / data.slack[r];
        }
      }
      ipm[i].push_back(data);
    }
  }
}

// ------------------------------------------------------------------ cost ----
void OCPSolver::qRef(double t, Mat& q_ref) const {
  q_ref = Mat(robot.dimq());
  for (int i = 0; i < robot.dimq(); ++i) q_ref[i] = cost.q_ref[i];
  if (!cost.use_trotting_ref || !(t > cost.t_start)) return;
  const double tau = t - cost.t_start;
  const int steps = (int)std::floor(tau / cost.t_period);
  const double rate = (tau - steps * cost.t_period) / cost.t_period;
  const double sin2 = std::sin(M_PI_2 * rate);
  q_ref[0] += (steps + rate) * cost.step_length;
  if (steps % 2 == 0) {
    q_ref[9] -= sin2 * cost.front_swing_knee;  q_ref[12] -= sin2 * cost.hip_stance_knee;
    q_ref[15] += sin2 * cost.front_stance_knee; q_ref[18] += sin2 * cost.hip_swing_knee;
  } else {
    q_ref[9] += sin2 * cost.front_stance_knee; q_ref[12] += sin2 * cost.hip_swing_knee;
    q_ref[15] -= sin2 * cost.front_swing_knee; q_ref[18] -= sin2 * cost.hip_stance_knee;
  }
}

// ---------------------------------------------------------------- stages ----
// SplitOCP::linearizeOCP (split_ocp.hxx:58-91) / computeKKTResidual (:189-214)
void OCPSolver::linearizeStage(int i, double t, const Mat& q_prev, bool residual_only) {
  const SplitSolutionC& si = s[i];
  const SplitSolutionC& sn = s[i + 1];
  SplitKKTMatrixC& M = kkt_matrix[i];
  SplitKKTResidualC& R = kkt_residual[i];
  ContactDynamicsDataC& D = cd[i];
  const ContactStatus& cs = contact_status;
  const int nv = nv_, nu = nu_, dimf = cs.dimf();
  const double dt = dt_;
  robot.updateKinematics(si.q, si.v, si.a);
  if (!residual_only) {
    M.Qxx.setZero(); M.Qxu_full.setZero(); M.Quu_full.setZero(); M.Qaa_diag.setZero(); M.Qff = Mat(dimf, dimf);
    M.Fvq.setZero(); M.Fvv.setZero(); M.Fvu.setZero();
  }
  R.Fq.setZero(); R.Fv.setZero(); R.lq.setZero(); R.lv.setZero(); R.la.setZero(); R.lf = Mat(dimf); R.lu.setZero(); R.lu_passive.setZero();
  // ---- cost: (Trotting)ConfigurationSpaceCost + ContactForceCost
  // (configuration_space_cost.cpp:292-310, trotting_configuration_space_cost.cpp:269-286, contact_force_cost.cpp:153-165)
  Mat q_ref, qdiff, Jq;
  qRef(t, q_ref);
  robot.subtractConfiguration(si.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(si.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = cost.q_weight[r] * qdiff[r];
  R.lq += dt * (Jq.t() * Wq);
  const double v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  for (int r = 0; r < nv; ++r) {
    R.lv[r] += dt * cost.v_weight[r] * (si.v[r] - (r == 0 ? v_ref0 : cost.v_ref[r]));
    R.la[r] += dt * cost.a_weight[r] * si.a[r];
  }
  for (int r = 0; r < nu; ++r) R.lu[r] += dt * cost.u_weight[r] * (si.u[r] - cost.u_ref[r]);
  {
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
      for (int r = 0; r < 3; ++r) R.lf[st + r] += dt * cost.f_weight[c][r] * (si.f[c][r] - cost.f_ref[c][r]);
      st += 3;
    }
  }
  // ---- constraints: [computePrimalAndDualResidual] + augmentDualResidual
  double Jc[5][3]; frictionJac(cons.mu, Jc);
  for (int c = 0; c < 7; ++c) {
    if (!componentValid(c, i)) continue;
    IpmData& data = ipm[i][c];
    if (c < 6) {
      const double sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : R.lu);
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        if (residual_only) {
          data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), c, r)) + data.slack[r];
          data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        }
        l[off + r] += sgn * dt * data.dual[r];
      }
    } else {
      if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        if (residual_only) {
          double res[5]; frictionConeResidual(cons.mu, si.f[cc], res);
          for (int r = 0; r < 5; ++r) {
            data.residual[5 * cc + r] = res[r] + data.slack[5 * cc + r];
            data.duality[5 * cc + r] = data.slack[5 * cc + r] * data.dual[5 * cc + r] - cons.barrier;
          }
        }
        for (int x = 0; x < 3; ++x) for (int r = 0; r < 5; ++r) R.lf[st + x] += dt * Jc[r][x] * data.dual[5 * cc + r];
        st += 3;
      }
    }
  }
  // ---- state equation: linearizeForwardEuler (state_equation.hxx:12-37, 210-221)
  Mat diff; robot.subtractConfiguration(si.q, sn.q, diff);
  for (int r = 0; r < nv; ++r) { R.Fq[r] = diff[r] + dt * si.v[r]; R.Fv[r] = si.v[r] + dt * si.a[r] - sn.v[r]; }
  Mat Fqq, Fqq_prev;
  robot.dSubtractdConfigurationPlus(si.q, sn.q, Fqq);
  robot.dSubtractdConfigurationMinus(q_prev, si.q, Fqq_prev);
  M.Fqq6 = Fqq.block(0, 0, 6, 6);
  M.Fqq_prev6 = Fqq_prev.block(0, 0, 6, 6);
  {
    Mat t1 = M.Fqq6.t() * sn.lmd.segment(0, 6) + M.Fqq_prev6.t() * si.lmd.segment(0, 6);
    for (int r = 0; r < 6; ++r) R.lq[r] += t1[r];
    for (int r = 6; r < nv; ++r) R.lq[r] += sn.lmd[r] - si.lmd[r];
    for (int r = 0; r < nv; ++r) { R.lv[r] += dt * sn.lmd[r] + sn.gmm[r] - si.gmm[r]; R.la[r] += dt * sn.gmm[r]; }
  }
  if (!residual_only) {
    // condenseForwardEuler (state_equation.hxx:40-63)
    Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_prev_inv);
    Mat Fm; robot.dSubtractdConfigurationMinus(si.q, sn.q, Fm);
    M.Fqq_prev6 = Fm.block(0, 0, 6, 6);
    Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_inv);
    M.Fqq_prev6 = M.Fqq6;
    R.Fq_prev = R.Fq.segment(0, 6);
    M.Fqq6 = -1.0 * (M.Fqq_inv * M.Fqq_prev6);
    M.Fqv6 = (-dt) * M.Fqq_inv;
    R.Fq.setSegment(0, -1.0 * (M.Fqq_inv * R.Fq_prev));
  }
  // ---- ContactDynamics::linearizeContactDynamics (contact_dynamics.hxx:48-102)
  robot.setContactForces(cs.active, si.f);
  Mat ID_full;
  robot.RNEA(si.q, si.v, si.a, ID_full);
  for (int r = 0; r < nu; ++r) ID_full[kP + r] -= si.u[r];
  Mat dIDdq, dIDdv;
  robot.RNEADerivatives(si.q, si.v, si.a, dIDdq, dIDdv, D.dIDda);
  Mat C, dCdq, dCdv;
  robot.computeBaumgarteResidual(cs.active, dt_, cs.points, C);      // baumgarte_time_step = T/N (hybrid_container.hpp:186-188)
  robot.computeBaumgarteDerivatives(cs.active, dt_, dCdq, dCdv, D.dCda);
  D.IDC = Mat(nv + dimf); D.IDC.setSegment(0, ID_full); D.IDC.setSegment(nv, C);
  D.dIDCdqv = Mat(nv + dimf, 2 * nv);
  D.dIDCdqv.setBlock(0, 0, dIDdq); D.dIDCdqv.setBlock(0, nv, dIDdv);
  D.dIDCdqv.setBlock(nv, 0, dCdq); D.dIDCdqv.setBlock(nv, nv, dCdv);
  R.lq += dt * (dIDdq.t() * si.beta);
  R.lv += dt * (dIDdv.t() * si.beta);
  R.la += dt * (D.dIDda.t() * si.beta);
  const Mat mu_stack = si.mu_stack(cs);
  if (dimf > 0) R.lf -= dt * (D.dCda * si.beta);
  for (int r = 0; r < 6; ++r) R.lu_passive[r] = dt * si.nu_passive[r] - dt * si.beta[r];
  for (int r = 0; r < nu; ++r) R.lu[r] -= dt * si.beta[kP + r];
  if (dimf > 0) {
    R.lq += dt * (dCdq.t() * mu_stack);
    R.lv += dt * (dCdv.t() * mu_stack);
    R.la += dt * (D.dCda.t() * mu_stack);
  }
  if (residual_only) return;
  // ---- cost Hessian (configuration_space_cost.cpp:351-365; contact_force_cost.cpp:182-194)
  {
    Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.q_weight[r];
    M.Qxx.addBlock(0, 0, Jq.t() * WJ, dt);
    for (int r = 0; r < nv; ++r) { M.Qxx(nv + r, nv + r) += dt * cost.v_weight[r]; M.Qaa_diag[r] += dt * cost.a_weight[r]; }
    for (int r = 0; r < nu; ++r) M.Quu_full(kP + r, kP + r) += dt * cost.u_weight[r];
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) { for (int r = 0; r < 3; ++r) M.Qff(st + r, st + r) += dt * cost.f_weight[c][r]; st += 3; }
  }
  // ---- Constraints::condenseSlackAndDual
  for (int c = 0; c < 7; ++c) {
    if (!componentValid(c, i)) continue;
    IpmData& data = ipm[i][c];
    if (c < 6) {
      const double sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : R.lu);
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        const double h = dt * data.dual[r] / data.slack[r];
        if (c < 2) M.Qxx(kP + r, kP + r) += h;
        else if (c < 4) M.Qxx(nv + kP + r, nv + kP + r) += h;
        else M.Quu_full(kP + r, kP + r) += h;
        data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), c, r)) + data.slack[r];
        data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        l[off + r] += sgn * dt * (data.dual[r] * data.residual[r] - data.duality[r]) / data.slack[r];
      }
    } else {
      // linearized_friction_cone.cpp:125-152, 184-202
      data.residual.setZero(); data.duality.setZero();
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        double res[5]; frictionConeResidual(cons.mu, si.f[cc], res);
        double rr[5], dd[5];
        for (int r = 0; r < 5; ++r) {
          const int idx = 5 * cc + r;
          data.residual[idx] = res[r] + data.slack[idx];
          data.duality[idx] = data.slack[idx] * data.dual[idx] - cons.barrier;
          rr[r] = (data.dual[idx] * data.residual[idx] - data.duality[idx]) / data.slack[idx];
          dd[r] = data.dual[idx] / data.slack[idx];
        }
        for (int x = 0; x < 3; ++x) {
          for (int r = 0; r < 5; ++r) R.lf[st + x] += dt * Jc[r][x] * rr[r];
          for (int y = 0; y < 3; ++y) { double acc = 0; for (int r = 0; r < 5; ++r) acc += Jc[r][x] * dd[r] * Jc[r][y]; M.Qff(st + x, st + y) += dt * acc; }
        }
        st += 3;
      }
    }
  }
  // ---- ContactDynamics::condenseContactDynamics (contact_dynamics.hxx:105-158)
  Robot::computeMJtJinv(D.dIDda, D.dCda, D.MJtJinv);
  D.MJtJinv_dIDCdqv = D.MJtJinv * D.dIDCdqv;
  D.MJtJinv_IDC = D.MJtJinv * D.IDC;
  D.Qafqv = Mat(nv + dimf, 2 * nv);
  D.Qafu_full = Mat(nv + dimf, nv);
  for (int c = 0; c < 2 * nv; ++c) for (int r = 0; r < nv; ++r) D.Qafqv(r, c) = -M.Qaa_diag[r] * D.MJtJinv_dIDCdqv(r, c);
  for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) D.Qafu_full(r, c) = M.Qaa_diag[r] * D.MJtJinv(r, c);
  if (dimf > 0) {
    D.Qafqv.setBlock(nv, 0, -1.0 * (M.Qff * D.MJtJinv_dIDCdqv.block(nv, 0, dimf, 2 * nv)));
    D.Qafu_full.setBlock(nv, 0, M.Qff * D.MJtJinv.block(nv, 0, dimf, nv));
  }
  D.laf = Mat(nv + dimf);
  for (int r = 0; r < nv; ++r) D.laf[r] = R.la[r] - M.Qaa_diag[r] * D.MJtJinv_IDC[r];
  if (dimf > 0) D.laf.setSegment(nv, -1.0 * R.lf - M.Qff * D.MJtJinv_IDC.segment(nv, dimf));
  M.Qxx -= D.MJtJinv_dIDCdqv.t() * D.Qafqv;
  M.Qxu_full -= D.MJtJinv_dIDCdqv.t() * D.Qafu_full;
  {
    Mat lx = D.MJtJinv_dIDCdqv.t() * D.laf;
    for (int r = 0; r < nv; ++r) { R.lq[r] -= lx[r]; R.lv[r] -= lx[nv + r]; }
  }
  M.Quu_full += D.MJtJinv.block(0, 0, nv, nv + dimf) * D.Qafu_full;
  {
    Mat t1 = D.MJtJinv.block(0, 0, nv, nv + dimf) * D.laf;
    for (int r = 0; r < 6; ++r) R.lu_passive[r] += t1[r];
    for (int r = 0; r < nu; ++r) R.lu[r] += t1[kP + r];
  }
  M.Fvq = (-dt) * D.MJtJinv_dIDCdqv.block(0, 0, nv, nv);
  M.Fvv = (-dt) * D.MJtJinv_dIDCdqv.block(0, nv, nv, nv) + Mat::Identity(nv);
  M.Fvu = dt * D.MJtJinv.block(0, kP, nv, nu);
  for (int r = 0; r < nv; ++r) R.Fv[r] -= dt * D.MJtJinv_IDC[r];
}

// TerminalOCP::linearizeOCP / computeKKTResidual (terminal_ocp.hxx:50-66, 118-131)
void OCPSolver::linearizeTerminal(double t, const Mat& q_prev, bool residual_only) {
  const SplitSolutionC& sN = s[N_];
  SplitKKTMatrixC& M = kkt_matrix[N_];
  SplitKKTResidualC& R = kkt_residual[N_];
  const int nv = nv_;
  R.lq.setZero(); R.lv.setZero();
  Mat q_ref, qdiff, Jq;
  qRef(t, q_ref);
  robot.subtractConfiguration(sN.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(sN.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = cost.qf_weight[r] * qdiff[r];
  R.lq += Jq.t() * Wq;
  const double v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  for (int r = 0; r < nv; ++r) R.lv[r] += cost.vf_weight[r] * (sN.v[r] - (r == 0 ? v_ref0 : cost.v_ref[r]));
  // linearizeForwardEulerTerminal (state_equation.hxx:66-83)
  Mat Fqq_prev; robot.dSubtractdConfigurationMinus(q_prev, sN.q, Fqq_prev);
  M.Fqq_prev6 = Fqq_prev.block(0, 0, 6, 6);
  Mat t1 = M.Fqq_prev6.t() * sN.lmd.segment(0, 6);
  for (int r = 0; r < 6; ++r) R.lq[r] += t1[r];
  for (int r = 6; r < nv; ++r) R.lq[r] -= sN.lmd[r];
  R.lv -= sN.gmm;
  if (residual_only) return;
  Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_prev_inv);     // condenseForwardEulerTerminal
  M.Qxx.setZero();
  Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.qf_weight[r];
  M.Qxx.addBlock(0, 0, Jq.t() * WJ);
  for (int r = 0; r < nv; ++r) M.Qxx(nv + r, nv + r) += cost.vf_weight[r];
}

void OCPSolver::linearizeOCP(double t, const Mat& q) {
  for (int i = 0; i <= N_; ++i) {
    const Mat& q_prev = (i == 0) ? q : s[i - 1].q;                      // ocp_linearizer.hxx:231-248
    if (i < N_) linearizeStage(i, t + i * dt_, q_prev, false);
    else linearizeTerminal(t + T_, q_prev, false);
  }
}

void OCPSolver::computeKKTResidual(double t, const Mat& q, const Mat& /*v*/) {
  for (int i = 0; i <= N_; ++i) {
    const Mat& q_prev = (i == 0) ? q : s[i - 1].q;
    if (i < N_) linearizeStage(i, t + i * dt_, q_prev, true);
    else linearizeTerminal(t + T_, q_prev, true);
  }
}

// OCPLinearizer::KKTError (ocp_linearizer.cpp:98-137); SplitOCP::squaredNormKKTResidual (split_ocp.hxx:251-267)
double OCPSolver::KKTError() {
  double sum = 0;
  for (int i = 0; i < N_; ++i) {
    const SplitKKTResidualC& R = kkt_residual[i];
    double e = R.lq.squaredNorm() + R.lv.squaredNorm() + R.la.squaredNorm() + R.lf.squaredNorm() + R.lu_passive.squaredNorm() +
               R.lu.squaredNorm() + R.Fq.squaredNorm() + R.Fv.squaredNorm() + dt_ * dt_ * cd[i].IDC.squaredNorm();
    double c2 = 0;
    for (int c = 0; c < 7; ++c) if (componentValid(c, i)) c2 += ipm[i][c].residual.squaredNorm() + ipm[i][c].duality.squaredNorm();
    sum += e + dt_ * dt_ * c2;
  }
  sum += kkt_residual[N_].lq.squaredNorm() + kkt_residual[N_].lv.squaredNorm();
  return std::sqrt(sum);
}

// ---------------------------------------------------------------- Riccati ----
// RiccatiRecursionSolver::backwardRiccatiRecursion without events (riccati_recursion_solver.cpp:48-107)
void OCPSolver::backwardRiccatiRecursion() {
  const int nv = nv_, nu = nu_, nj = nv - 6;
  const double dt = dt_;
  riccati[N_].Pqq = kkt_matrix[N_].Qxx.block(0, 0, nv, nv);
  riccati[N_].Pvv = kkt_matrix[N_].Qxx.block(nv, nv, nv, nv);
  riccati[N_].Pqv = Mat(nv, nv);
  riccati[N_].sq = -kkt_residual[N_].lq;
  riccati[N_].sv = -kkt_residual[N_].lv;
  for (int i = N_ - 1; i >= 0; --i) {
    const RiccatiC& rn = riccati[i + 1];
    SplitKKTMatrixC& M = kkt_matrix[i];
    SplitKKTResidualC& R = kkt_residual[i];
    // BackwardRiccatiRecursionFactorizer::factorizeKKTMatrix (backward_riccati_recursion_factorizer.hxx:44-114)
    Mat AtPqq(nv, nv), AtPqv(nv, nv), AtPvq(nv, nv), AtPvv(nv, nv);
    AtPqq.setBlock(0, 0, M.Fqq6.t() * rn.Pqq.block(0, 0, 6, nv)); AtPqq.setBlock(6, 0, rn.Pqq.block(6, 0, nj, nv));
    AtPqv.setBlock(0, 0, M.Fqq6.t() * rn.Pqv.block(0, 0, 6, nv)); AtPqv.setBlock(6, 0, rn.Pqv.block(6, 0, nj, nv));
    AtPvq.setBlock(0, 0, M.Fqv6.t() * rn.Pqq.block(0, 0, 6, nv)); AtPvq.setBlock(6, 0, dt * rn.Pqq.block(6, 0, nj, nv));
    AtPvv.setBlock(0, 0, M.Fqv6.t() * rn.Pqv.block(0, 0, 6, nv)); AtPvv.setBlock(6, 0, dt * rn.Pqv.block(6, 0, nj, nv));
    AtPqq += M.Fvq.t() * rn.Pqv.t();
    AtPqv += M.Fvq.t() * rn.Pvv;
    AtPvq += M.Fvv.t() * rn.Pqv.t();
    AtPvv += M.Fvv.t() * rn.Pvv;
    Mat BtPq = M.Fvu.t() * rn.Pqv.t();
    Mat BtPv = M.Fvu.t() * rn.Pvv;
    Mat Qqq = M.Qxx.block(0, 0, nv, nv), Qqv = M.Qxx.block(0, nv, nv, nv), Qvv = M.Qxx.block(nv, nv, nv, nv);
    Qqq.addBlock(0, 0, AtPqq.block(0, 0, nv, 6) * M.Fqq6); Qqq.addBlock(0, 6, AtPqq.block(0, 6, nv, nj));
    Qqv.addBlock(0, 0, AtPqq.block(0, 0, nv, 6) * M.Fqv6); Qqv.addBlock(0, 6, AtPqq.block(0, 6, nv, nj), dt);
    Qvv.addBlock(0, 0, AtPvq.block(0, 0, nv, 6) * M.Fqv6); Qvv.addBlock(0, 6, AtPvq.block(0, 6, nv, nj), dt);
    Qqq += AtPqv * M.Fvq;
    Qqv += AtPqv * M.Fvv;
    Qvv += AtPvv * M.Fvv;
    M.Qxx.setBlock(0, 0, Qqq); M.Qxx.setBlock(0, nv, Qqv); M.Qxx.setBlock(nv, nv, Qvv); M.Qxx.setBlock(nv, 0, Qqv.t());
    Mat Qqu = M.Qxu_full.block(0, kP, nv, nu), Qvu = M.Qxu_full.block(nv, kP, nv, nu);
    Qqu += AtPqv * M.Fvu;
    Qvu += AtPvv * M.Fvu;
    M.Qxu_full.setBlock(0, kP, Qqu); M.Qxu_full.setBlock(nv, kP, Qvu);
    Mat Quu = M.Quu_full.block(kP, kP, nu, nu);
    Quu += BtPv * M.Fvu;
    M.Quu_full.setBlock(kP, kP, Quu);
    R.lu += BtPq * R.Fq;
    R.lu += BtPv * R.Fv;
    R.lu -= M.Fvu.t() * rn.sv;
    // SplitRiccatiFactorizer::backwardRiccatiRecursion (split_riccati_factorizer.hxx:36-52)
    LLT llt;
    if (!llt.compute(Quu)) throw std::runtime_error("Riccati: Quu not positive definite at stage " + std::to_string(i));
    Mat Qxu(2 * nv, nu); Qxu.setBlock(0, 0, Qqu); Qxu.setBlock(nv, 0, Qvu);
    K[i] = -llt.solve(Qxu.t());
    k[i] = -llt.solve(R.lu);
    // factorizeRiccatiFactorization (backward_riccati_recursion_factorizer.hxx:117-161)
    RiccatiC& r = riccati[i];
    r.Pqq = Qqq; r.Pqv = Qqv; r.Pvv = Qvv;
    Mat GK = Quu * K[i];
    Mat Kq = K[i].block(0, 0, nu, nv), Kv = K[i].block(0, nv, nu, nv);
    r.Pqq -= Kq.t() * GK.block(0, 0, nu, nv);
    r.Pqv -= Kq.t() * GK.block(0, nv, nu, nv);
    r.Pvv -= Kv.t() * GK.block(0, nv, nu, nv);
    r.Pqq = 0.5 * (r.Pqq + r.Pqq.t());
    r.Pvv = 0.5 * (r.Pvv + r.Pvv.t());
    r.sq = Mat(nv); r.sv = Mat(nv);
    r.sq.setSegment(0, M.Fqq6.t() * rn.sq.segment(0, 6)); r.sq.setSegment(6, rn.sq.segment(6, nj));
    r.sv.setSegment(0, M.Fqv6.t() * rn.sq.segment(0, 6)); r.sv.setSegment(6, dt * rn.sq.segment(6, nj));
    r.sq += M.Fvq.t() * rn.sv;
    r.sv += M.Fvv.t() * rn.sv;
    r.sq -= AtPqq * R.Fq;
    r.sq -= AtPqv * R.Fv;
    r.sv -= AtPvq * R.Fq;
    r.sv -= AtPvv * R.Fv;
    r.sq -= R.lq;
    r.sv -= R.lv;
    r.sq -= Qqu * k[i];
    r.sv -= Qvu * k[i];
  }
}

// computeInitialStateDirection + forwardRiccatiRecursion (riccati_recursion_solver.cpp:110-162;
// split_riccati_factorizer.hxx:103-128)
void OCPSolver::forwardRiccatiRecursion(const Mat& q, const Mat& v) {
  const int nv = nv_, nj = nv - 6;
  robot.subtractConfiguration(q, s[0].q, d[0].dq);
  d[0].dq.setSegment(0, -1.0 * (kkt_matrix[0].Fqq_prev_inv * d[0].dq.segment(0, 6)));
  d[0].dv = v - s[0].v;
  for (int i = 0; i < N_; ++i) {
    const SplitKKTMatrixC& M = kkt_matrix[i];
    const SplitKKTResidualC& R = kkt_residual[i];
    Mat dx(2 * nv); dx.setSegment(0, d[i].dq); dx.setSegment(nv, d[i].dv);
    d[i].du = K[i] * dx + k[i];
    Mat dqn = R.Fq, dvn = R.Fv;
    Mat h = M.Fqq6 * d[i].dq.segment(0, 6) + M.Fqv6 * d[i].dv.segment(0, 6);
    for (int r = 0; r < 6; ++r) dqn[r] += h[r];
    for (int r = 0; r < nj; ++r) dqn[6 + r] += d[i].dq[6 + r] + dt_ * d[i].dv[6 + r];
    dvn += M.Fvq * d[i].dq;
    dvn += M.Fvv * d[i].dv;
    dvn += M.Fvu * d[i].du;
    d[i + 1].dq = dqn; d[i + 1].dv = dvn;
  }
}

static double fractionToBoundary(double rate, const Mat& vec, const Mat& dvec) {      // pdipm.hxx:52-73
  double m = 1;
  for (int i = 0; i < vec.size(); ++i) {
    const double f = -rate * (vec[i] / dvec[i]);
    if (f > 0 && f < 1 && f < m) m = f;
  }
  return m;
}

// RiccatiRecursionSolver::computeDirection (riccati_recursion_solver.cpp:165-251)
void OCPSolver::computeDirection() {
  const int nv = nv_, nu = nu_;
  double pmin = 1, dmin = 1;
  double Jc[5][3]; frictionJac(cons.mu, Jc);
  for (int i = 0; i <= N_; ++i) {
    const RiccatiC& r = riccati[i];
    d[i].dlmd = r.Pqq * d[i].dq + r.Pqv * d[i].dv - r.sq;
    d[i].dgmm = r.Pqv.t() * d[i].dq + r.Pvv * d[i].dv - r.sv;
    if (i == N_) continue;
    const ContactDynamicsDataC& D = cd[i];
    const int dimf = contact_status.dimf();
    // ContactDynamics::computeCondensedPrimalDirection (contact_dynamics.hxx:161-168)
    Mat dx(2 * nv); dx.setSegment(0, d[i].dq); dx.setSegment(nv, d[i].dv);
    d[i].daf = -1.0 * (D.MJtJinv_dIDCdqv * dx);
    d[i].daf += D.MJtJinv.block(0, kP, nv + dimf, nu) * d[i].du;
    d[i].daf -= D.MJtJinv_IDC;
    for (int r2 = 0; r2 < dimf; ++r2) d[i].daf[nv + r2] *= -1;
    // Constraints::computeSlackAndDualDirection + step sizes
    for (int c = 0; c < 7; ++c) {
      if (!componentValid(c, i)) continue;
      IpmData& data = ipm[i][c];
      if (c < 6) {
        const double sgn = (c & 1) ? 1.0 : -1.0;
        for (int r2 = 0; r2 < nu; ++r2) {
          const double dxr = c < 2 ? d[i].dq[kP + r2] : (c < 4 ? d[i].dv[kP + r2] : d[i].du[r2]);
          data.dslack[r2] = -sgn * dxr - data.residual[r2];
          data.ddual[r2] = -(data.dual[r2] * data.dslack[r2] + data.duality[r2]) / data.slack[r2];
        }
      } else {
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }   // linearized_friction_cone.cpp:162-163
        int st = 0;
        for (int cc = 0; cc < nc_; ++cc) if (contact_status.active[cc]) {
          for (int r2 = 0; r2 < 5; ++r2) {
            const int idx = 5 * cc + r2;
            double Jdf = 0; for (int x = 0; x < 3; ++x) Jdf += Jc[r2][x] * d[i].daf[nv + st + x];
            data.dslack[idx] = -Jdf - data.residual[idx];
            data.ddual[idx] = -(data.dual[idx] * data.dslack[idx] + data.duality[idx]) / data.slack[idx];
          }
          st += 3;
        }
      }
      pmin = std::min(pmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.slack, data.dslack));
      dmin = std::min(dmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.dual, data.ddual));
    }
  }
  primal_step_size = pmin; dual_step_size = dmin;
}

// OCPLinearizer::integrateSolution (ocp_linearizer.cpp:140-221)
void OCPSolver::integrateSolution() {
  const int nv = nv_, nu = nu_;
  const double ap = primal_step_size, ad = dual_step_size, dt = dt_;
  for (int i = 0; i <= N_; ++i) {
    SplitKKTMatrixC& M = kkt_matrix[i];
    if (i < N_) {
      ContactDynamicsDataC& D = cd[i];
      SplitKKTResidualC& R = kkt_residual[i];
      const int dimf = contact_status.dimf();
      Mat dx(2 * nv); dx.setSegment(0, d[i].dq); dx.setSegment(nv, d[i].dv);
      const Mat& dgmm = d[i + 1].dgmm;
      // ContactDynamics::computeCondensedDualDirection (contact_dynamics.hxx:171-190)
      d[i].dnu_passive = R.lu_passive;
      d[i].dnu_passive += M.Quu_full.block(0, kP, 6, nu) * d[i].du;
      d[i].dnu_passive += M.Qxu_full.block(0, 0, 2 * nv, 6).t() * dx;
      d[i].dnu_passive += dt * (D.MJtJinv.block(0, 0, 6, nv) * dgmm);
      d[i].dnu_passive = (-1.0 / dt) * d[i].dnu_passive;
      D.laf += D.Qafqv * dx;
      D.laf += D.Qafu_full.block(0, kP, nv + dimf, nu) * d[i].du;
      for (int r = 0; r < nv; ++r) D.laf[r] += dt * dgmm[r];
      d[i].dbetamu = (-1.0 / dt) * (D.MJtJinv * D.laf);
    }
    // stateequation::correctCostateDirectionForwardEuler (state_equation.hxx:96-108)
    d[i].dlmd.setSegment(0, -1.0 * (M.Fqq_prev_inv.t() * d[i].dlmd.segment(0, 6)));
    // SplitOCP::updatePrimal / TerminalOCP::updatePrimal -> SplitSolution::integrate (split_solution.hxx:215-240)
    SplitSolutionC& si = s[i];
    si.lmd += ap * d[i].dlmd;
    si.gmm += ap * d[i].dgmm;
    Mat qn; robot.integrateConfiguration(si.q, d[i].dq, ap, qn); si.q = qn;
    si.v += ap * d[i].dv;
    if (i == N_) continue;
    si.a += ap * d[i].daf.segment(0, nv);
    si.u += ap * d[i].du;
    si.beta += ap * d[i].dbetamu.segment(0, nv);
    si.nu_passive += ap * d[i].dnu_passive;
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (contact_status.active[c]) {
      for (int r = 0; r < 3; ++r) { si.f[c][r] += ap * d[i].daf[nv + st + r]; si.mu[c][r] += ap * d[i].dbetamu[nv + st + r]; }
      st += 3;
    }
    for (int c = 0; c < 7; ++c) {
      if (!componentValid(c, i)) continue;
      ipm[i][c].slack += ap * ipm[i][c].dslack;
      ipm[i][c].dual += ad * ipm[i][c].ddual;
    }
  }
}

void OCPSolver::updateSolution(double t, const Mat& q, const Mat& v) {
  linearizeOCP(t, q);
  auto t0 = std::chrono::steady_clock::now();
  backwardRiccatiRecursion();
  forwardRiccatiRecursion(q, v);
  riccati_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  computeDirection();
  integrateSolution();
}

}  // namespace oracle
