// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md and unocp.hpp).
#include "unocp.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <stdexcept>
#include <string>

#ifdef _OPENMP
#include <omp.h>
#define ORACLE_THREAD_NUM omp_get_thread_num()
#else
#define ORACLE_THREAD_NUM 0
#endif

namespace oracle {

// ------------------------------------------------------------ constraints ----
Constraints::Constraints(const Robot& robot, const idocp_constraints_t& c)
    : barrier(c.barrier), fraction_to_boundary_rate(c.fraction_to_boundary_rate) {
  const RModel& m = robot.model();
  const int nu = m.nu;
  auto vec = [&](const real* p, real sgn) { Mat v(nu); for (int i = 0; i < nu; ++i) v[i] = sgn * p[i]; return v; };
  // Constraints::push_back sorts components into position / velocity /
  // acceleration level lists (constraints.hxx:24-36); JointConstraintsFactory
  // pushes lower before upper (joint_constraints_factory.cpp:21-38).
  if (c.joint_position_limits) {
    components.push_back({JointLimit::Q, -1, vec(m.q_min, 1.0)});
    components.push_back({JointLimit::Q, +1, vec(m.q_max, 1.0)});
  }
  if (c.joint_velocity_limits) {
    components.push_back({JointLimit::V, -1, vec(m.v_max, -1.0)});
    components.push_back({JointLimit::V, +1, vec(m.v_max, 1.0)});
  }
  if (c.joint_torque_limits) {
    components.push_back({JointLimit::U, -1, vec(m.u_max, -1.0)});
    components.push_back({JointLimit::U, +1, vec(m.u_max, 1.0)});
  }
  // JointAccelerationLowerLimit / UpperLimit (joint_acceleration_{lower,upper}_limit.cpp): bounds of the components' own, acceleration level
  auto vecd = [&](const double* p) { Mat v(nu); for (int i = 0; i < nu; ++i) v[i] = p[i]; return v; };
  if (c.joint_acceleration_lower_limit) components.push_back({JointLimit::A, -1, vecd(c.a_min)});
  if (c.joint_acceleration_upper_limit) components.push_back({JointLimit::A, +1, vecd(c.a_max)});
}

int Constraints::dimc_total() const {
  int n = 0; for (auto& c : components) n += c.lim.size(); return n;
}

static const Mat& varOf(const JointLimit& jl, const SplitSolution& s) {
  return jl.var == JointLimit::Q ? s.q : (jl.var == JointLimit::V ? s.v : (jl.var == JointLimit::A ? s.a : s.u));
}
static const Mat& dvarOf(const JointLimit& jl, const SplitDirection& d) {
  return jl.var == JointLimit::Q ? d.dq : (jl.var == JointLimit::V ? d.dv : (jl.var == JointLimit::A ? d.da : d.du));
}

// pdipm::ComputeDuality + the component's computePrimalAndDualResidual
// (pdipm.hxx:26-31; e.g. joint_torques_upper_limit.cpp:82-86)
static void computePrimalAndDualResidual(const Constraints& cs, const JointLimit& jl,
                                         ConstraintComponentData& data, const SplitSolution& s) {
  const Mat& x = varOf(jl, s);
  const int n = jl.lim.size(), off = x.size() - n;      // .tail(dimc_)
  for (int i = 0; i < n; ++i) {
    data.residual[i] = jl.sign * (x[off + i] - jl.lim[i]) + data.slack[i];
    data.duality[i] = data.slack[i] * data.dual[i] - cs.barrier;
  }
}

// ------------------------------------------------------------- SplitUnOCP ----
SplitUnOCP::SplitUnOCP(int nv_)
    : nv(nv_), ID(nv_), dID_dq(nv_, nv_), dID_dv(nv_, nv_), dID_da(nv_, nv_), lu_condensed(nv_),
      Qqq(nv_, nv_), Qvv_diag(nv_), Qaa_diag(nv_), Quu_diag(nv_),
      Fq(nv_), Fv(nv_), lq(nv_), lv(nv_), la(nv_), lu(nv_) {}

UnOCPSolver::UnOCPSolver(const RModel& model, const RCost& cost_, const idocp_constraints_t& cons,
                         real T, int N)
    : robot(model), cost(cost_), constraints(robot, cons),
      s(N + 1, SplitSolution(robot)), d(N + 1, SplitDirection(robot)),
      ocp(N, SplitUnOCP(model.nv)),
      unkkt_matrix(N, SplitUnKKTMatrix(model.nv)), unkkt_residual(N, SplitUnKKTResidual(model.nv)),
      terminal_Qqq(model.nv, model.nv), terminal_Qvv(model.nv, model.nv), terminal_lq(model.nv), terminal_lv(model.nv),
      riccati(N + 1, SplitRiccatiFactorization(model.nv)),
      K(N, Mat(model.nv, 2 * model.nv)), k(N, Mat(model.nv)),
      task_robot_(model), N_(N), T_(T), dt_(T / N) {
  if (cost.task_dim != 0) {
    if (cost.task_dim != 3 && cost.task_dim != 6) throw std::invalid_argument("task_dim must be 0, 3 or 6");
    RModel mt = model;                      // the task frame as contact 0: framePlacement / getFrameJacobian come from the contact kinematics
    mt.ncontacts = 1; mt.contact_frame_id[0] = -1; mt.contact_joint[0] = cost.task_joint;
    for (int k2 = 0; k2 < 9; ++k2) mt.contact_R[0][k2] = cost.task_frame_R[k2];
    for (int k2 = 0; k2 < 3; ++k2) mt.contact_p[0][k2] = cost.task_frame_p[k2];
    task_robot_ = Robot(mt);
    std::array<real, 12> r0; for (int k2 = 0; k2 < 12; ++k2) r0[k2] = cost.task_ref[k2];
    task_refs.assign(N + 1, r0);
  }
  if (T <= 0) throw std::out_of_range("invalid value: T must be positive!");
  if (N <= 0) throw std::out_of_range("invalid value: N must be positive!");
  if (robot.hasFloatingBase() || robot.maxPointContacts() > 0)
    throw std::logic_error("robot has floating base or contacts: use OCPSolver");   // split_unocp.hxx:27-33
  initConstraints();
}

void UnOCPSolver::setTaskRefs(const double* refs) {
  for (int i = 0; i <= N_ && i < (int)task_refs.size(); ++i) for (int k2 = 0; k2 < 12; ++k2) task_refs[i][k2] = refs[12 * i + k2];
}
void UnOCPSolver::taskTerms(int i, const Mat& q, real& c, Mat& g, Mat& H) const {
  Robot rb = task_robot_;
  rb.taskSpaceTerms(cost.task_dim, task_refs[i].data(), i == N_ ? cost.task_weightf : cost.task_weight, q, c, g, H);
  for (int e = 1; e < cost.taskCount(); ++e) {      // further components on frames of their own (idocp_cost_t::task_extra), constant references
    RModel mt = task_robot_.model();
    mt.contact_frame_id[0] = -1; mt.contact_joint[0] = cost.taskJoint(e);
    for (int k2 = 0; k2 < 9; ++k2) mt.contact_R[0][k2] = cost.taskFrameR(e)[k2];
    for (int k2 = 0; k2 < 3; ++k2) mt.contact_p[0][k2] = cost.taskFrameP(e)[k2];
    Robot re(mt);
    real ce; Mat ge, He;
    re.taskSpaceTerms(cost.taskDim(e), cost.taskConstRef(e), cost.taskWeight(e, i == N_ ? 2 : 0), q, ce, ge, He);
    c += ce; g += ge; H += He;
  }
}

void UnOCPSolver::setSolution(const std::string& name, const Mat& value) {
  for (auto& e : s) {
    if (name == "q") e.q = value;
    else if (name == "v") e.v = value;
    else if (name == "a") e.a = value;
    else if (name == "u") e.u = value;
    else throw std::invalid_argument("invalid arugment: name must be q, v, a, or u!");
  }
  initConstraints();
}

// SplitUnOCP::initConstraints (split_unocp.hxx:61-66) -> createConstraintsData
// + setSlackAndDual -> pdipm::SetSlackAndDualPositive (pdipm.hxx:13-23)
void UnOCPSolver::initConstraints() {
  for (int i = 0; i < N_; ++i) {
    ConstraintsData& cd = ocp[i].cdata;
    cd.time_stage = i;
    cd.data.clear();
    for (const JointLimit& jl : constraints.components) {
      ConstraintComponentData data(jl.lim.size());
      if (constraints.valid(jl, i)) {
        const Mat& x = varOf(jl, s[i]);
        const int n = jl.lim.size(), off = x.size() - n;
        for (int r = 0; r < n; ++r) {
          data.slack[r] = -jl.sign * (x[off + r] - jl.lim[r]);
          while (data.slack[r] < constraints.barrier) data.slack[r] += constraints.barrier;
          data.dual[r] = constraints.barrier / data.slack[r];
        }
      }
      cd.data.push_back(data);
    }
  }
}

// ConfigurationSpaceCost::computeStageCostDerivatives (configuration_space_cost.cpp:292-310)
static void stageCostDerivatives(const RCost& c, real dt, const SplitSolution& s, SplitUnOCP& o) {
  const int nv = o.nv;
  for (int i = 0; i < nv; ++i) {
    o.lq[i] += dt * c.q_weight[i] * (s.q[i] - c.q_ref[i]);
    o.lv[i] += dt * c.v_weight[i] * (s.v[i] - c.v_ref[i]);
    o.la[i] += dt * c.a_weight[i] * s.a[i];
    o.lu[i] += dt * c.u_weight[i] * (s.u[i] - c.u_ref[i]);
  }
}

static Mat& residualOf(const JointLimit& jl, SplitUnOCP& o) {
  return jl.var == JointLimit::Q ? o.lq : (jl.var == JointLimit::V ? o.lv : (jl.var == JointLimit::A ? o.la : o.lu));
}
static Mat& hessianDiagOf(const JointLimit& jl, SplitUnOCP& o, Mat& qqdiag) {
  return jl.var == JointLimit::Q ? qqdiag : (jl.var == JointLimit::V ? o.Qvv_diag : (jl.var == JointLimit::A ? o.Qaa_diag : o.Quu_diag));
}

// stateequation::linearizeForwardEuler, fixed base (state_equation.hxx:12-37,210-221)
static void linearizeForwardEuler(real dt, const SplitSolution& s, const SplitSolution& sn, SplitUnOCP& o) {
  const int nv = o.nv;
  for (int i = 0; i < nv; ++i) {
    o.Fq[i] = s.q[i] - sn.q[i] + dt * s.v[i];
    o.Fv[i] = s.v[i] + dt * s.a[i] - sn.v[i];
    o.lq[i] += sn.lmd[i] - s.lmd[i];
    o.lv[i] += dt * sn.lmd[i] + sn.gmm[i] - s.gmm[i];
    o.la[i] += dt * sn.gmm[i];
  }
}

// SplitUnOCP::computeKKTResidual up to (and including) the dynamics
// (split_unocp.hxx:141-161); with_hessian=false.  linearizeOCP shares it.
void UnOCPSolver::computeStageResidual(Robot& robot, int i, real /*t*/) {
  SplitUnOCP& o = ocp[i];
  const SplitSolution& si = s[i];
  o.lq.setZero(); o.lv.setZero(); o.la.setZero(); o.lu.setZero(); o.Fq.setZero(); o.Fv.setZero();
  stageCostDerivatives(cost, dt_, si, o);
  if (cost.task_dim) { real c; Mat g, H; taskTerms(i, si.q, c, g, H); o.lq += dt_ * g; }      // TaskSpace*Cost::computeStageCostDerivatives
  // Constraints::computePrimalAndDualResidual + augmentDualResidual
  for (size_t c = 0; c < constraints.components.size(); ++c) {
    const JointLimit& jl = constraints.components[c];
    if (!constraints.valid(jl, i)) continue;
    ConstraintComponentData& data = o.cdata.data[c];
    computePrimalAndDualResidual(constraints, jl, data, si);
    Mat& l = residualOf(jl, o);
    const int n = jl.lim.size(), off = l.size() - n;
    for (int r = 0; r < n; ++r) l[off + r] += jl.sign * dt_ * data.dual[r];
  }
  linearizeForwardEuler(dt_, si, s[i + 1], o);
  // UnconstrainedDynamics::linearizeUnconstrainedDynamics (unconstrained_dynamics.hxx:55-65)
  robot.RNEA(si.q, si.v, si.a, o.ID);
  o.ID -= si.u;
  robot.RNEADerivatives(si.q, si.v, si.a, o.dID_dq, o.dID_dv, o.dID_da);
  o.lq += dt_ * (o.dID_dq.t() * si.beta);
  o.lv += dt_ * (o.dID_dv.t() * si.beta);
  o.la += dt_ * (o.dID_da.t() * si.beta);
  o.lu -= dt_ * si.beta;
}

// SplitUnOCP::linearizeOCP (split_unocp.hxx:69-99)
void UnOCPSolver::linearizeStage(Robot& robot, int i, real t, const Mat& /*q_prev*/) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  SplitUnOCP& o = ocp[i];
  const SplitSolution& si = s[i];
  const int nv = o.nv;
  o.Qqq.setZero(); o.Qvv_diag.setZero(); o.Qaa_diag.setZero(); o.Quu_diag.setZero();
  o.lq.setZero(); o.lv.setZero(); o.la.setZero(); o.lu.setZero();
  stageCostDerivatives(cost, dt_, si, o);
  Mat task_H;
  if (cost.task_dim) { real c; Mat g; taskTerms(i, si.q, c, g, task_H); o.lq += dt_ * g; }    // TaskSpace*Cost::computeStageCostDerivatives
  for (size_t c = 0; c < constraints.components.size(); ++c) {          // augmentDualResidual
    const JointLimit& jl = constraints.components[c];
    if (!constraints.valid(jl, i)) continue;
    const ConstraintComponentData& data = o.cdata.data[c];
    Mat& l = residualOf(jl, o);
    const int n = jl.lim.size(), off = l.size() - n;
    for (int r = 0; r < n; ++r) l[off + r] += jl.sign * dt_ * data.dual[r];
  }
  linearizeForwardEuler(dt_, si, s[i + 1], o);
  robot.RNEA(si.q, si.v, si.a, o.ID);
  o.ID -= si.u;
  robot.RNEADerivatives(si.q, si.v, si.a, o.dID_dq, o.dID_dv, o.dID_da);
  o.lq += dt_ * (o.dID_dq.t() * si.beta);
  o.lv += dt_ * (o.dID_dv.t() * si.beta);
  o.la += dt_ * (o.dID_da.t() * si.beta);
  o.lu -= dt_ * si.beta;
  // ConfigurationSpaceCost::computeStageCostHessian (configuration_space_cost.cpp:351-365)
  Mat Qqq_diag(nv);
  for (int r = 0; r < nv; ++r) {
    Qqq_diag[r] += dt_ * cost.q_weight[r];
    o.Qvv_diag[r] += dt_ * cost.v_weight[r];
    o.Qaa_diag[r] += dt_ * cost.a_weight[r];
    o.Quu_diag[r] += dt_ * cost.u_weight[r];
  }
  // Constraints::condenseSlackAndDual (e.g. joint_torques_upper_limit.cpp:62-73)
  for (size_t c = 0; c < constraints.components.size(); ++c) {
    const JointLimit& jl = constraints.components[c];
    if (!constraints.valid(jl, i)) continue;
    ConstraintComponentData& data = o.cdata.data[c];
    Mat& l = residualOf(jl, o);
    Mat& H = hessianDiagOf(jl, o, Qqq_diag);
    const int n = jl.lim.size(), off = l.size() - n;
    for (int r = 0; r < n; ++r) H[off + r] += dt_ * data.dual[r] / data.slack[r];
    computePrimalAndDualResidual(constraints, jl, data, si);
    for (int r = 0; r < n; ++r)
      l[off + r] += jl.sign * dt_ * (data.dual[r] * data.residual[r] - data.duality[r]) / data.slack[r];
  }
  for (int r = 0; r < nv; ++r) o.Qqq(r, r) = Qqq_diag[r];
  if (cost.task_dim) o.Qqq += dt_ * task_H;                        // TaskSpace*Cost::computeStageCostHessian (Gauss-Newton)
  // UnconstrainedDynamics::condenseUnconstrainedDynamics (unconstrained_dynamics.hxx:68-94)
  FLOP_REGION_SET(R_UNCONDENSE);
  SplitUnKKTMatrix& Q = unkkt_matrix[i];
  SplitUnKKTResidual& R = unkkt_residual[i];
  for (int r = 0; r < nv; ++r) o.lu_condensed[r] = o.lu[r] + o.Quu_diag[r] * o.ID[r];
  R.lq = o.lq + o.dID_dq.t() * o.lu_condensed;
  R.lv = o.lv + o.dID_dv.t() * o.lu_condensed;
  R.la = o.la + o.dID_da.t() * o.lu_condensed;
  R.Fq = o.Fq; R.Fv = o.Fv;
  Mat Quu_dq = o.dID_dq, Quu_dv = o.dID_dv, Quu_da = o.dID_da;
  for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) {
    Quu_dq(r, c) *= o.Quu_diag[r]; Quu_dv(r, c) *= o.Quu_diag[r]; Quu_da(r, c) *= o.Quu_diag[r];
  }
  // block order (a, q, v): Qaa=(0,0) Qaq=(0,1) Qav=(0,2) Qqq=(1,1) Qqv=(1,2) Qvv=(2,2)
  Q.Q.setZero();
  Q.set(1, 1, o.dID_dq.t() * Quu_dq);
  Q.set(1, 2, o.dID_dq.t() * Quu_dv);
  Q.set(2, 2, o.dID_dv.t() * Quu_dv);
  Q.set(0, 1, o.dID_da.t() * Quu_dq);
  Q.set(0, 2, o.dID_da.t() * Quu_dv);
  Q.set(0, 0, o.dID_da.t() * Quu_da);
  Q.add(1, 1, o.Qqq);
  for (int r = 0; r < nv; ++r) { Q.Q(2 * nv + r, 2 * nv + r) += o.Qvv_diag[r]; Q.Q(r, r) += o.Qaa_diag[r]; }
  (void)t;
}

// TerminalOCP::linearizeOCP (terminal_ocp.hxx:50-66), fixed base
void UnOCPSolver::linearizeTerminal(real /*t*/) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const SplitSolution& sN = s[N_];
  const int nv = robot.dimv();
  terminal_lq.setZero(); terminal_lv.setZero();
  for (int i = 0; i < nv; ++i) {            // computeTerminalCostDerivatives (configuration_space_cost.cpp:313-329)
    terminal_lq[i] += cost.qf_weight[i] * (sN.q[i] - cost.q_ref[i]);
    terminal_lv[i] += cost.vf_weight[i] * (sN.v[i] - cost.v_ref[i]);
  }
  terminal_lq -= sN.lmd;                    // linearizeForwardEulerTerminal (state_equation.hxx:66-83)
  terminal_lv -= sN.gmm;
  terminal_Qqq.setZero(); terminal_Qvv.setZero();
  for (int i = 0; i < nv; ++i) { terminal_Qqq(i, i) += cost.qf_weight[i]; terminal_Qvv(i, i) += cost.vf_weight[i]; }
  if (cost.task_dim) { real c; Mat g, H; taskTerms(N_, sN.q, c, g, H); terminal_lq += g; terminal_Qqq += H; }     // computeTerminalCostDerivatives / Hessian
}

void UnOCPSolver::linearizeOCP(real t, const Mat& q) {
  if ((int)robots_.size() != nthreads) setNumThreads(nthreads);
  #pragma omp parallel for num_threads(nthreads)
  for (int i = 0; i <= N_; ++i) {
    Robot& rb = robots_[ORACLE_THREAD_NUM];
    if (i == 0) linearizeStage(rb, 0, t, q);
    else if (i < N_) linearizeStage(rb, i, t + i * dt_, s[i - 1].q);
    else linearizeTerminal(t + T_);
  }
}

// UnRiccatiRecursion::backwardRiccatiRecursionTerminal + backwardRiccatiRecursion
// (unriccati_recursion.cpp:39-58)
void UnOCPSolver::backwardRiccatiRecursion() {
  FLOP_REGION(R_RICCATI_BWD);
  const int nv = robot.dimv();
  riccati[N_].Pqq = terminal_Qqq;
  riccati[N_].Pvv = terminal_Qvv;
  riccati[N_].sq = -terminal_lq;
  riccati[N_].sv = -terminal_lv;
  const real dt = dt_;
  for (int i = N_ - 1; i >= 0; --i) {
    const SplitRiccatiFactorization& rn = riccati[i + 1];
    SplitUnKKTMatrix& Q = unkkt_matrix[i];
    SplitUnKKTResidual& R = unkkt_residual[i];
    // BackwardUnRiccatiRecursionFactorizer::factorizeKKTMatrix
    // (backward_unriccati_recursion_factorizer.hxx:29-54)
    Q.add(1, 1, rn.Pqq);
    Q.add(1, 2, rn.Pqq, dt);
    Q.add(1, 2, rn.Pqv);
    Q.set(2, 1, Q.blk(1, 2).t());
    Q.add(2, 2, rn.Pqq, dt * dt);
    Q.add(2, 2, rn.Pqv, dt);
    Q.add(2, 2, rn.Pqv.t(), dt);
    Q.add(2, 2, rn.Pvv);
    Q.add(0, 1, rn.Pqv.t(), dt);               // Qaq^T += dt Pqv
    Q.add(0, 2, rn.Pqv.t(), dt * dt);          // Qav^T += dt^2 Pqv
    Q.add(0, 2, rn.Pvv.t(), dt);               // Qav^T += dt Pvv
    Q.add(0, 0, rn.Pvv, dt * dt);
    R.la += dt * (rn.Pqv.t() * R.Fq);
    R.la += dt * (rn.Pvv * R.Fv);
    R.la -= dt * rn.sv;
    // SplitUnRiccatiFactorizer::backwardRiccatiRecursion (split_unriccati_factorizer.hxx:30-46)
    LLT llt;
    if (!llt.compute(Q.blk(0, 0))) throw std::runtime_error("Riccati: Qaa not positive definite at stage " + std::to_string(i));
    Mat Qax = Q.Q.block(0, nv, nv, 2 * nv);
    K[i] = -llt.solve(Qax);
    k[i] = -llt.solve(R.la);
    // factorizeRiccatiFactorization (backward_unriccati_recursion_factorizer.hxx:57-89)
    SplitRiccatiFactorization& r = riccati[i];
    r.Pqq = Q.blk(1, 1); r.Pqv = Q.blk(1, 2); r.Pvv = Q.blk(2, 2);
    Mat GK = Q.blk(0, 0) * K[i];
    Mat Kq = K[i].block(0, 0, nv, nv), Kv = K[i].block(0, nv, nv, nv);
    Mat GKq = GK.block(0, 0, nv, nv), GKv = GK.block(0, nv, nv, nv);
    r.Pqq -= Kq.t() * GKq;
    r.Pqv -= Kq.t() * GKv;
    r.Pvv -= Kv.t() * GKv;
    r.Pvq = r.Pqv.t();
    r.Pqq = 0.5 * (r.Pqq + r.Pqq.t());
    r.Pvv = 0.5 * (r.Pvv + r.Pvv.t());
    r.sq = rn.sq;
    r.sq -= rn.Pqq * R.Fq;
    r.sq -= rn.Pqv * R.Fv;
    r.sv = rn.sv;
    r.sv += dt * r.sq;
    r.sv -= rn.Pqv.t() * R.Fq;
    r.sv -= rn.Pvv * R.Fv;
    r.sq -= R.lq;
    r.sv -= R.lv;
    r.sq -= Q.blk(0, 1).t() * k[i];
    r.sv -= Q.blk(0, 2).t() * k[i];
  }
}

// d[0] initial state (unocp_solver.cpp:100-101) + UnRiccatiRecursion::forwardRiccatiRecursion
// (unriccati_recursion.cpp:60-65; split_unriccati_factorizer.hxx:49-57)
void UnOCPSolver::forwardRiccatiRecursion(const Mat& q, const Mat& v) {
  FLOP_REGION(R_RICCATI_FWD);
  const int nv = robot.dimv();
  d[0].dq = q - s[0].q;
  d[0].dv = v - s[0].v;
  for (int i = 0; i < N_; ++i) {
    Mat dx(2 * nv); dx.setSegment(0, d[i].dq); dx.setSegment(nv, d[i].dv);
    d[i].da = K[i] * dx + k[i];
    d[i + 1].dq = unkkt_residual[i].Fq + d[i].dq;
    d[i + 1].dv = unkkt_residual[i].Fv + d[i].dv;
    d[i + 1].dq += dt_ * d[i].dv;
    d[i + 1].dv += dt_ * d[i].da;
  }
}

// pdipm::FractionToBoundary (pdipm.hxx:52-73)
static real fractionToBoundary(real rate, const Mat& vec, const Mat& dvec) {
  real m = 1;
  for (int i = 0; i < vec.size(); ++i) {
    const real f = -rate * (vec[i] / dvec[i]);
    if (f > 0 && f < 1 && f < m) m = f;
  }
  return m;
}

// Pieces of SplitUnOCP / SplitUnParNMPC::stageCost and constraintViolation (split_unocp.hxx:177-217,
// split_unparnmpc.hxx:179-227) at the trial point s + alpha d of one stage: the configuration-space cost
// (configuration_space_cost.cpp:241-256), the barrier on slack + alpha dslack (pdipm.hxx:84-87), dt |ID|_1 and
// dt |g(x_try) + slack|_1 with the CURRENT slack.
struct TrialPoint { Mat q, v, a, u; };
static TrialPoint trialPoint(const SplitSolution& s, const SplitDirection& d, real alpha) {
  return {s.q + alpha * d.dq, s.v + alpha * d.dv, s.a + alpha * d.da, s.u + alpha * d.du};
}
static real trialStageCost(const RCost& c, const Constraints& cs, const ConstraintsData& cd, int level, real dt,
                             const TrialPoint& x, real alpha, bool terminal_cost) {
  const int nv = x.v.size();
  real l = 0, lf = 0;
  for (int r = 0; r < nv; ++r) {
    l += c.q_weight[r] * (x.q[r] - c.q_ref[r]) * (x.q[r] - c.q_ref[r]) + c.v_weight[r] * (x.v[r] - c.v_ref[r]) * (x.v[r] - c.v_ref[r]) +
         c.a_weight[r] * x.a[r] * x.a[r] + c.u_weight[r] * (x.u[r] - c.u_ref[r]) * (x.u[r] - c.u_ref[r]);
    lf += c.qf_weight[r] * (x.q[r] - c.q_ref[r]) * (x.q[r] - c.q_ref[r]) + c.vf_weight[r] * (x.v[r] - c.v_ref[r]) * (x.v[r] - c.v_ref[r]);
  }
  real cost = 0.5 * dt * l + (terminal_cost ? 0.5 * lf : real(0.0));
  for (size_t j = 0; j < cs.components.size(); ++j) {
    if (!cs.valid(cs.components[j], level)) continue;
    const ConstraintComponentData& data = cd.data[j];
    real sum = 0;
    for (int r = 0; r < data.slack.size(); ++r) sum += std::log(data.slack[r] + alpha * data.dslack[r]);
    cost += dt * (-cs.barrier * sum);
  }
  return cost;
}
static real trialConstraintViolation(Robot& robot, const Constraints& cs, const ConstraintsData& cd, int level, real dt,
                                       const TrialPoint& x) {
  Mat ID(x.v.size());
  robot.RNEA(x.q, x.v, x.a, ID);
  real viol = 0;
  for (int r = 0; r < ID.size(); ++r) viol += dt * std::fabs(ID[r] - x.u[r]);
  for (size_t j = 0; j < cs.components.size(); ++j) {
    const JointLimit& jl = cs.components[j];
    if (!cs.valid(jl, level)) continue;
    const Mat& var = jl.var == JointLimit::Q ? x.q : (jl.var == JointLimit::V ? x.v : (jl.var == JointLimit::A ? x.a : x.u));
    const int n = jl.lim.size(), off = var.size() - n;
    for (int r = 0; r < n; ++r) viol += dt * std::fabs(jl.sign * (var[off + r] - jl.lim[r]) + cd.data[j].slack[r]);
  }
  return viol;
}

// second parallel loop of UnOCPSolver::updateSolution (unocp_solver.cpp:103-115)
void UnOCPSolver::computeDirection() {
  FLOP_REGION(R_EXPAND);
  real pmin = 1, dmin = 1;
  #pragma omp parallel for num_threads(nthreads) reduction(min : pmin, dmin)
  for (int i = 0; i <= N_; ++i) {
    const SplitRiccatiFactorization& r = riccati[i];
    // SplitUnRiccatiFactorizer::computeCostateDirection (split_unriccati_factorizer.hxx:60-68)
    d[i].dlmd = r.Pqq * d[i].dq;
    d[i].dlmd += r.Pqv * d[i].dv;
    d[i].dlmd -= r.sq;
    d[i].dgmm = r.Pqv.t() * d[i].dq;
    d[i].dgmm += r.Pvv * d[i].dv;
    d[i].dgmm -= r.sv;
    if (i < N_) {
      SplitUnOCP& o = ocp[i];
      // UnconstrainedDynamics::computeCondensedDirection (unconstrained_dynamics.hxx:97-106)
      d[i].du = o.ID;
      d[i].du += o.dID_dq * d[i].dq;
      d[i].du += o.dID_dv * d[i].dv;
      d[i].du += o.dID_da * d[i].da;
      for (int r2 = 0; r2 < o.nv; ++r2) d[i].dbeta[r2] = (o.lu[r2] + o.Quu_diag[r2] * d[i].du[r2]) / dt_;
      // Constraints::computeSlackAndDualDirection + step sizes
      for (size_t c = 0; c < constraints.components.size(); ++c) {
        const JointLimit& jl = constraints.components[c];
        if (!constraints.valid(jl, i)) continue;
        ConstraintComponentData& data = o.cdata.data[c];
        const Mat& dx = dvarOf(jl, d[i]);
        const int n = jl.lim.size(), off = dx.size() - n;
        for (int r2 = 0; r2 < n; ++r2) {
          data.dslack[r2] = -jl.sign * dx[off + r2] - data.residual[r2];
          data.ddual[r2] = -(data.dual[r2] * data.dslack[r2] + data.duality[r2]) / data.slack[r2];   // pdipm.hxx:76-81
        }
        const real ps = fractionToBoundary(constraints.fraction_to_boundary_rate, data.slack, data.dslack);
        const real ds = fractionToBoundary(constraints.fraction_to_boundary_rate, data.dual, data.ddual);
        if (ps < pmin) pmin = ps;
        if (ds < dmin) dmin = ds;
      }
    }
  }
  primal_step_size = pmin; dual_step_size = dmin;
}

// third parallel loop (unocp_solver.cpp:121-133): updatePrimal / updateDual
void UnOCPSolver::integrate() {
  FLOP_REGION(R_INTEGRATE);
  const real ap = primal_step_size, ad = dual_step_size;
  #pragma omp parallel for num_threads(nthreads)
  for (int i = 0; i <= N_; ++i) {
    s[i].lmd += ap * d[i].dlmd;
    s[i].gmm += ap * d[i].dgmm;
    s[i].q += ap * d[i].dq;
    s[i].v += ap * d[i].dv;
    if (i < N_) {
      s[i].a += ap * d[i].da;
      s[i].u += ap * d[i].du;
      s[i].beta += ap * d[i].dbeta;
      for (size_t c = 0; c < constraints.components.size(); ++c) {
        if (!constraints.valid(constraints.components[c], i)) continue;
        ConstraintComponentData& data = ocp[i].cdata.data[c];
        data.slack += ap * data.dslack;
        data.dual += ad * data.ddual;
      }
    }
  }
}

// UnLineSearch::computeCostAndViolation(UnOCP&, ...) (unline_search.cpp:55-82): forward-Euler residual against the trial
// point of the next stage, terminal cost of stage N
std::pair<real, real> UnOCPSolver::costAndViolation(real alpha) const {
  Robot rb = robot;
  real cost_sum = 0, viol = 0;
  for (int i = 0; i < N_; ++i) {
    const TrialPoint x = trialPoint(s[i], d[i], alpha);
    const Mat qn = s[i + 1].q + alpha * d[i + 1].dq, vn = s[i + 1].v + alpha * d[i + 1].dv;
    cost_sum += trialStageCost(cost, constraints, ocp[i].cdata, i, dt_, x, alpha, false);
    if (cost.task_dim) { real c; Mat g, H; taskTerms(i, x.q, c, g, H); cost_sum += dt_ * c; }
    for (int r = 0; r < rb.dimv(); ++r) viol += std::fabs(x.q[r] - qn[r] + dt_ * x.v[r]) + std::fabs(x.v[r] + dt_ * x.a[r] - vn[r]);
    viol += trialConstraintViolation(rb, constraints, ocp[i].cdata, i, dt_, x);
  }
  const Mat qN = s[N_].q + alpha * d[N_].dq, vN = s[N_].v + alpha * d[N_].dv;
  real lf = 0;
  for (int r = 0; r < rb.dimv(); ++r)
    lf += cost.qf_weight[r] * (qN[r] - cost.q_ref[r]) * (qN[r] - cost.q_ref[r]) + cost.vf_weight[r] * (vN[r] - cost.v_ref[r]) * (vN[r] - cost.v_ref[r]);
  if (cost.task_dim) { real c; Mat g, H; taskTerms(N_, qN, c, g, H); cost_sum += c; }
  return {cost_sum + 0.5 * lf, viol};
}

void UnOCPSolver::updateSolution(real t, const Mat& q, const Mat& v, bool use_line_search) {
  linearizeOCP(t, q);
  auto t0 = std::chrono::steady_clock::now();
  backwardRiccatiRecursion();
  forwardRiccatiRecursion(q, v);
  riccati_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  computeDirection();
  if (use_line_search)            // unocp_solver.cpp:116-120
    primal_step_size = line_search.computeStepSize([&](real a) { return costAndViolation(a); }, primal_step_size);
  integrate();
}

void UnOCPSolver::computeKKTResidual(real t, const Mat& /*q*/, const Mat& /*v*/) {
  if ((int)robots_.size() != nthreads) setNumThreads(nthreads);
  #pragma omp parallel for num_threads(nthreads)
  for (int i = 0; i < N_; ++i) computeStageResidual(robots_[ORACLE_THREAD_NUM], i, t + i * dt_);
  // TerminalOCP::computeKKTResidual (terminal_ocp.hxx:118-131)
  const SplitSolution& sN = s[N_];
  terminal_lq.setZero(); terminal_lv.setZero();
  for (int i = 0; i < robot.dimv(); ++i) {
    terminal_lq[i] += cost.qf_weight[i] * (sN.q[i] - cost.q_ref[i]);
    terminal_lv[i] += cost.vf_weight[i] * (sN.v[i] - cost.v_ref[i]);
  }
  if (cost.task_dim) { real c; Mat g, H; taskTerms(N_, sN.q, c, g, H); terminal_lq += g; }
  terminal_lq -= sN.lmd;
  terminal_lv -= sN.gmm;
}

// UnOCPSolver::KKTError (unocp_solver.cpp:190-202) / squaredNormKKTResidual (split_unocp.hxx:164-174)
// UnOCPSolver::isCurrentSolutionFeasible (unocp_solver.cpp:228-237) with the component tests of
// joint_{position,velocity,torques}_{lower,upper}_limit.cpp:36-47
int UnOCPSolver::isCurrentSolutionFeasible() const {
  for (int i = 0; i < N_; ++i)
    for (const JointLimit& jl : constraints.components) {
      if (!constraints.valid(jl, i)) continue;
      const Mat& x = varOf(jl, s[i]);
      const int n = jl.lim.size(), off = x.size() - n;
      for (int r = 0; r < n; ++r)
        if (jl.sign < 0 ? x[off + r] < jl.lim[r] : x[off + r] > jl.lim[r]) return i;
    }
  return -1;
}

real UnOCPSolver::KKTError() {
  real sum = 0;
  for (int i = 0; i < N_; ++i) {
    const SplitUnOCP& o = ocp[i];
    real e = o.lq.squaredNorm() + o.lv.squaredNorm() + o.la.squaredNorm() + o.lu.squaredNorm();
    e += o.Fq.squaredNorm() + o.Fv.squaredNorm();
    e += dt_ * dt_ * o.ID.squaredNorm();
    real c = 0;
    for (size_t j = 0; j < constraints.components.size(); ++j) {
      if (!constraints.valid(constraints.components[j], i)) continue;
      c += o.cdata.data[j].residual.squaredNorm() + o.cdata.data[j].duality.squaredNorm();
    }
    e += dt_ * dt_ * c;
    sum += e;
  }
  sum += terminal_lq.squaredNorm() + terminal_lv.squaredNorm();
  return std::sqrt(sum);
}


// ======================================================================= UnParNMPC ====
UnParNMPCSolver::UnParNMPCSolver(const RModel& model, const RCost& cost_, const idocp_constraints_t& cons, real T, int N)
    : robot(model), cost(cost_), constraints(robot, cons),
      s(N, SplitSolution(robot)), s_new(N, SplitSolution(robot)), d(N, SplitDirection(robot)), ocp(N, SplitUnOCP(model.nv)),
      unkkt_matrix(N, SplitUnKKTMatrix(model.nv)), unkkt_residual(N, SplitUnKKTResidual(model.nv)),
      aux_mat(N, Mat(2 * model.nv, 2 * model.nv)), kkt_inv(N, Mat(5 * model.nv, 5 * model.nv)), x_res(N, Mat(2 * model.nv)),
      N_(N), T_(T), dt_(T / N), lo_(0), hi_(N) {
  if (T <= 0) throw std::out_of_range("invalid value: T must be positive!");
  if (N <= 0) throw std::out_of_range("invalid value: N must be positive!");
  if (robot.hasFloatingBase() || robot.maxPointContacts() > 0)
    throw std::logic_error("robot has floating base or contacts: use ParNMPCSolver");   // split_unparnmpc.hxx:27-33
  if (cost.task_dim != 0) throw std::logic_error("task-space costs are restated for UnOCPSolver only");
  initConstraints();
}

void UnParNMPCSolver::setSolution(const std::string& name, const Mat& value) {
  for (auto& e : s) {
    if (name == "q") e.q = value;
    else if (name == "v") e.v = value;
    else if (name == "a") e.a = value;
    else if (name == "u") e.u = value;
    else throw std::invalid_argument("invalid arugment: name must be q, v, a, or u!");
  }
  initConstraints();
}

// stage i is created with time step i + 1 (unparnmpc_solver.cpp:55-66): only stage 0 lacks the position-level rows
void UnParNMPCSolver::initConstraints() {
  for (int i = 0; i < N_; ++i) {
    ConstraintsData& cd = ocp[i].cdata;
    cd.time_stage = i + 1;
    cd.data.clear();
    for (const JointLimit& jl : constraints.components) {
      ConstraintComponentData data(jl.lim.size());
      if (constraints.valid(jl, i + 1)) {
        const Mat& x = varOf(jl, s[i]);
        const int n = jl.lim.size(), off = x.size() - n;
        for (int r = 0; r < n; ++r) {
          data.slack[r] = -jl.sign * (x[off + r] - jl.lim[r]);
          while (data.slack[r] < constraints.barrier) data.slack[r] += constraints.barrier;
          data.dual[r] = constraints.barrier / data.slack[r];
        }
      }
      cd.data.push_back(data);
    }
  }
}

// UnBackwardCorrection::initAuxMat: the terminal cost Hessian (configuration_space_cost.cpp:368-380) on every stage
void UnParNMPCSolver::initBackwardCorrection(real /*t*/) {
  const int nv = robot.dimv();
  for (int i = 0; i < N_; ++i) {
    aux_mat[i].setZero();
    for (int r = 0; r < nv; ++r) { aux_mat[i](r, r) = cost.qf_weight[r]; aux_mat[i](nv + r, nv + r) = cost.vf_weight[r]; }
  }
}

// SplitUnParNMPC::linearizeOCP / computeKKTResidual (split_unparnmpc.hxx:69-102, 142-163) and the terminal twins
// (terminal_unparnmpc.hxx:69-102, 147-168): stage i < N - 1 couples to s[i + 1] through the costate, the last stage
// carries the terminal cost instead.
void UnParNMPCSolver::linearizeStage(int i, const Mat& q_prev, const Mat& v_prev, bool residual_only) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  SplitUnOCP& o = ocp[i];
  const SplitSolution& si = s[i];
  const bool terminal = (i == N_ - 1);
  const int nv = o.nv, level = i + 1;
  if (!residual_only) { o.Qqq.setZero(); o.Qvv_diag.setZero(); o.Qaa_diag.setZero(); o.Quu_diag.setZero(); }
  o.lq.setZero(); o.lv.setZero(); o.la.setZero(); o.lu.setZero();
  stageCostDerivatives(cost, dt_, si, o);
  if (terminal)                                        // computeTerminalCostDerivatives (configuration_space_cost.cpp:313-329)
    for (int r = 0; r < nv; ++r) {
      o.lq[r] += cost.qf_weight[r] * (si.q[r] - cost.q_ref[r]);
      o.lv[r] += cost.vf_weight[r] * (si.v[r] - cost.v_ref[r]);
    }
  for (size_t c = 0; c < constraints.components.size(); ++c) {          // [computePrimalAndDualResidual +] augmentDualResidual
    const JointLimit& jl = constraints.components[c];
    if (!constraints.valid(jl, level)) continue;
    ConstraintComponentData& data = o.cdata.data[c];
    if (residual_only) computePrimalAndDualResidual(constraints, jl, data, si);
    Mat& l = residualOf(jl, o);
    const int n = jl.lim.size(), off = l.size() - n;
    for (int r = 0; r < n; ++r) l[off + r] += jl.sign * dt_ * data.dual[r];
  }
  // stateequation::linearizeBackwardEuler / ...Terminal, fixed base (state_equation.hxx:111-167)
  for (int r = 0; r < nv; ++r) {
    o.Fq[r] = q_prev[r] - si.q[r] + dt_ * si.v[r];
    o.Fv[r] = v_prev[r] - si.v[r] + dt_ * si.a[r];
    if (terminal) { o.lq[r] -= si.lmd[r]; o.lv[r] += dt_ * si.lmd[r] - si.gmm[r]; }
    else { o.lq[r] += s[i + 1].lmd[r] - si.lmd[r]; o.lv[r] += dt_ * si.lmd[r] - si.gmm[r] + s[i + 1].gmm[r]; }
    o.la[r] += dt_ * si.gmm[r];
  }
  // UnconstrainedDynamics::linearizeUnconstrainedDynamics (unconstrained_dynamics.hxx:55-65)
  robot.RNEA(si.q, si.v, si.a, o.ID);
  o.ID -= si.u;
  robot.RNEADerivatives(si.q, si.v, si.a, o.dID_dq, o.dID_dv, o.dID_da);
  o.lq += dt_ * (o.dID_dq.t() * si.beta);
  o.lv += dt_ * (o.dID_dv.t() * si.beta);
  o.la += dt_ * (o.dID_da.t() * si.beta);
  o.lu -= dt_ * si.beta;
  if (residual_only) return;
  // computeStageCostHessian [+ computeTerminalCostHessian] (configuration_space_cost.cpp:351-380)
  Mat Qqq_diag(nv);
  for (int r = 0; r < nv; ++r) {
    Qqq_diag[r] += dt_ * cost.q_weight[r] + (terminal ? cost.qf_weight[r] : real(0.0));
    o.Qvv_diag[r] += dt_ * cost.v_weight[r] + (terminal ? cost.vf_weight[r] : real(0.0));
    o.Qaa_diag[r] += dt_ * cost.a_weight[r];
    o.Quu_diag[r] += dt_ * cost.u_weight[r];
  }
  for (size_t c = 0; c < constraints.components.size(); ++c) {          // condenseSlackAndDual
    const JointLimit& jl = constraints.components[c];
    if (!constraints.valid(jl, level)) continue;
    ConstraintComponentData& data = o.cdata.data[c];
    Mat& l = residualOf(jl, o);
    Mat& H = hessianDiagOf(jl, o, Qqq_diag);
    const int n = jl.lim.size(), off = l.size() - n;
    for (int r = 0; r < n; ++r) H[off + r] += dt_ * data.dual[r] / data.slack[r];
    computePrimalAndDualResidual(constraints, jl, data, si);
    for (int r = 0; r < n; ++r) l[off + r] += jl.sign * dt_ * (data.dual[r] * data.residual[r] - data.duality[r]) / data.slack[r];
  }
  for (int r = 0; r < nv; ++r) o.Qqq(r, r) = Qqq_diag[r];
  // condenseUnconstrainedDynamics (unconstrained_dynamics.hxx:68-94)
  SplitUnKKTMatrix& Q = unkkt_matrix[i];
  SplitUnKKTResidual& R = unkkt_residual[i];
  for (int r = 0; r < nv; ++r) o.lu_condensed[r] = o.lu[r] + o.Quu_diag[r] * o.ID[r];
  R.lq = o.lq + o.dID_dq.t() * o.lu_condensed;
  R.lv = o.lv + o.dID_dv.t() * o.lu_condensed;
  R.la = o.la + o.dID_da.t() * o.lu_condensed;
  R.Fq = o.Fq; R.Fv = o.Fv;
  Mat Quu_dq = o.dID_dq, Quu_dv = o.dID_dv, Quu_da = o.dID_da;
  for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) {
    Quu_dq(r, c) *= o.Quu_diag[r]; Quu_dv(r, c) *= o.Quu_diag[r]; Quu_da(r, c) *= o.Quu_diag[r];
  }
  Q.Q.setZero();
  Q.set(1, 1, o.dID_dq.t() * Quu_dq);
  Q.set(1, 2, o.dID_dq.t() * Quu_dv);
  Q.set(2, 2, o.dID_dv.t() * Quu_dv);
  Q.set(0, 1, o.dID_da.t() * Quu_dq);
  Q.set(0, 2, o.dID_da.t() * Quu_dv);
  Q.set(0, 0, o.dID_da.t() * Quu_da);
  Q.add(1, 1, o.Qqq);
  for (int r = 0; r < nv; ++r) { Q.Q(2 * nv + r, 2 * nv + r) += o.Qvv_diag[r]; Q.Q(r, r) += o.Qaa_diag[r]; }
}

// UnBackwardCorrection::coarseUpdate (unbackward_correction.cpp:67-97) with SplitUnBackwardCorrection::coarseUpdate
// (split_unbackward_correction.hxx:38-64) and SplitUnKKTMatrixInverter::invert (split_unkkt_matrix_inverter.hxx:37-80)
void UnParNMPCSolver::coarseUpdate(real /*t*/, const Mat& q, const Mat& v) {
  FLOP_REGION(R_KKT_INVERSE);
  const int nv = robot.dimv(), nx = 2 * nv, nq3 = 3 * nv, nk = 5 * nv;
  for (int i = lo_; i < hi_; ++i) {
    linearizeStage(i, i == 0 ? q : s[i - 1].q, i == 0 ? v : s[i - 1].v, false);
    Mat& Q = unkkt_matrix[i].Q;
    if (i < N_ - 1) Q.addBlock(nv, nv, aux_mat[i + 1]);                         // Qxx += aux_mat_next
    for (int c = 0; c < nq3; ++c) for (int r = c + 1; r < nq3; ++r) Q(r, c) = Q(c, r);     // Qvq = Qqv^T, Qxa = Qax^T
    Mat& Ki = kkt_inv[i];
    Ki.setZero();
    LLT lltQ;
    if (!lltQ.compute(Q)) throw std::runtime_error("UnParNMPC: Q not positive definite at stage " + std::to_string(i));
    const Mat Qinv = lltQ.solve(Mat::Identity(nq3));
    // F = [0 -I dt I; dt I 0 -I] in the (a, q, v) ordering
    Mat FQinv(nx, nq3);
    for (int c = 0; c < nq3; ++c) for (int r = 0; r < nv; ++r) {
      FQinv(r, c) = -Qinv(nv + r, c) + dt_ * Qinv(2 * nv + r, c);
      FQinv(nv + r, c) = dt_ * Qinv(r, c) - Qinv(2 * nv + r, c);
    }
    Mat S(nx, nx);
    for (int r = 0; r < nx; ++r) for (int c = 0; c < nv; ++c) {
      S(r, c) = -FQinv(r, nv + c) + dt_ * FQinv(r, 2 * nv + c);
      S(r, nv + c) = dt_ * FQinv(r, c) - FQinv(r, 2 * nv + c);
    }
    LLT lltS;
    if (!lltS.compute(S)) throw std::runtime_error("UnParNMPC: S not positive definite at stage " + std::to_string(i));
    const Mat TL = -1.0 * lltS.solve(Mat::Identity(nx));
    const Mat TR = -1.0 * (TL * FQinv);
    Ki.setBlock(0, 0, TL);
    Ki.setBlock(0, nx, TR);
    Ki.setBlock(nx, 0, TR.t());
    Ki.setBlock(nx, nx, Qinv - TR.t() * (S * TR));
    // d = KKT_inv * residual ; s_new = s - d
    const SplitUnKKTResidual& R = unkkt_residual[i];
    Mat res(nk);
    res.setSegment(0, R.Fq); res.setSegment(nv, R.Fv); res.setSegment(2 * nv, R.la); res.setSegment(3 * nv, R.lq); res.setSegment(4 * nv, R.lv);
    const Mat dd = Ki * res;
    d[i].dlmd = dd.segment(0, nv); d[i].dgmm = dd.segment(nv, nv); d[i].da = dd.segment(2 * nv, nv);
    d[i].dq = dd.segment(3 * nv, nv); d[i].dv = dd.segment(4 * nv, nv);
    s_new[i] = s[i];
    s_new[i].lmd -= d[i].dlmd; s_new[i].gmm -= d[i].dgmm; s_new[i].a -= d[i].da; s_new[i].q -= d[i].dq; s_new[i].v -= d[i].dv;
  }
}

// split_unbackward_correction.hxx:72-81
void UnParNMPCSolver::backwardCorrectionSerial() {
  FLOP_REGION(R_CORRECTION);
  const int nv = robot.dimv(), nx = 2 * nv, nk = 5 * nv;
  for (int i = std::min(hi_ - 1, N_ - 2); i >= lo_; --i) {
    x_res[i].setSegment(0, s_new[i + 1].lmd - s[i + 1].lmd);
    x_res[i].setSegment(nv, s_new[i + 1].gmm - s[i + 1].gmm);
    const Mat dx = kkt_inv[i].block(0, nk - nx, nx, nx) * x_res[i];
    s_new[i].lmd -= dx.segment(0, nv);
    s_new[i].gmm -= dx.segment(nv, nv);
  }
}

// split_unbackward_correction.hxx:84-92
void UnParNMPCSolver::backwardCorrectionParallel() {
  FLOP_REGION(R_CORRECTION);
  const int nv = robot.dimv(), nx = 2 * nv, nk = 5 * nv;
  for (int i = std::min(hi_ - 1, N_ - 2); i >= lo_; --i) {
    const Mat dd = kkt_inv[i].block(nx, nk - nx, nk - nx, nx) * x_res[i];
    s_new[i].a -= dd.segment(0, nv);
    s_new[i].q -= dd.segment(nv, nv);
    s_new[i].v -= dd.segment(2 * nv, nv);
  }
}

// split_unbackward_correction.hxx:95-104
void UnParNMPCSolver::forwardCorrectionSerial() {
  FLOP_REGION(R_CORRECTION);
  const int nv = robot.dimv(), nx = 2 * nv, nk = 5 * nv;
  for (int i = std::max(lo_, 1); i < hi_; ++i) {
    x_res[i].setSegment(0, s_new[i - 1].q - s[i - 1].q);
    x_res[i].setSegment(nv, s_new[i - 1].v - s[i - 1].v);
    const Mat dx = kkt_inv[i].block(nk - nx, 0, nx, nx) * x_res[i];
    s_new[i].q -= dx.segment(0, nv);
    s_new[i].v -= dx.segment(nv, nv);
  }
}

// last parallel loop of UnBackwardCorrection::backwardCorrection (unbackward_correction.cpp:114-131):
// forwardCorrectionParallel + aux_mat, computeDirection (split_unbackward_correction.hxx:107-123), the condensed
// direction (unconstrained_dynamics.hxx:97-106), slack / dual directions and the step sizes
void UnParNMPCSolver::forwardCorrectionParallel() {
  FLOP_REGION(R_CORRECTION);
  const int nv = robot.dimv(), nx = 2 * nv, nk = 5 * nv;
  real pmin = 1, dmin = 1;
  for (int i = lo_; i < hi_; ++i) {
    if (i > 0) {
      const Mat dd = kkt_inv[i].block(0, 0, nk - nx, nx) * x_res[i];
      s_new[i].lmd -= dd.segment(0, nv);
      s_new[i].gmm -= dd.segment(nv, nv);
      s_new[i].a -= dd.segment(2 * nv, nv);
      aux_mat[i] = -1.0 * kkt_inv[i].block(0, 0, nx, nx);
    }
    d[i].dlmd = s_new[i].lmd - s[i].lmd;
    d[i].dgmm = s_new[i].gmm - s[i].gmm;
    d[i].da = s_new[i].a - s[i].a;
    d[i].dq = s_new[i].q - s[i].q;
    d[i].dv = s_new[i].v - s[i].v;
    SplitUnOCP& o = ocp[i];
    d[i].du = o.ID;
    d[i].du += o.dID_dq * d[i].dq;
    d[i].du += o.dID_dv * d[i].dv;
    d[i].du += o.dID_da * d[i].da;
    for (int r2 = 0; r2 < nv; ++r2) d[i].dbeta[r2] = (o.lu[r2] + o.Quu_diag[r2] * d[i].du[r2]) / dt_;
    for (size_t c = 0; c < constraints.components.size(); ++c) {
      const JointLimit& jl = constraints.components[c];
      if (!constraints.valid(jl, i + 1)) continue;
      ConstraintComponentData& data = o.cdata.data[c];
      const Mat& dx = dvarOf(jl, d[i]);
      const int n = jl.lim.size(), off = dx.size() - n;
      for (int r2 = 0; r2 < n; ++r2) {
        data.dslack[r2] = -jl.sign * dx[off + r2] - data.residual[r2];
        data.ddual[r2] = -(data.dual[r2] * data.dslack[r2] + data.duality[r2]) / data.slack[r2];
      }
      const real ps = fractionToBoundary(constraints.fraction_to_boundary_rate, data.slack, data.dslack);
      const real ds = fractionToBoundary(constraints.fraction_to_boundary_rate, data.dual, data.ddual);
      if (ps < pmin) pmin = ps;
      if (ds < dmin) dmin = ds;
    }
  }
  primal_step_size = pmin; dual_step_size = dmin;
}

// updatePrimal / updateDual of every stage (unparnmpc_solver.cpp:88-102; split_solution.hxx:215-240)
void UnParNMPCSolver::integrate() {
  FLOP_REGION(R_INTEGRATE);
  const real ap = primal_step_size, ad = dual_step_size;
  for (int i = lo_; i < hi_; ++i) {
    s[i].lmd += ap * d[i].dlmd; s[i].gmm += ap * d[i].dgmm; s[i].q += ap * d[i].dq; s[i].v += ap * d[i].dv;
    s[i].a += ap * d[i].da; s[i].u += ap * d[i].du; s[i].beta += ap * d[i].dbeta;
    for (size_t c = 0; c < constraints.components.size(); ++c) {
      if (!constraints.valid(constraints.components[c], i + 1)) continue;
      ConstraintComponentData& data = ocp[i].cdata.data[c];
      data.slack += ap * data.dslack;
      data.dual += ad * data.ddual;
    }
  }
}

// UnLineSearch::computeCostAndViolation(UnParNMPC&, ...) (unline_search.cpp:85-121): backward-Euler residual against the
// trial point of the previous stage (the measured state for stage 0), terminal cost on the last stage
std::pair<real, real> UnParNMPCSolver::costAndViolation(real alpha, const Mat& q, const Mat& v) const {
  Robot rb = robot;
  real cost_sum = 0, viol = 0;
  for (int i = 0; i < N_; ++i) {
    const TrialPoint x = trialPoint(s[i], d[i], alpha);
    const Mat qp = i == 0 ? q : s[i - 1].q + alpha * d[i - 1].dq, vp = i == 0 ? v : s[i - 1].v + alpha * d[i - 1].dv;
    cost_sum += trialStageCost(cost, constraints, ocp[i].cdata, i + 1, dt_, x, alpha, i == N_ - 1);
    for (int r = 0; r < rb.dimv(); ++r) viol += std::fabs(qp[r] - x.q[r] + dt_ * x.v[r]) + std::fabs(vp[r] - x.v[r] + dt_ * x.a[r]);
    viol += trialConstraintViolation(rb, constraints, ocp[i].cdata, i + 1, dt_, x);
  }
  return {cost_sum, viol};
}

void UnParNMPCSolver::updateSolution(real t, const Mat& q, const Mat& v, bool use_line_search) {
  coarseUpdate(t, q, v);
  backwardCorrectionSerial();
  backwardCorrectionParallel();
  forwardCorrectionSerial();
  forwardCorrectionParallel();
  if (use_line_search)            // unparnmpc_solver.cpp:81-86
    primal_step_size = line_search.computeStepSize([&](real a) { return costAndViolation(a, q, v); }, primal_step_size);
  integrate();
}

void UnParNMPCSolver::computeKKTResidual(real /*t*/, const Mat& q, const Mat& v) {
  for (int i = lo_; i < hi_; ++i) linearizeStage(i, i == 0 ? q : s[i - 1].q, i == 0 ? v : s[i - 1].v, true);
}

// squaredNormKKTResidual of every stage (split_unparnmpc.hxx:166-176, terminal_unparnmpc.hxx:171-181)
real UnParNMPCSolver::KKTError() { return std::sqrt(KKTErrorSquared()); }
real UnParNMPCSolver::KKTErrorSquared() {
  real sum = 0;
  for (int i = lo_; i < hi_; ++i) {
    const SplitUnOCP& o = ocp[i];
    real e = o.lq.squaredNorm() + o.lv.squaredNorm() + o.la.squaredNorm() + o.lu.squaredNorm();
    e += o.Fq.squaredNorm() + o.Fv.squaredNorm();
    e += dt_ * dt_ * o.ID.squaredNorm();
    real c = 0;
    for (size_t j = 0; j < constraints.components.size(); ++j) {
      if (!constraints.valid(constraints.components[j], i + 1)) continue;
      c += o.cdata.data[j].residual.squaredNorm() + o.cdata.data[j].duality.squaredNorm();
    }
    sum += e + dt_ * dt_ * c;
  }
  return sum;
}

int UnParNMPCSolver::isCurrentSolutionFeasible() const {
  for (int i = 0; i < N_; ++i)
    for (const JointLimit& jl : constraints.components) {
      if (!constraints.valid(jl, i + 1)) continue;
      const Mat& x = varOf(jl, s[i]);
      const int n = jl.lim.size(), off = x.size() - n;
      for (int r = 0; r < n; ++r)
        if (jl.sign < 0 ? x[off + r] < jl.lim[r] : x[off + r] > jl.lim[r]) return i;
    }
  return -1;
}

}  // namespace oracle
