// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md and ocp.hpp).
#include "ocp.hpp"

#include <chrono>
#include <cmath>
#include <limits>
#include <stdexcept>

#ifdef _OPENMP
#include <omp.h>
#define ORACLE_THREAD_NUM omp_get_thread_num()
#else
#define ORACLE_THREAD_NUM 0
#endif

namespace oracle {

// kP (a member of both solvers): the passive rows of a floating base, Robot::dim_passive -- 6, or 0 on a fixed-base robot

SplitSolutionC::SplitSolutionC(const Robot& r)
    : lmd(r.dimv()), gmm(r.dimv()), q(r.dimq()), v(r.dimv()), a(r.dimv()), u(r.dimu()), beta(r.dimv()), nu_passive(r.dimv() - r.dimu()),
      xi(3 * r.maxPointContacts()), f(r.maxPointContacts(), Mat(3)), mu(r.maxPointContacts(), Mat(3)) {
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
    : dlmd(r.dimv()), dgmm(r.dimv()), du(r.dimu()), dq(r.dimv()), dv(r.dimv()), daf(r.dimv()), dbetamu(r.dimv()), dnu_passive(r.dimv() - r.dimu()),
      dxi(0) {}
SplitKKTMatrixC::SplitKKTMatrixC(int nv_, int nu_)
    : nv(nv_), nu(nu_), Qxx(2 * nv_, 2 * nv_), Qxu_full(2 * nv_, nv_), Quu_full(nv_, nv_), Qaa_diag(nv_), Qff(0, 0),
      Fqq6(nv_ - nu_, nv_ - nu_), Fqv6(nv_ - nu_, nv_ - nu_), Fvq(nv_, nv_), Fvv(nv_, nv_), Fvu(nv_, nu_), Fqq_prev6(nv_ - nu_, nv_ - nu_),
      Fqq_inv(nv_ - nu_, nv_ - nu_), Fqq_prev_inv(nv_ - nu_, nv_ - nu_) {}
SplitKKTResidualC::SplitKKTResidualC(int nv, int nu)
    : Fq(nv), Fv(nv), lq(nv), lv(nv), la(nv), lf(0), lu(nu), lu_passive(nv - nu), Fq_prev(nv - nu), P(0) {}

int ContactSequenceC::eventOfImpulse(int k) const {
  for (int e = 0, n = 0; e < numEvents(); ++e) if (is_impulse[e]) { if (n == k) return e; ++n; }
  return -1;
}
int ContactSequenceC::eventOfLift(int k) const {
  for (int e = 0, n = 0; e < numEvents(); ++e) if (!is_impulse[e]) { if (n == k) return e; ++n; }
  return -1;
}

// TaskSpace3DCost / TaskSpace6DCost on the floating-base solvers (task_space_3d_cost.cpp:60-157, task_space_6d_cost.cpp:68-178): a copy of the
// robot with the task frame as its contact 0 evaluates cost = 1/2 sum w diff^2, grad = JJ^T (w o diff), hess = JJ^T diag(w) JJ
static Robot makeTaskRobot(const RModel& model, const RCost& cost) {
  if (cost.task_dim == 0) return Robot(model);
  if (cost.task_dim != 3 && cost.task_dim != 6) throw std::invalid_argument("task_dim must be 0, 3 or 6");
  RModel mt = model;
  mt.contact_frame_id[0] = -1; mt.contact_joint[0] = cost.task_joint;
  for (int k2 = 0; k2 < 9; ++k2) mt.contact_R[0][k2] = cost.task_frame_R[k2];
  for (int k2 = 0; k2 < 3; ++k2) mt.contact_p[0][k2] = cost.task_frame_p[k2];
  return Robot(mt);
}
static void taskTerms(const Robot& task_robot, const RCost& cost, real t, const real* w, const Mat& q, real& c, Mat& g, Mat& H) {
  Robot rb = task_robot;
  rb.taskSpaceTerms(cost.task_dim, cost.taskRefAt(t), w, q, c, g, H);      // (TimeVarying variants: the pose at the stage's own time)
  // further components (CostFunction sums its components, cost_function.hxx): the same term on their own frames, with the weights of the same
  // kind (stage / impulse / terminal) as `w` is of the first component; constant references
  const int kind = w == cost.task_weight ? 0 : (w == cost.task_weighti ? 1 : 2);
  for (int e = 1; e < cost.taskCount(); ++e) {
    RModel mt = task_robot.model();
    mt.contact_frame_id[0] = -1; mt.contact_joint[0] = cost.taskJoint(e);
    for (int k2 = 0; k2 < 9; ++k2) mt.contact_R[0][k2] = cost.taskFrameR(e)[k2];
    for (int k2 = 0; k2 < 3; ++k2) mt.contact_p[0][k2] = cost.taskFrameP(e)[k2];
    Robot re(mt);
    real ce; Mat ge, He;
    re.taskSpaceTerms(cost.taskDim(e), cost.taskConstRef(e), cost.taskWeight(e, kind), q, ce, ge, He);
    c += ce; g += ge; H += He;
  }
}

OCPSolver::OCPSolver(const RModel& model, const RCost& cost_, const idocp_constraints_t& constraints, real T, int N,
                     int max_num_impulse)
    : robot(model), task_robot(makeTaskRobot(model, cost_)), cost(cost_), cons(constraints), N_ideal_(N), N_(N), nv_(model.nv), nu_(model.nu), nc_(model.ncontacts),
      max_events_(max_num_impulse), kP(model.nv - model.nu), T_(T), dt_(T / N) {
  if (T <= 0) throw std::out_of_range("invalid value: T must be positive!");
  if (N <= 0) throw std::out_of_range("invalid value: N must be positive!");
  if (max_num_impulse < 0) throw std::out_of_range("invalid value: max_num_impulse must be non-negative!");
  const int ns = nslots();
  s.assign(ns, SplitSolutionC(robot));
  d.assign(ns, SplitDirectionC(robot));
  kkt_matrix.assign(ns, SplitKKTMatrixC(model.nv, model.nu));
  kkt_residual.assign(ns, SplitKKTResidualC(model.nv, model.nu));
  cd.resize(ns); sw.resize(ns); ipm.resize(ns); cd_J.resize(ns);
  riccati.assign(ns, RiccatiC(model.nv));
  K.assign(ns, Mat(model.nu, 2 * model.nv)); k.assign(ns, Mat(model.nu));
  ContactStatus cs0;
  cs0.active.assign(nc_, false);
  cs0.points.assign(nc_, Mat(3));
  seq.phases.assign(1, cs0);                       // ContactSequence ctor: default (no contact) status
  discretize(0.0);
}

int OCPSolver::slotOf(int kind, int index) const {
  switch (kind) {
    case NodeC::Impulse: return N_ideal_ + 1 + index;
    case NodeC::Aux: return N_ideal_ + 1 + max_events_ + index;
    case NodeC::Lift: return N_ideal_ + 1 + 2 * max_events_ + index;
    default: return index;
  }
}

// ContactSequence::setContactStatusUniformly (contact_sequence.hxx:47-51)
void OCPSolver::setContactStatusUniformly(const std::vector<int>& active, const double* pts) {
  ContactStatus cs;
  cs.active.assign(nc_, false); cs.points.assign(nc_, Mat(3));
  for (int c = 0; c < nc_; ++c) {
    cs.active[c] = active[c] != 0;
    for (int k2 = 0; k2 < 3; ++k2) cs.points[c][k2] = pts[3 * c + k2];
  }
  seq.phases.assign(1, cs);
  seq.event_time.clear(); seq.is_impulse.clear(); seq.impulse_status.clear();
  discretized_ = false;
}

// ContactSequence::push_back (contact_sequence.hxx:52-104) + DiscreteEvent::setDiscreteEvent (discrete_event.hxx:57-84)
void OCPSolver::pushBackContactStatus(const std::vector<int>& active, const double* pts, real switching_time) {
  // the sequence itself holds up to N events (ocp_solver.cpp:16: contact_sequence_(robot, N)); the event stages live in
  // containers of max_num_impulse impulse / aux / lift entries each (hybrid_container.hpp:39-96), checked below
  if (seq.numEvents() + 1 > N_ideal_) throw std::runtime_error("Number of discrete events exceeds predefined max_num_events!");
  if (seq.numEvents() > 0 && switching_time <= seq.event_time.back()) throw std::runtime_error("event_time must be larger than the last event time!");
  const ContactStatus& pre = seq.phases.back();
  ContactStatus post, imp;
  post.active.assign(nc_, false); post.points.assign(nc_, Mat(3));
  imp.active.assign(nc_, false); imp.points.assign(nc_, Mat(3));
  bool exist_impulse = false, exist_lift = false;
  for (int c = 0; c < nc_; ++c) {
    post.active[c] = active[c] != 0;
    for (int k2 = 0; k2 < 3; ++k2) { post.points[c][k2] = pts[3 * c + k2]; imp.points[c][k2] = pts[3 * c + k2]; }
    if (pre.active[c]) { if (!post.active[c]) exist_lift = true; }
    else if (post.active[c]) { imp.active[c] = true; exist_impulse = true; }
  }
  if (!exist_impulse && !exist_lift) throw std::runtime_error("discrete_event.existDiscreteEvent() must be true!");
  if ((exist_impulse ? seq.numImpulse() : seq.numLift()) + 1 > max_events_) throw std::runtime_error("more impulse / lift events than max_num_impulse!");
  seq.phases.push_back(post);
  seq.event_time.push_back(switching_time);
  seq.is_impulse.push_back(exist_impulse);
  seq.impulse_status.push_back(imp);
  discretized_ = false;
}

// ContactSequence::pop_back / pop_front (contact_sequence.hxx:117-160) behind OCPSolver::popBackContactStatus /
// popFrontContactStatus (ocp_solver.cpp:187-194): the last (first) discrete event leaves together with the contact phase
// behind (in front of) it; a sequence without events falls back to the default status (no contact, points at the origin).
// The stage records do NOT move: impulse / aux / lift index k after a pop_front holds what index k held before
// (hybrid_container.hpp: the solver never shifts s.impulse / s.aux / s.lift).
static void seqPopBack(ContactSequenceC& seq, int nc) {
  if (seq.numEvents() > 0) {
    seq.event_time.pop_back(); seq.is_impulse.pop_back(); seq.impulse_status.pop_back(); seq.phases.pop_back();
  } else {
    ContactStatus cs0; cs0.active.assign(nc, false); cs0.points.assign(nc, Mat(3));
    seq.phases.assign(1, cs0);
  }
}
static void seqPopFront(ContactSequenceC& seq, int nc) {
  if (seq.numEvents() > 0) {
    seq.event_time.erase(seq.event_time.begin()); seq.is_impulse.erase(seq.is_impulse.begin());
    seq.impulse_status.erase(seq.impulse_status.begin()); seq.phases.erase(seq.phases.begin());
  } else {
    ContactStatus cs0; cs0.active.assign(nc, false); cs0.points.assign(nc, Mat(3));
    seq.phases.assign(1, cs0);
  }
}
void OCPSolver::popBackContactStatus() { seqPopBack(seq, nc_); discretized_ = false; }
void OCPSolver::popFrontContactStatus() { seqPopFront(seq, nc_); discretized_ = false; }

// ContactSequence::setContactPoints (contact_sequence.hxx:252-268)
void OCPSolver::setContactPoints(int phase, const double* pts) {
  if (phase >= (int)seq.phases.size()) throw std::runtime_error("contact_phase must be smaller than numContactPhases()!");
  for (int c = 0; c < nc_; ++c) for (int k2 = 0; k2 < 3; ++k2) {
    seq.phases[phase].points[c][k2] = pts[3 * c + k2];
    if (phase > 0 && seq.is_impulse[phase - 1]) seq.impulse_status[phase - 1].points[c][k2] = pts[3 * c + k2];
  }
}

// OCPSolver::setSolution (ocp_solver.cpp:95-165): every stage, including the event stages ("a" sets dv on impulse stages)
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

// ------------------------------------------------------------ discretiser ----
// OCPDiscretizer::discretizeOCP (ocp_discretizer.hxx:65-374), transcribed step by step.
void OCPSolver::discretize(real t) {
  const real min_dt = std::sqrt(std::numeric_limits<real>::epsilon());    // ocp_discretizer.hpp:108-109
  const real dt_ideal = T_ / N_ideal_, max_dt = dt_ideal - min_dt;
  const int Ni = seq.numImpulse(), Nl = seq.numLift();
  std::vector<int> tsbi(Ni + 1, -1), tsbl(Nl + 1, -1);
  std::vector<real> t_imp(Ni + 1, 0.0), t_lift(Nl + 1, 0.0), dt_aux(Ni + 1, 0.0), dt_lift(Nl + 1, 0.0);
  // countDiscreteEvents (:271-288)
  for (int kx = 0; kx < Ni; ++kx) { t_imp[kx] = seq.event_time[seq.eventOfImpulse(kx)]; tsbi[kx] = (int)std::floor((t_imp[kx] - t) / dt_ideal); }
  for (int kx = 0; kx < Nl; ++kx) { t_lift[kx] = seq.event_time[seq.eventOfLift(kx)]; tsbl[kx] = (int)std::floor((t_lift[kx] - t) / dt_ideal); }
  // countTimeSteps (:291-345)
  std::vector<real> dts(N_ideal_ + 1, dt_ideal), ts(N_ideal_ + 1, 0.0);
  int ii = 0, li = 0, on_grid = 0;
  for (int i = 0; i < N_ideal_; ++i) {
    const int stage = i - on_grid;
    if (ii < Ni && i == tsbi[ii]) {
      dts[stage] = t_imp[ii] - i * dt_ideal - t;
      if (dts[stage] <= min_dt) { tsbi[ii] = stage - 1; dt_aux[ii] = dt_ideal; ts[stage] = t + (i - 1) * dt_ideal; ++on_grid; ++ii; }
      else if (dts[stage] >= max_dt) { tsbi[ii] = i + 1; ts[stage] = t + i * dt_ideal; }
      else { tsbi[ii] = stage; dt_aux[ii] = dt_ideal - dts[stage]; ts[stage] = t + i * dt_ideal; ++ii; }
    } else if (li < Nl && i == tsbl[li]) {
      dts[stage] = t_lift[li] - i * dt_ideal - t;
      if (dts[stage] <= min_dt) { tsbl[li] = stage - 1; dt_lift[li] = dt_ideal; ts[stage] = t + (i - 1) * dt_ideal; ++on_grid; ++li; }
      else if (dts[stage] >= max_dt) { tsbl[li] = i + 1; ts[stage] = t + i * dt_ideal; }
      else { tsbl[li] = stage; dt_lift[li] = dt_ideal - dts[stage]; ts[stage] = t + i * dt_ideal; ++li; }
    } else {
      dts[stage] = dt_ideal; ts[stage] = t + i * dt_ideal;
    }
  }
  N_ = N_ideal_ - on_grid;
  ts[N_] = t + T_;
  // countTimeStages (:348-391) and countContactPhase (:394-404)
  std::vector<int> imp_after(N_ + 1, -1), lift_after(N_ + 1, -1), phase(N_ + 1, 0);
  ii = 0; li = 0;
  int num_events = 0;
  for (int i = 0; i < N_; ++i) {
    if (ii < Ni && i == tsbi[ii]) imp_after[i] = ii++;
    if (li < Nl && i == tsbl[li]) lift_after[i] = li++;
    phase[i] = num_events;
    if (imp_after[i] >= 0 || lift_after[i] >= 0) ++num_events;
  }
  phase[N_] = num_events;
  for (int i = 0; i < N_; ++i) if (imp_after[i] >= 0 && lift_after[i] >= 0) throw std::runtime_error("OCPDiscretizer: not well defined");
  // the chain
  chain.clear();
  auto node = [&](int kind, int index, real tt, real dtt, int ph, int level) {
    NodeC nd; nd.kind = kind; nd.index = index; nd.slot = slotOf(kind, index); nd.t = tt; nd.dt = dtt; nd.phase = ph; nd.level = level;
    chain.push_back(nd);
  };
  for (int i = 0; i < N_; ++i) {
    node(NodeC::Stage, i, ts[i], dts[i], phase[i], i);
    // switching constraint two steps ahead of an impulse (ocp_linearizer.hxx:152-163)
    if (imp_after[i] < 0 && lift_after[i] < 0 && i + 1 < N_ && imp_after[i + 1] >= 0) {
      chain.back().sw_event = seq.eventOfImpulse(imp_after[i + 1]);
      chain.back().sw_dt_next = dts[i + 1];
    }
    if (imp_after[i] >= 0) {
      const int kx = imp_after[i];
      node(NodeC::Impulse, kx, t_imp[kx], 0.0, seq.eventOfImpulse(kx), -1);
      node(NodeC::Aux, kx, t_imp[kx], dt_aux[kx], phase[i + 1], 0);
    } else if (lift_after[i] >= 0) {
      const int kx = lift_after[i];
      node(NodeC::Lift, kx, t_lift[kx], dt_lift[kx], phase[i + 1], 0);
      if (i + 1 < N_ && imp_after[i + 1] >= 0) {            // (:205-217)
        chain.back().sw_event = seq.eventOfImpulse(imp_after[i + 1]);
        chain.back().sw_dt_next = dts[i + 1];
      }
    }
  }
  node(NodeC::Terminal, N_, ts[N_], 0.0, phase[N_], N_);
  discretized_ = true;
}

const ContactStatus& OCPSolver::nodeContacts(int p) const {
  const NodeC& nd = chain[p];
  return nd.kind == NodeC::Impulse ? seq.impulse_status[nd.phase] : seq.phases[nd.phase];
}

// ------------------------------------------------------------ constraints ----
bool OCPSolver::componentEnabled(int c, bool impulse) const {
  if (impulse) return c == 6 && (cons.linearized_impulse_friction_cone != 0 || cons.impulse_friction_cone != 0);
  if (c < 2) return cons.joint_position_limits != 0;
  if (c < 4) return cons.joint_velocity_limits != 0;
  if (c < 6) return cons.joint_torque_limits != 0;
  if (c == 7) return false;
  if (c == 8) return cons.joint_acceleration_lower_limit != 0;       // acceleration level: every time stage (constraints_data.hpp:18-42)
  if (c == 9) return cons.joint_acceleration_upper_limit != 0;
  if (c == 10) return cons.contact_distance != 0;
  return cons.linearized_friction_cone != 0 || cons.friction_cone != 0;
}
bool OCPSolver::componentValid(int c, const NodeC& nd) const {     // constraints_data.hpp:18-42
  if (nd.kind == NodeC::Impulse) return componentEnabled(c, true);
  if (!componentEnabled(c, false)) return false;
  if (c < 2 || c == 10) return nd.level >= 2;                       // position level
  if (c < 4) return nd.level >= 1;
  return true;
}
int OCPSolver::dimc() const { int n = 0; for (int c = 0; c < NCOMP; ++c) if (componentEnabled(c, false)) n += componentDim(c); return n; }

static real limitOf(const RModel& m, const idocp_constraints_t& k, int c, int k2) {
  switch (c) {
    case 8: return k.a_min[k2];       // JointAccelerationLowerLimit / UpperLimit carry their own bounds (joint_acceleration_*_limit.cpp:6-13)
    case 9: return k.a_max[k2];
    case 0: return m.q_min[k2];
    case 1: return m.q_max[k2];
    case 2: return -m.v_max[k2];
    case 3: return m.v_max[k2];
    case 4: return -m.u_max[k2];
    default: return m.u_max[k2];
  }
}
// value of the limited variable (joint part)
static real limitedVar(const SplitSolutionC& s, int c, int k2, int nv, int nu) {
  if (c < 2) return s.q[s.q.size() - nu + k2];
  if (c < 4) return s.v[nv - nu + k2];
  if (c >= 8) return s.a[nv - nu + k2];
  return s.u[k2];
}
// The friction-cone component of one contact with force f: values g_r(f) <= 0 and gradients J_r(f) of its rows.
//   kind 0  LinearizedFrictionCone (5 rows): g = Jc f (linearized_friction_cone.hpp:72-84, linearized_friction_cone.cpp:25-29)
//   kind 1  FrictionCone (2 rows): g0 = -fz, g1 = fx^2 + fy^2 - mu^2 fz^2 (friction_cone.hpp:70-80); J1 is the reference's data.r[i] =
//           (2 fx, 2 fy, -2 mu^2 fz) (friction_cone.cpp:100-118); the Hessian contribution is the Gauss-Newton term J^T diag(dual / slack) J
//           in both cases (friction_cone.cpp:121-146), i.e. every formula below is the same for the two kinds
// (Linearized)ImpulseFrictionCone: the same rows on an impulse stage, with dt = 1.
ConeEval coneEval(int kind, real mu, const Mat& f) {
  ConeEval e;
  if (kind == 1) {
    e.nr = 2;
    e.res[0] = -f[2]; e.J[0][0] = 0; e.J[0][1] = 0; e.J[0][2] = -1;
    e.res[1] = f[0] * f[0] + f[1] * f[1] - mu * mu * f[2] * f[2];
    e.J[1][0] = 2 * f[0]; e.J[1][1] = 2 * f[1]; e.J[1][2] = -2 * mu * mu * f[2];
    return e;
  }
  e.nr = 5;
  const real m2 = mu / std::sqrt(2.0);
  const real Jc[5][3] = {{0, 0, -1}, {1, 0, -m2}, {-1, 0, -m2}, {0, 1, -m2}, {0, -1, -m2}};
  for (int r = 0; r < 5; ++r) { e.res[r] = 0; for (int c = 0; c < 3; ++c) { e.J[r][c] = Jc[r][c]; e.res[r] += Jc[r][c] * f[c]; } }
  return e;
}


// SplitOCP::initConstraints / ImpulseSplitOCP::initConstraints (split_ocp.hxx:50-55, impulse_split_ocp.hxx:33-37)
void OCPSolver::initNodeConstraints(const NodeC& nd) {
  const SplitSolutionC& sp = s[nd.slot];
  ipm[nd.slot].clear();
  for (int c = 0; c < NCOMP; ++c) {
    IpmData data(componentDim(c, nd.kind == NodeC::Impulse));
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    if (componentValid(c, nd)) {
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r = 0; r < nu_; ++r) data.slack[r] = -sgn * (limitedVar(sp, c, r, nv_, nu_) - limitOf(robot.model(), cons, c, r));
      } else if (c == 10) {
        // ContactDistance::setSlackAndDual (contact_distance.cpp:58-65): the height of every contact frame, active or not
        {
          const Mat zero(nv_);
          robot.updateKinematics(sp.q, zero, zero);
          for (int cc = 0; cc < nc_; ++cc) { real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr); data.slack[cc] = pw[2]; }
        }
      } else {
        for (int cc = 0; cc < nc_; ++cc) {      // all contacts, active or not (linearized_friction_cone.cpp:96-104)
          const ConeEval ce = coneEval(CK, cons.mu, sp.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
          for (int r = 0; r < CR; ++r) data.slack[CR * cc + r] = -res[r];
        }
      }
      for (int r = 0; r < data.slack.size(); ++r) {
        while (data.slack[r] < cons.barrier) data.slack[r] += cons.barrier;
        data.dual[r] = cons.barrier / data.slack[r];
      }
    }
    ipm[nd.slot].push_back(data);
  }
}

// OCPSolver::initConstraints (ocp_solver.cpp:60-64) -> OCPLinearizer::initConstraints (ocp_linearizer.cpp:40-70)
void OCPSolver::initConstraints(real t) {
  discretize(t);
  // every slot gets constraint data (the reference initialises all N_ideal stages and every event stage in use)
  for (int i = 0; i <= N_ideal_; ++i) { NodeC nd; nd.kind = i < N_ideal_ ? NodeC::Stage : NodeC::Terminal; nd.slot = i; nd.index = i; nd.level = i; initNodeConstraints(nd); }
  for (const NodeC& nd : chain) if (nd.kind == NodeC::Impulse || nd.kind == NodeC::Aux || nd.kind == NodeC::Lift) initNodeConstraints(nd);
}

// ------------------------------------------------------------------ cost ----
// TimeVaryingConfigurationSpaceCost::v_ref(t) (include/idocp/cost/time_varying_configuration_space_cost.hpp:111-118)
static real vRefScale(const RCost& cost, real t) {
  if (!cost.use_time_varying_ref) return 1.0;
  return (t > cost.tv_t_begin && t < cost.tv_t_end) ? 1.0 : 0.0;
}

void OCPSolver::qRef(real t, Mat& q_ref) const {
  q_ref = Mat(robot.dimq());
  for (int i = 0; i < robot.dimq(); ++i) q_ref[i] = cost.q_ref[i];
  if (cost.use_time_varying_ref) {      // set_q_ref (time_varying_configuration_space_cost.hpp:98-109)
    const real tau = t <= cost.tv_t_begin ? real(0.0) : ((t < cost.tv_t_end ? t : cost.tv_t_end) - cost.tv_t_begin);
    if (tau > 0.0) {
      Mat qb = q_ref, v(robot.dimv());
      for (int i = 0; i < robot.dimv(); ++i) v[i] = cost.v_ref[i];
      robot.integrateConfiguration(qb, v, tau, q_ref);
    }
    return;
  }
  if (!cost.use_trotting_ref || !(t > cost.t_start)) return;
  const real tau = t - cost.t_start;
  const int steps = (int)std::floor(tau / cost.t_period);
  const real rate = (tau - steps * cost.t_period) / cost.t_period;
  const real sin2 = std::sin(M_PI_2 * rate);
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
// SplitOCP::linearizeOCP (split_ocp.hxx:58-134) / computeKKTResidual (:189-248) for Stage / Aux / Lift nodes, and
// ImpulseSplitOCP::linearizeOCP / computeKKTResidual (impulse_split_ocp.hxx:40-66, 107-124) for Impulse nodes.
// The two are the same computation with (dt_dyn, dt_q) = (dt, dt) vs (1, 0), "a" playing the role of dv, no torque
// variables on the impulse stage, and the contact VELOCITY constraint instead of the Baumgarte constraint.
void OCPSolver::linearizeNode(Robot& robot, int p, const Mat& q_prev, bool residual_only) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const NodeC& nd = chain[p];
  const bool impulse = nd.kind == NodeC::Impulse;
  const SplitSolutionC& si = s[nd.slot];
  const SplitSolutionC& sn = s[chain[p + 1].slot];
  SplitKKTMatrixC& M = kkt_matrix[nd.slot];
  SplitKKTResidualC& R = kkt_residual[nd.slot];
  ContactDynamicsDataC& D = cd[nd.slot];
  const ContactStatus& cs = nodeContacts(p);
  const int nv = nv_, nu = nu_, dimf = cs.dimf();
  const real dt = impulse ? real(1.0) : nd.dt;          // scaling of cost / constraints / dynamics multipliers
  const real dtq = impulse ? real(0.0) : nd.dt;         // q+ = q (+) dtq v
  const real t = nd.t;
  if (impulse) robot.updateKinematics(si.q, si.v + si.a, Mat(nv));     // impulse_split_ocp.hxx:47
  else robot.updateKinematics(si.q, si.v, si.a);
  if (!residual_only) {
    M.Qxx.setZero(); M.Qxu_full.setZero(); M.Quu_full.setZero(); M.Qaa_diag.setZero(); M.Qff = Mat(dimf, dimf);
    M.Fvq.setZero(); M.Fvv.setZero(); M.Fvu.setZero();
  }
  R.Fq.setZero(); R.Fv.setZero(); R.lq.setZero(); R.lv.setZero(); R.la.setZero(); R.lf = Mat(dimf); R.lu.setZero(); R.lu_passive.setZero();
  R.P = Mat(0);
  const real* wq = impulse ? cost.qi_weight : cost.q_weight;
  const real* wv = impulse ? cost.vi_weight : cost.v_weight;
  const real* wa = impulse ? cost.dvi_weight : cost.a_weight;
  const real (*wf)[3] = impulse ? cost.fi_weight : cost.f_weight;
  const real (*rf)[3] = impulse ? cost.fi_ref : cost.f_ref;
  // ---- cost: (Trotting)ConfigurationSpaceCost + ContactForceCost
  // (configuration_space_cost.cpp:292-310, trotting_configuration_space_cost.cpp:269-327, contact_force_cost.cpp:153-179)
  Mat q_ref, qdiff, Jq;
  qRef(t, q_ref);
  robot.subtractConfiguration(si.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(si.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = wq[r] * qdiff[r];
  R.lq += dt * (Jq.t() * Wq);
  const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  const real vs = vRefScale(cost, t);      // TimeVaryingConfigurationSpaceCost::v_ref(t)
  for (int r = 0; r < nv; ++r) {
    R.lv[r] += dt * wv[r] * (si.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]));
    R.la[r] += dt * wa[r] * si.a[r];
  }
  if (!impulse) for (int r = 0; r < nu; ++r) R.lu[r] += dt * cost.u_weight[r] * (si.u[r] - cost.u_ref[r]);
  Mat task_H;
  if (cost.task_dim) { real c_; Mat g_; taskTerms(task_robot, cost, t, impulse ? cost.task_weighti : cost.task_weight, si.q, c_, g_, task_H); R.lq += dt * g_; }      // (dt = 1 on impulse stages)
  {
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
      for (int r = 0; r < 3; ++r) R.lf[st + r] += dt * wf[c][r] * (si.f[c][r] - rf[c][r]);
      st += 3;
    }
  }
  // ---- constraints: [computePrimalAndDualResidual] + augmentDualResidual
  for (int c = 0; c < NCOMP; ++c) {
    if (!componentValid(c, nd)) continue;
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    IpmData& data = ipm[nd.slot][c];
    if (jointComp(c)) {
      const real sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : (c < 6 ? R.lu : R.la));
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        if (residual_only) {
          data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), cons, c, r)) + data.slack[r];
          data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        }
        l[off + r] += sgn * dt * data.dual[r];
      }
    } else if (c == 10) {
      // ContactDistance::augmentDualResidual (contact_distance.cpp:68-78) [+ computePrimalAndDualResidual :132-146]: the contacts that are
      // NOT active keep their frame above the ground; J = LOCAL frame Jacobian (Robot::getFrameJacobian), its row 2
      if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
      robot.updateKinematics(si.q, si.v, si.a);
      cd_J[nd.slot] = Mat(nc_, nv);
      for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
        Mat vdq, adq, adv, J;
        robot.frameDerivatives(cc, vdq, adq, adv, J);
        for (int col = 0; col < nv; ++col) { cd_J[nd.slot](cc, col) = J(2, col); R.lq[col] -= dt * data.dual[cc] * J(2, col); }
        if (residual_only) {
          real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr);
          data.residual[cc] = -pw[2] + data.slack[cc];
          data.duality[cc] = data.slack[cc] * data.dual[cc] - cons.barrier;
        }
      }
    } else {
      if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J;
        if (residual_only) {
          for (int r = 0; r < CR; ++r) {
            data.residual[CR * cc + r] = res[r] + data.slack[CR * cc + r];
            data.duality[CR * cc + r] = data.slack[CR * cc + r] * data.dual[CR * cc + r] - cons.barrier;
          }
        }
        for (int x = 0; x < 3; ++x) for (int r = 0; r < CR; ++r) R.lf[st + x] += dt * Jc[r][x] * data.dual[CR * cc + r];
        st += 3;
      }
    }
  }
  // ---- state equation: linearizeForwardEuler (state_equation.hxx:12-37, 210-221) /
  //      linearizeImpulseForwardEuler (impulse_state_equation.hxx:10-34, 119-128)
  Mat diff; robot.subtractConfiguration(si.q, sn.q, diff);
  for (int r = 0; r < nv; ++r) { R.Fq[r] = diff[r] + dtq * si.v[r]; R.Fv[r] = si.v[r] + dt * si.a[r] - sn.v[r]; }
  Mat Fqq, Fqq_prev;
  robot.dSubtractdConfigurationPlus(si.q, sn.q, Fqq);
  robot.dSubtractdConfigurationMinus(q_prev, si.q, Fqq_prev);
  M.Fqq6 = Fqq.block(0, 0, kP, kP);
  M.Fqq_prev6 = Fqq_prev.block(0, 0, kP, kP);
  {
    Mat t1 = M.Fqq6.t() * sn.lmd.segment(0, kP) + M.Fqq_prev6.t() * si.lmd.segment(0, kP);
    for (int r = 0; r < kP; ++r) R.lq[r] += t1[r];
    for (int r = kP; r < nv; ++r) R.lq[r] += sn.lmd[r] - si.lmd[r];
    for (int r = 0; r < nv; ++r) { R.lv[r] += dtq * sn.lmd[r] + sn.gmm[r] - si.gmm[r]; R.la[r] += dt * sn.gmm[r]; }
  }
  if (!residual_only) {
    // condenseForwardEuler (state_equation.hxx:40-63) / condenseImpulseForwardEuler (impulse_state_equation.hxx:36-57)
    Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_prev_inv);
    Mat Fm; robot.dSubtractdConfigurationMinus(si.q, sn.q, Fm);
    M.Fqq_prev6 = Fm.block(0, 0, kP, kP);
    Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_inv);
    M.Fqq_prev6 = M.Fqq6;
    R.Fq_prev = R.Fq.segment(0, kP);
    M.Fqq6 = -1.0 * (M.Fqq_inv * M.Fqq_prev6);
    M.Fqv6 = (-dtq) * M.Fqq_inv;
    R.Fq.setSegment(0, -1.0 * (M.Fqq_inv * R.Fq_prev));
  }
  // ---- ContactDynamics::linearizeContactDynamics (contact_dynamics.hxx:48-102) /
  //      ImpulseDynamicsForwardEuler::linearizeImpulseDynamics (impulse_dynamics_forward_euler.hxx:18-58)
  robot.setContactForces(cs.active, si.f);
  Mat ID_full, dIDdq, dIDdv, C, dCdq, dCdv;
  if (impulse) {
    const Mat zero(nv);
    robot.RNEA(si.q, zero, si.a, ID_full, false);                       // RNEAImpulse (robot.hxx:505-517)
    robot.RNEADerivatives(si.q, zero, si.a, dIDdq, dIDdv, D.dIDda, false);
    dIDdv = Mat(nv, nv);                                               // dImD/dv = 0 (never formed by the reference)
    robot.computeImpulseVelocityResidual(cs.active, C);
    robot.computeImpulseVelocityDerivatives(cs.active, dCdq, dCdv);
    D.dCda = dCdv;                                                     // dC/ddv = dC/dv (:36-41)
  } else {
    robot.RNEA(si.q, si.v, si.a, ID_full);
    for (int r = 0; r < nu; ++r) ID_full[kP + r] -= si.u[r];
    robot.RNEADerivatives(si.q, si.v, si.a, dIDdq, dIDdv, D.dIDda);
    robot.computeBaumgarteResidual(cs.active, dt_, cs.points, C);      // baumgarte_time_step = T/N (hybrid_container.hpp:186-188)
    robot.computeBaumgarteDerivatives(cs.active, dt_, dCdq, dCdv, D.dCda);
  }
  D.IDC = Mat(nv + dimf); D.IDC.setSegment(0, ID_full); D.IDC.setSegment(nv, C);
  D.dIDCdqv = Mat(nv + dimf, 2 * nv);
  D.dIDCdqv.setBlock(0, 0, dIDdq); D.dIDCdqv.setBlock(0, nv, dIDdv);
  D.dIDCdqv.setBlock(nv, 0, dCdq); D.dIDCdqv.setBlock(nv, nv, dCdv);
  R.lq += dt * (dIDdq.t() * si.beta);
  R.lv += dt * (dIDdv.t() * si.beta);
  R.la += dt * (D.dIDda.t() * si.beta);
  const Mat mu_stack = si.mu_stack(cs);
  if (dimf > 0) R.lf -= dt * (D.dCda * si.beta);
  if (!impulse) {
    for (int r = 0; r < kP; ++r) R.lu_passive[r] = dt * si.nu_passive[r] - dt * si.beta[r];
    for (int r = 0; r < nu; ++r) R.lu[r] -= dt * si.beta[kP + r];
  }
  if (dimf > 0) {
    R.lq += dt * (dCdq.t() * mu_stack);
    R.lv += dt * (dCdv.t() * mu_stack);
    R.la += dt * (D.dCda.t() * mu_stack);
  }
  // ---- ForwardSwitchingConstraint::linearizeSwitchingConstraint (forward_switching_constraint.hxx:27-66)
  SwitchingC& W = sw[nd.slot];
  FLOP_REGION_SET(R_SWITCH);
  if (nd.sw_event >= 0) {
    const ContactStatus& is = seq.impulse_status[nd.sw_event];
    const int dimi = is.dimf();
    const real dt1 = nd.dt, dt2 = nd.sw_dt_next;
    Mat dq_ = (dt1 + dt2) * si.v + (dt1 * dt2) * si.a, q_;
    robot.integrateConfiguration(si.q, dq_, 1.0, q_);
    robot.updateKinematics(q_, Mat(nv), Mat(nv));
    robot.computeContactResidual(is.active, is.points, R.P);
    Mat Pq, dint_dq, dint_dv;
    robot.computeContactDerivative(is.active, Pq);
    robot.dIntegratedConfiguration(si.q, dq_, dint_dq);
    robot.dIntegratedVelocity(si.q, dq_, dint_dv);
    Mat Phiq = Pq * dint_dq, PqJ = Pq * dint_dv;
    W.Phix = Mat(dimi, 2 * nv);
    W.Phix.setBlock(0, 0, Phiq); W.Phix.setBlock(0, nv, (dt1 + dt2) * PqJ);
    W.Phia = (dt1 * dt2) * PqJ;
    const Mat xi = si.xi.segment(0, dimi);
    R.lq += Phiq.t() * xi;
    R.lv += W.Phix.block(0, nv, dimi, nv).t() * xi;
    R.la += W.Phia.t() * xi;
  } else {
    W = SwitchingC();
  }
  FLOP_REGION_SET(R_COST_CONSTRAINTS);
  if (residual_only) return;
  // ---- cost Hessian (configuration_space_cost.cpp:351-365; contact_force_cost.cpp:182-211)
  {
    Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= wq[r];
    M.Qxx.addBlock(0, 0, Jq.t() * WJ, dt);
    for (int r = 0; r < nv; ++r) { M.Qxx(nv + r, nv + r) += dt * wv[r]; M.Qaa_diag[r] += dt * wa[r]; }
    if (!impulse) for (int r = 0; r < nu; ++r) M.Quu_full(kP + r, kP + r) += dt * cost.u_weight[r];
    if (cost.task_dim) M.Qxx.addBlock(0, 0, task_H, dt);
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) { for (int r = 0; r < 3; ++r) M.Qff(st + r, st + r) += dt * wf[c][r]; st += 3; }
  }
  // ---- Constraints::condenseSlackAndDual
  for (int c = 0; c < NCOMP; ++c) {
    if (!componentValid(c, nd)) continue;
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    IpmData& data = ipm[nd.slot][c];
    if (jointComp(c)) {
      const real sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : (c < 6 ? R.lu : R.la));
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        const real h = dt * data.dual[r] / data.slack[r];
        if (c < 2) M.Qxx(kP + r, kP + r) += h;
        else if (c < 4) M.Qxx(nv + kP + r, nv + kP + r) += h;
        else if (c < 6) M.Quu_full(kP + r, kP + r) += h;
        else M.Qaa_diag[kP + r] += h;
        data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), cons, c, r)) + data.slack[r];
        data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        l[off + r] += sgn * dt * (data.dual[r] * data.residual[r] - data.duality[r]) / data.slack[r];
      }
    } else if (c == 10) {
      // ContactDistance::condenseSlackAndDual (contact_distance.cpp:81-102); the kinematics of s.q (the reference condenses the
      // constraints before it moves the robot to the predicted configuration of the switching constraint, split_ocp.hxx:124-129)
      data.residual.setZero(); data.duality.setZero();
      robot.updateKinematics(si.q, si.v, si.a);
      for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
        const Mat& Jm = cd_J[nd.slot];
        const real h = dt * data.dual[cc] / data.slack[cc];
        for (int r2 = 0; r2 < nv; ++r2) for (int c2 = 0; c2 < nv; ++c2) M.Qxx(r2, c2) += h * Jm(cc, r2) * Jm(cc, c2);
        real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr);
        data.residual[cc] = -pw[2] + data.slack[cc];
        data.duality[cc] = data.slack[cc] * data.dual[cc] - cons.barrier;
        const real g = dt * (data.dual[cc] * data.residual[cc] - data.duality[cc]) / data.slack[cc];
        for (int col = 0; col < nv; ++col) R.lq[col] -= g * Jm(cc, col);
      }
    } else {
      // linearized_friction_cone.cpp:125-152, 184-202 ; linearized_impulse_friction_cone.cpp (same with dt = 1)
      data.residual.setZero(); data.duality.setZero();
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
        real rr[5], dd[5];
        for (int r = 0; r < CR; ++r) {
          const int idx = CR * cc + r;
          data.residual[idx] = res[r] + data.slack[idx];
          data.duality[idx] = data.slack[idx] * data.dual[idx] - cons.barrier;
          rr[r] = (data.dual[idx] * data.residual[idx] - data.duality[idx]) / data.slack[idx];
          dd[r] = data.dual[idx] / data.slack[idx];
        }
        for (int x = 0; x < 3; ++x) {
          for (int r = 0; r < CR; ++r) R.lf[st + x] += dt * Jc[r][x] * rr[r];
          for (int y = 0; y < 3; ++y) { real acc = 0; for (int r = 0; r < CR; ++r) acc += Jc[r][x] * dd[r] * Jc[r][y]; M.Qff(st + x, st + y) += dt * acc; }
        }
        st += 3;
      }
    }
  }
  // ---- ContactDynamics::condenseContactDynamics (contact_dynamics.hxx:105-158) /
  //      ImpulseDynamicsForwardEuler::condenseImpulseDynamics (impulse_dynamics_forward_euler.hxx:59-105)
  if (keep_uncondensed) {      // test hook (ocp.hpp UncondensedC): everything the un-condensed Newton system of this stage consists of
    UncondensedC& U = unc[nd.slot];
    U.valid = true; U.kind = (int)nd.kind; U.dimf = dimf; U.dimi = (nd.sw_event >= 0) ? (int)R.P.size() : 0; U.has_u = impulse ? 0 : 1;
    U.dt = dt; U.dtq = dtq;
    U.active_mask = 0; for (int c = 0; c < nc_; ++c) if (cs.active[c]) U.active_mask |= 1 << c;
    U.Qxx = M.Qxx; U.Qaa = M.Qaa_diag; U.Qff = M.Qff; U.Quu = M.Quu_full;
    U.lq = R.lq; U.lv = R.lv; U.la = R.la; U.lf = R.lf; U.lu = R.lu; U.lu_passive = R.lu_passive;
    U.Fq = R.Fq; U.Fq.setSegment(0, R.Fq_prev);                       // the residual before condenseForwardEuler premultiplied its base rows
    U.Fv = R.Fv;
    U.Fqq = M.Fqq_prev6;                                              // dSubtractdConfigurationPlus(q, q_next) (parked there by condenseForwardEuler)
    U.Fqq_prev = Fqq_prev.block(0, 0, kP, kP);                          // dSubtractdConfigurationMinus(q_prev, q)
    U.dIDCdqv = D.dIDCdqv; U.M = D.dIDda; U.J = D.dCda; U.IDC = D.IDC;
    U.Phix = W.Phix; U.Phia = W.Phia; U.P = R.P;
  }
  FLOP_REGION_SET(R_CONDENSE);
  Robot::computeMJtJinv(D.dIDda, D.dCda, D.MJtJinv);
  D.MJtJinv_dIDCdqv = D.MJtJinv * D.dIDCdqv;
  D.MJtJinv_IDC = D.MJtJinv * D.IDC;
  D.Qafqv = Mat(nv + dimf, 2 * nv);
  D.Qafu_full = Mat(nv + dimf, nv);
  for (int c = 0; c < 2 * nv; ++c) for (int r = 0; r < nv; ++r) D.Qafqv(r, c) = -M.Qaa_diag[r] * D.MJtJinv_dIDCdqv(r, c);
  if (!impulse) for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) D.Qafu_full(r, c) = M.Qaa_diag[r] * D.MJtJinv(r, c);
  if (dimf > 0) {
    D.Qafqv.setBlock(nv, 0, -1.0 * (M.Qff * D.MJtJinv_dIDCdqv.block(nv, 0, dimf, 2 * nv)));
    if (!impulse) D.Qafu_full.setBlock(nv, 0, M.Qff * D.MJtJinv.block(nv, 0, dimf, nv));
  }
  D.laf = Mat(nv + dimf);
  for (int r = 0; r < nv; ++r) D.laf[r] = R.la[r] - M.Qaa_diag[r] * D.MJtJinv_IDC[r];
  if (dimf > 0) D.laf.setSegment(nv, -1.0 * R.lf - M.Qff * D.MJtJinv_IDC.segment(nv, dimf));
  M.Qxx -= D.MJtJinv_dIDCdqv.t() * D.Qafqv;
  {
    Mat lx = D.MJtJinv_dIDCdqv.t() * D.laf;
    for (int r = 0; r < nv; ++r) { R.lq[r] -= lx[r]; R.lv[r] -= lx[nv + r]; }
  }
  if (!impulse) {
    M.Qxu_full -= D.MJtJinv_dIDCdqv.t() * D.Qafu_full;
    M.Quu_full += D.MJtJinv.block(0, 0, nv, nv + dimf) * D.Qafu_full;
    Mat t1 = D.MJtJinv.block(0, 0, nv, nv + dimf) * D.laf;
    for (int r = 0; r < kP; ++r) R.lu_passive[r] += t1[r];
    for (int r = 0; r < nu; ++r) R.lu[r] += t1[kP + r];
    M.Fvu = dt * D.MJtJinv.block(0, kP, nv, nu);
  }
  M.Fvq = (-dt) * D.MJtJinv_dIDCdqv.block(0, 0, nv, nv);
  M.Fvv = (-dt) * D.MJtJinv_dIDCdqv.block(0, nv, nv, nv) + Mat::Identity(nv);
  for (int r = 0; r < nv; ++r) R.Fv[r] -= dt * D.MJtJinv_IDC[r];
  // ---- ContactDynamics::condenseSwitchingConstraint (contact_dynamics.hxx:193-199)
  FLOP_REGION_SET(R_SWITCH);
  if (nd.sw_event >= 0) {
    W.Phix -= W.Phia * D.MJtJinv_dIDCdqv.block(0, 0, nv, 2 * nv);
    W.Phiu = W.Phia * D.MJtJinv.block(0, kP, nv, nu);
    R.P -= W.Phia * D.MJtJinv_IDC.segment(0, nv);
  }
}

// TerminalOCP::linearizeOCP / computeKKTResidual (terminal_ocp.hxx:50-66, 118-131)
void OCPSolver::linearizeTerminal(Robot& robot, int p, const Mat& q_prev, bool residual_only) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const NodeC& nd = chain[p];
  const SplitSolutionC& sN = s[nd.slot];
  SplitKKTMatrixC& M = kkt_matrix[nd.slot];
  SplitKKTResidualC& R = kkt_residual[nd.slot];
  const int nv = nv_;
  R.lq.setZero(); R.lv.setZero();
  Mat q_ref, qdiff, Jq;
  qRef(nd.t, q_ref);
  robot.subtractConfiguration(sN.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(sN.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = cost.qf_weight[r] * qdiff[r];
  R.lq += Jq.t() * Wq;
  const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  const real vs = vRefScale(cost, nd.t);      // TimeVaryingConfigurationSpaceCost::v_ref(t)
  for (int r = 0; r < nv; ++r) R.lv[r] += cost.vf_weight[r] * (sN.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]));
  Mat task_H;
  if (cost.task_dim) { real c_; Mat g_; taskTerms(task_robot, cost, nd.t, cost.task_weightf, sN.q, c_, g_, task_H); R.lq += g_; }
  // linearizeForwardEulerTerminal (state_equation.hxx:66-83)
  Mat Fqq_prev; robot.dSubtractdConfigurationMinus(q_prev, sN.q, Fqq_prev);
  M.Fqq_prev6 = Fqq_prev.block(0, 0, kP, kP);
  Mat t1 = M.Fqq_prev6.t() * sN.lmd.segment(0, kP);
  for (int r = 0; r < kP; ++r) R.lq[r] += t1[r];
  for (int r = kP; r < nv; ++r) R.lq[r] -= sN.lmd[r];
  R.lv -= sN.gmm;
  if (residual_only) return;
  Robot::dSubtractdConfigurationInverse(M.Fqq_prev6, M.Fqq_prev_inv);     // condenseForwardEulerTerminal
  M.Qxx.setZero();
  Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.qf_weight[r];
  M.Qxx.addBlock(0, 0, Jq.t() * WJ);
  for (int r = 0; r < nv; ++r) M.Qxx(nv + r, nv + r) += cost.vf_weight[r];
  if (cost.task_dim) M.Qxx.addBlock(0, 0, task_H);
  if (keep_uncondensed) {
    UncondensedC& U = unc[nd.slot];
    U = UncondensedC();
    U.valid = true; U.kind = (int)nd.kind;
    U.Qxx = M.Qxx; U.lq = R.lq; U.lv = R.lv; U.Fqq_prev = M.Fqq_prev6;
  }
}

// OCPLinearizer::runParallel (ocp_linearizer.hxx:113-228); q_prev (:231-248) is the chain predecessor's q
void OCPSolver::linearizeOCP(real t, const Mat& q) {
  discretize(t);
  if ((int)robots_.size() != nthreads) setNumThreads(nthreads);
  const int Mc = M();
  if (keep_uncondensed) unc.assign(nslots(), UncondensedC());
  #pragma omp parallel for num_threads(nthreads)
  for (int p = 0; p < Mc; ++p) {
    Robot& rb = robots_[ORACLE_THREAD_NUM];
    const Mat& q_prev = (p == 0) ? q : s[chain[p - 1].slot].q;
    if (p < Mc - 1) linearizeNode(rb, p, q_prev, false);
    else linearizeTerminal(rb, p, q_prev, false);
  }
}

void OCPSolver::computeKKTResidual(real t, const Mat& q, const Mat& /*v*/) {
  discretize(t);
  if ((int)robots_.size() != nthreads) setNumThreads(nthreads);
  const int Mc = M();
  #pragma omp parallel for num_threads(nthreads)
  for (int p = 0; p < Mc; ++p) {
    Robot& rb = robots_[ORACLE_THREAD_NUM];
    const Mat& q_prev = (p == 0) ? q : s[chain[p - 1].slot].q;
    if (p < Mc - 1) linearizeNode(rb, p, q_prev, true);
    else linearizeTerminal(rb, p, q_prev, true);
  }
}

// OCPLinearizer::KKTError (ocp_linearizer.cpp:98-137); SplitOCP::squaredNormKKTResidual (split_ocp.hxx:251-267);
// ImpulseSplitOCP::squaredNormKKTResidual (impulse_split_ocp.hxx:127-137)
// OCPSolver::isCurrentSolutionFeasible (ocp_solver.cpp:216-248) with the component tests of joint_*_limit.cpp:36-47 and
// linearized_(impulse_)friction_cone.cpp:82-99, in the reference's order (stages; the terminal stage carries no constraints, impulses, aux, lifts)
int OCPSolver::isCurrentSolutionFeasible() const {
  for (int kind = NodeC::Stage; kind <= NodeC::Lift; ++kind)
    for (int p = 0; p < (int)chain.size(); ++p) {
      const auto& nd = chain[p];
      if (!(nd.kind == kind)) continue;
      const SplitSolutionC& si = s[nd.slot];
      for (int c = 0; c < NCOMP; ++c) {
        if (!jointComp(c) || !componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
        for (int r = 0; r < nu_; ++r) {
          const real x = limitedVar(si, c, r, nv_, nu_), lim = limitOf(robot.model(), cons, c, r);
          if ((c & 1) ? x > lim : x < lim) return p;
        }
      }
      if (componentValid(10, nd)) {                        // ContactDistance::isFeasible (contact_distance.cpp:44-55)
        Robot rb2 = robot;
        const Mat zero(nv_);
        rb2.updateKinematics(si.q, zero, zero);
        const ContactStatus& cs2 = nodeContacts(p);
        for (int cc = 0; cc < nc_; ++cc) if (!cs2.active[cc]) { real pw[3]; rb2.contactFrame(cc, pw, nullptr, nullptr, nullptr); if (pw[2] <= 0) return p; }
      }
      if (!componentValid(6, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      const ContactStatus& cs = nodeContacts(p);
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
        for (int r = 0; r < CR; ++r) if (res[r] > 0) return p;
      }
    }
  return -1;
}

real OCPSolver::KKTError() {
  real sum = 0;
  for (int p = 0; p < M() - 1; ++p) {
    const NodeC& nd = chain[p];
    const SplitKKTResidualC& R = kkt_residual[nd.slot];
    const real dt = nd.kind == NodeC::Impulse ? real(1.0) : nd.dt;
    real e = R.lq.squaredNorm() + R.lv.squaredNorm() + R.la.squaredNorm() + R.lf.squaredNorm() + R.lu_passive.squaredNorm() +
               R.lu.squaredNorm() + R.Fq.squaredNorm() + R.Fv.squaredNorm() + dt * dt * cd[nd.slot].IDC.squaredNorm();
    real c2 = 0;
    for (int c = 0; c < NCOMP; ++c) if (componentValid(c, nd)) c2 += ipm[nd.slot][c].residual.squaredNorm() + ipm[nd.slot][c].duality.squaredNorm();
    sum += e + dt * dt * c2 + R.P.squaredNorm();
  }
  const SplitKKTResidualC& RN = kkt_residual[chain.back().slot];
  sum += RN.lq.squaredNorm() + RN.lv.squaredNorm();
  return std::sqrt(sum);
}

// ---------------------------------------------------------------- Riccati ----
// RiccatiRecursionSolver::backwardRiccatiRecursion (riccati_recursion_solver.cpp:48-107): a walk down the chain.
// Stage / Aux / Lift: SplitRiccatiFactorizer::backwardRiccatiRecursion (split_riccati_factorizer.hxx:24-101), with the
// Schur-complement variant on stages that carry a switching constraint.  Impulse:
// ImpulseSplitRiccatiFactorizer::backwardRiccatiRecursion (impulse_backward_riccati_recursion_factorizer.hxx:30-107) --
// the same recursion with Fqv = 0 and no control.
void OCPSolver::backwardRiccatiRecursion() {
  FLOP_REGION(R_RICCATI_BWD);
  const int nv = nv_, nu = nu_, nj = nv - kP;
  {
    const int sl = chain.back().slot;
    riccati[sl].Pqq = kkt_matrix[sl].Qxx.block(0, 0, nv, nv);
    riccati[sl].Pvv = kkt_matrix[sl].Qxx.block(nv, nv, nv, nv);
    riccati[sl].Pqv = Mat(nv, nv);
    riccati[sl].sq = -kkt_residual[sl].lq;
    riccati[sl].sv = -kkt_residual[sl].lv;
  }
  for (int p = M() - 2; p >= 0; --p) {
    const NodeC& nd = chain[p];
    const bool impulse = nd.kind == NodeC::Impulse;
    const real dt = impulse ? real(0.0) : nd.dt;
    const int sl = nd.slot;
    const RiccatiC& rn = riccati[chain[p + 1].slot];
    SplitKKTMatrixC& Mx = kkt_matrix[sl];
    SplitKKTResidualC& R = kkt_residual[sl];
    // BackwardRiccatiRecursionFactorizer::factorizeKKTMatrix (backward_riccati_recursion_factorizer.hxx:44-114)
    Mat AtPqq(nv, nv), AtPqv(nv, nv), AtPvq(nv, nv), AtPvv(nv, nv);
    AtPqq.setBlock(0, 0, Mx.Fqq6.t() * rn.Pqq.block(0, 0, kP, nv)); AtPqq.setBlock(kP, 0, rn.Pqq.block(kP, 0, nj, nv));
    AtPqv.setBlock(0, 0, Mx.Fqq6.t() * rn.Pqv.block(0, 0, kP, nv)); AtPqv.setBlock(kP, 0, rn.Pqv.block(kP, 0, nj, nv));
    if (!impulse) {
      AtPvq.setBlock(0, 0, Mx.Fqv6.t() * rn.Pqq.block(0, 0, kP, nv)); AtPvq.setBlock(kP, 0, dt * rn.Pqq.block(kP, 0, nj, nv));
      AtPvv.setBlock(0, 0, Mx.Fqv6.t() * rn.Pqv.block(0, 0, kP, nv)); AtPvv.setBlock(kP, 0, dt * rn.Pqv.block(kP, 0, nj, nv));
    }
    AtPqq += Mx.Fvq.t() * rn.Pqv.t();
    AtPqv += Mx.Fvq.t() * rn.Pvv;
    AtPvq += Mx.Fvv.t() * rn.Pqv.t();
    AtPvv += Mx.Fvv.t() * rn.Pvv;
    Mat Qqq = Mx.Qxx.block(0, 0, nv, nv), Qqv = Mx.Qxx.block(0, nv, nv, nv), Qvv = Mx.Qxx.block(nv, nv, nv, nv);
    Qqq.addBlock(0, 0, AtPqq.block(0, 0, nv, kP) * Mx.Fqq6); Qqq.addBlock(0, kP, AtPqq.block(0, kP, nv, nj));
    if (!impulse) {
      Qqv.addBlock(0, 0, AtPqq.block(0, 0, nv, kP) * Mx.Fqv6); Qqv.addBlock(0, kP, AtPqq.block(0, kP, nv, nj), dt);
      Qvv.addBlock(0, 0, AtPvq.block(0, 0, nv, kP) * Mx.Fqv6); Qvv.addBlock(0, kP, AtPvq.block(0, kP, nv, nj), dt);
    }
    Qqq += AtPqv * Mx.Fvq;
    Qqv += AtPqv * Mx.Fvv;
    Qvv += AtPvv * Mx.Fvv;
    Mx.Qxx.setBlock(0, 0, Qqq); Mx.Qxx.setBlock(0, nv, Qqv); Mx.Qxx.setBlock(nv, nv, Qvv); Mx.Qxx.setBlock(nv, 0, Qqv.t());
    RiccatiC& r = riccati[sl];
    Mat Qqu(nv, nu), Qvu(nv, nu);
    if (impulse) {
      K[sl].setZero(); k[sl].setZero();
      r.Pqq = Qqq; r.Pqv = Qqv; r.Pvv = Qvv;
    } else {
      Mat BtPq = Mx.Fvu.t() * rn.Pqv.t();
      Mat BtPv = Mx.Fvu.t() * rn.Pvv;
      Qqu = Mx.Qxu_full.block(0, kP, nv, nu); Qvu = Mx.Qxu_full.block(nv, kP, nv, nu);
      Qqu += AtPqv * Mx.Fvu;
      Qvu += AtPvv * Mx.Fvu;
      Mx.Qxu_full.setBlock(0, kP, Qqu); Mx.Qxu_full.setBlock(nv, kP, Qvu);
      Mat Quu = Mx.Quu_full.block(kP, kP, nu, nu);
      Quu += BtPv * Mx.Fvu;
      Mx.Quu_full.setBlock(kP, kP, Quu);
      R.lu += BtPq * R.Fq;
      R.lu += BtPv * R.Fv;
      R.lu -= Mx.Fvu.t() * rn.sv;
      LLT llt;
      if (!llt.compute(Quu)) throw std::runtime_error("Riccati: Quu not positive definite at chain position " + std::to_string(p));
      Mat Qxu(2 * nv, nu); Qxu.setBlock(0, 0, Qqu); Qxu.setBlock(nv, 0, Qvu);
      SwitchingC& W = sw[sl];
      Mat DtM, KtDtM;
      if (nd.sw_event < 0) {
        // split_riccati_factorizer.hxx:24-41
        K[sl] = -llt.solve(Qxu.t());
        k[sl] = -llt.solve(R.lu);
      } else {
        // Schur complement w.r.t. the switching constraint (:43-101)
        Mat Ginv = llt.solve(Mat::Identity(nu));
        Mat DGinv = llt.solve(W.Phiu.t()).t();
        Mat S = DGinv * W.Phiu.t();
        LLT llt_s;
        if (!llt_s.compute(S)) throw std::runtime_error("Riccati: switching-constraint Schur complement not positive definite");
        Mat SinvDGinv = llt_s.solve(DGinv);
        Ginv -= SinvDGinv.t() * DGinv;
        K[sl] = -1.0 * (Ginv * Qxu.t());
        K[sl] -= SinvDGinv.t() * W.Phix;
        k[sl] = -1.0 * (Ginv * R.lu);
        k[sl] -= SinvDGinv.t() * R.P;
        W.M = llt_s.solve(W.Phix);
        W.M -= SinvDGinv * Qxu.t();
        W.m = llt_s.solve(R.P);
        W.m -= SinvDGinv * R.lu;
        DtM = W.Phiu.t() * W.M;
        KtDtM = K[sl].t() * DtM;
      }
      // factorizeRiccatiFactorization (backward_riccati_recursion_factorizer.hxx:117-161)
      r.Pqq = Qqq; r.Pqv = Qqv; r.Pvv = Qvv;
      Mat GK = Quu * K[sl];
      Mat Kq = K[sl].block(0, 0, nu, nv), Kv = K[sl].block(0, nv, nu, nv);
      r.Pqq -= Kq.t() * GK.block(0, 0, nu, nv);
      r.Pqv -= Kq.t() * GK.block(0, nv, nu, nv);
      r.Pvv -= Kv.t() * GK.block(0, nv, nu, nv);
    }
    r.Pqq = 0.5 * (r.Pqq + r.Pqq.t());
    r.Pvv = 0.5 * (r.Pvv + r.Pvv.t());
    r.sq = Mat(nv); r.sv = Mat(nv);
    r.sq.setSegment(0, Mx.Fqq6.t() * rn.sq.segment(0, kP)); r.sq.setSegment(kP, rn.sq.segment(kP, nj));
    if (!impulse) { r.sv.setSegment(0, Mx.Fqv6.t() * rn.sq.segment(0, kP)); r.sv.setSegment(kP, dt * rn.sq.segment(kP, nj)); }
    r.sq += Mx.Fvq.t() * rn.sv;
    r.sv += Mx.Fvv.t() * rn.sv;
    r.sq -= AtPqq * R.Fq;
    r.sq -= AtPqv * R.Fv;
    r.sv -= AtPvq * R.Fq;
    r.sv -= AtPvv * R.Fv;
    r.sq -= R.lq;
    r.sv -= R.lv;
    if (!impulse) {
      r.sq -= Qqu * k[sl];
      r.sv -= Qvu * k[sl];
      if (nd.sw_event >= 0) {
        // split_riccati_factorizer.hxx:88-100
        SwitchingC& W = sw[sl];
        Mat DtM = W.Phiu.t() * W.M;
        Mat KtDtM = K[sl].t() * DtM;
        r.Pqq -= KtDtM.block(0, 0, nv, nv);
        r.Pqq -= KtDtM.block(0, 0, nv, nv).t();
        r.Pqv -= KtDtM.block(0, nv, nv, nv);
        r.Pqv -= KtDtM.block(nv, 0, nv, nv).t();
        r.Pvv -= KtDtM.block(nv, nv, nv, nv);
        r.Pvv -= KtDtM.block(nv, nv, nv, nv).t();
        r.sq -= W.Phix.block(0, 0, W.Phix.r, nv).t() * W.m;
        r.sv -= W.Phix.block(0, nv, W.Phix.r, nv).t() * W.m;
      }
    }
  }
}

// computeInitialStateDirection + forwardRiccatiRecursion (riccati_recursion_solver.cpp:110-162;
// split_riccati_factorizer.hxx:103-128; impulse_split_riccati_factorizer.hxx:27-44)
void OCPSolver::forwardRiccatiRecursion(const Mat& q, const Mat& v) {
  FLOP_REGION(R_RICCATI_FWD);
  const int nv = nv_, nj = nv - kP;
  {
    const int s0 = chain[0].slot;
    robot.subtractConfiguration(q, s[s0].q, d[s0].dq);
    d[s0].dq.setSegment(0, -1.0 * (kkt_matrix[s0].Fqq_prev_inv * d[s0].dq.segment(0, kP)));
    d[s0].dv = v - s[s0].v;
  }
  for (int p = 0; p < M() - 1; ++p) {
    const NodeC& nd = chain[p];
    const int sl = nd.slot, sn = chain[p + 1].slot;
    const bool impulse = nd.kind == NodeC::Impulse;
    const real dt = impulse ? real(0.0) : nd.dt;
    const SplitKKTMatrixC& Mx = kkt_matrix[sl];
    const SplitKKTResidualC& R = kkt_residual[sl];
    Mat dx(2 * nv); dx.setSegment(0, d[sl].dq); dx.setSegment(nv, d[sl].dv);
    if (impulse) d[sl].du.setZero(); else d[sl].du = K[sl] * dx + k[sl];
    Mat dqn = R.Fq, dvn = R.Fv;
    Mat h = Mx.Fqq6 * d[sl].dq.segment(0, kP);
    if (!impulse) h += Mx.Fqv6 * d[sl].dv.segment(0, kP);
    for (int r = 0; r < kP; ++r) dqn[r] += h[r];
    for (int r = 0; r < nj; ++r) dqn[kP + r] += d[sl].dq[kP + r] + dt * d[sl].dv[kP + r];
    dvn += Mx.Fvq * d[sl].dq;
    dvn += Mx.Fvv * d[sl].dv;
    if (!impulse) dvn += Mx.Fvu * d[sl].du;
    d[sn].dq = dqn; d[sn].dv = dvn;
  }
}

static real fractionToBoundary(real rate, const Mat& vec, const Mat& dvec) {      // pdipm.hxx:52-73
  real m = 1;
  for (int i = 0; i < vec.size(); ++i) {
    const real f = -rate * (vec[i] / dvec[i]);
    if (f > 0 && f < 1 && f < m) m = f;
  }
  return m;
}


// RiccatiRecursionSolver::computeDirection (riccati_recursion_solver.cpp:165-251)
void OCPSolver::computeDirection() {
  FLOP_REGION(R_EXPAND);
  const int nv = nv_, nu = nu_;
  real pmin = 1, dmin = 1;
  const int Mc = M();
  #pragma omp parallel for num_threads(nthreads) reduction(min : pmin, dmin)
  for (int p = 0; p < Mc; ++p) {
    const NodeC& nd = chain[p];
    const int sl = nd.slot;
    const RiccatiC& r = riccati[sl];
    d[sl].dlmd = r.Pqq * d[sl].dq + r.Pqv * d[sl].dv - r.sq;
    d[sl].dgmm = r.Pqv.t() * d[sl].dq + r.Pvv * d[sl].dv - r.sv;
    if (nd.kind == NodeC::Terminal) continue;
    const bool impulse = nd.kind == NodeC::Impulse;
    const ContactDynamicsDataC& D = cd[sl];
    const ContactStatus& cs = nodeContacts(p);
    const int dimf = cs.dimf();
    // ContactDynamics::computeCondensedPrimalDirection (contact_dynamics.hxx:161-168) /
    // ImpulseDynamicsForwardEuler::expansionPrimal (impulse_dynamics_forward_euler.hxx:119-125)
    Mat dx(2 * nv); dx.setSegment(0, d[sl].dq); dx.setSegment(nv, d[sl].dv);
    d[sl].daf = -1.0 * (D.MJtJinv_dIDCdqv * dx);
    if (!impulse) d[sl].daf += D.MJtJinv.block(0, kP, nv + dimf, nu) * d[sl].du;
    d[sl].daf -= D.MJtJinv_IDC;
    for (int r2 = 0; r2 < dimf; ++r2) d[sl].daf[nv + r2] *= -1;
    // SplitRiccatiFactorizer::computeLagrangeMultiplierDirection (split_riccati_factorizer.hxx:139-145)
    if (nd.sw_event >= 0) d[sl].dxi = sw[sl].M * dx + sw[sl].m; else d[sl].dxi = Mat(0);
    // Constraints::computeSlackAndDualDirection + step sizes
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      IpmData& data = ipm[sl][c];
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r2 = 0; r2 < nu; ++r2) {
          const real dxr = c < 2 ? d[sl].dq[kP + r2] : (c < 4 ? d[sl].dv[kP + r2] : (c < 6 ? d[sl].du[r2] : d[sl].daf[kP + r2]));
          data.dslack[r2] = -sgn * dxr - data.residual[r2];
          data.ddual[r2] = -(data.dual[r2] * data.dslack[r2] + data.duality[r2]) / data.slack[r2];
        }
      } else if (c == 10) {
        // ContactDistance::computeSlackAndDualDirection (contact_distance.cpp:105-129)
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }
        for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
          real Jdq = 0; for (int col = 0; col < nv; ++col) Jdq += cd_J[sl](cc, col) * d[sl].dq[col];
          data.dslack[cc] = Jdq - data.residual[cc];
          data.ddual[cc] = -(data.dual[cc] * data.dslack[cc] + data.duality[cc]) / data.slack[cc];
        }
      } else {
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }   // linearized_friction_cone.cpp:162-163
        int st = 0;
        for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
          const ConeEval ce = coneEval(CK, cons.mu, s[sl].f[cc]); const real (*Jc)[3] = ce.J;     // (FrictionCone: data.r[i] of the linearisation)
          for (int r2 = 0; r2 < CR; ++r2) {
            const int idx = CR * cc + r2;
            real Jdf = 0; for (int x = 0; x < 3; ++x) Jdf += Jc[r2][x] * d[sl].daf[nv + st + x];
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
  FLOP_REGION(R_INTEGRATE);
  const int nv = nv_, nu = nu_;
  const real ap = primal_step_size, ad = dual_step_size;
  const int Mc = M();
  #pragma omp parallel for num_threads(nthreads)
  for (int p = 0; p < Mc; ++p) {
    const NodeC& nd = chain[p];
    const int sl = nd.slot;
    SplitKKTMatrixC& Mx = kkt_matrix[sl];
    const bool terminal = nd.kind == NodeC::Terminal, impulse = nd.kind == NodeC::Impulse;
    const ContactStatus& cs = nodeContacts(p);
    if (!terminal) {
      ContactDynamicsDataC& D = cd[sl];
      SplitKKTResidualC& R = kkt_residual[sl];
      const real dt = impulse ? real(1.0) : nd.dt;
      const int dimf = cs.dimf();
      Mat dx(2 * nv); dx.setSegment(0, d[sl].dq); dx.setSegment(nv, d[sl].dv);
      const Mat& dgmm = d[chain[p + 1].slot].dgmm;
      // ContactDynamics::computeCondensedDualDirection (contact_dynamics.hxx:171-190) /
      // ImpulseDynamicsForwardEuler::expansionDual (impulse_dynamics_forward_euler.hxx:127-137)
      if (!impulse) {
        d[sl].dnu_passive = R.lu_passive;
        d[sl].dnu_passive += Mx.Quu_full.block(0, kP, kP, nu) * d[sl].du;
        d[sl].dnu_passive += Mx.Qxu_full.block(0, 0, 2 * nv, kP).t() * dx;
        d[sl].dnu_passive += dt * (D.MJtJinv.block(0, 0, kP, nv) * dgmm);
        d[sl].dnu_passive = (-1.0 / dt) * d[sl].dnu_passive;
      } else {
        d[sl].dnu_passive.setZero();
      }
      D.laf += D.Qafqv * dx;
      if (!impulse) D.laf += D.Qafu_full.block(0, kP, nv + dimf, nu) * d[sl].du;
      for (int r = 0; r < nv; ++r) D.laf[r] += dt * dgmm[r];
      d[sl].dbetamu = (-1.0 / dt) * (D.MJtJinv * D.laf);
    }
    // stateequation::correctCostateDirectionForwardEuler (state_equation.hxx:96-108)
    d[sl].dlmd.setSegment(0, -1.0 * (Mx.Fqq_prev_inv.t() * d[sl].dlmd.segment(0, kP)));
    // SplitOCP / ImpulseSplitOCP / TerminalOCP::updatePrimal -> (Impulse)SplitSolution::integrate
    // (split_solution.hxx:215-240, impulse_split_solution.hxx:188-203)
    SplitSolutionC& si = s[sl];
    si.lmd += ap * d[sl].dlmd;
    si.gmm += ap * d[sl].dgmm;
    Mat qn; robot.integrateConfiguration(si.q, d[sl].dq, ap, qn); si.q = qn;
    si.v += ap * d[sl].dv;
    if (terminal) continue;
    si.a += ap * d[sl].daf.segment(0, nv);
    si.beta += ap * d[sl].dbetamu.segment(0, nv);
    if (!impulse) {
      si.u += ap * d[sl].du;
      si.nu_passive += ap * d[sl].dnu_passive;
    }
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
      for (int r = 0; r < 3; ++r) { si.f[c][r] += ap * d[sl].daf[nv + st + r]; si.mu[c][r] += ap * d[sl].dbetamu[nv + st + r]; }
      st += 3;
    }
    if (nd.sw_event >= 0) for (int r = 0; r < d[sl].dxi.size(); ++r) si.xi[r] += ap * d[sl].dxi[r];
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      ipm[sl][c].slack += ap * ipm[sl][c].dslack;
      ipm[sl][c].dual += ad * ipm[sl][c].ddual;
    }
  }
}

void OCPSolver::updateSolution(real t, const Mat& q, const Mat& v, bool use_line_search) {
  linearizeOCP(t, q);
  auto t0 = std::chrono::steady_clock::now();
  backwardRiccatiRecursion();
  forwardRiccatiRecursion(q, v);
  riccati_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  computeDirection();
  if (use_line_search)            // ocp_solver.cpp:84-90
    primal_step_size = line_search.computeStepSize([&](real a) { return costAndViolation(a); }, primal_step_size);
  integrateSolution();
}

// LineSearch::computeSolution + computeCostAndViolation for the OCP (src/line_search/line_search.cpp:63-196, 305-343;
// line_search.hpp:134-158).  Per stage of the chain:
//   cost      = SplitOCP::stageCost (split_ocp.hxx:270-289): stage cost + dt * barrier(slack + alpha dslack)
//               ImpulseSplitOCP::stageCost (impulse_split_ocp.hxx:155-170), TerminalOCP::terminalCost (terminal_ocp.hxx:81-87)
//   violation = SplitOCP::constraintViolation (split_ocp.hxx:292-346): |Fx|_1 + dt |[ID - u; C]|_1 + dt |g + slack|_1 (+ |P|_1 of the
//               switching constraint), with the CURRENT slack; the impulse stage with dt = 1 (impulse_split_ocp.hxx:173-192)
// The reference evaluates the state-equation residual of a grid stage in front of an event against the NEXT GRID STAGE (the
// value written for the event stage is overwritten a few lines further down, line_search.cpp:80-113); reproduced here.
std::pair<real, real> OCPSolver::costAndViolation(real alpha) {
  const int nv = nv_, nu = nu_;
  Robot rb = robot;
  real cost_sum = 0, viol_sum = 0;
  auto trial = [&](int p) {
    const NodeC& nd = chain[p];
    const int sl = nd.slot;
    SplitSolutionC x = s[sl];
    if (alpha > 0) {
      Mat qn; rb.integrateConfiguration(s[sl].q, d[sl].dq, alpha, qn); x.q = qn;
      x.v = s[sl].v + alpha * d[sl].dv;
      if (nd.kind != NodeC::Terminal) {
        const ContactStatus& cs = nodeContacts(p);
        x.a = s[sl].a + alpha * d[sl].daf.segment(0, nv);
        if (nd.kind != NodeC::Impulse) x.u = s[sl].u + alpha * d[sl].du;
        int st = 0;
        for (int c = 0; c < nc_; ++c) if (cs.active[c]) { for (int k2 = 0; k2 < 3; ++k2) x.f[c][k2] = s[sl].f[c][k2] + alpha * d[sl].daf[nv + st + k2]; st += 3; }
      }
    }
    return x;
  };
  for (int p = 0; p < M(); ++p) {
    const NodeC& nd = chain[p];
    const int sl = nd.slot;
    const SplitSolutionC x = trial(p);
    Mat q_ref, qdiff;
    qRef(nd.t, q_ref);
    rb.subtractConfiguration(x.q, q_ref, qdiff);
    const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
    const real vs = vRefScale(cost, nd.t);
    if (nd.kind == NodeC::Terminal) {
      real l = 0;
      for (int r = 0; r < nv; ++r) {
        const real dvr = x.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]);
        l += cost.qf_weight[r] * qdiff[r] * qdiff[r] + cost.vf_weight[r] * dvr * dvr;
      }
      cost_sum += 0.5 * l;
      if (cost.task_dim) { real c_; Mat g_, H_; taskTerms(task_robot, cost, nd.t, cost.task_weightf, x.q, c_, g_, H_); cost_sum += c_; }
      continue;
    }
    const bool impulse = nd.kind == NodeC::Impulse;
    const ContactStatus& cs = nodeContacts(p);
    const real dt = impulse ? real(1.0) : nd.dt, dtq = impulse ? real(0.0) : nd.dt;
    const real* wq = impulse ? cost.qi_weight : cost.q_weight;
    const real* wv = impulse ? cost.vi_weight : cost.v_weight;
    const real* wa = impulse ? cost.dvi_weight : cost.a_weight;
    const real (*wf)[3] = impulse ? cost.fi_weight : cost.f_weight;
    const real (*rf)[3] = impulse ? cost.fi_ref : cost.f_ref;
    // ---- cost
    real l = 0;
    for (int r = 0; r < nv; ++r) {
      const real dvr = x.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]);
      l += wq[r] * qdiff[r] * qdiff[r] + wv[r] * dvr * dvr + wa[r] * x.a[r] * x.a[r];
    }
    if (!impulse) for (int r = 0; r < nu; ++r) l += cost.u_weight[r] * (x.u[r] - cost.u_ref[r]) * (x.u[r] - cost.u_ref[r]);
    if (cost.task_dim) { real c_; Mat g_, H_; taskTerms(task_robot, cost, nd.t, impulse ? cost.task_weighti : cost.task_weight, x.q, c_, g_, H_); l += 2 * c_; }
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) for (int k2 = 0; k2 < 3; ++k2) l += wf[c][k2] * (x.f[c][k2] - rf[c][k2]) * (x.f[c][k2] - rf[c][k2]);
    real barrier = 0, primal = 0;
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      const IpmData& data = ipm[sl][c];
      for (int r = 0; r < data.slack.size(); ++r) barrier -= cons.barrier * std::log(data.slack[r] + alpha * data.dslack[r]);      // pdipm.hxx:84-87
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r = 0; r < nu; ++r) primal += std::fabs(sgn * (limitedVar(x, c, r, nv, nu) - limitOf(rb.model(), cons, c, r)) + data.slack[r]);
      } else if (c == 10) {
        // ContactDistance::computePrimalAndDualResidual at the trial configuration (contact_distance.cpp:132-146)
        rb.updateKinematics(x.q, x.v, x.a);
        for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) { real pw[3]; rb.contactFrame(cc, pw, nullptr, nullptr, nullptr); primal += std::fabs(-pw[2] + data.slack[cc]); }
      } else {
        for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
          const ConeEval ce = coneEval(CK, cons.mu, x.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
          for (int r = 0; r < CR; ++r) primal += std::fabs(res[r] + data.slack[CR * cc + r]);
        }
      }
    }
    cost_sum += 0.5 * dt * l + dt * barrier;
    // ---- violation: state equation against the successor the reference pairs the stage with
    int pn = p + 1;
    if (nd.kind == NodeC::Stage && (chain[pn].kind == NodeC::Impulse || chain[pn].kind == NodeC::Lift)) { ++pn; while (chain[pn].kind != NodeC::Stage && chain[pn].kind != NodeC::Terminal) ++pn; }
    const SplitSolutionC xn = trial(pn);
    Mat diff; rb.subtractConfiguration(x.q, xn.q, diff);
    real viol = 0;
    for (int r = 0; r < nv; ++r) viol += std::fabs(diff[r] + dtq * x.v[r]) + std::fabs(x.v[r] + dt * x.a[r] - xn.v[r]);
    // contact dynamics residual (contact_dynamics.hxx:202-217; impulse_dynamics_forward_euler.hxx:145-158)
    Mat ID, C, zero(nv);
    if (impulse) rb.updateKinematics(x.q, x.v + x.a, Mat(nv)); else rb.updateKinematics(x.q, x.v, x.a);
    rb.setContactForces(cs.active, x.f);
    if (impulse) { rb.RNEA(x.q, zero, x.a, ID, false); rb.computeImpulseVelocityResidual(cs.active, C); }
    else { rb.RNEA(x.q, x.v, x.a, ID); for (int r = 0; r < nu; ++r) ID[kP + r] -= x.u[r]; rb.computeBaumgarteResidual(cs.active, dt_, cs.points, C); }
    viol += dt * (ID.lpNorm1() + C.lpNorm1());
    viol += dt * primal;
    if (nd.sw_event >= 0) {        // forward_switching_constraint.hxx:27-47
      const ContactStatus& is = seq.impulse_status[nd.sw_event];
      const real dt1 = nd.dt, dt2 = nd.sw_dt_next;
      Mat dq_ = (dt1 + dt2) * x.v + (dt1 * dt2) * x.a, q_, Pm;
      rb.integrateConfiguration(x.q, dq_, 1.0, q_);
      rb.updateKinematics(q_, Mat(nv), Mat(nv));
      rb.computeContactResidual(is.active, is.points, Pm);
      viol += Pm.lpNorm1();
    }
    viol_sum += viol;
  }
  return {cost_sum, viol_sum};
}

// =============================================================================================== ParNMPC ====
ParNMPCSolver::ParNMPCSolver(const RModel& model, const RCost& cost_, const idocp_constraints_t& constraints, real T, int N,
                             int max_num_impulse)
    : robot(model), task_robot(makeTaskRobot(model, cost_)), cost(cost_), cons(constraints), next_s(robot), next_snew(robot), prev_s(robot), prev_snew(robot),
      N_ideal_(N), N_(N), nv_(model.nv), nu_(model.nu), nc_(model.ncontacts), max_events_(max_num_impulse), kP(model.nv - model.nu), T_(T), dt_(T / N) {
  if (T <= 0) throw std::out_of_range("invalid value: T must be positive!");
  if (N <= 0) throw std::out_of_range("invalid value: N must be positive!");
  if (max_num_impulse < 0) throw std::out_of_range("invalid value: max_num_impulse must be non-negative!");
  const int ns = nslots();
  s.assign(ns, SplitSolutionC(robot)); s_new = s;
  d.assign(ns, SplitDirectionC(robot));
  kkt_matrix.assign(ns, SplitKKTMatrixC(model.nv, model.nu));
  kkt_residual.assign(ns, SplitKKTResidualC(model.nv, model.nu));
  cd.resize(ns); ipm.resize(ns); cd_J.resize(ns);
  const int nx = 2 * nv_;
  KKT_mat_inv.assign(ns, Mat(2 * nx + nu_, 2 * nx + nu_));
  aux_mat.assign(ns, Mat(nx, nx));
  x_res.assign(ns, Mat(nx));
  sw_Pq.resize(ns);
  imp.resize(ns);
  contact_status.active.assign(nc_, false);
  contact_status.points.assign(nc_, Mat(3));
  seq.phases.assign(1, contact_status);
  next_aux = Mat(nx, nx);
}

int ParNMPCSolver::slotOf(int kind, int index) const {
  switch (kind) {
    case NodeC::Impulse: return N_ideal_ + index;
    case NodeC::Aux: return N_ideal_ + max_events_ + index;
    case NodeC::Lift: return N_ideal_ + 2 * max_events_ + index;
    default: return index;
  }
}

int ParNMPCSolver::haloSize(int kind) const {
  const int nq = robot.dimq(), nv = nv_, nx = 2 * nv;
  switch (kind) { case 0: case 4: return nq + nv; case 1: return 2 * nv + nq; case 2: return nx * nx; case 3: return 2 * nv; default: return nx * nx; }
}
void ParNMPCSolver::exportHalo(int kind, real* out) const {
  const int nq = robot.dimq(), nv = nv_;
  auto put = [&](const Mat& m, int off) { for (int k = 0; k < m.size(); ++k) out[off + k] = m[k]; };
  // first / last stage of this shard's chain (slots 0 and N - 1 of an event-free shard)
  const int first = chain.empty() ? 0 : chain.front().slot, last = chain.empty() ? N_ - 1 : chain.back().slot;
  switch (kind) {
    case 0: put(s[last].q, 0); put(s[last].v, nq); break;
    case 1: put(s[first].lmd, 0); put(s[first].gmm, nv); put(s[first].q, 2 * nv); break;
    case 2: put(aux_mat[first], 0); break;
    case 3: put(s_new[first].lmd, 0); put(s_new[first].gmm, nv); break;
    case 4: put(s_new[last].q, 0); put(s_new[last].v, nq); break;
    default: put(aux_mat[first], 0); break;
  }
}
void ParNMPCSolver::importHalo(int kind, const real* in) {
  const int nq = robot.dimq(), nv = nv_;
  auto get = [&](Mat& m, int off) { for (int k = 0; k < m.size(); ++k) m[k] = in[off + k]; };
  switch (kind) {
    case 0: get(prev_s.q, 0); get(prev_s.v, nq); break;                 // also handed to coarseUpdate as (q, v)
    case 1: get(next_s.lmd, 0); get(next_s.gmm, nv); get(next_s.q, 2 * nv); break;
    case 2: get(next_aux, 0); break;
    case 3: get(next_snew.lmd, 0); get(next_snew.gmm, nv); break;
    case 4: get(prev_snew.q, 0); get(prev_snew.v, nq); break;
    default: for (auto& a : aux_mat) get(a, 0); get(next_aux, 0); break;
  }
}

void ParNMPCSolver::setContactStatusUniformly(const std::vector<int>& active, const double* pts) {
  for (int c = 0; c < nc_; ++c) {
    contact_status.active[c] = active[c] != 0;
    for (int k2 = 0; k2 < 3; ++k2) contact_status.points[c][k2] = pts[3 * c + k2];
  }
  seq.phases.assign(1, contact_status);
  seq.event_time.clear(); seq.is_impulse.clear(); seq.impulse_status.clear();
  discretized_ = false;
}

// ParNMPCSolver::popBackContactStatus / popFrontContactStatus (parnmpc_solver.cpp:202-209) -> ContactSequence::pop_back / pop_front
void ParNMPCSolver::popBackContactStatus() { seqPopBack(seq, nc_); discretized_ = false; }
void ParNMPCSolver::popFrontContactStatus() { seqPopFront(seq, nc_); discretized_ = false; }

// ContactSequence::push_back (contact_sequence.hxx:63-117); capacities as in OCPSolver (see OCPSolver::pushBackContactStatus)
void ParNMPCSolver::pushBackContactStatus(const std::vector<int>& active, const double* pts, real switching_time) {
  if (seq.numEvents() + 1 > N_ideal_) throw std::runtime_error("Number of discrete events exceeds predefined max_num_events!");
  if (seq.numEvents() > 0 && switching_time <= seq.event_time.back()) throw std::runtime_error("event_time must be larger than the last event time!");
  const ContactStatus& pre = seq.phases.back();
  ContactStatus post, im;
  post.active.assign(nc_, false); post.points.assign(nc_, Mat(3));
  im.active.assign(nc_, false); im.points.assign(nc_, Mat(3));
  bool exist_impulse = false, exist_lift = false;
  for (int c = 0; c < nc_; ++c) {
    post.active[c] = active[c] != 0;
    for (int k2 = 0; k2 < 3; ++k2) { post.points[c][k2] = pts[3 * c + k2]; im.points[c][k2] = pts[3 * c + k2]; }
    if (pre.active[c]) { if (!post.active[c]) exist_lift = true; }
    else if (post.active[c]) { im.active[c] = true; exist_impulse = true; }
  }
  if (!exist_impulse && !exist_lift) throw std::runtime_error("discrete_event.existDiscreteEvent() must be true!");
  if ((exist_impulse ? seq.numImpulse() : seq.numLift()) + 1 > max_events_) throw std::runtime_error("more impulse / lift events than max_num_impulse!");
  seq.phases.push_back(post);
  seq.event_time.push_back(switching_time);
  seq.is_impulse.push_back(exist_impulse);
  seq.impulse_status.push_back(im);
  discretized_ = false;
}

// parnmpc_solver.cpp:106-160: every stage, including the event stages ("a" sets dv on impulse stages)
void ParNMPCSolver::setSolution(const std::string& name, const Mat& value) {
  for (auto& e : s) {
    if (name == "q") e.q = value;
    else if (name == "v") e.v = value;
    else if (name == "a") e.a = value;
    else if (name == "u") e.u = value;
    else if (name == "f") { for (auto& f : e.f) f = value; }
    else throw std::invalid_argument("invalid arugment: name must be q, v, a, f, or u!");
  }
}

// ParNMPCDiscretizer::discretizeOCP (parnmpc_discretizer.hxx:65-72): countDiscreteEvents (:246-262), countTimeSteps
// (:265-324), countTimeStages (:327-361), countContactPhase (:364-373)
void ParNMPCSolver::discretize(real t) {
  if (discretized_ && disc_t_ == t) return;
  chain.clear();
  const int Ne = seq.numEvents();
  if (Ne == 0) {
    N_ = N_ideal_;
    for (int i = 0; i < N_; ++i) {
      PNode nd;
      nd.kind = (has_terminal && i == N_ - 1) ? NodeC::Terminal : NodeC::Stage;
      nd.slot = i; nd.index = i; nd.t = t + (stage_offset + i + 1) * dt_; nd.dt = dt_; nd.phase = 0; nd.level = stage_offset + i + 1;
      chain.push_back(nd);
    }
    discretized_ = true; disc_t_ = t;
    return;
  }
  if (stage_offset != 0) throw std::logic_error("ParNMPC oracle: a horizon with discrete events is sharded by setChainSlice");
  const int Nid = N_ideal_, Ni = seq.numImpulse(), Nl = seq.numLift();
  const real dt_ideal = dt_, min_dt = std::sqrt(std::numeric_limits<real>::epsilon()), max_dt = dt_ideal - min_dt;
  std::vector<int> tsai(Ni + 1, -1), tsal(Nl + 1, -1);      // time stage AFTER the impulse / lift
  std::vector<real> t_imp(Ni + 1, 0.0), t_lift(Nl + 1, 0.0), dt_aux(Ni + 1, 0.0), dt_lift(Nl + 1, 0.0);
  for (int k = 0; k < Ni; ++k) { t_imp[k] = seq.event_time[seq.eventOfImpulse(k)]; tsai[k] = (int)std::floor((t_imp[k] - t) / dt_ideal); }
  for (int k = 0; k < Nl; ++k) { t_lift[k] = seq.event_time[seq.eventOfLift(k)]; tsal[k] = (int)std::floor((t_lift[k] - t) / dt_ideal); }
  std::vector<real> dts(Nid + 1, dt_ideal), ts(Nid + 1, 0.0);
  int ii = 0, li = 0, on_grid = 0;
  for (int i = 0; i < Nid; ++i) {
    const int stage = i - on_grid;
    if (ii < Ni && i == tsai[ii]) {
      dts[stage] = (i + 1) * dt_ideal + t - t_imp[ii];
      if (dts[stage] <= min_dt) { tsai[ii] = i + 1; ts[stage] = t + (i + 1) * dt_ideal; }
      else if (dts[stage] >= max_dt) { tsai[ii] = stage - 1; dt_aux[ii] = dt_ideal; ts[stage] = t + i * dt_ideal; ++on_grid; ++ii; }
      else { tsai[ii] = stage; dt_aux[ii] = dt_ideal - dts[stage]; ts[stage] = t + (i + 1) * dt_ideal; ++ii; }
    } else if (li < Nl && i == tsal[li]) {
      dts[stage] = (i + 1) * dt_ideal + t - t_lift[li];
      if (dts[stage] <= min_dt) { tsal[li] = i + 1; ts[stage] = t + (i + 1) * dt_ideal; }
      else if (dts[stage] >= max_dt) { tsal[li] = stage - 1; dt_lift[li] = dt_ideal; ts[stage] = t + i * dt_ideal; ++on_grid; ++li; }
      else { tsal[li] = stage; dt_lift[li] = dt_ideal - dts[stage]; ts[stage] = t + (i + 1) * dt_ideal; ++li; }
    } else {
      dts[stage] = dt_ideal; ts[stage] = t + (i + 1) * dt_ideal;
    }
  }
  N_ = Nid - on_grid;
  ts[N_ - 1] = t + T_;
  std::vector<int> imp_before(N_, -1), lift_before(N_, -1), phase(N_, 0);
  ii = 0; li = 0;
  int num_events = 0;
  for (int i = 0; i < N_; ++i) {
    if (ii < Ni && i == tsai[ii]) imp_before[i] = ii++;
    if (li < Nl && i == tsal[li]) lift_before[i] = li++;
    if (imp_before[i] >= 0 && lift_before[i] >= 0) throw std::runtime_error("ParNMPCDiscretizer: an impulse and a lift fall into the same time stage");
    if (imp_before[i] >= 0 || lift_before[i] >= 0) ++num_events;
    phase[i] = num_events;
  }
  if (ii != Ni || li != Nl) throw std::runtime_error("ParNMPCDiscretizer: a discrete event lies outside the horizon");
  for (int i = 0; i + 1 < N_; ++i) if (imp_before[i] >= 0 && imp_before[i + 1] >= 0) throw std::runtime_error("ParNMPCDiscretizer: impulses in consecutive time stages");
  // A lift or an impulse in front of the first time stage makes the event stages the first elements of the chain; their predecessor
  // is the measured state (backward_correction_solver.cpp:201-217, 232-246; the serial sweeps :283-292, :326-339 walk them like any
  // other).  For the impulse the reference's call at :203-211 leaves out the impulse status, so that SplitKKTMatrix::dimi() = 0
  // while SplitBackwardCorrection::coarseUpdate sizes the KKT inverse with s.aux.dimi() > 0 (split_backward_correction.hxx:46-52):
  // the two disagree and the inversion is not defined as written.  Restated here with the status passed -- the same aux stage as
  // everywhere else in the chain (:185-199), its predecessor being (q, v).
  for (int i = 0; i < N_; ++i) {
    const int phase_before = i > 0 ? phase[i - 1] : 0;
    if (imp_before[i] >= 0) {
      const int k = imp_before[i];
      PNode a; a.kind = NodeC::Aux; a.index = k; a.slot = slotOf(NodeC::Aux, k); a.t = t_imp[k]; a.dt = dt_aux[k]; a.phase = phase_before; a.level = 0;
      a.event = seq.eventOfImpulse(k);
      chain.push_back(a);
      PNode m; m.kind = NodeC::Impulse; m.index = k; m.slot = slotOf(NodeC::Impulse, k); m.t = t_imp[k]; m.dt = 0.0; m.phase = phase_before; m.level = -1;
      m.event = seq.eventOfImpulse(k);
      chain.push_back(m);
    } else if (lift_before[i] >= 0) {
      const int k = lift_before[i];
      PNode l; l.kind = NodeC::Lift; l.index = k; l.slot = slotOf(NodeC::Lift, k); l.t = t_lift[k]; l.dt = dt_lift[k]; l.phase = phase_before; l.level = 0;
      chain.push_back(l);
    }
    PNode nd;
    nd.kind = (i == N_ - 1) ? NodeC::Terminal : NodeC::Stage;
    nd.slot = i; nd.index = i; nd.t = ts[i]; nd.dt = dts[i]; nd.phase = phase[i];
    nd.level = (i == N_ - 1) ? N_ideal_ : i + 1;        // parnmpc_linearizer.cpp:43-58: terminal.initConstraints(robot, N_ideal, ...)
    chain.push_back(nd);
  }
  if (slice_end >= 0) {
    // a shard of the chain: the grid stages [slice_begin, slice_end) and the event stages in front of each of them
    const int lo = slice_begin, hi = std::min(slice_end, N_);
    std::vector<PNode> mine;
    int owner = 0;                                   // grid stage the current node belongs to
    for (size_t p = 0; p < chain.size(); ++p) {
      // event stages precede their grid stage: look ahead for it
      size_t g = p;
      while (chain[g].kind != NodeC::Stage && chain[g].kind != NodeC::Terminal) ++g;
      owner = chain[g].index;
      if (owner >= lo && owner < hi) mine.push_back(chain[p]);
    }
    chain.swap(mine);
    has_prev = lo > 0;
    has_terminal = hi >= N_;
  }
  discretized_ = true; disc_t_ = t;
}

bool ParNMPCSolver::componentValid(int c, const PNode& nd) const {     // constraints_data.hpp:18-42
  if (nd.kind == NodeC::Impulse) return c == 6 && (cons.linearized_impulse_friction_cone != 0 || cons.impulse_friction_cone != 0);
  if (c < 2) return cons.joint_position_limits != 0 && nd.level >= 2;
  if (c < 4) return cons.joint_velocity_limits != 0 && nd.level >= 1;
  if (c < 6) return cons.joint_torque_limits != 0;
  if (c == 7) return false;
  if (c == 8) return cons.joint_acceleration_lower_limit != 0;
  if (c == 9) return cons.joint_acceleration_upper_limit != 0;
  if (c == 10) return cons.contact_distance != 0 && nd.level >= 2;
  return cons.linearized_friction_cone != 0 || cons.friction_cone != 0;
}

void ParNMPCSolver::qRef(real t, Mat& q_ref) const {
  OCPSolver tmp_unused_guard(robot.model(), cost, cons, 1.0, 1);     // reuse the reference generator of the OCP oracle
  tmp_unused_guard.qRef(t, q_ref);
}

const SplitSolutionC* ParNMPCSolver::nextSolution(int p) const {
  if (p + 1 < (int)chain.size()) return &s[chain[p + 1].slot];
  return has_terminal ? nullptr : &next_s;
}

// BackwardCorrectionSolver::initAuxMat (backward_correction_solver.cpp:54-92): every aux_mat = terminal cost Hessian at s[N-1]
void ParNMPCSolver::initBackwardCorrection(real t) {
  discretize(t);
  const int nv = nv_;
  Mat q_ref, Jq;
  qRef(seq.numEvents() == 0 ? t + (stage_offset + N_) * dt_ : chain.back().t, q_ref);
  robot.dSubtractdConfigurationPlus(s[N_ - 1].q, q_ref, Jq);
  Mat Qxx(2 * nv, 2 * nv);
  Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.qf_weight[r];
  Qxx.addBlock(0, 0, Jq.t() * WJ);
  for (int r = 0; r < nv; ++r) Qxx(nv + r, nv + r) += cost.vf_weight[r];
  for (auto& a : aux_mat) a = Qxx;
}

// SplitParNMPC::initConstraints / ImpulseSplitParNMPC::initConstraints: slack = -g(s) pushed above the barrier, dual = barrier / slack
void ParNMPCSolver::initNodeConstraints(const PNode& nd) {
  const int i = nd.slot;
  const ContactStatus& cs = nodeContacts(nd);
  ipm[i].clear();
  for (int c = 0; c < NCOMP; ++c) {
    IpmData data(componentDim(c, nd.kind == NodeC::Impulse));
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    if (componentValid(c, nd)) {
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r = 0; r < nu_; ++r) data.slack[r] = -sgn * (limitedVar(s[i], c, r, nv_, nu_) - limitOf(robot.model(), cons, c, r));
      } else if (c == 10) {
        // ContactDistance::setSlackAndDual (contact_distance.cpp:58-65): the height of every contact frame, active or not
        {
          const Mat zero(nv_);
          robot.updateKinematics(s[i].q, zero, zero);
          for (int cc = 0; cc < nc_; ++cc) { real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr); data.slack[cc] = pw[2]; }
        }
      } else {
        for (int cc = 0; cc < nc_; ++cc) {      // all contacts, active or not (linearized_friction_cone.cpp:100-108)
          const ConeEval ce = coneEval(CK, cons.mu, s[i].f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
          for (int r = 0; r < CR; ++r) data.slack[CR * cc + r] = -res[r];
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

// ParNMPCLinearizer::initConstraints (parnmpc_linearizer.cpp:43-75): stage i with time step i + 1, aux / lift with 0
void ParNMPCSolver::initConstraints(real t) {
  discretize(t);
  if (seq.numEvents() == 0) {
    for (const PNode& nd : chain) initNodeConstraints(nd);
    return;
  }
  // every grid slot (the reference initialises all N_ideal stages), then the event stages in use
  for (int i = 0; i < N_ideal_; ++i) {
    PNode nd; nd.kind = i < N_ideal_ - 1 ? NodeC::Stage : NodeC::Terminal; nd.slot = i; nd.index = i; nd.level = i + 1; nd.phase = 0;
    for (const PNode& c : chain) if ((c.kind == NodeC::Stage || c.kind == NodeC::Terminal) && c.slot == i) nd.phase = c.phase;
    initNodeConstraints(nd);
  }
  for (const PNode& nd : chain) if (nd.kind == NodeC::Impulse || nd.kind == NodeC::Aux || nd.kind == NodeC::Lift) initNodeConstraints(nd);
}

// SplitParNMPC::linearizeOCP (split_parnmpc.hxx:50-84; with the switching constraint of an aux stage :86-124) /
// TerminalParNMPC::linearizeOCP (terminal_parnmpc.hxx:50-82) and the computeKKTResidual twins.
void ParNMPCSolver::linearizeNode(int p, const Mat& q_prev, const Mat& v_prev, bool residual_only) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const PNode& nd = chain[p];
  if (nd.kind == NodeC::Impulse) { linearizeImpulse(p, q_prev, v_prev, residual_only); return; }
  const int i = nd.slot;
  const bool terminal = nd.kind == NodeC::Terminal;
  const SplitSolutionC& si = s[i];
  SplitKKTMatrixC& M = kkt_matrix[i];
  SplitKKTResidualC& R = kkt_residual[i];
  ContactDynamicsDataC& D = cd[i];
  const ContactStatus& cs = nodeContacts(nd);
  const int nv = nv_, nu = nu_, dimf = cs.dimf();
  const real dt = nd.dt, t = nd.t;
  robot.updateKinematics(si.q, si.v, si.a);
  if (!residual_only) {
    M.Qxx.setZero(); M.Qxu_full.setZero(); M.Quu_full.setZero(); M.Qaa_diag.setZero(); M.Qff = Mat(dimf, dimf);
    M.Fvq.setZero(); M.Fvv.setZero(); M.Fvu.setZero();
  }
  R.Fq.setZero(); R.Fv.setZero(); R.lq.setZero(); R.lv.setZero(); R.la.setZero(); R.lf = Mat(dimf); R.lu.setZero(); R.lu_passive.setZero();
  R.P = Mat(0);
  // ---- cost (stage + terminal on the last stage)
  Mat q_ref, qdiff, Jq;
  qRef(t, q_ref);
  robot.subtractConfiguration(si.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(si.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = cost.q_weight[r] * qdiff[r];
  R.lq += dt * (Jq.t() * Wq);
  const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  const real vs = vRefScale(cost, t);      // TimeVaryingConfigurationSpaceCost::v_ref(t)
  for (int r = 0; r < nv; ++r) {
    R.lv[r] += dt * cost.v_weight[r] * (si.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]));
    R.la[r] += dt * cost.a_weight[r] * si.a[r];
  }
  for (int r = 0; r < nu; ++r) R.lu[r] += dt * cost.u_weight[r] * (si.u[r] - cost.u_ref[r]);
  Mat task_H, task_Hf;
  if (cost.task_dim) {
    real c_; Mat g_;
    taskTerms(task_robot, cost, t, cost.task_weight, si.q, c_, g_, task_H); R.lq += dt * g_;
    if (terminal) { taskTerms(task_robot, cost, t, cost.task_weightf, si.q, c_, g_, task_Hf); R.lq += g_; }
  }
  {
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
      for (int r = 0; r < 3; ++r) R.lf[st + r] += dt * cost.f_weight[c][r] * (si.f[c][r] - cost.f_ref[c][r]);
      st += 3;
    }
  }
  if (terminal) {
    Mat Wf(nv); for (int r = 0; r < nv; ++r) Wf[r] = cost.qf_weight[r] * qdiff[r];
    R.lq += Jq.t() * Wf;
    for (int r = 0; r < nv; ++r) R.lv[r] += cost.vf_weight[r] * (si.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]));
  }
  // ---- constraints
  for (int c = 0; c < NCOMP; ++c) {
    if (!componentValid(c, nd)) continue;
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    IpmData& data = ipm[i][c];
    if (jointComp(c)) {
      const real sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : (c < 6 ? R.lu : R.la));
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        if (residual_only) {
          data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), cons, c, r)) + data.slack[r];
          data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        }
        l[off + r] += sgn * dt * data.dual[r];
      }
    } else if (c == 10) {
      // ContactDistance::augmentDualResidual (contact_distance.cpp:68-78) [+ computePrimalAndDualResidual :132-146]: the contacts that are
      // NOT active keep their frame above the ground; J = LOCAL frame Jacobian (Robot::getFrameJacobian), its row 2
      if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
      robot.updateKinematics(si.q, si.v, si.a);
      cd_J[i] = Mat(nc_, nv);
      for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
        Mat vdq, adq, adv, J;
        robot.frameDerivatives(cc, vdq, adq, adv, J);
        for (int col = 0; col < nv; ++col) { cd_J[i](cc, col) = J(2, col); R.lq[col] -= dt * data.dual[cc] * J(2, col); }
        if (residual_only) {
          real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr);
          data.residual[cc] = -pw[2] + data.slack[cc];
          data.duality[cc] = data.slack[cc] * data.dual[cc] - cons.barrier;
        }
      }
    } else {
      if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J;
        if (residual_only) {
          for (int r = 0; r < CR; ++r) {
            data.residual[CR * cc + r] = res[r] + data.slack[CR * cc + r];
            data.duality[CR * cc + r] = data.slack[CR * cc + r] * data.dual[CR * cc + r] - cons.barrier;
          }
        }
        for (int x = 0; x < 3; ++x) for (int r = 0; r < CR; ++r) R.lf[st + x] += dt * Jc[r][x] * data.dual[CR * cc + r];
        st += 3;
      }
    }
  }
  // ---- state equation: linearizeBackwardEuler[Terminal] (state_equation.hxx:96-147, 222-233)
  Mat diff; robot.subtractConfiguration(q_prev, si.q, diff);
  for (int r = 0; r < nv; ++r) { R.Fq[r] = diff[r] + dt * si.v[r]; R.Fv[r] = v_prev[r] - si.v[r] + dt * si.a[r]; }
  Mat Fqq; robot.dSubtractdConfigurationMinus(q_prev, si.q, Fqq);
  M.Fqq6 = Fqq.block(0, 0, kP, kP);
  {
    Mat t1 = M.Fqq6.t() * si.lmd.segment(0, kP);
    if (!terminal) {
      const SplitSolutionC& sn = *nextSolution(p);
      Mat Fqq_next; robot.dSubtractdConfigurationPlus(si.q, sn.q, Fqq_next);
      M.Fqq_prev6 = Fqq_next.block(0, 0, kP, kP);
      t1 += M.Fqq_prev6.t() * sn.lmd.segment(0, kP);
      for (int r = kP; r < nv; ++r) R.lq[r] += sn.lmd[r] - si.lmd[r];
      for (int r = 0; r < nv; ++r) R.lv[r] += dt * si.lmd[r] - si.gmm[r] + sn.gmm[r];
    } else {
      for (int r = kP; r < nv; ++r) R.lq[r] -= si.lmd[r];
      for (int r = 0; r < nv; ++r) R.lv[r] += dt * si.lmd[r] - si.gmm[r];
    }
    for (int r = 0; r < kP; ++r) R.lq[r] += t1[r];
    for (int r = 0; r < nv; ++r) R.la[r] += dt * si.gmm[r];
  }
  if (!residual_only) {
    // condenseBackwardEuler (state_equation.hxx:149-170)
    Mat Fp; robot.dSubtractdConfigurationPlus(q_prev, si.q, Fp);
    Robot::dSubtractdConfigurationInverse(Fp.block(0, 0, kP, kP), M.Fqq_inv);
    M.Fqq_prev6 = M.Fqq6;
    R.Fq_prev = R.Fq.segment(0, kP);
    M.Fqq6 = M.Fqq_inv * M.Fqq_prev6;
    M.Fqv6 = dt * M.Fqq_inv;
    R.Fq.setSegment(0, M.Fqq_inv * R.Fq_prev);
  }
  // ---- ContactDynamics::linearizeContactDynamics
  robot.setContactForces(cs.active, si.f);
  Mat ID_full, dIDdq, dIDdv, C, dCdq, dCdv;
  robot.RNEA(si.q, si.v, si.a, ID_full);
  for (int r = 0; r < nu; ++r) ID_full[kP + r] -= si.u[r];
  robot.RNEADerivatives(si.q, si.v, si.a, dIDdq, dIDdv, D.dIDda);
  robot.computeBaumgarteResidual(cs.active, dt_, cs.points, C);
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
  for (int r = 0; r < kP; ++r) R.lu_passive[r] = dt * si.nu_passive[r] - dt * si.beta[r];
  for (int r = 0; r < nu; ++r) R.lu[r] -= dt * si.beta[kP + r];
  if (dimf > 0) {
    R.lq += dt * (dCdq.t() * mu_stack);
    R.lv += dt * (dCdv.t() * mu_stack);
    R.la += dt * (D.dCda.t() * mu_stack);
  }
  // ---- switchingconstraint::linearizeSwitchingConstraint (switching_constraint.hxx:8-21) on the aux stage: the feet that
  //      touch down at the impulse are already on their contact points
  if (nd.kind == NodeC::Aux) {
    const ContactStatus& is = seq.impulse_status[nd.event];
    const int dimi = is.dimf();
    robot.computeContactResidual(is.active, is.points, R.P);
    robot.computeContactDerivative(is.active, sw_Pq[i]);
    R.lq += sw_Pq[i].t() * si.xi.segment(0, dimi);
  }
  if (residual_only) return;
  // ---- cost Hessian
  {
    Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.q_weight[r];
    M.Qxx.addBlock(0, 0, Jq.t() * WJ, dt);
    for (int r = 0; r < nv; ++r) { M.Qxx(nv + r, nv + r) += dt * cost.v_weight[r]; M.Qaa_diag[r] += dt * cost.a_weight[r]; }
    for (int r = 0; r < nu; ++r) M.Quu_full(kP + r, kP + r) += dt * cost.u_weight[r];
    if (cost.task_dim) { M.Qxx.addBlock(0, 0, task_H, dt); if (terminal) M.Qxx.addBlock(0, 0, task_Hf); }
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) { for (int r = 0; r < 3; ++r) M.Qff(st + r, st + r) += dt * cost.f_weight[c][r]; st += 3; }
    if (terminal) {
      Mat WJf = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJf(r, c) *= cost.qf_weight[r];
      M.Qxx.addBlock(0, 0, Jq.t() * WJf);
      for (int r = 0; r < nv; ++r) M.Qxx(nv + r, nv + r) += cost.vf_weight[r];
    }
  }
  // ---- Constraints::condenseSlackAndDual
  for (int c = 0; c < NCOMP; ++c) {
    if (!componentValid(c, nd)) continue;
    const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
    IpmData& data = ipm[i][c];
    if (jointComp(c)) {
      const real sgn = (c & 1) ? 1.0 : -1.0;
      Mat& l = c < 2 ? R.lq : (c < 4 ? R.lv : (c < 6 ? R.lu : R.la));
      const int off = l.size() - nu;
      for (int r = 0; r < nu; ++r) {
        const real h = dt * data.dual[r] / data.slack[r];
        if (c < 2) M.Qxx(kP + r, kP + r) += h;
        else if (c < 4) M.Qxx(nv + kP + r, nv + kP + r) += h;
        else if (c < 6) M.Quu_full(kP + r, kP + r) += h;
        else M.Qaa_diag[kP + r] += h;
        data.residual[r] = sgn * (limitedVar(si, c, r, nv, nu) - limitOf(robot.model(), cons, c, r)) + data.slack[r];
        data.duality[r] = data.slack[r] * data.dual[r] - cons.barrier;
        l[off + r] += sgn * dt * (data.dual[r] * data.residual[r] - data.duality[r]) / data.slack[r];
      }
    } else if (c == 10) {
      // ContactDistance::condenseSlackAndDual (contact_distance.cpp:81-102); the kinematics of s.q (the reference condenses the
      // constraints before it moves the robot to the predicted configuration of the switching constraint, split_ocp.hxx:124-129)
      data.residual.setZero(); data.duality.setZero();
      robot.updateKinematics(si.q, si.v, si.a);
      for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
        const Mat& Jm = cd_J[i];
        const real h = dt * data.dual[cc] / data.slack[cc];
        for (int r2 = 0; r2 < nv; ++r2) for (int c2 = 0; c2 < nv; ++c2) M.Qxx(r2, c2) += h * Jm(cc, r2) * Jm(cc, c2);
        real pw[3]; robot.contactFrame(cc, pw, nullptr, nullptr, nullptr);
        data.residual[cc] = -pw[2] + data.slack[cc];
        data.duality[cc] = data.slack[cc] * data.dual[cc] - cons.barrier;
        const real g = dt * (data.dual[cc] * data.residual[cc] - data.duality[cc]) / data.slack[cc];
        for (int col = 0; col < nv; ++col) R.lq[col] -= g * Jm(cc, col);
      }
    } else {
      data.residual.setZero(); data.duality.setZero();
      int st = 0;
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
        real rr[5], dd[5];
        for (int r = 0; r < CR; ++r) {
          const int idx = CR * cc + r;
          data.residual[idx] = res[r] + data.slack[idx];
          data.duality[idx] = data.slack[idx] * data.dual[idx] - cons.barrier;
          rr[r] = (data.dual[idx] * data.residual[idx] - data.duality[idx]) / data.slack[idx];
          dd[r] = data.dual[idx] / data.slack[idx];
        }
        for (int x = 0; x < 3; ++x) {
          for (int r = 0; r < CR; ++r) R.lf[st + x] += dt * Jc[r][x] * rr[r];
          for (int y = 0; y < 3; ++y) { real acc = 0; for (int r = 0; r < CR; ++r) acc += Jc[r][x] * dd[r] * Jc[r][y]; M.Qff(st + x, st + y) += dt * acc; }
        }
        st += 3;
      }
    }
  }
  // ---- ContactDynamics::condenseContactDynamics(..., is_forward_euler = false) (contact_dynamics.hxx:105-158)
  if (keep_uncondensed) {      // test hook (ocp.hpp UncondensedC; tests/golden/gen_golden_kkt_parnmpc.py): the un-condensed Newton system of this backward-Euler stage
    if ((int)unc.size() != nslots()) unc.assign(nslots(), UncondensedC());
    UncondensedC& U = unc[i];
    U = UncondensedC();
    U.valid = true; U.kind = (int)nd.kind; U.dimf = dimf; U.dimi = (int)R.P.size(); U.has_u = 1; U.dt = dt; U.dtq = dt;
    U.active_mask = 0; for (int c = 0; c < nc_; ++c) if (cs.active[c]) U.active_mask |= 1 << c;
    U.Qxx = M.Qxx; U.Qaa = M.Qaa_diag; U.Qff = M.Qff; U.Quu = M.Quu_full;
    U.lq = R.lq; U.lv = R.lv; U.la = R.la; U.lf = R.lf; U.lu = R.lu; U.lu_passive = R.lu_passive;
    U.Fq = R.Fq; U.Fq.setSegment(0, R.Fq_prev);          // the residual before condenseBackwardEuler premultiplied its base rows
    U.Fv = R.Fv;
    U.Fqq = M.Fqq_prev6;                                 // dSubtractdConfigurationMinus(q_prev, q): d Fq / d q of THIS stage (parked there by condenseBackwardEuler)
    U.dIDCdqv = D.dIDCdqv; U.M = D.dIDda; U.J = D.dCda; U.IDC = D.IDC;
    if (nd.kind == NodeC::Aux) { U.Phix = sw_Pq[i]; U.P = R.P; }
  }
  FLOP_REGION_SET(R_CONDENSE);
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
    for (int r = 0; r < kP; ++r) R.lu_passive[r] += t1[r];
    for (int r = 0; r < nu; ++r) R.lu[r] += t1[kP + r];
  }
  M.Fvq = (-dt) * D.MJtJinv_dIDCdqv.block(0, 0, nv, nv);
  M.Fvv = (-dt) * D.MJtJinv_dIDCdqv.block(0, nv, nv, nv) - Mat::Identity(nv);
  M.Fvu = dt * D.MJtJinv.block(0, kP, nv, nu);
  for (int r = 0; r < nv; ++r) R.Fv[r] -= dt * D.MJtJinv_IDC[r];
}

// ImpulseSplitParNMPC::linearizeOCP / computeKKTResidual (impulse_split_parnmpc.hxx:33-60, 96-113):
// impulse cost + impulse friction cone, linearizeImpulseBackwardEuler / condenseImpulseBackwardEuler
// (impulse_state_equation.hxx:59-111), ImpulseDynamicsBackwardEuler::linearizeImpulseDynamics / condenseImpulseDynamics
// (impulse_dynamics_backward_euler.hxx:20-97).  Stored in the regular containers: a = dv, R.la = ldv, R.P = V,
// M.Qff, M.Qxx; the impulse-only blocks in imp[slot].
void ParNMPCSolver::linearizeImpulse(int p, const Mat& q_prev, const Mat& v_prev, bool residual_only) {
  FLOP_REGION(R_COST_CONSTRAINTS);
  const PNode& nd = chain[p];
  const int i = nd.slot;
  const SplitSolutionC& si = s[i];
  SplitKKTMatrixC& M = kkt_matrix[i];
  SplitKKTResidualC& R = kkt_residual[i];
  ImpulseDataC& I = imp[i];
  const ContactStatus& is = seq.impulse_status[nd.event];
  const int nv = nv_, dimf = is.dimf();
  robot.updateKinematics(si.q, si.v, Mat(nv));
  R.Fq.setZero(); R.Fv.setZero(); R.lq.setZero(); R.lv.setZero(); R.la.setZero(); R.lf = Mat(dimf); R.lu.setZero(); R.lu_passive.setZero();
  if (!residual_only) { M.Qxx.setZero(); M.Qff = Mat(dimf, dimf); I.Qdvdv = Mat(nv); I.Qqf = Mat(nv, dimf); }
  // ---- impulse cost (computeImpulseCostDerivatives of the cost components: no dt)
  Mat q_ref, qdiff, Jq;
  qRef(nd.t, q_ref);
  robot.subtractConfiguration(si.q, q_ref, qdiff);
  robot.dSubtractdConfigurationPlus(si.q, q_ref, Jq);
  Mat Wq(nv); for (int r = 0; r < nv; ++r) Wq[r] = cost.qi_weight[r] * qdiff[r];
  R.lq += Jq.t() * Wq;
  const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
  const real vs = vRefScale(cost, nd.t);
  for (int r = 0; r < nv; ++r) {
    R.lv[r] += cost.vi_weight[r] * (si.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]));
    R.la[r] += cost.dvi_weight[r] * si.a[r];
  }
  Mat task_Hi;      // TaskSpace3D / 6D cost on the impulse stage (computeImpulseCostDerivatives: its impulse weights, no dt)
  if (cost.task_dim) { real c_; Mat g_; taskTerms(task_robot, cost, nd.t, cost.task_weighti, si.q, c_, g_, task_Hi); R.lq += g_; }
  {
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (is.active[c]) {
      for (int r = 0; r < 3; ++r) R.lf[st + r] += cost.fi_weight[c][r] * (si.f[c][r] - cost.fi_ref[c][r]);
      st += 3;
    }
  }
  // ---- impulse friction cone: augmentDualResidual
  const bool cone = componentValid(6, nd);
  const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
  if (cone) {
    IpmData& data = ipm[i][6];
    if (residual_only) { data.residual.setZero(); data.duality.setZero(); }
    int st = 0;
    for (int cc = 0; cc < nc_; ++cc) if (is.active[cc]) {
      const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J;
      if (residual_only) {
        for (int r = 0; r < CR; ++r) {
          data.residual[CR * cc + r] = res[r] + data.slack[CR * cc + r];
          data.duality[CR * cc + r] = data.slack[CR * cc + r] * data.dual[CR * cc + r] - cons.barrier;
        }
      }
      for (int x = 0; x < 3; ++x) for (int r = 0; r < CR; ++r) R.lf[st + x] += Jc[r][x] * data.dual[CR * cc + r];
      st += 3;
    }
  }
  // ---- linearizeImpulseBackwardEuler (impulse_state_equation.hxx:59-85)
  const SplitSolutionC& sn = *nextSolution(p);
  Mat diff; robot.subtractConfiguration(q_prev, si.q, diff);
  for (int r = 0; r < nv; ++r) { R.Fq[r] = diff[r]; R.Fv[r] = v_prev[r] - si.v[r] + si.a[r]; }
  Mat Fqq; robot.dSubtractdConfigurationMinus(q_prev, si.q, Fqq);
  M.Fqq6 = Fqq.block(0, 0, kP, kP);
  {
    Mat Fqq_next; robot.dSubtractdConfigurationPlus(si.q, sn.q, Fqq_next);
    M.Fqq_prev6 = Fqq_next.block(0, 0, kP, kP);
    Mat t1 = M.Fqq_prev6.t() * sn.lmd.segment(0, kP);
    t1 += M.Fqq6.t() * si.lmd.segment(0, kP);
    for (int r = 0; r < kP; ++r) R.lq[r] += t1[r];
    for (int r = kP; r < nv; ++r) R.lq[r] += sn.lmd[r] - si.lmd[r];
    for (int r = 0; r < nv; ++r) { R.lv[r] += -si.gmm[r] + sn.gmm[r]; R.la[r] += si.gmm[r]; }
  }
  if (!residual_only) {
    // condenseImpulseBackwardEuler (:86-111)
    Mat Fp; robot.dSubtractdConfigurationPlus(q_prev, si.q, Fp);
    Robot::dSubtractdConfigurationInverse(Fp.block(0, 0, kP, kP), M.Fqq_inv);
    M.Fqq_prev6 = M.Fqq6;
    R.Fq_prev = R.Fq.segment(0, kP);
    M.Fqq6 = M.Fqq_inv * M.Fqq_prev6;
    R.Fq.setSegment(0, M.Fqq_inv * R.Fq_prev);
  }
  // ---- ImpulseDynamicsBackwardEuler::linearizeImpulseDynamics (:20-58)
  robot.setContactForces(is.active, si.f);
  {
    const Mat zero(nv);
    Mat dIDdv_unused;
    robot.RNEA(si.q, zero, si.a, I.ImD, false);                               // RNEAImpulse (robot.hxx:505-517)
    robot.RNEADerivatives(si.q, zero, si.a, I.dImDdq, dIDdv_unused, I.dImDddv, false);
  }
  robot.computeImpulseVelocityResidual(is.active, R.P);
  robot.computeImpulseVelocityDerivatives(is.active, I.Vq, I.Vv);
  const Mat mu_stack = si.mu_stack(is);
  R.lq += I.dImDdq.t() * si.beta;
  R.la += I.dImDddv.t() * si.beta;
  if (dimf > 0) {
    R.lf -= I.Vv * si.beta;
    R.lq += I.Vq.t() * mu_stack;
    R.lv += I.Vv.t() * mu_stack;
  }
  if (residual_only) return;
  // ---- impulse cost Hessian
  {
    Mat WJ = Jq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) WJ(r, c) *= cost.qi_weight[r];
    M.Qxx.addBlock(0, 0, Jq.t() * WJ);
    if (cost.task_dim) M.Qxx.addBlock(0, 0, task_Hi);
    for (int r = 0; r < nv; ++r) { M.Qxx(nv + r, nv + r) += cost.vi_weight[r]; I.Qdvdv[r] += cost.dvi_weight[r]; }
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (is.active[c]) { for (int r = 0; r < 3; ++r) M.Qff(st + r, st + r) += cost.fi_weight[c][r]; st += 3; }
  }
  // ---- condenseSlackAndDual of the impulse friction cone
  if (cone) {
    IpmData& data = ipm[i][6];
    data.residual.setZero(); data.duality.setZero();
    int st = 0;
    for (int cc = 0; cc < nc_; ++cc) if (is.active[cc]) {
      const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
      real rr[5], dd[5];
      for (int r = 0; r < CR; ++r) {
        const int idx = CR * cc + r;
        data.residual[idx] = res[r] + data.slack[idx];
        data.duality[idx] = data.slack[idx] * data.dual[idx] - cons.barrier;
        rr[r] = (data.dual[idx] * data.residual[idx] - data.duality[idx]) / data.slack[idx];
        dd[r] = data.dual[idx] / data.slack[idx];
      }
      for (int x = 0; x < 3; ++x) {
        for (int r = 0; r < CR; ++r) R.lf[st + x] += Jc[r][x] * rr[r];
        for (int y = 0; y < 3; ++y) { real acc = 0; for (int r = 0; r < CR; ++r) acc += Jc[r][x] * dd[r] * Jc[r][y]; M.Qff(st + x, st + y) += acc; }
      }
      st += 3;
    }
  }
  if (keep_uncondensed) {      // test hook (tests/golden/gen_golden_kkt_parnmpc_events.py): the un-condensed Newton system of this impulse stage
    if ((int)unc.size() != nslots()) unc.assign(nslots(), UncondensedC());
    UncondensedC& U = unc[i];
    U = UncondensedC();
    U.valid = true; U.kind = (int)nd.kind; U.dimf = dimf; U.dimi = dimf; U.has_u = 0; U.dt = 1; U.dtq = 0;
    U.active_mask = 0; for (int c = 0; c < nc_; ++c) if (is.active[c]) U.active_mask |= 1 << c;
    U.Qxx = M.Qxx; U.Qaa = I.Qdvdv; U.Qff = M.Qff;
    U.lq = R.lq; U.lv = R.lv; U.la = R.la; U.lf = R.lf;
    U.Fq = R.Fq; U.Fq.setSegment(0, R.Fq_prev);          // q_prev (-) q before condenseImpulseBackwardEuler premultiplied its base rows
    U.Fv = R.Fv;                                         // v_prev - v + dv
    U.Fqq = M.Fqq_prev6;                                 // dSubtractdConfigurationMinus(q_prev, q)
    U.dIDCdqv = Mat(nv + dimf, 2 * nv);                  // [d ImD / dq, 0; Vq, Vv]
    U.dIDCdqv.setBlock(0, 0, I.dImDdq);
    if (dimf > 0) { U.dIDCdqv.setBlock(nv, 0, I.Vq); U.dIDCdqv.setBlock(nv, nv, I.Vv); }
    U.M = I.dImDddv; U.J = I.Vv;
    U.IDC = Mat(nv + dimf); U.IDC.setSegment(0, I.ImD); if (dimf > 0) U.IDC.setSegment(nv, R.P);
  }
  // ---- condenseImpulseDynamics (:59-97)
  {
    LLT lltM;
    if (!lltM.compute(I.dImDddv)) throw std::runtime_error("ParNMPC: mass matrix not positive definite on an impulse stage");
    I.Minv = lltM.solve(Mat::Identity(nv));                                   // Robot::computeMinv
  }
  I.Fvq = -1.0 * (I.Minv * I.dImDdq);
  I.Fvf = I.Minv * I.Vv.t();
  I.Minv_ImD = I.Minv * I.ImD;
  I.Qdvq = I.Fvq; for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) I.Qdvq(r, c) *= I.Qdvdv[r];
  I.Qdvf = I.Fvf; for (int c = 0; c < dimf; ++c) for (int r = 0; r < nv; ++r) I.Qdvf(r, c) *= I.Qdvdv[r];
  I.ldv = R.la;
  for (int r = 0; r < nv; ++r) I.ldv[r] -= I.Qdvdv[r] * I.Minv_ImD[r];
  M.Qxx.addBlock(0, 0, I.Fvq.t() * I.Qdvq);
  I.Qqf += I.Fvq.t() * I.Qdvf;
  M.Qff += I.Fvf.t() * I.Qdvf;
  R.lq += I.Fvq.t() * I.ldv;
  R.lf += I.Fvf.t() * I.ldv;
  for (int r = 0; r < nv; ++r) R.Fv[r] -= I.Minv_ImD[r];
}

// BackwardCorrectionSolver::coarseUpdate (backward_correction_solver.cpp:95-250) -> SplitBackwardCorrection::coarseUpdate
// (split_backward_correction.hxx:30-72) / ImpulseSplitBackwardCorrection::coarseUpdate (impulse_split_backward_correction.hxx:30-56)
// -> SplitKKTMatrixInverter::invert (split_kkt_matrix_inverter.hxx:44-80; with Pq :110-166) /
// ImpulseSplitKKTMatrixInverter::invert (impulse_split_kkt_matrix_inverter.hxx:34-80).  One code path: the stage's KKT matrix is
//   [ 0  J ; J^T  Q ]  with  J = [F ; C] (C = Pq rows of an aux stage / V rows of an impulse stage) and Q over (w, q, v),
// w = u (regular, aux, lift, terminal) or f (impulse).
void ParNMPCSolver::coarseUpdate(real t, const Mat& q, const Mat& v) {
  FLOP_REGION(R_KKT_INVERSE);
  discretize(t);
  const int nv = nv_, nu = nu_, nx = 2 * nv, Mc = (int)chain.size();
  for (int p = 0; p < Mc; ++p) {
    const PNode& nd = chain[p];
    const int i = nd.slot;
    const Mat& q_prev = p == 0 ? q : s[chain[p - 1].slot].q;
    const Mat& v_prev = p == 0 ? v : s[chain[p - 1].slot].v;
    linearizeNode(p, q_prev, v_prev, false);
    SplitKKTMatrixC& M = kkt_matrix[i];
    const SplitKKTResidualC& R = kkt_residual[i];
    if (p + 1 < Mc) M.Qxx += aux_mat[chain[p + 1].slot];
    else if (!has_terminal) M.Qxx += next_aux;
    if (keep_uncondensed && (int)unc.size() == nslots()) unc[i].aux_next = p + 1 < Mc ? aux_mat[chain[p + 1].slot] : (!has_terminal ? next_aux : Mat(nx, nx));
    const bool impulse = nd.kind == NodeC::Impulse, aux = nd.kind == NodeC::Aux;
    const int ni = (impulse || aux) ? seq.impulse_status[nd.event].dimf() : 0;
    const int nw = impulse ? ni : nu, nQ = nw + nx, nr = nx + ni, nK = nr + nQ;
    Mat Q(nQ, nQ), J(nr, nQ);
    Mat res(nK);
    if (!impulse) {
      M.Qxx.setBlock(nv, 0, M.Qxx.block(0, nv, nv, nv).t());            // Qvq = Qqv^T
      // Qss = [Quu Qux; Qxu Qxx] in the order (u, q, v); F = [0 Fqq Fqv; Fvu Fvq Fvv]
      Mat Qxu = M.Qxu_full.block(0, kP, nx, nu);
      Q.setBlock(0, 0, M.Quu_full.block(kP, kP, nu, nu)); Q.setBlock(0, nu, Qxu.t()); Q.setBlock(nu, 0, Qxu); Q.setBlock(nu, nu, M.Qxx);
      Mat Fqq = -1.0 * Mat::Identity(nv), Fqv = nd.dt * Mat::Identity(nv);
      Fqq.setBlock(0, 0, M.Fqq6); Fqv.setBlock(0, 0, M.Fqv6);
      for (int r = 0; r < kP; ++r) for (int c = kP; c < nv; ++c) { Fqq(r, c) = 0; Fqq(c, r) = 0; Fqv(r, c) = 0; Fqv(c, r) = 0; }
      J.setBlock(0, nu, Fqq); J.setBlock(0, nu + nv, Fqv);
      J.setBlock(nv, 0, M.Fvu); J.setBlock(nv, nu, M.Fvq); J.setBlock(nv, nu + nv, M.Fvv);
      if (aux) J.setBlock(nx, nu, sw_Pq[i]);
      res.setSegment(0, R.Fq); res.setSegment(nv, R.Fv);
      if (aux) res.setSegment(nx, R.P);
      res.setSegment(nr, R.lu); res.setSegment(nr + nu, R.lq); res.setSegment(nr + nu + nv, R.lv);
    } else {
      // ImpulseSplitKKTMatrix (impulse_split_kkt_matrix.hxx:53-307): Qss over (f, q, v); Jac = [0 Fqq 0; Fvf Fvq -I; 0 Vq Vv]
      const ImpulseDataC& I = imp[i];
      M.Qxx.setBlock(nv, 0, M.Qxx.block(0, nv, nv, nv).t());
      Q.setBlock(0, 0, M.Qff); Q.setBlock(0, ni, I.Qqf.t()); Q.setBlock(ni, 0, I.Qqf); Q.setBlock(ni, ni, M.Qxx);
      Mat Fqq = -1.0 * Mat::Identity(nv);
      Fqq.setBlock(0, 0, M.Fqq6);
      for (int r = 0; r < kP; ++r) for (int c = kP; c < nv; ++c) { Fqq(r, c) = 0; Fqq(c, r) = 0; }
      J.setBlock(0, ni, Fqq);
      J.setBlock(nv, 0, I.Fvf); J.setBlock(nv, ni, I.Fvq); J.setBlock(nv, ni + nv, -1.0 * Mat::Identity(nv));
      J.setBlock(nx, ni, I.Vq); J.setBlock(nx, ni + nv, I.Vv);
      res.setSegment(0, R.Fq); res.setSegment(nv, R.Fv); res.setSegment(nx, R.P);
      res.setSegment(nr, R.lf); res.setSegment(nr + ni, R.lq); res.setSegment(nr + ni + nv, R.lv);
    }
    LLT lltQ;
    if (!lltQ.compute(Q)) throw std::runtime_error("ParNMPC: Qss not positive definite at chain position " + std::to_string(p));
    Mat Qinv = lltQ.solve(Mat::Identity(nQ));
    Mat JQinv = J * Qinv;
    Mat S = J * JQinv.t();
    LLT lltS;
    if (!lltS.compute(S)) throw std::runtime_error("ParNMPC: J Qss^-1 J^T not positive definite at chain position " + std::to_string(p));
    Mat TL = -1.0 * lltS.solve(Mat::Identity(nr));
    Mat TR = -1.0 * (TL * JQinv);
    Mat BR = Qinv - TR.t() * (S * TR);
    Mat& Ki = KKT_mat_inv[i];
    Ki = Mat(nK, nK);
    Ki.setBlock(0, 0, TL); Ki.setBlock(0, nr, TR); Ki.setBlock(nr, 0, TR.t()); Ki.setBlock(nr, nr, BR);
    Mat dir = Ki * res;
    SplitSolutionC& sn = s_new[i];
    sn = s[i];
    sn.lmd = s[i].lmd - dir.segment(0, nv);
    sn.gmm = s[i].gmm - dir.segment(nv, nv);
    if (aux) for (int k2 = 0; k2 < ni; ++k2) sn.xi[k2] = s[i].xi[k2] - dir[nx + k2];
    if (impulse) {
      const ContactStatus& is = seq.impulse_status[nd.event];
      int st = 0;
      for (int c = 0; c < nc_; ++c) if (is.active[c]) {
        for (int k2 = 0; k2 < 3; ++k2) { sn.mu[c][k2] = s[i].mu[c][k2] - dir[nx + st + k2]; sn.f[c][k2] = s[i].f[c][k2] - dir[nr + st + k2]; }
        st += 3;
      }
    } else {
      sn.u = s[i].u - dir.segment(nr, nu);
    }
    Mat qn; robot.integrateConfiguration(s[i].q, dir.segment(nr + nw, nv), -1.0, qn); sn.q = qn;
    sn.v = s[i].v - dir.segment(nr + nw + nv, nv);
  }
}

// backward_correction_solver.cpp:253-287; split_backward_correction.hxx:84-95
void ParNMPCSolver::backwardCorrectionSerial() {
  FLOP_REGION(R_CORRECTION);
  const int nv = nv_, nx = 2 * nv, Mc = (int)chain.size();
  for (int p = has_terminal ? Mc - 2 : Mc - 1; p >= 0; --p) {
    const int i = chain[p].slot;
    const SplitSolutionC& sn_next = (p == Mc - 1) ? next_snew : s_new[chain[p + 1].slot];
    const SplitSolutionC& s_next = (p == Mc - 1) ? next_s : s[chain[p + 1].slot];
    const int nK = KKT_mat_inv[i].r;
    x_res[i].setSegment(0, sn_next.lmd - s_next.lmd);
    x_res[i].setSegment(nv, sn_next.gmm - s_next.gmm);
    Mat dx = KKT_mat_inv[i].block(0, nK - nx, nx, nx) * x_res[i];
    s_new[i].lmd -= dx.segment(0, nv);
    s_new[i].gmm -= dx.segment(nv, nv);
  }
}
// :288-318; split_backward_correction.hxx:96-108; impulse_split_backward_correction.hxx:70-79
void ParNMPCSolver::backwardCorrectionParallel() {
  FLOP_REGION(R_CORRECTION);
  const int nv = nv_, nu = nu_, nx = 2 * nv, Mc = (int)chain.size();
  for (int p = 0; p < (has_terminal ? Mc - 1 : Mc); ++p) {
    const PNode& nd = chain[p];
    const int i = nd.slot;
    const int nK = KKT_mat_inv[i].r;
    const bool impulse = nd.kind == NodeC::Impulse, aux = nd.kind == NodeC::Aux;
    const int ni = (impulse || aux) ? seq.impulse_status[nd.event].dimf() : 0;
    const int nw = impulse ? ni : nu;
    Mat dz = KKT_mat_inv[i].block(nx, nK - nx, nK - nx, nx) * x_res[i];       // (dxi | dmu, du | df, dq, dv)
    if (aux) for (int k2 = 0; k2 < ni; ++k2) s_new[i].xi[k2] -= dz[k2];
    if (impulse) {
      const ContactStatus& is = seq.impulse_status[nd.event];
      int st = 0;
      for (int c = 0; c < nc_; ++c) if (is.active[c]) {
        for (int k2 = 0; k2 < 3; ++k2) { s_new[i].mu[c][k2] -= dz[st + k2]; s_new[i].f[c][k2] -= dz[ni + st + k2]; }
        st += 3;
      }
    } else {
      s_new[i].u -= dz.segment(ni, nu);
    }
    Mat qn; robot.integrateConfiguration(s_new[i].q, dz.segment(ni + nw, nv), -1.0, qn); s_new[i].q = qn;
    s_new[i].v -= dz.segment(ni + nw + nv, nv);
  }
}
// :319-352; split_backward_correction.hxx:109-120
void ParNMPCSolver::forwardCorrectionSerial() {
  FLOP_REGION(R_CORRECTION);
  const int nv = nv_, nx = 2 * nv, Mc = (int)chain.size();
  for (int p = has_prev ? 0 : 1; p < Mc; ++p) {
    const int i = chain[p].slot;
    const SplitSolutionC& snp = (p == 0) ? prev_snew : s_new[chain[p - 1].slot];
    const SplitSolutionC& sp = (p == 0) ? prev_s : s[chain[p - 1].slot];
    const int nK = KKT_mat_inv[i].r;
    Mat dq; robot.subtractConfiguration(snp.q, sp.q, dq);
    x_res[i].setSegment(0, dq);
    x_res[i].setSegment(nv, snp.v - sp.v);
    Mat dx = KKT_mat_inv[i].block(nK - nx, 0, nx, nx) * x_res[i];
    Mat qn; robot.integrateConfiguration(s_new[i].q, dx.segment(0, nv), -1.0, qn); s_new[i].q = qn;
    s_new[i].v -= dx.segment(nv, nv);
  }
}
// :353-470; split_backward_correction.hxx:121-155; impulse_split_backward_correction.hxx:93-125;
// SplitParNMPC / ImpulseSplitParNMPC::computeCondensed{Primal,Dual}Direction
void ParNMPCSolver::forwardCorrectionParallel() {
  FLOP_REGION(R_CORRECTION);
  const int nv = nv_, nu = nu_, nx = 2 * nv, Mc = (int)chain.size();
  real pmin = 1, dmin = 1;
  for (int p = 0; p < Mc; ++p) {
    const PNode& nd = chain[p];
    const int i = nd.slot;
    const int nK = KKT_mat_inv[i].r;
    const bool impulse = nd.kind == NodeC::Impulse, aux = nd.kind == NodeC::Aux;
    const int ni = (impulse || aux) ? seq.impulse_status[nd.event].dimf() : 0;
    const ContactStatus& cs = nodeContacts(nd);
    const int dimf = cs.dimf();
    const real dt = nd.dt;
    if (p > 0 || has_prev) {
      Mat dh = KKT_mat_inv[i].block(0, 0, nK - nx, nx) * x_res[i];             // (dlmd, dgmm, dxi | dmu, du | df)
      s_new[i].lmd -= dh.segment(0, nv);
      s_new[i].gmm -= dh.segment(nv, nv);
      if (aux) for (int k2 = 0; k2 < ni; ++k2) s_new[i].xi[k2] -= dh[nx + k2];
      if (impulse) {
        int st = 0;
        for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
          for (int k2 = 0; k2 < 3; ++k2) { s_new[i].mu[c][k2] -= dh[nx + st + k2]; s_new[i].f[c][k2] -= dh[nx + ni + st + k2]; }
          st += 3;
        }
      } else {
        s_new[i].u -= dh.segment(nx + ni, nu);
      }
    }
    aux_mat[i] = -1.0 * KKT_mat_inv[i].block(0, 0, nx, nx);
    // computeDirection
    d[i].dlmd = s_new[i].lmd - s[i].lmd;
    d[i].dgmm = s_new[i].gmm - s[i].gmm;
    robot.subtractConfiguration(s_new[i].q, s[i].q, d[i].dq);
    d[i].dv = s_new[i].v - s[i].v;
    Mat dx(nx); dx.setSegment(0, d[i].dq); dx.setSegment(nv, d[i].dv);
    SplitKKTMatrixC& M = kkt_matrix[i];
    SplitKKTResidualC& R = kkt_residual[i];
    if (impulse) {
      ImpulseDataC& I = imp[i];
      Mat df(dimf), dmu(dimf);
      int st = 0;
      for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
        for (int k2 = 0; k2 < 3; ++k2) { df[st + k2] = s_new[i].f[c][k2] - s[i].f[c][k2]; dmu[st + k2] = s_new[i].mu[c][k2] - s[i].mu[c][k2]; }
        st += 3;
      }
      // ImpulseDynamicsBackwardEuler::computeCondensedPrimalDirection (:98-104)
      Mat ddv = -1.0 * I.Minv_ImD;
      ddv += I.Fvq * d[i].dq;
      if (dimf > 0) ddv += I.Fvf * df;
      d[i].daf = Mat(nv + dimf); d[i].daf.setSegment(0, ddv); d[i].daf.setSegment(nv, df);
      if (componentValid(6, nd)) {
        const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
        IpmData& data = ipm[i][6];
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }
        st = 0;
        for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
          const ConeEval ce = coneEval(CK, cons.mu, s[i].f[cc]); const real (*Jc)[3] = ce.J;     // (FrictionCone: data.r[i] of the linearisation)
          for (int r2 = 0; r2 < CR; ++r2) {
            const int idx = CR * cc + r2;
            real Jdf = 0; for (int x = 0; x < 3; ++x) Jdf += Jc[r2][x] * df[st + x];
            data.dslack[idx] = -Jdf - data.residual[idx];
            data.ddual[idx] = -(data.dual[idx] * data.dslack[idx] + data.duality[idx]) / data.slack[idx];
          }
          st += 3;
        }
        pmin = std::min(pmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.slack, data.dslack));
        dmin = std::min(dmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.dual, data.ddual));
      }
      // computeCondensedDualDirection (:105-111), then the costate correction (state_equation.hxx:172-181)
      I.ldv += I.Qdvq * d[i].dq;
      if (dimf > 0) I.ldv += I.Qdvf * df;
      I.ldv += d[i].dgmm;
      Mat dbeta = -1.0 * (I.Minv * I.ldv);
      d[i].dbetamu = Mat(nv + dimf); d[i].dbetamu.setSegment(0, dbeta); d[i].dbetamu.setSegment(nv, dmu);
      d[i].dlmd.setSegment(0, M.Fqq_inv.t() * d[i].dlmd.segment(0, kP));
      continue;
    }
    d[i].du = s_new[i].u - s[i].u;
    if (aux) { d[i].dxi = Mat(ni); for (int k2 = 0; k2 < ni; ++k2) d[i].dxi[k2] = s_new[i].xi[k2] - s[i].xi[k2]; }
    // primal expansion
    const ContactDynamicsDataC& D = cd[i];
    d[i].daf = -1.0 * (D.MJtJinv_dIDCdqv * dx);
    d[i].daf += D.MJtJinv.block(0, kP, nv + dimf, nu) * d[i].du;
    d[i].daf -= D.MJtJinv_IDC;
    for (int r2 = 0; r2 < dimf; ++r2) d[i].daf[nv + r2] *= -1;
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      IpmData& data = ipm[i][c];
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r2 = 0; r2 < nu; ++r2) {
          const real dxr = c < 2 ? d[i].dq[kP + r2] : (c < 4 ? d[i].dv[kP + r2] : (c < 6 ? d[i].du[r2] : d[i].daf[kP + r2]));
          data.dslack[r2] = -sgn * dxr - data.residual[r2];
          data.ddual[r2] = -(data.dual[r2] * data.dslack[r2] + data.duality[r2]) / data.slack[r2];
        }
      } else if (c == 10) {
        // ContactDistance::computeSlackAndDualDirection (contact_distance.cpp:105-129)
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }
        for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) {
          real Jdq = 0; for (int col = 0; col < nv; ++col) Jdq += cd_J[i](cc, col) * d[i].dq[col];
          data.dslack[cc] = Jdq - data.residual[cc];
          data.ddual[cc] = -(data.dual[cc] * data.dslack[cc] + data.duality[cc]) / data.slack[cc];
        }
      } else {
        for (int r2 = 0; r2 < data.dslack.size(); ++r2) { data.dslack[r2] = 1.0; data.ddual[r2] = 1.0; }
        int st = 0;
        for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
          const ConeEval ce = coneEval(CK, cons.mu, s[i].f[cc]); const real (*Jc)[3] = ce.J;     // (FrictionCone: data.r[i] of the linearisation)
          for (int r2 = 0; r2 < CR; ++r2) {
            const int idx = CR * cc + r2;
            real Jdf = 0; for (int x = 0; x < 3; ++x) Jdf += Jc[r2][x] * d[i].daf[nv + st + x];
            data.dslack[idx] = -Jdf - data.residual[idx];
            data.ddual[idx] = -(data.dual[idx] * data.dslack[idx] + data.duality[idx]) / data.slack[idx];
          }
          st += 3;
        }
      }
      pmin = std::min(pmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.slack, data.dslack));
      dmin = std::min(dmin, fractionToBoundary(cons.fraction_to_boundary_rate, data.dual, data.ddual));
    }
    // dual expansion (contact_dynamics.hxx:171-190) with the stage's OWN dgmm, then the costate correction
    // (state_equation.hxx:172-181)
    ContactDynamicsDataC& Dm = cd[i];
    d[i].dnu_passive = R.lu_passive;
    d[i].dnu_passive += M.Quu_full.block(0, kP, kP, nu) * d[i].du;
    d[i].dnu_passive += M.Qxu_full.block(0, 0, nx, kP).t() * dx;
    d[i].dnu_passive += dt * (Dm.MJtJinv.block(0, 0, kP, nv) * d[i].dgmm);
    d[i].dnu_passive = (-1.0 / dt) * d[i].dnu_passive;
    Dm.laf += Dm.Qafqv * dx;
    Dm.laf += Dm.Qafu_full.block(0, kP, nv + dimf, nu) * d[i].du;
    for (int r = 0; r < nv; ++r) Dm.laf[r] += dt * d[i].dgmm[r];
    d[i].dbetamu = (-1.0 / dt) * (Dm.MJtJinv * Dm.laf);
    d[i].dlmd.setSegment(0, M.Fqq_inv.t() * d[i].dlmd.segment(0, kP));
  }
  primal_step_size = pmin; dual_step_size = dmin;
}

// ParNMPCLinearizer::integrateSolution (parnmpc_linearizer.cpp:248-300); SplitSolution::integrate / ImpulseSplitSolution::integrate
void ParNMPCSolver::integrateSolution() {
  FLOP_REGION(R_INTEGRATE);
  const int nv = nv_;
  const real ap = primal_step_size, ad = dual_step_size;
  for (const PNode& nd : chain) {
    const int i = nd.slot;
    const ContactStatus& cs = nodeContacts(nd);
    SplitSolutionC& si = s[i];
    si.lmd += ap * d[i].dlmd;
    si.gmm += ap * d[i].dgmm;
    Mat qn; robot.integrateConfiguration(si.q, d[i].dq, ap, qn); si.q = qn;
    si.v += ap * d[i].dv;
    si.a += ap * d[i].daf.segment(0, nv);
    si.beta += ap * d[i].dbetamu.segment(0, nv);
    if (nd.kind != NodeC::Impulse) {
      si.u += ap * d[i].du;
      si.nu_passive += ap * d[i].dnu_passive;
    }
    if (nd.kind == NodeC::Aux) for (int k2 = 0; k2 < d[i].dxi.size(); ++k2) si.xi[k2] += ap * d[i].dxi[k2];
    int st = 0;
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) {
      for (int r = 0; r < 3; ++r) { si.f[c][r] += ap * d[i].daf[nv + st + r]; si.mu[c][r] += ap * d[i].dbetamu[nv + st + r]; }
      st += 3;
    }
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      ipm[i][c].slack += ap * ipm[i][c].dslack;
      ipm[i][c].dual += ad * ipm[i][c].ddual;
    }
  }
}

void ParNMPCSolver::computeDirection(real t, const Mat& q, const Mat& v) {
  coarseUpdate(t, q, v);
  auto t0 = std::chrono::steady_clock::now();
  backwardCorrectionSerial();
  serial_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  backwardCorrectionParallel();
  t0 = std::chrono::steady_clock::now();
  forwardCorrectionSerial();
  serial_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  forwardCorrectionParallel();
}
void ParNMPCSolver::updateSolution(real t, const Mat& q, const Mat& v, bool use_line_search) {
  computeDirection(t, q, v);
  if (use_line_search)            // parnmpc_solver.cpp:94-99
    primal_step_size = line_search.computeStepSize([&](real a) { return costAndViolation(a, q, v); }, primal_step_size);
  integrateSolution();
}

// LineSearch::computeSolution + computeCostAndViolation for ParNMPC (src/line_search/line_search.cpp:199-301, 346-399).  Every stage
// of the chain is evaluated against the TRIAL iterate of its chain predecessor (line_search.hpp:224-264 for the grid stages,
// line_search.cpp:240-300 for the event stages: impulse <- aux, aux / lift <- the grid stage in front of them or the measured state):
//   grid / aux / lift stage (the last grid stage is TerminalParNMPC):
//     cost      = Split / TerminalParNMPC::stageCost (split_parnmpc.hxx:269-288, terminal_parnmpc.hxx:188-207): stage cost + dt *
//                 barrier(slack + alpha dslack) -- NOT the terminal cost, which the reference leaves out of its line search
//     violation = constraintViolation (split_parnmpc.hxx:291-342, terminal_parnmpc.hxx:210-229): |Fx|_1 + dt |[ID - u; C]|_1
//                 + dt |g + slack|_1, on an aux stage + |P|_1 of the switching constraint (no dt)
//   impulse stage:
//     cost      = ImpulseSplitParNMPC::stageCost (impulse_split_parnmpc.hxx:147-163): impulse cost + barrier, no dt
//     violation = constraintViolation (:166-194): |g + slack|_1 + |Fx|_1 + |ImD|_1 + |V|_1
std::pair<real, real> ParNMPCSolver::costAndViolation(real alpha, const Mat& q, const Mat& v) {
  const int nv = nv_, nu = nu_, kP = nv - nu;
  Robot rb = robot;
  auto trial = [&](int p) {
    const int sl = chain[p].slot;
    SplitSolutionC x = s[sl];
    if (alpha > 0) {
      const ContactStatus& cs = nodeContacts(chain[p]);
      Mat qn; rb.integrateConfiguration(s[sl].q, d[sl].dq, alpha, qn); x.q = qn;
      x.v = s[sl].v + alpha * d[sl].dv;
      x.a = s[sl].a + alpha * d[sl].daf.segment(0, nv);
      if (chain[p].kind != NodeC::Impulse) x.u = s[sl].u + alpha * d[sl].du;
      int st = 0;
      for (int c = 0; c < nc_; ++c) if (cs.active[c]) { for (int k2 = 0; k2 < 3; ++k2) x.f[c][k2] = s[sl].f[c][k2] + alpha * d[sl].daf[nv + st + k2]; st += 3; }
    }
    return x;
  };
  real cost_sum = 0, viol_sum = 0;
  SplitSolutionC xp(robot);
  for (int p = 0; p < (int)chain.size(); ++p) {
    const PNode& nd = chain[p];
    const int sl = nd.slot;
    const bool impulse = nd.kind == NodeC::Impulse;
    const SplitSolutionC x = trial(p);
    const ContactStatus& cs = nodeContacts(nd);
    const real dt = impulse ? real(1.0) : nd.dt;
    Mat q_ref, qdiff;
    qRef(nd.t, q_ref);
    rb.subtractConfiguration(x.q, q_ref, qdiff);
    const real v_ref0 = cost.use_trotting_ref ? cost.step_length / cost.t_period : cost.v_ref[0];
    const real vs = vRefScale(cost, nd.t);
    real l = 0;
    for (int r = 0; r < nv; ++r) {
      const real dvr = x.v[r] - vs * (r == 0 ? v_ref0 : cost.v_ref[r]);
      if (impulse) l += cost.qi_weight[r] * qdiff[r] * qdiff[r] + cost.vi_weight[r] * dvr * dvr + cost.dvi_weight[r] * x.a[r] * x.a[r];
      else l += cost.q_weight[r] * qdiff[r] * qdiff[r] + cost.v_weight[r] * dvr * dvr + cost.a_weight[r] * x.a[r] * x.a[r];
    }
    if (!impulse) for (int r = 0; r < nu; ++r) l += cost.u_weight[r] * (x.u[r] - cost.u_ref[r]) * (x.u[r] - cost.u_ref[r]);
    if (cost.task_dim) { real c_; Mat g_, H_; taskTerms(task_robot, cost, nd.t, impulse ? cost.task_weighti : cost.task_weight, x.q, c_, g_, H_); l += 2 * c_; }      // (stage part only: no terminal cost in the merit)
    for (int c = 0; c < nc_; ++c) if (cs.active[c]) for (int k2 = 0; k2 < 3; ++k2) {
      const real w = impulse ? cost.fi_weight[c][k2] : cost.f_weight[c][k2], fr = impulse ? cost.fi_ref[c][k2] : cost.f_ref[c][k2];
      l += w * (x.f[c][k2] - fr) * (x.f[c][k2] - fr);
    }
    real barrier = 0, primal = 0;
    for (int c = 0; c < NCOMP; ++c) {
      if (!componentValid(c, nd)) continue;
      const int CK = coneKind(impulse), CR = coneRows(impulse);
      const IpmData& data = ipm[sl][c];
      for (int r = 0; r < data.slack.size(); ++r) barrier -= cons.barrier * std::log(data.slack[r] + alpha * data.dslack[r]);      // pdipm.hxx:84-87
      if (jointComp(c)) {
        const real sgn = (c & 1) ? 1.0 : -1.0;
        for (int r = 0; r < nu; ++r) primal += std::fabs(sgn * (limitedVar(x, c, r, nv, nu) - limitOf(rb.model(), cons, c, r)) + data.slack[r]);
      } else if (c == 10) {
        // ContactDistance::computePrimalAndDualResidual at the trial configuration (contact_distance.cpp:132-146)
        rb.updateKinematics(x.q, x.v, x.a);
        for (int cc = 0; cc < nc_; ++cc) if (!cs.active[cc]) { real pw[3]; rb.contactFrame(cc, pw, nullptr, nullptr, nullptr); primal += std::fabs(-pw[2] + data.slack[cc]); }
      } else {
        for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
          const ConeEval ce = coneEval(CK, cons.mu, x.f[cc]);
          for (int r = 0; r < CR; ++r) primal += std::fabs(ce.res[r] + data.slack[CR * cc + r]);
        }
      }
    }
    cost_sum += 0.5 * dt * l + dt * barrier;
    const Mat& qpv = p == 0 ? q : xp.q;
    const Mat& vpv = p == 0 ? v : xp.v;
    Mat diff; rb.subtractConfiguration(qpv, x.q, diff);
    real viol = 0;
    if (impulse) {
      // impulse_state_equation.hxx:113-124, impulse_dynamics_backward_euler.hxx:117-134
      for (int r = 0; r < nv; ++r) viol += std::fabs(diff[r]) + std::fabs(vpv[r] - x.v[r] + x.a[r]);
      Mat ImD, V;
      const Mat zero(nv);
      rb.updateKinematics(x.q, x.v, zero);
      rb.setContactForces(cs.active, x.f);
      rb.RNEA(x.q, zero, x.a, ImD, false);
      rb.computeImpulseVelocityResidual(cs.active, V);
      viol += ImD.lpNorm1() + V.lpNorm1() + primal;
    } else {
      // backward-Euler residual against the trial predecessor (state_equation.hxx:225-236)
      for (int r = 0; r < nv; ++r) viol += std::fabs(diff[r] + dt * x.v[r]) + std::fabs(vpv[r] - x.v[r] + dt * x.a[r]);
      Mat ID, C;
      rb.updateKinematics(x.q, x.v, x.a);
      rb.setContactForces(cs.active, x.f);
      rb.RNEA(x.q, x.v, x.a, ID);
      for (int r = 0; r < nu; ++r) ID[kP + r] -= x.u[r];
      rb.computeBaumgarteResidual(cs.active, dt_, cs.points, C);
      viol += dt * (ID.lpNorm1() + C.lpNorm1()) + dt * primal;
      if (nd.kind == NodeC::Aux) {
        // switchingconstraint::computeSwitchingConstraintResidual on the aux stage's own configuration (switching_constraint.hxx:24-33)
        const ContactStatus& is = seq.impulse_status[nd.event];
        Mat Pr;
        rb.computeContactResidual(is.active, is.points, Pr);
        viol += Pr.lpNorm1();
      }
    }
    viol_sum += viol;
    xp = x;
  }
  return {cost_sum, viol_sum};
}

void ParNMPCSolver::computeKKTResidual(real t, const Mat& q, const Mat& v) {
  discretize(t);
  for (int p = 0; p < (int)chain.size(); ++p)
    linearizeNode(p, p == 0 ? q : s[chain[p - 1].slot].q, p == 0 ? v : s[chain[p - 1].slot].v, true);
}

// ParNMPCLinearizer::KKTError (parnmpc_linearizer.cpp:203-247); SplitParNMPC::squaredNormKKTResidual (split_parnmpc.hxx:250-266;
// the switching-constraint residual of an aux stage included), ImpulseSplitParNMPC::squaredNormKKTResidual
// (impulse_split_parnmpc.hxx:114-124): note that the constraint residuals are NOT weighted by dt^2 here, unlike SplitOCP
// ParNMPCSolver::isCurrentSolutionFeasible (parnmpc_solver.cpp:231-273) with the component tests of joint_*_limit.cpp:36-47 and
// linearized_(impulse_)friction_cone.cpp:82-99, in the reference's order (stages incl. the terminal one, impulses, aux, lifts)
int ParNMPCSolver::isCurrentSolutionFeasible() const {
  for (int kind = NodeC::Stage; kind <= NodeC::Lift; ++kind)
    for (int p = 0; p < (int)chain.size(); ++p) {
      const auto& nd = chain[p];
      if (!(nd.kind == kind || (kind == NodeC::Stage && nd.kind == NodeC::Terminal))) continue;
      const SplitSolutionC& si = s[nd.slot];
      for (int c = 0; c < NCOMP; ++c) {
        if (!jointComp(c) || !componentValid(c, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
        for (int r = 0; r < nu_; ++r) {
          const real x = limitedVar(si, c, r, nv_, nu_), lim = limitOf(robot.model(), cons, c, r);
          if ((c & 1) ? x > lim : x < lim) return p;
        }
      }
      if (componentValid(10, nd)) {                        // ContactDistance::isFeasible (contact_distance.cpp:44-55)
        Robot rb2 = robot;
        const Mat zero(nv_);
        rb2.updateKinematics(si.q, zero, zero);
        const ContactStatus& cs2 = nodeContacts(nd);
        for (int cc = 0; cc < nc_; ++cc) if (!cs2.active[cc]) { real pw[3]; rb2.contactFrame(cc, pw, nullptr, nullptr, nullptr); if (pw[2] <= 0) return p; }
      }
      if (!componentValid(6, nd)) continue;
      const int CK = coneKind(nd.kind == NodeC::Impulse), CR = coneRows(nd.kind == NodeC::Impulse); (void)CK; (void)CR;
      const ContactStatus& cs = nodeContacts(nd);
      for (int cc = 0; cc < nc_; ++cc) if (cs.active[cc]) {
        const ConeEval ce = coneEval(CK, cons.mu, si.f[cc]); const real* res = ce.res; const real (*Jc)[3] = ce.J; (void)Jc;
        for (int r = 0; r < CR; ++r) if (res[r] > 0) return p;
      }
    }
  return -1;
}

real ParNMPCSolver::KKTError() { return std::sqrt(KKTErrorSquared()); }
real ParNMPCSolver::KKTErrorSquared() {
  real sum = 0;
  for (const PNode& nd : chain) {
    const int i = nd.slot;
    const SplitKKTResidualC& R = kkt_residual[i];
    real e;
    if (nd.kind == NodeC::Impulse) {
      e = R.lq.squaredNorm() + R.lv.squaredNorm() + R.la.squaredNorm() + R.lf.squaredNorm() + R.Fq.squaredNorm() + R.Fv.squaredNorm() +
          imp[i].ImD.squaredNorm() + R.P.squaredNorm();
    } else {
      e = R.lq.squaredNorm() + R.lv.squaredNorm() + R.la.squaredNorm() + R.lf.squaredNorm() + R.lu_passive.squaredNorm() +
          R.lu.squaredNorm() + R.Fq.squaredNorm() + R.Fv.squaredNorm() + nd.dt * nd.dt * cd[i].IDC.squaredNorm();
      if (nd.kind == NodeC::Aux) e += R.P.squaredNorm();
    }
    real c2 = 0;
    for (int c = 0; c < NCOMP; ++c) if (componentValid(c, nd)) c2 += ipm[i][c].residual.squaredNorm() + ipm[i][c].duality.squaredNorm();
    sum += e + c2;
  }
  return sum;
}

}  // namespace oracle
