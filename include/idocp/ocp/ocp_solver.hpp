// idocp::OCPSolver -- drop-in facade over the HIP contact path.
//
// Same constructor signature and methods as the reference class
// (include/idocp/ocp/ocp_solver.hpp:32-232; src/ocp/ocp_solver.cpp).  Every
// method forwards to the C ABI (include/idocp_hip.h); the arithmetic runs in
// the HIP kernels K5a/K5b/S3/S4/K6/K7/K8.  `nthreads` is accepted for source
// compatibility and ignored (the GPU replaces the OpenMP team).  Argument
// errors: message on stderr + std::exit(EXIT_FAILURE), like the reference.
//
// Contact sequences with discrete events (pushBackContactStatus: lift stages,
// impulse + aux stages, switching constraints) run on the same kernels; the
// event stages are reachable through getSolution("...", "impulse" | "aux" | "lift").
//
// A FIXED-BASE robot without contact frames (examples/iiwa14/ocp_benchmark.cpp of the reference) is bound to the kernels of
// idocp::UnOCPSolver: with no contact rows the contact-dynamics formulation (control u, the acceleration eliminated through M^-1,
// contact_dynamics.hxx:105-158) and the unconstrained one (control a, the torque eliminated through u = ID(q, v, a),
// unconstrained_dynamics.hxx:55-106) condense ONE Newton system in two orders -- same direction, step sizes, iterates and KKT error
// (oracle restatement of both: tests/test_oracle_fixed_base.py, 1e-11; GPU against the OCPSolver restatement: tests/test_fixed_base_ocp_gpu.py).
// What differs is kept: setSolution leaves the slack / dual variables alone (idocp_unocp_set_solution_only), initConstraints(t) takes a time,
// getStateFeedbackGain returns the TORQUE policy (idocp_unocp_get_torque_feedback_gain).  A fixed-base robot WITH contact frames is refused.
#ifndef IDOCP_OCP_SOLVER_HPP_
#define IDOCP_OCP_SOLVER_HPP_

#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/eigen_shim.hpp"
#include "idocp/ocp/split_solution.hpp"
#include "idocp/robot/contact_status.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/unocp/unocp_solver.hpp"
#include "idocp_hip.h"

namespace idocp {

class OCPSolver {
 public:
  OCPSolver(const Robot& robot, const std::shared_ptr<CostFunction>& cost, const std::shared_ptr<Constraints>& constraints,
            const double T, const int N, const int max_num_impulse = 0, const int nthreads = 1, const int device = 0)
      : robot_(robot), N_(N), h_(nullptr), cost_(cost) {
    (void)nthreads;
    if (max_num_impulse < 0) {      // ocp_solver.cpp:34-36
      std::cerr << "invalid value: max_num_impulse must be non-negative!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    if (!robot.hasFloatingBase() && robot.maxPointContacts() == 0) {      // (no contact frame: no discrete event can ever be pushed)
      un_.reset(new UnOCPSolver(robot, cost, constraints, T, N, nthreads, device));
      return;
    }
    const idocp_cost_t c = cost->native();
    const idocp_constraints_t k = constraints->native();
    check(idocp_ocp_create_hybrid(&robot.model(), &c, &k, T, N, max_num_impulse, 1, device, &h_));
    last_cost_ = c;
    cache_.resize(N + 1);
  }
  // ocp_solver.hpp:44: an empty solver, to be assigned a constructed one before use
  OCPSolver() : robot_(), N_(0), h_(nullptr) {}
  ~OCPSolver() { idocp_ocp_destroy(h_); }
  // copyable and movable like the reference class (ocp_solver.hpp:171-186, `= default`): a copy is a DEEP copy of the solver
  // state on the device (idocp_ocp_clone)
  OCPSolver(const OCPSolver& other) : robot_(other.robot_), N_(other.N_), h_(nullptr), cost_(other.cost_), last_cost_(other.last_cost_), cache_(other.cache_) {
    if (other.h_) check(idocp_ocp_clone(other.h_, &h_));
    if (other.un_) un_.reset(new UnOCPSolver(*other.un_));
  }
  OCPSolver& operator=(const OCPSolver& other) {
    if (this != &other) {
      idocp_ocp_t* n = nullptr; if (other.h_) check(idocp_ocp_clone(other.h_, &n)); idocp_ocp_destroy(h_); h_ = n; robot_ = other.robot_; N_ = other.N_; cache_ = other.cache_; cost_ = other.cost_; last_cost_ = other.last_cost_;
      un_.reset(other.un_ ? new UnOCPSolver(*other.un_) : nullptr);
    }
    return *this;
  }
  OCPSolver(OCPSolver&& other) noexcept : robot_(other.robot_), N_(other.N_), h_(other.h_), cost_(std::move(other.cost_)), last_cost_(other.last_cost_), cache_(std::move(other.cache_)), un_(std::move(other.un_)) { other.h_ = nullptr; }
  OCPSolver& operator=(OCPSolver&& other) noexcept {
    if (this != &other) { idocp_ocp_destroy(h_); h_ = other.h_; other.h_ = nullptr; robot_ = other.robot_; N_ = other.N_; cache_ = std::move(other.cache_); cost_ = std::move(other.cost_); last_cost_ = other.last_cost_; un_ = std::move(other.un_); }
    return *this;
  }

  void initConstraints(const double t) {
    if (un_) { un_->initConstraints(); return; }      // (stage i with time step i in both solvers: ocp_linearizer.cpp:50, unocp_solver.cpp:64)
    syncTaskRefs(t);
    check(idocp_ocp_init_constraints(h_, t));
  }

  void updateSolution(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v, const bool line_search = false) {
    if (un_) { un_->updateSolution(t, q, v, line_search); return; }
    syncCost();
    syncTaskRefs(t);
    check(idocp_ocp_update_solution(h_, t, q.data(), v.data(), line_search ? 1 : 0));
  }

  std::vector<Eigen::VectorXd> getSolution(const std::string& name) const {
    if (un_) {
      if (name == "f" || name == "mu" || name == "nu_passive") return std::vector<Eigen::VectorXd>(N_, Eigen::VectorXd(0));      // (no contact, no passive joint)
      return un_->getSolution(name);
    }
    const int dim = dimOf(name);
    const bool per_stage = !(name == "q" || name == "v" || name == "lmd" || name == "gmm");
    const int n = per_stage ? N_ : N_ + 1;
    std::vector<double> buf((size_t)(N_ + 1) * dim);
    check(idocp_ocp_get_solution(h_, name.c_str(), 0, buf.data()));
    std::vector<Eigen::VectorXd> out(n, Eigen::VectorXd(dim));
    for (int i = 0; i < n; ++i) for (int j = 0; j < dim; ++j) out[i][j] = buf[(size_t)i * dim + j];
    return out;
  }

  // ocp_solver.hpp:97: const reference to the split solution of a time stage, e.g. getSolution(0).u -- ONE device-to-host copy of
  // the stage's record (idocp_ocp_get_split_solution); the reference stays valid until the next call for the same stage
  const SplitSolution& getSolution(const int stage) const {
    if (un_) return un_->getSolution(stage);
    SplitSolution& s = cache_.at(stage);
    const int nv = robot_.dimv(), nc = robot_.maxPointContacts();
    std::vector<double> rec((size_t)5 * nv + robot_.dimq() + robot_.dimu() + 6 * nc + 6);
    check(idocp_ocp_get_split_solution(h_, 0, stage, rec.data()));
    std::vector<int> active(nc > 0 ? nc : 1, 0);
    if (idocp_ocp_get_stage_contact_status(h_, stage, active.data()) < 0) check(IDOCP_E_ARG);
    s.assign(rec.data(), robot_.dimq(), nv, robot_.dimu(), nc, robot_.dim_passive(), active.data());
    return s;
  }

  // OCPSolver::isCurrentSolutionFeasible (ocp_solver.cpp:216-248)
  bool isCurrentSolutionFeasible() {
    if (un_) return un_->isCurrentSolutionFeasible();
    int ok = 0, where = -1;
    check(idocp_ocp_is_current_solution_feasible(h_, &ok, &where));
    if (!ok) std::cout << "INFEASIBLE at stage " << where << " of the discretised horizon" << std::endl;
    return ok != 0;
  }

  // OCPSolver::getStateFeedbackGain (ocp_solver.cpp:103-113): du = Kq dq + Kv dv
  void getStateFeedbackGain(const int time_stage, Eigen::MatrixXd& Kq, Eigen::MatrixXd& Kv) const {
    const int nv = robot_.dimv(), nu = robot_.dimu();
    Kq.resize(nu, nv); Kv.resize(nu, nv);
    if (un_) { check(idocp_unocp_get_torque_feedback_gain(un_->handle(), 0, time_stage, Kq.data(), Kv.data())); return; }
    check(idocp_ocp_get_state_feedback_gain(h_, 0, time_stage, Kq.data(), Kv.data()));
  }

  void setSolution(const std::string& name, const Eigen::VectorXd& value) {
    if (un_) { check(idocp_unocp_set_solution_only(un_->handle(), name.c_str(), value.data())); return; }
    check(idocp_ocp_set_solution(h_, name.c_str(), value.data()));
  }
  void setSolution(const std::string& name, const Eigen::Vector3d& value) {
    if (un_) {                                        // ocp_solver.cpp:143-153: "f" goes to every contact -- there is none
      if (name != "f") { std::cerr << "invalid arugment: name must be q, v, a, f, or u!" << '\n'; std::exit(EXIT_FAILURE); }
      return;
    }
    check(idocp_ocp_set_solution(h_, name.c_str(), value.data()));
  }

  void setContactStatusUniformly(const ContactStatus& contact_status) {
    std::vector<int> active;
    std::vector<double> pts;
    flatten(contact_status, active, pts);
    if (un_) return;                                  // a status without contacts
    check(idocp_ocp_set_contact_status_uniformly(h_, active.data(), pts.data()));
  }
  // ocp_solver.cpp:174-197
  void pushBackContactStatus(const ContactStatus& contact_status, const double switching_time) {
    std::vector<int> active;
    std::vector<double> pts;
    flatten(contact_status, active, pts);
    if (un_) {                                        // a status without contacts after a status without contacts (contact_sequence.hxx:69-72)
      std::cerr << "discrete_event.existDiscreteEvent() must be true!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    check(idocp_ocp_push_back_contact_status(h_, active.data(), pts.data(), switching_time));
  }
  void setContactPoints(const int contact_phase, const std::vector<Eigen::Vector3d>& contact_points) {
    std::vector<double> pts(3 * contact_points.size());
    for (size_t c = 0; c < contact_points.size(); ++c) for (int k = 0; k < 3; ++k) pts[3 * c + k] = contact_points[c][k];
    if (un_) return;
    check(idocp_ocp_set_contact_points(h_, contact_phase, pts.data()));
  }
  void popBackContactStatus() { if (!un_) check(idocp_ocp_pop_back_contact_status(h_)); }
  void popFrontContactStatus() { if (!un_) check(idocp_ocp_pop_front_contact_status(h_)); }
  void clearLineSearchFilter() { if (un_) un_->clearLineSearchFilter(); else check(idocp_ocp_clear_line_search_filter(h_)); }      // ocp_solver.cpp:196-199

  double KKTError() {
    if (un_) return un_->KKTError();
    double e = 0;
    check(idocp_ocp_kkt_error(h_, &e));
    return e;
  }
  void computeKKTResidual(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v) {
    if (un_) { un_->computeKKTResidual(t, q, v); return; }
    syncCost();
    syncTaskRefs(t);
    check(idocp_ocp_compute_kkt_residual(h_, t, q.data(), v.data()));
  }
  idocp_ocp_t* handle() { return h_; }
  idocp_ocp_t* handle() const { return h_; }
  // the solver a fixed-base robot without contacts is bound to (nullptr otherwise); handle() is null then
  UnOCPSolver* unconstrainedSolver() { return un_.get(); }

 private:
  Robot robot_;
  int N_;
  idocp_ocp_t* h_;
  // The reference's solver SHARES the cost function with the driver (ocp_solver.hpp:37-39): a reference or weight the driver changes
  // between two calls takes effect at the next one.  The device holds a copy; it is refreshed when the shared object has changed.
  std::shared_ptr<CostFunction> cost_;
  idocp_cost_t last_cost_{};
  // TimeVaryingTaskSpace3DCost / 6DCost on the floating base: the reference object is asked for its pose at the time of every stage of the
  // chain (time_varying_task_space_{3d,6d}_cost.cpp) -- up front, in front of every call that takes t
  void syncTaskRefs(const double t) {
    if (!cost_ || !h_ || !cost_->native().task_time_varying) return;
    if (chain_times_.size() < 4096) chain_times_.resize(4096);
    const int M = idocp_ocp_get_chain_times(h_, t, (int)chain_times_.size(), chain_times_.data());
    if (M <= 0) check(M < 0 ? M : IDOCP_E_ARG);
    const std::vector<double> times(chain_times_.begin(), chain_times_.begin() + M);
    if (cost_->taskRefsAt(times, task_refs_)) check(idocp_ocp_set_task_refs(h_, t, M, task_refs_.data()));
  }
  std::vector<double> chain_times_, task_refs_;
  void syncCost() {
    if (!cost_) return;
    const idocp_cost_t c = cost_->native();
    if (std::memcmp(&c, &last_cost_, sizeof(c)) != 0) { check(idocp_ocp_set_cost(h_, &c)); last_cost_ = c; }
  }
  mutable std::vector<SplitSolution> cache_;
  std::unique_ptr<UnOCPSolver> un_;
  int dimOf(const std::string& name) const {
    if (name == "q") return robot_.dimq();
    if (name == "u") return robot_.dimu();
    if (name == "f" || name == "mu") return robot_.max_dimf();
    if (name == "nu_passive") return robot_.dim_passive();
    return robot_.dimv();
  }
  static void check(int rc) {
    if (rc != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
  static void flatten(const ContactStatus& cs, std::vector<int>& active, std::vector<double>& pts) {
    const int nc = cs.maxPointContacts();
    active.resize(nc); pts.resize(3 * (size_t)nc);
    for (int c = 0; c < nc; ++c) {
      active[c] = cs.isContactActive(c) ? 1 : 0;
      for (int k = 0; k < 3; ++k) pts[3 * c + k] = cs.contactPoint(c)[k];
    }
  }
};

}  // namespace idocp
#endif  // IDOCP_OCP_SOLVER_HPP_
