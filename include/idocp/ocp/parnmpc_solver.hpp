// idocp::ParNMPCSolver -- drop-in facade over the HIP ParNMPC path.
//
// Same constructor signature and methods as the reference class
// (include/idocp/ocp/parnmpc_solver.hpp; src/ocp/parnmpc_solver.cpp).  Every method forwards to
// the C ABI (include/idocp_hip.h: idocp_parnmpc_* and the shared idocp_ocp_* entry points); the
// arithmetic runs in the HIP kernels K5a / K5b<BWD> / K9b / S5 / K10a / S6 / K10b / K6 / K7 and, for
// contact sequences with discrete events (pushBackContactStatus; max_num_impulse > 0), K5s / K9i / K9g
// for the aux and impulse stages of the ParNMPCDiscretizer chain.
//
// A FIXED-BASE robot without contact frames (examples/iiwa14/parnmpc_benchmark.cpp of the reference) is bound to the kernels of
// idocp::UnParNMPCSolver, for the reason given in ocp_solver.hpp: one Newton system per stage, condensed in two orders; the backward correction
// only sees its state and costate blocks, which do not depend on the order (tests/test_oracle_fixed_base.py, tests/test_fixed_base_ocp_gpu.py).
#ifndef IDOCP_PARNMPC_SOLVER_HPP_
#define IDOCP_PARNMPC_SOLVER_HPP_

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
#include "idocp/unocp/unparnmpc_solver.hpp"
#include "idocp_hip.h"

namespace idocp {

class ParNMPCSolver {
 public:
  ParNMPCSolver(const Robot& robot, const std::shared_ptr<CostFunction>& cost, const std::shared_ptr<Constraints>& constraints,
                const double T, const int N, const int max_num_impulse = 0, const int nthreads = 1, const int device = 0)
      : robot_(robot), N_(N), h_(nullptr), comm_(nullptr), cost_(cost) {
    (void)nthreads;
    if (!robot.hasFloatingBase() && robot.maxPointContacts() == 0) {
      un_.reset(new UnParNMPCSolver(robot, cost, constraints, T, N, nthreads, device));
      return;
    }
    const idocp_cost_t c = cost->native();
    last_cost_ = c;
    const idocp_constraints_t k = constraints->native();
    if (max_num_impulse > 0) check(idocp_parnmpc_create_hybrid(&robot.model(), &c, &k, T, N, max_num_impulse, 1, device, &h_));
    else check(idocp_parnmpc_create(&robot.model(), &c, &k, T, N, 1, device, &h_));
  }
  // Multi-GPU (BASELINE configs[3]): ONE process per GPU, every process constructs the solver with the same arguments plus its
  // communicator (idocp_comm_init_rank over an id made by idocp_comm_get_unique_id on rank 0).  The process then owns the grid
  // stages [rank N / world, (rank + 1) N / world) of the horizon; initBackwardCorrection / updateSolution / KKTError are collective
  // calls (RCCL halo exchange with the two neighbours, idocp_amd/csrc/parnmpc_dist.hip -- the counterpart of the reference's OpenMP
  // loops over one horizon, backward_correction_solver.cpp:255-366).  getSolution returns the local stages.
  ParNMPCSolver(const Robot& robot, const std::shared_ptr<CostFunction>& cost, const std::shared_ptr<Constraints>& constraints,
                const double T, const int N, idocp_comm_t* comm, const int max_num_impulse = 0, const int device = 0)
      : robot_(robot), N_(N), h_(nullptr), comm_(comm), cost_(cost) {
    const idocp_cost_t c = cost->native();
    last_cost_ = c;
    const idocp_constraints_t k = constraints->native();
    const int rank = idocp_comm_rank(comm), world = idocp_comm_world(comm);
    if (world < 1 || N % world != 0) { std::cerr << "invalid value: N must be divisible by the number of ranks!\n"; std::exit(EXIT_FAILURE); }
    const int Nl = N / world;
    if (max_num_impulse > 0) check(idocp_parnmpc_create_hybrid_shard(&robot.model(), &c, &k, T, N, max_num_impulse, rank * Nl, (rank + 1) * Nl, 1, device, &h_));
    else check(idocp_parnmpc_create_shard(&robot.model(), &c, &k, T / world, Nl, rank * Nl, rank == world - 1 ? 1 : 0, rank > 0 ? 1 : 0, 1, device, &h_));
    check(idocp_parnmpc_dist_attach(h_, comm_));
    N_ = Nl;
  }
  // parnmpc_solver.hpp:47: an empty solver, to be assigned a constructed one before use
  ParNMPCSolver() : robot_(), N_(0), h_(nullptr), comm_(nullptr) {}
  ~ParNMPCSolver() { if (comm_) idocp_parnmpc_dist_detach(h_); idocp_ocp_destroy(h_); }
  // copyable and movable like the reference class (a copy is a deep copy of the device state; a sharded solver is bound to its
  // communicator and can only be moved)
  ParNMPCSolver(const ParNMPCSolver& other) : robot_(other.robot_), N_(other.N_), h_(nullptr), comm_(nullptr), cost_(other.cost_), last_cost_(other.last_cost_) {
    if (other.comm_) { std::cerr << "a sharded ParNMPCSolver cannot be copied\n"; std::exit(EXIT_FAILURE); }
    if (other.h_) check(idocp_ocp_clone(other.h_, &h_));
    if (other.un_) un_.reset(new UnParNMPCSolver(*other.un_));
  }
  ParNMPCSolver& operator=(const ParNMPCSolver& other) {
    if (this != &other) {
      if (other.comm_ || comm_) { std::cerr << "a sharded ParNMPCSolver cannot be copied\n"; std::exit(EXIT_FAILURE); }
      idocp_ocp_t* n = nullptr; if (other.h_) check(idocp_ocp_clone(other.h_, &n)); idocp_ocp_destroy(h_); h_ = n; robot_ = other.robot_; N_ = other.N_;
      cost_ = other.cost_; last_cost_ = other.last_cost_;
      un_.reset(other.un_ ? new UnParNMPCSolver(*other.un_) : nullptr);
    }
    return *this;
  }
  ParNMPCSolver(ParNMPCSolver&& other) noexcept : robot_(other.robot_), N_(other.N_), h_(other.h_), comm_(other.comm_), cost_(std::move(other.cost_)), last_cost_(other.last_cost_), kkt_error_(other.kkt_error_), un_(std::move(other.un_)) { other.h_ = nullptr; other.comm_ = nullptr; }
  ParNMPCSolver& operator=(ParNMPCSolver&& other) noexcept {
    if (this != &other) {
      idocp_ocp_destroy(h_);                 // (drops this solver's own attachment, if any)
      h_ = other.h_; comm_ = other.comm_; robot_ = other.robot_; N_ = other.N_; kkt_error_ = other.kkt_error_;
      cost_ = std::move(other.cost_); last_cost_ = other.last_cost_;
      un_ = std::move(other.un_);
      other.h_ = nullptr; other.comm_ = nullptr;
    }
    return *this;
  }

  void initConstraints(const double t) {
    if (un_) { un_->initConstraints(); return; }      // (stage i with time step i + 1 in both solvers: parnmpc_linearizer.cpp:54, unparnmpc_solver.cpp:59)
    syncTaskRefs(t);
    check(idocp_ocp_init_constraints(h_, t));
  }
  void initBackwardCorrection(const double t) {
    if (un_) { un_->initBackwardCorrection(t); return; }
    syncTaskRefs(t);
    if (comm_) check(idocp_parnmpc_dist_init_backward_correction(h_, t));
    else check(idocp_parnmpc_init_backward_correction(h_, t));
  }

  void updateSolution(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v, const bool line_search = false) {
    if (un_) { un_->updateSolution(t, q, v, line_search); return; }
    syncCost();
    syncTaskRefs(t);
    if (comm_) {
      if (idocp_comm_rank(comm_) == 0) check(idocp_parnmpc_dist_set_initial_state(h_, q.data(), v.data(), robot_.dimq(), robot_.dimv()));
      // (line_search: the probes of the filter line search are evaluated collectively, every rank runs the same filter)
      check(line_search ? idocp_parnmpc_dist_update_solution_ls(h_, t) : idocp_parnmpc_dist_update_solution(h_, t));
      check(idocp_ocp_synchronize(h_));
      return;
    }
    check(idocp_parnmpc_update_solution(h_, t, q.data(), v.data(), line_search ? 1 : 0));
  }

  // stages 0 .. N-1 (stage i lives at t + (i + 1) T / N)
  std::vector<Eigen::VectorXd> getSolution(const std::string& name) const {
    if (un_) {
      if (name == "f" || name == "mu" || name == "nu_passive") return std::vector<Eigen::VectorXd>(N_, Eigen::VectorXd(0));
      return un_->getSolution(name);
    }
    const int dim = name == "q" ? robot_.dimq() : (name == "u" ? robot_.dimu() : ((name == "f" || name == "mu") ? robot_.max_dimf() : robot_.dimv()));
    std::vector<double> buf((size_t)N_ * dim);
    check(idocp_ocp_get_solution(h_, name.c_str(), 0, buf.data()));
    std::vector<Eigen::VectorXd> out(N_, Eigen::VectorXd(dim));
    for (int i = 0; i < N_; ++i) for (int j = 0; j < dim; ++j) out[i][j] = buf[(size_t)i * dim + j];
    return out;
  }

  // parnmpc_solver.hpp:103: const reference to the split solution of stage 0 .. N-1 (one device-to-host copy of its record)
  const SplitSolution& getSolution(const int stage) const {
    if (un_) return un_->getSolution(stage);
    if ((int)cache_.size() != N_) cache_.resize(N_);
    SplitSolution& s = cache_.at(stage);
    const int nv = robot_.dimv(), nc = robot_.maxPointContacts();
    std::vector<double> rec((size_t)5 * nv + robot_.dimq() + robot_.dimu() + 6 * nc + 6);
    check(idocp_ocp_get_split_solution(h_, 0, stage, rec.data()));
    std::vector<int> active(nc > 0 ? nc : 1, 0);
    if (idocp_ocp_get_stage_contact_status(h_, stage, active.data()) < 0) check(IDOCP_E_ARG);
    s.assign(rec.data(), robot_.dimq(), nv, robot_.dimu(), nc, robot_.dim_passive(), active.data());
    return s;
  }

  void setSolution(const std::string& name, const Eigen::VectorXd& value) {
    if (un_) { check(idocp_unocp_set_solution_only(un_->handle(), name.c_str(), value.data())); return; }      // (no re-initialisation of the constraints: parnmpc_solver.cpp:128-180)
    check(idocp_ocp_set_solution(h_, name.c_str(), value.data()));
  }
  void setSolution(const std::string& name, const Eigen::Vector3d& value) {
    if (un_) {
      if (name != "f") { std::cerr << "invalid arugment: name must be q, v, a, f, or u!" << '\n'; std::exit(EXIT_FAILURE); }
      return;
    }
    check(idocp_ocp_set_solution(h_, name.c_str(), value.data()));
  }

  // ParNMPCSolver::getStateFeedbackGain (parnmpc_solver.cpp:116-125): the reference checks its arguments and leaves Kq, Kv as they are
  void getStateFeedbackGain(const int, Eigen::MatrixXd&, Eigen::MatrixXd&) const {}

  void setContactStatusUniformly(const ContactStatus& contact_status) {
    const int nc = contact_status.maxPointContacts();
    std::vector<int> active(nc);
    std::vector<double> pts(3 * (size_t)nc);
    for (int c = 0; c < nc; ++c) {
      active[c] = contact_status.isContactActive(c) ? 1 : 0;
      for (int k = 0; k < 3; ++k) pts[3 * c + k] = contact_status.contactPoint(c)[k];
    }
    if (un_) return;
    check(idocp_ocp_set_contact_status_uniformly(h_, active.data(), pts.data()));
  }
  // parnmpc_solver.cpp:179-206
  void pushBackContactStatus(const ContactStatus& contact_status, const double switching_time) {
    const int nc = contact_status.maxPointContacts();
    std::vector<int> active(nc);
    std::vector<double> pts(3 * (size_t)nc);
    for (int c = 0; c < nc; ++c) {
      active[c] = contact_status.isContactActive(c) ? 1 : 0;
      for (int k = 0; k < 3; ++k) pts[3 * c + k] = contact_status.contactPoint(c)[k];
    }
    if (un_) { std::cerr << "discrete_event.existDiscreteEvent() must be true!" << '\n'; std::exit(EXIT_FAILURE); }      // contact_sequence.hxx:69-72
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
  void clearLineSearchFilter() { if (un_) un_->clearLineSearchFilter(); else check(idocp_ocp_clear_line_search_filter(h_)); }      // ParNMPCSolver::clearLineSearchFilter (parnmpc_solver.cpp:226-228)

  // ParNMPCSolver::isCurrentSolutionFeasible (parnmpc_solver.cpp:231-273)
  bool isCurrentSolutionFeasible() {
    if (un_) return un_->isCurrentSolutionFeasible();
    int ok = 0, where = -1;
    check(idocp_ocp_is_current_solution_feasible(h_, &ok, &where));
    if (!ok) std::cout << "INFEASIBLE at stage " << where << " of the discretised horizon" << std::endl;
    return ok != 0;
  }

  double KKTError() {
    if (un_) return un_->KKTError();
    if (comm_) return kkt_error_;
    double e = 0;
    check(idocp_ocp_kkt_error(h_, &e));
    return e;
  }
  void computeKKTResidual(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v) {
    if (un_) { un_->computeKKTResidual(t, q, v); return; }
    syncCost();
    syncTaskRefs(t);
    if (comm_) {
      if (idocp_comm_rank(comm_) == 0) check(idocp_parnmpc_dist_set_initial_state(h_, q.data(), v.data(), robot_.dimq(), robot_.dimv()));
      check(idocp_parnmpc_dist_kkt_error(h_, t, &kkt_error_));
      return;
    }
    check(idocp_parnmpc_compute_kkt_residual(h_, t, q.data(), v.data()));
  }
  idocp_ocp_t* handle() { return h_; }
  UnParNMPCSolver* unconstrainedSolver() { return un_.get(); }      // the solver a fixed-base robot without contacts is bound to (handle() is null then)

 private:
  Robot robot_;
  int N_;
  idocp_ocp_t* h_;
  idocp_comm_t* comm_ = nullptr;      // not owned
  // the cost object is shared with the caller (as in the reference, whose solver keeps the shared_ptr): weights edited through it after
  // construction reach the device with the next call (idocp_ocp_set_cost)
  std::shared_ptr<CostFunction> cost_;
  idocp_cost_t last_cost_{};
  double kkt_error_ = 0.0;
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
    if (!cost_ || !h_) return;
    const idocp_cost_t c = cost_->native();
    if (std::memcmp(&c, &last_cost_, sizeof(c)) != 0) { check(idocp_ocp_set_cost(h_, &c)); last_cost_ = c; }
  }
  mutable std::vector<SplitSolution> cache_;
  std::unique_ptr<UnParNMPCSolver> un_;
  static void check(int rc) {
    if (rc != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
};

}  // namespace idocp
#endif  // IDOCP_PARNMPC_SOLVER_HPP_
