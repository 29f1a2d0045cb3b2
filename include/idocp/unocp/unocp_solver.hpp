// idocp::UnOCPSolver -- drop-in facade over the HIP path.
//
// Same constructor signature and methods as the reference class
// (include/idocp/unocp/unocp_solver.hpp:25-188; src/unocp/unocp_solver.cpp).
// Every method forwards to the C ABI (include/idocp_hip.h); the arithmetic runs
// in the HIP kernels.  `nthreads` is accepted for source compatibility and
// ignored (the GPU replaces the OpenMP team).  Argument errors: message on
// stderr + std::exit(EXIT_FAILURE), like the reference.
#ifndef IDOCP_UNOCP_SOLVER_HPP_
#define IDOCP_UNOCP_SOLVER_HPP_

#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/eigen_shim.hpp"
#include "idocp/ocp/split_solution.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp_hip.h"

namespace idocp {

class UnOCPSolver {
 public:
  UnOCPSolver(const Robot& robot, const std::shared_ptr<CostFunction>& cost, const std::shared_ptr<Constraints>& constraints,
              const double T, const int N, const int nthreads = 1, const int device = 0)
      : robot_(robot), cost_(cost), N_(N), dt_(T / N), h_(nullptr) {
    (void)nthreads;
    const idocp_cost_t c = cost->native();
    const idocp_constraints_t k = constraints->native();
    check(idocp_unocp_create(&robot.model(), &c, &k, T, N, 1, device, &h_));
    last_cost_ = c;
    cache_.resize(N + 1);
  }
  // unocp_solver.hpp:49: an empty solver, to be assigned a constructed one before use
  UnOCPSolver() : robot_(), cost_(), N_(0), dt_(0.0), h_(nullptr) {}
  ~UnOCPSolver() { idocp_unocp_destroy(h_); }
  // copyable and movable like the reference class (unocp_solver.hpp:59-74, `= default`): a copy is a DEEP copy of the solver state on
  // the device (idocp_unocp_clone)
  UnOCPSolver(const UnOCPSolver& other) : robot_(other.robot_), cost_(other.cost_), N_(other.N_), dt_(other.dt_), h_(nullptr), cache_(other.cache_), last_cost_(other.last_cost_) {
    if (other.h_) check(idocp_unocp_clone(other.h_, &h_));
  }
  UnOCPSolver& operator=(const UnOCPSolver& other) {
    if (this != &other) {
      idocp_unocp_t* n = nullptr;
      if (other.h_) check(idocp_unocp_clone(other.h_, &n));
      idocp_unocp_destroy(h_);
      h_ = n; robot_ = other.robot_; cost_ = other.cost_; N_ = other.N_; dt_ = other.dt_; cache_ = other.cache_; last_cost_ = other.last_cost_;
    }
    return *this;
  }
  UnOCPSolver(UnOCPSolver&& other) noexcept
      : robot_(other.robot_), cost_(std::move(other.cost_)), N_(other.N_), dt_(other.dt_), h_(other.h_), cache_(std::move(other.cache_)), last_cost_(other.last_cost_) { other.h_ = nullptr; }
  UnOCPSolver& operator=(UnOCPSolver&& other) noexcept {
    if (this != &other) {
      idocp_unocp_destroy(h_);
      h_ = other.h_; other.h_ = nullptr; robot_ = other.robot_; cost_ = std::move(other.cost_); N_ = other.N_; dt_ = other.dt_; cache_ = std::move(other.cache_); last_cost_ = other.last_cost_;
    }
    return *this;
  }

  void initConstraints() { check(idocp_unocp_init_constraints(h_)); }

  void updateSolution(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v, const bool line_search = false) {
    syncCost();
    uploadTaskRefs(t);
    check(idocp_unocp_update_solution(h_, t, q.data(), v.data(), line_search ? 1 : 0));
  }

  // unocp_solver.hpp:96: const reference to the split solution of a time stage -- one device-to-host copy of the stage's record
  const SplitSolution& getSolution(const int stage) const {
    SplitSolution& s = cache_.at(stage);
    const int nv = robot_.dimv();
    std::vector<double> rec((size_t)7 * nv);
    check(idocp_unocp_get_split_solution(h_, 0, stage, rec.data()));
    s.assign(rec.data(), nv, nv, nv, 0, 0);
    return s;
  }

  std::vector<Eigen::VectorXd> getSolution(const std::string& name) const {
    const int dim = robot_.dimv();
    const bool per_stage = (name == "a" || name == "u" || name == "beta");
    const int n = per_stage ? N_ : N_ + 1;
    std::vector<double> buf((size_t)(N_ + 1) * dim);
    check(idocp_unocp_get_solution(h_, name.c_str(), 0, buf.data()));
    std::vector<Eigen::VectorXd> out(n, Eigen::VectorXd(dim));
    for (int i = 0; i < n; ++i) for (int j = 0; j < dim; ++j) out[i][j] = buf[(size_t)i * dim + j];
    return out;
  }

  // The reference leaves this unimplemented for the Un path (unocp_solver.cpp:144-154);
  // here it returns the acceleration gains Kq, Kv of the LQR policy da = K dx + k.
  void getStateFeedbackGain(const int time_stage, Eigen::MatrixXd& Kq, Eigen::MatrixXd& Kv) const {
    const int nv = robot_.dimv();
    std::vector<double> K((size_t)N_ * nv * 2 * nv);
    check(idocp_unocp_get_riccati(h_, 0, nullptr, nullptr, K.data(), nullptr));
    Kq.resize(nv, nv); Kv.resize(nv, nv);
    const double* Ki = &K[(size_t)time_stage * nv * 2 * nv];
    for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) { Kq(r, c) = Ki[c * nv + r]; Kv(r, c) = Ki[(nv + c) * nv + r]; }
  }

  void setSolution(const std::string& name, const Eigen::VectorXd& value) { check(idocp_unocp_set_solution(h_, name.c_str(), value.data())); }
  void clearLineSearchFilter() { check(idocp_unocp_clear_line_search_filter(h_)); }

  // UnOCPSolver::isCurrentSolutionFeasible (unocp_solver.cpp:228-237)
  bool isCurrentSolutionFeasible() {
    int ok = 0, where = -1;
    check(idocp_unocp_is_current_solution_feasible(h_, &ok, &where));
    if (!ok) std::cout << "INFEASIBLE at time stage " << where << std::endl;
    return ok != 0;
  }

  // UnOCPSolver::printSolution / saveSolution (unocp_solver.cpp:266-352): the stage-wise solution on stdout / as a text file, one stage per
  // line.  "end-effector" needs frame kinematics of arbitrary frames, which the HIP path does not carry.
  void printSolution(const std::string& name = "all", const std::vector<int> frames = {}) const {
    (void)frames;
    if (name == "end-effector") { std::cerr << "printSolution(\"end-effector\") is not supported by the HIP path" << std::endl; return; }
    const char* fields[4] = {"q", "v", "a", "u"};
    if (name == "all") {
      std::vector<std::vector<Eigen::VectorXd>> all;
      for (const char* f : fields) all.push_back(getSolution(f));
      for (size_t i = 0; i < all[0].size(); ++i)
        for (int f = 0; f < 4; ++f)
          if (i < all[f].size()) std::cout << fields[f] << "[" << i << "] = " << all[f][i] << std::endl;
      return;
    }
    for (const char* f : fields) {
      if (name != f) continue;
      const std::vector<Eigen::VectorXd> sol = getSolution(f);
      for (size_t i = 0; i < sol.size(); ++i) std::cout << f << "[" << i << "] = " << sol[i] << std::endl;
    }
  }
  void saveSolution(const std::string& path_to_file, const std::string& name) const {
    std::ofstream file(path_to_file);
    if (name == "q" || name == "v" || name == "a" || name == "u") {
      const std::vector<Eigen::VectorXd> sol = getSolution(name);
      for (const Eigen::VectorXd& x : sol) {
        for (int j = 0; j < (int)x.size(); ++j) file << x[j] << " ";
        file << "\n";
      }
    }
    file.close();
  }

  double KKTError() {
    double e = 0;
    check(idocp_unocp_kkt_error(h_, &e));
    return e;
  }
  void computeKKTResidual(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v) {
    syncCost();
    uploadTaskRefs(t);
    check(idocp_unocp_compute_kkt_residual(h_, t, q.data(), v.data()));
  }
  idocp_unocp_t* handle() { return h_; }

 private:
  Robot robot_;
  std::shared_ptr<CostFunction> cost_;
  int N_;
  double dt_;
  idocp_unocp_t* h_;
  mutable std::vector<SplitSolution> cache_;
  std::vector<double> task_refs_;
  // The reference's solver SHARES the cost function with the driver (unocp_solver.hpp: shared_ptr members): a reference or weight the driver
  // changes between two calls takes effect at the next one.  The device holds a copy; it is refreshed when the shared object has changed.
  idocp_cost_t last_cost_{};
  void syncCost() {
    if (!cost_ || !h_) return;
    const idocp_cost_t c = cost_->native();
    if (std::memcmp(&c, &last_cost_, sizeof(c)) != 0) { check(idocp_unocp_set_cost(h_, &c)); last_cost_ = c; }
  }
  // TimeVaryingTaskSpace*Cost: the reference asks the user's ref object at the time of every stage inside linearizeOCP
  // (unocp_solver.cpp:78-94 -> time_varying_task_space_6d_cost.cpp:65-67); here the poses are evaluated up front and uploaded
  void uploadTaskRefs(const double t) {
    if (cost_->taskRefs(t, dt_, N_, task_refs_)) check(idocp_unocp_set_task_refs(h_, task_refs_.data()));
  }
  static void check(int rc) {
    if (rc != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
};

}  // namespace idocp
#endif  // IDOCP_UNOCP_SOLVER_HPP_
