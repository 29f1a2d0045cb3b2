// idocp::UnParNMPCSolver -- drop-in facade over the HIP path.
//
// Same constructor signature and methods as the reference class
// (include/idocp/unocp/unparnmpc_solver.hpp:31-189; src/unocp/unparnmpc_solver.cpp).
// Every method forwards to the C ABI (include/idocp_hip.h, idocp_unparnmpc_* and the
// shared idocp_unocp_* accessors); the arithmetic runs in the HIP kernels.  `nthreads`
// is accepted for source compatibility and ignored.  Argument errors: message on
// stderr + std::exit(EXIT_FAILURE), like the reference.
#ifndef IDOCP_UNPARNMPC_SOLVER_HPP_
#define IDOCP_UNPARNMPC_SOLVER_HPP_

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
#include "idocp/unocp/unocp_solver.hpp"      // SplitSolution
#include "idocp_hip.h"

namespace idocp {

class UnParNMPCSolver {
 public:
  UnParNMPCSolver(const Robot& robot, const std::shared_ptr<CostFunction>& cost, const std::shared_ptr<Constraints>& constraints,
                  const double T, const int N, const int nthreads = 1, const int device = 0)
      : robot_(robot), N_(N), h_(nullptr), cost_(cost) {
    (void)nthreads;
    const idocp_cost_t c = cost->native();
    const idocp_constraints_t k = constraints->native();
    check(idocp_unparnmpc_create(&robot.model(), &c, &k, T, N, 1, device, &h_));
    last_cost_ = c;
    cache_.resize(N);
  }
  // unparnmpc_solver.hpp:48: an empty solver, to be assigned a constructed one before use
  UnParNMPCSolver() : robot_(), N_(0), h_(nullptr) {}
  ~UnParNMPCSolver() { idocp_unocp_destroy(h_); }
  // copyable and movable like the reference class (`= default` there): a copy is a DEEP copy of the device state (idocp_unocp_clone)
  UnParNMPCSolver(const UnParNMPCSolver& other) : robot_(other.robot_), N_(other.N_), h_(nullptr), cache_(other.cache_), cost_(other.cost_), last_cost_(other.last_cost_) {
    if (other.h_) check(idocp_unocp_clone(other.h_, &h_));
  }
  UnParNMPCSolver& operator=(const UnParNMPCSolver& other) {
    if (this != &other) {
      idocp_unocp_t* n = nullptr;
      if (other.h_) check(idocp_unocp_clone(other.h_, &n));
      idocp_unocp_destroy(h_);
      h_ = n; robot_ = other.robot_; N_ = other.N_; cache_ = other.cache_; cost_ = other.cost_; last_cost_ = other.last_cost_;
    }
    return *this;
  }
  UnParNMPCSolver(UnParNMPCSolver&& other) noexcept : robot_(other.robot_), N_(other.N_), h_(other.h_), cache_(std::move(other.cache_)), cost_(std::move(other.cost_)), last_cost_(other.last_cost_) { other.h_ = nullptr; }
  UnParNMPCSolver& operator=(UnParNMPCSolver&& other) noexcept {
    if (this != &other) { idocp_unocp_destroy(h_); h_ = other.h_; other.h_ = nullptr; robot_ = other.robot_; N_ = other.N_; cache_ = std::move(other.cache_); cost_ = std::move(other.cost_); last_cost_ = other.last_cost_; }
    return *this;
  }

  void initConstraints() { check(idocp_unocp_init_constraints(h_)); }
  void initBackwardCorrection(const double t) { syncCost(); check(idocp_unparnmpc_init_backward_correction(h_, t)); }

  void updateSolution(const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v, const bool line_search = false) {
    syncCost();
    check(idocp_unparnmpc_update_solution(h_, t, q.data(), v.data(), line_search ? 1 : 0));
  }

  // stage in [0, N): the N backward-Euler stages (stage i lives at t + (i + 1) dt)
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
    std::vector<double> buf((size_t)(N_ + 1) * dim);
    check(idocp_unocp_get_solution(h_, name.c_str(), 0, buf.data()));
    std::vector<Eigen::VectorXd> out(N_, Eigen::VectorXd(dim));
    for (int i = 0; i < N_; ++i) for (int j = 0; j < dim; ++j) out[i][j] = buf[(size_t)i * dim + j];
    return out;
  }

  // left unimplemented by the reference as well (unparnmpc_solver.cpp:107-118)
  void getStateFeedbackGain(const int, Eigen::MatrixXd&, Eigen::MatrixXd&) const {}

  void setSolution(const std::string& name, const Eigen::VectorXd& value) { check(idocp_unocp_set_solution(h_, name.c_str(), value.data())); }
  void clearLineSearchFilter() { check(idocp_unocp_clear_line_search_filter(h_)); }

  bool isCurrentSolutionFeasible() {
    int ok = 0, where = -1;
    check(idocp_unocp_is_current_solution_feasible(h_, &ok, &where));
    if (!ok) std::cout << "INFEASIBLE at time stage " << where << std::endl;
    return ok != 0;
  }

  // UnParNMPCSolver::printSolution / saveSolution (unparnmpc_solver.cpp:243-327): the stage-wise solution on stdout / as a text file, one stage per
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
    check(idocp_unparnmpc_compute_kkt_residual(h_, t, q.data(), v.data()));
  }
  idocp_unocp_t* handle() { return h_; }

 private:
  Robot robot_;
  int N_;
  idocp_unocp_t* h_;
  mutable std::vector<SplitSolution> cache_;
  // the cost object is shared with the caller (as in the reference, whose solver keeps the shared_ptr): weights and references edited through it
  // after construction reach the device with the next call (idocp_unocp_set_cost)
  std::shared_ptr<CostFunction> cost_;
  idocp_cost_t last_cost_{};
  void syncCost() {
    if (!cost_ || !h_) return;
    const idocp_cost_t c = cost_->native();
    if (std::memcmp(&c, &last_cost_, sizeof(c)) != 0) { check(idocp_unocp_set_cost(h_, &c)); last_cost_ = c; }
  }
  static void check(int rc) {
    if (rc != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
};

}  // namespace idocp
#endif  // IDOCP_UNPARNMPC_SOLVER_HPP_
