// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the fixed-base ("unconstrained") half of the idocp hot
// path: SplitUnOCP / TerminalOCP / UnconstrainedDynamics / UnRiccatiRecursion /
// UnOCPSolver.  Each function cites the reference lines it follows; the order
// of operations is the reference's.
#ifndef ORACLE_UNOCP_HPP_
#define ORACLE_UNOCP_HPP_

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <utility>
#include <array>
#include <vector>

#include "idocp_hip.h"
#include "mat.hpp"
#include "rbd.hpp"

namespace oracle {

// include/idocp/ocp/split_solution.hxx:10-31 (fixed-base fields)
struct SplitSolution {
  Mat lmd, gmm, q, v, a, u, beta;
  explicit SplitSolution(const Robot& r)
      : lmd(r.dimv()), gmm(r.dimv()), q(r.dimq()), v(r.dimv()), a(r.dimv()), u(r.dimu()), beta(r.dimv()) {}
};

// include/idocp/ocp/split_direction.hxx:8-23 (fixed-base fields)
struct SplitDirection {
  Mat dlmd, dgmm, dq, dv, da, du, dbeta;
  explicit SplitDirection(const Robot& r)
      : dlmd(r.dimv()), dgmm(r.dimv()), dq(r.dimv()), dv(r.dimv()), da(r.dimv()), du(r.dimu()), dbeta(r.dimv()) {}
};

// include/idocp/constraints/constraint_component_data.hpp
struct ConstraintComponentData {
  Mat slack, dual, residual, duality, dslack, ddual;
  explicit ConstraintComponentData(int dimc = 0)
      : slack(dimc), dual(dimc), residual(dimc), duality(dimc), dslack(dimc), ddual(dimc) {}
};

// One joint-limit component (src/constraints/joint_*_limit.cpp); the six
// reference classes differ only in the variable, the bound and a sign.
struct JointLimit {
  enum Var { Q = 0, V = 1, U = 2, A = 3 };
  Var var;
  int sign;            // -1 lower limit, +1 upper limit
  Mat lim;             // bound (xmin for lower, xmax for upper)
  int level() const { return var == Q ? 0 : (var == V ? 1 : 2); }  // position / velocity / acceleration
};

// include/idocp/constraints/constraints.hxx + constraints_data.hpp:18-42
struct Constraints {
  std::vector<JointLimit> components;   // stored level by level like the reference
  real barrier = 1.0e-04, fraction_to_boundary_rate = 0.995;
  Constraints(const Robot& robot, const idocp_constraints_t& c);
  int dimc_total() const;
  bool valid(const JointLimit& jl, int time_stage) const {
    // ConstraintsData(time_stage): stage 0 -> acceleration level only,
    // stage 1 -> velocity + acceleration, >= 2 -> all
    if (time_stage >= 2) return true;
    if (time_stage == 1) return jl.level() >= 1;
    return jl.level() >= 2;
  }
};

struct ConstraintsData {
  int time_stage = 0;
  std::vector<ConstraintComponentData> data;   // one per component (invalid ones keep dimc 0 semantics)
};

// include/idocp/unocp/split_unkkt_matrix.hxx:31-147 -- Q ordered (a, q, v)
struct SplitUnKKTMatrix {
  int nv; Mat Q;
  explicit SplitUnKKTMatrix(int nv_) : nv(nv_), Q(3 * nv_, 3 * nv_) {}
  Mat blk(int bi, int bj) const { return Q.block(bi * nv, bj * nv, nv, nv); }
  void set(int bi, int bj, const Mat& m) { Q.setBlock(bi * nv, bj * nv, m); }
  void add(int bi, int bj, const Mat& m, real al = 1.0) { Q.addBlock(bi * nv, bj * nv, m, al); }
};
// include/idocp/unocp/split_unkkt_residual.hxx:29-103 -- (Fq, Fv, la, lq, lv)
struct SplitUnKKTResidual {
  Mat Fq, Fv, la, lq, lv;
  explicit SplitUnKKTResidual(int nv) : Fq(nv), Fv(nv), la(nv), lq(nv), lv(nv) {}
};

// include/idocp/ocp/split_riccati_factorization.hpp:15-134
struct SplitRiccatiFactorization {
  Mat Pqq, Pqv, Pvq, Pvv, sq, sv;
  explicit SplitRiccatiFactorization(int nv) : Pqq(nv, nv), Pqv(nv, nv), Pvq(nv, nv), Pvv(nv, nv), sq(nv), sv(nv) {}
};

// LineSearchFilter (src/line_search/line_search_filter.cpp:33-63) with the defaults of line_search_filter.hpp:16-17 and the
// step-size schedule of UnLineSearch::computeStepSize (include/idocp/line_search/unline_search.hpp:62-92, defaults
// line_search.hpp:25-26)
struct LineSearchFilterC {
  std::vector<std::pair<real, real>> filter;
  real cost_reduction_rate = 0.005, constraints_reduction_rate = 0.005;
  real step_size_reduction_rate = 0.75, min_step_size = 0.05;
  bool isAccepted(real cost, real violation) const {
    for (const auto& p : filter) if (cost >= p.first && violation >= p.second) return false;
    return true;
  }
  void augment(real cost, real violation) {
    for (auto it = filter.begin(); it != filter.end();) {
      if (cost <= it->first && violation <= it->second) it = filter.erase(it); else ++it;
    }
    filter.push_back({cost - cost_reduction_rate * violation, (1 - constraints_reduction_rate) * violation});
  }
  // eval(alpha) -> (total cost, total violation) of the trial iterate s + alpha d (alpha = 0: the iterate itself)
  template <typename Eval>
  real computeStepSize(Eval eval, real max_primal_step_size) {
    if (filter.empty()) { const auto cv = eval(0.0); augment(cv.first, cv.second); }
    real a = max_primal_step_size;
    while (a > min_step_size) {
      const auto cv = eval(a);
      if (getenv("ORACLE_LS_DEBUG")) fprintf(stderr, "  ls try a=%.4f cost=%.10g viol=%.10g accepted=%d (filter %zu)\n", (double)a, (double)cv.first, (double)cv.second, (int)isAccepted(cv.first, cv.second), filter.size());
      if (isAccepted(cv.first, cv.second)) { augment(cv.first, cv.second); break; }
      a *= step_size_reduction_rate;
    }
    return a > min_step_size ? a : min_step_size;
  }
};

// SplitUnOCP (include/idocp/unocp/split_unocp.hxx) with its private
// kkt_matrix_/kkt_residual_, UnconstrainedDynamics and constraints data.
struct SplitUnOCP {
  int nv;
  ConstraintsData cdata;
  Mat ID, dID_dq, dID_dv, dID_da, lu_condensed;     // UnconstrainedDynamics members
  Mat Qqq, Qvv_diag, Qaa_diag, Quu_diag;            // used parts of kkt_matrix_
  Mat Fq, Fv, lq, lv, la, lu;                       // kkt_residual_
  explicit SplitUnOCP(int nv_);
};

class UnOCPSolver {
 public:
  UnOCPSolver(const RModel& model, const RCost& cost, const idocp_constraints_t& constraints,
              real T, int N);
  void setSolution(const std::string& name, const Mat& value);   // unocp_solver.cpp:157-181
  void initConstraints();                                        // unocp_solver.cpp:59-70
  void updateSolution(real t, const Mat& q, const Mat& v, bool line_search = false);      // unocp_solver.cpp:73-134
  std::pair<real, real> costAndViolation(real alpha) const;  // UnLineSearch::computeCostAndViolation (unline_search.cpp:55-82)
  LineSearchFilterC line_search;                                   // clearLineSearchFilter = line_search.filter.clear()
  void computeKKTResidual(real t, const Mat& q, const Mat& v);  // unocp_solver.cpp:205-225
  real KKTError();                                              // unocp_solver.cpp:190-202
  int isCurrentSolutionFeasible() const;                          // unocp_solver.cpp:228-237: first offending stage or -1

  // hot-path pieces, exposed so kernel-level parity tests can stop in between
  void linearizeOCP(real t, const Mat& q);                      // K1
  void backwardRiccatiRecursion();                                // S1
  void forwardRiccatiRecursion(const Mat& q, const Mat& v);       // S2
  void computeDirection();                                        // K2 (+ step sizes)
  void integrate();                                               // K3

  int N() const { return N_; }
  Robot robot;
  RCost cost;
  Constraints constraints;
  std::vector<SplitSolution> s;
  std::vector<SplitDirection> d;
  std::vector<SplitUnOCP> ocp;               // stages 0..N-1
  std::vector<SplitUnKKTMatrix> unkkt_matrix;
  std::vector<SplitUnKKTResidual> unkkt_residual;
  Mat terminal_Qqq, terminal_Qvv, terminal_lq, terminal_lv;
  std::vector<SplitRiccatiFactorization> riccati;
  std::vector<Mat> K, k;                      // LQRStateFeedbackPolicy per stage
  real primal_step_size = 1, dual_step_size = 1;
  real riccati_seconds = 0;                 // accumulated wall time of S1+S2

 private:
  int N_; real T_, dt_;
  // stage loops under `#pragma omp parallel for num_threads(nthreads)` like unocp_solver.cpp:78-94, 103-115, 121-133; every
  // thread works on a Robot of its own (the parameter shadows the member on purpose)
  void linearizeStage(Robot& robot, int i, real t, const Mat& q_prev);
  void linearizeTerminal(real t);
  void computeStageResidual(Robot& robot, int i, real t);
  std::vector<Robot> robots_;
 public:
  int nthreads = 1;
  void setNumThreads(int n) { nthreads = n < 1 ? 1 : n; robots_.assign(nthreads, robot); }
  // TaskSpace3DCost / TaskSpace6DCost (+ TimeVarying variants): the references of stages 0 .. N (rotation row-major, position)
  std::vector<std::array<real, 12>> task_refs;
  void setTaskRefs(const double* refs);          // [N + 1][12]
  // cost (without dt), gradient and Gauss-Newton Hessian of the task term at stage i (terminal weights for i = N)
  void taskTerms(int i, const Mat& q, real& c, Mat& g, Mat& H) const;
  Robot task_robot_;                             // the robot with the task frame as its contact 0
 private:
};


// UnParNMPCSolver (src/unocp/unparnmpc_solver.cpp:55-243): backward-Euler stages (SplitUnParNMPC / TerminalUnParNMPC,
// include/idocp/unocp/split_unparnmpc.hxx:60-110, terminal_unparnmpc.hxx:60-113), the per-stage KKT inverse
// (split_unkkt_matrix_inverter.hxx:37-80) and the coarse update + four correction sweeps of UnBackwardCorrection
// (src/unocp/unbackward_correction.cpp:55-132, split_unbackward_correction.hxx:38-123).  Stage i (0 <= i < N) lives at
// time t + (i + 1) dt and is created with constraint time step i + 1; the state before stage 0 is the measured (q, v).
// KKT ordering (lmd, gmm, a, q, v): rows [Fq, Fv, la, lq, lv].
class UnParNMPCSolver {
 public:
  UnParNMPCSolver(const RModel& model, const RCost& cost, const idocp_constraints_t& constraints, real T, int N);
  void setSolution(const std::string& name, const Mat& value);    // unparnmpc_solver.cpp:121-146
  void initConstraints();                                         // unparnmpc_solver.cpp:55-66
  void initBackwardCorrection(real t);                          // unbackward_correction.cpp:55-64
  void updateSolution(real t, const Mat& q, const Mat& v, bool line_search = false);       // unparnmpc_solver.cpp:74-103
  std::pair<real, real> costAndViolation(real alpha, const Mat& q, const Mat& v) const;   // unline_search.cpp:85-121
  LineSearchFilterC line_search;
  void computeKKTResidual(real t, const Mat& q, const Mat& v);   // unparnmpc_solver.cpp:169-187
  real KKTError();                                               // unparnmpc_solver.cpp:154-166
  int isCurrentSolutionFeasible() const;                           // unparnmpc_solver.cpp:190-209
  // horizon shard (test twin of idocp_unparnmpc_create_shard): every loop runs over the stages [lo, hi) only; the
  // neighbours' stages lo - 1 and hi are filled by the halo imports (oracle_unparnmpc_import)
  void setSlice(int lo, int hi) { lo_ = lo; hi_ = hi; }
  int lo() const { return lo_; }
  int hi() const { return hi_; }
  real KKTErrorSquared();
  // the phases of updateSolution, separately callable
  void coarseUpdate(real t, const Mat& q, const Mat& v);         // unbackward_correction.cpp:67-97
  void backwardCorrectionSerial();                                 // :104-106
  void backwardCorrectionParallel();                               // :107-110
  void forwardCorrectionSerial();                                  // :111-113
  void forwardCorrectionParallel();                                // :114-131 (+ direction, condensed direction, step sizes)
  void integrate();                                                // unparnmpc_solver.cpp:88-102

  int N() const { return N_; }
  Robot robot;
  RCost cost;
  Constraints constraints;
  std::vector<SplitSolution> s, s_new;        // N stages
  std::vector<SplitDirection> d;
  std::vector<SplitUnOCP> ocp;                // per-stage private data of SplitUnParNMPC (same members as SplitUnOCP)
  std::vector<SplitUnKKTMatrix> unkkt_matrix;
  std::vector<SplitUnKKTResidual> unkkt_residual;
  std::vector<Mat> aux_mat, kkt_inv, x_res;   // 2nv x 2nv, 5nv x 5nv, 2nv
  real primal_step_size = 1, dual_step_size = 1;

 private:
  int N_; real T_, dt_;
  int lo_ = 0, hi_ = 0;
  void linearizeStage(int i, const Mat& q_prev, const Mat& v_prev, bool residual_only);
};

}  // namespace oracle
#endif  // ORACLE_UNOCP_HPP_
