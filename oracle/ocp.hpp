// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the contact-capable half of the idocp hot path for a
// horizon WITHOUT discrete events (uniform contact status): SplitOCP /
// TerminalOCP / ContactDynamics / state equation with floating base /
// SplitRiccatiFactorizer / RiccatiRecursionSolver / OCPLinearizer / OCPSolver.
// Impulse, aux and lift stages and the switching constraint are not restated
// yet.  Every function cites the reference lines it follows.
#ifndef ORACLE_OCP_HPP_
#define ORACLE_OCP_HPP_

#include <string>
#include <vector>

#include "idocp_hip.h"
#include "mat.hpp"
#include "rbd.hpp"

namespace oracle {

// include/idocp/robot/contact_status.hxx
struct ContactStatus {
  std::vector<bool> active;
  std::vector<Mat> points;       // world contact points
  int dimf() const { int n = 0; for (bool a : active) n += a ? 3 : 0; return n; }
  bool hasActiveContacts() const { return dimf() > 0; }
};

// include/idocp/ocp/split_solution.hxx:10-31
struct SplitSolutionC {
  Mat lmd, gmm, q, v, a, u, beta, nu_passive;
  std::vector<Mat> f, mu;        // per contact (3)
  explicit SplitSolutionC(const Robot& r);
  Mat f_stack(const ContactStatus& cs) const;
  Mat mu_stack(const ContactStatus& cs) const;
};

// include/idocp/ocp/split_direction.hxx:8-23
struct SplitDirectionC {
  Mat dlmd, dgmm, du, dq, dv, daf, dbetamu, dnu_passive;
  explicit SplitDirectionC(const Robot& r);
};

struct IpmData {                  // ConstraintComponentData
  Mat slack, dual, residual, duality, dslack, ddual;
  explicit IpmData(int n = 0) : slack(n), dual(n), residual(n), duality(n), dslack(n), ddual(n) {}
};

// SplitKKTMatrix / SplitKKTResidual (include/idocp/ocp/split_kkt_matrix.hxx:11-30,75-497;
// split_kkt_residual.hxx:10-26) with the blocks kept as separate matrices.
struct SplitKKTMatrixC {
  int nv, nu;
  Mat Qxx, Qxu_full, Quu_full;    // (2nv x 2nv), (2nv x nv), (nv x nv); u_full = [passive(6) ; u]
  Mat Qaa_diag, Qff;              // nv ; dimf x dimf
  Mat Fqq6, Fqv6;                 // leading 6x6 blocks (rest is I, dt I implicitly)
  Mat Fvq, Fvv, Fvu;              // nv x nv, nv x nv, nv x nu
  Mat Fqq_prev6, Fqq_inv, Fqq_prev_inv;   // 6x6
  SplitKKTMatrixC(int nv_, int nu_);
};
struct SplitKKTResidualC {
  Mat Fq, Fv, lq, lv, la, lf, lu, lu_passive, Fq_prev;
  explicit SplitKKTResidualC(int nv, int nu);
};

// ContactDynamicsData (include/idocp/ocp/contact_dynamics_data.hxx)
struct ContactDynamicsDataC {
  Mat dIDda, dCda, dIDCdqv, MJtJinv, MJtJinv_dIDCdqv, Qafqv, Qafu_full, IDC, MJtJinv_IDC, laf;
};

struct RiccatiC {
  Mat Pqq, Pqv, Pvv, sq, sv;
  explicit RiccatiC(int nv) : Pqq(nv, nv), Pqv(nv, nv), Pvv(nv, nv), sq(nv), sv(nv) {}
};

class OCPSolver {
 public:
  OCPSolver(const idocp_model_t& model, const idocp_cost_t& cost, const idocp_constraints_t& constraints, double T, int N);
  void setContactStatusUniformly(const std::vector<int>& active, const double* contact_points /*[nc][3]*/);
  void setSolution(const std::string& name, const Mat& value);      // ocp_solver.cpp:95-165
  void initConstraints(double t);                                   // ocp_solver.cpp:60-64
  void updateSolution(double t, const Mat& q, const Mat& v);         // ocp_solver.cpp:67-92
  void computeKKTResidual(double t, const Mat& q, const Mat& v);     // ocp_solver.cpp:202-207
  double KKTError();                                                 // ocp_linearizer.cpp:98-137

  void linearizeOCP(double t, const Mat& q);                         // K5
  void backwardRiccatiRecursion();                                   // S3
  void forwardRiccatiRecursion(const Mat& q, const Mat& v);          // S4 (+ initial state direction)
  void computeDirection();                                           // K6
  void integrateSolution();                                          // K7

  int N() const { return N_; }
  double stepDt() const { return dt_; }
  int dimc() const;
  Robot robot;
  idocp_cost_t cost;
  idocp_constraints_t cons;
  ContactStatus contact_status;
  std::vector<SplitSolutionC> s;
  std::vector<SplitDirectionC> d;
  std::vector<SplitKKTMatrixC> kkt_matrix;
  std::vector<SplitKKTResidualC> kkt_residual;
  std::vector<ContactDynamicsDataC> cd;
  std::vector<std::vector<IpmData>> ipm;    // [stage][component]
  std::vector<RiccatiC> riccati;
  std::vector<Mat> K, k;
  double primal_step_size = 1, dual_step_size = 1;
  double riccati_seconds = 0;
  void qRef(double t, Mat& q_ref) const;                              // trotting_configuration_space_cost.hpp:126-164

 private:
  int N_, nv_, nu_, nc_;
  double T_, dt_;
  // components: 0..5 joint limits (q lo/up, v lo/up, u lo/up), 6 friction cone
  bool componentEnabled(int c) const;
  bool componentValid(int c, int stage) const;
  int componentDim(int c) const;
  void linearizeStage(int i, double t, const Mat& q_prev, bool residual_only);
  void linearizeTerminal(double t, const Mat& q_prev, bool residual_only);
};

}  // namespace oracle
#endif  // ORACLE_OCP_HPP_
