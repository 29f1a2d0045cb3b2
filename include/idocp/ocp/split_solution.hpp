// idocp::SplitSolution -- the split solution of one time stage as the solvers' getSolution(stage) returns it.
//
// Member names follow the reference class (include/idocp/ocp/split_solution.hpp:192-237): lmd, gmm, q, v, a, u, beta, the
// per-contact 3-vectors f and mu, nu_passive, and the stacked views f_stack() / mu_stack() (:93-122; every contact of the
// robot, active or not, three entries each).  On a fixed-base robot f, mu and nu_passive are empty.
#ifndef IDOCP_SPLIT_SOLUTION_HPP_
#define IDOCP_SPLIT_SOLUTION_HPP_

#include <vector>

#include "idocp/eigen_shim.hpp"

namespace idocp {

class SplitSolution {
 public:
  Eigen::VectorXd lmd, gmm, q, v, a, u, beta, nu_passive;
  std::vector<Eigen::Vector3d> f, mu;

  const Eigen::VectorXd& f_stack() const { return f_stack_; }
  const Eigen::VectorXd& mu_stack() const { return mu_stack_; }

  // fill from one record of the C ABI (idocp_ocp_get_split_solution / idocp_unocp_get_split_solution):
  // lmd gmm q v a u beta [f mu nu_passive]
  void assign(const double* rec, int nq, int nv, int nu, int ncontacts, int npassive) {
    const double* p = rec;
    auto take = [&](Eigen::VectorXd& dst, int n) { dst.resize(n); for (int i = 0; i < n; ++i) dst[i] = p[i]; p += n; };
    take(lmd, nv); take(gmm, nv); take(q, nq); take(v, nv); take(a, nv); take(u, nu); take(beta, nv);
    take(f_stack_, 3 * ncontacts); take(mu_stack_, 3 * ncontacts); take(nu_passive, npassive);
    f.resize(ncontacts); mu.resize(ncontacts);
    for (int c = 0; c < ncontacts; ++c)
      for (int k = 0; k < 3; ++k) { f[c][k] = f_stack_[3 * c + k]; mu[c][k] = mu_stack_[3 * c + k]; }
  }

 private:
  Eigen::VectorXd f_stack_, mu_stack_;
};

using SplitSolutionOCP = SplitSolution;      // (rounds 1-3 had a second struct for the floating-base solvers)

}  // namespace idocp
#endif  // IDOCP_SPLIT_SOLUTION_HPP_
