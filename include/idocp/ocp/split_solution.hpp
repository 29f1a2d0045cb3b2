// idocp::SplitSolution -- the split solution of one time stage as the solvers' getSolution(stage) returns it.
//
// Member names follow the reference class (include/idocp/ocp/split_solution.hpp:192-237): lmd, gmm, q, v, a, u, beta, the
// per-contact 3-vectors f and mu (one per contact of the robot, active or not), nu_passive, and the stacked vectors f_stack() /
// mu_stack() of the ACTIVE contacts only, dimf() entries (:93-122, split_solution.hxx:41-57, 215-240: set_f_stack() packs the
// active f[i] in contact order), with isContactActive(i) / dimf() as the stage's contact status.  On a fixed-base robot f, mu and
// nu_passive are empty and dimf() = 0.
#ifndef IDOCP_SPLIT_SOLUTION_HPP_
#define IDOCP_SPLIT_SOLUTION_HPP_

#include <vector>

#include "idocp/eigen_shim.hpp"

namespace idocp {

class SplitSolution {
 public:
  Eigen::VectorXd lmd, gmm, q, v, a, u, beta, nu_passive;
  std::vector<Eigen::Vector3d> f, mu;

  const Eigen::VectorXd& f_stack() const { return f_stack_; }        // [dimf()]: the active contacts' forces, contact order
  const Eigen::VectorXd& mu_stack() const { return mu_stack_; }
  int dimf() const { return static_cast<int>(f_stack_.size()); }
  bool isContactActive(const int contact_index) const { return is_contact_active_.at(contact_index); }
  const std::vector<bool>& isContactActive() const { return is_contact_active_; }
  bool hasActiveContacts() const { return f_stack_.size() > 0; }      // split_solution.hxx:52-54

  // fill from one record of the C ABI (idocp_ocp_get_split_solution / idocp_unocp_get_split_solution):
  // lmd gmm q v a u beta [f mu nu_passive]
  // lmd gmm q v a u beta [f mu nu_passive]; active[ncontacts] (or null: no contact active) = the stage's contact status
  // (idocp_ocp_get_stage_contact_status)
  void assign(const double* rec, int nq, int nv, int nu, int ncontacts, int npassive, const int* active = nullptr) {
    const double* p = rec;
    auto take = [&](Eigen::VectorXd& dst, int n) { dst.resize(n); for (int i = 0; i < n; ++i) dst[i] = p[i]; p += n; };
    take(lmd, nv); take(gmm, nv); take(q, nq); take(v, nv); take(a, nv); take(u, nu); take(beta, nv);
    Eigen::VectorXd f_all, mu_all;
    take(f_all, 3 * ncontacts); take(mu_all, 3 * ncontacts); take(nu_passive, npassive);
    f.resize(ncontacts); mu.resize(ncontacts);
    is_contact_active_.assign(ncontacts, false);
    int dimf = 0;
    for (int c = 0; c < ncontacts; ++c) {
      for (int k = 0; k < 3; ++k) { f[c][k] = f_all[3 * c + k]; mu[c][k] = mu_all[3 * c + k]; }
      if (active && active[c]) { is_contact_active_[c] = true; dimf += 3; }
    }
    f_stack_.resize(dimf); mu_stack_.resize(dimf);
    int row = 0;
    for (int c = 0; c < ncontacts; ++c) if (is_contact_active_[c]) {
      for (int k = 0; k < 3; ++k) { f_stack_[row + k] = f[c][k]; mu_stack_[row + k] = mu[c][k]; }
      row += 3;
    }
  }

 private:
  Eigen::VectorXd f_stack_, mu_stack_;
  std::vector<bool> is_contact_active_;
};

using SplitSolutionOCP = SplitSolution;      // (rounds 1-3 had a second struct for the floating-base solvers)

}  // namespace idocp
#endif  // IDOCP_SPLIT_SOLUTION_HPP_
