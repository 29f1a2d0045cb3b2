// idocp::ContactForceCost -- facade (include/idocp/cost/contact_force_cost.hpp:15-100,
// src/cost/contact_force_cost.cpp:153-194 of the reference): 1/2 (f - f_ref)^T W (f - f_ref)
// per active contact; evaluated inside K5b / K8.
#ifndef IDOCP_CONTACT_FORCE_COST_HPP_
#define IDOCP_CONTACT_FORCE_COST_HPP_

#include <vector>

#include "idocp/cost/cost_function.hpp"

namespace idocp {

class ContactForceCost final : public CostFunctionComponentBase {
 public:
  explicit ContactForceCost(const Robot& robot) : n_(robot.maxPointContacts()) { idocp_cost_init(&c_); }
  Kind kind() const override { return ContactForce; }
  void set_f_ref(const std::vector<Eigen::Vector3d>& f_ref) { put(c_.f_ref, f_ref, "f_ref"); }
  // f_ref = (0, 0, weight / max_point_contacts) for every contact (contact_force_cost.cpp:76-86)
  void set_f_ref(const Robot& robot) {
    for (int i = 0; i < n_; ++i) { c_.f_ref[i][0] = 0.0; c_.f_ref[i][1] = 0.0; c_.f_ref[i][2] = robot.totalWeight() / n_; }
  }
  void set_f_weight(const std::vector<Eigen::Vector3d>& f_weight) { put(c_.f_weight, f_weight, "f_weight"); }
  // impulse-force terms (contact_force_cost.cpp:167-211): act on the impulse stages
  void set_fi_ref(const std::vector<Eigen::Vector3d>& fi_ref) { put(c_.fi_ref, fi_ref, "fi_ref"); }
  void set_fi_weight(const std::vector<Eigen::Vector3d>& fi_weight) { put(c_.fi_weight, fi_weight, "fi_weight"); }
  bool exportTo(idocp_cost_t& cost) const override {
    for (int i = 0; i < n_; ++i) for (int k = 0; k < 3; ++k) {
      cost.f_weight[i][k] = c_.f_weight[i][k]; cost.f_ref[i][k] = c_.f_ref[i][k];
      cost.fi_weight[i][k] = c_.fi_weight[i][k]; cost.fi_ref[i][k] = c_.fi_ref[i][k];
    }
    return true;
  }

 private:
  int n_;
  idocp_cost_t c_;
  void put(double (*dst)[3], const std::vector<Eigen::Vector3d>& v, const char* name) {
    if ((int)v.size() != n_) {   // contact_force_cost.cpp:60-150: throw -> catch -> exit
      std::cerr << "invalid size: " << name << ".size() must be " << n_ << "!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    for (int i = 0; i < n_; ++i) for (int k = 0; k < 3; ++k) dst[i][k] = v[i][k];
  }
};

}  // namespace idocp
#endif  // IDOCP_CONTACT_FORCE_COST_HPP_
