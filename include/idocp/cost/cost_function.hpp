// idocp::CostFunction / ConfigurationSpaceCost -- facade.
//
// The reference's cost is an open plug-in system (CostFunctionComponentBase with
// virtual dispatch, include/idocp/cost/cost_function.hxx).  The HIP path
// evaluates the cost inside the stage kernel, so only the components it carries
// natively can be pushed: ConfigurationSpaceCost
// (src/cost/configuration_space_cost.cpp:241-397).  Anything else is rejected
// at push_back -- there is no silent CPU fallback.
#ifndef IDOCP_COST_FUNCTION_HPP_
#define IDOCP_COST_FUNCTION_HPP_

#include <cstdlib>
#include <iostream>
#include <memory>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp_hip.h"

namespace idocp {

class CostFunctionComponentBase {
 public:
  virtual ~CostFunctionComponentBase() {}
  // adds this component's parameters to the flat cost block; false = cannot be represented
  virtual bool exportTo(idocp_cost_t& cost) const = 0;
};

class ConfigurationSpaceCost final : public CostFunctionComponentBase {
 public:
  explicit ConfigurationSpaceCost(const Robot& robot) : dimq_(robot.dimq()), dimv_(robot.dimv()), dimu_(robot.dimu()) {
    idocp_cost_init(&c_);
  }
  void set_q_ref(const Eigen::VectorXd& v) { put(c_.q_ref, v, dimq_, "q_ref"); }
  void set_v_ref(const Eigen::VectorXd& v) { put(c_.v_ref, v, dimv_, "v_ref"); }
  void set_u_ref(const Eigen::VectorXd& v) { put(c_.u_ref, v, dimu_, "u_ref"); }
  void set_q_weight(const Eigen::VectorXd& v) { put(c_.q_weight, v, dimv_, "q_weight"); }
  void set_v_weight(const Eigen::VectorXd& v) { put(c_.v_weight, v, dimv_, "v_weight"); }
  void set_a_weight(const Eigen::VectorXd& v) { put(c_.a_weight, v, dimv_, "a_weight"); }
  void set_u_weight(const Eigen::VectorXd& v) { put(c_.u_weight, v, dimu_, "u_weight"); }
  void set_qf_weight(const Eigen::VectorXd& v) { put(c_.qf_weight, v, dimv_, "qf_weight"); }
  void set_vf_weight(const Eigen::VectorXd& v) { put(c_.vf_weight, v, dimv_, "vf_weight"); }
  bool exportTo(idocp_cost_t& cost) const override { cost = c_; return true; }

 private:
  int dimq_, dimv_, dimu_;
  idocp_cost_t c_;
  static void put(double* dst, const Eigen::VectorXd& v, int n, const char* name) {
    if (v.size() != n) {   // configuration_space_cost.cpp:64-240: throw -> catch -> exit
      std::cerr << "invalid size: " << name << ".size() must be " << n << "!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    for (int i = 0; i < n; ++i) dst[i] = v[i];
  }
};

class CostFunction {
 public:
  CostFunction() : n_(0) { idocp_cost_init(&c_); }
  void push_back(const std::shared_ptr<CostFunctionComponentBase>& c) {
    if (n_ > 0 || !c->exportTo(c_)) {
      std::cerr << "unsupported cost: the HIP path carries exactly one ConfigurationSpaceCost component" << '\n';
      std::exit(EXIT_FAILURE);
    }
    ++n_;
  }
  const idocp_cost_t& native() const { return c_; }

 private:
  int n_;
  idocp_cost_t c_;
};

}  // namespace idocp
#endif  // IDOCP_COST_FUNCTION_HPP_
