// idocp::TimeVaryingConfigurationSpaceCost -- facade
// (include/idocp/cost/time_varying_configuration_space_cost.hpp:20-127,
// src/cost/time_varying_configuration_space_cost.cpp of the reference): the
// configuration-space cost whose reference moves with a constant velocity inside
// a time window, q_ref(t) = q_begin (+) (t - t_begin) v_ref, and rests before and
// after it; the velocity reference is v_ref inside the window and zero outside.
// The per-stage references are generated on the host when the horizon is
// discretised and read by K5b / K8.
#ifndef IDOCP_TIME_VARYING_CONFIGURATION_SPACE_COST_HPP_
#define IDOCP_TIME_VARYING_CONFIGURATION_SPACE_COST_HPP_

#include "idocp/cost/cost_function.hpp"

namespace idocp {

class TimeVaryingConfigurationSpaceCost final : public CostFunctionComponentBase {
 public:
  explicit TimeVaryingConfigurationSpaceCost(const Robot& robot) : dimq_(robot.dimq()), dimv_(robot.dimv()) {
    idocp_cost_init(&c_);
    c_.use_time_varying_ref = 1;
  }
  // time_varying_configuration_space_cost.cpp:58-85
  void set_ref(const Robot& /*robot*/, const double t_begin, const double t_end, const Eigen::VectorXd& q_begin,
               const Eigen::VectorXd& v) {
    if (t_begin >= t_end) {
      std::cerr << "invalid argment: t_begin < t_end must be hold!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    put(c_.q_ref, q_begin, dimq_, "q_begin");
    put(c_.v_ref, v, dimv_, "v");
    c_.tv_t_begin = t_begin; c_.tv_t_end = t_end;
  }
  void set_q_weight(const Eigen::VectorXd& v) { put(c_.q_weight, v, dimv_, "q_weight"); }
  void set_v_weight(const Eigen::VectorXd& v) { put(c_.v_weight, v, dimv_, "v_weight"); }
  void set_a_weight(const Eigen::VectorXd& v) { put(c_.a_weight, v, dimv_, "a_weight"); }
  void set_qf_weight(const Eigen::VectorXd& v) { put(c_.qf_weight, v, dimv_, "qf_weight"); }
  void set_vf_weight(const Eigen::VectorXd& v) { put(c_.vf_weight, v, dimv_, "vf_weight"); }
  void set_qi_weight(const Eigen::VectorXd& v) { put(c_.qi_weight, v, dimv_, "qi_weight"); }
  void set_vi_weight(const Eigen::VectorXd& v) { put(c_.vi_weight, v, dimv_, "vi_weight"); }
  void set_dvi_weight(const Eigen::VectorXd& v) { put(c_.dvi_weight, v, dimv_, "dvi_weight"); }
  bool exportTo(idocp_cost_t& cost) const override {
    idocp_cost_t keep = cost;
    cost = c_;
    for (int i = 0; i < IDOCP_MAX_CONTACTS; ++i) for (int k = 0; k < 3; ++k) {
      cost.f_weight[i][k] = keep.f_weight[i][k]; cost.f_ref[i][k] = keep.f_ref[i][k];
      cost.fi_weight[i][k] = keep.fi_weight[i][k]; cost.fi_ref[i][k] = keep.fi_ref[i][k];
    }
    keepTaskFields(keep, cost);
    return true;
  }

 private:
  int dimq_, dimv_;
  idocp_cost_t c_;
  static void put(double* dst, const Eigen::VectorXd& v, int n, const char* name) {
    if (v.size() != n) {
      std::cerr << "invalid size: " << name << ".size() must be " << n << "!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    for (int i = 0; i < n; ++i) dst[i] = v[i];
  }
};

}  // namespace idocp
#endif  // IDOCP_TIME_VARYING_CONFIGURATION_SPACE_COST_HPP_
