// idocp::TrottingConfigurationSpaceCost -- facade
// (include/idocp/cost/trotting_configuration_space_cost.hpp:19-164,
// src/cost/trotting_configuration_space_cost.cpp of the reference): the
// configuration-space cost whose q_ref follows a trotting gait in time.  The
// per-stage reference configurations are generated on the host when the solver
// is updated (idocp_ocp_update_solution) and read by K5b / K8.
#ifndef IDOCP_TROTTING_CONFIGURATION_SPACE_COST_HPP_
#define IDOCP_TROTTING_CONFIGURATION_SPACE_COST_HPP_

#include "idocp/cost/cost_function.hpp"

namespace idocp {

struct TrottingSwingAngles {   // trotting_configuration_space_cost.hpp:12-17
  double front_swing_thigh = 0, front_swing_knee = 0, hip_swing_thigh = 0, hip_swing_knee = 0;
  double front_stance_thigh = 0, front_stance_knee = 0, hip_stance_thigh = 0, hip_stance_knee = 0;
};

class TrottingConfigurationSpaceCost final : public CostFunctionComponentBase {
 public:
  explicit TrottingConfigurationSpaceCost(const Robot& robot) : dimq_(robot.dimq()), dimv_(robot.dimv()) {
    idocp_cost_init(&c_);
    c_.use_trotting_ref = 1;
  }
  void set_ref(const double t_start, const double t_period, const Eigen::VectorXd& q_standing, const double step_length,
               const TrottingSwingAngles& swing_angles) {
    if (t_period <= 0) {
      std::cerr << "invalid argument: t_period must be positive!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    put(c_.q_ref, q_standing, dimq_, "q_standing");
    c_.t_start = t_start; c_.t_period = t_period; c_.step_length = step_length;
    c_.v_ref[0] = step_length / t_period;           // trotting_configuration_space_cost.cpp set_ref
    c_.front_swing_knee = swing_angles.front_swing_knee; c_.hip_swing_knee = swing_angles.hip_swing_knee;
    c_.front_stance_knee = swing_angles.front_stance_knee; c_.hip_stance_knee = swing_angles.hip_stance_knee;
    // the thigh angles are unused by the reference as well (update_q_ref, :126-164)
  }
  void set_q_weight(const Eigen::VectorXd& v) { put(c_.q_weight, v, dimv_, "q_weight"); }
  void set_v_weight(const Eigen::VectorXd& v) { put(c_.v_weight, v, dimv_, "v_weight"); }
  void set_a_weight(const Eigen::VectorXd& v) { put(c_.a_weight, v, dimv_, "a_weight"); }
  void set_qf_weight(const Eigen::VectorXd& v) { put(c_.qf_weight, v, dimv_, "qf_weight"); }
  void set_vf_weight(const Eigen::VectorXd& v) { put(c_.vf_weight, v, dimv_, "vf_weight"); }
  // impulse-stage weights (computeImpulseCostDerivatives / Hessian, trotting_configuration_space_cost.cpp:308-376)
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
#endif  // IDOCP_TROTTING_CONFIGURATION_SPACE_COST_HPP_
