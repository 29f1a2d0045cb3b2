// idocp::CostFunction / ConfigurationSpaceCost -- facade.
//
// The reference's cost is an open plug-in system (CostFunctionComponentBase with
// virtual dispatch, include/idocp/cost/cost_function.hxx).  The HIP path
// evaluates the cost inside the stage kernels, so only the components it carries
// natively can be pushed: one configuration-space cost (ConfigurationSpaceCost,
// src/cost/configuration_space_cost.cpp:241-397, or TrottingConfigurationSpaceCost,
// src/cost/trotting_configuration_space_cost.cpp), one ContactForceCost
// (src/cost/contact_force_cost.cpp:153-194) and up to four task-space costs
// (task_space_cost.hpp; one of them may be time-varying).  Anything else -- or a
// second component of one of the first two kinds -- is rejected at push_back:
// there is no silent CPU fallback.
#ifndef IDOCP_COST_FUNCTION_HPP_
#define IDOCP_COST_FUNCTION_HPP_

#include <cstdlib>
#include <iostream>
#include <memory>
#include <vector>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp_hip.h"

namespace idocp {

class CostFunctionComponentBase {
 public:
  virtual ~CostFunctionComponentBase() {}
  enum Kind { ConfigurationSpace = 0, ContactForce = 1, TaskSpace = 2 };
  virtual Kind kind() const { return ConfigurationSpace; }
  // writes this component's parameters into its fields of the flat cost block;
  // false = cannot be represented
  virtual bool exportTo(idocp_cost_t& cost) const = 0;
  // time-varying task-space costs: the reference poses at t + i dt, i = 0 .. N ([N + 1][12]); false = constant reference
  virtual bool stageRefs(const double /*t*/, const double /*dt*/, const int /*N*/, std::vector<double>& /*refs*/) const { return false; }
  // the same at arbitrary times (the stages of a chain with discrete events, idocp_ocp_get_chain_times): refs[times.size()][12]
  virtual bool refsAt(const std::vector<double>& /*times*/, std::vector<double>& /*refs*/) const { return false; }
};

// the task-space fields of the flat cost block (written by the components of task_space_cost.hpp)
inline void keepTaskFields(const idocp_cost_t& from, idocp_cost_t& to) {
  to.task_dim = from.task_dim; to.task_joint = from.task_joint; to.task_time_varying = from.task_time_varying;
  for (int k = 0; k < 9; ++k) to.task_frame_R[k] = from.task_frame_R[k];
  for (int k = 0; k < 3; ++k) to.task_frame_p[k] = from.task_frame_p[k];
  for (int k = 0; k < 6; ++k) { to.task_weight[k] = from.task_weight[k]; to.task_weightf[k] = from.task_weightf[k]; to.task_weighti[k] = from.task_weighti[k]; }
  for (int k = 0; k < 12; ++k) to.task_ref[k] = from.task_ref[k];
  to.task_extra_count = from.task_extra_count;
  for (int e = 0; e < IDOCP_MAX_EXTRA_TASKS; ++e) to.task_extra[e] = from.task_extra[e];
}
// the task_* block of a cost as a task_extra component (a second, third ... task-space cost pushed into one CostFunction)
inline idocp_task_component_t taskBlockAsComponent(const idocp_cost_t& c) {
  idocp_task_component_t t;
  t.dim = c.task_dim; t.joint = c.task_joint;
  for (int k = 0; k < 9; ++k) t.frame_R[k] = c.task_frame_R[k];
  for (int k = 0; k < 3; ++k) t.frame_p[k] = c.task_frame_p[k];
  for (int k = 0; k < 6; ++k) { t.weight[k] = c.task_weight[k]; t.weightf[k] = c.task_weightf[k]; t.weighti[k] = c.task_weighti[k]; }
  for (int k = 0; k < 12; ++k) t.ref[k] = c.task_ref[k];
  return t;
}

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
  // impulse-stage weights (computeImpulseCostDerivatives / Hessian, configuration_space_cost.cpp)
  void set_qi_weight(const Eigen::VectorXd& v) { put(c_.qi_weight, v, dimv_, "qi_weight"); }
  void set_vi_weight(const Eigen::VectorXd& v) { put(c_.vi_weight, v, dimv_, "vi_weight"); }
  void set_dvi_weight(const Eigen::VectorXd& v) { put(c_.dvi_weight, v, dimv_, "dvi_weight"); }
  bool exportTo(idocp_cost_t& cost) const override {
    idocp_cost_t keep = cost;                       // contact-force and task-space fields belong to other components
    cost = c_;
    for (int i = 0; i < IDOCP_MAX_CONTACTS; ++i) for (int k = 0; k < 3; ++k) {
      cost.f_weight[i][k] = keep.f_weight[i][k]; cost.f_ref[i][k] = keep.f_ref[i][k];
      cost.fi_weight[i][k] = keep.fi_weight[i][k]; cost.fi_ref[i][k] = keep.fi_ref[i][k];
    }
    keepTaskFields(keep, cost);
    return true;
  }

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
  CostFunction() : have_{false, false, false} { idocp_cost_init(&c_); }
  // CostFunction::push_back (cost_function.hxx:19-22).  The component is KEPT (shared with the driver, as in the reference): native() exports
  // it again on every call, so a weight or reference the driver changes through the component afterwards reaches the solver at its next call.
  void push_back(const std::shared_ptr<CostFunctionComponentBase>& c) {
    if (!place(c, c_, have_, task_)) {
      std::cerr << "unsupported cost: the HIP path carries one configuration-space cost, one ContactForceCost and up to " << 1 + IDOCP_MAX_EXTRA_TASKS
                << " task-space costs, one of them time-varying" << '\n';
      std::exit(EXIT_FAILURE);
    }
    comps_.push_back(c);
  }
  // CostFunction::clear (cost_function.hxx:25-27)
  void clear() { comps_.clear(); task_.reset(); have_[0] = have_[1] = have_[2] = false; idocp_cost_init(&c_); }
  // the flat cost block of the C ABI, exported afresh from the components in the order they were pushed
  const idocp_cost_t& native() const {
    idocp_cost_init(&c_);
    bool have[3] = {false, false, false};
    std::shared_ptr<CostFunctionComponentBase> task;
    for (const auto& c : comps_) place(c, c_, have, task);      // (accepted by push_back: the kind of a component does not change)
    return c_;
  }
  // reference poses of a time-varying task-space cost at the stage times; false: none pushed / constant reference
  bool taskRefs(const double t, const double dt, const int N, std::vector<double>& refs) const {
    return task_ ? task_->stageRefs(t, dt, N, refs) : false;
  }
  bool taskRefsAt(const std::vector<double>& times, std::vector<double>& refs) const { return task_ ? task_->refsAt(times, refs) : false; }

 private:
  bool have_[3];
  mutable idocp_cost_t c_;
  std::shared_ptr<CostFunctionComponentBase> task_;      // the task-space component in the task_* block (the one asked for reference poses)
  std::vector<std::shared_ptr<CostFunctionComponentBase>> comps_;
  // One component into the flat block: the first of its kind through exportTo; a second, third ... task-space component (the reference's
  // CostFunction takes any number of components, cost_function.hpp:67) into the task_extra block.  Only the component in the task_* block can be
  // time-varying (it is the one the solver asks for reference poses); a time-varying component pushed behind a constant one takes that place and
  // the constant one moves to task_extra.  false: cannot be represented.
  static bool place(const std::shared_ptr<CostFunctionComponentBase>& c, idocp_cost_t& blk, bool (&have)[3], std::shared_ptr<CostFunctionComponentBase>& task) {
    const int k = (int)c->kind();
    if (!(k == (int)CostFunctionComponentBase::TaskSpace && have[k])) {
      if (have[k] || !c->exportTo(blk)) return false;
      have[k] = true;
      if (k == (int)CostFunctionComponentBase::TaskSpace) task = c;
      return true;
    }
    idocp_cost_t one;
    idocp_cost_init(&one);
    if (!c->exportTo(one) || blk.task_extra_count >= IDOCP_MAX_EXTRA_TASKS || (one.task_time_varying && blk.task_time_varying)) return false;
    if (one.task_time_varying) {
      blk.task_extra[blk.task_extra_count++] = taskBlockAsComponent(blk);
      idocp_cost_t keep = blk;
      keepTaskFields(one, blk);
      blk.task_extra_count = keep.task_extra_count;
      for (int e = 0; e < IDOCP_MAX_EXTRA_TASKS; ++e) blk.task_extra[e] = keep.task_extra[e];
      task = c;
    } else {
      blk.task_extra[blk.task_extra_count++] = taskBlockAsComponent(one);
    }
    return true;
  }
};

}  // namespace idocp
#endif  // IDOCP_COST_FUNCTION_HPP_
