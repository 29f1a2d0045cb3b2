// idocp::TaskSpace3DCost / TaskSpace6DCost / TimeVaryingTaskSpace3DCost / TimeVaryingTaskSpace6DCost -- facade
// (include/idocp/cost/task_space_3d_cost.hpp:17-120, task_space_6d_cost.hpp:20-125, time_varying_task_space_3d_cost.hpp:19-140,
// time_varying_task_space_6d_cost.hpp:21-150 of the reference).  The cost of the position (3D) or pose (6D, log6 of the pose
// error) of one frame of a fixed-base robot.  The HIP path evaluates it inside K1 (UnOCPSolver only; dev_task.hpp), so the
// component exports its frame, weights and reference into the flat cost block; the time-varying variants are asked by the
// solver for the reference at the time of every stage (CostFunction::taskRefs) before each update.
#ifndef IDOCP_TASK_SPACE_COST_HPP_
#define IDOCP_TASK_SPACE_COST_HPP_

#include <memory>
#include <vector>

#include "idocp/cost/cost_function.hpp"
#include "idocp/pinocchio_shim.hpp"

namespace idocp {

namespace taskcost {
// frame_id -> parent joint + placement (Robot::framePlacement needs them; robot.hxx:166-178)
inline void bindFrame(const Robot& robot, const int frame_id, const int dim, idocp_cost_t& c) {
  idocp_cost_init(&c);
  c.task_dim = dim;
  if (idocp_model_frame_placement(robot.pathToUrdf().c_str(), frame_id, &c.task_joint, c.task_frame_R, c.task_frame_p) != IDOCP_OK) {
    std::cerr << idocp_last_error() << '\n';
    std::exit(EXIT_FAILURE);
  }
  for (int k = 0; k < 9; ++k) c.task_ref[k] = (k % 4 == 0) ? 1.0 : 0.0;
}
// the reference stores [rotation_weight; position_weight] against log6 = [linear; angular] (task_space_6d_cost.cpp:48-60)
inline void put6(double* dst, const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) {
  for (int k = 0; k < 3; ++k) { dst[k] = rotation_weight[k]; dst[3 + k] = position_weight[k]; }
}
inline void putPose(double* dst, const pinocchio::SE3& M) {
  for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dst[3 * r + c] = M.rotation()(r, c);
  for (int k = 0; k < 3; ++k) dst[9 + k] = M.translation()[k];
}
}  // namespace taskcost

class TaskSpace3DCost final : public CostFunctionComponentBase {
 public:
  TaskSpace3DCost(const Robot& robot, const int frame_id) { taskcost::bindFrame(robot, frame_id, 3, c_); }
  Kind kind() const override { return TaskSpace; }
  void set_q_3d_ref(const Eigen::Vector3d& q_3d_ref) { for (int k = 0; k < 3; ++k) c_.task_ref[9 + k] = q_3d_ref[k]; }
  void set_q_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weight[k] = w[k]; }
  void set_qf_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weightf[k] = w[k]; }
  void set_qi_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weighti[k] = w[k]; }
  bool exportTo(idocp_cost_t& cost) const override { keepTaskFields(c_, cost); return true; }
 private:
  idocp_cost_t c_;
};

class TaskSpace6DCost final : public CostFunctionComponentBase {
 public:
  TaskSpace6DCost(const Robot& robot, const int frame_id) { taskcost::bindFrame(robot, frame_id, 6, c_); }
  Kind kind() const override { return TaskSpace; }
  void set_q_6d_ref(const Eigen::Vector3d& position_ref, const Eigen::Matrix3d& rotation_mat_ref) {
    taskcost::putPose(c_.task_ref, pinocchio::SE3(rotation_mat_ref, position_ref));
  }
  void set_q_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weight, position_weight, rotation_weight); }
  void set_qf_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weightf, position_weight, rotation_weight); }
  void set_qi_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weighti, position_weight, rotation_weight); }
  bool exportTo(idocp_cost_t& cost) const override { keepTaskFields(c_, cost); return true; }
 private:
  idocp_cost_t c_;
};

class TimeVaryingTaskSpace3DRefBase {
 public:
  TimeVaryingTaskSpace3DRefBase() {}
  virtual ~TimeVaryingTaskSpace3DRefBase() {}
  virtual void compute_q_3d_ref(const double t, Eigen::VectorXd& q_3d_ref) const = 0;
};

class TimeVaryingTaskSpace3DCost final : public CostFunctionComponentBase {
 public:
  TimeVaryingTaskSpace3DCost(const Robot& robot, const int frame_id, const std::shared_ptr<TimeVaryingTaskSpace3DRefBase>& ref) : ref_(ref) {
    taskcost::bindFrame(robot, frame_id, 3, c_);
    c_.task_time_varying = 1;
  }
  Kind kind() const override { return TaskSpace; }
  void set_ref(const std::shared_ptr<TimeVaryingTaskSpace3DRefBase>& ref) { ref_ = ref; }
  void set_q_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weight[k] = w[k]; }
  void set_qf_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weightf[k] = w[k]; }
  void set_qi_3d_weight(const Eigen::Vector3d& w) { for (int k = 0; k < 3; ++k) c_.task_weighti[k] = w[k]; }
  bool exportTo(idocp_cost_t& cost) const override { keepTaskFields(c_, cost); return true; }
  bool stageRefs(const double t, const double dt, const int N, std::vector<double>& refs) const override {
    std::vector<double> times((size_t)N + 1);
    for (int i = 0; i <= N; ++i) times[i] = t + i * dt;
    return refsAt(times, refs);
  }
  bool refsAt(const std::vector<double>& times, std::vector<double>& refs) const override {
    refs.assign(times.size() * 12, 0.0);
    Eigen::VectorXd p(3);
    for (size_t i = 0; i < times.size(); ++i) {
      ref_->compute_q_3d_ref(times[i], p);
      double* r = &refs[12 * i];
      r[0] = r[4] = r[8] = 1.0;
      for (int k = 0; k < 3; ++k) r[9 + k] = p[k];
    }
    return true;
  }
 private:
  idocp_cost_t c_;
  std::shared_ptr<TimeVaryingTaskSpace3DRefBase> ref_;
};

class TimeVaryingTaskSpace6DRefBase {
 public:
  TimeVaryingTaskSpace6DRefBase() {}
  virtual ~TimeVaryingTaskSpace6DRefBase() {}
  virtual void compute_q_6d_ref(const double t, pinocchio::SE3& se3_ref) const = 0;
};

class TimeVaryingTaskSpace6DCost final : public CostFunctionComponentBase {
 public:
  TimeVaryingTaskSpace6DCost(const Robot& robot, const int frame_id, const std::shared_ptr<TimeVaryingTaskSpace6DRefBase>& ref) : ref_(ref) {
    taskcost::bindFrame(robot, frame_id, 6, c_);
    c_.task_time_varying = 1;
  }
  Kind kind() const override { return TaskSpace; }
  void set_ref(const std::shared_ptr<TimeVaryingTaskSpace6DRefBase>& ref) { ref_ = ref; }
  void set_q_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weight, position_weight, rotation_weight); }
  void set_qf_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weightf, position_weight, rotation_weight); }
  void set_qi_6d_weight(const Eigen::Vector3d& position_weight, const Eigen::Vector3d& rotation_weight) { taskcost::put6(c_.task_weighti, position_weight, rotation_weight); }
  bool exportTo(idocp_cost_t& cost) const override { keepTaskFields(c_, cost); return true; }
  bool stageRefs(const double t, const double dt, const int N, std::vector<double>& refs) const override {
    std::vector<double> times((size_t)N + 1);
    for (int i = 0; i <= N; ++i) times[i] = t + i * dt;
    return refsAt(times, refs);
  }
  bool refsAt(const std::vector<double>& times, std::vector<double>& refs) const override {
    refs.assign(times.size() * 12, 0.0);
    pinocchio::SE3 M;
    for (size_t i = 0; i < times.size(); ++i) {
      ref_->compute_q_6d_ref(times[i], M);
      taskcost::putPose(&refs[12 * i], M);
    }
    return true;
  }
 private:
  idocp_cost_t c_;
  std::shared_ptr<TimeVaryingTaskSpace6DRefBase> ref_;
};

}  // namespace idocp
#endif  // IDOCP_TASK_SPACE_COST_HPP_
