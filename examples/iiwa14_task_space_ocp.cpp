// iiwa14 end-effector tracking, idocp::UnOCPSolver with a TimeVaryingTaskSpace6DCost on the HIP path.
// Workload: the one of the reference's examples/iiwa14/task_space_ocp.cpp (the end-effector frame follows a circle of radius
// 0.1 m in the y-z plane with a fixed orientation; weights 1000 on the pose error, 0.01 on v and a; T = 6, N = 120).
//   usage: iiwa14_task_space_ocp [iiwa14.urdf]
#include <cmath>

#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/time_varying_task_space_6d_cost.hpp"
#include "idocp/unocp/unocp_solver.hpp"
#include "idocp/utils/joint_constraints_factory.hpp"

// the pose the end effector has to track at time t
class CircleRef final : public idocp::TimeVaryingTaskSpace6DRefBase {
 public:
  CircleRef() : radius_(0.1), centre_(0.546, 0, 0.76) {
    rotm_ << 0, 0, 1,
             0, 1, 0,
            -1, 0, 0;
  }
  void compute_q_6d_ref(const double t, pinocchio::SE3& se3_ref) const override {
    Eigen::Vector3d pos(centre_);
    pos.coeffRef(1) += radius_ * std::sin(M_PI * t);
    pos.coeffRef(2) += radius_ * std::cos(M_PI * t);
    se3_ref = pinocchio::SE3(rotm_, pos);
  }
 private:
  double radius_;
  Eigen::Vector3d centre_;
  Eigen::Matrix3d rotm_;
};

int main(int argc, char** argv) {
  idocp::Robot robot(argc > 1 ? argv[1] : "tests/golden/urdf/iiwa14.urdf");
  const int n = robot.dimv();
  robot.setJointEffortLimit(ex::filled(robot.dimu(), 50));
  robot.setJointVelocityLimit(ex::filled(n, M_PI_2));

  auto cost = std::make_shared<idocp::CostFunction>();
  auto smooth = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  smooth->set_q_weight(ex::filled(n, 0));
  smooth->set_qf_weight(ex::filled(n, 0));
  smooth->set_v_weight(ex::filled(n, 0.01));
  smooth->set_vf_weight(ex::filled(n, 0.01));
  smooth->set_a_weight(ex::filled(n, 0.01));
  smooth->set_u_weight(ex::filled(n, 0.0));
  cost->push_back(smooth);
  const int ee_frame = 22;
  auto track = std::make_shared<idocp::TimeVaryingTaskSpace6DCost>(robot, ee_frame, std::make_shared<CircleRef>());
  track->set_q_6d_weight(Eigen::Vector3d::Constant(1000), Eigen::Vector3d::Constant(1000));
  track->set_qf_6d_weight(Eigen::Vector3d::Constant(1000), Eigen::Vector3d::Constant(1000));
  cost->push_back(track);

  const double T = 6;
  const int horizon = 120;
  idocp::UnOCPSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), T, horizon, 4);
  ex::Vec q(robot.dimq());
  q << 0, M_PI_2, 0, M_PI_2, 0, M_PI_2, 0;
  const ex::Vec v = ex::Vec::Zero(n);
  solver.setSolution("q", q);
  solver.setSolution("v", v);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, q, v, 30, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, q, v, 1000, false);
  std::cout << "q at the terminal stage: " << solver.getSolution(horizon).q << std::endl;
  return 0;
}
