// ANYmal ParNMPCSolver benchmark through the drop-in facade: the driver of the reference's
// examples/anymal/parnmpc_benchmark.cpp:25-125 (N = 20, T = 0.5, 4 active point contacts), with the
// LinearizedFrictionCone in place of the nonlinear one.
//
//   usage: anymal_parnmpc_benchmark <path/to/anymal.urdf> [num_iteration]
#include <iostream>
#include <memory>
#include <string>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/contact_force_cost.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/ocp/parnmpc_solver.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/utils/ocp_benchmarker.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " <anymal.urdf> [num_iteration]" << std::endl;
    return 2;
  }
  const int num_iteration = (argc > 2) ? std::atoi(argv[2]) : 1000;
  std::vector<int> contact_frames = {14, 24, 34, 44};
  idocp::Robot robot(argv[1], contact_frames);
  auto cost = std::make_shared<idocp::CostFunction>();
  Eigen::VectorXd q_ref(robot.dimq());
  q_ref << 0, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0;
  auto config_cost = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  config_cost->set_q_weight(Eigen::VectorXd::Constant(robot.dimv(), 10));
  config_cost->set_q_ref(q_ref);
  config_cost->set_qf_weight(Eigen::VectorXd::Constant(robot.dimv(), 10));
  config_cost->set_v_weight(Eigen::VectorXd::Constant(robot.dimv(), 1));
  config_cost->set_vf_weight(Eigen::VectorXd::Constant(robot.dimv(), 1));
  config_cost->set_a_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.01));
  auto contact_cost = std::make_shared<idocp::ContactForceCost>(robot);
  std::vector<Eigen::Vector3d> f_weight(contact_frames.size(), Eigen::Vector3d(0.001, 0.001, 0.001));
  std::vector<Eigen::Vector3d> f_ref(contact_frames.size(), Eigen::Vector3d(0, 0, 70));
  contact_cost->set_f_weight(f_weight);
  contact_cost->set_f_ref(f_ref);
  cost->push_back(config_cost);
  cost->push_back(contact_cost);
  auto constraints = std::make_shared<idocp::Constraints>();
  constraints->push_back(std::make_shared<idocp::JointPositionLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointPositionUpperLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointVelocityLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointVelocityUpperLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointTorquesLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointTorquesUpperLimit>(robot));
  constraints->push_back(std::make_shared<idocp::LinearizedFrictionCone>(robot, 0.7));

  const double T = 0.5;
  const int N = 20;
  idocp::ParNMPCSolver parnmpc_solver(robot, cost, constraints, T, N, 4, 4);
  const double t = 0;
  Eigen::VectorXd q = q_ref;
  Eigen::VectorXd v = Eigen::VectorXd::Zero(robot.dimv());
  auto contact_status = robot.createContactStatus();
  contact_status.activateContacts({0, 1, 2, 3});
  robot.updateFrameKinematics(q);
  robot.setContactPoints(contact_status);
  parnmpc_solver.setContactStatusUniformly(contact_status);
  parnmpc_solver.setSolution("q", q);
  parnmpc_solver.setSolution("v", v);
  Eigen::Vector3d f_init(0, 0, 0.25 * robot.totalWeight());
  parnmpc_solver.setSolution("f", f_init);
  parnmpc_solver.initBackwardCorrection(t);
  parnmpc_solver.initConstraints(t);
  idocp::ocpbenchmarker::Convergence(parnmpc_solver, t, q, v, 20, false);
  idocp::ocpbenchmarker::CPUTime(parnmpc_solver, t, q, v, num_iteration, false);
  return 0;
}
