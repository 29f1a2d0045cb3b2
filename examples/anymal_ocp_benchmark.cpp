// ANYmal OCPSolver benchmark through the drop-in facade: same driver shape as the reference's
// examples/anymal/ocp_benchmark.cpp:25-121 (ConfigurationSpaceCost + ContactForceCost, six joint
// limits, friction cone, 4 active point contacts on every stage, N=20, T=0.5) -- with the
// LinearizedFrictionCone (the component the HIP path carries) in place of the nonlinear one.
//
//   usage: anymal_ocp_benchmark <path/to/anymal.urdf> [num_iteration]
#include <iostream>
#include <memory>
#include <string>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/contact_force_cost.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/ocp/ocp_solver.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/utils/ocp_benchmarker.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " <anymal.urdf> [num_iteration]" << std::endl;
    return 2;
  }
  const int num_iteration = (argc > 2) ? std::atoi(argv[2]) : 1000;
  std::vector<int> contact_frames = {14, 24, 34, 44};   // LF, LH, RF, RH
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
  std::vector<Eigen::Vector3d> f_weight, f_ref;
  for (size_t i = 0; i < contact_frames.size(); ++i) {
    f_weight.push_back(Eigen::Vector3d(0.001, 0.001, 0.001));
    f_ref.push_back(Eigen::Vector3d(0, 0, 70));
  }
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
  const double mu = 0.7;
  constraints->push_back(std::make_shared<idocp::LinearizedFrictionCone>(robot, mu));

  const double T = 0.5;
  const int N = 20;
  const int max_num_impulse_phase = 4;
  const int nthreads = 4;
  idocp::OCPSolver ocp_solver(robot, cost, constraints, T, N, max_num_impulse_phase, nthreads);

  const double t = 0;
  Eigen::VectorXd q = q_ref;
  Eigen::VectorXd v = Eigen::VectorXd::Zero(robot.dimv());

  auto contact_status = robot.createContactStatus();
  contact_status.activateContacts({0, 1, 2, 3});
  robot.updateFrameKinematics(q);
  robot.setContactPoints(contact_status);
  ocp_solver.setContactStatusUniformly(contact_status);
  ocp_solver.setSolution("q", q);
  ocp_solver.setSolution("v", v);
  Eigen::Vector3d f_init(0, 0, 0.25 * robot.totalWeight());
  ocp_solver.setSolution("f", f_init);
  ocp_solver.initConstraints(t);

  idocp::ocpbenchmarker::Convergence(ocp_solver, t, q, v, 10, false);
  idocp::ocpbenchmarker::CPUTime(ocp_solver, t, q, v, num_iteration, false);
  return 0;
}
