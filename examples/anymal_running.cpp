// ANYmal running (bounding with flight phases) through the drop-in facade: the driver of the reference's
// examples/anymal/anymal_running.cpp:29-231 -- TimeVaryingConfigurationSpaceCost + ContactForceCost, six joint limits,
// linearized (impulse) friction cones (mu = 0.8), contact sequence
//   all feet -> hind feet {LH, RH} -> flight -> front feet {LF, RF} -> hind feet -> ... -> all feet,
// N = 240, T = 7, up to 26 discrete events (touch-downs are impulse stages, lift-offs lift stages, flight phases have
// no contact rows at all).  BASELINE.json configs[4] runs the same schedule at N = 200.
//
//   usage: anymal_running <path/to/anymal.urdf> [sqp_iterations = 50] [N = 240]
#include <cstdlib>
#include <iostream>
#include <memory>
#include <string>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/contact_force_cost.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/cost/time_varying_configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/utils/ocp_benchmarker.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    std::cerr << "usage: " << argv[0] << " <anymal.urdf> [sqp_iterations] [N]" << std::endl;
    return 2;
  }
  const int num_iteration = argc > 2 ? std::atoi(argv[2]) : 50;
  const int N = argc > 3 ? std::atoi(argv[3]) : 240;
  std::vector<int> contact_frames = {14, 24, 34, 44};   // LF, LH, RF, RH
  idocp::Robot robot(argv[1], contact_frames);

  const double stride = 0.4;
  const double additive_stride_hip = 0.2;
  const double t_start = 1.0;
  const double t_front_swing = 0.135;
  const double t_front_hip_swing = 0.05;
  const double t_hip_swing = 0.165;
  const double t_period = t_front_swing + t_front_hip_swing + t_hip_swing;
  const int steps = 10;

  auto cost = std::make_shared<idocp::CostFunction>();
  Eigen::VectorXd q_standing(robot.dimq());
  q_standing << -3, 0, 0.4792, 0, 0, 0, 1, -0.1, 0.7, -1.0, -0.1, -0.7, 1.0, 0.1, 0.7, -1.0, 0.1, -0.7, 1.0;
  Eigen::VectorXd q_weight(robot.dimv());
  q_weight << 1, 1, 1, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10, 10;
  Eigen::VectorXd v_weight(robot.dimv());
  v_weight << 0.01, 0.01, 0.01, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1;
  Eigen::VectorXd a_weight = Eigen::VectorXd::Constant(robot.dimv(), 0.01);
  auto config_cost = std::make_shared<idocp::TimeVaryingConfigurationSpaceCost>(robot);
  Eigen::VectorXd v_ref = Eigen::VectorXd::Zero(robot.dimv());
  v_ref[0] = stride / t_period;
  config_cost->set_ref(robot, t_start, t_start + (0.5 + steps) * t_period, q_standing, v_ref);
  config_cost->set_q_weight(q_weight);
  config_cost->set_qf_weight(q_weight);
  config_cost->set_qi_weight(q_weight);
  config_cost->set_v_weight(v_weight);
  config_cost->set_vf_weight(v_weight);
  config_cost->set_vi_weight(v_weight);
  config_cost->set_a_weight(a_weight);
  config_cost->set_dvi_weight(a_weight);
  cost->push_back(config_cost);
  auto contact_cost = std::make_shared<idocp::ContactForceCost>(robot);
  std::vector<Eigen::Vector3d> f_weight(contact_frames.size(), Eigen::Vector3d(1e-01, 1e-01, 1.0e-07));
  std::vector<Eigen::Vector3d> f_ref(contact_frames.size(), Eigen::Vector3d(0, 0, 70));
  contact_cost->set_f_weight(f_weight);
  contact_cost->set_fi_weight(f_weight);
  contact_cost->set_f_ref(f_ref);
  cost->push_back(contact_cost);

  auto constraints = std::make_shared<idocp::Constraints>();
  constraints->push_back(std::make_shared<idocp::JointPositionLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointPositionUpperLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointVelocityLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointVelocityUpperLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointTorquesLowerLimit>(robot));
  constraints->push_back(std::make_shared<idocp::JointTorquesUpperLimit>(robot));
  const double mu = 0.8;
  constraints->push_back(std::make_shared<idocp::LinearizedFrictionCone>(robot, mu));
  constraints->push_back(std::make_shared<idocp::LinearizedImpulseFrictionCone>(robot, mu));

  const double T = 7;
  const int max_num_impulse_phase = (steps + 3) * 2;
  const int nthreads = 4;
  const double t = 0;
  idocp::OCPSolver ocp_solver(robot, cost, constraints, T, N, max_num_impulse_phase, nthreads);

  robot.updateFrameKinematics(q_standing);
  std::vector<Eigen::Vector3d> contact_points(robot.maxPointContacts(), Eigen::Vector3d::Zero());
  robot.getContactPoints(contact_points);
  auto contact_status_initial = robot.createContactStatus();
  contact_status_initial.activateContacts({0, 1, 2, 3});
  auto contact_status_front_swing = robot.createContactStatus();
  contact_status_front_swing.activateContacts({1, 3});
  auto contact_status_hip_swing = robot.createContactStatus();
  contact_status_hip_swing.activateContacts({0, 2});
  auto contact_status_front_hip_swing = robot.createContactStatus();     // flight: no active contact

  contact_status_initial.setContactPoints(contact_points);
  ocp_solver.setContactStatusUniformly(contact_status_initial);

  const double t_initial_front_swing = 0.125, t_initial_front_hip_swing = 0.05, t_initial_hip_swing = 0.125;
  const double t_initial = t_initial_front_swing + t_initial_front_hip_swing + t_initial_hip_swing;
  const double t_initial_front_swing2 = 0.135, t_initial_front_hip_swing2 = 0.055, t_initial_hip_swing2 = 0.15;
  const double t_initial2 = t_initial_front_swing2 + t_initial_front_hip_swing2 + t_initial_hip_swing2;

  contact_status_front_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_front_swing, t_start);
  contact_status_front_hip_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_front_hip_swing, t_start + t_initial_front_swing);

  contact_points[0].coeffRef(0) += 0.25 * stride;
  contact_points[1].coeffRef(0) += 0.25 * stride + 0.5 * additive_stride_hip;
  contact_points[2].coeffRef(0) += 0.25 * stride;
  contact_points[3].coeffRef(0) += 0.25 * stride + 0.5 * additive_stride_hip;
  contact_status_hip_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_hip_swing, t_start + t_initial_front_swing + t_initial_front_hip_swing);

  contact_status_front_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_front_swing, t_start + t_initial);
  contact_status_front_hip_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_front_hip_swing, t_start + t_initial + t_initial_front_swing2);

  contact_points[0].coeffRef(0) += 0.5 * stride;
  contact_points[1].coeffRef(0) += 0.5 * stride + 0.5 * additive_stride_hip;
  contact_points[2].coeffRef(0) += 0.5 * stride;
  contact_points[3].coeffRef(0) += 0.5 * stride + 0.5 * additive_stride_hip;
  contact_status_hip_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_hip_swing, t_start + t_initial + t_initial_front_swing2 + t_initial_front_hip_swing2);
  const double t_end_init = t_start + t_initial + t_initial2;

  for (int i = 0; i < steps; ++i) {
    contact_status_front_swing.setContactPoints(contact_points);
    ocp_solver.pushBackContactStatus(contact_status_front_swing, t_end_init + i * t_period);
    ocp_solver.pushBackContactStatus(contact_status_front_hip_swing, t_end_init + i * t_period + t_front_swing);
    for (int c = 0; c < 4; ++c) contact_points[c].coeffRef(0) += stride;
    contact_status_hip_swing.setContactPoints(contact_points);
    ocp_solver.pushBackContactStatus(contact_status_hip_swing, t_end_init + i * t_period + t_front_swing + t_front_hip_swing);
  }
  contact_status_front_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_front_swing, t_end_init + steps * t_period);

  const double t_end_front_swing = 0.15, t_end_front_hip_swing = 0.05, t_end_hip_swing = 0.15;
  const double t_end = t_end_front_swing + t_end_front_hip_swing + t_end_hip_swing;
  ocp_solver.pushBackContactStatus(contact_status_front_hip_swing, t_end_init + steps * t_period + t_end_front_swing);
  contact_points[0].coeffRef(0) += 0.5 * stride;
  contact_points[2].coeffRef(0) += 0.5 * stride;
  contact_points[1].coeffRef(0) += 0.5 * stride - additive_stride_hip;
  contact_points[3].coeffRef(0) += 0.5 * stride - additive_stride_hip;
  contact_status_hip_swing.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_hip_swing, t_end_init + steps * t_period + t_end_front_swing + t_end_front_hip_swing);
  contact_status_initial.setContactPoints(contact_points);
  ocp_solver.pushBackContactStatus(contact_status_initial, t_end_init + steps * t_period + t_end);

  Eigen::VectorXd q = q_standing;
  Eigen::VectorXd v = Eigen::VectorXd::Zero(robot.dimv());
  ocp_solver.setSolution("q", q);
  ocp_solver.setSolution("v", v);
  Eigen::Vector3d f_init(0, 0, 0.25 * robot.totalWeight());
  ocp_solver.setSolution("f", f_init);
  ocp_solver.initConstraints(t);

  const bool line_search = false;
  idocp::ocpbenchmarker::Convergence(ocp_solver, t, q, v, num_iteration, line_search);
  return 0;
}
