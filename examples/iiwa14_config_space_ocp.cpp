// iiwa14 swinging between two configurations, idocp::UnOCPSolver: the workload of the reference's examples/iiwa14/config_space_ocp.cpp
// (BASELINE.json configs[0]).  Torque limits 50 Nm, velocity limits pi / 2 rad/s; from (pi/2, 0, pi/2, 0, pi/2, 0, pi/2) to
// (0, pi/2, 0, pi/2, 0, pi/2, 0) with weights 10 / 0.01 / 0.01 on q / v / a; T = 3, N = 60, 30 iterations.  (The reference's viewer
// branch is not carried.)
//   usage: iiwa14_config_space_ocp [iiwa14.urdf] [file for the q trajectory]
#include <cmath>

#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/unocp/unocp_solver.hpp"
#include "idocp/utils/joint_constraints_factory.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(argc > 1 ? argv[1] : "tests/golden/urdf/iiwa14.urdf");
  const int n = robot.dimv();
  robot.setJointEffortLimit(ex::filled(robot.dimu(), 50));
  robot.setJointVelocityLimit(ex::filled(n, M_PI_2));
  ex::Vec from(n), to(n);
  for (int j = 0; j < n; ++j) { from[j] = (j % 2 == 0) ? M_PI_2 : 0.0; to[j] = (j % 2 == 0) ? 0.0 : M_PI_2; }
  auto swing = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  swing->set_q_ref(to);
  ex::attachWeights(*swing, ex::filled(n, 10), ex::filled(n, 0.01), ex::filled(n, 0.01), false);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(swing);

  const int horizon = 60;
  idocp::UnOCPSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 3.0, horizon, 4);
  const ex::Vec v = ex::Vec::Zero(n);
  solver.setSolution("q", from);
  solver.setSolution("v", v);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, from, v, 30, false);
  if (argc > 2) solver.saveSolution(argv[2], "q");
  std::cout << "feasible: " << (solver.isCurrentSolutionFeasible() ? "yes" : "no") << ", q at the terminal stage: " << solver.getSolution(horizon).q << std::endl;
  return 0;
}
