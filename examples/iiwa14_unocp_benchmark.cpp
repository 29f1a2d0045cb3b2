// iiwa14 reaching, idocp::UnOCPSolver on the HIP path: convergence and time per update.
// Workload: the one of the reference's examples/iiwa14/unocp_benchmark.cpp (configuration-space cost pulling every joint
// towards -5 rad at -9 rad/s, joint limits with 200 Nm torque limits; N = 20, T = 1, start at 2 rad on every joint).
//   usage: iiwa14_unocp_benchmark [iiwa14.urdf]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/unocp/unocp_solver.hpp"
#include "idocp/utils/joint_constraints_factory.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(argc > 1 ? argv[1] : "tests/golden/urdf/iiwa14.urdf");
  const int n = robot.dimv();
  robot.setJointEffortLimit(ex::filled(robot.dimu(), 200));
  auto reach = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  reach->set_q_ref(ex::filled(n, -5));
  reach->set_v_ref(ex::filled(n, -9));
  ex::attachWeights(*reach, ex::filled(n, 10), ex::filled(n, 0.1), ex::filled(n, 0.01), false);
  reach->set_u_weight(ex::filled(n, 0.0));
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(reach);

  const int horizon = 20;
  idocp::UnOCPSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 1.0, horizon, 4);
  const ex::Vec q = ex::filled(robot.dimq(), 2), v = ex::Vec::Zero(n);
  solver.setSolution("q", q);
  solver.setSolution("v", v);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, q, v, 50, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, q, v, 1000, false);
  std::cout << "q at the terminal stage: " << solver.getSolution(horizon).q << std::endl;
  return 0;
}
