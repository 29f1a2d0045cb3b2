// iiwa14 reaching, idocp::UnParNMPCSolver on the HIP path: convergence, time per update, solution output.
// Workload: the one of the reference's examples/iiwa14/unparnmpc_benchmark.cpp (the cost and limits of
// iiwa14_unocp_benchmark; N = 20, T = 1, 100 iterations).
//   usage: iiwa14_unparnmpc_benchmark [iiwa14.urdf] [prefix of the solution files]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/unocp/unparnmpc_solver.hpp"
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
  idocp::UnParNMPCSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 1.0, horizon, 4);
  const ex::Vec q = ex::filled(robot.dimq(), 2), v = ex::Vec::Zero(n);
  solver.setSolution("q", q);
  solver.setSolution("v", v);
  solver.initBackwardCorrection(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, q, v, 100, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, q, v, 1000, false);
  if (argc > 2) {                      // text output of the stage-wise solution, one stage per line
    const std::string prefix(argv[2]);
    solver.saveSolution(prefix + "_q.txt", "q");
    solver.saveSolution(prefix + "_u.txt", "u");
    solver.printSolution("v");
  }
  std::cout << "feasible: " << (solver.isCurrentSolutionFeasible() ? "yes" : "no") << ", q at the last stage: " << solver.getSolution(horizon - 1).q
            << std::endl;
  return 0;
}
