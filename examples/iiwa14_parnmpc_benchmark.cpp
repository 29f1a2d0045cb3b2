// iiwa14 reaching through idocp::ParNMPCSolver on a fixed-base arm without contact frames: convergence and time per update.
// Workload: the one of the reference's examples/iiwa14/parnmpc_benchmark.cpp (cost and limits of iiwa14_ocp_benchmark; N = 20, T = 1,
// 100 iterations after initBackwardCorrection).  The facade binds such a robot to the kernels of UnParNMPCSolver
// (include/idocp/ocp/parnmpc_solver.hpp).  Like iiwa14_ocp_benchmark -- and unlike the reference's driver, whose slack and dual
// variables stay at zero -- it initialises the constraints before the first iteration.
//   usage: iiwa14_parnmpc_benchmark [iiwa14.urdf] [iterations timed]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/ocp/parnmpc_solver.hpp"
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

  const int horizon = 20, no_impulse = 0;
  idocp::ParNMPCSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 1.0, horizon, no_impulse, 4);
  const ex::Vec q = ex::filled(robot.dimq(), 2), v = ex::Vec::Zero(n);
  solver.setSolution("q", q);
  solver.setSolution("v", v);
  solver.initConstraints(0.0);
  solver.initBackwardCorrection(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, q, v, 100, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, q, v, ex::argInt(argc, argv, 2, 1000), false);
  std::cout << "feasible: " << (solver.isCurrentSolutionFeasible() ? "yes" : "no") << ", q at the last stage: " << solver.getSolution(horizon - 1).q
            << std::endl;
  return 0;
}
