// ANYmal standing on four feet, idocp::OCPSolver on the HIP path: convergence and time per update.
// Workload: the one of the reference's examples/anymal/ocp_benchmark.cpp (configuration-space cost around the standing
// pose, contact-force cost around 70 N per foot, joint limits, FrictionCone mu = 0.7; N = 20, T = 0.5).
//   usage: anymal_ocp_benchmark <anymal.urdf> [timed updates = 1000]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(ex::needUrdf(argc, argv, "[timed updates]"), ex::anymalFeet());
  const int timed = ex::argInt(argc, argv, 2, 1000);
  const ex::Vec stand = ex::anymalStanding();

  auto pose_cost = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  pose_cost->set_q_ref(stand);
  ex::attachWeights(*pose_cost, ex::filled(18, 10), ex::filled(18, 1), ex::filled(18, 0.01), false);
  const ex::V3 share(0, 0, 70);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(pose_cost);
  cost->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), false, &share));

  idocp::OCPSolver solver(robot, cost, ex::jointLimits(robot, 0.7, false, true), 0.5, 20, 4, 4);      // FrictionCone(robot, 0.7), as in the reference driver (:76)
  ex::Schedule standing(ex::footholds(robot, stand));
  standing.add({0, 1, 2, 3}, 0.0);
  standing.install(solver, robot);
  ex::restingGuess(solver, robot, stand);
  solver.initConstraints(0.0);

  const ex::Vec v = ex::Vec::Zero(robot.dimv());
  idocp::ocpbenchmarker::Convergence(solver, 0.0, stand, v, 10, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, stand, v, timed, false);
  return 0;
}
