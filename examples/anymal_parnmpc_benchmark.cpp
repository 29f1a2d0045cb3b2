// ANYmal standing on four feet, idocp::ParNMPCSolver on the HIP path: convergence and time per update.
// Workload: the one of the reference's examples/anymal/parnmpc_benchmark.cpp (same cost and constraints as
// anymal_ocp_benchmark; N = 20, T = 0.5, 20 iterations).
//   usage: anymal_parnmpc_benchmark <anymal.urdf> [timed updates = 1000]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/ocp/parnmpc_solver.hpp"

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

  idocp::ParNMPCSolver solver(robot, cost, ex::jointLimits(robot, 0.7, false, true), 0.5, 20, 4, 4);      // FrictionCone(robot, 0.7), as in the reference driver (:76)
  ex::Schedule standing(ex::footholds(robot, stand));
  standing.add({0, 1, 2, 3}, 0.0);
  standing.install(solver, robot);
  ex::restingGuess(solver, robot, stand);
  solver.initBackwardCorrection(0.0);
  solver.initConstraints(0.0);

  const ex::Vec v = ex::Vec::Zero(robot.dimv());
  idocp::ocpbenchmarker::Convergence(solver, 0.0, stand, v, 20, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, stand, v, timed, false);
  return 0;
}
