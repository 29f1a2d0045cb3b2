// ANYmal trotting on the HIP path, through idocp::ParNMPCSolver (backward-Euler stages, backward correction) on a horizon
// with a lift-off and a touch-down.  Workload: the cost of examples/anymal/anymal_trotting.cpp of the reference with the
// six joint limits (no friction cones), N = 60, T = 1.55, schedule all feet -> {LH, RF} at 0.5 s -> {LF, RH} at 1.0 s.
//   usage: anymal_trotting_parnmpc <anymal.urdf> [iterations = 200]
#include "common.hpp"
#include "idocp/cost/trotting_configuration_space_cost.hpp"
#include "idocp/ocp/parnmpc_solver.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(ex::needUrdf(argc, argv, "[iterations]"), ex::anymalFeet());
  const int iterations = ex::argInt(argc, argv, 2, 200);
  const double step = 0.15, t0 = 0.5, period = 0.5;
  const ex::Vec stand = ex::anymalStanding();

  auto gait_cost = std::make_shared<idocp::TrottingConfigurationSpaceCost>(robot);
  idocp::TrottingSwingAngles swing;
  swing.front_swing_knee = swing.hip_swing_knee = 1.7;
  gait_cost->set_ref(t0, period, stand, step, swing);
  ex::attachWeights(*gait_cost, ex::filled(18, 10), ex::runs({{6, 1}, {12, 0.1}}), ex::runs({{6, 0.1}, {12, 0.01}}), true);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(gait_cost);
  cost->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), true, nullptr));

  idocp::ParNMPCSolver solver(robot, cost, ex::jointLimits(robot), 1.55, 60, 3, 4);

  ex::Schedule gait(ex::footholds(robot, stand));
  gait.add({0, 1, 2, 3}, 0.0);
  gait.add({1, 2}, t0);
  gait.advance({0, 3}, 0.5 * step);
  gait.add({0, 3}, t0 + period);
  gait.install(solver, robot);

  ex::restingGuess(solver, robot, stand);
  solver.initConstraints(0.0);
  solver.initBackwardCorrection(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, stand, ex::Vec::Zero(robot.dimv()), iterations, false);
  return 0;
}
