// ANYmal trotting on the HIP path, through idocp::OCPSolver.
// Workload: the one of the reference's examples/anymal/anymal_trotting.cpp (trotting reference cost + contact-force cost,
// joint limits, friction cones with mu = 0.7 on stages and impulses; all feet -> {LH, RF} -> {LF, RH} -> ..., steps of
// 0.15 m every 0.5 s from t = 0.5; N = 30, T = 1.55, two touch-down phases, 25 iterations).
//   usage: anymal_trotting <anymal.urdf>
#include "common.hpp"
#include "idocp/cost/trotting_configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(ex::needUrdf(argc, argv), ex::anymalFeet());
  const double step = 0.15, t0 = 0.5, period = 0.5;
  const int touch_downs = 2, horizon = 30;
  const ex::Vec stand = ex::anymalStanding();

  auto gait_cost = std::make_shared<idocp::TrottingConfigurationSpaceCost>(robot);
  idocp::TrottingSwingAngles swing;
  swing.front_swing_knee = swing.hip_swing_knee = 1.7;
  gait_cost->set_ref(t0, period, stand, step, swing);
  ex::attachWeights(*gait_cost, ex::filled(18, 10), ex::runs({{6, 1}, {12, 0.1}}), ex::runs({{6, 0.1}, {12, 0.01}}), true);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(gait_cost);
  cost->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), true, nullptr));

  idocp::OCPSolver solver(robot, cost, ex::jointLimits(robot, 0.7, true), t0 + touch_downs * period + 0.05, horizon, touch_downs + 1, 4);

  // diagonal pairs alternate; the first swing covers half a step
  ex::Schedule gait(ex::footholds(robot, stand));
  gait.add({0, 1, 2, 3}, 0.0);
  gait.add({1, 2}, t0);
  for (int k = 1; k <= touch_downs; ++k) {
    const bool lf_rh_land = (k % 2 == 1);
    if (lf_rh_land) { gait.advance({0, 3}, k == 1 ? 0.5 * step : step); gait.add({0, 3}, t0 + k * period); }
    else { gait.advance({1, 2}, step); gait.add({1, 2}, t0 + k * period); }
  }
  gait.install(solver, robot);

  ex::restingGuess(solver, robot, stand);
  solver.initConstraints(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, stand, ex::Vec::Zero(robot.dimv()), 25, false);
  return 0;
}
