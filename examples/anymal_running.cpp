// ANYmal running (a bound with flight phases) on the HIP path, through idocp::OCPSolver.
// Workload: the one of the reference's examples/anymal/anymal_running.cpp -- time-varying configuration reference moving at
// stride / period, contact-force cost, joint limits, friction cones with mu = 0.8; 40 discrete events (26 touch-downs as
// impulse stages, 14 lift-offs as lift stages, no contact rows during flight); N = 240, T = 7.  BASELINE.json configs[4]
// runs the same schedule at N = 200.
//   usage: anymal_running <anymal.urdf> [iterations = 50] [N = 240]
#include "common.hpp"
#include "idocp/cost/time_varying_configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"

namespace {
// one bound = hind stance {LH, RH} for `hind`, flight for `air`, front stance {LF, RF} for `front`
struct Bound { double hind, air, front; double length() const { return hind + air + front; } };
}

int main(int argc, char** argv) {
  idocp::Robot robot(ex::needUrdf(argc, argv, "[iterations] [N]"), ex::anymalFeet());
  const int iterations = ex::argInt(argc, argv, 2, 50), horizon = ex::argInt(argc, argv, 3, 240);
  const double stride = 0.4, hind_extra = 0.2, t0 = 1.0;
  const int bounds = 10;
  const Bound first{0.125, 0.05, 0.125}, second{0.135, 0.055, 0.15}, cruise{0.135, 0.05, 0.165}, last{0.15, 0.05, 0.15};
  const ex::Vec start = ex::anymalStanding(-3.0);

  auto motion_cost = std::make_shared<idocp::TimeVaryingConfigurationSpaceCost>(robot);
  ex::Vec speed = ex::Vec::Zero(robot.dimv());
  speed[0] = stride / cruise.length();
  motion_cost->set_ref(robot, t0, t0 + (0.5 + bounds) * cruise.length(), start, speed);
  ex::attachWeights(*motion_cost, ex::runs({{3, 1}, {15, 10}}), ex::runs({{3, 0.01}, {15, 0.1}}), ex::filled(18, 0.01), true);
  const ex::V3 share(0, 0, 70);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(motion_cost);
  cost->push_back(ex::forceCost(robot, ex::V3(1e-01, 1e-01, 1.0e-07), true, &share));

  idocp::OCPSolver solver(robot, cost, ex::jointLimits(robot, 0.8, true), 7.0, horizon, (bounds + 3) * 2, 4);

  // Footholds move while the feet are in the air, i.e. between the lift-off of the hind pair and the touch-down of the front
  // pair: front feet {0, 2} by `lf`, hind feet {1, 3} by `lh`.
  ex::Schedule gait(ex::footholds(robot, start));
  gait.add({0, 1, 2, 3}, 0.0);
  auto bound = [&gait](double begin, const Bound& b, double front_dx, double hind_dx) {
    gait.add({1, 3}, begin);
    gait.add({}, begin + b.hind);
    gait.advance({0, 2}, front_dx);
    gait.advance({1, 3}, hind_dx);
    gait.add({0, 2}, begin + b.hind + b.air);
  };
  bound(t0, first, 0.25 * stride, 0.25 * stride + 0.5 * hind_extra);
  bound(t0 + first.length(), second, 0.5 * stride, 0.5 * stride + 0.5 * hind_extra);
  const double cruise_begin = t0 + first.length() + second.length();
  for (int k = 0; k < bounds; ++k) bound(cruise_begin + k * cruise.length(), cruise, stride, stride);
  bound(cruise_begin + bounds * cruise.length(), last, 0.5 * stride, 0.5 * stride - hind_extra);
  gait.add({0, 1, 2, 3}, cruise_begin + bounds * cruise.length() + last.length());
  gait.install(solver, robot);

  ex::restingGuess(solver, robot, start);
  solver.initConstraints(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, start, ex::Vec::Zero(robot.dimv()), iterations, false);
  return 0;
}
