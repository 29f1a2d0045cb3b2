// ANYmal jumping on the HIP path, through idocp::OCPSolver.
// Workload: the one of the reference's examples/anymal/anymal_jumping.cpp -- ConfigurationSpaceCost around the standing posture,
// contact-force cost (f_ref = (0, 0, 70)), joint limits, friction cones with mu = 0.7; three jumps: all four feet leave the
// ground at t = 1 + k 1.15 (a lift stage, then 0.15 s of flight without any contact row) and land together 0.25 m further
// (an impulse stage with all twelve impulse rows and a twelve-row switching constraint two stages ahead); N = 100, T = 5.
//   usage: anymal_jumping <anymal.urdf> [iterations = 155]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"

int main(int argc, char** argv) {
  idocp::Robot robot(ex::needUrdf(argc, argv, "[iterations]"), ex::anymalFeet());
  const int iterations = ex::argInt(argc, argv, 2, 155);
  const double hop = 0.25, first_takeoff = 1.0, airborne = 0.15, grounded = 1.0;
  const int jumps = 3;
  const ex::Vec stand = ex::anymalStanding();

  auto posture = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  posture->set_q_ref(stand);
  ex::attachWeights(*posture, ex::runs({{3, 1}, {15, 10}}), ex::runs({{3, 0.01}, {15, 0.1}}), ex::filled(18, 0.01), true);
  const ex::V3 share(0, 0, 70);
  auto cost = std::make_shared<idocp::CostFunction>();
  cost->push_back(posture);
  cost->push_back(ex::forceCost(robot, ex::V3(1, 1, 0.1), true, &share));

  idocp::OCPSolver solver(robot, cost, ex::jointLimits(robot, 0.7, true), 5.0, 100, jumps, 4);

  ex::Schedule gait(ex::footholds(robot, stand));
  gait.add({0, 1, 2, 3}, 0.0);
  for (int k = 0; k < jumps; ++k) {
    const double takeoff = first_takeoff + k * (airborne + grounded);
    gait.add({}, takeoff);                       // the footholds of the flight phase are the ones just left
    gait.advance({0, 1, 2, 3}, hop);
    gait.add({0, 1, 2, 3}, takeoff + airborne);
  }
  gait.install(solver, robot);

  ex::restingGuess(solver, robot, stand);
  solver.initConstraints(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, stand, ex::Vec::Zero(robot.dimv()), iterations, false);
  return 0;
}
