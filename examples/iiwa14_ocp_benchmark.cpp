// iiwa14 reaching through idocp::OCPSolver -- the solver of the contact path on a fixed-base arm that has no contact frame.
// Workload: the one of the reference's examples/iiwa14/ocp_benchmark.cpp (= iiwa14_unocp_benchmark: every joint pulled towards
// -5 rad at -9 rad/s, joint limits with 200 Nm torque limits; N = 20, T = 1, start at 2 rad, 50 iterations).  The facade binds
// such a robot to the fixed-base kernels (include/idocp/ocp/ocp_solver.hpp), so the KKT errors printed here are those of
// iiwa14_unocp_benchmark; the torque feedback gain at the end is the one thing OCPSolver has and UnOCPSolver does not.
// One deviation from the reference's driver: it never calls initConstraints(t), which leaves slack = dual = 0
// (constraint_component_data.hxx:13-14) and divides 0 by 0 in the first condensation; this driver initialises them the way the
// reference's ANYmal drivers do.
//   usage: iiwa14_ocp_benchmark [iiwa14.urdf] [iterations timed]
#include "common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"
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
  idocp::OCPSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 1.0, horizon, no_impulse, 4);
  const ex::Vec q = ex::filled(robot.dimq(), 2), v = ex::Vec::Zero(n);
  solver.setSolution("q", q);
  solver.setSolution("v", v);
  solver.initConstraints(0.0);
  idocp::ocpbenchmarker::Convergence(solver, 0.0, q, v, 50, false);
  idocp::ocpbenchmarker::CPUTime(solver, 0.0, q, v, ex::argInt(argc, argv, 2, 1000), false);
  Eigen::MatrixXd Kq, Kv;
  solver.getStateFeedbackGain(0, Kq, Kv);
  std::cout << "du/dq(0, 0) of the policy at stage 0: " << Kq(0, 0) << ", du/dv(0, 0): " << Kv(0, 0) << std::endl;
  std::cout << "q at the terminal stage: " << solver.getSolution(horizon).q << std::endl;
  return 0;
}
