// Driver for the drop-in facade: the workload of the reference's
// examples/iiwa14/unparnmpc_benchmark.cpp (same public API calls), running on the
// HIP path.  Build: make -C examples.   Usage: ./iiwa14_unparnmpc_benchmark <urdf>
#include <iostream>
#include <memory>
#include <string>

#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/unocp/unparnmpc_solver.hpp"
#include "idocp/utils/joint_constraints_factory.hpp"
#include "idocp/utils/ocp_benchmarker.hpp"

int main(int argc, char** argv) {
  const std::string path_to_urdf = argc > 1 ? argv[1] : "tests/golden/urdf/iiwa14.urdf";
  idocp::Robot robot(path_to_urdf);
  robot.setJointEffortLimit(Eigen::VectorXd::Constant(robot.dimu(), 200));
  auto cost = std::make_shared<idocp::CostFunction>();
  auto config_cost = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
  config_cost->set_q_ref(Eigen::VectorXd::Constant(robot.dimv(), -5));
  config_cost->set_v_ref(Eigen::VectorXd::Constant(robot.dimv(), -9));
  config_cost->set_q_weight(Eigen::VectorXd::Constant(robot.dimv(), 10));
  config_cost->set_qf_weight(Eigen::VectorXd::Constant(robot.dimv(), 10));
  config_cost->set_v_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.1));
  config_cost->set_vf_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.1));
  config_cost->set_a_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.01));
  config_cost->set_u_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.0));
  cost->push_back(config_cost);
  idocp::JointConstraintsFactory constraints_factory(robot);
  auto constraints = constraints_factory.create();

  const double T = 1;
  const int N = 20;
  const int nthreads = 4;
  const double t = 0;
  const Eigen::VectorXd q = Eigen::VectorXd::Constant(robot.dimq(), 2);
  const Eigen::VectorXd v = Eigen::VectorXd::Zero(robot.dimv());
  idocp::UnParNMPCSolver parnmpc_solver(robot, cost, constraints, T, N, nthreads);
  parnmpc_solver.setSolution("q", q);
  parnmpc_solver.setSolution("v", v);
  parnmpc_solver.initBackwardCorrection(t);
  idocp::ocpbenchmarker::Convergence(parnmpc_solver, t, q, v, 100, false);
  idocp::ocpbenchmarker::CPUTime(parnmpc_solver, t, q, v, 1000, false);
  if (argc > 2) {                      // solution I/O of the reference (unparnmpc_solver.cpp:286-327)
    parnmpc_solver.saveSolution(std::string(argv[2]) + "_q.txt", "q");
    parnmpc_solver.saveSolution(std::string(argv[2]) + "_u.txt", "u");
    parnmpc_solver.printSolution("v");
  }
  std::cout << "feasible: " << (parnmpc_solver.isCurrentSolutionFeasible() ? "yes" : "no")
            << ", q at the last stage: " << parnmpc_solver.getSolution(N - 1).q << std::endl;
  return 0;
}
