// idocp::ocpbenchmarker -- same protocol and output as the reference
// (include/idocp/utils/ocp_benchmarker.hxx:13-52).
#ifndef IDOCP_OCP_BENCHMARKER_HPP_
#define IDOCP_OCP_BENCHMARKER_HPP_

#include <chrono>
#include <iostream>

#include "idocp/eigen_shim.hpp"

namespace idocp {
namespace ocpbenchmarker {

template <typename OCPSolverType>
inline void CPUTime(OCPSolverType& ocp_solver, const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v,
                    const int num_iteration, const bool line_search = false) {
  const auto start_clock = std::chrono::system_clock::now();
  for (int i = 0; i < num_iteration; ++i) ocp_solver.updateSolution(t, q, v, line_search);
  const auto end_clock = std::chrono::system_clock::now();
  const double ms = 1e-03 * std::chrono::duration_cast<std::chrono::microseconds>(end_clock - start_clock).count();
  std::cout << "---------- OCP benchmark : CPU time ----------" << std::endl;
  std::cout << "total CPU time: " << ms << "[ms]" << std::endl;
  std::cout << "CPU time per update: " << ms / num_iteration << "[ms]" << std::endl;
  std::cout << "-----------------------------------" << std::endl << std::endl;
}

template <typename OCPSolverType>
inline void Convergence(OCPSolverType& ocp_solver, const double t, const Eigen::VectorXd& q, const Eigen::VectorXd& v,
                        const int num_iteration, const bool line_search = false) {
  std::cout << "---------- OCP benchmark : Convergence ----------" << std::endl;
  ocp_solver.computeKKTResidual(t, q, v);
  std::cout << "Initial KKT error = " << ocp_solver.KKTError() << std::endl;
  for (int i = 0; i < num_iteration; ++i) {
    ocp_solver.updateSolution(t, q, v, line_search);
    ocp_solver.computeKKTResidual(t, q, v);
    std::cout << "KKT error after iteration " << i + 1 << " = " << ocp_solver.KKTError() << std::endl;
  }
  std::cout << "-----------------------------------" << std::endl << std::endl;
}

}  // namespace ocpbenchmarker
}  // namespace idocp
#endif  // IDOCP_OCP_BENCHMARKER_HPP_
