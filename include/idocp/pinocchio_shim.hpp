// pinocchio::SE3 stand-in for the facade when pinocchio is not installed.
//
// The reference's TimeVaryingTaskSpace6DRefBase hands the reference pose over as a pinocchio::SE3
// (include/idocp/cost/time_varying_task_space_6d_cost.hpp:21-42).  If <pinocchio/spatial/se3.hpp> is available it is used
// unchanged; otherwise this header provides the part the user's reference class touches: SE3(R, p), Identity(),
// rotation(), translation().
#ifndef IDOCP_PINOCCHIO_SHIM_HPP_
#define IDOCP_PINOCCHIO_SHIM_HPP_

#include "idocp/eigen_shim.hpp"

#if defined(__has_include)
#if __has_include(<pinocchio/spatial/se3.hpp>)
#include <pinocchio/spatial/se3.hpp>
#define IDOCP_HAVE_PINOCCHIO 1
#endif
#endif

#ifndef IDOCP_HAVE_PINOCCHIO
namespace pinocchio {

class SE3 {
 public:
  SE3() : R_(Eigen::Matrix3d::Identity()), p_() {}
  SE3(const Eigen::Matrix3d& R, const Eigen::Vector3d& p) : R_(R), p_(p) {}
  static SE3 Identity() { return SE3(); }
  const Eigen::Matrix3d& rotation() const { return R_; }
  Eigen::Matrix3d& rotation() { return R_; }
  const Eigen::Vector3d& translation() const { return p_; }
  Eigen::Vector3d& translation() { return p_; }
 private:
  Eigen::Matrix3d R_;
  Eigen::Vector3d p_;
};

}  // namespace pinocchio
#endif  // !IDOCP_HAVE_PINOCCHIO
#endif  // IDOCP_PINOCCHIO_SHIM_HPP_
