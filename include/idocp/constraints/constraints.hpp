// idocp::Constraints + the six joint-limit components -- facade.
// (include/idocp/constraints/constraints.hxx; src/constraints/joint_*_limit.cpp)
// The limits themselves come from the Robot (qmin_/qmax_/vmax_/umax_ are read
// from it by the reference constructors too); a component only switches its
// limit family on and carries the IPM parameters.
#ifndef IDOCP_CONSTRAINTS_HPP_
#define IDOCP_CONSTRAINTS_HPP_

#include <cstdlib>
#include <iostream>
#include <memory>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp_hip.h"

namespace idocp {

class ConstraintComponentBase {
 public:
  enum Family { Position, Velocity, Torque, LinearFrictionCone, QuadraticFrictionCone, Acceleration, Distance };
  ConstraintComponentBase(Family f, bool upper_in, double barrier_in, double rate_in)
      : family(f), upper(upper_in), barrier(barrier_in), fraction_to_boundary_rate(rate_in) {}
  virtual ~ConstraintComponentBase() {}
  Family family;
  bool upper;
  double barrier, fraction_to_boundary_rate;
  double mu = 0.0;
  Eigen::VectorXd bound;      // JointAcceleration*Limit: amin / amax, one entry per actuated joint

 protected:
  // The reference constrains a.tail(amin.size()) (joint_acceleration_lower_limit.cpp:10-24); the kernels carry one row per
  // actuated joint, so a bound vector of any other length is rejected here instead of silently bounding the joints it does not
  // name at zero.
  static Eigen::VectorXd checkedBound(const Robot& robot, const Eigen::VectorXd& b, const char* what) {
    if (b.size() != robot.dimu()) {
      std::cerr << "invalid argument: " << what << ".size() must be Robot::dimu() = " << robot.dimu() << " on the HIP path" << '\n';
      std::exit(EXIT_FAILURE);
    }
    return b;
  }
};

#define IDOCP_LIMIT_CLASS(NAME, FAMILY, UPPER)                                                        \
  class NAME final : public ConstraintComponentBase {                                                 \
   public:                                                                                            \
    explicit NAME(const Robot&, double barrier_in = 1.0e-04, double rate_in = 0.995)   \
        : ConstraintComponentBase(FAMILY, UPPER, barrier_in, rate_in) {}               \
  }
IDOCP_LIMIT_CLASS(JointPositionLowerLimit, Position, false);
IDOCP_LIMIT_CLASS(JointPositionUpperLimit, Position, true);
IDOCP_LIMIT_CLASS(JointVelocityLowerLimit, Velocity, false);
IDOCP_LIMIT_CLASS(JointVelocityUpperLimit, Velocity, true);
IDOCP_LIMIT_CLASS(JointTorquesLowerLimit, Torque, false);
IDOCP_LIMIT_CLASS(JointTorquesUpperLimit, Torque, true);
#undef IDOCP_LIMIT_CLASS

// JointAccelerationLowerLimit / UpperLimit (src/constraints/joint_acceleration_{lower,upper}_limit.cpp): a.tail(dimc) >= amin resp.
// <= amax with the bounds passed to the constructor (the robot model has none); each may be used on its own.
class JointAccelerationLowerLimit final : public ConstraintComponentBase {
 public:
  JointAccelerationLowerLimit(const Robot& robot, const Eigen::VectorXd& amin, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(Acceleration, false, barrier_in, rate_in) { bound = checkedBound(robot, amin, "amin"); }
};
class JointAccelerationUpperLimit final : public ConstraintComponentBase {
 public:
  JointAccelerationUpperLimit(const Robot& robot, const Eigen::VectorXd& amax, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(Acceleration, true, barrier_in, rate_in) { bound = checkedBound(robot, amax, "amax"); }
};

// ContactDistance (src/constraints/contact_distance.cpp): the frames of the contacts that are not active on a stage stay above z = 0
// (floating-base solvers).
class ContactDistance final : public ConstraintComponentBase {
 public:
  explicit ContactDistance(const Robot&, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(Distance, false, barrier_in, rate_in) {}
};

// LinearizedFrictionCone (include/idocp/constraints/linearized_friction_cone.hpp:17-120,
// src/constraints/linearized_friction_cone.cpp): five rows per active contact, evaluated in K5b.
class LinearizedFrictionCone final : public ConstraintComponentBase {
 public:
  LinearizedFrictionCone(const Robot&, const double mu_in, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(LinearFrictionCone, false, barrier_in, rate_in) { setFrictionCoefficient(mu_in); }
  void setFrictionCoefficient(const double mu_in) {
    if (mu_in <= 0) {      // linearized_friction_cone.cpp:20-31
      std::cerr << "invalid argment: mu must be positive" << '\n';
      std::exit(EXIT_FAILURE);
    }
    mu = mu_in;
  }
};
// Impulse twin (src/constraints/linearized_impulse_friction_cone.cpp): the same cone on the impulse forces of an
// impulse stage (no time-step scaling).
class LinearizedImpulseFrictionCone final : public ConstraintComponentBase {
 public:
  LinearizedImpulseFrictionCone(const Robot&, const double mu_in, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(LinearFrictionCone, true, barrier_in, rate_in) { mu = mu_in; }
};

// FrictionCone (include/idocp/constraints/friction_cone.hpp, src/constraints/friction_cone.cpp; the cone of
// examples/anymal/ocp_benchmark.cpp:76): two rows per active contact, -fz <= 0 and fx^2 + fy^2 - mu^2 fz^2 <= 0.
class FrictionCone final : public ConstraintComponentBase {
 public:
  FrictionCone(const Robot&, const double mu_in, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(QuadraticFrictionCone, false, barrier_in, rate_in) { setFrictionCoefficient(mu_in); }
  void setFrictionCoefficient(const double mu_in) {
    if (mu_in <= 0) {      // friction_cone.cpp:13-22
      std::cerr << "invalid value: mu must be positive!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    mu = mu_in;
  }
};
// Impulse twin (src/constraints/impulse_friction_cone.cpp)
class ImpulseFrictionCone final : public ConstraintComponentBase {
 public:
  ImpulseFrictionCone(const Robot&, const double mu_in, double barrier_in = 1.0e-04, double rate_in = 0.995)
      : ConstraintComponentBase(QuadraticFrictionCone, true, barrier_in, rate_in) { mu = mu_in; }
};

class Constraints {
 public:
  Constraints() : lo_{0, 0, 0}, hi_{0, 0, 0} {
    idocp_constraints_init(&c_);
    c_.joint_position_limits = c_.joint_velocity_limits = c_.joint_torque_limits = 0;
    c_.linearized_friction_cone = 0;
  }
  // The reference keeps barrier and fraction-to-boundary rate per component (constraint_component_base.hxx:10-24); the kernels take
  // ONE pair for the whole stage.  Components that disagree are rejected loudly instead of letting the last one win.
  void push_back(const std::shared_ptr<ConstraintComponentBase>& c) {
    if (have_ipm_ && (c->barrier != c_.barrier || c->fraction_to_boundary_rate != c_.fraction_to_boundary_rate)) {
      std::cerr << "unsupported constraints: every component must use the same barrier and fraction_to_boundary_rate on the HIP path" << '\n';
      std::exit(EXIT_FAILURE);
    }
    c_.barrier = c->barrier;
    c_.fraction_to_boundary_rate = c->fraction_to_boundary_rate;
    have_ipm_ = true;
    if (c->family == ConstraintComponentBase::LinearFrictionCone) {
      if (!c->upper) c_.linearized_friction_cone = 1; else c_.linearized_impulse_friction_cone = 1;     // upper = impulse twin
      c_.mu = c->mu;
      return;
    }
    if (c->family == ConstraintComponentBase::QuadraticFrictionCone) {
      if (!c->upper) c_.friction_cone = 1; else c_.impulse_friction_cone = 1;
      c_.mu = c->mu;
      return;
    }
    if (c->family == ConstraintComponentBase::Distance) { c_.contact_distance = 1; return; }
    if (c->family == ConstraintComponentBase::Acceleration) {
      if (c->bound.size() > IDOCP_MAX_NV) { std::cerr << "invalid argument: too many acceleration bounds" << '\n'; std::exit(EXIT_FAILURE); }
      (c->upper ? c_.joint_acceleration_upper_limit : c_.joint_acceleration_lower_limit) = 1;
      for (int i = 0; i < c->bound.size(); ++i) (c->upper ? c_.a_max : c_.a_min)[i] = c->bound[i];
      return;
    }
    (c->upper ? hi_ : lo_)[c->family] = 1;
  }
  // Constraints::setBarrier / setFractionToBoundaryRate (constraints.hxx:474-492): one value for every component pushed so far -- which is all the
  // kernels take anyway (the reference asserts positivity, constraint_component_base.hxx:10-20; here a message, the solver would refuse the value
  // too).  Call them before the solver is constructed: the solver copies the parameters.
  void setBarrier(const double barrier) {
    if (!(barrier > 0)) { std::cerr << "invalid value: barrier must be positive!" << '\n'; std::exit(EXIT_FAILURE); }
    c_.barrier = barrier;
  }
  void setFractionToBoundaryRate(const double fraction_to_boundary_rate) {
    if (!(fraction_to_boundary_rate > 0)) { std::cerr << "invalid value: fraction_to_boundary_rate must be positive!" << '\n'; std::exit(EXIT_FAILURE); }
    c_.fraction_to_boundary_rate = fraction_to_boundary_rate;
  }
  // Constraints::clear (constraints.hxx:43-48): no component left
  void clear() { *this = Constraints(); }
  // The kernels treat a limit family as a lower+upper pair (what
  // JointConstraintsFactory::create() builds); a lone lower or upper limit is rejected.
  idocp_constraints_t native() const {
    idocp_constraints_t c = c_;
    int* flag[3] = {&c.joint_position_limits, &c.joint_velocity_limits, &c.joint_torque_limits};
    for (int f = 0; f < 3; ++f) {
      if (lo_[f] != hi_[f]) {
        std::cerr << "unsupported constraints: joint limits must come in lower/upper pairs on the HIP path" << '\n';
        std::exit(EXIT_FAILURE);
      }
      *flag[f] = lo_[f];
    }
    return c;
  }

 private:
  idocp_constraints_t c_;
  int lo_[3], hi_[3];
  bool have_ipm_ = false;
};

}  // namespace idocp
#endif  // IDOCP_CONSTRAINTS_HPP_
