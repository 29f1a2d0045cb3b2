// Forwarding header: the reference keeps every constraint component in a header of its own
// (include/idocp/constraints/joint_torques_upper_limit.hpp); here the components live in constraints.hpp.
#ifndef IDOCP_CONSTRAINTS_JOINT_TORQUES_UPPER_LIMIT_HPP_
#define IDOCP_CONSTRAINTS_JOINT_TORQUES_UPPER_LIMIT_HPP_
#include "idocp/constraints/constraints.hpp"
#endif  // IDOCP_CONSTRAINTS_JOINT_TORQUES_UPPER_LIMIT_HPP_
