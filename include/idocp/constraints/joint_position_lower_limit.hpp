// Forwarding header: the reference keeps every constraint component in a header of its own
// (include/idocp/constraints/joint_position_lower_limit.hpp); here the components live in constraints.hpp.
#ifndef IDOCP_CONSTRAINTS_JOINT_POSITION_LOWER_LIMIT_HPP_
#define IDOCP_CONSTRAINTS_JOINT_POSITION_LOWER_LIMIT_HPP_
#include "idocp/constraints/constraints.hpp"
#endif  // IDOCP_CONSTRAINTS_JOINT_POSITION_LOWER_LIMIT_HPP_
