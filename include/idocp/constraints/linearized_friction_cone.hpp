// Forwarding header: the reference keeps every constraint component in a header of its own
// (include/idocp/constraints/linearized_friction_cone.hpp); here the components live in constraints.hpp.
#ifndef IDOCP_CONSTRAINTS_LINEARIZED_FRICTION_CONE_HPP_
#define IDOCP_CONSTRAINTS_LINEARIZED_FRICTION_CONE_HPP_
#include "idocp/constraints/constraints.hpp"
#endif  // IDOCP_CONSTRAINTS_LINEARIZED_FRICTION_CONE_HPP_
