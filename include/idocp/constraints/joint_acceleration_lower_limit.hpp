// include-path compatibility with the reference (include/idocp/constraints/joint_acceleration_lower_limit.hpp); the component lives in
// constraints.hpp.
#include "idocp/constraints/constraints.hpp"
