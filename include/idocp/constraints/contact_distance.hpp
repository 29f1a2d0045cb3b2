// include-path compatibility with the reference (include/idocp/constraints/contact_distance.hpp); the component lives in constraints.hpp.
#include "idocp/constraints/constraints.hpp"
