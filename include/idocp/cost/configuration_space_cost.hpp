#include "idocp/cost/cost_function.hpp"
