// include/idocp/cost/task_space_6d_cost.hpp of the reference: the class lives in task_space_cost.hpp
#ifndef IDOCP_TASK_SPACE_6D_COST_FWD_HPP_
#define IDOCP_TASK_SPACE_6D_COST_FWD_HPP_
#include "idocp/cost/task_space_cost.hpp"
#endif  // IDOCP_TASK_SPACE_6D_COST_FWD_HPP_
