// include/idocp/cost/time_varying_task_space_6d_cost.hpp of the reference: the class lives in task_space_cost.hpp
#ifndef IDOCP_TIME_VARYING_TASK_SPACE_6D_COST_FWD_HPP_
#define IDOCP_TIME_VARYING_TASK_SPACE_6D_COST_FWD_HPP_
#include "idocp/cost/task_space_cost.hpp"
#endif  // IDOCP_TIME_VARYING_TASK_SPACE_6D_COST_FWD_HPP_
