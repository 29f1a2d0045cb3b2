// Prints the task-space fields the facade components export into the flat cost block (tests/test_oracle_task_space.py).
#include <cstdio>
#include <memory>

#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/task_space_3d_cost.hpp"
#include "idocp/cost/task_space_6d_cost.hpp"
#include "idocp/cost/time_varying_task_space_3d_cost.hpp"
#include "idocp/cost/time_varying_task_space_6d_cost.hpp"

class Line final : public idocp::TimeVaryingTaskSpace3DRefBase {
 public:
  void compute_q_3d_ref(const double t, Eigen::VectorXd& p) const override { p[0] = 0.5 + t; p[1] = -0.25 * t; p[2] = 0.75; }
};
class Pose final : public idocp::TimeVaryingTaskSpace6DRefBase {
 public:
  void compute_q_6d_ref(const double t, pinocchio::SE3& M) const override {
    Eigen::Matrix3d R;
    R << 0, 0, 1,
         0, 1, 0,
        -1, 0, 0;
    M = pinocchio::SE3(R, Eigen::Vector3d(0.5, 0.1 * t, 0.7));
  }
};

static void dump(const char* name, const idocp_cost_t& c, const std::vector<double>& refs) {
  std::printf("%s dim %d joint %d tv %d weight", name, c.task_dim, c.task_joint, c.task_time_varying);
  for (int k = 0; k < 6; ++k) std::printf(" %.17g", c.task_weight[k]);
  std::printf(" weightf");
  for (int k = 0; k < 6; ++k) std::printf(" %.17g", c.task_weightf[k]);
  std::printf(" ref");
  for (int k = 0; k < 12; ++k) std::printf(" %.17g", c.task_ref[k]);
  std::printf(" frame");
  for (int k = 0; k < 9; ++k) std::printf(" %.17g", c.task_frame_R[k]);
  for (int k = 0; k < 3; ++k) std::printf(" %.17g", c.task_frame_p[k]);
  std::printf(" vweight %.17g refs", c.v_weight[0]);
  for (double x : refs) std::printf(" %.17g", x);
  std::printf("\n");
}

int main(int argc, char** argv) {
  idocp::Robot robot(argv[1]);
  const int frame = 22;
  std::vector<double> refs;
  {
    auto cost = std::make_shared<idocp::CostFunction>();
    auto t3 = std::make_shared<idocp::TaskSpace3DCost>(robot, frame);
    t3->set_q_3d_ref(Eigen::Vector3d(0.1, 0.2, 0.3));
    t3->set_q_3d_weight(Eigen::Vector3d(1, 2, 3));
    t3->set_qf_3d_weight(Eigen::Vector3d(4, 5, 6));
    cost->push_back(t3);
    auto cs = std::make_shared<idocp::ConfigurationSpaceCost>(robot);     // pushed AFTER: must not wipe the task fields
    cs->set_v_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.5));
    cost->push_back(cs);
    dump("3d", cost->native(), cost->taskRefs(0.0, 0.1, 2, refs) ? refs : std::vector<double>());
  }
  {
    auto cost = std::make_shared<idocp::CostFunction>();
    auto cs = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
    cs->set_v_weight(Eigen::VectorXd::Constant(robot.dimv(), 0.5));
    cost->push_back(cs);
    auto t6 = std::make_shared<idocp::TaskSpace6DCost>(robot, frame);
    Eigen::Matrix3d R;
    R << 0, -1, 0,
         1, 0, 0,
         0, 0, 1;
    t6->set_q_6d_ref(Eigen::Vector3d(0.1, 0.2, 0.3), R);
    t6->set_q_6d_weight(Eigen::Vector3d(1, 2, 3), Eigen::Vector3d(7, 8, 9));       // (position, rotation)
    t6->set_qf_6d_weight(Eigen::Vector3d(4, 5, 6), Eigen::Vector3d(10, 11, 12));
    cost->push_back(t6);
    dump("6d", cost->native(), cost->taskRefs(0.0, 0.1, 2, refs) ? refs : std::vector<double>());
  }
  {
    auto cost = std::make_shared<idocp::CostFunction>();
    auto t3 = std::make_shared<idocp::TimeVaryingTaskSpace3DCost>(robot, frame, std::make_shared<Line>());
    t3->set_q_3d_weight(Eigen::Vector3d(1, 2, 3));
    t3->set_qf_3d_weight(Eigen::Vector3d(4, 5, 6));
    cost->push_back(t3);
    refs.clear();
    const bool tv = cost->taskRefs(1.0, 0.5, 2, refs);
    dump("tv3d", cost->native(), tv ? refs : std::vector<double>());
  }
  {
    auto cost = std::make_shared<idocp::CostFunction>();
    auto t6 = std::make_shared<idocp::TimeVaryingTaskSpace6DCost>(robot, frame, std::make_shared<Pose>());
    t6->set_q_6d_weight(Eigen::Vector3d::Constant(1000), Eigen::Vector3d::Constant(100));
    t6->set_qf_6d_weight(Eigen::Vector3d::Constant(10), Eigen::Vector3d::Constant(1));
    cost->push_back(t6);
    refs.clear();
    const bool tv = cost->taskRefs(1.0, 0.5, 2, refs);
    dump("tv6d", cost->native(), tv ? refs : std::vector<double>());
  }
  return 0;
}
