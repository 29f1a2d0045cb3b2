// idocp::JointConstraintsFactory (src/utils/joint_constraints_factory.cpp:11-38)
#ifndef IDOCP_JOINT_CONSTRAINTS_FACTORY_HPP_
#define IDOCP_JOINT_CONSTRAINTS_FACTORY_HPP_

#include <memory>

#include "idocp/constraints/constraints.hpp"

namespace idocp {

class JointConstraintsFactory {
 public:
  explicit JointConstraintsFactory(const Robot& robot) : robot_(robot) {}
  std::shared_ptr<Constraints> create() const {
    auto c = std::make_shared<Constraints>();
    c->push_back(std::make_shared<JointPositionLowerLimit>(robot_));
    c->push_back(std::make_shared<JointPositionUpperLimit>(robot_));
    c->push_back(std::make_shared<JointVelocityLowerLimit>(robot_));
    c->push_back(std::make_shared<JointVelocityUpperLimit>(robot_));
    c->push_back(std::make_shared<JointTorquesLowerLimit>(robot_));
    c->push_back(std::make_shared<JointTorquesUpperLimit>(robot_));
    return c;
  }

 private:
  Robot robot_;
};

}  // namespace idocp
#endif  // IDOCP_JOINT_CONSTRAINTS_FACTORY_HPP_
