// idocp::Robot -- facade over the C ABI (include/idocp_hip.h).
// Mirrors the part of the reference class the drivers of the hot path use
// (include/idocp/robot/robot.hpp:26; src/robot/robot.cpp:8-85,113-170).
#ifndef IDOCP_ROBOT_HPP_
#define IDOCP_ROBOT_HPP_

#include <cstdlib>
#include <iostream>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/contact_status.hpp"
#include "idocp/robot/impulse_status.hpp"
#include "idocp/pinocchio_shim.hpp"
#include "idocp_hip.h"

namespace idocp {

class Robot {
 public:
  // Robot(path_to_urdf) / Robot(path_to_urdf, contact_frames): errors follow the
  // reference convention -- message on stderr, std::exit(EXIT_FAILURE).
  explicit Robot(const std::string& path_to_urdf, const std::vector<int>& contact_frames = {}) : path_(path_to_urdf) {
    if (idocp_abi_check(sizeof(idocp_model_t), sizeof(idocp_cost_t), sizeof(idocp_constraints_t)) != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
    const int rc = idocp_model_from_urdf(path_to_urdf.c_str(), contact_frames.empty() ? nullptr : contact_frames.data(),
                                         (int)contact_frames.size(), &model_);
    if (rc != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
  Robot() : model_() {}

  int dimq() const { return model_.nq; }
  int dimv() const { return model_.nv; }
  int dimu() const { return model_.nu; }
  int dim_passive() const { return model_.has_floating_base ? 6 : 0; }
  int max_dimf() const { return 3 * model_.ncontacts; }
  bool hasFloatingBase() const { return model_.has_floating_base != 0; }
  int maxPointContacts() const { return model_.ncontacts; }
  double totalWeight() const { return -model_.total_mass * model_.gravity[2]; }

  Eigen::VectorXd jointEffortLimit() const { return get(model_.u_max); }
  Eigen::VectorXd jointVelocityLimit() const { return get(model_.v_max); }
  Eigen::VectorXd lowerJointPositionLimit() const { return get(model_.q_min); }
  Eigen::VectorXd upperJointPositionLimit() const { return get(model_.q_max); }
  void setJointEffortLimit(const Eigen::VectorXd& v) { set(model_.u_max, v, "invalid size of joint_effort_limit"); }
  void setJointVelocityLimit(const Eigen::VectorXd& v) { set(model_.v_max, v, "invalid size of joint_velocity_limit"); }
  void setLowerJointPositionLimit(const Eigen::VectorXd& v) { set(model_.q_min, v, "invalid size of lower_joint_position_limit"); }
  void setUpperJointPositionLimit(const Eigen::VectorXd& v) { set(model_.q_max, v, "invalid size of upper_joint_position_limit"); }

  // Robot::createContactStatus / updateFrameKinematics / setContactPoints / getContactPoints
  // (include/idocp/robot/robot.hxx:85-91, 262-283, 661-683).  Host arithmetic; see
  // idocp_model_contact_positions in idocp_hip.h.
  ContactStatus createContactStatus() const { return ContactStatus(model_.ncontacts); }
  void updateFrameKinematics(const Eigen::VectorXd& q) {
    if (q.size() != model_.nq) {
      std::cerr << "invalid size: q.size() must be " << model_.nq << "!" << '\n';
      std::exit(EXIT_FAILURE);
    }
    q_kin_.assign(q.data(), q.data() + model_.nq);
    points_.assign(3 * (size_t)model_.ncontacts, 0.0);
    if (model_.ncontacts == 0) return;
    if (idocp_model_contact_positions(&model_, q.data(), points_.data()) != IDOCP_OK) {
      std::cerr << idocp_last_error() << '\n';
      std::exit(EXIT_FAILURE);
    }
  }
  // Robot::updateKinematics(q[, v[, a]]) (robot.hxx:166-203): the facade keeps the configuration for the frame queries below; velocities and accelerations of
  // frames are evaluated inside the stage kernels and have no host-side query.
  void updateKinematics(const Eigen::VectorXd& q) { updateFrameKinematics(q); }
  void updateKinematics(const Eigen::VectorXd& q, const Eigen::VectorXd&) { updateFrameKinematics(q); }
  void updateKinematics(const Eigen::VectorXd& q, const Eigen::VectorXd&, const Eigen::VectorXd&) { updateFrameKinematics(q); }
  // Robot::framePosition / frameRotation / framePlacement (robot.hxx:206-233): of any frame of the URDF (pinocchio's frame numbering, as the
  // contact frames and the task-space costs use it), at the configuration of the last updateFrameKinematics / updateKinematics
  Eigen::Vector3d framePosition(const int frame_id) const { return framePlacement(frame_id).translation(); }
  Eigen::Matrix3d frameRotation(const int frame_id) const { return framePlacement(frame_id).rotation(); }
  pinocchio::SE3 framePlacement(const int frame_id) const {
    if (q_kin_.empty()) { std::cerr << "invalid function call: call updateFrameKinematics(q) first!" << '\n'; std::exit(EXIT_FAILURE); }
    auto it = frames_.find(frame_id);                      // (where the frame sits is read from the URDF once per frame)
    if (it == frames_.end()) {
      FrameLocation loc;
      ok(idocp_model_frame_placement(path_.c_str(), frame_id, &loc.joint, loc.R, loc.p));
      it = frames_.emplace(frame_id, loc).first;
    }
    double Rw[9], pw[3];
    ok(idocp_model_frame_world_placement(&model_, q_kin_.data(), it->second.joint, it->second.R, it->second.p, Rw, pw));
    Eigen::Matrix3d R;
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) R(r, c) = Rw[3 * r + c];
    return pinocchio::SE3(R, Eigen::Vector3d(pw[0], pw[1], pw[2]));
  }
  // Robot::generateFeasibleConfiguration (robot.hxx:618-626): uniform between the joint position limits; the base, if any, uniform in [-1, 1]^3 with a
  // uniformly random orientation
  Eigen::VectorXd generateFeasibleConfiguration() const {
    Eigen::VectorXd q(model_.nq);
    auto uni = [](const double lo, const double hi) { return lo + (hi - lo) * (std::rand() / (double)RAND_MAX); };
    const int nb = model_.has_floating_base ? 7 : 0;
    for (int k = 0; k < nb; ++k) q[k] = uni(-1.0, 1.0);
    for (int k = 0; k < model_.nu; ++k) q[nb + k] = uni(model_.q_min[k], model_.q_max[k]);      // (the model's limit arrays run over the actuated joints)
    if (nb) ok(idocp_model_normalize_configuration(&model_, q.data()));
    return q;
  }
  void getContactPoints(std::vector<Eigen::Vector3d>& contact_points) const {
    contact_points.resize(model_.ncontacts);
    for (int c = 0; c < model_.ncontacts; ++c) for (int k = 0; k < 3; ++k) contact_points[c][k] = points_.at(3 * c + k);
  }
  void setContactPoints(ContactStatus& contact_status) const {
    std::vector<Eigen::Vector3d> pts;
    getContactPoints(pts);
    contact_status.setContactPoints(pts);
  }

  // Robot::integrateConfiguration / subtractConfiguration / normalizeConfiguration (robot.hxx:96-147; robot.hpp:105-160): what a driver does
  // between two solver calls (advance the plant, measure a distance on the configuration manifold).  Host arithmetic behind the C ABI
  // (idocp_model_integrate_configuration ...).
  void integrateConfiguration(const Eigen::VectorXd& v, const double integration_length, Eigen::VectorXd& q) const {      // in place (robot.hpp:82-86)
    Eigen::VectorXd out(model_.nq);
    integrateConfiguration(q, v, integration_length, out);
    q = out;
  }
  void integrateConfiguration(const Eigen::VectorXd& q, const Eigen::VectorXd& v, const double integration_length, Eigen::VectorXd& q_integrated) const {
    sized(q, model_.nq, "q"); sized(v, model_.nv, "v");
    if (q_integrated.size() != model_.nq) q_integrated.resize(model_.nq);
    ok(idocp_model_integrate_configuration(&model_, q.data(), v.data(), integration_length, q_integrated.data()));
  }
  void subtractConfiguration(const Eigen::VectorXd& q_plus, const Eigen::VectorXd& q_minus, Eigen::VectorXd& difference) const {
    sized(q_plus, model_.nq, "q_plus"); sized(q_minus, model_.nq, "q_minus");
    if (difference.size() != model_.nv) difference.resize(model_.nv);
    ok(idocp_model_subtract_configuration(&model_, q_plus.data(), q_minus.data(), difference.data()));
  }
  void normalizeConfiguration(Eigen::VectorXd& q) const {
    sized(q, model_.nq, "q");
    ok(idocp_model_normalize_configuration(&model_, q.data()));
  }
  // Robot::setFrictionCoefficient / frictionCoefficient / setRestitutionCoefficient / restitutionCoefficient (robot.hxx:358-406): per-contact properties
  // the reference stores (defaults 0.8 and 0, point_contact.hpp) and nothing on the solver path reads -- the friction cones take their mu as a
  // constructor argument (friction_cone.hpp).  Kept so that a driver that sets them compiles and reads back what it set.
  void setFrictionCoefficient(const std::vector<double>& friction_coefficient) {
    for (int i = 0; i < model_.ncontacts && i < (int)friction_coefficient.size(); ++i) {
      if (friction_coefficient[i] <= 0) { std::cerr << "invalid argument: friction coefficient must be positive" << '\n'; std::exit(EXIT_FAILURE); }
      friction_[i] = friction_coefficient[i];
    }
  }
  double frictionCoefficient(const int contact_index) const { noContacts(); return friction_[contact_index]; }
  void setRestitutionCoefficient(const std::vector<double>& restitution_coefficient) {
    for (int i = 0; i < model_.ncontacts && i < (int)restitution_coefficient.size(); ++i) {
      if (restitution_coefficient[i] < 0 || restitution_coefficient[i] > 1) { std::cerr << "invalid argument: restitution coefficient must be in [0, 1]" << '\n'; std::exit(EXIT_FAILURE); }
      restitution_[i] = restitution_coefficient[i];
    }
  }
  double restitutionCoefficient(const int contact_index) const { noContacts(); return restitution_[contact_index]; }
  // Robot::contactFramesIndices (robot.hxx:655-661)
  std::vector<int> contactFramesIndices() const { return std::vector<int>(model_.contact_frame_id, model_.contact_frame_id + model_.ncontacts); }
  // Robot::createImpulseStatus (robot.hxx:672-675)
  ImpulseStatus createImpulseStatus() const { return ImpulseStatus(model_.ncontacts); }

  const idocp_model_t& model() const { return model_; }
  // the URDF this robot was built from (frame lookups of the task-space costs)
  const std::string& pathToUrdf() const { return path_; }

 private:
  idocp_model_t model_;
  std::string path_;
  std::vector<double> points_;     // contact-frame positions of the last updateFrameKinematics(q)
  std::vector<double> q_kin_;      // that configuration (frame queries)
  struct FrameLocation { int joint; double R[9], p[3]; };
  mutable std::map<int, FrameLocation> frames_;
  double friction_[IDOCP_MAX_CONTACTS] = {0.8, 0.8, 0.8, 0.8}, restitution_[IDOCP_MAX_CONTACTS] = {0.0, 0.0, 0.0, 0.0};
  void noContacts() const {
    if (model_.ncontacts == 0) { std::cerr << "invalid function call: robot has no point contacts!" << '\n'; std::exit(EXIT_FAILURE); }
  }
  static void ok(int rc) { if (rc != IDOCP_OK) { std::cerr << idocp_last_error() << '\n'; std::exit(EXIT_FAILURE); } }
  static void sized(const Eigen::VectorXd& x, int n, const char* name) {
    if (x.size() != n) { std::cerr << "invalid size: " << name << ".size() must be " << n << "!" << '\n'; std::exit(EXIT_FAILURE); }
  }
  Eigen::VectorXd get(const double* p) const {
    Eigen::VectorXd v(model_.nu);
    for (int i = 0; i < model_.nu; ++i) v[i] = p[i];
    return v;
  }
  void set(double* p, const Eigen::VectorXd& v, const char* msg) {
    if (v.size() != model_.nu) {   // robot.cpp:113-170: throw -> catch -> exit
      std::cerr << msg << '\n';
      std::exit(EXIT_FAILURE);
    }
    for (int i = 0; i < model_.nu; ++i) p[i] = v[i];
  }
};

}  // namespace idocp
#endif  // IDOCP_ROBOT_HPP_
