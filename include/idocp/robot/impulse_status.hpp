// idocp::ImpulseStatus -- facade (include/idocp/robot/impulse_status.hpp:18-150 of the reference): which point contacts become active at an impulse, and
// where.  The reference builds it on a ContactStatus (impulse_status.hpp: `ContactStatus contact_status_` member); so does this one.  The solvers take
// contact statuses (pushBackContactStatus) and derive the impulse of an event themselves, as the reference's do (discrete_event.hxx:57-84); the class is
// here for drivers that inspect or build one.
#ifndef IDOCP_IMPULSE_STATUS_HPP_
#define IDOCP_IMPULSE_STATUS_HPP_

#include <vector>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/contact_status.hpp"

namespace idocp {

class ImpulseStatus {
 public:
  explicit ImpulseStatus(const int max_point_contacts = 0) : status_(max_point_contacts) {}

  bool operator==(const ImpulseStatus& other) const { return status_ == other.status_; }      // impulse_status.hxx:36-56
  bool operator!=(const ImpulseStatus& other) const { return !(*this == other); }

  bool isImpulseActive(const int contact_index) const { return status_.isContactActive(contact_index); }
  const std::vector<bool>& isImpulseActive() const { return status_.isContactActive(); }
  bool hasActiveImpulse() const { return status_.hasActiveContacts(); }
  int dimf() const { return status_.dimf(); }
  int dimp() const { return status_.dimf(); }
  int maxPointContacts() const { return status_.maxPointContacts(); }

  void setActivity(const std::vector<bool>& is_impulse_active) { status_.setActivity(is_impulse_active); }
  // impulse_status.hxx:70-88: the impulse of the transition between two contact statuses = the contacts that become active
  void setActivity(const ContactStatus& pre_contact_status, const ContactStatus& post_contact_status) {
    std::vector<bool> on(status_.maxPointContacts(), false);
    for (int i = 0; i < status_.maxPointContacts(); ++i) on[i] = !pre_contact_status.isContactActive(i) && post_contact_status.isContactActive(i);
    status_.setActivity(on);
  }
  void activateImpulse(const int contact_index) { status_.activateContact(contact_index); }
  void deactivateImpulse(const int contact_index) { status_.deactivateContact(contact_index); }
  void activateImpulse(const std::vector<int>& impulse_indices) { status_.activateContacts(impulse_indices); }
  void deactivateImpulse(const std::vector<int>& impulse_indices) { status_.deactivateContacts(impulse_indices); }
  void activateImpulse() { status_.activateContacts(); }
  void deactivateImpulse() { status_.deactivateContacts(); }

  void setContactPoint(const int contact_index, const Eigen::Vector3d& contact_point) { status_.setContactPoint(contact_index, contact_point); }
  void setContactPoints(const std::vector<Eigen::Vector3d>& contact_points) { status_.setContactPoints(contact_points); }
  const Eigen::Vector3d& contactPoint(const int contact_index) const { return status_.contactPoint(contact_index); }
  const std::vector<Eigen::Vector3d>& contactPoints() const { return status_.contactPoints(); }

 private:
  ContactStatus status_;
};

}  // namespace idocp
#endif  // IDOCP_IMPULSE_STATUS_HPP_
