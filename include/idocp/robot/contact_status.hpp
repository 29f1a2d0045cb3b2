// idocp::ContactStatus -- facade (include/idocp/robot/contact_status.hpp:18-180 of the reference):
// which point contacts are active, and where the active feet stand in the world.
#ifndef IDOCP_CONTACT_STATUS_HPP_
#define IDOCP_CONTACT_STATUS_HPP_

#include <cassert>
#include <vector>

#include "idocp/eigen_shim.hpp"

namespace idocp {

class ContactStatus {
 public:
  explicit ContactStatus(const int max_point_contacts = 0)
      : is_contact_active_(max_point_contacts, false), contact_points_(max_point_contacts), dimf_(0),
        max_point_contacts_(max_point_contacts) {}

  // contact_status.hxx:34-50: equal = the same contacts active AND the same contact points (isApprox)
  bool operator==(const ContactStatus& other) const {
    assert(other.maxPointContacts() == max_point_contacts_);
    for (int i = 0; i < max_point_contacts_; ++i) {
      if (other.isContactActive(i) != isContactActive(i)) return false;
      if (!other.contactPoints()[i].isApprox(contactPoints()[i])) return false;
    }
    return true;
  }
  bool operator!=(const ContactStatus& other) const { return !(*this == other); }

  bool isContactActive(const int contact_index) const { return is_contact_active_.at(contact_index); }
  const std::vector<bool>& isContactActive() const { return is_contact_active_; }
  bool hasActiveContacts() const { return dimf_ > 0; }
  int dimf() const { return dimf_; }
  int maxPointContacts() const { return max_point_contacts_; }

  // ContactStatus::setActivity (contact_status.hxx:81-92): all flags at once
  void setActivity(const std::vector<bool>& is_contact_active) {
    assert((int)is_contact_active.size() == max_point_contacts_);
    is_contact_active_ = is_contact_active;
    dimf_ = 0;
    for (const bool on : is_contact_active_) if (on) dimf_ += 3;
  }
  void activateContact(const int contact_index) {
    if (!is_contact_active_.at(contact_index)) { is_contact_active_[contact_index] = true; dimf_ += 3; }
  }
  void deactivateContact(const int contact_index) {
    if (is_contact_active_.at(contact_index)) { is_contact_active_[contact_index] = false; dimf_ -= 3; }
  }
  void activateContacts(const std::vector<int>& contact_indices) { for (const int i : contact_indices) activateContact(i); }
  void activateContacts() { for (int i = 0; i < max_point_contacts_; ++i) activateContact(i); }
  void deactivateContacts(const std::vector<int>& contact_indices) { for (const int i : contact_indices) deactivateContact(i); }
  void deactivateContacts() { for (int i = 0; i < max_point_contacts_; ++i) deactivateContact(i); }

  void setContactPoint(const int contact_index, const Eigen::Vector3d& contact_point) { contact_points_.at(contact_index) = contact_point; }
  void setContactPoints(const std::vector<Eigen::Vector3d>& contact_points) {
    assert((int)contact_points.size() == max_point_contacts_);
    contact_points_ = contact_points;
  }
  const Eigen::Vector3d& contactPoint(const int contact_index) const { return contact_points_.at(contact_index); }
  const std::vector<Eigen::Vector3d>& contactPoints() const { return contact_points_; }

 private:
  std::vector<bool> is_contact_active_;
  std::vector<Eigen::Vector3d> contact_points_;
  int dimf_, max_point_contacts_;
};

}  // namespace idocp
#endif  // IDOCP_CONTACT_STATUS_HPP_
