// idocp::DiscreteEvent -- facade (include/idocp/hybrid/discrete_event.hpp:14-140 of the reference): the change between two contact statuses.  A contact
// that becomes active makes the event an impulse (and is a row of its ImpulseStatus), one that becomes inactive makes it a lift; an event can be both, and
// is then handled as an impulse (contact_sequence.hxx:97-109).  Host-side value type; the solvers' own sequence lives behind the C ABI
// (idocp_ocp_push_back_contact_status, ocp_capi.hip) and follows the same rules.
#ifndef IDOCP_DISCRETE_EVENT_HPP_
#define IDOCP_DISCRETE_EVENT_HPP_

#include <cassert>
#include <vector>

#include "idocp/eigen_shim.hpp"
#include "idocp/robot/contact_status.hpp"
#include "idocp/robot/impulse_status.hpp"

namespace idocp {

class DiscreteEvent {
 public:
  explicit DiscreteEvent(const int max_point_contacts = 0)
      : pre_(max_point_contacts), post_(max_point_contacts), impulse_(max_point_contacts), max_point_contacts_(max_point_contacts),
        exist_impulse_(false), exist_lift_(false) {}
  DiscreteEvent(const ContactStatus& pre_contact_status, const ContactStatus& post_contact_status)
      : DiscreteEvent(pre_contact_status.maxPointContacts()) {
    setDiscreteEvent(pre_contact_status, post_contact_status);
  }

  bool existDiscreteEvent() const { return exist_impulse_ || exist_lift_; }
  bool existImpulse() const { return exist_impulse_; }
  bool existLift() const { return exist_lift_; }
  const ImpulseStatus& impulseStatus() const { return impulse_; }
  const ContactStatus& preContactStatus() const { return pre_; }
  const ContactStatus& postContactStatus() const { return post_; }
  int maxPointContacts() const { return max_point_contacts_; }

  // discrete_event.hxx:77-103: the impulse rows are the contacts inactive before and active after; the impulse takes the contact points of the status after
  void setDiscreteEvent(const ContactStatus& pre_contact_status, const ContactStatus& post_contact_status) {
    assert(pre_contact_status.maxPointContacts() == max_point_contacts_);
    assert(post_contact_status.maxPointContacts() == max_point_contacts_);
    impulse_.setActivity(pre_contact_status, post_contact_status);
    exist_impulse_ = impulse_.hasActiveImpulse();
    exist_lift_ = false;
    for (int i = 0; i < max_point_contacts_; ++i)
      if (pre_contact_status.isContactActive(i) && !post_contact_status.isContactActive(i)) exist_lift_ = true;
    impulse_.setContactPoints(post_contact_status.contactPoints());
    pre_ = pre_contact_status;
    post_ = post_contact_status;
  }
  void setContactPoint(const int contact_index, const Eigen::Vector3d& contact_point) { impulse_.setContactPoint(contact_index, contact_point); }
  void setContactPoints(const std::vector<Eigen::Vector3d>& contact_points) { impulse_.setContactPoints(contact_points); }

 private:
  ContactStatus pre_, post_;
  ImpulseStatus impulse_;
  int max_point_contacts_;
  bool exist_impulse_, exist_lift_;
};

}  // namespace idocp
#endif  // IDOCP_DISCRETE_EVENT_HPP_
