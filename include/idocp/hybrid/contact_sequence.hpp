// idocp::ContactSequence -- facade (include/idocp/hybrid/contact_sequence.hpp:18-200 of the reference): the contact statuses of the horizon, one per contact
// phase, and the discrete events between them with their times -- impulses (a contact becomes active) and lifts counted separately, in time order.
// Host-side value type with the reference's rules and messages (contact_sequence.hxx:62-290): an event must change something, must start from the last
// status, must come later than the last event, and there is room for max_num_events of them; popping the last phase leaves the default status.  The
// solvers keep a sequence of their own behind the C ABI (idocp_ocp_push_back_contact_status / pop_back / pop_front, ocp_capi.hip: same rules, checked
// against the oracle on random sequences, tests/test_discretiser_fuzz_gpu.py); this class is for drivers that plan a gait before handing it over:
//   for (int i = 0; i < seq.numDiscreteEvents(); ++i) solver.pushBackContactStatus(seq.contactStatus(i + 1), seq.eventTime(i));
// Two places where this class does what the reference's text says rather than what its code does on inputs that break the sequence anyway:
// updateImpulseTime / updateLiftTime check BOTH neighbours of the event (contact_sequence.hxx:183-197 checks the later one only for the first event), and
// setContactPoints moves the impulse that opens the phase (contact_sequence.hxx:262-264 indexes the impulses by phase, right only while no lift precedes).
#ifndef IDOCP_CONTACT_SEQUENCE_HPP_
#define IDOCP_CONTACT_SEQUENCE_HPP_

#include <cassert>
#include <cstdlib>
#include <deque>
#include <iostream>
#include <string>
#include <vector>

#include "idocp/eigen_shim.hpp"
#include "idocp/hybrid/discrete_event.hpp"
#include "idocp/robot/contact_status.hpp"
#include "idocp/robot/impulse_status.hpp"
#include "idocp/robot/robot.hpp"

namespace idocp {

class ContactSequence {
 public:
  ContactSequence(const Robot& robot, const int max_num_events)
      : max_num_events_(max_num_events), default_contact_status_(robot.createContactStatus()) {
    if (max_num_events <= 0) die("invalid argument: max_num_events must be positive!");
    phases_.push_back(default_contact_status_);
  }
  ContactSequence() : max_num_events_(0) {}

  void setContactStatusUniformly(const ContactStatus& contact_status) {
    clear_all();
    phases_.push_back(contact_status);
  }

  void push_back(const DiscreteEvent& discrete_event, const double event_time) {
    if (numContactPhases() == 0) die("Call setContactStatusUniformly() before calling push_back()!");
    if (!discrete_event.existDiscreteEvent()) die("discrete_event.existDiscreteEvent() must be true!");
    if (discrete_event.preContactStatus() != phases_.back()) die("discrete_event.preContactStatus() is not consistent with the last contact status!");
    if (numDiscreteEvents() + 1 > max_num_events_)
      die("Number of discrete events=" + std::to_string(numDiscreteEvents() + 1) + " exceeds predefined max_num_events=" + std::to_string(max_num_events_) + "!");
    if (numDiscreteEvents() > 0 && event_time <= events_.back().time)
      die("event_time=" + std::to_string(event_time) + " must be larger than the last event time=" + std::to_string(events_.back().time) + "!");
    phases_.push_back(discrete_event.postContactStatus());
    events_.push_back({event_time, discrete_event.existImpulse(), discrete_event});
  }
  void push_back(const ContactStatus& contact_status, const double event_time) { push_back(DiscreteEvent(phases_.back(), contact_status), event_time); }

  void pop_back() {
    if (numDiscreteEvents() > 0) {
      events_.pop_back();
      phases_.pop_back();
    } else if (numContactPhases() > 0) {
      phases_.back() = default_contact_status_;
    }
  }
  void pop_front() {
    if (numDiscreteEvents() > 0) {
      events_.pop_front();
      phases_.pop_front();
    } else if (numContactPhases() > 0) {
      phases_.front() = default_contact_status_;
    }
  }

  void updateImpulseTime(const int impulse_index, const double impulse_time) { updateTime(true, impulse_index, impulse_time); }
  void updateLiftTime(const int lift_index, const double lift_time) { updateTime(false, lift_index, lift_time); }

  // the contact points of a phase; the impulse that opens the phase takes them too
  void setContactPoints(const int contact_phase, const std::vector<Eigen::Vector3d>& contact_points) {
    if (contact_phase < 0 || contact_phase >= numContactPhases())
      die("contact_phase=" + std::to_string(contact_phase) + " must be smaller than numContactPhases()" + std::to_string(numContactPhases()) + "!");
    phases_[contact_phase].setContactPoints(contact_points);
    if (contact_phase > 0 && events_[contact_phase - 1].impulse) events_[contact_phase - 1].event.setContactPoints(contact_points);
  }

  int numImpulseEvents() const { int n = 0; for (const Event& e : events_) n += e.impulse ? 1 : 0; return n; }
  int numLiftEvents() const { return numDiscreteEvents() - numImpulseEvents(); }
  int numDiscreteEvents() const { return (int)events_.size(); }
  int numContactPhases() const { return (int)phases_.size(); }
  const ContactStatus& contactStatus(const int contact_phase) const { return phases_.at(contact_phase); }
  const ImpulseStatus& impulseStatus(const int impulse_index) const { return events_.at(eventOf(true, impulse_index)).event.impulseStatus(); }
  double impulseTime(const int impulse_index) const { return events_.at(eventOf(true, impulse_index)).time; }
  double liftTime(const int lift_index) const { return events_.at(eventOf(false, lift_index)).time; }
  // (not in the reference's public interface: the events in time order, whatever their kind -- what a driver that replays the sequence needs)
  double eventTime(const int event_index) const { return events_.at(event_index).time; }
  bool isImpulseEvent(const int event_index) const { return events_.at(event_index).impulse; }

  void clear_all() {
    phases_.clear();
    events_.clear();
  }

 private:
  struct Event { double time; bool impulse; DiscreteEvent event; };
  int max_num_events_;
  ContactStatus default_contact_status_;
  std::deque<ContactStatus> phases_;      // phases_[i + 1] follows events_[i]
  std::deque<Event> events_;

  [[noreturn]] static void die(const std::string& what) {      // the reference prints the message and exits (contact_sequence.hxx:93-96)
    std::cerr << what << '\n';
    std::exit(EXIT_FAILURE);
  }
  // index in time order of the k-th impulse (or lift); the number of events when there is none
  int eventOf(const bool impulse, const int k) const {
    int seen = 0;
    for (int e = 0; e < numDiscreteEvents(); ++e)
      if (events_[e].impulse == impulse && seen++ == k) return e;
    return numDiscreteEvents();
  }
  // contact_sequence.hxx:165-247: the new time must stay between the neighbouring events
  void updateTime(const bool impulse, const int index, const double time) {
    const char* kind = impulse ? "impulse" : "lift";
    const int count = impulse ? numImpulseEvents() : numLiftEvents();
    if (count <= 0) die(std::string(impulse ? "numImpulseEvents()" : "numLiftEvents()") + " must be positive when calling this method!");
    if (index < 0) die(std::string(kind) + "_index must be non-negative!");
    if (index >= count)
      die(std::string(kind) + "_index=" + std::to_string(index) + " must be less than " + (impulse ? "numImpulseEvents()=" : "numLiftEvents()=") + std::to_string(count) + "!");
    const int e = eventOf(impulse, index);
    if (e > 0 && events_[e - 1].time >= time)
      die(std::string(kind) + "_time=" + std::to_string(time) + " must be larger than event_time_[event_index-1]=" + std::to_string(events_[e - 1].time) + "!");
    if (e + 1 < numDiscreteEvents() && events_[e + 1].time <= time)
      die(std::string(kind) + "_time=" + std::to_string(time) + " must be smaller than event_time_[event_index+1]=" + std::to_string(events_[e + 1].time) + "!");
    events_[e].time = time;
  }
};

}  // namespace idocp
#endif  // IDOCP_CONTACT_SEQUENCE_HPP_
