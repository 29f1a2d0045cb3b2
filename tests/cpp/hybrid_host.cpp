// idocp::DiscreteEvent / idocp::ContactSequence (include/idocp/hybrid/*.hpp) on the host: the rules of the reference's contact_sequence.hxx:62-290 and
// discrete_event.hxx:77-103 on a scripted gait.  No device work: runs in the CPU suite (tests/test_hybrid_host.py), linked against the product library
// for the URDF reader only.  A second mode checks that a rule violation prints the reference's message and exits with EXIT_FAILURE.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "idocp/hybrid/contact_sequence.hpp"
#include "idocp/hybrid/discrete_event.hpp"
#include "idocp/robot/robot.hpp"

static int failures = 0;
#define REQUIRE(cond) do { if (!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); ++failures; } } while (0)

int main(int argc, char** argv) {
  if (argc < 2) { std::printf("usage: %s anymal.urdf [violation]\n", argv[0]); return 2; }
  const std::vector<int> frames = {14, 24, 34, 44};
  idocp::Robot robot(argv[1], frames);
  const std::string violation = argc > 2 ? argv[2] : "";

  idocp::ContactStatus all = robot.createContactStatus(), trot_a = all, trot_b = all, none = all;
  all.activateContacts();
  trot_a.activateContacts({0, 3});
  trot_b.activateContacts({1, 2});
  std::vector<Eigen::Vector3d> pts(4), moved(4);
  for (int i = 0; i < 4; ++i) { pts[i] = Eigen::Vector3d(0.1 * i, -0.2, 0.0); moved[i] = Eigen::Vector3d(0.1 * i + 0.05, -0.2, 0.0); }
  for (idocp::ContactStatus* s : {&all, &trot_a, &trot_b, &none}) s->setContactPoints(pts);

  // ---- DiscreteEvent
  {
    idocp::DiscreteEvent e(4);
    REQUIRE(!e.existDiscreteEvent() && e.maxPointContacts() == 4);
    e.setDiscreteEvent(all, trot_a);                       // two feet leave: a lift
    REQUIRE(e.existDiscreteEvent() && e.existLift() && !e.existImpulse() && e.impulseStatus().dimp() == 0);
    e.setDiscreteEvent(trot_a, all);                       // two feet land: an impulse with 6 rows, on feet 1 and 2
    REQUIRE(e.existImpulse() && !e.existLift() && e.impulseStatus().dimp() == 6 && e.impulseStatus().isImpulseActive(1) && !e.impulseStatus().isImpulseActive(0));
    e.setDiscreteEvent(trot_a, trot_b);                    // both at once
    REQUIRE(e.existImpulse() && e.existLift() && e.impulseStatus().dimp() == 6);
    e.setDiscreteEvent(all, all);
    REQUIRE(!e.existDiscreteEvent());
    REQUIRE(e.preContactStatus() == all && e.postContactStatus() == all && all != trot_a);
    idocp::ContactStatus shifted = all;
    shifted.setContactPoints(moved);
    REQUIRE(shifted != all);                               // (equality looks at the contact points too: contact_status.hxx:34-45)
    const idocp::DiscreteEvent landing(none, all);
    REQUIRE(landing.existImpulse() && landing.impulseStatus().dimp() == 12 && landing.impulseStatus().contactPoint(2).isApprox(pts[2]));
  }

  // ---- ContactSequence: stand, lift (0.3), land (0.5), swap diagonal pairs (0.9: impulse and lift at once), land (1.2)
  idocp::ContactSequence seq(robot, 5);
  REQUIRE(seq.numContactPhases() == 1 && seq.numDiscreteEvents() == 0 && !seq.contactStatus(0).hasActiveContacts());
  seq.setContactStatusUniformly(all);
  REQUIRE(seq.numContactPhases() == 1 && seq.contactStatus(0) == all);
  seq.push_back(trot_a, 0.3);
  seq.push_back(all, 0.5);
  seq.push_back(idocp::DiscreteEvent(all, trot_a), 0.7);
  seq.push_back(trot_b, 0.9);
  seq.push_back(all, 1.2);
  REQUIRE(seq.numContactPhases() == 6 && seq.numDiscreteEvents() == 5 && seq.numImpulseEvents() == 3 && seq.numLiftEvents() == 2);
  REQUIRE(seq.impulseTime(0) == 0.5 && seq.impulseTime(1) == 0.9 && seq.impulseTime(2) == 1.2 && seq.liftTime(0) == 0.3 && seq.liftTime(1) == 0.7);
  REQUIRE(seq.impulseStatus(0).dimp() == 6 && seq.impulseStatus(0).isImpulseActive(1) && seq.impulseStatus(1).isImpulseActive(2) && !seq.impulseStatus(1).isImpulseActive(0));
  REQUIRE(seq.contactStatus(4) == trot_b && seq.isImpulseEvent(3) && !seq.isImpulseEvent(2) && seq.eventTime(3) == 0.9);
  seq.updateImpulseTime(1, 0.95);
  seq.updateLiftTime(0, 0.25);
  REQUIRE(seq.impulseTime(1) == 0.95 && seq.eventTime(3) == 0.95 && seq.liftTime(0) == 0.25 && seq.eventTime(0) == 0.25);
  seq.setContactPoints(2, moved);                          // the phase opened by the first impulse: the impulse moves with it
  REQUIRE(seq.contactStatus(2).contactPoint(1).isApprox(moved[1]) && seq.impulseStatus(0).contactPoint(1).isApprox(moved[1]) && seq.impulseStatus(1).contactPoint(1).isApprox(pts[1]));

  if (violation == "no_event") seq.push_back(seq.contactStatus(5), 2.0);
  if (violation == "time_order") { seq.pop_back(); seq.push_back(all, 0.9); }
  if (violation == "inconsistent") { seq.pop_back(); seq.push_back(idocp::DiscreteEvent(all, trot_a), 2.0); }
  if (violation == "update_past_next") seq.updateLiftTime(0, 0.6);
  if (violation == "bad_phase") seq.setContactPoints(6, moved);
  if (violation == "too_many") { idocp::ContactSequence small(robot, 1); small.setContactStatusUniformly(all); small.push_back(trot_a, 0.1); small.push_back(all, 0.2); }
  if (violation == "max_num_events") idocp::ContactSequence bad(robot, 0);
  if (!violation.empty()) { std::printf("the violation '%s' was accepted\n", violation.c_str()); return 3; }

  seq.pop_front();                                         // the receding horizon leaves the first event behind
  REQUIRE(seq.numContactPhases() == 5 && seq.numLiftEvents() == 1 && seq.numImpulseEvents() == 3 && seq.contactStatus(0) == trot_a && seq.liftTime(0) == 0.7);
  seq.pop_back();
  REQUIRE(seq.numDiscreteEvents() == 3 && seq.numImpulseEvents() == 2 && seq.contactStatus(3) == trot_b);
  seq.pop_back(); seq.pop_back(); seq.pop_back();
  REQUIRE(seq.numDiscreteEvents() == 0 && seq.numContactPhases() == 1 && seq.contactStatus(0) == trot_a);
  seq.pop_front();                                         // nothing left but a phase: the default status takes its place
  REQUIRE(seq.numContactPhases() == 1 && !seq.contactStatus(0).hasActiveContacts());
  seq.clear_all();
  REQUIRE(seq.numContactPhases() == 0 && seq.numDiscreteEvents() == 0);

  if (failures) { std::printf("%d check(s) failed\n", failures); return 1; }
  std::printf("hybrid_host: all checks passed\n");
  return 0;
}
