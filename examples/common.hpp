// Helpers shared by the example drivers of this repository.  The drivers run the workloads of the reference's examples
// (parameter values quoted in each driver) through the facade classes; everything here is plumbing: building weight
// vectors from runs, attaching the same weights to the stage / terminal / impulse slots of a cost, the standard joint-limit
// set, and a small table type for contact schedules.
#ifndef IDOCP_EXAMPLES_COMMON_HPP_
#define IDOCP_EXAMPLES_COMMON_HPP_

#include <cstdlib>
#include <initializer_list>
#include <iostream>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "idocp/constraints/constraints.hpp"
#include "idocp/cost/contact_force_cost.hpp"
#include "idocp/cost/cost_function.hpp"
#include "idocp/robot/robot.hpp"
#include "idocp/utils/ocp_benchmarker.hpp"

namespace ex {

using Vec = Eigen::VectorXd;
using V3 = Eigen::Vector3d;

inline int argInt(int argc, char** argv, int index, int fallback) { return argc > index ? std::atoi(argv[index]) : fallback; }

inline const char* needUrdf(int argc, char** argv, const char* more = "") {
  if (argc >= 2) return argv[1];
  std::cerr << "usage: " << argv[0] << " <robot.urdf> " << more << std::endl;
  std::exit(2);
}

// a vector made of runs: runs({{6, 1.0}, {12, 0.1}}) = six ones followed by twelve 0.1
inline Vec runs(std::initializer_list<std::pair<int, double>> pieces) {
  int n = 0;
  for (const auto& p : pieces) n += p.first;
  Vec out(n);
  int at = 0;
  for (const auto& p : pieces) for (int k = 0; k < p.first; ++k) out[at++] = p.second;
  return out;
}
inline Vec filled(int n, double x) { return runs({{n, x}}); }

// ANYmal: contact frames LF, LH, RF, RH of the URDF in tests/golden/urdf and the nominal standing configuration
inline std::vector<int> anymalFeet() { return {14, 24, 34, 44}; }
inline Vec anymalStanding(double x = 0.0) {
  const double hip = 0.1, thigh = 0.7, knee = 1.0;
  Vec q(19);
  q[0] = x; q[1] = 0; q[2] = 0.4792;
  q[3] = 0; q[4] = 0; q[5] = 0; q[6] = 1;
  const double side[4] = {-1, -1, 1, 1}, fore[4] = {1, -1, 1, -1};          // LF, LH, RF, RH
  for (int leg = 0; leg < 4; ++leg) { q[7 + 3 * leg] = side[leg] * hip; q[8 + 3 * leg] = fore[leg] * thigh; q[9 + 3 * leg] = -fore[leg] * knee; }
  return q;
}

// the same weights on the running, terminal and (optionally) impulse slots of a configuration-space cost
template <typename CostT>
void attachWeights(CostT& c, const Vec& wq, const Vec& wv, const Vec& wa, bool impulse_slots) {
  c.set_q_weight(wq); c.set_qf_weight(wq);
  c.set_v_weight(wv); c.set_vf_weight(wv);
  c.set_a_weight(wa);
  if (impulse_slots) { c.set_qi_weight(wq); c.set_vi_weight(wv); c.set_dvi_weight(wa); }
}

// contact-force cost with one weight for every foot; ref == nullptr: every foot carries its share of the weight
inline std::shared_ptr<idocp::ContactForceCost> forceCost(const idocp::Robot& robot, const V3& weight, bool impulse_slots, const V3* ref) {
  auto c = std::make_shared<idocp::ContactForceCost>(robot);
  const std::vector<V3> w(robot.maxPointContacts(), weight);
  c->set_f_weight(w);
  if (impulse_slots) c->set_fi_weight(w);
  if (ref) c->set_f_ref(std::vector<V3>(robot.maxPointContacts(), *ref)); else c->set_f_ref(robot);
  return c;
}

// position / velocity / torque limits of the actuated joints, plus the linearised friction cones when mu > 0
// quadratic = true: FrictionCone / ImpulseFrictionCone (the benchmark drivers of the reference) instead of the linearized pair
inline std::shared_ptr<idocp::Constraints> jointLimits(const idocp::Robot& robot, double mu = 0.0, bool impulse_cone = false, bool quadratic = false) {
  auto k = std::make_shared<idocp::Constraints>();
  k->push_back(std::make_shared<idocp::JointPositionLowerLimit>(robot));
  k->push_back(std::make_shared<idocp::JointPositionUpperLimit>(robot));
  k->push_back(std::make_shared<idocp::JointVelocityLowerLimit>(robot));
  k->push_back(std::make_shared<idocp::JointVelocityUpperLimit>(robot));
  k->push_back(std::make_shared<idocp::JointTorquesLowerLimit>(robot));
  k->push_back(std::make_shared<idocp::JointTorquesUpperLimit>(robot));
  if (mu > 0.0 && !quadratic) k->push_back(std::make_shared<idocp::LinearizedFrictionCone>(robot, mu));
  if (mu > 0.0 && !quadratic && impulse_cone) k->push_back(std::make_shared<idocp::LinearizedImpulseFrictionCone>(robot, mu));
  if (mu > 0.0 && quadratic) k->push_back(std::make_shared<idocp::FrictionCone>(robot, mu));
  if (mu > 0.0 && quadratic && impulse_cone) k->push_back(std::make_shared<idocp::ImpulseFrictionCone>(robot, mu));
  return k;
}

// where the feet are at configuration q
inline std::vector<V3> footholds(idocp::Robot& robot, const Vec& q) {
  robot.updateFrameKinematics(q);
  std::vector<V3> p(robot.maxPointContacts(), V3::Zero());
  robot.getContactPoints(p);
  return p;
}

// a contact schedule as a table: the first row holds from the start, every further row starts at its time
struct Phase {
  std::vector<int> stance;       // indices of the feet on the ground
  std::vector<V3> points;        // footholds of all feet
  double since;
};
struct Schedule {
  std::vector<Phase> rows;
  std::vector<V3> feet;          // running footholds, advanced by the builders below
  explicit Schedule(const std::vector<V3>& start) : feet(start) {}
  void advance(std::initializer_list<int> which, double dx) { for (int f : which) feet[f][0] += dx; }
  void add(std::initializer_list<int> stance, double since) { rows.push_back({std::vector<int>(stance), feet, since}); }
  template <typename Solver>
  void install(Solver& solver, idocp::Robot& robot) const {
    for (size_t r = 0; r < rows.size(); ++r) {
      auto status = robot.createContactStatus();
      status.activateContacts(rows[r].stance);
      status.setContactPoints(rows[r].points);
      if (r == 0) solver.setContactStatusUniformly(status); else solver.pushBackContactStatus(status, rows[r].since);
    }
  }
};

// initial guess: the given configuration at rest, every foot carrying a quarter of the weight
template <typename Solver>
void restingGuess(Solver& solver, const idocp::Robot& robot, const Vec& q) {
  solver.setSolution("q", q);
  solver.setSolution("v", Vec::Zero(robot.dimv()));
  solver.setSolution("f", V3(0, 0, 0.25 * robot.totalWeight()));
}

}  // namespace ex
#endif  // IDOCP_EXAMPLES_COMMON_HPP_
