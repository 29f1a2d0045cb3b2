// The parts of the reference's solver interface that are about C++ object semantics rather than arithmetic
// (ocp_solver.hpp:44-74,97; unocp_solver.hpp:49-74,96; parnmpc_solver.hpp; unparnmpc_solver.hpp): default construction, copy and
// move of empty and of live solvers, `getSolution(stage) const` returning a const reference to a SplitSolution.
//   usage: facade_surface              -> the part that needs no GPU
//          facade_surface <iiwa14.urdf> <anymal.urdf>   -> also the live solvers
#include <cmath>
#include <utility>

#include "../../examples/common.hpp"
#include "idocp/cost/configuration_space_cost.hpp"
#include "idocp/cost/task_space_3d_cost.hpp"
#include "idocp/cost/time_varying_task_space_3d_cost.hpp"
#include "idocp/ocp/ocp_solver.hpp"
#include "idocp/ocp/parnmpc_solver.hpp"
#include "idocp/unocp/unocp_solver.hpp"
#include "idocp/unocp/unparnmpc_solver.hpp"
#include "idocp/utils/joint_constraints_factory.hpp"

#define REQUIRE(c) do { if (!(c)) { std::cerr << "FAILED: " #c " (line " << __LINE__ << ")\n"; return 1; } } while (0)

template <typename S>
static int emptySolverSemantics() {
  S a;                    // default constructor
  S b(a);                 // copy of an empty solver
  S c(std::move(b));      // move
  a = c;                  // copy assignment
  c = std::move(a);       // move assignment
  return 0;
}

static double maxDiff(const ex::Vec& a, const ex::Vec& b) {
  if (a.size() != b.size()) return 1e300;
  double m = 0.0;
  for (int i = 0; i < a.size(); ++i) m = std::fmax(m, std::fabs(a[i] - b[i]));
  return m;
}

int main(int argc, char** argv) {
  if (emptySolverSemantics<idocp::OCPSolver>() || emptySolverSemantics<idocp::UnOCPSolver>() ||
      emptySolverSemantics<idocp::ParNMPCSolver>() || emptySolverSemantics<idocp::UnParNMPCSolver>()) return 1;
  std::cout << "empty solvers: ok" << std::endl;
  if (argc < 3) return 0;

  {  // ---- fixed base
    idocp::Robot robot(argv[1]);
    const int n = robot.dimv();
    auto reach = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
    reach->set_q_ref(ex::filled(n, -1));
    ex::attachWeights(*reach, ex::filled(n, 10), ex::filled(n, 0.1), ex::filled(n, 0.01), false);
    auto cost = std::make_shared<idocp::CostFunction>();
    cost->push_back(reach);
    const int N = 12;
    idocp::UnOCPSolver solver(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
    const ex::Vec q = ex::filled(n, 0.5), v = ex::Vec::Zero(n);
    solver.setSolution("q", q);
    solver.setSolution("v", v);
    solver.initConstraints();
    solver.updateSolution(0.0, q, v);
    const idocp::UnOCPSolver& cs = solver;
    const idocp::SplitSolution& s3 = cs.getSolution(3);                       // const overload, const reference
    REQUIRE(maxDiff(s3.u, cs.getSolution("u")[3]) == 0.0 && maxDiff(s3.q, cs.getSolution("q")[3]) == 0.0);
    REQUIRE(maxDiff(cs.getSolution(N).q, cs.getSolution("q")[N]) == 0.0);
    idocp::UnOCPSolver copy(solver);                                          // deep copy: same iterate, independent afterwards
    REQUIRE(maxDiff(copy.getSolution(3).u, s3.u) == 0.0);
    copy.updateSolution(0.0, q, v);
    solver.updateSolution(0.0, q, v);
    REQUIRE(maxDiff(copy.getSolution(5).a, solver.getSolution(5).a) == 0.0);  // the copy carries slack / dual as well: same second step
    idocp::UnOCPSolver assigned;
    assigned = solver;
    REQUIRE(maxDiff(assigned.getSolution(5).a, solver.getSolution(5).a) == 0.0);
    idocp::UnParNMPCSolver pn(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
    pn.setSolution("q", q);
    pn.setSolution("v", v);
    pn.initConstraints();
    pn.initBackwardCorrection(0.0);
    pn.updateSolution(0.0, q, v);
    const idocp::UnParNMPCSolver& cpn = pn;
    REQUIRE(maxDiff(cpn.getSolution(2).v, cpn.getSolution("v")[2]) == 0.0);
    idocp::UnParNMPCSolver pn2(pn);
    REQUIRE(maxDiff(pn2.getSolution(2).lmd, cpn.getSolution(2).lmd) == 0.0);
    {  // OCPSolver / ParNMPCSolver on this robot (no contact frame): bound to the kernels above (ocp_solver.hpp, parnmpc_solver.hpp) -- the step of
       // the unconstrained solver from the same start, bit for bit; setSolution without re-initialising the constraints; copies; the torque gain
      idocp::UnOCPSolver un(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      idocp::OCPSolver oc(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N, 0, 2);
      REQUIRE(oc.unconstrainedSolver() != nullptr && oc.handle() == nullptr);
      un.setSolution("q", q); un.setSolution("v", v);
      oc.setSolution("q", q); oc.setSolution("v", v);
      oc.setSolution("f", ex::V3(0, 0, 1));                                   // (every contact: there is none)
      oc.setContactStatusUniformly(robot.createContactStatus());
      oc.initConstraints(0.0);
      oc.computeKKTResidual(0.0, q, v); un.computeKKTResidual(0.0, q, v);
      REQUIRE(oc.KKTError() == un.KKTError());
      oc.updateSolution(0.0, q, v); un.updateSolution(0.0, q, v);
      const idocp::OCPSolver& coc = oc;
      REQUIRE(maxDiff(coc.getSolution(4).u, un.getSolution(4).u) == 0.0 && maxDiff(coc.getSolution("lmd")[N], un.getSolution("lmd")[N]) == 0.0);
      REQUIRE((int)coc.getSolution("f").size() == N && coc.getSolution("f")[0].size() == 0 && coc.getSolution(4).dimf() == 0);
      Eigen::MatrixXd Kq, Kv, Aq, Av;
      coc.getStateFeedbackGain(2, Kq, Kv);
      un.getStateFeedbackGain(2, Aq, Av);                                       // (the acceleration gain: another matrix)
      REQUIRE(Kq.rows() == n && Kq.cols() == n && Kv.rows() == n && std::isfinite(Kq(0, 0)) && std::fabs(Kq(0, 0) - Aq(0, 0)) > 1e-9);
      idocp::OCPSolver oc2(oc), oc3;
      oc3 = oc;
      oc.updateSolution(0.0, q, v); oc2.updateSolution(0.0, q, v); oc3.updateSolution(0.0, q, v); un.updateSolution(0.0, q, v);
      REQUIRE(maxDiff(oc2.getSolution(5).a, oc.getSolution(5).a) == 0.0 && maxDiff(oc3.getSolution(5).a, oc.getSolution(5).a) == 0.0);
      REQUIRE(maxDiff(un.getSolution(5).a, oc.getSolution(5).a) == 0.0);
      REQUIRE(oc.isCurrentSolutionFeasible());
      idocp::OCPSolver moved(std::move(oc2));
      REQUIRE(moved.unconstrainedSolver() != nullptr && maxDiff(moved.getSolution(5).a, oc.getSolution(5).a) == 0.0);
      // setSolution after initConstraints keeps slack / dual (OCPSolver) -- UnOCPSolver's re-initialises them: the next steps differ
      oc.setSolution("v", ex::filled(n, 0.05)); un.setSolution("v", ex::filled(n, 0.05));
      oc.updateSolution(0.0, q, v); un.updateSolution(0.0, q, v);
      REQUIRE(maxDiff(un.getSolution(5).a, oc.getSolution(5).a) > 0.0);

      idocp::UnParNMPCSolver upn(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      idocp::ParNMPCSolver opn(robot, cost, idocp::JointConstraintsFactory(robot).create(), 0.6, N, 0, 2);
      REQUIRE(opn.unconstrainedSolver() != nullptr);
      upn.setSolution("q", q); upn.setSolution("v", v);
      opn.setSolution("q", q); opn.setSolution("v", v);
      upn.initConstraints(); opn.initConstraints(0.0);
      upn.initBackwardCorrection(0.0); opn.initBackwardCorrection(0.0);
      upn.updateSolution(0.0, q, v); opn.updateSolution(0.0, q, v);
      const idocp::ParNMPCSolver& copn = opn;
      REQUIRE(maxDiff(copn.getSolution(3).q, upn.getSolution(3).q) == 0.0 && maxDiff(copn.getSolution("gmm")[N - 1], upn.getSolution("gmm")[N - 1]) == 0.0);
      opn.computeKKTResidual(0.0, q, v); upn.computeKKTResidual(0.0, q, v);
      REQUIRE(opn.KKTError() == upn.KKTError());
      idocp::ParNMPCSolver opn2(opn);
      opn.updateSolution(0.0, q, v); opn2.updateSolution(0.0, q, v);
      REQUIRE(maxDiff(opn2.getSolution(3).u, opn.getSolution(3).u) == 0.0);
      std::cout << "contact-path solvers on the fixed-base robot: ok" << std::endl;
    }
    {  // the cost function is SHARED with the driver (unocp_solver.hpp: shared_ptr members): a reference moved after construction takes effect
       // at the next call -- equal, bit for bit, to a solver constructed after the move
      auto reach2 = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
      reach2->set_q_ref(ex::filled(n, -1));
      ex::attachWeights(*reach2, ex::filled(n, 10), ex::filled(n, 0.1), ex::filled(n, 0.01), false);
      auto cost2 = std::make_shared<idocp::CostFunction>();
      cost2->push_back(reach2);
      idocp::UnOCPSolver early(robot, cost2, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      idocp::UnParNMPCSolver early_pn(robot, cost2, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      reach2->set_q_ref(ex::filled(n, 0.3));                                    // the driver moves the goal
      reach2->set_qf_weight(ex::filled(n, 25));
      auto reach3 = std::make_shared<idocp::ConfigurationSpaceCost>(robot);       // the moved problem, stated afresh
      reach3->set_q_ref(ex::filled(n, 0.3));
      ex::attachWeights(*reach3, ex::filled(n, 10), ex::filled(n, 0.1), ex::filled(n, 0.01), false);
      reach3->set_qf_weight(ex::filled(n, 25));
      auto cost3 = std::make_shared<idocp::CostFunction>();
      cost3->push_back(reach3);
      idocp::UnOCPSolver late(robot, cost3, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      idocp::UnParNMPCSolver late_pn(robot, cost3, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
      for (idocp::UnOCPSolver* sp : {&early, &late}) { sp->setSolution("q", q); sp->setSolution("v", v); sp->updateSolution(0.0, q, v); }
      REQUIRE(maxDiff(early.getSolution(4).a, late.getSolution(4).a) == 0.0 && maxDiff(early.getSolution(N).lmd, late.getSolution(N).lmd) == 0.0);
      {                                                                         // (and the move matters: a solver of the problem as it was)
        auto reach0 = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
        reach0->set_q_ref(ex::filled(n, -1));
        ex::attachWeights(*reach0, ex::filled(n, 10), ex::filled(n, 0.1), ex::filled(n, 0.01), false);
        auto cost0 = std::make_shared<idocp::CostFunction>();
        cost0->push_back(reach0);
        idocp::UnOCPSolver before(robot, cost0, idocp::JointConstraintsFactory(robot).create(), 0.6, N);
        before.setSolution("q", q); before.setSolution("v", v); before.updateSolution(0.0, q, v);
        REQUIRE(maxDiff(early.getSolution(4).a, before.getSolution(4).a) > 1e-6);
      }
      for (idocp::UnParNMPCSolver* sp : {&early_pn, &late_pn}) { sp->setSolution("q", q); sp->setSolution("v", v); sp->initBackwardCorrection(0.0); sp->updateSolution(0.0, q, v); }
      REQUIRE(maxDiff(early_pn.getSolution(4).a, late_pn.getSolution(4).a) == 0.0);
      std::cout << "shared cost function: ok" << std::endl;
    }
    std::cout << "fixed-base solvers: ok" << std::endl;
    {  // Robot::integrateConfiguration (both overloads) / subtractConfiguration / normalizeConfiguration on the fixed-base arm (robot.hpp:82-147)
      ex::Vec qa = ex::filled(n, 0.25), va = ex::filled(n, -2.0), qb, d;
      robot.integrateConfiguration(qa, va, 0.1, qb);
      REQUIRE(qb.size() == n && std::fabs(qb[3] - 0.05) < 1e-15);
      robot.subtractConfiguration(qb, qa, d);
      REQUIRE(d.size() == n && std::fabs(d[2] + 0.2) < 1e-15);
      robot.integrateConfiguration(va, 0.1, qa);
      REQUIRE(maxDiff(qa, qb) == 0.0);
      robot.normalizeConfiguration(qa);
      REQUIRE(maxDiff(qa, qb) == 0.0);
    }
  }
  {  // ---- floating base
    idocp::Robot robot(argv[2], ex::anymalFeet());
    const ex::Vec stand = ex::anymalStanding();
    auto pose_cost = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
    pose_cost->set_q_ref(stand);
    ex::attachWeights(*pose_cost, ex::filled(18, 10), ex::filled(18, 1), ex::filled(18, 0.01), false);
    const ex::V3 share(0, 0, 70);
    auto cost = std::make_shared<idocp::CostFunction>();
    cost->push_back(pose_cost);
    cost->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), false, &share));
    const int N = 10;
    idocp::OCPSolver solver(robot, cost, ex::jointLimits(robot, 0.7, false, true), 0.25, N);
    ex::Schedule standing(ex::footholds(robot, stand));
    standing.add({0, 1, 2, 3}, 0.0);
    standing.install(solver, robot);
    ex::restingGuess(solver, robot, stand);
    solver.initConstraints(0.0);
    const ex::Vec v = ex::Vec::Zero(robot.dimv());
    solver.updateSolution(0.0, stand, v);
    {  // ... and on the floating base: one sampling period of a twist forwards, the way back through subtractConfiguration, a unit quaternion throughout
      ex::Vec tw = ex::runs({{3, 0.3}, {3, -0.8}, {12, 0.5}}), q1, back;
      robot.integrateConfiguration(stand, tw, 0.2, q1);
      robot.subtractConfiguration(q1, stand, back);
      double worst = 0.0, nq2 = 0.0;
      for (int k = 0; k < 18; ++k) worst = std::fmax(worst, std::fabs(back[k] - 0.2 * tw[k]));
      for (int k = 3; k < 7; ++k) nq2 += q1[k] * q1[k];
      REQUIRE(worst < 1e-13 && std::fabs(nq2 - 1.0) < 1e-14);
      ex::Vec q2 = q1;
      for (int k = 3; k < 7; ++k) q2[k] *= 3.0;
      robot.normalizeConfiguration(q2);
      REQUIRE(maxDiff(q2, q1) < 1e-15);
      {  // frame queries (robot.hxx:206-233): a contact frame's position is its contact point; a rotation is orthonormal; random configurations stay inside the limits
        robot.updateKinematics(q1, v);
        std::vector<Eigen::Vector3d> feet;
        robot.getContactPoints(feet);
        const std::vector<int> ids = robot.contactFramesIndices();
        double off = 0.0, ortho = 0.0;
        for (int c = 0; c < 4; ++c) {
          const Eigen::Vector3d p = robot.framePosition(ids[c]);
          for (int k = 0; k < 3; ++k) off = std::fmax(off, std::fabs(p[k] - feet[c][k]));
        }
        const pinocchio::SE3 X = robot.framePlacement(ids[2]);
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
          double acc = 0.0;
          for (int k = 0; k < 3; ++k) acc += X.rotation()(k, i) * X.rotation()(k, j);
          ortho = std::fmax(ortho, std::fabs(acc - (i == j ? 1.0 : 0.0)));
        }
        REQUIRE(off < 1e-15 && ortho < 1e-14 && std::fabs(robot.frameRotation(ids[2])(1, 2) - X.rotation()(1, 2)) == 0.0);
        const ex::Vec qr = robot.generateFeasibleConfiguration(), lo = robot.lowerJointPositionLimit(), hi = robot.upperJointPositionLimit();
        bool inside = qr.size() == 19;
        double nq = 0.0;
        for (int k = 3; k < 7; ++k) nq += qr[k] * qr[k];
        for (int k = 0; k < 12; ++k) inside = inside && qr[7 + k] >= lo[k] && qr[7 + k] <= hi[k];
        REQUIRE(inside && std::fabs(nq - 1.0) < 1e-14);
      }
      {  // ImpulseStatus (impulse_status.hpp): the contacts that BECOME active between two statuses
        idocp::ImpulseStatus imp = robot.createImpulseStatus();
        idocp::ContactStatus pre = robot.createContactStatus(), post = robot.createContactStatus();
        pre.activateContacts({1, 2});
        post.activateContacts({0, 1, 2, 3});
        imp.setActivity(pre, post);
        REQUIRE(imp.maxPointContacts() == robot.maxPointContacts() && imp.dimf() == 6 && imp.isImpulseActive(0) && !imp.isImpulseActive(1) && imp.isImpulseActive(3));
        imp.deactivateImpulse();
        REQUIRE(!imp.hasActiveImpulse());
      }
      // per-contact properties the reference stores (robot_test.cpp:106: the default friction coefficient is 0.8) and the frame indices
      REQUIRE(robot.frictionCoefficient(2) == 0.8 && robot.restitutionCoefficient(0) == 0.0);
      robot.setFrictionCoefficient({0.5, 0.6, 0.7, 0.9});
      REQUIRE(robot.frictionCoefficient(2) == 0.7 && robot.contactFramesIndices() == ex::anymalFeet());
    }
    const idocp::OCPSolver& cs = solver;
    const idocp::SplitSolution& s0 = cs.getSolution(0);
    REQUIRE(maxDiff(s0.u, cs.getSolution("u")[0]) == 0.0 && maxDiff(s0.q, cs.getSolution("q")[0]) == 0.0);
    // all four feet on the ground: f_stack() = the four contacts' forces, dimf() = 12 (split_solution.hpp:93-122)
    REQUIRE(s0.dimf() == 12 && s0.isContactActive(2));
    REQUIRE(maxDiff(s0.f_stack(), cs.getSolution("f")[0]) == 0.0 && maxDiff(s0.mu_stack(), cs.getSolution("mu")[0]) == 0.0);
    REQUIRE(maxDiff(s0.beta, cs.getSolution("beta")[0]) == 0.0 && maxDiff(s0.nu_passive, cs.getSolution("nu_passive")[0]) == 0.0);
    REQUIRE((int)s0.f.size() == robot.maxPointContacts() && s0.f[1][2] == s0.f_stack()[5]);
    REQUIRE(maxDiff(cs.getSolution(N).v, cs.getSolution("v")[N]) == 0.0);
    {  // the shared cost function on the floating base (ocp_solver.hpp:37-39): weights changed through the component after construction
       // reach the solver at its next call -- the step of a solver constructed with those weights
      auto pc1 = std::make_shared<idocp::ConfigurationSpaceCost>(robot), pc2 = std::make_shared<idocp::ConfigurationSpaceCost>(robot);
      for (auto& pc : {pc1, pc2}) { pc->set_q_ref(stand); ex::attachWeights(*pc, ex::filled(18, 10), ex::filled(18, 1), ex::filled(18, 0.01), false); }
      ex::attachWeights(*pc2, ex::filled(18, 4), ex::filled(18, 2), ex::filled(18, 0.05), false);
      auto c1 = std::make_shared<idocp::CostFunction>(), c2 = std::make_shared<idocp::CostFunction>();
      c1->push_back(pc1); c1->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), false, &share));
      c2->push_back(pc2); c2->push_back(ex::forceCost(robot, ex::V3(0.001, 0.001, 0.001), false, &share));
      idocp::OCPSolver s1(robot, c1, ex::jointLimits(robot, 0.7, false, true), 0.25, N), s2(robot, c2, ex::jointLimits(robot, 0.7, false, true), 0.25, N);
      ex::attachWeights(*pc1, ex::filled(18, 4), ex::filled(18, 2), ex::filled(18, 0.05), false);       // ... after s1 was constructed
      for (idocp::OCPSolver* sp : {&s1, &s2}) {
        standing.install(*sp, robot);
        ex::restingGuess(*sp, robot, stand);
        sp->initConstraints(0.0);
        sp->updateSolution(0.0, stand, v);
      }
      REQUIRE(maxDiff(s1.getSolution(3).u, s2.getSolution(3).u) == 0.0 && maxDiff(s1.getSolution(N).lmd, s2.getSolution(N).lmd) == 0.0);
      REQUIRE(maxDiff(s1.getSolution(3).u, solver.getSolution(3).u) > 1e-9);
    }
    {  // TimeVaryingTaskSpace3DCost on the floating base: a reference that does not move gives the constant-reference solver's step
      struct Still : idocp::TimeVaryingTaskSpace3DRefBase {
        ex::V3 p;
        void compute_q_3d_ref(const double, Eigen::VectorXd& r) const override { r.resize(3); for (int k = 0; k < 3; ++k) r[k] = p[k]; }
      };
      auto still = std::make_shared<Still>();
      still->p = ex::V3(0.35, 0.2, 0.05);
      const int foot = ex::anymalFeet()[0];
      auto tv = std::make_shared<idocp::TimeVaryingTaskSpace3DCost>(robot, foot, still);
      tv->set_q_3d_weight(ex::V3(10, 10, 10));
      auto fixed = std::make_shared<idocp::TaskSpace3DCost>(robot, foot);
      fixed->set_q_3d_ref(still->p);
      fixed->set_q_3d_weight(ex::V3(10, 10, 10));
      auto c1 = std::make_shared<idocp::CostFunction>(), c2 = std::make_shared<idocp::CostFunction>();
      c1->push_back(pose_cost); c1->push_back(tv);
      c2->push_back(pose_cost); c2->push_back(fixed);
      idocp::OCPSolver s1(robot, c1, ex::jointLimits(robot, 0.7, false, true), 0.25, N), s2(robot, c2, ex::jointLimits(robot, 0.7, false, true), 0.25, N);
      for (idocp::OCPSolver* sp : {&s1, &s2}) {
        standing.install(*sp, robot);
        ex::restingGuess(*sp, robot, stand);
        sp->initConstraints(0.1);
        sp->updateSolution(0.1, stand, v);
      }
      REQUIRE(maxDiff(s1.getSolution(3).q, s2.getSolution(3).q) == 0.0 && maxDiff(s1.getSolution(3).u, s2.getSolution(3).u) == 0.0);
      REQUIRE(maxDiff(s1.getSolution(3).q, solver.getSolution(3).q) > 1e-6);      // (and the task cost does act)
    }
    {  // popFrontContactStatus / popBackContactStatus / pushBackContactStatus (ocp_solver.hpp:113-129): a receding horizon through the
       // facade.  Solver A is given the gait [stand | lift LF+RH at 0.12 | touch-down at 0.27 | LH+RF swing at 0.27 ... ] and, once
       // the first event has passed (t = 0.15), pops it and pushes the next phase; solver B is built with the shifted sequence
       // directly.  From the same guess both must take the SAME step (bitwise: same chain, same records), and f_stack() of a trot
       // stage holds the two stance feet only.
      const int Nt = 16;
      const double Tt = 0.8;
      auto limits = ex::jointLimits(robot, 0.7, true);
      idocp::OCPSolver A(robot, cost, limits, Tt, Nt, 3), B(robot, cost, limits, Tt, Nt, 3);
      ex::Schedule gait(ex::footholds(robot, stand));
      gait.add({0, 1, 2, 3}, 0.0);
      gait.add({1, 2}, 0.12);
      gait.advance({0, 3}, 0.05);
      gait.add({0, 3}, 0.37);
      gait.advance({1, 2}, 0.1);
      gait.add({1, 2}, 0.62);
      gait.install(A, robot);
      ex::restingGuess(A, robot, stand);
      A.initConstraints(0.0);
      {
        idocp::OCPSolver D(A);                                                // (A itself takes no step before the shift)
        const idocp::OCPSolver& cA = D;
        D.updateSolution(0.0, stand, v);
        const idocp::SplitSolution& trot = cA.getSolution(5);                 // t = 0.25: LH, RF on the ground
        REQUIRE(trot.dimf() == 6 && !trot.isContactActive(0) && trot.isContactActive(1) && trot.isContactActive(2) && !trot.isContactActive(3));
        REQUIRE(trot.f_stack().size() == 6 && trot.f_stack()[2] == trot.f[1][2] && trot.f_stack()[5] == trot.f[2][2]);
        REQUIRE((int)trot.f.size() == 4);                                     // the per-contact vectors cover every contact
        REQUIRE(cA.getSolution(1).dimf() == 12);
      }
      // shift: the lift at 0.12 has passed at t = 0.15
      ex::Schedule next = gait;
      next.advance({0, 3}, 0.1);
      auto status = robot.createContactStatus();
      status.activateContacts({0, 3});
      status.setContactPoints(next.feet);
      A.popFrontContactStatus();
      A.pushBackContactStatus(status, 0.87);
      ex::Schedule shifted(gait.rows[1].points);
      shifted.rows.push_back(gait.rows[1]);
      shifted.rows.push_back(gait.rows[2]);
      shifted.rows.push_back(gait.rows[3]);
      shifted.rows.push_back({{0, 3}, next.feet, 0.87});
      shifted.install(B, robot);
      ex::restingGuess(B, robot, stand);
      A.initConstraints(0.15);
      B.initConstraints(0.15);
      A.updateSolution(0.15, stand, v);
      B.updateSolution(0.15, stand, v);
      for (int i : {0, 3, 7, 12, Nt}) {
        REQUIRE(maxDiff(A.getSolution(i).q, B.getSolution(i).q) == 0.0 && maxDiff(A.getSolution(i).lmd, B.getSolution(i).lmd) == 0.0);
      }
      REQUIRE(A.getSolution(0).dimf() == 6 && maxDiff(A.getSolution(0).f_stack(), B.getSolution(0).f_stack()) == 0.0);
      // popBack: the phase pushed last leaves again; a solver built without it takes the same step
      idocp::OCPSolver Cs(robot, cost, limits, Tt, Nt, 3);
      shifted.rows.pop_back();
      shifted.install(Cs, robot);
      idocp::OCPSolver A2(robot, cost, limits, Tt, Nt, 3);                     // A has stepped: the same pops on a fresh solver
      gait.install(A2, robot);
      A2.popFrontContactStatus();
      A2.pushBackContactStatus(status, 0.87);
      A2.popBackContactStatus();
      A = A2;
      ex::restingGuess(A, robot, stand);
      ex::restingGuess(Cs, robot, stand);
      A.initConstraints(0.15);
      Cs.initConstraints(0.15);
      A.updateSolution(0.15, stand, v);
      Cs.updateSolution(0.15, stand, v);
      REQUIRE(maxDiff(A.getSolution(9).q, Cs.getSolution(9).q) == 0.0 && maxDiff(A.getSolution(9).u, Cs.getSolution(9).u) == 0.0);
      std::cout << "receding horizon through the facade: ok" << std::endl;
    }
    std::cout << "floating-base solver: ok" << std::endl;
  }
  return 0;
}
