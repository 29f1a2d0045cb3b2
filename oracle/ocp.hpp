// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// CPU restatement of the contact-capable half of the idocp hot path, including
// horizons with discrete events: SplitOCP / ImpulseSplitOCP / TerminalOCP,
// ContactDynamics / ImpulseDynamicsForwardEuler, the state equations with a
// floating base, ForwardSwitchingConstraint, the (constrained) Riccati
// factorizers, RiccatiRecursionSolver, OCPLinearizer, ContactSequence,
// OCPDiscretizer and OCPSolver.  Every function cites the reference lines it follows.
//
// Storage: like the reference (hybrid_container.hpp:60-168) every stage owns a fixed
// SLOT -- grid stage i -> i, impulse k -> N+1+k, aux k -> N+1+E+k, lift k -> N+1+2E+k
// (E = max number of events) -- so nothing moves when the discretisation changes.
// The discretiser produces the CHAIN: the slots in time order
//   stage, [impulse, aux | lift], stage, ..., terminal.
// Every neighbour relation of the reference (q_prev, s_next, d_next, riccati_next;
// ocp_linearizer.hxx:113-248, riccati_recursion_solver.cpp:48-162) is the chain
// predecessor / successor.  The HIP path uses the same slots and the same chain.
#ifndef ORACLE_OCP_HPP_
#define ORACLE_OCP_HPP_

#include <string>
#include <vector>

#include "idocp_hip.h"
#include "mat.hpp"
#include "rbd.hpp"
#include "unocp.hpp"      // LineSearchFilterC

namespace oracle {

// include/idocp/robot/contact_status.hxx, impulse_status.hxx: a set of active point contacts + their world points
struct ContactStatus {
  std::vector<bool> active;
  std::vector<Mat> points;       // world contact points
  int dimf() const { int n = 0; for (bool a : active) n += a ? 3 : 0; return n; }
  bool hasActiveContacts() const { return dimf() > 0; }
};

// include/idocp/hybrid/contact_sequence.hxx:56-333
struct ContactSequenceC {
  std::vector<ContactStatus> phases;           // contact_statuses_
  std::vector<real> event_time;              // one per discrete event
  std::vector<bool> is_impulse;                // DiscreteEvent::existImpulse (discrete_event.hxx:57-84)
  std::vector<ContactStatus> impulse_status;   // per EVENT (only meaningful where is_impulse)
  int numEvents() const { return (int)event_time.size(); }
  int numImpulse() const { int n = 0; for (bool b : is_impulse) n += b ? 1 : 0; return n; }
  int numLift() const { return numEvents() - numImpulse(); }
  int eventOfImpulse(int k) const;             // event index of the k-th impulse
  int eventOfLift(int k) const;
};

// One stage of the chain.
struct NodeC {
  enum Kind { Stage = 0, Impulse = 1, Aux = 2, Lift = 3, Terminal = 4 };
  int kind = Stage;
  int slot = 0;           // storage slot of this stage
  int index = 0;          // grid stage (Stage / Terminal), impulse index (Impulse / Aux), lift index (Lift)
  real t = 0, dt = 0;   // dt = 0 for Impulse / Terminal
  int phase = 0;          // contact phase (Stage / Aux / Lift); for Impulse: the event index
  int level = 0;          // time step handed to Constraints::createConstraintsData: grid stage, 0 (aux, lift), -1 (impulse)
  int sw_event = -1;      // event index of the impulse whose switching constraint sits on this stage (or -1)
  real sw_dt_next = 0;  // dt of the stage between this one and the impulse
};

// include/idocp/ocp/split_solution.hxx:10-31, impulse/impulse_split_solution.hxx (a = dv on impulse stages)
struct SplitSolutionC {
  Mat lmd, gmm, q, v, a, u, beta, nu_passive, xi;
  std::vector<Mat> f, mu;        // per contact (3)
  explicit SplitSolutionC(const Robot& r);
  Mat f_stack(const ContactStatus& cs) const;
  Mat mu_stack(const ContactStatus& cs) const;
};

// include/idocp/ocp/split_direction.hxx:8-23
struct SplitDirectionC {
  Mat dlmd, dgmm, du, dq, dv, daf, dbetamu, dnu_passive, dxi;
  explicit SplitDirectionC(const Robot& r);
};

// rows of the friction-cone component of one contact (coneEval, ocp.cpp)
struct ConeEval { int nr; real res[5]; real J[5][3]; };
ConeEval coneEval(int kind, real mu, const Mat& f);

// IPM components: 0/1 joint position lower/upper, 2/3 velocity, 4/5 torque, 6 (linearized) friction cone, 7 unused, 8/9 joint
// acceleration lower/upper; odd = upper bound (sign +1); 10 contact distance
constexpr int NCOMP = 11;      // 10: ContactDistance (one row per contact)
inline bool jointComp(int c) { return c < 6 || c == 8 || c == 9; }
struct IpmData {                  // ConstraintComponentData
  Mat slack, dual, residual, duality, dslack, ddual;
  explicit IpmData(int n = 0) : slack(n), dual(n), residual(n), duality(n), dslack(n), ddual(n) {}
};

// SplitKKTMatrix / SplitKKTResidual (include/idocp/ocp/split_kkt_matrix.hxx:11-30,75-497;
// split_kkt_residual.hxx:10-26; impulse twins) with the blocks kept as separate matrices.
struct SplitKKTMatrixC {
  int nv, nu;
  Mat Qxx, Qxu_full, Quu_full;    // (2nv x 2nv), (2nv x nv), (nv x nv); u_full = [passive(6) ; u]
  Mat Qaa_diag, Qff;              // nv ; dimf x dimf
  Mat Fqq6, Fqv6;                 // leading 6x6 blocks (rest is I, dt I implicitly)
  Mat Fvq, Fvv, Fvu;              // nv x nv, nv x nv, nv x nu
  Mat Fqq_prev6, Fqq_inv, Fqq_prev_inv;   // 6x6
  SplitKKTMatrixC(int nv_, int nu_);
};
struct SplitKKTResidualC {
  Mat Fq, Fv, lq, lv, la, lf, lu, lu_passive, Fq_prev, P;
  explicit SplitKKTResidualC(int nv, int nu);
};

// ContactDynamicsData / ImpulseDynamicsForwardEulerData (include/idocp/ocp/contact_dynamics_data.hxx)
struct ContactDynamicsDataC {
  Mat dIDda, dCda, dIDCdqv, MJtJinv, MJtJinv_dIDCdqv, Qafqv, Qafu_full, IDC, MJtJinv_IDC, laf;
};

// SplitStateConstraintJacobian + SplitConstrainedRiccatiFactorization
// (split_state_constraint_jacobian.hxx, split_constrained_riccati_factorization.hxx)
struct SwitchingC {
  Mat Phix, Phia, Phiu;          // dimi x 2nv, dimi x nv, dimi x nu
  Mat M, m;                      // dxi = M dx + m
};

// Test hook (tests/golden/gen_golden_kkt.py): the UN-CONDENSED Newton system of a stage, captured by linearizeNode / linearizeTerminal just
// before condenseContactDynamics when OCPSolver::keep_uncondensed is set -- cost + IPM Hessians and gradients in all of (q, v, a, f, u), the raw
// state-equation Jacobians, [dID; dC] / d(q, v, a), [ID; C], the switching-constraint rows.  A dense solve of the whole horizon's KKT system
// assembled from these shares no formula with condensation, Riccati recursion and expansion.
struct UncondensedC {
  bool valid = false;
  int kind = 0, dimf = 0, dimi = 0, has_u = 0, active_mask = 0;      // active_mask: bit c = contact c carries rows (impulse stages: the contacts of the impulse)
  real dt = 0, dtq = 0;
  Mat Qxx, Qaa, Qff, Quu, lq, lv, la, lf, lu, lu_passive, Fq, Fv, Fqq, Fqq_prev, dIDCdqv, M, J, IDC, Phix, Phia, P;
  Mat aux_next;      // ParNMPC: what the coarse update adds to Qxx for the stage behind this one (BackwardCorrectionSolver::aux_mat_ of the next stage)
};

struct RiccatiC {
  Mat Pqq, Pqv, Pvv, sq, sv;
  explicit RiccatiC(int nv) : Pqq(nv, nv), Pqv(nv, nv), Pvv(nv, nv), sq(nv), sv(nv) {}
};

class OCPSolver {
 public:
  OCPSolver(const RModel& model, const RCost& cost, const idocp_constraints_t& constraints, real T, int N,
            int max_num_impulse = 0);
  void setContactStatusUniformly(const std::vector<int>& active, const double* contact_points /*[nc][3]*/);   // ocp_solver.cpp:169-171
  void pushBackContactStatus(const std::vector<int>& active, const double* contact_points, real switching_time);   // :174-177
  void setContactPoints(int contact_phase, const double* contact_points);                                     // :180-184
  void popBackContactStatus();                                        // :187-189 -> ContactSequence::pop_back (contact_sequence.hxx:117-136)
  void popFrontContactStatus();                                       // :192-194 -> ContactSequence::pop_front (contact_sequence.hxx:139-160)
  void setSolution(const std::string& name, const Mat& value);      // ocp_solver.cpp:95-165
  void initConstraints(real t);                                   // ocp_solver.cpp:60-64
  void updateSolution(real t, const Mat& q, const Mat& v, bool line_search = false);         // ocp_solver.cpp:67-92
  // LineSearch::computeCostAndViolation of the trial iterate s (+) alpha d (src/line_search/line_search.cpp:63-196;
  // alpha = 0: the current iterate with the current slacks) and the filter of OCPSolver (ocp_solver.cpp:84-90, 196-199)
  std::pair<real, real> costAndViolation(real alpha);
  LineSearchFilterC line_search;
  void computeKKTResidual(real t, const Mat& q, const Mat& v);     // ocp_solver.cpp:202-207
  real KKTError();                                                 // ocp_linearizer.cpp:98-137
  int isCurrentSolutionFeasible() const;                             // ocp_solver.cpp:216-248: first offending chain position or -1

  void discretize(real t);                                         // OCPDiscretizer::discretizeOCP (ocp_discretizer.hxx:65-374)
  void linearizeOCP(real t, const Mat& q);                         // K5
  void backwardRiccatiRecursion();                                   // S3
  void forwardRiccatiRecursion(const Mat& q, const Mat& v);          // S4 (+ initial state direction)
  void computeDirection();                                           // K6
  void integrateSolution();                                          // K7

  int N() const { return N_; }                 // grid stages after discretisation (N_ideal minus events on the grid)
  int M() const { return (int)chain.size(); }  // chain length = N + 1 + 2 N_impulse + N_lift
  real stepDt() const { return dt_; }
  int dimc() const;
  const ContactStatus& nodeContacts(int p) const;     // contact (or impulse) status a node is linearised with
  Robot robot;
  Robot task_robot;                          // the robot with the task frame of a TaskSpace3D / 6D cost as its contact 0
  RCost cost;
  idocp_constraints_t cons;
  ContactSequenceC seq;
  std::vector<NodeC> chain;
  int slotOf(int kind, int index) const;
  int posOfSlot(int slot) const { for (int p = 0; p < (int)chain.size(); ++p) if (chain[p].slot == slot) return p; return -1; }
  int nslots() const { return N_ideal_ + 1 + 3 * max_events_; }
  // all per-stage arrays are indexed by SLOT; chain[p].slot maps a chain position to it
  std::vector<SplitSolutionC> s;
  std::vector<SplitDirectionC> d;
  std::vector<SplitKKTMatrixC> kkt_matrix;
  std::vector<SplitKKTResidualC> kkt_residual;
  std::vector<ContactDynamicsDataC> cd;
  std::vector<SwitchingC> sw;
  std::vector<std::vector<IpmData>> ipm;    // [node][component]
  std::vector<Mat> cd_J;                    // ContactDistance: row 2 of the LOCAL frame Jacobians of the linearisation (nc x nv per slot)
  bool keep_uncondensed = false;            // test hook: capture the un-condensed stage systems (UncondensedC) during linearizeOCP
  std::vector<UncondensedC> unc;            // [slot]
  std::vector<RiccatiC> riccati;
  std::vector<Mat> K, k;
  real primal_step_size = 1, dual_step_size = 1;
  real riccati_seconds = 0;
  void qRef(real t, Mat& q_ref) const;                              // trotting_configuration_space_cost.hpp:126-164

 private:
  int N_ideal_, N_, nv_, nu_, nc_, max_events_;
  int kP;                    // passive rows of the floating base (6), 0 on a fixed-base robot
  real T_, dt_;
  bool discretized_ = false;
  // components: 0..5 joint limits (q lo/up, v lo/up, u lo/up), 6 friction cone (impulse cone on impulse stages)
  bool componentEnabled(int c, bool impulse) const;
  bool componentValid(int c, const NodeC& nd) const;
  int componentDim(int c, bool impulse = false) const { return jointComp(c) ? nu_ : (c == 6 ? coneRows(impulse) * nc_ : (c == 10 ? nc_ : 0)); }
  int coneKind(bool impulse) const { return (impulse ? cons.impulse_friction_cone : cons.friction_cone) ? 1 : 0; }
  int coneRows(bool impulse) const { return coneKind(impulse) == 1 ? 2 : 5; }
  void initNodeConstraints(const NodeC& nd);
  // the stage loops run under `#pragma omp parallel for num_threads(nthreads)` where the reference's do (ocp_linearizer.cpp:47,
  // 74-83, 104, 152; riccati_recursion_solver.cpp:174-239), each thread with a Robot of its own like the reference's
  // robots[omp_get_thread_num()] (the parameter shadows the member on purpose)
  void linearizeNode(Robot& robot, int p, const Mat& q_prev, bool residual_only);
  void linearizeTerminal(Robot& robot, int p, const Mat& q_prev, bool residual_only);
  std::vector<Robot> robots_;
 public:
  int nthreads = 1;
  void setNumThreads(int n) { nthreads = n < 1 ? 1 : n; robots_.assign(nthreads, robot); }
};

// ParNMPCSolver (src/ocp/parnmpc_solver.cpp:66-103): backward-Euler stages (SplitParNMPC / TerminalParNMPC,
// include/idocp/ocp/split_parnmpc.hxx, terminal_parnmpc.hxx), per-stage KKT inverse (SplitKKTMatrixInverter,
// split_kkt_matrix_inverter.hxx:44-166), coarse update and the four correction sweeps of BackwardCorrectionSolver
// (src/ocp/backward_correction_solver.cpp:62-490, split_backward_correction.hxx:30-155).
// Stage i (0 <= i < N) lives at time t + (i + 1) dt; the state before stage 0 is the measured (q, v).
//
// Horizons with discrete events (ParNMPCDiscretizer, include/idocp/hybrid/parnmpc_discretizer.hxx:65-397): the event
// stages sit IN FRONT of the grid stage that follows the event,
//   ..., stage i-1, [aux k, impulse k | lift k], stage i, ...
// aux / lift are ordinary backward-Euler stages with their own time step (the aux stage also carries the switching
// constraint P(q) = 0 of the impulse, switching_constraint.hxx:8-21, as rows Pq of its KKT matrix); the impulse stage is
// ImpulseSplitParNMPC (impulse_split_parnmpc.hxx:33-60) with ImpulseDynamicsBackwardEuler (impulse_dynamics_backward_euler
// .hxx:20-128) and its own KKT matrix in the variables (lmd, gmm, mu | f, q, v) (impulse_split_kkt_matrix_inverter.hxx:34-120,
// impulse_split_backward_correction.hxx:30-127).  As in the OCP oracle every neighbour relation of the reference is the
// chain predecessor / successor, and every stage owns a fixed slot: grid stage i -> i, impulse k -> N + k,
// aux k -> N + E + k, lift k -> N + 2E + k.
class ParNMPCSolver {
 public:
  ParNMPCSolver(const RModel& model, const RCost& cost, const idocp_constraints_t& constraints, real T, int N,
                int max_num_impulse = 0);
  void setContactStatusUniformly(const std::vector<int>& active, const double* contact_points);
  void pushBackContactStatus(const std::vector<int>& active, const double* contact_points, real switching_time);
  void popBackContactStatus();                                        // parnmpc_solver.cpp:202-204
  void popFrontContactStatus();                                       // parnmpc_solver.cpp:207-209
  void setSolution(const std::string& name, const Mat& value);
  void initBackwardCorrection(real t);                              // parnmpc_solver.cpp:66-70
  void initConstraints(real t);                                     // parnmpc_linearizer.cpp:43-75
  void updateSolution(real t, const Mat& q, const Mat& v, bool line_search = false);           // parnmpc_solver.cpp:73-103
  void computeDirection(real t, const Mat& q, const Mat& v);          // the same without integrateSolution (line-search tests)
  // LineSearch::computeCostAndViolation of the trial iterate s (+) alpha d for ParNMPC (src/line_search/line_search.cpp:199-237;
  // event-free horizons only: throws std::logic_error when the chain holds impulse / aux / lift stages)
  std::pair<real, real> costAndViolation(real alpha, const Mat& q, const Mat& v);
  LineSearchFilterC line_search;
  void computeKKTResidual(real t, const Mat& q, const Mat& v);
  real KKTError();                                                   // parnmpc_linearizer.cpp:203-247
  int isCurrentSolutionFeasible() const;                               // parnmpc_solver.cpp:231-273: first offending chain position or -1
  // the phases of updateSolution, separately callable
  void coarseUpdate(real t, const Mat& q, const Mat& v);
  void backwardCorrectionSerial();
  void backwardCorrectionParallel();
  void forwardCorrectionSerial();
  void forwardCorrectionParallel();
  void integrateSolution();
  int N() const { return N_; }                 // grid stages of the current discretisation
  int M() const { return (int)chain.size(); }  // chain length = N + 2 N_impulse + N_lift
  struct PNode {
    int kind = NodeC::Stage;   // Stage, Impulse, Aux, Lift, Terminal (= the last grid stage, which carries the terminal cost)
    int slot = 0, index = 0;   // index: grid stage / impulse index / lift index
    real t = 0, dt = 0;
    int phase = 0;             // contact phase (Stage / Aux / Lift / Terminal)
    int level = 0;             // time step of Constraints::createConstraintsData: i + 1 (stage i), 0 (aux, lift), -1 (impulse)
    int event = -1;            // Aux / Impulse: event index of the impulse
  };
  void discretize(real t);                                           // ParNMPCDiscretizer::discretizeOCP
  bool keep_uncondensed = false;            // test hook: capture the un-condensed stage systems (UncondensedC) in linearizeNode
  std::vector<UncondensedC> unc;            // [slot]
  std::vector<PNode> chain;
  ContactSequenceC seq;
  const ContactStatus& nodeContacts(const PNode& nd) const { return nd.kind == NodeC::Impulse ? seq.impulse_status[nd.event] : seq.phases[nd.phase]; }
  int slotOf(int kind, int index) const;
  int nslots() const { return N_ideal_ + 3 * max_events_; }
  Robot robot;
  Robot task_robot;                          // the robot with the task frame of a TaskSpace3D / 6D cost as its contact 0
  RCost cost;
  idocp_constraints_t cons;
  ContactStatus contact_status;      // the first contact phase (the only one of an event-free horizon)
  // all per-stage arrays are indexed by SLOT
  std::vector<SplitSolutionC> s, s_new;
  std::vector<SplitDirectionC> d;
  std::vector<SplitKKTMatrixC> kkt_matrix;
  std::vector<SplitKKTResidualC> kkt_residual;
  std::vector<ContactDynamicsDataC> cd;
  std::vector<std::vector<IpmData>> ipm;
  std::vector<Mat> cd_J;                             // ContactDistance rows of the linearisation (nc x nv per slot)
  std::vector<Mat> KKT_mat_inv, aux_mat, x_res;      // dimKKT^2, nx^2, nx per stage
  std::vector<Mat> sw_Pq;                            // aux stages: Pq (dimi x nv); the residual P sits in kkt_residual.P
  // ImpulseDynamicsBackwardEulerData + the impulse blocks of ImpulseSplitKKTMatrix (per impulse slot)
  struct ImpulseDataC { Mat ImD, dImDdq, dImDddv, Minv, Minv_ImD, Qdvq, Qdvf, ldv, Fvq, Fvf, Vq, Vv, Qqf, Qdvdv; };
  std::vector<ImpulseDataC> imp;
  real primal_step_size = 1, dual_step_size = 1;
  real serial_seconds = 0;
  // ---- horizon sharding (stages of one horizon spread over several processes; event-free
  // horizons only) ----
  // This object then owns the stages [stage_offset, stage_offset + N) of a longer horizon.  What it needs from its
  // neighbours arrives through importHalo: the state in front of its first stage (has_prev), the first stage of the
  // right neighbour (lmd, gmm, q for the coupling terms; aux_mat; corrected lmd, gmm for the backward sweep) and the
  // corrected state of the left neighbour's last stage for the forward sweep.
  int stage_offset = 0;
  bool has_terminal = true, has_prev = false;
  // Horizons WITH discrete events are sharded by a slice of the chain instead: this object discretises the whole horizon and
  // keeps the grid stages [slice_begin, slice_end) together with the event stages in front of each of them (slots and
  // constraint levels stay the global ones); has_prev / has_terminal follow from the slice.  slice_end < 0: the whole chain.
  int slice_begin = 0, slice_end = -1;
  void setChainSlice(int stage_begin, int stage_end) { slice_begin = stage_begin; slice_end = stage_end; discretized_ = false; }
  SplitSolutionC next_s, next_snew, prev_s, prev_snew;
  Mat next_aux;
  // kinds: 0 state_last (q, v of the last stage -> right), 1 costate_first (lmd, gmm, q of the first stage -> left),
  //        2 aux_first (-> left), 3 bwd_first (s_new.lmd, s_new.gmm of the first stage -> left),
  //        4 fwd_last (s_new.q, s_new.v of the last stage -> right), 5 aux_all (aux_mat of every stage, init only)
  int haloSize(int kind) const;
  void exportHalo(int kind, real* out) const;
  void importHalo(int kind, const real* in);
  real KKTErrorSquared();

 private:
  int N_ideal_, N_, nv_, nu_, nc_, max_events_;
  int kP;                    // passive rows of the floating base (6), 0 on a fixed-base robot
  real T_, dt_;
  real disc_t_ = 0;
  bool discretized_ = false;
  bool componentValid(int c, const PNode& nd) const;
  int componentDim(int c, bool impulse = false) const { return jointComp(c) ? nu_ : (c == 6 ? coneRows(impulse) * nc_ : (c == 10 ? nc_ : 0)); }
  int coneKind(bool impulse) const { return (impulse ? cons.impulse_friction_cone : cons.friction_cone) ? 1 : 0; }
  int coneRows(bool impulse) const { return coneKind(impulse) == 1 ? 2 : 5; }
  void qRef(real t, Mat& q_ref) const;
  void initNodeConstraints(const PNode& nd);
  void linearizeNode(int p, const Mat& q_prev, const Mat& v_prev, bool residual_only);
  void linearizeImpulse(int p, const Mat& q_prev, const Mat& v_prev, bool residual_only);
  const SplitSolutionC* nextSolution(int p) const;       // s of the chain successor (next_s behind the last stage of a shard)
};

}  // namespace oracle
#endif  // ORACLE_OCP_HPP_
