// C-ABI implementation of the OCPSolver path (include/idocp_hip.h).
//
// Sequences the contact-path kernels like OCPSolver::updateSolution
// (src/ocp/ocp_solver.cpp:67-92): [host: per-stage reference q_ref(t_i)] ->
// tangent RNEA (K5a) -> condensation (K5b) -> backward / forward Riccati (S3, S4)
// -> expand primal + step sizes (K6) -> expand dual + integrate (K7).
// The hybrid discretisation (OCPDiscretizer) is host-side index logic; this
// build accepts only event-free horizons, so the schedule is the plain grid.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <string>
#include <vector>

#include "dev_lie.hpp"
#include "host_util.hpp"
#include "idocp_hip.h"
#include "ocp_launch.hpp"

using namespace idocp_dev;
using idocp_host::set_last_error;

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess) {                                                                   \
      set_last_error(std::string(#expr) + ": " + hipGetErrorString(e_));                      \
      (void)hipGetLastError(); /* HIP keeps a failed call as the thread's "last error": reported here, it must not fail the next handle's launches */ \
      return IDOCP_E_DEVICE;                                                                  \
    }                                                                                         \
  } while (0)

namespace {

using DQ = LeggedDims<4, 3>;
using LQ = OcpLayout<DQ>;

void toDevModelOcp(const idocp_model_t& m, DevModel& d) {
  std::memset(&d, 0, sizeof(d));
  d.njoints = m.njoints; d.nq = m.nq; d.nv = m.nv; d.nu = m.nu; d.has_floating_base = m.has_floating_base;
  for (int i = 0; i < m.njoints; ++i) {
    d.parent[i] = m.parent[i]; d.jtype[i] = m.jtype[i]; d.idx_q[i] = m.idx_q[i]; d.idx_v[i] = m.idx_v[i];
    std::memcpy(d.axis[i], m.axis[i], sizeof(double) * 3);
    std::memcpy(d.R[i], m.plc_R[i], sizeof(double) * 9);
    std::memcpy(d.p[i], m.plc_p[i], sizeof(double) * 3);
    d.mass[i] = m.mass[i];
    const double* c = m.com[i];
    const double* I = m.inertia[i];
    const double ms = m.mass[i];
    for (int k = 0; k < 3; ++k) d.mc[i][k] = ms * c[k];
    const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    d.Io[i][0] = I[0] + ms * (cc - c[0] * c[0]); d.Io[i][1] = I[1] - ms * c[0] * c[1]; d.Io[i][2] = I[2] - ms * c[0] * c[2];
    d.Io[i][3] = I[4] + ms * (cc - c[1] * c[1]); d.Io[i][4] = I[5] - ms * c[1] * c[2]; d.Io[i][5] = I[8] + ms * (cc - c[2] * c[2]);
  }
  std::memcpy(d.gravity, m.gravity, sizeof(double) * 3);
}

// free-flyer (identity placement) + 4 chains of 3 revolute joints, contact c on the tip joint of leg c
bool isQuadruped(const idocp_model_t& m) {
  if (!m.has_floating_base || m.njoints != DQ::NJ || m.nv != DQ::NV || m.nq != DQ::NQ || m.ncontacts != DQ::NC) return false;
  if (m.jtype[0] != IDOCP_JOINT_FREEFLYER || m.parent[0] != -1) return false;
  const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  for (int k = 0; k < 9; ++k) if (std::fabs(m.plc_R[0][k] - I3[k]) > 1e-14) return false;
  for (int k = 0; k < 3; ++k) if (std::fabs(m.plc_p[0][k]) > 1e-14) return false;
  for (int leg = 0; leg < DQ::NL; ++leg)
    for (int j = 0; j < DQ::LJ; ++j) {
      const int ji = 1 + leg * DQ::LJ + j;
      if (m.jtype[ji] != IDOCP_JOINT_REVOLUTE || m.parent[ji] != (j == 0 ? 0 : ji - 1) || m.idx_v[ji] != 6 + leg * DQ::LJ + j) return false;
    }
  for (int c = 0; c < DQ::NC; ++c) if (m.contact_joint[c] != DQ::LJ * (c + 1)) return false;
  return true;
}

}  // namespace

// idocp::ContactStatus / ImpulseStatus on the host (include/idocp/robot/contact_status.hxx)
struct HostStatus {
  int active[IDOCP_MAX_CONTACTS] = {0, 0, 0, 0};
  double points[IDOCP_MAX_CONTACTS][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  int dimf() const { int n = 0; for (int c = 0; c < DQ::NC; ++c) n += active[c] ? 3 : 0; return n; }
};

struct idocp_ocp {
  idocp_model_t model;
  idocp_cost_t cost;
  idocp_constraints_t cons;
  int N, E, NS, batch, device;         // N = grid intervals (N_ideal), E = max number of discrete events, NS = slots per instance
  double T;
  hipStream_t stream = nullptr;
  OcpBuffers B{};
  OcpProblem prob;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;   // parallel to allocs (idocp_ocp_clone copies buffer by buffer)
  double *d_q0 = nullptr, *d_v0 = nullptr, *d_tmp = nullptr, *d_qref = nullptr, *d_taskref = nullptr;
  // filter line search on a shard of a ParNMPC horizon (idocp_parnmpc_set_line_search_hooks, parnmpc_dist.hip): the trial iterate of the
  // state in front of the shard's first stage, and the two collective steps of one probe
  double *d_qtry = nullptr, *d_vtry = nullptr;
  int (*ls_pre)(idocp_ocp_t*) = nullptr;       // after the trial iterate is formed: exchange its boundary state
  int (*ls_post)(idocp_ocp_t*) = nullptr;      // after the shard's sums are formed: all-reduce them
  std::vector<double> task_refs_host;      // idocp_ocp_set_task_refs: [M][12] for the chain discretised at task_refs_t
  double task_refs_t = 0.0;
  bool task_refs_lenient = false;          // creation, clone and the chain getters discretise without poses (constant pose in the table)
  bool task_refs_stale = false;            // ... and leave the table to be redone by the next strict discretisation
  void* d_prob = nullptr;
  OcpNode* d_nodes = nullptr;
  bool contact_status_set = false;
  bool parnmpc = false;               // backward-Euler stages + backward correction instead of the Riccati sweep
  int stage_offset = 0;               // ParNMPC horizon sharding: global index of the first stage of this shard
  bool has_terminal = true, has_prev = false;
  // ContactSequence (include/idocp/hybrid/contact_sequence.hxx:56-333)
  std::vector<HostStatus> phases;
  std::vector<double> event_time;
  std::vector<char> is_impulse;
  std::vector<HostStatus> impulse_status;          // per event
  // chain of the last discretisation (OCPDiscretizer, ocp_discretizer.hxx:65-374)
  std::vector<OcpNode> chain;
  std::vector<int> chain_index;
  std::vector<double> chain_t;
  int Ngrid = 0;                      // grid stages after discretisation
  double disc_time = NAN;
  bool seq_dirty = true, has_switch = false;
  int n_impulse = 0;
  int* d_impulse_pos = nullptr;
  int* d_switch_pos = nullptr;
  int* d_general_pos = nullptr;     // ParNMPC: chain positions of the aux (switching rows) / impulse stages
  int* d_cond_pos = nullptr;        // chain positions by stage class of K5b (all feet | half of them | the rest)
  int cond_n[5] = {0, 0, 0, 0, 0};      // all feet | half of them | the rest | event stages with half of the feet | flight stages
  int n_general = 0;
  int slice_begin = 0, slice_end = -1;   // ParNMPC with events: this handle keeps the grid stages [slice_begin, slice_end) of the chain (-1: all)
  int uniform_dimf = -1;              // dimf shared by all stages of an event-free chain, else -1
  int M() const { return (int)chain.size(); }
  // hipGraph of one updateSolution (idocp_ocp_update_solution_graph): valid while the discretisation and the input buffers stay
  long disc_epoch = 0, graph_epoch = -1;
  hipGraph_t graph = nullptr;
  hipGraphExec_t graph_exec = nullptr;
  const double *graph_q = nullptr, *graph_v = nullptr;
  bool launched_eagerly = false;
  // filter line search (LineSearchFilter, src/line_search/line_search_filter.cpp): one filter per instance
  std::vector<std::vector<std::pair<double, double>>> filters;
  OcpNode* d_nodes_ls = nullptr;
  double* ext_try = nullptr;
  int fused_forward_mode = -1;        // idocp_ocp_set_fused_forward: -1 by batch size, 0 S4 + K6, 1 the fused forward sweep
  int riccati_sweep_mode = -1;        // idocp_ocp_set_riccati_sweep: -1 by batch size, 0 one wavefront per instance (register-resident), 1 eight per instance
  // staging of the per-stage setters (idocp_ocp_set_solution_stages / _chain, the aux-matrix setters): an MPC loop warm-starts every tick, so
  // the device scratch and its pinned host twin are kept (grown on demand) and the setter returns without a stream synchronisation -- the
  // copy and the fill kernel are stream-ordered in front of whatever the caller launches next; fill_done guards the reuse of the host twin
  double *d_fill = nullptr, *h_fill = nullptr;
  size_t fill_cap = 0;
  hipEvent_t fill_done = nullptr;
  // fork / join inside an iteration (round 6): the switching-constraint kernel K5s -- one latency-bound wavefront per stage that carries a
  // switching constraint, 0.09 ms on configs[2] with most of the chip idle -- runs on a stream of its own NEXT TO the nominal sweeps K5n (both
  // read the iterate only and write records of their own); the condensation launches wait for both.  IDOCP_SIDE_STREAM=0 keeps everything on
  // the handle's stream.
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool side_pending = false;          // a kernel is in flight on `side` that the next condensation launch has to wait for
};

namespace {

int allocBufO(idocp_ocp* h, double** p, size_t n) {
  HIP_TRY(hipMalloc((void**)p, n * sizeof(double)));
  h->allocs.push_back(*p); h->alloc_bytes.push_back(n * sizeof(double));
  HIP_TRY(hipMemsetAsync(*p, 0, n * sizeof(double), h->stream));
  return IDOCP_OK;
}
int setDev(const idocp_ocp* h) { HIP_TRY(hipSetDevice(h->device)); return IDOCP_OK; }

// (Trotting)ConfigurationSpaceCost reference of stage time t
// (include/idocp/cost/trotting_configuration_space_cost.hpp:126-164)
void qRefAt(const idocp_cost_t& c, int nq, double t, double* q_ref) {
  for (int i = 0; i < nq; ++i) q_ref[i] = c.q_ref[i];
  if (c.use_time_varying_ref) {
    // TimeVaryingConfigurationSpaceCost::set_q_ref (time_varying_configuration_space_cost.hpp:98-109):
    // q_begin (+) tau v_ref with tau = t - t_begin clamped to the window (q_end = q_begin (+) (t_end - t_begin) v_ref)
    const double tau = t <= c.tv_t_begin ? 0.0 : ((t < c.tv_t_end ? t : c.tv_t_end) - c.tv_t_begin);
    if (tau > 0.0) {
      double v6[6] = {c.v_ref[0], c.v_ref[1], c.v_ref[2], c.v_ref[3], c.v_ref[4], c.v_ref[5]}, qb[7];
      idocp_dev::lieIntegrateBase(c.q_ref, v6, tau, qb);
      for (int i = 0; i < 7; ++i) q_ref[i] = qb[i];
      for (int i = 7; i < nq; ++i) q_ref[i] = c.q_ref[i] + tau * c.v_ref[i - 1];
    }
    return;
  }
  if (!c.use_trotting_ref || !(t > c.t_start)) return;
  const double tau = t - c.t_start;
  const int steps = (int)std::floor(tau / c.t_period);
  const double rate = (tau - steps * c.t_period) / c.t_period;
  const double sin2 = std::sin(M_PI_2 * rate);
  q_ref[0] += (steps + rate) * c.step_length;
  if (steps % 2 == 0) {
    q_ref[9] -= sin2 * c.front_swing_knee;  q_ref[12] -= sin2 * c.hip_stance_knee;
    q_ref[15] += sin2 * c.front_stance_knee; q_ref[18] += sin2 * c.hip_swing_knee;
  } else {
    q_ref[9] += sin2 * c.front_stance_knee; q_ref[12] += sin2 * c.hip_swing_knee;
    q_ref[15] -= sin2 * c.front_swing_knee; q_ref[18] -= sin2 * c.hip_stance_knee;
  }
}

// TimeVaryingConfigurationSpaceCost::v_ref(t) (time_varying_configuration_space_cost.hpp:111-118): zero outside the window
double vRefOnAt(const idocp_cost_t& c, double t) {
  if (!c.use_time_varying_ref) return 1.0;
  return (t > c.tv_t_begin && t < c.tv_t_end) ? 1.0 : 0.0;
}

int slotOf(const idocp_ocp* h, int kind, int index) {
  switch (kind) {
    case 1: return h->N + 1 + index;
    case 2: return h->N + 1 + h->E + index;
    case 3: return h->N + 1 + 2 * h->E + index;
    default: return index;
  }
}

void fillStatus(OcpNode& nd, const HostStatus& st) {
  int row = 0;
  for (int c = 0; c < DQ::NC; ++c) {
    nd.active[c] = st.active[c] ? 1 : 0;
    nd.row_of[c] = st.active[c] ? row : -1;
    if (st.active[c]) row += 3;
    for (int k = 0; k < 3; ++k) nd.contact_point[c][k] = st.points[c][k];
  }
  nd.dimf = row;
}

// OCPDiscretizer::discretizeOCP(contact_sequence, t) (ocp_discretizer.hxx:65-374): event times -> time stages, time
// steps of the stages around an event, contact phase of every stage -- and from those the chain of stages in time
// order.  Host-side index logic, re-run only when the initial time or the contact sequence changes.
int discretizeParNMPC(idocp_ocp* h, double t);

// Reference poses of the task-space cost for the M stages of the chain just built: the constant pose of the cost, or -- TimeVarying
// variants -- the poses the caller evaluated at the stage times (idocp_ocp_get_chain_times -> idocp_ocp_set_task_refs).
// The host half of a discretisation as it stood before a discretiser started to rewrite it.  The one recoverable error of a
// discretiser that is only known once the new chain exists -- a TimeVarying task-space cost without reference poses for THIS chain --
// restores it, so that the handle keeps describing the discretisation its device tables (d_nodes, class lists, d_prob) still hold.
struct DiscSnapshot {
  idocp_ocp* h;
  std::vector<OcpNode> chain;
  std::vector<int> chain_index;
  std::vector<double> chain_t;
  OcpProblem prob;
  int Ngrid, uniform_dimf, n_impulse, n_general, BM, BNS, Bnimp, Bnsw;
  bool has_switch, has_terminal, has_prev;
  explicit DiscSnapshot(idocp_ocp* hh)
      : h(hh), chain(hh->chain), chain_index(hh->chain_index), chain_t(hh->chain_t), prob(hh->prob), Ngrid(hh->Ngrid), uniform_dimf(hh->uniform_dimf),
        n_impulse(hh->n_impulse), n_general(hh->n_general), BM(hh->B.M), BNS(hh->B.NS), Bnimp(hh->B.n_impulse_fe), Bnsw(hh->B.n_switch),
        has_switch(hh->has_switch), has_terminal(hh->has_terminal), has_prev(hh->has_prev) {}
  void restore() {
    h->chain.swap(chain); h->chain_index.swap(chain_index); h->chain_t.swap(chain_t);
    h->prob = prob; h->Ngrid = Ngrid; h->uniform_dimf = uniform_dimf; h->n_impulse = n_impulse; h->n_general = n_general;
    h->B.M = BM; h->B.NS = BNS; h->B.n_impulse_fe = Bnimp; h->B.n_switch = Bnsw;
    h->has_switch = has_switch; h->has_terminal = has_terminal; h->has_prev = has_prev;
  }
};

// reference poses of a TimeVarying task-space cost must exist for the chain (t, M) unless the caller only asks for the chain's shape
static int taskRefsAvailable(idocp_ocp* h, double t, int M) {
  if (h->cost.task_dim == 0 || !h->cost.task_time_varying || h->task_refs_lenient) return IDOCP_OK;
  if (h->task_refs_host.size() != (size_t)M * 12 || h->task_refs_t != t) {
    set_last_error("TimeVarying task-space cost: no reference poses for this chain (idocp_ocp_set_task_refs with the same t, M = the chain's length)");
    return IDOCP_E_ARG;
  }
  return IDOCP_OK;
}

static int uploadTaskRefs(idocp_ocp* h, double t, int M) {
  if (h->cost.task_dim == 0) return IDOCP_OK;
  std::vector<double> tab((size_t)M * 12);
  if (h->cost.task_time_varying) {
    if (h->task_refs_host.size() != (size_t)M * 12 || h->task_refs_t != t) {
      if (!h->task_refs_lenient) {                       // (not reached: every discretiser asks taskRefsAvailable before it touches the device)
        set_last_error("TimeVarying task-space cost: no reference poses for this chain (idocp_ocp_set_task_refs with the same t, M = the chain's length)");
        return IDOCP_E_ARG;
      }
      for (int p = 0; p < M; ++p) for (int k = 0; k < 12; ++k) tab[(size_t)p * 12 + k] = h->cost.task_ref[k];
      h->task_refs_stale = true;
    } else {
      tab = h->task_refs_host;
      h->task_refs_stale = false;
    }
  } else {
    for (int p = 0; p < M; ++p) for (int k = 0; k < 12; ++k) tab[(size_t)p * 12 + k] = h->cost.task_ref[k];
  }
  HIP_TRY(hipMemcpyAsync(h->d_taskref, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

int discretize(idocp_ocp* h, double t) {
  if (!h->seq_dirty && h->disc_time == t && !(h->task_refs_stale && !h->task_refs_lenient)) return IDOCP_OK;
  if (h->parnmpc) return discretizeParNMPC(h, t);
  const int N_ideal = h->N;
  const double min_dt = std::sqrt(std::numeric_limits<double>::epsilon());       // ocp_discretizer.hpp:108-109
  const double dt_ideal = h->T / N_ideal, max_dt = dt_ideal - min_dt;
  std::vector<int> ev_imp, ev_lift;
  for (int e = 0; e < (int)h->event_time.size(); ++e) (h->is_impulse[e] ? ev_imp : ev_lift).push_back(e);
  const int Ni = (int)ev_imp.size(), Nl = (int)ev_lift.size();
  std::vector<int> tsbi(Ni + 1, -1), tsbl(Nl + 1, -1);
  std::vector<double> t_imp(Ni + 1, 0.0), t_lift(Nl + 1, 0.0), dt_aux(Ni + 1, 0.0), dt_lift(Nl + 1, 0.0);
  for (int k = 0; k < Ni; ++k) { t_imp[k] = h->event_time[ev_imp[k]]; tsbi[k] = (int)std::floor((t_imp[k] - t) / dt_ideal); }    // countDiscreteEvents
  for (int k = 0; k < Nl; ++k) { t_lift[k] = h->event_time[ev_lift[k]]; tsbl[k] = (int)std::floor((t_lift[k] - t) / dt_ideal); }
  std::vector<double> dts(N_ideal + 1, dt_ideal), ts(N_ideal + 1, 0.0);
  int ii = 0, li = 0, on_grid = 0;
  for (int i = 0; i < N_ideal; ++i) {                                                                                           // countTimeSteps
    const int stage = i - on_grid;
    if (ii < Ni && i == tsbi[ii]) {
      dts[stage] = t_imp[ii] - i * dt_ideal - t;
      if (dts[stage] <= min_dt) { tsbi[ii] = stage - 1; dt_aux[ii] = dt_ideal; ts[stage] = t + (i - 1) * dt_ideal; ++on_grid; ++ii; }
      else if (dts[stage] >= max_dt) { tsbi[ii] = i + 1; ts[stage] = t + i * dt_ideal; }
      else { tsbi[ii] = stage; dt_aux[ii] = dt_ideal - dts[stage]; ts[stage] = t + i * dt_ideal; ++ii; }
    } else if (li < Nl && i == tsbl[li]) {
      dts[stage] = t_lift[li] - i * dt_ideal - t;
      if (dts[stage] <= min_dt) { tsbl[li] = stage - 1; dt_lift[li] = dt_ideal; ts[stage] = t + (i - 1) * dt_ideal; ++on_grid; ++li; }
      else if (dts[stage] >= max_dt) { tsbl[li] = i + 1; ts[stage] = t + i * dt_ideal; }
      else { tsbl[li] = stage; dt_lift[li] = dt_ideal - dts[stage]; ts[stage] = t + i * dt_ideal; ++li; }
    } else {
      dts[stage] = dt_ideal; ts[stage] = t + i * dt_ideal;
    }
  }
  const int Ng = N_ideal - on_grid;
  ts[Ng] = t + h->T;
  std::vector<int> imp_after(Ng + 1, -1), lift_after(Ng + 1, -1), phase(Ng + 1, 0);                                             // countTimeStages / countContactPhase
  ii = 0; li = 0;
  int num_events = 0;
  for (int i = 0; i < Ng; ++i) {
    if (ii < Ni && i == tsbi[ii]) imp_after[i] = ii++;
    if (li < Nl && i == tsbl[li]) lift_after[i] = li++;
    phase[i] = num_events;
    if (imp_after[i] >= 0 && lift_after[i] >= 0) { set_last_error("OCPDiscretizer: an impulse and a lift fall into the same time stage"); return IDOCP_E_ARG; }
    if (imp_after[i] >= 0 || lift_after[i] >= 0) ++num_events;
  }
  phase[Ng] = num_events;
  if (num_events > (int)h->phases.size() - 1) { set_last_error("OCPDiscretizer: inconsistent contact sequence"); return IDOCP_E_ARG; }
  DiscSnapshot before(h);
  h->chain.clear(); h->chain_index.clear(); h->chain_t.clear();
  auto node = [&](int kind, int index, double tt, double dtt, const HostStatus& st, int level) {
    OcpNode nd;
    std::memset(&nd, 0, sizeof(nd));
    nd.kind = kind; nd.slot = slotOf(h, kind, index); nd.level = level;
    nd.has_u = (kind == 1) ? 0 : 1;
    nd.dt = (kind == 1) ? 1.0 : dtt;
    nd.dtq = (kind == 1) ? 0.0 : dtt;
    fillStatus(nd, st);
    h->chain.push_back(nd); h->chain_index.push_back(index); h->chain_t.push_back(tt);
  };
  auto addSwitch = [&](OcpNode& nd, int impulse_index, double dt_next) {              // ocp_linearizer.hxx:152-163, 205-217
    const HostStatus& is = h->impulse_status[ev_imp[impulse_index]];
    int row = 0;
    for (int c = 0; c < DQ::NC; ++c) {
      nd.sw_active[c] = is.active[c] ? 1 : 0;
      nd.sw_row[c] = is.active[c] ? row : -1;
      if (is.active[c]) row += 3;
      for (int k = 0; k < 3; ++k) nd.sw_point[c][k] = is.points[c][k];
    }
    nd.sw_dimi = row;
    nd.sw_dt1 = nd.dtq; nd.sw_dt2 = dt_next;
  };
  h->has_switch = false;
  for (int i = 0; i < Ng; ++i) {
    node(0, i, ts[i], dts[i], h->phases[phase[i]], i);
    if (imp_after[i] < 0 && lift_after[i] < 0 && i + 1 < Ng && imp_after[i + 1] >= 0) { addSwitch(h->chain.back(), imp_after[i + 1], dts[i + 1]); h->has_switch = true; }
    if (imp_after[i] >= 0) {
      const int k = imp_after[i];
      node(1, k, t_imp[k], 0.0, h->impulse_status[ev_imp[k]], -1);
      node(2, k, t_imp[k], dt_aux[k], h->phases[phase[i + 1]], 0);
    } else if (lift_after[i] >= 0) {
      const int k = lift_after[i];
      node(3, k, t_lift[k], dt_lift[k], h->phases[phase[i + 1]], 0);
      if (i + 1 < Ng && imp_after[i + 1] >= 0) { addSwitch(h->chain.back(), imp_after[i + 1], dts[i + 1]); h->has_switch = true; }
    }
  }
  node(4, Ng, ts[Ng], 0.0, h->phases[phase[Ng]], Ng);
  const int M = h->M();
  if (taskRefsAvailable(h, t, M)) { before.restore(); return IDOCP_E_ARG; }
  for (int p = 0; p < M; ++p) {
    h->chain[p].prev = p > 0 ? h->chain[p - 1].slot : -1;
    h->chain[p].next = p + 1 < M ? h->chain[p + 1].slot : -1;
  }
  h->Ngrid = Ng;
  h->uniform_dimf = -1;
  if (h->event_time.empty()) h->uniform_dimf = h->chain[0].dimf;
  // upload: chain, per-stage cost references, problem header
  std::vector<double> tab((size_t)M * DQ::NQ);
  for (int p = 0; p < M; ++p) { qRefAt(h->cost, DQ::NQ, h->chain_t[p], &tab[(size_t)p * DQ::NQ]); h->chain[p].vref_on = vRefOnAt(h->cost, h->chain_t[p]); }
  h->prob.M = M; h->prob.NS = h->NS;
  h->B.M = M; h->B.NS = h->NS;
  HIP_TRY(hipMemcpyAsync(h->d_qref, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  { const int rct = uploadTaskRefs(h, t, M); if (rct) return rct; }
  HIP_TRY(hipMemcpyAsync(h->d_nodes, h->chain.data(), sizeof(OcpNode) * M, hipMemcpyHostToDevice, h->stream));
  // The chain as the line search pairs it (line_search.cpp:80-113): the state-equation residual of a grid stage in front of an
  // impulse / lift is evaluated against the NEXT GRID STAGE (the value computed against the event stage is overwritten there).
  std::vector<OcpNode> chain_ls = h->chain;
  for (int p = 0; p + 1 < M; ++p)
    if (chain_ls[p].kind == 0 && (chain_ls[p + 1].kind == 1 || chain_ls[p + 1].kind == 3)) {
      int pn = p + 2;
      while (pn < M && chain_ls[pn].kind != 0 && chain_ls[pn].kind != 4) ++pn;
      if (pn < M) chain_ls[p].next = chain_ls[pn].slot;
    }
  HIP_TRY(hipMemcpyAsync(h->d_nodes_ls, chain_ls.data(), sizeof(OcpNode) * M, hipMemcpyHostToDevice, h->stream));
  std::vector<int> ipos;
  for (int p = 0; p < M; ++p) if (h->chain[p].kind == 1) ipos.push_back(p);
  h->n_impulse = (int)ipos.size();
  h->B.n_impulse_fe = h->parnmpc ? 0 : h->n_impulse;
  if (!ipos.empty()) HIP_TRY(hipMemcpyAsync(h->d_impulse_pos, ipos.data(), sizeof(int) * ipos.size(), hipMemcpyHostToDevice, h->stream));
  std::vector<int> spos;
  for (int p = 0; p + 1 < M; ++p) if (h->chain[p].sw_dimi > 0) spos.push_back(p);
  h->B.n_switch = (int)spos.size();
  if (!spos.empty()) HIP_TRY(hipMemcpyAsync(h->d_switch_pos, spos.data(), sizeof(int) * spos.size(), hipMemcpyHostToDevice, h->stream));
  // stage classes of K5b (OcpLaunch::condenseMixed)
  std::vector<int> cls[5];
  for (int p = 0; p < M; ++p) {
    const OcpNode& nd = h->chain[p];
    const bool grid = (nd.kind == 0 || nd.kind == 2 || nd.kind == 3), plain = grid && nd.sw_dimi == 0;
    const bool event_half = !h->parnmpc && !plain && (grid || nd.kind == 1) && nd.dimf == DQ::NF / 2;      // an impulse / a switching constraint on half of the feet
    const bool flight = !h->parnmpc && plain && nd.dimf == 0;
    cls[plain && nd.dimf == DQ::NF ? 0 : (plain && nd.dimf == DQ::NF / 2 ? 1 : (event_half ? 3 : (flight ? 4 : 2)))].push_back(p);
  }
  std::vector<int> cpos;
  for (int c = 0; c < 5; ++c) { h->cond_n[c] = (int)cls[c].size(); cpos.insert(cpos.end(), cls[c].begin(), cls[c].end()); }
  HIP_TRY(hipMemcpyAsync(h->d_cond_pos, cpos.data(), sizeof(int) * cpos.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_prob, &h->prob, sizeof(OcpProblem), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));     // tab is a stack temporary
  h->disc_time = t; h->seq_dirty = false; ++h->disc_epoch;
  return IDOCP_OK;
}

// ParNMPCDiscretizer::discretizeOCP with discrete events (parnmpc_discretizer.hxx:65-72: countDiscreteEvents :246-262,
// countTimeSteps :265-324, countTimeStages :327-361, countContactPhase :364-373).  The event stages sit IN FRONT of the grid
// stage that follows the event:  ..., stage i-1, [aux k, impulse k | lift k], stage i, ...; the aux stage carries the
// switching constraint of its impulse (sw_* fields of the node, with sw_dt1 = sw_dt2 = 0: the constraint acts on the aux
// stage's own configuration).
int discretizeParNMPCHybrid(idocp_ocp* h, double t) {
  if (h->stage_offset != 0) { set_last_error("ParNMPC: a horizon with discrete events is sharded by idocp_parnmpc_create_hybrid_shard"); return IDOCP_E_UNSUPPORTED; }
  const int Nid = h->N;
  const double dt_ideal = h->T / Nid, min_dt = std::sqrt(std::numeric_limits<double>::epsilon()), max_dt = dt_ideal - min_dt;
  std::vector<int> ev_imp, ev_lift;
  for (int e = 0; e < (int)h->event_time.size(); ++e) (h->is_impulse[e] ? ev_imp : ev_lift).push_back(e);
  const int Ni = (int)ev_imp.size(), Nl = (int)ev_lift.size();
  std::vector<int> tsai(Ni + 1, -1), tsal(Nl + 1, -1);        // time stage AFTER the impulse / lift
  std::vector<double> t_imp(Ni + 1, 0.0), t_lift(Nl + 1, 0.0), dt_aux(Ni + 1, 0.0), dt_lift(Nl + 1, 0.0);
  for (int k = 0; k < Ni; ++k) { t_imp[k] = h->event_time[ev_imp[k]]; tsai[k] = (int)std::floor((t_imp[k] - t) / dt_ideal); }
  for (int k = 0; k < Nl; ++k) { t_lift[k] = h->event_time[ev_lift[k]]; tsal[k] = (int)std::floor((t_lift[k] - t) / dt_ideal); }
  std::vector<double> dts(Nid + 1, dt_ideal), ts(Nid + 1, 0.0);
  int ii = 0, li = 0, on_grid = 0;
  for (int i = 0; i < Nid; ++i) {
    const int stage = i - on_grid;
    if (ii < Ni && i == tsai[ii]) {
      dts[stage] = (i + 1) * dt_ideal + t - t_imp[ii];
      if (dts[stage] <= min_dt) { tsai[ii] = i + 1; ts[stage] = t + (i + 1) * dt_ideal; }
      else if (dts[stage] >= max_dt) { tsai[ii] = stage - 1; dt_aux[ii] = dt_ideal; ts[stage] = t + i * dt_ideal; ++on_grid; ++ii; }
      else { tsai[ii] = stage; dt_aux[ii] = dt_ideal - dts[stage]; ts[stage] = t + (i + 1) * dt_ideal; ++ii; }
    } else if (li < Nl && i == tsal[li]) {
      dts[stage] = (i + 1) * dt_ideal + t - t_lift[li];
      if (dts[stage] <= min_dt) { tsal[li] = i + 1; ts[stage] = t + (i + 1) * dt_ideal; }
      else if (dts[stage] >= max_dt) { tsal[li] = stage - 1; dt_lift[li] = dt_ideal; ts[stage] = t + i * dt_ideal; ++on_grid; ++li; }
      else { tsal[li] = stage; dt_lift[li] = dt_ideal - dts[stage]; ts[stage] = t + (i + 1) * dt_ideal; ++li; }
    } else {
      dts[stage] = dt_ideal; ts[stage] = t + (i + 1) * dt_ideal;
    }
  }
  const int Ng = Nid - on_grid;
  ts[Ng - 1] = t + h->T;
  std::vector<int> imp_before(Ng, -1), lift_before(Ng, -1), phase(Ng, 0);
  ii = 0; li = 0;
  int num_events = 0;
  for (int i = 0; i < Ng; ++i) {
    if (ii < Ni && i == tsai[ii]) imp_before[i] = ii++;
    if (li < Nl && i == tsal[li]) lift_before[i] = li++;
    if (imp_before[i] >= 0 && lift_before[i] >= 0) { set_last_error("ParNMPCDiscretizer: an impulse and a lift fall into the same time stage"); return IDOCP_E_ARG; }
    if (imp_before[i] >= 0 || lift_before[i] >= 0) ++num_events;
    phase[i] = num_events;
  }
  if (ii != Ni || li != Nl) { set_last_error("ParNMPCDiscretizer: a discrete event lies outside the horizon"); return IDOCP_E_ARG; }
  for (int i = 0; i + 1 < Ng; ++i)
    if (imp_before[i] >= 0 && imp_before[i + 1] >= 0) { set_last_error("ParNMPCDiscretizer: impulses in consecutive time stages"); return IDOCP_E_ARG; }
  // a lift or an impulse in front of the first time stage: the event stages are the first elements of the chain and their
  // predecessor is the measured state (backward_correction_solver.cpp:201-217, 232-246).  The aux stage carries the switching
  // constraint like every other aux stage: the reference's call at :203-211 omits the impulse status and then sizes the KKT
  // inverse with it (split_backward_correction.hxx:46-52), which is not defined as written (oracle/ocp.cpp, same place)
  DiscSnapshot before(h);
  h->chain.clear(); h->chain_index.clear(); h->chain_t.clear();
  auto node = [&](int kind, int index, double tt, double dtt, const HostStatus& st, int level) {
    OcpNode nd;
    std::memset(&nd, 0, sizeof(nd));
    nd.kind = kind; nd.slot = slotOf(h, kind, index); nd.level = level;
    nd.has_u = (kind == 1) ? 0 : 1;
    nd.dt = (kind == 1) ? 1.0 : dtt;
    nd.dtq = (kind == 1) ? 0.0 : dtt;
    fillStatus(nd, st);
    h->chain.push_back(nd); h->chain_index.push_back(index); h->chain_t.push_back(tt);
  };
  h->has_switch = false;
  for (int i = 0; i < Ng; ++i) {
    const int phase_before = i > 0 ? phase[i - 1] : 0;
    if (imp_before[i] >= 0) {
      const int k = imp_before[i];
      const HostStatus& is = h->impulse_status[ev_imp[k]];
      node(2, k, t_imp[k], dt_aux[k], h->phases[phase_before], 0);
      {
        OcpNode& nd = h->chain.back();                              // switchingconstraint::linearizeSwitchingConstraint on the aux stage
        int row = 0;
        for (int c = 0; c < DQ::NC; ++c) {
          nd.sw_active[c] = is.active[c] ? 1 : 0;
          nd.sw_row[c] = is.active[c] ? row : -1;
          if (is.active[c]) row += 3;
          for (int k2 = 0; k2 < 3; ++k2) nd.sw_point[c][k2] = is.points[c][k2];
        }
        nd.sw_dimi = row; nd.sw_dt1 = 0.0; nd.sw_dt2 = 0.0;
        h->has_switch = true;
      }
      node(1, k, t_imp[k], 0.0, is, -1);
    } else if (lift_before[i] >= 0) {
      const int k = lift_before[i];
      node(3, k, t_lift[k], dt_lift[k], h->phases[phase_before], 0);
    }
    node(0, i, ts[i], dts[i], h->phases[phase[i]], (i == Ng - 1) ? Nid : i + 1);      // parnmpc_linearizer.cpp:43-58
  }
  bool is_last_shard = true, is_first_shard = true;
  if (h->slice_end >= 0) {
    // A shard of the chain (idocp_parnmpc_create_hybrid_shard): the grid stages [slice_begin, slice_end) and the event stages in
    // front of each of them; slots and constraint levels stay the global ones.  The placeholder behind the slice is the slot of
    // the right neighbour's first stage, where the imported halos (lmd, gmm, q, aux_mat, corrected lmd, gmm) land.
    const int lo = h->slice_begin, hi = std::min(h->slice_end, Ng);
    std::vector<OcpNode> nodes;
    std::vector<int> idx;
    std::vector<double> tt;
    int next_slot = -1;
    for (size_t p = 0; p < h->chain.size(); ++p) {
      size_t g = p;
      while (h->chain[g].kind != 0) ++g;                            // event stages precede their grid stage
      const int owner = h->chain_index[g];
      if (owner >= lo && owner < hi) { nodes.push_back(h->chain[p]); idx.push_back(h->chain_index[p]); tt.push_back(h->chain_t[p]); }
      else if (owner >= hi && next_slot < 0) next_slot = h->chain[p].slot;
    }
    if (nodes.empty()) { before.restore(); set_last_error("ParNMPC: empty shard of the chain"); return IDOCP_E_ARG; }
    h->chain.swap(nodes); h->chain_index.swap(idx); h->chain_t.swap(tt);
    is_first_shard = lo == 0; is_last_shard = hi >= Ng;
    if (!is_last_shard) {
      OcpNode nd;
      std::memset(&nd, 0, sizeof(nd));
      nd.kind = 4; nd.slot = next_slot; nd.level = Nid; nd.has_u = 1; nd.dt = dt_ideal; nd.dtq = dt_ideal;
      fillStatus(nd, h->phases[0]);
      h->chain.push_back(nd); h->chain_index.push_back(hi); h->chain_t.push_back(t + h->T);
    }
    h->has_switch = false;
    for (const OcpNode& nd : h->chain) if (nd.sw_dimi > 0) h->has_switch = true;
  }
  if (is_last_shard) {
    OcpNode nd;                                                     // placeholder behind the last stage (see below)
    std::memset(&nd, 0, sizeof(nd));
    nd.kind = 4; nd.slot = Nid; nd.level = Nid; nd.has_u = 1; nd.dt = dt_ideal; nd.dtq = dt_ideal;
    fillStatus(nd, h->phases[phase[Ng - 1]]);
    h->chain.push_back(nd); h->chain_index.push_back(Ng); h->chain_t.push_back(t + h->T);
  }
  h->has_terminal = is_last_shard; h->has_prev = !is_first_shard;
  const int M = h->M();
  if (taskRefsAvailable(h, t, M)) { before.restore(); return IDOCP_E_ARG; }
  for (int p = 0; p < M; ++p) {
    h->chain[p].prev = p > 0 ? h->chain[p - 1].slot : -1;
    h->chain[p].next = p + 1 < M ? h->chain[p + 1].slot : -1;
  }
  h->prob.has_terminal = h->has_terminal ? 1 : 0; h->prob.has_prev = h->has_prev ? 1 : 0; h->prob.stage_offset = 0;
  h->Ngrid = Ng - 1;
  h->uniform_dimf = -1;
  std::vector<double> tab((size_t)M * DQ::NQ);
  for (int p = 0; p < M; ++p) { qRefAt(h->cost, DQ::NQ, h->chain_t[p], &tab[(size_t)p * DQ::NQ]); h->chain[p].vref_on = vRefOnAt(h->cost, h->chain_t[p]); }
  h->prob.M = M; h->prob.NS = h->NS;
  h->B.M = M; h->B.NS = h->NS;
  HIP_TRY(hipMemcpyAsync(h->d_qref, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  { const int rct = uploadTaskRefs(h, t, M); if (rct) return rct; }
  HIP_TRY(hipMemcpyAsync(h->d_nodes, h->chain.data(), sizeof(OcpNode) * M, hipMemcpyHostToDevice, h->stream));
  std::vector<int> ipos;
  for (int p = 0; p < M; ++p) if (h->chain[p].kind == 1) ipos.push_back(p);
  h->n_impulse = (int)ipos.size();
  h->B.n_impulse_fe = h->parnmpc ? 0 : h->n_impulse;
  if (!ipos.empty()) HIP_TRY(hipMemcpyAsync(h->d_impulse_pos, ipos.data(), sizeof(int) * ipos.size(), hipMemcpyHostToDevice, h->stream));
  std::vector<int> spos;
  for (int p = 0; p + 1 < M; ++p) if (h->chain[p].sw_dimi > 0) spos.push_back(p);
  h->B.n_switch = (int)spos.size();
  if (!spos.empty()) HIP_TRY(hipMemcpyAsync(h->d_switch_pos, spos.data(), sizeof(int) * spos.size(), hipMemcpyHostToDevice, h->stream));
  std::vector<int> gpos;
  for (int p = 0; p < M; ++p) if (parnmpcShape<LQ>(h->chain[p]).general) gpos.push_back(p);
  h->n_general = (int)gpos.size();
  if (!gpos.empty()) HIP_TRY(hipMemcpyAsync(h->d_general_pos, gpos.data(), sizeof(int) * gpos.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_prob, &h->prob, sizeof(OcpProblem), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->disc_time = t; h->seq_dirty = false;
  return IDOCP_OK;
}

// ParNMPCDiscretizer for a horizon without events (include/idocp/hybrid/parnmpc_discretizer.hxx): N backward-Euler stages,
// stage i at time t + (i + 1) dt with constraint level i + 1 (parnmpc_linearizer.cpp:43-58), followed by a placeholder so
// that the per-stage kernels see the usual "M - 1 stages + one more" chain.
int discretizeParNMPC(idocp_ocp* h, double t) {
  if (!h->event_time.empty()) return discretizeParNMPCHybrid(h, t);
  const int N = h->N;
  const double dt = h->T / N;
  if (taskRefsAvailable(h, t, N + 1)) return IDOCP_E_ARG;          // before anything of the handle is rewritten
  h->chain.clear(); h->chain_index.clear(); h->chain_t.clear();
  for (int i = 0; i <= N; ++i) {
    OcpNode nd;
    std::memset(&nd, 0, sizeof(nd));
    nd.kind = i < N ? 0 : 4; nd.slot = i; nd.level = h->stage_offset + i + 1; nd.has_u = 1; nd.dt = dt; nd.dtq = dt;
    nd.prev = i - 1; nd.next = i < N ? i + 1 : -1;
    fillStatus(nd, h->phases[0]);
    h->chain.push_back(nd); h->chain_index.push_back(i); h->chain_t.push_back(t + (h->stage_offset + i + 1) * dt);
  }
  h->chain_t[N] = t + (h->stage_offset + N) * dt;
  h->prob.has_terminal = h->has_terminal ? 1 : 0; h->prob.has_prev = h->has_prev ? 1 : 0; h->prob.stage_offset = h->stage_offset;
  h->Ngrid = N - 1;                  // getters: stages 0 .. N-1
  h->uniform_dimf = -1; h->has_switch = false; h->n_impulse = 0; h->B.n_impulse_fe = 0; h->n_general = 0; h->B.n_switch = 0;
  const int M = N + 1;
  std::vector<double> tab((size_t)M * DQ::NQ);
  for (int p = 0; p < M; ++p) { qRefAt(h->cost, DQ::NQ, h->chain_t[p], &tab[(size_t)p * DQ::NQ]); h->chain[p].vref_on = vRefOnAt(h->cost, h->chain_t[p]); }
  h->prob.M = M; h->prob.NS = h->NS;
  h->B.M = M; h->B.NS = h->NS;
  HIP_TRY(hipMemcpyAsync(h->d_qref, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
  { const int rct = uploadTaskRefs(h, t, M); if (rct) return rct; }
  HIP_TRY(hipMemcpyAsync(h->d_nodes, h->chain.data(), sizeof(OcpNode) * M, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_prob, &h->prob, sizeof(OcpProblem), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->disc_time = t; h->seq_dirty = false;
  return IDOCP_OK;
}

struct Field { int offset, dim, extra; };
bool solFieldO(const std::string& n, Field& f) {
  if (n == "lmd") f = {LQ::S_LMD, DQ::NV, 1};
  else if (n == "gmm") f = {LQ::S_GMM, DQ::NV, 1};
  else if (n == "q") f = {LQ::S_Q, DQ::NQ, 1};
  else if (n == "v") f = {LQ::S_V, DQ::NV, 1};
  else if (n == "a") f = {LQ::S_A, DQ::NV, 0};
  else if (n == "u") f = {LQ::S_U, DQ::NU, 0};
  else if (n == "beta") f = {LQ::S_BETA, DQ::NV, 0};
  else if (n == "f") f = {LQ::S_F, DQ::NF, 0};
  else if (n == "mu") f = {LQ::S_MU, DQ::NF, 0};
  else if (n == "nu_passive") f = {LQ::S_NUP, 6, 0};
  else if (n == "xi") f = {LQ::S_XI, DQ::NF, 0};
  else return false;
  return true;
}
bool dirFieldO(const std::string& n, Field& f) {
  if (n == "dlmd") f = {LQ::D_LMD, DQ::NV, 1};
  else if (n == "dgmm") f = {LQ::D_GMM, DQ::NV, 1};
  else if (n == "dq") f = {LQ::D_Q, DQ::NV, 1};
  else if (n == "dv") f = {LQ::D_V, DQ::NV, 1};
  else if (n == "da") f = {LQ::D_A, DQ::NV, 0};
  else if (n == "du") f = {LQ::D_U, DQ::NU, 0};
  else if (n == "dbeta") f = {LQ::D_BETA, DQ::NV, 0};
  else if (n == "df") f = {LQ::D_F, DQ::NF, 0};
  else if (n == "dmu") f = {LQ::D_MU, DQ::NF, 0};
  else if (n == "dnu_passive") f = {LQ::D_NUP, 6, 0};
  else if (n == "dxi") f = {LQ::D_XI, DQ::NF, 0};
  else return false;
  return true;
}

int copyField(idocp_ocp* h, const double* base, size_t stride, size_t nrec, const Field& f, double* out) {
  HIP_TRY(hipMemcpy2DAsync(out, f.dim * sizeof(double), base + f.offset, stride * sizeof(double), f.dim * sizeof(double), nrec,
                           hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

}  // namespace

extern "C" {

// the cost-dependent fields of the device problem block (idocp_ocp_create*, idocp_ocp_set_cost)
static void fillCostFields(OcpProblem& p, const idocp_cost_t& cost) {
  for (int i = 0; i < DQ::NV; ++i) {
    p.v_ref[i] = cost.v_ref[i]; p.q_weight[i] = cost.q_weight[i]; p.v_weight[i] = cost.v_weight[i]; p.a_weight[i] = cost.a_weight[i];
    p.qf_weight[i] = cost.qf_weight[i]; p.vf_weight[i] = cost.vf_weight[i];
    p.qi_weight[i] = cost.qi_weight[i]; p.vi_weight[i] = cost.vi_weight[i]; p.dvi_weight[i] = cost.dvi_weight[i];
  }
  if (cost.use_trotting_ref) p.v_ref[0] = cost.step_length / cost.t_period;     // trotting_configuration_space_cost.cpp:81-83
  for (int i = 0; i < DQ::NU; ++i) { p.u_ref[i] = cost.u_ref[i]; p.u_weight[i] = cost.u_weight[i]; }
  for (int c = 0; c < DQ::NC; ++c)
    for (int k = 0; k < 3; ++k) {
      p.f_weight[c][k] = cost.f_weight[c][k]; p.f_ref[c][k] = cost.f_ref[c][k];
      p.fi_weight[c][k] = cost.fi_weight[c][k]; p.fi_ref[c][k] = cost.fi_ref[c][k];
    }
  p.task_dim = cost.task_dim; p.task_joint = cost.task_joint;
  for (int k = 0; k < 9; ++k) p.task_R[k] = cost.task_frame_R[k];
  for (int k = 0; k < 3; ++k) p.task_p[k] = cost.task_frame_p[k];
  for (int k = 0; k < 6; ++k) { p.task_weight[k] = cost.task_weight[k]; p.task_weightf[k] = cost.task_weightf[k]; p.task_weighti[k] = cost.task_weighti[k]; }
  for (int k = 0; k < 12; ++k) p.task_ref[k] = cost.task_ref[k];
  const int nextra = cost.task_dim ? cost.task_extra_count : 0;
  p.task_n = cost.task_dim ? 1 + nextra : 0;
  for (int e = 0; e < IDOCP_MAX_EXTRA_TASKS; ++e) {
    OcpProblem::TaskExtra& o = p.task_extra[e];
    std::memset(&o, 0, sizeof(o));
    if (e >= nextra) continue;
    const idocp_task_component_t& t = cost.task_extra[e];
    o.dim = t.dim; o.joint = t.joint;
    for (int k = 0; k < 9; ++k) o.R[k] = t.frame_R[k];
    for (int k = 0; k < 3; ++k) o.p[k] = t.frame_p[k];
    for (int k = 0; k < 6; ++k) { o.weight[k] = t.weight[k]; o.weightf[k] = t.weightf[k]; o.weighti[k] = t.weighti[k]; }
    for (int k = 0; k < 12; ++k) o.ref[k] = t.ref[k];
  }
}
// the task_extra components of a cost block (idocp_cost_t): count, dimensions and frames
static int checkTaskExtras(const idocp_cost_t* cost) {
  if (cost->task_extra_count < 0 || cost->task_extra_count > IDOCP_MAX_EXTRA_TASKS) { set_last_error("invalid value: task_extra_count must be 0 .. " + std::to_string(IDOCP_MAX_EXTRA_TASKS)); return IDOCP_E_ARG; }
  if (cost->task_extra_count > 0 && cost->task_dim == 0) { set_last_error("invalid value: task_extra components need the first task-space component (task_dim != 0)"); return IDOCP_E_ARG; }
  for (int e = 0; e < cost->task_extra_count; ++e) {
    const idocp_task_component_t& t = cost->task_extra[e];
    if (t.dim != 3 && t.dim != 6) { set_last_error("invalid value: task_extra[" + std::to_string(e) + "].dim must be 3 or 6"); return IDOCP_E_ARG; }
    if (t.joint < 0 || t.joint > DQ::NU) { set_last_error("invalid value: the frame of task_extra[" + std::to_string(e) + "] must sit on the floating base or on a leg link"); return IDOCP_E_ARG; }
  }
  return IDOCP_OK;
}

static int createOcpImpl(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                         int N, int max_num_impulse, int batch, int device, bool parnmpc, idocp_ocp_t** out) {
  if (!model || !cost || !constraints || !out) { set_last_error("idocp_ocp_create: null argument"); return IDOCP_E_ARG; }
  if (!(T > 0)) { set_last_error("invalid value: T must be positive!"); return IDOCP_E_ARG; }       // ocp_solver.cpp:27-44
  if (N <= 0) { set_last_error("invalid value: N must be positive!"); return IDOCP_E_ARG; }
  if (max_num_impulse < 0) { set_last_error("invalid value: max_num_impulse must be non-negative!"); return IDOCP_E_ARG; }
  if (batch <= 0) { set_last_error("invalid value: batch must be positive!"); return IDOCP_E_ARG; }
  // ConstraintComponentBase::setBarrier / setFractionToBoundaryRate (constraint_component_base.hxx:10-24) assert these; a
  // non-positive barrier would make the slack initialisation (pdipm.hxx:13-24) loop forever on the device
  if (!(constraints->barrier > 0)) { set_last_error("invalid value: barrier must be positive!"); return IDOCP_E_ARG; }
  // JointAcceleration*Limit bounds: finite, and a_min < a_max where both are in use (an empty interval has no interior point to start the
  // barrier method from)
  for (int r = 0; r < model->nu; ++r) {
    const bool lo = constraints->joint_acceleration_lower_limit != 0, hi = constraints->joint_acceleration_upper_limit != 0;
    if ((lo && !std::isfinite(constraints->a_min[r])) || (hi && !std::isfinite(constraints->a_max[r]))) { set_last_error("invalid value: joint acceleration bounds must be finite!"); return IDOCP_E_ARG; }
    if (lo && hi && !(constraints->a_min[r] < constraints->a_max[r])) { set_last_error("invalid value: a_min must be smaller than a_max!"); return IDOCP_E_ARG; }
  }
  if ((constraints->linearized_friction_cone && constraints->friction_cone) || (constraints->linearized_impulse_friction_cone && constraints->impulse_friction_cone)) {
    set_last_error("unsupported constraints: LinearizedFrictionCone and FrictionCone (or their impulse twins) together; the stage kernels carry one cone per kind of stage");
    return IDOCP_E_UNSUPPORTED;
  }
  if (!(constraints->fraction_to_boundary_rate > 0 && constraints->fraction_to_boundary_rate <= 1)) {
    set_last_error("invalid value: fraction_to_boundary_rate must be in (0, 1]!"); return IDOCP_E_ARG;
  }
  if (cost->task_dim != 0) {
    if (cost->task_dim != 3 && cost->task_dim != 6) { set_last_error("invalid value: task_dim must be 0, 3 or 6"); return IDOCP_E_ARG; }
    if (cost->task_joint < 0 || cost->task_joint > DQ::NU) { set_last_error("invalid value: the task frame must sit on the floating base or on a leg link"); return IDOCP_E_ARG; }
  }
  { const int rc_t = checkTaskExtras(cost); if (rc_t) return rc_t; }
  if (!isQuadruped(*model)) {
    set_last_error("idocp_ocp_create: this build carries OCP kernels for a floating-base quadruped (4 legs x 3 joints, 4 point contacts) only");
    return IDOCP_E_UNSUPPORTED;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: the idocp HIP path has no CPU fallback");
    return IDOCP_E_DEVICE;
  }
  if (device < 0 || device >= ndev) { set_last_error("invalid device ordinal"); return IDOCP_E_ARG; }
  idocp_ocp* h = new idocp_ocp();
  h->model = *model; h->cost = *cost; h->cons = *constraints; h->N = N; h->E = max_num_impulse; h->batch = batch; h->device = device; h->T = T;
  h->NS = N + 1 + 3 * max_num_impulse;
  auto fail = [&](int code) { (void)hipGetLastError(); idocp_ocp_destroy(h); return code; };      // (clears HIP's sticky last error: see HIP_TRY)
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) { set_last_error("hipStreamCreate failed"); return fail(IDOCP_E_DEVICE); }
  {
    // (a batch of instances only: at batch 1 the two event hand-offs cost more than the 9 wavefronts of K5s -- 1.09 against 1.06 ms per iteration)
    // (read at every creation, not once per process: tests/test_forward_expand_gpu.py runs the oracle parity tests with the side stream forced on small batches)
    const bool use_side = !(getenv("IDOCP_SIDE_STREAM") && atoi(getenv("IDOCP_SIDE_STREAM")) == 0);
    const int side_min_batch = getenv("IDOCP_SIDE_STREAM_MIN_BATCH") ? atoi(getenv("IDOCP_SIDE_STREAM_MIN_BATCH")) : 128;
    if (use_side && !parnmpc && batch >= side_min_batch) {
      if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) != hipSuccess ||
          hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming) != hipSuccess) { set_last_error("side stream: hipStreamCreate / hipEventCreate failed"); return fail(IDOCP_E_DEVICE); }
    }
  }
  const size_t ns = (size_t)batch * h->NS;
  OcpBuffers& B = h->B;
  int rc;
  double* tmp;
  if ((rc = allocBufO(h, &B.sol, ns * LQ::SOL))) return fail(rc);
  if ((rc = allocBufO(h, &B.dir, ns * LQ::DIR))) return fail(rc);
  if ((rc = allocBufO(h, &B.slack, ns * LQ::CON))) return fail(rc);
  if ((rc = allocBufO(h, &B.dual, ns * LQ::CON))) return fail(rc);
  if ((rc = allocBufO(h, &B.lin, ns * LQ::LIN))) return fail(rc);
  if ((rc = allocBufO(h, &B.lie, ns * LQ::LIE))) return fail(rc);
  if ((rc = allocBufO(h, &B.nom, ns * LQ::NOM))) return fail(rc);
  if ((rc = allocBufO(h, &B.kkt, ns * LQ::KKT))) return fail(rc);
  if ((rc = allocBufO(h, &B.exp, ns * LQ::EXP))) return fail(rc);
  if ((rc = allocBufO(h, &B.ric, ns * LQ::RIC))) return fail(rc);
  if ((rc = allocBufO(h, &B.gain, ns * LQ::GAIN))) return fail(rc);
  if ((rc = allocBufO(h, &B.swc, max_num_impulse > 0 ? ns * LQ::SWC : 16))) return fail(rc);
  h->parnmpc = parnmpc;
  if (parnmpc) {
    if ((rc = allocBufO(h, &B.snew, ns * LQ::SNEW))) return fail(rc);
    if ((rc = allocBufO(h, &B.kinv, ns * LQ::KINV))) return fail(rc);
    if ((rc = allocBufO(h, &B.aux, ns * LQ::AUX))) return fail(rc);
    if ((rc = allocBufO(h, &B.xres, ns * LQ::XRES))) return fail(rc);
    if ((rc = allocBufO(h, &B.fwd_prev, (size_t)batch * (DQ::NQ + DQ::NV)))) return fail(rc);
  }
  if ((rc = allocBufO(h, &B.step_stage, ns * 2))) return fail(rc);
  if ((rc = allocBufO(h, &B.step, (size_t)batch * 2))) return fail(rc);
  if ((rc = allocBufO(h, &B.err_stage, ns))) return fail(rc);
  if ((rc = allocBufO(h, &B.err, (size_t)batch))) return fail(rc);
  if ((rc = allocBufO(h, &B.sol_try, ns * LQ::SOL))) return fail(rc);
  B.ext = nullptr; h->ext_try = nullptr;
  if (constraints->contact_distance || cost->task_dim != 0) {      // terms with frame Jacobians of their own (ocp_ext_kernel.hip), of the iterate and of the line search's trial iterate
    if ((rc = allocBufO(h, &B.ext, ns * LQ::EXT))) return fail(rc);
    if ((rc = allocBufO(h, &h->ext_try, ns * LQ::EXT))) return fail(rc);
  }
  if ((rc = allocBufO(h, &B.merit_stage, ns * 4))) return fail(rc);
  if ((rc = allocBufO(h, &B.merit, (size_t)batch * 2))) return fail(rc);
  if ((rc = allocBufO(h, &B.ls_alpha, (size_t)batch))) return fail(rc);
  { double* tmp_ls = nullptr; if ((rc = allocBufO(h, &tmp_ls, ((size_t)h->NS * sizeof(OcpNode) + 7) / 8))) return fail(rc); h->d_nodes_ls = reinterpret_cast<OcpNode*>(tmp_ls); B.nodes_ls = h->d_nodes_ls; }
  h->filters.assign(batch, {});
  if ((rc = allocBufO(h, &h->d_q0, (size_t)batch * DQ::NQ))) return fail(rc);
  if ((rc = allocBufO(h, &h->d_v0, (size_t)batch * DQ::NV))) return fail(rc);
  if (parnmpc) {
    if ((rc = allocBufO(h, &h->d_qtry, (size_t)batch * DQ::NQ))) return fail(rc);
    if ((rc = allocBufO(h, &h->d_vtry, (size_t)batch * DQ::NV))) return fail(rc);
  }
  if ((rc = allocBufO(h, &h->d_tmp, (size_t)batch * IDOCP_MAX_NQ))) return fail(rc);
  if ((rc = allocBufO(h, &h->d_qref, (size_t)h->NS * DQ::NQ))) return fail(rc);
  if ((rc = allocBufO(h, &h->d_taskref, (size_t)h->NS * 12))) return fail(rc);
  if ((rc = allocBufO(h, &tmp, ((size_t)batch * sizeof(int) + 7) / 8))) return fail(rc);
  B.status = reinterpret_cast<int*>(tmp);
  if ((rc = allocBufO(h, &tmp, 64))) return fail(rc);
  B.prof = reinterpret_cast<long long*>(tmp);
  B.prof_dimf = getenv("IDOCP_PROF_DIMF") ? atoi(getenv("IDOCP_PROF_DIMF")) : 12;
  if ((rc = allocBufO(h, &tmp, ((size_t)h->NS * sizeof(OcpNode) + 7) / 8))) return fail(rc);
  h->d_nodes = reinterpret_cast<OcpNode*>(tmp);
  B.nodes = h->d_nodes;
  if ((rc = allocBufO(h, &tmp, ((size_t)(max_num_impulse + 1) * sizeof(int) + 7) / 8))) return fail(rc);
  h->d_impulse_pos = reinterpret_cast<int*>(tmp);
  B.impulse_pos = h->d_impulse_pos;
  if ((rc = allocBufO(h, &tmp, ((size_t)(max_num_impulse + 1) * sizeof(int) + 7) / 8))) return fail(rc);
  h->d_switch_pos = reinterpret_cast<int*>(tmp);
  B.switch_pos = h->d_switch_pos;
  B.n_switch = 0;
  if ((rc = allocBufO(h, &tmp, ((size_t)(2 * max_num_impulse + 1) * sizeof(int) + 7) / 8))) return fail(rc);
  h->d_general_pos = reinterpret_cast<int*>(tmp);
  B.general_pos = h->d_general_pos;
  if ((rc = allocBufO(h, &tmp, ((size_t)h->NS * sizeof(int) + 7) / 8))) return fail(rc);
  h->d_cond_pos = reinterpret_cast<int*>(tmp);
  B.cond_pos = h->d_cond_pos;
  B.q_ref = h->d_qref;
  B.task_refs = h->d_taskref;
  B.leg_axes_xyy = 1;
  for (int leg = 0; leg < DQ::NL; ++leg)
    for (int j = 0; j < DQ::LJ; ++j) {
      const int ji = 1 + leg * DQ::LJ + j;
      const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      const double ax[3] = {j == 0 ? 1.0 : 0.0, j == 0 ? 0.0 : 1.0, 0.0};
      for (int k = 0; k < 9; ++k) if (model->plc_R[ji][k] != I3[k]) B.leg_axes_xyy = 0;
      for (int k = 0; k < 3; ++k) if (model->axis[ji][k] != ax[k]) B.leg_axes_xyy = 0;
    }
  for (int c = 0; c < DQ::NC; ++c) {      // ... and contact frames that are not rotated against their joint
    const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int k = 0; k < 9; ++k) if (model->contact_R[c][k] != I3[k]) B.leg_axes_xyy = 0;
  }
  if (std::getenv("IDOCP_GENERAL_AXES")) B.leg_axes_xyy = 0;      // tests: the general instantiations on a model that qualifies for the special ones
  DevModel dm; toDevModelOcp(*model, dm);
  OcpProblem& p = h->prob;
  std::memset(&p, 0, sizeof(p));
  p.N = N; p.batch = batch; p.T = T; p.dt = T / N; p.NS = h->NS; p.E = max_num_impulse; p.backward_euler = parnmpc ? 1 : 0; p.has_terminal = 1; p.has_prev = 0;
  p.baumgarte_time_step = T / N;                           // hybrid_container.hpp:186-188
  fillCostFields(p, *cost);
  for (int i = 0; i < DQ::NU; ++i) { p.q_min[i] = model->q_min[i]; p.q_max[i] = model->q_max[i]; p.v_max[i] = model->v_max[i]; p.u_max[i] = model->u_max[i]; }
  for (int c = 0; c < DQ::NC; ++c) {
    for (int k = 0; k < 3; ++k) p.contact_p[c][k] = model->contact_p[c][k];
    std::memcpy(p.contact_R[c], model->contact_R[c], sizeof(double) * 9);
  }
  p.use_q_limits = constraints->joint_position_limits; p.use_v_limits = constraints->joint_velocity_limits;
  p.use_u_limits = constraints->joint_torque_limits;
  p.use_a_lower = constraints->joint_acceleration_lower_limit ? 1 : 0;
  p.use_contact_distance = constraints->contact_distance ? 1 : 0;
  p.use_a_upper = constraints->joint_acceleration_upper_limit ? 1 : 0;
  for (int r = 0; r < IDOCP_MAX_NV; ++r) { p.a_min[r] = constraints->a_min[r]; p.a_max[r] = constraints->a_max[r]; }
  p.use_friction_cone = (constraints->linearized_friction_cone || constraints->friction_cone) ? 1 : 0;
  p.use_impulse_friction_cone = (constraints->linearized_impulse_friction_cone || constraints->impulse_friction_cone) ? 1 : 0;
  p.cone_kind = constraints->friction_cone ? 1 : 0; p.impulse_cone_kind = constraints->impulse_friction_cone ? 1 : 0;
  p.mu = constraints->mu; p.barrier = constraints->barrier; p.fraction_rate = constraints->fraction_to_boundary_rate;
  void* d_model = nullptr;
  if (hipMalloc(&d_model, sizeof(DevModel)) != hipSuccess || hipMalloc(&h->d_prob, sizeof(OcpProblem)) != hipSuccess) {
    set_last_error("hipMalloc failed"); return fail(IDOCP_E_DEVICE);
  }
  h->allocs.push_back(d_model); h->alloc_bytes.push_back(sizeof(DevModel));
  h->allocs.push_back(h->d_prob); h->alloc_bytes.push_back(sizeof(OcpProblem));
  if (hipMemcpyAsync(d_model, &dm, sizeof(dm), hipMemcpyHostToDevice, h->stream) != hipSuccess) { set_last_error("hipMemcpy failed"); return fail(IDOCP_E_DEVICE); }
  B.model = static_cast<const DevModel*>(d_model);
  B.prob = static_cast<const OcpProblem*>(h->d_prob);
  h->phases.assign(1, HostStatus());                      // ContactSequence ctor: default (no contact) status
  h->task_refs_lenient = true;
  rc = discretize(h, 0.0);
  h->task_refs_lenient = false;
  if (rc) return fail(rc);
  // identity quaternion in every q so that an unset solution is a valid configuration
  {
    std::vector<double> q(DQ::NQ, 0.0); q[6] = 1.0;
    if (hipMemcpyAsync(h->d_tmp, q.data(), sizeof(double) * DQ::NQ, hipMemcpyHostToDevice, h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
    ocpFillField(B.sol, LQ::SOL, LQ::S_Q, DQ::NQ, h->NS, batch, h->d_tmp, 0, 1, h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  }
  *out = h;
  return IDOCP_OK;
}

int idocp_ocp_create_hybrid(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                            int N, int max_num_impulse, int batch, int device, idocp_ocp_t** out) {
  return createOcpImpl(model, cost, constraints, T, N, max_num_impulse, batch, device, false, out);
}
int idocp_ocp_create(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                     int N, int batch, int device, idocp_ocp_t** out) {
  return createOcpImpl(model, cost, constraints, T, N, 0, batch, device, false, out);
}

extern "C" void idocp_parnmpc_dist_on_destroy(idocp_ocp_t* h);      // parnmpc_dist.hip: drops the handle's communicator attachment, if any
void idocp_ocp_destroy(idocp_ocp_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  idocp_parnmpc_dist_on_destroy(h);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->graph_exec) (void)hipGraphExecDestroy(h->graph_exec);
  if (h->graph) (void)hipGraphDestroy(h->graph);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->side) { (void)hipStreamSynchronize(h->side); (void)hipStreamDestroy(h->side); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  if (h->ev_join) (void)hipEventDestroy(h->ev_join);
  if (h->d_fill) (void)hipFree(h->d_fill);
  if (h->h_fill) (void)hipHostFree(h->h_fill);
  if (h->fill_done) (void)hipEventDestroy(h->fill_done);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static HostStatus makeStatus(const int* active, const double* contact_points) {
  HostStatus st;
  for (int c = 0; c < DQ::NC; ++c) {
    st.active[c] = active[c] ? 1 : 0;
    for (int k = 0; k < 3; ++k) st.points[c][k] = contact_points[3 * c + k];
  }
  return st;
}

int idocp_ocp_set_contact_status_uniformly(idocp_ocp_t* h, const int* active, const double* contact_points) {
  if (!h || !active || !contact_points) return IDOCP_E_ARG;
  h->phases.assign(1, makeStatus(active, contact_points));            // contact_sequence.hxx:47-51
  h->event_time.clear(); h->is_impulse.clear(); h->impulse_status.clear();
  h->contact_status_set = true;
  h->seq_dirty = true;
  return IDOCP_OK;
}

int idocp_ocp_push_back_contact_status(idocp_ocp_t* h, const int* active, const double* contact_points, double switching_time) {
  if (!h || !active || !contact_points) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("Call setContactStatusUniformly() before calling push_back()!"); return IDOCP_E_ARG; }
  // the sequence holds up to N events (ocp_solver.cpp:16: contact_sequence_(robot, N)); the event stages live in
  // max_num_impulse impulse / aux / lift slots each (hybrid_container.hpp:39-96), checked below
  if ((int)h->event_time.size() + 1 > h->N) {
    set_last_error("Number of discrete events=" + std::to_string(h->event_time.size() + 1) + " exceeds predefined max_num_events=" + std::to_string(h->N) + "!");
    return IDOCP_E_ARG;
  }
  if (!h->event_time.empty() && switching_time <= h->event_time.back()) {
    set_last_error("event_time=" + std::to_string(switching_time) + " must be larger than the last event time=" + std::to_string(h->event_time.back()) + "!");
    return IDOCP_E_ARG;
  }
  // DiscreteEvent::setDiscreteEvent (discrete_event.hxx:57-84)
  const HostStatus& pre = h->phases.back();
  const HostStatus post = makeStatus(active, contact_points);
  HostStatus imp = post;
  bool exist_impulse = false, exist_lift = false;
  for (int c = 0; c < DQ::NC; ++c) {
    imp.active[c] = 0;
    if (pre.active[c]) { if (!post.active[c]) exist_lift = true; }
    else if (post.active[c]) { imp.active[c] = 1; exist_impulse = true; }
  }
  if (!exist_impulse && !exist_lift) { set_last_error("discrete_event.existDiscreteEvent() must be true!"); return IDOCP_E_ARG; }
  {
    int n_same = 0;
    for (int e : h->is_impulse) n_same += ((e != 0) == exist_impulse) ? 1 : 0;
    if (n_same + 1 > h->E) {
      set_last_error(std::string("Number of ") + (exist_impulse ? "impulse" : "lift") + " events=" + std::to_string(n_same + 1) + " exceeds max_num_impulse=" + std::to_string(h->E) + "!");
      return IDOCP_E_ARG;
    }
  }
  h->phases.push_back(post);
  h->event_time.push_back(switching_time);
  h->is_impulse.push_back(exist_impulse ? 1 : 0);
  h->impulse_status.push_back(imp);
  h->seq_dirty = true;
  return IDOCP_OK;
}

int idocp_ocp_set_contact_points(idocp_ocp_t* h, int contact_phase, const double* contact_points) {
  if (!h || !contact_points) return IDOCP_E_ARG;
  if (contact_phase < 0 || contact_phase >= (int)h->phases.size()) {
    set_last_error("contact_phase=" + std::to_string(contact_phase) + " must be smaller than numContactPhases()" + std::to_string(h->phases.size()) + "!");
    return IDOCP_E_ARG;
  }
  for (int c = 0; c < DQ::NC; ++c) for (int k = 0; k < 3; ++k) {
    h->phases[contact_phase].points[c][k] = contact_points[3 * c + k];
    if (contact_phase > 0 && h->is_impulse[contact_phase - 1]) h->impulse_status[contact_phase - 1].points[c][k] = contact_points[3 * c + k];
  }
  h->seq_dirty = true;
  return IDOCP_OK;
}

int idocp_ocp_pop_back_contact_status(idocp_ocp_t* h) {              // contact_sequence.hxx:105-125
  if (!h) return IDOCP_E_ARG;
  if (!h->event_time.empty()) {
    h->event_time.pop_back(); h->is_impulse.pop_back(); h->impulse_status.pop_back(); h->phases.pop_back();
  } else {
    h->phases.assign(1, HostStatus());
  }
  h->seq_dirty = true;
  return IDOCP_OK;
}

int idocp_ocp_pop_front_contact_status(idocp_ocp_t* h) {             // contact_sequence.hxx:126-146
  if (!h) return IDOCP_E_ARG;
  if (!h->event_time.empty()) {
    h->event_time.erase(h->event_time.begin()); h->is_impulse.erase(h->is_impulse.begin());
    h->impulse_status.erase(h->impulse_status.begin()); h->phases.erase(h->phases.begin());
  } else {
    h->phases.assign(1, HostStatus());
  }
  h->seq_dirty = true;
  return IDOCP_OK;
}

int idocp_ocp_get_chain(idocp_ocp_t* h, double t, int capacity, int* kind, int* index, int* slot, double* dt, int* dimf, int* sw_dimi) {
  if (!h) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  h->task_refs_lenient = true;
  rc = discretize(h, t);
  h->task_refs_lenient = false;
  if (rc) return rc;
  const int M = h->M();
  if (capacity < M) { set_last_error("idocp_ocp_get_chain: capacity too small"); return IDOCP_E_ARG; }
  for (int p = 0; p < M; ++p) {
    const OcpNode& nd = h->chain[p];
    if (kind) kind[p] = nd.kind;
    if (index) index[p] = h->chain_index[p];
    if (slot) slot[p] = nd.slot;
    if (dt) dt[p] = nd.dtq;
    if (dimf) dimf[p] = nd.kind == 4 ? 0 : nd.dimf;
    if (sw_dimi) sw_dimi[p] = nd.sw_dimi;
  }
  return M;
}

int idocp_ocp_get_chain_times(idocp_ocp_t* h, double t, int capacity, double* times) {
  if (!h || !times) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  // (the chain's shape and times do not depend on the reference poses: a TimeVarying task cost without poses yet must not stop the
  //  discretisation that tells the caller where to evaluate them)
  h->task_refs_lenient = true;
  rc = discretize(h, t);
  h->task_refs_lenient = false;
  if (rc) return rc;
  const int M = h->M();
  if (capacity < M) { set_last_error("idocp_ocp_get_chain_times: capacity too small"); return IDOCP_E_ARG; }
  for (int p = 0; p < M; ++p) times[p] = h->chain_t[p];
  return M;
}
int idocp_ocp_set_task_refs(idocp_ocp_t* h, double t, int M, const double* refs) {
  if (!h || !refs || M <= 0) return IDOCP_E_ARG;
  if (h->cost.task_dim == 0 || !h->cost.task_time_varying) { set_last_error("idocp_ocp_set_task_refs: the solver carries no TimeVarying task-space cost"); return IDOCP_E_ARG; }
  const bool same = h->task_refs_t == t && h->task_refs_host.size() == (size_t)M * 12 &&
                    std::memcmp(h->task_refs_host.data(), refs, sizeof(double) * (size_t)M * 12) == 0;
  if (same && !h->task_refs_stale) return IDOCP_OK;          // the poses the device table already holds
  h->task_refs_host.assign(refs, refs + (size_t)M * 12);
  h->task_refs_t = t;
  if (!h->seq_dirty && h->disc_time == t && M == h->M()) {
    // the chain is the current one (idocp_ocp_get_chain_times has just discretised at t, the facade's order of calls): the table
    // goes straight to the device -- no second discretisation with its host-synchronous uploads in the MPC loop
    int rc = setDev(h); if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(h->d_taskref, h->task_refs_host.data(), sizeof(double) * (size_t)M * 12, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->task_refs_stale = false;
    return IDOCP_OK;
  }
  h->seq_dirty = true;                       // uploaded with the next discretisation at t
  return IDOCP_OK;
}

static int setSolutionO(idocp_ocp_t* h, const char* name, const double* values, int per_instance) {
  if (!h || !name || !values) return IDOCP_E_ARG;
  const std::string n(name);
  Field f;
  if (!(n == "q" || n == "v" || n == "a" || n == "f" || n == "u") || !solFieldO(n, f)) {
    set_last_error("invalid arugment: name must be q, v, a, f, or u!");
    return IDOCP_E_ARG;
  }
  int rc = setDev(h); if (rc) return rc;
  const int dim = (n == "f") ? 3 : f.dim, repeat = (n == "f") ? DQ::NC : 1;
  const size_t cnt = (size_t)(per_instance ? h->batch : 1) * dim;
  HIP_TRY(hipMemcpyAsync(h->d_tmp, values, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  // every stage incl. the event stages (ocp_solver.cpp:95-165; "a" sets dv on impulse stages)
  ocpFillField(h->B.sol, LQ::SOL, f.offset, dim, h->NS, h->batch, h->d_tmp, per_instance, repeat, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_ocp_set_solution(idocp_ocp_t* h, const char* name, const double* value) { return setSolutionO(h, name, value, 0); }

// Warm start of an MPC loop: a field of every grid stage (slots 0 .. nstages - 1), the same for all instances.
static int fillStagesO(idocp_ocp_t* h, double* rec, int stride, int offset, int dim, int nstages, const double* values, bool chain = false) {
  const size_t bytes = (size_t)nstages * dim * sizeof(double);
  if (h->fill_done) HIP_TRY(hipEventSynchronize(h->fill_done));      // the previous setter's copy has left the host twin (normally long ago)
  else HIP_TRY(hipEventCreateWithFlags(&h->fill_done, hipEventDisableTiming));
  if (bytes > h->fill_cap) {
    if (h->d_fill) { HIP_TRY(hipStreamSynchronize(h->stream)); (void)hipFree(h->d_fill); h->d_fill = nullptr; }
    if (h->h_fill) { (void)hipHostFree(h->h_fill); h->h_fill = nullptr; }
    h->fill_cap = 0;
    const size_t cap = bytes + bytes / 2;
    HIP_TRY(hipMalloc((void**)&h->d_fill, cap));
    HIP_TRY(hipHostMalloc((void**)&h->h_fill, cap, hipHostMallocDefault));
    h->fill_cap = cap;
  }
  std::memcpy(h->h_fill, values, bytes);                              // the caller's buffer is free when the call returns
  HIP_TRY(hipMemcpyAsync(h->d_fill, h->h_fill, bytes, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipEventRecord(h->fill_done, h->stream));
  ocpFillStages(rec, stride, offset, dim, h->NS, nstages, h->batch, h->d_fill, h->stream, chain ? h->d_nodes : nullptr);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_ocp_set_solution_stages(idocp_ocp_t* h, const char* name, int nstages, const double* values) {
  if (!h || !name || !values) return IDOCP_E_ARG;
  Field f;
  if (!solFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  const int nmax = h->parnmpc ? h->N : h->N + f.extra;        // grid stages that carry the field (the terminal stage: lmd gmm q v)
  if (nstages <= 0 || nstages > nmax) { set_last_error("idocp_ocp_set_solution_stages: nstages out of range"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  return fillStagesO(h, h->B.sol, LQ::SOL, f.offset, f.dim, nstages, values);
}
int idocp_parnmpc_set_aux_mat(idocp_ocp_t* h, int nstages, const double* values) {
  if (!h || !values || !h->parnmpc) { set_last_error("idocp_parnmpc_set_aux_mat: not a ParNMPC handle"); return IDOCP_E_ARG; }
  if (nstages <= 0 || nstages > h->N) { set_last_error("idocp_parnmpc_set_aux_mat: nstages out of range"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  return fillStagesO(h, h->B.aux, LQ::AUX, 0, DQ::NX * DQ::NX, nstages, values);
}
// warm start along the CHAIN of the current discretisation (event stages included): values[M][dim] in the order of idocp_ocp_get_chain
static int chainLength(idocp_ocp_t* h, int M) {
  if (h->chain.empty()) { set_last_error("no chain yet: discretise first (initConstraints / initBackwardCorrection / updateSolution)"); return IDOCP_E_ARG; }
  if (M <= 0 || M > h->M()) { set_last_error("chain setter: M out of range"); return IDOCP_E_ARG; }
  return IDOCP_OK;
}
int idocp_ocp_set_solution_chain(idocp_ocp_t* h, const char* name, int M, const double* values) {
  if (!h || !name || !values) return IDOCP_E_ARG;
  Field f;
  if (!solFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  if ((rc = chainLength(h, M))) return rc;
  return fillStagesO(h, h->B.sol, LQ::SOL, f.offset, f.dim, M, values, true);
}
int idocp_parnmpc_set_aux_mat_chain(idocp_ocp_t* h, int M, const double* values) {
  if (!h || !values || !h->parnmpc) { set_last_error("idocp_parnmpc_set_aux_mat_chain: not a ParNMPC handle"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  if ((rc = chainLength(h, M))) return rc;
  return fillStagesO(h, h->B.aux, LQ::AUX, 0, DQ::NX * DQ::NX, M, values, true);
}
int idocp_ocp_set_solution_batch(idocp_ocp_t* h, const char* name, const double* values) { return setSolutionO(h, name, values, 1); }

int idocp_ocp_init_constraints(idocp_ocp_t* h, double t) {
  if (!h) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;                 // ocp_solver.cpp:60-64
  OcpLaunch<DQ>::initConstraints(h->B, h->batch, h->NS, h->stream);
  OcpLaunch<DQ>::extInit(h->B, h->batch, h->NS, h->stream);      // ContactDistance rows
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// K5b: one launch on an event-free chain with all feet in contact, else one launch per stage class
static void launchCondenseO(idocp_ocp_t* h, int M, const double* d_q, int part = 0, hipStream_t st_imp = nullptr) {
  if (h->uniform_dimf == DQ::NF || h->cond_n[0] + h->cond_n[1] + h->cond_n[3] + h->cond_n[4] == 0) OcpLaunch<DQ>::condense(h->B, h->batch, M, h->uniform_dimf, d_q, h->stream, part);
  else OcpLaunch<DQ>::condenseMixed(h->B, h->batch, M, h->cond_n, d_q, h->stream, part, st_imp);
}

// The forward sweep that expands as it walks (ocp_forward_expand_kernel) pays when the stage-parallel expansion it absorbs is bandwidth: a
// batch of instances.  A few instances are the opposite case -- K6 spreads their stages over the whole chip while the fused walk strings the
// expansion of all stages of an instance onto one wavefront's chain -- so small batches keep S4 + K6 (latency mode).  IDOCP_FUSED_FORWARD=0 / 1
// forces one or the other; IDOCP_FUSED_FORWARD_MIN_BATCH moves the threshold.
static bool fusedForward(const idocp_ocp_t* h) {
  static const int forced = getenv("IDOCP_FUSED_FORWARD") ? atoi(getenv("IDOCP_FUSED_FORWARD")) : -1;
  static const int min_batch = getenv("IDOCP_FUSED_FORWARD_MIN_BATCH") ? atoi(getenv("IDOCP_FUSED_FORWARD_MIN_BATCH")) : 192;      // measured on configs[2]: S4 + K6 0.315 / 0.405 / 0.587 ms at batch 128 / 256 / 512, the fused walk 0.346 / 0.356 / 0.447
  if (h->M() > OcpForwardExpandMaxChain) return false;             // the walk keeps the chain in LDS
  if (h->fused_forward_mode >= 0) return h->fused_forward_mode != 0;
  if (forced >= 0) return forced != 0;
  return h->batch >= min_batch;
}
// The backward sweep of a handful of instances (latency mode: one reference solver object is batch 1) runs eight wavefronts per instance --
// measured on configs[2]: 0.79 / 0.83 against 0.855 / 0.858 ms at batch 1 / 16, 0.88 against 0.86 at 64.  IDOCP_RICCATI_WIDE_MAX_BATCH moves
// the threshold, IDOCP_RICCATI_NT (ocp_riccati_kernel.hip) overrides everything.
static bool wideSweep(const idocp_ocp_t* h) {
  static const int max_batch = getenv("IDOCP_RICCATI_WIDE_MAX_BATCH") ? atoi(getenv("IDOCP_RICCATI_WIDE_MAX_BATCH")) : 16;
  if (h->riccati_sweep_mode >= 0) return h->riccati_sweep_mode != 0;
  return h->batch <= max_batch;
}
int idocp_ocp_set_riccati_sweep(idocp_ocp_t* h, int mode) {
  if (!h || h->parnmpc || mode < -1 || mode > 1) return IDOCP_E_ARG;
  h->riccati_sweep_mode = mode;
  ++h->disc_epoch;                      // a captured hipGraph holds the other kernel
  return IDOCP_OK;
}
int idocp_ocp_riccati_sweep(idocp_ocp_t* h) { return (h && !h->parnmpc && wideSweep(h)) ? 1 : 0; }
// per handle: mode -1 = by batch size (the default), 0 = S4 + K6 + reduction, 1 = the fused forward sweep
int idocp_ocp_set_fused_forward(idocp_ocp_t* h, int mode) {
  if (!h || h->parnmpc || mode < -1 || mode > 1) return IDOCP_E_ARG;
  h->fused_forward_mode = mode;
  ++h->disc_epoch;                      // a captured hipGraph holds the other kernels
  return IDOCP_OK;
}
// 1 when the forward sweep and the primal expansion of this OCPSolver handle run as one kernel, 0 when they are S4, K6 and the reduction
int idocp_ocp_fused_forward(idocp_ocp_t* h) { return (h && !h->parnmpc && fusedForward(h)) ? 1 : 0; }
static void launchForwardO(idocp_ocp_t* h, int M, const double* d_q, const double* d_v) {
  if (fusedForward(h)) { OcpLaunch<DQ>::forwardExpand(h->B, h->batch, M, d_q, d_v, h->stream); return; }
  OcpLaunch<DQ>::riccatiForward(h->B, h->batch, M, d_q, d_v, h->stream);
  OcpLaunch<DQ>::expandPrimal(h->B, h->batch, M, h->stream);
}

// one SQP iteration on the handle's stream: K5 (+ K5a on impulse stages, K5s), S3, the forward sweep with the primal expansion (S4 + K6), K7
// K5s (and ParNMPC's impulse kernel K5a) in front of the condensation: on the side stream where the handle has one and the chain carries
// switching constraints (fork here, join in joinSideO before the condensation launches), else on the handle's stream
static int launchSwitchO(idocp_ocp_t* h, int M) {
  OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
  if (!h->has_switch) return IDOCP_OK;
  if (!h->side) { OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream); return IDOCP_OK; }
  HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  OcpLaunch<DQ>::switching(h->B, h->batch, M, h->side);
  HIP_TRY(hipEventRecord(h->ev_join, h->side));        // (recorded again by launchNominalO, behind the impulse stages' nominal launch)
  h->side_pending = true;
  return IDOCP_OK;
}
// the nominal sweeps K5n on the handle's stream; with a fork open (launchSwitchO) the small launch over the impulse stages goes beside them too
static int launchNominalO(idocp_ocp_t* h, int M, const double* d_q) {
  launchCondenseO(h, M, d_q, 1, h->side_pending ? h->side : nullptr);
  if (h->side_pending) HIP_TRY(hipEventRecord(h->ev_join, h->side));
  return IDOCP_OK;
}
static int joinSideO(idocp_ocp_t* h) {
  if (h->side_pending) { HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0)); h->side_pending = false; }
  return IDOCP_OK;
}
// K7 and, independent of it, the base poses K7b: beside each other where the handle has a side stream (joined at once: the next iteration
// reads both)
static int launchIntegrateO(idocp_ocp_t* h, int M) {
  if (!h->side) { OcpLaunch<DQ>::expandDualIntegrate(h->B, h->batch, M, h->stream); return IDOCP_OK; }
  HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  OcpLaunch<DQ>::expandDualIntegrate(h->B, h->batch, M, h->stream, h->side);
  HIP_TRY(hipEventRecord(h->ev_join, h->side));
  HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  return IDOCP_OK;
}
// the condensation launches proper: on a chain with several stage classes the largest class on the handle's stream and the others beside it on
// the side stream (their tails blend instead of adding up); joined before the external Hessian / the Riccati sweep
static int launchCondenseClassesO(idocp_ocp_t* h, int M, const double* d_q) {
  int rc;
  if ((rc = joinSideO(h))) return rc;
  const bool mixed = !(h->uniform_dimf == DQ::NF || h->cond_n[0] + h->cond_n[1] + h->cond_n[3] + h->cond_n[4] == 0);
  // OFF by default: measured (round 6, profiles/experiments/r06_side_stream.md) -- two different instantiations of the 57 kB condensation kernel
  // resident together take 3.2 instead of 2.24 ms (configs[2]); IDOCP_SIDE_STREAM_K5=1 repeats the experiment
  static const bool split = getenv("IDOCP_SIDE_STREAM_K5") && atoi(getenv("IDOCP_SIDE_STREAM_K5")) != 0;
  if (!mixed || !h->side || !split || h->cond_n[1] == 0) { launchCondenseO(h, M, d_q, 2); return IDOCP_OK; }
  HIP_TRY(hipEventRecord(h->ev_fork, h->stream));
  HIP_TRY(hipStreamWaitEvent(h->side, h->ev_fork, 0));
  OcpLaunch<DQ>::condenseMixed(h->B, h->batch, M, h->cond_n, d_q, h->stream, 3);
  OcpLaunch<DQ>::condenseMixed(h->B, h->batch, M, h->cond_n, d_q, h->side, 4);
  HIP_TRY(hipEventRecord(h->ev_join, h->side));
  HIP_TRY(hipStreamWaitEvent(h->stream, h->ev_join, 0));
  OcpLaunch<DQ>::condenseMixed(h->B, h->batch, M, h->cond_n, d_q, h->stream, 5);
  return IDOCP_OK;
}
static int launchUpdateO(idocp_ocp_t* h, int M, const double* d_q, const double* d_v) {
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  int rc;
  if ((rc = launchSwitchO(h, M))) return rc;
  if ((rc = launchNominalO(h, M, d_q))) return rc;        // nominal sweeps (+ Lie tasks, external rows): beside K5s
  if ((rc = launchCondenseClassesO(h, M, d_q))) return rc;
  OcpLaunch<DQ>::riccatiBackward(h->B, h->batch, M, h->has_switch, h->stream, wideSweep(h));
  launchForwardO(h, M, d_q, d_v);
  if ((rc = launchIntegrateO(h, M))) return rc;
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}

int idocp_ocp_launch_kernel(idocp_ocp_t* h, int kernel_id, const double* d_q, const double* d_v) {
  if (!h || kernel_id < 0 || kernel_id > 8 || !d_q || !d_v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (h->seq_dirty || h->disc_time != h->disc_time) { if ((rc = discretize(h, h->disc_time == h->disc_time ? h->disc_time : 0.0))) return rc; }
  const int M = h->M();
  switch (kernel_id) {
    case 0: if ((rc = launchSwitchO(h, M))) return rc; break;      // (with a side stream: K5s starts there and id 8 / 1 waits for it)
    case 1: if ((rc = launchNominalO(h, M, d_q))) return rc; if ((rc = launchCondenseClassesO(h, M, d_q))) return rc; break;
    case 7: if ((rc = launchNominalO(h, M, d_q))) return rc; break;      // the two halves of 1: the nominal rigid-body sweeps (+ external rows) ...
    case 8: if ((rc = launchCondenseClassesO(h, M, d_q))) return rc; break;      // ... and the condensation launches proper
    case 2: if ((rc = joinSideO(h))) return rc;      // (a caller that skips the condensation ids still gets K5s in front of the sweep)
            OcpLaunch<DQ>::riccatiBackward(h->B, h->batch, M, h->has_switch, h->stream, wideSweep(h)); break;
    // 3: the forward sweep.  Since round 5 it expands as it walks (S4 + K6 + the step-size reduction in one kernel, ocp_forward_expand_kernel);
    // ids 4 and 5 are then empty.  IDOCP_FUSED_FORWARD=0 restores the three kernels behind ids 3, 4, 5.
    case 3: if (fusedForward(h)) OcpLaunch<DQ>::forwardExpand(h->B, h->batch, M, d_q, d_v, h->stream); else OcpLaunch<DQ>::riccatiForward(h->B, h->batch, M, d_q, d_v, h->stream); break;
    case 4: case 5: if (!fusedForward(h)) OcpLaunch<DQ>::single(kernel_id, h->B, h->batch, M, h->stream); break;
    case 6: if ((rc = launchIntegrateO(h, M))) return rc; break;
    default: OcpLaunch<DQ>::single(kernel_id, h->B, h->batch, M, h->stream); break;
  }
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}

int idocp_ocp_update_solution_device(idocp_ocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("idocp_ocp_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;                 // ocp_.discretize(contact_sequence_, t) (ocp_solver.cpp:72)
  const int M = h->M();
  if ((rc = launchUpdateO(h, M, d_q, d_v))) return rc;
  h->launched_eagerly = true;
  return IDOCP_OK;
}

// The same iteration replayed from a hipGraph: at small batch sizes the nine launches of an iteration are launch-bound (latency
// mode: one OCP instance, 2 ms per iteration), the graph submits them in one call.  Captured on first use and again whenever the
// discretisation (chain length, stage classes, events) or the input buffers change; the host-side discretiser itself stays outside.
int idocp_ocp_update_solution_graph(idocp_ocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("idocp_ocp_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  if (!h->launched_eagerly) return idocp_ocp_update_solution_device(h, t, d_q, d_v);      // first call: plain launches (one-time kernel attributes)
  if ((rc = discretize(h, t))) return rc;
  if (!h->graph_exec || h->graph_epoch != h->disc_epoch || h->graph_q != d_q || h->graph_v != d_v) {
    if (h->graph_exec) { (void)hipGraphExecDestroy(h->graph_exec); h->graph_exec = nullptr; }
    if (h->graph) { (void)hipGraphDestroy(h->graph); h->graph = nullptr; }
    HIP_TRY(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    rc = launchUpdateO(h, h->M(), d_q, d_v);
    const hipError_t e = hipStreamEndCapture(h->stream, &h->graph);
    if (rc) return rc;
    HIP_TRY(e);
    HIP_TRY(hipGraphInstantiate(&h->graph_exec, h->graph, nullptr, nullptr, 0));
    h->graph_epoch = h->disc_epoch; h->graph_q = d_q; h->graph_v = d_v;
  }
  HIP_TRY(hipGraphLaunch(h->graph_exec, h->stream));
  return IDOCP_OK;
}

int idocp_ocp_synchronize(idocp_ocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  if (h->side) HIP_TRY(hipStreamSynchronize(h->side));      // (a fork left open by idocp_ocp_launch_kernel id 0 alone)
  return IDOCP_OK;
}
void* idocp_ocp_stream(idocp_ocp_t* h) { return h ? (void*)h->stream : nullptr; }

// LineSearch::computeCostAndViolation (src/line_search/line_search.cpp:63-196) of s (+) alpha[b] d for every instance: trial
// iterate + barrier cost (ocp_trial_kernel), then the rigid-body residual kernels and the MERIT variant of the condensation
// kernel on a copy of the buffers whose `sol` is the trial iterate, then the sums over the chain.  out[2 b] = cost, [2 b + 1] = violation.
// The filter line search of ParNMPCSolver runs on one shard (horizons with discrete events included: every stage of the chain is
// evaluated against the trial iterate of its chain predecessor, src/line_search/line_search.cpp:199-301, line_search.hpp:224-264).
static int parnmpcLineSearchSupported(const idocp_ocp_t* h) {
  if ((h->has_prev || !h->has_terminal) && !(h->ls_pre && h->ls_post)) {
    set_last_error("line_search=true on a shard of a ParNMPC horizon needs the sharded driver (idocp_parnmpc_dist_update_solution_ls)");
    return IDOCP_E_UNSUPPORTED;
  }
  return IDOCP_OK;
}
static int lineSearchEvalO(idocp_ocp_t* h, const std::vector<double>& alpha, const double* d_q, std::vector<double>& out) {
  const int M = h->M();
  HIP_TRY(hipMemcpyAsync(h->B.ls_alpha, alpha.data(), sizeof(double) * h->batch, hipMemcpyHostToDevice, h->stream));
  OcpLaunch<DQ>::trialIterate(h->B, h->batch, M, h->stream);
  OcpBuffers Bt = h->B;
  Bt.sol = h->B.sol_try;
  Bt.ext = h->ext_try;                              // (the trial iterate's heights / rows; the expansion kernels keep reading the linearisation's)
  if (h->parnmpc) {
    // ParNMPC: backward-Euler stages against the trial predecessor (the measured state in front of the first element of the chain;
    // on a shard with a left neighbour: that neighbour's trial iterate of its last stage, fetched by the driver's hook);
    // aux stages add the l1 norm of their switching constraint (K5s on the trial iterate), impulse stages have a kernel of their own
    const double *pq = d_q, *pv = h->d_v0;
    if (h->ls_pre) {
      const int rcp = h->ls_pre(h);
      if (rcp) return rcp;
      if (h->has_prev) { pq = h->d_qtry; pv = h->d_vtry; }
    }
    OcpLaunch<DQ>::rnea(Bt, h->batch, M, h->n_impulse, h->stream);
    if (h->has_switch) OcpLaunch<DQ>::switching(Bt, h->batch, M, h->stream);
    OcpLaunch<DQ>::meritBackwardEuler(Bt, h->batch, M, pq, pv, h->stream);
    OcpLaunch<DQ>::parnmpcImpulseMerit(Bt, h->batch, h->n_impulse, pq, pv, h->stream);
  } else {
    Bt.nodes = h->B.nodes_ls;
    OcpLaunch<DQ>::rnea(Bt, h->batch, M, h->n_impulse, h->stream);
    if (h->has_switch) OcpLaunch<DQ>::switching(Bt, h->batch, M, h->stream);
    OcpLaunch<DQ>::merit(Bt, h->batch, M, d_q, h->stream);
  }
  OcpLaunch<DQ>::meritReduce(h->B, h->batch, h->stream);
  HIP_TRY(hipGetLastError());
  if (h->parnmpc && h->ls_post) { const int rcp = h->ls_post(h); if (rcp) return rcp; }      // sum over the shards
  out.resize((size_t)h->batch * 2);
  HIP_TRY(hipMemcpyAsync(out.data(), h->B.merit, sizeof(double) * out.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
// LineSearch::computeStepSize (include/idocp/line_search/line_search.hpp:62-92) for every instance of the batch; the filter
// (line_search_filter.cpp:33-63, defaults line_search_filter.hpp:16-17, line_search.hpp:25-26) runs on the host.  On return
// B.step holds the accepted primal step of every instance.
static int runLineSearchO(idocp_ocp_t* h, const double* d_q) {
  const double cost_rate = 0.005, con_rate = 0.005, reduction = 0.75, min_step = 0.05;
  const int B = h->batch;
  auto accepted = [](const std::vector<std::pair<double, double>>& f, double c, double v) {
    for (const auto& p : f) if (c >= p.first && v >= p.second) return false;
    return true;
  };
  auto augment = [&](std::vector<std::pair<double, double>>& f, double c, double v) {
    for (auto it = f.begin(); it != f.end();) { if (c <= it->first && v <= it->second) it = f.erase(it); else ++it; }
    f.push_back({c - cost_rate * v, (1 - con_rate) * v});
  };
  std::vector<double> step((size_t)B * 2), alpha(B, 0.0), cv;
  int rc;
  bool any_empty = false;
  for (int b = 0; b < B; ++b) any_empty = any_empty || h->filters[b].empty();
  if (any_empty) {                                  // "if filter is empty, augment the current solution to the filter"
    if ((rc = lineSearchEvalO(h, alpha, d_q, cv))) return rc;
    for (int b = 0; b < B; ++b) if (h->filters[b].empty()) augment(h->filters[b], cv[2 * b], cv[2 * b + 1]);
  }
  HIP_TRY(hipMemcpyAsync(step.data(), h->B.step, sizeof(double) * step.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<char> done(B, 0);
  int open = 0;
  for (int b = 0; b < B; ++b) { alpha[b] = step[2 * b]; if (!(alpha[b] > min_step)) done[b] = 1; else ++open; }
  while (open > 0) {
    if ((rc = lineSearchEvalO(h, alpha, d_q, cv))) return rc;
    for (int b = 0; b < B; ++b) {
      if (done[b]) continue;
      if (accepted(h->filters[b], cv[2 * b], cv[2 * b + 1])) { augment(h->filters[b], cv[2 * b], cv[2 * b + 1]); done[b] = 1; --open; continue; }
      alpha[b] *= reduction;
      if (!(alpha[b] > min_step)) { done[b] = 1; --open; }
    }
  }
  for (int b = 0; b < B; ++b) step[2 * b] = alpha[b] > min_step ? alpha[b] : min_step;
  HIP_TRY(hipMemcpyAsync(h->B.step, step.data(), sizeof(double) * step.size(), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));         // `step` is a stack temporary
  return IDOCP_OK;
}

// linearise + Riccati + direction + step sizes WITHOUT integrating (kernels 0 .. 5): the state in which
// idocp_ocp_line_search_eval probes trial steps
int idocp_ocp_compute_direction(idocp_ocp_t* h, double t, const double* q, const double* v) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("idocp_ocp_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * DQ::NV, hipMemcpyHostToDevice, h->stream));
  if ((rc = discretize(h, t))) return rc;
  const int M = h->M();
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
  if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);
  launchCondenseO(h, M, h->d_q0);
  OcpLaunch<DQ>::riccatiBackward(h->B, h->batch, M, h->has_switch, h->stream, wideSweep(h));
  launchForwardO(h, M, h->d_q0, h->d_v0);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
// (total cost, total constraint violation) of s (+) alpha[b] d for every instance (alpha = 0: the iterate itself, current slacks)
int idocp_ocp_line_search_eval(idocp_ocp_t* h, const double* alpha, double* cost, double* violation) {
  if (!h || !alpha || !cost || !violation) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (h->parnmpc && (rc = parnmpcLineSearchSupported(h))) return rc;
  std::vector<double> a(alpha, alpha + h->batch), cv;
  if ((rc = lineSearchEvalO(h, a, h->d_q0, cv))) return rc;
  for (int b = 0; b < h->batch; ++b) { cost[b] = cv[2 * b]; violation[b] = cv[2 * b + 1]; }
  return IDOCP_OK;
}
int idocp_ocp_clear_line_search_filter(idocp_ocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  for (auto& f : h->filters) f.clear();
  return IDOCP_OK;
}

int idocp_ocp_update_solution(idocp_ocp_t* h, double t, const double* q, const double* v, int line_search) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * DQ::NV, hipMemcpyHostToDevice, h->stream));
  if (line_search) {
    // OCPSolver::updateSolution(t, q, v, true) (ocp_solver.cpp:67-92): direction, filter line search on the primal step, integration
    if (!h->contact_status_set) { set_last_error("idocp_ocp_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
    if ((rc = discretize(h, t))) return rc;
    const int M = h->M();
    HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
    OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
    if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);
    launchCondenseO(h, M, h->d_q0);
    OcpLaunch<DQ>::riccatiBackward(h->B, h->batch, M, h->has_switch, h->stream, wideSweep(h));
    launchForwardO(h, M, h->d_q0, h->d_v0);
    HIP_TRY(hipGetLastError());
    if ((rc = runLineSearchO(h, h->d_q0))) return rc;
    OcpLaunch<DQ>::expandDualIntegrate(h->B, h->batch, M, h->stream);
    HIP_TRY(hipGetLastError());
  } else
  if ((rc = idocp_ocp_update_solution_device(h, t, h->d_q0, h->d_v0))) return rc;
  std::vector<int> st(h->batch);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.status, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b)
    if (st[b] != 0) { set_last_error("factorisation failed (M, J M^-1 J^T, Quu or the switching-constraint Schur complement not positive definite), instance " + std::to_string(b)); return st[b]; }
  return IDOCP_OK;
}

int idocp_ocp_compute_kkt_residual(idocp_ocp_t* h, double t, const double* q, const double* v) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;
  const int M = h->M();
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
  OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
  if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);
  OcpLaunch<DQ>::residual(h->B, h->batch, M, h->d_q0, h->stream);
  ocpKktErrorReduce(h->B, h->batch, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_ocp_kkt_error(idocp_ocp_t* h, double* kkt_error) {
  if (!h || !kkt_error) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(kkt_error, h->B.err, sizeof(double) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// grid stages 0..N live in slots 0..N, so the stage-wise getters are plain strided copies
int idocp_ocp_get_solution(idocp_ocp_t* h, const char* name, int instance, double* out) {
  if (!h || !name || !out || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  Field f;
  if (!solFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  return copyField(h, h->B.sol + (size_t)instance * h->NS * LQ::SOL, LQ::SOL, h->parnmpc ? h->N : h->Ngrid + f.extra, f, out);
}
// OCPSolver::getSolution(stage) (ocp_solver.hpp:97): the whole split solution of ONE grid stage in one device-to-host copy --
// what an MPC loop reads every cycle (getSolution(0).u).  out: lmd gmm q v a u beta f mu nu_passive, the order of the sol record
// (split_solution.hxx:10-31); on the terminal stage only lmd gmm q v are meaningful.
int idocp_ocp_get_split_solution(idocp_ocp_t* h, int instance, int stage, double* out) {
  if (!h || !out || instance < 0 || instance >= h->batch || stage < 0 || stage > (h->parnmpc ? h->N - 1 : h->Ngrid)) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  static_assert(LQ::S_LMD == 0 && LQ::S_NUP > LQ::S_MU, "record order = output order");
  const size_t n = LQ::S_NUP + 6;
  HIP_TRY(hipMemcpyAsync(out, h->B.sol + ((size_t)instance * h->NS + stage) * LQ::SOL, n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
// The contact status the last discretisation linearises grid stage `stage` with (SplitSolution::setContactStatus,
// split_solution.hxx:41-57: isContactActive(i), dimf): active[ncontacts] flags; returns dimf = 3 * (number of active contacts) or a
// negative error code.  What sizes SplitSolution::f_stack() / mu_stack() of getSolution(stage) (split_solution.hpp:93-122).
int idocp_ocp_get_stage_contact_status(idocp_ocp_t* h, int stage, int* active) {
  if (!h || !active || stage < 0 || stage > (h->parnmpc ? h->N - 1 : h->Ngrid)) return IDOCP_E_ARG;
  for (const OcpNode& nd : h->chain) {
    if ((nd.kind == 0 || nd.kind == 4) && nd.slot == stage) {
      for (int c = 0; c < DQ::NC; ++c) active[c] = nd.kind == 4 ? 0 : nd.active[c];
      return nd.kind == 4 ? 0 : nd.dimf;
    }
  }
  set_last_error("idocp_ocp_get_stage_contact_status: stage " + std::to_string(stage) + " is not a grid stage of the current discretisation");
  return IDOCP_E_ARG;
}
// Direction components that exist only for the ACTIVE contacts of a stage (SplitDirection::df / dmu are daf().tail(dimf) /
// dbetamu().tail(dimf), split_direction.hxx:150-229) or only on a stage that carries a switching constraint (dxi, dimi rows): the record
// keeps whatever an earlier discretisation left in the other slots -- a stage whose contact status has changed since -- and the getters
// report zeros there, like an absent entry (the integration never reads them: K7 steps the active contacts only).
static void maskAbsentDirectionRows(const std::string& name, const OcpNode& nd, double* row) {
  if (name == "df" || name == "dmu") {
    for (int c = 0; c < DQ::NC; ++c) if (!nd.active[c]) for (int k = 0; k < 3; ++k) row[3 * c + k] = 0.0;
  } else if (name == "dxi") {
    for (int k = nd.sw_dimi; k < DQ::NF; ++k) row[k] = 0.0;
  }
}
int idocp_ocp_get_direction(idocp_ocp_t* h, const char* name, int instance, double* out) {
  if (!h || !name || !out || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  Field f;
  if (!dirFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  const size_t nrec = h->parnmpc ? h->N : h->Ngrid + f.extra;
  rc = copyField(h, h->B.dir + (size_t)instance * h->NS * LQ::DIR, LQ::DIR, nrec, f, out);
  if (rc) return rc;
  const std::string n(name);
  if (n == "df" || n == "dmu" || n == "dxi")
    for (const OcpNode& nd : h->chain)
      if ((nd.kind == 0 || nd.kind == 4) && nd.slot >= 0 && (size_t)nd.slot < nrec) maskAbsentDirectionRows(n, nd, out + (size_t)nd.slot * f.dim);
  return IDOCP_OK;
}

// the same fields for every stage of the chain, in chain order: out[M][dim] (rows of stages that do not carry the
// field, e.g. "u" on an impulse stage, hold whatever the slot holds)
static int getChainField(idocp_ocp_t* h, const double* base, size_t stride, const Field& f, int instance, double* out) {
  int rc = setDev(h); if (rc) return rc;
  const int M = h->M();
  std::vector<double> all((size_t)h->NS * stride);
  HIP_TRY(hipMemcpyAsync(all.data(), base + (size_t)instance * h->NS * stride, all.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int p = 0; p < M; ++p) std::memcpy(out + (size_t)p * f.dim, &all[(size_t)h->chain[p].slot * stride + f.offset], sizeof(double) * f.dim);
  return IDOCP_OK;
}
// the correction state along the chain: out[M][nx * nx] (column-major per stage), the counterpart of idocp_parnmpc_set_aux_mat_chain --
// together with idocp_ocp_get_solution_chain it is what carries a converged ParNMPC solver over into another handle (another batch
// size, another shard of the horizon)
int idocp_parnmpc_get_aux_mat_chain(idocp_ocp_t* h, int instance, double* out) {
  if (!h || !out || !h->parnmpc || instance < 0 || instance >= h->batch) { set_last_error("idocp_parnmpc_get_aux_mat_chain: not a ParNMPC handle"); return IDOCP_E_ARG; }
  Field f{0, DQ::NX * DQ::NX, 0};
  return getChainField(h, h->B.aux, LQ::AUX, f, instance, out);
}
// the coarse / corrected iterate s_new of the backward correction along the chain (SplitBackwardCorrection's s_new: lmd gmm u q v; "xi": the switching
// multiplier of an aux stage).  On an impulse stage "u" holds the impulse forces f and "xi" the multipliers mu of the velocity constraint, both packed
// (active contacts first).  out[M][dim], dim = nv (lmd gmm v), nq (q), nu (u), 3 nc (xi).
int idocp_parnmpc_get_new_solution_chain(idocp_ocp_t* h, const char* name, int instance, double* out) {
  if (!h || !name || !out || !h->parnmpc || instance < 0 || instance >= h->batch) { set_last_error("idocp_parnmpc_get_new_solution_chain: not a ParNMPC handle, or a null argument"); return IDOCP_E_ARG; }
  const std::string n(name);
  Field f;
  if (n == "lmd") f = {LQ::N_LMD, DQ::NV, 0};
  else if (n == "gmm") f = {LQ::N_GMM, DQ::NV, 0};
  else if (n == "u") f = {LQ::N_U, DQ::NU, 0};
  else if (n == "q") f = {LQ::N_Q, DQ::NQ, 0};
  else if (n == "v") f = {LQ::N_V, DQ::NV, 0};
  else if (n == "xi") f = {LQ::N_XI, DQ::NF, 0};
  else { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  return getChainField(h, h->B.snew, LQ::SNEW, f, instance, out);
}
int idocp_ocp_get_solution_chain(idocp_ocp_t* h, const char* name, int instance, double* out) {
  if (!h || !name || !out || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  Field f;
  if (!solFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  return getChainField(h, h->B.sol, LQ::SOL, f, instance, out);
}
int idocp_ocp_get_direction_chain(idocp_ocp_t* h, const char* name, int instance, double* out) {
  if (!h || !name || !out || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  Field f;
  if (!dirFieldO(name, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  int rc = getChainField(h, h->B.dir, LQ::DIR, f, instance, out);
  if (rc) return rc;
  const std::string n(name);
  if (n == "df" || n == "dmu" || n == "dxi")
    for (int p = 0; p < h->M(); ++p) maskAbsentDirectionRows(n, h->chain[p], out + (size_t)p * f.dim);
  return IDOCP_OK;
}

int idocp_ocp_get_step_sizes(idocp_ocp_t* h, double* primal, double* dual) {
  if (!h || !primal || !dual) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  std::vector<double> st((size_t)h->batch * 2);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.step, sizeof(double) * st.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b) { primal[b] = st[2 * b]; dual[b] = st[2 * b + 1]; }
  return IDOCP_OK;
}

// chain != 0: one entry per stage of the chain (P, s: M entries; K, k: M - 1); chain == 0: grid stages 0..N
static int getRiccati(idocp_ocp_t* h, int instance, int chain, double* P, double* s, double* K, double* k) {
  if (!h || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  const int nv = DQ::NV, nx = DQ::NX, nu = DQ::NU;
  const int n = chain ? h->M() : h->Ngrid + 1;
  std::vector<double> ric((size_t)h->NS * LQ::RIC), gain((size_t)h->NS * LQ::GAIN);
  HIP_TRY(hipMemcpyAsync(ric.data(), h->B.ric + (size_t)instance * h->NS * LQ::RIC, ric.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(gain.data(), h->B.gain + (size_t)instance * h->NS * LQ::GAIN, gain.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i < n; ++i) {
    const int slot = chain ? h->chain[i].slot : i;
    const double* r = &ric[(size_t)slot * LQ::RIC];
    if (P) {
      double* Pm = P + (size_t)i * nx * nx;
      for (int c = 0; c < nv; ++c) for (int rr = 0; rr < nv; ++rr) {
        Pm[c * nx + rr] = r[LQ::R_PQQ + LQ::psym(rr, c)];          // (packed upper triangle)
        Pm[(nv + c) * nx + rr] = r[LQ::R_PQV + c * nv + rr];
        Pm[c * nx + nv + rr] = r[LQ::R_PQV + rr * nv + c];
        Pm[(nv + c) * nx + nv + rr] = r[LQ::R_PVV + LQ::psym(rr, c)];
      }
    }
    if (s) { std::memcpy(s + (size_t)i * nx, r + LQ::R_SQ, sizeof(double) * nv); std::memcpy(s + (size_t)i * nx + nv, r + LQ::R_SV, sizeof(double) * nv); }
    if (i < n - 1) {
      const double* g = &gain[(size_t)slot * LQ::GAIN];
      if (K) std::memcpy(K + (size_t)i * nu * nx, g + LQ::G_K, sizeof(double) * nu * nx);
      if (k) std::memcpy(k + (size_t)i * nu, g + LQ::G_k, sizeof(double) * nu);
    }
  }
  return IDOCP_OK;
}
int idocp_ocp_get_riccati(idocp_ocp_t* h, int instance, double* P, double* s, double* K, double* k) { return getRiccati(h, instance, 0, P, s, K, k); }
int idocp_ocp_get_riccati_chain(idocp_ocp_t* h, int instance, double* P, double* s, double* K, double* k) { return getRiccati(h, instance, 1, P, s, K, k); }

int idocp_ocp_get_state_feedback_gain(idocp_ocp_t* h, int instance, int stage, double* Kq, double* Kv) {
  if (!h || instance < 0 || instance >= h->batch || stage < 0 || stage >= h->Ngrid || !Kq || !Kv) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  std::vector<double> g(LQ::GAIN);
  HIP_TRY(hipMemcpyAsync(g.data(), h->B.gain + ((size_t)instance * h->NS + stage) * LQ::GAIN, g.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::memcpy(Kq, &g[LQ::G_K], sizeof(double) * DQ::NU * DQ::NV);                      // K.leftCols(nv)
  std::memcpy(Kv, &g[LQ::G_K + DQ::NU * DQ::NV], sizeof(double) * DQ::NU * DQ::NV);   // K.rightCols(nv)
  return IDOCP_OK;
}

// OCPSolver::isCurrentSolutionFeasible (ocp_solver.cpp:216-248) / ParNMPCSolver (parnmpc_solver.cpp:231-273): the primal
// iterate against the inequality constraints, stage by stage along the chain.  A host-side check on the downloaded solution
// records (a few kB per instance): joint limits with the time-step gating of constraints_data.hpp:18-42, the linearised
// (impulse) friction cone on the active contacts.  feasible[b] = 1 / 0; where[b] = chain position of the first offending
// stage in the reference's order (stages, impulses, aux, lifts), -1 if none.
int idocp_ocp_is_current_solution_feasible(idocp_ocp_t* h, int* feasible, int* where) {
  if (!h || !feasible) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (h->seq_dirty || h->disc_time != h->disc_time) { if ((rc = discretize(h, h->disc_time == h->disc_time ? h->disc_time : 0.0))) return rc; }
  const int M = h->M(), nu = DQ::NU;
  const OcpProblem& P = h->prob;
  std::vector<double> sol((size_t)h->batch * h->NS * LQ::SOL);
  HIP_TRY(hipMemcpyAsync(sol.data(), h->B.sol, sol.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  auto stageOk = [&](const OcpNode& nd, const double* s) {
    if (nd.kind == 1) {                                               // linearized_impulse_friction_cone.cpp:82-94
      if (!P.use_impulse_friction_cone) return true;
    } else {
      if (P.use_q_limits && nd.level >= 2)                            // joint_position_{lower,upper}_limit.cpp:38-47
        for (int r = 0; r < nu; ++r) { const double q = s[LQ::S_Q + DQ::NQ - nu + r]; if (q < P.q_min[r] || q > P.q_max[r]) return false; }
      if (P.use_v_limits && nd.level >= 1)
        for (int r = 0; r < nu; ++r) { const double v = s[LQ::S_V + DQ::NV - nu + r]; if (v < -P.v_max[r] || v > P.v_max[r]) return false; }
      if (P.use_u_limits && nd.has_u)
        for (int r = 0; r < nu; ++r) { const double u = s[LQ::S_U + r]; if (u < -P.u_max[r] || u > P.u_max[r]) return false; }
      if (nd.kind != 4)                                               // joint_acceleration_{lower,upper}_limit.cpp:38-47
        for (int r = 0; r < nu; ++r) {
          const double a = s[LQ::S_A + DQ::NV - nu + r];
          if ((P.use_a_lower && a < P.a_min[r]) || (P.use_a_upper && a > P.a_max[r])) return false;
        }
      if (P.use_contact_distance && nd.level >= 2 && nd.kind != 4) {   // contact_distance.cpp:44-55: the frames of the contacts that are not active
        double pts[DQ::NC * 3];
        idocp_model_contact_positions(&h->model, s + LQ::S_Q, pts);
        for (int c = 0; c < DQ::NC; ++c) if (!nd.active[c] && pts[3 * c + 2] <= 0) return false;
      }
      if (!P.use_friction_cone) return true;
    }
    const int ck = nd.kind == 1 ? P.impulse_cone_kind : P.cone_kind;
    for (int c = 0; c < DQ::NC; ++c) {                                // linearized_friction_cone.cpp:87-99, friction_cone.cpp:70-84
      if (!nd.active[c]) continue;
      double J[3];
      for (int r = 0; r < coneRows(ck); ++r) if (coneRow(ck, P.mu, r, s + LQ::S_F + 3 * c, J) > 0) return false;
    }
    return true;
  };
  for (int b = 0; b < h->batch; ++b) {
    int bad = -1;
    for (int kind = 0; kind < 4 && bad < 0; ++kind)
      for (int p = 0; p < M && bad < 0; ++p) {
        const OcpNode& nd = h->chain[p];
        if (nd.kind != kind) continue;
        if (!stageOk(nd, &sol[((size_t)b * h->NS + nd.slot) * LQ::SOL])) bad = p;
      }
    feasible[b] = bad < 0 ? 1 : 0;
    if (where) where[b] = bad;
  }
  return IDOCP_OK;
}

int idocp_ocp_dimc(const idocp_ocp_t* h) {
  if (!h) return 0;
  const idocp_constraints_t& c = h->cons;
  return 2 * DQ::NU * ((c.joint_position_limits ? 1 : 0) + (c.joint_velocity_limits ? 1 : 0) + (c.joint_torque_limits ? 1 : 0)) +
         (c.linearized_friction_cone ? 5 * DQ::NC : 0) + (c.friction_cone ? 2 * DQ::NC : 0) +
         DQ::NU * ((c.joint_acceleration_lower_limit ? 1 : 0) + (c.joint_acceleration_upper_limit ? 1 : 0)) + (c.contact_distance ? DQ::NC : 0);
}

int idocp_ocp_get_constraint_data(idocp_ocp_t* h, int instance, double* slack, double* dual) {
  if (!h || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  const int N = h->Ngrid, dimc = idocp_ocp_dimc(h);
  std::vector<double> sl((size_t)N * LQ::CON), du((size_t)N * LQ::CON);
  HIP_TRY(hipMemcpyAsync(sl.data(), h->B.slack + (size_t)instance * h->NS * LQ::CON, sl.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(du.data(), h->B.dual + (size_t)instance * h->NS * LQ::CON, du.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const idocp_constraints_t& c = h->cons;
  // components in the order 0 .. 5 joint position / velocity / torque (lower, upper), 6 cone, 8 / 9 joint acceleration lower / upper, 10 contact distance
  const int use[11] = {c.joint_position_limits, c.joint_position_limits, c.joint_velocity_limits, c.joint_velocity_limits, c.joint_torque_limits,
                       c.joint_torque_limits, c.linearized_friction_cone || c.friction_cone, 0, c.joint_acceleration_lower_limit, c.joint_acceleration_upper_limit,
                       c.contact_distance};
  const int cr = coneRows(h->prob.cone_kind);        // rows per contact of the cone in use (the records keep five slots per contact)
  for (int i = 0; i < N; ++i) {
    int off = 0;
    for (int comp = 0; comp < 11; ++comp) {
      if (!use[comp]) continue;
      const int n = comp == 10 ? DQ::NC : (comp != 6 ? DQ::NU : cr * DQ::NC);
      const bool valid = (comp < 2 || comp == 10) ? i >= 2 : (comp < 4 ? i >= 1 : true);
      for (int r = 0; r < n; ++r) {
        const int src = comp == 10 ? LQ::C_CD + r : (comp != 6 ? ipmCompRow<LQ>(comp) + r : LQ::C_FRIC + 5 * (r / cr) + r % cr);
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? sl[(size_t)i * LQ::CON + src] : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? du[(size_t)i * LQ::CON + src] : 0.0;
      }
      off += n;
    }
  }
  return IDOCP_OK;
}

int idocp_ocp_get_profile(idocp_ocp_t* h, long long* out, int n) {
  if (!h || !out || n <= 0 || n > 64) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemcpy(out, h->B.prof, sizeof(long long) * n, hipMemcpyDeviceToHost));
  return IDOCP_OK;
}

// The inverse of idocp_ocp_get_lqr_stage, for every instance: the condensed LQR stage the backward sweep reads (its kkt record), so that a test can run
// the sweep alone on a problem whose answer it knows (tests/test_golden_riccati.py: the dense KKT solution of a random LQR problem in the reference's
// block structure).  Column-major blocks; terminal != 0: Qxx and lx only.
int idocp_ocp_set_lqr_stage(idocp_ocp_t* h, int stage, int terminal, const double* Qxx, const double* Qxu, const double* Quu, const double* Fqq6,
                            const double* Fqv6, const double* Fvq, const double* Fvv, const double* Fvu, const double* lx, const double* lu,
                            const double* Fx) {
  if (!h || !Qxx || !lx || stage < 0 || stage > h->Ngrid) return IDOCP_E_ARG;
  if (!terminal && (!Qxu || !Quu || !Fqq6 || !Fqv6 || !Fvq || !Fvv || !Fvu || !lu || !Fx)) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  const int nv = DQ::NV, nx = DQ::NX, nu = DQ::NU;
  std::vector<double> k(LQ::KKT, 0.0);
  for (int c = 0; c < nx; ++c) for (int r = 0; r <= c; ++r) k[LQ::K_QXX + LQ::xsym(r, c)] = Qxx[c * nx + r];
  std::memcpy(&k[LQ::K_LX], lx, sizeof(double) * nx);
  if (!terminal) {
    std::memcpy(&k[LQ::K_QXU], Qxu, sizeof(double) * nx * nu);
    std::memcpy(&k[LQ::K_QUU], Quu, sizeof(double) * nu * nu);
    std::memcpy(&k[LQ::K_FQQ], Fqq6, sizeof(double) * 36);
    std::memcpy(&k[LQ::K_FQV], Fqv6, sizeof(double) * 36);
    std::memcpy(&k[LQ::K_FVQ], Fvq, sizeof(double) * nv * nv);
    std::memcpy(&k[LQ::K_FVV], Fvv, sizeof(double) * nv * nv);
    std::memcpy(&k[LQ::K_FVU], Fvu, sizeof(double) * nv * nu);
    std::memcpy(&k[LQ::K_LU], lu, sizeof(double) * nu);
    std::memcpy(&k[LQ::K_FX], Fx, sizeof(double) * nx);
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (long b = 0; b < h->batch; ++b)
    HIP_TRY(hipMemcpy(h->B.kkt + ((size_t)b * h->NS + stage) * LQ::KKT, k.data(), k.size() * sizeof(double), hipMemcpyHostToDevice));
  return IDOCP_OK;
}

// ContactDynamicsData of a grid stage as the condensation kernel left it (contact_dynamics_data.hxx:8-29: MJtJinv, MJtJinv_dIDCdqv, MJtJinv_IDC), dense and
// column-major with n = nv + dimf rows (the active contacts packed): what the parity tests hold to the INDEPENDENT rigid-body vectors of tests/golden
// (M, J and the derivatives of [ID; C] follow from these three by one inverse).  Returns dimf, or a negative error code.
static int getContactDynamics(idocp_ocp_t* h, int instance, int stage, int dimf, double* MJtJinv, double* MJtJinv_dIDCdqv, double* MJtJinv_IDC);
int idocp_ocp_get_contact_dynamics(idocp_ocp_t* h, int instance, int stage, double* MJtJinv, double* MJtJinv_dIDCdqv, double* MJtJinv_IDC) {
  if (!h || !MJtJinv || !MJtJinv_dIDCdqv || !MJtJinv_IDC || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  int dimf = -1;
  for (const OcpNode& nd : h->chain) if (nd.kind == 0 && nd.slot == stage) dimf = nd.dimf;
  if (dimf < 0) { set_last_error("idocp_ocp_get_contact_dynamics: stage " + std::to_string(stage) + " is not a grid stage of the current discretisation"); return IDOCP_E_ARG; }
  return getContactDynamics(h, instance, stage, dimf, MJtJinv, MJtJinv_dIDCdqv, MJtJinv_IDC);
}
// the same by chain position: any stage that has dynamics -- grid, aux, lift, and IMPULSE stages (ImpulseDynamicsForwardEulerData,
// impulse_dynamics_forward_euler_data.hxx: MJtJinv of the impulse's contacts, the blocks of [ImD; V] in place of [ID; C])
int idocp_ocp_get_contact_dynamics_chain(idocp_ocp_t* h, int instance, int position, double* MJtJinv, double* MJtJinv_dIDCdqv, double* MJtJinv_IDC) {
  if (!h || !MJtJinv || !MJtJinv_dIDCdqv || !MJtJinv_IDC || instance < 0 || instance >= h->batch || position < 0 || position >= h->M()) return IDOCP_E_ARG;
  const OcpNode& nd = h->chain[position];
  if (nd.kind == 4) { set_last_error("idocp_ocp_get_contact_dynamics_chain: the terminal stage has no dynamics"); return IDOCP_E_ARG; }
  return getContactDynamics(h, instance, nd.slot, nd.dimf, MJtJinv, MJtJinv_dIDCdqv, MJtJinv_IDC);
}
static int getContactDynamics(idocp_ocp_t* h, int instance, int stage, int dimf, double* MJtJinv, double* MJtJinv_dIDCdqv, double* MJtJinv_IDC) {
  int rc = setDev(h); if (rc) return rc;
  std::vector<double> e(LQ::EXP);
  HIP_TRY(hipMemcpyAsync(e.data(), h->B.exp + ((size_t)instance * h->NS + stage) * LQ::EXP, e.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int n = DQ::NV + dimf;
  for (int c = 0; c < n; ++c) for (int r = 0; r < n; ++r) MJtJinv[r + n * c] = r >= c ? e[LQ::E_MJ + r * (r + 1) / 2 + c] : e[LQ::E_MJ + c * (c + 1) / 2 + r];      // (packed lower triangle)
  for (int c = 0; c < DQ::NX; ++c) for (int r = 0; r < n; ++r) MJtJinv_dIDCdqv[r + n * c] = e[LQ::E_MJD + r + DQ::NVF * c];
  for (int r = 0; r < n; ++r) MJtJinv_IDC[r] = e[LQ::E_MJIDC + r];
  return dimf;
}

int idocp_ocp_get_lqr_stage(idocp_ocp_t* h, int instance, int stage, double* Qxx, double* Qxu, double* Quu, double* A, double* Bm,
                            double* lx, double* lu, double* Fx) {
  if (!h || instance < 0 || instance >= h->batch || stage < 0 || stage >= h->Ngrid) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  const int nv = DQ::NV, nx = DQ::NX, nu = DQ::NU;
  std::vector<double> k(LQ::KKT);
  HIP_TRY(hipMemcpyAsync(k.data(), h->B.kkt + ((size_t)instance * h->NS + stage) * LQ::KKT, k.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int c = 0; c < nx; ++c) for (int r = 0; r < nx; ++r) Qxx[c * nx + r] = k[LQ::K_QXX + LQ::xsym(r, c)];      // (the record holds the upper triangle)
  std::memcpy(Qxu, &k[LQ::K_QXU], sizeof(double) * nx * nu);
  std::memcpy(Quu, &k[LQ::K_QUU], sizeof(double) * nu * nu);
  std::memset(A, 0, sizeof(double) * nx * nx);
  std::memset(Bm, 0, sizeof(double) * nx * nu);
  double dt = h->T / h->N;
  for (const OcpNode& nd : h->chain) if (nd.slot == stage) dt = nd.dtq;
  for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) {
    double fqq = (r == c) ? 1.0 : 0.0, fqv = (r == c) ? dt : 0.0;
    if (r < 6 && c < 6) { fqq = k[LQ::K_FQQ + r + 6 * c]; fqv = k[LQ::K_FQV + r + 6 * c]; }
    else if (r < 6 || c < 6) { fqq = 0.0; fqv = 0.0; }
    A[r + nx * c] = fqq; A[r + nx * (nv + c)] = fqv;
    A[(nv + r) + nx * c] = k[LQ::K_FVQ + r + nv * c]; A[(nv + r) + nx * (nv + c)] = k[LQ::K_FVV + r + nv * c];
  }
  for (int j = 0; j < nu; ++j) for (int r = 0; r < nv; ++r) Bm[(nv + r) + nx * j] = k[LQ::K_FVU + r + nv * j];
  std::memcpy(lx, &k[LQ::K_LX], sizeof(double) * nx);
  std::memcpy(lu, &k[LQ::K_LU], sizeof(double) * nu);
  std::memcpy(Fx, &k[LQ::K_FX], sizeof(double) * nx);
  return IDOCP_OK;
}

// ---------------------------------------------------------------------------------------------- ParNMPC ----
// idocp::ParNMPCSolver (src/ocp/parnmpc_solver.cpp) for horizons without discrete events.  The handle is an idocp_ocp with
// backward-Euler stages; solution / direction getters, setters and the KKT error go through the idocp_ocp_* entry points
// with stage index 0 .. N-1 (there is no separate terminal stage: stage N-1 carries the terminal cost).
int idocp_parnmpc_create(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                         int N, int batch, int device, idocp_ocp_t** out) {
  return createOcpImpl(model, cost, constraints, T, N, 0, batch, device, true, out);
}

int idocp_parnmpc_create_hybrid(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                                int N, int max_num_impulse, int batch, int device, idocp_ocp_t** out) {
  return createOcpImpl(model, cost, constraints, T, N, max_num_impulse, batch, device, true, out);
}

// ParNMPCSolver::initBackwardCorrection (parnmpc_solver.cpp:66-70)
int idocp_parnmpc_init_backward_correction(idocp_ocp_t* h, double t) {
  if (!h || !h->parnmpc) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;
  OcpLaunch<DQ>::parnmpcPhase(4, h->B, h->batch, h->M(), h->has_terminal, h->d_q0, h->d_v0, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// phases of ParNMPCSolver::updateSolution (parnmpc_solver.cpp:73-103), separately launchable (bench, tests):
// 0 K5a tangent RNEA, 1 K9a backward-Euler condensation, 2 K9b KKT inverse + coarse update, 3 S5 backward serial,
// 4 K10a backward parallel, 5 S6 forward serial, 6 K10b forward parallel + direction, 7 K6 expansion + step sizes,
// 8 step-size reduction, 9 K7 dual expansion + integration
int idocp_parnmpc_launch_phase(idocp_ocp_t* h, int phase, const double* d_q, const double* d_v) {
  if (!h || !h->parnmpc || phase < 0 || phase > 9 || !d_q || !d_v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (h->seq_dirty || h->disc_time != h->disc_time) { if ((rc = discretize(h, h->disc_time == h->disc_time ? h->disc_time : 0.0))) return rc; }
  const int M = h->M();
  switch (phase) {
    case 0: OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream); break;
    case 1:
      if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);          // P, Pq of the aux stages
      OcpLaunch<DQ>::condenseBackwardEuler(h->B, h->batch, M, d_q, d_v, false, h->stream);
      OcpLaunch<DQ>::parnmpcImpulseCondense(h->B, h->batch, h->n_impulse, false, d_q, d_v, h->stream);
      break;
    case 2:
      OcpLaunch<DQ>::parnmpcInverse(h->B, h->batch, M, h->stream);
      OcpLaunch<DQ>::parnmpcEventInverse(h->B, h->batch, h->n_general, h->stream);
      break;
    case 3: case 4: case 5: case 6: OcpLaunch<DQ>::parnmpcPhase(phase - 3, h->B, h->batch, M, h->has_terminal, d_q, d_v, h->stream); break;
    case 7: OcpLaunch<DQ>::single(4, h->B, h->batch, M, h->stream); break;
    case 8: OcpLaunch<DQ>::single(5, h->B, h->batch, M, h->stream); break;
    default: OcpLaunch<DQ>::single(6, h->B, h->batch, M, h->stream); break;
  }
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}

int idocp_parnmpc_update_solution_device(idocp_ocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h || !h->parnmpc || !d_q || !d_v) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("idocp_parnmpc_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  for (int phase = 0; phase <= 9; ++phase) if ((rc = idocp_parnmpc_launch_phase(h, phase, d_q, d_v))) return rc;
  return IDOCP_OK;
}

// coarse update + the four correction sweeps + direction + step sizes WITHOUT integrating (phases 0 .. 8): the state in which
// idocp_ocp_line_search_eval probes trial steps
int idocp_parnmpc_compute_direction(idocp_ocp_t* h, double t, const double* q, const double* v) {
  if (!h || !h->parnmpc || !q || !v) return IDOCP_E_ARG;
  if (!h->contact_status_set) { set_last_error("idocp_parnmpc_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * DQ::NV, hipMemcpyHostToDevice, h->stream));
  if ((rc = discretize(h, t))) return rc;
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  for (int phase = 0; phase <= 8; ++phase) if ((rc = idocp_parnmpc_launch_phase(h, phase, h->d_q0, h->d_v0))) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

int idocp_parnmpc_update_solution(idocp_ocp_t* h, double t, const double* q, const double* v, int line_search) {
  if (!h || !h->parnmpc || !q || !v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (line_search) {
    // ParNMPCSolver::updateSolution(t, q, v, true) (parnmpc_solver.cpp:73-103): direction, filter line search on the primal step, integration
    // (the support check needs the chain of this t; nothing has touched the direction / Riccati records when it refuses)
    if (!h->contact_status_set) { set_last_error("idocp_parnmpc_update_solution: call setContactStatusUniformly first"); return IDOCP_E_ARG; }
    if ((rc = discretize(h, t))) return rc;
    if ((rc = parnmpcLineSearchSupported(h))) return rc;
    if ((rc = idocp_parnmpc_compute_direction(h, t, q, v))) return rc;
    if ((rc = runLineSearchO(h, h->d_q0))) return rc;
    if ((rc = idocp_parnmpc_launch_phase(h, 9, h->d_q0, h->d_v0))) return rc;
  } else {
    HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * DQ::NV, hipMemcpyHostToDevice, h->stream));
    if ((rc = idocp_parnmpc_update_solution_device(h, t, h->d_q0, h->d_v0))) return rc;
  }
  std::vector<int> st(h->batch);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.status, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b)
    if (st[b] != 0) { set_last_error("factorisation failed (a stage matrix is not positive definite), instance " + std::to_string(b)); return st[b]; }
  return IDOCP_OK;
}

int idocp_parnmpc_compute_kkt_residual(idocp_ocp_t* h, double t, const double* q, const double* v) {
  if (!h || !h->parnmpc || !q || !v) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;
  const int M = h->M();
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * DQ::NQ, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * DQ::NV, hipMemcpyHostToDevice, h->stream));
  OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
  if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);
  OcpLaunch<DQ>::condenseBackwardEuler(h->B, h->batch, M, h->d_q0, h->d_v0, true, h->stream);
  OcpLaunch<DQ>::parnmpcImpulseCondense(h->B, h->batch, h->n_impulse, true, h->d_q0, h->d_v0, h->stream);
  ocpKktErrorReduce(h->B, h->batch, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// ---- horizon sharding (idocp_amd/parnmpc_dist.py; SURVEY.md 8e, BASELINE.json configs[3]) ----
// One handle per shard: the stages [stage_offset, stage_offset + N) of a longer horizon (T = N * dt of the shard).
int idocp_parnmpc_create_shard(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                               int N, int stage_offset, int has_terminal, int has_prev, int batch, int device, idocp_ocp_t** out) {
  if (stage_offset < 0) { set_last_error("invalid value: stage_offset must be non-negative!"); return IDOCP_E_ARG; }
  int rc = createOcpImpl(model, cost, constraints, T, N, 0, batch, device, true, out);
  if (rc) return rc;
  (*out)->stage_offset = stage_offset; (*out)->has_terminal = has_terminal != 0; (*out)->has_prev = has_prev != 0;
  (*out)->seq_dirty = true;
  return IDOCP_OK;
}
// A shard of a horizon WITH discrete events: the handle discretises the whole horizon (T, N, the contact sequence pushed on
// every rank alike) and keeps the grid stages [stage_begin, stage_end) of the chain together with the event stages in front
// of each of them; the halos are those of an event-free shard (they only carry q, v, lmd, gmm and aux_mat).
int idocp_parnmpc_create_hybrid_shard(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints, double T,
                                      int N, int max_num_impulse, int stage_begin, int stage_end, int batch, int device, idocp_ocp_t** out) {
  if (stage_begin < 0 || stage_end <= stage_begin || stage_end > N) { set_last_error("invalid value: 0 <= stage_begin < stage_end <= N must hold!"); return IDOCP_E_ARG; }
  int rc = createOcpImpl(model, cost, constraints, T, N, max_num_impulse, batch, device, true, out);
  if (rc) return rc;
  (*out)->slice_begin = stage_begin; (*out)->slice_end = stage_end;
  (*out)->has_terminal = stage_end == N; (*out)->has_prev = stage_begin > 0;
  (*out)->seq_dirty = true;
  return IDOCP_OK;
}
// doubles per instance of a halo: 0 state_last, 1 costate_first, 2 aux_first, 3 bwd_first, 4 fwd_last, 5 aux_all
int idocp_parnmpc_halo_size(int kind) {
  switch (kind) {
    case 0: case 4: return DQ::NQ + DQ::NV;
    case 1: return 2 * DQ::NV + DQ::NQ;
    case 3: return 2 * DQ::NV;
    case 2: case 5: return DQ::NX * DQ::NX;
    default: return -1;
  }
}
// d_buf[batch][halo_size(kind)] in device memory.  Importing kind 0 sets the state in front of the first stage (what the
// *_device entry points otherwise take as d_q, d_v): use idocp_parnmpc_prev_state to get those pointers.
static int haloImpl(idocp_ocp_t* h, int kind, bool do_import, double* d_buf, bool sync) {
  if (!h || !h->parnmpc || !d_buf || idocp_parnmpc_halo_size(kind) < 0) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if (h->seq_dirty || h->disc_time != h->disc_time) { if ((rc = discretize(h, h->disc_time == h->disc_time ? h->disc_time : 0.0))) return rc; }
  OcpLaunch<DQ>::parnmpcHalo(h->B, h->batch, kind, do_import, d_buf, h->d_q0, h->d_v0, h->stream);
  HIP_TRY(hipGetLastError());
  if (sync) HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_parnmpc_export_halo(idocp_ocp_t* h, int kind, double* d_buf) { return haloImpl(h, kind, false, d_buf, true); }
int idocp_parnmpc_import_halo(idocp_ocp_t* h, int kind, const double* d_buf) { return haloImpl(h, kind, true, const_cast<double*>(d_buf), true); }
// the same pack / unpack kernels ENQUEUED on the handle's stream without waiting for them: what a stream-ordered transport needs
// (idocp_amd/csrc/parnmpc_dist.hip: pack -> ncclSend, ncclRecv -> unpack, all on this stream)
int idocp_parnmpc_export_halo_async(idocp_ocp_t* h, int kind, double* d_buf) { return haloImpl(h, kind, false, d_buf, false); }
int idocp_parnmpc_import_halo_async(idocp_ocp_t* h, int kind, const double* d_buf) { return haloImpl(h, kind, true, const_cast<double*>(d_buf), false); }
// device pointers of the state in front of the first stage, q[batch][nq], v[batch][nv] (rank 0 uploads the measured state
// there; the other shards receive it through import_halo(0))
int idocp_parnmpc_prev_state(idocp_ocp_t* h, double** d_q, double** d_v) {
  if (!h || !h->parnmpc || !d_q || !d_v) return IDOCP_E_ARG;
  *d_q = h->d_q0; *d_v = h->d_v0;
  return IDOCP_OK;
}
// ---- filter line search on a sharded horizon: what the driver (parnmpc_dist.hip) needs from the handle ----
int idocp_parnmpc_set_line_search_hooks(idocp_ocp_t* h, int (*pre)(idocp_ocp_t*), int (*post)(idocp_ocp_t*)) {
  if (!h || !h->parnmpc) return IDOCP_E_ARG;
  h->ls_pre = pre; h->ls_post = post;
  return IDOCP_OK;
}
// the state_last halo of the TRIAL iterate: export (q, v) of the shard's last stage from sol_try; import into the trial state in front of
// the shard's first stage (d_buf[batch][idocp_parnmpc_halo_size(0)], enqueued on the handle's stream)
int idocp_parnmpc_trial_halo_async(idocp_ocp_t* h, int do_import, double* d_buf) {
  if (!h || !h->parnmpc || !d_buf) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  OcpBuffers Bt = h->B;
  if (!do_import) Bt.sol = h->B.sol_try;
  OcpLaunch<DQ>::parnmpcHalo(Bt, h->batch, 0, do_import != 0, d_buf, h->d_qtry, h->d_vtry, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
// cost and violation sums of the last probe, [batch][2] in device memory (what the driver all-reduces)
int idocp_parnmpc_merit_device(idocp_ocp_t* h, double** d_merit) {
  if (!h || !d_merit) return IDOCP_E_ARG;
  *d_merit = h->B.merit;
  return IDOCP_OK;
}
// LineSearch::computeStepSize on the direction of phases 0 .. 8 (with the hooks above: collectively on every shard); B.step then holds the
// accepted primal steps
int idocp_parnmpc_line_search(idocp_ocp_t* h) {
  if (!h || !h->parnmpc) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = parnmpcLineSearchSupported(h))) return rc;
  return runLineSearchO(h, h->d_q0);
}
// step sizes [batch][2] (primal, dual) in device memory: read after phase 8, overwrite with the global minimum before phase 9
int idocp_parnmpc_step_sizes_device(idocp_ocp_t* h, double** d_steps) {
  if (!h || !d_steps) return IDOCP_E_ARG;
  *d_steps = h->B.step;
  return IDOCP_OK;
}
// discretise at time t (uploads the stage references) without running anything: call before the first phase of an iteration
int idocp_parnmpc_discretize(idocp_ocp_t* h, double t) {
  if (!h || !h->parnmpc) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  return discretize(h, t);
}

// KKT residual with the state in front of the first stage already in device memory (sharded horizons); the squared
// error of the shard's stages is then d_err2[batch]
int idocp_parnmpc_kkt_error_squared_device(idocp_ocp_t* h, double t, double* d_err2) {
  if (!h || !h->parnmpc || !d_err2) return IDOCP_E_ARG;
  int rc = setDev(h); if (rc) return rc;
  if ((rc = discretize(h, t))) return rc;
  const int M = h->M();
  OcpLaunch<DQ>::rnea(h->B, h->batch, M, h->n_impulse, h->stream);
  if (h->has_switch) OcpLaunch<DQ>::switching(h->B, h->batch, M, h->stream);
  OcpLaunch<DQ>::condenseBackwardEuler(h->B, h->batch, M, h->d_q0, h->d_v0, true, h->stream);
  OcpLaunch<DQ>::parnmpcImpulseCondense(h->B, h->batch, h->n_impulse, true, h->d_q0, h->d_v0, h->stream);
  ocpKktErrorReduce(h->B, h->batch, h->stream, d_err2);      // stays on the device and on the stream
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_ocp_batch(idocp_ocp_t* h) { return h ? h->batch : 0; }
// Storage precision of the Riccati factorisation P, s (64 = default; 32 = rounded to single precision after every stage of the
// backward sweep, the device side of BASELINE configs[4]'s tolerance study).  Everything else stays FP64.
int idocp_ocp_set_riccati_storage(idocp_ocp_t* h, int bits) {
  if (!h || (bits != 32 && bits != 64)) return IDOCP_E_ARG;
  if (h->parnmpc) { set_last_error("idocp_ocp_set_riccati_storage: the Riccati sweep belongs to OCPSolver"); return IDOCP_E_UNSUPPORTED; }
  int rc = setDev(h); if (rc) return rc;
  h->prob.ric_fp32 = bits == 32 ? 1 : 0;
  HIP_TRY(hipMemcpyAsync(h->d_prob, &h->prob, sizeof(OcpProblem), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_ocp_state_dims(idocp_ocp_t* h, int* nq, int* nv) {
  if (!h || !nq || !nv) return IDOCP_E_ARG;
  *nq = DQ::NQ; *nv = DQ::NV;
  return IDOCP_OK;
}

// Robot::integrateConfiguration / subtractConfiguration / normalizeConfiguration (include/idocp/robot/robot.hxx:96-147) on the HOST, joint by joint:
// what an MPC loop does between two solver calls -- advance the plant by one sampling period, measure the distance to a goal, keep the base
// quaternion on the unit sphere.  Either side of the hot path, not on it (the kernels carry their own copies: dev_lie.hpp, the same functions).
int idocp_model_integrate_configuration(const idocp_model_t* m, const double* q, const double* v, double length, double* q_out) {
  if (!m || !q || !v || !q_out) { set_last_error("idocp_model_integrate_configuration: null argument"); return IDOCP_E_ARG; }
  for (int i = 0; i < m->njoints; ++i) {
    const int iq = m->idx_q[i], iv = m->idx_v[i];
    if (m->jtype[i] == IDOCP_JOINT_FREEFLYER) idocp_dev::lieIntegrateBase(q + iq, v + iv, length, q_out + iq);      // pinocchio::integrate on SE(3)
    else q_out[iq] = q[iq] + length * v[iv];
  }
  return IDOCP_OK;
}
// diff[nv] = q_plus (-) q_minus: log6(M_minus^-1 M_plus) on the free-flyer (pinocchio::difference(q_minus, q_plus)), q_plus - q_minus on a revolute joint
int idocp_model_subtract_configuration(const idocp_model_t* m, const double* q_plus, const double* q_minus, double* diff) {
  if (!m || !q_plus || !q_minus || !diff) { set_last_error("idocp_model_subtract_configuration: null argument"); return IDOCP_E_ARG; }
  for (int i = 0; i < m->njoints; ++i) {
    const int iq = m->idx_q[i], iv = m->idx_v[i];
    if (m->jtype[i] == IDOCP_JOINT_FREEFLYER) {
      double R[9], p[3];
      idocp_dev::lieRelative(q_minus + iq, q_plus + iq, R, p);
      idocp_dev::lieLog6(R, p, diff + iv);
    } else {
      diff[iv] = q_plus[iq] - q_minus[iq];
    }
  }
  return IDOCP_OK;
}
int idocp_model_normalize_configuration(const idocp_model_t* m, double* q) {
  if (!m || !q) { set_last_error("idocp_model_normalize_configuration: null argument"); return IDOCP_E_ARG; }
  for (int i = 0; i < m->njoints; ++i) {
    if (m->jtype[i] != IDOCP_JOINT_FREEFLYER) continue;
    double* qu = q + m->idx_q[i] + 3;
    const double n = std::sqrt(qu[0] * qu[0] + qu[1] * qu[1] + qu[2] * qu[2] + qu[3] * qu[3]);
    if (!(n > 0)) { set_last_error("idocp_model_normalize_configuration: zero quaternion"); return IDOCP_E_ARG; }
    for (int c = 0; c < 4; ++c) qu[c] /= n;
  }
  return IDOCP_OK;
}

// Deep copy (the reference's solver classes are copyable, `= default`: ocp_solver.hpp:171-186): a new handle of the same
// configuration whose device records, contact sequence and discretisation state equal the source's.
// The reference's solvers hold a shared_ptr to the CostFunction (ocp_solver.hpp:37-39): weights and references a driver changes
// between two updateSolution calls -- an MPC loop moving its goal -- take effect at the next call.  Here the cost was copied at
// creation; this is the way to change it afterwards.  The KIND of cost must stay what it was (a task-space term cannot appear or
// disappear: its record is allocated at creation).
int idocp_ocp_set_cost(idocp_ocp_t* h, const idocp_cost_t* cost) {
  if (!h || !cost) return IDOCP_E_ARG;
  if ((cost->task_dim != 0) != (h->cost.task_dim != 0) || cost->task_time_varying != h->cost.task_time_varying) {
    set_last_error("idocp_ocp_set_cost: a task-space cost cannot be added or removed after creation");
    return IDOCP_E_UNSUPPORTED;
  }
  if (cost->task_dim != 0 && (cost->task_dim != 3 && cost->task_dim != 6)) { set_last_error("invalid value: task_dim must be 0, 3 or 6"); return IDOCP_E_ARG; }
  { const int rc_t = checkTaskExtras(cost); if (rc_t) return rc_t; }
  h->cost = *cost;
  fillCostFields(h->prob, *cost);
  h->seq_dirty = true;              // the problem block and the per-stage reference table are uploaded with the next discretisation
  return IDOCP_OK;
}
int idocp_ocp_clone(idocp_ocp_t* src, idocp_ocp_t** out) {
  if (!src || !out) return IDOCP_E_ARG;
  int rc = setDev(src); if (rc) return rc;
  idocp_ocp_t* h = nullptr;
  if ((rc = createOcpImpl(&src->model, &src->cost, &src->cons, src->T, src->N, src->E, src->batch, src->device, src->parnmpc, &h))) return rc;
  auto fail = [&](int code) { (void)hipGetLastError(); idocp_ocp_destroy(h); return code; };      // (clears HIP's sticky last error: see HIP_TRY)
  if (h->allocs.size() != src->allocs.size()) { set_last_error("idocp_ocp_clone: allocation tables differ"); return fail(IDOCP_E_DEVICE); }
  if (hipStreamSynchronize(src->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  for (size_t i = 0; i < h->allocs.size(); ++i) {
    if (h->alloc_bytes[i] != src->alloc_bytes[i]) { set_last_error("idocp_ocp_clone: allocation tables differ"); return fail(IDOCP_E_DEVICE); }
    if (hipMemcpyAsync(h->allocs[i], src->allocs[i], h->alloc_bytes[i], hipMemcpyDeviceToDevice, h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  }
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  h->task_refs_host = src->task_refs_host; h->task_refs_t = src->task_refs_t;
  h->contact_status_set = src->contact_status_set; h->stage_offset = src->stage_offset; h->has_terminal = src->has_terminal; h->has_prev = src->has_prev;
  h->phases = src->phases; h->event_time = src->event_time; h->is_impulse = src->is_impulse; h->impulse_status = src->impulse_status;
  h->slice_begin = src->slice_begin; h->slice_end = src->slice_end;
  h->fused_forward_mode = src->fused_forward_mode;
  h->riccati_sweep_mode = src->riccati_sweep_mode;
  h->filters = src->filters;                 // the line-search filter is part of the solver's state (LineSearch is a member of the reference's solvers)
  h->prob = src->prob;
  h->seq_dirty = true;                       // the chain is rebuilt (and uploaded) on first use
  if (src->disc_time == src->disc_time) {
    h->task_refs_lenient = true;
    rc = discretize(h, src->disc_time);
    h->task_refs_lenient = false;
    if (rc) return fail(rc);
  }
  *out = h;
  return IDOCP_OK;
}
extern "C" void idocp_set_last_error_string(const char* msg) { set_last_error(msg ? msg : ""); }
int idocp_device_copy(void* d_dst, const void* d_src, unsigned long nbytes) {
  if (!d_dst || !d_src) return IDOCP_E_ARG;
  HIP_TRY(hipMemcpy(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice));
  return IDOCP_OK;
}

}  // extern "C"
