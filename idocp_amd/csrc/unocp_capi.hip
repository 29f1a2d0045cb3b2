// C-ABI implementation of the UnOCPSolver path (include/idocp_hip.h).
//
// Host side of the boundary: owns the device buffers and the HIP stream of a
// handle, converts the flat model/cost/constraint structs into the device
// parameter blocks and sequences the kernels exactly like
// UnOCPSolver::updateSolution does (src/unocp/unocp_solver.cpp:73-134):
//   linearize (K1) -> backward/forward Riccati (S1,S2) -> expand + step sizes
//   (K2) -> integrate (K3).
// There is NO CPU fallback: without a GPU every entry point that needs the
// device returns IDOCP_E_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cmath>
#include <cstring>
#include <string>
#include <utility>
#include <vector>

#include "host_util.hpp"
#include "idocp_hip.h"
#include "unocp_launch.hpp"

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

void toDevModel(const idocp_model_t& m, DevModel& d) {
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
    // Io = Ic + m (c.c 1 - c c^T)   (inertia about the joint-frame origin)
    const double cc = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
    d.Io[i][0] = I[0] + ms * (cc - c[0] * c[0]);
    d.Io[i][1] = I[1] - ms * c[0] * c[1];
    d.Io[i][2] = I[2] - ms * c[0] * c[2];
    d.Io[i][3] = I[4] + ms * (cc - c[1] * c[1]);
    d.Io[i][4] = I[5] - ms * c[1] * c[2];
    d.Io[i][5] = I[8] + ms * (cc - c[2] * c[2]);
  }
  std::memcpy(d.gravity, m.gravity, sizeof(double) * 3);
}

// The UnOCP kernels are compiled for a serial chain of NV revolute joints.
bool isRevoluteChain(const idocp_model_t& m, int nv) {
  if (m.njoints != nv || m.nv != nv || m.nq != nv || m.has_floating_base || m.ncontacts != 0) return false;
  for (int i = 0; i < nv; ++i)
    if (m.parent[i] != i - 1 || m.jtype[i] != IDOCP_JOINT_REVOLUTE || m.idx_v[i] != i) return false;
  return true;
}

}  // namespace

struct idocp_unocp {
  idocp_model_t model;
  idocp_cost_t cost;
  idocp_constraints_t cons;
  int N, batch, device, nv;
  double T;
  hipStream_t stream = nullptr;
  UnBuffers B{};
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;
  double shard_dt = 0.0; int shard_offset = 0, shard_terminal = 1, shard_prev = 0;      // the shard arguments of createImpl (idocp_unocp_clone)
  double *d_q0 = nullptr, *d_v0 = nullptr, *d_tmp = nullptr;   // staging for host-pointer entry points
  bool has_direction = false;
  int bwd = 0;                  // 1: UnParNMPC handle (backward-Euler stages, idocp_unparnmpc_*)
  int level_offset = 0;         // constraint time step of local stage i = i + level_offset
  int shard = 0;                // 1: a horizon shard of UnParNMPC (idocp_unparnmpc_create_shard)
  // filter line search (LineSearchFilter, src/line_search/line_search_filter.cpp): one filter per instance
  std::vector<std::vector<std::pair<double, double>>> filters;
};

namespace {

using L7 = UnLayout<7>;

int allocBuf(idocp_unocp* h, double** p, size_t n) {
  HIP_TRY(hipMalloc((void**)p, n * sizeof(double)));
  h->allocs.push_back(*p);
  h->alloc_bytes.push_back(n * sizeof(double));
  HIP_TRY(hipMemsetAsync(*p, 0, n * sizeof(double), h->stream));
  return IDOCP_OK;
}

struct FieldRef { int offset, dim, nstages_extra; };   // nstages = N + nstages_extra

bool solField(const std::string& n, int nv, FieldRef& f) {
  if (n == "lmd") f = {L7::S_LMD, nv, 1};
  else if (n == "gmm") f = {L7::S_GMM, nv, 1};
  else if (n == "q") f = {L7::S_Q, nv, 1};
  else if (n == "v") f = {L7::S_V, nv, 1};
  else if (n == "a") f = {L7::S_A, nv, 0};
  else if (n == "u") f = {L7::S_U, nv, 0};
  else if (n == "beta") f = {L7::S_BETA, nv, 0};
  else return false;
  return true;
}

int setDevice(const idocp_unocp* h) {
  HIP_TRY(hipSetDevice(h->device));
  return IDOCP_OK;
}

int copyRecords(idocp_unocp* h, const double* d_base, size_t rec_stride, size_t nrec, int offset, int dim, double* out) {
  // strided device -> host copy of one field of consecutive records
  HIP_TRY(hipMemcpy2DAsync(out, dim * sizeof(double), d_base + offset, rec_stride * sizeof(double), dim * sizeof(double),
                           nrec, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

}  // namespace

extern "C" {

int idocp_device_count(int* count) {
  if (!count) return IDOCP_E_ARG;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return IDOCP_OK;
}

int idocp_device_alloc(void** d_ptr, unsigned long long nbytes) {
  if (!d_ptr) return IDOCP_E_ARG;
  HIP_TRY(hipMalloc(d_ptr, nbytes));
  return IDOCP_OK;
}
int idocp_device_free(void* d_ptr) { HIP_TRY(hipFree(d_ptr)); return IDOCP_OK; }
int idocp_device_upload(void* d_dst, const void* h_src, unsigned long long nbytes) {
  HIP_TRY(hipMemcpy(d_dst, h_src, nbytes, hipMemcpyHostToDevice));
  return IDOCP_OK;
}

// The problem block of a handle (cost, limits, IPM parameters, shard description) from the structs of the C ABI: at creation and when the cost is
// replaced (idocp_unocp_set_cost).
static void fillUnProblem(const idocp_model_t& model, const idocp_cost_t& cost, const idocp_constraints_t& constraints, double T, int N, int batch, int bwd,
                          double dt, int stage_offset, int has_terminal, int has_prev, UnProblem& up) {
  std::memset(&up, 0, sizeof(up));
  up.N = N; up.batch = batch; up.T = T; up.dt = dt > 0.0 ? dt : T / N;
  up.stage_offset = stage_offset; up.has_terminal = has_terminal; up.has_prev = has_prev;
  for (int i = 0; i < model.nv; ++i) {
    up.q_ref[i] = cost.q_ref[i]; up.v_ref[i] = cost.v_ref[i]; up.u_ref[i] = cost.u_ref[i];
    up.q_weight[i] = cost.q_weight[i]; up.v_weight[i] = cost.v_weight[i]; up.a_weight[i] = cost.a_weight[i];
    up.u_weight[i] = cost.u_weight[i]; up.qf_weight[i] = cost.qf_weight[i]; up.vf_weight[i] = cost.vf_weight[i];
    up.q_min[i] = model.q_min[i]; up.q_max[i] = model.q_max[i]; up.v_max[i] = model.v_max[i]; up.u_max[i] = model.u_max[i];
  }
  up.use_q_limits = constraints.joint_position_limits; up.use_v_limits = constraints.joint_velocity_limits;
  up.use_u_limits = constraints.joint_torque_limits;
  up.use_a_lower = constraints.joint_acceleration_lower_limit ? 1 : 0;
  up.use_a_upper = constraints.joint_acceleration_upper_limit ? 1 : 0;
  for (int i = 0; i < IDOCP_MAX_NV; ++i) { up.a_min[i] = constraints.a_min[i]; up.a_max[i] = constraints.a_max[i]; }
  up.barrier = constraints.barrier; up.fraction_rate = constraints.fraction_to_boundary_rate;
  up.backward_euler = bwd;
  up.task.dim = cost.task_dim; up.task.joint = cost.task_joint;
  std::memcpy(up.task.R, cost.task_frame_R, sizeof(up.task.R)); std::memcpy(up.task.p, cost.task_frame_p, sizeof(up.task.p));
  std::memcpy(up.task.weight, cost.task_weight, sizeof(up.task.weight)); std::memcpy(up.task.weightf, cost.task_weightf, sizeof(up.task.weightf));
  if (cost.task_dim == 3) for (int k = 3; k < 6; ++k) up.task.weight[k] = up.task.weightf[k] = 0.0;
  std::memcpy(up.task.ref, cost.task_ref, sizeof(up.task.ref));
  up.task_n = cost.task_dim ? 1 + cost.task_extra_count : 0;
  for (int e = 0; e < IDOCP_MAX_EXTRA_TASKS; ++e) {
    TaskCost& tc = up.task_extra[e];
    std::memset(&tc, 0, sizeof(tc));
    if (cost.task_dim == 0 || e >= cost.task_extra_count) continue;
    const idocp_task_component_t& t = cost.task_extra[e];
    tc.dim = t.dim; tc.joint = t.joint;
    std::memcpy(tc.R, t.frame_R, sizeof(tc.R)); std::memcpy(tc.p, t.frame_p, sizeof(tc.p));
    std::memcpy(tc.weight, t.weight, sizeof(tc.weight)); std::memcpy(tc.weightf, t.weightf, sizeof(tc.weightf)); std::memcpy(tc.ref, t.ref, sizeof(tc.ref));
    if (t.dim == 3) for (int k = 3; k < 6; ++k) tc.weight[k] = tc.weightf[k] = 0.0;
  }
}

// bwd: 0 UnOCP, 1 UnParNMPC.  N, dt: the stages this handle holds and their time step; stage_offset / has_terminal / has_prev
// describe a horizon shard of UnParNMPC (0 / 1 / 0 for a whole horizon).
static int createImpl(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints,
                      double T, int N, int batch, int device, int bwd, idocp_unocp_t** out, double dt = 0.0, int stage_offset = 0,
                      int has_terminal = 1, int has_prev = 0) {
  if (!model || !cost || !constraints || !out) { set_last_error("idocp_unocp_create: null argument"); return IDOCP_E_ARG; }
  // argument checks of UnOCPSolver::UnOCPSolver (unocp_solver.cpp:33-47) and SplitUnOCP (split_unocp.hxx:26-33)
  if (!(T > 0)) { set_last_error("invalid value: T must be positive!"); return IDOCP_E_ARG; }
  if (N <= 0) { set_last_error("invalid value: N must be positive!"); return IDOCP_E_ARG; }
  if (batch <= 0) { set_last_error("invalid value: batch must be positive!"); return IDOCP_E_ARG; }
  if (!(constraints->barrier > 0)) { set_last_error("invalid value: barrier must be positive!"); return IDOCP_E_ARG; }      // constraint_component_base.hxx:10-24
  // JointAcceleration*Limit bounds: finite, and a_min < a_max where both are in use (an empty interval has no interior point to start the
  // barrier method from)
  for (int r = 0; r < model->nu; ++r) {
    const bool lo = constraints->joint_acceleration_lower_limit != 0, hi = constraints->joint_acceleration_upper_limit != 0;
    if ((lo && !std::isfinite(constraints->a_min[r])) || (hi && !std::isfinite(constraints->a_max[r]))) { set_last_error("invalid value: joint acceleration bounds must be finite!"); return IDOCP_E_ARG; }
    if (lo && hi && !(constraints->a_min[r] < constraints->a_max[r])) { set_last_error("invalid value: a_min must be smaller than a_max!"); return IDOCP_E_ARG; }
  }
  if (constraints->contact_distance) {
    set_last_error("unsupported constraints: ContactDistance belongs to the floating-base solvers (a fixed-base chain has no contacts here)");
    return IDOCP_E_UNSUPPORTED;
  }
  if (!(constraints->fraction_to_boundary_rate > 0 && constraints->fraction_to_boundary_rate <= 1)) {
    set_last_error("invalid value: fraction_to_boundary_rate must be in (0, 1]!"); return IDOCP_E_ARG;
  }
  if (model->has_floating_base) { set_last_error("robot has floating base: robot should have no constraints!"); return IDOCP_E_ARG; }
  if (model->ncontacts > 0) { set_last_error("robot can have contacts: robot should have no constraints!"); return IDOCP_E_ARG; }
  if (!isRevoluteChain(*model, 7)) {
    set_last_error("idocp_unocp_create: this build carries UnOCP kernels for a 7-dof revolute chain (iiwa14) only");
    return IDOCP_E_UNSUPPORTED;
  }
  if (cost->task_dim != 0) {
    if (cost->task_dim != 3 && cost->task_dim != 6) { set_last_error("invalid value: task_dim must be 0, 3 or 6!"); return IDOCP_E_ARG; }
    if (cost->task_joint < 0 || cost->task_joint >= model->njoints) { set_last_error("invalid value: task_joint is not a joint of the model!"); return IDOCP_E_ARG; }
    if (bwd) { set_last_error("idocp_unparnmpc_create: the task-space costs are carried by UnOCPSolver only"); return IDOCP_E_UNSUPPORTED; }
  }
  // further task-space components (idocp_cost_t::task_extra)
  if (cost->task_extra_count < 0 || cost->task_extra_count > IDOCP_MAX_EXTRA_TASKS) { set_last_error("invalid value: task_extra_count must be 0 .. " + std::to_string(IDOCP_MAX_EXTRA_TASKS) + "!"); return IDOCP_E_ARG; }
  if (cost->task_extra_count > 0 && cost->task_dim == 0) { set_last_error("invalid value: task_extra components need the first task-space component (task_dim != 0)!"); return IDOCP_E_ARG; }
  for (int e = 0; e < cost->task_extra_count; ++e) {
    if (cost->task_extra[e].dim != 3 && cost->task_extra[e].dim != 6) { set_last_error("invalid value: task_extra[" + std::to_string(e) + "].dim must be 3 or 6!"); return IDOCP_E_ARG; }
    if (cost->task_extra[e].joint < 0 || cost->task_extra[e].joint >= model->njoints) { set_last_error("invalid value: task_extra[" + std::to_string(e) + "].joint is not a joint of the model!"); return IDOCP_E_ARG; }
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    set_last_error("no HIP device available: the idocp HIP path has no CPU fallback");
    return IDOCP_E_DEVICE;
  }
  if (device < 0 || device >= ndev) { set_last_error("invalid device ordinal"); return IDOCP_E_ARG; }
  idocp_unocp* h = new idocp_unocp();
  h->model = *model; h->cost = *cost; h->cons = *constraints;
  h->N = N; h->batch = batch; h->device = device; h->T = T; h->nv = model->nv; h->bwd = bwd;
  h->level_offset = bwd ? 1 + stage_offset : 0; h->shard = (stage_offset != 0 || !has_terminal || has_prev) ? 1 : 0;
  h->shard_dt = dt; h->shard_offset = stage_offset; h->shard_terminal = has_terminal; h->shard_prev = has_prev;
  int rc = IDOCP_OK;
  auto fail = [&](int code) { (void)hipGetLastError(); idocp_unocp_destroy(h); return code; };      // (clears HIP's sticky last error: see HIP_TRY)
  if (hipSetDevice(device) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess) {
    set_last_error("hipSetDevice/hipStreamCreate failed");
    return fail(IDOCP_E_DEVICE);
  }
  const size_t nrec1 = (size_t)batch * (N + 1), nrec0 = (size_t)batch * N;
  UnBuffers& B = h->B;
  double* tmp;
  if ((rc = allocBuf(h, &B.sol, nrec1 * L7::SOL))) return fail(rc);
  if ((rc = allocBuf(h, &B.dir, nrec1 * L7::SOL))) return fail(rc);
  if ((rc = allocBuf(h, &B.slack, nrec0 * L7::CON))) return fail(rc);
  if ((rc = allocBuf(h, &B.dual, nrec0 * L7::CON))) return fail(rc);
  B.slack_a = nullptr; B.dual_a = nullptr;
  if (constraints->joint_acceleration_lower_limit || constraints->joint_acceleration_upper_limit) {      // rows of components 6, 7
    if ((rc = allocBuf(h, &B.slack_a, nrec0 * 2 * model->nv))) return fail(rc);
    if ((rc = allocBuf(h, &B.dual_a, nrec0 * 2 * model->nv))) return fail(rc);
  }
  if ((rc = allocBuf(h, &B.kkt, nrec0 * L7::KKT))) return fail(rc);
  if ((rc = allocBuf(h, &B.dyn, nrec0 * L7::DYN))) return fail(rc);
  if ((rc = allocBuf(h, &B.ric, nrec1 * L7::RIC))) return fail(rc);
  if ((rc = allocBuf(h, &B.gain, nrec0 * L7::GAIN))) return fail(rc);
  if ((rc = allocBuf(h, &B.step_stage, nrec0 * 2))) return fail(rc);
  if ((rc = allocBuf(h, &B.step, (size_t)batch * 2))) return fail(rc);
  if ((rc = allocBuf(h, &B.err_stage, nrec1))) return fail(rc);
  if ((rc = allocBuf(h, &B.err, (size_t)batch))) return fail(rc);
  if ((rc = allocBuf(h, &h->d_q0, (size_t)batch * model->nq))) return fail(rc);
  if ((rc = allocBuf(h, &h->d_v0, (size_t)batch * model->nv))) return fail(rc);
  if ((rc = allocBuf(h, &h->d_tmp, (size_t)batch * IDOCP_MAX_NQ))) return fail(rc);
  if ((rc = allocBuf(h, &tmp, ((size_t)batch * sizeof(int) + 7) / 8))) return fail(rc);
  B.status = reinterpret_cast<int*>(tmp);
  if ((rc = allocBuf(h, &B.ls_alpha, (size_t)batch))) return fail(rc);
  if ((rc = allocBuf(h, &B.ls_stage, nrec1 * 2))) return fail(rc);
  if ((rc = allocBuf(h, &B.ls_out, (size_t)batch * 2))) return fail(rc);
  h->filters.assign(batch, {});
  if (bwd) {
    if ((rc = allocBuf(h, &B.kinv, nrec0 * L7::KINV))) return fail(rc);
    if ((rc = allocBuf(h, &B.snew, nrec1 * L7::SOL))) return fail(rc);
    if ((rc = allocBuf(h, &B.aux, nrec1 * L7::AUX))) return fail(rc);
    if ((rc = allocBuf(h, &B.xres, nrec1 * L7::XRES))) return fail(rc);
    if ((rc = allocBuf(h, &B.xprev, (size_t)batch * L7::NX))) return fail(rc);
  }
  if (cost->task_dim != 0) {
    if ((rc = allocBuf(h, &B.task_ref, (size_t)(N + 1) * 12))) return fail(rc);
    if ((rc = allocBuf(h, &B.task_term, (size_t)batch * L7::TASK))) return fail(rc);
    B.task = 1; B.task_stride = L7::TASK;
    if (cost->task_extra_count > 0 && (rc = allocBuf(h, &B.task_xs, (size_t)batch * N * L7::TASK))) return fail(rc);      // stage terms of the further components
    std::vector<double> refs((size_t)(N + 1) * 12);
    for (int i = 0; i <= N; ++i) std::memcpy(&refs[12 * i], cost->task_ref, sizeof(double) * 12);
    if (hipMemcpyAsync(B.task_ref, refs.data(), refs.size() * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess) { set_last_error("hipMemcpy failed"); return fail(IDOCP_E_DEVICE); }
  }
  B.zaxes = 1;
  for (int i = 0; i < model->njoints; ++i) if (!(model->axis[i][0] == 0.0 && model->axis[i][1] == 0.0 && model->axis[i][2] == 1.0)) B.zaxes = 0;
  if (std::getenv("IDOCP_GENERAL_AXES")) B.zaxes = 0;      // tests: the general instantiation on a chain that qualifies for the special one
  DevModel dm; toDevModel(*model, dm);
  UnProblem up;
  fillUnProblem(*model, *cost, *constraints, T, N, batch, bwd, dt, stage_offset, has_terminal, has_prev, up);
  void *d_model = nullptr, *d_prob = nullptr;
  if (hipMalloc(&d_model, sizeof(DevModel)) != hipSuccess || hipMalloc(&d_prob, sizeof(UnProblem)) != hipSuccess) {
    set_last_error("hipMalloc failed"); return fail(IDOCP_E_DEVICE);
  }
  h->allocs.push_back(d_model); h->allocs.push_back(d_prob);
  h->alloc_bytes.push_back(sizeof(DevModel)); h->alloc_bytes.push_back(sizeof(UnProblem));
  if (hipMemcpyAsync(d_model, &dm, sizeof(dm), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipMemcpyAsync(d_prob, &up, sizeof(up), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess) {
    set_last_error("hipMemcpy failed"); return fail(IDOCP_E_DEVICE);
  }
  B.model = static_cast<const DevModel*>(d_model);
  B.prob = static_cast<const UnProblem*>(d_prob);
  // the reference constructor ends with initConstraints() (unocp_solver.cpp:47)
  UnLaunch<7>::initConstraints(B, batch, N, h->stream);
  if (bwd) UnLaunch<7>::parnmpcInitAux(B, batch, N, h->stream);
  if (hipStreamSynchronize(h->stream) != hipSuccess) { set_last_error("initConstraints launch failed"); return fail(IDOCP_E_DEVICE); }
  *out = h;
  return IDOCP_OK;
}

int idocp_unocp_create(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints,
                       double T, int N, int batch, int device, idocp_unocp_t** out) {
  return createImpl(model, cost, constraints, T, N, batch, device, 0, out);
}
// UnParNMPCSolver::UnParNMPCSolver (src/unocp/unparnmpc_solver.cpp:11-44): same arguments and checks
int idocp_unparnmpc_create(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints,
                           double T, int N, int batch, int device, idocp_unocp_t** out) {
  return createImpl(model, cost, constraints, T, N, batch, device, 1, out);
}

// The reference's solvers SHARE the CostFunction with the driver (unocp_solver.hpp: shared_ptr members), so a reference or a weight changed
// between two updateSolution calls takes effect at the next one; here the cost is copied at creation and this call is how an MPC loop moves
// its goal (the counterpart of idocp_ocp_set_cost).  The KIND of cost must stay what it was: a task-space component cannot appear or disappear,
// nor can the first task_extra component (their records are allocated at creation).  Stream-ordered: the next launch sees the new block.
int idocp_unocp_set_cost(idocp_unocp_t* h, const idocp_cost_t* cost) {
  if (!h || !cost) { set_last_error("idocp_unocp_set_cost: null argument"); return IDOCP_E_ARG; }
  if ((cost->task_dim != 0) != (h->cost.task_dim != 0) || cost->task_time_varying != h->cost.task_time_varying ||
      (cost->task_extra_count > 0) != (h->cost.task_extra_count > 0)) {
    set_last_error("idocp_unocp_set_cost: a task-space cost cannot be added or removed after creation");
    return IDOCP_E_UNSUPPORTED;
  }
  if (cost->task_dim != 0) {
    if (cost->task_dim != 3 && cost->task_dim != 6) { set_last_error("invalid value: task_dim must be 0, 3 or 6!"); return IDOCP_E_ARG; }
    if (cost->task_joint < 0 || cost->task_joint >= h->model.njoints) { set_last_error("invalid value: task_joint is not a joint of the model!"); return IDOCP_E_ARG; }
  }
  if (cost->task_extra_count < 0 || cost->task_extra_count > IDOCP_MAX_EXTRA_TASKS) { set_last_error("invalid value: task_extra_count must be 0 .. " + std::to_string(IDOCP_MAX_EXTRA_TASKS) + "!"); return IDOCP_E_ARG; }
  for (int e = 0; e < cost->task_extra_count; ++e) {
    if (cost->task_extra[e].dim != 3 && cost->task_extra[e].dim != 6) { set_last_error("invalid value: task_extra[" + std::to_string(e) + "].dim must be 3 or 6!"); return IDOCP_E_ARG; }
    if (cost->task_extra[e].joint < 0 || cost->task_extra[e].joint >= h->model.njoints) { set_last_error("invalid value: task_extra[" + std::to_string(e) + "].joint is not a joint of the model!"); return IDOCP_E_ARG; }
  }
  int rc = setDevice(h); if (rc) return rc;
  UnProblem up;
  fillUnProblem(h->model, *cost, h->cons, h->T, h->N, h->batch, h->bwd, h->shard_dt, h->shard_offset, h->shard_terminal, h->shard_prev, up);
  // (the block is read by kernels already queued on the stream: the copy is ordered behind them; `up` must outlive the asynchronous copy)
  HIP_TRY(hipMemcpyAsync(const_cast<UnProblem*>(h->B.prob), &up, sizeof(up), hipMemcpyHostToDevice, h->stream));
  if (cost->task_dim != 0 && !cost->task_time_varying) {      // the constant reference pose of every stage (time-varying: idocp_unocp_set_task_refs)
    std::vector<double> refs((size_t)(h->N + 1) * 12);
    for (int i = 0; i <= h->N; ++i) std::memcpy(&refs[12 * i], cost->task_ref, sizeof(double) * 12);
    HIP_TRY(hipMemcpyAsync(h->B.task_ref, refs.data(), refs.size() * sizeof(double), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->cost = *cost;
  return IDOCP_OK;
}

static int wrongKind(const idocp_unocp_t* h, int want_bwd);

// A deep copy of a solver (the reference's UnOCPSolver / UnParNMPCSolver are copyable, unocp_solver.hpp:59-62): the same problem on
// the same device, every device buffer copied -- iterate, slack / dual, Riccati factors, task references --, the line-search filter too
int idocp_unocp_clone(idocp_unocp_t* src, idocp_unocp_t** out) {
  if (!src || !out) return IDOCP_E_ARG;
  if (hipSetDevice(src->device) != hipSuccess) return IDOCP_E_DEVICE;
  idocp_unocp_t* h = nullptr;
  int rc = createImpl(&src->model, &src->cost, &src->cons, src->T, src->N, src->batch, src->device, src->bwd, &h, src->shard_dt, src->shard_offset,
                      src->shard_terminal, src->shard_prev);
  if (rc) return rc;
  auto fail = [&](int code) { (void)hipGetLastError(); idocp_unocp_destroy(h); return code; };      // (clears HIP's sticky last error: see HIP_TRY)
  if (h->allocs.size() != src->allocs.size() || h->alloc_bytes != src->alloc_bytes) { set_last_error("idocp_unocp_clone: allocation tables differ"); return fail(IDOCP_E_DEVICE); }
  if (hipStreamSynchronize(src->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  for (size_t i = 0; i < h->allocs.size(); ++i)
    if (hipMemcpyAsync(h->allocs[i], src->allocs[i], h->alloc_bytes[i], hipMemcpyDeviceToDevice, h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(IDOCP_E_DEVICE);
  h->filters = src->filters; h->has_direction = src->has_direction;
  *out = h;
  return IDOCP_OK;
}

// One shard of the horizon of UnParNMPCSolver: the stages [stage_begin, stage_end) of N (include/idocp_hip.h)
int idocp_unparnmpc_create_shard(const idocp_model_t* model, const idocp_cost_t* cost, const idocp_constraints_t* constraints,
                                 double T, int N, int stage_begin, int stage_end, int batch, int device, idocp_unocp_t** out) {
  if (N <= 0 || stage_begin < 0 || stage_end > N || stage_begin >= stage_end) { set_last_error("idocp_unparnmpc_create_shard: invalid stage range"); return IDOCP_E_ARG; }
  if (!(T > 0)) { set_last_error("invalid value: T must be positive!"); return IDOCP_E_ARG; }
  return createImpl(model, cost, constraints, T, stage_end - stage_begin, batch, device, 1, out, T / N, stage_begin, stage_end == N ? 1 : 0,
                    stage_begin > 0 ? 1 : 0);
}
// halo kinds as in idocp_parnmpc_halo_size: 0 state_last (q, v), 1 costate_first (lmd, gmm), 2 aux_first, 3 bwd_first (corrected
// lmd, gmm), 4 fwd_last (corrected q, v)
int idocp_unparnmpc_halo_size(int kind) {
  switch (kind) {
    case 0: case 1: case 3: case 4: return L7::NX;
    case 2: return L7::NX * L7::NX;
    default: return 0;
  }
}
int idocp_unparnmpc_export_halo(idocp_unocp_t* h, int kind, double* d_buf) {
  if (!h || !d_buf || kind < 0 || kind > 4) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  const long rs = (long)(h->N + 1) * L7::SOL, last = (long)(h->N - 1) * L7::SOL;
  const int nx = L7::NX;
  switch (kind) {
    case 0: stridedCopy(d_buf, nx, 0, h->B.sol, rs, last + L7::S_Q, nx, h->batch, h->stream); break;
    case 1: stridedCopy(d_buf, nx, 0, h->B.sol, rs, L7::S_LMD, nx, h->batch, h->stream); break;
    case 2: stridedCopy(d_buf, nx * nx, 0, h->B.aux, (long)(h->N + 1) * L7::AUX, 0, nx * nx, h->batch, h->stream); break;
    case 3: stridedCopy(d_buf, nx, 0, h->B.snew, rs, L7::S_LMD, nx, h->batch, h->stream); break;
    default: stridedCopy(d_buf, nx, 0, h->B.snew, rs, last + L7::S_Q, nx, h->batch, h->stream); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unparnmpc_import_halo(idocp_unocp_t* h, int kind, const double* d_buf) {
  if (!h || !d_buf || kind < 0 || kind > 4) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  const long rs = (long)(h->N + 1) * L7::SOL, next = (long)h->N * L7::SOL;
  const int nx = L7::NX, nv = h->nv;
  switch (kind) {
    case 0:   // the left neighbour's last state becomes this shard's "measured" state
      stridedCopy(h->d_q0, nv, 0, d_buf, nx, 0, nv, h->batch, h->stream);
      stridedCopy(h->d_v0, nv, 0, d_buf, nx, nv, nv, h->batch, h->stream);
      break;
    case 1: stridedCopy(h->B.sol, rs, next + L7::S_LMD, d_buf, nx, 0, nx, h->batch, h->stream); break;
    case 2: stridedCopy(h->B.aux, (long)(h->N + 1) * L7::AUX, (long)h->N * L7::AUX, d_buf, nx * nx, 0, nx * nx, h->batch, h->stream); break;
    case 3: stridedCopy(h->B.snew, rs, next + L7::S_LMD, d_buf, nx, 0, nx, h->batch, h->stream); break;
    default: stridedCopy(h->B.xprev, nx, 0, d_buf, nx, 0, nx, h->batch, h->stream); break;
  }
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unparnmpc_prev_state(idocp_unocp_t* h, double** d_q, double** d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  *d_q = h->d_q0; *d_v = h->d_v0;
  return IDOCP_OK;
}
int idocp_unparnmpc_step_sizes_device(idocp_unocp_t* h, double** d_steps) {
  if (!h || !d_steps) return IDOCP_E_ARG;
  *d_steps = h->B.step;
  return IDOCP_OK;
}
// squared KKT error of the local stages with the resident previous state (d_q0, d_v0): d_err2[batch] on the device
int idocp_unparnmpc_kkt_error_squared_device(idocp_unocp_t* h, double t, double* d_err2) {
  if (!h || !d_err2) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  (void)t;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::parnmpcResidual(h->B, h->batch, h->N, h->d_q0, h->d_v0, h->stream);
  squareInto(d_err2, h->B.err, h->batch, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

void idocp_unocp_destroy(idocp_unocp_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

static int setSolutionImpl(idocp_unocp_t* h, const char* name, const double* values, int per_instance, bool init_constraints = true) {
  if (!h || !name || !values) return IDOCP_E_ARG;
  FieldRef f;
  const std::string n(name);
  if (!(n == "q" || n == "v" || n == "a" || n == "u") || !solField(n, h->nv, f)) {
    set_last_error("invalid arugment: name must be q, v, a, or u!");
    return IDOCP_E_ARG;
  }
  int rc = setDevice(h); if (rc) return rc;
  const size_t cnt = (size_t)(per_instance ? h->batch : 1) * f.dim;
  HIP_TRY(hipMemcpyAsync(h->d_tmp, values, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  fillField(h->B.sol, L7::SOL, f.offset, f.dim, h->N + 1, h->batch, h->d_tmp, per_instance, h->stream);
  if (init_constraints) UnLaunch<7>::initConstraints(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unocp_set_solution(idocp_unocp_t* h, const char* name, const double* value) { return setSolutionImpl(h, name, value, 0); }
int idocp_unocp_set_solution_batch(idocp_unocp_t* h, const char* name, const double* values) { return setSolutionImpl(h, name, values, 1); }
// OCPSolver::setSolution / ParNMPCSolver::setSolution (ocp_solver.cpp:116-165, parnmpc_solver.cpp:128-180) leave the slack and dual variables alone
// (the driver calls initConstraints(t) itself): the setter of the facade's OCPSolver / ParNMPCSolver on a fixed-base robot without contacts,
// which are bound to these kernels (include/idocp/ocp/ocp_solver.hpp)
int idocp_unocp_set_solution_only(idocp_unocp_t* h, const char* name, const double* value) { return setSolutionImpl(h, name, value, 0, false); }

// TimeVaryingTaskSpace{3D,6D}Cost: the reference asks its TimeVaryingTaskSpace*RefBase for the pose at the time of every
// stage (time_varying_task_space_6d_cost.cpp:65-67 with t = t0 + i dt from unocp_solver.cpp:78-94); here the caller evaluates
// them and hands over the N + 1 poses.  refs: host, [N + 1][12] (rotation row-major, position).
int idocp_unocp_set_task_refs(idocp_unocp_t* h, const double* refs) {
  if (!h || !refs) { set_last_error("idocp_unocp_set_task_refs: null argument"); return IDOCP_E_ARG; }
  if (!h->B.task) { set_last_error("idocp_unocp_set_task_refs: the cost of this solver has no task-space component"); return IDOCP_E_ARG; }
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->B.task_ref, refs, sizeof(double) * 12 * (h->N + 1), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

int idocp_unocp_init_constraints(idocp_unocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::initConstraints(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// UnLineSearch::computeStepSize (include/idocp/line_search/unline_search.hpp:62-92) for every instance of the batch: the
// trial iterates are evaluated on the device (un_line_search_kernel), the filter logic (line_search_filter.cpp:33-63,
// defaults line_search_filter.hpp:16-17, line_search.hpp:25-26) runs here.  On return B.step holds the accepted primal
// step of every instance.  d_q, d_v: the measured state of the update (device).
static int lineSearchEval(idocp_unocp_t* h, const std::vector<double>& alpha, const double* d_q, const double* d_v, std::vector<double>& out) {
  HIP_TRY(hipMemcpyAsync(h->B.ls_alpha, alpha.data(), sizeof(double) * h->batch, hipMemcpyHostToDevice, h->stream));
  UnLaunch<7>::lineSearchEval(h->B, h->batch, h->N, h->bwd != 0, d_q, d_v, h->stream);
  HIP_TRY(hipGetLastError());
  out.resize((size_t)h->batch * 2);
  HIP_TRY(hipMemcpyAsync(out.data(), h->B.ls_out, sizeof(double) * out.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
static int runLineSearch(idocp_unocp_t* h, const double* d_q, const double* d_v) {
  if (h->shard) { set_last_error("line_search=true is not supported on a horizon shard"); return IDOCP_E_UNSUPPORTED; }
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
    if ((rc = lineSearchEval(h, alpha, d_q, d_v, cv))) return rc;
    for (int b = 0; b < B; ++b) if (h->filters[b].empty()) augment(h->filters[b], cv[2 * b], cv[2 * b + 1]);
  }
  HIP_TRY(hipMemcpyAsync(step.data(), h->B.step, sizeof(double) * step.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<char> done(B, 0);
  int open = 0;
  for (int b = 0; b < B; ++b) { alpha[b] = step[2 * b]; if (!(alpha[b] > min_step)) done[b] = 1; else ++open; }
  while (open > 0) {
    if ((rc = lineSearchEval(h, alpha, d_q, d_v, cv))) return rc;
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

static int wrongKind(const idocp_unocp_t* h, int want_bwd) {
  if (h->bwd == want_bwd) return 0;
  set_last_error(want_bwd ? "this handle is an UnOCPSolver (idocp_unocp_create): use idocp_unocp_*"
                          : "this handle is an UnParNMPCSolver (idocp_unparnmpc_create): use idocp_unparnmpc_*");
  return IDOCP_E_ARG;
}

int idocp_unocp_update_solution_device(idocp_unocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  (void)t;   // ConfigurationSpaceCost is time-invariant
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  UnLaunch<7>::linearize(h->B, h->batch, h->N, h->stream);
  UnLaunch<7>::riccati(h->B, h->batch, h->N, d_q, d_v, h->stream);
  UnLaunch<7>::expand(h->B, h->batch, h->N, h->stream);
  UnLaunch<7>::integrate(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  h->has_direction = true;
  return IDOCP_OK;
}

static int statusOf(idocp_unocp_t* h) {
  std::vector<int> st(h->batch);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.status, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b)
    if (st[b] != 0) { set_last_error("Riccati: Qaa not positive definite (instance " + std::to_string(b) + ")"); return st[b]; }
  return IDOCP_OK;
}

int idocp_unocp_synchronize(idocp_unocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

void* idocp_unocp_stream(idocp_unocp_t* h) { return h ? (void*)h->stream : nullptr; }

int idocp_unocp_update_solution(idocp_unocp_t* h, double t, const double* q, const double* v, int line_search) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * h->model.nq, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * h->model.nv, hipMemcpyHostToDevice, h->stream));
  if (!line_search) {
    rc = idocp_unocp_update_solution_device(h, t, h->d_q0, h->d_v0);
    if (rc) return rc;
    return statusOf(h);
  }
  // unocp_solver.cpp:116-120: the filter line search sits between the direction and the update
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  UnLaunch<7>::linearize(h->B, h->batch, h->N, h->stream);
  UnLaunch<7>::riccati(h->B, h->batch, h->N, h->d_q0, h->d_v0, h->stream);
  UnLaunch<7>::expand(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  if ((rc = statusOf(h))) return rc;
  if ((rc = runLineSearch(h, h->d_q0, h->d_v0))) return rc;
  UnLaunch<7>::integrate(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->has_direction = true;
  return IDOCP_OK;
}

// UnOCPSolver / UnParNMPCSolver::clearLineSearchFilter (unocp_solver.cpp:185-187)
int idocp_unocp_clear_line_search_filter(idocp_unocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  for (auto& f : h->filters) f.clear();
  return IDOCP_OK;
}
// (total cost, total constraint violation) of s + alpha[b] d for every instance (UnLineSearch::computeCostAndViolation;
// alpha = 0: the iterate itself), with the measured state of the last host-pointer update / residual call
int idocp_unocp_line_search_eval(idocp_unocp_t* h, const double* alpha, double* cost, double* violation) {
  if (!h || !alpha || !cost || !violation) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  std::vector<double> a(alpha, alpha + h->batch), cv;
  if ((rc = lineSearchEval(h, a, h->d_q0, h->d_v0, cv))) return rc;
  for (int b = 0; b < h->batch; ++b) { cost[b] = cv[2 * b]; violation[b] = cv[2 * b + 1]; }
  return IDOCP_OK;
}

int idocp_unocp_launch_linearize(idocp_unocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h) return IDOCP_E_ARG;
  (void)t; (void)d_q; (void)d_v;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::linearize(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_unocp_launch_riccati(idocp_unocp_t* h, const double* d_q, const double* d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::riccati(h->B, h->batch, h->N, d_q, d_v, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_unocp_launch_expand(idocp_unocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::expand(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_unocp_launch_integrate(idocp_unocp_t* h) {
  if (!h) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::integrate(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}

int idocp_unocp_launch_kernel(idocp_unocp_t* h, int kernel_id, const double* d_q, const double* d_v) {
  if (!h || kernel_id < 0 || kernel_id > 5 || !d_q || !d_v) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::single(kernel_id, h->B, h->batch, h->N, d_q, d_v, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}

int idocp_unocp_compute_kkt_residual(idocp_unocp_t* h, double t, const double* q, const double* v) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  if (wrongKind(h, 0)) return IDOCP_E_ARG;
  (void)t;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::residual(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unocp_kkt_error(idocp_unocp_t* h, double* kkt_error) {
  if (!h || !kkt_error) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(kkt_error, h->B.err, sizeof(double) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

// ---- UnParNMPCSolver (src/unocp/unparnmpc_solver.cpp) on an idocp_unparnmpc_create handle ----
int idocp_unparnmpc_init_backward_correction(idocp_unocp_t* h, double t) {          // unparnmpc_solver.cpp:69-71
  if (!h) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  (void)t;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::parnmpcInitAux(h->B, h->batch, h->N, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unparnmpc_launch_phase(idocp_unocp_t* h, int phase, const double* d_q, const double* d_v) {
  if (!h || phase < 0 || phase > 6 || !d_q || !d_v) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  UnLaunch<7>::parnmpcPhase(phase, h->B, h->batch, h->N, d_q, d_v, h->stream);
  HIP_TRY(hipGetLastError());
  return IDOCP_OK;
}
int idocp_unparnmpc_update_solution_device(idocp_unocp_t* h, double t, const double* d_q, const double* d_v) {
  if (!h || !d_q || !d_v) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  (void)t;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
  for (int phase = 0; phase <= 6; ++phase) UnLaunch<7>::parnmpcPhase(phase, h->B, h->batch, h->N, d_q, d_v, h->stream);
  HIP_TRY(hipGetLastError());
  h->has_direction = true;
  return IDOCP_OK;
}
int idocp_unparnmpc_update_solution(idocp_unocp_t* h, double t, const double* q, const double* v, int line_search) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * h->model.nq, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * h->model.nv, hipMemcpyHostToDevice, h->stream));
  if (!line_search) {
    if ((rc = idocp_unparnmpc_update_solution_device(h, t, h->d_q0, h->d_v0))) return rc;
  } else {                                           // unparnmpc_solver.cpp:81-86
    HIP_TRY(hipMemsetAsync(h->B.status, 0, sizeof(int) * h->batch, h->stream));
    for (int phase = 0; phase <= 5; ++phase) UnLaunch<7>::parnmpcPhase(phase, h->B, h->batch, h->N, h->d_q0, h->d_v0, h->stream);
    HIP_TRY(hipGetLastError());
    if ((rc = runLineSearch(h, h->d_q0, h->d_v0))) return rc;
    UnLaunch<7>::parnmpcPhase(6, h->B, h->batch, h->N, h->d_q0, h->d_v0, h->stream);
    HIP_TRY(hipGetLastError());
    h->has_direction = true;
  }
  std::vector<int> st(h->batch);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.status, sizeof(int) * h->batch, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b)
    if (st[b] != 0) { set_last_error("UnParNMPC: stage KKT matrix not invertible (Q or F Q^-1 F^T not positive definite), instance " + std::to_string(b)); return st[b]; }
  return IDOCP_OK;
}
int idocp_unparnmpc_compute_kkt_residual(idocp_unocp_t* h, double t, const double* q, const double* v) {
  if (!h || !q || !v) return IDOCP_E_ARG;
  if (wrongKind(h, 1)) return IDOCP_E_ARG;
  (void)t;
  int rc = setDevice(h); if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(h->d_q0, q, sizeof(double) * h->batch * h->model.nq, hipMemcpyHostToDevice, h->stream));
  HIP_TRY(hipMemcpyAsync(h->d_v0, v, sizeof(double) * h->batch * h->model.nv, hipMemcpyHostToDevice, h->stream));
  UnLaunch<7>::parnmpcResidual(h->B, h->batch, h->N, h->d_q0, h->d_v0, h->stream);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}

static int getRecords(idocp_unocp_t* h, const double* base, const char* name, int instance, double* out, bool direction) {
  if (!h || !name || !out || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  std::string n(name);
  if (direction) { if (n.size() < 2 || n[0] != 'd') return IDOCP_E_ARG; n = n.substr(1); }
  FieldRef f;
  if (!solField(n, h->nv, f)) { set_last_error(std::string("unknown field name: ") + name); return IDOCP_E_ARG; }
  int rc = setDevice(h); if (rc) return rc;
  const size_t nst = h->N + f.nstages_extra;
  return copyRecords(h, base + (size_t)instance * (h->N + 1) * L7::SOL, L7::SOL, nst, f.offset, f.dim, out);
}
int idocp_unocp_get_solution(idocp_unocp_t* h, const char* name, int instance, double* out) {
  return getRecords(h, h ? h->B.sol : nullptr, name, instance, out, false);
}
// UnOCPSolver / UnParNMPCSolver::getSolution(stage): the split solution of one stage in one device-to-host copy;
// out: lmd gmm q v a u beta (7 nv doubles, the order of the sol record)
int idocp_unocp_get_split_solution(idocp_unocp_t* h, int instance, int stage, double* out) {
  if (!h || !out || instance < 0 || instance >= h->batch || stage < 0 || stage > h->N) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  static_assert(L7::S_LMD == 0 && L7::S_BETA == 6 * 7, "record order = output order");
  HIP_TRY(hipMemcpyAsync(out, h->B.sol + ((size_t)instance * (h->N + 1) + stage) * L7::SOL, sizeof(double) * 7 * h->nv, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  return IDOCP_OK;
}
int idocp_unocp_get_direction(idocp_unocp_t* h, const char* name, int instance, double* out) {
  return getRecords(h, h ? h->B.dir : nullptr, name, instance, out, true);
}
int idocp_unparnmpc_get_new_solution(idocp_unocp_t* h, const char* name, int instance, double* out) {
  if (!h || wrongKind(h, 1)) return IDOCP_E_ARG;
  return getRecords(h, h->B.snew, name, instance, out, false);
}

int idocp_unocp_get_step_sizes(idocp_unocp_t* h, double* primal, double* dual) {
  if (!h || !primal || !dual) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  std::vector<double> st((size_t)h->batch * 2);
  HIP_TRY(hipMemcpyAsync(st.data(), h->B.step, sizeof(double) * st.size(), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int b = 0; b < h->batch; ++b) { primal[b] = st[2 * b]; dual[b] = st[2 * b + 1]; }
  return IDOCP_OK;
}

int idocp_unocp_get_riccati(idocp_unocp_t* h, int instance, double* P, double* s, double* K, double* k) {
  if (!h || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  const int nv = h->nv, nx = 2 * nv, N = h->N;
  std::vector<double> ric((size_t)(N + 1) * L7::RIC), gain((size_t)N * L7::GAIN);
  HIP_TRY(hipMemcpyAsync(ric.data(), h->B.ric + (size_t)instance * (N + 1) * L7::RIC, ric.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(gain.data(), h->B.gain + (size_t)instance * N * L7::GAIN, gain.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  for (int i = 0; i <= N; ++i) {
    const double* r = &ric[(size_t)i * L7::RIC];
    if (P) {
      double* Pm = P + (size_t)i * nx * nx;
      for (int c = 0; c < nv; ++c) for (int rr = 0; rr < nv; ++rr) {
        Pm[c * nx + rr] = r[L7::R_PQQ + L7::sym(rr, c)];           // Pqq (packed upper triangle)
        Pm[(nv + c) * nx + rr] = r[L7::R_PQV + c * nv + rr];       // Pqv
        Pm[c * nx + nv + rr] = r[L7::R_PQV + rr * nv + c];         // Pvq = Pqv^T
        Pm[(nv + c) * nx + nv + rr] = r[L7::R_PVV + L7::sym(rr, c)];  // Pvv
      }
    }
    if (s) { std::memcpy(s + (size_t)i * nx, r + L7::R_SQ, sizeof(double) * nv); std::memcpy(s + (size_t)i * nx + nv, r + L7::R_SV, sizeof(double) * nv); }
    if (i < N) {
      const double* g = &gain[(size_t)i * L7::GAIN];
      if (K) std::memcpy(K + (size_t)i * nv * nx, g + L7::G_K, sizeof(double) * nv * nx);
      if (k) std::memcpy(k + (size_t)i * nv, g + L7::G_k, sizeof(double) * nv);
    }
  }
  return IDOCP_OK;
}

// OCPSolver::getStateFeedbackGain (ocp_solver.cpp:101-111) for the fixed-base robot without contacts: the torque policy du = Kq dq + Kv dv of the
// contact-dynamics formulation is the acceleration policy da = Ka dx + ka of this one mapped through the linearised inverse dynamics,
// du = ID + dID/dq dq + dID/dv dv + M da (unconstrained_dynamics.hxx:84-92):  Kq = dID/dq + M Ka_q,  Kv = dID/dv + M Ka_v, with the
// derivatives of the SAME linearisation the gains belong to (the dyn record of the last updateSolution).  Kq, Kv: nv x nv col-major.
int idocp_unocp_get_torque_feedback_gain(idocp_unocp_t* h, int instance, int stage, double* Kq, double* Kv) {
  if (!h || !Kq || !Kv || instance < 0 || instance >= h->batch || stage < 0 || stage >= h->N) { set_last_error("idocp_unocp_get_torque_feedback_gain: invalid argument"); return IDOCP_E_ARG; }
  if (h->bwd || !h->B.gain) { set_last_error("idocp_unocp_get_torque_feedback_gain: the LQR policy belongs to UnOCPSolver (UnParNMPCSolver keeps none)"); return IDOCP_E_UNSUPPORTED; }
  int rc = setDevice(h); if (rc) return rc;
  const int nv = h->nv;
  std::vector<double> dyn(L7::DYN), gain(L7::GAIN);
  const size_t rec = (size_t)instance * h->N + stage;
  HIP_TRY(hipMemcpyAsync(dyn.data(), h->B.dyn + rec * L7::DYN, dyn.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(gain.data(), h->B.gain + rec * L7::GAIN, gain.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const double *dq = &dyn[L7::D_DQ], *dv = &dyn[L7::D_DV], *M = &dyn[L7::D_DA], *Ka = &gain[L7::G_K];
  for (int c = 0; c < nv; ++c) for (int r = 0; r < nv; ++r) {
    double aq = dq[c * nv + r], av = dv[c * nv + r];
    for (int m = 0; m < nv; ++m) { aq += M[m * nv + r] * Ka[c * nv + m]; av += M[m * nv + r] * Ka[(nv + c) * nv + m]; }
    Kq[c * nv + r] = aq; Kv[c * nv + r] = av;
  }
  return IDOCP_OK;
}

// UnOCPSolver::isCurrentSolutionFeasible (unocp_solver.cpp:228-237): joint limits of stages 0..N-1 with the time-step
// gating of constraints_data.hpp:18-42, checked on the host from the downloaded solution records.
int idocp_unocp_is_current_solution_feasible(idocp_unocp_t* h, int* feasible, int* where) {
  if (!h || !feasible) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  const int nv = h->nv, N = h->N;
  std::vector<double> sol((size_t)h->batch * (N + 1) * L7::SOL);
  HIP_TRY(hipMemcpyAsync(sol.data(), h->B.sol, sol.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  const idocp_model_t& m = h->model;
  for (int b = 0; b < h->batch; ++b) {
    int bad = -1;
    for (int i = 0; i < N && bad < 0; ++i) {
      const double* s = &sol[((size_t)b * (N + 1) + i) * L7::SOL];
      for (int r = 0; r < nv && bad < 0; ++r) {
        if (h->cons.joint_position_limits && i + h->level_offset >= 2 && (s[L7::S_Q + r] < m.q_min[r] || s[L7::S_Q + r] > m.q_max[r])) bad = i;
        if (h->cons.joint_velocity_limits && i + h->level_offset >= 1 && (s[L7::S_V + r] < -m.v_max[r] || s[L7::S_V + r] > m.v_max[r])) bad = i;
        if (h->cons.joint_torque_limits && (s[L7::S_U + r] < -m.u_max[r] || s[L7::S_U + r] > m.u_max[r])) bad = i;
        if (h->cons.joint_acceleration_lower_limit && s[L7::S_A + r] < h->cons.a_min[r]) bad = i;      // joint_acceleration_lower_limit.cpp:38-47
        if (h->cons.joint_acceleration_upper_limit && s[L7::S_A + r] > h->cons.a_max[r]) bad = i;
      }
    }
    feasible[b] = bad < 0 ? 1 : 0;
    if (where) where[b] = bad;
  }
  return IDOCP_OK;
}

int idocp_unocp_dimc(const idocp_unocp_t* h) {
  if (!h) return 0;
  return h->nv * 2 * ((h->cons.joint_position_limits ? 1 : 0) + (h->cons.joint_velocity_limits ? 1 : 0) + (h->cons.joint_torque_limits ? 1 : 0)) +
         h->nv * ((h->cons.joint_acceleration_lower_limit ? 1 : 0) + (h->cons.joint_acceleration_upper_limit ? 1 : 0));
}

// [N][dimc] with the enabled components in the reference's order; rows that are
// not valid at a stage (constraints_data.hpp:18-42) read 0.
int idocp_unocp_get_constraint_data(idocp_unocp_t* h, int instance, double* slack, double* dual) {
  if (!h || instance < 0 || instance >= h->batch) return IDOCP_E_ARG;
  int rc = setDevice(h); if (rc) return rc;
  const int nv = h->nv, N = h->N, dimc = idocp_unocp_dimc(h);
  std::vector<double> sl((size_t)N * L7::CON), du((size_t)N * L7::CON);
  HIP_TRY(hipMemcpyAsync(sl.data(), h->B.slack + (size_t)instance * N * L7::CON, sl.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipMemcpyAsync(du.data(), h->B.dual + (size_t)instance * N * L7::CON, du.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> sa, da;
  if (h->B.slack_a) {
    sa.resize((size_t)N * 2 * nv); da.resize(sa.size());
    HIP_TRY(hipMemcpyAsync(sa.data(), h->B.slack_a + (size_t)instance * N * 2 * nv, sa.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(da.data(), h->B.dual_a + (size_t)instance * N * 2 * nv, da.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  const int use[8] = {h->cons.joint_position_limits, h->cons.joint_position_limits, h->cons.joint_velocity_limits, h->cons.joint_velocity_limits,
                      h->cons.joint_torque_limits, h->cons.joint_torque_limits, h->cons.joint_acceleration_lower_limit, h->cons.joint_acceleration_upper_limit};
  for (int i = 0; i < N; ++i) {
    int off = 0;
    for (int c = 0; c < 8; ++c) {
      if (!use[c]) continue;
      const bool valid = (c < 2) ? i + h->level_offset >= 2 : ((c < 4) ? i + h->level_offset >= 1 : true);
      for (int r = 0; r < nv; ++r) {
        const double sv = c < 6 ? sl[(size_t)i * L7::CON + c * nv + r] : sa[(size_t)i * 2 * nv + (c - 6) * nv + r];
        const double dv = c < 6 ? du[(size_t)i * L7::CON + c * nv + r] : da[(size_t)i * 2 * nv + (c - 6) * nv + r];
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? sv : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? dv : 0.0;
      }
      off += nv;
    }
  }
  return IDOCP_OK;
}

int idocp_rnea_derivatives(const idocp_model_t* model, int n, const double* q, const double* v, const double* a,
                           double* tau, double* dtau_dq, double* dtau_dv, double* dtau_da, int device) {
  if (!model || n <= 0 || !q || !v || !a || !tau || !dtau_dq || !dtau_dv || !dtau_da) return IDOCP_E_ARG;
  if (!isRevoluteChain(*model, 7)) { set_last_error("idocp_rnea_derivatives: 7-dof revolute chain only in this build"); return IDOCP_E_UNSUPPORTED; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) { set_last_error("no HIP device available"); return IDOCP_E_DEVICE; }
  HIP_TRY(hipSetDevice(device));
  const int nv = model->nv;
  DevModel dm; toDevModel(*model, dm);
  // one allocation, released on every way out of this function
  struct Scratch { void* p = nullptr; ~Scratch() { if (p) (void)hipFree(p); } } scratch;
  const size_t nvec = (size_t)n * nv, nmat = nvec * nv, model_doubles = (sizeof(dm) + sizeof(double) - 1) / sizeof(double);
  HIP_TRY(hipMalloc(&scratch.p, sizeof(double) * (model_doubles + 4 * nvec + 3 * nmat)));
  double* base = static_cast<double*>(scratch.p);
  DevModel* d_m = reinterpret_cast<DevModel*>(base);
  double *d_q = base + model_doubles, *d_v = d_q + nvec, *d_a = d_v + nvec, *d_tau = d_a + nvec, *d_dq = d_tau + nvec, *d_dv = d_dq + nmat, *d_da = d_dv + nmat;
  HIP_TRY(hipMemcpy(d_m, &dm, sizeof(dm), hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_q, q, sizeof(double) * n * nv, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_v, v, sizeof(double) * n * nv, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(d_a, a, sizeof(double) * n * nv, hipMemcpyHostToDevice));
  bool zaxes = !std::getenv("IDOCP_GENERAL_AXES");      // (the same choice of the sweep's instantiation as the solvers make)
  for (int i = 0; i < model->njoints; ++i) if (!(model->axis[i][0] == 0.0 && model->axis[i][1] == 0.0 && model->axis[i][2] == 1.0)) zaxes = false;
  UnLaunch<7>::rneaDerivatives(d_m, n, d_q, d_v, d_a, d_tau, d_dq, d_dv, d_da, zaxes, nullptr);
  HIP_TRY(hipGetLastError());
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(tau, d_tau, sizeof(double) * n * nv, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dtau_dq, d_dq, sizeof(double) * n * nv * nv, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dtau_dv, d_dv, sizeof(double) * n * nv * nv, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dtau_da, d_da, sizeof(double) * n * nv * nv, hipMemcpyDeviceToHost));
  return IDOCP_OK;
}

}  // extern "C"
