// Horizon-sharded ParNMPC: the multi-GPU driver behind the C ABI (BASELINE.json configs[3]).
//
// Counterpart of BackwardCorrectionSolver (src/ocp/backward_correction_solver.cpp:255-366), whose four correction sweeps the
// reference runs over ONE horizon with OpenMP: here rank r owns the stages [r N/G, (r+1) N/G) of every instance (one process per
// GPU, idocp_parnmpc_create_shard / _create_hybrid_shard) and per iteration exchanges with its two neighbours only
//     state_last (q, v -> right)   costate_first (lmd, gmm, q -> left)   aux_first (aux_mat -> left)
//     bwd_first (corrected lmd, gmm, right -> left pipeline)             fwd_last (corrected q, v, left -> right pipeline)
// plus an all-reduce(min) of the step sizes and an all-reduce(sum) of the squared KKT error.  Everything is enqueued on the
// shard's own HIP stream: pack kernel -> ncclSend / ncclRecv (RCCL, point-to-point over xGMI) -> unpack kernel -> phase kernels,
// with persistent halo buffers and NO host synchronisation inside an iteration (round 1 drove the same protocol from Python
// with a stream sync and a blocking send / recv per phase).  The halos are small (2.7 MB per neighbour and iteration at batch
// 256, dominated by aux_first), so the boundary exchange is simply issued at the head of the iteration; the two serial sweeps are
// pipelines across the ranks by construction and cost G hops of latency.
//
// RCCL is loaded with dlopen on first use (no link-time dependency: a host process that already carries another copy of the
// library, e.g. torch's, keeps its own).  A second transport, `local`, connects several shard handles living on ONE GPU in one
// process (one host thread per endpoint, host-side rendezvous): it exists so that this very driver is covered by a test on a
// single-GPU box (tests/test_parnmpc_gpu.py) -- it is not a fallback of the product path.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <cmath>
#include <condition_variable>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "idocp_hip.h"

extern "C" void idocp_set_last_error_string(const char* msg);

namespace {

struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Broadcast)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*GetVersion)(int*) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*CommUserRank)(const ncclComm_t, int*) = nullptr;
  bool load() {
    if (lib) return true;
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
    if (!lib) return false;
#define SYM(f) f = reinterpret_cast<decltype(f)>(dlsym(lib, "nccl" #f)); if (!f) return false;
    SYM(GetUniqueId) SYM(CommInitRank) SYM(CommDestroy) SYM(Send) SYM(Recv) SYM(AllReduce) SYM(Broadcast) SYM(GroupStart) SYM(GroupEnd) SYM(GetErrorString)
    SYM(GetVersion) SYM(CommCount) SYM(CommUserRank)
#undef SYM
    return true;
  }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;

// in-process transport (tests): mailboxes in device memory, host-side rendezvous between the endpoints' threads
struct LocalHub {
  int world;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::vector<std::pair<double*, size_t>>> box;      // (src, dst) -> queue of device buffers
  std::vector<std::vector<double>> reduce_in;     // all-reduce staging (host)
  int reduce_count = 0, reduce_gen = 0;
  std::vector<double> reduce_out;
  int refs = 0;
};

}  // namespace

struct idocp_comm {
  int rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  LocalHub* hub = nullptr;
  bool force_collectives = false;     // world == 1: issue the collectives through RCCL anyway (idocp_comm_set_force_collectives)
  // third transport (idocp_comm_init_callbacks): host-staged point-to-point / collectives through caller-supplied functions.  It follows the
  // RCCL branch of the driver call by call -- same grouping, same order, same lifetime -- so that two PROCESSES can run the driver on a box
  // where RCCL cannot connect them (one GPU: tests/test_parnmpc_gpu.py backs it with gloo)
  bool has_cb = false;
  idocp_comm_callbacks_t cb{};
  bool in_group = false;
  std::vector<int> pending_recv;      // halo kinds received inside the open group: copied to the device when the group ends
};

namespace {

enum { STATE_LAST = 0, COSTATE_FIRST = 1, AUX_FIRST = 2, BWD_FIRST = 3, FWD_LAST = 4, AUX_ALL = 5, NKINDS = 6 };

struct DistState {
  idocp_comm* comm = nullptr;
  int batch = 0;
  hipStream_t stream = nullptr;
  double *sendb[NKINDS] = {}, *recvb[NKINDS] = {};
  size_t count[NKINDS] = {};
  double *d_q = nullptr, *d_v = nullptr, *d_steps = nullptr, *d_err2 = nullptr;
  std::vector<double> hsend[NKINDS], hrecv[NKINDS], hred;      // host staging of the callback transport
};
std::map<idocp_ocp_t*, DistState> g_dist;
std::mutex g_dist_mutex;

int fail(int code, const std::string& msg) { idocp_set_last_error_string(msg.c_str()); return code; }
#define HIPC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { (void)hipGetLastError(); return fail(IDOCP_E_DEVICE, std::string(#x " failed: ") + hipGetErrorString(e_)); } } while (0)      // (the sticky last error is cleared: ocp_capi.hip HIP_TRY)
#define NCCLC(x) do { ncclResult_t r_ = (x); if (r_ != ncclSuccess) return fail(IDOCP_E_DEVICE, std::string(#x " failed: ") + g_rccl.GetErrorString(r_)); } while (0)
#define RC(x) do { int rc_ = (x); if (rc_) return rc_; } while (0)

DistState* stateOf(idocp_ocp_t* h) {
  std::lock_guard<std::mutex> lk(g_dist_mutex);
  auto it = g_dist.find(h);
  return it == g_dist.end() ? nullptr : &it->second;
}

// ---- transport: point-to-point and collectives, stream-ordered (RCCL) or host-rendezvous (local) ----
int xsend(DistState& s, int kind, int peer) {
  idocp_comm* c = s.comm;
  if (c->nccl) { NCCLC(g_rccl.Send(s.sendb[kind], s.count[kind], ncclDouble, peer, c->nccl, s.stream)); return IDOCP_OK; }
  if (c->has_cb) {
    s.hsend[kind].resize(s.count[kind]);
    HIPC(hipMemcpyAsync(s.hsend[kind].data(), s.sendb[kind], s.count[kind] * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    if (c->cb.send(c->cb.ctx, s.hsend[kind].data(), (unsigned long)s.count[kind], peer, c->in_group ? 1 : 0)) return fail(IDOCP_E_DEVICE, "callback transport: send failed");
    return IDOCP_OK;
  }
  LocalHub* hub = c->hub;
  double* copy = nullptr;
  HIPC(hipMalloc((void**)&copy, s.count[kind] * sizeof(double)));
  HIPC(hipMemcpyAsync(copy, s.sendb[kind], s.count[kind] * sizeof(double), hipMemcpyDeviceToDevice, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  { std::lock_guard<std::mutex> lk(hub->m); hub->box[{c->rank, peer}].push_back({copy, s.count[kind]}); }
  hub->cv.notify_all();
  return IDOCP_OK;
}
int xrecv(DistState& s, int kind, int peer) {
  idocp_comm* c = s.comm;
  if (c->nccl) { NCCLC(g_rccl.Recv(s.recvb[kind], s.count[kind], ncclDouble, peer, c->nccl, s.stream)); return IDOCP_OK; }
  if (c->has_cb) {
    s.hrecv[kind].resize(s.count[kind]);
    if (c->cb.recv(c->cb.ctx, s.hrecv[kind].data(), (unsigned long)s.count[kind], peer, c->in_group ? 1 : 0)) return fail(IDOCP_E_DEVICE, "callback transport: recv failed");
    if (c->in_group) { c->pending_recv.push_back(kind); return IDOCP_OK; }      // (posted: the data is there when the group ends)
    HIPC(hipMemcpyAsync(s.recvb[kind], s.hrecv[kind].data(), s.count[kind] * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    return IDOCP_OK;
  }
  LocalHub* hub = c->hub;
  std::pair<double*, size_t> msg;
  {
    std::unique_lock<std::mutex> lk(hub->m);
    auto& q = hub->box[{peer, c->rank}];
    hub->cv.wait(lk, [&] { return !q.empty(); });
    msg = q.front(); q.erase(q.begin());
  }
  if (msg.second != s.count[kind]) return fail(IDOCP_E_ARG, "local transport: halo size mismatch");
  HIPC(hipMemcpyAsync(s.recvb[kind], msg.first, msg.second * sizeof(double), hipMemcpyDeviceToDevice, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  HIPC(hipFree(msg.first));
  return IDOCP_OK;
}
int xgroupStart(DistState& s) {
  if (s.comm->nccl) NCCLC(g_rccl.GroupStart());
  if (s.comm->has_cb) {
    s.comm->in_group = true; s.comm->pending_recv.clear();
    if (s.comm->cb.group_start && s.comm->cb.group_start(s.comm->cb.ctx)) return fail(IDOCP_E_DEVICE, "callback transport: group_start failed");
  }
  return IDOCP_OK;
}
int xgroupEnd(DistState& s) {
  if (s.comm->nccl) NCCLC(g_rccl.GroupEnd());
  if (s.comm->has_cb) {
    idocp_comm* c = s.comm;
    c->in_group = false;
    const int rc = c->cb.group_end ? c->cb.group_end(c->cb.ctx) : 0;      // completes every transfer posted since group_start
    std::vector<int> kinds;
    kinds.swap(c->pending_recv);
    if (rc) return fail(IDOCP_E_DEVICE, "callback transport: group_end failed");
    for (int kind : kinds) HIPC(hipMemcpyAsync(s.recvb[kind], s.hrecv[kind].data(), s.count[kind] * sizeof(double), hipMemcpyHostToDevice, s.stream));
    if (!kinds.empty()) HIPC(hipStreamSynchronize(s.stream));
  }
  return IDOCP_OK;
}
// Runs `body` between ncclGroupStart and ncclGroupEnd.  The group is ALWAYS closed: a failed send / recv inside an open group would
// otherwise leave every later RCCL call of this thread queued in a group that never ends (and the peers hanging).  The first error wins.
template <typename Body>
int xgrouped(DistState& s, Body body) {
  RC(xgroupStart(s));
  const int rc_body = body();
  const int rc_end = xgroupEnd(s);
  return rc_body ? rc_body : rc_end;
}
// in place on a device buffer of n doubles; op: 0 sum, 1 min
int xallreduce(DistState& s, double* d_buf, size_t n, int op) {
  idocp_comm* c = s.comm;
  if (c->world == 1 && !c->force_collectives) return IDOCP_OK;
  if (c->nccl) { NCCLC(g_rccl.AllReduce(d_buf, d_buf, n, ncclDouble, op == 0 ? ncclSum : ncclMin, c->nccl, s.stream)); return IDOCP_OK; }
  if (c->has_cb) {
    s.hred.resize(n);
    HIPC(hipMemcpyAsync(s.hred.data(), d_buf, n * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    if (c->cb.allreduce(c->cb.ctx, s.hred.data(), (unsigned long)n, op)) return fail(IDOCP_E_DEVICE, "callback transport: allreduce failed");
    HIPC(hipMemcpyAsync(d_buf, s.hred.data(), n * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    return IDOCP_OK;
  }
  LocalHub* hub = c->hub;
  std::vector<double> mine(n);
  HIPC(hipMemcpyAsync(mine.data(), d_buf, n * sizeof(double), hipMemcpyDeviceToHost, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  std::vector<double> out;
  {
    std::unique_lock<std::mutex> lk(hub->m);
    const int gen = hub->reduce_gen;
    hub->reduce_in.push_back(mine);
    if (++hub->reduce_count == hub->world) {
      hub->reduce_out.assign(n, op == 0 ? 0.0 : 1e300);
      for (const auto& v : hub->reduce_in) for (size_t i = 0; i < n; ++i) hub->reduce_out[i] = op == 0 ? hub->reduce_out[i] + v[i] : (v[i] < hub->reduce_out[i] ? v[i] : hub->reduce_out[i]);
      hub->reduce_in.clear(); hub->reduce_count = 0; ++hub->reduce_gen;
      hub->cv.notify_all();
    } else {
      hub->cv.wait(lk, [&] { return hub->reduce_gen != gen; });
    }
    out = hub->reduce_out;
  }
  HIPC(hipMemcpyAsync(d_buf, out.data(), n * sizeof(double), hipMemcpyHostToDevice, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  return IDOCP_OK;
}
int xbroadcast(DistState& s, int kind, int root) {      // sendb[kind] of `root` -> recvb[kind] of everybody
  idocp_comm* c = s.comm;
  if (c->nccl) { NCCLC(g_rccl.Broadcast(s.sendb[kind], s.recvb[kind], s.count[kind], ncclDouble, root, c->nccl, s.stream)); return IDOCP_OK; }
  if (c->has_cb) {
    s.hrecv[kind].resize(s.count[kind]);
    if (c->rank == root) {
      HIPC(hipMemcpyAsync(s.hrecv[kind].data(), s.sendb[kind], s.count[kind] * sizeof(double), hipMemcpyDeviceToHost, s.stream));
      HIPC(hipStreamSynchronize(s.stream));
    }
    if (c->cb.broadcast(c->cb.ctx, s.hrecv[kind].data(), (unsigned long)s.count[kind], root)) return fail(IDOCP_E_DEVICE, "callback transport: broadcast failed");
    HIPC(hipMemcpyAsync(s.recvb[kind], s.hrecv[kind].data(), s.count[kind] * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    return IDOCP_OK;
  }
  if (c->rank == root) {
    for (int p = 0; p < c->world; ++p) if (p != root) RC(xsend(s, kind, p));
    HIPC(hipMemcpyAsync(s.recvb[kind], s.sendb[kind], s.count[kind] * sizeof(double), hipMemcpyDeviceToDevice, s.stream));
  } else {
    RC(xrecv(s, kind, root));
  }
  return IDOCP_OK;
}

int phases(idocp_ocp_t* h, DistState& s, std::initializer_list<int> ids) {
  for (int ph : ids) RC(idocp_parnmpc_launch_phase(h, ph, s.d_q, s.d_v));
  return IDOCP_OK;
}

// state_last -> right, costate_first and aux_first -> left; everything of the previous iterate, so it is simply issued first
int exchangeBoundary(idocp_ocp_t* h, DistState& s) {
  const int rank = s.comm->rank, world = s.comm->world;
  if (world == 1) return IDOCP_OK;
  const bool left = rank > 0, right = rank < world - 1;
  if (right) RC(idocp_parnmpc_export_halo_async(h, STATE_LAST, s.sendb[STATE_LAST]));
  if (left) { RC(idocp_parnmpc_export_halo_async(h, COSTATE_FIRST, s.sendb[COSTATE_FIRST])); RC(idocp_parnmpc_export_halo_async(h, AUX_FIRST, s.sendb[AUX_FIRST])); }
  RC(xgrouped(s, [&]() -> int {
    if (s.comm->nccl || s.comm->has_cb) {
      if (right) RC(xsend(s, STATE_LAST, rank + 1));
      if (left) { RC(xrecv(s, STATE_LAST, rank - 1)); RC(xsend(s, COSTATE_FIRST, rank - 1)); RC(xsend(s, AUX_FIRST, rank - 1)); }
      if (right) { RC(xrecv(s, COSTATE_FIRST, rank + 1)); RC(xrecv(s, AUX_FIRST, rank + 1)); }
    } else {
      // host rendezvous: all sends first (they never block), then the receives
      if (right) RC(xsend(s, STATE_LAST, rank + 1));
      if (left) { RC(xsend(s, COSTATE_FIRST, rank - 1)); RC(xsend(s, AUX_FIRST, rank - 1)); }
      if (left) RC(xrecv(s, STATE_LAST, rank - 1));
      if (right) { RC(xrecv(s, COSTATE_FIRST, rank + 1)); RC(xrecv(s, AUX_FIRST, rank + 1)); }
    }
    return IDOCP_OK;
  }));
  if (left) RC(idocp_parnmpc_import_halo_async(h, STATE_LAST, s.recvb[STATE_LAST]));
  if (right) { RC(idocp_parnmpc_import_halo_async(h, COSTATE_FIRST, s.recvb[COSTATE_FIRST])); RC(idocp_parnmpc_import_halo_async(h, AUX_FIRST, s.recvb[AUX_FIRST])); }
  return IDOCP_OK;
}

}  // namespace

extern "C" {

int idocp_comm_get_unique_id(void* id) {
  if (!id) return IDOCP_E_ARG;
  std::lock_guard<std::mutex> lk(g_rccl_mutex);
  if (!g_rccl.load()) return fail(IDOCP_E_DEVICE, "idocp_comm: cannot load librccl.so");
  ncclUniqueId uid;
  NCCLC(g_rccl.GetUniqueId(&uid));
  static_assert(sizeof(uid) == IDOCP_COMM_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id, &uid, sizeof(uid));
  return IDOCP_OK;
}

int idocp_comm_init_rank(const void* id, int rank, int world, int device, idocp_comm_t** out) {
  if (!id || !out || world < 1 || rank < 0 || rank >= world) return fail(IDOCP_E_ARG, "idocp_comm_init_rank: invalid argument");
  {
    std::lock_guard<std::mutex> lk(g_rccl_mutex);
    if (!g_rccl.load()) return fail(IDOCP_E_DEVICE, "idocp_comm: cannot load librccl.so");
  }
  HIPC(hipSetDevice(device));
  idocp_comm* c = new idocp_comm();
  c->rank = rank; c->world = world; c->device = device;
  ncclUniqueId uid;
  std::memcpy(&uid, id, sizeof(uid));
  const ncclResult_t r = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
  if (r != ncclSuccess) { const std::string msg = std::string("ncclCommInitRank failed: ") + g_rccl.GetErrorString(r); delete c; return fail(IDOCP_E_DEVICE, msg); }
  *out = c;
  return IDOCP_OK;
}

int idocp_comm_init_local(int world, int device, idocp_comm_t** out) {
  if (!out || world < 1) return fail(IDOCP_E_ARG, "idocp_comm_init_local: invalid argument");
  LocalHub* hub = new LocalHub();
  hub->world = world; hub->refs = world;
  for (int r = 0; r < world; ++r) {
    idocp_comm* c = new idocp_comm();
    c->rank = r; c->world = world; c->device = device; c->hub = hub;
    out[r] = c;
  }
  return IDOCP_OK;
}

int idocp_comm_init_callbacks(int rank, int world, int device, const idocp_comm_callbacks_t* cb, idocp_comm_t** out) {
  if (!cb || !out || world < 1 || rank < 0 || rank >= world) return fail(IDOCP_E_ARG, "idocp_comm_init_callbacks: invalid argument");
  if (!cb->send || !cb->recv || !cb->allreduce || !cb->broadcast) return fail(IDOCP_E_ARG, "idocp_comm_init_callbacks: send, recv, allreduce and broadcast are required");
  idocp_comm* c = new idocp_comm();
  c->rank = rank; c->world = world; c->device = device; c->has_cb = true; c->cb = *cb;
  *out = c;
  return IDOCP_OK;
}

void idocp_comm_destroy(idocp_comm_t* c) {
  if (!c) return;
  if (c->has_cb && c->cb.destroy) c->cb.destroy(c->cb.ctx);
  if (c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->hub) {
    bool last;
    { std::lock_guard<std::mutex> lk(c->hub->m); last = (--c->hub->refs == 0); }
    if (last) delete c->hub;
  }
  delete c;
}

// What the communicator itself reports (not what it was asked for): ncclCommCount, ncclCommUserRank, ncclGetVersion -- the bench prints
// them so that a scaling line can be read against the ranks RCCL really connected.  transport: 1 RCCL, 0 the in-process test transport, 2 caller-supplied callbacks.
int idocp_comm_info(const idocp_comm_t* c, int* nranks, int* user_rank, int* rccl_version, int* transport) {
  if (!c) return IDOCP_E_ARG;
  int n = c->world, r = c->rank, ver = 0;
  if (c->nccl) {
    NCCLC(g_rccl.CommCount(c->nccl, &n));
    NCCLC(g_rccl.CommUserRank(c->nccl, &r));
    NCCLC(g_rccl.GetVersion(&ver));
  }
  if (nranks) *nranks = n;
  if (user_rank) *user_rank = r;
  if (rccl_version) *rccl_version = ver;
  if (transport) *transport = c->nccl ? 1 : (c->has_cb ? 2 : 0);
  return IDOCP_OK;
}
int idocp_comm_rank(const idocp_comm_t* c) { return c ? c->rank : -1; }
int idocp_comm_world(const idocp_comm_t* c) { return c ? c->world : -1; }

int idocp_parnmpc_dist_attach(idocp_ocp_t* h, idocp_comm_t* comm) {
  if (!h || !comm) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_attach: null argument");
  if (stateOf(h)) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_attach: the handle is attached already (detach it first)");
  DistState s;
  s.comm = comm;
  s.batch = idocp_ocp_batch(h);
  if (s.batch <= 0) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_attach: not a solver handle");
  s.stream = static_cast<hipStream_t>(idocp_ocp_stream(h));
  HIPC(hipSetDevice(comm->device));
  for (int k = 0; k < NKINDS; ++k) {
    const int sz = idocp_parnmpc_halo_size(k);
    if (sz <= 0) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_attach: unknown halo kind");
    s.count[k] = (size_t)s.batch * sz;
    HIPC(hipMalloc((void**)&s.sendb[k], s.count[k] * sizeof(double)));
    HIPC(hipMalloc((void**)&s.recvb[k], s.count[k] * sizeof(double)));
  }
  HIPC(hipMalloc((void**)&s.d_err2, (size_t)s.batch * sizeof(double)));
  RC(idocp_parnmpc_prev_state(h, &s.d_q, &s.d_v));
  RC(idocp_parnmpc_step_sizes_device(h, &s.d_steps));
  std::lock_guard<std::mutex> lk(g_dist_mutex);
  g_dist[h] = s;
  return IDOCP_OK;
}

int idocp_parnmpc_dist_detach(idocp_ocp_t* h) {
  std::lock_guard<std::mutex> lk(g_dist_mutex);
  auto it = g_dist.find(h);
  if (it == g_dist.end()) return IDOCP_E_ARG;
  for (int k = 0; k < NKINDS; ++k) { (void)hipFree(it->second.sendb[k]); (void)hipFree(it->second.recvb[k]); }
  (void)hipFree(it->second.d_err2);
  g_dist.erase(it);
  return IDOCP_OK;
}

// called by idocp_ocp_destroy: a handle that dies attached must not leave its halo buffers (and a key that a later handle at the
// same address would inherit) behind
void idocp_parnmpc_dist_on_destroy(idocp_ocp_t* h) { (void)idocp_parnmpc_dist_detach(h); }

int idocp_comm_set_force_collectives(idocp_comm_t* c, int on) {
  if (!c) return IDOCP_E_ARG;
  c->force_collectives = on != 0;
  return IDOCP_OK;
}

// Exercises every RCCL entry point the driver uses on THIS rank alone: grouped ncclSend / ncclRecv to itself for every halo kind
// (buffers filled by a device pattern), all-reduce (sum, min) and broadcast, all on the shard's stream, and compares what came back.
// With world == 1 the all-reduce / broadcast results must equal the inputs.  max_abs_diff: largest deviation seen (0 expected).
// world > 1: the same calls across the REAL neighbours, in the grouping the driver uses (exchangeBoundary: everything for the right
// neighbour and everything from the left one in one group, then the other way round), with a pattern that names kind, element and
// SENDING rank, so that a halo that arrives from the wrong peer, in the wrong buffer or shifted is seen; then all-reduce (sum, min) and
// the broadcast from the last rank against their closed forms.  A collective call: every rank of the communicator makes it.
static double selftestPattern(int kind, size_t i, int rank) { return 1.0 + kind + 1e-3 * (double)(i % 9973) + 16.0 * rank; }
static int neighbourSelftest(DistState& s, double* max_abs_diff) {
  idocp_comm* c = s.comm;
  const int rank = c->rank, world = c->world;
  const bool left = rank > 0, right = rank < world - 1;
  double worst = 0.0;
  std::vector<double> buf;
  auto fill = [&](int k) -> int {
    buf.resize(s.count[k]);
    for (size_t i = 0; i < s.count[k]; ++i) buf[i] = selftestPattern(k, i, rank);
    HIPC(hipMemcpyAsync(s.sendb[k], buf.data(), s.count[k] * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPC(hipMemsetAsync(s.recvb[k], 0, s.count[k] * sizeof(double), s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    return IDOCP_OK;
  };
  auto check = [&](int k, int from) -> int {
    buf.resize(s.count[k]);
    HIPC(hipMemcpyAsync(buf.data(), s.recvb[k], s.count[k] * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    for (size_t i = 0; i < s.count[k]; ++i) worst = std::fmax(worst, std::fabs(buf[i] - selftestPattern(k, i, from)));
    return IDOCP_OK;
  };
  for (int dir = 0; dir < 2; ++dir) {                  // 0: to the right neighbour, 1: to the left one
    const bool snd = dir == 0 ? right : left, rcv = dir == 0 ? left : right;
    const int to = dir == 0 ? rank + 1 : rank - 1, from = dir == 0 ? rank - 1 : rank + 1;
    for (int k = 0; k < NKINDS; ++k) RC(fill(k));
    RC(xgrouped(s, [&]() -> int {
      if (snd) for (int k = 0; k < NKINDS; ++k) RC(xsend(s, k, to));
      if (rcv) for (int k = 0; k < NKINDS; ++k) RC(xrecv(s, k, from));
      return IDOCP_OK;
    }));
    if (rcv) for (int k = 0; k < NKINDS; ++k) RC(check(k, from));
  }
  for (int op = 0; op < 2; ++op) {
    RC(fill(STATE_LAST));
    RC(xallreduce(s, s.sendb[STATE_LAST], s.count[STATE_LAST], op));
    buf.resize(s.count[STATE_LAST]);
    HIPC(hipMemcpyAsync(buf.data(), s.sendb[STATE_LAST], buf.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    for (size_t i = 0; i < buf.size(); ++i) {
      const double base = selftestPattern(STATE_LAST, i, 0);
      const double want = op == 0 ? world * base + 16.0 * (0.5 * world * (world - 1)) : base;
      worst = std::fmax(worst, std::fabs(buf[i] - want) / (op == 0 ? world : 1));
    }
  }
  RC(fill(AUX_ALL));
  RC(xbroadcast(s, AUX_ALL, world - 1));
  RC(check(AUX_ALL, world - 1));
  *max_abs_diff = worst;
  return IDOCP_OK;
}

int idocp_parnmpc_dist_transport_selftest(idocp_ocp_t* h, double* max_abs_diff) {
  DistState* sp = stateOf(h);
  if (!sp || !max_abs_diff) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_transport_selftest: attach a communicator first");
  DistState& s = *sp;
  idocp_comm* c = s.comm;
  if (c->world > 1) return neighbourSelftest(s, max_abs_diff);
  if (!c->nccl) return fail(IDOCP_E_UNSUPPORTED, "idocp_parnmpc_dist_transport_selftest: at world size 1 it needs the RCCL transport (idocp_comm_init_rank)");
  double worst = 0.0;
  std::vector<std::vector<double>> sent(NKINDS);
  for (int k = 0; k < NKINDS; ++k) {
    sent[k].resize(s.count[k]);
    for (size_t i = 0; i < s.count[k]; ++i) sent[k][i] = 1.0 + k + 1e-3 * (double)(i % 9973) + c->rank;
    HIPC(hipMemcpyAsync(s.sendb[k], sent[k].data(), s.count[k] * sizeof(double), hipMemcpyHostToDevice, s.stream));
    HIPC(hipMemsetAsync(s.recvb[k], 0, s.count[k] * sizeof(double), s.stream));
  }
  // every halo kind to ourselves, all in ONE group (a send to self only completes next to its receive)
  RC(xgrouped(s, [&]() -> int {
    for (int k = 0; k < NKINDS; ++k) { RC(xsend(s, k, c->rank)); RC(xrecv(s, k, c->rank)); }
    return IDOCP_OK;
  }));
  std::vector<double> got;
  for (int k = 0; k < NKINDS; ++k) {
    got.resize(s.count[k]);
    HIPC(hipMemcpyAsync(got.data(), s.recvb[k], s.count[k] * sizeof(double), hipMemcpyDeviceToHost, s.stream));
    HIPC(hipStreamSynchronize(s.stream));
    for (size_t i = 0; i < s.count[k]; ++i) worst = std::fmax(worst, std::fabs(got[i] - sent[k][i]));
  }
  // collectives: in place on a scratch copy of the first halo buffer
  const bool forced = c->force_collectives;
  c->force_collectives = true;
  int rc = IDOCP_OK;
  for (int op = 0; op < 2 && !rc; ++op) {
    rc = xallreduce(s, s.sendb[STATE_LAST], s.count[STATE_LAST], op);
    if (rc) break;
    got.resize(s.count[STATE_LAST]);
    if (hipMemcpyAsync(got.data(), s.sendb[STATE_LAST], got.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream) != hipSuccess ||
        hipStreamSynchronize(s.stream) != hipSuccess) { rc = fail(IDOCP_E_DEVICE, "selftest: copy back failed"); break; }
    if (c->world == 1) for (size_t i = 0; i < got.size(); ++i) worst = std::fmax(worst, std::fabs(got[i] - sent[STATE_LAST][i]));
  }
  c->force_collectives = forced;
  RC(rc);
  HIPC(hipMemsetAsync(s.recvb[AUX_ALL], 0, s.count[AUX_ALL] * sizeof(double), s.stream));
  RC(xbroadcast(s, AUX_ALL, c->world - 1));
  got.resize(s.count[AUX_ALL]);
  HIPC(hipMemcpyAsync(got.data(), s.recvb[AUX_ALL], got.size() * sizeof(double), hipMemcpyDeviceToHost, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  if (c->world == 1) for (size_t i = 0; i < got.size(); ++i) worst = std::fmax(worst, std::fabs(got[i] - sent[AUX_ALL][i]));
  *max_abs_diff = worst;
  return IDOCP_OK;
}

// rank 0: the measured state q[batch][nq], v[batch][nv] (host buffers); the other ranks receive theirs through the halos
int idocp_parnmpc_dist_set_initial_state(idocp_ocp_t* h, const double* q, const double* v, int nq, int nv) {
  DistState* s = stateOf(h);
  if (!s || !q || !v) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_set_initial_state: attach a communicator first");
  int mq = 0, mv = 0;
  RC(idocp_ocp_state_dims(h, &mq, &mv));
  if (nq != mq || nv != mv) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist_set_initial_state: nq / nv do not match the model");
  HIPC(hipMemcpyAsync(s->d_q, q, sizeof(double) * s->batch * nq, hipMemcpyHostToDevice, s->stream));
  HIPC(hipMemcpyAsync(s->d_v, v, sizeof(double) * s->batch * nv, hipMemcpyHostToDevice, s->stream));
  HIPC(hipStreamSynchronize(s->stream));
  return IDOCP_OK;
}

// ParNMPCSolver::initBackwardCorrection: aux_mat = terminal cost Hessian at the LAST stage of the horizon -> computed by the last
// rank, broadcast to everybody
int idocp_parnmpc_dist_init_backward_correction(idocp_ocp_t* h, double t) {
  DistState* s = stateOf(h);
  if (!s) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  RC(idocp_parnmpc_init_backward_correction(h, t));
  if (s->comm->world == 1 && !(s->comm->nccl && s->comm->force_collectives)) return IDOCP_OK;
  RC(idocp_parnmpc_export_halo_async(h, AUX_ALL, s->sendb[AUX_ALL]));
  RC(xbroadcast(*s, AUX_ALL, s->comm->world - 1));
  RC(idocp_parnmpc_import_halo_async(h, AUX_ALL, s->recvb[AUX_ALL]));
  return IDOCP_OK;
}

// One iteration (ParNMPCSolver::updateSolution, parnmpc_solver.cpp:73-103) of the sharded horizon.  Returns once everything is
// ENQUEUED (RCCL transport): synchronise with idocp_ocp_synchronize before reading results on the host.
int idocp_parnmpc_dist_update_solution(idocp_ocp_t* h, double t) {
  DistState* sp = stateOf(h);
  if (!sp) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  DistState& s = *sp;
  const int rank = s.comm->rank, world = s.comm->world;
  const bool left = rank > 0, right = rank < world - 1;
  RC(idocp_parnmpc_discretize(h, t));
  RC(exchangeBoundary(h, s));
  RC(phases(h, s, {0, 1, 2}));                                   // linearise, condense, KKT inverse + coarse update
  if (right) { RC(xrecv(s, BWD_FIRST, rank + 1)); RC(idocp_parnmpc_import_halo_async(h, BWD_FIRST, s.recvb[BWD_FIRST])); }
  RC(phases(h, s, {3}));                                         // backward serial sweep, right -> left across the ranks
  if (left) { RC(idocp_parnmpc_export_halo_async(h, BWD_FIRST, s.sendb[BWD_FIRST])); RC(xsend(s, BWD_FIRST, rank - 1)); }
  RC(phases(h, s, {4}));                                         // backward parallel: overlaps the left neighbours' serial sweeps
  if (left) { RC(xrecv(s, FWD_LAST, rank - 1)); RC(idocp_parnmpc_import_halo_async(h, FWD_LAST, s.recvb[FWD_LAST])); }
  RC(phases(h, s, {5}));                                         // forward serial sweep, left -> right
  if (right) { RC(idocp_parnmpc_export_halo_async(h, FWD_LAST, s.sendb[FWD_LAST])); RC(xsend(s, FWD_LAST, rank + 1)); }
  RC(phases(h, s, {6, 7, 8}));                                   // forward parallel, expansion, local step sizes
  RC(xallreduce(s, s.d_steps, (size_t)s.batch * 2, 1));          // min over the horizon
  RC(phases(h, s, {9}));                                         // dual expansion + integration
  return IDOCP_OK;
}

// ---- filter line search across the shards ----
// One probe of LineSearch::computeCostAndViolation (src/line_search/line_search.cpp:199-301) needs, per shard, the trial iterate of the state
// in front of its first stage -- the left neighbour's trial (q, v) of its last stage -- and the sums over ALL shards.  The handle calls
// these two hooks from inside its probe (ocp_capi.hip, lineSearchEvalO); the state_last buffers are free by then (the boundary exchange of
// the iteration is long done).
static int lsPre(idocp_ocp_t* h) {
  DistState* sp = stateOf(h);
  if (!sp) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  DistState& s = *sp;
  const int rank = s.comm->rank, world = s.comm->world;
  if (world == 1) return IDOCP_OK;
  const bool left = rank > 0, right = rank < world - 1;
  if (right) RC(idocp_parnmpc_trial_halo_async(h, 0, s.sendb[STATE_LAST]));
  RC(xgrouped(s, [&]() -> int {
    if (right) RC(xsend(s, STATE_LAST, rank + 1));
    if (left) RC(xrecv(s, STATE_LAST, rank - 1));
    return IDOCP_OK;
  }));
  if (left) RC(idocp_parnmpc_trial_halo_async(h, 1, s.recvb[STATE_LAST]));
  return IDOCP_OK;
}
static int lsPost(idocp_ocp_t* h) {
  DistState* sp = stateOf(h);
  if (!sp) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  double* d_merit = nullptr;
  RC(idocp_parnmpc_merit_device(h, &d_merit));
  return xallreduce(*sp, d_merit, (size_t)sp->batch * 2, 0);
}

// ParNMPCSolver::updateSolution(t, q, v, true) of the sharded horizon: the iteration up to the step sizes, the filter line search on the
// primal step with every probe evaluated collectively, the integration.  Every rank runs the same filter on the same all-reduced sums.
int idocp_parnmpc_dist_update_solution_ls(idocp_ocp_t* h, double t) {
  DistState* sp = stateOf(h);
  if (!sp) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  DistState& s = *sp;
  const int rank = s.comm->rank, world = s.comm->world;
  const bool left = rank > 0, right = rank < world - 1;
  // the hooks live on the handle for the duration of THIS call only: left installed, a later single-handle line search on the shard
  // (idocp_parnmpc_update_solution(.., 1), idocp_ocp_line_search_eval) would issue RCCL calls the other ranks do not match
  struct HookGuard {
    idocp_ocp_t* h;
    ~HookGuard() { idocp_parnmpc_set_line_search_hooks(h, nullptr, nullptr); }
  } guard{h};
  RC(idocp_parnmpc_set_line_search_hooks(h, lsPre, lsPost));
  RC(idocp_parnmpc_discretize(h, t));
  RC(exchangeBoundary(h, s));
  RC(phases(h, s, {0, 1, 2}));
  if (right) { RC(xrecv(s, BWD_FIRST, rank + 1)); RC(idocp_parnmpc_import_halo_async(h, BWD_FIRST, s.recvb[BWD_FIRST])); }
  RC(phases(h, s, {3}));
  if (left) { RC(idocp_parnmpc_export_halo_async(h, BWD_FIRST, s.sendb[BWD_FIRST])); RC(xsend(s, BWD_FIRST, rank - 1)); }
  RC(phases(h, s, {4}));
  if (left) { RC(xrecv(s, FWD_LAST, rank - 1)); RC(idocp_parnmpc_import_halo_async(h, FWD_LAST, s.recvb[FWD_LAST])); }
  RC(phases(h, s, {5}));
  if (right) { RC(idocp_parnmpc_export_halo_async(h, FWD_LAST, s.sendb[FWD_LAST])); RC(xsend(s, FWD_LAST, rank + 1)); }
  RC(phases(h, s, {6, 7, 8}));
  RC(xallreduce(s, s.d_steps, (size_t)s.batch * 2, 1));
  RC(idocp_parnmpc_line_search(h));                              // collective: lsPre / lsPost inside every probe
  RC(phases(h, s, {9}));
  return IDOCP_OK;
}

// ParNMPCSolver::computeKKTResidual + KKTError of the whole horizon: sqrt(sum over ranks of the shards' squared errors)
int idocp_parnmpc_dist_kkt_error(idocp_ocp_t* h, double t, double* kkt_error) {
  DistState* sp = stateOf(h);
  if (!sp || !kkt_error) return fail(IDOCP_E_ARG, "idocp_parnmpc_dist: attach a communicator first");
  DistState& s = *sp;
  RC(idocp_parnmpc_discretize(h, t));
  RC(exchangeBoundary(h, s));
  RC(idocp_parnmpc_kkt_error_squared_device(h, t, s.d_err2));
  RC(xallreduce(s, s.d_err2, (size_t)s.batch, 0));
  HIPC(hipMemcpyAsync(kkt_error, s.d_err2, sizeof(double) * s.batch, hipMemcpyDeviceToHost, s.stream));
  HIPC(hipStreamSynchronize(s.stream));
  for (int b = 0; b < s.batch; ++b) kkt_error[b] = std::sqrt(kkt_error[b]);
  return IDOCP_OK;
}

}  // extern "C"
