// ORACLE -- TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// Flat C interface of the CPU restatement so tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg can drive it through ctypes.  Nothing in the
// product (idocp_amd/, include/) links or loads this library.
#include <chrono>
#include <cstring>
#include <string>

#include "idocp_hip.h"
#include "rbd.hpp"
#include "unocp.hpp"

using namespace oracle;

// copy between the caller's FP64 buffers and the restatement's scalar (`nbytes` counts FP64 bytes, as in the memcpy it replaces)
template <typename A, typename Bt>
static inline void xcpy(A* dst, const Bt* src, size_t nbytes) { const size_t n = nbytes / sizeof(double); for (size_t i = 0; i < n; ++i) dst[i] = (A)src[i]; }
static Mat toVec(const double* p, int n) { Mat v(n); xcpy(v.d.data(), p, sizeof(double) * n); return v; }

static std::string g_oracle_error;
extern "C" {
const char* oracle_last_error(void) { return g_oracle_error.c_str(); }

int oracle_rnea(const idocp_model_t* m, const double* q, const double* v, const double* a,
                const double* fext_local /* [ncontacts][3] or NULL */, int gravity, double* tau) {
  Robot r(*m);
  if (fext_local && m->ncontacts > 0) {
    std::vector<bool> act(m->ncontacts, true); std::vector<Mat> f;
    for (int c = 0; c < m->ncontacts; ++c) f.push_back(toVec(fext_local + 3 * c, 3));
    r.setContactForces(act, f);
  }
  Mat t;
  r.RNEA(toVec(q, m->nq), toVec(v, m->nv), toVec(a, m->nv), t, gravity != 0);
  xcpy(tau, t.d.data(), sizeof(double) * m->nv);
  return 0;
}

int oracle_rnea_derivatives(const idocp_model_t* m, const double* q, const double* v, const double* a,
                            const double* fext_local, int gravity, double* dq, double* dv, double* da) {
  Robot r(*m);
  if (fext_local && m->ncontacts > 0) {
    std::vector<bool> act(m->ncontacts, true); std::vector<Mat> f;
    for (int c = 0; c < m->ncontacts; ++c) f.push_back(toVec(fext_local + 3 * c, 3));
    r.setContactForces(act, f);
  }
  Mat Dq, Dv, Da;
  r.RNEADerivatives(toVec(q, m->nq), toVec(v, m->nv), toVec(a, m->nv), Dq, Dv, Da, gravity != 0);
  const size_t n = sizeof(double) * m->nv * m->nv;
  xcpy(dq, Dq.d.data(), n); xcpy(dv, Dv.d.data(), n); xcpy(da, Da.d.data(), n);
  return 0;
}

// Contact-path rigid-body routines for golden tests.  All outputs col-major.
// C[3nc], dCdq/dCdv/dCda[3nc x nv], frames: p[nc][3], R[nc][9], v[nc][6], a[nc][6],
// vdq/adq/adv/ada [nc][6 x nv], MJtJinv[(nv+3nc)^2]
int oracle_contact_kinematics(const idocp_model_t* m, const double* q, const double* v, const double* a,
                              const double* contact_points, double time_step, double* C, double* dCdq, double* dCdv,
                              double* dCda, double* fp, double* fR, double* fv, double* fa, double* vdq, double* adq,
                              double* adv, double* ada, double* MJtJinv) {
  Robot r(*m);
  const int nv = m->nv, nc = m->ncontacts;
  Mat Q = toVec(q, m->nq), V = toVec(v, nv), A = toVec(a, nv);
  r.updateKinematics(Q, V, A);
  std::vector<bool> act(nc, true); std::vector<Mat> cp;
  for (int c = 0; c < nc; ++c) cp.push_back(toVec(contact_points + 3 * c, 3));
  Mat Cm, Dq, Dv, Da;
  r.computeBaumgarteResidual(act, time_step, cp, Cm);
  r.computeBaumgarteDerivatives(act, time_step, Dq, Dv, Da);
  xcpy(C, Cm.d.data(), sizeof(double) * 3 * nc);
  xcpy(dCdq, Dq.d.data(), sizeof(double) * 3 * nc * nv);
  xcpy(dCdv, Dv.d.data(), sizeof(double) * 3 * nc * nv);
  xcpy(dCda, Da.d.data(), sizeof(double) * 3 * nc * nv);
  for (int c = 0; c < nc; ++c) {
    real tp[3], tR[9], tv[6], ta[6];
    r.contactFrame(c, tp, tR, tv, ta);
    xcpy(fp + 3 * c, tp, sizeof(double) * 3); xcpy(fR + 9 * c, tR, sizeof(double) * 9); xcpy(fv + 6 * c, tv, sizeof(double) * 6); xcpy(fa + 6 * c, ta, sizeof(double) * 6);
    Mat a1, a2, a3, a4;
    r.frameDerivatives(c, a1, a2, a3, a4);
    const size_t n = (size_t)6 * nv;
    xcpy(vdq + c * n, a1.d.data(), sizeof(double) * n); xcpy(adq + c * n, a2.d.data(), sizeof(double) * n);
    xcpy(adv + c * n, a3.d.data(), sizeof(double) * n); xcpy(ada + c * n, a4.d.data(), sizeof(double) * n);
  }
  if (MJtJinv) {
    Mat dq, dv, Mm, out;
    r.RNEADerivatives(Q, V, A, dq, dv, Mm);
    Robot::computeMJtJinv(Mm, Da, out);
    xcpy(MJtJinv, out.d.data(), sizeof(double) * out.size());
  }
  return 0;
}

// ForwardSwitchingConstraint terms for all contacts (forward_switching_constraint.hxx:27-66), col-major:
// P[3nc] = foot positions at q (+) ((dt1+dt2) v + dt1 dt2 a) minus points; Phiq, Phiv, Phia [3nc x nv]
int oracle_switching_terms(const idocp_model_t* m, const double* q, const double* v, const double* a, double dt1, double dt2,
                           const double* points, double* Pout, double* Phiq, double* Phiv, double* Phia) {
  Robot r(*m);
  const int nv = m->nv, nc = m->ncontacts;
  Mat Q = toVec(q, m->nq), V = toVec(v, nv), A = toVec(a, nv);
  Mat dq = (dt1 + dt2) * V + (dt1 * dt2) * A, q2;
  r.integrateConfiguration(Q, dq, 1.0, q2);
  r.updateKinematics(q2, Mat(nv), Mat(nv));
  std::vector<bool> act(nc, true); std::vector<Mat> cp;
  for (int c = 0; c < nc; ++c) cp.push_back(toVec(points + 3 * c, 3));
  Mat Pm, Pq, Jq, Jv;
  r.computeContactResidual(act, cp, Pm);
  r.computeContactDerivative(act, Pq);
  r.dIntegratedConfiguration(Q, dq, Jq);
  r.dIntegratedVelocity(Q, dq, Jv);
  Mat a1 = Pq * Jq, a2 = (dt1 + dt2) * (Pq * Jv), a3 = (dt1 * dt2) * (Pq * Jv);
  xcpy(Pout, Pm.d.data(), sizeof(double) * 3 * nc);
  xcpy(Phiq, a1.d.data(), sizeof(double) * 3 * nc * nv);
  xcpy(Phiv, a2.d.data(), sizeof(double) * 3 * nc * nv);
  xcpy(Phia, a3.d.data(), sizeof(double) * 3 * nc * nv);
  return 0;
}

// Impulse-stage rigid-body terms (impulse_dynamics_forward_euler.hxx:40-58), col-major, all contacts active:
// ImD[nv] = rnea(q, 0, dv) without gravity with the impulse forces f[nc][3]; dImDdq, dImDddv [nv x nv];
// C[3nc] = LOCAL linear velocity of the feet at (q, v + dv); dCdq, dCdv [3nc x nv]
int oracle_impulse_terms(const idocp_model_t* m, const double* q, const double* v, const double* dv, const double* f, double* ImD,
                         double* dImDdq, double* dImDddv, double* C, double* dCdq, double* dCdv) {
  Robot r(*m);
  const int nv = m->nv, nc = m->ncontacts;
  Mat Q = toVec(q, m->nq), V = toVec(v, nv), DV = toVec(dv, nv), zero(nv);
  std::vector<bool> act(nc, true); std::vector<Mat> fs;
  for (int c = 0; c < nc; ++c) fs.push_back(toVec(f + 3 * c, 3));
  r.updateKinematics(Q, V + DV, zero);
  r.setContactForces(act, fs);
  Mat tau, dq, dvv, da, Cm, Cq, Cv;
  r.RNEA(Q, zero, DV, tau, false);
  r.RNEADerivatives(Q, zero, DV, dq, dvv, da, false);
  r.computeImpulseVelocityResidual(act, Cm);
  r.computeImpulseVelocityDerivatives(act, Cq, Cv);
  xcpy(ImD, tau.d.data(), sizeof(double) * nv);
  xcpy(dImDdq, dq.d.data(), sizeof(double) * nv * nv);
  xcpy(dImDddv, da.d.data(), sizeof(double) * nv * nv);
  xcpy(C, Cm.d.data(), sizeof(double) * 3 * nc);
  xcpy(dCdq, Cq.d.data(), sizeof(double) * 3 * nc * nv);
  xcpy(dCdv, Cv.d.data(), sizeof(double) * 3 * nc * nv);
  return 0;
}

// Lie operations: q_int = q (+) dv ; diff = q1 (-) q ; J0 = d diff / d q (ARG0), J1 = d diff / d q1 (ARG1)
int oracle_lie_ops(const idocp_model_t* m, const double* q, const double* q1, const double* dv, double* q_int,
                   double* diff, double* J0, double* J1) {
  Robot r(*m);
  Mat Q = toVec(q, m->nq), Q1 = toVec(q1, m->nq), DV = toVec(dv, m->nv), out, d, j0, j1;
  r.integrateConfiguration(Q, DV, 1.0, out);
  r.subtractConfiguration(Q1, Q, d);
  r.dSubtractdConfigurationMinus(Q1, Q, j0);
  r.dSubtractdConfigurationPlus(Q1, Q, j1);
  xcpy(q_int, out.d.data(), sizeof(double) * m->nq);
  xcpy(diff, d.d.data(), sizeof(double) * m->nv);
  xcpy(J0, j0.d.data(), sizeof(double) * m->nv * m->nv);
  xcpy(J1, j1.d.data(), sizeof(double) * m->nv * m->nv);
  return 0;
}

// ---- UnOCPSolver ---------------------------------------------------------
void* oracle_unocp_create(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k,
                          double T, int N) {
  try { return new UnOCPSolver(*m, *c, *k, T, N); } catch (...) { return nullptr; }
}
void oracle_unocp_destroy(void* h) { delete static_cast<UnOCPSolver*>(h); }

int oracle_unocp_set_solution(void* h, const char* name, const double* value) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  const std::string n(name);
  try { s->setSolution(n, toVec(value, n == "q" ? s->robot.dimq() : s->robot.dimv())); } catch (...) { return -1; }
  return 0;
}
int oracle_unocp_init_constraints(void* h) { static_cast<UnOCPSolver*>(h)->initConstraints(); return 0; }

int oracle_unocp_update_solution(void* h, double t, const double* q, const double* v) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); } catch (...) { return 1; }
  return 0;
}
// updateSolution(t, q, v, line_search = true) and the pieces of UnLineSearch
int oracle_unocp_update_solution_ls(void* h, double t, const double* q, const double* v) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()), true); } catch (...) { return 1; }
  return 0;
}
void oracle_unocp_clear_line_search_filter(void* h) { static_cast<UnOCPSolver*>(h)->line_search.filter.clear(); }
int oracle_unocp_cost_and_violation(void* h, double alpha, double* out) {
  const auto cv = static_cast<UnOCPSolver*>(h)->costAndViolation(alpha);
  out[0] = cv.first; out[1] = cv.second; return 0;
}
// staged execution (kernel-level parity): 0 linearize, 1 backward+forward Riccati, 2 direction, 3 integrate
int oracle_unocp_stage(void* h, int what, double t, const double* q, const double* v) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
  try {
    if (what == 0) s->linearizeOCP(t, Q);
    else if (what == 1) { s->backwardRiccatiRecursion(); s->forwardRiccatiRecursion(Q, V); }
    else if (what == 2) s->computeDirection();
    else if (what == 3) s->integrate();
    else return -1;
  } catch (...) { return 1; }
  return 0;
}
int oracle_unocp_compute_kkt_residual(void* h, double t, const double* q, const double* v) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return 0;
}
int oracle_unocp_is_current_solution_feasible(void* h) { return static_cast<UnOCPSolver*>(h)->isCurrentSolutionFeasible(); }
double oracle_unocp_kkt_error(void* h) { return static_cast<UnOCPSolver*>(h)->KKTError(); }

static const Mat* solField(const SplitSolution& s, const std::string& n) {
  if (n == "q") return &s.q; if (n == "v") return &s.v; if (n == "a") return &s.a; if (n == "u") return &s.u;
  if (n == "lmd") return &s.lmd; if (n == "gmm") return &s.gmm; if (n == "beta") return &s.beta;
  return nullptr;
}
static const Mat* dirField(const SplitDirection& d, const std::string& n) {
  if (n == "dq") return &d.dq; if (n == "dv") return &d.dv; if (n == "da") return &d.da; if (n == "du") return &d.du;
  if (n == "dlmd") return &d.dlmd; if (n == "dgmm") return &d.dgmm; if (n == "dbeta") return &d.dbeta;
  return nullptr;
}
// out[(N+1)][dim]
int oracle_unocp_get_solution(void* h, const char* name, double* out) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  for (int i = 0; i <= s->N(); ++i) {
    const Mat* f = solField(s->s[i], name);
    if (!f) return -1;
    xcpy(out + (size_t)i * f->size(), f->d.data(), sizeof(double) * f->size());
  }
  return 0;
}
int oracle_unocp_get_direction(void* h, const char* name, double* out) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  for (int i = 0; i <= s->N(); ++i) {
    const Mat* f = dirField(s->d[i], name);
    if (!f) return -1;
    xcpy(out + (size_t)i * f->size(), f->d.data(), sizeof(double) * f->size());
  }
  return 0;
}
int oracle_unocp_get_step_sizes(void* h, double* primal, double* dual) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  *primal = s->primal_step_size; *dual = s->dual_step_size; return 0;
}
// P[N+1][2nv*2nv] col-major [Pqq Pqv; Pvq Pvv], s[N+1][2nv], K[N][nv*2nv], k[N][nv]
int oracle_unocp_get_riccati(void* h, double* P, double* sv, double* K, double* k) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  const int nv = s->robot.dimv(), nx = 2 * nv;
  for (int i = 0; i <= s->N(); ++i) {
    const SplitRiccatiFactorization& r = s->riccati[i];
    if (P) {
      Mat Pm(nx, nx);
      Pm.setBlock(0, 0, r.Pqq); Pm.setBlock(0, nv, r.Pqv); Pm.setBlock(nv, 0, r.Pqv.t()); Pm.setBlock(nv, nv, r.Pvv);
      xcpy(P + (size_t)i * nx * nx, Pm.d.data(), sizeof(double) * nx * nx);
    }
    if (sv) { xcpy(sv + (size_t)i * nx, r.sq.d.data(), sizeof(double) * nv); xcpy(sv + (size_t)i * nx + nv, r.sv.d.data(), sizeof(double) * nv); }
    if (i < s->N()) {
      if (K) xcpy(K + (size_t)i * nv * nx, s->K[i].d.data(), sizeof(double) * nv * nx);
      if (k) xcpy(k + (size_t)i * nv, s->k[i].d.data(), sizeof(double) * nv);
    }
  }
  return 0;
}
// ---- UnParNMPCSolver -----------------------------------------------------
int oracle_unocp_set_num_threads(void* h, int n) { static_cast<UnOCPSolver*>(h)->setNumThreads(n); return 0; }
int oracle_openmp_enabled(void) {
#ifdef _OPENMP
  return 1;
#else
  return 0;
#endif
}
// ---- exact FLOP counter (flops.hpp) -- only the liboracle_flops.so build counts --------------
//   oracle_flops_enabled()            1 in the counting build
//   oracle_flops_shape(&nregion, &nkind)
//   oracle_flops_region_name(i) / oracle_flops_kind_name(k)
//   oracle_flops_reset(); oracle_flops_get(out[nregion * nkind])      counts since the last reset, row-major [region][kind]
int oracle_flops_enabled(void) {
#ifdef ORACLE_COUNT_FLOPS
  return 1;
#else
  return 0;
#endif
}
static const char* const kFlopRegionNames[] = {"other", "kinematics", "rnea", "rnea_derivatives", "baumgarte_contact", "mjtjinv", "lie", "cost_constraints_multipliers",
                                               "condense", "switching_constraint", "unconstrained_dynamics", "riccati_backward", "riccati_forward", "expand_direction",
                                               "integrate", "parnmpc_kkt_inverse", "parnmpc_corrections"};
static const char* const kFlopKindNames[] = {"add", "mul", "div", "sqrt", "transcendental"};
void oracle_flops_shape(int* nregion, int* nkind) { *nregion = (int)(sizeof(kFlopRegionNames) / sizeof(kFlopRegionNames[0])); *nkind = 5; }
const char* oracle_flops_region_name(int i) { return kFlopRegionNames[i]; }
const char* oracle_flops_kind_name(int k) { return kFlopKindNames[k]; }
#ifdef ORACLE_COUNT_FLOPS
}      // extern "C"
namespace oracle {
thread_local int flop_region = R_OTHER;
unsigned long long flop_count[R_NREGION][F_NKIND];
}
extern "C" {
static_assert(sizeof(kFlopRegionNames) / sizeof(kFlopRegionNames[0]) == oracle::R_NREGION && oracle::F_NKIND == 5, "names of the FLOP regions / kinds");
void oracle_flops_reset(void) { std::memset(oracle::flop_count, 0, sizeof(oracle::flop_count)); oracle::flop_region = oracle::R_OTHER; }
void oracle_flops_get(unsigned long long* out) { std::memcpy(out, oracle::flop_count, sizeof(oracle::flop_count)); }
#else
void oracle_flops_reset(void) {}
void oracle_flops_get(unsigned long long* out) { (void)out; }
#endif
// TaskSpace*Cost: references of stages 0 .. N, refs[N + 1][12] (rotation row-major, position)
int oracle_unocp_set_task_refs(void* h, const double* refs) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  if (s->cost.task_dim == 0) return 1;
  s->setTaskRefs(refs);
  return 0;
}
// cost (no dt), gradient[nv], Gauss-Newton Hessian[nv x nv] (row-major) of the task term of stage i at configuration q
int oracle_unocp_task_terms(void* h, int stage, const double* q, double* c, double* g, double* H) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  const int nv = s->robot.dimv();
  try {
    real cc; Mat gg, HH;
    s->taskTerms(stage, toVec(q, nv), cc, gg, HH);
    *c = (double)cc;
    for (int r = 0; r < nv; ++r) { g[r] = (double)gg[r]; for (int k2 = 0; k2 < nv; ++k2) H[r * nv + k2] = (double)HH(r, k2); }
  } catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
void* oracle_unparnmpc_create(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k, double T, int N) {
  try { return new UnParNMPCSolver(*m, *c, *k, T, N); } catch (...) { return nullptr; }
}
void oracle_unparnmpc_destroy(void* h) { delete static_cast<UnParNMPCSolver*>(h); }
int oracle_unparnmpc_set_solution(void* h, const char* name, const double* value) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const std::string n(name);
  try { s->setSolution(n, toVec(value, n == "q" ? s->robot.dimq() : s->robot.dimv())); } catch (...) { return -1; }
  return 0;
}
int oracle_unparnmpc_init(void* h, double t) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  s->initConstraints(); s->initBackwardCorrection(t); return 0;
}
int oracle_unparnmpc_update_solution(void* h, double t, const double* q, const double* v) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); } catch (...) { return 1; }
  return 0;
}
int oracle_unparnmpc_update_solution_ls(void* h, double t, const double* q, const double* v) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()), true); } catch (...) { return 1; }
  return 0;
}
void oracle_unparnmpc_clear_line_search_filter(void* h) { static_cast<UnParNMPCSolver*>(h)->line_search.filter.clear(); }
int oracle_unparnmpc_cost_and_violation(void* h, double alpha, const double* q, const double* v, double* out) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const auto cv = s->costAndViolation(alpha, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  out[0] = cv.first; out[1] = cv.second; return 0;
}
// staged execution: 0 coarse update, 1 backward serial, 2 backward parallel, 3 forward serial, 4 forward parallel
// (+ direction and step sizes), 5 integrate
int oracle_unparnmpc_stage(void* h, int what, double t, const double* q, const double* v) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  try {
    if (what == 0) s->coarseUpdate(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
    else if (what == 1) s->backwardCorrectionSerial();
    else if (what == 2) s->backwardCorrectionParallel();
    else if (what == 3) s->forwardCorrectionSerial();
    else if (what == 4) s->forwardCorrectionParallel();
    else if (what == 5) s->integrate();
    else return -1;
  } catch (...) { return 1; }
  return 0;
}
double oracle_unparnmpc_kkt_error(void* h, double t, const double* q, const double* v) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return s->KKTError();
}
// horizon shard of UnParNMPC (test twin of idocp_unparnmpc_create_shard + the halo entry points; kinds as in include/idocp_hip.h)
int oracle_unparnmpc_set_slice(void* h, int lo, int hi) { static_cast<UnParNMPCSolver*>(h)->setSlice(lo, hi); return 0; }
int oracle_unparnmpc_export(void* h, int kind, double* out) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const int nv = s->robot.dimv(), lo = s->lo(), hi = s->hi();
  auto put2 = [&](const Mat& a, const Mat& b) { xcpy(out, a.d.data(), sizeof(double) * nv); xcpy(out + nv, b.d.data(), sizeof(double) * nv); };
  switch (kind) {
    case 0: put2(s->s[hi - 1].q, s->s[hi - 1].v); break;
    case 1: put2(s->s[lo].lmd, s->s[lo].gmm); break;
    case 2: xcpy(out, s->aux_mat[lo].d.data(), sizeof(double) * 4 * nv * nv); break;
    case 3: put2(s->s_new[lo].lmd, s->s_new[lo].gmm); break;
    case 4: put2(s->s_new[hi - 1].q, s->s_new[hi - 1].v); break;
    default: return -1;
  }
  return 0;
}
int oracle_unparnmpc_import(void* h, int kind, const double* in) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const int nv = s->robot.dimv(), lo = s->lo(), hi = s->hi();
  auto get2 = [&](Mat& a, Mat& b) { a = toVec(in, nv); b = toVec(in + nv, nv); };
  switch (kind) {
    case 0: get2(s->s[lo - 1].q, s->s[lo - 1].v); break;
    case 1: get2(s->s[hi].lmd, s->s[hi].gmm); break;
    case 2: xcpy(s->aux_mat[hi].d.data(), in, sizeof(double) * 4 * nv * nv); break;
    case 3: get2(s->s_new[hi].lmd, s->s_new[hi].gmm); break;
    case 4: get2(s->s_new[lo - 1].q, s->s_new[lo - 1].v); break;
    default: return -1;
  }
  return 0;
}
void oracle_unparnmpc_set_step_sizes(void* h, double primal, double dual) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  s->primal_step_size = primal; s->dual_step_size = dual;
}
double oracle_unparnmpc_kkt_error_squared(void* h, double t, const double* q, const double* v) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return s->KKTErrorSquared();
}
// `iters` updateSolution calls at fixed (t, q, v); returns total seconds (the sweeps are not timed separately)
double oracle_unparnmpc_bench(void* h, double t, const double* q, const double* v, int iters, double* sweep_seconds) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) s->updateSolution(t, Q, V);
  if (sweep_seconds) *sweep_seconds = 0.0;
  return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}
int oracle_unparnmpc_is_current_solution_feasible(void* h) { return static_cast<UnParNMPCSolver*>(h)->isCurrentSolutionFeasible(); }
// out[N][dim]; names: the solution fields, "d" + field for the direction, "new_" + field for the coarse / corrected iterate
int oracle_unparnmpc_get(void* h, const char* name, double* out) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const std::string n(name);
  for (int i = 0; i < s->N(); ++i) {
    const Mat* f = n.rfind("new_", 0) == 0 ? solField(s->s_new[i], n.substr(4)) : (n[0] == 'd' ? dirField(s->d[i], n) : solField(s->s[i], n));
    if (!f) return -1;
    xcpy(out + (size_t)i * f->size(), f->d.data(), sizeof(double) * f->size());
  }
  return 0;
}
int oracle_unparnmpc_get_step_sizes(void* h, double* primal, double* dual) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  *primal = s->primal_step_size; *dual = s->dual_step_size; return 0;
}
// per stage: KKT inverse [N][5nv*5nv] (col-major), aux matrix [N][2nv*2nv]
int oracle_unparnmpc_get_matrices(void* h, double* kkt_inv, double* aux) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const int nv = s->robot.dimv(), nk = 5 * nv, nx = 2 * nv;
  for (int i = 0; i < s->N(); ++i) {
    if (kkt_inv) xcpy(kkt_inv + (size_t)i * nk * nk, s->kkt_inv[i].d.data(), sizeof(double) * nk * nk);
    if (aux) xcpy(aux + (size_t)i * nx * nx, s->aux_mat[i].d.data(), sizeof(double) * nx * nx);
  }
  return 0;
}
// slack / dual [N][dimc]; rows of components that are not valid at a stage are 0
int oracle_unparnmpc_get_constraint_data(void* h, double* slack, double* dual) {
  UnParNMPCSolver* s = static_cast<UnParNMPCSolver*>(h);
  const int dimc = s->constraints.dimc_total();
  for (int i = 0; i < s->N(); ++i) {
    int off = 0;
    for (size_t c = 0; c < s->constraints.components.size(); ++c) {
      const ConstraintComponentData& data = s->ocp[i].cdata.data[c];
      const int n = s->constraints.components[c].lim.size();
      const bool valid = s->constraints.valid(s->constraints.components[c], i + 1);
      for (int r = 0; r < n; ++r) {
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? (double)data.slack[r] : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? (double)data.dual[r] : 0.0;
      }
      off += n;
    }
  }
  return 0;
}

int oracle_unocp_dimc(void* h) { return static_cast<UnOCPSolver*>(h)->constraints.dimc_total(); }
// slack/dual [N][dimc]; rows of components that are not valid at a stage are 0
int oracle_unocp_get_constraint_data(void* h, double* slack, double* dual) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  const int dimc = s->constraints.dimc_total();
  for (int i = 0; i < s->N(); ++i) {
    int off = 0;
    for (size_t c = 0; c < s->constraints.components.size(); ++c) {
      const ConstraintComponentData& data = s->ocp[i].cdata.data[c];
      const int n = s->constraints.components[c].lim.size();
      const bool valid = s->constraints.valid(s->constraints.components[c], i);
      for (int r = 0; r < n; ++r) {
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? (double)data.slack[r] : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? (double)data.dual[r] : 0.0;
      }
      off += n;
    }
  }
  return 0;
}
// condensed stage KKT blocks after linearize: Q[N][3nv*3nv] col-major (a,q,v order),
// res[N][5nv] (Fq,Fv,la,lq,lv)
int oracle_unocp_get_unkkt(void* h, double* Q, double* res) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  const int nv = s->robot.dimv(), n3 = 3 * nv;
  for (int i = 0; i < s->N(); ++i) {
    if (Q) xcpy(Q + (size_t)i * n3 * n3, s->unkkt_matrix[i].Q.d.data(), sizeof(double) * n3 * n3);
    if (res) {
      const SplitUnKKTResidual& r = s->unkkt_residual[i];
      const Mat* parts[5] = {&r.Fq, &r.Fv, &r.la, &r.lq, &r.lv};
      for (int p = 0; p < 5; ++p) xcpy(res + (size_t)i * 5 * nv + p * nv, parts[p]->d.data(), sizeof(double) * nv);
    }
  }
  return 0;
}

// ocpbenchmarker::CPUTime protocol (include/idocp/utils/ocp_benchmarker.hxx:13-34):
// `iters` updateSolution calls at fixed (t,q,v); returns total seconds, and the
// seconds spent inside the Riccati sweeps through *riccati_seconds.
double oracle_unocp_bench(void* h, double t, const double* q, const double* v, int iters, double* riccati_seconds) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
  s->riccati_seconds = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) s->updateSolution(t, Q, V);
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (riccati_seconds) *riccati_seconds = s->riccati_seconds;
  return el;
}

}  // extern "C"

// ---- OCPSolver (contact path, uniform contact status) ------------------------
#include "ocp.hpp"
extern "C" {

void* oracle_ocp_create(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k, double T, int N) {
  try { return new OCPSolver(*m, *c, *k, T, N); } catch (...) { return nullptr; }
}
// OCPSolver(robot, cost, constraints, T, N, max_num_impulse, nthreads) (ocp_solver.cpp:10-47)
void* oracle_ocp_create_hybrid(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k, double T, int N,
                               int max_num_impulse) {
  try { return new OCPSolver(*m, *c, *k, T, N, max_num_impulse); } catch (...) { return nullptr; }
}
// OCPSolver::pushBackContactStatus / setContactPoints (ocp_solver.cpp:174-184)
int oracle_ocp_push_back_contact_status(void* h, const int* active, const double* points, double switching_time) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  std::vector<int> a(active, active + s->robot.maxPointContacts());
  try { s->pushBackContactStatus(a, points, switching_time); } catch (...) { return -1; }
  return 0;
}
// OCPSolver::popBackContactStatus / popFrontContactStatus (ocp_solver.cpp:187-194)
int oracle_ocp_pop_back_contact_status(void* h) { static_cast<OCPSolver*>(h)->popBackContactStatus(); return 0; }
int oracle_ocp_pop_front_contact_status(void* h) { static_cast<OCPSolver*>(h)->popFrontContactStatus(); return 0; }
int oracle_ocp_set_contact_points(void* h, int phase, const double* points) {
  try { static_cast<OCPSolver*>(h)->setContactPoints(phase, points); } catch (...) { return -1; }
  return 0;
}
// the chain produced by the last discretisation: returns its length M; arrays (may be NULL) of length M
// TimeVarying task-space cost: the reference poses tabulated at M times (looked up by stage time, RCost::taskRefAt)
static void fillTaskRefTable(RCost& cost, int M, const double* times, const double* refs) {
  cost.task_tab_t.resize(M); cost.task_tab.resize(M);
  for (int k = 0; k < M; ++k) { cost.task_tab_t[k] = times[k]; for (int j = 0; j < 12; ++j) cost.task_tab[k][j] = refs[12 * k + j]; }
}
int oracle_ocp_set_task_refs(void* h, int M, const double* times, const double* refs) { fillTaskRefTable(static_cast<OCPSolver*>(h)->cost, M, times, refs); return 0; }
int oracle_parnmpc_set_task_refs(void* h, int M, const double* times, const double* refs) { fillTaskRefTable(static_cast<ParNMPCSolver*>(h)->cost, M, times, refs); return 0; }
int oracle_ocp_chain(void* h, double t, int* kind, int* index, int* slot, double* tt, double* dt, int* sw_event, int* dimf) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  try { s->discretize(t); } catch (const std::exception& e) { g_oracle_error = e.what(); return -1; }      // (a sequence the discretiser refuses)
  for (int p = 0; p < s->M(); ++p) {
    const NodeC& nd = s->chain[p];
    if (kind) kind[p] = nd.kind;
    if (index) index[p] = nd.index;
    if (slot) slot[p] = nd.slot;
    if (tt) tt[p] = nd.t;
    if (dt) dt[p] = nd.dt;
    if (sw_event) sw_event[p] = nd.sw_event;
    if (dimf) dimf[p] = nd.kind == NodeC::Terminal ? 0 : s->nodeContacts(p).dimf();
  }
  return s->M();
}
void oracle_ocp_destroy(void* h) { delete static_cast<OCPSolver*>(h); }
// OCPSolver(..., nthreads) of the reference: the stage loops run on `n` OpenMP threads (1 without -fopenmp)
int oracle_ocp_update_solution_ls(void* h, double t, const double* q, const double* v) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()), true); }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
// linearise + Riccati + direction, WITHOUT integrating (so that cost / violation of trial steps can be probed)
int oracle_ocp_compute_direction(void* h, double t, const double* q, const double* v) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  try {
    const Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
    s->linearizeOCP(t, Q); s->backwardRiccatiRecursion(); s->forwardRiccatiRecursion(Q, V); s->computeDirection();
  } catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
void oracle_ocp_clear_line_search_filter(void* h) { static_cast<OCPSolver*>(h)->line_search.filter.clear(); }
// cost and l1 constraint violation of s (+) alpha d (the direction of the last linearisation); out[2]
int oracle_ocp_cost_and_violation(void* h, double alpha, double* out) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  try { const auto cv = s->costAndViolation(alpha); out[0] = (double)cv.first; out[1] = (double)cv.second; }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
int oracle_ocp_set_num_threads(void* h, int n) { static_cast<OCPSolver*>(h)->setNumThreads(n); return 0; }
int oracle_ocp_set_contact_status(void* h, const int* active, const double* points) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  std::vector<int> a(active, active + s->robot.maxPointContacts());
  s->setContactStatusUniformly(a, points);
  return 0;
}
int oracle_ocp_set_solution(void* h, const char* name, const double* value) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const std::string n(name);
  const int dim = n == "q" ? s->robot.dimq() : (n == "u" ? s->robot.dimu() : (n == "f" ? 3 : s->robot.dimv()));
  try { s->setSolution(n, toVec(value, dim)); } catch (...) { return -1; }
  return 0;
}
int oracle_ocp_init_constraints(void* h, double t) { static_cast<OCPSolver*>(h)->initConstraints(t); return 0; }
int oracle_ocp_update_solution(void* h, double t, const double* q, const double* v) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); } catch (...) { return 1; }
  return 0;
}
// 0 linearize, 1 Riccati backward+forward, 2 direction, 3 integrate
int oracle_ocp_stage(void* h, int what, double t, const double* q, const double* v) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
  try {
    if (what == 0) s->linearizeOCP(t, Q);
    else if (what == 1) { s->backwardRiccatiRecursion(); s->forwardRiccatiRecursion(Q, V); }
    else if (what == 2) s->computeDirection();
    else if (what == 3) s->integrateSolution();
    else return -1;
  } catch (...) { return 1; }
  return 0;
}
// Test hook of the Riccati layer (tests/test_golden_riccati.py): overwrite the condensed LQR data of chain stage i -- the blocks
// RiccatiRecursionSolver consumes, in the reference's storage (Fqq / Fqv: leading 6 x 6 blocks only) -- with the caller's, so that the
// backward sweep can be held to an INDEPENDENTLY computed solution of the dense KKT system (tests/golden/gen_golden_riccati.py).
// Matrices column-major.  terminal != 0: stage i is the terminal stage (Qxx, lx only).
int oracle_ocp_inject_lqr_stage(void* h, int i, int terminal, const double* Qxx, const double* Qxu, const double* Quu, const double* Fqq6,
                                const double* Fqv6, const double* Fvq, const double* Fvv, const double* Fvu, const double* lx,
                                const double* lu, const double* Fx) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const int nv = s->robot.dimv(), nu = s->robot.dimu(), nx = 2 * nv;
  if (i < 0 || i >= (int)s->kkt_matrix.size()) return -1;
  SplitKKTMatrixC& M = s->kkt_matrix[i];
  SplitKKTResidualC& R = s->kkt_residual[i];
  auto put = [](Mat& m, const double* src, int r, int c) { Mat t(r, c); xcpy(t.d.data(), src, sizeof(double) * r * c); m = t; };
  put(M.Qxx, Qxx, nx, nx);
  for (int r = 0; r < nv; ++r) { R.lq[r] = lx[r]; R.lv[r] = lx[nv + r]; }
  if (terminal) return 0;
  Mat qxu, quu;
  put(qxu, Qxu, nx, nu); put(quu, Quu, nu, nu);
  M.Qxu_full.setBlock(0, 6, qxu); M.Quu_full.setBlock(6, 6, quu);
  put(M.Fqq6, Fqq6, 6, 6); put(M.Fqv6, Fqv6, 6, 6); put(M.Fvq, Fvq, nv, nv); put(M.Fvv, Fvv, nv, nv); put(M.Fvu, Fvu, nv, nu);
  for (int r = 0; r < nv; ++r) { R.Fq[r] = Fx[r]; R.Fv[r] = Fx[nv + r]; }
  for (int r = 0; r < nu; ++r) R.lu[r] = lu[r];
  return 0;
}
int oracle_ocp_backward_riccati_only(void* h) {
  try { static_cast<OCPSolver*>(h)->backwardRiccatiRecursion(); } catch (...) { return 1; }
  return 0;
}
int oracle_ocp_compute_kkt_residual(void* h, double t, const double* q, const double* v) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return 0;
}
int oracle_ocp_is_current_solution_feasible(void* h) { return static_cast<OCPSolver*>(h)->isCurrentSolutionFeasible(); }
double oracle_ocp_kkt_error(void* h) { return static_cast<OCPSolver*>(h)->KKTError(); }
// reference configuration of the configuration-space cost at time t (q_ref[nq])
void oracle_ocp_q_ref(void* h, double t, double* q_ref) {
  OCPSolver* o = static_cast<OCPSolver*>(h);
  Mat q;
  o->qRef(t, q);
  for (int i = 0; i < o->robot.dimq(); ++i) q_ref[i] = q[i];
}

// field of every stage, padded to `stride` doubles per stage: out[(N+1)][stride].
// solution names: q v a u f lmd gmm beta mu nu_passive ; direction names: dq dv da df du dlmd dgmm dbeta dmu dnu_passive
// f / mu / df / dmu are reported per contact slot ([nc][3], inactive slots 0 for directions).
static int ocpGetOne(OCPSolver* s, const std::string& n, int p, int stride, double* o) {
  const int nv = s->robot.dimv(), nc = s->robot.maxPointContacts();
  const NodeC& nd = s->chain[p];
  const bool terminal = nd.kind == NodeC::Terminal;
  const SplitSolutionC& x = s->s[nd.slot];
  const SplitDirectionC& d = s->d[nd.slot];
  auto put = [&](const Mat& m) { for (int k = 0; k < m.size() && k < stride; ++k) o[k] = m[k]; };
  auto putSlots = [&](const Mat& stack, int off) {   // stacked active rows -> contact slots
    const ContactStatus& cs = s->nodeContacts(p);
    int st = 0;
    for (int c = 0; c < nc; ++c) if (cs.active[c]) { for (int k = 0; k < 3; ++k) o[3 * c + k] = stack[off + st + k]; st += 3; }
  };
  if (n == "q") put(x.q); else if (n == "v") put(x.v); else if (n == "a") put(x.a); else if (n == "u") put(x.u);
  else if (n == "lmd") put(x.lmd); else if (n == "gmm") put(x.gmm); else if (n == "beta") put(x.beta);
  else if (n == "nu_passive") put(x.nu_passive);
  else if (n == "xi") put(x.xi);
  else if (n == "f") { for (int c = 0; c < nc; ++c) for (int k = 0; k < 3; ++k) o[3 * c + k] = x.f[c][k]; }
  else if (n == "mu") { for (int c = 0; c < nc; ++c) for (int k = 0; k < 3; ++k) o[3 * c + k] = x.mu[c][k]; }
  else if (n == "dq") put(d.dq); else if (n == "dv") put(d.dv); else if (n == "du") put(d.du);
  else if (n == "dlmd") put(d.dlmd); else if (n == "dgmm") put(d.dgmm); else if (n == "dnu_passive") put(d.dnu_passive);
  else if (n == "dxi") put(d.dxi);
  else if (n == "da") { if (!terminal) put(d.daf.segment(0, nv)); }
  else if (n == "dbeta") { if (!terminal) put(d.dbetamu.segment(0, nv)); }
  else if (n == "df") { if (!terminal) putSlots(d.daf, nv); }
  else if (n == "dmu") { if (!terminal) putSlots(d.dbetamu, nv); }
  else return -1;
  return 0;
}
int oracle_ocp_get(void* h, const char* name, int stride, double* out) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const std::string n(name);
  for (int i = 0; i <= s->N(); ++i) {
    const int p = s->posOfSlot(i);
    if (p < 0) return -1;
    if (ocpGetOne(s, n, p, stride, out + (size_t)i * stride) != 0) return -1;
  }
  return 0;
}
// the same fields for every stage of the chain, in chain order: out[M][stride]
int oracle_ocp_get_chain(void* h, const char* name, int stride, double* out) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const std::string n(name);
  for (int p = 0; p < s->M(); ++p) if (ocpGetOne(s, n, p, stride, out + (size_t)p * stride) != 0) return -1;
  return 0;
}
// Riccati factorisation along the chain: P[M][2nv*2nv], s[M][2nv], K[M-1][nu*2nv], k[M-1][nu]
int oracle_ocp_get_riccati_chain(void* h, double* P, double* sv, double* K, double* k) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const int nv = s->robot.dimv(), nx = 2 * nv, nu = s->robot.dimu();
  for (int p = 0; p < s->M(); ++p) {
    const int sl = s->chain[p].slot;
    const RiccatiC& r = s->riccati[sl];
    if (P) {
      Mat Pm(nx, nx);
      Pm.setBlock(0, 0, r.Pqq); Pm.setBlock(0, nv, r.Pqv); Pm.setBlock(nv, 0, r.Pqv.t()); Pm.setBlock(nv, nv, r.Pvv);
      xcpy(P + (size_t)p * nx * nx, Pm.d.data(), sizeof(double) * nx * nx);
    }
    if (sv) { xcpy(sv + (size_t)p * nx, r.sq.d.data(), sizeof(double) * nv); xcpy(sv + (size_t)p * nx + nv, r.sv.d.data(), sizeof(double) * nv); }
    if (p < s->M() - 1) {
      if (K) xcpy(K + (size_t)p * nu * nx, s->K[sl].d.data(), sizeof(double) * nu * nx);
      if (k) xcpy(k + (size_t)p * nu, s->k[sl].d.data(), sizeof(double) * nu);
    }
  }
  return 0;
}
int oracle_ocp_get_step_sizes(void* h, double* primal, double* dual) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  *primal = s->primal_step_size; *dual = s->dual_step_size; return 0;
}
// P[N+1][2nv*2nv] col-major, s[N+1][2nv], K[N][nu*2nv] col-major (nu x 2nv), k[N][nu]
int oracle_ocp_get_riccati(void* h, double* P, double* sv, double* K, double* k) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const int nv = s->robot.dimv(), nx = 2 * nv, nu = s->robot.dimu();
  for (int i = 0; i <= s->N(); ++i) {
    const RiccatiC& r = s->riccati[i];
    if (P) {
      Mat Pm(nx, nx);
      Pm.setBlock(0, 0, r.Pqq); Pm.setBlock(0, nv, r.Pqv); Pm.setBlock(nv, 0, r.Pqv.t()); Pm.setBlock(nv, nv, r.Pvv);
      xcpy(P + (size_t)i * nx * nx, Pm.d.data(), sizeof(double) * nx * nx);
    }
    if (sv) { xcpy(sv + (size_t)i * nx, r.sq.d.data(), sizeof(double) * nv); xcpy(sv + (size_t)i * nx + nv, r.sv.d.data(), sizeof(double) * nv); }
    if (i < s->N()) {
      if (K) xcpy(K + (size_t)i * nu * nx, s->K[i].d.data(), sizeof(double) * nu * nx);
      if (k) xcpy(k + (size_t)i * nu, s->k[i].d.data(), sizeof(double) * nu);
    }
  }
  return 0;
}
// rows of the friction-cone component of one contact (coneEval, ocp.cpp): returns the number of rows, res[5], J[5][3] row-major
int oracle_cone_eval(int kind, double mu, const double* f, double* res, double* J) {
  Mat fm(3); for (int k = 0; k < 3; ++k) fm[k] = f[k];
  const ConeEval e = coneEval(kind, mu, fm);
  for (int r = 0; r < e.nr; ++r) { res[r] = (double)e.res[r]; for (int c = 0; c < 3; ++c) J[3 * r + c] = (double)e.J[r][c]; }
  return e.nr;
}
int oracle_ocp_dimc(void* h) { return static_cast<OCPSolver*>(h)->dimc(); }
// slack / dual [N][dimc]: enabled components in order; rows invalid at a stage read 0
int oracle_ocp_get_constraint_data(void* h, double* slack, double* dual) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const int dimc = s->dimc();
  const idocp_constraints_t& c = s->cons;
  const int en[NCOMP] = {c.joint_position_limits, c.joint_position_limits, c.joint_velocity_limits, c.joint_velocity_limits,
                         c.joint_torque_limits, c.joint_torque_limits, c.linearized_friction_cone || c.friction_cone, 0,
                         c.joint_acceleration_lower_limit, c.joint_acceleration_upper_limit, c.contact_distance};
  for (int i = 0; i < s->N(); ++i) {
    int off = 0;
    for (int comp = 0; comp < NCOMP; ++comp) {
      if (!en[comp]) continue;
      const IpmData& data = s->ipm[i][comp];
      const bool valid = comp < 2 ? i >= 2 : (comp < 4 ? i >= 1 : true);
      for (int r = 0; r < data.slack.size(); ++r) {
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? (double)data.slack[r] : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? (double)data.dual[r] : 0.0;
      }
      off += data.slack.size();
    }
  }
  return 0;
}
// condensed stage LQR data after linearize (before the Riccati sweep modifies it), col-major:
// Qxx[2nv x 2nv], Qxu[2nv x nu], Quu[nu x nu], A[2nv x 2nv] = [Fqq Fqv; Fvq Fvv], B[2nv x nu] = [0; Fvu],
// lx[2nv], lu[nu], Fx[2nv]
int oracle_ocp_get_lqr_stage(void* h, int i, double* Qxx, double* Qxu, double* Quu, double* A, double* B, double* lx,
                             double* lu, double* Fx) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  const int nv = s->robot.dimv(), nu = s->robot.dimu(), nx = 2 * nv;
  const SplitKKTMatrixC& M = s->kkt_matrix[i];
  const SplitKKTResidualC& R = s->kkt_residual[i];
  xcpy(Qxx, M.Qxx.d.data(), sizeof(double) * nx * nx);
  Mat qxu = M.Qxu_full.block(0, 6, nx, nu), quu = M.Quu_full.block(6, 6, nu, nu);
  xcpy(Qxu, qxu.d.data(), sizeof(double) * nx * nu);
  xcpy(Quu, quu.d.data(), sizeof(double) * nu * nu);
  Mat Am(nx, nx), Bm(nx, nu);
  // implicit parts: Fqq = I, Fqv = dt I outside the leading 6x6 blocks
  // (backward_riccati_recursion_factorizer.hxx:66-71, 96-100)
  Mat Fqq = Mat::Identity(nv);
  Fqq.setBlock(0, 0, M.Fqq6);
  const int pos = s->posOfSlot(i);
  const double dt_node = (pos >= 0 && s->chain[pos].kind != NodeC::Impulse) ? (double)s->chain[pos].dt : 0.0;
  Mat Fqv_full = dt_node * Mat::Identity(nv);
  Fqv_full.setBlock(0, 0, M.Fqv6);
  Am.setBlock(0, 0, Fqq); Am.setBlock(0, nv, Fqv_full); Am.setBlock(nv, 0, M.Fvq); Am.setBlock(nv, nv, M.Fvv);
  Bm.setBlock(nv, 0, M.Fvu);
  xcpy(A, Am.d.data(), sizeof(double) * nx * nx);
  xcpy(B, Bm.d.data(), sizeof(double) * nx * nu);
  for (int r = 0; r < nv; ++r) { lx[r] = R.lq[r]; lx[nv + r] = R.lv[r]; Fx[r] = R.Fq[r]; Fx[nv + r] = R.Fv[r]; }
  xcpy(lu, R.lu.d.data(), sizeof(double) * nu);
  return 0;
}
// ---- test hook: the un-condensed Newton system of chain position `pos` (UncondensedC, ocp.hpp) --------------------------------
//   oracle_ocp_keep_uncondensed(h, 1) before compute_direction / update_solution; then per position
//   oracle_ocp_get_uncondensed(h, pos, "meta", out) -> [valid, kind, dimf, dimi, has_u, dt, dtq, active_mask]; any other name: the block, column-major;
//   returns the number of doubles (out may be NULL to ask for it), -1 for an unknown name
int oracle_ocp_keep_uncondensed(void* h, int on) { static_cast<OCPSolver*>(h)->keep_uncondensed = on != 0; return 0; }
static int uncondensedField(const UncondensedC& U, const char* name, double* out);
int oracle_ocp_get_uncondensed(void* h, int pos, const char* name, double* out) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  if (pos < 0 || pos >= s->M() || (int)s->unc.size() != s->nslots()) return -1;
  return uncondensedField(s->unc[s->chain[pos].slot], name, out);
}
// the same for a ParNMPCSolver (backward-Euler stages; Fqq = d Fq / d q of the stage itself, no Fqq_prev)
int oracle_parnmpc_keep_uncondensed(void* h, int on) { static_cast<ParNMPCSolver*>(h)->keep_uncondensed = on != 0; return 0; }
int oracle_parnmpc_get_uncondensed(void* h, int pos, const char* name, double* out) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  if (pos < 0 || pos >= s->M() || (int)s->unc.size() != s->nslots()) return -1;
  return uncondensedField(s->unc[s->chain[pos].slot], name, out);
}
static int uncondensedField(const UncondensedC& U, const char* name, double* out) {
  const std::string n(name);
  if (n == "meta") {
    if (out) { out[0] = U.valid; out[1] = U.kind; out[2] = U.dimf; out[3] = U.dimi; out[4] = U.has_u; out[5] = (double)U.dt; out[6] = (double)U.dtq; out[7] = U.active_mask; }
    return 8;
  }
  const Mat* m = n == "Qxx" ? &U.Qxx : n == "Qaa" ? &U.Qaa : n == "Qff" ? &U.Qff : n == "Quu" ? &U.Quu : n == "lq" ? &U.lq : n == "lv" ? &U.lv : n == "la" ? &U.la
               : n == "lf" ? &U.lf : n == "lu" ? &U.lu : n == "lu_passive" ? &U.lu_passive : n == "Fq" ? &U.Fq : n == "Fv" ? &U.Fv : n == "Fqq" ? &U.Fqq
               : n == "Fqq_prev" ? &U.Fqq_prev : n == "dIDCdqv" ? &U.dIDCdqv : n == "M" ? &U.M : n == "J" ? &U.J : n == "IDC" ? &U.IDC : n == "Phix" ? &U.Phix
               : n == "Phia" ? &U.Phia : n == "P" ? &U.P : n == "aux_next" ? &U.aux_next : nullptr;
  if (!m) return -1;
  if (out) for (int i = 0; i < m->size(); ++i) out[i] = (double)m->d[i];
  return m->size();
}
double oracle_ocp_bench(void* h, double t, const double* q, const double* v, int iters, double* riccati_seconds) {
  OCPSolver* s = static_cast<OCPSolver*>(h);
  Mat Q = toVec(q, s->robot.dimq()), V = toVec(v, s->robot.dimv());
  s->riccati_seconds = 0;
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) s->updateSolution(t, Q, V);
  const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (riccati_seconds) *riccati_seconds = s->riccati_seconds;
  return el;
}

// ---- ParNMPCSolver (event-free horizons) -------------------------------------------------
void* oracle_parnmpc_create(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k, double T, int N) {
  try { return new ParNMPCSolver(*m, *c, *k, T, N); } catch (...) { return nullptr; }
}
void* oracle_parnmpc_create_hybrid(const idocp_model_t* m, const idocp_cost_t* c, const idocp_constraints_t* k, double T, int N,
                                   int max_num_impulse) {
  try { return new ParNMPCSolver(*m, *c, *k, T, N, max_num_impulse); } catch (...) { return nullptr; }
}
int oracle_parnmpc_pop_back_contact_status(void* h) { static_cast<ParNMPCSolver*>(h)->popBackContactStatus(); return 0; }
int oracle_parnmpc_pop_front_contact_status(void* h) { static_cast<ParNMPCSolver*>(h)->popFrontContactStatus(); return 0; }
int oracle_parnmpc_push_back_contact_status(void* h, const int* active, const double* points, double switching_time) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  std::vector<int> a(active, active + s->robot.maxPointContacts());
  try { s->pushBackContactStatus(a, points, switching_time); } catch (...) { return -1; }
  return 0;
}
// the chain of the discretisation at time t (ParNMPCDiscretizer): returns its length; arrays of `capacity` entries
int oracle_parnmpc_chain(void* h, double t, int capacity, int* kind, int* index, int* slot, double* tt, double* dt, int* dimf, int* level) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try { s->discretize(t); } catch (...) { return -1; }
  const int M = s->M();
  for (int p = 0; p < M && p < capacity; ++p) {
    const ParNMPCSolver::PNode& nd = s->chain[p];
    kind[p] = nd.kind; index[p] = nd.index; slot[p] = nd.slot; tt[p] = nd.t; dt[p] = nd.dt; dimf[p] = s->nodeContacts(nd).dimf(); level[p] = nd.level;
  }
  return M;
}
// solution / direction field along the chain: out[M][stride]  (a = dv, da = ddv on impulse stages)
int oracle_parnmpc_get_chain(void* h, const char* name, int stride, double* out) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  const std::string name_in(name);
  const int nv = s->robot.dimv(), nc = s->robot.maxPointContacts();
  for (int p = 0; p < s->M(); ++p) {
    const ParNMPCSolver::PNode& nd = s->chain[p];
    double* o = out + (size_t)p * stride;
    for (int k = 0; k < stride; ++k) o[k] = 0.0;
    std::string n = name_in;
    const bool coarse = n.rfind("new_", 0) == 0;      // "new_" + field: the coarse / corrected iterate s_new of the backward correction
    const SplitSolutionC& x = coarse ? s->s_new[nd.slot] : s->s[nd.slot];
    const SplitDirectionC& d = s->d[nd.slot];
    const ContactStatus& cs = s->nodeContacts(nd);
    auto put = [&](const Mat& m) { for (int k = 0; k < m.size() && k < stride; ++k) o[k] = m[k]; };
    if (coarse) n = n.substr(4);
    if (n == "q") put(x.q); else if (n == "v") put(x.v); else if (n == "a") put(x.a);
    else if (n == "u") { if (nd.kind != NodeC::Impulse) put(x.u); }
    else if (n == "lmd") put(x.lmd); else if (n == "gmm") put(x.gmm); else if (n == "beta") put(x.beta);
    else if (n == "xi") { if (nd.kind == NodeC::Aux) put(x.xi); }
    else if (n == "f" || n == "mu") { for (int c = 0; c < nc; ++c) for (int k = 0; k < 3; ++k) o[3 * c + k] = (n == "f" ? x.f : x.mu)[c][k]; }      // every contact slot (inactive ones keep their initial guess)
    else if (n == "nu_passive") { if (nd.kind != NodeC::Impulse) put(x.nu_passive); }
    else if (n == "dq") put(d.dq); else if (n == "dv") put(d.dv);
    else if (n == "du") { if (nd.kind != NodeC::Impulse) put(d.du); }
    else if (n == "dlmd") put(d.dlmd); else if (n == "dgmm") put(d.dgmm);
    else if (n == "dnu_passive") { if (nd.kind != NodeC::Impulse) put(d.dnu_passive); }
    else if (n == "dxi") { if (nd.kind == NodeC::Aux) put(d.dxi); }
    else if (n == "da") put(d.daf.segment(0, nv));
    else if (n == "dbeta") put(d.dbetamu.segment(0, nv));
    else if (n == "df" || n == "dmu") {
      const Mat& st = n == "df" ? d.daf : d.dbetamu;
      int k0 = 0;
      for (int c = 0; c < nc; ++c) if (cs.active[c]) { for (int k = 0; k < 3; ++k) o[3 * c + k] = st[nv + k0 + k]; k0 += 3; }
    }
    else return -1;
  }
  return 0;
}
void oracle_parnmpc_destroy(void* h) { delete static_cast<ParNMPCSolver*>(h); }
int oracle_parnmpc_set_contact_status(void* h, const int* active, const double* points) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  std::vector<int> a(active, active + s->robot.maxPointContacts());
  s->setContactStatusUniformly(a, points);
  return 0;
}
int oracle_parnmpc_set_solution(void* h, const char* name, const double* value) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  const std::string n(name);
  const int dim = n == "q" ? s->robot.dimq() : (n == "u" ? s->robot.dimu() : (n == "f" ? 3 : s->robot.dimv()));
  try { s->setSolution(n, toVec(value, dim)); } catch (...) { return -1; }
  return 0;
}
// Warm start (the MPC use of the solver: the iterate and the correction state persist between calls): one field of ONE stage,
// name in q v a u f (all contacts stacked) lmd gmm beta mu (stacked); and BackwardCorrectionSolver::aux_mat_ of one stage
int oracle_parnmpc_set_stage(void* h, int i, const char* name, const double* value) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  if (i < 0 || i >= (int)s->s.size()) return -1;
  const std::string n(name);
  SplitSolutionC& x = s->s[i];
  const int nv = s->robot.dimv(), nc = s->robot.maxPointContacts();
  if (n == "q") x.q = toVec(value, s->robot.dimq());
  else if (n == "v") x.v = toVec(value, nv);
  else if (n == "a") x.a = toVec(value, nv);
  else if (n == "u") x.u = toVec(value, s->robot.dimu());
  else if (n == "lmd") x.lmd = toVec(value, nv);
  else if (n == "gmm") x.gmm = toVec(value, nv);
  else if (n == "beta") x.beta = toVec(value, nv);
  else if (n == "f") for (int c = 0; c < nc; ++c) x.f[c] = toVec(value + 3 * c, 3);
  else if (n == "mu") for (int c = 0; c < nc; ++c) x.mu[c] = toVec(value + 3 * c, 3);
  else return -1;
  return 0;
}
int oracle_parnmpc_set_aux_mat(void* h, int i, const double* mat /* nx x nx col-major */) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  if (i < 0 || i >= (int)s->aux_mat.size()) return -1;
  const int nx = 2 * s->robot.dimv();
  Mat m(nx, nx);
  xcpy(m.d.data(), mat, sizeof(double) * nx * nx);
  s->aux_mat[i] = m;
  return 0;
}
int oracle_parnmpc_init(void* h, double t) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  s->initBackwardCorrection(t);
  s->initConstraints(t);
  return 0;
}
int oracle_parnmpc_update_solution(void* h, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
// filter line search of ParNMPCSolver (event-free horizons): updateSolution(t, q, v, true); the direction alone; cost / violation of a trial step
int oracle_parnmpc_update_solution_ls(void* h, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try { s->updateSolution(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()), true); }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
int oracle_parnmpc_compute_direction(void* h, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try { s->computeDirection(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
int oracle_parnmpc_cost_and_violation(void* h, double alpha, const double* q, const double* v, double* out2) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try { const auto cv = s->costAndViolation(alpha, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); out2[0] = (double)cv.first; out2[1] = (double)cv.second; }
  catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
void oracle_parnmpc_clear_line_search_filter(void* h) { static_cast<ParNMPCSolver*>(h)->line_search.filter.clear(); }
int oracle_parnmpc_is_current_solution_feasible(void* h) { return static_cast<ParNMPCSolver*>(h)->isCurrentSolutionFeasible(); }
double oracle_parnmpc_kkt_error(void* h, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return s->KKTError();
}
// solution / direction field of every stage: out[N][stride]
int oracle_parnmpc_get(void* h, const char* name, int stride, double* out) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  const std::string n(name);
  const int nv = s->robot.dimv(), nc = s->robot.maxPointContacts();
  for (int i = 0; i < s->N(); ++i) {
    double* o = out + (size_t)i * stride;
    const SplitSolutionC& x = s->s[i];
    const SplitDirectionC& d = s->d[i];
    auto put = [&](const Mat& m) { for (int k = 0; k < m.size() && k < stride; ++k) o[k] = m[k]; };
    if (n == "q") put(x.q); else if (n == "v") put(x.v); else if (n == "a") put(x.a); else if (n == "u") put(x.u);
    else if (n == "lmd") put(x.lmd); else if (n == "gmm") put(x.gmm); else if (n == "beta") put(x.beta);
    else if (n == "f") { for (int c = 0; c < nc; ++c) for (int k = 0; k < 3; ++k) o[3 * c + k] = x.f[c][k]; }
    else if (n == "mu") { for (int c = 0; c < nc; ++c) for (int k = 0; k < 3; ++k) o[3 * c + k] = x.mu[c][k]; }
    else if (n == "nu_passive") put(x.nu_passive);
    else if (n == "dq") put(d.dq); else if (n == "dv") put(d.dv); else if (n == "du") put(d.du);
    else if (n == "dlmd") put(d.dlmd); else if (n == "dgmm") put(d.dgmm); else if (n == "dnu_passive") put(d.dnu_passive);
    else if (n == "da") put(d.daf.segment(0, nv));
    else if (n == "dbeta") put(d.dbetamu.segment(0, nv));
    else if (n == "df" || n == "dmu") {
      const Mat& st = n == "df" ? d.daf : d.dbetamu;
      int k0 = 0;
      for (int c = 0; c < nc; ++c) if (s->contact_status.active[c]) { for (int k = 0; k < 3; ++k) o[3 * c + k] = st[nv + k0 + k]; k0 += 3; }
    }
    else return -1;
  }
  return 0;
}
int oracle_parnmpc_get_step_sizes(void* h, double* primal, double* dual) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  *primal = s->primal_step_size; *dual = s->dual_step_size; return 0;
}
// ---- horizon sharding of the ParNMPC oracle (see ParNMPCSolver in ocp.hpp) ----
int oracle_parnmpc_set_shard(void* h, int stage_offset, int has_terminal, int has_prev) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  s->stage_offset = stage_offset; s->has_terminal = has_terminal != 0; s->has_prev = has_prev != 0;
  return 0;
}
// a horizon WITH discrete events: this object keeps the grid stages [stage_begin, stage_end) of the whole horizon and the event
// stages in front of them
int oracle_parnmpc_set_chain_slice(void* h, int stage_begin, int stage_end) {
  static_cast<ParNMPCSolver*>(h)->setChainSlice(stage_begin, stage_end);
  return 0;
}
// 0 linearize + coarse update, 1 backward serial, 2 backward parallel, 3 forward serial, 4 forward parallel (+ directions,
// local step sizes), 5 integrate
int oracle_parnmpc_phase(void* h, int phase, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  try {
    switch (phase) {
      case 0: s->coarseUpdate(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv())); break;
      case 1: s->backwardCorrectionSerial(); break;
      case 2: s->backwardCorrectionParallel(); break;
      case 3: s->forwardCorrectionSerial(); break;
      case 4: s->forwardCorrectionParallel(); break;
      default: s->integrateSolution(); break;
    }
  } catch (const std::exception& e) { g_oracle_error = e.what(); return 1; } catch (...) { g_oracle_error = "unknown"; return 1; }
  return 0;
}
int oracle_parnmpc_init_constraints_only(void* h, double t) { static_cast<ParNMPCSolver*>(h)->initConstraints(t); return 0; }
int oracle_parnmpc_init_aux_only(void* h, double t) { static_cast<ParNMPCSolver*>(h)->initBackwardCorrection(t); return 0; }
int oracle_parnmpc_halo_size(void* h, int kind) { return static_cast<ParNMPCSolver*>(h)->haloSize(kind); }
int oracle_parnmpc_export(void* h, int kind, double* out) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  std::vector<real> t(s->haloSize(kind));
  s->exportHalo(kind, t.data());
  xcpy(out, t.data(), sizeof(double) * t.size());
  return 0;
}
int oracle_parnmpc_import(void* h, int kind, const double* in) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  std::vector<real> t(s->haloSize(kind));
  xcpy(t.data(), in, sizeof(double) * t.size());
  s->importHalo(kind, t.data());
  return 0;
}
int oracle_parnmpc_set_step_sizes(void* h, double primal, double dual) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  s->primal_step_size = primal; s->dual_step_size = dual; return 0;
}
double oracle_parnmpc_kkt_error_squared(void* h, double t, const double* q, const double* v) {
  ParNMPCSolver* s = static_cast<ParNMPCSolver*>(h);
  s->computeKKTResidual(t, toVec(q, s->robot.dimq()), toVec(v, s->robot.dimv()));
  return s->KKTErrorSquared();
}
}  // extern "C"
