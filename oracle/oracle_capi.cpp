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

static Mat toVec(const double* p, int n) { Mat v(n); std::memcpy(v.d.data(), p, sizeof(double) * n); return v; }

extern "C" {

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
  std::memcpy(tau, t.d.data(), sizeof(double) * m->nv);
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
  std::memcpy(dq, Dq.d.data(), n); std::memcpy(dv, Dv.d.data(), n); std::memcpy(da, Da.d.data(), n);
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
  std::memcpy(C, Cm.d.data(), sizeof(double) * 3 * nc);
  std::memcpy(dCdq, Dq.d.data(), sizeof(double) * 3 * nc * nv);
  std::memcpy(dCdv, Dv.d.data(), sizeof(double) * 3 * nc * nv);
  std::memcpy(dCda, Da.d.data(), sizeof(double) * 3 * nc * nv);
  for (int c = 0; c < nc; ++c) {
    r.contactFrame(c, fp + 3 * c, fR + 9 * c, fv + 6 * c, fa + 6 * c);
    Mat a1, a2, a3, a4;
    r.frameDerivatives(c, a1, a2, a3, a4);
    const size_t n = (size_t)6 * nv;
    std::memcpy(vdq + c * n, a1.d.data(), sizeof(double) * n); std::memcpy(adq + c * n, a2.d.data(), sizeof(double) * n);
    std::memcpy(adv + c * n, a3.d.data(), sizeof(double) * n); std::memcpy(ada + c * n, a4.d.data(), sizeof(double) * n);
  }
  if (MJtJinv) {
    Mat dq, dv, Mm, out;
    r.RNEADerivatives(Q, V, A, dq, dv, Mm);
    Robot::computeMJtJinv(Mm, Da, out);
    std::memcpy(MJtJinv, out.d.data(), sizeof(double) * out.size());
  }
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
  std::memcpy(q_int, out.d.data(), sizeof(double) * m->nq);
  std::memcpy(diff, d.d.data(), sizeof(double) * m->nv);
  std::memcpy(J0, j0.d.data(), sizeof(double) * m->nv * m->nv);
  std::memcpy(J1, j1.d.data(), sizeof(double) * m->nv * m->nv);
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
    std::memcpy(out + (size_t)i * f->size(), f->d.data(), sizeof(double) * f->size());
  }
  return 0;
}
int oracle_unocp_get_direction(void* h, const char* name, double* out) {
  UnOCPSolver* s = static_cast<UnOCPSolver*>(h);
  for (int i = 0; i <= s->N(); ++i) {
    const Mat* f = dirField(s->d[i], name);
    if (!f) return -1;
    std::memcpy(out + (size_t)i * f->size(), f->d.data(), sizeof(double) * f->size());
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
      std::memcpy(P + (size_t)i * nx * nx, Pm.d.data(), sizeof(double) * nx * nx);
    }
    if (sv) { std::memcpy(sv + (size_t)i * nx, r.sq.d.data(), sizeof(double) * nv); std::memcpy(sv + (size_t)i * nx + nv, r.sv.d.data(), sizeof(double) * nv); }
    if (i < s->N()) {
      if (K) std::memcpy(K + (size_t)i * nv * nx, s->K[i].d.data(), sizeof(double) * nv * nx);
      if (k) std::memcpy(k + (size_t)i * nv, s->k[i].d.data(), sizeof(double) * nv);
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
        if (slack) slack[(size_t)i * dimc + off + r] = valid ? data.slack[r] : 0.0;
        if (dual) dual[(size_t)i * dimc + off + r] = valid ? data.dual[r] : 0.0;
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
    if (Q) std::memcpy(Q + (size_t)i * n3 * n3, s->unkkt_matrix[i].Q.d.data(), sizeof(double) * n3 * n3);
    if (res) {
      const SplitUnKKTResidual& r = s->unkkt_residual[i];
      const Mat* parts[5] = {&r.Fq, &r.Fv, &r.la, &r.lq, &r.lv};
      for (int p = 0; p < 5; ++p) std::memcpy(res + (size_t)i * 5 * nv + p * nv, parts[p]->d.data(), sizeof(double) * nv);
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
