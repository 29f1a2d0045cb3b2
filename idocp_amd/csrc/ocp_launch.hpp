// Host-visible launch wrappers of the contact-path kernels.
#ifndef IDOCP_OCP_LAUNCH_HPP_
#define IDOCP_OCP_LAUNCH_HPP_

#include <hip/hip_runtime.h>

#include "ocp_device.hpp"

namespace idocp_dev {

// longest chain ocp_forward_expand_kernel keeps in LDS (slot, status word, two time steps per node); longer chains run S4 + K6
constexpr int OcpForwardExpandMaxChain = 320;

template <typename D>
struct OcpLaunch {
  // M = length of the chain (stages in time order incl. event stages and the terminal stage)
  static void rnea(const OcpBuffers& B, long batch, int M, int n_impulse, hipStream_t st);      // K5a (M - 1 stages; the n_impulse impulse stages in a launch of their own)
  static void switching(const OcpBuffers& B, long batch, int M, hipStream_t st);     // K5s: switching-constraint terms (all stages; no-op where absent)
  // q0_lie != nullptr: with the Lie-group tasks of the forward-Euler stages (else the caller launches a lie kernel itself)
  static void nominal(const OcpBuffers& B, long batch, int M, hipStream_t st, const double* q0_lie = nullptr, hipStream_t st_imp = nullptr);      // st_imp: the launch over the impulse stages of a forward-Euler chain on a stream of its own      // K5n: nominal Newton-Euler sweeps -> nom record (every K5 / K8 / merit launch is preceded by it)
  static void condense(const OcpBuffers& B, long batch, int M, int dimf, const double* q0, hipStream_t st, int part = 0);   // K5b (+ terminal); dimf = -1: mixed chain
  // part 0: everything; 1: the nominal sweeps + external rows; 2: the class launches + external Hessian; 3: the LARGEST class only; 4: the other
  // classes only (parts 3 and 4 let the caller put them on two streams: the launches are independent, every stage writes its own records);
  // 5: the external Hessian only
  static void condenseMixed(const OcpBuffers& B, long batch, int M, const int n[5], const double* q0, hipStream_t st, int part = 0, hipStream_t st_imp = nullptr);   // K5b on a chain with events, per stage class
  static void residual(const OcpBuffers& B, long batch, int M, const double* q0, hipStream_t st);   // K8
  static void condenseBackwardEuler(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, bool residual,
                                    hipStream_t st);                                          // K9a: ParNMPC stage (K5b with backward Euler)
  // S3.  wide: eight wavefronts per instance with P staged in LDS (latency mode, a handful of instances) instead of the register-resident
  // sweep of one wavefront per instance
  static void riccatiBackward(const OcpBuffers& B, long batch, int M, bool hybrid, hipStream_t st, bool wide = false);
  static void riccatiForward(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, hipStream_t st);  // S4
  static void parnmpcInverse(const OcpBuffers& B, long batch, int M, hipStream_t st);            // K9w, or K9b with IDOCP_K9_WAVE=0
  static void parnmpcInverseWave(const OcpBuffers& B, long batch, int M, hipStream_t st);        // K9w (parnmpc_kkt_wave_kernel.hip)
  // terms with frame Jacobians of their own (ocp_ext_kernel.hip): no-ops when B.ext == nullptr
  static void extRows(const OcpBuffers& B, long batch, int M, bool residual, hipStream_t st);
  static void extHessian(const OcpBuffers& B, long batch, int M, hipStream_t st);
  static void extInit(const OcpBuffers& B, long batch, int NS, hipStream_t st);
  static void parnmpcImpulseMerit(const OcpBuffers& Btry, long batch, int n_impulse, const double* q0, const double* v0, hipStream_t st);      // line search on the impulse stages
  static void parnmpcImpulseCondense(const OcpBuffers& B, long batch, int n_impulse, bool residual, const double* q0, const double* v0,
                                     hipStream_t st);                                               // K9i: impulse stages of a ParNMPC chain
  static void parnmpcEventInverse(const OcpBuffers& B, long batch, int n_general, hipStream_t st); // aux (switching rows) and impulse stages: K9w's general instantiation, or K9g with IDOCP_K9_WAVE=0
  static void parnmpcEventInverseWave(const OcpBuffers& B, long batch, int n_general, hipStream_t st);
  static void parnmpcPhase(int phase, const OcpBuffers& B, long batch, int M, bool has_terminal, const double* q0, const double* v0,
                           hipStream_t st);                                                       // 0 S5, 1 K10a, 2 S6, 3 K10b, 4 init aux_mat
  static void parnmpcHalo(const OcpBuffers& B, long batch, int kind, bool do_import, double* buf, double* q0, double* v0, hipStream_t st);
  // filter line search of the floating-base solvers (src/line_search/line_search.cpp)
  static void trialIterate(const OcpBuffers& B, long batch, int M, hipStream_t st);             // s (+) alpha d -> sol_try, barrier cost
  static void merit(const OcpBuffers& Btry, long batch, int M, const double* q0, hipStream_t st);  // per-stage cost + l1 violation of sol_try
  static void meritBackwardEuler(const OcpBuffers& Btry, long batch, int M, const double* q0, const double* v0, hipStream_t st);      // the same for ParNMPC (event-free)
  static void meritReduce(const OcpBuffers& B, long batch, hipStream_t st);
  static void expandPrimal(const OcpBuffers& B, long batch, int M, hipStream_t st);   // K6 (+ step-size reduction)
  static void forwardExpand(const OcpBuffers& B, long batch, int M, const double* q0, const double* v0, hipStream_t st);   // S4 + K6 + reduction fused (round 5)
  static void expandDualIntegrate(const OcpBuffers& B, long batch, int M, hipStream_t st, hipStream_t st_base = nullptr);   // K7 (+ K7b, the base poses: independent of K7, on st_base if given)
  static void initConstraints(const OcpBuffers& B, long batch, int NS, hipStream_t st);     // every slot
  static void single(int kernel_id, const OcpBuffers& B, long batch, int M, hipStream_t st);   // ids 4, 5, 6
};

void ocpKktErrorReduce(const OcpBuffers& B, long batch, hipStream_t st, double* squared_out = nullptr);
void ocpFillStages(double* rec, int stride, int offset, int dim, long NS, int nstages, long batch, const double* values, hipStream_t st,
                   const OcpNode* nodes = nullptr);
void ocpFillField(double* sol, int stride, int offset, int dim, long nrec_per_inst, long batch, const double* value,
                  int per_instance, int repeat, hipStream_t st);

}  // namespace idocp_dev
#endif  // IDOCP_OCP_LAUNCH_HPP_
