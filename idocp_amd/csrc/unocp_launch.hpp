// Host-visible launch wrappers of the UnOCP kernels (defined in unocp_kernels.hip).
#ifndef IDOCP_UNOCP_LAUNCH_HPP_
#define IDOCP_UNOCP_LAUNCH_HPP_

#include <hip/hip_runtime.h>

#include "unocp_device.hpp"

namespace idocp_dev {

template <int NV>
struct UnLaunch {
  static void linearize(const UnBuffers& B, long batch, int N, hipStream_t st);
  static void residual(const UnBuffers& B, long batch, int N, hipStream_t st);
  static void riccati(const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st);
  static void single(int kernel_id, const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st);
  static void expand(const UnBuffers& B, long batch, int N, hipStream_t st);
  static void integrate(const UnBuffers& B, long batch, int N, hipStream_t st);
  static void initConstraints(const UnBuffers& B, long batch, int N, hipStream_t st);
  // UnParNMPC (backward-Euler stages + backward correction)
  static void parnmpcPhase(int phase, const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st);
  static void parnmpcResidual(const UnBuffers& B, long batch, int N, const double* q0, const double* v0, hipStream_t st);
  static void parnmpcInitAux(const UnBuffers& B, long batch, int N, hipStream_t st);
  // UnLineSearch::computeCostAndViolation at the trial steps B.ls_alpha -> B.ls_out
  static void lineSearchEval(const UnBuffers& B, long batch, int N, bool bwd, const double* q0, const double* v0, hipStream_t st);
  static void rneaDerivatives(const DevModel* m, int n, const double* q, const double* v, const double* a, double* tau,
                              double* dq, double* dv, double* da, bool zaxes, hipStream_t st);
};

void stridedCopy(double* dst, long dst_stride, long dst_off, const double* src, long src_stride, long src_off, int n, long batch, hipStream_t st);
void squareInto(double* out, const double* err, long batch, hipStream_t st);
void fillField(double* sol, int stride, int offset, int dim, long nrec_per_inst, long batch, const double* value,
               int per_instance, hipStream_t st);

}  // namespace idocp_dev
#endif  // IDOCP_UNOCP_LAUNCH_HPP_
