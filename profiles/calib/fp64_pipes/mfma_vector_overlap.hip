#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NV, int NM>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = (threadIdx.x + i) * 1e-3;
  d4 acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  const double b = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NM; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[m & 1], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i & 7] = a[i & 7] * b + c;
  }
  double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = s + acc[0][0] + acc[1][1];
}
template <int NV, int NM> void run() {
  double* o; hipMalloc(&o, 8 * 64 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000, blocks = 1024;      // one wavefront per SIMD
  hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(64), 0, 0, o, iters); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<NV, NM>), dim3(blocks), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("per iteration: %d mfma_f64_16x16x4 + %2d v_fma_f64: %.1f ns = %.0f cycles at 2.4 GHz\n", NM, NV, ms * 1e6 / iters, ms * 1e6 / iters * 2.4);
  hipFree(o);
}
int main() { run<0, 1>(); run<8, 0>(); run<8, 1>(); run<16, 1>(); run<0, 2>(); run<8, 2>(); run<16, 2>(); run<32, 2>(); return 0; }
