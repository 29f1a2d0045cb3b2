#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NA>
__global__ __launch_bounds__(64) void k(double* out, int iters) {
  d4 acc[NA];
  for (int m = 0; m < NA; ++m) acc[m] = d4{0, 0, 0, 0};
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < NA; ++m) acc[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc[m], 0, 0, 0);
  }
  double s = 0; for (int m = 0; m < NA; ++m) s += acc[m][0] + acc[m][3];
  out[(size_t)blockIdx.x * 64 + threadIdx.x] = s;
}
template <int NA> void run(int blocks) {
  double* o; hipMalloc(&o, 8 * 64 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  hipLaunchKernelGGL((k<NA>), dim3(blocks), dim3(64), 0, 0, o, iters); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL((k<NA>), dim3(blocks), dim3(64), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%d wavefronts/SIMD, %d independent accumulators: %.1f ns per MFMA = %.0f cycles at 2.4 GHz\n", blocks / 1024, NA, ms * 1e6 / iters / NA, ms * 1e6 / iters / NA * 2.4);
  hipFree(o);
}
int main() { run<1>(1024); run<2>(1024); run<3>(1024); run<4>(1024); run<6>(1024); run<8>(1024); run<1>(2048); run<2>(2048); run<1>(4096); run<2>(4096); return 0; }
