#include <hip/hip_runtime.h>
#include <cstdio>
template <typename T>
__global__ __launch_bounds__(256) void k(T* out, int iters) {
  T a[16];
  for (int i = 0; i < 16; ++i) a[i] = (T)(threadIdx.x + i) * (T)1e-3;
  T b = (T)1.0000001, c = (T)1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = a[i] * b + c;
  }
  T s = 0; for (int i = 0; i < 16; ++i) s += a[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename T> void run(const char* name) {
  T* o; hipMalloc(&o, sizeof(T) * 256 * 4096);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000, blocks = 4096;
  hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, o, iters); hipDeviceSynchronize();
  hipEventRecord(e0); hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, o, iters); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fma = (double)blocks * 256 * iters * 16;
  printf("%s: %.3f ms, %.1f TFLOP/s (2 flops per FMA)\n", name, ms, 2 * fma / (ms * 1e-3) / 1e12);
}
int main() { run<double>("f64"); run<float>("f32"); return 0; }
