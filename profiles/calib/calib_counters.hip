// Calibration of the rocprofv3 FETCH_SIZE / WRITE_SIZE counters on known byte counts, in the access widths the solver's kernels
// use (8 B and 16 B per lane, coalesced), on buffers far beyond the 256 MiB Infinity Cache (MI355X_MICROARCH.md, HBM section:
// FETCH_SIZE reads half of a wide coalesced stream on gfx950; other widths and WRITE_SIZE are uncalibrated -- this program is the
// calibration profiles/make_traffic_json.py's factors come from).  Each kernel moves exactly N * 8 bytes per direction it touches.
//   hipcc --offload-arch=gfx950 -O3 -o calib_counters calib_counters.hip
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- ./calib_counters ;  rocprofv3 --kernel-trace --pmc WRITE_SIZE -- ./calib_counters
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void calib_read8(const double* __restrict__ a, double* __restrict__ out, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i < n; i += (long)gridDim.x * blockDim.x) acc += a[i];
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ void calib_read16(const double2* __restrict__ a, double* __restrict__ out, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  double acc = 0.0;
  for (; i < n2; i += (long)gridDim.x * blockDim.x) { const double2 x = a[i]; acc += x.x + x.y; }
  if (acc == 1.2345e300) out[0] = acc;
}
__global__ void calib_write8(double* __restrict__ a, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) a[i] = (double)i;
}
__global__ void calib_write16(double2* __restrict__ a, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n2; i += (long)gridDim.x * blockDim.x) a[i] = make_double2((double)i, 1.0);
}
__global__ void calib_copy8(const double* __restrict__ a, double* __restrict__ b, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}

int main() {
  const long n = 1L << 28;      // 2 GiB of doubles per buffer
  double *a, *b;
  if (hipMalloc(&a, n * 8) != hipSuccess || hipMalloc(&b, n * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(a, 0, n * 8); hipMemset(b, 0, n * 8);
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(calib_read8, grid, block, 0, 0, a, b, n);
    hipLaunchKernelGGL(calib_read16, grid, block, 0, 0, (const double2*)a, b, n / 2);
    hipLaunchKernelGGL(calib_write8, grid, block, 0, 0, b, n);
    hipLaunchKernelGGL(calib_write16, grid, block, 0, 0, (double2*)b, n / 2);
    hipLaunchKernelGGL(calib_copy8, grid, block, 0, 0, a, b, n);
  }
  hipDeviceSynchronize();
  printf("bytes per kernel and direction: %ld\n", n * 8);
  return 0;
}
