#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned long long* out) {      // block (la, lb): A one-hot at lane la, B one-hot at lane lb; out = mask of nonzero D lanes
  const int l = threadIdx.x, la = blockIdx.x, lb = blockIdx.y;
  const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(l == la ? 1.0 : 0.0, l == lb ? 1.0 : 0.0, 0.0, 0, 0, 0);
  const unsigned long long m = __ballot(d != 0.0);
  if (l == 0) out[la * 64 + lb] = m;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 8 * 4096); static unsigned long long h[4096];
  hipLaunchKernelGGL(k, dim3(64, 64), dim3(64), 0, 0, d); hipMemcpy(h, d, 8 * 4096, hipMemcpyDeviceToHost);
  // for every A lane: which B lanes pair with it (same block, same k), and where the product lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d pairs with B lanes:", la);
    for (int lb = 0; lb < 64; ++lb) if (h[la * 64 + lb]) { int dl = __builtin_ctzll(h[la * 64 + lb]); printf(" %d->D%d%s", lb, dl, __builtin_popcountll(h[la * 64 + lb]) > 1 ? "+" : ""); }
    printf("\n");
    if (la == 19) { printf("...\n"); la = 43; }
  }
  return 0;
}
