#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  unsigned lo = 1000 + lane, hi = 2000 + lane;
  u32x2 sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);
  out[lane] = sw[0]; out[64 + lane] = sw[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("sw0: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[0], h[1], h[32], h[33]);
  printf("sw1: lane0 %u lane1 %u lane32 %u lane33 %u\n", h[64], h[65], h[96], h[97]);
  return 0;
}
