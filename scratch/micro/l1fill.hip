// How many bytes per second can ONE CU pull from the L2, and how does it depend on the loads it keeps in flight?
// Every block walks the same 2-MB buffer (L2-resident, 64x the 32-KB L1) from its own start; a thread issues DEPTH independent
// 16-byte loads before it consumes any.  Stand-alone: hipcc --offload-arch=gfx950 -O3 -o scratch/micro/l1fill scratch/micro/l1fill.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int NT>
__global__ __launch_bounds__(NT) void fill_kernel(const u32x4* __restrict__ buf, unsigned* __restrict__ out, int n16, int iters) {
  const int stride = NT;                                   // a wave covers 1 KB = 8 lines per load instruction
  int pos = (blockIdx.x * 7919 * NT + threadIdx.x) % n16;
  u32x4 acc = {0u, 0u, 0u, 0u};
  for (int it = 0; it < iters; it++) {
    u32x4 v[DEPTH];
#pragma unroll
    for (int d = 0; d < DEPTH; d++) {
      v[d] = buf[pos];
      pos += stride; if (pos >= n16) pos -= n16;
    }
#pragma unroll
    for (int d = 0; d < DEPTH; d++) acc ^= v[d];
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1u;
}

template <int DEPTH, int NT>
static void run(const u32x4* buf, unsigned* out, int n16, int blocks, long bytes_per_block) {
  const int iters = (int)(bytes_per_block / (16L * NT * DEPTH));
  hipEvent_t a, b;
  (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  hipLaunchKernelGGL((fill_kernel<DEPTH, NT>), dim3(blocks), dim3(NT), 0, 0, buf, out, n16, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(a, 0);
  for (int r = 0; r < 5; r++) hipLaunchKernelGGL((fill_kernel<DEPTH, NT>), dim3(blocks), dim3(NT), 0, 0, buf, out, n16, iters);
  (void)hipEventRecord(b, 0);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  const double t = ms / 5.0 * 1e-3;
  const double total = (double)iters * 16.0 * NT * DEPTH * blocks;
  printf("threads %4d depth %2d blocks %4d: %7.1f us  %6.2f TB/s chip  %6.1f GB/s per CU (256 CUs)  in flight per block %5.1f KB\n", NT, DEPTH, blocks,
         t * 1e6, total / t / 1e12, total / t / 256.0 / 1e9, 16.0 * NT * DEPTH / 1024.0);
}

int main() {
  const int n16 = (2 << 20) / 16;
  u32x4* buf; unsigned* out;
  (void)hipMalloc(&buf, (size_t)n16 * 16); (void)hipMalloc(&out, 4);
  (void)hipMemset(buf, 1, (size_t)n16 * 16);
  const long per_block = 8L << 20;
  for (int blocks : {256, 512, 1024}) {
    run<1, 256>(buf, out, n16, blocks, per_block);
    run<2, 256>(buf, out, n16, blocks, per_block);
    run<4, 256>(buf, out, n16, blocks, per_block);
    run<8, 256>(buf, out, n16, blocks, per_block);
    run<16, 256>(buf, out, n16, blocks, per_block);
    run<4, 1024>(buf, out, n16, blocks, per_block);
    run<8, 1024>(buf, out, n16, blocks, per_block);
  }
  return 0;
}
