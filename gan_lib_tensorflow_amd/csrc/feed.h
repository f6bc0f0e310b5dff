// Device code shared by the critic-feed launch (loss_opt.hip) and the spectral-norm forward launch that carries the feed as a
// block range of its own (sn.hip): the Philox generator and the body of the feed.
#pragma once
#include "gank_common.h"

// Philox4x32-10 counter-based RNG; state = {seed, offset} in device memory, advanced on the device
struct u4 { unsigned x, y, z, w; };

__device__ __forceinline__ u4 philox4x32_10(unsigned long long ctr, unsigned long long stream_off, unsigned long long seed) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = (unsigned)stream_off, c3 = (unsigned)(stream_off >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return {c0, c1, c2, c3};
}
__device__ __forceinline__ float u01(unsigned x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }  // [0,1)


struct CriticFeedArgs {
  const unsigned char* real_all;
  const int* labels_all;
  const bf16* fake_all;
  bf16* both;
  int* labels2;
  int* slot;
  unsigned long long* state;
  unsigned* done;
  int B, n_slots;
  int blocks;          // block count of the feed (a range of a larger launch, or its whole grid)
};

// Block `bid` of `a.blocks` (256 threads each).  both[0:B] = 2*(real/256 - .5) + U[0,1/128) for slot *slot of the iteration's real
// batches (CHW rows -> HWC), both[B:2B] = the generator output kept for that slot, labels2 = the slot's labels twice; then the slot
// counter and the RNG offset advance, by the LAST block to finish (every block has read them by then).
__device__ __forceinline__ void critic_feed_block(const CriticFeedArgs& a, int bid) {
  const unsigned long long seed = a.state[0], off = a.state[1];
  const int sl = a.slot[0];
  const int B = a.B;
  const unsigned char* data = a.real_all + (long)sl * B * 3072;
  const long n = (long)B * 3072, n4 = n >> 2;
  const long stride = (long)a.blocks * 256, t = bid * 256L + threadIdx.x;
  for (long i = t; i < n4; i += stride) {                       // identical arithmetic to preprocess_kernel
    const u4 r = philox4x32_10((unsigned long long)i, off, seed);
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 4; e++) {
      const long o = i * 4 + e;
      const int b = (int)(o / 3072), rem = (int)(o - (long)b * 3072);
      const int c = rem % 3, hw = rem / 3;
      const float px = (float)data[(long)b * 3072 + c * 1024 + hw];
      a.both[o] = f2bf(2.f * (px / 256.f - .5f) + u01(v[e]) * (1.f / 128.f));
    }
  }
  const u32x4* fs = reinterpret_cast<const u32x4*>(a.fake_all + (long)sl * n);
  u32x4* fd = reinterpret_cast<u32x4*>(a.both + n);
  for (long i = t; i < n / 8; i += stride) fd[i] = fs[i];
  for (long i = t; i < B; i += stride) {
    const int lb = a.labels_all[(long)sl * B + i];
    a.labels2[i] = lb;
    a.labels2[B + i] = lb;
  }
  // no fence: the counters only have to be READ by every block before the last one rewrites them, and each
  // block's loads of them are consumed (addresses of everything above) before it reaches its atomic
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned prev = atomicAdd(a.done, 1u);
    if (prev == (unsigned)a.blocks - 1) {
      a.done[0] = 0u;
      a.slot[0] = sl + 1 < a.n_slots ? sl + 1 : 0;
      a.state[1] = off + 1;
    }
  }
}

static inline int critic_feed_blocks(int B) {       // the standalone launch's grid (rgrid of loss_opt.hip)
  long g = ((long)B * 768 + 255) / 256;
  if (g > 2048) g = 2048;
  return (int)(g < 1 ? 1 : g);
}
