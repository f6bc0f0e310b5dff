// Device code of the per-label tap sums (label_conv.hip), shared with the all-taps filter-gradient launch that can carry them as extra
// blocks (conv_wgrad.hip: both read the same dy; the sums then cost no launch of their own).
#pragma once
#include "gank_common.h"

constexpr int LABEL_TAP_SUMS_LDS = 32 * 9 * 64 * (int)sizeof(float);

__device__ __forceinline__ void label_conv_tap_sums_block(const bf16* __restrict__ dy, const int* __restrict__ lists, float* __restrict__ S,
                                                          int N, int V, int H, int W, int Cout, int block, float* red) {
  // (a 512-thread form -- 64 pixel lanes, 16 samples per batch, the lanes of a wave meeting by shuffles -- measured 18.8 us against 11.4)
  // red: [32 pixel lanes][9][64] floats of dynamic LDS
  const int nchunk = Cout >> 6;
  int b = block;
  const int cb = (b % nchunk) * 64; b /= nchunk;
  const int half = b & 1, v = b >> 1;
  const int tid = threadIdx.x, g = tid & 7, pl = tid >> 3;
  const int* list = lists + (long)v * (N + 1) + 1;
  const int cnt = lists[(long)v * (N + 1)];
  const int HW = H * W, hp = HW >> 1, p0 = half * hp;
  const int KP = hp >> 5;                                             // pixels per thread (4 at 16 x 16)
  float dl[4][8];
#pragma unroll
  for (int k = 0; k < 4; k++)
#pragma unroll
    for (int e = 0; e < 8; e++) dl[k][e] = 0.f;
  for (int sb = 0; sb < cnt; sb += 8) {
    bf16x8 x[8][4];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int n = list[sb + u < cnt ? sb + u : cnt - 1];
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const int p = p0 + pl + 32 * (k < KP ? k : 0);
        x[u][k] = *reinterpret_cast<const bf16x8*>(dy + ((long)n * HW + p) * Cout + cb + g * 8);
      }
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const float on = sb + u < cnt ? 1.f : 0.f;
#pragma unroll
      for (int k = 0; k < 4; k++)
#pragma unroll
        for (int e = 0; e < 8; e++) dl[k][e] += on * bf2f(x[u][k][e]);
    }
  }
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 8; e++) acc[t][e] = 0.f;
#pragma unroll
  for (int k = 0; k < 4; k++) {
    if (k < KP) {
      const int p = p0 + pl + 32 * k, y = p / W, x = p - y * W;
      // tap (kh, kw) is valid at this pixel when its input pixel (y + kh - 1, x + kw - 1) is inside
      const bool rv[3] = {y > 0, true, y < H - 1}, cv[3] = {x > 0, true, x < W - 1};
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const float m = (rv[t / 3] && cv[t % 3]) ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < 8; e++) acc[t][e] += m * dl[k][e];
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 8; e++) red[((long)pl * 9 + t) * 64 + g * 8 + e] = acc[t][e];
  __syncthreads();
  for (int i = tid; i < 9 * 64; i += 256) {
    float sum = 0.f;
    for (int l = 0; l < 32; l++) sum += red[(long)l * 9 * 64 + i];
    const int t = i >> 6, c = i & 63;
    S[(((long)half * V + v) * 9 + t) * Cout + cb + c] = sum;
  }
}

