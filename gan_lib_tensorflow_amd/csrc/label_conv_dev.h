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


// ---- the label gradients (gank_label_conv3x3_bwd's second launch), also carried by the image-resident input-gradient launch ----
constexpr int LCB_CT = 16, LCB_V = 16;
struct LabelBwdArgs {
  const float* S;          // [2 row halves][V][9][Cout] per-label tap sums
  const bf16* T;           // [V][C2]
  const float* w;          // the whole fp32 filter [3,3,Cin_total,Cout]
  float* dw;               // its gradient (rows c0 .. c0 + C2 - 1 of every tap accumulated here)
  float* de_parts;         // [9][V][C2]
  float* dw_feat_tmp;      // optional staging buffer of the other rows (see gank_label_conv3x3_bwd)
  int V, Cin_total, c0, C2, Cout, merge_blocks;
  int blocks;              // 9 * (C2 / LCB_CT) + merge_blocks + pool_blocks
  // optional tenth part: the pooled shortcut branch's gradient of the tiled vector, summed per label -- de_parts[9][v][c] = sum over
  // the samples n of label v and the pooled pixels p of gp[n][p][gp_c0 + c] (pool_blocks = V * (C2 / 32) extra blocks)
  const bf16* gp;          // [N][HWp][gp_pitch]
  const int* lists;        // [V][N + 1]
  int N, HWp, gp_pitch, gp_c0, pool_blocks;
};
// one block of the label gradients; NT = threads of the calling launch (>= 256: the first 256 work, all of them meet at the barriers)
__device__ __forceinline__ void label_conv_bwd_block(const LabelBwdArgs& q, int block, float* sm) {
  const float* __restrict__ S = q.S;
  const bf16* __restrict__ T = q.T;
  const float* __restrict__ w = q.w;
  float* __restrict__ dw = q.dw;
  float* __restrict__ de_parts = q.de_parts;
  float* __restrict__ dw_feat_tmp = q.dw_feat_tmp;
  const int V = q.V, Cin_total = q.Cin_total, c0 = q.c0, C2 = q.C2, Cout = q.Cout, merge_blocks = q.merge_blocks;
  const int tid = threadIdx.x < 256 ? threadIdx.x : 256 + (threadIdx.x & 255);      // threads past 256 fall outside every loop bound below
  const bool worker = threadIdx.x < 256;
  const int tiles = C2 / LCB_CT;
  if (block >= 9 * tiles + merge_blocks) {
    // block = (label v, 32 channels): thread = (8-channel group, pooled pixel lane); the label's samples in list order, 16 loads in flight
    const int pb = block - 9 * tiles - merge_blocks, nch = C2 >> 5;
    const int v = pb / nch, ch = pb - v * nch;
    const int g = threadIdx.x & 3, pl = (threadIdx.x >> 2) & 63;
    const int* list = q.lists + (long)v * (q.N + 1) + 1;
    const int cnt = q.lists[(long)v * (q.N + 1)];
    float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (worker)
      for (int p = pl; p < q.HWp; p += 64)
        for (int sb = 0; sb < cnt; sb += 16) {
          bf16x8 x[16];
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const int n = list[sb + u < cnt ? sb + u : cnt - 1];
            x[u] = *reinterpret_cast<const bf16x8*>(q.gp + ((long)n * q.HWp + p) * q.gp_pitch + q.gp_c0 + ch * 32 + g * 8);
          }
#pragma unroll
          for (int u = 0; u < 16; u++) {
            const float on = sb + u < cnt ? 1.f : 0.f;
#pragma unroll
            for (int e = 0; e < 8; e++) a8[e] += on * bf2f(x[u][e]);
          }
        }
    if (worker) {
#pragma unroll
      for (int e = 0; e < 8; e++) sm[(pl * 4 + g) * 8 + e] = a8[e];
    }
    __syncthreads();
    if (threadIdx.x < 32) {
      const int c = threadIdx.x;
      float t = 0.f;
      for (int l = 0; l < 64; l++) t += sm[(l * 4 + (c >> 3)) * 8 + (c & 7)];
      de_parts[(9L * V + v) * C2 + ch * 32 + c] = t;
    }
    return;
  }
  if (block >= 9 * tiles) {
    // the feature half's filter gradient, accumulated by the ordinary filter-gradient launch into a contiguous [9][c0][Cout] buffer:
    // added into rows [0, c0) of every tap of dw, the buffer cleared for the next pass (no fill launch)
    const long total = 9L * c0 * Cout;
    for (long i = ((long)block - 9 * tiles) * 256 + threadIdx.x; worker && i < total; i += (long)merge_blocks * 256) {
      const long t = i / ((long)c0 * Cout), rem = i - t * (long)c0 * Cout;
      dw[t * (long)Cin_total * Cout + rem] += dw_feat_tmp[i];
      dw_feat_tmp[i] = 0.f;
    }
    return;
  }
  const int t = block / tiles, ct = block - t * tiles;
  const int SP = Cout + 4;                               // padded rows (16-byte aligned)
  float* Sl = sm;                                        // [V][SP]     per-label sums of this tap
  float* Ws = Sl + (long)V * SP;                         // [CT][SP]    (bf16-rounded, as the MFMA operand was)
  float* Rs = Ws + (long)LCB_CT * SP;                    // [V][CT]     relu(T)
  // Cout <= 256 (host check): one output channel per thread.  EVERY global load of the block is requested before the first wait:
  // the tile's filter rows, the per-label sums of both row halves, the gradient rows this block adds to (three dependent round
  // trips of 2 us each otherwise)
  const int co = tid < Cout ? tid : Cout - 1;
  float wreg[LCB_CT], dold[LCB_CT], sl[LCB_V];
#pragma unroll
  for (int j = 0; j < LCB_CT; j++) wreg[j] = w[((long)t * Cin_total + c0 + ct * LCB_CT + j) * Cout + co];
#pragma unroll
  for (int v = 0; v < LCB_V; v++) {
    const int vv = v < V ? v : V - 1;
    const float a = S[((long)vv * 9 + t) * Cout + co] + S[(((long)V + vv) * 9 + t) * Cout + co];      // the two row halves
    sl[v] = v < V ? a : 0.f;
  }
#pragma unroll
  for (int j = 0; j < LCB_CT; j++) dold[j] = dw[((long)t * Cin_total + c0 + ct * LCB_CT + j) * Cout + co];
  for (int i = tid; i < V * LCB_CT; i += 256) Rs[i] = fmaxf(bf2f(T[(long)(i / LCB_CT) * C2 + ct * LCB_CT + (i % LCB_CT)]), 0.f);
  if (tid < Cout) {
#pragma unroll
    for (int j = 0; j < LCB_CT; j++) Ws[(long)j * SP + tid] = bf2f(f2bf(wreg[j]));
#pragma unroll
    for (int v = 0; v < LCB_V; v++)
      if (v < V) Sl[(long)v * SP + tid] = sl[v];
  }
  __syncthreads();
  if (tid < Cout) {
    // filter gradient of this block's CT rows: dw[t][c0 + c][co] += sum_v relu(T[v][c]) Sl[v][co]   (this block is their only writer)
#pragma unroll
    for (int v = 0; v < LCB_V; v++)
      if (v < V) {
#pragma unroll
        for (int j = 0; j < LCB_CT; j++) dold[j] += Rs[v * LCB_CT + j] * sl[v];
      }
#pragma unroll
    for (int j = 0; j < LCB_CT; j++) dw[((long)t * Cin_total + c0 + ct * LCB_CT + j) * Cout + tid] = dold[j];
  }
  // gradient of the tiled vector, this tap's and tile's share per LABEL: thread (v, channel j of the tile)
  if (tid < V * LCB_CT) {
    const int v = tid / LCB_CT, j = tid - v * LCB_CT;
    f32x4 a4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4* wr = reinterpret_cast<const f32x4*>(Ws + (long)j * SP);
    const f32x4* sr = reinterpret_cast<const f32x4*>(Sl + (long)v * SP);
#pragma unroll 8
    for (int q = 0; q < (Cout >> 2); q++) {
      const f32x4 a = wr[q], b2 = sr[q];
#pragma unroll
      for (int u = 0; u < 4; u++) a4[u] += a[u] * b2[u];
    }
    de_parts[((long)t * V + v) * C2 + ct * LCB_CT + j] = Rs[v * LCB_CT + j] > 0.f ? (a4[0] + a4[1]) + (a4[2] + a4[3]) : 0.f;
  }
}

