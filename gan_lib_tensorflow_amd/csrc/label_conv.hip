// The spatially CONSTANT half of a 3x3 conv's input, factored out (round 5).
//
// The critic concatenates the label embedding, tiled over the 16x16 grid, to its 128 feature channels in front of D.Block.2
// (SNGAN/gan_cifar_resnet.py:276-284: expand_dims x2 + tile + concat), so half of the 256 input channels of D.Block.2.Conv1
// (:186-190, 302 of the critic's 855 MFLOP per sample) hold ONE vector per sample, r_n = relu(T[label_n]) after the block's
// pre-activation.  For those channels
//     conv(x)[n, p, :] = sum_{taps t whose input pixel p + t - 1 lies inside the image} W[t, c0:c0+C2, :]^T r_n
// depends on the sample only through its LABEL and on the pixel only through its BORDER CLASS (3 row classes x 3 column classes:
// first / inner / last): a table of V x 9 x Cout numbers replaces half of the layer's multiply-adds in the forward pass, the
// input gradient and the filter gradient.  Exact algebra, no approximation; the fp32 table takes the place of an fp32 MFMA
// accumulation of the same bf16 products.
//   forward : bias_table[v][cls][co] = bias[co] + sum_{t valid in cls} sum_c bf16(W[t][c0+c][co]) * relu(T[v][c])
//             (gank_label_conv3x3_table); the image-resident conv on the feature half adds row (label_n, cls(p)) in its epilogue
//             (gank_img16_conv3x3_label_bias);
//   backward: S[n][t][co] = sum of dy[n, p, co] over the pixels p where tap t is valid (one pass over dy), then
//             dW[t][c0+c][co] += sum_n r_n[c] S[n][t][co]   and   de[n][c] = relu'(T[label_n][c]) sum_{t,co} bf16(W[t][c0+c][co]) S[n][t][co]
//             (gank_label_conv3x3_bwd; de leaves as 9 per-tap partials the consumer adds in a fixed order).
#include "gank_common.h"

namespace {
// tap t = (kh, kw) reads input pixel (y + kh - 1, x + kw - 1): invalid in the first row / column class for kh / kw == 0 and in the
// last one for kh / kw == 2.  cls = 3 * row class + column class, classes 0 first, 1 inner, 2 last.
__device__ __forceinline__ bool tap_valid(int t, int cls) {
  const int kh = t / 3, kw = t - 3 * kh, rc = cls / 3, cc = cls - 3 * rc;
  return !(kh == 0 && rc == 0) && !(kh == 2 && rc == 2) && !(kw == 0 && cc == 0) && !(kw == 2 && cc == 2);
}
}  // namespace

// grid = V * 9 (label, class), 256 threads over the output channels
__global__ __launch_bounds__(256) void label_conv_table_kernel(const float* __restrict__ w, int Cin_total, int c0, int C2, int Cout,
                                                             const bf16* __restrict__ T, const float* __restrict__ bias, float* __restrict__ out) {
  __shared__ float r[1024];
  const int v = blockIdx.x / 9, cls = blockIdx.x - 9 * v;
  for (int c = threadIdx.x; c < C2; c += 256) r[c] = fmaxf(bf2f(T[(long)v * C2 + c]), 0.f);
  __syncthreads();
  for (int co = threadIdx.x; co < Cout; co += 256) {
    float acc = 0.f;
    for (int t = 0; t < 9; t++) {
      if (!tap_valid(t, cls)) continue;          // block-uniform
      const float* wp = w + ((long)t * Cin_total + c0) * Cout + co;
      float a4[4] = {0.f, 0.f, 0.f, 0.f};
      for (int c = 0; c < C2; c += 4) {          // four independent chains (C2 % 4 == 0)
#pragma unroll
        for (int u = 0; u < 4; u++) a4[u] += bf2f(f2bf(wp[(long)(c + u) * Cout])) * r[c + u];
      }
      acc += (a4[0] + a4[1]) + (a4[2] + a4[3]);
    }
    out[((long)v * 9 + cls) * Cout + co] = acc + (bias ? bias[co] : 0.f);
  }
}

extern "C" int gank_label_conv3x3_table(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                        float* bias_table, void* stream) {
  GANK_REQUIRE(w && T && bias_table && V > 0 && C2 > 0 && C2 <= 1024 && C2 % 4 == 0 && c0 >= 0 && c0 + C2 <= Cin_total && Cout > 0,
               "label_conv3x3_table: bad arguments");
  hipLaunchKernelGGL(label_conv_table_kernel, dim3(V * 9), dim3(256), 0, (hipStream_t)stream, w, Cin_total, c0, C2, Cout, (const bf16*)T, bias, bias_table);
  GANK_LAUNCH_OK("label_conv3x3_table");
  return 0;
}

// S[n][t][co]: block = sample, thread = (8-channel group g, pixel lane pl); 16-byte loads, nine predicated accumulators, the pixel
// lanes of a channel group meet in LDS in a fixed order.  H x W pixels per sample (row-major), Cout % 8 == 0, 256 % (Cout / 8) == 0.
__global__ __launch_bounds__(256) void label_conv_tap_sums_kernel(const bf16* __restrict__ dy, float* __restrict__ S, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float red[];         // [PL][9][Cout]
  const int n = blockIdx.x, cg = Cout >> 3, PL = 256 / cg;
  const int g = threadIdx.x % cg, pl = threadIdx.x / cg;
  float acc[9][8];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 8; e++) acc[t][e] = 0.f;
  const int HW = H * W;
  for (int p = pl; p < HW; p += PL) {
    const int y = p / W, x = p - y * W;
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + ((long)n * HW + p) * Cout + g * 8);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; e++) f[e] = bf2f(v[e]);
    // tap (kh, kw) is valid at this pixel when its input pixel (y + kh - 1, x + kw - 1) is inside
    const bool rv[3] = {y > 0, true, y < H - 1}, cv[3] = {x > 0, true, x < W - 1};
#pragma unroll
    for (int t = 0; t < 9; t++) {
      const float m = (rv[t / 3] && cv[t % 3]) ? 1.f : 0.f;
#pragma unroll
      for (int e = 0; e < 8; e++) acc[t][e] += m * f[e];
    }
  }
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 8; e++) red[((long)pl * 9 + t) * Cout + g * 8 + e] = acc[t][e];
  __syncthreads();
  for (int i = threadIdx.x; i < 9 * Cout; i += 256) {
    float s = 0.f;
    for (int l = 0; l < PL; l++) s += red[(long)l * 9 * Cout + i];
    S[(long)n * 9 * Cout + i] = s;
  }
}

// block = (tap t, tile of CT channels of the constant half): the whole S[:, t, :] (N x Cout fp32, padded rows) and the tile's
// weights in LDS.  Threads = output channels for the filter gradient, then (sample, channel group) for the vector gradient.
constexpr int LCB_CT = 16;
__global__ __launch_bounds__(256) void label_conv_bwd_kernel(const float* __restrict__ S, const int* __restrict__ labels, const bf16* __restrict__ T, int V,
                                                           const float* __restrict__ w, int Cin_total, int c0, int C2, int Cout, int N,
                                                           float* __restrict__ dw, float* __restrict__ de_parts,
                                                           float* __restrict__ dw_feat_tmp, int merge_blocks) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const int tiles = C2 / LCB_CT;
  if ((int)blockIdx.x >= 9 * tiles) {
    // the feature half's filter gradient, accumulated by the ordinary filter-gradient launch into a contiguous [9][c0][Cout] buffer:
    // added into rows [0, c0) of every tap of dw, the buffer cleared for the next pass (no fill launch)
    const long total = 9L * c0 * Cout;
    for (long i = ((long)blockIdx.x - 9 * tiles) * 256 + tid; i < total; i += (long)merge_blocks * 256) {
      const long t = i / ((long)c0 * Cout), rem = i - t * (long)c0 * Cout;
      dw[t * (long)Cin_total * Cout + rem] += dw_feat_tmp[i];
      dw_feat_tmp[i] = 0.f;
    }
    return;
  }
  const int t = blockIdx.x / tiles, ct = blockIdx.x - t * tiles;
  const int SP = Cout + 1;                               // padded row: thread n walks row n without bank conflicts
  float* Ss = sm;                                        // [N][Cout + 1]
  float* Ws = Ss + (long)N * SP;                         // [CT][Cout]  (bf16-rounded, as the MFMA operand was)
  float* Rs = Ws + LCB_CT * Cout;                        // [V][CT]     relu(T), and the relu mask below
  int* Lb = reinterpret_cast<int*>(Rs + V * LCB_CT);     // [N]
  for (int i = tid; i < N * Cout; i += 256) {
    const int n = i / Cout, co = i - n * Cout;
    Ss[(long)n * SP + co] = S[((long)n * 9 + t) * Cout + co];
  }
  for (int i = tid; i < LCB_CT * Cout; i += 256) {
    const int j = i / Cout, co = i - j * Cout;
    Ws[i] = bf2f(f2bf(w[((long)t * Cin_total + c0 + ct * LCB_CT + j) * Cout + co]));
  }
  for (int i = tid; i < V * LCB_CT; i += 256) Rs[i] = fmaxf(bf2f(T[(long)(i / LCB_CT) * C2 + ct * LCB_CT + (i % LCB_CT)]), 0.f);
  for (int i = tid; i < N; i += 256) {
    const int lb = labels[i];
    Lb[i] = lb < 0 ? 0 : (lb >= V ? V - 1 : lb);
  }
  __syncthreads();
  // filter gradient: dw[t][c0 + c][co] += sum_n r_n[c] S[n][t][co]   (this block is the only writer of its rows)
  for (int co = tid; co < Cout; co += 256) {
    float acc[LCB_CT];
#pragma unroll
    for (int j = 0; j < LCB_CT; j++) acc[j] = 0.f;
    for (int n = 0; n < N; n++) {
      const float s = Ss[(long)n * SP + co];
      const float* rr = Rs + Lb[n] * LCB_CT;
#pragma unroll
      for (int j = 0; j < LCB_CT; j++) acc[j] += rr[j] * s;
    }
#pragma unroll
    for (int j = 0; j < LCB_CT; j++) dw[((long)t * Cin_total + c0 + ct * LCB_CT + j) * Cout + co] += acc[j];
  }
  // vector gradient, this tap's share: de_parts[t][n][c] = [T[label_n][c] > 0] sum_co W[t][c0 + c][co] S[n][t][co]
  for (int i = tid; i < N * (LCB_CT / 8); i += 256) {
    const int n = i % N, jh = i / N;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; j++) acc[j] = 0.f;
    const float* sr = Ss + (long)n * SP;
    for (int co = 0; co < Cout; co++) {
      const float s = sr[co];
#pragma unroll
      for (int j = 0; j < 8; j++) acc[j] += Ws[(jh * 8 + j) * Cout + co] * s;
    }
    const float* rr = Rs + Lb[n] * LCB_CT + jh * 8;
#pragma unroll
    for (int j = 0; j < 8; j++) de_parts[((long)t * N + n) * C2 + ct * LCB_CT + jh * 8 + j] = rr[j] > 0.f ? acc[j] : 0.f;
  }
}

extern "C" long gank_label_conv3x3_bwd_ws_floats(int N, int Cout) { return (long)N * 9 * Cout; }

extern "C" int gank_label_conv3x3_bwd(const void* dy, const int32_t* labels, const void* T, int V, const float* w, int Cin_total, int c0, int C2,
                                      int Cout, int N, int H, int W, float* dw, float* dw_feat_tmp, float* de_parts, float* ws, void* stream) {
  GANK_REQUIRE(dy && labels && T && w && dw && de_parts && ws && N > 0 && H > 1 && W > 1 && V > 0, "label_conv3x3_bwd: bad arguments");
  GANK_REQUIRE(Cout % 8 == 0 && 256 % (Cout / 8) == 0 && C2 % LCB_CT == 0 && c0 >= 0 && c0 + C2 <= Cin_total, "label_conv3x3_bwd: unsupported channel counts");
  hipStream_t s = (hipStream_t)stream;
  const int PL = 256 / (Cout / 8);
  const size_t lds1 = (size_t)PL * 9 * Cout * sizeof(float);
  const size_t lds2 = ((size_t)N * (Cout + 1) + (size_t)LCB_CT * Cout + (size_t)V * LCB_CT) * sizeof(float) + (size_t)N * sizeof(int);
  GANK_REQUIRE(lds1 <= 160 * 1024 && lds2 <= 160 * 1024, "label_conv3x3_bwd: N = %d, Cout = %d do not fit the LDS", N, Cout);
  GANK_MAX_DYNAMIC_LDS(label_conv_tap_sums_kernel, (int)lds1, "label_conv3x3_bwd");
  hipLaunchKernelGGL(label_conv_tap_sums_kernel, dim3(N), dim3(256), lds1, s, (const bf16*)dy, ws, H, W, Cout);
  const int merge_blocks = dw_feat_tmp ? 64 : 0;
  GANK_MAX_DYNAMIC_LDS(label_conv_bwd_kernel, (int)lds2, "label_conv3x3_bwd");
  hipLaunchKernelGGL(label_conv_bwd_kernel, dim3(9 * (C2 / LCB_CT) + merge_blocks), dim3(256), lds2, s, ws, labels, (const bf16*)T, V, w, Cin_total, c0, C2, Cout,
                     N, dw, de_parts, dw_feat_tmp, merge_blocks);
  GANK_LAUNCH_OK("label_conv3x3_bwd");
  return 0;
}
