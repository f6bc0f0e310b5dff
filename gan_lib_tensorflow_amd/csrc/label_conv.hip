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
//             (gank_label_conv3x3_bwd; de leaves as 9 per-tap partials [9][V][C2] SUMMED OVER THE SAMPLES OF A LABEL -- r and the mask
//             depend on the label alone, and so do de's consumers; gank_concat_label_unpool_bwd_factored adds them, in a fixed order,
//             to the row of the label's first sample).
#include "gank_common.h"
#include "label_conv_dev.h"

namespace {
// tap t = (kh, kw) reads input pixel (y + kh - 1, x + kw - 1): invalid in the first row / column class for kh / kw == 0 and in the
// last one for kh / kw == 2.  cls = 3 * row class + column class, classes 0 first, 1 inner, 2 last.
__device__ __forceinline__ bool tap_valid(int t, int cls) {
  const int kh = t / 3, kw = t - 3 * kh, rc = cls / 3, cc = cls - 3 * rc;
  return !(kh == 0 && rc == 0) && !(kh == 2 && rc == 2) && !(kw == 0 && cc == 0) && !(kw == 2 && cc == 2);
}
}  // namespace

// block = (label v, 64 output channels), thread = (tap t, channel co): 128-deep dot products with 16 loads in flight (the first form --
// a block per (label, class) walking every valid tap, 1152 dependent-ish loads per thread -- took 46 us), the nine per-tap sums meet
// in LDS and thread (class, co) adds the taps valid in its class.
// The pooled half of functional.concat_label_fork_pool as a rider of the table launch (the block's shortcut reads
// mean_pool2x2(concat(a, tile(T[labels]))); the full-resolution concatenation is not built): concat_label_pool_fwd_kernel's
// arithmetic for its pooled output, expression for expression (elementwise.hip), grid-stride over `nblocks` blocks of `nthreads`.
struct LabelPoolRider {
  const bf16* a;          // [N, 2Hp, 2Wp, C1]
  bf16* yp;               // [N, Hp, Wp, C1 + C2]
  long total8;
  int Hp, Wp, C1, blocks;
};
__device__ __forceinline__ void label_pool_rider_block(const LabelPoolRider& q, const bf16* __restrict__ T, const int* __restrict__ labels, int C2, int V,
                                                      int bid, int nthreads) {
  const int C = q.C1 + C2, cg = C >> 3, cg1 = q.C1 >> 3;
  const int W = 2 * q.Wp;
  for (long i = bid * (long)nthreads + threadIdx.x; i < q.total8; i += (long)q.blocks * nthreads) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int pw = (int)(p % q.Wp); p /= q.Wp;
    const int ph = (int)(p % q.Hp);
    const int n = (int)(p / q.Hp);
    const long hi = ((long)n * 2 * q.Hp + 2 * ph) * W + 2 * pw;          // top-left high-resolution pixel
    bf16x8 m;
    if (g < cg1) {
      const bf16x8 v0 = *reinterpret_cast<const bf16x8*>(q.a + hi * q.C1 + g * 8);
      const bf16x8 v1 = *reinterpret_cast<const bf16x8*>(q.a + (hi + 1) * q.C1 + g * 8);
      const bf16x8 v2 = *reinterpret_cast<const bf16x8*>(q.a + (hi + W) * q.C1 + g * 8);
      const bf16x8 v3 = *reinterpret_cast<const bf16x8*>(q.a + (hi + W + 1) * q.C1 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) m[e] = f2bf((bf2f(v0[e]) + bf2f(v2[e]) + bf2f(v1[e]) + bf2f(v3[e])) * 0.25f);   // order of tf.add_n at :120-121
    } else {
      const int l = labels[n];
      const bool ok = l >= 0 && l < V;
      m = *reinterpret_cast<const bf16x8*>(T + (long)(ok ? l : 0) * C2 + (g - cg1) * 8);
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; e++) m[e] = f2bf(0.f);
      }
    }
    *reinterpret_cast<bf16x8*>(q.yp + (((long)n * q.Hp + ph) * q.Wp + pw) * C + g * 8) = m;
  }
}

// The pooled concatenation of ONE sample per block, kept in LDS, and the block's 1x1 shortcut conv on it (gan_cifar_resnet.py:172-184:
// Shortcut = Conv2D 1x1 on the mean-pooled input): sc[n] = yp[n] W_s + b_s by 8 waves = (4 tiles of 32 output channels) x (2 tiles of
// 32 pooled pixels), A fragments straight from the plain-conv operand (bf16 [Cs][wpitch], rows = output channels: lane (r, h) reads 16
// bytes of row r), B fragments from the LDS image.  The launch of the 1x1 conv (6.4 us at the launch floor) is gone.
struct LabelShortcut {
  const bf16* wf;          // [Cs][wpitch]
  const float* bias;       // [Cs] or null
  bf16* out;               // [N, Hp, Wp, Cs]; null = no shortcut (the element-wise rider runs instead)
  int Cs, wpitch;
};
constexpr int LSC_MAXC = 256, LSC_PITCH = 2 * LSC_MAXC + 16;      // bytes per pooled pixel in LDS (33 sixteen-byte units: odd)
__device__ __forceinline__ void label_pool_shortcut_block(const LabelPoolRider& q, const LabelShortcut& sc, const bf16* __restrict__ T,
                                                          const int* __restrict__ labels, int C2, int V, int n, char* img) {
  const int C = q.C1 + C2, cg = C >> 3, cg1 = q.C1 >> 3;
  const int W = 2 * q.Wp, HWp = q.Hp * q.Wp;                        // HWp == 64 (host check)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ct = wave & 3, pt = (wave >> 2) & 1, r = lane & 31, h = lane >> 5;
  // the A fragments of this wave's 32 output channels: requested before the pooling pass
  bf16x8 fa[LSC_MAXC / 16];
  if (wave < 8) {
    const bf16* wrow = sc.wf + (long)(ct * 32 + r) * sc.wpitch + h * 8;
#pragma unroll
    for (int kk = 0; kk < LSC_MAXC / 16; kk++)
      if (kk * 16 < C) fa[kk] = *reinterpret_cast<const bf16x8*>(wrow + kk * 16);
  }
  // every load of the block is requested before the first wait: the sample's C1 feature channels (two 8-channel groups of one pooled
  // pixel per thread of waves 0-7 at C1 = 128: 8 sixteen-byte loads) and the label's table row (one piece per thread)
  const int l = labels[n];
  const bool ok = l >= 0 && l < V;
  constexpr int FI = 2;                                             // feature items per thread (64 pixels x C1 / 8 groups <= 2 x 512: host check)
  bf16x8 fv[FI][4];
  const int nfeat = HWp * cg1;
#pragma unroll
  for (int u = 0; u < FI; u++) {
    const int i = tid + u * 512;
    if (tid < 512 && i < nfeat) {
      const int g = i % cg1, p = i / cg1;
      const int pw = p % q.Wp, ph = p / q.Wp;
      const long hi = ((long)n * 2 * q.Hp + 2 * ph) * W + 2 * pw;
      fv[u][0] = *reinterpret_cast<const bf16x8*>(q.a + hi * q.C1 + g * 8);
      fv[u][1] = *reinterpret_cast<const bf16x8*>(q.a + (hi + 1) * q.C1 + g * 8);
      fv[u][2] = *reinterpret_cast<const bf16x8*>(q.a + (hi + W) * q.C1 + g * 8);
      fv[u][3] = *reinterpret_cast<const bf16x8*>(q.a + (hi + W + 1) * q.C1 + g * 8);
    }
  }
  const int cg2 = cg - cg1;
  bf16x8 tv;
#pragma unroll
  for (int e = 0; e < 8; e++) tv[e] = f2bf(0.f);
  const int tg = tid % cg2;
  if (ok) tv = *reinterpret_cast<const bf16x8*>(T + (long)l * C2 + tg * 8);
#pragma unroll
  for (int u = 0; u < FI; u++) {
    const int i = tid + u * 512;
    if (tid < 512 && i < nfeat) {
      const int g = i % cg1, p = i / cg1;
      bf16x8 m;
#pragma unroll
      for (int e = 0; e < 8; e++) m[e] = f2bf((bf2f(fv[u][0][e]) + bf2f(fv[u][2][e]) + bf2f(fv[u][1][e]) + bf2f(fv[u][3][e])) * 0.25f);   // label_pool_rider_block's arithmetic
      *reinterpret_cast<bf16x8*>(q.yp + ((long)n * HWp + p) * C + g * 8) = m;
      *reinterpret_cast<bf16x8*>(img + p * LSC_PITCH + g * 16) = m;
    }
  }
  for (int i = tid; i < HWp * cg2; i += 576) {                      // 576 % cg2 == 0 (host check): a thread keeps its group
    const int p = i / cg2;
    *reinterpret_cast<bf16x8*>(q.yp + ((long)n * HWp + p) * C + (cg1 + tg) * 8) = tv;
    *reinterpret_cast<bf16x8*>(img + p * LSC_PITCH + (cg1 + tg) * 16) = tv;
  }
  __syncthreads();
  if (wave >= 8) return;
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
  for (int kk = 0; kk < LSC_MAXC / 16; kk++)
    if (kk * 16 < C) {
      const bf16x8 fb = *reinterpret_cast<const bf16x8*>(img + (pt * 32 + r) * LSC_PITCH + kk * 32 + h * 16);
      acc = GANK_MFMA32(fa[kk], fb, acc);
    }
#pragma unroll
  for (int qq = 0; qq < 2; qq++) {
    const int co = ct * 32 + 16 * qq + 8 * h;
    float v[8];
    acc_widen(acc, qq, 1.0f, v);
    if (sc.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sc.bias + co), b1 = *reinterpret_cast<const f32x4*>(sc.bias + co + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x8*>(sc.out + ((long)n * HWp + pt * 32 + r) * sc.Cs + co) = o;
  }
}

// lists (optional, int32 [V][N + 1]): row v = {count, the samples of label v in ascending order} -- what the backward launches walk
// (built by the first channel block of every label: a deterministic rank per sample)
__global__ __launch_bounds__(576) void label_conv_table_kernel(const float* __restrict__ w, int Cin_total, int c0, int C2, int Cout,
                                                             const bf16* __restrict__ T, const float* __restrict__ bias, float* __restrict__ out,
                                                             const int* __restrict__ labels, int N, int V, int* __restrict__ lists, LabelPoolRider pr,
                                                             LabelShortcut sc) {
  __shared__ __attribute__((aligned(16))) char lds_[64 * LSC_PITCH];      // the table blocks' arrays / one pooled sample of a shortcut block
  float* r = reinterpret_cast<float*>(lds_);                               // [1024]
  float (*P)[64] = reinterpret_cast<float (*)[64]>(lds_ + 4096);            // [9][64]
  int* lab = reinterpret_cast<int*>(lds_ + 4096 + 9 * 64 * 4);              // [1024]
  static_assert(4096 + 9 * 64 * 4 + 4096 <= 64 * LSC_PITCH, "LDS plan");
  const int nco = (Cout + 63) / 64;
  if ((int)blockIdx.x >= V * nco) {
    if (sc.out) label_pool_shortcut_block(pr, sc, T, labels, C2, V, blockIdx.x - V * nco, lds_);
    else label_pool_rider_block(pr, T, labels, C2, V, blockIdx.x - V * nco, 576);
    return;
  }
  const int v = blockIdx.x / nco, cb = (blockIdx.x - v * nco) * 64;
  const int t = threadIdx.x >> 6, col = threadIdx.x & 63;
  if (lists && cb == 0) {
    for (int i = threadIdx.x; i < N; i += 576) {
      const int lb = labels[i];
      lab[i] = lb < 0 ? 0 : (lb >= V ? V - 1 : lb);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < N; i += 576)
      if (lab[i] == v) {
        int rank = 0;
        for (int j = 0; j < i; j++) rank += lab[j] == v ? 1 : 0;
        lists[(long)v * (N + 1) + 1 + rank] = i;
      }
    if (threadIdx.x == 0) {
      int cnt = 0;
      for (int j = 0; j < N; j++) cnt += lab[j] == v ? 1 : 0;
      lists[(long)v * (N + 1)] = cnt;
    }
  }
  const int co = cb + col < Cout ? cb + col : Cout - 1;
  for (int c = threadIdx.x; c < C2; c += 576) r[c] = fmaxf(bf2f(T[(long)v * C2 + c]), 0.f);
  __syncthreads();
  const float* wp = w + ((long)t * Cin_total + c0) * Cout + co;
  float acc = 0.f;
  for (int c = 0; c < C2; c += 16) {            // C2 % 16 == 0 (64 loads in flight measured slower: 13.8 against 9.4 us)
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; u++) x[u] = wp[(long)(c + u) * Cout];
#pragma unroll
    for (int u = 0; u < 16; u++) acc += bf2f(f2bf(x[u])) * r[c + u];
  }
  P[t][col] = acc;
  __syncthreads();
  const int cls = t;                            // the same 576 threads as (class, channel)
  float sum = 0.f;
#pragma unroll
  for (int tt = 0; tt < 9; tt++) sum += tap_valid(tt, cls) ? P[tt][col] : 0.f;
  if (cb + col < Cout) out[((long)v * 9 + cls) * Cout + cb + col] = sum + (bias ? bias[cb + col] : 0.f);
}

extern "C" int gank_label_conv3x3_table(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                        float* bias_table, const int32_t* labels, int N, int32_t* lists, void* stream) {
  GANK_REQUIRE(w && T && bias_table && V > 0 && C2 > 0 && C2 <= 1024 && C2 % 16 == 0 && c0 >= 0 && c0 + C2 <= Cin_total && Cout > 0,
               "label_conv3x3_table: bad arguments");
  GANK_REQUIRE(!lists || (labels && N > 0 && N <= 1024), "label_conv3x3_table: the sample lists need the labels of 1..1024 samples");
  hipLaunchKernelGGL(label_conv_table_kernel, dim3(V * ((Cout + 63) / 64)), dim3(576), 0, (hipStream_t)stream, w, Cin_total, c0, C2, Cout, (const bf16*)T, bias,
                     bias_table, labels, N, V, lists, LabelPoolRider{}, LabelShortcut{});
  GANK_LAUNCH_OK("label_conv3x3_table");
  return 0;
}
// ... with mean_pool2x2(concat(a, tile(T[labels]))) -> y_pooled [N, H/2, W/2, C1 + C2] (gank_concat_label_pool_fwd with y = NULL, bit for
// bit) computed by extra workgroups of the same launch: the two launches in front of D.Block.2 become one
extern "C" int gank_label_conv3x3_table_pooled(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                               float* bias_table, const int32_t* labels, int N, int32_t* lists, const void* a, void* y_pooled,
                                               int H, int W, int C1, void* stream) {
  GANK_REQUIRE(w && T && bias_table && labels && a && y_pooled && V > 0 && C2 > 0 && C2 <= 1024 && C2 % 16 == 0 && c0 >= 0 && c0 + C2 <= Cin_total && Cout > 0,
               "label_conv3x3_table_pooled: bad arguments");
  GANK_REQUIRE(N > 0 && N <= 1024 && C1 % 8 == 0 && H % 2 == 0 && W % 2 == 0, "label_conv3x3_table_pooled: 1..1024 samples, C1 %% 8 == 0, even sizes");
  LabelPoolRider pr{(const bf16*)a, (bf16*)y_pooled, (long)N * (H / 2) * (W / 2) * ((C1 + C2) / 8), H / 2, W / 2, C1, 0};
  long blocks = (pr.total8 + 575) / 576;
  pr.blocks = (int)(blocks > 2048 ? 2048 : blocks);
  hipLaunchKernelGGL(label_conv_table_kernel, dim3(V * ((Cout + 63) / 64) + pr.blocks), dim3(576), 0, (hipStream_t)stream, w, Cin_total, c0, C2, Cout,
                     (const bf16*)T, bias, bias_table, labels, N, V, lists, pr, LabelShortcut{});
  GANK_LAUNCH_OK("label_conv3x3_table_pooled");
  return 0;
}
// ... and the block's 1x1 shortcut conv on the pooled concatenation in the same launch (one workgroup per sample pools into LDS and
// multiplies from there): shortcut [N, H/2, W/2, Cs] = y_pooled x ws_f + bias_s with ws_f the plain-conv operand of the 1x1 filter
// (bf16 [Cs][ws_pitch], rows = output channels: gank_conv2d_prep_weights' wf).  (H/2) * (W/2) == 64, Cs == 128, C1 + C2 <= 256 and a
// multiple of 16.
extern "C" int gank_label_conv3x3_table_pooled_shortcut(const float* w, int Cin_total, int c0, int C2, int Cout, const void* T, int V, const float* bias,
                                                        float* bias_table, const int32_t* labels, int N, int32_t* lists, const void* a, void* y_pooled,
                                                        int H, int W, int C1, const void* ws_f, int ws_pitch, const float* bias_s, int Cs,
                                                        void* shortcut, void* stream) {
  GANK_REQUIRE(w && T && bias_table && labels && a && y_pooled && ws_f && shortcut && V > 0 && C2 > 0 && C2 <= 1024 && C2 % 16 == 0 && c0 >= 0 &&
               c0 + C2 <= Cin_total && Cout > 0, "label_conv3x3_table_pooled_shortcut: bad arguments");
  GANK_REQUIRE(N > 0 && N <= 1024 && C1 % 8 == 0 && H % 2 == 0 && W % 2 == 0 && (H / 2) * (W / 2) == 64 && Cs == 128 && (C1 + C2) % 16 == 0 &&
               C1 + C2 <= LSC_MAXC && ws_pitch >= C1 + C2 && ws_pitch % 8 == 0 && 64 * (C1 / 8) <= 1024 && 576 % (C2 / 8) == 0,
               "label_conv3x3_table_pooled_shortcut: 64 pooled pixels, 128 shortcut channels, at most %d input channels (multiple of 16)", LSC_MAXC);
  LabelPoolRider pr{(const bf16*)a, (bf16*)y_pooled, (long)N * (H / 2) * (W / 2) * ((C1 + C2) / 8), H / 2, W / 2, C1, N};
  const LabelShortcut sc{(const bf16*)ws_f, bias_s, (bf16*)shortcut, Cs, ws_pitch};
  hipLaunchKernelGGL(label_conv_table_kernel, dim3(V * ((Cout + 63) / 64) + N), dim3(576), 0, (hipStream_t)stream, w, Cin_total, c0, C2, Cout,
                     (const bf16*)T, bias, bias_table, labels, N, V, lists, pr, sc);
  GANK_LAUNCH_OK("label_conv3x3_table_pooled_shortcut");
  return 0;
}

// Sl[half][v][t][co] = sum over the samples n of label v and the pixels p of the row half where tap t is valid of dy[n, p, co].
// block = (label v, half of the rows, 64 output channels): it lists the label's samples in ascending order (deterministic), adds
// their rows per PIXEL first (thread = (8-channel group, pixel lane), four samples = 16 sixteen-byte loads in flight) and applies
// the nine tap masks once at the end; the pixel lanes meet in LDS in a fixed order.  The per-sample form of this pass (a block
// per sample, the label sums left to the consumer) made the consumer walk 128 samples per block: 12 + 29 us; this one feeds it
// V x 9 x Cout numbers.  H x W pixels per sample (H even, (H / 2) * W % 32 == 0), Cout % 64 == 0.
__global__ __launch_bounds__(256) void label_conv_tap_sums_kernel(const bf16* __restrict__ dy, const int* __restrict__ lists, float* __restrict__ S,
                                                                int N, int V, int H, int W, int Cout) {
  extern __shared__ __attribute__((aligned(16))) float red_[];
  label_conv_tap_sums_block(dy, lists, S, N, V, H, W, Cout, blockIdx.x, red_);
}

// block = (tap t, tile of CT channels of the constant half), 256 threads.  The per-tap sums enter only through their sums per LABEL
// (r and the relu mask depend on a sample through its label alone): thread co adds S[n][t][co] of the samples of each label (registers,
// predicated; loads in batches of 16), then
//   dw[t][c0 + c][co] += sum_v relu(T[v][c]) Sl[v][co]                                                   (this block is the only writer of its rows)
//   de_parts[t][first sample of label v][c] = [T[v][c] > 0] sum_co bf16(w[t][c0 + c][co]) Sl[v][co]      (zero for every other sample)
// -- the gradient of the tiled vector leaves SUMMED PER LABEL, parked at the label's first sample: its consumers (the label branch's
// dense layer and embedding table) add the samples of a label anyway.  (The first form kept S[:, t, :] in LDS and walked it per
// sample: 59 us, LDS-bound.)
__global__ __launch_bounds__(256) void label_conv_bwd_kernel(LabelBwdArgs q) {
  extern __shared__ __attribute__((aligned(16))) float sm_[];
  label_conv_bwd_block(q, blockIdx.x, sm_);
}

extern "C" long gank_label_conv3x3_bwd_ws_floats(int N, int Cout) { (void)N; return 2L * 16 * 9 * Cout; }      // [2 row halves][<= 16 labels][9][Cout]

extern "C" int gank_label_conv3x3_bwd(const void* dy, const int32_t* lists, const void* T, int V, const float* w, int Cin_total, int c0, int C2,
                                      int Cout, int N, int H, int W, float* dw, float* dw_feat_tmp, float* de_parts, float* ws, void* stream) {
  GANK_REQUIRE(lists && T && w && dw && de_parts && ws && N > 0 && H > 1 && W > 1 && V > 0, "label_conv3x3_bwd: bad arguments");      // dy NULL: the sums are in ws already
  GANK_REQUIRE(Cout % 8 == 0 && 256 % (Cout / 8) == 0 && C2 % LCB_CT == 0 && c0 >= 0 && c0 + C2 <= Cin_total, "label_conv3x3_bwd: unsupported channel counts");
  hipStream_t s = (hipStream_t)stream;
  GANK_REQUIRE(V <= LCB_V && H % 2 == 0 && ((H / 2) * W) % 32 == 0 && (H / 2) * W <= 128 && Cout % 64 == 0 && N <= 1024,
               "label_conv3x3_bwd: at most %d labels, 1024 samples, (H / 2) * W a multiple of 32 up to 128, Cout %% 64 == 0", LCB_V);
  const size_t lds1 = (size_t)32 * 9 * 64 * sizeof(float);
  GANK_MAX_DYNAMIC_LDS(label_conv_tap_sums_kernel, (int)lds1, "label_conv3x3_bwd");
  const size_t lds2 = ((size_t)V * (Cout + 4) + (size_t)LCB_CT * (Cout + 4) + (size_t)V * LCB_CT) * sizeof(float);
  GANK_REQUIRE(lds2 <= 64 * 1024, "label_conv3x3_bwd: N = %d, Cout = %d do not fit the LDS", N, Cout);
  if (dy) hipLaunchKernelGGL(label_conv_tap_sums_kernel, dim3(V * 2 * (Cout / 64)), dim3(256), lds1, s, (const bf16*)dy, lists, ws, N, V, H, W, Cout);
  const int merge_blocks = dw_feat_tmp ? 64 : 0;
  const LabelBwdArgs q{ws, (const bf16*)T, w, dw, de_parts, dw_feat_tmp, V, Cin_total, c0, C2, Cout, merge_blocks, 9 * (C2 / LCB_CT) + merge_blocks,
                       nullptr, nullptr, 0, 0, 0, 0, 0};
  hipLaunchKernelGGL(label_conv_bwd_kernel, dim3(q.blocks), dim3(256), lds2, s, q);
  GANK_LAUNCH_OK("label_conv3x3_bwd");
  return 0;
}

// the second launch alone (the tap sums are in `tap_sums` already: gank_conv2d_wgrad_slabs_rows_tap_sums) with a TENTH part computed by
// extra blocks: de_parts [10][V][C2], part 9 = the gradient the tiled vector receives through the POOLED shortcut branch, summed per
// label -- sum over the samples of label v and the Hp x Wp pooled pixels of g_pooled[n][p][c0g + c] (g_pooled [N,Hp,Wp,pitch]: each of
// the four pixels under a pooled pixel receives 0.25 g, so the sum over the sample is the sum of the pooled gradient).  With it the
// join launch of the two branches (gank_concat_label_unpool_bwd_factored) has nothing left to do when the feature channels join in
// gank_img16_conv3x3_dgrad_unpool; the consumer is gank_label_dense_bwd_parts.
extern "C" int gank_label_conv3x3_bwd_pooled(const float* tap_sums, const int32_t* lists, const void* T, int V, const float* w, int Cin_total, int c0,
                                             int C2, int Cout, int N, float* dw, float* dw_feat_tmp, float* de_parts, const void* g_pooled, int HWp,
                                             int pitch, int c0g, void* stream) {
  GANK_REQUIRE(tap_sums && lists && T && w && dw && de_parts && g_pooled && N > 0 && N <= 1024 && V > 0 && V <= LCB_V && HWp > 0,
               "label_conv3x3_bwd_pooled: bad arguments");
  GANK_REQUIRE(Cout % 4 == 0 && Cout <= 256 && C2 % 32 == 0 && c0 >= 0 && c0 + C2 <= Cin_total && pitch % 8 == 0 && c0g % 8 == 0 && c0g + C2 <= pitch,
               "label_conv3x3_bwd_pooled: unsupported channel counts");
  size_t lds2 = ((size_t)V * (Cout + 4) + (size_t)LCB_CT * (Cout + 4) + (size_t)V * LCB_CT) * sizeof(float);
  if (lds2 < 64 * 4 * 8 * sizeof(float)) lds2 = 64 * 4 * 8 * sizeof(float);
  GANK_REQUIRE(lds2 <= 64 * 1024, "label_conv3x3_bwd_pooled: Cout = %d does not fit the LDS", Cout);
  const int merge_blocks = dw_feat_tmp ? 64 : 0, pool_blocks = V * (C2 / 32);
  const LabelBwdArgs q{tap_sums, (const bf16*)T, w, dw, de_parts, dw_feat_tmp, V, Cin_total, c0, C2, Cout, merge_blocks,
                       9 * (C2 / LCB_CT) + merge_blocks + pool_blocks, (const bf16*)g_pooled, lists, N, HWp, pitch, c0g, pool_blocks};
  hipLaunchKernelGGL(label_conv_bwd_kernel, dim3(q.blocks), dim3(256), lds2, (hipStream_t)stream, q);
  GANK_LAUNCH_OK("label_conv3x3_bwd_pooled");
  return 0;
}
