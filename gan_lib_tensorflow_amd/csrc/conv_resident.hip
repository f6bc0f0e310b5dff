// "Resident" convolutions for the critic's small layers (gfx950): the whole input of a workgroup -- every channel of
// its pixels plus the halo -- sits in LDS for the whole kernel, the weights stream from L2 straight into registers in
// MFMA-fragment order, and there is NO barrier inside a convolution.
//
// Why: the critic works on 128 samples per update.  Its 8x8x128 residual blocks (D.Block.3 / D.Block.4,
// SNGAN/gan_cifar_resnet.py:156-209 with resample=None) are 2.4 GFLOP per conv; as four implicit-GEMM launches per
// block pair they ran 10-13 us each (one 4-wave workgroup per CU, a barrier and a dependent global->LDS round trip
// per 64-deep K-step, 0.08 of the MFMA peak, 2.5x the algorithmic HBM bytes).  An 8x8x128 sample is 16 KB, so one
// workgroup per sample keeps  relu -> conv1 -> relu -> conv2 -> + shortcut  of BOTH blocks (and the final
// relu + spatial mean) on chip: the activations make one trip to HBM per tensor the backward pass needs, the 24
// launches per critic pass become 2 (forward chain, backward chain).
//
// Layout: LDS image = 10 x 10 halo pixels, pixel pitch 272 B (128 channels + 16 B), row pitch 2944 B; with these
// pitches the 16-lane groups of a ds_read_b128 B-fragment read (32 lanes = 4 image rows x 8 columns, one tap) fall on
// 16 distinct 16-byte bank slots.  The border stays zero (SAME padding); images hold RAW values and the relu is
// applied to the fragment after the read (v_pk_max_i16), so the shortcut operand and the tensors saved for the
// backward pass come from the same image.  Weight operand ("rfrag", prep kind 4): [32-row tile][tap][k/16][lane][8]
// bf16, one coalesced 1 KB request per MFMA A-fragment, a register ring of 8 requests in flight per wave.
// Wave ct of the 4: output channels 32*ct .. +31 of all 64 pixels (two MFMA tiles: image rows 0-3, 4-7).
#include "gank_common.h"
#include "label_conv_dev.h"
#ifdef GANK_TUNING
// timing-only experiment (GANK_STATS_DBG=1): the statistics epilogues skip their atomics (set once per process, before the first launch)
static __device__ int gank_stats_dbg = 0;
static void gank_stats_dbg_init() {
  static const int v = gank_tune("GANK_STATS_DBG", 0);
  static bool done = false;
  if (!done) { done = true; if (v) (void)hipMemcpyToSymbol(HIP_SYMBOL(gank_stats_dbg), &v, sizeof(int)); }
}
#else
static inline void gank_stats_dbg_init() {}
#endif
#include <stdlib.h>

namespace {
constexpr int RB_C = 128;                    // channels
constexpr int RB_PPB = RB_C * 2 + 16;        // pixel pitch (bytes): 17 sixteen-byte units
constexpr int RB_RPB = 2944;                 // halo row pitch (bytes): 184 units = 8 mod 16
constexpr int RB_IMG = 10 * RB_RPB;          // one halo image
constexpr int RB_STEPS = 9 * (RB_C / 16);    // MFMA K-steps per conv (tap-major)
constexpr int RB_WBYTES = RB_C * 9 * RB_C * 2;
constexpr int RB_LDS = 2 * RB_IMG;

struct ResFwdArgs {
  const bf16* x;          // [N,8,8,C] chain input
  const bf16* w[4];       // rfrag fprop operands: block0.conv1, block0.conv2, block1.conv1, block1.conv2
  const float* bias[4];   // may be null
  bf16* h1[2];            // conv1 output (bias added, before the relu) per block; null = not kept
  bf16* y[2];             // block outputs; null = not kept
  bf16* pooled;           // optional [N,C]: mean over the 64 pixels of relu(y_last)   (gan_cifar_resnet.py:299-301)
  // optional fused critic head (with pooled): logits[n] = bf16(pooled[n] . head_w + head_b[0])   (D.Output, :303-304)
  const float* head_w;    // [C] fp32 (the spectrally normalised D.Output weight)
  const float* head_b;    // [1] fp32 or null
  bf16* logits;           // [N] or null (no head)
  int N, nblocks;
};

struct ResBwdArgs {
  const bf16* dy;         // [N,8,8,C] gradient of the chain output, or null with dpool
  const bf16* dpool;      // optional [N,C]: gradient of `pooled`; dy = dpool/64 * [y_last > 0]
  const bf16* ylast;      // y of the last block (with dpool)
  bf16* dy_out;           // optional: where the dy built from dpool is kept (operand of the last conv2's filter gradient)
  const bf16* wd[4];      // rfrag dgrad operands in the order applied: last block conv2, conv1, then the block before
  const bf16* h1[2];      // relu masks, same order
  const bf16* xin[2];     // block inputs (mask of the pre-activation relu), same order
  bf16* g1[2];            // out: gradient of conv1's output (dy operand of conv1's filter gradient); null = not kept
  bf16* dx[2];            // out: gradient of the block input; null = not kept (the last one is the chain's result)
  // optional fused critic head (instead of dpool): d loss / d pooled[n][c] = bf16(dl[n] * head_w[c]) with dl[n] the hinge
  // derivative at logits[n] (gan_cifar_resnet.py:379-381 mode 0, :492 mode 1), exactly gank_critic_head_hinge's arithmetic;
  // workgroup N of the grid (one more than samples) writes the loss and accumulates the layer's weight / bias gradients
  const bf16* logits;     // [N] from the forward launch, or null
  const float* head_w;    // [C]
  const bf16* pooled;     // [N,C] from the forward launch (weight gradient operand)
  float* loss;            // [1]
  float* w_grad;          // [C] accumulated, or null
  float* b_grad;          // [1] accumulated, or null
  int n_real, mode;
  float loss_scale;
  int N, nblocks;
};

// hinge derivative / loss term of one logit: the arithmetic of critic_head_hinge_kernel (loss_opt.hip)
__device__ __forceinline__ void res_head_term(float v, int m, int M, int n_real, int mode, float loss_scale, float& dl, float& l) {
  const int n_fake = M - n_real;
  float d;
  if (mode == 1) { l = -v / (float)M; d = -1.f / (float)M; }
  else if (m < n_real) { const float u = 1.f - v; l = fmaxf(u, 0.f) / (float)n_real; d = u > 0.f ? -1.f / (float)n_real : 0.f; }
  else { const float u = 1.f + v; l = fmaxf(u, 0.f) / (float)n_fake; d = u > 0.f ? 1.f / (float)n_fake : 0.f; }
  dl = bf2f(f2bf(d * loss_scale));
}

// TPW = 32-pixel MFMA tiles per wave (1: 8 waves per sample, 2: 4 waves, every weight fragment feeds two MFMAs);
// PF = weight fragments in flight per wave.  A wave consumes one 1 KB fragment per TPW x 32 MFMA cycles and an L2 hit
// takes 500+ cycles under load, so PF x TPW x 32 must cover that: 8 in flight left the matrix pipe waiting.
// The weight stream never stops: the ring is filled once (res_ring_fill) and the last PF steps of a conv already
// request the first PF fragments of the NEXT conv (`rn`), so the L2 round trip of a conv's first fragments hides behind
// the previous conv's tail, its epilogue and the barrier instead of opening every conv with an idle matrix pipe.
template <int PF>
__device__ __forceinline__ void res_ring_fill(u32x4 (&ring)[PF], const __amdgpu_buffer_rsrc_t rw, int lane16, int wbase) {
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, wbase + s * 1024, 0);
}
// S0, NS: this wave's share of the K-steps (all of them, or one half when the reduction is split over two wave groups:
// `wbase` then already points at step S0 of the wave's fragment stream)
template <bool RELU, int TPW, int PF, int S0 = 0, int NS = RB_STEPS>
__device__ __forceinline__ void res_conv3x3(f32x16 (&acc)[TPW], u32x4 (&ring)[PF], const char* smem, int b_base, const __amdgpu_buffer_rsrc_t rw,
                                            const __amdgpu_buffer_rsrc_t rn, bool has_next, int lane16, int wbase) {
  static_assert(NS % PF == 0, "the ring position must be the same at the start of every conv");
  constexpr int PB = 2;                              // pixel fragments are read PB steps ahead of their MFMAs
  u32x4 bq[PB + 1][TPW];
  auto read_b = [&](int s, u32x4 (&dst)[TPW]) {
    const int tap = (S0 + s) >> 3, kk = (S0 + s) & 7;
#pragma unroll
    for (int t = 0; t < TPW; t++)
      dst[t] = *reinterpret_cast<const u32x4*>(smem + b_base + (4 * t + tap / 3) * RB_RPB + (tap % 3) * RB_PPB + kk * 32);
  };
#pragma unroll
  for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
  for (int s = 0; s < NS; s++) {
    // sched_barrier: hipcc otherwise sinks the LDS reads to just in front of their MFMAs (lgkmcnt(0) before every pair,
    // the whole LDS latency exposed) and lets only ~5 weight requests stay in flight
    if (s + PB < NS) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
    const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      u32x4 bv = bq[s % (PB + 1)][t];
      if constexpr (RELU) bv = relu_bf16x8(bv);
      acc[t] = GANK_MFMA32(fa, __builtin_bit_cast(bf16x8, bv), acc[t]);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s + PF < NS) ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane16, wbase + (s + PF) * 1024, 0);
    else if (has_next) ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rn, lane16, wbase + (s + PF - NS) * 1024, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
}
// one conv of a chain kernel: KS == 1 -> every wave runs all K-steps; KS == 2 -> wave group kg runs its half (a wave-uniform branch
// between two fully unrolled bodies, so the LDS offsets stay immediates), the two partial tiles meet in `part` (LDS behind the
// images: [4 channel tiles][2 * TPW * 4 quads][64 lanes] x 16 B, consecutive lanes contiguous): group 1 parks its sums, a
// barrier, group 0 adds them -- after this call only kg == 0 holds the result
template <bool RELU, int TPW, int PF, int KS>
__device__ __forceinline__ void res_conv3x3_ks(f32x16 (&acc)[TPW], u32x4 (&ring)[PF], const char* smem, int b_base, const __amdgpu_buffer_rsrc_t rw,
                                               const __amdgpu_buffer_rsrc_t rn, bool has_next, int lane16, int wbase, int kg, f32x4* part, int lane) {
  if constexpr (KS == 1) {
    res_conv3x3<RELU, TPW, PF>(acc, ring, smem, b_base, rw, rn, has_next, lane16, wbase);
  } else {
    constexpr int NS = RB_STEPS / 2;
    if (kg == 0) res_conv3x3<RELU, TPW, PF, 0, NS>(acc, ring, smem, b_base, rw, rn, has_next, lane16, wbase);
    else res_conv3x3<RELU, TPW, PF, NS, NS>(acc, ring, smem, b_base, rw, rn, has_next, lane16, wbase);
    if (kg == 1) {
#pragma unroll
      for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) part[(t * 4 + g) * 64 + lane] = f32x4{acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]};
    }
    __syncthreads();
    if (kg == 0) {
#pragma unroll
      for (int t = 0; t < TPW; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 o = part[(t * 4 + g) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; e++) acc[t][4 * g + e] += o[e];
        }
    }
  }
}

// interior of an image <-> [64 pixels][C] in HBM, 16 bytes per lane, whole 256-byte pixel rows
template <int NT>
__device__ __forceinline__ void res_store_image(const char* img, bf16* dst, int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / NT; it++) {
    const int q = tid + it * NT, px = q >> 4, c16 = q & 15;
    *reinterpret_cast<u32x4*>(dst + px * RB_C + c16 * 8) =
        *reinterpret_cast<const u32x4*>(img + ((px >> 3) + 1) * RB_RPB + ((px & 7) + 1) * RB_PPB + c16 * 16);
  }
}
template <int NT>
__device__ __forceinline__ void res_load_image(char* img, const bf16* src, int tid) {
#pragma unroll
  for (int it = 0; it < 1024 / NT; it++) {
    const int q = tid + it * NT, px = q >> 4, c16 = q & 15;
    *reinterpret_cast<u32x4*>(img + ((px >> 3) + 1) * RB_RPB + ((px & 7) + 1) * RB_PPB + c16 * 16) =
        *reinterpret_cast<const u32x4*>(src + px * RB_C + c16 * 8);
  }
}
}  // namespace

// KS = 2 (with TPW = 2): the 8 waves are (4 channel tiles) x (2 halves of the reduction) instead of x (2 pixel tiles): every
// weight fragment is then requested by ONE wave of the workgroup and feeds two MFMAs, instead of being requested by two waves
// for one MFMA each -- the CU's L1 path carried every weight byte twice, and that path (not L2, not the matrix pipe) bounds
// these kernels.  The halves meet in LDS after each conv (res_conv3x3_ks).
template <int TPW, int PF, int KS>
__global__ __launch_bounds__(512 / TPW * KS) void res8_chain_fwd_kernel(ResFwdArgs a) {
  constexpr int NT = 512 / TPW * KS;
  static_assert(KS == 1 || TPW == 2, "the K split needs both pixel tiles in one wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3;                                                        // 32 output channels per wave
  const int kg = KS == 2 ? wave >> 2 : 0;
  const int r = lane & 31, h = lane >> 5;
  const long n = blockIdx.x;
  const int row = (KS == 2 ? 0 : 4 * (wave >> 2)) + (r >> 3), col = r & 7;        // pixel of this wave's first tile; tile t = 4 t rows below
  const int b_base = row * RB_RPB + col * RB_PPB + h * 16;                        // tap (0,0), kk 0 of this lane's pixel
  const int own = (row + 1) * RB_RPB + (col + 1) * RB_PPB + ct * 64 + h * 8;      // this lane's 4 channels of quad g: + 16 g (+ 4 rows for tile 1)
  const int wbase = (ct * RB_STEPS + kg * (RB_STEPS / 2)) * 1024;
  f32x4* part = reinterpret_cast<f32x4*>(smem + RB_LDS) + ct * (TPW * 4 * 64);

  u32x4 ring[PF];
  res_ring_fill<PF>(ring, __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w[0]), 0, RB_WBYTES, 0x00020000), lane * 16, wbase);
  for (int i = tid; i < RB_LDS / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  res_load_image<NT>(smem, a.x + n * 64 * RB_C, tid);
  __syncthreads();

#pragma unroll 1
  for (int b = 0; b < a.nblocks; b++) {
    // static indexing of the by-value argument arrays (a dynamic index spills the struct to scratch)
    const bf16* w1 = b == 0 ? a.w[0] : a.w[2];
    const bf16* w2 = b == 0 ? a.w[1] : a.w[3];
    const float* bias1 = b == 0 ? a.bias[0] : a.bias[2];
    const float* bias2 = b == 0 ? a.bias[1] : a.bias[3];
    bf16* h1 = b == 0 ? a.h1[0] : a.h1[1];
    bf16* y = b == 0 ? a.y[0] : a.y[1];
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(w1), 0, RB_WBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(w2), 0, RB_WBYTES, 0x00020000);
    const bool more = b + 1 < a.nblocks;             // a.w[2] is the next block's conv_1 (a valid pointer is needed only then)
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(more ? a.w[2] : w2), 0, RB_WBYTES, 0x00020000);

    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    res_conv3x3_ks<true, TPW, PF, KS>(acc, ring, smem, b_base, r1, r2, true, lane * 16, wbase, kg, part, lane);   // conv_1(relu(x))      (:186-190)
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (bias1) bb = *reinterpret_cast<const f32x4*>(bias1 + ct * 32 + 8 * g + 4 * h);
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = f2bf(acc[t][4 * g + e] + bb[e]);
        *reinterpret_cast<bf16x4*>(smem + RB_IMG + own + 4 * t * RB_RPB + 16 * g) = o;
      }
    }
    __syncthreads();                                                              // image B = h1 complete
    if (h1) res_store_image<NT>(smem + RB_IMG, h1 + n * 64 * RB_C, tid);
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    res_conv3x3_ks<true, TPW, PF, KS>(acc, ring, smem + RB_IMG, b_base, r2, r3, more, lane * 16, wbase, kg, part, lane);   // conv_2(relu(h1))     (:198-207)
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        f32x4 bb = {0.f, 0.f, 0.f, 0.f};
        if (bias2) bb = *reinterpret_cast<const f32x4*>(bias2 + ct * 32 + 8 * g + 4 * h);
        const bf16x4 xs = *reinterpret_cast<const bf16x4*>(smem + own + 4 * t * RB_RPB + 16 * g);    // shortcut + output   (:209)
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = f2bf(acc[t][4 * g + e] + bb[e] + bf2f(xs[e]));
        *reinterpret_cast<bf16x4*>(smem + own + 4 * t * RB_RPB + 16 * g) = o;     // own elements only: no other wave reads or writes them here
      }
    }
    __syncthreads();                                                              // image A = y complete
    if (y) res_store_image<NT>(smem, y + n * 64 * RB_C, tid);
  }
  float* sp = reinterpret_cast<float*>(smem + RB_IMG);                            // image B is dead here: [C] pooled values for the head
  if (a.pooled && tid < RB_C) {                                                   // relu + mean over the 8 x 8 pixels (:299-301)
    float s = 0.f;
    for (int px = 0; px < 64; px++)
      s += fmaxf(bf2f(*reinterpret_cast<const bf16*>(smem + ((px >> 3) + 1) * RB_RPB + ((px & 7) + 1) * RB_PPB + tid * 2)), 0.f);
    const bf16 pv = f2bf(s * (1.f / 64.f));
    a.pooled[n * RB_C + tid] = pv;
    if (a.logits) sp[tid] = bf2f(pv);
  }
  if (a.logits) {                                                                 // D.Output on the pooled row (:303-304): one wave, the
    __syncthreads();                                                              // summation order of critic_head_hinge_kernel
    if (wave == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < RB_C; k += 64) t += sp[k + lane] * a.head_w[k + lane];
      t = wave_sum(t);
      if (lane == 0) a.logits[n] = f2bf(t + (a.head_b ? a.head_b[0] : 0.f));
    }
  }
}

template <int TPW, int PF, int KS>
__global__ __launch_bounds__(512 / TPW * KS) void res8_chain_bwd_kernel(ResBwdArgs a) {
  constexpr int NT = 512 / TPW * KS;
  static_assert(KS == 1 || TPW == 2, "the K split needs both pixel tiles in one wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3;
  const int kg = KS == 2 ? wave >> 2 : 0;
  const int r = lane & 31, h = lane >> 5;
  const long n = blockIdx.x;
  if (a.logits && n == a.N) {                        // the workgroup behind the samples: loss value, D.Output's weight / bias gradient
    float* s_dl = reinterpret_cast<float*>(smem);    // [N]
    float* red = s_dl + a.N;                         // [16]
    float* part = red + 16;                          // [8][C]
    float acc = 0.f;
    for (int m = tid; m < a.N; m += NT) {
      float dl, l;
      res_head_term(bf2f(a.logits[m]), m, a.N, a.n_real, a.mode, a.loss_scale, dl, l);
      s_dl[m] = dl;
      acc += l;
    }
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      float t = 0.f;
      for (int i = 0; i < NT / 64; i++) t += red[i];
      a.loss[0] = t;
    }
    if (a.w_grad) {                                  // w_grad[k] += sum_m pooled[m][k] dl[m]: 8 row slices per column, summed in slice order
      const int nsl = 8, per = (a.N + nsl - 1) / nsl;
      for (int i = tid; i < nsl * RB_C; i += NT) {
        const int sl = i / RB_C, k = i - sl * RB_C;
        const int m0 = sl * per, m1 = min(a.N, m0 + per);
        float t = 0.f;
        for (int m = m0; m < m1; m++) t += bf2f(a.pooled[(long)m * RB_C + k]) * s_dl[m];
        part[i] = t;
      }
      __syncthreads();
      if (tid < RB_C) {
        float t = 0.f;
#pragma unroll
        for (int sl = 0; sl < 8; sl++) t += part[sl * RB_C + tid];
        a.w_grad[tid] += t;
      }
    }
    if (a.b_grad && tid < 64) {
      float t = 0.f;
      for (int m = tid; m < a.N; m += 64) t += s_dl[m];
      t = wave_sum(t);
      if (tid == 0) a.b_grad[0] += t;
    }
    return;
  }
  const int row = (KS == 2 ? 0 : 4 * (wave >> 2)) + (r >> 3), col = r & 7;
  const int b_base = row * RB_RPB + col * RB_PPB + h * 16;
  const int own = (row + 1) * RB_RPB + (col + 1) * RB_PPB + ct * 64 + h * 8;
  const long own_g = (n * 64 + row * 8 + col) * RB_C + ct * 32 + 4 * h;           // the same elements in HBM: + 8 g (+ 32 pixels for tile 1)
  const int wbase = (ct * RB_STEPS + kg * (RB_STEPS / 2)) * 1024;
  f32x4* part = reinterpret_cast<f32x4*>(smem + RB_LDS) + ct * (TPW * 4 * 64);

  u32x4 ring[PF];
  res_ring_fill<PF>(ring, __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.wd[0]), 0, RB_WBYTES, 0x00020000), lane * 16, wbase);
  for (int i = tid; i < RB_LDS / 16; i += NT) reinterpret_cast<u32x4*>(smem)[i] = u32x4{0u, 0u, 0u, 0u};
  __syncthreads();
  if (a.dpool || a.logits) {                         // gradient of relu + spatial mean: dpool / 64 where y_last > 0
    float dl = 0.f, lterm;
    if (a.logits) res_head_term(bf2f(a.logits[n]), (int)n, a.N, a.n_real, a.mode, a.loss_scale, dl, lterm);
#pragma unroll
    for (int it = 0; it < 1024 / NT; it++) {
      const int q = tid + it * NT, px = q >> 4, c16 = q & 15;
      const bf16x8 yv = *reinterpret_cast<const bf16x8*>(a.ylast + (n * 64 + px) * RB_C + c16 * 8);
      bf16x8 dp;
      if (a.logits) {                                // the head's input gradient, bf16(dl * w) as the head kernel rounds it
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.head_w + c16 * 8), w1 = *reinterpret_cast<const f32x4*>(a.head_w + c16 * 8 + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { dp[e] = f2bf(dl * w0[e]); dp[4 + e] = f2bf(dl * w1[e]); }
      } else {
        dp = *reinterpret_cast<const bf16x8*>(a.dpool + n * RB_C + c16 * 8);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(yv[e]) > 0.f ? bf2f(dp[e]) * (1.f / 64.f) : 0.f);
      *reinterpret_cast<bf16x8*>(smem + ((px >> 3) + 1) * RB_RPB + ((px & 7) + 1) * RB_PPB + c16 * 16) = o;
      if (a.dy_out) *reinterpret_cast<bf16x8*>(a.dy_out + (n * 64 + px) * RB_C + c16 * 8) = o;
    }
  } else {
    res_load_image<NT>(smem, a.dy + n * 64 * RB_C, tid);
  }
  __syncthreads();

#pragma unroll 1
  for (int b = 0; b < a.nblocks; b++) {
    const bf16* wd2 = b == 0 ? a.wd[0] : a.wd[2];
    const bf16* wd1 = b == 0 ? a.wd[1] : a.wd[3];
    const bf16* h1 = b == 0 ? a.h1[0] : a.h1[1];
    const bf16* xin = b == 0 ? a.xin[0] : a.xin[1];
    bf16* g1 = b == 0 ? a.g1[0] : a.g1[1];
    bf16* dx = b == 0 ? a.dx[0] : a.dx[1];
    const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(wd2), 0, RB_WBYTES, 0x00020000);
    const __amdgpu_buffer_rsrc_t r1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(wd1), 0, RB_WBYTES, 0x00020000);
    const bool more = b + 1 < a.nblocks;
    const __amdgpu_buffer_rsrc_t r3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(more ? a.wd[2] : wd1), 0, RB_WBYTES, 0x00020000);

    bf16x4 m[TPW][4];
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) m[t][g] = *reinterpret_cast<const bf16x4*>(h1 + own_g + t * 32 * RB_C + 8 * g);     // in flight during the conv
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    res_conv3x3_ks<false, TPW, PF, KS>(acc, ring, smem, b_base, r2, r1, true, lane * 16, wbase, kg, part, lane);   // input gradient of conv_2
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = f2bf(bf2f(m[t][g][e]) > 0.f ? acc[t][4 * g + e] : 0.f);       // relu'(h1)
        *reinterpret_cast<bf16x4*>(smem + RB_IMG + own + 4 * t * RB_RPB + 16 * g) = o;
      }
    }
    __syncthreads();                                                              // image B = g1 complete
    if (g1) res_store_image<NT>(smem + RB_IMG, g1 + n * 64 * RB_C, tid);
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) m[t][g] = *reinterpret_cast<const bf16x4*>(xin + own_g + t * 32 * RB_C + 8 * g);
    }
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    res_conv3x3_ks<false, TPW, PF, KS>(acc, ring, smem + RB_IMG, b_base, r1, r3, more, lane * 16, wbase, kg, part, lane);   // input gradient of conv_1
    if (kg == 0) {
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const bf16x4 ds = *reinterpret_cast<const bf16x4*>(smem + own + 4 * t * RB_RPB + 16 * g);    // + dy along the identity shortcut
        bf16x4 o;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = f2bf((bf2f(m[t][g][e]) > 0.f ? acc[t][4 * g + e] : 0.f) + bf2f(ds[e]));
        *reinterpret_cast<bf16x4*>(smem + own + 4 * t * RB_RPB + 16 * g) = o;
      }
    }
    __syncthreads();                                                              // image A = dx complete
    if (dx) res_store_image<NT>(smem, dx + n * 64 * RB_C, tid);
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------
#ifdef GANK_TUNING
static int res8_cfg() {          // experiment knob GANK_RES8_CFG = 10*TPW + {1: 8, 2: 12, 3: 24 fragments in flight}; + 100: reduction split over two wave groups
  static const int v = gank_tune("GANK_RES8_CFG", 122);
  return v;
}
#endif
constexpr int RB_PART = 4 * 2 * 4 * 64 * 16;      // partial tiles of the K-split form: [4 channel tiles][2 pixel tiles x 4 quads][64 lanes] x 16 B
template <int TPW, int PF, int KS>
static int res8_launch_fwd(const ResFwdArgs& a, hipStream_t s) {
  constexpr int LDS = RB_LDS + (KS == 2 ? RB_PART : 0);
  GANK_MAX_DYNAMIC_LDS((res8_chain_fwd_kernel<TPW, PF, KS>), LDS, "res8_chain_fwd");
  hipLaunchKernelGGL((res8_chain_fwd_kernel<TPW, PF, KS>), dim3(a.N), dim3(512 / TPW * KS), LDS, s, a);
  return 0;
}
template <int TPW, int PF, int KS>
static int res8_launch_bwd(const ResBwdArgs& a, hipStream_t s) {
  constexpr int LDS = RB_LDS + (KS == 2 ? RB_PART : 0);
  GANK_MAX_DYNAMIC_LDS((res8_chain_bwd_kernel<TPW, PF, KS>), LDS, "res8_chain_bwd");
  hipLaunchKernelGGL((res8_chain_bwd_kernel<TPW, PF, KS>), dim3(a.N + (a.logits ? 1 : 0)), dim3(512 / TPW * KS), LDS, s, a);
  return 0;
}
#ifdef GANK_TUNING
#define RES8_DISPATCH(fn, a, s)                                      \
  switch (res8_cfg()) {                                              \
    case 11: rc = fn<1, 8, 1>(a, s); break;                          \
    case 12: rc = fn<1, 12, 1>(a, s); break;                         \
    case 13: rc = fn<1, 24, 1>(a, s); break;                         \
    case 21: rc = fn<2, 8, 1>(a, s); break;                          \
    case 22: rc = fn<2, 12, 1>(a, s); break;                         \
    case 23: rc = fn<2, 24, 1>(a, s); break;                         \
    case 121: rc = fn<2, 6, 2>(a, s); break;                         \
    case 123: rc = fn<2, 18, 2>(a, s); break;                        \
    default: rc = fn<2, 12, 2>(a, s); break;                         \
  }
#else          // (4 channel tiles) x (2 halves of the reduction), two pixel tiles per wave, 12 weight fragments in flight
#define RES8_DISPATCH(fn, a, s) rc = fn<2, 12, 2>(a, s)
#endif

extern "C" int gank_res8_chain_fwd_head(const void* x, const void* const* w_rfrag, const float* const* bias, void* const* h1,
                                        void* const* y, void* pooled, const float* head_w, const float* head_b, void* logits, int N, int C,
                                        int nblocks, void* stream) {
  GANK_REQUIRE(x && w_rfrag && bias && h1 && y && N > 0, "res8_chain_fwd: null pointer");
  GANK_REQUIRE(!logits || (pooled && head_w), "res8_chain_fwd: the fused head needs the pooled output and the head weight");
  GANK_REQUIRE(C == RB_C, "res8_chain_fwd: built for %d channels (got %d)", RB_C, C);
  GANK_REQUIRE(nblocks == 1 || nblocks == 2, "res8_chain_fwd: 1 or 2 blocks per launch (got %d)", nblocks);
  ResFwdArgs a{};
  a.x = (const bf16*)x; a.pooled = (bf16*)pooled; a.N = N; a.nblocks = nblocks;
  a.head_w = head_w; a.head_b = head_b; a.logits = (bf16*)logits;
  for (int i = 0; i < 2 * nblocks; i++) {
    GANK_REQUIRE(w_rfrag[i], "res8_chain_fwd: null weight operand %d", i);
    a.w[i] = (const bf16*)w_rfrag[i];
    a.bias[i] = bias[i];
  }
  for (int b = 0; b < nblocks; b++) { a.h1[b] = (bf16*)h1[b]; a.y[b] = (bf16*)y[b]; }
  GANK_REQUIRE(a.y[nblocks - 1] || a.pooled, "res8_chain_fwd: no output requested");
  hipStream_t s = (hipStream_t)stream;
  // algorithmic bytes: x, every weight once, every tensor written once
  gank_prof_begin(0, 2.0 * nblocks * 2.0 * N * 64.0 * RB_C * 9.0 * RB_C, s,
                  2.0 * N * 64.0 * RB_C * (1 + 2 * nblocks) + 2.0 * nblocks * RB_WBYTES);
  gank_prof_tag(0, "res8_chain_fwd_kernel");
  int rc = 0;
  RES8_DISPATCH(res8_launch_fwd, a, s);
  gank_prof_end(0, s);
  if (rc) return rc;
  GANK_LAUNCH_OK("res8_chain_fwd");
  return 0;
}
extern "C" int gank_res8_chain_fwd(const void* x, const void* const* w_rfrag, const float* const* bias, void* const* h1,
                                   void* const* y, void* pooled, int N, int C, int nblocks, void* stream) {
  return gank_res8_chain_fwd_head(x, w_rfrag, bias, h1, y, pooled, nullptr, nullptr, nullptr, N, C, nblocks, stream);
}

static int res8_chain_bwd_impl(const void* dy, const void* dpool, const void* ylast, void* dy_out, const void* const* wd_rfrag,
                               const void* const* h1, const void* const* xin, void* const* g1, void* const* dx, int N, int C,
                               int nblocks, const gank_res8_head* head, void* stream) {
  GANK_REQUIRE((dy || ((dpool || head) && ylast)) && wd_rfrag && h1 && xin && g1 && dx && N > 0, "res8_chain_bwd: null pointer");
  if (head) {
    GANK_REQUIRE(!dy && !dpool, "res8_chain_bwd_head: the head replaces dy / dpool");
    GANK_REQUIRE(head->logits && head->head_w && head->pooled && head->loss, "res8_chain_bwd_head: null head tensor");
    GANK_REQUIRE(head->mode == 0 || head->mode == 1, "res8_chain_bwd_head: mode 0 (hinge_d) or 1 (hinge_g)");
    GANK_REQUIRE(head->mode == 1 || (head->n_real > 0 && head->n_real < N), "res8_chain_bwd_head: n_real must split the batch");
    GANK_REQUIRE(head->loss_scale > 0.f, "res8_chain_bwd_head: loss_scale must be positive");
    GANK_REQUIRE(N <= 8192, "res8_chain_bwd_head: at most 8192 samples (got %d)", N);
  }
  GANK_REQUIRE(C == RB_C, "res8_chain_bwd: built for %d channels (got %d)", RB_C, C);
  GANK_REQUIRE(nblocks == 1 || nblocks == 2, "res8_chain_bwd: 1 or 2 blocks per launch (got %d)", nblocks);
  ResBwdArgs a{};
  a.dy = (const bf16*)dy; a.dpool = dy ? nullptr : (const bf16*)dpool; a.ylast = (const bf16*)ylast; a.dy_out = (bf16*)dy_out;
  a.N = N; a.nblocks = nblocks;
  if (head) {
    a.logits = (const bf16*)head->logits; a.head_w = head->head_w; a.pooled = (const bf16*)head->pooled; a.loss = head->loss;
    a.w_grad = head->w_grad; a.b_grad = head->b_grad; a.n_real = head->n_real; a.mode = head->mode; a.loss_scale = head->loss_scale;
  }
  for (int i = 0; i < 2 * nblocks; i++) {
    GANK_REQUIRE(wd_rfrag[i], "res8_chain_bwd: null weight operand %d", i);
    a.wd[i] = (const bf16*)wd_rfrag[i];
  }
  for (int b = 0; b < nblocks; b++) {
    GANK_REQUIRE(h1[b] && xin[b], "res8_chain_bwd: null mask tensor of block %d", b);
    a.h1[b] = (const bf16*)h1[b]; a.xin[b] = (const bf16*)xin[b]; a.g1[b] = (bf16*)g1[b]; a.dx[b] = (bf16*)dx[b];
  }
  GANK_REQUIRE(a.dx[nblocks - 1], "res8_chain_bwd: the chain's input gradient has no destination");
  hipStream_t s = (hipStream_t)stream;
  gank_prof_begin(0, 2.0 * nblocks * 2.0 * N * 64.0 * RB_C * 9.0 * RB_C, s,
                  2.0 * N * 64.0 * RB_C * (1 + 4 * nblocks) + 2.0 * nblocks * RB_WBYTES);
  gank_prof_tag(0, "res8_chain_bwd_kernel");
  int rc = 0;
  RES8_DISPATCH(res8_launch_bwd, a, s);
  gank_prof_end(0, s);
  if (rc) return rc;
  GANK_LAUNCH_OK("res8_chain_bwd");
  return 0;
}
extern "C" int gank_res8_chain_bwd(const void* dy, const void* dpool, const void* ylast, void* dy_out, const void* const* wd_rfrag,
                                   const void* const* h1, const void* const* xin, void* const* g1, void* const* dx, int N, int C,
                                   int nblocks, void* stream) {
  return res8_chain_bwd_impl(dy, dpool, ylast, dy_out, wd_rfrag, h1, xin, g1, dx, N, C, nblocks, nullptr, stream);
}
extern "C" int gank_res8_chain_bwd_head(const gank_res8_head* head, const void* ylast, void* dy_out, const void* const* wd_rfrag,
                                        const void* const* h1, const void* const* xin, void* const* g1, void* const* dx, int N, int C,
                                        int nblocks, void* stream) {
  GANK_REQUIRE(head, "res8_chain_bwd_head: null head");
  return res8_chain_bwd_impl(nullptr, nullptr, ylast, dy_out, wd_rfrag, h1, xin, g1, dx, N, C, nblocks, head, stream);
}

// ==================================================================================================================
// ConvMeanPool 3x3 (gan_cifar_resnet.py:112-123) on resident images: fprop = 4x4 stride-2 conv, dgrad = its transposed
// conv as 4 output phases of 2x2 taps (same algebra as gank_convpool3x3_*; operands: prep kind 5).
//
// The implicit-GEMM forms gathered every tap from HBM/L2 again: 16 taps x 33.5 MB for D.Block.1.Conv2 at n = 128
// (51.7 us, 0.33 PFLOP/s; 89.8 MB of HBM traffic for 42 MB of operands).  Here a workgroup owns a patch of POOLED pixels
// (8 x 16 or 8 x 8), stages the input region it needs ONCE per 64-channel chunk -- as four parity planes
// plane[(r & 1, c & 1)][r >> 1][c >> 1], so a stride-2 tap becomes a unit-stride read of one plane -- and runs all 16
// taps of the chunk out of LDS with no barrier in between; weights stream from L2 in fragment order (one 1 KB request
// per A fragment, ring of 12 in flight).  Wave (ct, pg): 32 output channels x TPW 32-pixel tiles.
// Pitches: pixel 144 B (64 ch + 16 B = 9 sixteen-byte units, odd => consecutive pixels take distinct bank slots), plane
// row pitch = 0 mod 256 B for 16-wide tiles (2 rows x 16 columns) and 128 mod 256 B for 8-wide tiles (4 rows x 8
// columns): every 16-lane group of a ds_read_b128 then covers 16 distinct slots.
// ==================================================================================================================
namespace {
constexpr int CP_PP = 144;                           // pixel pitch (bytes) of a 64-channel chunk image

template <int PW> struct CpGeom {
  static constexpr int PHH = 8;                      // pooled patch rows
  static constexpr int PLW = PW + 1, PLH = PHH + 1;  // plane size (pixels)
  static constexpr int RP = PW == 16 ? 2560 : 1408;  // plane row pitch: 17*144 = 2448 -> 2560 (0 mod 256); 9*144 = 1296 -> 1408 (128 mod 256)
  static constexpr int PLANE = PLH * RP;
  static constexpr int IMG = 4 * PLANE;              // 92160 / 50688 bytes
  static constexpr int TROWS = 32 / PW;              // patch rows per 32-pixel tile
};

struct CpFwdArgs {
  const bf16* x;          // [N, 2Hp, 2Wp, Cin]
  const bf16* w;          // rfrag: [Cout/32][Cin/64][16 taps][4 kk][64][8]
  const float* bias;      // optional [Cout]
  const bf16* res;        // optional residual at pooled resolution [N,Hp,Wp,Cout]
  bf16* y;                // [N,Hp,Wp,Cout]
  int N, Hp, Wp, Cin, Cout, relu;
  int xcd;                // XCD-aware block order (resident_xcd_env): workgroups that share an input patch share an L2
  int dbg;                // TUNING builds (GANK_CPOOL_DBG, timing only): 1 = input loads out of range (zero fill, no traffic), 2 = no output
};

struct CpBwdArgs {
  const bf16* dy;         // [N,Hp,Wp,Cout]
  const bf16* w;          // rfrag: [4 phases][Cin/32][Cout/64][4 taps][4 kk][64][8]
  const bf16* mask;       // optional relu reference [N,2Hp,2Wp,Cin]
  bf16* dx;               // [N,2Hp,2Wp,Cin]
  int N, Hp, Wp, Cin, Cout;
  int xcd;
};
}  // namespace

// fprop: 8 waves = (Cout tile ct = wave & 3) x (pixel group pg = wave >> 2); a workgroup covers 128 output channels
template <int PW, int TPW, int PF, int NTAP = 16>      // NTAP != 16: timing experiments only (GANK_CPOOL_TAPS, tuning builds)
__global__ __launch_bounds__(512) void cpool_res_fprop_kernel(CpFwdArgs a) {
  using G = CpGeom<PW>;
  constexpr int NSTEP = NTAP * 4;
  static_assert(2 * TPW * G::TROWS == G::PHH, "8 waves = 4 channel tiles x 2 pixel groups must tile the patch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, pg = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int pw_n = a.Wp / PW, ph_n = a.Hp / G::PHH, cgroups = a.Cout >> 7;
  int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cg = bid % cgroups; bid /= cgroups;
  const int n = bid / (pw_n * ph_n), pr = bid - n * pw_n * ph_n;
  const int py0 = (pr / pw_n) * G::PHH, px0 = (pr % pw_n) * PW;          // pooled patch origin
  const int nchunks = a.Cin >> 6;
  const int H2 = 2 * a.Hp, W2 = 2 * a.Wp;

  // this lane's pixel inside a tile, and its B-fragment base inside plane (0,0): (row, col) + h*16 bytes
  const int trow = PW == 16 ? (r >> 4) : (r >> 3), tcol = PW == 16 ? (r & 15) : (r & 7);
  const int b_base = (pg * TPW * G::TROWS + trow) * G::RP + tcol * CP_PP + h * 16;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, a.Cout * 16 * a.Cin * 2, 0x00020000);
  const int wbase = (cg * 4 + ct) * nchunks * 64 * 1024;                  // 64 steps (16 taps x 4 kk) of 1 KB per chunk
  const int nsteps = nchunks * NSTEP;
  u32x4 ring[PF];
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wbase + (s < nsteps ? s : nsteps - 1) * 1024, 0);

  f32x16 acc[TPW];
#pragma unroll
  for (int t = 0; t < TPW; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

  // staging: the (2*PHH + 2) x (2*PW + 2) input pixels of the patch, 8 sixteen-byte pieces (64 channels) per pixel
  constexpr int HR = 2 * G::PHH + 2, HC = 2 * PW + 2, NPIECE = HR * HC * 8;
  constexpr int NLD = (NPIECE + 511) / 512;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * H2 * W2 * a.Cin * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;

  // the next chunk's pieces are requested before the current chunk's MFMA loop (in flight behind the weight stream, as in the
  // two-group kernel below): with the loads in front of each loop the workgroup spent 16.8 of its 24.6 us outside the loops
  // (profiles/r05_cpool_9tap_bound.txt: 64 -> 36 steps per chunk took 3.4 us off)
  int p_off[NLD], p_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * 512;
    const bool on = NPIECE % 512 == 0 || q < NPIECE;
    const int hp = q >> 3, c16 = q & 7;
    const int hr = hp / HC, hc = hp - hr * HC;                            // halo row / column: input pixel (2*py0 - 1 + hr, 2*px0 - 1 + hc)
    const int iy = 2 * py0 - 1 + hr, ix = 2 * px0 - 1 + hc;
    const bool ok = on && (unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2;
    p_off[j] = ok ? (((n * H2 + iy) * W2 + ix) * a.Cin + c16 * 8) * 2 : OOB;
#ifdef GANK_TUNING
    if (a.dbg & 1) p_off[j] = OOB;
#endif
    p_lds[j] = on ? ((hr & 1) * 2 + (hc & 1)) * G::PLANE + (hr >> 1) * G::RP + (hc >> 1) * CP_PP + c16 * 16 : -1;
  }
  u32x4 rP[NLD];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int j = 0; j < NLD; j++) rP[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, p_off[j] == OOB ? OOB : p_off[j] + c * 128, 0, 0);
  };
  load_chunk(0);

  int step = 0;
#pragma unroll 1
  for (int c = 0; c < nchunks; c++) {
    if (c > 0) __syncthreads();                                           // every wave is done reading the previous chunk's image
#pragma unroll
    for (int j = 0; j < NLD; j++)
      if (p_lds[j] >= 0) {
        u32x4 v = rP[j];
        if (a.relu) v = relu_bf16x8(v);
        *reinterpret_cast<u32x4*>(smem + p_lds[j]) = v;
      }
    __syncthreads();
    if (c + 1 < nchunks) load_chunk(c + 1);
    static_assert(NSTEP % PF == 0, "ring position is chunk-invariant");
    constexpr int PB = 2;                                                   // pixel fragments are read PB steps ahead of their MFMAs
    bf16x8 bq[PB + 1][TPW];
    auto read_b = [&](int s, bf16x8 (&dst)[TPW]) {
      const int tap = s >> 2, kk = s & 3, ta = tap >> 2, tb = tap & 3;
#pragma unroll
      for (int t = 0; t < TPW; t++)
        dst[t] = *reinterpret_cast<const bf16x8*>(smem + b_base + ((ta & 1) * 2 + (tb & 1)) * G::PLANE +
                                                   (t * G::TROWS + (ta >> 1)) * G::RP + (tb >> 1) * CP_PP + kk * 32);
    };
#pragma unroll
    for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
    for (int s = 0; s < NSTEP; s++, step++) {
      if (s + PB < NSTEP) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
      __builtin_amdgcn_sched_barrier(0);                                    // see res_conv3x3: keeps reads early and the ring deep
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = GANK_MFMA32(fa, bq[s % (PB + 1)][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int nx = step + PF;
        ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wbase + (nx < nsteps ? nx : nsteps - 1) * 1024, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: after acc_widen the lane holds channels 16q + 8h .. +7 of one pooled pixel: 16-byte pieces
#ifdef GANK_TUNING
  if (a.dbg & 2) return;
#endif
#pragma unroll
  for (int t = 0; t < TPW; t++) {
    const int py = py0 + (pg * TPW + t) * G::TROWS + trow, px = px0 + tcol;
    const long m = ((long)n * a.Hp + py) * a.Wp + px;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int co = cg * 128 + ct * 32 + 16 * q + 8 * h;
      float v[8];
      acc_widen(acc[t], q, 1.0f, v);
      if (a.bias) {
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
        for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
      }
      if (a.res) {
        const bf16x8 rs = *reinterpret_cast<const bf16x8*>(a.res + m * a.Cout + co);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += bf2f(rs[e]);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
      *reinterpret_cast<bf16x8*>(a.y + m * a.Cout + co) = o;
    }
  }
}

// The same for 8-wide pooled patches (one 16x16 -> 8x8 image per workgroup: D.Block.2.Conv2 at n = 128 is 128 patches) with
// the chip filled: a workgroup covers 64 output channels instead of 128 (twice the workgroups) and its 8 waves are
// (ct = 2 channel tiles) x (pg = 2 pixel tiles) x (kg = 2 halves of the reduction): K group kg runs the input-channel chunks
// 2r + kg out of its OWN image (two chunk images resident at once), the two partial sums meet in LDS at the end.  What bounded
// the one-group form was each CU streaming the whole 16 * Cin * 128 * 2-byte operand through its L1 (1 MB at 40-70 GB/s);
// here a CU streams half of it and a wave a quarter; the next round's input pieces are requested before the current round's
// MFMAs (in flight behind the weight stream instead of in front of it).
template <int PF, int NTAP = 16>                       // NTAP != 16: timing experiments only
__global__ __launch_bounds__(512) void cpool_res_fprop_k2_kernel(CpFwdArgs a) {
  using G = CpGeom<8>;
  constexpr int PW = 8, NSTEP = NTAP * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 K groups][IMG]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 1, pg = (wave >> 1) & 1, kg = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int ph_n = a.Hp / G::PHH, cgroups = a.Cout >> 6;
  int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cg = bid % cgroups; bid /= cgroups;
  const int n = bid / ph_n, pr = bid - n * ph_n;
  const int py0 = pr * G::PHH, px0 = 0;
  const int nchunks = a.Cin >> 6, nrounds = nchunks >> 1;
  const int H2 = 2 * a.Hp, W2 = 2 * a.Wp;
  const int trow = r >> 3, tcol = r & 7;
  const int b_base = kg * G::IMG + (pg * G::TROWS + trow) * G::RP + tcol * CP_PP + h * 16;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, a.Cout * 16 * a.Cin * 2, 0x00020000);
  const int tile = cg * 2 + ct;
  // this wave's weight stream: chunks kg, kg + 2, ... of its channel tile, 64 steps (16 taps x 4 kk) of 1 KB each
  auto wofs = [&](int st) {                                             // st = 64 * round + s
    const int rd = st / NSTEP, s = st - rd * NSTEP;
    return ((tile * nchunks + 2 * rd + kg) * 64 + s) * 1024;
  };
  const int nsteps = nrounds * NSTEP;
  u32x4 ring[PF];
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(s < nsteps ? s : nsteps - 1), 0);

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;

  // staging of one round: the 18 x 18 input pixels of the patch for TWO 64-channel chunks, 8 sixteen-byte pieces per pixel and chunk
  constexpr int HR = 2 * G::PHH + 2, HC = 2 * PW + 2, NPIECE = 2 * HR * HC * 8;
  constexpr int NLD = (NPIECE + 511) / 512;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * H2 * W2 * a.Cin * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
  int p_off[NLD], p_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * 512;
    const bool on = q < NPIECE;
    const int hp = q >> 4, c16 = q & 15;                                // 16 pieces per pixel: chunk (c16 >> 3), piece (c16 & 7)
    const int hr = hp / HC, hc = hp - hr * HC;
    const int iy = 2 * py0 - 1 + hr, ix = 2 * px0 - 1 + hc;
    const bool ok = on && (unsigned)iy < (unsigned)H2 && (unsigned)ix < (unsigned)W2;
    p_off[j] = ok ? (((n * H2 + iy) * W2 + ix) * a.Cin + c16 * 8) * 2 : OOB;
    p_lds[j] = on ? (c16 >> 3) * G::IMG + ((hr & 1) * 2 + (hc & 1)) * G::PLANE + (hr >> 1) * G::RP + (hc >> 1) * CP_PP + (c16 & 7) * 16 : -1;
  }
  u32x4 rP[NLD];
  auto load_round = [&](int rd) {
#pragma unroll
    for (int j = 0; j < NLD; j++) rP[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, p_off[j] == OOB ? OOB : p_off[j] + rd * 256, 0, 0);
  };
  load_round(0);

  int step = 0;
#pragma unroll 1
  for (int rd = 0; rd < nrounds; rd++) {
    if (rd > 0) __syncthreads();                                        // every wave is done reading the previous round's images
#pragma unroll
    for (int j = 0; j < NLD; j++)
      if (p_lds[j] >= 0) {
        u32x4 v = rP[j];
        if (a.relu) v = relu_bf16x8(v);
        *reinterpret_cast<u32x4*>(smem + p_lds[j]) = v;
      }
    __syncthreads();
    if (rd + 1 < nrounds) load_round(rd + 1);                           // in flight behind this round's weight stream
    static_assert(NSTEP % PF == 0, "ring position is round-invariant");
    constexpr int PB = 2;
    bf16x8 bq[PB + 1];
    auto read_b = [&](int s, bf16x8& dst) {
      const int tap = s >> 2, kk = s & 3, ta = tap >> 2, tb = tap & 3;
      dst = *reinterpret_cast<const bf16x8*>(smem + b_base + ((ta & 1) * 2 + (tb & 1)) * G::PLANE + (ta >> 1) * G::RP + (tb >> 1) * CP_PP + kk * 32);
    };
#pragma unroll
    for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
    for (int s = 0; s < NSTEP; s++, step++) {
      if (s + PB < NSTEP) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
      __builtin_amdgcn_sched_barrier(0);                                // see res_conv3x3: keeps reads early and the ring deep
      acc = GANK_MFMA32(fa, bq[s % (PB + 1)], acc);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int nx = step + PF;
        ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(nx < nsteps ? nx : nsteps - 1), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // the two K groups meet: group 1 parks its partial tile in LDS (the images are dead), group 0 adds it and finishes
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem) + ((wave & 3) * 64 + lane) * 16;
  if (kg == 1) {
#pragma unroll
    for (int e = 0; e < 16; e += 4) *reinterpret_cast<f32x4*>(red + e) = f32x4{acc[e], acc[e + 1], acc[e + 2], acc[e + 3]};
  }
  __syncthreads();
  if (kg == 1) return;
#pragma unroll
  for (int e = 0; e < 16; e += 4) {
    const f32x4 o = *reinterpret_cast<const f32x4*>(red + e);
    acc[e] += o[0]; acc[e + 1] += o[1]; acc[e + 2] += o[2]; acc[e + 3] += o[3];
  }
  const int py = py0 + pg * G::TROWS + trow, px = px0 + tcol;
  const long m = ((long)n * a.Hp + py) * a.Wp + px;
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int co = cg * 64 + ct * 32 + 16 * q + 8 * h;
    float v[8];
    acc_widen(acc, q, 1.0f, v);
    if (a.bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
    }
    if (a.res) {
      const bf16x8 rs = *reinterpret_cast<const bf16x8*>(a.res + m * a.Cout + co);
#pragma unroll
      for (int e = 0; e < 8; e++) v[e] += bf2f(rs[e]);
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
    *reinterpret_cast<bf16x8*>(a.y + m * a.Cout + co) = o;
  }
}

// dgrad: dx[n, 2y+pa, 2x+pb, ci] = sum_{i,j in {0,1}} sum_co dy[n, y+i-(1-pa), x+j-(1-pb), co] * Wph[pa,pb][ci][(i,j),co].
// The dy patch (+1 halo) with Cout = 128 channels stays in LDS for all four phases (pixel pitch 272 B); a workgroup covers
// 128 of the Cin "output" channels; wave (ct, pg) as above.
namespace {
template <int PW> struct CdGeom {
  static constexpr int PHH = 8;
  static constexpr int PP = 272;                                   // 128 channels + 16 B = 17 units
  static constexpr int RP = PW == 16 ? 5120 : 2944;                // 18*272 = 4896 -> 5120 (0 mod 256); 10*272 = 2720 -> 2944 (128 mod 256)
  static constexpr int IMG = (PHH + 2) * RP;                       // 51200 / 29440
  static constexpr int TROWS = 32 / PW;
};
}  // namespace

template <int PW, int TPW, int PF>
__global__ __launch_bounds__(512) void cpool_res_dgrad_kernel(CpBwdArgs a) {
  using G = CdGeom<PW>;
  static_assert(2 * TPW * G::TROWS == G::PHH, "8 waves = 4 channel tiles x 2 pixel groups must tile the patch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, pg = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int pw_n = a.Wp / PW, ph_n = a.Hp / G::PHH, cgroups = a.Cin >> 7;
  int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cg = bid % cgroups; bid /= cgroups;
  const int n = bid / (pw_n * ph_n), pr = bid - n * pw_n * ph_n;
  const int py0 = (pr / pw_n) * G::PHH, px0 = (pr % pw_n) * PW;
  const int trow = PW == 16 ? (r >> 4) : (r >> 3), tcol = PW == 16 ? (r & 15) : (r & 7);
  const int b_base = (pg * TPW * G::TROWS + trow) * G::RP + tcol * G::PP + h * 16;      // halo pixel (row, col): dy pixel (py0-1+row, px0-1+col)

  // operand: [phase][Cin/32][4 taps][Cout/16 kk][64][8]; Cout == 128 -> 32 steps of 1 KB per (phase, tile)
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, 4 * a.Cin * 4 * a.Cout * 2, 0x00020000);
  const int tiles = a.Cin >> 5, tile = cg * 4 + ct;
  auto wofs = [&](int phase, int s) { return ((phase * tiles + tile) * 32 + s) * 1024; };
  u32x4 ring[PF];
  static_assert(32 % PF == 0, "ring position is phase-invariant");
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(0, s), 0);

  // stage the dy halo: (PHH + 2) x (PW + 2) pixels x 16 pieces
  constexpr int HR = G::PHH + 2, HC = PW + 2, NPIECE = HR * HC * 16;
  constexpr int NLD = (NPIECE + 511) / 512;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.N * a.Hp * a.Wp * a.Cout * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * 512;
    if (NPIECE % 512 == 0 || q < NPIECE) {
      const int hp = q >> 4, c16 = q & 15;
      const int hr = hp / HC, hc = hp - hr * HC;
      const int iy = py0 - 1 + hr, ix = px0 - 1 + hc;
      const bool ok = (unsigned)iy < (unsigned)a.Hp && (unsigned)ix < (unsigned)a.Wp;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ry, ok ? (((n * a.Hp + iy) * a.Wp + ix) * a.Cout + c16 * 8) * 2 : OOB, 0, 0);
      *reinterpret_cast<u32x4*>(smem + hr * G::RP + hc * G::PP + c16 * 16) = v;
    }
  }
  __syncthreads();

  const int H2 = 2 * a.Hp, W2 = 2 * a.Wp;
#pragma unroll 1
  for (int phase = 0; phase < 4; phase++) {
    const int pa = phase >> 1, pb = phase & 1;
    const int poff = pa * G::RP + pb * G::PP;                  // tap (i, j) reads halo pixel (y + i + pa, x + j + pb)
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    // relu masks of this phase's outputs: requested now, consumed after the K loop
    bf16x8 mk[TPW][2];
    if (a.mask) {
#pragma unroll
      for (int t = 0; t < TPW; t++) {
        const int y = py0 + (pg * TPW + t) * G::TROWS + trow, x = px0 + tcol;
        const long m = ((long)n * H2 + 2 * y + pa) * W2 + 2 * x + pb;
#pragma unroll
        for (int q = 0; q < 2; q++) mk[t][q] = *reinterpret_cast<const bf16x8*>(a.mask + m * a.Cin + cg * 128 + ct * 32 + 16 * q + 8 * h);
      }
    }
    constexpr int PB = 2;
    bf16x8 bq[PB + 1][TPW];
    auto read_b = [&](int s, bf16x8 (&dst)[TPW]) {
      const int tap = s >> 3, kk = s & 7, ti = tap >> 1, tj = tap & 1;
#pragma unroll
      for (int t = 0; t < TPW; t++)
        dst[t] = *reinterpret_cast<const bf16x8*>(smem + b_base + poff + (t * G::TROWS + ti) * G::RP + tj * G::PP + kk * 32);
    };
#pragma unroll
    for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
    for (int s = 0; s < 32; s++) {
      if (s + PB < 32) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < TPW; t++) acc[t] = GANK_MFMA32(fa, bq[s % (PB + 1)][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int nx = s + PF;
        const int np = nx < 32 ? phase : (phase < 3 ? phase + 1 : 3), ns = nx < 32 ? nx : (phase < 3 ? nx - 32 : 31);
        ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(np, ns), 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < TPW; t++) {
      const int y = py0 + (pg * TPW + t) * G::TROWS + trow, x = px0 + tcol;
      const long m = ((long)n * H2 + 2 * y + pa) * W2 + 2 * x + pb;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int ci = cg * 128 + ct * 32 + 16 * q + 8 * h;
        float v[8];
        acc_widen(acc[t], q, 1.0f, v);
        if (a.mask) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = bf2f(mk[t][q][e]) > 0.f ? v[e] : 0.f;
        }
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x8*>(a.dx + m * a.Cin + ci) = o;
      }
    }
  }
}

// The same with the 8 waves as (4 channel tiles) x (2 halves of each phase's reduction) instead of x (2 pixel groups): a wave
// owns ALL pixel tiles of the patch, so every weight fragment is requested by one wave of the workgroup (and feeds TW MFMAs)
// instead of by two -- the CU's L1 path, which bounds these kernels, carries half the bytes.  After a phase's K loop the two
// halves swap partial tiles through LDS: group kg finishes (masks, rounds, stores) the tiles t with t / (TW/2) == kg.
template <int PW, int PF>
__global__ __launch_bounds__(512) void cpool_res_dgrad_ks_kernel(CpBwdArgs a) {
  using G = CdGeom<PW>;
  constexpr int TW = G::PHH / G::TROWS, TH = TW / 2;       // pixel tiles per wave / tiles a wave finishes
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, kg = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int pw_n = a.Wp / PW, ph_n = a.Hp / G::PHH, cgroups = a.Cin >> 7;
  int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cg = bid % cgroups; bid /= cgroups;
  const int n = bid / (pw_n * ph_n), pr = bid - n * pw_n * ph_n;
  const int py0 = (pr / pw_n) * G::PHH, px0 = (pr % pw_n) * PW;
  const int trow = PW == 16 ? (r >> 4) : (r >> 3), tcol = PW == 16 ? (r & 15) : (r & 7);
  const int b_base = trow * G::RP + tcol * G::PP + h * 16;
  f32x4* part = reinterpret_cast<f32x4*>(smem + G::IMG) + (kg * 4 + ct) * (TH * 4 * 64);      // [writer kg][ct][TH tiles x 4 quads][64 lanes]
  f32x4* part_in = reinterpret_cast<f32x4*>(smem + G::IMG) + ((kg ^ 1) * 4 + ct) * (TH * 4 * 64);

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, 4 * a.Cin * 4 * a.Cout * 2, 0x00020000);
  const int tiles = a.Cin >> 5, tile = cg * 4 + ct;
  auto wofs = [&](int phase, int s) { return ((phase * tiles + tile) * 32 + 16 * kg + s) * 1024; };      // this wave's 16 steps of a phase
  u32x4 ring[PF];
  static_assert(16 % PF == 0, "ring position is phase-invariant");
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(0, s), 0);

  constexpr int HR = G::PHH + 2, HC = PW + 2, NPIECE = HR * HC * 16;
  constexpr int NLD = (NPIECE + 511) / 512;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.N * a.Hp * a.Wp * a.Cout * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * 512;
    if (NPIECE % 512 == 0 || q < NPIECE) {
      const int hp = q >> 4, c16 = q & 15;
      const int hr = hp / HC, hc = hp - hr * HC;
      const int iy = py0 - 1 + hr, ix = px0 - 1 + hc;
      const bool ok = (unsigned)iy < (unsigned)a.Hp && (unsigned)ix < (unsigned)a.Wp;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ry, ok ? (((n * a.Hp + iy) * a.Wp + ix) * a.Cout + c16 * 8) * 2 : OOB, 0, 0);
      *reinterpret_cast<u32x4*>(smem + hr * G::RP + hc * G::PP + c16 * 16) = v;
    }
  }
  __syncthreads();

  const int H2 = 2 * a.Hp, W2 = 2 * a.Wp;
#pragma unroll 1
  for (int phase = 0; phase < 4; phase++) {
    const int pa = phase >> 1, pb = phase & 1;
    const int poff = pa * G::RP + pb * G::PP;
    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    // relu masks of the tiles this wave finishes: requested now, consumed after the K loop
    bf16x8 mk[TH][2];
    if (a.mask) {
#pragma unroll
      for (int t = 0; t < TH; t++) {
        const int y = py0 + (kg * TH + t) * G::TROWS + trow, x = px0 + tcol;
        const long m = ((long)n * H2 + 2 * y + pa) * W2 + 2 * x + pb;
#pragma unroll
        for (int q = 0; q < 2; q++) mk[t][q] = *reinterpret_cast<const bf16x8*>(a.mask + m * a.Cin + cg * 128 + ct * 32 + 16 * q + 8 * h);
      }
    }
    constexpr int PB = 2;
    bf16x8 bq[PB + 1][TW];
    auto body = [&](auto half) {
      constexpr int S0 = decltype(half)::value * 16;
      auto read_b = [&](int s, bf16x8 (&dst)[TW]) {
        const int tap = (S0 + s) >> 3, kk = (S0 + s) & 7, ti = tap >> 1, tj = tap & 1;
#pragma unroll
        for (int t = 0; t < TW; t++)
          dst[t] = *reinterpret_cast<const bf16x8*>(smem + b_base + poff + (t * G::TROWS + ti) * G::RP + tj * G::PP + kk * 32);
      };
#pragma unroll
      for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
      for (int s = 0; s < 16; s++) {
        if (s + PB < 16) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
        const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TW; t++) acc[t] = GANK_MFMA32(fa, bq[s % (PB + 1)][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        {
          const int nx = s + PF;
          const int np = nx < 16 ? phase : (phase < 3 ? phase + 1 : 3), ns = nx < 16 ? nx : (phase < 3 ? nx - 16 : 15);
          ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(np, ns), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (kg == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    // swap: park the partial sums of the tiles the OTHER group finishes (compile-time tile indices in both branches: a
    // run-time index into the accumulator array would move it to scratch memory)
    if (phase > 0) __syncthreads();                    // the previous phase's partials have been read
    auto finish = [&](auto group) {
      constexpr int KG = decltype(group)::value;
#pragma unroll
      for (int t = 0; t < TH; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x16& src = acc[(KG ^ 1) * TH + t];
          part[(t * 4 + g) * 64 + lane] = f32x4{src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TH; t++) {
        f32x16& fin = acc[KG * TH + t];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 o = part_in[(t * 4 + g) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; e++) fin[4 * g + e] += o[e];
        }
        const int y = py0 + (KG * TH + t) * G::TROWS + trow, x = px0 + tcol;
        const long m = ((long)n * H2 + 2 * y + pa) * W2 + 2 * x + pb;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int ci = cg * 128 + ct * 32 + 16 * q + 8 * h;
          float v[8];
          acc_widen(fin, q, 1.0f, v);
          if (a.mask) {
#pragma unroll
            for (int e = 0; e < 8; e++) v[e] = bf2f(mk[t][q][e]) > 0.f ? v[e] : 0.f;
          }
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
          *reinterpret_cast<bf16x8*>(a.dx + m * a.Cin + ci) = o;
        }
      }
    };
    if (kg == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The same input gradient with the FILTER GRADIENT OF THE 3-CHANNEL-INPUT CONV IN FRONT OF IT in its epilogue (round 5).
// In the critic's first block (gan_cifar_resnet.py:212-234) the tensor this kernel produces -- d loss / d relu'(Conv1 output),
// 128 x 32 x 32 x 128 = 33.5 MB -- has ONE consumer in a critic update: D.Block.1.Conv1's filter and bias gradient (the image
// needs no gradient).  Written, it cost this kernel its store phase and the streaming filter-gradient kernel a 33.5-MB read at
// 1.3 TB/s (25 us).  Here the finished (masked, bf16-rounded -- the values the stored tensor held) tile of a wave goes through
// the wave's OWN 2-KB LDS region ([32 pixels][32 channels], ordered by the wave's LDS counter alone) and comes back through
// ds_read_b64_tr_b16 as the B operand of  dW1[k][c] += sum_p Xcol[k][p] dh[p][c]:  k = (kh, kw, cin) of the 3x3x3 filter (27 rows),
// row 27 = ones (the bias gradient), 4 MFMAs per wave and phase.  Xcol ([128 pixels of the phase][32 k], 8 KB, two buffers) is
// built by all threads from the workgroup's image rows (18 x 32 x 3 bf16, zero-padded in LDS) one phase ahead.  The 1x1
// shortcut conv on the pooled image (D.Block.1.Shortcut) reads the SAME dy this kernel stages: its filter and bias gradient are
// rows 28..31 of the same accumulator tile, 4 more MFMAs per wave on the resident dy halo.  The tensor is never stored.
// One [32][Cin] tile per workgroup leaves as fp32 atomics (the two K groups of a channel tile meet in LDS first).
// Geometry: Wp == 16 (a patch spans the image width, so the image rows need no column halo beyond the zero padding).
// ------------------------------------------------------------------------------------------------------------------
namespace {
struct CpBwdWgArgs {
  const bf16* dy;         // [N,Hp,Wp,Cout]
  const bf16* w;          // rfrag (as CpBwdArgs)
  const bf16* mask;       // relu reference [N,2Hp,2Wp,Cin] (the front conv's output)
  const bf16* ximg;       // [N,2Hp,2Wp,3]: input of the front 3x3 conv
  const bf16* xpool;      // [N,Hp,Wp,3] or null: input of the 1x1 shortcut conv whose output gradient is dy
  float* dw1;             // [3,3,3,Cin] accumulated
  float* db1;             // [Cin] or null
  float* dws;             // [1,1,3,Cout] or null
  float* dbs;             // [Cout] or null
  float* slabs;           // null: fp32 atomics into the four targets; else [grid][32][128] plain stores (Cin == 128), summed by the caller
  int N, Hp, Wp, Cin, Cout;
  int xcd;
  int dbg;                // TUNING builds (GANK_IMGWG_DBG): 1 = no final atomics, 2 = no filter-gradient section, 4 = no Xcol build
};
constexpr int WG_XI_PITCH = 112;                       // image row in LDS (bf16): 8 zeros, 96 values, 8 zeros
constexpr int WG_XI = CdGeom<16>::IMG + 2 * 4 * 2 * 4 * 64 * 16;     // byte offsets behind the halo image and the partial-tile area
constexpr int WG_XP = WG_XI + 4096;
constexpr int WG_XCOL = WG_XP + 768;
constexpr int WG_DH = WG_XCOL + 2 * 8192;
constexpr int WG_LDS = WG_DH + 8 * 2048;
static_assert(WG_XCOL % 16 == 0 && WG_DH % 16 == 0 && WG_LDS <= 160 * 1024, "LDS plan of the fused filter gradient");
__device__ __forceinline__ s16x4 res_tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}
}  // namespace

template <int PF>
__global__ __launch_bounds__(512) void cpool_res_dgrad_imgwg_kernel(CpBwdWgArgs a) {
  constexpr int PW = 16;
  using G = CdGeom<PW>;
  constexpr int TW = G::PHH / G::TROWS, TH = TW / 2;       // 4 pixel tiles per wave, 2 finished per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, kg = wave >> 2;
  const int r = lane & 31, h = lane >> 5;
  const int ph_n = a.Hp / G::PHH, cgroups = a.Cin >> 7;
  int bid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int cg = bid % cgroups; bid /= cgroups;
  const int n = bid / ph_n, py0 = (bid - n * ph_n) * G::PHH, px0 = 0;
  const int trow = r >> 4, tcol = r & 15;
  const int b_base = trow * G::RP + tcol * G::PP + h * 16;
  f32x4* part = reinterpret_cast<f32x4*>(smem + G::IMG) + (kg * 4 + ct) * (TH * 4 * 64);
  f32x4* part_in = reinterpret_cast<f32x4*>(smem + G::IMG) + ((kg ^ 1) * 4 + ct) * (TH * 4 * 64);
  const bool shortcut = a.xpool != nullptr && cg == 0;

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, 4 * a.Cin * 4 * a.Cout * 2, 0x00020000);
  const int tiles = a.Cin >> 5, tile = cg * 4 + ct;
  auto wofs = [&](int phase, int s) { return ((phase * tiles + tile) * 32 + 16 * kg + s) * 1024; };
  u32x4 ring[PF];
  static_assert(16 % PF == 0, "ring position is phase-invariant");
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(0, s), 0);

  // ---- stage: dy halo, the image rows 2 py0 - 1 .. 2 py0 + 16 (zero outside the image / in the pads), the pooled-image rows
  constexpr int HR = G::PHH + 2, HC = PW + 2, NPIECE = HR * HC * 16;
  constexpr int NLD = (NPIECE + 511) / 512;
  const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.N * a.Hp * a.Wp * a.Cout * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * 512;
    if (NPIECE % 512 == 0 || q < NPIECE) {
      const int hp = q >> 4, c16 = q & 15;
      const int hr = hp / HC, hc = hp - hr * HC;
      const int iy = py0 - 1 + hr, ix = px0 - 1 + hc;
      const bool ok = (unsigned)iy < (unsigned)a.Hp && (unsigned)ix < (unsigned)a.Wp;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(ry, ok ? (((n * a.Hp + iy) * a.Wp + ix) * a.Cout + c16 * 8) * 2 : OOB, 0, 0);
      *reinterpret_cast<u32x4*>(smem + hr * G::RP + hc * G::PP + c16 * 16) = v;
    }
  }
  const int H2 = 2 * a.Hp, W2 = 2 * a.Wp;      // W2 == 32
  {
    const u32x4 z4 = {0u, 0u, 0u, 0u};
    if (tid < 216) {                           // 18 rows x 12 sixteen-byte pieces (a 32-pixel row of 3 channels = 192 B)
      const int lr = tid / 12, j = tid - lr * 12, gr = 2 * py0 - 1 + lr;
      const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.ximg), 0, a.N * H2 * W2 * 3 * 2, 0x00020000);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rx, (unsigned)gr < (unsigned)H2 ? ((n * H2 + gr) * W2 * 3) * 2 + j * 16 : OOB, 0, 0);
      *reinterpret_cast<u32x4*>(smem + WG_XI + lr * (WG_XI_PITCH * 2) + 16 + j * 16) = v;
    } else if (tid < 252) {                    // the two 16-byte pads of each row
      const int q = tid - 216, lr = q >> 1;
      *reinterpret_cast<u32x4*>(smem + WG_XI + lr * (WG_XI_PITCH * 2) + (q & 1) * 208) = z4;
    } else if (tid >= 256 && tid < 304 && shortcut) {      // 8 pooled rows x 16 pixels x 3 channels = 768 contiguous bytes
      const __amdgpu_buffer_rsrc_t rp = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.xpool), 0, a.N * a.Hp * a.Wp * 3 * 2, 0x00020000);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rp, ((n * a.Hp + py0) * a.Wp * 3) * 2 + (tid - 256) * 16, 0, 0);
      *reinterpret_cast<u32x4*>(smem + WG_XP + (tid - 256) * 16) = v;
    }
  }
  __syncthreads();

  // ---- Xcol: thread (p = tid / 4, kq = tid % 4) writes rows k = 8 kq .. + 7 of phase pixel p.  k = 9 kh + (3 kw + cin) reads the
  // image row (2 y + pa + kh) at element 5 + 3 (2 x + pb) + (3 kw + cin): nine consecutive values per filter row.
  const unsigned short* xi = reinterpret_cast<const unsigned short*>(smem + WG_XI);
  int xoff[8];
  {
    const int p = tid >> 2, kq = tid & 3, t = p >> 5, rr = p & 31, y = 2 * t + (rr >> 4), xq = rr & 15;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = 8 * kq + j, kh = (k * 57) >> 9, jj = k - 9 * kh;       // k / 9 for k < 32
      xoff[j] = k < 27 ? (2 * y + kh) * WG_XI_PITCH + 5 + 6 * xq + jj : (k == 27 ? -1 : -2);
    }
  }
  auto build_xcol = [&](int phase) {
    const int sh = (phase >> 1) * WG_XI_PITCH + (phase & 1) * 3;
    unsigned short v[8];
#pragma unroll
    for (int j = 0; j < 8; j++) v[j] = xoff[j] >= 0 ? xi[xoff[j] + sh] : (xoff[j] == -1 ? (unsigned short)0x3F80 : (unsigned short)0);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; j++) o[j] = (unsigned)v[2 * j] | ((unsigned)v[2 * j + 1] << 16);
    *reinterpret_cast<u32x4*>(smem + WG_XCOL + (phase & 1) * 8192 + tid * 16) = o;       // row p = tid / 4: 64 B, piece kq
  };
  build_xcol(0);
  if (shortcut) {      // Xs [128 pooled pixels][32]: rows 28..30 = the pooled image's channels, row 31 = ones (in the DH area, before any wave uses it)
    const unsigned short* xp = reinterpret_cast<const unsigned short*>(smem + WG_XP);
    const int p = tid >> 2, kq = tid & 3;
    u32x4 o = {0u, 0u, 0u, 0u};
    if (kq == 3) {
      o[2] = (unsigned)xp[3 * p] | ((unsigned)xp[3 * p + 1] << 16);
      o[3] = (unsigned)xp[3 * p + 2] | (0x3F80u << 16);
    }
    *reinterpret_cast<u32x4*>(smem + WG_DH + tid * 16) = o;
  }
  __syncthreads();

  f32x16 accw;
#pragma unroll
  for (int e = 0; e < 16; e++) accw[e] = 0.f;
  const int g4 = lane >> 4, li = lane & 15;
  const int tr_row = 8 * (g4 >> 1) + (li >> 2), tr_col = 16 * (g4 & 1) + 4 * (li & 3);      // transposed-read lane geometry (conv_wgrad.hip)
  const int tr64 = (tr_row * 32 + tr_col) * 2;                                              // bytes, 64-byte rows
  if (shortcut) {      // this wave: channel tile ct, pooled rows 4 kg .. + 3 (a K-step = one pooled row of 16 pixels)
#pragma unroll
    for (int jj = 0; jj < 4; jj++) {
      const int j = 4 * kg + jj;
      const char* pa_ = smem + WG_DH + j * 16 * 64 + tr64;
      const char* pb_ = smem + (1 + j) * G::RP + (1 + tr_row) * G::PP + (ct * 32 + tr_col) * 2;
      const s16x4 al = res_tr_read(pa_), ah = res_tr_read(pa_ + 4 * 64);
      const s16x4 bl = res_tr_read(pb_), bh = res_tr_read(pb_ + 4 * G::PP);
      const s16x8 ta = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
      const s16x8 tb = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
      accw = GANK_MFMA32(__builtin_bit_cast(bf16x8, ta), __builtin_bit_cast(bf16x8, tb), accw);
    }
  }

  char* dh = smem + WG_DH + wave * 2048;
#pragma unroll 1
  for (int phase = 0; phase < 4; phase++) {
    const int pa = phase >> 1, pb = phase & 1;
    const int poff = pa * G::RP + pb * G::PP;
    f32x16 acc[TW];
#pragma unroll
    for (int t = 0; t < TW; t++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
    bf16x8 mk[TH][2];
#pragma unroll
    for (int t = 0; t < TH; t++) {
      const int y = py0 + (kg * TH + t) * G::TROWS + trow, x = px0 + tcol;
      const long m = ((long)n * H2 + 2 * y + pa) * W2 + 2 * x + pb;
#pragma unroll
      for (int q = 0; q < 2; q++) mk[t][q] = *reinterpret_cast<const bf16x8*>(a.mask + m * a.Cin + cg * 128 + ct * 32 + 16 * q + 8 * h);
    }
    constexpr int PB = 2;
    bf16x8 bq[PB + 1][TW];
    auto body = [&](auto half) {
      constexpr int S0 = decltype(half)::value * 16;
      auto read_b = [&](int s, bf16x8 (&dst)[TW]) {
        const int tap = (S0 + s) >> 3, kk = (S0 + s) & 7, ti = tap >> 1, tj = tap & 1;
#pragma unroll
        for (int t = 0; t < TW; t++)
          dst[t] = *reinterpret_cast<const bf16x8*>(smem + b_base + poff + (t * G::TROWS + ti) * G::RP + tj * G::PP + kk * 32);
      };
#pragma unroll
      for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
      for (int s = 0; s < 16; s++) {
        if (s + PB < 16) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
        const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < TW; t++) acc[t] = GANK_MFMA32(fa, bq[s % (PB + 1)][t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        {
          const int nx = s + PF;
          const int np = nx < 16 ? phase : (phase < 3 ? phase + 1 : 3), ns = nx < 16 ? nx : (phase < 3 ? nx - 16 : 15);
          ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wofs(np, ns), 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (kg == 0) body(std::integral_constant<int, 0>{});
    else body(std::integral_constant<int, 1>{});
    if (phase > 0) __syncthreads();                    // the previous phase's partials have been read; every wave has left phase - 1
#ifdef GANK_TUNING
    if (!(a.dbg & 4))
#endif
    if (phase < 3) build_xcol(phase + 1);              // (its buffer was last read in phase - 1; visible after this phase's barrier below)
    auto finish = [&](auto group) {
      constexpr int KG = decltype(group)::value;
#pragma unroll
      for (int t = 0; t < TH; t++)
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x16& src = acc[(KG ^ 1) * TH + t];
          part[(t * 4 + g) * 64 + lane] = f32x4{src[4 * g], src[4 * g + 1], src[4 * g + 2], src[4 * g + 3]};
        }
      __syncthreads();
#pragma unroll
      for (int t = 0; t < TH; t++) {
        f32x16& fin = acc[KG * TH + t];
#pragma unroll
        for (int g = 0; g < 4; g++) {
          const f32x4 o = part_in[(t * 4 + g) * 64 + lane];
#pragma unroll
          for (int e = 0; e < 4; e++) fin[4 * g + e] += o[e];
        }
        // the finished tile (32 pixels x this wave's 32 channels), masked and rounded as the stored tensor was, through the wave's own LDS rows
#pragma unroll
        for (int q = 0; q < 2; q++) {
          float v[8];
          acc_widen(fin, q, 1.0f, v);
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] = bf2f(mk[t][q][e]) > 0.f ? v[e] : 0.f;
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
          *reinterpret_cast<bf16x8*>(dh + r * 64 + 32 * q + 16 * h) = o;
        }
#ifdef GANK_TUNING
        if (a.dbg & 2) continue;
#endif
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own writes: program order + this wait, no barrier
        const char* xc = smem + WG_XCOL + (phase & 1) * 8192 + (KG * TH + t) * 32 * 64 + tr64;
#pragma unroll
        for (int kk = 0; kk < 2; kk++) {
          const s16x4 al = res_tr_read(xc + kk * 16 * 64), ah = res_tr_read(xc + kk * 16 * 64 + 4 * 64);
          const s16x4 bl = res_tr_read(dh + kk * 16 * 64 + tr64), bh = res_tr_read(dh + kk * 16 * 64 + 4 * 64 + tr64);
          const s16x8 ta = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
          const s16x8 tb = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
          accw = GANK_MFMA32(__builtin_bit_cast(bf16x8, ta), __builtin_bit_cast(bf16x8, tb), accw);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads above have returned before the next tile overwrites the rows
      }
    };
    if (kg == 0) finish(std::integral_constant<int, 0>{});
    else finish(std::integral_constant<int, 1>{});
  }

  // ---- the two K groups of a channel tile meet in LDS (each keeps one half of the rows), then one [32][32] tile per wave pair
  // leaves as fp32 atomics: rows 0..26 -> dw1, 27 -> db1, 28..30 -> dws, 31 -> dbs
#ifdef GANK_TUNING
  if (a.dbg & 1) { if (accw[0] == 123.456f) a.dw1[0] = 1.f; return; }
#endif
  __syncthreads();
  float* ex = reinterpret_cast<float*>(smem + G::IMG) + ((kg * 4 + ct) * 8) * 64;
  const float* ex_in = reinterpret_cast<const float*>(smem + G::IMG) + (((kg ^ 1) * 4 + ct) * 8) * 64;
#pragma unroll
  for (int e = 0; e < 8; e++) ex[e * 64 + lane] = kg == 0 ? accw[8 + e] : accw[e];      // the half the OTHER group finishes
  __syncthreads();
  const int col = ct * 32 + r;
  if (a.slabs) {            // this workgroup's own [32][128] tile: no atomics, no contention (gank_sum_slabs adds the tiles up)
    float* sl = a.slabs + (long)blockIdx.x * 4096;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int ee = kg == 0 ? e : 8 + e;
      const int k = (ee & 3) + 8 * (ee >> 2) + 4 * h;
      sl[k * 128 + col] = (kg == 0 ? accw[e] : accw[8 + e]) + ex_in[e * 64 + lane];
    }
    return;
  }
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int ee = kg == 0 ? e : 8 + e;
    const float v = (kg == 0 ? accw[e] : accw[8 + e]) + ex_in[e * 64 + lane];
    const int k = (ee & 3) + 8 * (ee >> 2) + 4 * h;
    float* dst = nullptr;
    if (k < 27) {
      dst = a.dw1 + (long)k * a.Cin + cg * 128 + col;
#ifdef GANK_TUNING          // contention experiments: one copy of the tile per XCD (8) / per workgroup (16); the caller's buffer is that large
      if (a.dbg & 8) dst += (blockIdx.x & 7) * 8192;
      if (a.dbg & 16) dst += blockIdx.x * 8192;
#endif
    } else if (k == 27) dst = a.db1 ? a.db1 + cg * 128 + col : nullptr;
    else if (k < 31) dst = (shortcut && a.dws) ? a.dws + (long)(k - 28) * a.Cout + col : nullptr;
    else dst = (shortcut && a.dbs) ? a.dbs + col : nullptr;
    if (dst) atomicAdd(dst, v);
  }
}

static int resident_xcd_env() {
  static const int v = gank_tune("GANK_RESIDENT_XCD", 1);   // experiment knob: 0 = hardware block order (neighbouring ids round-robin over the XCDs)
  return v;
}
static bool cpool_res_geom_ok(int Hp, int Wp) { return Hp % 8 == 0 && (Wp % 16 == 0 || Wp == 8); }
static int cpool_k2_env() {
  static const int v = gank_tune("GANK_CPOOL_K2", 1);   // experiment knob: GANK_CPOOL_K2=0 keeps the one-group kernel for 8-wide patches
  return v;
}

extern "C" int gank_cpool_res_fprop(const void* x, const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Hp,
                                    int Wp, int Cin, int Cout, int flags, void* stream) {
  GANK_REQUIRE(x && w_rfrag && y && N > 0, "cpool_res_fprop: null pointer");
  GANK_REQUIRE(Cin % 64 == 0 && Cout % 128 == 0 && cpool_res_geom_ok(Hp, Wp),
               "cpool_res_fprop: needs Cin %% 64 == 0, Cout %% 128 == 0, Hp %% 8 == 0 and Wp %% 16 == 0 or Wp == 8 (got %d, %d, %dx%d)", Cin, Cout, Hp, Wp);
  GANK_REQUIRE((long)N * 4 * Hp * Wp * Cin < (1L << 30) && (long)Cout * 16 * Cin * 2 < (1L << 31), "cpool_res_fprop: tensor too large (32-bit byte offsets)");
  CpFwdArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)w_rfrag; a.bias = bias; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.N = N; a.Hp = Hp; a.Wp = Wp; a.Cin = Cin; a.Cout = Cout; a.relu = (flags & GANK_IN_RELU) ? 1 : 0;
  a.xcd = resident_xcd_env();
#ifdef GANK_TUNING
  a.dbg = gank_tune("GANK_CPOOL_DBG", 0);
#endif
  hipStream_t s = (hipStream_t)stream;
  const double M = (double)N * Hp * Wp;
  gank_prof_begin(0, 2.0 * M * Cout * 16.0 * Cin, s, 2.0 * (4.0 * M * Cin + 16.0 * Cin * Cout + M * Cout + (residual ? M * Cout : 0.0)));
#ifdef GANK_TUNING
  static const int taps_env = gank_tune("GANK_CPOOL_TAPS", 16);   // timing only: 9 = the step count of a box-filter + 3x3 stride-2 form (results are wrong)
  if (taps_env == 9 && Wp % 16 == 0) {
    const int grid = N * (Hp / 8) * (Wp / 16) * (Cout / 128);
    GANK_MAX_DYNAMIC_LDS((cpool_res_fprop_kernel<16, 2, 12, 9>), CpGeom<16>::IMG, "cpool_res_fprop");
    hipLaunchKernelGGL((cpool_res_fprop_kernel<16, 2, 12, 9>), dim3(grid), dim3(512), CpGeom<16>::IMG, s, a);
  } else if (taps_env == 9 && Cin % 128 == 0 && N * (Hp / 8) * (Cout / 128) < 256) {
    const int grid = N * (Hp / 8) * (Cout / 64);
    GANK_MAX_DYNAMIC_LDS((cpool_res_fprop_k2_kernel<12, 9>), 2 * CpGeom<8>::IMG, "cpool_res_fprop");
    hipLaunchKernelGGL((cpool_res_fprop_k2_kernel<12, 9>), dim3(grid), dim3(512), 2 * CpGeom<8>::IMG, s, a);
  } else
#endif
  if (Wp % 16 == 0) {
    const int grid = N * (Hp / 8) * (Wp / 16) * (Cout / 128);
    GANK_MAX_DYNAMIC_LDS((cpool_res_fprop_kernel<16, 2, 8>), CpGeom<16>::IMG, "cpool_res_fprop");
    gank_prof_tag(0, "cpool_res_fprop_kernel<16, 2, 8>");
    hipLaunchKernelGGL((cpool_res_fprop_kernel<16, 2, 8>), dim3(grid), dim3(512), CpGeom<16>::IMG, s, a);
  } else if (Cin % 128 == 0 && cpool_k2_env() && N * (Hp / 8) * (Cout / 128) < 256) {
    // fewer one-per-CU workgroups than CUs: 64 output channels per workgroup, the reduction split over two wave groups
    const int grid = N * (Hp / 8) * (Cout / 64);
    GANK_MAX_DYNAMIC_LDS((cpool_res_fprop_k2_kernel<8>), 2 * CpGeom<8>::IMG, "cpool_res_fprop");
    gank_prof_tag(0, "cpool_res_fprop_k2_kernel<8>");
    hipLaunchKernelGGL((cpool_res_fprop_k2_kernel<8>), dim3(grid), dim3(512), 2 * CpGeom<8>::IMG, s, a);
  } else {
    const int grid = N * (Hp / 8) * (Cout / 128);
    GANK_MAX_DYNAMIC_LDS((cpool_res_fprop_kernel<8, 1, 8>), CpGeom<8>::IMG, "cpool_res_fprop");
    gank_prof_tag(0, "cpool_res_fprop_kernel<8, 1, 8>");
    hipLaunchKernelGGL((cpool_res_fprop_kernel<8, 1, 8>), dim3(grid), dim3(512), CpGeom<8>::IMG, s, a);
  }
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("cpool_res_fprop");
  return 0;
}

extern "C" int gank_cpool_res_dgrad(const void* dy, const void* w_rfrag, const void* relu_ref, void* dx, int N, int Hp, int Wp, int Cin,
                                    int Cout, void* stream) {
  GANK_REQUIRE(dy && w_rfrag && dx && N > 0, "cpool_res_dgrad: null pointer");
  GANK_REQUIRE(Cout == 128 && Cin % 128 == 0 && cpool_res_geom_ok(Hp, Wp),
               "cpool_res_dgrad: needs Cout == 128, Cin %% 128 == 0, Hp %% 8 == 0 and Wp %% 16 == 0 or Wp == 8 (got %d, %d, %dx%d)", Cout, Cin, Hp, Wp);
  GANK_REQUIRE((long)N * 4 * Hp * Wp * Cin < (1L << 30), "cpool_res_dgrad: tensor too large (32-bit byte offsets)");
  CpBwdArgs a{};
  a.dy = (const bf16*)dy; a.w = (const bf16*)w_rfrag; a.mask = (const bf16*)relu_ref; a.dx = (bf16*)dx;
  a.N = N; a.Hp = Hp; a.Wp = Wp; a.Cin = Cin; a.Cout = Cout;
  a.xcd = resident_xcd_env();
  hipStream_t s = (hipStream_t)stream;
  const double M = (double)N * Hp * Wp;
  gank_prof_begin(0, 2.0 * M * 4.0 * Cin * 4.0 * Cout, s, 2.0 * (M * Cout + 16.0 * Cin * Cout + 4.0 * M * Cin + (relu_ref ? 4.0 * M * Cin : 0.0)));
  static const int ks_env = gank_tune("GANK_CPOOL_DGRAD_KS", 1);   // experiment knob: 0 keeps the (channel tile) x (pixel group) wave layout
  if (Wp % 16 == 0) {
    const int grid = N * (Hp / 8) * (Wp / 16) * (Cin / 128);
    if (ks_env == 2) {
      constexpr int LDS = CdGeom<16>::IMG + 2 * 4 * 2 * 4 * 64 * 16;
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_ks_kernel<16, 16>), LDS, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_ks_kernel<16, 16>");
      hipLaunchKernelGGL((cpool_res_dgrad_ks_kernel<16, 16>), dim3(grid), dim3(512), LDS, s, a);
    } else if (ks_env) {
      constexpr int LDS = CdGeom<16>::IMG + 2 * 4 * 2 * 4 * 64 * 16;      // + [2 groups][4 channel tiles][2 tiles x 4 quads][64 lanes] x 16 B
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_ks_kernel<16, 8>), LDS, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_ks_kernel<16, 8>");
      hipLaunchKernelGGL((cpool_res_dgrad_ks_kernel<16, 8>), dim3(grid), dim3(512), LDS, s, a);
    } else {
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_kernel<16, 2, 8>), CdGeom<16>::IMG, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_kernel<16, 2, 8>");
      hipLaunchKernelGGL((cpool_res_dgrad_kernel<16, 2, 8>), dim3(grid), dim3(512), CdGeom<16>::IMG, s, a);
    }
  } else {
    const int grid = N * (Hp / 8) * (Cin / 128);
    if (ks_env == 2) {
      constexpr int LDS = CdGeom<8>::IMG + 2 * 4 * 1 * 4 * 64 * 16;
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_ks_kernel<8, 16>), LDS, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_ks_kernel<8, 16>");
      hipLaunchKernelGGL((cpool_res_dgrad_ks_kernel<8, 16>), dim3(grid), dim3(512), LDS, s, a);
    } else if (ks_env) {
      constexpr int LDS = CdGeom<8>::IMG + 2 * 4 * 1 * 4 * 64 * 16;
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_ks_kernel<8, 8>), LDS, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_ks_kernel<8, 8>");
      hipLaunchKernelGGL((cpool_res_dgrad_ks_kernel<8, 8>), dim3(grid), dim3(512), LDS, s, a);
    } else {
      GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_kernel<8, 1, 8>), CdGeom<8>::IMG, "cpool_res_dgrad");
      gank_prof_tag(0, "cpool_res_dgrad_kernel<8, 1, 8>");
      hipLaunchKernelGGL((cpool_res_dgrad_kernel<8, 1, 8>), dim3(grid), dim3(512), CdGeom<8>::IMG, s, a);
    }
  }
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("cpool_res_dgrad");
  return 0;
}

// dy [N,Hp,16,128] -> NO dx: the filter / bias gradients of the 3x3 conv on the 3-channel image in front (dw1 [3,3,3,Cin], db1) and,
// with x_pooled, of the 1x1 shortcut conv on the pooled image whose output gradient dy also is (dws [1,1,3,128], dbs), accumulated.
extern "C" int gank_cpool_res_dgrad_image_wgrad(const void* dy, const void* w_rfrag, const void* relu_ref, const void* x_image,
                                                float* dw1, float* db1, const void* x_pooled, float* dws, float* dbs, int N, int Hp, int Wp,
                                                int Cin, int Cout, float* slabs, void* stream) {
  GANK_REQUIRE(dy && w_rfrag && relu_ref && x_image && dw1 && N > 0, "cpool_res_dgrad_image_wgrad: null pointer");
  GANK_REQUIRE(Cout == 128 && Cin % 128 == 0 && Hp % 8 == 0 && Wp == 16,
               "cpool_res_dgrad_image_wgrad: needs Cout == 128, Cin %% 128 == 0, Hp %% 8 == 0 and Wp == 16 (got %d, %d, %dx%d)", Cout, Cin, Hp, Wp);
  GANK_REQUIRE(!x_pooled || (dws != nullptr), "cpool_res_dgrad_image_wgrad: x_pooled without dws");
  GANK_REQUIRE((long)N * 4 * Hp * Wp * Cin < (1L << 30), "cpool_res_dgrad_image_wgrad: tensor too large (32-bit byte offsets)");
  CpBwdWgArgs a{};
  a.dy = (const bf16*)dy; a.w = (const bf16*)w_rfrag; a.mask = (const bf16*)relu_ref; a.ximg = (const bf16*)x_image;
  GANK_REQUIRE(!slabs || Cin == 128, "cpool_res_dgrad_image_wgrad: the slab form takes Cin == 128 (got %d)", Cin);
  a.xpool = (const bf16*)x_pooled; a.dw1 = dw1; a.db1 = db1; a.dws = dws; a.dbs = dbs; a.slabs = slabs;
  a.N = N; a.Hp = Hp; a.Wp = Wp; a.Cin = Cin; a.Cout = Cout;
  a.xcd = resident_xcd_env();
  static const int dbg = gank_tune("GANK_IMGWG_DBG", 0);
  a.dbg = dbg;
  hipStream_t s = (hipStream_t)stream;
  const double M = (double)N * Hp * Wp;
  // the input gradient's multiply-adds + the two filter gradients'; bytes: dy, the operand, the relu reference, the images (no dx)
  gank_prof_begin(0, 2.0 * M * 4.0 * Cin * 4.0 * Cout + 2.0 * 4.0 * M * 28.0 * Cin + (x_pooled ? 2.0 * M * 4.0 * Cout : 0.0), s,
                  2.0 * (M * Cout + 16.0 * Cin * Cout + 4.0 * M * Cin + 4.0 * M * 3 + (x_pooled ? M * 3 : 0.0)));
  const int grid = N * (Hp / 8) * (Cin / 128);
  GANK_MAX_DYNAMIC_LDS((cpool_res_dgrad_imgwg_kernel<8>), WG_LDS, "cpool_res_dgrad_image_wgrad");
  gank_prof_tag(0, "cpool_res_dgrad_imgwg_kernel<8>");
  hipLaunchKernelGGL((cpool_res_dgrad_imgwg_kernel<8>), dim3(grid), dim3(512), WG_LDS, s, a);
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("cpool_res_dgrad_image_wgrad");
  return 0;
}

// ==================================================================================================================
// 3x3 SAME convolution on 16x16 images, one image x 128 output channels per workgroup: the critic's D.Block.2.Conv1
// (256 -> 256 at 16x16, gan_cifar_resnet.py:186-190 with resample='down') forward and input gradient, 2 x 38.6 GFLOP per
// update at n = 128.  On the LDS-patch kernel these ran at 0.92 PFLOP/s with the matrix pipe 34 % busy: a barrier and a
// weight tile through LDS per (tap, 64-channel chunk) step, 36 barriers per block.  Here, as in the kernels above: the image
// (18 x 18 halo pixels of a 64-channel chunk, 51 KB; TWO chunks resident, the next one requested before the current one's
// MFMAs and stored behind them) stays in LDS for all 9 taps with ONE barrier per chunk, and the weights stream from L2 in
// MFMA-fragment order ("rfrag", prep kind 4) straight into a register ring.  256 pixels per image make the weight stream
// small against the arithmetic (1.2 MB requested per workgroup for 576 MFMAs per wave), so, unlike the 8x8 kernels, this one
// is bound by the matrix pipe: timing-only builds without the weight reloads and without the pixel-fragment reads (garbage
// results, same MFMAs) ran 32.9 / 33.5 / 31.2 us (neither / either / both removed: 36.2 us) at n = 128 -- the MFMA loop at the
// ~1.5 GHz the chip holds under it plus ~6 us of prologue, epilogue and launch; one wave per SIMD with 8 tiles per wave was slower
// (39.7 us), deeper or shallower weight rings made no difference.  Waves: (ct = 4 tiles of 32 output channels) x (pg = upper / lower 8 image rows), 4 pixel
// tiles (2 rows x 16 columns) per wave.  Pixel pitch 144 B, row pitch 2816 B (= 11 x 256: the conflict-free pair of the patch kernel).
// ==================================================================================================================
namespace {
constexpr int I16_PP = 144, I16_RP = 2816, I16_IMG = 18 * I16_RP;      // 50688 bytes per chunk image
struct I16Args {
  const bf16* x;          // [N,16,16,Cin]
  const bf16* w;          // rfrag kind 4: [Cout/32][9 taps][Cin/16][64 lanes][8]
  const float* bias;      // optional [Cout]; with bias_labels: [V][9][Cout], row (label of the sample, border class of the pixel)
  const int* bias_labels; // optional [N] (label_conv.hip: the spatially constant input channels of the layer, factored out)
  int bias_V;
  // rider (gank_img16_conv3x3_label_bwd): lb.blocks extra workgroups behind the main_blocks of the conv compute the label gradients of
  // the factored layer (label_conv_dev.h) -- independent of this launch's own work, no launch of their own
  LabelBwdArgs lb;
  int main_blocks;
  const bf16* mask;       // optional [N,16,16,Cout]: result zeroed where mask <= 0 (relu backward)
  const bf16* res;        // optional [N,16,16,Cout]: added last (res_pitch > 0: rows of res_pitch channels, the first Cout of them; scaled by res_scale)
  int res_pitch;
  float res_scale;
  bf16* y;                // [N,16,16,Cout]
  float* stat_sums;       // optional [groups][GANK_STAT_SLOTS][2][Cout]: batch-norm statistics of (y - bias), as gank_res8_conv3x3
  int N, Cin, Cout, relu; // relu: on the input operand while it is staged
  int res_up, stat_n_per_group;      // res_up: res is [N,8,8,Cout], added nearest-neighbour upsampled
  int xcd;                // XCD-aware block order: the Cout/128 workgroups of an image share an L2
  // NORM: relu(cond_batchnorm(x)) applied while a chunk is staged (the workgroup owns ONE sample: one (scale, shift) row pair)
  const float* cbn_stats; // [groups][2][Cin] (mean, invstd)
  const float* cbn_gamma; // [n_labels][Cin]
  const float* cbn_beta;
  const int* cbn_labels;  // [N]
  int cbn_n_per_group, cbn_n_labels;
};
}  // namespace

// TW = pixel tiles per wave: 4 -> 8 waves = (4 channel tiles) x (upper / lower half of the image); 8 -> 4 waves, one per SIMD, each
// weight fragment feeding 8 MFMAs (half the weight requests, no second wave contending for the SIMD's matrix pipe).
// HALF: a workgroup owns 8 of the 16 image rows (10 halo rows resident) -- for launches whose whole-image grid leaves CUs idle
// (the critic's 128 -> 128 layer at n = 128 is 128 workgroups on 256 CUs).  Its 8 waves are (4 channel tiles) x (2 halves of
// the reduction: K-steps 0,1 / 2,3 of every tap), 4 pixel tiles each, so every weight fragment is still requested once per
// workgroup and feeds 4 MFMAs; the two partial sums of a tile meet through LDS after the last chunk and each wave of a pair
// finishes two of the four tiles.
template <int PF, int TW, bool HALF = false, bool NORM = false>
__global__ __launch_bounds__(HALF ? 512 : 2048 / TW) void img16_conv3x3_kernel(I16Args a) {
  static_assert(!HALF || TW == 4, "half-image form: 4 pixel tiles per wave");
  constexpr int NT = HALF ? 512 : 2048 / TW;
  constexpr int HROWS = HALF ? 10 : 18, IMG = HROWS * I16_RP;            // resident halo rows, bytes per chunk image
  constexpr int STEPS = HALF ? 18 : 36, KPT = HALF ? 2 : 4;              // K-steps of a chunk this wave executes; per tap
  extern __shared__ __attribute__((aligned(16))) char smem[];          // [2][IMG] (HALF: at least the 64-KB exchange area)
  if (a.lb.blocks > 0 && (int)blockIdx.x >= a.main_blocks) {
    label_conv_bwd_block(a.lb, blockIdx.x - a.main_blocks, reinterpret_cast<float*>(smem));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, pg = (HALF || TW == 8) ? 0 : wave >> 2, kh = HALF ? wave >> 2 : 0;
  const int r = lane & 31, h = lane >> 5;
  const int cgroups = a.Cout >> 7;
  const int lid = a.xcd ? xcd_remap(blockIdx.x, a.lb.blocks > 0 ? a.main_blocks : (int)gridDim.x) : blockIdx.x;
  const int bid = HALF ? lid >> 1 : lid, row0 = HALF ? (lid & 1) * 8 : 0;
  const int cg = bid % cgroups, n = bid / cgroups;
  const int nchunks = a.Cin >> 6, kq = a.Cin >> 4;                      // 64-channel chunks; 16-channel K-steps per tap
  const int trow = r >> 4, tcol = r & 15;
  const int b_base = (pg * 2 * TW + trow) * I16_RP + tcol * I16_PP + h * 16 + kh * 64; // tile t adds 2 t rows; tap (ty, tx) adds ty rows, tx pixels

  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, a.Cout * 9 * a.Cin * 2, 0x00020000);
  const int tile = cg * 4 + ct;
  // fragment stream of this wave: chunk-major, then tap, then the chunk's K-steps of that tap (4, or this wave's 2): step s of
  // chunk c sits at wbase + 4096 c + toff[s / KPT] + 1024 (s % KPT).  toff[] is scalar state computed once (a division per step
  // cost 18 scalar instructions per MFMA group: SQ_INSTS_SALU was 4.6 x SQ_INSTS_MFMA)
  const int wbase = tile * 9 * kq * 1024 + kh * 2048;
  int toff[9];
#pragma unroll
  for (int t = 0; t < 9; t++) toff[t] = t * kq * 1024;
  static_assert(PF <= STEPS, "the ring spans at most one chunk");
  u32x4 ring[PF];
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wbase + toff[s / KPT] + (s % KPT) * 1024, 0);

  f32x16 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

  // staging: HROWS x 18 halo pixels x 8 sixteen-byte pieces of the chunk
  constexpr int NPIECE = HROWS * 18 * 8, NLD = (NPIECE + NT - 1) / NT;
  const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * 256 * a.Cin * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
  int p_off[NLD], p_lds[NLD];
#pragma unroll
  for (int j = 0; j < NLD; j++) {
    const int q = tid + j * NT;
    const bool on = q < NPIECE;
    const int hp = q >> 3, c16 = q & 7;
    const int hr = hp / 18, hc = hp - hr * 18;
    const bool ok = on && (unsigned)(row0 + hr - 1) < 16u && (unsigned)(hc - 1) < 16u;
    p_off[j] = ok ? (((n * 16 + row0 + hr - 1) * 16 + hc - 1) * a.Cin + c16 * 8) * 2 : OOB;
    p_lds[j] = on ? hr * I16_RP + hc * I16_PP + c16 * 16 : -1;
  }
  u32x4 rP[NLD];
  // NORM: this thread's pieces are channels 8 (tid & 7) .. + 7 of every chunk (NT is a multiple of 8): the sample's mean / invstd /
  // gamma / beta for them ride along with the chunk's loads; the arithmetic and its order are the forward CBN kernel's
  // ((x - mean) * invstd * gamma + beta, relu, ONE rounding -- gank_cbn_relu_conv3x3_fprop's staging, expression for expression),
  // so the result equals cbn_apply + this conv bit for bit and the normalised tensor is never stored.  Padding stays zero.
  static_assert(!NORM || NT % 8 == 0, "a thread's pieces must share their channel group");
  f32x4 nm[NORM ? 8 : 1];
  const float* np_mu = nullptr;
  const float* np_ga = nullptr;
  const float* np_be = nullptr;
  if constexpr (NORM) {
    int lb = a.cbn_labels[n];
    lb = lb < 0 ? 0 : (lb >= a.cbn_n_labels ? a.cbn_n_labels - 1 : lb);
    np_mu = a.cbn_stats + (long)(n / a.cbn_n_per_group) * 2 * a.Cin + (tid & 7) * 8;
    np_ga = a.cbn_gamma + (long)lb * a.Cin + (tid & 7) * 8;
    np_be = a.cbn_beta + (long)lb * a.Cin + (tid & 7) * 8;
  }
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int j = 0; j < NLD; j++) rP[j] = __builtin_amdgcn_raw_buffer_load_b128(rx, p_off[j] == OOB ? OOB : p_off[j] + c * 128, 0, 0);
    if constexpr (NORM) {
      nm[0] = *reinterpret_cast<const f32x4*>(np_mu + c * 64); nm[1] = *reinterpret_cast<const f32x4*>(np_mu + c * 64 + 4);
      nm[2] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64); nm[3] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64 + 4);
      nm[4] = *reinterpret_cast<const f32x4*>(np_ga + c * 64); nm[5] = *reinterpret_cast<const f32x4*>(np_ga + c * 64 + 4);
      nm[6] = *reinterpret_cast<const f32x4*>(np_be + c * 64); nm[7] = *reinterpret_cast<const f32x4*>(np_be + c * 64 + 4);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int j = 0; j < NLD; j++)
      if (p_lds[j] >= 0) {
        u32x4 v = rP[j];
        if constexpr (NORM) {
          if (p_off[j] != OOB) {
            const bf16x8 xv = __builtin_bit_cast(bf16x8, v);
            bf16x8 o;
#pragma unroll
            for (int e = 0; e < 8; e++) {
              const float mu = e < 4 ? nm[0][e] : nm[1][e - 4], iv = e < 4 ? nm[2][e] : nm[3][e - 4];
              const float ga = e < 4 ? nm[4][e] : nm[5][e - 4], be = e < 4 ? nm[6][e] : nm[7][e - 4];
              const float t = (bf2f(xv[e]) - mu) * iv * ga + be;
              o[e] = f2bf(fmaxf(t, 0.f));
            }
            v = __builtin_bit_cast(u32x4, o);
          }
        } else {
          if (a.relu) v = relu_bf16x8(v);
        }
        *reinterpret_cast<u32x4*>(smem + buf * IMG + p_lds[j]) = v;
      }
  };
  load_chunk(0);
  store_chunk(0);
  __syncthreads();

#pragma unroll 1
  for (int c = 0; c < nchunks; c++) {
    const char* img = smem + (c & 1) * IMG;
    const bool more = c + 1 < nchunks;
    const int cbase = wbase + c * 4096;
    if (more) load_chunk(c + 1);                                       // in flight during this chunk's steps
    static_assert(STEPS % PF == 0, "ring position is chunk-invariant");
    constexpr int PB = 2;                                               // pixel fragments are read PB steps ahead of their MFMAs
    bf16x8 bq[PB + 1][TW];
    auto read_b = [&](int s, bf16x8 (&dst)[TW]) {
      const int tap = s / KPT, kk = s % KPT, ty = tap / 3, tx = tap - 3 * ty;
#pragma unroll
      for (int t = 0; t < TW; t++)
        dst[t] = *reinterpret_cast<const bf16x8*>(img + b_base + (2 * t + ty) * I16_RP + tx * I16_PP + kk * 32);
    };
#pragma unroll
    for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
    for (int s = 0; s < STEPS; s++) {
      if (s + PB < STEPS) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
      __builtin_amdgcn_sched_barrier(0);                                // see res_conv3x3: keeps reads early and the ring deep
#pragma unroll
      for (int t = 0; t < TW; t++) acc[t] = GANK_MFMA32(fa, bq[s % (PB + 1)][t], acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      {
        const int nx = s + PF;                                          // compile-time after unrolling
        if (nx < STEPS) ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, cbase + toff[nx / KPT] + (nx % KPT) * 1024, 0);
        else if (more) ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, cbase + 4096 + toff[(nx - STEPS) / KPT] + ((nx - STEPS) % KPT) * 1024, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      store_chunk((c + 1) & 1);          // that image was last read in chunk c - 1: every wave is past it (the barrier below, one chunk ago)
      __syncthreads();
    }
  }

  if constexpr (HALF) {
    // the two halves of the reduction meet: wave (ct, kh) hands the partial sums of the tiles its partner finishes (2 (1 - kh), +1)
    // to LDS [wave][2 tiles][16][64 lanes] and adds the partner's partial sums of its own tiles (2 kh, +1).  Register arrays are
    // indexed with constants under the wave-uniform branch (a run-time index would move acc[] to scratch)
    __syncthreads();                                 // every wave is done reading the images
    float* xch = reinterpret_cast<float*>(smem);
    auto send = [&](auto T0) {
      constexpr int t0 = decltype(T0)::value;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) xch[((wave * 2 + j) * 16 + e) * 64 + lane] = acc[t0 + j][e];
    };
    auto recv = [&](auto T0) {
      constexpr int t0 = decltype(T0)::value;
      const int partner = wave ^ 4;
#pragma unroll
      for (int j = 0; j < 2; j++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t0 + j][e] += xch[((partner * 2 + j) * 16 + e) * 64 + lane];
    };
    if (kh == 0) send(std::integral_constant<int, 2>{}); else send(std::integral_constant<int, 0>{});
    __syncthreads();
    if (kh == 0) recv(std::integral_constant<int, 0>{}); else recv(std::integral_constant<int, 2>{});
  }

  // epilogue: after acc_widen the lane holds channels 16q + 8h .. +7 of one pixel: 16-byte pieces.  Statistics (optional): the
  // sums of v = acc + residual (the result without its bias) and v^2 per channel, as the 8x8 kernel accumulates them
  const bool stats = a.stat_sums != nullptr;
  float s1[2][8], s2[2][8];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int e = 0; e < 8; e++) { s1[q][e] = 0.f; s2[q][e] = 0.f; }
#pragma unroll
  for (int t = 0; t < TW; t++) {
    if (HALF && (t >> 1) != kh) continue;            // wave-uniform: this wave finishes tiles 2 kh, 2 kh + 1
    const int py = row0 + pg * 2 * TW + 2 * t + trow;
    const long m = (long)n * 256 + py * 16 + tcol;
    const long mr = a.res_up ? (long)n * 64 + (py >> 1) * 8 + (tcol >> 1) : m;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int co = cg * 128 + ct * 32 + 16 * q + 8 * h;
      float v[8];
      acc_widen(acc[t], q, 1.0f, v);
      if (a.mask) {
        const bf16x8 mk = *reinterpret_cast<const bf16x8*>(a.mask + m * a.Cout + co);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = bf2f(mk[e]) > 0.f ? v[e] : 0.f;
      }
      if (a.res) {
        const bf16x8 rs = *reinterpret_cast<const bf16x8*>(a.res + mr * (a.res_pitch > 0 ? a.res_pitch : a.Cout) + co);
        if (a.res_pitch > 0) {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] += a.res_scale * bf2f(rs[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 8; e++) v[e] += bf2f(rs[e]);
        }
      }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) {
        const float* bp = a.bias + co;
        if (a.bias_labels) {
          int lb = a.bias_labels[n];
          lb = lb < 0 ? 0 : (lb >= a.bias_V ? a.bias_V - 1 : lb);
          const int cls = (py == 0 ? 0 : (py == 15 ? 2 : 1)) * 3 + (tcol == 0 ? 0 : (tcol == 15 ? 2 : 1));
          bp += ((long)lb * 9 + cls) * a.Cout;
        }
        b0 = *reinterpret_cast<const f32x4*>(bp); b1 = *reinterpret_cast<const f32x4*>(bp + 4);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        s1[q][e] += v[e];
        s2[q][e] += v[e] * v[e];
        o[e] = f2bf(v[e] + (e < 4 ? b0[e] : b1[e - 4]));
      }
      *reinterpret_cast<bf16x8*>(a.y + m * a.Cout + co) = o;
    }
  }
  if (stats) {
    // per wave: 32 channels x 2 statistics, each the sum over the 32 lanes r of a half-wave (channel 16q + 8h + e); through LDS
    // ([stat][q][h][e][r] floats per wave), then lane (stat, channel) adds its 32 values -> ONE full-width atomic per wave
    __syncthreads();                                 // every wave is done reading the images
    float* red = reinterpret_cast<float*>(smem) + wave * 2048;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        red[((0 * 2 + q) * 2 + h) * 256 + e * 32 + r] = s1[q][e];
        red[((1 * 2 + q) * 2 + h) * 256 + e * 32 + r] = s2[q][e];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int st = lane >> 5, cw = lane & 31, qq = cw >> 4, hh = (cw >> 3) & 1, ee = cw & 7;
    const float* src = red + ((st * 2 + qq) * 2 + hh) * 256 + ee * 32;
    float tsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + 4 * i);
      tsum += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    float* dst = a.stat_sums + ((long)(n / a.stat_n_per_group) * GANK_STAT_SLOTS + (blockIdx.x % GANK_STAT_SLOTS)) * 2 * a.Cout;
#ifdef GANK_TUNING
    if (gank_stats_dbg) return;
#endif
    atomicAdd(dst + st * a.Cout + cg * 128 + ct * 32 + cw, tsum);
  }
}

__global__ void i16_zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

namespace {
struct I16Cbn { const int* labels; const float* gamma; const float* beta; const float* stats; int groups, n_labels; };
}
static int img16_conv3x3_impl(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                              int N, int Cin, int Cout, int flags, float* stat_sums, int stat_groups, const I16Cbn* cbn, void* stream,
                              const int32_t* bias_labels = nullptr, int bias_V = 0, const LabelBwdArgs* lb = nullptr, int res_pitch = 0, float res_scale = 1.f);
// the layer's spatially constant input channels factored out (label_conv.hip): bias_table [V][9][Cout] from gank_label_conv3x3_table
// holds, per label and border class of a pixel, the layer's bias plus what those channels contribute; x and w_rfrag are the
// remaining (feature) channels only
extern "C" int gank_img16_conv3x3_label_bias(const void* x, const void* w_rfrag, const float* bias_table, const int32_t* labels, int V, void* y,
                                             int N, int Cin, int Cout, int flags, void* stream) {
  GANK_REQUIRE(bias_table && labels && V > 0, "img16_conv3x3_label_bias: null table / labels");
  return img16_conv3x3_impl(x, w_rfrag, bias_table, nullptr, nullptr, y, N, Cin, Cout, flags, nullptr, 0, nullptr, stream, labels, V);
}
// the input gradient of the factored layer (x = dy, w_rfrag = the feature half's dgrad operand, relu_ref = the features) with the label
// gradients -- gank_label_conv3x3_bwd's second launch on tap sums that are already in `tap_sums` (gank_conv2d_wgrad_slabs_rows_tap_sums) --
// computed by extra workgroups of the same launch
extern "C" int gank_img16_conv3x3_label_bwd(const void* x, const void* w_rfrag, const void* relu_ref, void* y, int N, int Cin, int Cout, int flags,
                                            const float* tap_sums, const void* T, int V, const float* w, int Cin_total, int c0, int C2, int CoutW,
                                            float* dw, float* de_parts, void* stream) {
  GANK_REQUIRE(tap_sums && T && w && dw && de_parts && V > 0 && V <= LCB_V && C2 % LCB_CT == 0 && CoutW <= 256 && CoutW % 4 == 0 && c0 >= 0 && c0 + C2 <= Cin_total,
               "img16_conv3x3_label_bwd: bad label-gradient arguments");
  const LabelBwdArgs lb{tap_sums, (const bf16*)T, w, dw, de_parts, nullptr, V, Cin_total, c0, C2, CoutW, 0, 9 * (C2 / LCB_CT)};
  return img16_conv3x3_impl(x, w_rfrag, nullptr, relu_ref, nullptr, y, N, Cin, Cout, flags, nullptr, 0, nullptr, stream, nullptr, 0, &lb);
}
// the input gradient behind a fork whose other branch is a 2x2 mean pool (D.Block.2's fan-out: gan_cifar_resnet.py:166-184 with the pooled
// shortcut): dx = relu_mask(conv(dy)) + res_scale * unpool2x(g_pooled[..., :Cout]) in the conv's epilogue -- g_pooled [N,8,8,res_pitch]
// holds the pooled branch's gradient (its first Cout channels belong to this tensor), so the join of the two branch gradients costs no pass
extern "C" int gank_img16_conv3x3_dgrad_unpool(const void* x, const void* w_rfrag, const void* relu_ref, const void* g_pooled, int res_pitch,
                                               float res_scale, void* y, int N, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(g_pooled && res_pitch >= Cout && res_pitch % 8 == 0, "img16_conv3x3_dgrad_unpool: the pooled gradient needs at least Cout channels per pixel");
  return img16_conv3x3_impl(x, w_rfrag, nullptr, relu_ref, g_pooled, y, N, Cin, Cout, GANK_RES_UPSAMPLE2X, nullptr, 0, nullptr, stream, nullptr, 0, nullptr,
                            res_pitch, res_scale);
}
extern "C" int gank_img16_conv3x3_stats(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                                        int N, int Cin, int Cout, int flags, float* stat_sums, int stat_groups, void* stream) {
  return img16_conv3x3_impl(x, w_rfrag, bias, relu_ref, residual, y, N, Cin, Cout, flags, stat_sums, stat_groups, nullptr, stream);
}
// conv3x3_SAME(relu(cond_batchnorm(x))) + bias (+ residual) on 16x16 images: the normalisation of common/ops/normalization.py:47-57 and
// the nonlinearity (gan_cifar_resnet.py:186) applied while the image-resident kernel stages its operand -- for passes that keep
// nothing for a backward pass (the 320-sample generator pass for the critic's fakes, sampling): the normalised tensor is never
// stored.  stats [groups][2][Cin] = (mean, invstd) per tower (gank_cbn_stats / _from_sums); bit-identical to gank_cbn_fwd* +
// gank_img16_conv3x3_stats.
extern "C" int gank_cbn_relu_img16_conv3x3(const void* x, const int32_t* labels, const float* gamma, const float* beta, const float* stats,
                                           const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Cin, int Cout,
                                           int groups, int n_labels, int flags, float* stat_sums, int stat_groups, void* stream) {
  GANK_REQUIRE(labels && gamma && beta && stats, "cbn_relu_img16_conv3x3: null normalisation pointers");
  GANK_REQUIRE(groups > 0 && N % groups == 0 && n_labels > 0, "cbn_relu_img16_conv3x3: batch %d / %d towers", N, groups);
  GANK_REQUIRE(!(flags & GANK_IN_RELU), "cbn_relu_img16_conv3x3: the relu is part of the fused normalisation");
  const I16Cbn cbn{labels, gamma, beta, stats, groups, n_labels};
  return img16_conv3x3_impl(x, w_rfrag, bias, nullptr, residual, y, N, Cin, Cout, flags, stat_sums, stat_groups, &cbn, stream);
}
static int img16_conv3x3_impl(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                              int N, int Cin, int Cout, int flags, float* stat_sums, int stat_groups, const I16Cbn* cbn, void* stream,
                              const int32_t* bias_labels, int bias_V, const LabelBwdArgs* lb, int res_pitch, float res_scale) {
  GANK_REQUIRE(x && w_rfrag && y && N > 0, "img16_conv3x3: null pointer");
  GANK_REQUIRE(Cin % 64 == 0 && Cout % 128 == 0, "img16_conv3x3: needs Cin %% 64 == 0 and Cout %% 128 == 0 (got %d, %d)", Cin, Cout);
  GANK_REQUIRE((flags & ~(GANK_IN_RELU | GANK_RES_UPSAMPLE2X | GANK_STATS_PREZEROED)) == 0, "img16_conv3x3: flags: GANK_IN_RELU, GANK_RES_UPSAMPLE2X, GANK_STATS_PREZEROED");
  GANK_REQUIRE(!(flags & GANK_RES_UPSAMPLE2X) || residual, "img16_conv3x3: GANK_RES_UPSAMPLE2X without a residual");
  GANK_REQUIRE(!stat_sums || (stat_groups > 0 && N % stat_groups == 0), "img16_conv3x3: batch %d not divisible by %d towers", N, stat_groups);
  GANK_REQUIRE(!(bias && relu_ref), "img16_conv3x3: bias and relu_ref together (a forward layer has the bias, an input gradient the mask)");
  GANK_REQUIRE((long)N * 256 * (Cin > Cout ? Cin : Cout) < (1L << 30) && (long)Cout * 9 * Cin * 2 < (1L << 31), "img16_conv3x3: tensor too large (32-bit byte offsets)");
  I16Args a{};
  a.x = (const bf16*)x; a.w = (const bf16*)w_rfrag; a.bias = bias; a.mask = (const bf16*)relu_ref; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.relu = (flags & GANK_IN_RELU) ? 1 : 0;
  a.res_up = (flags & GANK_RES_UPSAMPLE2X) ? 1 : 0; a.stat_sums = stat_sums; a.stat_n_per_group = stat_sums ? N / stat_groups : 1;
  a.xcd = resident_xcd_env();
  a.bias_labels = bias_labels; a.bias_V = bias_V;
  if (lb) a.lb = *lb;
  a.res_pitch = res_pitch; a.res_scale = res_scale;
  const bool half_form = !cbn && (gank_tune("GANK_IMG16_HALF", 1) == 2 || (gank_tune("GANK_IMG16_HALF", 1) == 1 && N * (Cout / 128) < 256));
  a.main_blocks = (half_form ? 2 : 1) * N * (Cout / 128);
  if (cbn) {
    a.cbn_labels = cbn->labels; a.cbn_gamma = cbn->gamma; a.cbn_beta = cbn->beta; a.cbn_stats = cbn->stats;
    a.cbn_n_per_group = N / cbn->groups; a.cbn_n_labels = cbn->n_labels;
  }
  hipStream_t s = (hipStream_t)stream;
  if (stat_sums) gank_stats_dbg_init();
  if (stat_sums && !(flags & GANK_STATS_PREZEROED)) {
    const int nz = stat_groups * GANK_STAT_SLOTS * 2 * Cout;
    hipLaunchKernelGGL(i16_zero_kernel, dim3((nz + 255) / 256), dim3(256), 0, s, stat_sums, nz);
  }
  const double M = (double)N * 256;
  gank_prof_begin(0, 2.0 * M * Cout * 9.0 * Cin, s, 2.0 * (M * Cin + 9.0 * Cin * Cout + M * Cout * (1 + (relu_ref ? 1 : 0) + (residual ? 1 : 0))));
  static const int cfg_env = gank_tune("GANK_IMG16_CFG", 412);   // experiment knob: 100 * (pixel tiles per wave) + weight fragments in flight
  static const int half_env = gank_tune("GANK_IMG16_HALF", 1);   // experiment knob: 0 = whole images only, 1 = half images when the whole-image grid leaves CUs idle, 2 = always
  if (cbn) {          // whole-image form only (the passes this serves have N * Cout / 128 >= 256 workgroups)
    gank_prof_tag(0, "img16_conv3x3_kernel<12, 4, false, true>");
    GANK_MAX_DYNAMIC_LDS((img16_conv3x3_kernel<12, 4, false, true>), 2 * I16_IMG, "cbn_relu_img16_conv3x3");
    hipLaunchKernelGGL((img16_conv3x3_kernel<12, 4, false, true>), dim3(a.main_blocks + a.lb.blocks), dim3(512), 2 * I16_IMG, s, a);
    gank_prof_end(0, s);
    GANK_LAUNCH_OK("cbn_relu_img16_conv3x3");
    return 0;
  }
  if (half_env == 2 || (half_env == 1 && N * (Cout / 128) < 256)) {
    constexpr int HALF_LDS = 2 * 10 * I16_RP > 65536 ? 2 * 10 * I16_RP : 65536;
    static const int hpf_env = gank_tune("GANK_IMG16_HALF_PF", 9);
#define IMG16_LAUNCH_HALF(PF)                                                                             \
  do {                                                                                                    \
    gank_prof_tag(0, "img16_conv3x3_kernel<" #PF ", 4, true, false>");                                         \
    GANK_MAX_DYNAMIC_LDS((img16_conv3x3_kernel<PF, 4, true>), HALF_LDS, "img16_conv3x3");                 \
    hipLaunchKernelGGL((img16_conv3x3_kernel<PF, 4, true>), dim3(a.main_blocks + a.lb.blocks), dim3(512), HALF_LDS, s, a); \
  } while (0)
    switch (hpf_env) {
#ifdef GANK_TUNING
      case 6: IMG16_LAUNCH_HALF(6); break;
      case 18: IMG16_LAUNCH_HALF(18); break;
#endif
      default: IMG16_LAUNCH_HALF(9); break;
    }
#undef IMG16_LAUNCH_HALF
    GANK_LAUNCH_OK("img16_conv3x3");
    gank_prof_end(0, s);
    return 0;
  }
#define IMG16_LAUNCH(PF, TW)                                                                              \
  do {                                                                                                    \
    gank_prof_tag(0, "img16_conv3x3_kernel<" #PF ", " #TW ", false, false>");                                         \
    GANK_MAX_DYNAMIC_LDS((img16_conv3x3_kernel<PF, TW>), 2 * I16_IMG, "img16_conv3x3");                   \
    hipLaunchKernelGGL((img16_conv3x3_kernel<PF, TW>), dim3(a.main_blocks + a.lb.blocks), dim3(2048 / TW), 2 * I16_IMG, s, a); \
  } while (0)
  switch (cfg_env) {
#ifdef GANK_TUNING
    case 406: IMG16_LAUNCH(6, 4); break;
    case 418: IMG16_LAUNCH(18, 4); break;
    case 806: IMG16_LAUNCH(6, 8); break;
    case 812: IMG16_LAUNCH(12, 8); break;
    case 818: IMG16_LAUNCH(18, 8); break;
#endif
    default: IMG16_LAUNCH(12, 4); break;
  }
#undef IMG16_LAUNCH
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("img16_conv3x3");
  return 0;
}
extern "C" int gank_img16_conv3x3(const void* x, const void* w_rfrag, const float* bias, const void* relu_ref, const void* residual, void* y,
                                  int N, int Cin, int Cout, int flags, void* stream) {
  return gank_img16_conv3x3_stats(x, w_rfrag, bias, relu_ref, residual, y, N, Cin, Cout, flags, nullptr, 0, stream);
}

// ==================================================================================================================
// 3x3 SAME convolution on 8x8 images with C = 256 (or 128) input channels, one image per workgroup: the generator's first
// block (gan_cifar_resnet.py:179-207 with resample='up' at 4x4 -> 8x8: Conv1 = NN-upsample + 3x3, Conv2 = 3x3) and their
// input gradients.  These layers are 2048-8192 pixels against 2.4 MB of weights: on the generic gather they ran at
// 0.12-0.4 PFLOP/s (one dependent global -> LDS round trip and a barrier per 64-deep K-step, 16-36 K-steps per block).
// Here, as in the critic's chain above: the image (+ zero halo) is staged ONCE -- the NN-upsample is done by the loader, so
// UpsampleConv needs no phase operands --, the weights stream from L2 in MFMA-fragment order ("rfrag", prep kind 4), a ring
// of 12 one-KB requests in flight per wave, no barrier inside the conv.  A workgroup = 1 image x 128 output channels
// (4 waves x 32 channels x two 32-pixel tiles); grid = N x Cout/128.
// Epilogues: y = acc + residual (full size or half size added NN-upsampled) + bias, with the batch-norm statistics of
// (y - bias) for the layer that consumes y (normalization.py:47); or, for the input gradient of UpsampleConv, the 2x2 sums
// of the 8x8 result (the gradient of the NN-upsample) -> [N,4,4,Cout].
// ==================================================================================================================
namespace {
template <int C> struct G8Geom {
  static constexpr int PPB = C * 2 + 16;                                   // pixel pitch (bytes): odd number of 16-byte units
  static constexpr int RPB = ((10 * (PPB / 16) + 7) / 16 * 16 + 8) * 16;   // halo row pitch: = 8 units mod 16 (two image rows cover 16 distinct bank slots)
  static constexpr int IMG = 10 * RPB;
  static constexpr int KK = C / 16;                                        // MFMA K-steps per tap
  static constexpr int STEPS = 9 * KK;
};
static_assert(G8Geom<128>::RPB == RB_RPB && G8Geom<128>::PPB == RB_PPB, "same image layout as the chain kernels at 128 channels");

constexpr int G8_RED_BYTES = 4 * 2048 * 4;     // statistics epilogue: 2048 floats per wave (reuses the image region)

struct G8Args {
  const bf16* x;       // [N,8,8,C], or [N,4,4,C] with up_in
  const bf16* w;       // rfrag [Cout/32][9][C/16][64][8]
  const float* bias;   // [Cout] or null
  const bf16* res;     // [N,8,8,Cout], [N,4,4,Cout] with res_up, or null
  bf16* y;             // [N,8,8,Cout], or [N,4,4,Cout] with pool_out
  float* stat_sums;    // [groups][GANK_STAT_SLOTS][2][Cout] or null
  int N, Cout, up_in, res_up, pool_out, stat_n_per_group;
  int xcd;
};

template <int C, int PF>
__global__ __launch_bounds__(256) void res8_conv3x3_kernel(G8Args a) {
  using G = G8Geom<C>;
  static_assert(G::STEPS % PF == 0, "ring position");
  constexpr int NT = 256, U = C / 8;                 // U = 16-byte units per pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int ct = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int ntc = a.Cout >> 7;
  const int lid = a.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
  const int n = lid / ntc, half = lid - n * ntc;
  const int co0 = half * 128 + ct * 32;
  const int b_base = (r >> 3) * G::RPB + (r & 7) * G::PPB + h * 16;
  const int wbase = (half * 4 + ct) * G::STEPS * 1024;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, (a.Cout >> 5) * G::STEPS * 1024, 0x00020000);

  u32x4 ring[PF];
#pragma unroll
  for (int s = 0; s < PF; s++) ring[s] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wbase + s * 1024, 0);

  // zero halo: rows 0 and 9 whole, columns 0 and 9 of rows 1..8
  constexpr int RU = G::RPB / 16, PU = G::PPB / 16;
  for (int i = tid; i < 2 * RU + 16 * PU; i += NT) {
    int off;
    if (i < 2 * RU) off = (i < RU ? 0 : 9 * G::RPB - RU * 16) + i * 16;
    else {
      const int j = i - 2 * RU, rr = 1 + j / (2 * PU), k = j % (2 * PU);
      off = rr * G::RPB + (k < PU ? 0 : 9 * G::PPB - PU * 16) + k * 16;
    }
    *reinterpret_cast<u32x4*>(smem + off) = u32x4{0u, 0u, 0u, 0u};
  }
  if (a.up_in) {                                     // NN-upsample in the loader: source pixel (y, x) -> (2y..2y+1, 2x..2x+1)
    const bf16* src = a.x + (long)n * 16 * C;
#pragma unroll
    for (int it = 0; it < 16 * U / NT; it++) {
      const int q = tid + it * NT, px = q / U, u = q % U;
      const u32x4 v = *reinterpret_cast<const u32x4*>(src + px * C + u * 8);
      char* d = smem + (2 * (px >> 2) + 1) * G::RPB + (2 * (px & 3) + 1) * G::PPB + u * 16;
      *reinterpret_cast<u32x4*>(d) = v;
      *reinterpret_cast<u32x4*>(d + G::PPB) = v;
      *reinterpret_cast<u32x4*>(d + G::RPB) = v;
      *reinterpret_cast<u32x4*>(d + G::RPB + G::PPB) = v;
    }
  } else {
    const bf16* src = a.x + (long)n * 64 * C;
#pragma unroll
    for (int it = 0; it < 64 * U / NT; it++) {
      const int q = tid + it * NT, px = q / U, u = q % U;
      *reinterpret_cast<u32x4*>(smem + ((px >> 3) + 1) * G::RPB + ((px & 7) + 1) * G::PPB + u * 16) =
          *reinterpret_cast<const u32x4*>(src + px * C + u * 8);
    }
  }
  __syncthreads();

  f32x16 acc[2];
#pragma unroll
  for (int t = 0; t < 2; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0.f;
  {
    constexpr int PB = 2;                            // pixel fragments are read PB steps ahead of their MFMAs
    u32x4 bq[PB + 1][2];
    auto read_b = [&](int s, u32x4 (&dst)[2]) {
      const int tap = s / G::KK, kk = s % G::KK;
#pragma unroll
      for (int t = 0; t < 2; t++)
        dst[t] = *reinterpret_cast<const u32x4*>(smem + b_base + (4 * t + tap / 3) * G::RPB + (tap % 3) * G::PPB + kk * 32);
    };
#pragma unroll
    for (int s = 0; s < PB; s++) read_b(s, bq[s]);
#pragma unroll
    for (int s = 0; s < G::STEPS; s++) {
      if (s + PB < G::STEPS) read_b(s + PB, bq[(s + PB) % (PB + 1)]);
      const bf16x8 fa = __builtin_bit_cast(bf16x8, ring[s % PF]);
      __builtin_amdgcn_sched_barrier(0);             // keeps the LDS reads ahead of their MFMAs and the weight ring full (see res_conv3x3)
#pragma unroll
      for (int t = 0; t < 2; t++) acc[t] = GANK_MFMA32(fa, __builtin_bit_cast(bf16x8, bq[s % (PB + 1)][t]), acc[t]);
      __builtin_amdgcn_sched_barrier(0);
      if (s + PF < G::STEPS) ring[s % PF] = __builtin_amdgcn_raw_buffer_load_b128(rw, lane * 16, wbase + (s + PF) * 1024, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // epilogue: after acc_widen the lane holds channels co0 + 16q + 8h .. +7 of pixel 32t + r
  if (a.pool_out) {
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int q = 0; q < 2; q++) {
        float v[8];
        acc_widen(acc[t], q, 1.0f, v);
#pragma unroll
        for (int e = 0; e < 8; e++) {
          v[e] += __shfl_xor(v[e], 1);
          v[e] += __shfl_xor(v[e], 8);
        }
        if ((r & 9) == 0) {
          bf16x8 o;
#pragma unroll
          for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
          const int py = 2 * t + (r >> 4), px = (r & 7) >> 1;
          *reinterpret_cast<bf16x8*>(a.y + ((long)n * 16 + py * 4 + px) * a.Cout + co0 + 16 * q + 8 * h) = o;
        }
      }
    return;
  }
  const bool stats = a.stat_sums != nullptr;
  float s1[2][8], s2[2][8];
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int e = 0; e < 8; e++) { s1[q][e] = 0.f; s2[q][e] = 0.f; }
#pragma unroll
  for (int t = 0; t < 2; t++) {
    const int p = 32 * t + r;
    const long rpix = a.res_up ? (long)n * 16 + (p >> 4) * 4 + ((p & 7) >> 1) : (long)n * 64 + p;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int c = co0 + 16 * q + 8 * h;
      float v[8];
      acc_widen(acc[t], q, 1.0f, v);
      if (a.res) {
        const bf16x8 rv = *reinterpret_cast<const bf16x8*>(a.res + rpix * a.Cout + c);
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] += bf2f(rv[e]);
      }
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (a.bias) { b0 = *reinterpret_cast<const f32x4*>(a.bias + c); b1 = *reinterpret_cast<const f32x4*>(a.bias + c + 4); }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        s1[q][e] += v[e];
        s2[q][e] += v[e] * v[e];
        o[e] = f2bf(v[e] + (e < 4 ? b0[e] : b1[e - 4]));
      }
      *reinterpret_cast<bf16x8*>(a.y + ((long)n * 64 + p) * a.Cout + c) = o;
    }
  }
  if (stats) {
    // per wave: 32 channels x 2 statistics, each the sum over the 32 lanes r of one half-wave h (channel 16q + 8h + e).
    // Through LDS: [stat][q][h][e][r] floats per wave, then lane (stat, channel) adds its 32 values -> ONE full-width atomic.
    __syncthreads();                                 // every wave is done reading the image
    float* red = reinterpret_cast<float*>(smem) + ct * 2048;
#pragma unroll
    for (int q = 0; q < 2; q++)
#pragma unroll
      for (int e = 0; e < 8; e++) {
        red[((0 * 2 + q) * 2 + h) * 256 + e * 32 + r] = s1[q][e];
        red[((1 * 2 + q) * 2 + h) * 256 + e * 32 + r] = s2[q][e];
      }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // lane = stat * 32 + channel-in-wave; channel-in-wave = 16 q + 8 h + e
    const int st = lane >> 5, cw = lane & 31, qq = cw >> 4, hh = (cw >> 3) & 1, ee = cw & 7;
    const float* src = red + ((st * 2 + qq) * 2 + hh) * 256 + ee * 32;
    float tsum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; i++) {
      const f32x4 t4 = *reinterpret_cast<const f32x4*>(src + 4 * i);
      tsum += (t4[0] + t4[1]) + (t4[2] + t4[3]);
    }
    float* dst = a.stat_sums + ((long)(n / a.stat_n_per_group) * GANK_STAT_SLOTS + (blockIdx.x % GANK_STAT_SLOTS)) * 2 * a.Cout;
#ifdef GANK_TUNING
    if (gank_stats_dbg) return;
#endif
    atomicAdd(dst + st * a.Cout + co0 + cw, tsum);
  }
}

__global__ void g8_zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
}  // namespace

extern "C" int gank_res8_conv3x3(const void* x, const void* w_rfrag, const float* bias, const void* residual, void* y, int N, int Cin,
                                 int Cout, int flags, float* stat_sums, int stat_groups, void* stream) {
  GANK_REQUIRE(x && w_rfrag && y && N > 0, "res8_conv3x3: null pointer");
  GANK_REQUIRE((Cin == 256 || Cin == 128) && Cout % 128 == 0, "res8_conv3x3: built for Cin 128 | 256 and Cout %% 128 == 0 (got %d, %d)", Cin, Cout);
  const int known = GANK_IN_UPSAMPLE2X | GANK_RES_UPSAMPLE2X | GANK_STATS_PREZEROED | GANK_OUT_POOLSUM2X;
  GANK_REQUIRE((flags & ~known) == 0, "res8_conv3x3: unsupported flags 0x%x", flags & ~known);
  const bool pool_out = (flags & GANK_OUT_POOLSUM2X) != 0;
  GANK_REQUIRE(!pool_out || (!bias && !residual && !stat_sums), "res8_conv3x3: the pooled-sum output takes no bias / residual / statistics");
  GANK_REQUIRE(!(flags & GANK_RES_UPSAMPLE2X) || residual, "res8_conv3x3: GANK_RES_UPSAMPLE2X without a residual");
  GANK_REQUIRE(!stat_sums || (stat_groups > 0 && N % stat_groups == 0), "res8_conv3x3: batch %d not divisible by %d towers", N, stat_groups);
  GANK_REQUIRE((long)N * 64 * (Cin > Cout ? Cin : Cout) < (1L << 30), "res8_conv3x3: tensor too large");
  G8Args a{};
  a.x = (const bf16*)x; a.w = (const bf16*)w_rfrag; a.bias = bias; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.stat_sums = stat_sums; a.N = N; a.Cout = Cout;
  a.up_in = (flags & GANK_IN_UPSAMPLE2X) ? 1 : 0; a.res_up = (flags & GANK_RES_UPSAMPLE2X) ? 1 : 0; a.pool_out = pool_out ? 1 : 0;
  a.stat_n_per_group = stat_sums ? N / stat_groups : 1;
  a.xcd = resident_xcd_env();
  hipStream_t s = (hipStream_t)stream;
  const double px_in = a.up_in ? 16.0 : 64.0, px_out = pool_out ? 16.0 : 64.0;
  gank_prof_begin(0, 2.0 * N * 64.0 * 9.0 * Cin * Cout, s,
                  2.0 * (N * px_in * Cin + 9.0 * Cin * Cout + N * px_out * Cout + (residual ? N * (a.res_up ? 16.0 : 64.0) * Cout : 0.0)));
  gank_prof_tag(0, Cin == 256 ? "res8_conv3x3_kernel<256, 12>" : "res8_conv3x3_kernel<128, 12>");
  if (stat_sums) gank_stats_dbg_init();
  if (stat_sums && !(flags & GANK_STATS_PREZEROED)) {
    const int nz = stat_groups * GANK_STAT_SLOTS * 2 * Cout;
    hipLaunchKernelGGL(g8_zero_kernel, dim3((nz + 255) / 256), dim3(256), 0, s, stat_sums, nz);
  }
  const int grid = N * (Cout / 128);
  if (Cin == 256) {
    constexpr int LDS = G8Geom<256>::IMG > G8_RED_BYTES ? G8Geom<256>::IMG : G8_RED_BYTES;
    GANK_MAX_DYNAMIC_LDS((res8_conv3x3_kernel<256, 12>), LDS, "res8_conv3x3");
    hipLaunchKernelGGL((res8_conv3x3_kernel<256, 12>), dim3(grid), dim3(256), LDS, s, a);
  } else {
    constexpr int LDS = G8Geom<128>::IMG > G8_RED_BYTES ? G8Geom<128>::IMG : G8_RED_BYTES;
    GANK_MAX_DYNAMIC_LDS((res8_conv3x3_kernel<128, 12>), LDS, "res8_conv3x3");
    hipLaunchKernelGGL((res8_conv3x3_kernel<128, 12>), dim3(grid), dim3(256), LDS, s, a);
  }
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("res8_conv3x3");
  return 0;
}
