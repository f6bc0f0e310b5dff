// Implicit-GEMM convolution on v_mfma_f32_32x32x16_bf16 (gfx950): fprop and dgrad.
//
// GEMM view:  Y[m, co] = sum_k  P[m, k] * Wt[co, k]      m = output pixel (n,oh,ow), k = (tap, ci)
//   * MFMA "A" operand (32 rows)  = 32 output channels, fragment = 8 consecutive k of one channel
//   * MFMA "B" operand (32 lanes) = 32 output pixels,   fragment = 8 consecutive k of one pixel
//   so both fragments are one ds_read_b128 from [row][k] LDS tiles, and each lane ends up holding
//   4 consecutive output channels of ONE pixel per accumulator quad -> 8-byte NHWC stores.
//   * K-step = 64 (one tap x 64 input channels on the fast path); LDS rows are 64 bf16 + 8 pad
//     = 144 B, which makes every ds_read_b128 lane group hit 16 distinct 4-bank slots.
//   * generic and LDS-patch kernels: global -> register -> LDS staging: the staging pass is where zero padding,
//     nearest-neighbour upsample, zero insertion, stride and the pre-activation relu are applied,
//     so none of those tensors is ever materialised in HBM;
//   * two-group kernel (conv_igemm_pp_kernel, the large 256-channel layers): both operands by LDS-DMA
//     (buffer_load ... lds), 64-B swizzled rows, one block per CU -- its own header below.
// Replaces tf.nn.conv2d / Conv2DBackpropInput at common/ops/conv2d.py:180-187 and the
// surrounding block-library glue (SNGAN/gan_cifar_resnet.py:112-153,186,198,209,261).
#include "gank_common.h"
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
#include <type_traits>

#ifndef GANK_KMODE
#define GANK_KMODE 0
#endif
#define IG_IN_ZEROINS2X 16
#define IG_IN_STRIDE2 32
#define IG_RES_UP2X 128       // residual is [N,H/2,W/2,Cout]: read at (oh>>1, ow>>1) (public GANK_RES_UPSAMPLE2X)

struct IgemmArgs {
  const bf16* x;
  const bf16* w;
  const float* bias;
  const bf16* res;
  const bf16* mask;
  bf16* y;
  int N, H, W;      // output spatial size
  int Hin, Win;     // stored input spatial size
  int Cin, Cout, CoutPad, Kpad;
  int ks, pad, taps;
  int M;            // N*H*W
  int flags;
  float scale;
  int nsteps;       // Kpad / 64
  int tiles_m, tiles_n;
  int shw, sw;      // log2(H*W), log2(W) or -1
  int korder;       // bit0: tap-inner K order
  int tiles_pp;     // phase mode: pixel tiles per phase
  // conditional-batch-norm statistics of the OUTPUT, accumulated by the epilogue (two-group kernel only):
  // stat_sums [groups][2][Cout] += (sum, sum of squares) of (y - bias) over the samples of each tower; null = off
  float* stat_sums;
  int stat_n_per_group, stat_groups, stat_prezeroed;
  int phase_inner;  // phase mode: block -> (tile, phase) with the phase fastest (default) or slowest
  // patch kernel, MODE bit 3: the input is normalised on its way into LDS -- conditional batch norm + relu of the layer in
  // front (normalization.py:47-57, gan_cifar_resnet.py:257-258) fused into this conv's operand staging
  const float* cbn_stats;    // [groups][2][Cin] (mean, invstd)
  const float* cbn_gamma;    // [n_labels][Cin]
  const float* cbn_beta;
  const int* cbn_labels;     // [N]
  int cbn_n_per_group, cbn_n_labels;
  bf16* aux_out;             // narrow-input kernel, POOL: the 2x2 mean-pooled input it gathered, [N,H,W,CIN] (or null)
};

static int phase_inner_env() {
  static const int v = gank_tune("GANK_PHASE_INNER", 1);   // experiment knob: GANK_PHASE_INNER=0 restores the phase-slowest block order
  return v;
}

constexpr int LROW = 72;  // LDS row length in bf16 (64 + 8 pad) = 144 B

template <int CTRL>
__device__ __forceinline__ float pp_dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// all 16 lanes of a DPP row end up with the row's sum: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float pp_row_sum(float v) {
  v = pp_dpp_add<0xB1>(v);
  v = pp_dpp_add<0x4E>(v);
  v = pp_dpp_add<0x141>(v);
  v = pp_dpp_add<0x140>(v);
  return v;
}

// Epilogue of ONE 32x32 accumulator tile with 16-byte pieces.  An accumulator quad g of lane (r, h) is channels 8g+4h..+3 of
// pixel r; v_permlane32_swap trades quad 2q+1 of the h = 0 half-wave for quad 2q of the h = 1 half-wave, after which the lane
// owns 8 CONSECUTIVE channels 16q + 8h .. +7: half as many store (and mask / residual load) instructions and L2 requests as
// the 8-byte form, and the four pieces of a pixel's 64-byte run leave back to back.  y_px / mask_px / res_px: the pixel's row
// (+ the tile's first channel); bias likewise.  Needs 8 | Cout.  d (optional): the values before the bias, for statistics.
template <bool WANT_D>
__device__ __forceinline__ void epi_tile_wide(const f32x16& acc, float scale, const float* __restrict__ bias, const bf16* __restrict__ mask_px,
                                              const bf16* __restrict__ res_px, bf16* __restrict__ y_px, int h, bool otanh, float (*d)[8]) {
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int c = 16 * q + 8 * h;
    float v[8];
    acc_widen(acc, q, scale, v);
    float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (bias) {
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(bias + c), b1 = *reinterpret_cast<const f32x4*>(bias + c + 4);
#pragma unroll
      for (int e = 0; e < 4; e++) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
    }
#pragma unroll
    for (int e = 0; e < 8; e++) {
      if constexpr (WANT_D) d[q][e] = v[e];
      v[e] += bb[e];
    }
    if (mask_px) {
      const bf16x8 mk = *reinterpret_cast<const bf16x8*>(mask_px + c);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const bool on = bf2f(mk[e]) > 0.f;
        v[e] = on ? v[e] : 0.f;
        if constexpr (WANT_D) d[q][e] = on ? d[q][e] : -bb[e];
      }
    }
    if (res_px) {
      const bf16x8 rs = *reinterpret_cast<const bf16x8*>(res_px + c);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float t = bf2f(rs[e]);
        v[e] += t;
        if constexpr (WANT_D) d[q][e] += t;
      }
    }
    bf16x8 out;
#pragma unroll
    for (int e = 0; e < 8; e++) out[e] = f2bf(otanh ? tanhf(v[e]) : v[e]);
    *reinterpret_cast<bf16x8*>(y_px + c) = out;
  }
}

// PF   = register prefetch slots: global loads for K-step s+PF-1 are in flight while step s is computed
//        (plain loads survive the per-step barrier; hipcc emits counted vmcnt waits for the oldest slot).
// MODE = compile-time gather mode (bit0: relu on the input operand, bit1: shifted index = NN-upsample or
//        zero insertion).  The staging pass runs once per K-step per wave and competes with the MFMAs for
//        issue slots, so its instruction count is what bounds this kernel: runtime flags cost selects,
//        64-bit address arithmetic cost 3x the VALU work, tap decoding by division 100+ SALU per step.
template <int WM, int WN, int TM, int TN, bool PACKED, int PF, int MODE, bool STATS = false>
__global__ __launch_bounds__(WM* WN * 64) void conv_igemm_kernel(IgemmArgs a) {
  static_assert(!STATS || (!PACKED && TN * 16 <= 32), "statistics epilogue: plain operands, at most 32 channels per half-wave");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32;
  constexpr int BN = WN * TN * 32;
  constexpr int CP = BM * 8 / NT;  // 16-byte pixel chunks per thread per step
  constexpr int CW = BN * 8 / NT;  // 16-byte weight chunks per thread per step
  static_assert(CP >= 1 && CW >= 1, "tile too small for the thread count");
  static_assert(!PACKED || (PF == 1 && MODE == 0), "the packed path takes its flags at run time");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sP = reinterpret_cast<bf16*>(smem);             // [2][BM][LROW]
  bf16* sW = sP + 2 * BM * LROW;                        // [2][BN][LROW]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int r = lane & 31, h = lane >> 5;

  const int nwg = a.tiles_m * a.tiles_n;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % a.tiles_n;
  int tile_m = lid / a.tiles_n;
  // MODE bit 2: output-phase decomposition of a stride-2 transposed conv / NN-upsample+3x3 conv.  The grid
  // holds 4 x tiles_pp pixel tiles; phase (a,b) = output parity.  Each phase is a 2x2-tap stride-1 conv over
  // the LOW-RES input with its own weight matrix and pad (1-a, 1-b), written to the (2y+a, 2x+b) positions:
  // 4 taps per output instead of 9 (or 16 with zero insertion).
  constexpr bool PHASE = !PACKED && (MODE & 4) != 0;
  int phase = 0;
  if constexpr (PHASE) {
    // phase fastest: the four phases of a pixel tile get consecutive logical ids = the same XCD, back to back -- they read the
    // same input tile from that L2 and their interleaved output pixels (each phase writes every other pixel of a row) meet
    // in it before they reach HBM.  Phase-slowest put them on four different XCDs: 1.7x the algorithmic write bytes.
    if (a.phase_inner) { phase = tile_m & 3; tile_m >>= 2; }
    else { phase = tile_m / a.tiles_pp; tile_m -= phase * a.tiles_pp; }
  }
  const int pad_h = PHASE ? 1 - (phase >> 1) : a.pad, pad_w = PHASE ? 1 - (phase & 1) : a.pad;

  const int st = (a.flags & IG_IN_STRIDE2) ? 2 : 1;
  const bool shr = PACKED ? (a.flags & (GANK_IN_UPSAMPLE2X | IG_IN_ZEROINS2X)) != 0 : (MODE & 2) != 0;
  const bool zins = (a.flags & IG_IN_ZEROINS2X) != 0;
  const bool inrelu = PACKED ? (a.flags & GANK_IN_RELU) != 0 : (MODE & 1) != 0;
  const int LH = shr ? 2 * a.Hin : a.Hin, LW = shr ? 2 * a.Win : a.Win;

  // per-thread pixel chunk descriptors (fixed for the whole K loop)
  int p_base[CP], p_oh[CP], p_ow[CP];
#pragma unroll
  for (int j = 0; j < CP; j++) {
    const int q = tid + NT * j;
    const int m = tile_m * BM + (q >> 3);
    if (m < a.M) {
      int n, oh, ow;
      pix_decomp(m, a.H, a.W, a.shw, a.sw, n, oh, ow);
      p_base[j] = n * a.Hin * a.Win;
      p_oh[j] = oh * st;
      p_ow[j] = ow * st;
    } else {
      p_base[j] = 0;
      p_oh[j] = -(1 << 20);
      p_ow[j] = 0;
    }
  }

  u32x4 rP[PF][CP], rW[PF][CW];

  // Fast path: buffer loads with 32-bit byte offsets.  Out-of-image taps get an out-of-range voffset,
  // for which the hardware bounds check returns ZEROS: padding costs no branch, no select and no mask,
  // and every load is unconditional (a branch around a load makes hipcc's vmcnt bookkeeping assume the
  // worst path and drain the prefetch ring at every step).
  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(a.x), 0, a.N * a.Hin * a.Win * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(a.w), 0, a.CoutPad * a.Kpad * 2 * (PHASE ? 4 : 1), 0x00020000);
  int p_off[CP];   // byte offset of (n, oh*st, ow*st, cc*8) in x   (non-shifted-index modes)
  int w_off[CW];   // byte offset of (co, cc*8) in w
#pragma unroll
  for (int j = 0; j < CP; j++)
    p_off[j] = ((p_base[j] + p_oh[j] * a.Win + p_ow[j]) * a.Cin + ((tid + NT * j) & 7) * 8) * 2;
#pragma unroll
  for (int j = 0; j < CW; j++) {
    const int q = tid + NT * j;
    w_off[j] = (((PHASE ? phase * a.CoutPad : 0) + tile_n * BN + (q >> 3)) * a.Kpad + (q & 7) * 8) * 2;
  }

  // K-step cursor, advanced incrementally (tap-major, then 64-channel chunk): no divisions in the loop.
  // After the last step it stays put, so the PF-1 trailing refills of the ring re-load the last step.
  const int last = a.nsteps - 1;
  int cur = 0, cur_c0 = 0, cur_tap = 0, cur_dh = -pad_h, cur_dw = -pad_w;
  constexpr bool TAP_INNER = (GANK_KMODE & 1) != 0;   // compile-time experiment knob (build.py GANK_KMODE)
  constexpr int W_AUX = (GANK_KMODE >> 1) == 1 ? 2 : ((GANK_KMODE >> 1) == 2 ? 16 : 0);   // nt / sc1 / default

  auto load_next = [&](u32x4 (&rP)[CP], u32x4 (&rW)[CW]) {
    if constexpr (!PACKED) {
      const int delta = ((cur_dh * a.Win + cur_dw) * a.Cin + cur_c0) * 2;     // wave-uniform (SALU)
#pragma unroll
      for (int j = 0; j < CP; j++) {
        const int ih = p_oh[j] + cur_dh, iw = p_ow[j] + cur_dw;
        bool ok = (unsigned)ih < (unsigned)LH && (unsigned)iw < (unsigned)LW;
        int off;
        if constexpr ((MODE & 2) != 0) {   // upsample / zero-insertion: the stored index is (ih>>1, iw>>1)
          if (zins) ok = ok && (((ih | iw) & 1) == 0);
          off = ((p_base[j] + (ih >> 1) * a.Win + (iw >> 1)) * a.Cin + cur_c0 + ((tid + NT * j) & 7) * 8) * 2;
        } else {
          off = p_off[j] + delta;
        }
        rP[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0);
      }
    } else {
      const int ktot = a.taps * a.Cin;
#pragma unroll
      for (int j = 0; j < CP; j++) {
        const int cc = (tid + NT * j) & 7;
        const int kb = cur * 64 + cc * 8;
        // (tap, channel) of the chunk's first k by ONE division, then counted up: three run-time divisions per ELEMENT made
        // this gather 290 us for a 1.2-GFLOP layer (PGGAN's 513-channel conv behind minibatch-std)
        int tap = kb / a.Cin, ci = kb - tap * a.Cin;
        int th = tap / a.ks;
        int dh = th - a.pad, dw = tap - th * a.ks - a.pad;
        bf16x8 v;
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float val = 0.f;
          if (kb + e < ktot) {
            int ih = p_oh[j] + dh, iw = p_ow[j] + dw;
            bool ok = (unsigned)ih < (unsigned)LH && (unsigned)iw < (unsigned)LW;
            if (zins) ok = ok && (((ih | iw) & 1) == 0);
            if (shr) { ih >>= 1; iw >>= 1; }
            if (ok) {
              val = bf2f(a.x[((long)(p_base[j] + ih * a.Win + iw)) * a.Cin + ci]);
              if (inrelu) val = fmaxf(val, 0.f);
            }
          }
          v[e] = f2bf(val);
          if (++ci == a.Cin) {            // next tap
            ci = 0;
            if (++dw > a.ks - 1 - a.pad) { dw = -a.pad; dh++; }
          }
        }
        rP[j] = __builtin_bit_cast(u32x4, v);
      }
    }
    if constexpr (!PACKED) {
      // weights bypass the CU's 32 KB L1 (nt): they are streamed once per block, while the activation patch of
      // a 64-channel chunk is re-read by all taps of that chunk (tap-inner order below) and should stay in L1
      const int wk = (cur_tap * a.Cin + cur_c0) * 2;
#pragma unroll
      for (int j = 0; j < CW; j++) rW[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[j], wk, W_AUX);
      if (cur < last) {
        cur++;
        if constexpr (TAP_INNER) {  // K order: channel chunk outer, tap inner
          cur_tap++;
          if (++cur_dw > a.ks - 1 - pad_w) { cur_dw = -pad_w; cur_dh++; }
          if (cur_tap >= a.taps) { cur_tap = 0; cur_dh = -pad_h; cur_dw = -pad_w; cur_c0 += 64; }
        } else {                    // K order: tap outer, channel chunk inner
          cur_c0 += 64;
          if (cur_c0 >= a.Cin) {
            cur_c0 = 0; cur_tap++;
            if (++cur_dw > a.ks - 1 - pad_w) { cur_dw = -pad_w; cur_dh++; }
          }
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < CW; j++) rW[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[j], cur * 128, 0);
      if (cur < last) cur++;
    }
  };

  auto store_step = [&](int buf, u32x4 (&rP)[CP], u32x4 (&rW)[CW]) {
#pragma unroll
    for (int j = 0; j < CP; j++) {
      const int q = tid + NT * j;
      u32x4 v = rP[j];
      if constexpr (!PACKED && (MODE & 1) != 0) v = relu_bf16x8(v);
      *reinterpret_cast<u32x4*>(sP + (buf * BM + (q >> 3)) * LROW + (q & 7) * 8) = v;
    }
#pragma unroll
    for (int j = 0; j < CW; j++) {
      const int q = tid + NT * j;
      *reinterpret_cast<u32x4*>(sW + (buf * BN + (q >> 3)) * LROW + (q & 7) * 8) = rW[j];
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; i++)
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

#pragma unroll
  for (int d = 0; d < PF; d++) load_next(rP[d], rW[d]);     // steps 0 .. PF-1 (clamped)
  store_step(0, rP[0], rW[0]);
  __syncthreads();

  // one K-step on ring slot D (compile-time): refill the slot, MFMAs on LDS buffer s&1, stage step s+1
  auto step = [&](int s, auto slot) {
    constexpr int D = decltype(slot)::value;
    const int buf = s & 1;
    if constexpr (PF > 1) {
      load_next(rP[D], rW[D]);     // slot D held step s (already in LDS): now step s+PF
      // stage step s+1 BEFORE this step's MFMAs: its data was requested a full step ago, the other LDS
      // buffer is free since the last barrier, and the write latency now hides behind the MFMAs instead
      // of sitting between them and the barrier
      if (s + 1 < a.nsteps) store_step(buf ^ 1, rP[(D + 1) % PF], rW[(D + 1) % PF]);
    }
    const bf16* pW = sW + (buf * BN + wave_n * TN * 32 + r) * LROW + h * 8;
    const bf16* pP = sP + (buf * BM + wave_m * TM * 32 + r) * LROW + h * 8;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      bf16x8 fa[TN], fb[TM];
#pragma unroll
      for (int i = 0; i < TN; i++) fa[i] = *reinterpret_cast<const bf16x8*>(pW + i * 32 * LROW + kk * 16);
#pragma unroll
      for (int j = 0; j < TM; j++) fb[j] = *reinterpret_cast<const bf16x8*>(pP + j * 32 * LROW + kk * 16);
#pragma unroll
      for (int i = 0; i < TN; i++)
#pragma unroll
        for (int j = 0; j < TM; j++)
          acc[i][j] = GANK_MFMA32(fa[i], fb[j], acc[i][j]);
    }
    if constexpr (PF == 1) {
      if (s + 1 < a.nsteps) {
        load_next(rP[0], rW[0]);
        store_step(buf ^ 1, rP[0], rW[0]);
      }
    }
    __syncthreads();
  };

  // full ring revolutions (no exits inside the unrolled body), then the < PF leftover steps
  int s0 = 0;
  for (; s0 + PF <= a.nsteps; s0 += PF) {
    if constexpr (PF >= 1) step(s0 + 0, std::integral_constant<int, 0>{});
    if constexpr (PF >= 2) step(s0 + 1, std::integral_constant<int, 1 % PF>{});
    if constexpr (PF >= 3) step(s0 + 2, std::integral_constant<int, 2 % PF>{});
    if constexpr (PF >= 4) step(s0 + 3, std::integral_constant<int, 3 % PF>{});
  }
  if constexpr (PF >= 2) { if (s0 + 0 < a.nsteps) step(s0 + 0, std::integral_constant<int, 0>{}); }
  if constexpr (PF >= 3) { if (s0 + 1 < a.nsteps) step(s0 + 1, std::integral_constant<int, 1 % PF>{}); }
  if constexpr (PF >= 4) { if (s0 + 2 < a.nsteps) step(s0 + 2, std::integral_constant<int, 2 % PF>{}); }

  // epilogue: lane holds, per accumulator quad g, channels co0+8g+4h .. +3 of pixel m
  const bool vec = (a.Cout & 3) == 0;
  const bool wide = (a.Cout & 31) == 0;         // whole 32-channel tiles: 16-byte pieces (epi_tile_wide); always with STATS
  const bool otanh = (a.flags & GANK_OUT_TANH) != 0;
  // STATS (host guarantees: full tiles, every tile inside one tower, Cout % BN == 0): per-lane sums of (y - bias) and
  // its square per channel, reduced over the wave's pixels after the stores -- the batch-norm statistics of the layer
  // that consumes y (normalization.py:47), as in the two-group kernel's epilogue
  float st1[STATS ? TN * 16 : 1], st2[STATS ? TN * 16 : 1];
  if constexpr (STATS) {
#pragma unroll
    for (int q = 0; q < TN * 16; q++) { st1[q] = 0.f; st2[q] = 0.f; }
  }
#pragma unroll
  for (int j = 0; j < TM; j++) {
    int m = tile_m * BM + (wave_m * TM + j) * 32 + r;
    if (m >= a.M) continue;
    if constexpr (PHASE) {      // low-res pixel (n,y,x) of this phase -> output pixel (n, 2y+a, 2x+b)
      int n_, y_, x_;
      pix_decomp(m, a.H, a.W, a.shw, a.sw, n_, y_, x_);
      m = (n_ * 2 * a.H + 2 * y_ + (phase >> 1)) * 2 * a.W + 2 * x_ + (phase & 1);
    }
    long mr = m;                 // residual pixel: the output pixel, or its low-resolution parent
    if (!PHASE && (a.flags & IG_RES_UP2X)) {
      int n_, y_, x_;
      pix_decomp(m, a.H, a.W, a.shw, a.sw, n_, y_, x_);
      mr = ((long)(n_ * (a.H >> 1) + (y_ >> 1))) * (a.W >> 1) + (x_ >> 1);
    }
    if (wide) {
#pragma unroll
      for (int i = 0; i < TN; i++) {
        const int ct = tile_n * BN + (wave_n * TN + i) * 32;
        float d[2][8];
        epi_tile_wide<STATS>(acc[i][j], a.scale, a.bias ? a.bias + ct : nullptr, a.mask ? a.mask + (long)m * a.Cout + ct : nullptr,
                             a.res ? a.res + mr * a.Cout + ct : nullptr, a.y + (long)m * a.Cout + ct, h, otanh, d);
        if constexpr (STATS) {
#pragma unroll
          for (int q = 0; q < 2; q++)
#pragma unroll
            for (int e = 0; e < 8; e++) { st1[i * 16 + q * 8 + e] += d[q][e]; st2[i * 16 + q * 8 + e] += d[q][e] * d[q][e]; }
        }
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < TN; i++) {
      const int co0 = tile_n * BN + (wave_n * TN + i) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int co = co0 + 8 * g;
        if (co >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] * a.scale;
        const long o = (long)m * a.Cout + co;
        if (vec) {
          f32x4 b = {0.f, 0.f, 0.f, 0.f};
          if (a.bias) {
            b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += b[e];
          }
          if (a.mask) {
            const bf16x4 mk = *reinterpret_cast<const bf16x4*>(a.mask + o);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (bf2f(mk[e]) > 0.f) ? v[e] : 0.f;
          }
          if (a.res) {
            const bf16x4 rs = *reinterpret_cast<const bf16x4*>(a.res + mr * a.Cout + co);
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] += bf2f(rs[e]);
          }
          bf16x4 out;
#pragma unroll
          for (int e = 0; e < 4; e++) out[e] = f2bf(otanh ? tanhf(v[e]) : v[e]);
          *reinterpret_cast<bf16x4*>(a.y + o) = out;
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (co + e >= a.Cout) break;
            float t = v[e];
            if (a.bias) t += a.bias[co + e];
            if (a.mask) t = (bf2f(a.mask[o + e]) > 0.f) ? t : 0.f;
            if (a.res) t += bf2f(a.res[mr * a.Cout + co + e]);
            a.y[o + e] = f2bf(otanh ? tanhf(t) : t);
          }
        }
      }
    }
  }
  if constexpr (STATS) {
    float keep1 = 0.f, keep2 = 0.f;
#pragma unroll
    for (int q = 0; q < TN * 16; q++) {
      // sum over the 32 lanes of the half-wave (its 32 pixels): DPP inside the 16-lane rows, one shuffle across them
      float s1 = pp_row_sum(st1[q]), s2 = pp_row_sum(st2[q]);
      s1 += __shfl_xor(s1, 16, 64);
      s2 += __shfl_xor(s2, 16, 64);
      keep1 = r == q ? s1 : keep1;
      keep2 = r == q ? s2 : keep2;
    }
    if (r < TN * 16) {          // lane (r, h) holds channel (i, q, e) = (r >> 4, (r >> 3) & 1, r & 7) of its half (epi_tile_wide)
      const int m0 = tile_m * BM;                      // the tile's first pixel (phase mode: low-resolution pixel index)
      const int n0 = a.shw >= 0 ? (m0 >> a.shw) : m0 / (a.H * a.W);
      float* dst = a.stat_sums + ((long)(n0 / a.stat_n_per_group) * GANK_STAT_SLOTS + (blockIdx.x % GANK_STAT_SLOTS)) * 2 * a.Cout;
      const int co = tile_n * BN + (wave_n * TN + (r >> 4)) * 32 + 16 * ((r >> 3) & 1) + 8 * h + (r & 7);
#ifdef GANK_TUNING
      if (gank_stats_dbg) return;
#endif
      atomicAdd(dst + co, keep1);
      atomicAdd(dst + a.Cout + co, keep2);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// LDS-staged input patch variant for plain 3x3 / stride 1 / SAME convolutions (Cin % 64 == 0, W % 16 == 0,
// H % 8 == 0, Cout tile 128).  Evidence (rocprof, same shape 64x32x32x256->256): the NN-upsampled conv, whose
// gather touches 4x fewer distinct cache lines, runs at 1.2-1.3 PFLOP/s while the plain conv runs at 0.76 --
// identical MFMA/LDS/instruction streams, so the plain conv is bound by L1/L2 line requests of the pixel
// operand (every tap re-fetches the shifted 128-pixel tile).  Here one block owns an 8x16 output patch; per
// 64-channel chunk it stages the 10x18 halo ONCE into LDS (23 KB) and all 9 taps read their B fragments
// from it at a shifted row offset: 9x fewer activation line requests, 45% fewer global loads and ds_writes
// per MFMA.  Weights stream exactly as in the generic kernel (2-slot register ring, double-buffered LDS).
// ------------------------------------------------------------------------------------------------------
constexpr int PHALO = 180;   // 10 x 18 halo pixels
// Halo image row pitch (bf16 elements).  A B-fragment ds_read_b128 covers 16 pixels of one halo row (lanes 0-15)
// and 16 of the next (lanes 16-31); its 16-lane groups {0-3,12-15,20-27}, ... mix both rows, and with the natural
// pitch of 18 x 144 B = 2592 B (= 32 mod 256) two lanes of every group share a bank: SQ_LDS_BANK_CONFLICT showed 4
// extra cycles on every B read (8 instead of 4).  A pitch that is a multiple of 256 B keeps the 144-B pixel stride's
// conflict-free slot permutation across the row change.
constexpr int HROWP = 1408;  // 2816 B = 11 x 256 B  (>= 18 * LROW = 1296)

// BN = 128: 2 x 2 waves of 64 pixels x 64 couts.  BN = 32: 4 x 1 waves of 32 pixels x 32 couts, for the 3-channel
// output conv of the generator (G.Output 256 -> 3, padded to one 32-row MFMA tile): the generic 256x32 kernel
// re-fetched the 256-channel pixel operand per tap; here it is staged once per chunk like any other 3x3 conv.
// MODE bit 2 (PHASE): one output phase (a, b) of a stride-2 transposed conv -- UpsampleConv 3x3 fprop, ConvMeanPool
// input gradient -- over an 8x16 LOW-RES patch: 2x2 taps at offsets (i - (1-a), j - (1-b)) out of the same halo,
// weights of phase p at w + p*CoutPad*Kpad (Kpad = 4*Cin), results written to (2y+a, 2x+b).  The per-tap gather of
// the generic phase kernel fetched 16 KB of pixels per 16 KB of weights per step; the halo makes it 23 KB per 4 steps.
template <int MODE, int BN>   // MODE bit0: relu on the input operand
__global__ __launch_bounds__(256) void conv_igemm_patch_kernel(IgemmArgs a) {
  constexpr int NT = 256;
  constexpr bool PHASE = (MODE & 4) != 0;
  constexpr int NTAPS = PHASE ? 4 : 9, KS = PHASE ? 2 : 3;
  constexpr int WN = BN == 128 ? 2 : 1, WM = 4 / WN, TN = BN / (32 * WN), TM = 4 / WM;   // wave tile TM x TN MFMA tiles
  constexpr int CW = BN * 8 / NT;                       // weight chunks per thread per step
  static_assert(BN == 128 || BN == 32, "cout tile");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sP = reinterpret_cast<bf16*>(smem);             // [10][HROWP]  (single buffer; pixel stride LROW inside a row)
  bf16* sW = sP + 10 * HROWP;                           // [2][BN][LROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_m = wave % WM, wave_n = wave / WM;
  const int r = lane & 31, h = lane >> 5;

  const int nwg = a.tiles_m * a.tiles_n;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % a.tiles_n;
  int tile_m = lid / a.tiles_n;
  int phase = 0;
  if constexpr (PHASE) {
    // phase fastest: the four phases of a pixel tile get consecutive logical ids = the same XCD, back to back -- they read the
    // same input tile from that L2 and their interleaved output pixels (each phase writes every other pixel of a row) meet
    // in it before they reach HBM.  Phase-slowest put them on four different XCDs: 1.7x the algorithmic write bytes.
    if (a.phase_inner) { phase = tile_m & 3; tile_m >>= 2; }
    else { phase = tile_m / a.tiles_pp; tile_m -= phase * a.tiles_pp; }
  }
  const int pad_h = PHASE ? 1 - (phase >> 1) : 1, pad_w = PHASE ? 1 - (phase & 1) : 1;
  const int pw = a.W >> 4, ph = a.H >> 3;               // patches per row / column
  const int n = tile_m / (pw * ph), pr = tile_m - n * pw * ph;
  const int py0 = (pr / pw) << 3, px0 = (pr % pw) << 4;

  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * a.H * a.W * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w) + (PHASE ? (long)phase * a.CoutPad * a.Kpad : 0L), 0,
                                                                        a.CoutPad * a.Kpad * 2, 0x00020000);

  // halo chunks of this thread: 1440 16-byte chunks over 256 threads = 6 slots (the last one partial)
  int h_off[6], h_lds[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int q = tid + NT * j;
    const int hp = q >> 3, cc = q & 7;
    const int iy = py0 - 1 + hp / 18, ix = px0 - 1 + hp % 18;
    const bool ok = q < PHALO * 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    h_off[j] = ok ? (((n * a.H + iy) * a.W + ix) * a.Cin + cc * 8) * 2 : OOB;
    h_lds[j] = (q < PHALO * 8) ? (hp / 18) * HROWP + (hp % 18) * LROW + cc * 8 : -1;
  }
  int w_off[CW];
#pragma unroll
  for (int j = 0; j < CW; j++) {
    const int q = tid + NT * j;
    w_off[j] = ((tile_n * BN + (q >> 3)) * a.Kpad + (q & 7) * 8) * 2;
  }

  u32x4 rH[6], rW[CW];
  const int nchunks = a.Cin >> 6;
  const int last = nchunks * NTAPS - 1;
  int wcur = 0, wtap = 0, wc0 = 0;      // weight cursor (chunk outer, tap inner): k offset = tap*Cin + c0
  auto load_w = [&]() {
    const int wk = (wtap * a.Cin + wc0) * 2;
#pragma unroll
    for (int j = 0; j < CW; j++) rW[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, w_off[j], wk, 0);
    if (wcur < last) {
      wcur++;
      if (++wtap == NTAPS) { wtap = 0; wc0 += 64; }
    }
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int j = 0; j < CW; j++) {
      const int q = tid + NT * j;
      *reinterpret_cast<u32x4*>(sW + (buf * BN + (q >> 3)) * LROW + (q & 7) * 8) = rW[j];
    }
  };
  // MODE bit 3: y = relu((x - mean) * invstd * gamma[label] + beta[label]) applied to every in-image element while it is
  // staged (the conditional batch norm + relu in front of this conv, in the forward kernel's own expression and order, so the
  // operand is bit-identical to the tensor the unfused path would have stored); padding stays zero.  A thread's chunks all
  // carry the same 8 channels of a 64-channel chunk (cc = tid & 7): 4 x 8 parameters per chunk, reloaded per chunk.
  constexpr bool NORM = (MODE & 8) != 0;
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
  auto load_halo = [&](int c) {
#pragma unroll
    for (int j = 0; j < 6; j++)
      rH[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, h_off[j] == OOB ? OOB : h_off[j] + c * 128, 0, 0);
    if constexpr (NORM) {
      nm[0] = *reinterpret_cast<const f32x4*>(np_mu + c * 64); nm[1] = *reinterpret_cast<const f32x4*>(np_mu + c * 64 + 4);
      nm[2] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64); nm[3] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64 + 4);
      nm[4] = *reinterpret_cast<const f32x4*>(np_ga + c * 64); nm[5] = *reinterpret_cast<const f32x4*>(np_ga + c * 64 + 4);
      nm[6] = *reinterpret_cast<const f32x4*>(np_be + c * 64); nm[7] = *reinterpret_cast<const f32x4*>(np_be + c * 64 + 4);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (h_lds[j] >= 0) {
        u32x4 v = rH[j];
        if constexpr (NORM) {
          if (h_off[j] != OOB) {
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
        } else if constexpr ((MODE & 1) != 0) {
          v = relu_bf16x8(v);
        }
        *reinterpret_cast<u32x4*>(sP + h_lds[j]) = v;
      }
    }
  };

  f32x16 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; i++)
#pragma unroll
    for (int j = 0; j < TM; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // B-fragment base of this lane inside the halo image: pixel (wave_m*4 + 2j + (r>>4), r&15), centre tap
  int pb[TM];
#pragma unroll
  for (int j = 0; j < TM; j++) pb[j] = (wave_m * 2 * TM + 2 * j + (r >> 4) + 1) * HROWP + ((r & 15) + 1) * LROW + h * 8;
  const int ab = (wave_n * TN * 32 + r) * LROW + h * 8;

  load_halo(0);
  load_w();                 // step 0
  store_halo();
  store_w(0);
  __syncthreads();

  int s = 0;
  for (int c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks) load_halo(c + 1);           // in flight during the 9 tap-steps of this chunk
    int dh = -pad_h, dw = -pad_w;
#pragma unroll 1
    for (int tap = 0; tap < NTAPS; tap++, s++) {
      const int buf = s & 1;
      if (s < last) load_w();                        // weights of step s+1
      const int toff = dh * HROWP + dw * LROW;
      const bf16* pW = sW + buf * BN * LROW + ab;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        bf16x8 fa[TN], fb[TM];
#pragma unroll
        for (int i = 0; i < TN; i++) fa[i] = *reinterpret_cast<const bf16x8*>(pW + i * 32 * LROW + kk * 16);
#pragma unroll
        for (int j = 0; j < TM; j++) fb[j] = *reinterpret_cast<const bf16x8*>(sP + pb[j] + toff + kk * 16);
#pragma unroll
        for (int i = 0; i < TN; i++)
#pragma unroll
          for (int j = 0; j < TM; j++)
            acc[i][j] = GANK_MFMA32(fa[i], fb[j], acc[i][j]);
      }
      if (s < last) store_w(buf ^ 1);
      __syncthreads();
      if (++dw > KS - 1 - pad_w) { dw = -pad_w; dh++; }
    }
    if (c + 1 < nchunks) {                           // every wave is past its last read of the patch (barrier)
      store_halo();
      __syncthreads();
    }
  }

  // epilogue (same order as the generic kernel): lane holds channels co0+8g+4h..+3 of one pixel per quad
  const bool otanh = (a.flags & GANK_OUT_TANH) != 0;
  const bool vec = (a.Cout & 3) == 0;
  const bool wide = (a.Cout & 31) == 0;         // whole 32-channel tiles: 16-byte pieces (epi_tile_wide)
#pragma unroll
  for (int j = 0; j < TM; j++) {
    const int py = wave_m * 2 * TM + 2 * j + (r >> 4), px = r & 15;
    const long m = PHASE ? ((long)(n * 2 * a.H + 2 * (py0 + py) + (phase >> 1))) * (2 * a.W) + 2 * (px0 + px) + (phase & 1)
                         : ((long)(n * a.H + py0 + py)) * a.W + px0 + px;
    const long mr = (!PHASE && (a.flags & IG_RES_UP2X)) ? ((long)(n * (a.H >> 1) + ((py0 + py) >> 1))) * (a.W >> 1) + ((px0 + px) >> 1) : m;
    if (wide) {
#pragma unroll
      for (int i = 0; i < TN; i++) {
        const int ct = tile_n * BN + (wave_n * TN + i) * 32;
        epi_tile_wide<false>(acc[i][j], a.scale, a.bias ? a.bias + ct : nullptr, a.mask ? a.mask + m * a.Cout + ct : nullptr,
                             a.res ? a.res + mr * a.Cout + ct : nullptr, a.y + m * a.Cout + ct, h, otanh, nullptr);
      }
      continue;
    }
#pragma unroll
    for (int i = 0; i < TN; i++) {
      const int co0 = tile_n * BN + (wave_n * TN + i) * 32 + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; g++) {
        const int co = co0 + 8 * g;
        if (co >= a.Cout) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) v[e] = acc[i][j][4 * g + e] * a.scale;
        const long o = m * a.Cout + co;
        if (!vec) {          // narrow outputs (Cout = 3): element-wise tail
#pragma unroll
          for (int e = 0; e < 4; e++) {
            if (co + e >= a.Cout) break;
            float t = v[e];
            if (a.bias) t += a.bias[co + e];
            if (a.mask) t = (bf2f(a.mask[o + e]) > 0.f) ? t : 0.f;
            if (a.res) t += bf2f(a.res[mr * a.Cout + co + e]);
            a.y[o + e] = f2bf(otanh ? tanhf(t) : t);
          }
          continue;
        }
        if (a.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += b[e];
        }
        if (a.mask) {
          const bf16x4 mk = *reinterpret_cast<const bf16x4*>(a.mask + o);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = (bf2f(mk[e]) > 0.f) ? v[e] : 0.f;
        }
        if (a.res) {
          const bf16x4 rs = *reinterpret_cast<const bf16x4*>(a.res + mr * a.Cout + co);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += bf2f(rs[e]);
        }
        bf16x4 out;
        if (otanh) {
#pragma unroll
          for (int e = 0; e < 4; e++) out[e] = f2bf(tanhf(v[e]));
        } else {
#pragma unroll
          for (int e = 0; e < 4; e++) out[e] = f2bf(v[e]);
        }
        *reinterpret_cast<bf16x4*>(a.y + o) = out;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / SAME convolution with at most 4 output channels: the generator's image layer G.Output (256 -> 3 + tanh,
// gan_cifar_resnet.py:260-261).  On the 32-row instance of the patch kernel above it ran 36 tap-steps per block, each with a
// weight tile through LDS and a barrier for FOUR MFMAs per wave (87 us for the 320-sample pass against a 35 us HBM floor:
// latency, not arithmetic).  The whole filter is 4 rows x 9 x Cin: it is loaded into LDS ONCE per block ([4][Kpad + 8], row 3
// zero), the 10 x 18 halo of a 64-channel chunk is staged as before (next chunk's requests in flight during the taps), and a
// chunk's 36 MFMA steps run with no barrier in between.  MODE bits as the patch kernel: 0 relu on the input, 3 conditional batch
// norm + relu applied while staging.
// ------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void conv3x3_few_kernel(IgemmArgs a) {
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sP = reinterpret_cast<bf16*>(smem);             // [10][HROWP]
  bf16* sW = sP + 10 * HROWP;                           // [4][Kpad + 8]
  const int wrow = a.Kpad + 8;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int lid = xcd_remap(blockIdx.x, a.tiles_m);
  const int pw = a.W >> 4, ph = a.H >> 3;
  const int n = lid / (pw * ph), pr = lid - n * pw * ph;
  const int py0 = (pr / pw) << 3, px0 = (pr % pw) << 4;

  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * a.H * a.W * a.Cin * 2, 0x00020000);

  // the filter: rows 0 .. Cout-1 of the [CoutPad][Kpad] operand, row 3 (and any row >= Cout) zero
  for (int q = tid; q < 4 * (a.Kpad >> 3); q += NT) {
    const int row = q / (a.Kpad >> 3), c8 = q - row * (a.Kpad >> 3);
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row < a.Cout && row < 3) v = *reinterpret_cast<const u32x4*>(a.w + (long)row * a.Kpad + c8 * 8);
    *reinterpret_cast<u32x4*>(sW + row * wrow + c8 * 8) = v;
  }

  int h_off[6], h_lds[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int q = tid + NT * j;
    const int hp = q >> 3, cc = q & 7;
    const int iy = py0 - 1 + hp / 18, ix = px0 - 1 + hp % 18;
    const bool ok = q < PHALO * 8 && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    h_off[j] = ok ? (((n * a.H + iy) * a.W + ix) * a.Cin + cc * 8) * 2 : OOB;
    h_lds[j] = (q < PHALO * 8) ? (hp / 18) * HROWP + (hp % 18) * LROW + cc * 8 : -1;
  }
  u32x4 rH[6];
  constexpr bool NORM = (MODE & 8) != 0;
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
  auto load_halo = [&](int c) {
#pragma unroll
    for (int j = 0; j < 6; j++)
      rH[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, h_off[j] == OOB ? OOB : h_off[j] + c * 128, 0, 0);
    if constexpr (NORM) {
      nm[0] = *reinterpret_cast<const f32x4*>(np_mu + c * 64); nm[1] = *reinterpret_cast<const f32x4*>(np_mu + c * 64 + 4);
      nm[2] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64); nm[3] = *reinterpret_cast<const f32x4*>(np_mu + a.Cin + c * 64 + 4);
      nm[4] = *reinterpret_cast<const f32x4*>(np_ga + c * 64); nm[5] = *reinterpret_cast<const f32x4*>(np_ga + c * 64 + 4);
      nm[6] = *reinterpret_cast<const f32x4*>(np_be + c * 64); nm[7] = *reinterpret_cast<const f32x4*>(np_be + c * 64 + 4);
    }
  };
  auto store_halo = [&]() {           // the patch kernel's staging, expression for expression (MODE bit 3: the forward CBN kernel's)
#pragma unroll
    for (int j = 0; j < 6; j++) {
      if (h_lds[j] >= 0) {
        u32x4 v = rH[j];
        if constexpr (NORM) {
          if (h_off[j] != OOB) {
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
        } else if constexpr ((MODE & 1) != 0) {
          v = relu_bf16x8(v);
        }
        *reinterpret_cast<u32x4*>(sP + h_lds[j]) = v;
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  // B fragment of this lane: pixel (2 * wave + (r >> 4), r & 15) of the patch, centre tap; A fragment: filter row min(r, 3)
  const int pb = (wave * 2 + (r >> 4) + 1) * HROWP + ((r & 15) + 1) * LROW + h * 8;
  const int ab = (r < 3 ? r : 3) * wrow + h * 8;

  const int nchunks = a.Cin >> 6;
  load_halo(0);
  store_halo();
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < nchunks; c++) {
    if (c + 1 < nchunks) load_halo(c + 1);           // in flight during the 36 steps of this chunk
    const bf16* pW = sW + ab + c * 64;
#pragma unroll
    for (int tap = 0; tap < 9; tap++) {
      const int toff = (tap / 3 - 1) * HROWP + (tap % 3 - 1) * LROW;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        const bf16x8 fa = *reinterpret_cast<const bf16x8*>(pW + tap * a.Cin + kk * 16);
        const bf16x8 fb = *reinterpret_cast<const bf16x8*>(sP + pb + toff + kk * 16);
        acc = GANK_MFMA32(fa, fb, acc);
      }
    }
    if (c + 1 < nchunks) {
      __syncthreads();                               // every wave is past its last read of the patch
      store_halo();
      __syncthreads();
    }
  }

  // epilogue: rows 0 .. Cout-1 of the tile sit in the first accumulator quad of the h = 0 lanes
  if (h == 0) {
    const bool otanh = (a.flags & GANK_OUT_TANH) != 0;
    const int py = wave * 2 + (r >> 4), px = r & 15;
    const long m = ((long)(n * a.H + py0 + py)) * a.W + px0 + px;
#pragma unroll
    for (int e = 0; e < 3; e++) {
      if (e >= a.Cout) break;
      float t = acc[e] * a.scale;
      if (a.bias) t += a.bias[e];
      if (a.mask) t = (bf2f(a.mask[m * a.Cout + e]) > 0.f) ? t : 0.f;
      if (a.res) t += bf2f(a.res[m * a.Cout + e]);
      a.y[m * a.Cout + e] = f2bf(otanh ? tanhf(t) : t);
    }
  }
}

template <int MODE>
static int launch_few(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  a.tiles_m = a.N * (a.H / 8) * (a.W / 16);
  a.tiles_n = 1;
  const size_t lds = ((size_t)10 * HROWP + (size_t)4 * (a.Kpad + 8)) * sizeof(bf16);
  auto kern = conv3x3_few_kernel<MODE>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv3x3_few");
  static const std::string tag = gank_format("conv3x3_few_kernel<%d>", MODE);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  hipLaunchKernelGGL(kern, dim3(a.tiles_m), dim3(256), lds, s, a);
  GANK_LAUNCH_OK("conv3x3_few");
  return 0;
}
static bool few_ok(const IgemmArgs& a) {
  static const int env = gank_tune("GANK_IGEMM_FEW", 1);   // experiment knob: 0 keeps the 32-row patch kernel
  // (channel-inner K order only: the kernel walks [tap][channel]; it accumulates no statistics -- `tl_stats_done` stays 0 and the
  //  caller's batch-norm pass computes them, as after the 32-row patch kernel it replaces)
  return env && a.korder == 0 && a.Cout <= 3 && a.CoutPad == 32 && a.Cin % 64 == 0 && a.Kpad == 9 * a.Cin && !(a.flags & IG_RES_UP2X) &&
         (size_t)(10 * HROWP + 4 * (a.Kpad + 8)) * sizeof(bf16) <= 160 * 1024;
}

template <int MODE>    // MODE has bit 2 set
static int launch_patch_phase(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  a.tiles_pp = a.N * (a.H / 8) * (a.W / 16);
  a.phase_inner = phase_inner_env();
  a.tiles_m = 4 * a.tiles_pp;
  a.tiles_n = a.CoutPad / 128;
  const size_t lds = ((size_t)10 * HROWP + (size_t)2 * 128 * LROW) * sizeof(bf16);
  auto kern = conv_igemm_patch_kernel<MODE, 128>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm_patch");
  static const std::string tag = gank_format("conv_igemm_patch_kernel<%d, 128>", MODE);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
  GANK_LAUNCH_OK("conv_igemm_patch_phase");
  return 0;
}
static bool patch_phase_ok(const IgemmArgs& a) {   // a.H, a.W = low-res grid; a.Cin % 64 == 0 is checked by the callers
  static const int env = gank_tune("GANK_IGEMM_PATCH_PHASE", 1);   // GANK_IGEMM_PATCH_PHASE=0 keeps the per-tap phase kernel
  return env && a.W % 16 == 0 && a.H % 8 == 0 && a.CoutPad % 128 == 0 && a.Cout % 4 == 0 && a.Cin % 64 == 0;
}

// ------------------------------------------------------------------------------------------------------
// Narrow-input convolutions: Cin <= 4 with taps*Cin <= 32 -- the image side of the critic (D.Block.1.Conv1 3->128
// 3x3, D.Block.1.Shortcut 3->128 1x1, gan_cifar_resnet.py:212-234) and the input gradient of G.Output (3 <- 256).
// The whole reduction is ONE 32-deep MFMA K-step, so there is nothing to pipeline and nothing to share: no LDS,
// no barriers.  A wave owns 32 pixels x 128 couts; each lane gathers its own 16 im2col values (2-byte loads from a
// tensor that lives in L1/L2: 6 KB per image) and reads its weight fragments straight from the row-major operand.
// The kernel is bound by writing the output (33.5 MB for D.Block.1.Conv1 at N=128); the K-packed generic path spent
// 45-55 us per call in per-element index arithmetic for it, this one a few microseconds above the write time.
// ------------------------------------------------------------------------------------------------------
// POOL (KS = 1): the input is stored at twice the resolution, [N,2H,2W,CIN], and each gathered value is the 2x2 mean of
// gan_cifar_resnet.py:129-130 (MeanPoolConv: pool, THEN the 1x1 conv) rounded to the element type -- the arithmetic of
// pool2x2_kernel<1> + this kernel, without the launch in between; the pooled tensor (which the filter gradient needs) is a
// side output of the lanes that gathered it.
constexpr int NARROW_LDS = 4 * 32 * 272;      // the four waves' private transpose regions
template <int KS, int CIN, bool POOL>
__device__ __forceinline__ void narrow_in_body(const IgemmArgs& a, const int block, char* s_t) {
  constexpr int TAPS = KS * KS, KTOT = TAPS * CIN, PAD = (KS - 1) / 2;
  static_assert(KTOT <= 32, "one 32-deep K-step");
  static_assert(!POOL || KS == 1, "pooled gather: 1x1 only");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int tile_n = block % a.tiles_n, tile_m = block / a.tiles_n;
  const int m = tile_m * 128 + wave * 32 + r;
  const bool inrelu = (a.flags & GANK_IN_RELU) != 0;

  int n = 0, oh = 0, ow = 0;
  if (m < a.M) pix_decomp(m, a.H, a.W, a.shw, a.sw, n, oh, ow);
  // B fragments: k = kk*16 + 8h + j  ->  (tap, c) = (k / CIN, k % CIN)
  bf16x8 fb[2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int k = kk * 16 + 8 * h + j;            // h is runtime: (tap, c) need real arithmetic, CIN / KS are constants
      const int tap = k / CIN, c = k - tap * CIN;
      const int ih = oh + tap / KS - PAD, iw = ow + tap % KS - PAD;
      const bool ok = m < a.M && k < KTOT && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
      // UNCONDITIONAL load from a clamped address, then select: a branch around each load serialises the 16 gathers
      // (one L2 round trip each: this was 20 of the kernel's 23 us)
      float val;
      if constexpr (POOL) {
        const int rs = 2 * a.W * CIN;
        const int idx = ok ? ((n * 2 * a.H + 2 * ih) * 2 * a.W + 2 * iw) * CIN + c : 0;
        val = bf2f(f2bf((bf2f(a.x[idx]) + bf2f(a.x[idx + rs]) + bf2f(a.x[idx + CIN]) + bf2f(a.x[idx + rs + CIN])) * 0.25f));   // tf.add_n order (:129-130)
      } else {
        const int idx = ok ? ((n * a.H + ih) * a.W + iw) * CIN + c : 0;
        val = bf2f(a.x[idx]);
      }
      val = ok ? val : 0.f;
      if (inrelu) val = fmaxf(val, 0.f);
      fb[kk][j] = f2bf(val);
    }
  }
  if constexpr (POOL) {
    if (a.aux_out && tile_n == 0 && h == 0 && m < a.M) {
#pragma unroll
      for (int j = 0; j < CIN; j++) a.aux_out[(long)m * CIN + j] = fb[0][j];
    }
  }
  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[i][e] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const bf16* wr = a.w + (long)(tile_n * 128 + i * 32 + r) * a.Kpad + h * 8;     // rows < CoutPad (CoutPad % 128 == 0)
#pragma unroll
    for (int kk = 0; kk < 2; kk++) {
      const bf16x8 fa = *reinterpret_cast<const bf16x8*>(wr + kk * 16);
      acc[i] = GANK_MFMA32(fa, fb[kk], acc[i]);
    }
  }
  // Epilogue.  Whole 32-channel tiles leave in 16-byte pieces, a pixel's 256-byte run in eight back-to-back stores that
  // merge in the L2 (epi_tile_wide); an earlier LDS transpose of the wave's tile cost two barriers and 2-way bank conflicts
  // on both sides (SQ_LDS_BANK_CONFLICT = 50 % of the LDS cycles) in a kernel that is one latency chain per block.
  const bool otanh = (a.flags & GANK_OUT_TANH) != 0;
  if (a.korder == 0 && !a.mask && !a.res && !otanh && (a.Cout & 127) == 0) {
    // Wave-private transpose: the wave's 32 pixels x 128 channels (8 KB) go through its OWN LDS region, so nothing but the
    // wave's LDS counter orders the two sides (no s_barrier: the block-wide transpose this kernel once had cost two), and a store
    // instruction then covers FOUR whole 256-byte pixel runs instead of a 32-byte piece of 32 different ones.  Pitch 272 B (16 B off
    // a multiple of 256 B; measured SQ_LDS_BANK_CONFLICT: 25 % of this kernel's LDS cycles, which are few).
    char* mine = s_t + wave * 32 * 272;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int ct = tile_n * 128 + i * 32;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        float v[8];
        acc_widen(acc[i], q, a.scale, v);
        bf16x8 o;
        if (a.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + ct + 16 * q + 8 * h), b1 = *reinterpret_cast<const f32x4*>(a.bias + ct + 16 * q + 8 * h + 4);
#pragma unroll
          for (int e = 0; e < 4; e++) { v[e] += b0[e]; v[4 + e] += b1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x8*>(mine + r * 272 + (i * 32 + 16 * q + 8 * h) * 2) = o;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the wave's own writes: program order + this wait, no barrier
    const int m_w = tile_m * 128 + wave * 32;                // first pixel of the wave
#pragma unroll
    for (int it = 0; it < 8; it++) {
      const int p = it * 4 + (lane >> 4), c16 = lane & 15;
      const u32x4 v = *reinterpret_cast<const u32x4*>(mine + p * 272 + c16 * 16);
      if (m_w + p < a.M) *reinterpret_cast<u32x4*>(a.y + (long)(m_w + p) * a.Cout + tile_n * 128 + c16 * 8) = v;
    }
    return;
  }
  if ((a.Cout & 31) == 0) {
    if (m < a.M) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int ct = tile_n * 128 + i * 32;
        if (ct < a.Cout)
          epi_tile_wide<false>(acc[i], a.scale, a.bias ? a.bias + ct : nullptr, a.mask ? a.mask + (long)m * a.Cout + ct : nullptr,
                               a.res ? a.res + (long)m * a.Cout + ct : nullptr, a.y + (long)m * a.Cout + ct, h, otanh, nullptr);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int co0 = tile_n * 128 + i * 32 + 4 * h;
#pragma unroll
    for (int g = 0; g < 4; g++) {
      const int co = co0 + 8 * g;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; e++) v[e] = acc[i][4 * g + e] * a.scale;
      if (m < a.M && co < a.Cout) {
        const long o = (long)m * a.Cout + co;
        if (a.bias) {
          const f32x4 b = *reinterpret_cast<const f32x4*>(a.bias + co);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += b[e];
        }
        if (a.mask) {
          const bf16x4 mk = *reinterpret_cast<const bf16x4*>(a.mask + o);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] = (bf2f(mk[e]) > 0.f) ? v[e] : 0.f;
        }
        if (a.res) {
          const bf16x4 rs = *reinterpret_cast<const bf16x4*>(a.res + o);
#pragma unroll
          for (int e = 0; e < 4; e++) v[e] += bf2f(rs[e]);
        }
        bf16x4 out;
#pragma unroll
        for (int e = 0; e < 4; e++) out[e] = f2bf(otanh ? tanhf(v[e]) : v[e]);
        *reinterpret_cast<bf16x4*>(a.y + o) = out;
      }
    }
  }
}

template <int KS, int CIN, bool POOL = false>
__global__ __launch_bounds__(256) void conv_narrow_in_kernel(IgemmArgs a) {
  __shared__ __attribute__((aligned(16))) char s_t[NARROW_LDS];
  narrow_in_body<KS, CIN, POOL>(a, blockIdx.x, s_t);
}
// the two image-side layers of the critic's first block (gan_cifar_resnet.py:212-234: conv_1 3x3 on the image, Shortcut = 1x1 on its 2x2
// mean) read the same 3-channel image: one launch, the shortcut's workgroups behind conv_1's (no wave takes both paths)
__global__ __launch_bounds__(256) void conv_narrow_in_pair_kernel(IgemmArgs a, IgemmArgs b, int blocks_a) {
  __shared__ __attribute__((aligned(16))) char s_t[NARROW_LDS];
  if ((int)blockIdx.x < blocks_a) narrow_in_body<3, 3, false>(a, blockIdx.x, s_t);
  else narrow_in_body<1, 3, true>(b, blockIdx.x - blocks_a, s_t);
}

template <int KS, int CIN, bool POOL = false>
static int launch_narrow_in(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  a.tiles_m = cdiv(a.M, 128);
  a.tiles_n = a.CoutPad / 128;
  static const std::string tag = gank_format("conv_narrow_in_kernel<%d, %d%s>", KS, CIN, POOL ? ", pool" : "");     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  hipLaunchKernelGGL((conv_narrow_in_kernel<KS, CIN, POOL>), dim3(a.tiles_m * a.tiles_n), dim3(256), 0, s, a);
  GANK_LAUNCH_OK("conv_narrow_in");
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Two-group ("ping-pong") patch kernel for 3x3 / stride 1 / SAME convolutions and for the four output phases of the
// stride-2 transposed convolutions (UpsampleConv 3x3 fprop, ConvMeanPool 3x3 input gradient), Cin % 32 == 0,
// Cout % 256 == 0, on 8 x 32 (PW = 32) or 16 x 16 (PW = 16) pixel patches.
// Every kernel above sits at ~0.9 PFLOP/s: a 128-wide tile, 2-3 blocks per CU, one __syncthreads (with its vmcnt(0)
// drain) per K-step.  This one has the structure that gets past that on a GEMM:
//   * one 512-thread block per CU owns 256 pixels x 256 couts; wave (wm, wn) = 128 pixels x 64 couts = 8 x 4 tiles of
//     v_mfma_f32_16x16x32 (GANK_PP_M16, default; 4 x 2 of 32x32x16 otherwise), 128 accumulator registers, 12 fragment reads
//     (ds_read_b128) per K-step;
//   * a phase is one K-step of 32 channels of one tap: {12 fragment reads R, LDS-DMA issues, counted vmcnt, 32 (16) MFMAs M}
//     with ONE s_barrier, which group 0 (waves 0-3) takes between R and M and group 1 (waves 4-7: the other wave of
//     each SIMD) at the top of the phase.  After barrier p group 0 therefore runs {M_p, R_p+1} while group 1 runs
//     {R_p, M_p}: each SIMD's matrix pipe has one group's MFMAs beside the other group's reads and staging;
//   * both operands arrive by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPRs; out-of-range lanes deliver the
//     zero padding).  Weights: a ring of 6 slots (16 KB = 256 couts x 32 k), every wave issues its 2 pieces of K-step
//     p+4 in phase p.  Pixels: the halo of the next 32-channel chunk goes into the other of two halo images, issued by
//     group 1 only, in the first two phases of the current chunk and AHEAD of those phases' weight pieces.
//     Nothing in the loop waits for vmcnt(0): a phase's wait leaves the weight pieces of that phase and the one before
//     in flight.  What makes this safe (q = a K-step, interval j = between barriers j and j+1; group 0 reads R_q in
//     interval q-1, group 1 in interval q): a piece of step q is retired by group 0's wait in phase q-1 and group 1's
//     in phase q-2, both ahead of barrier q-1; a slot is rewritten in phase q+2 at the earliest, two barriers after its
//     last read; the halo image of chunk c+1 is rewritten after barrier (first phase of chunk c), one barrier after
//     group 1's last read of it, and is retired by group 1's wait in the chunk's last-but-one phase (hence "ahead of
//     the weight pieces": vmcnt retires in issue order);
//   * LDS images are lane-linear per DMA (1 KB = 16 pixels or couts x 64 B); the 16-byte chunk inside a 64-B row is
//     XOR-swizzled with (row >> 2) & 3 on the SOURCE address and on the read: conflict-free ds_read_b128 groups for
//     the weights and for the pixel tiles (SQ_LDS_BANK_CONFLICT = 0; the two-row tiles of PW = 16 swizzle with hx >> 1);
//   * the epilogue widens its stores with v_permlane32_swap (8 consecutive channels = 16 B per lane): with one block
//     per CU nothing else overlaps them.
// Measured (256 -> 256, 32 x 32, n = 128, random data): 1.10-1.14 PFLOP/s against 0.81-0.86 for conv_igemm_patch_kernel.
// The matrix pipe is then 61-67 % busy at the clock the chip holds under this load (1.5-1.8 GHz; MFMAs alone: 2.3).
// ------------------------------------------------------------------------------------------------------
constexpr int PP_HALO_BYTES = 24576;               // 10 x 34 (or 18 x 18) px x 64 B <= 24 DMA pieces of 1 KB
constexpr int PP_WSLOT_BYTES = 16384;              // 256 couts x 32 k x 2 B = 16 pieces
constexpr int PP_NSLOT = 6, PP_PD = 4;             // ring slots, prefetch distance in K-steps
constexpr int PP_WRING = 2 * PP_HALO_BYTES;
constexpr int PP_LDS_BYTES = PP_WRING + PP_NSLOT * PP_WSLOT_BYTES;   // 144 KB

template <int N>
__device__ __forceinline__ void pp_wait_vmcnt() {
  static_assert(N == 0 || N == 4 || N == 6 || N == 7, "vmcnt value not instantiated");
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
}

// Matrix instruction of the two-group kernel.  1 (default): v_mfma_f32_16x16x32 -- the same multiply-adds in twice as many
// instructions of half the size, the same 12 fragment reads per K-step.  On this power-bound kernel it runs 6-10 % faster than
// the 32x32x16 form (timing experiment at equal memory traffic: 145 -> 130 us at n = 128, 312 -> 288 us at n = 320; the guide
// reports 1.12-1.15x for bare loops).  0: the 32x32x16 form (kept for A/B runs: -DGANK_PP_M16=0).
#ifndef GANK_PP_M16
#define GANK_PP_M16 1
#endif
#ifdef GANK_ACT_F16
#define GANK_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
#else
#define GANK_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif
template <int MODE, int PW, bool STATS = false>   // MODE bit0: relu on the input operand; bit2: one output phase of a stride-2 transposed conv
__global__ __launch_bounds__(512) void conv_igemm_pp_kernel(IgemmArgs a) {
  constexpr bool PHASE = (MODE & 4) != 0;
  constexpr int NT = PHASE ? 4 : 9;                  // taps (K-steps) per 32-channel chunk
  constexpr int PHH = 256 / PW;                      // patch rows
  constexpr int HW_ = PW + 2, HROW = HW_ * 64;       // halo width (pixels), halo row pitch (bytes)
  constexpr int HPX = (PHH + 2) * HW_;               // halo pixels
  constexpr int RS = 32 / PW;                        // patch rows per 32-pixel MFMA tile
  // halo swizzle: 16-byte chunk c of halo pixel (hy, hx) sits in slot c ^ ((hx >> HSW) & 3).  ds_read_b128 is served in four
  // fixed 16-lane groups ({0-3,12-15,20-27}, ...); with the 18-pixel rows of PW = 16 (pitch = 8 slots mod 16) a group mixes
  // two patch rows and (hx >> 2) left two lanes of most groups on one slot: 6.7 LDS cycles per read instead of 4 by the bank
  // rule, SQ_LDS_BANK_CONFLICT = 30 % of SQ_LDS_IDX_ACTIVE.  (hx >> 1) is conflict-free there, (hx >> 2) for one-row tiles.
  constexpr int HSW = PW == 16 ? 1 : 2;
#if GANK_PP_M16
  // 16x16x32 fragments: lane l reads 16-byte chunk l >> 4 of row l & 15 (a cout, or a pixel of one 16-pixel run of a patch
  // row: no two-row tiles here).  chunk ^ ((row >> 1) & 2) puts the 16 lanes of every ds_read_b128 lane group on 16 different
  // slots for every start column of the run (brute force over the four groups and all shifts 0..18), weights included.
  auto pp_hswz = [](int hx) { return (hx >> 1) & 2; };
  auto pp_wswz = [](int co) { return (co >> 1) & 2; };
#else
  auto pp_hswz = [](int hx) { return (hx >> HSW) & 3; };
  auto pp_wswz = [](int co) { return (co >> 2) & 3; };
#endif
  static_assert(PW == 32 || PW == 16, "patch width");
  static_assert(HPX * 64 <= PP_HALO_BYTES, "halo image");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  typedef __attribute__((address_space(3))) void* lds_t;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int r = lane & 31, h = lane >> 5;
  const int prow = RS == 1 ? 0 : (r >> 4), pcol = RS == 1 ? r : (r & 15);   // this lane's pixel inside a tile

  const int nwg = a.tiles_m * a.tiles_n;
  const int lid = xcd_remap(blockIdx.x, nwg);
  const int tile_n = lid % a.tiles_n;
  int tile_m = lid / a.tiles_n;
  int phase = 0;
  if constexpr (PHASE) {
    // phase fastest: the four phases of a pixel tile get consecutive logical ids = the same XCD, back to back -- they read the
    // same input tile from that L2 and their interleaved output pixels (each phase writes every other pixel of a row) meet
    // in it before they reach HBM.  Phase-slowest put them on four different XCDs: 1.7x the algorithmic write bytes.
    if (a.phase_inner) { phase = tile_m & 3; tile_m >>= 2; }
    else { phase = tile_m / a.tiles_pp; tile_m -= phase * a.tiles_pp; }
  }
  const int pad_h = PHASE ? 1 - (phase >> 1) : 1, pad_w = PHASE ? 1 - (phase & 1) : 1;
  const int pw = a.W / PW, ph = a.H / PHH;
  const int n = tile_m / (pw * ph), pr = tile_m - n * pw * ph;
  const int py0 = (pr / pw) * PHH, px0 = (pr % pw) * PW;

  constexpr unsigned OOB = 0x7FFFFFF0u;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * a.H * a.W * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w) + (PHASE ? (long)phase * a.CoutPad * a.Kpad : 0L), 0,
                                                                        a.CoutPad * a.Kpad * 2, 0x00020000);

  // DMA source offsets.  Halo (group 1 only, 6 pieces per wave): 16-byte unit q of the image = pixel q>>2 (row-major
  // (PHH+2) x (PW+2), origin (py0-1, px0-1)), slot q&3 holding channel chunk (q&3) ^ ((hx>>HSW)&3).  Weights: unit q of a
  // slot = cout q>>2, slot q&3 holding k chunk (q&3) ^ ((co>>2)&3).
  unsigned h_off[6];
#pragma unroll
  for (int j = 0; j < 6; j++) {
    const int q = (j * 4 + wn) * 64 + lane;
    const int hp = q >> 2, hy = hp / HW_, hx = hp - hy * HW_;
    const int iy = py0 - 1 + hy, ix = px0 - 1 + hx;
    const bool ok = hp < HPX && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
    h_off[j] = ok ? (unsigned)((((n * a.H + iy) * a.W + ix) * a.Cin + (((q & 3) ^ pp_hswz(hx)) << 3)) * 2) : OOB;
  }
  int w_off[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int q = (j * 8 + wave) * 64 + lane;
    const int co = q >> 2;
    w_off[j] = ((tile_n * 256 + co) * a.Kpad + (((q & 3) ^ pp_wswz(co)) << 3)) * 2;
  }
  // fragment read offsets of this lane (bytes): weights row r of the wave's cout range; pixels: halo column pcol + d of
  // the wave's first patch row (+ the tap's row offset as an immediate), d = the tap's column offset 0..2
  constexpr int ND = PHASE ? 2 : 3;
#if GANK_PP_M16
  const int p16 = lane & 15, kc = lane >> 4;         // row / pixel inside a 16-wide tile, 16-byte K chunk
  constexpr int RPT = PW / 16;                       // 16-pixel tiles per patch row
  int a_lane, b_lane[ND];
  a_lane = PP_WRING + (wn * 64 + p16) * 64 + ((kc ^ pp_wswz(p16)) << 4);          // + i * 1024 for cout tile i (16 rows)
#pragma unroll
  for (int d = 0; d < ND; d++) {
    const int hx = p16 + d + (PHASE ? 1 - pad_w : 0);
    const int hy = wm * (PHH / 2) + (PHASE ? 1 - pad_h : 0);
    b_lane[d] = hy * HROW + hx * 64 + ((kc ^ pp_hswz(hx)) << 4);                   // + (j / RPT) * HROW + (j % RPT) * 1024 for pixel tile j
  }
#else
  int a_lane[2], b_lane[ND][2];
#pragma unroll
  for (int kk = 0; kk < 2; kk++) {
    a_lane[kk] = PP_WRING + (wn * 64 + r) * 64 + (((kk * 2 + h) ^ pp_wswz(r)) << 4);
#pragma unroll
    for (int d = 0; d < ND; d++) {
      const int hx = pcol + d + (PHASE ? 1 - pad_w : 0);
      const int hy = wm * (PHH / 2) + prow + (PHASE ? 1 - pad_h : 0);
      b_lane[d][kk] = hy * HROW + hx * 64 + (((kk * 2 + h) ^ pp_hswz(hx)) << 4);
    }
  }
#endif

  const int nch = a.Cin >> 5;
  auto issue_w = [&](int slot_bytes, int c, int tap) {      // K-step (chunk c, tap) -> ring slot
    const int koff = (tap * a.Cin + c * 32) * 2;
#pragma unroll
    for (int j = 0; j < 2; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_t)(smem + PP_WRING + slot_bytes + (j * 8 + wave) * 1024), 16, w_off[j], koff, 0, 0);
  };
  auto issue_h = [&](int buf, int c, int j0) {              // pieces j0..j0+2 of chunk c's halo -> image buf (group 1)
#pragma unroll
    for (int j = j0; j < j0 + 3; j++)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_t)(smem + buf * PP_HALO_BYTES + (j * 4 + wn) * 1024), 16, (int)(h_off[j] + (unsigned)c * 64u), 0, 0, 0);
  };

#if GANK_PP_M16
  f32x4 acc[4][8];                                   // [cout tile of 16][pixel tile of 16]
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#else
  f32x16 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;
#endif

  // prologue: halo of chunk 0, weights of K-steps 0..3 (issue order matters: the counted waits rely on it)
  if (wm == 1) { issue_h(0, 0, 0); issue_h(0, 0, 3); }
  {
    int c = 0, t = 0;
#pragma unroll
    for (int q = 0; q < PP_PD; q++) {
      issue_w(q * PP_WSLOT_BYTES, c, t);
      if (++t == NT) { t = 0; c++; }
      if (c >= nch) { c = nch - 1; t = NT - 1; }
    }
  }
  // Steps 0 AND 1 (and the halo) must have landed here: group 0 reads step 1 right after barrier 0, and group 1's first
  // in-loop wait comes after that barrier ("group 1's wait in phase q-2" is this one for q = 1).  Steps 2-3 stay in flight.
  // (A vmcnt(6) here passed every parity test and failed the run-to-run identity screen in 1-30 % of the launches.)
  pp_wait_vmcnt<4>();
  __builtin_amdgcn_s_barrier();

  int rslot = 0, wslot = PP_PD * PP_WSLOT_BYTES;
  for (int c = 0; c < nch; c++) {
    const int hb = (c & 1) * PP_HALO_BYTES;
    const int cn = (c + 1 < nch) ? c + 1 : nch - 1;
    auto phase_step = [&](auto tc) {
      constexpr int t = decltype(tc)::value;
      constexpr int dh1 = PHASE ? t / 2 : t / 3, dw1 = PHASE ? t % 2 : t % 3;
      if (wm == 1) __builtin_amdgcn_s_barrier();
#if GANK_PP_M16
      bf16x8 fa[4], fb[8];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = *reinterpret_cast<const bf16x8*>(smem + rslot + i * 1024 + a_lane);
#pragma unroll
      for (int j = 0; j < 8; j++) fb[j] = *reinterpret_cast<const bf16x8*>(smem + hb + (j / RPT + dh1) * HROW + (j % RPT) * 1024 + b_lane[dw1]);
#else
      bf16x8 fa[2][2], fb[4][2];
#pragma unroll
      for (int kk = 0; kk < 2; kk++) {
#pragma unroll
        for (int i = 0; i < 2; i++) fa[i][kk] = *reinterpret_cast<const bf16x8*>(smem + rslot + i * 2048 + a_lane[kk]);
#pragma unroll
        for (int j = 0; j < 4; j++) fb[j][kk] = *reinterpret_cast<const bf16x8*>(smem + hb + (j * RS + dh1) * HROW + b_lane[dw1][kk]);
      }
#endif
      // group 1: half of the next chunk's halo, ahead of the weight pieces (clamped past the end: a spare image)
      if constexpr (t < 2) { if (wm == 1) issue_h((c + 1) & 1, cn, 3 * t); }
      {                                                // K-step p + 4 (clamped to the last one: a spare write into a dead slot)
        constexpr int t4 = (t + PP_PD) % NT;
        int c4 = c + (t + PP_PD) / NT, tt = t4;
        if (c4 >= nch) { c4 = nch - 1; tt = NT - 1; }
        issue_w(wslot, c4, tt);
      }
      if constexpr (t < 2) { if (wm == 1) pp_wait_vmcnt<7>(); else pp_wait_vmcnt<4>(); }
      else pp_wait_vmcnt<4>();
      if (wm == 0) __builtin_amdgcn_s_barrier();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#if GANK_PP_M16
      if constexpr ((MODE & 1) != 0) {
#pragma unroll
        for (int j = 0; j < 8; j++) fb[j] = __builtin_bit_cast(bf16x8, relu_bf16x8(__builtin_bit_cast(u32x4, fb[j])));
      }
      __builtin_amdgcn_sched_barrier(0);               // keeps the MFMA cluster behind the barrier (hipcc hoists it otherwise)
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 8; j++)
          acc[i][j] = GANK_MFMA16(fa[i], fb[j], acc[i][j]);
#else
      if constexpr ((MODE & 1) != 0) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int kk = 0; kk < 2; kk++) fb[j][kk] = __builtin_bit_cast(bf16x8, relu_bf16x8(__builtin_bit_cast(u32x4, fb[j][kk])));
      }
      __builtin_amdgcn_sched_barrier(0);               // keeps the MFMA cluster behind the barrier (hipcc hoists it otherwise)
#pragma unroll
      for (int kk = 0; kk < 2; kk++)
#pragma unroll
        for (int i = 0; i < 2; i++)
#pragma unroll
          for (int j = 0; j < 4; j++)
            acc[i][j] = GANK_MFMA32(fa[i][kk], fb[j][kk], acc[i][j]);
#endif
      __builtin_amdgcn_sched_barrier(0);
      rslot += PP_WSLOT_BYTES; if (rslot == PP_NSLOT * PP_WSLOT_BYTES) rslot = 0;
      wslot += PP_WSLOT_BYTES; if (wslot == PP_NSLOT * PP_WSLOT_BYTES) wslot = 0;
    };
    phase_step(std::integral_constant<int, 0>{}); phase_step(std::integral_constant<int, 1>{});
    phase_step(std::integral_constant<int, 2>{}); phase_step(std::integral_constant<int, 3>{});
    if constexpr (!PHASE) {
      phase_step(std::integral_constant<int, 4>{}); phase_step(std::integral_constant<int, 5>{}); phase_step(std::integral_constant<int, 6>{});
      phase_step(std::integral_constant<int, 7>{}); phase_step(std::integral_constant<int, 8>{});
    }
  }
  pp_wait_vmcnt<0>();                                 // the clamped spare DMAs must land before the LDS is released

  // epilogue.  An accumulator quad g of lane (r, h) is channels 8g+4h..+3 of pixel r: 8-byte pieces.  v_permlane32_swap
  // trades quad 2q+1 of the h = 0 half-wave for quad 2q of the h = 1 half-wave, after which every lane owns 8
  // CONSECUTIVE channels (16q + 8h ..) and writes, and reads mask / residual, 16 bytes at a time.
  const bool otanh = (a.flags & GANK_OUT_TANH) != 0;
  // Pixel-outer loop order.  A wave owns 64 channels = ONE 128-byte line of each of its pixels, in four 32-byte pieces (i, q).
  // With (i, q) outside the pieces of a line were a whole pixel loop apart; the L2 -- turned over completely by the 33 MB
  // that the resident blocks write at the same time -- evicted the line in between and every piece went to the fabric as its
  // own request: WRITE_SIZE 165 MB for a 67-MB output.  Computed and stored back to back they merge (67.1 MB; 69.2 MB with
  // the statistics atomics) and the kernel is 5-9 % faster.
  // With STATS the epilogue also accumulates the batch-norm statistics of what it writes (normalization.py:47: tf.nn.moments
  // of the NEXT layer's input).  Sums are of (y - bias): the mean of a conv output is mostly its bias, and E[x^2] - E[x]^2
  // in fp32 wants a small mean.  A wave reduces its 32 pixels by shuffles; lanes r = 0 add to one of GANK_STAT_SLOTS copies
  // of the tower's sums (blocks of a tower spread over the copies: same-address float atomics serialise at the memory side).
  float keep1 = 0.f, keep2 = 0.f;
#if GANK_PP_M16
  // 16x16x32 accumulators: lane (p16, kc) of tile (i, j) holds couts 4 kc .. 4 kc + 3 of pixel p16 (8 bytes).  For a PAIR of
  // cout tiles (2P, 2P+1), v_permlane16_swap with vdst = tile 2P's register and src = tile 2P+1's trades the odd 16-lane rows
  // of the one for the even rows of the other: afterwards row kc owns 8 CONSECUTIVE couts 8 (kc >> 1) .. + 7 of tile
  // 2P + (kc & 1) -- 16 bytes -- and the two stores of a pixel (P = 0, 1: 64 bytes each) leave back to back.
  {
    float st1[STATS ? 2 : 1][8], st2[STATS ? 2 : 1][8];
    if constexpr (STATS) {
#pragma unroll
      for (int P = 0; P < 2; P++)
#pragma unroll
        for (int e = 0; e < 8; e++) { st1[P][e] = 0.f; st2[P][e] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const int py = py0 + wm * (PHH / 2) + j / RPT, px = px0 + (j % RPT) * 16 + p16;
      const long m = PHASE ? ((long)(n * 2 * a.H + 2 * py + (phase >> 1))) * (2 * a.W) + 2 * px + (phase & 1) : ((long)(n * a.H + py)) * a.W + px;
      const long mr = (!PHASE && (a.flags & IG_RES_UP2X)) ? ((long)(n * (a.H >> 1) + (py >> 1))) * (a.W >> 1) + (px >> 1) : m;
      bf16x8 outl[2];
#pragma unroll
      for (int P = 0; P < 2; P++) {
        const int co = tile_n * 256 + wn * 64 + (2 * P + (kc & 1)) * 16 + 8 * (kc >> 1);
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; e++) {
          // inline asm as for v_permlane32_swap (the builtin's second result was dropped by hipcc 7.2); s_nop 1 = the 2 wait
          // states a VALU write needs before the swap reads it
          float lo = acc[2 * P][j][e] * a.scale, hi = acc[2 * P + 1][j][e] * a.scale;
          asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
          v[e] = lo;
          v[4 + e] = hi;
        }
        const long o = m * a.Cout + co;
        // the statistics are of d = y - bias, carried beside y without ever adding the bias
        float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[8];
        if (a.bias) {
          const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
          for (int e = 0; e < 4; e++) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) { d[e] = v[e]; v[e] += bb[e]; }
        if (a.mask) {
          const bf16x8 mk = *reinterpret_cast<const bf16x8*>(a.mask + o);
#pragma unroll
          for (int e = 0; e < 8; e++) { const bool on = bf2f(mk[e]) > 0.f; v[e] = on ? v[e] : 0.f; d[e] = on ? d[e] : -bb[e]; }
        }
        if (a.res) {
          const bf16x8 rs = *reinterpret_cast<const bf16x8*>(a.res + mr * a.Cout + co);
#pragma unroll
          for (int e = 0; e < 8; e++) { const float t = bf2f(rs[e]); v[e] += t; d[e] += t; }
        }
        if constexpr (STATS) {
#pragma unroll
          for (int e = 0; e < 8; e++) { st1[P][e] += d[e]; st2[P][e] += d[e] * d[e]; }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) outl[P][e] = f2bf(otanh ? tanhf(v[e]) : v[e]);
      }
      bf16* line = a.y + m * a.Cout + tile_n * 256 + wn * 64 + (kc & 1) * 16 + 8 * (kc >> 1);
#pragma unroll
      for (int P = 0; P < 2; P++) *reinterpret_cast<bf16x8*>(line + 32 * P) = outl[P];
    }
    if constexpr (STATS) {
      // a 16-lane row holds 16 pixels of the same 16 channels (2 pairs x 8): DPP row sums, lane p16 keeps channel p16
#pragma unroll
      for (int P = 0; P < 2; P++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          const float s1 = pp_row_sum(st1[P][e]), s2 = pp_row_sum(st2[P][e]);
          const bool mine = p16 == P * 8 + e;
          keep1 = mine ? s1 : keep1;
          keep2 = mine ? s2 : keep2;
        }
    }
  }
  if constexpr (STATS) {
    // ONE full-width atomic per statistic and wave: lane (p16, kc) carries channel (P, e) = (p16 >> 3, p16 & 7) of its row
    float* dst = a.stat_sums + ((long)(n / a.stat_n_per_group) * GANK_STAT_SLOTS + (blockIdx.x % GANK_STAT_SLOTS)) * 2 * a.Cout;
    const int co = tile_n * 256 + wn * 64 + (2 * (p16 >> 3) + (kc & 1)) * 16 + 8 * (kc >> 1) + (p16 & 7);
#ifdef GANK_TUNING
    if (gank_stats_dbg) return;
#endif
    atomicAdd(dst + co, keep1);
    atomicAdd(dst + a.Cout + co, keep2);
  }
#else
  {
    // all 64 running sums are live beside the accumulators (200 VGPRs; holding the packed results for a separate store pass
    // instead spilled 22-25 of them)
    float st1[STATS ? 2 : 1][2][8], st2[STATS ? 2 : 1][2][8];
    if constexpr (STATS) {
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int q = 0; q < 2; q++)
#pragma unroll
          for (int e = 0; e < 8; e++) { st1[i][q][e] = 0.f; st2[i][q][e] = 0.f; }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int py = py0 + wm * (PHH / 2) + j * RS + prow, px = px0 + pcol;
      const long m = PHASE ? ((long)(n * 2 * a.H + 2 * py + (phase >> 1))) * (2 * a.W) + 2 * px + (phase & 1) : ((long)(n * a.H + py)) * a.W + px;
      const long mr = (!PHASE && (a.flags & IG_RES_UP2X)) ? ((long)(n * (a.H >> 1) + (py >> 1))) * (a.W >> 1) + (px >> 1) : m;
      bf16x8 outl[2][2];
#pragma unroll
      for (int i = 0; i < 2; i++) {
#pragma unroll
        for (int q = 0; q < 2; q++) {
          const int co = tile_n * 256 + (wn * 2 + i) * 32 + 16 * q + 8 * h;
          float v[8];
#pragma unroll
          for (int e = 0; e < 4; e++) {
            float lo = acc[i][j][8 * q + e] * a.scale, hi = acc[i][j][8 * q + 4 + e] * a.scale;
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
            v[e] = lo;
            v[4 + e] = hi;
          }
          const long o = m * a.Cout + co;
          // y in the operation order of the plain variant (bit-identical outputs); the statistics are of d = y - bias,
          // carried beside it without ever adding the bias
          float bb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, d[8];
          if (a.bias) {
            const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.bias + co), b1 = *reinterpret_cast<const f32x4*>(a.bias + co + 4);
#pragma unroll
            for (int e = 0; e < 4; e++) { bb[e] = b0[e]; bb[4 + e] = b1[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; e++) { d[e] = v[e]; v[e] += bb[e]; }
          if (a.mask) {
            const bf16x8 mk = *reinterpret_cast<const bf16x8*>(a.mask + o);
#pragma unroll
            for (int e = 0; e < 8; e++) { const bool on = bf2f(mk[e]) > 0.f; v[e] = on ? v[e] : 0.f; d[e] = on ? d[e] : -bb[e]; }
          }
          if (a.res) {
            const bf16x8 rs = *reinterpret_cast<const bf16x8*>(a.res + mr * a.Cout + co);
#pragma unroll
            for (int e = 0; e < 8; e++) { const float t = bf2f(rs[e]); v[e] += t; d[e] += t; }
          }
          if constexpr (STATS) {
#pragma unroll
            for (int e = 0; e < 8; e++) { st1[i][q][e] += d[e]; st2[i][q][e] += d[e] * d[e]; }
          }
#pragma unroll
          for (int e = 0; e < 8; e++) outl[i][q][e] = f2bf(otanh ? tanhf(v[e]) : v[e]);
        }
      }
      bf16* line = a.y + m * a.Cout + tile_n * 256 + wn * 64 + 8 * h;
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int q = 0; q < 2; q++) *reinterpret_cast<bf16x8*>(line + i * 32 + 16 * q) = outl[i][q];
    }
    if constexpr (STATS) {
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
      for (int q = 0; q < 2; q++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
          float s1 = pp_row_sum(st1[i][q][e]), s2 = pp_row_sum(st2[i][q][e]);
          s1 += __shfl_xor(s1, 16, 64);
          s2 += __shfl_xor(s2, 16, 64);
          const bool mine = r == (2 * i + q) * 8 + e;
          keep1 = mine ? s1 : keep1;
          keep2 = mine ? s2 : keep2;
        }
    }
  }
  if constexpr (STATS) {
    // ONE full-width atomic per statistic and wave (an atomic instruction costs the memory pipe the same with 2 lanes as
    // with 64: 512 two-lane atomics per block made this epilogue slower than the statistics pass it replaces)
    float* dst = a.stat_sums + ((long)(n / a.stat_n_per_group) * GANK_STAT_SLOTS + (blockIdx.x % GANK_STAT_SLOTS)) * 2 * a.Cout;
    const int ci = r >> 4, cq = (r >> 3) & 1, ce = r & 7;
    const int co = tile_n * 256 + (wn * 2 + ci) * 32 + 16 * cq + 8 * h + ce;
#ifdef GANK_TUNING
    if (gank_stats_dbg) return;
#endif
    atomicAdd(dst + co, keep1);
    atomicAdd(dst + a.Cout + co, keep2);
  }
#endif
}

static thread_local int tl_stats_done = 0;     // did the kernel chosen by the last dispatch accumulate a.stat_sums?
__global__ void ig_zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

static void stats_zero(const IgemmArgs& a, hipStream_t s);
// 128x128 tiles from this many tiles on; fewer run as 64x64 (4x the workgroups: the grid rounds better on 256 CUs)
static int t128_min() {
  static const int v = gank_tune("GANK_IGEMM_T128_MIN", 192);   // experiment knob
  return v;
}
// generic kernel with the statistics epilogue: every pixel tile full and inside one tower, every channel tile full
// (a.M, a.H, a.W: the grid the tiles walk -- the low-resolution grid in phase mode)
static bool igemm_stats_ok(const IgemmArgs& a, int BM, int BN) {
  static const int env = gank_tune("GANK_IGEMM_STATS", 1);   // experiment knob: GANK_IGEMM_STATS=0 leaves the statistics to the batch-norm kernels
  return env && a.stat_sums != nullptr && !(a.flags & GANK_OUT_TANH) && a.M % BM == 0 && a.Cout % BN == 0 &&
         ((long)a.stat_n_per_group * a.H * a.W) % BM == 0;
}
template <int MODE, int PW>
static int launch_pp(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  const bool stats = a.stat_sums != nullptr;
  if (stats) { tl_stats_done = 1; stats_zero(a, s); }
  constexpr bool PHASE = (MODE & 4) != 0;
  const int tiles = a.N * (a.H / (256 / PW)) * (a.W / PW);
  a.tiles_pp = tiles;
  a.phase_inner = phase_inner_env();
  a.tiles_m = PHASE ? 4 * tiles : tiles;
  a.tiles_n = a.Cout / 256;
  if (PHASE) { a.Kpad = 4 * a.Cin; a.Hin = a.H; a.Win = a.W; }
  auto kern = stats ? conv_igemm_pp_kernel<MODE, PW, true> : conv_igemm_pp_kernel<MODE, PW, false>;
  if (stats) { GANK_MAX_DYNAMIC_LDS((conv_igemm_pp_kernel<MODE, PW, true>), PP_LDS_BYTES, "conv_igemm_pp"); }
  else { GANK_MAX_DYNAMIC_LDS((conv_igemm_pp_kernel<MODE, PW, false>), PP_LDS_BYTES, "conv_igemm_pp"); }
  static const std::string tag = gank_format("conv_igemm_pp_kernel<%d, %d>", MODE, PW);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(512), PP_LDS_BYTES, s, a);
  GANK_LAUNCH_OK("conv_igemm_pp");
  return 0;
}
// geometry both PP forms need (H, W = the grid the patches tile: the output for a plain conv, the low-res grid in phase mode)
static int pp_patch_width(int H, int W) { return (W % 32 == 0 && H % 8 == 0) ? 32 : (W % 16 == 0 && H % 16 == 0) ? 16 : 0; }
static int pp_env() {          // GANK_IGEMM_PP: 0 off, 1 32-wide patches of plain convs only, 2 (default) every form
  static const int v = gank_tune("GANK_IGEMM_PP", 2);
  return v;
}
// One 512-thread block per CU: below ~a full round of blocks the 128-wide kernels (2-3 blocks per CU) fill the chip better
// (16 x 16 images, n = 128: 128 blocks, 49.6 us against 43.9).
static bool pp_enough_blocks(const IgemmArgs& a, int phases) {
  const int pw = pp_patch_width(a.H, a.W);
  return pw != 0 && (long)phases * a.N * (a.H / (256 / pw)) * (a.W / pw) * (a.Cout / 256) >= 224;
}
static bool pp_phase_ok(const IgemmArgs& a) {   // a.H, a.W = low-res grid
  return pp_env() >= 2 && a.Cout % 256 == 0 && a.Cin % 32 == 0 && pp_enough_blocks(a, 4);
}
static int launch_pp_phase(const IgemmArgs& a, hipStream_t s) {
  const bool relu = (a.flags & GANK_IN_RELU) != 0;
  if (pp_patch_width(a.H, a.W) == 32) return relu ? launch_pp<5, 32>(a, s) : launch_pp<4, 32>(a, s);
  return relu ? launch_pp<5, 16>(a, s) : launch_pp<4, 16>(a, s);
}

template <int MODE, int BN>
static int launch_patch(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  a.tiles_m = a.N * (a.H / 8) * (a.W / 16);
  a.tiles_n = a.CoutPad / BN;
  const size_t lds = ((size_t)10 * HROWP + (size_t)2 * BN * LROW) * sizeof(bf16);
  auto kern = conv_igemm_patch_kernel<MODE, BN>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm_patch");
  static const std::string tag = gank_format("conv_igemm_patch_kernel<%d, %d>", MODE, BN);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(256), lds, s, a);
  GANK_LAUNCH_OK("conv_igemm_patch");
  return 0;
}

template <int WM, int WN, int TM, int TN, bool PACKED, int PF, int MODE>
static int launch_cfg(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  a.tiles_m = cdiv(a.M, BM);
  a.tiles_n = a.CoutPad / BN;
  const size_t lds = (size_t)2 * (BM + BN) * LROW * sizeof(bf16);
  static const std::string tag = gank_format("conv_igemm_kernel<%d, %d, %d, %d, %s, %d, %d>", WM, WN, TM, TN, PACKED ? "true" : "false", PF, MODE);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  if constexpr (!PACKED && MODE == 0 && TN * 16 <= 32) {
    if (igemm_stats_ok(a, BM, BN)) {      // the layer that consumes y is a batch norm: its statistics ride on this epilogue
      auto kern = conv_igemm_kernel<WM, WN, TM, TN, PACKED, PF, MODE, true>;
      GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm");
      tl_stats_done = 1;
      stats_zero(a, s);
      hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), lds, s, a);
      GANK_LAUNCH_OK("conv_igemm");
      return 0;
    }
  }
  auto kern = conv_igemm_kernel<WM, WN, TM, TN, PACKED, PF, MODE>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm");
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), lds, s, a);
  GANK_LAUNCH_OK("conv_igemm");
  return 0;
}

template <int WM, int WN, int TM, int TN, int PF>
static int launch_phase(const IgemmArgs& a0, hipStream_t s) {
  IgemmArgs a = a0;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  a.tiles_pp = cdiv(a.M, BM);
  a.phase_inner = phase_inner_env();
  a.tiles_m = 4 * a.tiles_pp;
  a.tiles_n = a.CoutPad / BN;
  const size_t lds = (size_t)2 * (BM + BN) * LROW * sizeof(bf16);
  static const std::string tag = gank_format("conv_igemm_kernel<%d, %d, %d, %d, false, %d, 4>", WM, WN, TM, TN, PF);     // magic static: built once, thread-safe
  gank_prof_tag(0, tag.c_str());
  if constexpr (TN * 16 <= 32) {
    if (igemm_stats_ok(a, BM, BN)) {
      auto kern = conv_igemm_kernel<WM, WN, TM, TN, false, PF, 4, true>;
      GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm");
      tl_stats_done = 1;
      stats_zero(a, s);
      hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), lds, s, a);
      GANK_LAUNCH_OK("conv_igemm_phase");
      return 0;
    }
  }
  auto kern = conv_igemm_kernel<WM, WN, TM, TN, false, PF, 4>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_igemm");
  hipLaunchKernelGGL(kern, dim3(a.tiles_m * a.tiles_n), dim3(WM * WN * 64), lds, s, a);
  GANK_LAUNCH_OK("conv_igemm_phase");
  return 0;
}

template <int WM, int WN, int TM, int TN, int PF>
static int launch_mode(const IgemmArgs& a, hipStream_t s) {
  const int mode = ((a.flags & GANK_IN_RELU) ? 1 : 0) | ((a.flags & (GANK_IN_UPSAMPLE2X | IG_IN_ZEROINS2X)) ? 2 : 0);
  switch (mode) {
    case 0: return launch_cfg<WM, WN, TM, TN, false, PF, 0>(a, s);
    case 1: return launch_cfg<WM, WN, TM, TN, false, PF, 1>(a, s);
    case 2: return launch_cfg<WM, WN, TM, TN, false, PF, 2>(a, s);
    default: return launch_cfg<WM, WN, TM, TN, false, PF, 3>(a, s);
  }
}

// x [N,Hin,Win,Cin] -> y [N,H,W,Cout];  w [CoutPad][Kpad] bf16
int gank_igemm_dispatch(IgemmArgs a, hipStream_t s) {
  GANK_REQUIRE(a.x && a.w && a.y, "conv_igemm: null pointer");
  GANK_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "conv_igemm: bad shape");
  GANK_REQUIRE(a.ks >= 1 && a.ks <= 7, "conv_igemm: unsupported filter size %d", a.ks);
  GANK_REQUIRE((long)a.N * a.Hin * a.Win * a.Cin < (1L << 30) && (long)a.N * a.H * a.W * a.Cout < (1L << 40), "conv_igemm: tensor too large (32-bit byte offsets)");
  a.taps = a.ks * a.ks;
  a.CoutPad = roundup(a.Cout, 32);
  a.Kpad = roundup(a.taps * a.Cin, 64);
  a.nsteps = a.Kpad / 64;
  a.M = a.N * a.H * a.W;
  a.sw = log2_or_neg(a.W);
  a.shw = log2_or_neg(a.H * a.W);
  const bool packed = (a.Cin % 64) != 0;
  const double flops = 2.0 * a.M * (double)a.Cout * a.taps * a.Cin;
  // algorithmic bytes: every input pixel-channel and weight read once, every output written once
  // (+ the epilogue's operands: the residual -- a quarter of it when it is added upsampled -- and the relu mask)
  gank_prof_begin(0, flops, s, 2.0 * ((double)a.N * a.Hin * a.Win * a.Cin + (double)a.taps * a.Cin * a.Cout + (double)a.M * a.Cout +
                                      (a.res ? (double)a.M * a.Cout / ((a.flags & IG_RES_UP2X) ? 4.0 : 1.0) : 0.0) + (a.mask ? (double)a.M * a.Cout : 0.0)));
  int rc;
  const long tiles128 = (long)cdiv(a.M, 128) * (a.CoutPad / 128);
  static const int ko_env = gank_tune("GANK_IGEMM_KORDER", 0);
  a.korder = ko_env;
  static const int pf_env = gank_tune("GANK_IGEMM_PF", 0);   // experiment knob: GANK_IGEMM_PF=1|2|3 overrides the prefetch depth
  static const int patch_env = gank_tune("GANK_IGEMM_PATCH", 1);   // experiment knob: GANK_IGEMM_PATCH=0 disables the LDS-patch kernel
  const bool patch_geom = patch_env && !packed && a.ks == 3 && a.pad == 1 &&
                          !(a.flags & (GANK_IN_UPSAMPLE2X | IG_IN_ZEROINS2X | IG_IN_STRIDE2)) &&
                          a.W % 16 == 0 && a.H % 8 == 0 && a.Hin == a.H && a.Win == a.W;
  const bool patch_ok = patch_geom && (a.Cout % 4) == 0 && a.CoutPad % 128 == 0;
  const bool patch32_ok = patch_geom && a.CoutPad == 32;
  const bool narrow_ok = !(a.flags & IG_RES_UP2X) && a.Cin == 3 && (a.ks == 1 || a.ks == 3) && a.pad == (a.ks - 1) / 2 && a.CoutPad % 128 == 0 && a.Cout % 4 == 0 &&
                         !(a.flags & (GANK_IN_UPSAMPLE2X | IG_IN_ZEROINS2X | IG_IN_STRIDE2)) && a.Hin == a.H && a.Win == a.W;
  const bool pp_ok = pp_env() && patch_geom && a.Cout % 256 == 0 && a.Kpad == a.taps * a.Cin && pp_enough_blocks(a, 1) &&
                     (pp_env() >= 2 || pp_patch_width(a.H, a.W) == 32);
  if (narrow_ok) {
    rc = a.ks == 3 ? launch_narrow_in<3, 3>(a, s) : launch_narrow_in<1, 3>(a, s);
  } else if (pp_ok) {
    const bool relu = (a.flags & GANK_IN_RELU) != 0;
    if (pp_patch_width(a.H, a.W) == 32) rc = relu ? launch_pp<1, 32>(a, s) : launch_pp<0, 32>(a, s);
    else rc = relu ? launch_pp<1, 16>(a, s) : launch_pp<0, 16>(a, s);
  } else if (patch_ok) {
    rc = (a.flags & GANK_IN_RELU) ? launch_patch<1, 128>(a, s) : launch_patch<0, 128>(a, s);
  } else if (patch32_ok && few_ok(a)) {
    rc = (a.flags & GANK_IN_RELU) ? launch_few<1>(a, s) : launch_few<0>(a, s);
  } else if (patch32_ok) {
    rc = (a.flags & GANK_IN_RELU) ? launch_patch<1, 32>(a, s) : launch_patch<0, 32>(a, s);
  } else if (packed) {
    if (a.CoutPad % 64 == 0) rc = launch_cfg<2, 2, 1, 1, true, 1, 0>(a, s);
    else rc = launch_cfg<4, 1, 2, 1, true, 1, 0>(a, s);
#ifdef GANK_TUNING          // tile shapes and prefetch depths that lost their A/B runs: only an experiment build instantiates them
  } else if (a.CoutPad % 128 == 0 && tiles128 >= t128_min() && pf_env >= 20) {
    if (pf_env == 20) rc = launch_mode<4, 2, 2, 2, 1>(a, s);          // 256 x 128, 8 waves
    else if (pf_env == 22 && a.CoutPad % 256 == 0) rc = launch_mode<2, 4, 2, 2, 2>(a, s);   // 128 x 256
    else if (pf_env == 23 && a.CoutPad % 256 == 0) rc = launch_mode<4, 2, 2, 4, 2>(a, s);   // 256 x 256, wave 64x128
    else rc = launch_mode<4, 2, 2, 2, 2>(a, s);
  } else if (a.CoutPad % 128 == 0 && tiles128 >= t128_min() && (pf_env == 1 || pf_env == 3)) {
    rc = pf_env == 1 ? launch_mode<2, 2, 2, 2, 1>(a, s) : launch_mode<2, 2, 2, 2, 3>(a, s);
  } else if (a.CoutPad % 64 == 0 && !(a.CoutPad % 128 == 0 && tiles128 >= t128_min()) && (pf_env == 1 || pf_env == 11 || pf_env == 12 || pf_env == 3)) {
    rc = (pf_env == 1 || pf_env == 11) ? launch_mode<2, 2, 1, 1, 1>(a, s) : launch_mode<2, 2, 1, 1, 2>(a, s);
#endif
  } else if (a.CoutPad % 128 == 0 && tiles128 >= t128_min()) {
    rc = launch_mode<2, 2, 2, 2, 2>(a, s);               // 128 x 128 tiles, prefetch depth 2
  } else if (a.CoutPad % 64 == 0) {
    rc = launch_mode<2, 2, 1, 1, 4>(a, s);               // 64 x 64 tiles (4x the workgroups: small grids round better on 256 CUs), depth 4
  } else {
    rc = launch_mode<4, 1, 2, 1, 2>(a, s);
  }
  gank_prof_end(0, s);
  return rc;
}

static int stats_setup(IgemmArgs& a, float* stat_sums, int groups, int N, int Cout, hipStream_t s, int flags) {
  tl_stats_done = 0;
  if (!stat_sums) return 0;
  gank_stats_dbg_init();
  a.stat_prezeroed = (flags & GANK_STATS_PREZEROED) != 0;
  GANK_REQUIRE(groups > 0 && N % groups == 0, "conv statistics: batch %d not divisible by %d towers", N, groups);
  a.stat_sums = stat_sums;
  a.stat_n_per_group = N / groups;
  a.stat_groups = groups;
  return 0;
}
// the sums are cleared by the launcher of a kernel that accumulates them (a kernel that does not never pays for the fill)
static void stats_zero(const IgemmArgs& a, hipStream_t s) {
  if (a.stat_prezeroed) return;          // GANK_STATS_PREZEROED: the caller cleared the sums (one fill for a whole pass)
  const int n = a.stat_groups * GANK_STAT_SLOTS * 2 * a.Cout;
  hipLaunchKernelGGL(ig_zero_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a.stat_sums, n);
}

static int conv2d_fprop_impl(const void* x, const void* wf, const float* bias, const void* residual,
                             const void* relu_ref, void* y, int N, int H, int W, int Cin, int Cout, int ksize,
                             int flags, float scale, float* stat_sums, int groups, void* stream) {
  GANK_REQUIRE(ksize % 2 == 1, "conv2d_fprop: even filter sizes are not on this path (ksize=%d)", ksize);
  GANK_REQUIRE(!(flags & GANK_IN_UPSAMPLE2X) || (H % 2 == 0 && W % 2 == 0), "conv2d_fprop: upsample needs even output size");
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wf; a.bias = bias; a.res = (const bf16*)residual;
  a.mask = (const bf16*)relu_ref; a.y = (bf16*)y;
  a.N = N; a.H = H; a.W = W;
  const bool up = flags & GANK_IN_UPSAMPLE2X;
  a.Hin = up ? H / 2 : H; a.Win = up ? W / 2 : W;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.flags = (flags & (GANK_IN_UPSAMPLE2X | GANK_IN_RELU | GANK_OUT_TANH)) |
            ((flags & GANK_RES_UPSAMPLE2X) ? IG_RES_UP2X : 0);
  GANK_REQUIRE(!(flags & GANK_RES_UPSAMPLE2X) || (residual && H % 2 == 0 && W % 2 == 0), "conv2d_fprop: RES_UPSAMPLE2X needs a residual and even output size");
  a.scale = scale;
  if (stats_setup(a, stat_sums, groups, N, Cout, (hipStream_t)stream, flags)) return 1;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int gank_conv2d_fprop(const void* x, const void* wf, const float* bias, const void* residual,
                                 const void* relu_ref, void* y, int N, int H, int W, int Cin, int Cout, int ksize,
                                 int flags, float scale, void* stream) {
  return conv2d_fprop_impl(x, wf, bias, residual, relu_ref, y, N, H, W, Cin, Cout, ksize, flags, scale, nullptr, 0, stream);
}
extern "C" int gank_conv2d_fprop_stats(const void* x, const void* wf, const float* bias, const void* residual,
                                       const void* relu_ref, void* y, int N, int H, int W, int Cin, int Cout, int ksize,
                                       int flags, float scale, float* stat_sums, int groups, int* produced, void* stream) {
  GANK_REQUIRE(stat_sums && produced, "conv2d_fprop_stats: null statistics pointers");
  const int rc = conv2d_fprop_impl(x, wf, bias, residual, relu_ref, y, N, H, W, Cin, Cout, ksize, flags, scale, stat_sums, groups, stream);
  *produced = tl_stats_done;
  return rc;
}

// y = conv3x3_SAME(relu(cond_batchnorm(x))) + bias [tanh]: the normalisation rides on the conv's operand staging (patch
// kernel, MODE bit 3) -- for passes that keep nothing for a backward pass (the fakes of the critic updates, sampling): the
// normalised tensor is never written or read (G.OutputNorm + G.Output: 168 MB each way at 320 samples).
extern "C" int gank_cbn_relu_conv3x3_fprop(const void* x, const int32_t* labels, const float* gamma, const float* beta, const float* stats,
                                           const void* wf, const float* bias, void* y, int N, int H, int W, int Cin, int Cout,
                                           int groups, int n_labels, int flags, void* stream) {
  GANK_REQUIRE(x && labels && gamma && beta && stats && wf && y, "cbn_relu_conv3x3_fprop: null pointer");
  GANK_REQUIRE(groups > 0 && N % groups == 0 && n_labels > 0, "cbn_relu_conv3x3_fprop: batch %d / %d towers", N, groups);
  GANK_REQUIRE(Cin % 64 == 0 && W % 16 == 0 && H % 8 == 0, "cbn_relu_conv3x3_fprop: needs Cin %% 64 == 0, W %% 16 == 0, H %% 8 == 0");
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wf; a.bias = bias; a.y = (bf16*)y;
  a.N = N; a.H = H; a.W = W; a.Hin = H; a.Win = W; a.Cin = Cin; a.Cout = Cout; a.ks = 3; a.pad = 1;
  a.flags = flags & GANK_OUT_TANH;
  a.scale = 1.f;
  a.taps = 9; a.CoutPad = roundup(Cout, 32); a.Kpad = roundup(9 * Cin, 64); a.nsteps = a.Kpad / 64;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  GANK_REQUIRE(a.CoutPad == 32 || a.CoutPad % 128 == 0, "cbn_relu_conv3x3_fprop: Cout must pad to 32 or a multiple of 128 (got %d)", Cout);
  GANK_REQUIRE((long)N * H * W * Cin < (1L << 30), "cbn_relu_conv3x3_fprop: tensor too large (32-bit byte offsets)");
  a.cbn_stats = stats; a.cbn_gamma = gamma; a.cbn_beta = beta; a.cbn_labels = labels;
  a.cbn_n_per_group = N / groups; a.cbn_n_labels = n_labels;
  hipStream_t s = (hipStream_t)stream;
  gank_prof_begin(0, 2.0 * a.M * (double)Cout * 9 * Cin, s, 2.0 * ((double)a.M * Cin + 9.0 * Cin * Cout + (double)a.M * Cout));
  const int rc = a.CoutPad == 32 ? (few_ok(a) ? launch_few<8>(a, s) : launch_patch<8, 32>(a, s)) : launch_patch<8, 128>(a, s);
  gank_prof_end(0, s);
  return rc;
}

// MeanPoolConv with a 1x1 filter on a 3-channel image (D.Block.1.Shortcut, gan_cifar_resnet.py:125-137,218-221): the 2x2
// mean is taken inside the conv's gather; `pooled` (optional) receives the pooled image for the filter gradient.
extern "C" int gank_meanpool_conv1x1_fprop(const void* x, const void* wf, const float* bias, void* y, void* pooled,
                                           int N, int H, int W, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(x && wf && y && N > 0 && H > 0 && W > 0, "meanpool_conv1x1_fprop: bad arguments");
  GANK_REQUIRE(Cin == 3, "meanpool_conv1x1_fprop: built for 3-channel images (Cin = %d)", Cin);
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wf; a.bias = bias; a.y = (bf16*)y; a.aux_out = (bf16*)pooled;
  a.N = N; a.H = H; a.W = W; a.Hin = 2 * H; a.Win = 2 * W; a.Cin = Cin; a.Cout = Cout; a.ks = 1; a.pad = 0;
  a.scale = 1.f;
  a.taps = 1; a.CoutPad = roundup(Cout, 32); a.Kpad = roundup(Cin, 64); a.nsteps = a.Kpad / 64;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  GANK_REQUIRE(a.CoutPad % 128 == 0 && Cout % 4 == 0, "meanpool_conv1x1_fprop: Cout must be a multiple of 4 that pads to a multiple of 128 (got %d)", Cout);
  GANK_REQUIRE((long)N * 4 * H * W * Cin < (1L << 30), "meanpool_conv1x1_fprop: tensor too large (32-bit offsets)");
  hipStream_t s = (hipStream_t)stream;
  gank_prof_begin(0, 2.0 * a.M * (double)Cout * Cin, s, 2.0 * (4.0 * a.M * Cin + (double)Cin * Cout + (double)a.M * Cout));
  const int rc = launch_narrow_in<1, 3, true>(a, s);
  gank_prof_end(0, s);
  return rc;
}

// gank_conv2d_fprop(x [N,H,W,3], wf1: 3x3 -> Cout1, no flags) and gank_meanpool_conv1x1_fprop(x, wfs: -> Couts at H/2 x W/2, pooled side
// output) in ONE launch -- the two image-side layers of OptimizedResBlockDisc1 (gan_cifar_resnet.py:212-234); results bit for bit those
// of the two entries.
extern "C" int gank_image_conv_pair_fprop(const void* x, const void* wf1, const float* bias1, void* y1, const void* wfs, const float* biass, void* ys,
                                          void* pooled, int N, int H, int W, int Cout1, int Couts, void* stream) {
  GANK_REQUIRE(x && wf1 && y1 && wfs && ys && N > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0, "image_conv_pair_fprop: bad arguments");
  GANK_REQUIRE(Cout1 % 128 == 0 && Couts % 128 == 0, "image_conv_pair_fprop: channel counts must be multiples of 128 (got %d, %d)", Cout1, Couts);
  GANK_REQUIRE((long)N * H * W * 3 < (1L << 30) && (long)N * H * W * Cout1 < (1L << 31), "image_conv_pair_fprop: tensor too large (32-bit offsets)");
  IgemmArgs a{}, b{};
  a.x = (const bf16*)x; a.w = (const bf16*)wf1; a.bias = bias1; a.y = (bf16*)y1;
  a.N = N; a.H = H; a.W = W; a.Hin = H; a.Win = W; a.Cin = 3; a.Cout = Cout1; a.ks = 3; a.pad = 1; a.scale = 1.f;
  a.taps = 9; a.CoutPad = Cout1; a.Kpad = roundup(27, 64); a.nsteps = 1;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  a.tiles_m = cdiv(a.M, 128); a.tiles_n = a.CoutPad / 128;
  { static const int ko_env = gank_tune("GANK_IGEMM_KORDER", 0); a.korder = ko_env; }
  const int Hp = H / 2, Wp = W / 2;
  b.x = (const bf16*)x; b.w = (const bf16*)wfs; b.bias = biass; b.y = (bf16*)ys; b.aux_out = (bf16*)pooled;
  b.N = N; b.H = Hp; b.W = Wp; b.Hin = H; b.Win = W; b.Cin = 3; b.Cout = Couts; b.ks = 1; b.pad = 0; b.scale = 1.f;
  b.taps = 1; b.CoutPad = Couts; b.Kpad = roundup(3, 64); b.nsteps = 1;
  b.M = N * Hp * Wp; b.sw = log2_or_neg(Wp); b.shw = log2_or_neg(Hp * Wp);
  b.tiles_m = cdiv(b.M, 128); b.tiles_n = b.CoutPad / 128;
  b.korder = 0;                       // (as gank_meanpool_conv1x1_fprop)
  hipStream_t s = (hipStream_t)stream;
  const int blocks_a = a.tiles_m * a.tiles_n, blocks_b = b.tiles_m * b.tiles_n;
  gank_prof_begin(0, 2.0 * a.M * (double)Cout1 * 27 + 2.0 * b.M * (double)Couts * 3, s,
                  2.0 * ((double)a.M * 3 + 27.0 * Cout1 + (double)a.M * Cout1 + 3.0 * Couts + (double)b.M * Couts));
  gank_prof_tag(0, "conv_narrow_in_pair_kernel");
  hipLaunchKernelGGL(conv_narrow_in_pair_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, s, a, b, blocks_a);
  gank_prof_end(0, s);
  GANK_LAUNCH_OK("image_conv_pair_fprop");
  return 0;
}

// dgrad of the stride-1 SAME conv = the same engine on dy with the flipped/transposed operand (wd)
extern "C" int gank_conv2d_dgrad(const void* dy, const void* wd, const void* residual, const void* relu_ref, void* dx,
                                 int N, int H, int W, int Cin, int Cout, int ksize, int flags, float scale, void* stream) {
  GANK_REQUIRE(ksize % 2 == 1, "conv2d_dgrad: even filter sizes are not on this path (ksize=%d)", ksize);
  IgemmArgs a{};
  a.x = (const bf16*)dy; a.w = (const bf16*)wd; a.bias = nullptr; a.res = (const bf16*)residual;
  a.mask = (const bf16*)relu_ref; a.y = (bf16*)dx;
  a.N = N; a.H = H; a.W = W;
  const bool up = flags & GANK_IN_UPSAMPLE2X;
  a.Hin = up ? H / 2 : H; a.Win = up ? W / 2 : W;
  a.Cin = Cout; a.Cout = Cin; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.flags = (flags & (GANK_IN_UPSAMPLE2X | GANK_IN_RELU));
  a.scale = scale;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}


// ---- general convolution: any filter size <= 7 (even sizes too), stride 1 or 2, explicit leading pad, input and output
// sizes given separately -- TF SAME with an even filter (pad_before = (k-1)/2 ... the surplus goes after), explicit tf.pad +
// VALID, and the 1x1 -> 1x1 bottom of a U-Net are all the same gather with out-of-image taps read as zeros.  Runs on the
// generic implicit-GEMM kernel (Pix2Pix/networks.py:366-470: 4x4 stride-2 encoders, 4x4 stride-1 decoders on NN-upsampled
// inputs, the PatchGAN critic's padded VALID convs).  flags: GANK_IN_RELU, GANK_IN_UPSAMPLE2X (stride 1; Hin, Win are the
// STORED input sizes), GANK_OUT_TANH.
extern "C" int gank_conv2d_general_fprop(const void* x, const void* wf, const float* bias, void* y, int N, int Hin, int Win,
                                         int Hout, int Wout, int Cin, int Cout, int ksize, int stride, int pad, int flags, void* stream) {
  GANK_REQUIRE(stride == 1 || stride == 2, "conv2d_general_fprop: stride %d (1 or 2)", stride);
  GANK_REQUIRE(pad >= 0 && pad < ksize, "conv2d_general_fprop: pad %d outside [0, ksize)", pad);
  GANK_REQUIRE(!((flags & GANK_IN_UPSAMPLE2X) && stride == 2), "conv2d_general_fprop: upsampled input with stride 2");
  GANK_REQUIRE(Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "conv2d_general_fprop: bad sizes");
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wf; a.bias = bias; a.y = (bf16*)y;
  a.N = N; a.H = Hout; a.W = Wout; a.Hin = Hin; a.Win = Win;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = pad;
  a.flags = (flags & (GANK_IN_UPSAMPLE2X | GANK_IN_RELU | GANK_OUT_TANH)) | (stride == 2 ? IG_IN_STRIDE2 : 0);
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}
// input gradient of the stride-1 form: the same engine on dy with the flipped operand (wd of gank_conv2d_prep_weights) and
// the complementary pad.  dx [N,Hx,Wx,Cin] (for an upsampled-input conv: the gradient at the UPSAMPLED size -- the caller
// sums 2x2), relu_ref (optional, like dx) masks the result.  The stride-2 input gradient is the transposed conv:
// gank_deconv2d_prep_phases + gank_upconv3x3_fprop on the same filter memory.
extern "C" int gank_conv2d_general_dgrad(const void* dy, const void* wd, const void* relu_ref, void* dx, int N, int Hx, int Wx,
                                         int Hdy, int Wdy, int Cin, int Cout, int ksize, int pad, void* stream) {
  GANK_REQUIRE(pad >= 0 && pad < ksize, "conv2d_general_dgrad: pad %d outside [0, ksize)", pad);
  IgemmArgs a{};
  a.x = (const bf16*)dy; a.w = (const bf16*)wd; a.mask = (const bf16*)relu_ref; a.y = (bf16*)dx;
  a.N = N; a.H = Hx; a.W = Wx; a.Hin = Hdy; a.Win = Wdy;
  a.Cin = Cout; a.Cout = Cin; a.ks = ksize; a.pad = ksize - 1 - pad;
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}

// NN-upsample(2x) + 3x3 SAME conv == stride-2 transposed conv with a 4x4 kernel: computed as 4 output phases
// of 2x2 taps over the low-res input (2.25x fewer MACs than 9 taps at high resolution).  wph comes from
// gank_upconv3x3_prep_weights.  Epilogue as gank_conv2d_fprop (bias, residual at OUTPUT resolution, tanh).
static int upconv3x3_fprop_impl(const void* x, const void* wph, const float* bias, const void* residual, void* y,
                                int N, int Hl, int Wl, int Cin, int Cout, int flags, float* stat_sums, int groups, void* stream) {
  GANK_REQUIRE(Cin % 64 == 0, "upconv3x3_fprop: Cin must be a multiple of 64 (got %d)", Cin);
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wph; a.bias = bias; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.N = N; a.H = Hl; a.W = Wl; a.Hin = Hl; a.Win = Wl;
  a.Cin = Cin; a.Cout = Cout; a.ks = 2; a.pad = 0;
  a.flags = flags & GANK_OUT_TANH;
  a.scale = 1.f;
  hipStream_t s = (hipStream_t)stream;
  a.taps = 4; a.CoutPad = roundup(Cout, 32); a.Kpad = 4 * Cin; a.nsteps = a.Kpad / 64;
  a.M = N * Hl * Wl; a.sw = log2_or_neg(Wl); a.shw = log2_or_neg(Hl * Wl);
  GANK_REQUIRE((long)N * Hl * Wl * Cin < (1L << 30) && (long)a.M * 4 * Cout < (1L << 31), "upconv3x3_fprop: tensor too large");
  if (stats_setup(a, stat_sums, groups, N, Cout, s, flags)) return 1;
  gank_prof_begin(0, 2.0 * a.M * 4.0 * (double)Cout * 4 * Cin, s, 2.0 * ((double)a.M * Cin + 16.0 * Cin * Cout + 4.0 * a.M * Cout + (residual ? 4.0 * a.M * Cout : 0.0)));
  int rc;
  const long tiles128 = 4L * cdiv(a.M, 128) * (a.CoutPad / 128);
  if (pp_phase_ok(a)) rc = launch_pp_phase(a, s);
  else if (patch_phase_ok(a)) rc = launch_patch_phase<4>(a, s);
  else if (a.CoutPad % 128 == 0 && tiles128 >= t128_min()) rc = launch_phase<2, 2, 2, 2, 2>(a, s);
  else if (a.CoutPad % 64 == 0) rc = launch_phase<2, 2, 1, 1, 4>(a, s);
  else rc = launch_phase<4, 1, 2, 1, 2>(a, s);
  gank_prof_end(0, s);
  return rc;
}

extern "C" int gank_upconv3x3_fprop(const void* x, const void* wph, const float* bias, const void* residual, void* y,
                                    int N, int Hl, int Wl, int Cin, int Cout, int flags, void* stream) {
  return upconv3x3_fprop_impl(x, wph, bias, residual, y, N, Hl, Wl, Cin, Cout, flags, nullptr, 0, stream);
}
extern "C" int gank_upconv3x3_fprop_stats(const void* x, const void* wph, const float* bias, const void* residual, void* y,
                                          int N, int Hl, int Wl, int Cin, int Cout, int flags, float* stat_sums, int groups, int* produced,
                                          void* stream) {
  GANK_REQUIRE(stat_sums && produced, "upconv3x3_fprop_stats: null statistics pointers");
  const int rc = upconv3x3_fprop_impl(x, wph, bias, residual, y, N, Hl, Wl, Cin, Cout, flags, stat_sums, groups, stream);
  *produced = tl_stats_done;
  return rc;
}

// its input gradient: dx_low = stride-2 SAME conv (4x4 taps) of dy with the combined kernel (wd4 layout from
// gank_upconv3x3_prep_weights): 16 taps per LOW-RES pixel = 4 per output pixel, and no 2x2 sum pass.
extern "C" int gank_upconv3x3_dgrad(const void* dy, const void* wd4, const void* relu_ref, void* dx, int N, int Hl, int Wl,
                                    int Cin, int Cout, void* stream) {
  IgemmArgs a{};
  a.x = (const bf16*)dy; a.w = (const bf16*)wd4; a.mask = (const bf16*)relu_ref; a.y = (bf16*)dx;
  a.N = N; a.H = Hl; a.W = Wl; a.Hin = 2 * Hl; a.Win = 2 * Wl;
  a.Cin = Cout; a.Cout = Cin; a.ks = 4; a.pad = 1;
  a.flags = IG_IN_STRIDE2;
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}

// ---- ConvMeanPool 3x3 (gan_cifar_resnet.py:112-123) as ONE 4x4 stride-2 conv ----------------------------
// mean_pool(conv3x3(x)) has 16 taps per POOLED pixel = 4 per conv output instead of 9, and no full-resolution
// intermediate.  Operands from gank_convpool3x3_prep_weights (the 1/4 is folded into them).
extern "C" int gank_convpool3x3_fprop(const void* x, const void* wp4, const float* bias, const void* residual, void* y,
                                      int N, int Hp, int Wp, int Cin, int Cout, int flags, void* stream) {
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wp4; a.bias = bias; a.res = (const bf16*)residual; a.y = (bf16*)y;
  a.N = N; a.H = Hp; a.W = Wp; a.Hin = 2 * Hp; a.Win = 2 * Wp;
  a.Cin = Cin; a.Cout = Cout; a.ks = 4; a.pad = 1;
  a.flags = IG_IN_STRIDE2 | (flags & GANK_IN_RELU);
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}

// its input gradient: the stride-2 transposed conv, as 4 output phases of 2x2 taps over dy (pooled resolution);
// relu_ref (optional, [N,2Hp,2Wp,Cin]) masks the result with relu'(x) of the pre-activation input.
extern "C" int gank_convpool3x3_dgrad(const void* dy, const void* wphd, const void* relu_ref, void* dx, int N, int Hp, int Wp,
                                      int Cin, int Cout, void* stream) {
  GANK_REQUIRE(Cout % 64 == 0, "convpool3x3_dgrad: Cout must be a multiple of 64 (got %d)", Cout);
  IgemmArgs a{};
  a.x = (const bf16*)dy; a.w = (const bf16*)wphd; a.mask = (const bf16*)relu_ref; a.y = (bf16*)dx;
  a.N = N; a.H = Hp; a.W = Wp; a.Hin = Hp; a.Win = Wp;
  a.Cin = Cout; a.Cout = Cin; a.ks = 2; a.pad = 0;
  a.flags = 0;
  a.scale = 1.f;
  hipStream_t s = (hipStream_t)stream;
  a.taps = 4; a.CoutPad = roundup(Cin, 32); a.Kpad = 4 * Cout; a.nsteps = a.Kpad / 64;
  a.M = N * Hp * Wp; a.sw = log2_or_neg(Wp); a.shw = log2_or_neg(Hp * Wp);
  GANK_REQUIRE((long)N * Hp * Wp * Cout < (1L << 30) && (long)a.M * 4 * Cin < (1L << 31), "convpool3x3_dgrad: tensor too large");
  gank_prof_begin(0, 2.0 * a.M * 4.0 * (double)Cin * 4 * Cout, s, 2.0 * ((double)a.M * Cout + 16.0 * Cin * Cout + 4.0 * a.M * Cin + (relu_ref ? 4.0 * a.M * Cin : 0.0)));
  int rc;
  const long tiles128 = 4L * cdiv(a.M, 128) * (a.CoutPad / 128);
  if (pp_phase_ok(a)) rc = launch_pp_phase(a, s);
  else if (patch_phase_ok(a)) rc = launch_patch_phase<4>(a, s);
  else if (a.CoutPad % 128 == 0 && tiles128 >= t128_min()) rc = launch_phase<2, 2, 2, 2, 2>(a, s);
  else if (a.CoutPad % 64 == 0) rc = launch_phase<2, 2, 1, 1, 4>(a, s);
  else rc = launch_phase<4, 1, 2, 1, 2>(a, s);
  gank_prof_end(0, s);
  return rc;
}

// ---- Deconv2D (common/ops/deconv2d.py:99-114): conv2d_transpose, stride 2, SAME ---------------------
// fprop = zero-insertion of x + stride-1 conv with the flipped filter (the dgrad operand layout of the
// filter viewed as HWIO [k,k,Cout,Cin]); dgrad = the stride-2 SAME conv itself.
static int deconv_pad_before(int ks) { int t = ks - 2; if (t < 0) t = 0; return t / 2; }

extern "C" int gank_deconv2d_fprop(const void* x, const void* wz, const float* bias, void* y, int N, int H, int W,
                                   int Cin, int Cout, int ksize, void* stream) {
  IgemmArgs a{};
  a.x = (const bf16*)x; a.w = (const bf16*)wz; a.bias = bias; a.y = (bf16*)y;
  a.N = N; a.H = 2 * H; a.W = 2 * W; a.Hin = H; a.Win = W;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize;
  a.pad = ksize - 1 - deconv_pad_before(ksize);
  a.flags = IG_IN_ZEROINS2X;
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}

extern "C" int gank_deconv2d_dgrad(const void* dy, const void* wfz, void* dx, int N, int H, int W, int Cin, int Cout,
                                   int ksize, void* stream) {
  IgemmArgs a{};
  a.x = (const bf16*)dy; a.w = (const bf16*)wfz; a.y = (bf16*)dx;
  a.N = N; a.H = H; a.W = W; a.Hin = 2 * H; a.Win = 2 * W;
  a.Cin = Cout; a.Cout = Cin; a.ks = ksize; a.pad = deconv_pad_before(ksize);
  a.flags = IG_IN_STRIDE2;
  a.scale = 1.f;
  return gank_igemm_dispatch(a, (hipStream_t)stream);
}
