// Filter-gradient (wgrad) implicit GEMM on v_mfma_f32_32x32x16_bf16 (gfx950).
//
//   dW[tap][ci][co] += scale * sum_m  Xg[m, tap][ci] * dY[m][co]          m = output pixel
// GEMM view per tap: rows = ci (MFMA A operand), cols = co (MFMA B operand, on the lanes),
// reduction = pixels.  Both operands live in HBM as [pixel][channel] (NHWC) but the MFMA wants 8
// consecutive *pixels* per lane, so tiles are staged [pixel][32 channels] (64-byte rows, which makes
// every 32-lane half of a transposed read cover 256 contiguous bytes = all 64 banks once) and read
// with ds_read_b64_tr_b16 -- the hardware transpose, no shuffles.
// The reduction over N*H*W pixels is split over blocks; partial tiles are combined with fp32
// atomics into dW (caller zeroes / accumulates).
// Replaces Conv2DBackpropFilter for common/ops/conv2d.py:180-187; the gather fuses NN-upsample,
// relu, the mean-pool gradient (dy stored at half size) and the stride-2 form used by Deconv2D.
#include "gank_common.h"
#include "label_conv_dev.h"
#include <type_traits>
#include <stdlib.h>

#define WG_X_ZEROINS2X 16  // (unused here; kept aligned with conv_igemm flags)
#define WG_X_STRIDE2 32

struct WgradArgs {
  const bf16* x;
  const bf16* dy;
  float* dw;
  float* dbias;       // optional: dbias[co] += scale * sum_pixels dy[p][co] (fused bias gradient)
  float* ws;          // optional workspace for split partial slabs (all-taps kernel)
  long ws_elems;
  int N, H, W;        // pixel grid the reduction runs over (conv output size)
  int Hx, Wx;         // stored x spatial
  int Hdy, Wdy;       // stored dy spatial
  int Cin, Cout;
  int ks, pad, taps;
  int M;
  int flags;
  float scale;
  int tiles_ci, tiles_co, splits, steps_per_split;
  int shw, sw;
  // block order (filter-row and all-taps kernels): 1 = XCD-aware -- one-dimensional grid, xcd_remap, (channel tile, filter row)
  // fastest, so the workgroups that read the SAME pixel range of x and dy (every tile of one split of one layer) have
  // consecutive logical ids = share an XCD's L2; 0 = the split index fastest (tiles of a split land on all eight XCDs unless
  // the split count happens to be a multiple of 8, and every XCD then pulls the whole operand through the fabric)
  int xcd;
  int dbg;            // TUNING builds only (GANK_WGRAD_DBG): 1 = no output (timing of everything but the partial-tile stores / atomics), 2 = one step per block; rows kernel: 4 = no MFMA section, 8 = no global loads in the loop, 16 = no LDS stores, 32 = no barrier
  // batched launch of up to 4 same-shape layers (gank_conv2d_wgrad_batched): blockIdx.y (xcd: the logical id) picks the operand set
  int nbatch;
  const bf16* xs[4];
  const bf16* dys[4];
  float* dws[4];
  float* dbs[4];
  float* wss[4];      // slab workspaces of a batched launch (filter-row kernel; all null = atomics into dws)
  // split-K slabs of the per-tap / lean / packed kernels (gank_conv2d_wgrad_slabs): block `split` stores its tiles into
  // slab[split] = slab + split * slab_stride (a private copy of dw, every element written exactly once) instead of adding them to
  // dw with fp32 atomics; the caller sums the slabs later (gank_sum_slabs, job filled by the launcher).  Host-side request:
  float* slab;
  long slab_stride;
  float* slab_ws;     // (host) workspace offered by the caller
  long slab_elems;
  gank_slab_job* slab_job;   // (host) out: the job that sums what the launch wrote; nslabs = 0 when the launch used atomics
  // rider of the all-taps launch (gank_conv2d_wgrad_slabs_rows_tap_sums): rider_blocks extra workgroups behind the main_blocks of
  // the filter gradient compute the per-label tap sums of the same dy (label_conv_dev.h) -- no launch of their own
  const int* rider_lists;
  float* rider_S;
  int rider_blocks, rider_V, main_blocks;
  // rider of the lean launch on a 1x1 layer (gank_conv1x1_wgrad_dgrad): d1_blocks extra workgroups compute the layer's INPUT gradient
  // dx[p][ci] = sum_co dy[p][co] wd[ci][co] (32 pixels x all input channels each) -- the two launches of a 1x1 conv's backward become one
  const bf16* d1_wd;  // bf16 [Cin][d1_pitch], rows = input channels (the plain-conv dgrad operand)
  bf16* d1_dx;        // [M][Cin]
  int d1_pitch, d1_blocks;
};

// one rider block of the 1x1 input gradient: 4 waves, wave w takes the 64-channel groups w, w + 4, ... (two 32-row A tiles each) of the
// block's 32 pixels; A fragments straight from wd (lane (r, h): 16 bytes of row r), B fragments straight from dy (lane (r, h): 16 bytes of
// pixel r) -- every load of a group is requested before its first MFMA; Cout <= 128 (8 K-steps), Cin % 64 == 0, M % 32 == 0
__device__ __forceinline__ void dgrad1x1_rider_block(const WgradArgs& a, int block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = lane & 31, h = lane >> 5;
  const long px = (long)block * 32 + r;
  bf16x8 fb[8];
#pragma unroll
  for (int kk = 0; kk < 8; kk++)
    if (kk * 16 < a.Cout) fb[kk] = *reinterpret_cast<const bf16x8*>(a.dy + px * a.Cout + kk * 16 + h * 8);
  for (int cgp = wave; cgp < (a.Cin >> 6); cgp += 4) {
    bf16x8 fa[2][8];
#pragma unroll
    for (int t = 0; t < 2; t++)
#pragma unroll
      for (int kk = 0; kk < 8; kk++)
        if (kk * 16 < a.Cout) fa[t][kk] = *reinterpret_cast<const bf16x8*>(a.d1_wd + (long)(cgp * 64 + t * 32 + r) * a.d1_pitch + kk * 16 + h * 8);
#pragma unroll
    for (int t = 0; t < 2; t++) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; e++) acc[e] = 0.f;
#pragma unroll
      for (int kk = 0; kk < 8; kk++)
        if (kk * 16 < a.Cout) acc = GANK_MFMA32(fa[t][kk], fb[kk], acc);
#pragma unroll
      for (int q = 0; q < 2; q++) {
        float v[8];
        acc_widen(acc, q, 1.0f, v);
        bf16x8 o;
#pragma unroll
        for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
        *reinterpret_cast<bf16x8*>(a.d1_dx + px * a.Cin + cgp * 64 + t * 32 + 16 * q + 8 * h) = o;
      }
    }
  }
}
static thread_local bool g_d1_carried = false;       // (host) set by the lean launcher when it carried the rider

__device__ __forceinline__ s16x4 lds_tr_read(const bf16* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
}

// WA x WB waves; each wave TA x TB MFMA tiles of 32(ci) x 32(co); 64 pixels per step.
// FAST (Cin%8==0 && Cout%8==0): 16-byte UNCONDITIONAL loads (out-of-range -> dummy address, zeroed at
// LDS-store time) into a PF-deep register ring, so loads for step s+PF-1 are in flight during step s and
// hipcc can emit counted vmcnt waits (a branch around a load would force a full drain every step).
// !FAST: element-wise predicated gather for the narrow operands (3-channel image side, Cout=1/3, K=300).
template <int WA, int WB, int TA, int TB, bool FAST, int PF>
__global__ __launch_bounds__(WA* WB * 64) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int NT = WA * WB * 64;
  constexpr int CiT = WA * TA * 32, CoT = WB * TB * 32;
  constexpr int SUBA = CiT / 32, SUBB = CoT / 32;     // 32-channel sub-tiles, each [64 pixels][32 ch] = 4 KB
  constexpr int CHA = 64 * CiT / 8, CHB = 64 * CoT / 8;  // 16-byte chunks per step
  constexpr int CPA = (CHA + NT - 1) / NT, CPB = (CHB + NT - 1) / NT;
  static_assert(FAST || PF == 1, "the narrow path consumes its loads immediately");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sA = reinterpret_cast<bf16*>(smem);          // [2][SUBA][64][32]
  bf16* sB = sA + 2 * SUBA * 2048;                   // [2][SUBB][64][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_a = wave % WA, wave_b = wave / WA;

  int bid = blockIdx.x;
  const int split = bid % a.splits; bid /= a.splits;
  const int tco = bid % a.tiles_co; bid /= a.tiles_co;
  const int tci = bid % a.tiles_ci; bid /= a.tiles_ci;
  const int tap = bid;
  const int dh = tap / a.ks - a.pad, dw = tap % a.ks - a.pad;
  const int ci0 = tci * CiT, co0 = tco * CoT;

  const bool xup = (a.flags & GANK_IN_UPSAMPLE2X) != 0;
  const bool xrelu = (a.flags & GANK_IN_RELU) != 0;
  const bool dyup = (a.flags & GANK_DY_UPSAMPLE2X) != 0;
  const int st = (a.flags & WG_X_STRIDE2) ? 2 : 1;
  const int LH = xup ? 2 * a.Hx : a.Hx, LW = xup ? 2 * a.Wx : a.Wx;
  const bool do_bias = a.dbias != nullptr && tap == 0 && tci == 0;

  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M + 63) / 64 - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;   // uniform per block; cannot happen with the host's split computation

  u32x4 rA[PF][CPA], rB[PF][CPB];
  unsigned okA[PF], okB[PF];
  float bsum[CPB][8];
#pragma unroll
  for (int j = 0; j < CPB; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bsum[j][e] = 0.f;

  auto load_step = [&](int s, u32x4 (&rA)[CPA], u32x4 (&rB)[CPB], unsigned& oka, unsigned& okb) {
    const int mbase = (step0 + s) * 64;
    oka = 0u; okb = 0u;
#pragma unroll
    for (int j = 0; j < CPA; j++) {
      const int q = tid + NT * j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (CHA % NT == 0 || q < CHA) {
        const int p = q / (CiT / 8), cc = q % (CiT / 8);
        const int m = mbase + p;
        const int c = ci0 + cc * 8;
        int n, oh, ow;
        pix_decomp(m, a.H, a.W, a.shw, a.sw, n, oh, ow);
        int ih = oh * st + dh, iw = ow * st + dw;
        const bool ok = m < a.M && c < a.Cin && (unsigned)ih < (unsigned)LH && (unsigned)iw < (unsigned)LW;
        if (xup) { ih >>= 1; iw >>= 1; }
        if constexpr (FAST) {
          const long off = ok ? ((long)(n * a.Hx + ih) * a.Wx + iw) * a.Cin + c : 0L;
          v = *reinterpret_cast<const u32x4*>(a.x + off);
          oka |= (ok ? 1u : 0u) << j;
        } else if (ok) {
          const long off = ((long)(n * a.Hx + ih) * a.Wx + iw) * a.Cin + c;
          bf16x8 t;
#pragma unroll
          for (int e = 0; e < 8; e++) t[e] = (c + e < a.Cin) ? a.x[off + e] : f2bf(0.f);
          v = __builtin_bit_cast(u32x4, t);
          oka |= 1u << j;
        }
      }
      rA[j] = v;
    }
#pragma unroll
    for (int j = 0; j < CPB; j++) {
      const int q = tid + NT * j;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (CHB % NT == 0 || q < CHB) {
        const int p = q / (CoT / 8), cc = q % (CoT / 8);
        const int m = mbase + p;
        const int c = co0 + cc * 8;
        int n, oh, ow;
        pix_decomp(m, a.H, a.W, a.shw, a.sw, n, oh, ow);
        if (dyup) { oh >>= 1; ow >>= 1; }
        const bool ok = m < a.M && c < a.Cout;
        if constexpr (FAST) {
          const long off = ok ? ((long)(n * a.Hdy + oh) * a.Wdy + ow) * a.Cout + c : 0L;
          v = *reinterpret_cast<const u32x4*>(a.dy + off);
          okb |= (ok ? 1u : 0u) << j;
        } else if (ok) {
          const long off = ((long)(n * a.Hdy + oh) * a.Wdy + ow) * a.Cout + c;
          bf16x8 t;
#pragma unroll
          for (int e = 0; e < 8; e++) t[e] = (c + e < a.Cout) ? a.dy[off + e] : f2bf(0.f);
          v = __builtin_bit_cast(u32x4, t);
          okb |= 1u << j;
        }
      }
      rB[j] = v;
    }
  };

  auto store_step = [&](int buf, u32x4 (&rA)[CPA], u32x4 (&rB)[CPB], unsigned oka, unsigned okb) {
    const u32x4 z = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int j = 0; j < CPA; j++) {
      const int q = tid + NT * j;
      if (CHA % NT == 0 || q < CHA) {
        const int p = q / (CiT / 8), cc = q % (CiT / 8);
        u32x4 v = ((oka >> j) & 1u) ? rA[j] : z;
        if (xrelu) v = relu_bf16x8(v);
        *reinterpret_cast<u32x4*>(sA + ((buf * SUBA + (cc >> 2)) * 64 + p) * 32 + (cc & 3) * 8) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < CPB; j++) {
      const int q = tid + NT * j;
      if (CHB % NT == 0 || q < CHB) {
        const int p = q / (CoT / 8), cc = q % (CoT / 8);
        const u32x4 v = ((okb >> j) & 1u) ? rB[j] : z;
        *reinterpret_cast<u32x4*>(sB + ((buf * SUBB + (cc >> 2)) * 64 + p) * 32 + (cc & 3) * 8) = v;
        if (do_bias) {   // bias gradient = column sums of dy, fused into the pass that already streams dy
          const bf16x8 t = __builtin_bit_cast(bf16x8, v);
#pragma unroll
          for (int e = 0; e < 8; e++) bsum[j][e] += bf2f(t[e]);
        }
      }
    }
  };

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  // transposed-read lane geometry: 16-lane group g=(lane>>4): channel half g&1, k half g>>1;
  // lane i of the group supplies row (pixel) i>>2, channels 4*(i&3).. of the 4x16 block
  const int g = lane >> 4, li = lane & 15;
  const int tr_off = ((8 * (g >> 1) + (li >> 2)) * 32) + 16 * (g & 1) + 4 * (li & 3);  // in bf16 elements

  const int last = nsteps - 1;
#pragma unroll
  for (int d = 0; d < PF; d++) load_step(d < last ? d : last, rA[d], rB[d], okA[d], okB[d]);
  store_step(0, rA[0], rB[0], okA[0], okB[0]);
  __syncthreads();

  for (int s0 = 0; s0 < nsteps; s0 += PF) {
#pragma unroll
    for (int d = 0; d < PF; d++) {
      const int s = s0 + d;
      if (s >= nsteps) break;
      const int buf = s & 1;
      if constexpr (PF > 1) load_step(s + PF < last ? s + PF : last, rA[d], rB[d], okA[d], okB[d]);
      const bf16* pA = sA + (buf * SUBA + wave_a * TA) * 2048 + tr_off;
      const bf16* pB = sB + (buf * SUBB + wave_b * TB) * 2048 + tr_off;
#pragma unroll
      for (int kk = 0; kk < 4; kk++) {
        bf16x8 fa[TA], fb[TB];
#pragma unroll
        for (int i = 0; i < TA; i++) {
          const s16x4 lo = lds_tr_read(pA + i * 2048 + kk * 16 * 32);
          const s16x4 hi = lds_tr_read(pA + i * 2048 + kk * 16 * 32 + 4 * 32);
          const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fa[i] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int j = 0; j < TB; j++) {
          const s16x4 lo = lds_tr_read(pB + j * 2048 + kk * 16 * 32);
          const s16x4 hi = lds_tr_read(pB + j * 2048 + kk * 16 * 32 + 4 * 32);
          const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          fb[j] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int i = 0; i < TA; i++)
#pragma unroll
          for (int j = 0; j < TB; j++)
            acc[i][j] = GANK_MFMA32(fa[i], fb[j], acc[i][j]);
      }
      if (s + 1 < nsteps) {
        if constexpr (PF == 1) load_step(s + 1, rA[0], rB[0], okA[0], okB[0]);
        store_step(buf ^ 1, rA[(d + 1) % PF], rB[(d + 1) % PF], okA[(d + 1) % PF], okB[(d + 1) % PF]);
      }
      __syncthreads();
    }
  }

  // D[i][j]: row i = ci (reg&3)+8*(reg>>2)+4*(lane>>5), col j = co = lane&31
  const int r = lane & 31, h = lane >> 5;
#ifdef GANK_TUNING
  if (a.dbg & 16) { if (acc[0][0][0] == 123.456f) a.dw[0] = 1.f; return; }
#endif
#pragma unroll
  for (int i = 0; i < TA; i++) {
#pragma unroll
    for (int j = 0; j < TB; j++) {
      const int co = co0 + (wave_b * TB + j) * 32 + r;
      if (co >= a.Cout) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ci = ci0 + (wave_a * TA + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ci < a.Cin) {
          const long o = ((long)tap * a.Cin + ci) * a.Cout + co;
          if (a.slab) a.slab[(long)split * a.slab_stride + o] = acc[i][j][e];
          else atomicAdd(a.dw + o, acc[i][j][e] * a.scale);
        }
      }
    }
  }
  if (do_bias) {
    // block-level reduction through LDS (the staging buffers are free now), then ONE atomic per channel:
    // contended same-address float atomics serialise at the memory side
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [NT][CPB*8]
#pragma unroll
    for (int j = 0; j < CPB; j++)
#pragma unroll
      for (int e = 0; e < 8; e++) red[(tid * CPB + j) * 8 + e] = bsum[j][e];
    __syncthreads();
    for (int c = tid; c < CoT; c += NT) {
      if (co0 + c >= a.Cout) continue;
      const int cc = c >> 3, e = c & 7;
      float t = 0.f;
      // chunk (p, cc) lives in thread q % NT, slot q / NT with q = p * (CoT/8) + cc
      for (int p = 0; p < 64; p++) {
        const int q = p * (CoT / 8) + cc;
        if (q < CHB) t += red[((q % NT) * CPB + q / NT) * 8 + e];
      }
      atomicAdd(a.dbias + co0 + c, t * a.scale);
    }
  }
}

// ------------------------------------------------------------------------------------------------------
// Lean variant for the layers that carry the FLOPs: Cin%8==0, Cout%8==0, power-of-two H and W, stride 1,
// M%64==0.  The general kernel above spends ~500 VALU instructions per 16 MFMAs on gather arithmetic
// (rocprof: 20 VALU per MFMA, VALU-issue-bound at 186 TFLOP/s); here the staging pass is ~10 per 16-byte
// chunk: 32-bit byte offsets into buffer loads (out-of-image -> out-of-range offset -> hardware returns
// zeros), pixel coordinates by shift/mask, compile-time MODE (bit0 relu(x), bit1 x stored at half size,
// bit2 dy stored at half size), PF-slot register ring with no exits inside the unrolled body.
// Sub-tiles are 64 B apart mod 256 B so the ds_write_b128 of one pixel's 128 channels is conflict-free.
// ------------------------------------------------------------------------------------------------------
// Launchers of the atomics-based kernels: take the caller's slab workspace when one is offered and the launch's splits fit it.
static void wgrad_slab_setup(WgradArgs& a) {
  a.slab = nullptr;
  a.slab_stride = 0;
  const long dw_elems = (long)a.taps * a.Cin * a.Cout;
  if (a.slab_job) *a.slab_job = gank_slab_job{nullptr, a.dw, dw_elems, dw_elems, 0, a.scale, 0};
  if (!a.slab_ws || !a.slab_job || a.nbatch > 0 || a.splits < 2 || (long)a.splits * dw_elems > a.slab_elems) return;
  a.slab = a.slab_ws;
  a.slab_stride = dw_elems;
  a.slab_job->slabs = a.slab_ws;
  a.slab_job->nslabs = a.splits;
}
static int wgrad_xcd_env() {
  static const int v = gank_tune("GANK_WGRAD_XCD", 1);   // experiment knob: 0 = split-fastest block order without the XCD remap (WgradArgs.xcd)
  return v;
}
static int wgrad_round_down_env() {
  static const int v = gank_tune("GANK_WGRAD_ROUND_DOWN", 1);   // experiment knob: GANK_WGRAD_ROUND_DOWN=0 restores ceil(target / tiles) pixel splits
  return v;
}
constexpr int SUBS = 2048 + 32;   // sub-tile stride in bf16 elements

template <int WA, int WB, int TA, int TB, int PF, int MODE>
__global__ __launch_bounds__(WA* WB * 64) void conv_wgrad_lean_kernel(WgradArgs a) {
  constexpr int NT = WA * WB * 64;
  constexpr int CiT = WA * TA * 32, CoT = WB * TB * 32;
  constexpr int SUBA = CiT / 32, SUBB = CoT / 32;
  constexpr int CHA = 64 * CiT / 8, CHB = 64 * CoT / 8;
  static_assert(CHA % NT == 0 && CHB % NT == 0, "tile/threads mismatch");
  constexpr int CPA = CHA / NT, CPB = CHB / NT;
  constexpr bool XRELU = (MODE & 1) != 0, XUP = (MODE & 2) != 0, DYUP = (MODE & 4) != 0;
  constexpr bool XS2 = (MODE & 8) != 0;   // x is read at (2*oh+dh, 2*ow+dw): stride-2 conv (pooled ConvMeanPool form)
  static_assert(!(XS2 && (XUP || DYUP)), "stride-2 x excludes the 2x flags");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sA = reinterpret_cast<bf16*>(smem);          // [2][SUBA] sub-tiles of [64 pixels][32 ch]
  bf16* sB = sA + 2 * SUBA * SUBS;                   // [2][SUBB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_a = wave % WA, wave_b = wave / WA;
  if constexpr (NT == 256 && TA == 1 && TB == 1 && MODE == 0) {      // (the instantiation the 1x1 shortcuts run on: the rider's registers stay out of the others)
    if (a.d1_blocks > 0 && (int)blockIdx.x >= a.main_blocks) {      // rider: the 1x1 layer's input gradient (no barrier on this path)
      dgrad1x1_rider_block(a, blockIdx.x - a.main_blocks);
      return;
    }
  }
  // operand set of this block: the single layer, or layer blockIdx.y of a same-shape batch.  Static indices only: a
  // dynamically indexed (or modified) by-value argument struct is spilled to scratch and every field read pays for it
  // (measured: every instantiation of this kernel ran 2x slower).
  const bf16* X = a.x;
  const bf16* DY = a.dy;
  float* DW = a.dw;
  float* DB = a.dbias;
  if (a.nbatch > 0) {
    const int bi = blockIdx.y;
    X = bi == 0 ? a.xs[0] : (bi == 1 ? a.xs[1] : (bi == 2 ? a.xs[2] : a.xs[3]));
    DY = bi == 0 ? a.dys[0] : (bi == 1 ? a.dys[1] : (bi == 2 ? a.dys[2] : a.dys[3]));
    DW = bi == 0 ? a.dws[0] : (bi == 1 ? a.dws[1] : (bi == 2 ? a.dws[2] : a.dws[3]));
    DB = bi == 0 ? a.dbs[0] : (bi == 1 ? a.dbs[1] : (bi == 2 ? a.dbs[2] : a.dbs[3]));
  }

  int bid = blockIdx.x;
  const int split = bid % a.splits; bid /= a.splits;
  const int tco = bid % a.tiles_co; bid /= a.tiles_co;
  const int tci = bid % a.tiles_ci; bid /= a.tiles_ci;
  const int tap = bid;
  const int dh = tap / a.ks - a.pad, dw = tap % a.ks - a.pad;
  const int ci0 = tci * CiT, co0 = tco * CoT;
  const bool do_bias = DB != nullptr && tap == 0 && tci == 0;

  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M >> 6) - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;

  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(X), 0, a.N * a.Hx * a.Wx * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<bf16*>(DY), 0, a.N * a.Hdy * a.Wdy * a.Cout * 2, 0x00020000);
  const int Wm = a.W - 1, Hm = a.H - 1, sw = a.sw, shw = a.shw;

  // per-thread chunk constants: pixel-in-step p, channel byte offset (or OOB when the chunk is past C)
  int a_p[CPA], a_c[CPA], b_p[CPB], b_c[CPB], a_lds[CPA], b_lds[CPB];
#pragma unroll
  for (int j = 0; j < CPA; j++) {
    const int q = tid + NT * j, p = q / (CiT / 8), cc = q % (CiT / 8);
    a_p[j] = p;
    a_c[j] = (ci0 + cc * 8 < a.Cin) ? (ci0 + cc * 8) * 2 : OOB;
    a_lds[j] = (cc >> 2) * SUBS + p * 32 + (cc & 3) * 8;
  }
#pragma unroll
  for (int j = 0; j < CPB; j++) {
    const int q = tid + NT * j, p = q / (CoT / 8), cc = q % (CoT / 8);
    b_p[j] = p;
    b_c[j] = (co0 + cc * 8 < a.Cout) ? (co0 + cc * 8) * 2 : OOB;
    b_lds[j] = (cc >> 2) * SUBS + p * 32 + (cc & 3) * 8;
  }

  u32x4 rA[PF][CPA], rB[PF][CPB];
  float bsum[CPB][8];
#pragma unroll
  for (int j = 0; j < CPB; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bsum[j][e] = 0.f;

  const int last = nsteps - 1;
  int cur = 0;   // step cursor; parks on the last step so trailing ring refills are harmless re-loads

  auto load_next = [&](u32x4 (&rA)[CPA], u32x4 (&rB)[CPB]) {
    const int mbase = (step0 + cur) << 6;
#pragma unroll
    for (int j = 0; j < CPA; j++) {
      const int m = mbase + a_p[j];
      const int ow = m & Wm, t = m >> sw, oh = t & Hm;
      const int ih = XS2 ? 2 * oh + dh : oh + dh, iw = XS2 ? 2 * ow + dw : ow + dw;
      const bool ok = XS2 ? ((unsigned)ih < (unsigned)a.Hx && (unsigned)iw < (unsigned)a.Wx)
                          : ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W);
      int off;
      if constexpr (XS2) {
        const int n = m >> shw;
        off = ((n * a.Hx + ih) * a.Wx + iw) * a.Cin * 2 + a_c[j];
      } else if constexpr (XUP) {
        const int n = m >> shw;
        off = ((n * a.Hx + (ih >> 1)) * a.Wx + (iw >> 1)) * a.Cin * 2 + a_c[j];
      } else {
        off = (m + dh * a.W + dw) * a.Cin * 2 + a_c[j];
      }
      rA[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < CPB; j++) {
      const int m = mbase + b_p[j];
      int off;
      if constexpr (DYUP) {
        const int ow = m & Wm, t = m >> sw, oh = t & Hm, n = m >> shw;
        off = ((n * a.Hdy + (oh >> 1)) * a.Wdy + (ow >> 1)) * a.Cout * 2 + b_c[j];
      } else {
        off = m * a.Cout * 2 + b_c[j];
      }
      rB[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0);   // b_c == OOB pushes it out of range
    }
    if (cur < last) cur++;
  };

  auto store_step = [&](int buf, u32x4 (&rA)[CPA], u32x4 (&rB)[CPB]) {
#pragma unroll
    for (int j = 0; j < CPA; j++) {
      u32x4 v = rA[j];
      if constexpr (XRELU) v = relu_bf16x8(v);
      *reinterpret_cast<u32x4*>(sA + buf * SUBA * SUBS + a_lds[j]) = v;
    }
#pragma unroll
    for (int j = 0; j < CPB; j++) {
      *reinterpret_cast<u32x4*>(sB + buf * SUBB * SUBS + b_lds[j]) = rB[j];
      if (do_bias) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, rB[j]);
#pragma unroll
        for (int e = 0; e < 8; e++) bsum[j][e] += bf2f(t[e]);
      }
    }
  };

  f32x16 acc[TA][TB];
#pragma unroll
  for (int i = 0; i < TA; i++)
#pragma unroll
    for (int j = 0; j < TB; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.f;

  const int g = lane >> 4, li = lane & 15;
  const int tr_off = ((8 * (g >> 1) + (li >> 2)) * 32) + 16 * (g & 1) + 4 * (li & 3);

#pragma unroll
  for (int d = 0; d < PF; d++) load_next(rA[d], rB[d]);
  store_step(0, rA[0], rB[0]);
  __syncthreads();

  auto step = [&](int s, auto slot) {
    constexpr int D = decltype(slot)::value;
    const int buf = s & 1;
    if constexpr (PF > 1) load_next(rA[D], rB[D]);
    const bf16* pA = sA + (buf * SUBA + wave_a * TA) * SUBS + tr_off;
    const bf16* pB = sB + (buf * SUBB + wave_b * TB) * SUBS + tr_off;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      bf16x8 fa[TA], fb[TB];
#pragma unroll
      for (int i = 0; i < TA; i++) {
        const s16x4 lo = lds_tr_read(pA + i * SUBS + kk * 16 * 32);
        const s16x4 hi = lds_tr_read(pA + i * SUBS + kk * 16 * 32 + 4 * 32);
        const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fa[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int j = 0; j < TB; j++) {
        const s16x4 lo = lds_tr_read(pB + j * SUBS + kk * 16 * 32);
        const s16x4 hi = lds_tr_read(pB + j * SUBS + kk * 16 * 32 + 4 * 32);
        const s16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        fb[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < TA; i++)
#pragma unroll
        for (int j = 0; j < TB; j++)
          acc[i][j] = GANK_MFMA32(fa[i], fb[j], acc[i][j]);
    }
    if (s + 1 < nsteps) {
      if constexpr (PF == 1) load_next(rA[0], rB[0]);
      store_step(buf ^ 1, rA[(D + 1) % PF], rB[(D + 1) % PF]);
    }
    __syncthreads();
  };

  int s0 = 0;
  for (; s0 + PF <= nsteps; s0 += PF) {
    if constexpr (PF >= 1) step(s0 + 0, std::integral_constant<int, 0>{});
    if constexpr (PF >= 2) step(s0 + 1, std::integral_constant<int, 1 % PF>{});
    if constexpr (PF >= 3) step(s0 + 2, std::integral_constant<int, 2 % PF>{});
    if constexpr (PF >= 4) step(s0 + 3, std::integral_constant<int, 3 % PF>{});
  }
  if constexpr (PF >= 2) { if (s0 + 0 < nsteps) step(s0 + 0, std::integral_constant<int, 0>{}); }
  if constexpr (PF >= 3) { if (s0 + 1 < nsteps) step(s0 + 1, std::integral_constant<int, 1 % PF>{}); }
  if constexpr (PF >= 4) { if (s0 + 2 < nsteps) step(s0 + 2, std::integral_constant<int, 2 % PF>{}); }

  const int r = lane & 31, h = lane >> 5;
#ifdef GANK_TUNING
  if (a.dbg & 16) { if (acc[0][0][0] == 123.456f) DW[0] = 1.f; return; }
#endif
#pragma unroll
  for (int i = 0; i < TA; i++) {
#pragma unroll
    for (int j = 0; j < TB; j++) {
      const int co = co0 + (wave_b * TB + j) * 32 + r;
      if (co >= a.Cout) continue;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ci = ci0 + (wave_a * TA + i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (ci < a.Cin) {
          const long o = ((long)tap * a.Cin + ci) * a.Cout + co;
          if (a.slab) a.slab[(long)split * a.slab_stride + o] = acc[i][j][e];
          else atomicAdd(DW + o, acc[i][j][e] * a.scale);
        }
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [NT][CPB*8]
#pragma unroll
    for (int j = 0; j < CPB; j++)
#pragma unroll
      for (int e = 0; e < 8; e++) red[(tid * CPB + j) * 8 + e] = bsum[j][e];
    __syncthreads();
    for (int c = tid; c < CoT; c += NT) {
      if (co0 + c >= a.Cout) continue;
      const int cc = c >> 3, e = c & 7;
      float t = 0.f;
      for (int p = 0; p < 64; p++) {
        const int q = p * (CoT / 8) + cc;
        t += red[((q % NT) * CPB + q / NT) * 8 + e];
      }
      atomicAdd(DB + co0 + c, t * a.scale);
    }
  }
}

template <int WA, int WB, int TA, int TB, int PF, int MODE>
static int launch_wgrad_lean_mode(WgradArgs a, hipStream_t s) {
  constexpr int CiT = WA * TA * 32, CoT = WB * TB * 32;
  a.tiles_ci = cdiv(a.Cin, CiT);
  a.tiles_co = cdiv(a.Cout, CoT);
  const int total_steps = a.M / 64;
  const long tiles = (long)a.taps * a.tiles_ci * a.tiles_co * (a.nbatch > 0 ? a.nbatch : 1);   // grid.y layers fill the chip too
  static const int target_env = gank_tune("GANK_WGRAD_SPLIT_TARGET", 256), minsteps_env = gank_tune("GANK_WGRAD_MIN_STEPS", 8);
  int splits = (int)((target_env + tiles - 1) / tiles);
  if (splits > total_steps / minsteps_env) splits = total_steps / minsteps_env;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
  size_t lds = (size_t)2 * (CiT / 32 + CoT / 32) * SUBS * sizeof(bf16);
  const size_t red = (size_t)WA * WB * 64 * (64 * CoT / 8 / (WA * WB * 64)) * 8 * sizeof(float);
  if (red > lds) lds = red;
  auto kern = conv_wgrad_lean_kernel<WA, WB, TA, TB, PF, MODE>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_wgrad");
  const long grid = tiles / (a.nbatch > 0 ? a.nbatch : 1) * a.splits;
  GANK_REQUIRE(grid < (1L << 30), "conv_wgrad: grid too large");
  static const std::string tag = gank_format("conv_wgrad_lean_kernel<%d, %d, %d, %d, %d, %d>", WA, WB, TA, TB, PF, MODE);     // magic static: built once, thread-safe
  gank_prof_tag(1, tag.c_str());
  wgrad_slab_setup(a);
  { static const int dbg_ = gank_tune("GANK_WGRAD_DBG", 0); a.dbg = dbg_; }
  long gx = grid;
  a.d1_blocks = 0;
  if (a.d1_dx && a.nbatch <= 0 && WA * WB * 64 == 256 && TA == 1 && TB == 1 && MODE == 0 && a.ks == 1) {
    a.main_blocks = (int)grid;
    a.d1_blocks = a.M / 32;
    gx += a.d1_blocks;
    g_d1_carried = true;
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)gx, a.nbatch > 0 ? a.nbatch : 1), dim3(WA * WB * 64), lds, s, a);
  GANK_LAUNCH_OK("conv_wgrad_lean");
  return 0;
}

template <int WA, int WB, int TA, int TB, int PF>
static int launch_wgrad_lean(const WgradArgs& a, hipStream_t s) {
  const int mode = ((a.flags & GANK_IN_RELU) ? 1 : 0) | ((a.flags & GANK_IN_UPSAMPLE2X) ? 2 : 0) |
                   ((a.flags & GANK_DY_UPSAMPLE2X) ? 4 : 0) | ((a.flags & WG_X_STRIDE2) ? 8 : 0);
  switch (mode) {
    case 8: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 8>(a, s);
    case 9: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 9>(a, s);
    case 0: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 0>(a, s);
    case 1: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 1>(a, s);
    case 2: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 2>(a, s);
    case 4: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 4>(a, s);
    case 5: return launch_wgrad_lean_mode<WA, WB, TA, TB, PF, 5>(a, s);
    default: return -1;   // combination not instantiated: caller falls back to the general kernel
  }
}

// ------------------------------------------------------------------------------------------------------
// Packed-narrow variant for the image-side layers (Cin=3: D.Block.1.Conv1/Shortcut) and the 3- or
// 1-channel outputs (G.Output, D.Output).  The per-tap kernels above re-read the WIDE operand once per tap
// (9 x 33 MB for D.Block.1.Conv1); here the narrow operand is im2col-packed on the fly into <= 32 columns
// k = (tap, c) -- x[p+tap][ci] when PACK_X, dy[p-tap][co] otherwise -- so the wide operand streams through
// ONCE and all taps come out of one MFMA pass.  4 waves, each a 32x32 tile along the wide channel axis.
// ------------------------------------------------------------------------------------------------------
template <bool PACK_X>
__global__ __launch_bounds__(256) void conv_wgrad_packed_kernel(WgradArgs a) {
  constexpr int NT = 256;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sN = reinterpret_cast<bf16*>(smem);        // narrow operand [2][1 sub-tile]
  bf16* sWd = sN + 2 * SUBS;                       // wide operand   [2][4 sub-tiles]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int split = blockIdx.x % a.splits, tw = blockIdx.x / a.splits;   // wide-channel tile
  const int Cn = PACK_X ? a.Cin : a.Cout;          // narrow channels (<= 4)
  const int Cw = PACK_X ? a.Cout : a.Cin;          // wide channels (% 8 == 0)
  const bf16* pn = PACK_X ? a.x : a.dy;
  const bf16* pw = PACK_X ? a.dy : a.x;
  const int w0 = tw * 128;
  const int kcols = a.taps * Cn;                   // <= 32
  const int sgn = PACK_X ? 1 : -1;

  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M + 63) / 64 - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;

  const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(pw), 0, a.M * Cw * 2, 0x00020000);
  constexpr int OOB = 0x7FFFFFF0;
  // narrow chunk of this thread: pixel np, columns ncc*8 .. +7 ; per-column tap offsets and channel
  const int np = tid >> 2, ncc = tid & 3;
  int k_dh[8], k_dw[8], k_c[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = ncc * 8 + e;
    const int tap = k / Cn;
    k_c[e] = (k < kcols) ? k - tap * Cn : -1;
    k_dh[e] = sgn * (tap / a.ks - a.pad);
    k_dw[e] = sgn * (tap % a.ks - a.pad);
  }
  // wide chunks: 4 per thread
  int w_p[4], w_c[4], w_lds[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int q = tid + NT * j, p = q >> 4, cc = q & 15;
    w_p[j] = p;
    w_c[j] = (w0 + cc * 8 < Cw) ? (w0 + cc * 8) * 2 : OOB;
    w_lds[j] = (cc >> 2) * SUBS + p * 32 + (cc & 3) * 8;
  }
  const bool do_bias = PACK_X && a.dbias != nullptr;   // every wide-channel tile sums its own 128 columns of dy (!PACK_X: the host adds a column-sum launch)
  float bw[4][8];
#pragma unroll
  for (int j = 0; j < 4; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bw[j][e] = 0.f;

  // Register ring PFD steps deep.  The grid is one block per CU (4 waves) and a step is only 16 KB of the wide operand: one
  // step of lookahead kept 16 KB per CU in flight = 2 TB/s at the latency under load, and the kernel ran at exactly that
  // (24 us for the critic's 33.5-MB dy; MFMA pipe 1 % busy).  Loads are unconditional -- steps past the end re-load the last
  // one, out-of-image taps read element 0 and are zeroed by a select -- so the compiler's vmcnt counting sees straight-line code.
  constexpr int PFD = 3;
  u32x4 rN[PFD], rW[PFD][4];
  const int last = nsteps - 1;
  auto load_step = [&](int s, u32x4& rn, u32x4 (&rw)[4]) {
    const int mbase = (step0 + (s < last ? s : last)) * 64;
    {
      const int m = mbase + np;
      int n, oh, ow;
      pix_decomp(m < a.M ? m : 0, a.H, a.W, a.shw, a.sw, n, oh, ow);
      bf16x8 v;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const int ih = oh + k_dh[e], iw = ow + k_dw[e];
        const bool ok = m < a.M && k_c[e] >= 0 && (unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W;
        const bf16 t = pn[ok ? ((long)(n * a.H + ih) * a.W + iw) * Cn + k_c[e] : 0L];
        v[e] = ok ? t : f2bf(0.f);
      }
      rn = __builtin_bit_cast(u32x4, v);
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int m = mbase + w_p[j];
      rw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, m < a.M ? m * Cw * 2 + w_c[j] : OOB, 0, 0);
    }
  };
  auto store_step = [&](int buf, const u32x4& rn, const u32x4 (&rw)[4]) {
    *reinterpret_cast<u32x4*>(sN + buf * SUBS + np * 32 + ncc * 8) = rn;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      *reinterpret_cast<u32x4*>(sWd + buf * 4 * SUBS + w_lds[j]) = rw[j];
      if (PACK_X && do_bias) {      // bias gradient of the wide dy
        const bf16x8 t = __builtin_bit_cast(bf16x8, rw[j]);
#pragma unroll
        for (int e = 0; e < 8; e++) bw[j][e] += bf2f(t[e]);
      }
    }
  };

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  const int g = lane >> 4, li = lane & 15;
  const int tr_off = ((8 * (g >> 1) + (li >> 2)) * 32) + 16 * (g & 1) + 4 * (li & 3);

#pragma unroll
  for (int d = 0; d < PFD; d++) load_step(d, rN[d], rW[d]);     // steps 0 .. PFD-1 (clamped)
  store_step(0, rN[0], rW[0]);
  __syncthreads();
  // one step on ring slot D (compile-time): slot D held step s (already in LDS) and is refilled with step s + PFD
  auto step = [&](int s, auto slot) {
    constexpr int D = decltype(slot)::value;
    const int buf = s & 1;
    load_step(s + PFD, rN[D], rW[D]);
    const bf16* pN = sN + buf * SUBS + tr_off;
    const bf16* pW = sWd + (buf * 4 + wave) * SUBS + tr_off;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const s16x4 nl = lds_tr_read(pN + kk * 16 * 32), nh = lds_tr_read(pN + kk * 16 * 32 + 4 * 32);
      const s16x4 wl = lds_tr_read(pW + kk * 16 * 32), wh = lds_tr_read(pW + kk * 16 * 32 + 4 * 32);
      const s16x8 tn = {nl[0], nl[1], nl[2], nl[3], nh[0], nh[1], nh[2], nh[3]};
      const s16x8 tw8 = {wl[0], wl[1], wl[2], wl[3], wh[0], wh[1], wh[2], wh[3]};
      const bf16x8 fn = __builtin_bit_cast(bf16x8, tn), fw = __builtin_bit_cast(bf16x8, tw8);
      // rows (MFMA A) = ci side, cols (MFMA B, lanes) = co side
      if constexpr (PACK_X) acc = GANK_MFMA32(fn, fw, acc);
      else acc = GANK_MFMA32(fw, fn, acc);
    }
    if (s + 1 < nsteps) store_step(buf ^ 1, rN[(D + 1) % PFD], rW[(D + 1) % PFD]);
    __syncthreads();
  };
  int s0 = 0;
  for (; s0 + PFD <= nsteps; s0 += PFD) {
    step(s0 + 0, std::integral_constant<int, 0>{});
    step(s0 + 1, std::integral_constant<int, 1>{});
    step(s0 + 2, std::integral_constant<int, 2>{});
  }
  if (s0 + 0 < nsteps) step(s0 + 0, std::integral_constant<int, 0>{});
  if (s0 + 1 < nsteps) step(s0 + 1, std::integral_constant<int, 1>{});

  const int r = lane & 31, h = lane >> 5;
#ifdef GANK_TUNING
  if (a.dbg & 16) { if (acc[0] == 123.456f) a.dw[0] = 1.f; return; }
#endif
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * h;      // MFMA row index 0..31, col = r
    if constexpr (PACK_X) {
      const int k = row, co = w0 + wave * 32 + r;        // dw[(tap*Cin+ci)][co] == dw[k][co]
      if (k < kcols && co < a.Cout) {
        if (a.slab) a.slab[(long)split * a.slab_stride + (long)k * a.Cout + co] = acc[e];
        else atomicAdd(a.dw + (long)k * a.Cout + co, acc[e] * a.scale);
      }
    } else {
      const int ci = w0 + wave * 32 + row, k = r;        // k = tap*Cout + co
      if (k < kcols && ci < a.Cin) {
        const int tap = k / a.Cout, co = k - tap * a.Cout;
        const long o = ((long)tap * a.Cin + ci) * a.Cout + co;
        if (a.slab) a.slab[(long)split * a.slab_stride + o] = acc[e];
        else atomicAdd(a.dw + o, acc[e] * a.scale);
      }
    }
  }
  if (PACK_X && do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [256][32]
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
      for (int e = 0; e < 8; e++) red[(tid * 4 + j) * 8 + e] = bw[j][e];
    __syncthreads();
    for (int c = tid; c < 128; c += NT) {
      if (w0 + c >= a.Cout) continue;
      const int cc = c >> 3, e = c & 7;
      float t = 0.f;
      for (int p = 0; p < 64; p++) {
        const int q = p * 16 + cc;
        t += red[((q % NT) * 4 + q / NT) * 8 + e];
      }
      atomicAdd(a.dbias + w0 + c, t * a.scale);
    }
  }
}

template <bool PACK_X>
static int launch_wgrad_packed(WgradArgs a, hipStream_t s) {
  const int Cw = PACK_X ? a.Cout : a.Cin;
  const int tiles = cdiv(Cw, 128);
  const int total_steps = cdiv(a.M, 64);
  static const int target = gank_tune("GANK_WGRAD_PACKED_BLOCKS", 256);   // experiment knob: GANK_WGRAD_PACKED_BLOCKS (blocks per launch; every block ends in 32 x 128 float atomics)
  int splits = target / tiles;
  static const int minsteps = gank_tune("GANK_WGRAD_PACKED_MINSTEPS", 8);   // experiment knob: GANK_WGRAD_PACKED_MINSTEPS (64-pixel steps per block, at least)
  if (splits > total_steps / minsteps) splits = total_steps / minsteps;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
  const size_t lds = (size_t)2 * 5 * SUBS * sizeof(bf16);   // 41.6 KB (>= the 32 KB bias-reduction scratch)
  auto kern = conv_wgrad_packed_kernel<PACK_X>;
  gank_prof_tag(1, PACK_X ? "conv_wgrad_packed_kernel<true>" : "conv_wgrad_packed_kernel<false>");
  wgrad_slab_setup(a);
  { static const int dbg_ = gank_tune("GANK_WGRAD_DBG", 0); a.dbg = dbg_; }
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles * a.splits)), dim3(256), lds, s, a);
  GANK_LAUNCH_OK("conv_wgrad_packed");
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// Streaming form of the narrow-input filter gradient (round 3): dW [k*k*CIN (<= 31)][Cout] = Xcol^T dY over ALL pixels,
// i.e. a [32 x M] x [M x 128] product whose whole cost is reading dy once (33.5 MB for D.Block.1.Conv1 at 128 samples).
// The packed kernel above walks it in 64-pixel steps behind a 3-deep register ring with a barrier per step and ran at
// 1.25 TB/s (27 us).  Here a block owns 512 consecutive pixels x 128 couts and requests its ENTIRE dy slice (128 KB) by
// LDS-DMA before doing anything else -- every CU has 128 KB in flight at once -- and builds the im2col operand (32 KB) from
// the L2-resident image while the slice streams in.  Wave w issues, waits for and consumes the 32-channel sub-tile w, so
// the dy side needs no barrier at all; one barrier publishes the im2col tile.  Row k*k*CIN of the operand is all ones: its
// output row is the bias gradient.  LDS: the full 160 KB (dy [4 subs][512 px][32 ch] | Xcol [512 px][32 k]).
// Two layers of different geometry can share a launch (the critic's D.Block.1.Conv1 3x3 and D.Block.1.Shortcut 1x1).
// ------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* wg_lds_t;
constexpr int NARROW_STREAM_LDS = (4 * 512 * 32 + 512 * 32) * 2;     // 160 KB
constexpr int NARROW_STAGE_OFF = NARROW_STREAM_LDS - 4096;           // the last 4 KB double as the image stage (read out before the operand tile is written)

struct NarrowWgArgs {
  const bf16* x;       // [N,H,W,CIN]
  const bf16* dy;      // [N,H,W,Cout]
  float* dw;           // [ks*ks*CIN][Cout]
  float* dbias;        // [Cout] or null
  int N, H, W, Cout, M, shw, sw;
  float scale;
  int blocks;          // ceil(M / 512) * (Cout / 128)
};

template <int N>
__device__ __forceinline__ void wg_wait_vmcnt() {
  if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// NPX = pixels per block: 512 (the whole LDS) or 128 (the second, smaller layer of a folded pair: see the kernel below)
template <int KS, int CIN, int NPX>
__device__ __forceinline__ void wgrad_narrow_stream_body(const NarrowWgArgs& a, int blk, char* smem, f32x16& out) {
  constexpr int KTOT = KS * KS * CIN, PAD = (KS - 1) / 2;
  static_assert(KTOT <= 31, "one 32-row operand tile with a row left for the bias gradient");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cgroups = a.Cout >> 7;
  const int cg = blk % cgroups, pb = blk / cgroups;
  constexpr int NPC = NPX / 16;                              // DMA pieces (= MFMA K-steps) per wave
  constexpr int NCH = NPX * 4 / 256;                         // operand chunks per thread
  const int m0 = pb * NPX, co0 = cg * 128;
  bf16* sG = reinterpret_cast<bf16*>(smem);                  // [4][NPX][32]
  bf16* sX = sG + 4 * 512 * 32;                              // [NPX][32]  (at its NPX = 512 place either way)
  char* stage = smem + NARROW_STAGE_OFF;                     // 4 KB at the end of sX: the image bytes this block's pixels touch

  // ---- (1) the image region of the block's pixels and their halo, ONE coalesced 16-byte load per thread, BEFORE any LDS-DMA is
  // in flight (beside a pending DMA hipcc drains vmcnt(0) at the first use of an ordinary load: the whole slice would land first).
  // x is [N,H,W,CIN] with CIN innermost: flat pixels m0 - W - 1 .. m0 + 512 + W are one contiguous byte range.
  const int sb = max(0, (m0 - PAD * (a.W + 1)) * CIN * 2) & ~15;        // first staged byte of x  (NPX + 2 W + 2 pixels <= 4 KB: checked by the host)
  {
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.M * CIN * 2, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, sb + 16 * tid, 0, 0);      // past the end: zeros
    *reinterpret_cast<u32x4*>(stage + 16 * tid) = v;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- (2) the dy slice: wave w requests sub-tile w (couts co0 + 32 w ..), 32 pieces of 16 pixels x 64 B
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.M * a.Cout * 2, 0x00020000);
  {
    const int voff = ((lane >> 2) * a.Cout + (lane & 3) * 8) * 2;        // pixel l/4 of the piece, 16-byte chunk l%4 of its 64 B
#pragma unroll
    for (int pc = 0; pc < NPC; pc++) {
      const int soff = ((m0 + pc * 16) * a.Cout + co0 + wave * 32) * 2;   // pixels past M fall outside the buffer: zeros
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (wg_lds_t)(smem + (wave * NPX + pc * 16) * 64), 16, voff, soff, 0, 0);
    }
  }
  __builtin_amdgcn_s_barrier();          // raw barriers from here on: __syncthreads() would drain the DMA

  // ---- (3) im2col operand from the staged bytes, while the slice streams in: chunk q = (pixel q>>2, columns 8 (q&3) .. +7),
  // 8 chunks per thread; q & 3 is the same for all of a thread's chunks, so the tap geometry of its 8 columns is computed once
  const int cc = tid & 3;
  int k_d[8], k_dh[8], k_dw[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = cc * 8 + e, tap = k / CIN, c = k - tap * CIN;
    k_dh[e] = k < KTOT ? tap / KS - PAD : 1 << 20;                      // out-of-range column: never inside the image
    k_dw[e] = tap % KS - PAD;
    k_d[e] = ((tap / KS - PAD) * a.W + tap % KS - PAD) * CIN + c;       // element offset from the pixel's first channel
  }
  const int ones_e = KTOT - cc * 8;                                     // column of the all-ones row, if it is one of this thread's
  bf16x8 ch[NCH];
#pragma unroll
  for (int j = 0; j < NCH; j++) {
    const int p = (tid >> 2) + 64 * j, m = m0 + p;
    int n, oh, ow;
    pix_decomp(m < a.M ? m : 0, a.H, a.W, a.shw, a.sw, n, oh, ow);
    const int base = m * CIN * 2 - sb;                                  // byte offset of the pixel in the stage
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const bool ok = m < a.M && (unsigned)(oh + k_dh[e]) < (unsigned)a.H && (unsigned)(ow + k_dw[e]) < (unsigned)a.W;
      const bf16 t = *reinterpret_cast<const bf16*>(stage + (ok ? base + 2 * k_d[e] : 0));
      ch[j][e] = ok ? t : f2bf((e == ones_e && m < a.M) ? 1.f : 0.f);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();          // every thread has read the stage: the operand tile may overwrite it
#pragma unroll
  for (int j = 0; j < NCH; j++) *reinterpret_cast<bf16x8*>(sX + ((tid >> 2) + 64 * j) * 32 + cc * 8) = ch[j];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; e++) acc[e] = 0.f;
  const int g = lane >> 4, li = lane & 15;
  const int tr_off = ((8 * (g >> 1) + (li >> 2)) * 32) + 16 * (g & 1) + 4 * (li & 3);
  const bf16* pX = sX + tr_off;
  const bf16* pG = sG + wave * NPX * 32 + tr_off;
  auto stage8 = [&](int s0) {
#pragma unroll
    for (int s = s0; s < s0 + 8; s++) {             // K-step s = pixels 16 s .. 16 s + 15 = DMA piece s of this wave
      const s16x4 xl = lds_tr_read(pX + s * 16 * 32), xh = lds_tr_read(pX + s * 16 * 32 + 4 * 32);
      const s16x4 gl = lds_tr_read(pG + s * 16 * 32), gh = lds_tr_read(pG + s * 16 * 32 + 4 * 32);
      const s16x8 tx = {xl[0], xl[1], xl[2], xl[3], xh[0], xh[1], xh[2], xh[3]};
      const s16x8 tg = {gl[0], gl[1], gl[2], gl[3], gh[0], gh[1], gh[2], gh[3]};
      acc = GANK_MFMA32(__builtin_bit_cast(bf16x8, tx), __builtin_bit_cast(bf16x8, tg), acc);
    }
  };
  // the wave's own pieces retire in issue order: 8 at a time
  if constexpr (NPC == 32) {
    wg_wait_vmcnt<24>(); __builtin_amdgcn_sched_barrier(0); stage8(0);
    wg_wait_vmcnt<16>(); __builtin_amdgcn_sched_barrier(0); stage8(8);
    wg_wait_vmcnt<8>();  __builtin_amdgcn_sched_barrier(0); stage8(16);
    wg_wait_vmcnt<0>();  __builtin_amdgcn_sched_barrier(0); stage8(24);
  } else {
    static_assert(NPC == 8, "512 or 128 pixels per block");
    wg_wait_vmcnt<0>();  __builtin_amdgcn_sched_barrier(0); stage8(0);
  }

  out = acc;
}

// D rows = k (tap, channel; row KTOT = column sums of dy), cols = couts co0 + 32 wave + r
template <int KS, int CIN>
__device__ __forceinline__ void wgrad_narrow_stream_epilogue(const NarrowWgArgs& a, int blk, const f32x16& acc) {
  constexpr int KTOT = KS * KS * CIN;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int co0 = (blk % (a.Cout >> 7)) * 128;
  const int r = lane & 31, h = lane >> 5;
  const int co = co0 + wave * 32 + r;
#pragma unroll
  for (int e = 0; e < 16; e++) {
    const int k = (e & 3) + 8 * (e >> 2) + 4 * h;
    if (k < KTOT) atomicAdd(a.dw + (long)k * a.Cout + co, acc[e] * a.scale);
    else if (k == KTOT && a.dbias) atomicAdd(a.dbias + co, acc[e] * a.scale);
  }
}

template <int KS0, int KS1, int CIN>
__global__ __launch_bounds__(256) void conv_wgrad_narrow_stream_kernel(NarrowWgArgs a0, NarrowWgArgs a1, int folded) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  f32x16 acc0, acc1;
  if (folded) {
    // a pair whose second layer has a quarter of the first one's pixels (the critic's Conv1 on 32x32, Shortcut on the pooled
    // 16x16): every block takes 512 pixels of the first AND then 128 of the second -- one round of blocks on the chip, where
    // 256 + 64 blocks of 160 KB LDS ran as two.  Both layers' atomics leave at the very end: issued between the layers they
    // would sit in front of the second layer's first vmcnt wait.
    wgrad_narrow_stream_body<KS0, CIN, 512>(a0, blockIdx.x, smem, acc0);
    __builtin_amdgcn_s_barrier();        // every wave is past its last LDS read of the first layer
    wgrad_narrow_stream_body<KS1, CIN, 128>(a1, blockIdx.x, smem, acc1);
    wgrad_narrow_stream_epilogue<KS0, CIN>(a0, blockIdx.x, acc0);
    wgrad_narrow_stream_epilogue<KS1, CIN>(a1, blockIdx.x, acc1);
    return;
  }
  if ((int)blockIdx.x < a0.blocks) {
    wgrad_narrow_stream_body<KS0, CIN, 512>(a0, blockIdx.x, smem, acc0);
    wgrad_narrow_stream_epilogue<KS0, CIN>(a0, blockIdx.x, acc0);
  } else {
    wgrad_narrow_stream_body<KS1, CIN, 512>(a1, blockIdx.x - a0.blocks, smem, acc1);
    wgrad_narrow_stream_epilogue<KS1, CIN>(a1, blockIdx.x - a0.blocks, acc1);
  }
}

static bool wgrad_narrow_stream_ok(int N, int H, int W, int Cin, int Cout, int ks) {
  static const int env = gank_tune("GANK_WGRAD_STREAM", 1);   // experiment knob: GANK_WGRAD_STREAM=0 keeps the packed kernel
  return env && Cin == 3 && (ks == 1 || ks == 3) && Cout % 128 == 0 && (512 + 2 * W + 2) * 6 + 16 <= 4096 && (long)N * H * W * Cout * 2 < (1L << 31) && (long)N * H * W < (1L << 30);
}
static NarrowWgArgs narrow_args(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int Cout, float scale) {
  NarrowWgArgs q{};
  q.x = (const bf16*)x; q.dy = (const bf16*)dy; q.dw = dw; q.dbias = dbias;
  q.N = N; q.H = H; q.W = W; q.Cout = Cout; q.M = N * H * W; q.sw = log2_or_neg(W); q.shw = log2_or_neg(H * W); q.scale = scale;
  q.blocks = cdiv(q.M, 512) * (Cout / 128);
  return q;
}
// one or two layers (ks1 == 0: one) with Cin == 3
static int launch_wgrad_narrow_stream(const NarrowWgArgs& a0, int ks0, const NarrowWgArgs& a1, int ks1, hipStream_t s) {
  NarrowWgArgs b1 = a1;
  if (!ks1) b1.blocks = 0;
  // fold the second layer into the first one's blocks when it has exactly a quarter of the pixels (and the same couts)
  // GANK_NARROW_FOLD=1: every block takes 512 pixels of the first layer and then 128 of the second (one round of blocks instead of
  // 256 + 64).  Isolated it is faster (scratch/micro/narrow.hip: 16.8 vs 18.6 us warm, 24.2 vs 28.3 us from cold caches); inside the
  // critic update it measured 0.5 % SLOWER per iteration (interleaved A/B on one box), so separate block ranges stay the default.
  static const int fold_env = gank_tune("GANK_NARROW_FOLD", 0);
  const int folded = fold_env && ks1 && a0.Cout == a1.Cout && a0.M == 4 * a1.M && a0.M % 512 == 0 ? 1 : 0;
  const int grid = folded ? a0.blocks : a0.blocks + b1.blocks;
#define NARROW_LAUNCH(K0, K1)                                                                                      \
  do {                                                                                                             \
    auto kern = conv_wgrad_narrow_stream_kernel<K0, K1, 3>;                                                        \
    GANK_MAX_DYNAMIC_LDS(kern, NARROW_STREAM_LDS, "conv_wgrad_narrow_stream");                                     \
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), NARROW_STREAM_LDS, s, a0, b1, folded);               \
  } while (0)
  if (ks0 == 3 && (ks1 == 1 || ks1 == 0)) NARROW_LAUNCH(3, 1);
  else if (ks0 == 1 && (ks1 == 3)) NARROW_LAUNCH(1, 3);
  else if (ks0 == 3 && ks1 == 3) NARROW_LAUNCH(3, 3);
  else NARROW_LAUNCH(1, 1);
#undef NARROW_LAUNCH
  GANK_LAUNCH_OK("conv_wgrad_narrow_stream");
  return 0;
}

// ------------------------------------------------------------------------------------------------------
// All-taps variant for 3x3 / stride 1 / power-of-two images / Cin%64==0 / Cout%64==0 -- the layers that
// carry the wgrad FLOPs.  One block owns a 64(ci) x 64(co) tile for ALL 9 taps: per step it stages ONE
// 8x8 patch of dy and the 10x10 halo of x, then every tap's A fragments are transposed reads of the same
// halo image at a shifted pixel offset (ds_read_b64_tr_b16 takes a per-lane row address, so a shifted
// window is just address arithmetic).  Versus one block per tap: 3.5x less L2->LDS traffic, 4x fewer
// ds_write, and 36 instead of 16 MFMAs per wave between barriers.  Partial tiles of the pixel splits go to
// fp32 slabs with plain coalesced stores and are summed by a second tiny kernel (deterministic; the atomic
// form would burst 9x more atomic bytes per block at the end of the kernel).
// ------------------------------------------------------------------------------------------------------
constexpr int XSUB = 100 * 32 + 32;   // halo sub-tile stride (bf16): 64 B off a 256 B multiple

template <int MODE, int PF>
__global__ __launch_bounds__(256) void conv_wgrad_taps_kernel(WgradArgs a) {
  constexpr int NT = 256;
  constexpr bool XRELU = (MODE & 1) != 0, XUP = (MODE & 2) != 0, DYUP = (MODE & 4) != 0;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sX = reinterpret_cast<bf16*>(smem);        // [2][2 subs][100 pix][32 ch]
  bf16* sD = sX + 2 * 2 * XSUB;                    // [2][2 subs][64 pix][32 ch]

  if (a.rider_blocks > 0 && (int)blockIdx.x >= a.main_blocks) {
    label_conv_tap_sums_block(a.dy, a.rider_lists, a.rider_S, a.N, a.rider_V, a.H, a.W, a.Cout, blockIdx.x - a.main_blocks, reinterpret_cast<float*>(smem));
    return;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wave_a = wave & 1, wave_b = wave >> 1;
  int bid = blockIdx.x, split;
  const int nmain = a.rider_blocks > 0 ? a.main_blocks : (int)gridDim.x;
  if (a.xcd) {            // tiles fastest behind the XCD remap: the tiles of one pixel split share an L2 (WgradArgs.xcd)
    const int ntile = a.tiles_co * a.tiles_ci;
    bid = xcd_remap(blockIdx.x, nmain);
    split = bid / ntile; bid -= split * ntile;
  } else {
    split = bid % a.splits; bid /= a.splits;
  }
  const int tco = bid % a.tiles_co, tci = bid / a.tiles_co;
  const int ci0 = tci * 64, co0 = tco * 64;
  const bool do_bias = a.dbias != nullptr && tci == 0;

  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M >> 6) - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;

  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, a.N * a.Hx * a.Wx * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, a.N * a.Hdy * a.Wdy * a.Cout * 2, 0x00020000);
  const int pw_shift = a.sw - 3, pi_shift = a.shw - 6;   // patches per row / per image (log2)

  // per-thread chunk constants
  int x_hy[4], x_hx[4], x_c[4], x_lds[4];
  bool x_on[4];
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const int q = tid + NT * j;
    x_on[j] = q < 800;
    const int hp = x_on[j] ? q >> 3 : 0, cc = q & 7;
    x_hy[j] = hp / 10 - 1;
    x_hx[j] = hp % 10 - 1;
    x_c[j] = x_on[j] ? (ci0 + cc * 8) * 2 : OOB;
    x_lds[j] = (cc >> 2) * XSUB + hp * 32 + (cc & 3) * 8;
  }
  int d_y[2], d_x[2], d_c[2], d_lds[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int q = tid + NT * j, p = q >> 3, cc = q & 7;
    d_y[j] = p >> 3; d_x[j] = p & 7;
    d_c[j] = (co0 + cc * 8) * 2;
    d_lds[j] = (cc >> 2) * SUBS + p * 32 + (cc & 3) * 8;
  }

  u32x4 rX[PF][4], rD[PF][2];
  float bsum[2][8];
#pragma unroll
  for (int j = 0; j < 2; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bsum[j][e] = 0.f;

  const int last = nsteps - 1;
  int cur = 0;
  auto load_next = [&](u32x4 (&rX)[4], u32x4 (&rD)[2]) {
    const int patch = step0 + cur;                       // wave-uniform (SALU)
    const int n = patch >> pi_shift;
    const int pin = patch & ((1 << pi_shift) - 1);
    const int py0 = (pin >> pw_shift) << 3, px0 = (pin & ((1 << pw_shift) - 1)) << 3;
#pragma unroll
    for (int j = 0; j < 4; j++) {
      const int iy = py0 + x_hy[j], ix = px0 + x_hx[j];
      const bool ok = (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
      int off;
      if constexpr (XUP) off = ((n * a.Hx + (iy >> 1)) * a.Wx + (ix >> 1)) * a.Cin * 2 + x_c[j];
      else off = ((n * a.H + iy) * a.W + ix) * a.Cin * 2 + x_c[j];
      rX[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      const int oy = py0 + d_y[j], ox = px0 + d_x[j];
      int off;
      if constexpr (DYUP) off = ((n * a.Hdy + (oy >> 1)) * a.Wdy + (ox >> 1)) * a.Cout * 2 + d_c[j];
      else off = ((n * a.H + oy) * a.W + ox) * a.Cout * 2 + d_c[j];
      rD[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0);
    }
    if (cur < last) cur++;
  };
  auto store_step = [&](int buf, u32x4 (&rX)[4], u32x4 (&rD)[2]) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      if (x_on[j]) {
        u32x4 v = rX[j];
        if constexpr (XRELU) v = relu_bf16x8(v);
        *reinterpret_cast<u32x4*>(sX + buf * 2 * XSUB + x_lds[j]) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      *reinterpret_cast<u32x4*>(sD + buf * 2 * SUBS + d_lds[j]) = rD[j];
      if (do_bias) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, rD[j]);
#pragma unroll
        for (int e = 0; e < 8; e++) bsum[j][e] += bf2f(t[e]);
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

  const int g = lane >> 4, li = lane & 15;
  // lane part of the transposed-read addresses: k half (g>>1) = patch row within the pair, (li>>2) = pixel
  const int xl = ((g >> 1) * 10 + (li >> 2)) * 32 + 16 * (g & 1) + 4 * (li & 3);
  const int dl = (8 * (g >> 1) + (li >> 2)) * 32 + 16 * (g & 1) + 4 * (li & 3);

#pragma unroll
  for (int d = 0; d < PF; d++) load_next(rX[d], rD[d]);
  store_step(0, rX[0], rD[0]);
  __syncthreads();

  auto step = [&](int s, auto slot) {
    constexpr int D = decltype(slot)::value;
    const int buf = s & 1;
    if constexpr (PF > 1) load_next(rX[D], rD[D]);
    else if (s + 1 < nsteps) load_next(rX[0], rD[0]);
    const bf16* pX = sX + (buf * 2 + wave_a) * XSUB + xl;
    const bf16* pD = sD + (buf * 2 + wave_b) * SUBS + dl;
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
      const s16x4 bl = lds_tr_read(pD + kk * 16 * 32), bh = lds_tr_read(pD + kk * 16 * 32 + 4 * 32);
      const s16x8 tb = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
      const bf16x8 fb = __builtin_bit_cast(bf16x8, tb);
#pragma unroll
      for (int t = 0; t < 9; t++) {
        const int dh = t / 3 - 1, dw = t % 3 - 1;
        const int o = ((2 * kk + 1 + dh) * 10 + 1 + dw) * 32;
        const s16x4 al = lds_tr_read(pX + o), ah = lds_tr_read(pX + o + 4 * 32);
        const s16x8 ta = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
        acc[t] = GANK_MFMA32(__builtin_bit_cast(bf16x8, ta), fb, acc[t]);
      }
    }
    if (s + 1 < nsteps) store_step(buf ^ 1, rX[(D + 1) % PF], rD[(D + 1) % PF]);
    __syncthreads();
  };
  int s0 = 0;
  for (; s0 + PF <= nsteps; s0 += PF) {
    step(s0 + 0, std::integral_constant<int, 0>{});
    if constexpr (PF >= 2) step(s0 + 1, std::integral_constant<int, 1 % PF>{});
  }
  if constexpr (PF >= 2) { if (s0 < nsteps) step(s0, std::integral_constant<int, 0>{}); }

  // partial tile -> slab [split][tap][Cin][Cout] (plain stores) or atomics straight into dw
  const int r = lane & 31, h = lane >> 5;
  const int co = co0 + wave_b * 32 + r;
#pragma unroll
  for (int t = 0; t < 9; t++) {
#pragma unroll
    for (int e = 0; e < 16; e++) {
      const int ci = ci0 + wave_a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
      const long o = ((long)t * a.Cin + ci) * a.Cout + co;
      if (a.ws) a.ws[(long)split * 9 * a.Cin * a.Cout + o] = acc[t][e] * a.scale;
      else atomicAdd(a.dw + o, acc[t][e] * a.scale);
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [256][16]
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 8; e++) red[(tid * 2 + j) * 8 + e] = bsum[j][e];
    __syncthreads();
    if (tid < 64) {
      const int cc = tid >> 3, e = tid & 7;
      float tsum = 0.f;
      for (int p = 0; p < 64; p++) {
        const int q = p * 8 + cc;
        tsum += red[((q % NT) * 2 + q / NT) * 8 + e];
      }
      atomicAdd(a.dbias + co0 + tid, tsum * a.scale);
    }
  }
}

// dw[i] += sum_split ws[split][i]
__global__ void wgrad_reduce_slabs_kernel(const float* __restrict__ ws, float* __restrict__ dw, long n4, int splits) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 t = reinterpret_cast<const f32x4*>(dw)[i];
    for (int sb = 0; sb < splits; sb += 8) {          // 8 independent slab loads per batch, summed in slab order
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) v[u] = reinterpret_cast<const f32x4*>(ws)[(long)min(sb + u, splits - 1) * n4 + i];
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (sb + u < splits) { t[0] += v[u][0]; t[1] += v[u][1]; t[2] += v[u][2]; t[3] += v[u][3]; }
    }
    reinterpret_cast<f32x4*>(dw)[i] = t;
  }
}

static int wgrad_taps_min_m() {
  static const int v = gank_tune("GANK_WGRAD_TAPS_MINM", 16384);   // experiment knob
  return v;
}
static bool wgrad_taps_ok(const WgradArgs& a) {
  return a.ks == 3 && a.pad == 1 && !(a.flags & WG_X_STRIDE2) && a.sw >= 3 && a.shw >= 6 && a.H >= 8 && a.W >= 8 &&
         (a.M >= wgrad_taps_min_m() || (a.M >= 8192 && (a.Cin / 64) * (a.Cout / 64) >= 64)) &&   // small reductions go to the filter-row kernel unless the channel tiles alone fill the chip
         a.Cin % 64 == 0 && a.Cout % 64 == 0 && (a.M % 64 == 0) &&
         (long)a.N * a.Hx * a.Wx * a.Cin < (1L << 30) && (long)a.N * a.Hdy * a.Wdy * a.Cout < (1L << 30);
}

static void wgrad_taps_geometry(WgradArgs& a) {
  a.tiles_ci = a.Cin / 64;
  a.tiles_co = a.Cout / 64;
  const int total_steps = a.M / 64;
  const int tiles = a.tiles_ci * a.tiles_co;
  static const int target = gank_tune("GANK_WGRAD_TAPS_TARGET", 256);   // experiment knob // one workgroup per CU: 384 leaves a half-empty second round, 512 doubles the slab traffic
  int splits = (target + tiles - 1) / tiles;
  if (splits > total_steps / 4) splits = total_steps / 4;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
}

template <int MODE>
static int launch_wgrad_taps_mode(WgradArgs a, hipStream_t s) {
  wgrad_taps_geometry(a);
  const long slab = 9L * a.Cin * a.Cout;
  if (a.ws && a.ws_elems >= slab * a.splits && a.splits > 1) {
    // slab mode
  } else {
    a.ws = nullptr;   // atomics
  }
  size_t lds = (size_t)2 * 2 * (XSUB + SUBS) * sizeof(bf16);
  static const int tpf = gank_tune("GANK_WGRAD_TAPS_PF", 2);
#ifdef GANK_TUNING
  auto kern = tpf == 1 ? conv_wgrad_taps_kernel<MODE, 1> : conv_wgrad_taps_kernel<MODE, 2>;
#else
  (void)tpf;
  auto kern = conv_wgrad_taps_kernel<MODE, 2>;
#endif
  static const std::string tag = gank_format("conv_wgrad_taps_kernel<%d, 2> + wgrad_reduce_slabs_kernel", MODE);     // magic static: built once, thread-safe
  static const std::string tag_deferred = gank_format("conv_wgrad_taps_kernel<%d, 2>", MODE);      // (the slab reduction is a job of the caller's summing launch)
  gank_prof_tag(1, (a.ws && a.slab_job && a.splits <= 32) ? tag_deferred.c_str() : tag.c_str());
  a.xcd = wgrad_xcd_env();
  a.main_blocks = a.tiles_ci * a.tiles_co * a.splits;
  if (a.rider_blocks > 0) {
    GANK_REQUIRE(a.ws && a.slab_job && a.splits <= 32 && a.Hdy == a.H && a.Wdy == a.W, "conv_wgrad_taps: the tap-sums rider needs the deferred slab form");
    if (lds < (size_t)LABEL_TAP_SUMS_LDS) lds = LABEL_TAP_SUMS_LDS;
    GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_wgrad_taps");
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(a.main_blocks + (a.rider_blocks > 0 ? a.rider_blocks : 0))), dim3(256), lds, s, a);
  GANK_LAUNCH_OK("conv_wgrad_taps");
  if (a.ws && a.slab_job && a.splits <= 32) {       // the caller sums the slabs later, with other producers' (gank_sum_slabs)
    *a.slab_job = gank_slab_job{a.ws, a.dw, slab, slab, a.splits, 1.0f, 0};
  } else if (a.ws) {
    const long n4 = slab / 4;
    long blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(wgrad_reduce_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a.ws, a.dw, n4, a.splits);
    GANK_LAUNCH_OK("wgrad_reduce_slabs");
  }
  return 0;
}

static int launch_wgrad_taps(const WgradArgs& a, hipStream_t s) {
  const int mode = ((a.flags & GANK_IN_RELU) ? 1 : 0) | ((a.flags & GANK_IN_UPSAMPLE2X) ? 2 : 0) | ((a.flags & GANK_DY_UPSAMPLE2X) ? 4 : 0);
  switch (mode) {
    case 0: return launch_wgrad_taps_mode<0>(a, s);
    case 1: return launch_wgrad_taps_mode<1>(a, s);
    case 2: return launch_wgrad_taps_mode<2>(a, s);
    case 4: return launch_wgrad_taps_mode<4>(a, s);
    case 5: return launch_wgrad_taps_mode<5>(a, s);
    default: return -1;
  }
}

// ------------------------------------------------------------------------------------------------------
// ConvMeanPool 3x3 filter gradient, filter-row form.  The 16-tap stride-2 gradient through the per-tap kernel above
// re-reads x and dy from L2 once per tap and lands at ~0.2-0.4 PFLOP/s with a zero fill in front and a fold behind it.
// Here one block owns a 64(ci) x 64(co) tile for the FOUR taps of one filter row a over a range of 8x8 patches of the
// pooled grid: per patch it stages dy (64 pixels) and the 8 x-rows 2*py + a - 1 that this filter row touches (18
// columns), and the four taps b read their A fragments from that image at shifted columns.  The image is stored with
// the columns DE-INTERLEAVED by parity ([row][parity][9]), so a tap's stride-2 pixel walk is a unit-stride walk inside
// one parity plane and the transposed reads stay conflict-free (interleaved, pixels 0 and 2 of a 16-lane group would
// share banks).  Partial tiles go to slabs [split][16 taps][Cin][Cout]; ONE kernel then sums the splits, folds 4x4 -> 3x3
// (dW3[i][j] = 1/4 sum_{s,t} dW4[i+s][j+t]) and accumulates into dw: two launches instead of fill + 16-tap kernel + fold,
// 4 instead of 16 operand passes through L2.
// ------------------------------------------------------------------------------------------------------
constexpr int CPR_XSUB = 8 * 18 * 32 + 32;   // sub-tile stride (bf16) of the widest image: 64 B off a 256 B multiple

// S2 = true: the ConvMeanPool form above (4 taps per filter row, x read at stride 2, slabs).
// S2 = false: the same structure for a plain 3x3 stride-1 filter gradient (3 taps per filter row, 8 x 10 halo pixels per
// patch) -- for layers too small for the all-taps kernel (its 9-tap partial tiles, one per pixel split, cost more slab
// traffic than the layer has FLOPs); partial tiles are added to dw with fp32 atomics (a third of the all-taps volume per
// block) or written to slabs when a workspace is given.  grid.y = layer of a same-shape batch (nbatch > 0).
// KG = 2: 512 threads, the two wave groups take the first / second pixel-row pairs of every patch (two waves per SIMD: one
// group's transposed reads run under the other's MFMAs; with one wave per SIMD a step took ~1300 cycles against 512 of LDS
// reads and 384 of MFMA) and the second group's partial tile is added through LDS at the end -- same slab volume.
// KG = 3: 512 threads in two ROLES: waves 0-3 only read LDS and issue MFMAs, waves 4-7 only load the next patches and store them
// to the other LDS buffer (one barrier per step as before).  In the one-role form a step was the SUM of its MFMA section (0.25 us),
// its 20 buffer loads (0.2 us, the same with out-of-range addresses: issue through the CU's one address unit, not memory) and
// its 20 16-byte LDS stores (0.2 us): every wave stalled on the load issue in the middle of its MFMA stream.
template <int MODE, int PF, bool S2, int KG = 1>
__global__ __launch_bounds__(KG == 3 ? 512 : 256 * KG) void conv_wgrad_rows_kernel(WgradArgs a) {
  constexpr bool SPEC = KG == 3;
  constexpr int NT = SPEC ? 256 : 256 * KG, NDC = 512 / NT, KPG = SPEC ? 4 : 4 / KG;      // NT = staging threads
  constexpr bool XRELU = (MODE & 1) != 0;
  constexpr int COLS = S2 ? 18 : 10, NTAP = S2 ? 4 : 3, ST = S2 ? 2 : 1;
  constexpr int XCHUNKS = 8 * COLS * 8, NXC = (XCHUNKS + NT - 1) / NT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16* sX = reinterpret_cast<bf16*>(smem);        // [2][2 subs][8 * COLS pix][32 ch]
  bf16* sD = sX + 2 * 2 * CPR_XSUB;                // [2][2 subs][64 pix][32 ch]

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave_a = wave & 1, wave_b = (wave >> 1) & 1, kgrp = SPEC ? 0 : wave >> 2;
  const bool feeds = !SPEC || wave >= 4, computes = !SPEC || wave < 4;      // wave-uniform roles
  const int tid = SPEC ? (threadIdx.x & 255) : threadIdx.x;                   // index among the staging (and among the computing) threads
  const bf16* X = a.x;
  const bf16* DY = a.dy;
  float* DW = a.dw;
  float* DB = a.dbias;
  int bid = blockIdx.x, bi = blockIdx.y;
  int split, tco, tci, frow;
  if (a.xcd) {
    const int ntile = a.tiles_co * a.tiles_ci * (S2 ? 4 : 3), per_layer = ntile * a.splits;
    bid = xcd_remap(blockIdx.x, gridDim.x);
    bi = bid / per_layer; bid -= bi * per_layer;
    split = bid / ntile; bid -= split * ntile;
    tco = bid % a.tiles_co; bid /= a.tiles_co;
    tci = bid % a.tiles_ci;
    frow = bid / a.tiles_ci;
  } else {
    split = bid % a.splits; bid /= a.splits;
    tco = bid % a.tiles_co; bid /= a.tiles_co;
    tci = bid % a.tiles_ci;
    frow = bid / a.tiles_ci;               // filter row
  }
  float* WS = a.ws;
  if (a.nbatch > 0) {      // static indices only (a dynamically indexed by-value struct is spilled to scratch)
    WS = bi == 0 ? a.wss[0] : (bi == 1 ? a.wss[1] : (bi == 2 ? a.wss[2] : a.wss[3]));
    X = bi == 0 ? a.xs[0] : (bi == 1 ? a.xs[1] : (bi == 2 ? a.xs[2] : a.xs[3]));
    DY = bi == 0 ? a.dys[0] : (bi == 1 ? a.dys[1] : (bi == 2 ? a.dys[2] : a.dys[3]));
    DW = bi == 0 ? a.dws[0] : (bi == 1 ? a.dws[1] : (bi == 2 ? a.dws[2] : a.dws[3]));
    DB = bi == 0 ? a.dbs[0] : (bi == 1 ? a.dbs[1] : (bi == 2 ? a.dbs[2] : a.dbs[3]));
  }
  const int ci0 = tci * 64, co0 = tco * 64;
  const bool do_bias = DB != nullptr && tci == 0 && frow == 0;

  const int step0 = split * a.steps_per_split;
  int nsteps = (a.M >> 6) - step0;
  if (nsteps > a.steps_per_split) nsteps = a.steps_per_split;
  if (nsteps <= 0) return;
#ifdef GANK_TUNING
  if (a.dbg & 2) nsteps = 1;
#endif

  constexpr int OOB = 0x7FFFFFF0;
  const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(X), 0, a.N * a.Hx * a.Wx * a.Cin * 2, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(DY), 0, a.N * a.H * a.W * a.Cout * 2, 0x00020000);
  const int pw_shift = a.sw - 3, pi_shift = a.shw - 6;   // patches per row / per image of the dy grid (log2)

  int x_hy[NXC], x_hx[NXC], x_c[NXC], x_lds[NXC];
  bool x_on[NXC];
#pragma unroll
  for (int j = 0; j < NXC; j++) {
    const int q = tid + NT * j;
    x_on[j] = q < XCHUNKS;
    const int hp = x_on[j] ? q >> 3 : 0, cc = q & 7;
    const int hyi = hp / COLS, hx = hp % COLS;
    x_hy[j] = ST * hyi + frow - 1;
    x_hx[j] = hx - 1;
    x_c[j] = x_on[j] ? (ci0 + cc * 8) * 2 : OOB;
    const int slot = S2 ? hyi * COLS + (hx & 1) * (COLS / 2) + (hx >> 1) : hyi * COLS + hx;
    x_lds[j] = (cc >> 2) * CPR_XSUB + slot * 32 + (cc & 3) * 8;
  }
  int d_y[NDC], d_x[NDC], d_c[NDC], d_lds[NDC];
#pragma unroll
  for (int j = 0; j < NDC; j++) {
    const int q = tid + NT * j, p = q >> 3, cc = q & 7;
    d_y[j] = p >> 3; d_x[j] = p & 7;
    d_c[j] = (co0 + cc * 8) * 2;
    d_lds[j] = (cc >> 2) * SUBS + p * 32 + (cc & 3) * 8;
  }

  u32x4 rX[PF][NXC], rD[PF][NDC];
  float bsum[NDC][8];
#pragma unroll
  for (int j = 0; j < NDC; j++)
#pragma unroll
    for (int e = 0; e < 8; e++) bsum[j][e] = 0.f;

  const int last = nsteps - 1;
  int cur = 0;
  auto load_next = [&](u32x4 (&rX)[NXC], u32x4 (&rD)[NDC]) {
    const int patch = step0 + cur;                       // wave-uniform
    const int n = patch >> pi_shift;
    const int pin = patch & ((1 << pi_shift) - 1);
    const int py0 = (pin >> pw_shift) << 3, px0 = (pin & ((1 << pw_shift) - 1)) << 3;
#pragma unroll
    for (int j = 0; j < NXC; j++) {
      const int iy = ST * py0 + x_hy[j], ix = ST * px0 + x_hx[j];
      bool ok = (unsigned)iy < (unsigned)a.Hx && (unsigned)ix < (unsigned)a.Wx;
#ifdef GANK_TUNING
      if (a.dbg & 64) ok = false;            // timing only: x reads out of range (zero fill, no memory traffic)
#endif
      const int off = ((n * a.Hx + iy) * a.Wx + ix) * a.Cin * 2 + x_c[j];
      rX[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, ok ? off : OOB, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < NDC; j++) {
      int off = ((n * a.H + py0 + d_y[j]) * a.W + px0 + d_x[j]) * a.Cout * 2 + d_c[j];
#ifdef GANK_TUNING
      if (a.dbg & 128) off = OOB;            // timing only: dy reads out of range
#endif
      rD[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_dy, off, 0, 0);
    }
    if (cur < last) cur++;
  };
  auto store_step = [&](int buf, u32x4 (&rX)[NXC], u32x4 (&rD)[NDC]) {
#pragma unroll
    for (int j = 0; j < NXC; j++) {
      if (x_on[j]) {
        u32x4 v = rX[j];
        if constexpr (XRELU) v = relu_bf16x8(v);
        *reinterpret_cast<u32x4*>(sX + buf * 2 * CPR_XSUB + x_lds[j]) = v;
      }
    }
#pragma unroll
    for (int j = 0; j < NDC; j++) {
      *reinterpret_cast<u32x4*>(sD + buf * 2 * SUBS + d_lds[j]) = rD[j];
      if (do_bias) {
        const bf16x8 t = __builtin_bit_cast(bf16x8, rD[j]);
#pragma unroll
        for (int e = 0; e < 8; e++) bsum[j][e] += bf2f(t[e]);
      }
    }
  };

  f32x16 acc[NTAP];
#pragma unroll
  for (int t = 0; t < NTAP; t++)
#pragma unroll
    for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

  const int g = lane >> 4, li = lane & 15;
  // lane part of the transposed-read addresses: k half (g>>1) = dy row within the pair, (li>>2) = dy column
  const int xl = ((g >> 1) * COLS + (li >> 2)) * 32 + 16 * (g & 1) + 4 * (li & 3) + kgrp * KPG * 2 * COLS * 32;     // (+ the wave group's first row pair)
  const int dl = (8 * (g >> 1) + (li >> 2)) * 32 + 16 * (g & 1) + 4 * (li & 3) + kgrp * KPG * 16 * 32;

  if (feeds) {
#pragma unroll
    for (int d = 0; d < PF; d++) load_next(rX[d], rD[d]);
    store_step(0, rX[0], rD[0]);
  }
  __syncthreads();

  auto step = [&](int s, auto slot) {
    constexpr int D = decltype(slot)::value;
    const int buf = s & 1;
#ifdef GANK_TUNING
    if (!(a.dbg & 8))
#endif
    if (feeds) {
      if constexpr (PF > 1) load_next(rX[D], rD[D]);
      else if (s + 1 < nsteps) load_next(rX[0], rD[0]);
    }
    const bf16* pX = sX + (buf * 2 + wave_a) * CPR_XSUB + xl;
    const bf16* pD = sD + (buf * 2 + wave_b) * SUBS + dl;
#ifdef GANK_TUNING
    if (!(a.dbg & 4))
#endif
    if (computes)
#pragma unroll
    for (int kk = 0; kk < KPG; kk++) {
      const s16x4 bl = lds_tr_read(pD + kk * 16 * 32), bh = lds_tr_read(pD + kk * 16 * 32 + 4 * 32);
      const s16x8 tb = {bl[0], bl[1], bl[2], bl[3], bh[0], bh[1], bh[2], bh[3]};
      const bf16x8 fb = __builtin_bit_cast(bf16x8, tb);
#pragma unroll
      for (int t = 0; t < NTAP; t++) {
        const int o = S2 ? (2 * kk * COLS + (t & 1) * (COLS / 2) + (t >> 1)) * 32 : (2 * kk * COLS + t) * 32;
        const s16x4 al = lds_tr_read(pX + o), ah = lds_tr_read(pX + o + 4 * 32);
        const s16x8 ta = {al[0], al[1], al[2], al[3], ah[0], ah[1], ah[2], ah[3]};
        acc[t] = GANK_MFMA32(__builtin_bit_cast(bf16x8, ta), fb, acc[t]);
      }
    }
#ifdef GANK_TUNING
    if (!(a.dbg & 16))
#endif
    if (feeds && s + 1 < nsteps) store_step(buf ^ 1, rX[(D + 1) % PF], rD[(D + 1) % PF]);
#ifdef GANK_TUNING
    if (!(a.dbg & 32))
#endif
    __syncthreads();
  };
  int s0 = 0;
  for (; s0 + PF <= nsteps; s0 += PF) {
    step(s0 + 0, std::integral_constant<int, 0>{});
    if constexpr (PF >= 2) step(s0 + 1, std::integral_constant<int, 1 % PF>{});
    if constexpr (PF >= 3) step(s0 + 2, std::integral_constant<int, 2 % PF>{});
    if constexpr (PF >= 4) step(s0 + 3, std::integral_constant<int, 3 % PF>{});
  }
  if constexpr (PF >= 2) { if (s0 + 0 < nsteps) step(s0 + 0, std::integral_constant<int, 0>{}); }
  if constexpr (PF >= 3) { if (s0 + 1 < nsteps) step(s0 + 1, std::integral_constant<int, 1 % PF>{}); }
  if constexpr (PF >= 4) { if (s0 + 2 < nsteps) step(s0 + 2, std::integral_constant<int, 2 % PF>{}); }

#ifdef GANK_TUNING
  if (a.dbg & 1) { if (acc[0][0] == 123.456f) DW[0] = 1.f; return; }
#endif
  if constexpr (KG == 2) {        // second wave group's partial tile -> first group's registers (the loop's last barrier has passed)
    float* xr = reinterpret_cast<float*>(smem);             // [NTAP * 16][256]
    if (kgrp == 1) {
#pragma unroll
      for (int t = 0; t < NTAP; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) xr[(t * 16 + e) * 256 + (tid & 255)] = acc[t][e];
    }
    __syncthreads();
    if (kgrp == 0) {
#pragma unroll
      for (int t = 0; t < NTAP; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] += xr[(t * 16 + e) * 256 + tid];
    }
  }
  // partial tile -> slab [split][frow*NTAP + t][Cin][Cout], or atomics into dw
  const int r = lane & 31, h = lane >> 5;
  const int co = co0 + wave_b * 32 + r;
  const long plane = (long)a.Cin * a.Cout;
  if (kgrp == 0 && computes) {
#pragma unroll
    for (int t = 0; t < NTAP; t++) {
      float* dst = WS ? WS + ((long)split * NTAP * NTAP + frow * NTAP + t) * plane : DW + (long)(frow * NTAP + t) * plane;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int ci = ci0 + wave_a * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
        if (WS) dst[(long)ci * a.Cout + co] = acc[t][e];
        else atomicAdd(dst + (long)ci * a.Cout + co, acc[t][e] * a.scale);
      }
    }
  }
  if (do_bias) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem);            // [512 chunks][8]
    if (feeds) {
#pragma unroll
      for (int j = 0; j < NDC; j++)
#pragma unroll
        for (int e = 0; e < 8; e++) red[(tid * NDC + j) * 8 + e] = bsum[j][e];
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      const int cc = tid >> 3, e = tid & 7;
      float tsum = 0.f;
      for (int p = 0; p < 64; p++) {
        const int q = p * 8 + cc;
        tsum += red[((q % NT) * NDC + q / NT) * 8 + e];
      }
      atomicAdd(DB + co0 + tid, tsum * a.scale);
    }
  }
}

static int wgrad_rows_kg() {
  static const int v = gank_tune("GANK_WGRAD_ROWS_KG", 3);   // experiment knob: 1 = 256 threads, every wave feeds and computes; 2 = two K groups (512 threads); 3 = feeding + computing waves (512 threads)
  return (v == 2 || v == 3) ? v : 1;
}
template <bool S2>
static size_t wgrad_rows_lds(int kg) {
  const size_t stage = (size_t)2 * 2 * (CPR_XSUB + SUBS) * sizeof(bf16), xr = kg == 2 ? (size_t)(S2 ? 4 : 3) * 16 * 256 * sizeof(float) : 0;
  return stage > xr ? stage : xr;
}
// filter-row kernel for plain 3x3 stride-1 layers below the all-taps kernel's size (single layer or a same-shape batch)
static bool wgrad_rows_ok(const WgradArgs& a) {
  static const int env = gank_tune("GANK_WGRAD_ROWS", 1);   // experiment knob: GANK_WGRAD_ROWS=0 keeps these layers on the per-tap kernel
  return env && a.ks == 3 && a.pad == 1 && (a.flags & ~GANK_IN_RELU) == 0 && a.sw >= 3 && a.shw >= 6 && a.H >= 8 && a.W >= 8 &&
         a.Hx == a.H && a.Wx == a.W && a.Hdy == a.H && a.Wdy == a.W && a.Cin % 64 == 0 && a.Cout % 64 == 0 && a.M % 64 == 0 &&
         (long)a.N * a.H * a.W * a.Cin < (1L << 30) && (long)a.N * a.H * a.W * a.Cout < (1L << 30);
}
static int wgrad_rows_splits(const WgradArgs& a, int nb) {     // the pixel splits launch_wgrad_rows will use for a batch of nb layers
  const int total_steps = a.M / 64, tiles = (a.Cin / 64) * (a.Cout / 64) * 3 * nb;
  static const int target = gank_tune("GANK_WGRAD_ROWS_TARGET", 256);
  int splits = wgrad_round_down_env() ? target / tiles : (target + tiles - 1) / tiles;
  if (splits > total_steps / 4) splits = total_steps / 4;
  if (splits < 1) splits = 1;
  return cdiv(total_steps, cdiv(total_steps, splits));
}
static int launch_wgrad_rows(WgradArgs a, hipStream_t s) {
  const int nb = a.nbatch > 0 ? a.nbatch : 1;
  a.tiles_ci = a.Cin / 64;
  a.tiles_co = a.Cout / 64;
  const int total_steps = a.M / 64;
  const int tiles = a.tiles_ci * a.tiles_co * 3 * nb;
  static const int target = gank_tune("GANK_WGRAD_ROWS_TARGET", 256);   // experiment knob
  // at most `target` workgroups: 6 splits x 48 tiles = 288 blocks ran as two rounds on 256 CUs (31 us), 5 x 48 = 240 as one
  int splits = wgrad_round_down_env() ? target / tiles : (target + tiles - 1) / tiles;
  if (splits > total_steps / 4) splits = total_steps / 4;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
  a.ws = nullptr;          // partial tiles by fp32 atomics, or (a batched launch with wss[] set) to per-layer slabs [split][9][Cin][Cout]
  if (a.nbatch <= 0) for (int j = 0; j < 4; j++) a.wss[j] = nullptr;
  const int kg = wgrad_rows_kg();
  const size_t lds = wgrad_rows_lds<false>(kg);
  const bool relu = (a.flags & GANK_IN_RELU) != 0;
  static const int pf = gank_tune("GANK_WGRAD_ROWS_PF", 2);   // experiment knob: register prefetch depth
#ifdef GANK_TUNING
  auto kern = relu ? (pf == 4 ? conv_wgrad_rows_kernel<1, 4, false> : pf == 3 ? conv_wgrad_rows_kernel<1, 3, false> : conv_wgrad_rows_kernel<1, 2, false>)
                   : (pf == 4 ? conv_wgrad_rows_kernel<0, 4, false> : pf == 3 ? conv_wgrad_rows_kernel<0, 3, false> : conv_wgrad_rows_kernel<0, 2, false>);
#else
  (void)pf;
  auto kern = relu ? conv_wgrad_rows_kernel<1, 2, false> : conv_wgrad_rows_kernel<0, 2, false>;
#endif
  if (kg == 2) kern = relu ? conv_wgrad_rows_kernel<1, 2, false, 2> : conv_wgrad_rows_kernel<0, 2, false, 2>;
  if (kg == 3) kern = relu ? conv_wgrad_rows_kernel<1, 2, false, 3> : conv_wgrad_rows_kernel<0, 2, false, 3>;
  static_assert((size_t)2 * 2 * (CPR_XSUB + SUBS) * sizeof(bf16) <= 65536, "below the default dynamic-LDS limit: no attribute call");
  gank_prof_tag(1, relu ? "conv_wgrad_rows_kernel<1, 2, false>" : "conv_wgrad_rows_kernel<0, 2, false>");
  a.xcd = wgrad_xcd_env();
  static const int dbg = gank_tune("GANK_WGRAD_DBG", 0);
  a.dbg = dbg;
  if (a.xcd) hipLaunchKernelGGL(kern, dim3((unsigned)(3 * a.tiles_ci * a.tiles_co * a.splits * nb)), dim3(kg == 1 ? 256 : 512), lds, s, a);
  else hipLaunchKernelGGL(kern, dim3((unsigned)(3 * a.tiles_ci * a.tiles_co * a.splits), nb), dim3(kg == 1 ? 256 : 512), lds, s, a);
  GANK_LAUNCH_OK("conv_wgrad_rows");
  return 0;
}

template <int WA, int WB, int TA, int TB, bool FAST, int PF>
static int launch_wgrad(WgradArgs a, hipStream_t s) {
  constexpr int CiT = WA * TA * 32, CoT = WB * TB * 32;
  a.tiles_ci = cdiv(a.Cin, CiT);
  a.tiles_co = cdiv(a.Cout, CoT);
  const int total_steps = cdiv(a.M, 64);
  const long tiles = (long)a.taps * a.tiles_ci * a.tiles_co;
  // enough blocks to fill 256 CUs ~3x, but at least 4 pixel steps per block
  int splits = (int)((768 + tiles - 1) / tiles);
  if (splits > total_steps / 4) splits = total_steps / 4;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
  const size_t lds = (size_t)2 * (CiT / 32 + CoT / 32) * 2048 * sizeof(bf16);
  auto kern = conv_wgrad_kernel<WA, WB, TA, TB, FAST, PF>;
  GANK_MAX_DYNAMIC_LDS(kern, (int)lds, "conv_wgrad");
  const long grid = tiles * a.splits;
  GANK_REQUIRE(grid < (1L << 30), "conv_wgrad: grid too large");
  static const std::string tag = gank_format("conv_wgrad_kernel<%d, %d, %d, %d, %s, %d>", WA, WB, TA, TB, FAST ? "true" : "false", PF);     // magic static: built once, thread-safe
  gank_prof_tag(1, tag.c_str());
  wgrad_slab_setup(a);
  { static const int dbg_ = gank_tune("GANK_WGRAD_DBG", 0); a.dbg = dbg_; }
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(WA * WB * 64), lds, s, a);
  GANK_LAUNCH_OK("conv_wgrad");
  return 0;
}

int gank_wgrad_dispatch(WgradArgs a, hipStream_t s) {
  GANK_REQUIRE(a.x && a.dy && a.dw, "conv_wgrad: null pointer");
  GANK_REQUIRE(a.N > 0 && a.H > 0 && a.W > 0 && a.Cin > 0 && a.Cout > 0, "conv_wgrad: bad shape");
  GANK_REQUIRE(a.ks >= 1 && a.ks <= 7, "conv_wgrad: unsupported filter size %d", a.ks);
  a.taps = a.ks * a.ks;
  a.M = a.N * a.H * a.W;
  a.sw = log2_or_neg(a.W);
  a.shw = log2_or_neg(a.H * a.W);
  GANK_REQUIRE((long)a.N * a.H * a.W < (1L << 31), "conv_wgrad: too many pixels");
  const double flops = 2.0 * a.M * (double)a.Cout * a.taps * a.Cin;
  const bool fast = (a.Cin % 8 == 0) && (a.Cout % 8 == 0);
  gank_prof_begin(1, flops, s, 2.0 * ((double)a.N * a.Hx * a.Wx * a.Cin + (double)a.N * a.Hdy * a.Wdy * a.Cout) + 4.0 * a.taps * a.Cin * a.Cout);
  int rc = -1;
  const bool same_grid = (a.flags & (WG_X_STRIDE2 | GANK_IN_UPSAMPLE2X)) || (a.Hx == a.H && a.Wx == a.W);    // the lean kernel's plain gather assumes x on dy's grid
  const bool dy_grid = (a.flags & GANK_DY_UPSAMPLE2X) || (a.Hdy == a.H && a.Wdy == a.W);
  const bool lean = fast && same_grid && dy_grid && a.sw >= 0 && a.shw >= 0 && (a.M % 64 == 0) &&
                    (long)a.N * a.Hx * a.Wx * a.Cin < (1L << 30) && (long)a.N * a.Hdy * a.Wdy * a.Cout < (1L << 30);
  static const int taps_env = gank_tune("GANK_WGRAD_TAPS", 1);   // experiment knob: GANK_WGRAD_TAPS=0 disables the all-taps kernel
  if (taps_env && wgrad_taps_ok(a)) rc = launch_wgrad_taps(a, s);
  if (rc < 0 && wgrad_rows_ok(a)) rc = launch_wgrad_rows(a, s);
  static const int lpf_env = gank_tune("GANK_WGRAD_LEAN_PF", 2);   // experiment knob: prefetch depth of the lean kernel
  (void)lpf_env;
  // 1x1 layers on few pixels (the shortcuts: 2 K - 8 K pixels, 0.5 - 1 GFLOP): the partial tile a workgroup adds with fp32
  // atomics is what it costs (a CU retires one 256-byte atomic instruction per ~50 ns: 64 KB of a 128 x 128 tile = 13 us);
  // 64 x 64 tiles are 16 KB per workgroup and fill the chip four times better
  static const int small1x1_env = gank_tune("GANK_WGRAD_1X1_SMALL_TILE", 1);
  if (rc < 0 && lean) {
#ifdef GANK_TUNING
    if (a.Cin >= 128 && a.Cout >= 128 && lpf_env == 4) rc = launch_wgrad_lean<2, 2, 2, 2, 4>(a, s);
    else if (a.Cin >= 128 && a.Cout >= 128 && lpf_env == 3) rc = launch_wgrad_lean<2, 2, 2, 2, 3>(a, s);
    else
#endif
    if (small1x1_env && a.ks == 1 && a.M <= 16384 && a.Cin >= 64 && a.Cout >= 64) rc = launch_wgrad_lean<2, 2, 1, 1, 2>(a, s);
    else if (a.Cin >= 128 && a.Cout >= 128) rc = launch_wgrad_lean<2, 2, 2, 2, 2>(a, s);
    else if (a.Cin >= 64 && a.Cout >= 64) rc = launch_wgrad_lean<2, 2, 1, 1, 2>(a, s);
  }
  // narrow operands: pack (tap, channel) into <= 32 MFMA columns so the wide operand streams once
  const bool plain = (a.flags == 0) && a.Hx == a.H && a.Wx == a.W && a.Hdy == a.H && a.Wdy == a.W &&      // the packed kernels walk x and dy on one grid
                     (long)a.M * (a.Cin > a.Cout ? a.Cin : a.Cout) < (1L << 30);
  if (rc < 0 && plain && wgrad_narrow_stream_ok(a.N, a.H, a.W, a.Cin, a.Cout, a.ks)) {
    gank_prof_tag(1, a.ks == 3 ? "conv_wgrad_narrow_stream_kernel<3>" : "conv_wgrad_narrow_stream_kernel<1>");
    const NarrowWgArgs q = narrow_args(a.x, a.dy, a.dw, a.dbias, a.N, a.H, a.W, a.Cout, a.scale);
    rc = launch_wgrad_narrow_stream(q, a.ks, q, 0, s);
  }
  if (rc < 0 && plain && a.Cin <= 4 && a.taps * a.Cin <= 32 && a.Cout % 8 == 0) rc = launch_wgrad_packed<true>(a, s);
  if (rc < 0 && plain && a.Cout <= 4 && a.taps * a.Cout <= 32 && a.Cin % 8 == 0) {
    rc = launch_wgrad_packed<false>(a, s);
    if (rc == 0 && a.dbias) rc = gank_colsum_bf16(a.dy, a.dbias, a.M, a.Cout, a.scale, s);   // 3 columns: trivial
  }
  if (rc >= 0) {
    // done by a specialised kernel
  } else if (fast) {
    if (a.Cin >= 128 && a.Cout >= 128) rc = launch_wgrad<2, 2, 2, 2, true, 3>(a, s);
    else if (a.Cin <= 32) rc = launch_wgrad<1, 4, 1, 1, true, 3>(a, s);
    else if (a.Cout <= 32) rc = launch_wgrad<4, 1, 1, 1, true, 3>(a, s);
    else rc = launch_wgrad<2, 2, 1, 1, true, 3>(a, s);                    // 64 x 64
  } else {
    if (a.Cin >= 128 && a.Cout >= 128) rc = launch_wgrad<2, 2, 2, 2, false, 1>(a, s);
    else if (a.Cin <= 32) rc = launch_wgrad<1, 4, 1, 1, false, 1>(a, s);  // 32 ci x 128 co (image-side layers)
    else if (a.Cout <= 32) rc = launch_wgrad<4, 1, 1, 1, false, 1>(a, s); // 128 ci x 32 co (G.Output, D.Output)
    else rc = launch_wgrad<2, 2, 1, 1, false, 1>(a, s);
  }
  gank_prof_end(1, s);
  return rc;
}

extern "C" long gank_conv2d_wgrad_ws_elems(int N, int H, int W, int Cin, int Cout, int ksize, int flags) {
  WgradArgs a{};
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = (ksize - 1) / 2; a.flags = flags;
  const bool xup = flags & GANK_IN_UPSAMPLE2X, dyup = flags & GANK_DY_UPSAMPLE2X;
  a.Hx = xup ? H / 2 : H; a.Wx = xup ? W / 2 : W; a.Hdy = dyup ? H / 2 : H; a.Wdy = dyup ? W / 2 : W;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  if (!wgrad_taps_ok(a)) return 0;
  wgrad_taps_geometry(a);
  return a.splits > 1 ? 9L * Cin * Cout * a.splits : 0;
}

namespace { struct TapSumsRider { const int32_t* lists; float* S; int V; }; struct Dgrad1x1Rider { const void* wd; int pitch; void* dx; }; }
static int conv2d_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, float* ws, long ws_elems, int N, int H, int W, int Cin, int Cout,
                             int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream,
                             const TapSumsRider* rider = nullptr, const Dgrad1x1Rider* d1 = nullptr);
extern "C" int gank_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws, long ws_elems, int N,
                                 int H, int W, int Cin, int Cout, int ksize, int flags, float scale, void* stream) {
  return conv2d_wgrad_impl(x, dy, dw, dbias, ws, ws_elems, N, H, W, Cin, Cout, ksize, flags, scale, nullptr, 0, nullptr, stream);
}
// floats of slab workspace worth offering to gank_conv2d_wgrad_slabs (0: the filter is too large for per-split copies, or the
// layer runs on a kernel with its own slab reduction): up to 256 pixel splits of filters of <= 65536 elements
extern "C" long gank_conv2d_wgrad_slab_elems(int N, int H, int W, int Cin, int Cout, int ksize, int flags) {
  const long dw_elems = (long)ksize * ksize * Cin * Cout;
  const long taps_ws = gank_conv2d_wgrad_ws_elems(N, H, W, Cin, Cout, ksize, flags);
  if (taps_ws > 0) return taps_ws;              // all-taps kernel: its slabs, the reduction left to the caller's summing launch
  if (dw_elems > 65536 || (long)N * H * W < 4096) return 0;
  // Measured in the step (round 5): the narrow-channel layers (G.Output's 256 -> 3 filter gradient: 256 blocks x 6912 atomics onto the
  // same addresses) lose 21 of 41 us to their atomics and 18 of them come back with slabs; the 1x1 shortcuts' small tiles lose 3 of
  // 8.6 us and a slab store + their share of the summing launch costs as much; the 256 -> 256 shortcut's 64 slabs are 16.8 MB.
  if (!(Cin <= 4 || Cout <= 4) || ksize * ksize * (Cin <= 4 ? Cin : Cout) > 32) return 0;
  return 256 * dw_elems;
}
// pixel splits of the all-taps kernel's deferred slab form for this layer (0: another kernel, or more than gank_sum_slabs' wide path takes)
extern "C" int gank_conv2d_wgrad_slab_splits(int N, int H, int W, int Cin, int Cout, int ksize, int flags) {
  WgradArgs a{};
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = (ksize - 1) / 2; a.flags = flags;
  const bool xup = flags & GANK_IN_UPSAMPLE2X, dyup = flags & GANK_DY_UPSAMPLE2X;
  a.Hx = xup ? H / 2 : H; a.Wx = xup ? W / 2 : W; a.Hdy = dyup ? H / 2 : H; a.Wdy = dyup ? W / 2 : W;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  static const int taps_env = gank_tune("GANK_WGRAD_TAPS", 1);
  if (!taps_env || !wgrad_taps_ok(a)) return 0;
  wgrad_taps_geometry(a);
  return (a.splits > 1 && a.splits <= 32) ? a.splits : 0;
}
static int wgrad_slabs_rows_impl(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total,
                                 int Cout, int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream,
                                 const TapSumsRider* rider);
extern "C" int gank_conv2d_wgrad_slabs_rows(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total,
                                            int Cout, int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream) {
  return wgrad_slabs_rows_impl(x, dy, dw_full, dbias, N, H, W, Cin, Cin_total, Cout, ksize, flags, scale, slab_ws, slab_elems, job, stream, nullptr);
}
// ... with the per-label tap sums of the SAME dy (gank_label_conv3x3_bwd's first launch: lists from gank_label_conv3x3_table, V labels,
// tap_sums_ws of gank_label_conv3x3_bwd_ws_floats floats) computed by extra workgroups of the filter-gradient launch; pass dy = NULL to
// gank_label_conv3x3_bwd afterwards
extern "C" int gank_conv2d_wgrad_slabs_rows_tap_sums(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total,
                                                     int Cout, int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job,
                                                     const int32_t* lists, int V, float* tap_sums_ws, void* stream) {
  GANK_REQUIRE(lists && tap_sums_ws && V > 0 && V <= 16 && Cout % 64 == 0 && H % 2 == 0 && ((H / 2) * W) % 32 == 0 && (H / 2) * W <= 128 && N <= 1024,
               "conv2d_wgrad_slabs_rows_tap_sums: unsupported geometry for the tap-sums rider");
  const TapSumsRider rider{lists, tap_sums_ws, V};
  return wgrad_slabs_rows_impl(x, dy, dw_full, dbias, N, H, W, Cin, Cin_total, Cout, ksize, flags, scale, slab_ws, slab_elems, job, stream, &rider);
}
static int wgrad_slabs_rows_impl(const void* x, const void* dy, float* dw_full, float* dbias, int N, int H, int W, int Cin, int Cin_total,
                                 int Cout, int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream,
                                 const TapSumsRider* rider) {
  GANK_REQUIRE(slab_ws && job && slab_elems > 0 && dw_full && Cin_total >= Cin, "conv2d_wgrad_slabs_rows: bad arguments");
  const int splits = gank_conv2d_wgrad_slab_splits(N, H, W, Cin, Cout, ksize, flags);
  GANK_REQUIRE(splits > 0 && slab_elems >= (long)ksize * ksize * Cin * Cout * splits && (Cin * Cout) % 4 == 0 && ((long)Cin_total * Cout) % 4 == 0 && scale == 1.0f,
               "conv2d_wgrad_slabs_rows: this layer has no deferred slab form (gank_conv2d_wgrad_slab_splits = %d)", splits);
  *job = gank_slab_job{nullptr, dw_full, (long)ksize * ksize * Cin * Cout, (long)ksize * ksize * Cin * Cout, 0, scale, 0, 0, 0};
  if (conv2d_wgrad_impl(x, dy, dw_full, dbias, nullptr, 0, N, H, W, Cin, Cout, ksize, flags, scale, slab_ws, slab_elems, job, stream, rider)) return 1;
  GANK_REQUIRE(job->nslabs > 0, "conv2d_wgrad_slabs_rows: the launch did not leave its reduction to the job");
  job->out_run = (long)Cin * Cout;
  job->out_pitch = (long)Cin_total * Cout;
  return 0;
}
extern "C" int gank_conv2d_wgrad_slabs(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int Cin, int Cout, int ksize,
                                       int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream) {
  GANK_REQUIRE(slab_ws && job && slab_elems > 0, "conv2d_wgrad_slabs: null slab workspace / job");
  *job = gank_slab_job{nullptr, dw, (long)ksize * ksize * Cin * Cout, (long)ksize * ksize * Cin * Cout, 0, scale, 0};
  return conv2d_wgrad_impl(x, dy, dw, dbias, nullptr, 0, N, H, W, Cin, Cout, ksize, flags, scale, slab_ws, slab_elems, job, stream);
}
static int conv2d_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, float* ws, long ws_elems, int N, int H, int W, int Cin, int Cout,
                             int ksize, int flags, float scale, float* slab_ws, long slab_elems, gank_slab_job* job, void* stream,
                             const TapSumsRider* rider, const Dgrad1x1Rider* d1) {
  GANK_REQUIRE(ksize % 2 == 1, "conv2d_wgrad: even filter sizes are not on this path (ksize=%d)", ksize);
  WgradArgs a{};
  if (d1) { a.d1_wd = (const bf16*)d1->wd; a.d1_pitch = d1->pitch; a.d1_dx = (bf16*)d1->dx; }
  if (rider) { a.rider_lists = rider->lists; a.rider_S = rider->S; a.rider_V = rider->V; a.rider_blocks = rider->V * 2 * (Cout / 64); }
  a.slab_ws = slab_ws; a.slab_elems = slab_elems; a.slab_job = job;
  if (slab_ws && !ws) { ws = slab_ws; ws_elems = slab_elems; }        // (a layer on the all-taps kernel takes the offered space as its slab workspace)
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = dw; a.dbias = dbias; a.ws = ws; a.ws_elems = ws_elems;
  a.N = N; a.H = H; a.W = W;
  const bool xup = flags & GANK_IN_UPSAMPLE2X, dyup = flags & GANK_DY_UPSAMPLE2X;
  GANK_REQUIRE(!(xup || dyup) || (H % 2 == 0 && W % 2 == 0), "conv2d_wgrad: 2x flags need even size");
  a.Hx = xup ? H / 2 : H; a.Wx = xup ? W / 2 : W;
  a.Hdy = dyup ? H / 2 : H; a.Wdy = dyup ? W / 2 : W;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = (ksize - 1) / 2;
  a.flags = flags & (GANK_IN_UPSAMPLE2X | GANK_IN_RELU | GANK_DY_UPSAMPLE2X);
  a.scale = scale;
  return gank_wgrad_dispatch(a, (hipStream_t)stream);
}

// both gradients of a 1x1 conv (stride 1) in ONE launch: the filter / bias gradient as gank_conv2d_wgrad (ACCUMULATED into dw [Cin][Cout],
// dbias) and the input gradient dx [N,H,W,Cin] = dy wd^T (bf16, the bytes' meaning of gank_conv2d_dgrad with ksize 1, no flags) computed
// by extra workgroups of the filter-gradient launch.  wd: the plain-conv dgrad operand (bf16 [Cin][wd_pitch], rows = input channels).
// Cin % 64 == 0, Cout % 16 == 0, Cout <= 128, N * H * W % 32 == 0.  A layer whose filter gradient runs on a kernel without the rider
// gets its input gradient from a launch of its own (gank_conv2d_dgrad) -- the results are the same either way.
extern "C" int gank_conv2d_dgrad(const void* dy, const void* wd, const void* residual, const void* relu_ref, void* dx, int N, int H, int W, int Cin, int Cout,
                                 int ksize, int flags, float scale, void* stream);
extern "C" int gank_conv1x1_wgrad_dgrad(const void* x, const void* dy, float* dw, float* dbias, const void* wd, int wd_pitch, void* dx, int N, int H, int W,
                                        int Cin, int Cout, void* stream) {
  GANK_REQUIRE(x && dy && dw && wd && dx && N > 0 && H > 0 && W > 0, "conv1x1_wgrad_dgrad: null pointer");
  GANK_REQUIRE(Cin % 64 == 0 && Cout % 16 == 0 && Cout <= 128 && ((long)N * H * W) % 32 == 0 && wd_pitch >= Cout && wd_pitch % 8 == 0 &&
               (long)N * H * W * Cin < (1L << 31), "conv1x1_wgrad_dgrad: needs Cin %% 64 == 0, Cout %% 16 == 0, Cout <= 128, pixels %% 32 == 0 (got %d, %d, %ld)",
               Cin, Cout, (long)N * H * W);
  const Dgrad1x1Rider d1{wd, wd_pitch, dx};
  g_d1_carried = false;
  if (conv2d_wgrad_impl(x, dy, dw, dbias, nullptr, 0, N, H, W, Cin, Cout, 1, 0, 1.0f, nullptr, 0, nullptr, stream, nullptr, &d1)) return 1;
  if (g_d1_carried) return 0;
  return gank_conv2d_dgrad(dy, wd, nullptr, nullptr, dx, N, H, W, Cin, Cout, 1, 0, 1.0f, stream);
}

// filter gradient of the general convolution (gank_conv2d_general_fprop): x [N,Hx,Wx,Cin] as stored, dy [N,Hdy,Wdy,Cout];
// flags GANK_IN_RELU, GANK_IN_UPSAMPLE2X; ACCUMULATES into dw fp32 [k,k,Cin,Cout] (and dbias)
extern "C" int gank_conv2d_general_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int Hx, int Wx, int Hdy, int Wdy,
                                         int Cin, int Cout, int ksize, int stride, int pad, int flags, void* stream) {
  GANK_REQUIRE(stride == 1 || stride == 2, "conv2d_general_wgrad: stride %d (1 or 2)", stride);
  GANK_REQUIRE(pad >= 0 && pad < ksize, "conv2d_general_wgrad: pad %d outside [0, ksize)", pad);
  GANK_REQUIRE(!((flags & GANK_IN_UPSAMPLE2X) && stride == 2), "conv2d_general_wgrad: upsampled input with stride 2");
  WgradArgs a{};
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = dw; a.dbias = dbias;
  a.N = N; a.H = Hdy; a.W = Wdy; a.Hx = Hx; a.Wx = Wx; a.Hdy = Hdy; a.Wdy = Wdy;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = pad;
  a.flags = (flags & (GANK_IN_UPSAMPLE2X | GANK_IN_RELU)) | (stride == 2 ? WG_X_STRIDE2 : 0);
  a.scale = 1.f;
  return gank_wgrad_dispatch(a, (hipStream_t)stream);
}

// ---- ConvMeanPool 3x3 filter gradient ------------------------------------------------------------------
// dW3[i][j] = 1/4 sum_{s,t in {0,1}} dW4[i+s][j+t], where dW4 is the filter gradient of the equivalent 4x4
// stride-2 conv over the POOLED pixel grid (16 taps x M/4 pixels instead of 9 taps x M).  ws16: fp32 scratch of
// 16*Cin*Cout elements (zeroed here, by a kernel).
__global__ void wgrad_zero_kernel(float* __restrict__ p, long n4) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}
__global__ void wgrad_fold4x4_kernel(const float* __restrict__ w4, float* __restrict__ dw, long plane4) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;     // one float4 of the [Cin][Cout] plane
  if (i >= plane4) return;
  f32x4 t[16];
#pragma unroll
  for (int k = 0; k < 16; k++) t[k] = reinterpret_cast<const f32x4*>(w4)[k * plane4 + i];
#pragma unroll
  for (int a = 0; a < 3; a++)
#pragma unroll
    for (int b = 0; b < 3; b++) {
      const f32x4 v = (t[a * 4 + b] + t[a * 4 + b + 1] + t[(a + 1) * 4 + b] + t[(a + 1) * 4 + b + 1]) * 0.25f;
      reinterpret_cast<f32x4*>(dw)[(a * 3 + b) * plane4 + i] += v;
    }
}

// dw[i][j] += 1/4 sum_{s,t in {0,1}} sum_split slab[split][(i+s)*4 + j+t]      (one thread per (3x3 tap, float4 of the plane))
__global__ void wgrad_cpool_fold_slabs_kernel(const float* __restrict__ ws, float* __restrict__ dw, long plane4, int splits) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= plane4) return;
  const int tap = blockIdx.y, ti = tap / 3, tj = tap % 3;
  f32x4 t = {0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < splits; sp += 2) {          // 8 independent loads per batch (2 splits x 4 source taps), fixed order
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int s2 = min(sp + (u >> 2), splits - 1);
      const int src = (ti + ((u >> 1) & 1)) * 4 + tj + (u & 1);
      v[u] = reinterpret_cast<const f32x4*>(ws)[((long)s2 * 16 + src) * plane4 + i];
    }
#pragma unroll
    for (int u = 0; u < 8; u++)
      if (sp + (u >> 2) < splits) { t[0] += v[u][0]; t[1] += v[u][1]; t[2] += v[u][2]; t[3] += v[u][3]; }
  }
  f32x4 o = reinterpret_cast<f32x4*>(dw)[(long)tap * plane4 + i];
  o[0] += 0.25f * t[0]; o[1] += 0.25f * t[1]; o[2] += 0.25f * t[2]; o[3] += 0.25f * t[3];
  reinterpret_cast<f32x4*>(dw)[(long)tap * plane4 + i] = o;
}

static bool wgrad_cpool_rows_ok(int N, int Hp, int Wp, int Cin, int Cout) {
  static const int env = gank_tune("GANK_CPOOL_ROWS", 1);   // experiment knob: GANK_CPOOL_ROWS=0 falls back to the 16-tap per-tap kernel
  return env && Hp >= 8 && Wp >= 8 && log2_or_neg(Hp) >= 0 && log2_or_neg(Wp) >= 0 && Cin % 64 == 0 && Cout % 64 == 0 &&
         (long)N * 4 * Hp * Wp * Cin < (1L << 30) && (long)N * Hp * Wp * Cout < (1L << 30);
}
static void wgrad_cpool_rows_geometry(WgradArgs& a) {
  a.tiles_ci = a.Cin / 64;
  a.tiles_co = a.Cout / 64;
  const int total_steps = a.M / 64;
  const int tiles = a.tiles_ci * a.tiles_co * 4;
  static const int target = gank_tune("GANK_CPOOL_ROWS_TARGET", 256);   // experiment knob
  int splits = wgrad_round_down_env() ? target / tiles : (target + tiles - 1) / tiles;
  if (splits > total_steps / 4) splits = total_steps / 4;
  if (splits < 1) splits = 1;
  a.steps_per_split = cdiv(total_steps, splits);
  a.splits = cdiv(total_steps, a.steps_per_split);
}

extern "C" long gank_convpool3x3_wgrad_ws_elems(int N, int Hp, int Wp, int Cin, int Cout) {
  if (!wgrad_cpool_rows_ok(N, Hp, Wp, Cin, Cout)) return 16L * Cin * Cout;
  WgradArgs a{};
  a.Cin = Cin; a.Cout = Cout; a.M = N * Hp * Wp;
  wgrad_cpool_rows_geometry(a);
  return 16L * Cin * Cout * a.splits;
}

static int convpool3x3_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                                  int Cin, int Cout, int flags, gank_slab_job* job, void* stream);
extern "C" int gank_convpool3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                                      int Cin, int Cout, int flags, void* stream) {
  return convpool3x3_wgrad_impl(x, dy, dw, dbias, ws16, ws_elems, N, Hp, Wp, Cin, Cout, flags, nullptr, stream);
}
extern "C" int gank_convpool3x3_wgrad_job(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                                          int Cin, int Cout, int flags, gank_slab_job* job, void* stream) {
  GANK_REQUIRE(job, "convpool3x3_wgrad_job: null job");
  *job = gank_slab_job{nullptr, dw, 9L * Cin * Cout, 16L * Cin * Cout, 0, 0.25f, 1};
  return convpool3x3_wgrad_impl(x, dy, dw, dbias, ws16, ws_elems, N, Hp, Wp, Cin, Cout, flags, job, stream);
}
static int convpool3x3_wgrad_impl(const void* x, const void* dy, float* dw, float* dbias, float* ws16, long ws_elems, int N, int Hp, int Wp,
                                  int Cin, int Cout, int flags, gank_slab_job* job, void* stream) {
  GANK_REQUIRE(x && dy && dw && ws16, "convpool3x3_wgrad: null pointer");
  GANK_REQUIRE(Cin % 4 == 0 && Cout % 4 == 0, "convpool3x3_wgrad: channels must be multiples of 4");
  hipStream_t s = (hipStream_t)stream;
  if (wgrad_cpool_rows_ok(N, Hp, Wp, Cin, Cout)) {
    WgradArgs a{};
    a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = dw; a.dbias = dbias; a.ws = ws16;
    a.N = N; a.H = Hp; a.W = Wp; a.Hx = 2 * Hp; a.Wx = 2 * Wp; a.Hdy = Hp; a.Wdy = Wp;
    a.Cin = Cin; a.Cout = Cout; a.M = N * Hp * Wp;
    a.sw = log2_or_neg(Wp); a.shw = log2_or_neg(Hp * Wp);
    wgrad_cpool_rows_geometry(a);
    GANK_REQUIRE(ws_elems >= 16L * Cin * Cout * a.splits, "convpool3x3_wgrad: workspace of %ld floats, need %ld (gank_convpool3x3_wgrad_ws_elems)",
                 ws_elems, 16L * Cin * Cout * a.splits);
    gank_prof_begin(1, 2.0 * a.M * 16.0 * Cin * Cout, s, 2.0 * ((double)N * 4 * Hp * Wp * Cin + (double)a.M * Cout) + 36.0 * Cin * Cout);
    gank_prof_tag(1, (job && a.splits <= 16 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) ? "conv_wgrad_rows_kernel<1, 2, true>"
                                                                                           : "conv_wgrad_rows_kernel<1, 2, true> + wgrad_cpool_fold_slabs_kernel");
    const int kg = wgrad_rows_kg();
    const size_t lds = wgrad_rows_lds<true>(kg);
    a.scale = 1.f;
    static const int pf = gank_tune("GANK_CPOOL_ROWS_PF", 2);   // experiment knob: register prefetch depth
#ifdef GANK_TUNING
    auto kern = (flags & GANK_IN_RELU) ? (pf == 3 ? conv_wgrad_rows_kernel<1, 3, true> : conv_wgrad_rows_kernel<1, 2, true>)
                                       : (pf == 3 ? conv_wgrad_rows_kernel<0, 3, true> : conv_wgrad_rows_kernel<0, 2, true>);
#else
    (void)pf;
    auto kern = (flags & GANK_IN_RELU) ? conv_wgrad_rows_kernel<1, 2, true> : conv_wgrad_rows_kernel<0, 2, true>;
#endif
    static_assert((size_t)2 * 2 * (CPR_XSUB + SUBS) * sizeof(bf16) <= 65536, "below the default dynamic-LDS limit: no attribute call");
    if (kg == 2) kern = (flags & GANK_IN_RELU) ? conv_wgrad_rows_kernel<1, 2, true, 2> : conv_wgrad_rows_kernel<0, 2, true, 2>;
    if (kg == 3) kern = (flags & GANK_IN_RELU) ? conv_wgrad_rows_kernel<1, 2, true, 3> : conv_wgrad_rows_kernel<0, 2, true, 3>;
    a.xcd = wgrad_xcd_env();
    hipLaunchKernelGGL(kern, dim3((unsigned)(4 * a.tiles_ci * a.tiles_co * a.splits)), dim3(kg == 1 ? 256 : 512), lds, s, a);
    const long plane4 = (long)Cin * Cout / 4;
    if (job && a.splits <= 16 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0) {        // the fold is left to the caller's summing launch
      job->slabs = ws16;
      job->nslabs = a.splits;
    } else {
      hipLaunchKernelGGL(wgrad_cpool_fold_slabs_kernel, dim3((unsigned)cdiv(plane4, 256), 9), dim3(256), 0, s, ws16, dw, plane4, a.splits);
    }
    gank_prof_end(1, s);
    GANK_LAUNCH_OK("convpool3x3_wgrad");
    return 0;
  }
  GANK_REQUIRE(ws_elems >= 16L * Cin * Cout, "convpool3x3_wgrad: workspace of %ld floats, need %ld", ws_elems, 16L * Cin * Cout);
  const long n4 = 16L * Cin * Cout / 4;
  hipLaunchKernelGGL(wgrad_zero_kernel, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, s, ws16, n4);
  WgradArgs a{};
  a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.dw = ws16; a.dbias = dbias;
  a.N = N; a.H = Hp; a.W = Wp;
  a.Hx = 2 * Hp; a.Wx = 2 * Wp; a.Hdy = Hp; a.Wdy = Wp;
  a.Cin = Cin; a.Cout = Cout; a.ks = 4; a.pad = 1;
  a.flags = WG_X_STRIDE2 | (flags & GANK_IN_RELU);
  a.scale = 1.f;
  if (gank_wgrad_dispatch(a, s)) return 1;
  const long plane4 = (long)Cin * Cout / 4;
  hipLaunchKernelGGL(wgrad_fold4x4_kernel, dim3((unsigned)cdiv(plane4, 256)), dim3(256), 0, s, ws16, dw, plane4);
  GANK_LAUNCH_OK("convpool3x3_wgrad");
  return 0;
}

// ---- filter gradient of UpsampleConv 3x3 (nearest-neighbour 2x upsample, then the 3x3 conv: gan_cifar_resnet.py:138-153) in its
// phase form.  With u = the conv output position, the output phase (a, b) = (u_y & 1, u_x & 1) reads the LOW-resolution input
// through 2 x 2 taps (gank_upconv3x3_fprop), so the gradient of those 16 (phase, tap) matrices is
//     D4[kh][kw][co][ci] = sum_{n,y,x} dy[n, 2y + kh - 1, 2x + kw - 1, co] * x_low[n, y, x, ci],   kh = 3 - 2i - a, kw = 3 - 2j - b
// -- exactly the 16-tap stride-2 filter gradient of ConvMeanPool with the operands swapped (its "x" := dy at 2H x 2W, its "dy"
// := x at H x W): the rows kernel above computes it at 4/9 of the multiply-adds of the all-taps kernel on the upsampled input.
// The 3 x 3 taps then collect the phase taps they fed: dW3[t][s] += sum_{kh in {2-t, 3-t}} sum_{kw in {2-s, 3-s}} D4[kh][kw]^T.
// One thread block transposes a 32 x 32 (co, ci) tile through LDS so that both the slab reads and the dw updates are coalesced.
__global__ __launch_bounds__(256) void wgrad_upconv_fold_slabs_kernel(const float* __restrict__ ws, float* __restrict__ dw, int Cin, int Cout, int splits) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32;
  const int tap = blockIdx.z, t = tap / 3, sidx = tap % 3;
  const long plane = (long)Cin * Cout;
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int co = co0 + ty + 8 * rr, ci = ci0 + tx;
    float acc = 0.f;
    for (int sp0 = 0; sp0 < splits; sp0 += 4) {         // fixed order: deterministic; 16 loads requested per trip (one split per trip was a
      float v[4][4];                                    // chain of dependent round trips: 10.5 us for 17 MB)
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int sp = sp0 + u < splits ? sp0 + u : splits - 1;
        const float* base = ws + (long)sp * 16 * plane + (long)co * Cin + ci;
        v[u][0] = base[((2 - t) * 4 + (2 - sidx)) * plane]; v[u][1] = base[((2 - t) * 4 + (3 - sidx)) * plane];
        v[u][2] = base[((3 - t) * 4 + (2 - sidx)) * plane]; v[u][3] = base[((3 - t) * 4 + (3 - sidx)) * plane];
      }
#pragma unroll
      for (int u = 0; u < 4; u++)
        if (sp0 + u < splits) acc += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    tile[ty + 8 * rr][tx] = acc;
  }
  __syncthreads();
#pragma unroll
  for (int rr = 0; rr < 4; rr++) {
    const int ci = ci0 + ty + 8 * rr, co = co0 + tx;
    dw[((long)tap * Cin + ci) * Cout + co] += tile[tx][ty + 8 * rr];
  }
}

static bool wgrad_upconv_ok(int N, int H, int W, int Cin, int Cout) {
  static const int env = gank_tune("GANK_UPCONV_WGRAD_PHASE", 1);   // experiment knob: 0 keeps the all-taps kernel on the upsampled input
  return env && Cin % 64 == 0 && Cout % 64 == 0 && wgrad_cpool_rows_ok(N, H, W, Cout, Cin);
}

extern "C" long gank_upconv3x3_wgrad_ws_elems(int N, int H, int W, int Cin, int Cout) {
  if (!wgrad_upconv_ok(N, H, W, Cin, Cout)) return 0;
  WgradArgs a{};
  a.Cin = Cout; a.Cout = Cin; a.M = N * H * W;
  wgrad_cpool_rows_geometry(a);
  return 16L * Cin * Cout * a.splits;
}

extern "C" int gank_upconv3x3_wgrad(const void* x_low, const void* dy, float* dw, float* ws16, long ws_elems, int N, int H, int W, int Cin, int Cout,
                                    void* stream) {
  GANK_REQUIRE(x_low && dy && dw && ws16, "upconv3x3_wgrad: null pointer");
  GANK_REQUIRE(wgrad_upconv_ok(N, H, W, Cin, Cout), "upconv3x3_wgrad: unsupported shape (N %d, %d x %d, %d -> %d channels): gank_upconv3x3_wgrad_ws_elems returns 0 for it",
               N, H, W, Cin, Cout);
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a{};
  a.x = (const bf16*)dy; a.dy = (const bf16*)x_low; a.dw = dw; a.dbias = nullptr; a.ws = ws16;          // operands swapped (above)
  a.N = N; a.H = H; a.W = W; a.Hx = 2 * H; a.Wx = 2 * W; a.Hdy = H; a.Wdy = W;
  a.Cin = Cout; a.Cout = Cin; a.M = N * H * W;
  a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  wgrad_cpool_rows_geometry(a);
  GANK_REQUIRE(ws_elems >= 16L * Cin * Cout * a.splits, "upconv3x3_wgrad: workspace of %ld floats, need %ld (gank_upconv3x3_wgrad_ws_elems)", ws_elems,
               16L * Cin * Cout * a.splits);
  gank_prof_begin(1, 2.0 * a.M * 16.0 * Cin * Cout, s, 2.0 * ((double)a.M * Cin + 4.0 * a.M * Cout) + 36.0 * Cin * Cout);
  gank_prof_tag(1, "conv_wgrad_rows_kernel<0, 2, true> + wgrad_upconv_fold_slabs_kernel");
  const int kg = wgrad_rows_kg();
  const size_t lds = wgrad_rows_lds<true>(kg);
  a.scale = 1.f;
  a.xcd = wgrad_xcd_env();
  auto kern = conv_wgrad_rows_kernel<0, 2, true>;
  if (kg == 2) kern = conv_wgrad_rows_kernel<0, 2, true, 2>;
  if (kg == 3) kern = conv_wgrad_rows_kernel<0, 2, true, 3>;
  hipLaunchKernelGGL(kern, dim3((unsigned)(4 * a.tiles_ci * a.tiles_co * a.splits)), dim3(kg == 1 ? 256 : 512), lds, s, a);
  hipLaunchKernelGGL(wgrad_upconv_fold_slabs_kernel, dim3(Cin / 32, Cout / 32, 9), dim3(256), 0, s, ws16, dw, Cin, Cout, a.splits);
  gank_prof_end(1, s);
  GANK_LAUNCH_OK("upconv3x3_wgrad");
  return 0;
}

// Two 3-channel-input layers of different geometry in one launch (the streaming kernel above): the critic's D.Block.1.Conv1
// (3x3 on the 32x32 image) and D.Block.1.Shortcut (1x1 on the pooled 16x16 image).  Falls back to two gank_conv2d_wgrad calls
// when a layer is outside the kernel's shapes.
extern "C" int gank_conv2d_wgrad_narrow_pair(const void* x0, const void* dy0, float* dw0, float* db0, int N0, int H0, int W0, int Cout0, int ks0,
                                             const void* x1, const void* dy1, float* dw1, float* db1, int N1, int H1, int W1, int Cout1, int ks1,
                                             float scale, void* stream) {
  GANK_REQUIRE(x0 && dy0 && dw0 && x1 && dy1 && dw1, "conv2d_wgrad_narrow_pair: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (wgrad_narrow_stream_ok(N0, H0, W0, 3, Cout0, ks0) && wgrad_narrow_stream_ok(N1, H1, W1, 3, Cout1, ks1)) {
    const NarrowWgArgs a0 = narrow_args(x0, dy0, dw0, db0, N0, H0, W0, Cout0, scale), a1 = narrow_args(x1, dy1, dw1, db1, N1, H1, W1, Cout1, scale);
    const double m0 = (double)a0.M, m1 = (double)a1.M;
    gank_prof_begin(1, 2.0 * (m0 * Cout0 * ks0 * ks0 * 3 + m1 * Cout1 * ks1 * ks1 * 3), s,
                    2.0 * (m0 * (3 + Cout0) + m1 * (3 + Cout1)) + 4.0 * 3 * (ks0 * ks0 * Cout0 + ks1 * ks1 * Cout1));
    gank_prof_tag(1, "conv_wgrad_narrow_stream_kernel (pair)");
    const int rc = launch_wgrad_narrow_stream(a0, ks0, a1, ks1, s);
    gank_prof_end(1, s);
    return rc;
  }
  if (gank_conv2d_wgrad(x0, dy0, dw0, db0, nullptr, 0, N0, H0, W0, 3, Cout0, ks0, 0, scale, stream)) return 1;
  return gank_conv2d_wgrad(x1, dy1, dw1, db1, nullptr, 0, N1, H1, W1, 3, Cout1, ks1, 0, scale, stream);
}

// Same-shape layers in one launch.  The critic's 8x8x128 residual blocks (D.Block.3/4, four 3x3 128->128 convs) have
// filter gradients of 2.4 GFLOP each that fill 144 workgroups for a few microseconds; issued together (grid.y = layer)
// they overlap each other's latency instead of queueing behind one another.
static int wgrad_batched_impl(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout, int ksize, int flags, float scale,
                              float* ws, long ws_elems, gank_slab_job* jobs, void* stream);
extern "C" int gank_conv2d_wgrad_batched(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout,
                                         int ksize, int flags, float scale, void* stream) {
  return wgrad_batched_impl(items, count, N, H, W, Cin, Cout, ksize, flags, scale, nullptr, 0, nullptr, stream);
}
static bool wgrad_batched_geometry(WgradArgs& a, int N, int H, int W, int Cin, int Cout, int ksize, int flags, float scale, bool& rows) {
  a.N = N; a.H = H; a.W = W; a.Hx = H; a.Wx = W; a.Hdy = H; a.Wdy = W;
  a.Cin = Cin; a.Cout = Cout; a.ks = ksize; a.pad = (ksize - 1) / 2; a.taps = ksize * ksize;
  a.flags = flags & GANK_IN_RELU;
  a.scale = scale;
  a.M = N * H * W; a.sw = log2_or_neg(W); a.shw = log2_or_neg(H * W);
  const bool batchable = (flags & ~GANK_IN_RELU) == 0 && Cin % 8 == 0 && Cout % 8 == 0 && Cin >= 128 && Cout >= 128 && a.sw >= 0 &&
                         a.shw >= 0 && a.M % 64 == 0 && !wgrad_taps_ok(a) && (long)a.M * (Cin > Cout ? Cin : Cout) < (1L << 30);
  rows = batchable && wgrad_rows_ok(a);
  return batchable;
}
extern "C" long gank_conv2d_wgrad_batched_ws_elems(int count, int N, int H, int W, int Cin, int Cout, int ksize, int flags) {
  if (count <= 0 || ksize != 3) return 0;
  WgradArgs a{};
  bool rows = false;
  wgrad_batched_geometry(a, N, H, W, Cin, Cout, ksize, flags, 1.f, rows);
  if (!rows) return 0;
  long total = 0;
  for (int i = 0; i < count; i += 4) {
    const int nb = count - i < 4 ? count - i : 4;
    total += (long)nb * wgrad_rows_splits(a, nb) * 9 * Cin * Cout;
  }
  return total;
}
extern "C" int gank_conv2d_wgrad_batched_slabs(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout, int ksize,
                                               int flags, float scale, float* ws, long ws_elems, gank_slab_job* jobs, void* stream) {
  GANK_REQUIRE(ws && jobs, "conv2d_wgrad_batched_slabs: null workspace / job list");
  GANK_REQUIRE(ws_elems >= gank_conv2d_wgrad_batched_ws_elems(count, N, H, W, Cin, Cout, ksize, flags) &&
               gank_conv2d_wgrad_batched_ws_elems(count, N, H, W, Cin, Cout, ksize, flags) > 0,
               "conv2d_wgrad_batched_slabs: this geometry has no slab form, or the workspace is smaller than gank_conv2d_wgrad_batched_ws_elems");
  return wgrad_batched_impl(items, count, N, H, W, Cin, Cout, ksize, flags, scale, ws, ws_elems, jobs, stream);
}
static int wgrad_batched_impl(const gank_wgrad_item* items, int count, int N, int H, int W, int Cin, int Cout, int ksize, int flags, float scale,
                              float* ws, long ws_elems, gank_slab_job* jobs, void* stream) {
  GANK_REQUIRE(items && count > 0, "conv2d_wgrad_batched: empty list");
  GANK_REQUIRE(ksize % 2 == 1, "conv2d_wgrad_batched: even filter sizes are not on this path (ksize=%d)", ksize);
  hipStream_t s = (hipStream_t)stream;
  WgradArgs a{};
  bool rows = false;
  const bool batchable = wgrad_batched_geometry(a, N, H, W, Cin, Cout, ksize, flags, scale, rows);
  long ws_used = 0;
  int i = 0;
  while (batchable && count - i >= (rows ? 1 : 2)) {
    const int nb = count - i < 4 ? count - i : 4;
    const long slab = 9L * Cin * Cout;
    const int splits = (ws && rows) ? wgrad_rows_splits(a, nb) : 0;
    for (int j = 0; j < nb; j++) {
      GANK_REQUIRE(items[i + j].x && items[i + j].dy && items[i + j].dw, "conv2d_wgrad_batched: null pointer in item %d", i + j);
      a.xs[j] = (const bf16*)items[i + j].x; a.dys[j] = (const bf16*)items[i + j].dy;
      a.dws[j] = items[i + j].dw; a.dbs[j] = items[i + j].dbias;
      a.wss[j] = nullptr;
      if (ws && rows) {        // partial tiles to slabs [split][9][Cin][Cout]; the caller sums them (gank_sum_slabs)
        a.wss[j] = ws + ws_used;
        jobs[i + j] = gank_slab_job{a.wss[j], items[i + j].dw, slab, slab, splits, scale, 0};
        ws_used += (long)splits * slab;
      }
    }
    for (int j = nb; j < 4; j++) a.wss[j] = nullptr;
    a.nbatch = nb;
    a.x = a.xs[0]; a.dy = a.dys[0]; a.dw = a.dws[0]; a.dbias = a.dbs[0];
    gank_prof_begin(1, 2.0 * nb * a.M * (double)Cout * a.taps * Cin, s, nb * (2.0 * a.M * ((double)Cin + Cout) + 4.0 * a.taps * Cin * Cout));
    const int rc = rows ? launch_wgrad_rows(a, s) : launch_wgrad_lean<2, 2, 2, 2, 2>(a, s);
    gank_prof_end(1, s);
    if (rc) return rc > 0 ? rc : gank_set_error("conv2d_wgrad_batched: mode not instantiated");
    i += nb;
  }
  for (; i < count; i++)
    if (gank_conv2d_wgrad(items[i].x, items[i].dy, items[i].dw, items[i].dbias, nullptr, 0, N, H, W, Cin, Cout, ksize, flags, scale, stream)) return 1;
  return 0;
}

// Deconv2D filter gradient: dF[a,b,co,ci] = sum dy[n,2p+a-pt,2q+b-pl,co] * x[n,p,q,ci]
// = this engine with the gathered operand := dy (stride 2) and the plain operand := x.
extern "C" int gank_deconv2d_wgrad(const void* x, const void* dy, float* df, int N, int H, int W, int Cin, int Cout,
                                   int ksize, void* stream) {
  WgradArgs a{};
  a.x = (const bf16*)dy; a.dy = (const bf16*)x; a.dw = df;
  a.N = N; a.H = H; a.W = W;
  a.Hx = 2 * H; a.Wx = 2 * W; a.Hdy = H; a.Wdy = W;
  a.Cin = Cout; a.Cout = Cin; a.ks = ksize;
  int t = ksize - 2; if (t < 0) t = 0;
  a.pad = t / 2;
  a.flags = WG_X_STRIDE2;
  a.scale = 1.f;
  return gank_wgrad_dispatch(a, (hipStream_t)stream);
}
