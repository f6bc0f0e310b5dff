// Weight-operand preparation: device code shared by the standalone / batched preparation launches (elementwise.hip)
// and by the spectral-norm forward launch that scales and prepares in one pass (sn.hip).
//
// SC (template): every element read from the master weight is divided by `sg` first -- the spectral norm's sigma -- so
// the operands are built from W / sigma without W_bar being written and read back (same arithmetic: one fp32 division
// per element, then exactly the operations of the unscaled form).
#pragma once
#include "gank_common.h"

template <bool SC>
__device__ __forceinline__ float prep_ld(float x, float sg) {
  if constexpr (SC) return x / sg;
  else return x;
}

// ---- NN-upsample + 3x3 conv as a 4x4 stride-2 transposed conv: phase operand matrices and the
// combined 4x4 kernel for the input gradient (see gank_upconv3x3_fprop / _dgrad).
//   wph[p=(a,b)][co][(i*2+j)*Cin+ci] = sum_{dh in R(a,i)} sum_{dw in R(b,j)} w[dh][dw][ci][co]
//       R(0,0)={0}  R(0,1)={1,2}  R(1,0)={0,1}  R(1,1)={2}
//   wd4[ci][(u*4+v)*Cout+co]         = sum_{dh in S(u)} sum_{dw in S(v)} w[dh][dw][ci][co]
//       S(0)={2}  S(1)={1,2}  S(2)={0,1}  S(3)={0}
__device__ __forceinline__ void up_range_R(int a, int i, int& lo, int& hi) {
  if (a == 0) { lo = i == 0 ? 0 : 1; hi = i == 0 ? 0 : 2; } else { lo = i == 0 ? 0 : 2; hi = i == 0 ? 1 : 2; }
}
__device__ __forceinline__ void up_range_S(int u, int& lo, int& hi) {
  lo = u == 0 ? 2 : (u == 1 ? 1 : 0);
  hi = u == 0 ? 2 : (u == 1 ? 2 : (u == 2 ? 1 : 0));
}
// Sum of the 3x3 taps [h0..h1] x [w0..w1] (one or two per axis) of one (ci, co) element; p = &w[0][0][ci][co], plane = Cin*Cout.
// FOUR UNCONDITIONAL loads (a repeated address where a range holds one tap) and 0/1 factors, in the loop's order of addition
// (bit-identical): run-time loop bounds made each tap its own load -> wait -> add round trip, up to 4 in a row per element and
// 32 per thread of the one-chunk-per-thread layouts -- the critic's 1.7 M weights took 17 us, all of it latency.
template <bool SC = false>
__device__ __forceinline__ float sum_taps(const float* __restrict__ p, long plane, bool flip, int h0, int h1, int w0, int w1, float sg = 1.f) {
  const float mh = h1 > h0 ? 1.f : 0.f, mw = w1 > w0 ? 1.f : 0.f;
  const int t00 = flip ? (2 - h0) * 3 + (2 - w0) : h0 * 3 + w0, t01 = flip ? (2 - h0) * 3 + (2 - w1) : h0 * 3 + w1;
  const int t10 = flip ? (2 - h1) * 3 + (2 - w0) : h1 * 3 + w0, t11 = flip ? (2 - h1) * 3 + (2 - w1) : h1 * 3 + w1;
  const float a = prep_ld<SC>(p[t00 * plane], sg), b = prep_ld<SC>(p[t01 * plane], sg), c = prep_ld<SC>(p[t10 * plane], sg), d = prep_ld<SC>(p[t11 * plane], sg);
  return ((a + mw * b) + mh * c) + (mh * mw) * d;
}

// One kernel builds both operands.  `ph` = phase matrix [4][CrP pad][4*CkP], `d4` = combined 4x4 matrix
// [CrD pad][roundup(16*CkD,64)]; (sr, sk) are the strides of the row / inner channel in w's [ci][co] plane, so the
// same code serves UpsampleConv (ph rows = co, d4 rows = ci) and ConvMeanPool (ph rows = ci, d4 rows = co, the
// 3x3 taps flipped, everything scaled by 1/4 -- see gank_convpool3x3_prep_weights).
struct PrepUpArgs {
  const float* w;
  bf16* ph;
  bf16* d4;
  int CrP, CkP, srP, skP, CrPpad;
  int CrD, CkD, srD, skD, CrDpad, Kpad4;
  int flip, plane;     // plane = Cin*Cout (stride of one 3x3 tap)
  float scale;
};

template <bool SC = false>
__device__ __forceinline__ void prep_up_element(const PrepUpArgs& q, long idx, float sg = 1.f) {
  const long nph = 4L * q.CrPpad * 4 * q.CkP;
  if (idx < nph) {
    const int k = (int)(idx % (4 * q.CkP));
    long t = idx / (4 * q.CkP);
    const int r = (int)(t % q.CrPpad), p = (int)(t / q.CrPpad);
    const int tap = k / q.CkP, c = k - tap * q.CkP;
    float v = 0.f;
    if (r < q.CrP) {
      int h0, h1, w0, w1;
      up_range_R(p >> 1, tap >> 1, h0, h1);
      up_range_R(p & 1, tap & 1, w0, w1);
      v = sum_taps<SC>(q.w + (long)r * q.srP + (long)c * q.skP, q.plane, q.flip != 0, h0, h1, w0, w1, sg);
    }
    q.ph[idx] = f2bf(v * q.scale);
  } else {
    const long i2 = idx - nph;
    const int r = (int)(i2 / q.Kpad4), k = (int)(i2 - (long)r * q.Kpad4);
    float v = 0.f;
    if (r < q.CrD && k < 16 * q.CkD) {
      const int tap = k / q.CkD, c = k - tap * q.CkD;
      int h0, h1, w0, w1;
      up_range_S(tap >> 2, h0, h1);
      up_range_S(tap & 3, w0, w1);
      v = sum_taps<SC>(q.w + (long)r * q.srD + (long)c * q.skD, q.plane, q.flip != 0, h0, h1, w0, w1, sg);
    }
    q.d4[i2] = f2bf(v * q.scale);
  }
}
// Eight consecutive elements of the 4x4 matrix d4 per thread where its inner index runs along w's fast axis (skD == 1:
// UpsampleConv) and a tap's row is a whole number of 8-element pieces: 16-byte loads and one 16-byte store instead of
// eight 4-byte loads and 2-byte stores per tap (the generator's 8.4 M-element preparation was 43 of its 54 us).
__device__ __forceinline__ bool prep_up_vec8_ok(const PrepUpArgs& q) { return q.skD == 1 && (q.CkD & 7) == 0 && q.Kpad4 == 16 * q.CkD; }
template <bool SC = false>
__device__ __forceinline__ void prep_up_d4_vec8(const PrepUpArgs& q, long i2, float sg = 1.f) {       // i2 % 8 == 0, inside d4
  const int r = (int)(i2 / q.Kpad4), k = (int)(i2 - (long)r * q.Kpad4);
  bf16x8 o;
  if (r < q.CrD) {
    const int tap = k / q.CkD, c = k - tap * q.CkD;
    int h0, h1, w0, w1;
    up_range_S(tap >> 2, h0, h1);
    up_range_S(tap & 3, w0, w1);
    const float mh = h1 > h0 ? 1.f : 0.f, mw = w1 > w0 ? 1.f : 0.f;
    const bool flip = q.flip != 0;
    const int t00 = flip ? (2 - h0) * 3 + (2 - w0) : h0 * 3 + w0, t01 = flip ? (2 - h0) * 3 + (2 - w1) : h0 * 3 + w1;
    const int t10 = flip ? (2 - h1) * 3 + (2 - w0) : h1 * 3 + w0, t11 = flip ? (2 - h1) * 3 + (2 - w1) : h1 * 3 + w1;
    const float* p = q.w + (long)r * q.srD + c;
#pragma unroll
    for (int half = 0; half < 2; half++) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + (long)t00 * q.plane + 4 * half);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p + (long)t01 * q.plane + 4 * half);
      const f32x4 cc = *reinterpret_cast<const f32x4*>(p + (long)t10 * q.plane + 4 * half);
      const f32x4 d = *reinterpret_cast<const f32x4*>(p + (long)t11 * q.plane + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; e++)
        o[4 * half + e] = f2bf((((prep_ld<SC>(a[e], sg) + mw * prep_ld<SC>(b[e], sg)) + mh * prep_ld<SC>(cc[e], sg)) + (mh * mw) * prep_ld<SC>(d[e], sg)) * q.scale);   // sum_taps' order
    }
  } else {
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(0.f);
  }
  *reinterpret_cast<bf16x8*>(q.d4 + i2) = o;
}

__host__ __device__ inline long prep_up_total(const PrepUpArgs& q) { return 4L * q.CrPpad * 4 * q.CkP + (long)q.CrDpad * q.Kpad4; }

// The operand whose inner index runs along w's SLOW channel axis (sr == 1: the phase matrix of UpsampleConv, the
// 4x4 matrix of ConvMeanPool) read element-wise is a stride-Cout gather (the 16.8 M-element generator preparation took
// 91 us).  Here one block turns a 32 x 32 (row, inner) tile of one (phase, tap) / tap slice through LDS: coalesced reads
// along the rows, coalesced writes along the inner index.  Needs Ck % 64 == 0 and no K padding.
__host__ __device__ inline int prep_up_tiles(const PrepUpArgs& q, bool ph) {
  return ph ? 16 * (q.CrPpad / 32) * (q.CkP / 64) : 16 * (q.CrDpad / 32) * (q.CkD / 64);
}
template <bool SC = false>
__device__ __forceinline__ void prep_up_tile(const PrepUpArgs& q, bool ph, int tile, float (*tl)[33], float sg = 1.f) {     // tl: [64][33]
  // a 64 (inner index c) x 32 (row r) tile: reads along r (128-byte runs of w's fast axis), writes along c in 16-byte pieces
  const int Cr = ph ? q.CrP : q.CrD, Ck = ph ? q.CkP : q.CkD, CrPad = ph ? q.CrPpad : q.CrDpad, sk = ph ? q.skP : q.skD;
  const int tk = Ck / 64, tr = CrPad / 32;
  const int c0 = (tile % tk) * 64, r0 = ((tile / tk) % tr) * 32, slice = tile / (tk * tr);      // slice 0..15
  int h0, h1, w0, w1;
  if (ph) { up_range_R((slice >> 2) >> 1, (slice & 3) >> 1, h0, h1); up_range_R((slice >> 2) & 1, slice & 1, w0, w1); }
  else { up_range_S(slice >> 2, h0, h1); up_range_S(slice & 3, w0, w1); }
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
  for (int i = ty; i < 64; i += 8) {
    const int c = c0 + i, r = r0 + tx;
    float v = 0.f;
    if (r < Cr) v = sum_taps<SC>(q.w + r + (long)c * sk, q.plane, q.flip != 0, h0, h1, w0, w1, sg);
    tl[i][tx] = v * q.scale;
  }
  __syncthreads();
  const int ri = threadIdx.x >> 3, cp = (threadIdx.x & 7) * 8;
  const int r = r0 + ri, c = c0 + cp;
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; j++) o[j] = f2bf(tl[cp + j][ri]);
  if (ph) *reinterpret_cast<bf16x8*>(q.ph + ((long)((slice >> 2) * CrPad + r) * 4 + (slice & 3)) * Ck + c) = o;
  else *reinterpret_cast<bf16x8*>(q.d4 + (long)r * q.Kpad4 + (long)slice * Ck + c) = o;
}

// block split shared by the standalone and the batched launchers
struct PrepUpSplit { int ntiles, nelem; long lo, hi; };
static inline PrepUpSplit prep_up_split(const PrepUpArgs& q, int kind) {
  const bool ph = kind == 1;          // which matrix has the strided source: kind 1 the phase matrix, kind 2 the 4x4 one
  const bool tiled = ph ? (q.srP == 1 && q.CkP % 64 == 0) : (q.srD == 1 && q.CkD % 64 == 0 && q.Kpad4 == 16 * q.CkD);
  const long nph = 4L * q.CrPpad * 4 * q.CkP, total = prep_up_total(q);
  PrepUpSplit s;
  s.ntiles = tiled ? prep_up_tiles(q, ph) : 0;
  s.lo = tiled && ph ? nph : 0;
  s.hi = tiled && !ph ? nph : total;
  s.nelem = (int)cdiv(s.hi - s.lo, 2048);
  return s;
}

// kind 1 = UpsampleConv 3x3 (ph rows = co, d4 rows = ci); kind 2 = ConvMeanPool 3x3 (ph rows = ci, d4 rows = co, flipped, x 1/4)
__host__ __device__ inline PrepUpArgs prep_up_args(int kind, const float* w, void* ph, void* d4, int Cin, int Cout) {
  PrepUpArgs q{};
  q.w = w; q.ph = (bf16*)ph; q.d4 = (bf16*)d4; q.plane = Cin * Cout;
  if (kind == 1) {
    q.CrP = Cout; q.CkP = Cin; q.srP = 1; q.skP = Cout;
    q.CrD = Cin; q.CkD = Cout; q.srD = Cout; q.skD = 1;
    q.flip = 0; q.scale = 1.f;
  } else {
    q.CrP = Cin; q.CkP = Cout; q.srP = Cout; q.skP = 1;
    q.CrD = Cout; q.CkD = Cin; q.srD = 1; q.skD = Cout;
    q.flip = 1; q.scale = 0.25f;
  }
  q.CrPpad = (q.CrP + 31) / 32 * 32;
  q.CrDpad = (q.CrD + 31) / 32 * 32;
  q.Kpad4 = (16 * q.CkD + 63) / 64 * 64;
  return q;
}

// ---- batched: every conv/linear weight of a network in ONE launch (the per-layer form costs 2 tiny
// launches x ~5 us per layer per forward; a network has 11-12 weights).  Table by value in kernargs.
#define PREP_MAX 16
struct PrepTable {
  gank_prep_desc d[PREP_MAX];
  PrepUpArgs up[PREP_MAX];         // kinds 1/2: operand geometry, filled on the host
  int first_block[PREP_MAX + 1];   // prefix sum of blocks per entry
  int nwf[PREP_MAX];               // wf tiles of entry i (the rest of its blocks are wd work)
  int count;
};

// One block of the batched preparation: `bid` = block index inside the table's block range; sg = divisor of entry e's
// weight (SC) -- read by the caller from wherever its sigma lives.
__device__ __forceinline__ int prep_batch_entry(const PrepTable& t, int bid) {
  int e = 0;
  for (int i = 1; i < t.count; i++)
    if (bid >= t.first_block[i]) e = i;
  return e;
}
template <bool SC>
__device__ __forceinline__ void prep_batch_block(const PrepTable& t, int e, int bid, float sg) {
  const gank_prep_desc& d = t.d[e];
  const int b = bid - t.first_block[e];
  const int taps = d.ksize * d.ksize;
  if (d.kind == 1 || d.kind == 2) {       // UpsampleConv / ConvMeanPool 3x3 operands
    const PrepUpArgs& q = t.up[e];
    __shared__ float tt[64][33];
    const long nph = 4L * q.CrPpad * 4 * q.CkP, total = prep_up_total(q);
    if (b < t.nwf[e]) {                   // transposed tiles of the strided-source matrix (nwf = their count, or 0)
      prep_up_tile<SC>(q, d.kind == 1, b, tt, sg);
    } else {                              // the other matrix (or both when the tile path does not apply), element-wise
      long lo = 0, hi = total;
      if (t.nwf[e] > 0) { if (d.kind == 1) lo = nph; else hi = nph; }
      const long base = lo + (long)(b - t.nwf[e]) * 2048;
      if (lo >= nph && prep_up_vec8_ok(q)) {          // the block's 2048 elements all lie in d4 (lo = nph there, both multiples of 8)
        const long i = base + 8 * threadIdx.x;
        if (i < hi) prep_up_d4_vec8<SC>(q, i - nph, sg);
      } else {
        for (int j = 0; j < 8; j++) {
          const long i = base + j * 256 + threadIdx.x;
          if (i < hi) prep_up_element<SC>(q, i, sg);
        }
      }
    }
  } else if (d.kind == 4) {
    // "rfrag" operands of the resident kernels (conv_resident.hip): [32-row tile][tap][k/16][lane = h*32 + r][8], one
    // 16-byte chunk (8 consecutive k of one row) per thread.  wf rows = co, k = ci; wd rows = ci, k = co, taps flipped.
    const bool isf = b < t.nwf[e];
    const int rows = isf ? d.Cout : d.Cin, kc = isf ? d.Cin : d.Cout;
    const long nchunk = (long)(rows / 32) * taps * (kc / 16) * 64;
    bf16* dst = (bf16*)(isf ? d.wf : d.wd);
    const long base = (long)(isf ? b : b - t.nwf[e]) * 256;
    const long q = base + threadIdx.x;
    if (q < nchunk) {
      const int lane = (int)(q & 63);
      long u = q >> 6;
      const int kk = (int)(u % (kc / 16)); u /= (kc / 16);
      const int tap = (int)(u % taps);
      const int rt = (int)(u / taps);
      const int row = rt * 32 + (lane & 31), k0 = kk * 16 + (lane >> 5) * 8;
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const int k = k0 + j;
        const long cp = d.cin_pitch > 0 ? d.cin_pitch : d.Cin;       // the first Cin of cin_pitch input channels (a channel slice of a wider filter)
        o[j] = f2bf(prep_ld<SC>(isf ? d.w[((long)tap * cp + k) * d.Cout + row] : d.w[((long)(taps - 1 - tap) * cp + row) * d.Cout + k], sg));
      }
      *reinterpret_cast<bf16x8*>(dst + q * 8) = o;
    }
  } else if (d.kind == 5) {
    // ConvMeanPool 3x3 operands of the resident kernels (conv_resident.hip), one 16-byte chunk per thread:
    //   wf [Cout/32][Cin/64][16 taps][4 kk][64 lanes][8]   = W4[tap][ci][co]        (the 4x4 stride-2 kernel, x 1/4)
    //   wd [4 phases][Cin/32][4 taps][Cout/16 kk][64][8]   = Wph[phase][ci][tap,co]  (its transposed conv by output phase)
    // same tap algebra as kind 2 (gank_convpool3x3_prep_weights).
    const bool isf = b < t.nwf[e];
    const long q = (long)(isf ? b : b - t.nwf[e]) * 256 + threadIdx.x;
    const long nchunk = 2L * d.Cin * d.Cout;                 // 16 * Cin * Cout / 8 sixteen-byte chunks, both operands
    if (q < nchunk) {
      const int lane = (int)(q & 63), r = lane & 31, hh = lane >> 5;
      long u = q >> 6;
      bf16x8 o;
      if (isf) {
        const int kk = (int)(u & 3); u >>= 2;
        const int tap = (int)(u & 15); u >>= 4;
        const int nch = d.Cin >> 6;
        const int chunk = (int)(u % nch), co = (int)(u / nch) * 32 + r;
        int h0, h1, w0, w1;
        up_range_S(tap >> 2, h0, h1);
        up_range_S(tap & 3, w0, w1);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int ci = chunk * 64 + kk * 16 + hh * 8 + j;
          o[j] = f2bf(0.25f * sum_taps<SC>(d.w + (long)ci * d.Cout + co, (long)d.Cin * d.Cout, true, h0, h1, w0, w1, sg));
        }
        *reinterpret_cast<bf16x8*>((bf16*)d.wf + q * 8) = o;
      } else {
        const int nkk = d.Cout >> 4;
        const int kk = (int)(u % nkk); u /= nkk;
        const int tap = (int)(u & 3); u >>= 2;
        const int tiles = d.Cin >> 5;
        const int ci = (int)(u % tiles) * 32 + r, phase = (int)(u / tiles);
        int h0, h1, w0, w1;
        up_range_R(phase >> 1, tap >> 1, h0, h1);
        up_range_R(phase & 1, tap & 1, w0, w1);
#pragma unroll
        for (int j = 0; j < 8; j++) {
          const int co = kk * 16 + hh * 8 + j;
          o[j] = f2bf(0.25f * sum_taps<SC>(d.w + (long)ci * d.Cout + co, (long)d.Cin * d.Cout, true, h0, h1, w0, w1, sg));
        }
        *reinterpret_cast<bf16x8*>((bf16*)d.wd + q * 8) = o;
      }
    }
  } else if (b < t.nwf[e]) {
    // wf [CoutPad][Kpad] = w^T: a 64 (k) x 32 (cout) tile through LDS -- rows of w read along cout (128-byte runs), rows of wf
    // written along k in 16-byte pieces (one per thread; 2-byte stores before)
    const int K = taps * d.Cin, Kpad = (K + 63) / 64 * 64, CoutPad = (d.Cout + 31) / 32 * 32;
    __shared__ float tl[64][33];
    const int ntk = Kpad / 64;
    const int k0 = (b % ntk) * 64, c0 = (b / ntk) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int i = ty; i < 64; i += 8) {
      const int k = k0 + i, c = c0 + tx;
      tl[i][tx] = (k < K && c < d.Cout) ? prep_ld<SC>(d.w[(long)k * d.Cout + c], sg) : 0.f;
    }
    __syncthreads();
    bf16* wf = (bf16*)d.wf;
    const int c = c0 + (threadIdx.x >> 3), kp = (threadIdx.x & 7) * 8;       // CoutPad % 32 == 0, Kpad % 64 == 0: always inside
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(tl[kp + j][threadIdx.x >> 3]);
    *reinterpret_cast<bf16x8*>(wf + (long)c * Kpad + k0 + kp) = o;
  } else {
    const int Kpad2 = (taps * d.Cout + 63) / 64 * 64, CinPad = (d.Cin + 31) / 32 * 32;
    const long total = (long)CinPad * Kpad2;
    bf16* wd = (bf16*)d.wd;
    const long base = (long)(b - t.nwf[e]) * 2048;
    if ((d.Cout & 7) == 0) {
      // 8 consecutive k = 8 consecutive couts of one tap (or 8 pad columns): two 16-byte loads, one 16-byte store
      const long i = base + 8 * threadIdx.x;
      if (i < total) {
        const int ci = (int)(i / Kpad2), k = (int)(i - (long)ci * Kpad2);
        bf16x8 o;
        if (ci < d.Cin && k < taps * d.Cout) {
          const int tp = k / d.Cout, co = k - tp * d.Cout;
          const float* p = d.w + ((long)(taps - 1 - tp) * d.Cin + ci) * d.Cout + co;
          const f32x4 a = *reinterpret_cast<const f32x4*>(p), bb = *reinterpret_cast<const f32x4*>(p + 4);
#pragma unroll
          for (int u = 0; u < 4; u++) { o[u] = f2bf(prep_ld<SC>(a[u], sg)); o[4 + u] = f2bf(prep_ld<SC>(bb[u], sg)); }
        } else {
#pragma unroll
          for (int u = 0; u < 8; u++) o[u] = f2bf(0.f);
        }
        *reinterpret_cast<bf16x8*>(wd + i) = o;
      }
      return;
    }
    for (int j = 0; j < 8; j++) {
      const long i = base + j * 256 + threadIdx.x;
      if (i >= total) break;
      const int ci = (int)(i / Kpad2), k = (int)(i - (long)ci * Kpad2);
      float v = 0.f;
      if (ci < d.Cin && k < taps * d.Cout) {
        const int tp = k / d.Cout, co = k - tp * d.Cout;
        v = prep_ld<SC>(d.w[((long)(taps - 1 - tp) * d.Cin + ci) * d.Cout + co], sg);
      }
      wd[i] = f2bf(v);
    }
  }
}


// Host: fill a PrepTable from `count` (<= PREP_MAX) descriptors; returns the number of blocks, or -1 (error set).
static inline int prep_table_fill(PrepTable& t, const gank_prep_desc* table, int count, int base_index) {
  t = PrepTable{};
  t.count = count;
  int blocks = 0;
  for (int i = 0; i < t.count; i++) {
    const gank_prep_desc& d = table[i];
    const bool ok = d.w && (d.wf || d.wd) && d.ksize >= 1 && d.Cin > 0 && d.Cout > 0;
    if (!ok) { gank_set_error("prep_weights_batched: bad entry %d", base_index + i); return -1; }
    const bool kind_ok = d.kind == 0 || ((d.kind == 1 || d.kind == 2) && d.ksize == 3 && d.wf && d.wd) ||
                         (d.kind == 4 && d.Cin % 32 == 0 && d.Cout % 32 == 0 && (d.cin_pitch == 0 || d.cin_pitch >= d.Cin)) ||
                         (d.kind == 5 && d.ksize == 3 && d.wf && d.wd && d.Cin % 64 == 0 && d.Cout % 32 == 0);
    if (!kind_ok) {
      gank_set_error("prep_weights_batched: entry %d: kind %d needs ksize 3 and both outputs (1, 2, 5) / channels %% 32 == 0 (4) / Cin %% 64 == 0 (5)", base_index + i, d.kind);
      return -1;
    }
    t.d[i] = d;
    const int taps = d.ksize * d.ksize;
    int nwf = d.wf ? (roundup(taps * d.Cin, 64) / 64) * (roundup(d.Cout, 32) / 32) : 0;
    int nwd = d.wd ? cdiv((long)roundup(d.Cin, 32) * roundup(taps * d.Cout, 64), 2048) : 0;
    if (d.kind == 1 || d.kind == 2) {
      t.up[i] = d.kind == 1 ? prep_up_args(1, d.w, d.wf, d.wd, d.Cin, d.Cout) : prep_up_args(2, d.w, d.wd, d.wf, d.Cin, d.Cout);
      const PrepUpSplit sp = prep_up_split(t.up[i], d.kind);
      nwf = sp.ntiles;
      nwd = sp.nelem;
    }
    if (d.kind == 5) nwf = nwd = cdiv(2L * d.Cin * d.Cout, 256);
    if (d.kind == 4) {          // one 16-byte chunk per thread
      nwf = d.wf ? cdiv((long)d.Cout * taps * d.Cin / 8, 256) : 0;
      nwd = d.wd ? cdiv((long)d.Cin * taps * d.Cout / 8, 256) : 0;
    }
    t.first_block[i] = blocks;
    t.nwf[i] = nwf;
    blocks += nwf + nwd;
  }
  t.first_block[t.count] = blocks;
  return blocks;
}
