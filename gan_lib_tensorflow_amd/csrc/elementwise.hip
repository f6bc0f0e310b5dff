// HBM-bound glue kernels of the block library: weight layout preparation, resampling, activations,
// pooling, concat/tile, embedding, casts, column sums.  All bf16 traffic is 16 B per lane whenever the
// channel count allows (C % 8 == 0); a scalar path covers the 3-channel image side and odd sizes.
#include "gank_common.h"

static inline dim3 grid1d(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

// ------------------------------------------------------------------------------------------------
// weight preparation (see gank.h)
// ------------------------------------------------------------------------------------------------
// wf[co][k] = w[k][co]  (k = tap*Cin+ci), zero padded to [CoutPad][Kpad]: 32x32 LDS-tiled transpose
__global__ void prep_wf_kernel(const float* __restrict__ w, bf16* __restrict__ wf, int K, int Cout, int CoutPad, int Kpad) {
  __shared__ float t[32][33];
  const int k0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: 32 x 8
  for (int i = ty; i < 32; i += 8) {
    const int k = k0 + i, c = c0 + tx;
    t[i][tx] = (k < K && c < Cout) ? w[(long)k * Cout + c] : 0.f;
  }
  __syncthreads();
  for (int i = ty; i < 32; i += 8) {
    const int c = c0 + i, k = k0 + tx;
    if (c < CoutPad && k < Kpad) wf[(long)c * Kpad + k] = f2bf(t[tx][i]);
  }
}

// wd[ci][tap'*Cout+co] = w[taps-1-tap'][ci][co], zero padded to [CinPad][Kpad2]
__global__ void prep_wd_kernel(const float* __restrict__ w, bf16* __restrict__ wd, int taps, int Cin, int Cout, int CinPad, int Kpad2) {
  const long total = (long)CinPad * Kpad2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i / Kpad2), k = (int)(i - (long)ci * Kpad2);
    float v = 0.f;
    if (ci < Cin && k < taps * Cout) {
      const int tp = k / Cout, co = k - tp * Cout;
      v = w[((long)(taps - 1 - tp) * Cin + ci) * Cout + co];
    }
    wd[i] = f2bf(v);
  }
}

extern "C" int gank_conv2d_prep_weights(const float* w, void* wf, void* wd, int ksize, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(w && (wf || wd), "prep_weights: null pointer");
  GANK_REQUIRE(ksize >= 1 && Cin > 0 && Cout > 0, "prep_weights: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int taps = ksize * ksize;
  if (wf) {
    const int K = taps * Cin, Kpad = roundup(K, 64), CoutPad = roundup(Cout, 32);
    hipLaunchKernelGGL(prep_wf_kernel, dim3(Kpad / 32, CoutPad / 32), dim3(256), 0, s, w, (bf16*)wf, K, Cout, CoutPad, Kpad);
    GANK_LAUNCH_OK("prep_wf");
  }
  if (wd) {
    const int Kpad2 = roundup(taps * Cout, 64), CinPad = roundup(Cin, 32);
    hipLaunchKernelGGL(prep_wd_kernel, grid1d((long)CinPad * Kpad2), dim3(256), 0, s, w, (bf16*)wd, taps, Cin, Cout, CinPad, Kpad2);
    GANK_LAUNCH_OK("prep_wd");
  }
  return 0;
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
__device__ __forceinline__ float sum_taps(const float* __restrict__ p, long plane, bool flip, int h0, int h1, int w0, int w1) {
  const float mh = h1 > h0 ? 1.f : 0.f, mw = w1 > w0 ? 1.f : 0.f;
  const int t00 = flip ? (2 - h0) * 3 + (2 - w0) : h0 * 3 + w0, t01 = flip ? (2 - h0) * 3 + (2 - w1) : h0 * 3 + w1;
  const int t10 = flip ? (2 - h1) * 3 + (2 - w0) : h1 * 3 + w0, t11 = flip ? (2 - h1) * 3 + (2 - w1) : h1 * 3 + w1;
  const float a = p[t00 * plane], b = p[t01 * plane], c = p[t10 * plane], d = p[t11 * plane];
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

__device__ __forceinline__ void prep_up_element(const PrepUpArgs& q, long idx) {
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
      v = sum_taps(q.w + (long)r * q.srP + (long)c * q.skP, q.plane, q.flip != 0, h0, h1, w0, w1);
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
      v = sum_taps(q.w + (long)r * q.srD + (long)c * q.skD, q.plane, q.flip != 0, h0, h1, w0, w1);
    }
    q.d4[i2] = f2bf(v * q.scale);
  }
}
// Eight consecutive elements of the 4x4 matrix d4 per thread where its inner index runs along w's fast axis (skD == 1:
// UpsampleConv) and a tap's row is a whole number of 8-element pieces: 16-byte loads and one 16-byte store instead of
// eight 4-byte loads and 2-byte stores per tap (the generator's 8.4 M-element preparation was 43 of its 54 us).
__device__ __forceinline__ bool prep_up_vec8_ok(const PrepUpArgs& q) { return q.skD == 1 && (q.CkD & 7) == 0 && q.Kpad4 == 16 * q.CkD; }
__device__ __forceinline__ void prep_up_d4_vec8(const PrepUpArgs& q, long i2) {       // i2 % 8 == 0, inside d4
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
      for (int e = 0; e < 4; e++) o[4 * half + e] = f2bf((((a[e] + mw * b[e]) + mh * cc[e]) + (mh * mw) * d[e]) * q.scale);   // sum_taps' order
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
__device__ __forceinline__ void prep_up_tile(const PrepUpArgs& q, bool ph, int tile, float (*tl)[33]) {     // tl: [64][33]
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
    if (r < Cr) v = sum_taps(q.w + r + (long)c * sk, q.plane, q.flip != 0, h0, h1, w0, w1);
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

// standalone form: `ntiles` leading blocks take the transposed tiles of matrix `ph_tiled`, the rest go element-wise over [lo, hi)
__global__ __launch_bounds__(256) void prep_upconv_kernel(PrepUpArgs q, int ntiles, int ph_tiled, long lo, long hi) {
  __shared__ float tt[64][33];
  if ((int)blockIdx.x < ntiles) { prep_up_tile(q, ph_tiled != 0, blockIdx.x, tt); return; }
  const long base = lo + (long)(blockIdx.x - ntiles) * 2048;
  const long nph = 4L * q.CrPpad * 4 * q.CkP;
  if (lo >= nph && prep_up_vec8_ok(q)) {
    const long i = base + 8 * threadIdx.x;
    if (i < hi) prep_up_d4_vec8(q, i - nph);
    return;
  }
  for (int j = 0; j < 8; j++) {
    const long i = base + j * 256 + threadIdx.x;
    if (i < hi) prep_up_element(q, i);
  }
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

extern "C" int gank_upconv3x3_prep_weights(const float* w, void* wph, void* wd4, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(w && wph && wd4 && Cin > 0 && Cout > 0, "upconv3x3_prep_weights: bad arguments");
  const PrepUpArgs q = prep_up_args(1, w, wph, wd4, Cin, Cout);
  const PrepUpSplit sp = prep_up_split(q, 1);
  hipLaunchKernelGGL(prep_upconv_kernel, dim3(sp.ntiles + sp.nelem), dim3(256), 0, (hipStream_t)stream, q, sp.ntiles, 1, sp.lo, sp.hi);
  GANK_LAUNCH_OK("prep_upconv");
  return 0;
}

// ConvMeanPool 3x3 (gan_cifar_resnet.py:112-123): mean_pool2x2(conv3x3_SAME(x)) == 4x4 stride-2 conv (pad 1) with
//   W4[a][b] = 1/4 sum_{i in I(a)} sum_{j in I(b)} W3[i][j],   I(0)={0} I(1)={0,1} I(2)={1,2} I(3)={2}
// wp4 [roundup(Cout,32)][roundup(16*Cin,64)] is its fprop operand; wphd [4][roundup(Cin,32)][4*Cout] holds the 4
// output phases of the transposed conv that is its input gradient (2x2 taps of dy per high-res pixel).
extern "C" int gank_convpool3x3_prep_weights(const float* w, void* wp4, void* wphd, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(w && wp4 && wphd && Cin > 0 && Cout > 0, "convpool3x3_prep_weights: bad arguments");
  const PrepUpArgs q = prep_up_args(2, w, wphd, wp4, Cin, Cout);
  const PrepUpSplit sp = prep_up_split(q, 2);
  hipLaunchKernelGGL(prep_upconv_kernel, dim3(sp.ntiles + sp.nelem), dim3(256), 0, (hipStream_t)stream, q, sp.ntiles, 0, sp.lo, sp.hi);
  GANK_LAUNCH_OK("prep_convpool");
  return 0;
}

// ---- Deconv2D (tf.nn.conv2d_transpose, stride 2, SAME; common/ops/deconv2d.py:99-109) by output phase --------------
// out[n, 2y+a, 2x+b, co] = sum_{i,j in {0,1}} sum_ci x[n, y+i-(1-a), x+j-(1-b), ci] * f[ta][tb][co][ci],
//   ta = 2 - a - 2i + pb, tb = 2 - b - 2j + pb, pb = (k-2)/2 (TensorFlow's SAME pad-before of the stride-2 conv whose
//   gradient this op is); taps outside [0, k) are zero.  Exact for k = 3 (one or two taps per phase and axis) and k = 4
//   (two): the zero-insertion form spends k*k MACs per output on 3/4 zeros, this one 4.  (k = 5 needs three taps per axis
//   in the odd phases and stays on the zero-insertion form.)  wph layout = gank_upconv3x3_prep_weights': [4][CoutPad][4*Cin].
__global__ void deconv_phase_prep_kernel(const float* __restrict__ f, bf16* __restrict__ wph, int k, int Cin, int Cout, int CoutPad) {
  const long total = 4L * CoutPad * 4 * Cin;
  const int pb = (k - 2) / 2;
  for (long idx = blockIdx.x * (long)blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(idx % Cin);
    long t = idx / Cin;
    const int tap = (int)(t & 3); t >>= 2;
    const int co = (int)(t % CoutPad), p = (int)(t / CoutPad);
    const int ta = 2 - (p >> 1) - 2 * (tap >> 1) + pb, tb = 2 - (p & 1) - 2 * (tap & 1) + pb;
    float v = 0.f;
    if (co < Cout && ta >= 0 && ta < k && tb >= 0 && tb < k) v = f[(((long)ta * k + tb) * Cout + co) * Cin + ci];
    wph[idx] = f2bf(v);
  }
}
extern "C" int gank_deconv2d_prep_phases(const float* f, void* wph, int ksize, int Cin, int Cout, void* stream) {
  GANK_REQUIRE(f && wph && Cin > 0 && Cout > 0, "deconv2d_prep_phases: bad arguments");
  GANK_REQUIRE(ksize == 3 || ksize == 4, "deconv2d_prep_phases: 2x2 taps per phase cover filter sizes 3 and 4 (got %d)", ksize);
  const int CoutPad = roundup(Cout, 32);
  const long total = 4L * CoutPad * 4 * Cin;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(deconv_phase_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, f, (bf16*)wph, ksize, Cin, Cout, CoutPad);
  GANK_LAUNCH_OK("deconv2d_prep_phases");
  return 0;
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

// Fragment-major operand copy (kind 3): the 16 bytes lane l of a wave feeds to v_mfma_f32_32x32x16_bf16 as its A
// operand are contiguous, fragments ordered [32-row tile][K-step = 64-channel chunk outer, tap inner][kk][lane], so a
// wave loads one fragment as ONE coalesced 1 KB request straight into registers (conv_igemm_patch2_kernel).
// row: output row of the operand matrix, k = tap*C + c its column (C = channels per tap, C % 64 == 0).
__device__ __forceinline__ long frag_index(int row, int k, int C, int taps, int nsteps) {
  const int tap = k / C, ch = k - tap * C;
  const int chunk = ch >> 6, w64 = ch & 63;
  const int kk = w64 >> 4, hh = (w64 >> 3) & 1, j = w64 & 7;
  const int kstep = chunk * taps + tap;
  return ((((long)(row >> 5) * nsteps + kstep) * 4 + kk) * 64 + hh * 32 + (row & 31)) * 8 + j;
}

__global__ void prep_batch_kernel(PrepTable t) {
  int e = 0;
  for (int i = 1; i < t.count; i++)
    if ((int)blockIdx.x >= t.first_block[i]) e = i;
  const gank_prep_desc& d = t.d[e];
  const int b = blockIdx.x - t.first_block[e];
  const int taps = d.ksize * d.ksize;
  if (d.kind == 1 || d.kind == 2) {       // UpsampleConv / ConvMeanPool 3x3 operands
    const PrepUpArgs& q = t.up[e];
    __shared__ float tt[64][33];
    const long nph = 4L * q.CrPpad * 4 * q.CkP, total = prep_up_total(q);
    if (b < t.nwf[e]) {                   // transposed tiles of the strided-source matrix (nwf = their count, or 0)
      prep_up_tile(q, d.kind == 1, b, tt);
    } else {                              // the other matrix (or both when the tile path does not apply), element-wise
      long lo = 0, hi = total;
      if (t.nwf[e] > 0) { if (d.kind == 1) lo = nph; else hi = nph; }
      const long base = lo + (long)(b - t.nwf[e]) * 2048;
      if (lo >= nph && prep_up_vec8_ok(q)) {          // the block's 2048 elements all lie in d4 (lo = nph there, both multiples of 8)
        const long i = base + 8 * threadIdx.x;
        if (i < hi) prep_up_d4_vec8(q, i - nph);
      } else {
        for (int j = 0; j < 8; j++) {
          const long i = base + j * 256 + threadIdx.x;
          if (i < hi) prep_up_element(q, i);
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
        o[j] = f2bf(isf ? d.w[((long)tap * d.Cin + k) * d.Cout + row] : d.w[((long)(taps - 1 - tap) * d.Cin + row) * d.Cout + k]);
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
          o[j] = f2bf(0.25f * sum_taps(d.w + (long)ci * d.Cout + co, (long)d.Cin * d.Cout, true, h0, h1, w0, w1));
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
          o[j] = f2bf(0.25f * sum_taps(d.w + (long)ci * d.Cout + co, (long)d.Cin * d.Cout, true, h0, h1, w0, w1));
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
      tl[i][tx] = (k < K && c < d.Cout) ? d.w[(long)k * d.Cout + c] : 0.f;
    }
    __syncthreads();
    bf16* wf = (bf16*)d.wf;
    const int c = c0 + (threadIdx.x >> 3), kp = (threadIdx.x & 7) * 8;       // CoutPad % 32 == 0, Kpad % 64 == 0: always inside
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; j++) o[j] = f2bf(tl[kp + j][threadIdx.x >> 3]);
    *reinterpret_cast<bf16x8*>(wf + (long)c * Kpad + k0 + kp) = o;
    // fragment-major copy: 8 consecutive k of one 8-aligned group are consecutive there too (Cin % 64 == 0)
    if (d.kind == 3) *reinterpret_cast<bf16x8*>(wf + (long)CoutPad * Kpad + frag_index(c, k0 + kp, d.Cin, taps, Kpad / 64)) = o;
  } else {
    const int Kpad2 = (taps * d.Cout + 63) / 64 * 64, CinPad = (d.Cin + 31) / 32 * 32;
    const long total = (long)CinPad * Kpad2;
    bf16* wd = (bf16*)d.wd;
    const long base = (long)(b - t.nwf[e]) * 2048;
    if ((d.Cout & 7) == 0 && d.kind != 3) {
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
          for (int u = 0; u < 4; u++) { o[u] = f2bf(a[u]); o[4 + u] = f2bf(bb[u]); }
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
        v = d.w[((long)(taps - 1 - tp) * d.Cin + ci) * d.Cout + co];
      }
      wd[i] = f2bf(v);
      if (d.kind == 3) wd[total + frag_index(ci, k, d.Cout, taps, Kpad2 / 64)] = f2bf(v);
    }
  }
}

extern "C" int gank_conv2d_prep_weights_batched(const gank_prep_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0, "prep_weights_batched: empty table");
  for (int base = 0; base < count; base += PREP_MAX) {
    PrepTable t{};
    t.count = count - base < PREP_MAX ? count - base : PREP_MAX;
    int blocks = 0;
    for (int i = 0; i < t.count; i++) {
      const gank_prep_desc& d = table[base + i];
      GANK_REQUIRE(d.w && (d.wf || d.wd) && d.ksize >= 1 && d.Cin > 0 && d.Cout > 0, "prep_weights_batched: bad entry %d", base + i);
      GANK_REQUIRE(d.kind == 0 || ((d.kind == 1 || d.kind == 2) && d.ksize == 3 && d.wf && d.wd) ||
                   (d.kind == 3 && d.Cin % 64 == 0 && d.Cout % 64 == 0) || (d.kind == 4 && d.Cin % 32 == 0 && d.Cout % 32 == 0) ||
                   (d.kind == 5 && d.ksize == 3 && d.wf && d.wd && d.Cin % 64 == 0 && d.Cout % 32 == 0),
                   "prep_weights_batched: entry %d: kind %d needs ksize 3 and both outputs (1, 2, 5) / channels %% 64 == 0 (3) / %% 32 == 0 (4) / Cin %% 64 == 0 (5)", base + i, d.kind);
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
    hipLaunchKernelGGL(prep_batch_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, t);
    GANK_LAUNCH_OK("prep_weights_batched");
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------
// column sum  out[c] += scale * sum_r x[r][c]
// ------------------------------------------------------------------------------------------------
__global__ void colsum_vec_kernel(const bf16* __restrict__ x, float* __restrict__ out, long rows, int C, float scale, long rows_per_block) {
  // thread -> (8-channel group, row lane); C/8 groups must divide into 256 threads
  const int cg = C >> 3;
  const int RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const long r0 = blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  if (rl < RL) {
    for (long r = r0 + rl; r < r1; r += RL) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v[e]);
    }
  }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int gg = c >> 3, e = c & 7;
    float t = 0.f;
    for (int l = 0; l < RL; l++) t += red[(l * cg + gg) * 8 + e];
    atomicAdd(out + c, t * scale);
  }
}

template <int CMAX>
__global__ void colsum_small_kernel(const bf16* __restrict__ x, float* __restrict__ out, long rows, int C, float scale) {
  float acc[CMAX];
#pragma unroll
  for (int c = 0; c < CMAX; c++) acc[c] = 0.f;
  for (long r = blockIdx.x * (long)blockDim.x + threadIdx.x; r < rows; r += (long)gridDim.x * blockDim.x)
#pragma unroll
    for (int c = 0; c < CMAX; c++)
      if (c < C) acc[c] += bf2f(x[r * C + c]);
  __shared__ float red[16];
#pragma unroll
  for (int c = 0; c < CMAX; c++) {
    if (c < C) {
      const float t = block_sum(acc[c], red);
      if (threadIdx.x == 0) atomicAdd(out + c, t * scale);
    }
  }
}

__global__ void colsum_generic_kernel(const bf16* __restrict__ x, float* __restrict__ out, long rows, int C, float scale) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (long r = 0; r < rows; r++) t += bf2f(x[r * C + c]);
  out[c] += t * scale;
}

extern "C" int gank_colsum_bf16(const void* x, float* out, long rows, int C, float scale, void* stream) {
  GANK_REQUIRE(x && out && rows > 0 && C > 0, "colsum: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (C % 8 == 0 && 256 % (C / 8) == 0) {
    long blocks = (rows + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    const long rpb = (rows + blocks - 1) / blocks;
    blocks = (rows + rpb - 1) / rpb;
    hipLaunchKernelGGL(colsum_vec_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)x, out, rows, C, scale, rpb);
  } else if (C <= 8) {
    // 64 blocks: every block ends in C same-address atomics, which serialise (512 blocks: 15 of the kernel's 22 us)
    hipLaunchKernelGGL(colsum_small_kernel<8>, grid1d(rows, 256, 64), dim3(256), 0, s, (const bf16*)x, out, rows, C, scale);
  } else {
    hipLaunchKernelGGL(colsum_generic_kernel, dim3(cdiv(C, 64)), dim3(64), 0, s, (const bf16*)x, out, rows, C, scale);
  }
  GANK_LAUNCH_OK("colsum");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// 2x2 pooling / unpooling
// ------------------------------------------------------------------------------------------------
template <int V>  // V = 8 (vector) or 1 (scalar)
__global__ void pool2x2_kernel(const bf16* __restrict__ x, const bf16* __restrict__ res, bf16* __restrict__ y,
                               int N, int Ho, int Wo, int C, float scale) {
  const int cg = C / V;
  const long total = (long)N * Ho * Wo * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const long ib = (((long)n * 2 * Ho + 2 * oh) * 2 * Wo + 2 * ow) * C + g * V;
    const long rs = (long)2 * Wo * C;
    const long ob = (((long)n * Ho + oh) * Wo + ow) * C + g * V;
    if constexpr (V == 8) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(x + ib);
      const bf16x8 b = *reinterpret_cast<const bf16x8*>(x + ib + C);
      const bf16x8 c = *reinterpret_cast<const bf16x8*>(x + ib + rs);
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(x + ib + rs + C);
      bf16x8 r8;
      if (res) r8 = *reinterpret_cast<const bf16x8*>(res + ob);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float v = (bf2f(a[e]) + bf2f(c[e]) + bf2f(b[e]) + bf2f(d[e])) * scale;  // order of tf.add_n at :120-121
        if (res) v += bf2f(r8[e]);
        o[e] = f2bf(v);
      }
      *reinterpret_cast<bf16x8*>(y + ob) = o;
    } else {
      float v = (bf2f(x[ib]) + bf2f(x[ib + rs]) + bf2f(x[ib + C]) + bf2f(x[ib + rs + C])) * scale;
      if (res) v += bf2f(res[ob]);
      y[ob] = f2bf(v);
    }
  }
}

extern "C" int gank_pool2x2(const void* x, const void* residual, void* y, int N, int Hout, int Wout, int C, float scale, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && Hout > 0 && Wout > 0 && C > 0, "pool2x2: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (C % 8 == 0)
    hipLaunchKernelGGL(pool2x2_kernel<8>, grid1d((long)N * Hout * Wout * (C / 8)), dim3(256), 0, s, (const bf16*)x, (const bf16*)residual, (bf16*)y, N, Hout, Wout, C, scale);
  else
    hipLaunchKernelGGL(pool2x2_kernel<1>, grid1d((long)N * Hout * Wout * C), dim3(256), 0, s, (const bf16*)x, (const bf16*)residual, (bf16*)y, N, Hout, Wout, C, scale);
  GANK_LAUNCH_OK("pool2x2");
  return 0;
}

template <int V>
__global__ void unpool2x2_add_kernel(const bf16* __restrict__ g, const bf16* __restrict__ base, bf16* __restrict__ y,
                                     int N, int Hi, int Wi, int C, float scale) {
  const int cg = C / V;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const long total = (long)N * Ho * Wo * cg;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int gi = (int)(i % cg);
    long p = i / cg;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    const long ob = i * V;
    const long ib = (((long)n * Hi + (oh >> 1)) * Wi + (ow >> 1)) * C + gi * V;
    if constexpr (V == 8) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(g + ib);
      bf16x8 b;
      if (base) b = *reinterpret_cast<const bf16x8*>(base + ob);
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(a[e]) * scale + (base ? bf2f(b[e]) : 0.f));
      *reinterpret_cast<bf16x8*>(y + ob) = o;
    } else {
      y[ob] = f2bf(bf2f(g[ib]) * scale + (base ? bf2f(base[ob]) : 0.f));
    }
  }
}

extern "C" int gank_unpool2x2_add(const void* g, const void* base, void* y, int N, int Hin, int Win, int C, float scale, void* stream) {
  GANK_REQUIRE(g && y && N > 0 && Hin > 0 && Win > 0 && C > 0, "unpool2x2_add: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (C % 8 == 0)
    hipLaunchKernelGGL(unpool2x2_add_kernel<8>, grid1d((long)N * Hin * Win * 4 * (C / 8)), dim3(256), 0, s, (const bf16*)g, (const bf16*)base, (bf16*)y, N, Hin, Win, C, scale);
  else
    hipLaunchKernelGGL(unpool2x2_add_kernel<1>, grid1d((long)N * Hin * Win * 4 * C), dim3(256), 0, s, (const bf16*)g, (const bf16*)base, (bf16*)y, N, Hin, Win, C, scale);
  GANK_LAUNCH_OK("unpool2x2_add");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// flat elementwise: processed 8 bf16 per lane with a scalar tail
// ------------------------------------------------------------------------------------------------
enum { EW_ADD = 0, EW_RELU_FWD, EW_RELU_BWD, EW_TANH_BWD };

template <int OP>
__device__ __forceinline__ float ew_apply(float a, float b, float p) {
  if constexpr (OP == EW_ADD) return a + b;
  if constexpr (OP == EW_RELU_FWD) return fmaxf(a, p * a);                // tf.maximum(x, leak*x); leak=0 -> relu
  if constexpr (OP == EW_RELU_BWD) return (b > 0.f) ? a : p * a;          // a=dy, b=x
  if constexpr (OP == EW_TANH_BWD) return a * (1.f - b * b);              // a=dy, b=y
  return 0.f;
}

template <int OP, bool BIN>
__global__ void ew_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y, long n, float p) {
  const long nv = n >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nv; i += (long)gridDim.x * blockDim.x) {
    const bf16x8 va = reinterpret_cast<const bf16x8*>(a)[i];
    bf16x8 vb;
    if constexpr (BIN) vb = reinterpret_cast<const bf16x8*>(b)[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(ew_apply<OP>(bf2f(va[e]), BIN ? bf2f(vb[e]) : 0.f, p));
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
  if (blockIdx.x == 0) {
    for (long i = (nv << 3) + threadIdx.x; i < n; i += blockDim.x)
      y[i] = f2bf(ew_apply<OP>(bf2f(a[i]), BIN ? bf2f(b[i]) : 0.f, p));
  }
}

#define EW_ENTRY(NAME, OP, BIN, A, B, P)                                                                   \
  GANK_REQUIRE(A && y && n > 0, NAME ": bad arguments");                                                   \
  hipLaunchKernelGGL((ew_kernel<OP, BIN>), grid1d(n / 8 + 1), dim3(256), 0, (hipStream_t)stream, (const bf16*)A, (const bf16*)B, (bf16*)y, n, P); \
  GANK_LAUNCH_OK(NAME);                                                                                    \
  return 0;

extern "C" int gank_add_bf16(const void* a, const void* b, void* y, long n, void* stream) { EW_ENTRY("add", EW_ADD, true, a, b, 0.f) }
extern "C" int gank_relu_fwd(const void* x, void* y, long n, float leak, void* stream) { EW_ENTRY("relu_fwd", EW_RELU_FWD, false, x, x, leak) }
extern "C" int gank_relu_bwd(const void* dy, const void* x, void* y, long n, float leak, void* stream) { EW_ENTRY("relu_bwd", EW_RELU_BWD, true, dy, x, leak) }
extern "C" int gank_tanh_bwd(const void* dy, const void* yy, void* y, long n, void* stream) { EW_ENTRY("tanh_bwd", EW_TANH_BWD, true, dy, yy, 0.f) }

__global__ void scale_f32_kernel(const float* __restrict__ x, const float* __restrict__ s, float* __restrict__ y, long n) {
  const float k = s[0];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = x[i] * k;
}
extern "C" int gank_scale_f32(const float* x, const float* sc, float* y, long n, void* stream) {
  GANK_REQUIRE(x && sc && y && n > 0, "scale_f32: bad arguments");
  hipLaunchKernelGGL(scale_f32_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, sc, y, n);
  GANK_LAUNCH_OK("scale_f32");
  return 0;
}

// Device-to-device copy as a KERNEL.  Inside a captured hipGraph the memset/memcpy nodes that hipMemsetAsync /
// hipMemcpyAsync turn into were observed to lose their ordering against neighbouring kernel nodes (gank_cbn_bwd
// history), so everything the captured training step moves goes through kernels.
__global__ void copy_bytes_kernel(const unsigned char* __restrict__ src, unsigned char* __restrict__ dst, long n16, long nbytes) {
  const long stride = (long)gridDim.x * blockDim.x, t = blockIdx.x * (long)blockDim.x + threadIdx.x;
  for (long i = t; i < n16; i += stride) reinterpret_cast<u32x4*>(dst)[i] = reinterpret_cast<const u32x4*>(src)[i];
  for (long i = n16 * 16 + t; i < nbytes; i += stride) dst[i] = src[i];
}
extern "C" int gank_copy_bytes(void* dst, const void* src, long nbytes, void* stream) {
  GANK_REQUIRE(dst && src && nbytes > 0, "copy_bytes: bad arguments");
  const bool al = (((uintptr_t)dst | (uintptr_t)src) & 15) == 0;
  const long n16 = al ? nbytes / 16 : 0;
  hipLaunchKernelGGL(copy_bytes_kernel, grid1d(al ? n16 + 15 : nbytes), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned char*)src, (unsigned char*)dst, n16, nbytes);
  GANK_LAUNCH_OK("copy_bytes");
  return 0;
}

// `count` equal-sized device buffers -> consecutive slots of one buffer, ONE launch (grid.y = the source); the pointer list
// travels by value in the kernel arguments, so the call is hipGraph-capturable and needs no device-side table
#define COPY_GATHER_MAX 16
struct CopyGatherArgs { const unsigned char* src[COPY_GATHER_MAX]; };
__global__ void copy_gather_kernel(CopyGatherArgs a, unsigned char* __restrict__ dst, long n16, long nbytes) {
  // static selection of the by-value pointer (a dynamic index would spill the struct to scratch)
  const unsigned char* src = a.src[0];
#pragma unroll
  for (int i = 1; i < COPY_GATHER_MAX; i++) src = (int)blockIdx.y == i ? a.src[i] : src;
  unsigned char* d = dst + (long)blockIdx.y * nbytes;
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (long i = t; i < n16; i += stride) reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(src)[i];
  for (long i = n16 * 16 + t; i < nbytes; i += stride) d[i] = src[i];
}
extern "C" int gank_copy_bytes_gather(void* dst, const void* const* srcs, int count, long nbytes_each, void* stream) {
  GANK_REQUIRE(dst && srcs && count > 0 && count <= COPY_GATHER_MAX && nbytes_each > 0, "copy_bytes_gather: bad arguments (1..%d sources)", COPY_GATHER_MAX);
  CopyGatherArgs a{};
  bool al = (((uintptr_t)dst | (uintptr_t)nbytes_each) & 15) == 0;
  for (int i = 0; i < count; i++) {
    GANK_REQUIRE(srcs[i], "copy_bytes_gather: source %d is null", i);
    a.src[i] = (const unsigned char*)srcs[i];
    al = al && (((uintptr_t)srcs[i]) & 15) == 0;
  }
  const long n16 = al ? nbytes_each / 16 : 0;
  long blocks = ((al ? n16 : nbytes_each) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_gather_kernel, dim3((unsigned)blocks, (unsigned)count), dim3(256), 0, (hipStream_t)stream, a, (unsigned char*)dst, n16, nbytes_each);
  GANK_LAUNCH_OK("copy_bytes_gather");
  return 0;
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ x, bf16* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = f2bf(x[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16* __restrict__ x, float* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) y[i] = bf2f(x[i]);
}
extern "C" int gank_cast_f32_bf16(const float* x, void* y, long n, void* stream) {
  GANK_REQUIRE(x && y && n > 0, "cast_f32_bf16: bad arguments");
  hipLaunchKernelGGL(cast_f32_bf16_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, (bf16*)y, n);
  GANK_LAUNCH_OK("cast_f32_bf16");
  return 0;
}
extern "C" int gank_cast_bf16_f32(const void* x, float* y, long n, void* stream) {
  GANK_REQUIRE(x && y && n > 0, "cast_bf16_f32: bad arguments");
  hipLaunchKernelGGL(cast_bf16_f32_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, y, n);
  GANK_LAUNCH_OK("cast_bf16_f32");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// relu + global mean pool over H,W:  x [N,HW,C] -> y [N,C]          (gan_cifar_resnet.py:299-301)
// one block per sample; threads = (C/8 channel groups) x (row lanes)
// ------------------------------------------------------------------------------------------------
__global__ void relu_meanpool_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int HW, int C) {
  const int n = blockIdx.x;
  const int cg = C >> 3, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < RL)
    for (int r = rl; r < HW; r += RL) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)n * HW + r) * C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += fmaxf(bf2f(v[e]), 0.f);
    }
  __shared__ float red[256 * 8];
#pragma unroll
  for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int l = 0; l < RL; l++) t += red[(l * cg + (c >> 3)) * 8 + (c & 7)];
    y[(long)n * C + c] = f2bf(t / (float)HW);
  }
}

__global__ void relu_meanpool_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ dx, long total8, int HW, int C) {
  const int cg = C >> 3;
  const float inv = 1.f / (float)HW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long n = (i / cg) / HW;
    const bf16x8 xv = reinterpret_cast<const bf16x8*>(x)[i];
    const bf16x8 gv = *reinterpret_cast<const bf16x8*>(dy + n * C + g * 8);
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(bf2f(xv[e]) > 0.f ? bf2f(gv[e]) * inv : 0.f);
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  }
}

extern "C" int gank_relu_meanpool_hw_fwd(const void* x, void* y, int N, int HW, int C, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && HW > 0, "relu_meanpool_fwd: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && 256 % (C / 8) == 0, "relu_meanpool_fwd: C=%d unsupported (need C%%8==0 and (C/8) | 256)", C);
  hipLaunchKernelGGL(relu_meanpool_fwd_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, HW, C);
  GANK_LAUNCH_OK("relu_meanpool_fwd");
  return 0;
}
extern "C" int gank_relu_meanpool_hw_bwd(const void* dy, const void* x, void* dx, int N, int HW, int C, void* stream) {
  GANK_REQUIRE(dy && x && dx && N > 0 && HW > 0 && C % 8 == 0, "relu_meanpool_bwd: bad arguments");
  const long total8 = (long)N * HW * (C / 8);
  hipLaunchKernelGGL(relu_meanpool_bwd_kernel, grid1d(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)x, (bf16*)dx, total8, HW, C);
  GANK_LAUNCH_OK("relu_meanpool_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// concat(a, tile(e)) on channels                                     (gan_cifar_resnet.py:282-284)
// ------------------------------------------------------------------------------------------------
__global__ void concat_tile_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ e, bf16* __restrict__ y, long total8, int HW, int C1, int C2) {
  const int cg = (C1 + C2) >> 3, cg1 = C1 >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long p = i / cg;  // n*HW + hw
    bf16x8 v;
    if (g < cg1) v = *reinterpret_cast<const bf16x8*>(a + p * C1 + g * 8);
    else v = *reinterpret_cast<const bf16x8*>(e + (p / HW) * C2 + (g - cg1) * 8);
    reinterpret_cast<bf16x8*>(y)[i] = v;
  }
}

// da = dy[..., :C1] ; de[n, c] = sum_hw dy[n, hw, C1 + c]   (block of 1024 threads per sample)
__global__ __launch_bounds__(1024) void concat_tile_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ da, bf16* __restrict__ de,
                                                              int HW, int C1, int C2) {
  constexpr int NT = 1024;
  const int n = blockIdx.x, C = C1 + C2;
  const int cg1 = C1 >> 3;
  for (int i = threadIdx.x; i < HW * cg1; i += NT) {
    const int g = i % cg1, r = i / cg1;
    *reinterpret_cast<bf16x8*>(da + ((long)n * HW + r) * C1 + g * 8) = *reinterpret_cast<const bf16x8*>(dy + ((long)n * HW + r) * C + g * 8);
  }
  const int cg2 = C2 >> 3, RL = NT / cg2;
  const int g = threadIdx.x % cg2, rl = threadIdx.x / cg2;
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < RL)
    for (int r = rl; r < HW; r += RL) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(dy + ((long)n * HW + r) * C + C1 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(v[e]);
    }
  __shared__ float red[NT * 8];
#pragma unroll
  for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C2; c += NT) {
    float t = 0.f;
    for (int l = 0; l < RL; l++) t += red[(l * cg2 + (c >> 3)) * 8 + (c & 7)];
    de[(long)n * C2 + c] = f2bf(t);
  }
}

extern "C" int gank_concat_tile_fwd(const void* a, const void* e, void* y, int N, int HW, int C1, int C2, void* stream) {
  GANK_REQUIRE(a && e && y && N > 0 && HW > 0, "concat_tile_fwd: bad arguments");
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0, "concat_tile_fwd: channel counts must be multiples of 8");
  const long total8 = (long)N * HW * ((C1 + C2) / 8);
  hipLaunchKernelGGL(concat_tile_fwd_kernel, grid1d(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)e, (bf16*)y, total8, HW, C1, C2);
  GANK_LAUNCH_OK("concat_tile_fwd");
  return 0;
}
extern "C" int gank_concat_tile_bwd(const void* dy, void* da, void* de, int N, int HW, int C1, int C2, void* stream) {
  GANK_REQUIRE(dy && da && de && N > 0 && HW > 0, "concat_tile_bwd: bad arguments");
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && 1024 % (C2 / 8) == 0, "concat_tile_bwd: unsupported channel counts %d,%d", C1, C2);
  hipLaunchKernelGGL(concat_tile_bwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)da, (bf16*)de, HW, C1, C2);
  GANK_LAUNCH_OK("concat_tile_bwd");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// embedding lookup / dense gradient                                 (common/ops/embedding.py:51)
// ------------------------------------------------------------------------------------------------
__global__ void embedding_fwd_kernel(const float* __restrict__ table, const int* __restrict__ idx, bf16* __restrict__ y, long total, int D, int vocab) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / D), d = (int)(i - (long)n * D);
    const int r = idx[n];
    y[i] = f2bf((r >= 0 && r < vocab) ? table[(long)r * D + d] : 0.f);
  }
}
__global__ void embedding_bwd_kernel(const bf16* __restrict__ dy, const int* __restrict__ idx, float* __restrict__ dtable, long total, int D, int vocab) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int n = (int)(i / D), d = (int)(i - (long)n * D);
    const int r = idx[n];
    if (r >= 0 && r < vocab) atomicAdd(dtable + (long)r * D + d, bf2f(dy[i]));
  }
}
extern "C" int gank_embedding_fwd(const float* table, const int32_t* idx, void* y, int N, int D, int vocab, void* stream) {
  GANK_REQUIRE(table && idx && y && N > 0 && D > 0 && vocab > 0, "embedding_fwd: bad arguments");
  hipLaunchKernelGGL(embedding_fwd_kernel, grid1d((long)N * D), dim3(256), 0, (hipStream_t)stream, table, idx, (bf16*)y, (long)N * D, D, vocab);
  GANK_LAUNCH_OK("embedding_fwd");
  return 0;
}
extern "C" int gank_embedding_bwd(const void* dy, const int32_t* idx, float* dtable, int N, int D, int vocab, void* stream) {
  GANK_REQUIRE(dy && idx && dtable && N > 0 && D > 0 && vocab > 0, "embedding_bwd: bad arguments");
  hipLaunchKernelGGL(embedding_bwd_kernel, grid1d((long)N * D), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, idx, dtable, (long)N * D, D, vocab);
  GANK_LAUNCH_OK("embedding_bwd");
  return 0;
}
