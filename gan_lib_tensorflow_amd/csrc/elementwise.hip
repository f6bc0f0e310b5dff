// HBM-bound glue kernels of the block library: weight layout preparation, resampling, activations,
// pooling, concat/tile, embedding, casts, column sums.  All bf16 traffic is 16 B per lane whenever the
// channel count allows (C % 8 == 0); a scalar path covers the 3-channel image side and odd sizes.
#include "gank_common.h"
#include "label_conv_dev.h"

static inline dim3 grid1d(long n, int block = 256, int cap = 4096) {
  long g = (n + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

#include "prep_weights.h"

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


// ---- batched: every conv/linear weight of a network in ONE launch (table by value in the kernel arguments)
__global__ void prep_batch_kernel(PrepTable t) {
  const int e = prep_batch_entry(t, blockIdx.x);
  prep_batch_block<false>(t, e, blockIdx.x, 1.f);
}

extern "C" int gank_conv2d_prep_weights_batched(const gank_prep_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0, "prep_weights_batched: empty table");
  for (int base = 0; base < count; base += PREP_MAX) {
    PrepTable t;
    const int blocks = prep_table_fill(t, table + base, count - base < PREP_MAX ? count - base : PREP_MAX, base);
    if (blocks < 0) return 1;
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
    // batches of 8 independent 16-byte loads (one load per loop trip was one L2 / HBM round trip per trip: 2.8 TB/s on a 67-MB tensor);
    // the additions stay in ascending row order
    for (long r = r0 + rl; r < r1; r += 8L * RL) {
      bf16x8 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const long rr = r + (long)u * RL;
        v[u] = *reinterpret_cast<const bf16x8*>(x + (rr < r1 ? rr : r) * C + g * 8);
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (r + (long)u * RL < r1) {
#pragma unroll
          for (int e = 0; e < 8; e++) acc[e] += bf2f(v[u][e]);
        }
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
// y = wa a + wb b + wc c + wd d over n floats (null terms skipped): the weighted sums of scalar losses in the train steps
// (ACGAN/train.py:108-121, Pix2Pix/model.py: gen_loss = gan_weight GAN + l1_weight L1) and fp32 accumulations y = y + x
// (no __restrict__: callers accumulate in place, y aliasing one of the terms -- element i is read before it is written)
__global__ void wsum4_f32_kernel(const float* a, const float* b, const float* c, const float* d,
                                 float wa, float wb, float wc, float wd, float* y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = wa * a[i];
    if (b) t += wb * b[i];
    if (c) t += wc * c[i];
    if (d) t += wd * d[i];
    y[i] = t;
  }
}
extern "C" int gank_weighted_sum4_f32(const float* a, const float* b, const float* c, const float* d, float wa, float wb, float wc, float wd,
                                      float* y, long n, void* stream) {
  GANK_REQUIRE(a && y && n > 0, "weighted_sum4_f32: bad arguments");
  hipLaunchKernelGGL(wsum4_f32_kernel, grid1d(n), dim3(256), 0, (hipStream_t)stream, a, b, c, d, wa, wb, wc, wd, y, n);
  GANK_LAUNCH_OK("weighted_sum4_f32");
  return 0;
}
// out[i] += scale * sum_{s < nslabs} slabs[s * stride + i]   for up to 8 jobs per launch, slabs summed in ascending order
// (deterministic).  Split-K filter gradients whose partial tiles used to leave as fp32 atomics write plain slabs instead and
// are summed here: in the critic update the atomics of the 3-channel-input fused gradient (256 workgroups x 4096 floats onto
// the SAME 4096 addresses) and of the batched 8x8 layers (240 x 12288) cost 17 + 12 us of their kernels' 40 + 30 us.
// Few slabs (<= 16): a thread owns 4 consecutive outputs and walks the slabs; many slabs (the 256 of the fused gradient): a
// block owns 64 outputs, its four waves take every fourth slab each and meet in LDS.
constexpr int SLAB_JOBS = 12;
struct SlabJobTable {
  gank_slab_job j[SLAB_JOBS];
  int first_block[SLAB_JOBS + 1];
  int count;
};
__device__ __forceinline__ void sum_slabs_block(const SlabJobTable& t, float (*part)[64]) {
  int ji = 0;
#pragma unroll
  for (int i = 1; i < SLAB_JOBS; i++) ji += (i < t.count && (int)blockIdx.x >= t.first_block[i]) ? 1 : 0;
  const gank_slab_job& jb = t.j[ji];
  const int lb = blockIdx.x - t.first_block[ji], tid = threadIdx.x;
  const float* __restrict__ sl = jb.slabs;
  if (jb.fold) {
    // 4x4 -> 3x3 fold of the ConvMeanPool filter gradient (wgrad_cpool_fold_slabs_kernel's arithmetic and order): one thread per
    // (3x3 tap, float4 of the plane); host checked plane % 4 == 0 and 16-byte alignment
    const long plane4 = jb.n / 36, i = (long)lb * 256 + tid;
    if (i >= 9 * plane4) return;
    const int tap = (int)(i / plane4), ti = tap / 3, tj = tap - 3 * ti;
    const long e = i - (long)tap * plane4;
    const long stride4 = jb.stride >> 2;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int sp = 0; sp < jb.nslabs; sp += 2) {          // 8 independent loads per batch (2 splits x 4 source taps), fixed order
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int s2 = min(sp + (u >> 2), jb.nslabs - 1);
        const int src = (ti + ((u >> 1) & 1)) * 4 + tj + (u & 1);
        v[u] = reinterpret_cast<const f32x4*>(sl)[(long)s2 * stride4 + (long)src * plane4 + e];
      }
#pragma unroll
      for (int u = 0; u < 8; u++)
        if (sp + (u >> 2) < jb.nslabs) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
    f32x4 o = reinterpret_cast<f32x4*>(jb.out)[(long)tap * plane4 + e];
#pragma unroll
    for (int q = 0; q < 4; q++) o[q] += jb.scale * acc[q];
    reinterpret_cast<f32x4*>(jb.out)[(long)tap * plane4 + e] = o;
    return;
  }
  if (jb.nslabs <= 32) {
    const long i0 = ((long)lb * 256 + tid) * 4;
    if (i0 >= jb.n) return;
    // (out_run: runs of out_run outputs, out_pitch apart -- both multiples of 4, so a thread's four outputs share a run)
    float* op = jb.out + (jb.out_run > 0 ? (i0 / jb.out_run) * jb.out_pitch + i0 % jb.out_run : i0);
    if (i0 + 4 <= jb.n && (jb.stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(sl) | reinterpret_cast<uintptr_t>(jb.out)) & 15) == 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      for (int b0 = 0; b0 < jb.nslabs; b0 += 16) {          // batches of 16 independent loads, slabs in ascending order
        f32x4 v[16];
#pragma unroll
        for (int u = 0; u < 16; u++) v[u] = *reinterpret_cast<const f32x4*>(sl + (long)min(b0 + u, jb.nslabs - 1) * jb.stride + i0);
        if (b0 == 0) acc = v[0];
#pragma unroll
        for (int u = 0; u < 16; u++)
          if ((b0 > 0 || u > 0) && b0 + u < jb.nslabs) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
      }
      f32x4 o = *reinterpret_cast<const f32x4*>(op);
#pragma unroll
      for (int e = 0; e < 4; e++) o[e] += jb.scale * acc[e];
      *reinterpret_cast<f32x4*>(op) = o;
    } else {
      for (long i = i0; i < min(i0 + 4, jb.n); i++) {
        float acc = 0.f;
        for (int u = 0; u < jb.nslabs; u++) acc += sl[(long)u * jb.stride + i];
        op[i - i0] += jb.scale * acc;
      }
    }
    return;
  }
  const int o = tid & 63, sg = tid >> 6;
  const long i = (long)lb * 64 + o;
  float acc = 0.f;
  if (i < jb.n) {
    for (int sb = sg; sb < jb.nslabs; sb += 4 * 16) {          // 16 independent loads per batch, this wave's slabs in ascending order
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; u++) v[u] = sl[(long)min(sb + 4 * u, jb.nslabs - 1) * jb.stride + i];
#pragma unroll
      for (int u = 0; u < 16; u++) acc += (sb + 4 * u < jb.nslabs) ? v[u] : 0.f;
    }
  }
  part[sg][o] = acc;
  __syncthreads();
  if (sg == 0 && i < jb.n) jb.out[i] += jb.scale * (((part[0][o] + part[1][o]) + part[2][o]) + part[3][o]);
}
__global__ __launch_bounds__(256) void sum_slabs_kernel(SlabJobTable t) {
  __shared__ float part[4][64];
  sum_slabs_block(t, part);
}
// ... with the label-gradient launch of a critic update (gank_label_conv3x3_bwd_pooled: independent of every slab, small) as extra
// workgroups behind the summing ones: one launch less per update
__global__ __launch_bounds__(256) void sum_slabs_label_kernel(SlabJobTable t, LabelBwdArgs q, int main_blocks) {
  __shared__ float part[4][64];
  extern __shared__ __attribute__((aligned(16))) float sm_[];
  if ((int)blockIdx.x >= main_blocks) {
    label_conv_bwd_block(q, blockIdx.x - main_blocks, sm_);
    return;
  }
  sum_slabs_block(t, part);
}
static int sum_slabs_impl(const gank_slab_job* jobs, int count, const LabelBwdArgs* lq, size_t lds, void* stream);
extern "C" int gank_sum_slabs(const gank_slab_job* jobs, int count, void* stream) { return sum_slabs_impl(jobs, count, nullptr, 0, stream); }
// gank_sum_slabs(jobs, count) + gank_label_conv3x3_bwd_pooled(...) in ONE launch (the label gradients are extra workgroups; dw_feat_tmp is
// not offered here: the feature rows arrive through a slab job).  count <= 12.
extern "C" int gank_sum_slabs_label_bwd(const gank_slab_job* jobs, int count, const float* tap_sums, const int32_t* lists, const void* T, int V, const float* w,
                                        int Cin_total, int c0, int C2, int Cout, int N, float* dw, float* de_parts, const void* g_pooled, int HWp, int pitch,
                                        int c0g, void* stream) {
  GANK_REQUIRE(jobs && count > 0 && count <= SLAB_JOBS, "sum_slabs_label_bwd: 1..%d jobs", SLAB_JOBS);
  GANK_REQUIRE(tap_sums && lists && T && w && dw && de_parts && g_pooled && N > 0 && N <= 1024 && V > 0 && V <= LCB_V && HWp > 0,
               "sum_slabs_label_bwd: bad arguments");
  GANK_REQUIRE(Cout % 4 == 0 && Cout <= 256 && C2 % 32 == 0 && c0 >= 0 && c0 + C2 <= Cin_total && pitch % 8 == 0 && c0g % 8 == 0 && c0g + C2 <= pitch,
               "sum_slabs_label_bwd: unsupported channel counts");
  size_t lds2 = ((size_t)V * (Cout + 4) + (size_t)LCB_CT * (Cout + 4) + (size_t)V * LCB_CT) * sizeof(float);
  if (lds2 < 64 * 4 * 8 * sizeof(float)) lds2 = 64 * 4 * 8 * sizeof(float);
  GANK_REQUIRE(lds2 <= 48 * 1024, "sum_slabs_label_bwd: Cout = %d does not fit the LDS", Cout);
  const int pool_blocks = V * (C2 / 32);
  const LabelBwdArgs q{tap_sums, (const bf16*)T, w, dw, de_parts, nullptr, V, Cin_total, c0, C2, Cout, 0, 9 * (C2 / LCB_CT) + pool_blocks,
                       (const bf16*)g_pooled, lists, N, HWp, pitch, c0g, pool_blocks};
  return sum_slabs_impl(jobs, count, &q, lds2, stream);
}
static int sum_slabs_impl(const gank_slab_job* jobs, int count, const LabelBwdArgs* lq, size_t lds, void* stream) {
  GANK_REQUIRE(jobs && count > 0, "sum_slabs: empty list");
  for (int base = 0; base < count; base += SLAB_JOBS) {
    SlabJobTable t{};
    t.count = count - base < SLAB_JOBS ? count - base : SLAB_JOBS;
    int blocks = 0;
    for (int i = 0; i < t.count; i++) {
      const gank_slab_job& j = jobs[base + i];
      GANK_REQUIRE(j.slabs && j.out && j.n > 0 && j.nslabs > 0 && (j.fold ? j.stride >= j.n / 9 * 16 : j.stride >= j.n), "sum_slabs: bad job %d", base + i);
      GANK_REQUIRE(!j.fold || (j.n % 36 == 0 && (j.stride & 3) == 0 && ((reinterpret_cast<uintptr_t>(j.slabs) | reinterpret_cast<uintptr_t>(j.out)) & 15) == 0),
                   "sum_slabs: fold job %d needs n %% 36 == 0 and 16-byte aligned buffers", base + i);
      t.j[i] = j;
      t.first_block[i] = blocks;
      GANK_REQUIRE(j.out_run == 0 || (!j.fold && j.nslabs <= 32 && j.out_run % 4 == 0 && j.out_pitch % 4 == 0 && j.out_pitch >= j.out_run && j.n % j.out_run == 0),
                   "sum_slabs: job %d: strided outputs need a plain sum of <= 32 slabs and runs / pitches that are multiples of 4", base + i);
      blocks += j.fold ? (int)((j.n / 4 + 255) / 256) : (j.nslabs <= 32 ? (int)((j.n + 1023) / 1024) : (int)((j.n + 63) / 64));
    }
    t.first_block[t.count] = blocks;
    if (lq) hipLaunchKernelGGL(sum_slabs_label_kernel, dim3((unsigned)(blocks + lq->blocks)), dim3(256), lds, (hipStream_t)stream, t, *lq, blocks);
    else hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t);
  }
  GANK_LAUNCH_OK("sum_slabs");
  return 0;
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

// two gathers in one launch (the iteration's image batches and their label vectors into the two feed rings): grid.y = count_a + count_b,
// the second group's rows behind the first's
struct CopyGather2Args { const unsigned char* src[2 * COPY_GATHER_MAX]; };
__global__ void copy_gather2_kernel(CopyGather2Args a, unsigned char* __restrict__ dst_a, long n16_a, long nbytes_a, int count_a,
                                    unsigned char* __restrict__ dst_b, long n16_b, long nbytes_b) {
  const unsigned char* src = a.src[0];
#pragma unroll
  for (int i = 1; i < 2 * COPY_GATHER_MAX; i++) src = (int)blockIdx.y == i ? a.src[i] : src;
  const bool second = (int)blockIdx.y >= count_a;
  const int row = second ? blockIdx.y - count_a : blockIdx.y;
  const long nbytes = second ? nbytes_b : nbytes_a, n16 = second ? n16_b : n16_a;
  unsigned char* d = (second ? dst_b : dst_a) + (long)row * nbytes;
  const long t = blockIdx.x * (long)blockDim.x + threadIdx.x, stride = (long)gridDim.x * blockDim.x;
  for (long i = t; i < n16; i += stride) reinterpret_cast<u32x4*>(d)[i] = reinterpret_cast<const u32x4*>(src)[i];
  for (long i = n16 * 16 + t; i < nbytes; i += stride) d[i] = src[i];
}
extern "C" int gank_copy_bytes_gather2(void* dst_a, const void* const* srcs_a, int count_a, long nbytes_a, void* dst_b, const void* const* srcs_b, int count_b,
                                       long nbytes_b, void* stream) {
  GANK_REQUIRE(dst_a && srcs_a && dst_b && srcs_b && count_a > 0 && count_b > 0 && count_a <= COPY_GATHER_MAX && count_b <= COPY_GATHER_MAX && nbytes_a > 0 &&
               nbytes_b > 0, "copy_bytes_gather2: bad arguments (1..%d sources per group)", COPY_GATHER_MAX);
  CopyGather2Args a{};
  bool al_a = (((uintptr_t)dst_a | (uintptr_t)nbytes_a) & 15) == 0, al_b = (((uintptr_t)dst_b | (uintptr_t)nbytes_b) & 15) == 0;
  for (int i = 0; i < count_a; i++) {
    GANK_REQUIRE(srcs_a[i], "copy_bytes_gather2: source %d of the first group is null", i);
    a.src[i] = (const unsigned char*)srcs_a[i];
    al_a = al_a && (((uintptr_t)srcs_a[i]) & 15) == 0;
  }
  for (int i = 0; i < count_b; i++) {
    GANK_REQUIRE(srcs_b[i], "copy_bytes_gather2: source %d of the second group is null", i);
    a.src[count_a + i] = (const unsigned char*)srcs_b[i];
    al_b = al_b && (((uintptr_t)srcs_b[i]) & 15) == 0;
  }
  const long n16_a = al_a ? nbytes_a / 16 : 0, n16_b = al_b ? nbytes_b / 16 : 0;
  const long work_a = al_a ? n16_a : nbytes_a, work_b = al_b ? n16_b : nbytes_b;
  long blocks = ((work_a > work_b ? work_a : work_b) + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  hipLaunchKernelGGL(copy_gather2_kernel, dim3((unsigned)blocks, (unsigned)(count_a + count_b)), dim3(256), 0, (hipStream_t)stream, a, (unsigned char*)dst_a, n16_a,
                     nbytes_a, count_a, (unsigned char*)dst_b, n16_b, nbytes_b);
  GANK_LAUNCH_OK("copy_bytes_gather2");
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
// The critic's label branch through a per-label table (gank.h: gank_concat_label_*, gank_label_dense_bwd):
// embed_y -> Linear -> tile -> concat (gan_cifar_resnet.py:276-284) depends on the sample only through its label, so the
// dense layer runs on the V = 10 table rows (sn.hip: second launch of the batched spectral norm) and the concat gathers.
// ------------------------------------------------------------------------------------------------
__global__ void concat_label_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ T, const int* __restrict__ labels,
                                        bf16* __restrict__ y, long total8, int HW, int C1, int C2, int V) {
  const int cg = (C1 + C2) >> 3, cg1 = C1 >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long p = i / cg;  // n*HW + hw
    bf16x8 v;
    if (g < cg1) {
      v = *reinterpret_cast<const bf16x8*>(a + p * C1 + g * 8);
    } else {
      const int l = labels[p / HW];
      const bool ok = l >= 0 && l < V;
      v = *reinterpret_cast<const bf16x8*>(T + (long)(ok ? l : 0) * C2 + (g - cg1) * 8);
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; e++) v[e] = f2bf(0.f);
      }
    }
    reinterpret_cast<bf16x8*>(y)[i] = v;
  }
}

// da = dy[..., :C1] ; de32[n, c] = sum_hw dy[n, hw, C1 + c]   (block of 1024 threads per sample; fp32 sums)
__global__ __launch_bounds__(1024) void concat_label_bwd_kernel(const bf16* __restrict__ dy, bf16* __restrict__ da, float* __restrict__ de,
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
    de[(long)n * C2 + c] = t;
  }
}

// concat + the fan-out of the down-sampling block that follows (its main path reads y, its pooled shortcut mean_pool2x2(y)):
// one pass writes both.  The tiled half is constant over the pixels, so its 2x2 mean is the table row itself (exact); the
// other half is pool2x2_kernel's arithmetic.  Thread = (pooled pixel, 8 channels).
__global__ void concat_label_pool_fwd_kernel(const bf16* __restrict__ a, const bf16* __restrict__ T, const int* __restrict__ labels,
                                             bf16* __restrict__ y, bf16* __restrict__ yp, long total8, int Hp, int Wp, int C1, int C2, int V) {
  const int C = C1 + C2, cg = C >> 3, cg1 = C1 >> 3;
  const int W = 2 * Wp;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int pw = (int)(p % Wp); p /= Wp;
    const int ph = (int)(p % Hp);
    const int n = (int)(p / Hp);
    const long hi = ((long)n * 2 * Hp + 2 * ph) * W + 2 * pw;          // top-left high-resolution pixel
    bf16x8 v[4], m;
    if (g < cg1) {
      v[0] = *reinterpret_cast<const bf16x8*>(a + hi * C1 + g * 8);
      v[1] = *reinterpret_cast<const bf16x8*>(a + (hi + 1) * C1 + g * 8);
      v[2] = *reinterpret_cast<const bf16x8*>(a + (hi + W) * C1 + g * 8);
      v[3] = *reinterpret_cast<const bf16x8*>(a + (hi + W + 1) * C1 + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) m[e] = f2bf((bf2f(v[0][e]) + bf2f(v[2][e]) + bf2f(v[1][e]) + bf2f(v[3][e])) * 0.25f);   // order of tf.add_n at :120-121
    } else {
      const int l = labels[n];
      const bool ok = l >= 0 && l < V;
      m = *reinterpret_cast<const bf16x8*>(T + (long)(ok ? l : 0) * C2 + (g - cg1) * 8);
      if (!ok) {
#pragma unroll
        for (int e = 0; e < 8; e++) m[e] = f2bf(0.f);
      }
      v[0] = v[1] = v[2] = v[3] = m;
    }
    if (y) {           // (null: the consumer of the full-resolution tensor reads `a` and the table itself -- label_conv.hip)
      *reinterpret_cast<bf16x8*>(y + hi * C + g * 8) = v[0];
      *reinterpret_cast<bf16x8*>(y + (hi + 1) * C + g * 8) = v[1];
      *reinterpret_cast<bf16x8*>(y + (hi + W) * C + g * 8) = v[2];
      *reinterpret_cast<bf16x8*>(y + (hi + W + 1) * C + g * 8) = v[3];
    }
    *reinterpret_cast<bf16x8*>(yp + (((long)n * Hp + ph) * Wp + pw) * C + g * 8) = m;
  }
}

// backward of the pair: dy = g_main + 0.25 * unpool(g_pool) (unpool2x2_add's arithmetic) is never written -- its first C1
// channels go straight to da, the tiled half is summed over the sample's pixels into de32 (block of 1024 threads per sample)
// gm_c1: g_main holds the first C1 channels only (the consumer of the tiled half was factored out: label_conv.hip) and that
// consumer's gradient of the tiled vector arrives SUMMED PER LABEL as de_parts partial sums [de_parts][V][C2]: added, in ascending
// order, to the row of the label's first sample (lists: row v = {count, samples of label v ascending}; de's consumers add the
// samples of a label anyway).
__global__ __launch_bounds__(1024) void concat_label_unpool_bwd_kernel(const bf16* __restrict__ gm, const bf16* __restrict__ gp, bf16* __restrict__ da,
                                                                      float* __restrict__ de, int H, int W, int C1, int C2,
                                                                      int gm_c1, const float* __restrict__ de_add, int de_parts, int N,
                                                                      const int* __restrict__ labels, const int* __restrict__ lists, int V) {
  constexpr int NT = 1024;
  const int n = blockIdx.x, C = C1 + C2, HW = H * W, Wp = W >> 1;
  const int cg = C >> 3, cg1 = C1 >> 3, cg2 = C2 >> 3;
  __shared__ float red[NT * 8];
  if (!da) {
    // only the tiled vector's gradient (the first C1 channels' join happens in gank_img16_conv3x3_dgrad_unpool): each of the four
    // full-resolution pixels under a pooled pixel contributes bf16(0.25 g) = 0.25 g exactly, so the sum over the sample is the sum of
    // the pooled gradient's tiled half -- thread = (8-channel group, pooled pixel lane), ONE batch of loads, lanes meet in LDS
    const int PL = NT / cg2, g2 = threadIdx.x % cg2, pl = threadIdx.x / cg2;
    const int HWp = (H >> 1) * Wp;
    float a8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int p = pl; p < HWp; p += PL) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(gp + ((long)n * HWp + p) * C + C1 + g2 * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) a8[e] += bf2f(v[e]);
    }
#pragma unroll
    for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = a8[e];
    __syncthreads();
    for (int c = threadIdx.x; c < C2; c += NT) {
      float t = 0.f;
      for (int l = 0; l < PL; l++) t += red[(l * cg2 + (c >> 3)) * 8 + (c & 7)];
      if (de_parts > 0) {
        int lb = labels[n];
        lb = lb < 0 ? 0 : (lb >= V ? V - 1 : lb);
        if (lists[(long)lb * (N + 1) + 1] == n)
          for (int p = 0; p < de_parts; p++) t += de_add[((long)p * V + lb) * C2 + c];
      }
      de[(long)n * C2 + c] = t;
    }
    return;
  }
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // thread -> a fixed 8-channel group (NT % cg == 0), pixels strided: the tiled half's partial sums stay in registers
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg, RL = NT / cg;
  for (int r = rl; r < HW; r += RL) {
    const int oh = r / W, ow = r - oh * W;
    const long pi = ((long)n * (H >> 1) + (oh >> 1)) * Wp + (ow >> 1);
    const bf16x8 a = *reinterpret_cast<const bf16x8*>(gp + pi * C + g * 8);
    bf16x8 b;
    const bool has_main = gm != nullptr && (!gm_c1 || g < cg1);
    if (has_main) b = *reinterpret_cast<const bf16x8*>(gm + ((long)n * HW + r) * (gm_c1 ? C1 : C) + g * 8);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; e++) v[e] = bf2f(a[e]) * 0.25f + (has_main ? bf2f(b[e]) : 0.f);
    if (g < cg1) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) o[e] = f2bf(v[e]);
      *reinterpret_cast<bf16x8*>(da + ((long)n * HW + r) * C1 + g * 8) = o;
    } else {
#pragma unroll
      for (int e = 0; e < 8; e++) acc[e] += bf2f(f2bf(v[e]));       // the rounded value the separate launches summed
    }
  }
#pragma unroll
  for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = acc[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C2; c += NT) {
    const int gg = cg1 + (c >> 3);
    float t = 0.f;
    for (int l = 0; l < RL; l++) t += red[(l * cg + gg) * 8 + (c & 7)];
    if (de_parts > 0) {
      int lb = labels[n];
      lb = lb < 0 ? 0 : (lb >= V ? V - 1 : lb);
      if (lists[(long)lb * (N + 1) + 1] == n)
        for (int p = 0; p < de_parts; p++) t += de_add[((long)p * V + lb) * C2 + c];
    }
    de[(long)n * C2 + c] = t;
  }
  (void)cg2;
}

// dT[l] = sum_{n: labels[n] = l} de32[n], then this block's 8 rows k of dW [D,C2] += bf16(emb)^T dT and of
// demb [V,D] += dT W^T; block 0 also owns dbias.  Every block builds dT itself, 1024 threads = (column j, sample group g):
// a thread requests its group's samples of column j in one burst, adds each into its group's table row of that sample's
// label (an LDS read-modify-write chain of N / groups steps), and the group tables are summed in group order -- no atomics,
// the same bits every run.  (Measured forms this replaced: a thread per (label, column) pair scanning all N samples with a
// compare per pair, 57 k cycles of one-wave-per-SIMD latency; a wave walking 20 dependent L2 round trips for the table rows.)
// Dynamic LDS: dTg [1024*V] | dT [V*C2] | labels [N] | Ws [8*C2] | Es [8*V] | dt_out [8*V].
__global__ __launch_bounds__(1024) void label_dense_bwd_kernel(const float* __restrict__ de, const int* __restrict__ labels,
                                                               const float* __restrict__ table, const float* __restrict__ W,
                                                               float* __restrict__ dW, float* __restrict__ dbias, float* __restrict__ dtable,
                                                               int N, int V, int D, int C2, const float* __restrict__ de_add, int de_parts) {
  // de_add [de_parts][V][C2]: gradient rows that arrive summed per label already (label_conv.hip), added in part order behind the
  // per-sample rows; de may be null then (no per-sample rows at all: N = 0)
  extern __shared__ __attribute__((aligned(16))) float lds_f[];
  float* dTg = lds_f;
  float* dT = dTg + 1024 * V;
  int* lbs = reinterpret_cast<int*>(dT + ((V * C2 + 3) & ~3));
  float* Ws = reinterpret_cast<float*>(lbs + ((N + 3) & ~3));
  float* Es = Ws + 8 * C2;
  float* dto = Es + 8 * V;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int NG = 1024 / C2, j = tid % C2, g = tid / C2;
  const int k0 = blockIdx.x * 8;
  // the values this block adds to (first element of each thread's strided loops) are requested with the block's first loads: the
  // read-modify-write round trips at the end were a third of the kernel
  float dw_old = 0.f, dt_old = 0.f, db_old = 0.f;
  if (dW && tid < 8 * C2 && k0 + tid / C2 < D) dw_old = dW[(long)(k0 + tid / C2) * C2 + tid % C2];
  if (dtable && tid < 8 * V && k0 + (tid & 7) < D) dt_old = dtable[(long)(tid >> 3) * D + k0 + (tid & 7)];
  if (dbias && blockIdx.x == 0 && tid < C2) db_old = dbias[tid];
  if (N == 0) {                                  // rows summed per label already: dT needs nothing but them
    for (int idx = tid; idx < V * C2; idx += 1024) {
      float t = 0.f;
      for (int p = 0; p < de_parts; p++) t += de_add[(long)p * V * C2 + idx];
      dT[idx] = t;
    }
  }
  if (N > 0)
    for (int l = 0; l < V; l++) dTg[(g * V + l) * C2 + j] = 0.f;
  for (int i = tid; i < N; i += 1024) lbs[i] = labels[i];
  if (W)
    for (int i = tid; i < 8 * C2; i += 1024) { const int k = k0 + i / C2; Ws[i] = k < D ? W[(long)k * C2 + i % C2] : 0.f; }
  if (table)
    for (int i = tid; i < 8 * V; i += 1024) { const int k = k0 + (i & 7), l = i >> 3; Es[i] = k < D ? bf2f(f2bf(table[(long)l * D + k])) : 0.f; }
  __syncthreads();
  for (int nb = g; nb < N; nb += 16 * NG) {
    float x[16];
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int n = nb + u * NG;
      x[u] = n < N ? de[(long)n * C2 + j] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 16; u++) {
      const int n = nb + u * NG;
      const int l = n < N ? lbs[n] : -1;
      if (l >= 0 && l < V) dTg[(g * V + l) * C2 + j] += x[u];
    }
  }
  __syncthreads();
  if (N > 0) {
    for (int idx = tid; idx < V * C2; idx += 1024) {
      const int l = idx / C2, jj = idx % C2;
      float t = 0.f;
      for (int gg = 0; gg < NG; gg++) t += dTg[(gg * V + l) * C2 + jj];
      for (int p = 0; p < de_parts; p++) t += de_add[((long)p * V + l) * C2 + jj];
      dT[idx] = t;
    }
    __syncthreads();
  }
  if (dW)
    for (int i = tid; i < 8 * C2; i += 1024) {
      const int kk = i / C2, jj = i % C2;
      if (k0 + kk < D) {
        float t = 0.f;
        for (int l = 0; l < V; l++) t += Es[l * 8 + kk] * dT[l * C2 + jj];
        dW[(long)(k0 + kk) * C2 + jj] = (i == tid ? dw_old : dW[(long)(k0 + kk) * C2 + jj]) + t;
      }
    }
  if (dtable) {
    for (int p = wave; p < 8 * V; p += 16) {        // pair p = (label l, row kk): a wave per pair, lanes over the columns
      const int kk = p & 7, l = p >> 3;
      float t = 0.f;
      for (int jj = lane; jj < C2; jj += 64) t += dT[l * C2 + jj] * Ws[kk * C2 + jj];
      t = wave_sum(t);
      if (lane == 0) dto[p] = t;
    }
    __syncthreads();
    for (int p = tid; p < 8 * V; p += 1024) {
      const int kk = p & 7, l = p >> 3;
      if (k0 + kk < D) dtable[(long)l * D + k0 + kk] = (p == tid ? dt_old : dtable[(long)l * D + k0 + kk]) + dto[p];
    }
  }
  if (dbias && blockIdx.x == 0)
    for (int jj = tid; jj < C2; jj += 1024) {
      float t = 0.f;
      for (int l = 0; l < V; l++) t += dT[l * C2 + jj];
      dbias[jj] = (jj == tid ? db_old : dbias[jj]) + t;
    }
}

extern "C" int gank_concat_label_fwd(const void* a, const void* T, const int32_t* labels, void* y, int N, int HW, int C1, int C2, int V,
                                     void* stream) {
  GANK_REQUIRE(a && T && labels && y && N > 0 && HW > 0 && V > 0, "concat_label_fwd: bad arguments");
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0, "concat_label_fwd: channel counts must be multiples of 8");
  const long total8 = (long)N * HW * ((C1 + C2) / 8);
  hipLaunchKernelGGL(concat_label_fwd_kernel, grid1d(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)T, labels, (bf16*)y,
                     total8, HW, C1, C2, V);
  GANK_LAUNCH_OK("concat_label_fwd");
  return 0;
}
extern "C" int gank_concat_label_bwd(const void* dy, void* da, float* de32, int N, int HW, int C1, int C2, void* stream) {
  GANK_REQUIRE(dy && da && de32 && N > 0 && HW > 0, "concat_label_bwd: bad arguments");
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && 1024 % (C2 / 8) == 0, "concat_label_bwd: unsupported channel counts %d,%d", C1, C2);
  hipLaunchKernelGGL(concat_label_bwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)dy, (bf16*)da, de32, HW, C1, C2);
  GANK_LAUNCH_OK("concat_label_bwd");
  return 0;
}
extern "C" int gank_concat_label_pool_fwd(const void* a, const void* T, const int32_t* labels, void* y, void* y_pooled, int N, int H, int W,
                                          int C1, int C2, int V, void* stream) {
  GANK_REQUIRE(a && T && labels && y_pooled && N > 0 && H > 0 && W > 0 && V > 0, "concat_label_pool_fwd: bad arguments");      // y may be NULL: pooled output only
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && H % 2 == 0 && W % 2 == 0, "concat_label_pool_fwd: channel counts must be multiples of 8, the size even");
  const long total8 = (long)N * (H / 2) * (W / 2) * ((C1 + C2) / 8);
  hipLaunchKernelGGL(concat_label_pool_fwd_kernel, grid1d(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)T, labels,
                     (bf16*)y, (bf16*)y_pooled, total8, H / 2, W / 2, C1, C2, V);
  GANK_LAUNCH_OK("concat_label_pool_fwd");
  return 0;
}
extern "C" int gank_concat_label_unpool_bwd(const void* g_main, const void* g_pooled, void* da, float* de32, int N, int H, int W, int C1, int C2,
                                            void* stream) {
  GANK_REQUIRE(g_pooled && da && de32 && N > 0 && H > 0 && W > 0, "concat_label_unpool_bwd: bad arguments");
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && H % 2 == 0 && W % 2 == 0 && 1024 % ((C1 + C2) / 8) == 0,
               "concat_label_unpool_bwd: unsupported shape (%d + %d channels, %d x %d)", C1, C2, H, W);
  hipLaunchKernelGGL(concat_label_unpool_bwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)g_main, (const bf16*)g_pooled,
                     (bf16*)da, de32, H, W, C1, C2, 0, (const float*)nullptr, 0, N, (const int*)nullptr, (const int*)nullptr, 0);
  GANK_LAUNCH_OK("concat_label_unpool_bwd");
  return 0;
}
// the same where the consumer of the tiled half was factored out (gank_label_conv3x3_*): g_main_c1 [N,H,W,C1] is the gradient of the
// first C1 channels only, de_add [de_parts][V][C2] that consumer's gradient of the tiled vector per LABEL (partial sums, added in
// order to the label's first sample: labels [N], lists from gank_label_conv3x3_table)
extern "C" int gank_concat_label_unpool_bwd_factored(const void* g_main_c1, const void* g_pooled, void* da, float* de32, const float* de_add, int de_parts,
                                                     const int32_t* labels, const int32_t* lists, int V, int N, int H, int W, int C1, int C2, void* stream) {
  GANK_REQUIRE(g_pooled && de32 && N > 0 && H > 0 && W > 0 && (de_parts == 0 || (de_add && labels && lists && V > 0)) && (da == nullptr || g_main_c1),
               "concat_label_unpool_bwd_factored: bad arguments");      // da NULL (then g_main_c1 is ignored): only the tiled vector's gradient
  GANK_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && H % 2 == 0 && W % 2 == 0 && 1024 % ((C1 + C2) / 8) == 0 && 1024 % (C2 / 8) == 0,
               "concat_label_unpool_bwd_factored: unsupported shape (%d + %d channels, %d x %d)", C1, C2, H, W);
  hipLaunchKernelGGL(concat_label_unpool_bwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)g_main_c1, (const bf16*)g_pooled,
                     (bf16*)da, de32, H, W, C1, C2, 1, de_add, de_parts, N, labels, lists, V);
  GANK_LAUNCH_OK("concat_label_unpool_bwd_factored");
  return 0;
}
extern "C" int gank_label_dense_bwd(const float* de32, const int32_t* labels, const float* table, const float* W, float* dW, float* dbias,
                                    float* dtable, int N, int V, int D, int C2, void* stream) {
  GANK_REQUIRE(de32 && labels && N > 0 && V > 0 && D > 0 && C2 > 0, "label_dense_bwd: bad arguments");
  GANK_REQUIRE((!dW || table) && (!dtable || W), "label_dense_bwd: dW needs the table, dtable needs W");
  GANK_REQUIRE(C2 <= 1024 && 1024 % C2 == 0, "label_dense_bwd: C2 = %d must divide 1024", C2);
  const size_t lds = (1024 * (size_t)V + (((size_t)V * C2 + 3) & ~(size_t)3) + (((size_t)N + 3) & ~(size_t)3) + 8 * (size_t)C2 + 16 * (size_t)V) * 4;
  GANK_REQUIRE(lds <= 160 * 1024, "label_dense_bwd: V = %d labels, N = %d samples do not fit the LDS", V, N);
  GANK_MAX_DYNAMIC_LDS(label_dense_bwd_kernel, 160 * 1024, "label_dense_bwd");
  hipLaunchKernelGGL(label_dense_bwd_kernel, dim3(cdiv(D, 8)), dim3(1024), lds, (hipStream_t)stream, de32, labels, table, W, dW, dbias, dtable,
                     N, V, D, C2, (const float*)nullptr, 0);
  GANK_LAUNCH_OK("label_dense_bwd");
  return 0;
}
// the same from gradient rows that are summed per label already: dT[l] = sum_p de_parts_rows[p][l] (fp32 [parts][V][C2], added in part
// order: gank_label_conv3x3_bwd_pooled's ten parts) -- no per-sample rows, no labels
extern "C" int gank_label_dense_bwd_parts(const float* de_parts_rows, int parts, const float* table, const float* W, float* dW, float* dbias,
                                          float* dtable, int V, int D, int C2, void* stream) {
  GANK_REQUIRE(de_parts_rows && parts > 0 && V > 0 && D > 0 && C2 > 0, "label_dense_bwd_parts: bad arguments");
  GANK_REQUIRE((!dW || table) && (!dtable || W), "label_dense_bwd_parts: dW needs the table, dtable needs W");
  GANK_REQUIRE(C2 <= 1024 && 1024 % C2 == 0, "label_dense_bwd_parts: C2 = %d must divide 1024", C2);
  const size_t lds = (1024 * (size_t)V + (((size_t)V * C2 + 3) & ~(size_t)3) + 8 * (size_t)C2 + 16 * (size_t)V) * 4;
  GANK_REQUIRE(lds <= 160 * 1024, "label_dense_bwd_parts: V = %d labels do not fit the LDS", V);
  GANK_MAX_DYNAMIC_LDS(label_dense_bwd_kernel, 160 * 1024, "label_dense_bwd_parts");
  hipLaunchKernelGGL(label_dense_bwd_kernel, dim3(cdiv(D, 8)), dim3(1024), lds, (hipStream_t)stream, (const float*)nullptr, (const int*)nullptr, table, W,
                     dW, dbias, dtable, 0, V, D, C2, de_parts_rows, parts);
  GANK_LAUNCH_OK("label_dense_bwd_parts");
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
