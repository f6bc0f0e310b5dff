// Conditional batch norm (common/ops/normalization.py:27-59), forward and backward, HBM-bound.
//   moments over (N/groups, H, W) per tower, biased (tf.nn.moments): ONE pass of shifted sums per part
//   + a deterministic Chan merge of the parts, eps = 1e-5; y = (x-mean)*invstd*gamma[label] + beta[label]; optional fused relu.
// x is [N, HW, C] bf16 with C % 8 == 0: every lane moves 16 B; thread = (8-channel group, row lane).
#include <atomic>
#include "gank_common.h"

#define BN_EPS 1e-5f

extern "C" int gank_cbn_parts(long rows_per_group) {
  long p = (rows_per_group + 63) / 64;     // small tensors still get enough blocks to hide latency
  if (p < 1) p = 1;
  if (p > 256) p = 256;
  return (int)p;
}

struct CbnGeom {
  int N, HW, C, groups, parts, n_labels, relu;
  long rows_per_group, rows_per_part;
  float eps;
};

constexpr int CBN_NT = 1024;   // 16 waves per block: these kernels are latency-bound streams otherwise

// pass 1 (ONE read of x): per part, shifted sums around the part's first row -> (mean_p, M2_p).
// Shifting by a sample of the data keeps S2 - S1^2/n free of catastrophic cancellation in fp32.
__global__ __launch_bounds__(CBN_NT) void cbn_stats_kernel(const bf16* __restrict__ x, float* __restrict__ ws, CbnGeom q) {
  const int cg = q.C >> 3, RL = CBN_NT / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int grp = blockIdx.x / q.parts, part = blockIdx.x % q.parts;
  const long r0 = grp * q.rows_per_group + part * q.rows_per_part;
  long r1 = r0 + q.rows_per_part;
  const long rend = (grp + 1) * q.rows_per_group;
  if (r1 > rend) r1 = rend;
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0}, k[8];
  {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r0 * q.C + g * 8);
#pragma unroll
    for (int e = 0; e < 8; e++) k[e] = bf2f(v[e]);
  }
  if (rl < RL)
    for (long r = r0 + rl; r < r1; r += RL) {
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + r * q.C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) { const float d = bf2f(v[e]) - k[e]; s1[e] += d; s2[e] += d * d; }
    }
  __shared__ float red[CBN_NT * 16];
#pragma unroll
  for (int e = 0; e < 8; e++) { red[threadIdx.x * 16 + e] = s1[e]; red[threadIdx.x * 16 + 8 + e] = s2[e]; }
  __syncthreads();
  const float n = (float)(r1 - r0);
  for (int c = threadIdx.x; c < q.C; c += CBN_NT) {
    float t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < RL; l++) {
      t1 += red[(l * cg + (c >> 3)) * 16 + (c & 7)];
      t2 += red[(l * cg + (c >> 3)) * 16 + 8 + (c & 7)];
    }
    const float kk = bf2f(x[r0 * q.C + c]);
    float* o = ws + (((long)grp * q.parts + part) * 2) * q.C;
    o[c] = kk + t1 / n;                 // mean of the part
    o[q.C + c] = t2 - t1 * t1 / n;      // sum of squared deviations from that mean
  }
}

// pass 2: merge the parts (Chan et al. pairwise update; fixed merge order: deterministic) -> mean, invstd.
// Block = 16 channels x 16 part-slices: each thread merges its strided slice, then slice 0 merges the 16
// partial results through LDS (sequential depth parts/16 + 16 instead of parts).
__device__ __forceinline__ void chan_merge(float& na, float& mean, float& m2, float nb, float mb, float m2b) {
  if (nb <= 0.f) return;
  const float d = mb - mean, nt = na + nb;
  mean += d * nb / nt;
  m2 += m2b + d * d * na * nb / nt;
  na = nt;
}

__global__ __launch_bounds__(256) void cbn_finalize_kernel(const float* __restrict__ ws, float* __restrict__ stats, CbnGeom q) {
  const int cl = threadIdx.x & 15, ps = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const int grp = blockIdx.y;
  __shared__ float sh[3][16][17];
  float na = 0.f, mean = 0.f, m2 = 0.f;
  if (c < q.C) {
    for (int pb = ps; pb < q.parts; pb += 64) {          // 4 parts per batch: their loads are issued together
      float pm[4], pv[4];
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int p = min(pb + 16 * u, q.parts - 1);
        const float* o = ws + (((long)grp * q.parts + p) * 2) * q.C;
        pm[u] = o[c];
        pv[u] = o[q.C + c];
      }
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int p = pb + 16 * u;
        if (p < q.parts) {
          long r0 = (long)p * q.rows_per_part, r1 = r0 + q.rows_per_part;
          if (r1 > q.rows_per_group) r1 = q.rows_per_group;
          chan_merge(na, mean, m2, (float)(r1 - r0), pm[u], pv[u]);
        }
      }
    }
  }
  sh[0][ps][cl] = na; sh[1][ps][cl] = mean; sh[2][ps][cl] = m2;
  __syncthreads();
  if (ps == 0 && c < q.C) {
    for (int j = 1; j < 16; j++) chan_merge(na, mean, m2, sh[0][j][cl], sh[1][j][cl], sh[2][j][cl]);
    stats[((long)grp * 2 + 0) * q.C + c] = mean;
    stats[((long)grp * 2 + 1) * q.C + c] = 1.f / sqrtf(m2 / na + q.eps);   // biased variance (tf.nn.moments)
  }
}

// The apply loop of both forward kernels.  Rows of one sample share the label's gamma / beta rows: they are loaded once per
// sample (not per row, behind a dependent label load), and four rows are in flight per thread -- the loop was 5 loads, a
// 64-bit division and one 16-byte row per trip.  The expression and its order are the forward pass's own (the backward pass
// recomputes the relu mask from it, bit for bit).
__device__ __forceinline__ void cbn_apply_rows(const bf16* __restrict__ x, const int* __restrict__ labels, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, bf16* __restrict__ y, const float (&mu)[8], const float (&iv)[8],
                                               const CbnGeom& q, long r0, long r1, int rl, int RL, int g) {
  if (r1 <= r0) return;
  const int n0 = (int)(r0 / q.HW), n1 = (int)((r1 - 1) / q.HW);
  for (int n = n0; n <= n1; n++) {
    const long lo = r0 > (long)n * q.HW ? r0 : (long)n * q.HW;
    const long hi = r1 < (long)(n + 1) * q.HW ? r1 : (long)(n + 1) * q.HW;
    int lb = labels[n];
    lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
    const f32x4 g0 = *reinterpret_cast<const f32x4*>(gamma + (long)lb * q.C + g * 8);
    const f32x4 g1 = *reinterpret_cast<const f32x4*>(gamma + (long)lb * q.C + g * 8 + 4);
    const f32x4 b0 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + g * 8);
    const f32x4 b1 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + g * 8 + 4);
    auto one = [&](const bf16x8& v, long r) {
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
        float t = (bf2f(v[e]) - mu[e]) * iv[e] * ga + be;
        if (q.relu) t = fmaxf(t, 0.f);
        o[e] = f2bf(t);
      }
      *reinterpret_cast<bf16x8*>(y + r * q.C + g * 8) = o;
    };
    long r = lo + rl;
    for (; r + 3L * RL < hi; r += 4L * RL) {
      bf16x8 v[4];
#pragma unroll
      for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const bf16x8*>(x + (r + (long)u * RL) * q.C + g * 8);
#pragma unroll
      for (int u = 0; u < 4; u++) one(v[u], r + (long)u * RL);
    }
    for (; r < hi; r += RL) one(*reinterpret_cast<const bf16x8*>(x + r * q.C + g * 8), r);
  }
}

// pass 3: normalise + gamma/beta gather (+relu)
__global__ __launch_bounds__(CBN_NT) void cbn_apply_kernel(const bf16* __restrict__ x, const int* __restrict__ labels,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         bf16* __restrict__ y, const float* __restrict__ stats, CbnGeom q) {
  const int cg = q.C >> 3, RL = CBN_NT / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int grp = blockIdx.x / q.parts, part = blockIdx.x % q.parts;
  const long r0 = grp * q.rows_per_group + part * q.rows_per_part;
  long r1 = r0 + q.rows_per_part;
  const long rend = (grp + 1) * q.rows_per_group;
  if (r1 > rend) r1 = rend;
  if (rl >= RL) return;
  float mu[8], iv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    mu[e] = stats[((long)grp * 2) * q.C + g * 8 + e];
    iv[e] = stats[((long)grp * 2 + 1) * q.C + g * 8 + e];
  }
  cbn_apply_rows(x, labels, gamma, beta, y, mu, iv, q, r0, r1, rl, RL, g);
}

// apply pass fed by statistics the PRODUCING conv accumulated in its epilogue (gank_conv2d_fprop_stats): every block
// turns the shifted sums of its 8 channels into (mean, invstd) itself -- no statistics pass over x, no finalize launch.
// sums [groups][GANK_STAT_SLOTS][2][C] = partial sums / sums of squares of (x - shift[c]) per tower; shift (the conv bias) may be null.
__global__ __launch_bounds__(CBN_NT) void cbn_apply_sums_kernel(const bf16* __restrict__ x, const int* __restrict__ labels,
                                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              bf16* __restrict__ y, float* __restrict__ stats, const float* __restrict__ sums,
                                                              const float* __restrict__ shift, CbnGeom q) {
  const int cg = q.C >> 3, RL = CBN_NT / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int grp = blockIdx.x / q.parts, part = blockIdx.x % q.parts;
  const long r0 = grp * q.rows_per_group + part * q.rows_per_part;
  long r1 = r0 + q.rows_per_part;
  const long rend = (grp + 1) * q.rows_per_group;
  if (r1 > rend) r1 = rend;
  // one thread per channel adds the partial copies and leaves (mean, invstd) in LDS for the block
  __shared__ float s_mu[2048], s_iv[2048];
  const float M = (float)q.rows_per_group;
  for (int c = threadIdx.x; c < q.C; c += CBN_NT) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int sl = 0; sl < GANK_STAT_SLOTS; sl++) {
      t1 += sums[(((long)grp * GANK_STAT_SLOTS + sl) * 2) * q.C + c];
      t2 += sums[(((long)grp * GANK_STAT_SLOTS + sl) * 2 + 1) * q.C + c];
    }
    const float m1 = t1 / M, m2 = t2 / M;
    const float mean = m1 + (shift ? shift[c] : 0.f), is = 1.f / sqrtf(fmaxf(m2 - m1 * m1, 0.f) + q.eps);   // biased variance (tf.nn.moments)
    s_mu[c] = mean;
    s_iv[c] = is;
    if (part == 0) {                                                 // what the backward pass reads
      stats[((long)grp * 2) * q.C + c] = mean;
      stats[((long)grp * 2 + 1) * q.C + c] = is;
    }
  }
  __syncthreads();
  if (rl >= RL) return;
  float mu[8], iv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) { mu[e] = s_mu[g * 8 + e]; iv[e] = s_iv[g * 8 + e]; }
  cbn_apply_rows(x, labels, gamma, beta, y, mu, iv, q, r0, r1, rl, RL, g);
}

static int cbn_geom(CbnGeom& q, int N, int HW, int C, int groups, int n_labels, int relu) {
  GANK_REQUIRE(N > 0 && HW > 0 && C > 0 && groups > 0 && n_labels > 0, "cbn: bad shape");
  GANK_REQUIRE(N % groups == 0, "cbn: batch %d not divisible by %d towers", N, groups);
  GANK_REQUIRE(C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, "cbn: C=%d unsupported (need C%%8==0, (C/8) | 256)", C);
  q.N = N; q.HW = HW; q.C = C; q.groups = groups; q.n_labels = n_labels; q.relu = relu;
  q.rows_per_group = (long)(N / groups) * HW;
  q.parts = gank_cbn_parts(q.rows_per_group);
  q.rows_per_part = (q.rows_per_group + q.parts - 1) / q.parts;
  q.parts = (int)((q.rows_per_group + q.rows_per_part - 1) / q.rows_per_part);   // no empty parts
  return 0;
}

// Small towers (<= 1024 rows: the generator's first conditional batch norm, on the 4x4 map behind G.Input, whose statistics no conv
// epilogue can deliver) in ONE launch instead of three at the launch floor: block = (tower, 64 channels = 128 bytes of every row),
// thread = (8-channel group, row lane); the tower's rows of the block stay in REGISTERS (<= 16 sixteen-byte pieces per thread, all
// requested before the first use) for the mean, the sum of squared deviations from that exact mean (two passes over registers: no
// shifted sums, no part merge) and the apply -- the forward expression of cbn_apply_rows, operand for operand.  The 64 row lanes meet
// in LDS in a fixed order (deterministic).
constexpr int CFS_ROWS = 16, CFS_LANES = 64;        // rows per thread, row lanes (512 threads: 256 VGPRs for the 16 resident pieces)
__global__ __launch_bounds__(512) void cbn_fwd_small_kernel(const bf16* __restrict__ x, const int* __restrict__ labels, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16* __restrict__ y, float* __restrict__ stats, CbnGeom q) {
  __shared__ float red[512 * 8];
  __shared__ float mean_s[64], iv_s[64];
  const int tid = threadIdx.x, g = tid & 7, rl = tid >> 3;
  const int cgroups = q.C >> 6;
  const int grp = blockIdx.x / cgroups, cb = blockIdx.x - grp * cgroups;
  const long r0 = (long)grp * q.rows_per_group;
  const int rows = (int)q.rows_per_group;
  const int c0 = cb * 64 + g * 8;
  float v[CFS_ROWS][8];                 // as floats: the three passes below would otherwise each convert (or the compiler keep both forms)
  {
    bf16x8 raw[CFS_ROWS];
#pragma unroll
    for (int u = 0; u < CFS_ROWS; u++) {
      const int r = rl * CFS_ROWS + u;
      raw[u] = *reinterpret_cast<const bf16x8*>(x + (r0 + (r < rows ? r : 0)) * q.C + c0);       // unconditional, clamped
    }
#pragma unroll
    for (int u = 0; u < CFS_ROWS; u++)
#pragma unroll
      for (int e = 0; e < 8; e++) v[u][e] = bf2f(raw[u][e]);
  }
  float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < CFS_ROWS; u++)
    if (rl * CFS_ROWS + u < rows) {
#pragma unroll
      for (int e = 0; e < 8; e++) s1[e] += v[u][e];
    }
#pragma unroll
  for (int e = 0; e < 8; e++) red[tid * 8 + e] = s1[e];
  __syncthreads();
  if (tid < 64) {
    float p[CFS_LANES];                 // all loads first (a load per add in a chain was 2.7 us per reduction), the adds in lane order
#pragma unroll
    for (int l = 0; l < CFS_LANES; l++) p[l] = red[l * 64 + tid];
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < CFS_LANES; l++) t += p[l];
    mean_s[tid] = t / (float)rows;
  }
  __syncthreads();
  float mu[8], s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
  for (int e = 0; e < 8; e++) mu[e] = mean_s[g * 8 + e];
#pragma unroll
  for (int u = 0; u < CFS_ROWS; u++)
    if (rl * CFS_ROWS + u < rows) {
#pragma unroll
      for (int e = 0; e < 8; e++) { const float d = v[u][e] - mu[e]; s2[e] += d * d; }
    }
#pragma unroll
  for (int e = 0; e < 8; e++) red[tid * 8 + e] = s2[e];
  __syncthreads();
  if (tid < 64) {
    float p[CFS_LANES];
#pragma unroll
    for (int l = 0; l < CFS_LANES; l++) p[l] = red[l * 64 + tid];
    float t = 0.f;
#pragma unroll
    for (int l = 0; l < CFS_LANES; l++) t += p[l];
    const float iv = 1.f / sqrtf(t / (float)rows + q.eps);       // biased variance (tf.nn.moments)
    iv_s[tid] = iv;
    stats[((long)grp * 2 + 0) * q.C + cb * 64 + tid] = mean_s[tid];
    stats[((long)grp * 2 + 1) * q.C + cb * 64 + tid] = iv;
  }
  if (!y) return;                       // statistics only (gank_cbn_stats: the same numbers as the forward pass's, bit for bit)
  __syncthreads();
  float iv[8];
#pragma unroll
  for (int e = 0; e < 8; e++) iv[e] = iv_s[g * 8 + e];
  // a thread's rows are consecutive: the label's gamma / beta rows are reloaded only when the sample changes (once per thread at HW >= 16)
  int n_cur = -1;
  f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, b0 = g0, b1 = g0;
#pragma unroll
  for (int u = 0; u < CFS_ROWS; u++) {
    const int r = rl * CFS_ROWS + u;
    if (r < rows) {
      const int n = (int)((r0 + r) / q.HW);
      if (n != n_cur) {
        n_cur = n;
        int lb = labels[n];
        lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
        g0 = *reinterpret_cast<const f32x4*>(gamma + (long)lb * q.C + c0); g1 = *reinterpret_cast<const f32x4*>(gamma + (long)lb * q.C + c0 + 4);
        b0 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + c0); b1 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + c0 + 4);
      }
      bf16x8 o;
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float ga = e < 4 ? g0[e] : g1[e - 4], be = e < 4 ? b0[e] : b1[e - 4];
        float t = (v[u][e] - mu[e]) * iv[e] * ga + be;
        if (q.relu) t = fmaxf(t, 0.f);
        o[e] = f2bf(t);
      }
      *reinterpret_cast<bf16x8*>(y + (r0 + r) * q.C + c0) = o;
    }
  }
}
static bool cbn_fwd_small_ok(const CbnGeom& q) {
  static const int env = gank_tune("GANK_CBN_SMALL", 1);   // experiment knob: 0 keeps the three-launch form for small towers
  return env && q.rows_per_group <= CFS_LANES * CFS_ROWS && q.C % 64 == 0;
}

extern "C" int gank_cbn_fwd_eps(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y,
                            float* stats, float* ws, int N, int HW, int C, int groups, int n_labels, int relu, float eps, void* stream) {
  GANK_REQUIRE(x && labels && gamma && beta && y && stats && ws, "cbn_fwd: null pointer");
  CbnGeom q;
  if (cbn_geom(q, N, HW, C, groups, n_labels, relu)) return 1;
  GANK_REQUIRE(eps > 0.f, "cbn_fwd: eps must be positive");
  q.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (cbn_fwd_small_ok(q)) {
    hipLaunchKernelGGL(cbn_fwd_small_kernel, dim3(groups * (C / 64)), dim3(512), 0, s, (const bf16*)x, labels, gamma, beta, (bf16*)y, stats, q);
    GANK_LAUNCH_OK("cbn_fwd");
    return 0;
  }
  const dim3 grid(groups * q.parts);
  hipLaunchKernelGGL(cbn_stats_kernel, grid, dim3(CBN_NT), 0, s, (const bf16*)x, ws, q);
  hipLaunchKernelGGL(cbn_finalize_kernel, dim3(cdiv(C, 16), groups), dim3(256), 0, s, ws, stats, q);
  hipLaunchKernelGGL(cbn_apply_kernel, grid, dim3(CBN_NT), 0, s, (const bf16*)x, labels, gamma, beta, (bf16*)y, stats, q);
  GANK_LAUNCH_OK("cbn_fwd");
  return 0;
}

extern "C" int gank_cbn_fwd_from_sums(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y, float* stats,
                                      const float* sums, const float* shift, int N, int HW, int C, int groups, int n_labels, int relu,
                                      float eps, void* stream) {
  GANK_REQUIRE(x && labels && gamma && beta && y && stats && sums, "cbn_fwd_from_sums: null pointer");
  CbnGeom q;
  if (cbn_geom(q, N, HW, C, groups, n_labels, relu)) return 1;
  GANK_REQUIRE(eps > 0.f, "cbn_fwd_from_sums: eps must be positive");
  q.eps = eps;
  hipLaunchKernelGGL(cbn_apply_sums_kernel, dim3(groups * q.parts), dim3(CBN_NT), 0, (hipStream_t)stream, (const bf16*)x, labels, gamma, beta,
                     (bf16*)y, stats, sums, shift, q);
  GANK_LAUNCH_OK("cbn_fwd_from_sums");
  return 0;
}

extern "C" int gank_cbn_fwd(const void* x, const int32_t* labels, const float* gamma, const float* beta, void* y,
                            float* stats, float* ws, int N, int HW, int C, int groups, int n_labels, int relu, void* stream) {
  return gank_cbn_fwd_eps(x, labels, gamma, beta, y, stats, ws, N, HW, C, groups, n_labels, relu, BN_EPS, stream);
}

// ---- backward -------------------------------------------------------------------------------------
// b1: per-sample sums S1[n][c] = sum_hw dym, S2[n][c] = sum_hw dym * xhat   (dym = dy masked by y>0)
// The relu mask is y > 0.  With `beta` given it is RECOMPUTED from x -- (x - mean) * invstd * gamma + beta, the forward pass's
// own expression in the forward pass's order -- instead of read: one of the three tensor reads of this pass (and one of the
// four accesses of the apply pass) less.
__global__ void cbn_bwd_sums_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const bf16* __restrict__ y,
                                    const float* __restrict__ stats, float* __restrict__ S, CbnGeom q, int hw_parts,
                                    const int* __restrict__ labels, const float* __restrict__ gamma, const float* __restrict__ beta,
                                    float* __restrict__ Sp, unsigned* __restrict__ tickets) {
  const int cg = q.C >> 3, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const int n = blockIdx.x / hw_parts, hp = blockIdx.x % hw_parts;
  const int grp = n / (q.N / q.groups);
  const int per = (q.HW + hw_parts - 1) / hw_parts;
  const int h0 = hp * per, h1 = min(q.HW, h0 + per);
  float a1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, a2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (rl < RL) {
    float mu[8], iv[8];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      mu[e] = stats[((long)grp * 2) * q.C + g * 8 + e];
      iv[e] = stats[((long)grp * 2 + 1) * q.C + g * 8 + e];
    }
    float ga[8], be[8];
    if (q.relu && beta) {
      int lb = labels[n];
      lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
#pragma unroll
      for (int e = 0; e < 8; e++) { ga[e] = gamma[(long)lb * q.C + g * 8 + e]; be[e] = beta[(long)lb * q.C + g * 8 + e]; }
    }
    for (int r = h0 + rl; r < h1; r += RL) {
      const long o = ((long)n * q.HW + r) * q.C + g * 8;
      const bf16x8 d = *reinterpret_cast<const bf16x8*>(dy + o);
      const bf16x8 xv = *reinterpret_cast<const bf16x8*>(x + o);
      bf16x8 yv;
      if (q.relu && !beta) yv = *reinterpret_cast<const bf16x8*>(y + o);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        float dd = bf2f(d[e]);
        if (q.relu) {
          const bool on = beta ? ((bf2f(xv[e]) - mu[e]) * iv[e] * ga[e] + be[e] > 0.f) : (bf2f(yv[e]) > 0.f);
          if (!on) dd = 0.f;
        }
        a1[e] += dd;
        a2[e] += dd * (bf2f(xv[e]) - mu[e]) * iv[e];
      }
    }
  }
  __shared__ float red[256 * 16];
#pragma unroll
  for (int e = 0; e < 8; e++) { red[threadIdx.x * 16 + e] = a1[e]; red[threadIdx.x * 16 + 8 + e] = a2[e]; }
  __syncthreads();
  for (int c = threadIdx.x; c < q.C; c += 256) {
    float t1 = 0.f, t2 = 0.f;
    for (int l = 0; l < RL; l++) {
      t1 += red[(l * cg + (c >> 3)) * 16 + (c & 7)];
      t2 += red[(l * cg + (c >> 3)) * 16 + 8 + (c & 7)];
    }
    if (hw_parts == 1) {
      S[((long)n * 2) * q.C + c] = t1;
      S[((long)n * 2 + 1) * q.C + c] = t2;
    } else if (Sp) {            // rows of this part's own, written through to memory (visible to every XCD once vmcnt has drained)
      __hip_atomic_store(Sp + (((long)hp * q.N + n) * 2) * q.C + c, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(Sp + (((long)hp * q.N + n) * 2 + 1) * q.C + c, t2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      atomicAdd(S + ((long)n * 2) * q.C + c, t1);
      atomicAdd(S + ((long)n * 2 + 1) * q.C + c, t2);
    }
  }
  if (hw_parts > 1 && Sp) {
    // the last part of sample n to arrive adds the parts' rows in ascending order (the same bits every run): no fill launch, no
    // atomics on the sums (idiom of sn.hip: write-through stores, drained; one ticket per block; one acquire by the last arriver)
    __shared__ int last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned tk = __hip_atomic_fetch_add(tickets + n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int last = tk == (unsigned)(hw_parts - 1);
      if (last) {
        __hip_atomic_store(tickets + n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // zero again for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    for (int i = threadIdx.x; i < 2 * q.C; i += 256) {
      float v[8];
#pragma unroll
      for (int p = 0; p < 8; p++) v[p] = p < hw_parts ? Sp[((long)p * q.N + n) * 2 * q.C + i] : 0.f;
      float t = v[0];
#pragma unroll
      for (int p = 1; p < 8; p++) t += p < hw_parts ? v[p] : 0.f;
      S[(long)n * 2 * q.C + i] = t;
    }
  }
}

__global__ void cbn_zero_kernel(float* __restrict__ p, int n4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n4) reinterpret_cast<f32x4*>(p)[i] = f32x4{0.f, 0.f, 0.f, 0.f};
}

// b2: table gradients and per-tower means.  grid = (C/64, n_labels + 1): block row l < n_labels owns label l
// (sequential over n: deterministic, no atomics, registers only); the last block row computes the means.
__global__ __launch_bounds__(256) void cbn_bwd_tables_kernel(const float* __restrict__ S, const int* __restrict__ labels,
                                                            const float* __restrict__ gamma, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, float* __restrict__ M, CbnGeom q) {
  // 64 channels x 4 waves: wave w takes the samples n = w (mod 4) -- the sample loop was the whole run time of the
  // 64-thread version (N dependent load pairs in a row); the 4 partial sums meet in LDS, in a fixed order
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int cc = c < q.C ? c : q.C - 1;
  const int l = blockIdx.y;
  __shared__ float red[4][2][64];
  if (l < q.n_labels) {
    float a1 = 0.f, a2 = 0.f;
#pragma unroll 8
    for (int n = wv; n < q.N; n += 4) {      // unconditional loads (pipelined), predicated adds
      int lb = labels[n];
      lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
      const float s1 = S[((long)n * 2) * q.C + cc], s2 = S[((long)n * 2 + 1) * q.C + cc];
      a1 += (lb == l) ? s1 : 0.f;
      a2 += (lb == l) ? s2 : 0.f;
    }
    red[wv][0][lane] = a1;
    red[wv][1][lane] = a2;
    __syncthreads();
    if (wv == 0 && c < q.C) {
      dbeta[(long)l * q.C + c] += red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
      dgamma[(long)l * q.C + c] += red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
    }
  } else {
    const int gs = q.N / q.groups;
    for (int grp = 0; grp < q.groups; grp++) {
      float m1 = 0.f, m2 = 0.f;
#pragma unroll 4
      for (int i = wv; i < gs; i += 4) {
        const int n = grp * gs + i;
        int lb = labels[n];
        lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
        const float ga = gamma[(long)lb * q.C + cc];
        m1 += ga * S[((long)n * 2) * q.C + cc];
        m2 += ga * S[((long)n * 2 + 1) * q.C + cc];
      }
      __syncthreads();                       // the previous group's sums have been read
      red[wv][0][lane] = m1;
      red[wv][1][lane] = m2;
      __syncthreads();
      if (wv == 0 && c < q.C) {
        M[((long)grp * 2) * q.C + c] = (red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane]) / (float)q.rows_per_group;
        M[((long)grp * 2 + 1) * q.C + c] = (red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane]) / (float)q.rows_per_group;
      }
    }
  }
}

// b3: dx = invstd * (g - mean(g) - xhat * mean(g xhat)),  g = dym * gamma[label]
__global__ void cbn_bwd_apply_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const bf16* __restrict__ y,
                                     const int* __restrict__ labels, const float* __restrict__ gamma, const float* __restrict__ stats,
                                     const float* __restrict__ M, bf16* __restrict__ dx, CbnGeom q, long total8, const float* __restrict__ beta) {
  // A thread keeps ONE channel group g (blockDim is a multiple of C/8) and walks rows; the 40 parameter values
  // of a row depend on its sample only and are reloaded when the sample changes, not per 16 bytes of data behind a label load
  // (the loop was 5 tensor-independent 16-byte loads, two 64-bit divisions and a modulo per trip); two rows are in flight.
  const int cg = q.C >> 3;
  const int gs = q.N / q.groups;
  // block b owns the rows [b * per, (b + 1) * per): consecutive rows of a thread are blockDim / cg apart, in the same sample
  const long rows_all = total8 / cg;
  const long per = (rows_all + gridDim.x - 1) / gridDim.x;
  const int g = threadIdx.x % cg;
  const long rstride = blockDim.x / cg;
  const long rbeg = blockIdx.x * per + threadIdx.x / cg;
  const long rows = (blockIdx.x + 1) * per < rows_all ? (blockIdx.x + 1) * per : rows_all;
  int n_cur = -1;
  f32x4 mu0, mu1, iv0, iv1, ma0, ma1, mb0, mb1, ga0, ga1, be0 = {0.f, 0.f, 0.f, 0.f}, be1 = be0;
  auto params = [&](int n) {
    const int grp = n / gs;
    int lb = labels[n];
    lb = lb < 0 ? 0 : (lb >= q.n_labels ? q.n_labels - 1 : lb);
    const float* sp = stats + ((long)grp * 2) * q.C + g * 8;
    const float* mp = M + ((long)grp * 2) * q.C + g * 8;
    const float* gp = gamma + (long)lb * q.C + g * 8;
    mu0 = *reinterpret_cast<const f32x4*>(sp); mu1 = *reinterpret_cast<const f32x4*>(sp + 4);
    iv0 = *reinterpret_cast<const f32x4*>(sp + q.C); iv1 = *reinterpret_cast<const f32x4*>(sp + q.C + 4);
    ma0 = *reinterpret_cast<const f32x4*>(mp); ma1 = *reinterpret_cast<const f32x4*>(mp + 4);
    mb0 = *reinterpret_cast<const f32x4*>(mp + q.C); mb1 = *reinterpret_cast<const f32x4*>(mp + q.C + 4);
    ga0 = *reinterpret_cast<const f32x4*>(gp); ga1 = *reinterpret_cast<const f32x4*>(gp + 4);
    if (q.relu && beta) {
      be0 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + g * 8);
      be1 = *reinterpret_cast<const f32x4*>(beta + (long)lb * q.C + g * 8 + 4);
    }
  };
  auto one = [&](long i, const bf16x8& d, const bf16x8& xv, const bf16x8& yv) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float mu = e < 4 ? mu0[e] : mu1[e - 4], iv = e < 4 ? iv0[e] : iv1[e - 4];
      const float ga = e < 4 ? ga0[e] : ga1[e - 4];
      float dd = bf2f(d[e]);
      if (q.relu) {
        const bool on = beta ? ((bf2f(xv[e]) - mu) * iv * ga + (e < 4 ? be0[e] : be1[e - 4]) > 0.f) : (bf2f(yv[e]) > 0.f);
        if (!on) dd = 0.f;
      }
      const float gg = dd * ga;
      const float xh = (bf2f(xv[e]) - mu) * iv;
      o[e] = f2bf(iv * (gg - (e < 4 ? ma0[e] : ma1[e - 4]) - xh * (e < 4 ? mb0[e] : mb1[e - 4])));
    }
    reinterpret_cast<bf16x8*>(dx)[i] = o;
  };
  const bool ry = q.relu && !beta;
  long row = rbeg;
  for (; row + rstride < rows; row += 2 * rstride) {
    const long ia = row * cg + g, ib = (row + rstride) * cg + g;
    const bf16x8 da = reinterpret_cast<const bf16x8*>(dy)[ia], xa = reinterpret_cast<const bf16x8*>(x)[ia];
    const bf16x8 db = reinterpret_cast<const bf16x8*>(dy)[ib], xb = reinterpret_cast<const bf16x8*>(x)[ib];
    bf16x8 ya = da, yb = db;
    if (ry) { ya = reinterpret_cast<const bf16x8*>(y)[ia]; yb = reinterpret_cast<const bf16x8*>(y)[ib]; }
    int n = (int)(row / q.HW);
    if (n != n_cur) { params(n); n_cur = n; }
    one(ia, da, xa, ya);
    n = (int)((row + rstride) / q.HW);
    if (n != n_cur) { params(n); n_cur = n; }
    one(ib, db, xb, yb);
  }
  for (; row < rows; row += rstride) {
    const long ia = row * cg + g;
    const bf16x8 da = reinterpret_cast<const bf16x8*>(dy)[ia], xa = reinterpret_cast<const bf16x8*>(x)[ia];
    bf16x8 ya = da;
    if (ry) ya = reinterpret_cast<const bf16x8*>(y)[ia];
    const int n = (int)(row / q.HW);
    if (n != n_cur) { params(n); n_cur = n; }
    one(ia, da, xa, ya);
  }
}

static int cbn_bwd_parts(int HW) {
  int hw_parts = HW / 64;
  return hw_parts < 1 ? 1 : (hw_parts > 8 ? 8 : hw_parts);
}
// ws_floats: what the caller's workspace holds.  N*2*C + groups*2*C (the contract of gank_cbn_bwd / _remask): the pixel parts of a
// sample meet by fp32 atomics in a zero-filled buffer; gank_cbn_bwd_ws_floats(): every part writes rows of its own and the last
// part of a sample to finish adds them in a fixed order -- no fill launch, no atomics on the sums (in a train step they cost ~3x
// their micro-benchmark price), the same bits every run.  (Adding the parts inside the table launch instead -- 8x its loads on a
// 44-block grid -- cost 20 us per call.)  Ticket words: self-resetting, a group per call in rotation (calls on one stream follow
// each other; N <= CBN_TICKETS).
constexpr int CBN_TICKETS = 4096, CBN_TICKET_GROUPS = 8;
__device__ unsigned cbn_tickets[CBN_TICKETS * CBN_TICKET_GROUPS];
static int cbn_bwd_impl(const void* dy, const void* x, const void* y, const float* beta, const int32_t* labels, const float* gamma,
                        const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                        int groups, int n_labels, int relu, void* stream, long ws_floats = 0) {
  GANK_REQUIRE(dy && x && labels && gamma && stats && dx && dgamma && dbeta && ws, "cbn_bwd: null pointer");
  GANK_REQUIRE(!relu || y || beta, "cbn_bwd: relu backward needs y or beta");
  CbnGeom q;
  if (cbn_geom(q, N, HW, C, groups, n_labels, relu)) return 1;
  hipStream_t s = (hipStream_t)stream;
  const int hw_parts = cbn_bwd_parts(HW);
  const bool slab = hw_parts > 1 && N <= CBN_TICKETS && ws_floats >= (long)(hw_parts + 1) * N * 2 * C + (long)groups * 2 * C;
  float* S = ws;                                       // [N][2][C]
  float* M = ws + (long)N * 2 * C;                     // [groups][2][C]
  float* Sp = slab ? M + (long)groups * 2 * C : nullptr;      // [hw_parts][N][2][C]
  unsigned* tickets = nullptr;
  if (slab) {
    static std::atomic<unsigned*> ticket_addr[64];     // per device, looked up once (the first call of a process is an eager one)
    static std::atomic<unsigned> ticket_group{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    unsigned* base = ticket_addr[dev & 63].load(std::memory_order_relaxed);
    if (!base) {
      if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(cbn_tickets)) != hipSuccess || !base) return gank_set_error("cbn_bwd: ticket words not found");
      ticket_addr[dev & 63].store(base, std::memory_order_relaxed);
    }
    tickets = base + (ticket_group.fetch_add(1, std::memory_order_relaxed) % CBN_TICKET_GROUPS) * CBN_TICKETS;
  }
  // zeroed by a kernel, not hipMemsetAsync: inside a captured hipGraph the memset node was observed to race with
  // its kernel neighbours (intermittent NaN generator gradients under graph replay, never in eager mode)
  if (hw_parts > 1 && !slab) hipLaunchKernelGGL(cbn_zero_kernel, dim3(cdiv(N * 2 * C / 4, 256)), dim3(256), 0, s, S, N * 2 * C / 4);
  hipLaunchKernelGGL(cbn_bwd_sums_kernel, dim3(N * hw_parts), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (const bf16*)y, stats, S, q, hw_parts, labels, gamma, beta,
                     Sp, tickets);
  hipLaunchKernelGGL(cbn_bwd_tables_kernel, dim3(cdiv(C, 64), n_labels + 1), dim3(256), 0, s, S, labels, gamma, dgamma, dbeta, M, q);
  const long total8 = (long)N * HW * (C / 8);
  long blocks = (total8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(cbn_bwd_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, (const bf16*)y, labels, gamma, stats, M, (bf16*)dx, q, total8, beta);
  GANK_LAUNCH_OK("cbn_bwd");
  return 0;
}

extern "C" int gank_cbn_bwd(const void* dy, const void* x, const void* y, const int32_t* labels, const float* gamma,
                            const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                            int groups, int n_labels, int relu, void* stream) {
  return cbn_bwd_impl(dy, x, y, nullptr, labels, gamma, stats, dx, dgamma, dbeta, ws, N, HW, C, groups, n_labels, relu, stream);
}
// the same with the relu mask recomputed from x, gamma and beta (what the forward pass computed, bit for bit) instead of read from y
extern "C" int gank_cbn_bwd_remask(const void* dy, const void* x, const float* beta, const int32_t* labels, const float* gamma,
                                   const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, int N, int HW, int C,
                                   int groups, int n_labels, int relu, void* stream) {
  GANK_REQUIRE(!relu || beta, "cbn_bwd_remask: relu backward needs beta");
  return cbn_bwd_impl(dy, x, nullptr, beta, labels, gamma, stats, dx, dgamma, dbeta, ws, N, HW, C, groups, n_labels, relu, stream);
}

extern "C" long gank_cbn_bwd_ws_floats(int N, int HW, int C, int groups) {
  const int parts = cbn_bwd_parts(HW);
  return (long)(parts > 1 ? parts + 1 : 1) * N * 2 * C + (long)groups * 2 * C;
}
// either mask form (y, or beta with y = NULL) on a workspace of ws_floats floats
extern "C" int gank_cbn_bwd_ws(const void* dy, const void* x, const void* y, const float* beta, const int32_t* labels, const float* gamma,
                               const float* stats, void* dx, float* dgamma, float* dbeta, float* ws, long ws_floats, int N, int HW, int C,
                               int groups, int n_labels, int relu, void* stream) {
  GANK_REQUIRE(ws_floats >= (long)N * 2 * C + (long)groups * 2 * C, "cbn_bwd_ws: workspace of %ld floats, need at least %ld", ws_floats,
               (long)N * 2 * C + (long)groups * 2 * C);
  return cbn_bwd_impl(dy, x, beta ? nullptr : y, beta, labels, gamma, stats, dx, dgamma, dbeta, ws, N, HW, C, groups, n_labels, relu, stream, ws_floats);
}

// (mean, invstd) per tower and channel from the sums a conv epilogue accumulated (gank_conv2d_fprop_stats): what
// cbn_apply_sums_kernel computes in its prologue, as a launch of its own for consumers that normalise inside another kernel
__global__ void cbn_stats_from_sums_kernel(const float* __restrict__ sums, const float* __restrict__ shift, float* __restrict__ stats,
                                           int C, int groups, float M, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x, grp = blockIdx.y;
  if (c >= C) return;
  float t1 = 0.f, t2 = 0.f;
#pragma unroll
  for (int sl = 0; sl < GANK_STAT_SLOTS; sl++) {
    t1 += sums[(((long)grp * GANK_STAT_SLOTS + sl) * 2) * C + c];
    t2 += sums[(((long)grp * GANK_STAT_SLOTS + sl) * 2 + 1) * C + c];
  }
  const float m1 = t1 / M, m2 = t2 / M;
  stats[((long)grp * 2) * C + c] = m1 + (shift ? shift[c] : 0.f);
  stats[((long)grp * 2 + 1) * C + c] = 1.f / sqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
}
extern "C" int gank_cbn_stats_from_sums(const float* sums, const float* shift, float* stats, int C, int groups, long rows_per_group,
                                        float eps, void* stream) {
  GANK_REQUIRE(sums && stats && C > 0 && groups > 0 && rows_per_group > 0 && eps > 0.f, "cbn_stats_from_sums: bad arguments");
  hipLaunchKernelGGL(cbn_stats_from_sums_kernel, dim3(cdiv(C, 256), groups), dim3(256), 0, (hipStream_t)stream, sums, shift, stats, C, groups,
                     (float)rows_per_group, eps);
  GANK_LAUNCH_OK("cbn_stats_from_sums");
  return 0;
}
// the statistics pass alone (stats [groups][2][C]; ws as gank_cbn_fwd)
extern "C" int gank_cbn_stats(const void* x, float* stats, float* ws, int N, int HW, int C, int groups, float eps, void* stream) {
  GANK_REQUIRE(x && stats && ws, "cbn_stats: null pointer");
  CbnGeom q;
  if (cbn_geom(q, N, HW, C, groups, 1, 0)) return 1;
  q.eps = eps;
  hipStream_t s = (hipStream_t)stream;
  if (cbn_fwd_small_ok(q)) {            // small towers: the forward pass's one-launch kernel without its apply (the same statistics, bit for bit)
    hipLaunchKernelGGL(cbn_fwd_small_kernel, dim3(groups * (C / 64)), dim3(512), 0, s, (const bf16*)x, (const int*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (bf16*)nullptr, stats, q);
    GANK_LAUNCH_OK("cbn_stats");
    return 0;
  }
  hipLaunchKernelGGL(cbn_stats_kernel, dim3(groups * q.parts), dim3(CBN_NT), 0, s, (const bf16*)x, ws, q);
  hipLaunchKernelGGL(cbn_finalize_kernel, dim3(cdiv(C, 16), groups), dim3(256), 0, s, ws, stats, q);
  GANK_LAUNCH_OK("cbn_stats");
  return 0;
}
