// The remaining normalisation layers of common/ops/normalization.py, at op level:
//   layer_norm  (:62-102, tf.contrib.layers.layer_norm, begin_norm_axis=1, begin_params_axis=-1): moments over
//               (H,W,C) per sample, gamma/beta over C -- what Normalize() dispatches to for 'D.' names when
//               NORMALIZATION_D is set (gan_cifar_resnet.py:98-99);
//   pixel_norm  (:125-140, PGGAN): x * rsqrt(mean_c(x^2) + eps) per pixel.
// instance_norm (:105-122) is the conditional-batch-norm kernel set with one tower per sample (cbn.hip).
// All are HBM-bound wavefront reductions over bf16 NHWC tensors: 16 bytes per lane, fp32 statistics.
#include "gank_common.h"

// ---- layer norm: one workgroup (1024 threads) per sample, three passes over a row that stays in L2 ---------
__global__ __launch_bounds__(1024) void layer_norm_fwd_kernel(const bf16* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, bf16* __restrict__ y,
                                                             float* __restrict__ stats, long R, int C, float eps) {
  __shared__ float red[16];
  const long n = blockIdx.x;
  const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + n * R);
  const long R8 = R >> 3;
  float s = 0.f;
  for (long i = threadIdx.x; i < R8; i += blockDim.x) {
    const bf16x8 v = xr[i];
#pragma unroll
    for (int e = 0; e < 8; e++) s += bf2f(v[e]);
  }
  const float mean = block_sum(s, red) / (float)R;
  float q = 0.f;
  for (long i = threadIdx.x; i < R8; i += blockDim.x) {
    const bf16x8 v = xr[i];
#pragma unroll
    for (int e = 0; e < 8; e++) { const float d = bf2f(v[e]) - mean; q += d * d; }
  }
  const float invstd = rsqrtf(block_sum(q, red) / (float)R + eps);
  if (threadIdx.x == 0) { stats[2 * n] = mean; stats[2 * n + 1] = invstd; }
  bf16x8* yr = reinterpret_cast<bf16x8*>(y + n * R);
  const int C8 = C >> 3;
  for (long i = threadIdx.x; i < R8; i += blockDim.x) {
    const int c0 = (int)(i % C8) * 8;
    const bf16x8 v = xr[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf((bf2f(v[e]) - mean) * invstd * gamma[c0 + e] + beta[c0 + e]);
    yr[i] = o;
  }
}

// dx = invstd (g - mean(g) - xhat mean(g xhat)),  g = dy gamma[c];  dgamma[c] += sum dy xhat, dbeta[c] += sum dy.
// Channel sums meet in LDS (C <= 2048) and leave as one atomic per channel per sample.
__global__ __launch_bounds__(1024) void layer_norm_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                             const float* __restrict__ gamma, const float* __restrict__ stats,
                                                             bf16* __restrict__ dx, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta, long R, int C) {
  __shared__ float red[16];
  __shared__ float cs[2][2048];
  const long n = blockIdx.x;
  const float mean = stats[2 * n], invstd = stats[2 * n + 1];
  const bf16x8* xr = reinterpret_cast<const bf16x8*>(x + n * R);
  const bf16x8* gr = reinterpret_cast<const bf16x8*>(dy + n * R);
  const long R8 = R >> 3;
  const int C8 = C >> 3;
  for (int c = threadIdx.x; c < C; c += blockDim.x) { cs[0][c] = 0.f; cs[1][c] = 0.f; }
  __syncthreads();
  float s1 = 0.f, s2 = 0.f;
  for (long i = threadIdx.x; i < R8; i += blockDim.x) {
    const int c0 = (int)(i % C8) * 8;
    const bf16x8 v = xr[i], g = gr[i];
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xh = (bf2f(v[e]) - mean) * invstd, d = bf2f(g[e]);
      const float gg = d * gamma[c0 + e];
      s1 += gg;
      s2 += gg * xh;
      atomicAdd(&cs[0][c0 + e], d * xh);      // LDS atomics: 1024 threads spread over C channels
      atomicAdd(&cs[1][c0 + e], d);
    }
  }
  const float m1 = block_sum(s1, red) / (float)R;
  const float m2 = block_sum(s2, red) / (float)R;     // block_sum's barriers also order the LDS atomics above
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    atomicAdd(dgamma + c, cs[0][c]);
    atomicAdd(dbeta + c, cs[1][c]);
  }
  bf16x8* dr = reinterpret_cast<bf16x8*>(dx + n * R);
  for (long i = threadIdx.x; i < R8; i += blockDim.x) {
    const int c0 = (int)(i % C8) * 8;
    const bf16x8 v = xr[i], g = gr[i];
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const float xh = (bf2f(v[e]) - mean) * invstd;
      o[e] = f2bf(invstd * (bf2f(g[e]) * gamma[c0 + e] - m1 - xh * m2));
    }
    dr[i] = o;
  }
}

extern "C" int gank_layer_norm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* stats, int N, int HW,
                                   int C, float eps, void* stream) {
  GANK_REQUIRE(x && gamma && beta && y && stats && N > 0 && HW > 0, "layer_norm_fwd: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && C <= 2048, "layer_norm: C=%d unsupported (need C %% 8 == 0, C <= 2048)", C);
  hipLaunchKernelGGL(layer_norm_fwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)x, gamma, beta, (bf16*)y, stats,
                     (long)HW * C, C, eps);
  GANK_LAUNCH_OK("layer_norm_fwd");
  return 0;
}

extern "C" int gank_layer_norm_bwd(const void* dy, const void* x, const float* gamma, const float* stats, void* dx, float* dgamma,
                                   float* dbeta, int N, int HW, int C, void* stream) {
  GANK_REQUIRE(dy && x && gamma && stats && dx && dgamma && dbeta && N > 0 && HW > 0, "layer_norm_bwd: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && C <= 2048, "layer_norm: C=%d unsupported (need C %% 8 == 0, C <= 2048)", C);
  hipLaunchKernelGGL(layer_norm_bwd_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, (const bf16*)dy, (const bf16*)x, gamma, stats,
                     (bf16*)dx, dgamma, dbeta, (long)HW * C, C);
  GANK_LAUNCH_OK("layer_norm_bwd");
  return 0;
}

// ---- pixel norm: one wave per pixel (C <= 512: 8 channels per lane) -------------------------------------------
// fwd: y = a x, a = rsqrt(mean_c x^2 + eps).   bwd: dx = a dy - x a^3 (sum_c dy x) / C   (a recomputed, nothing saved)
template <bool BWD>
__global__ __launch_bounds__(256) void pixel_norm_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, bf16* __restrict__ out,
                                                         long pixels, int C, float eps) {
  const long p = blockIdx.x * 4L + (threadIdx.x >> 6);
  if (p >= pixels) return;
  const int lane = threadIdx.x & 63, c0 = lane * 8;
  const bool on = c0 < C;
  bf16x8 v, g;
#pragma unroll
  for (int e = 0; e < 8; e++) { v[e] = f2bf(0.f); g[e] = f2bf(0.f); }
  if (on) {
    v = *reinterpret_cast<const bf16x8*>(x + p * C + c0);
    if (BWD) g = *reinterpret_cast<const bf16x8*>(dy + p * C + c0);
  }
  float sq = 0.f, dot = 0.f;
#pragma unroll
  for (int e = 0; e < 8; e++) { const float t = bf2f(v[e]); sq += t * t; dot += bf2f(g[e]) * t; }
  sq = wave_sum(sq);
  const float a = rsqrtf(sq / (float)C + eps);
  float k = 0.f;
  if (BWD) k = wave_sum(dot) * a * a * a / (float)C;
  if (on) {
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(BWD ? a * bf2f(g[e]) - bf2f(v[e]) * k : a * bf2f(v[e]));
    *reinterpret_cast<bf16x8*>(out + p * C + c0) = o;
  }
}

extern "C" int gank_pixel_norm_fwd(const void* x, void* y, long pixels, int C, float eps, void* stream) {
  GANK_REQUIRE(x && y && pixels > 0, "pixel_norm_fwd: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && C <= 512, "pixel_norm: C=%d unsupported (need C %% 8 == 0, C <= 512)", C);
  hipLaunchKernelGGL(pixel_norm_kernel<false>, dim3((unsigned)cdiv(pixels, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)nullptr,
                     (const bf16*)x, (bf16*)y, pixels, C, eps);
  GANK_LAUNCH_OK("pixel_norm_fwd");
  return 0;
}

extern "C" int gank_pixel_norm_bwd(const void* dy, const void* x, void* dx, long pixels, int C, float eps, void* stream) {
  GANK_REQUIRE(dy && x && dx && pixels > 0, "pixel_norm_bwd: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && C <= 512, "pixel_norm: C=%d unsupported (need C %% 8 == 0, C <= 512)", C);
  hipLaunchKernelGGL(pixel_norm_kernel<true>, dim3((unsigned)cdiv(pixels, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy,
                     (const bf16*)x, (bf16*)dx, pixels, C, eps);
  GANK_LAUNCH_OK("pixel_norm_bwd");
  return 0;
}
