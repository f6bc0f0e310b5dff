// Kernels only the ACGAN configuration needs (BASELINE.json config 3; ACGAN/train.py:89-121, ACGAN/model.py:49-90):
//   * the second derivative of train-mode batch normalisation (the WGAN-GP term differentiates the critic's INPUT
//     gradient with respect to the critic's weights, ACGAN/train.py:99-107: every op of the critic is differentiated
//     twice; conv / dense / pooling / leaky-relu compose from their own first-order kernels, batch norm does not);
//   * the moving-statistics update of tf.contrib.layers.batch_norm (common/ops/normalization.py:8-24) as one launch;
//   * the gradient-penalty reduction, the real/fake interpolation, spatial mean and its adjoint, uniform random numbers.
// All HBM / latency bound; 16 bytes per lane where the shape allows.
#include "gank_common.h"

// ---------------------------------------------------------------------------------------------------------------
// batch norm, second order.  First backward (per channel, M = N*HW rows, xh = (x - mu) * s, s = invstd):
//     dx = gamma * s * (dy - mean(dy) - xh * mean(dy * xh))
// Given a = dL/d(dx), this returns  gI = dL/dx,  ggO = dL/d(dy)  and accumulates  gG = dL/dgamma  (the terms of
// upstream gradients with respect to dgamma / dbeta do not occur: the penalty only uses the input gradient).
// With  A0 = sum a, A1 = sum a*(x-mu), G0 = sum dy, G1 = sum dy*(x-mu), AG = sum a*dy :
//     ggO = gamma*s/M * (M*a - A0 - (x-mu)*s^2*A1)
//     gI  = gamma * [ (x-mu)*s^3/M * (A0*G0/M - AG + 3*s^2*G1*A1/M) + A1*s^3/M * (G0/M - dy) + G1*s^3/M * (A0/M - a) ]
//     gG  = s * (AG - A0*G0/M - s^2*A1*G1/M)
// ---------------------------------------------------------------------------------------------------------------
__global__ void bn2_zero_kernel(float* __restrict__ p, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0.f;
}

__global__ __launch_bounds__(256) void bn2_sums_kernel(const bf16* __restrict__ a, const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                       const float* __restrict__ stats, float* __restrict__ ws, long rows, int C, long rows_per_block) {
  // thread = (8-channel group, row lane); ws [5][C] (atomics: a few hundred blocks x C adds)
  const int cg = C >> 3, RL = 256 / cg;
  const int g = threadIdx.x % cg, rl = threadIdx.x / cg;
  const long r0 = blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block;
  if (r1 > rows) r1 = rows;
  float mu[8], s[5][8];
#pragma unroll
  for (int e = 0; e < 8; e++) {
    mu[e] = stats[g * 8 + e];
#pragma unroll
    for (int k = 0; k < 5; k++) s[k][e] = 0.f;
  }
  if (rl < RL)
    for (long r = r0 + rl; r < r1; r += RL) {
      const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + r * C + g * 8);
      const bf16x8 vg = *reinterpret_cast<const bf16x8*>(dy + r * C + g * 8);
      const bf16x8 vx = *reinterpret_cast<const bf16x8*>(x + r * C + g * 8);
#pragma unroll
      for (int e = 0; e < 8; e++) {
        const float fa = bf2f(va[e]), fg = bf2f(vg[e]), d = bf2f(vx[e]) - mu[e];
        s[0][e] += fa; s[1][e] += fa * d; s[2][e] += fg; s[3][e] += fg * d; s[4][e] += fa * fg;
      }
    }
  __shared__ float red[256 * 8];
  for (int k = 0; k < 5; k++) {
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; e++) red[threadIdx.x * 8 + e] = s[k][e];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = 0.f;
      for (int l = 0; l < RL; l++) t += red[(l * cg + (c >> 3)) * 8 + (c & 7)];
      atomicAdd(ws + k * C + c, t);
    }
  }
}

__global__ __launch_bounds__(256) void bn2_apply_kernel(const bf16* __restrict__ a, const bf16* __restrict__ dy, const bf16* __restrict__ x,
                                                        const float* __restrict__ gamma, const float* __restrict__ stats, const float* __restrict__ ws,
                                                        bf16* __restrict__ gI, bf16* __restrict__ ggO, float* __restrict__ gG, long rows, int C) {
  const long n8 = rows * (C >> 3);
  const float M = (float)rows;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const int c0 = (int)(i % (C >> 3)) * 8;
    const bf16x8 va = *reinterpret_cast<const bf16x8*>(a + i * 8);
    const bf16x8 vg = *reinterpret_cast<const bf16x8*>(dy + i * 8);
    const bf16x8 vx = *reinterpret_cast<const bf16x8*>(x + i * 8);
    bf16x8 oI, oO;
#pragma unroll
    for (int e = 0; e < 8; e++) {
      const int c = c0 + e;
      const float mu = stats[c], s = stats[C + c], gm = gamma[c];
      const float A0 = ws[c], A1 = ws[C + c], G0 = ws[2 * C + c], G1 = ws[3 * C + c], AG = ws[4 * C + c];
      const float fa = bf2f(va[e]), fg = bf2f(vg[e]), d = bf2f(vx[e]) - mu;
      const float s2 = s * s, s3 = s2 * s;
      oO[e] = f2bf(gm * s / M * (M * fa - A0 - d * s2 * A1));
      const float all_sub = A0 * G0 / M - AG + 3.f * s2 * G1 * A1 / M;
      oI[e] = f2bf(gm * (d * s3 / M * all_sub + A1 * s3 / M * (G0 / M - fg) + G1 * s3 / M * (A0 / M - fa)));
    }
    *reinterpret_cast<bf16x8*>(gI + i * 8) = oI;
    *reinterpret_cast<bf16x8*>(ggO + i * 8) = oO;
  }
  if (gG && blockIdx.x == 0)
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
      const float s = stats[C + c];
      gG[c] += s * (ws[4 * C + c] - ws[c] * ws[2 * C + c] / M - s * s * ws[C + c] * ws[3 * C + c] / M);
    }
}

extern "C" int gank_bn_bwd_bwd(const void* ggI, const void* dy, const void* x, const float* gamma, const float* stats, void* gI, void* ggO,
                               float* gG, float* ws, long rows, int C, void* stream) {
  GANK_REQUIRE(ggI && dy && x && gamma && stats && gI && ggO && ws && rows > 0, "bn_bwd_bwd: null pointer");
  GANK_REQUIRE(C % 8 == 0 && C <= 2048 && 256 % (C / 8) == 0, "bn_bwd_bwd: unsupported channel count %d", C);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn2_zero_kernel, dim3((5 * C + 255) / 256), dim3(256), 0, s, ws, 5 * C);
  long blocks = (rows + 255) / 256;
  if (blocks > 1024) blocks = 1024;
  const long rpb = (rows + blocks - 1) / blocks;
  hipLaunchKernelGGL(bn2_sums_kernel, dim3((unsigned)((rows + rpb - 1) / rpb)), dim3(256), 0, s, (const bf16*)ggI, (const bf16*)dy, (const bf16*)x, stats,
                     ws, rows, C, rpb);
  long ab = (rows * (C / 8) + 255) / 256;
  if (ab > 2048) ab = 2048;
  hipLaunchKernelGGL(bn2_apply_kernel, dim3((unsigned)ab), dim3(256), 0, s, (const bf16*)ggI, (const bf16*)dy, (const bf16*)x, gamma, stats, ws,
                     (bf16*)gI, (bf16*)ggO, gG, rows, C);
  GANK_LAUNCH_OK("bn_bwd_bwd");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// tf.contrib.layers.batch_norm moving statistics (decay, zero_debias_moving_mean=True; normalization.py:10-22), one
// launch per call for all towers:  for each tower g in order:
//   moving_variance <- decay*mv + (1-decay) * var_g * n/(n-1);  biased <- decay*biased + (1-decay)*mean_g;
//   local_step += 1;  moving_mean <- biased / (1 - decay^local_step)
// stats [groups][2][C] = (mean, invstd) as the forward kernels leave them; var = 1/invstd^2 - eps.
// ---------------------------------------------------------------------------------------------------------------
__global__ void bn_moving_kernel(const float* __restrict__ stats, float* __restrict__ mm, float* __restrict__ mv, float* __restrict__ biased,
                                 float* __restrict__ step, int C, int groups, float decay, float eps, float unbias) {
  // ONE block: every thread reads local_step before thread 0 rewrites it (barrier in between, no early exit)
  const int c = threadIdx.x;
  float st = step[0];
  if (c < C) {
    float b = biased[c], v = mv[c], m = mm[c];
    for (int g = 0; g < groups; g++) {
      const float mean = stats[(g * 2) * C + c], is = stats[(g * 2 + 1) * C + c];
      v = decay * v + (1.f - decay) * (1.f / (is * is) - eps) * unbias;
      b = decay * b + (1.f - decay) * mean;
      st += 1.f;
      m = b / (1.f - powf(decay, st));
    }
    mm[c] = m; mv[c] = v; biased[c] = b;
  }
  __syncthreads();
  if (c == 0) step[0] = step[0] + (float)groups;
}

extern "C" int gank_bn_moving_update(const float* stats, float* moving_mean, float* moving_var, float* biased, float* local_step, int C,
                                     int groups, long count, float decay, float eps, void* stream) {
  GANK_REQUIRE(stats && moving_mean && moving_var && biased && local_step && C > 0 && groups > 0 && count > 0, "bn_moving_update: bad arguments");
  GANK_REQUIRE(C <= 1024, "bn_moving_update: one block handles all channels (C = %d > 1024)", C);   // one block: step[0] is read before it is written
  const float unbias = count > 1 ? (float)count / (float)(count - 1) : 1.f;
  hipLaunchKernelGGL(bn_moving_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stats, moving_mean, moving_var, biased, local_step, C, groups,
                     decay, eps, unbias);
  GANK_LAUNCH_OK("bn_moving_update");
  return 0;
}

// ---------------------------------------------------------------------------------------------------------------
// gradient penalty  lambda * mean_n (sqrt(sum_d g[n,d]^2 + 1e-10) - 1)^2   (ACGAN/train.py:104-106; misc.py WGAN-GP)
// and its derivative with respect to g; one block per sample, then one block sums the per-sample terms.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gp_rows_kernel(const bf16* __restrict__ g, float* __restrict__ pen, float* __restrict__ dg, int N, long D, float lambda) {
  __shared__ float red[16];
  const long n = blockIdx.x;
  float acc = 0.f;
  for (long i = threadIdx.x; i < D; i += blockDim.x) { const float v = bf2f(g[n * D + i]); acc += v * v; }
  const float tot = block_sum(acc, red);
  const float slope = sqrtf(tot + 1e-10f);
  if (threadIdx.x == 0) pen[n] = lambda * (slope - 1.f) * (slope - 1.f) / (float)N;
  const float k = lambda * 2.f * (slope - 1.f) / (slope * (float)N);
  for (long i = threadIdx.x; i < D; i += blockDim.x) dg[n * D + i] = k * bf2f(g[n * D + i]);
}
__global__ __launch_bounds__(256) void sum_small_kernel(const float* __restrict__ v, float* __restrict__ out, int n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += v[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = tot;
}
extern "C" int gank_gp_loss(const void* grad, float* loss, void* dgrad, float* ws, int N, long D, float lambda, void* stream) {
  GANK_REQUIRE(grad && loss && dgrad && ws && N > 0 && D > 0, "gp_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gp_rows_kernel, dim3(N), dim3(256), 0, s, (const bf16*)grad, ws, (float*)dgrad, N, D, lambda);
  hipLaunchKernelGGL(sum_small_kernel, dim3(1), dim3(256), 0, s, ws, loss, N);
  GANK_LAUNCH_OK("gp_loss");
  return 0;
}

// interpolates = real + alpha[n] * (fake - real)     (ACGAN/train.py:99-101)
__global__ void lerp_rows_kernel(const bf16* __restrict__ real, const bf16* __restrict__ fake, const float* __restrict__ alpha, bf16* __restrict__ out,
                                 long D, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const float a = alpha[i / D], r = bf2f(real[i]);
    out[i] = f2bf(r + a * (bf2f(fake[i]) - r));
  }
}
extern "C" int gank_lerp_rows(const void* real, const void* fake, const float* alpha, void* out, int N, long D, void* stream) {
  GANK_REQUIRE(real && fake && alpha && out && N > 0 && D > 0, "lerp_rows: bad arguments");
  const long total = (long)N * D;
  long blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(lerp_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)real, (const bf16*)fake, alpha,
                     (bf16*)out, D, total);
  GANK_LAUNCH_OK("lerp_rows");
  return 0;
}

// y[n,c] = scale * sum_hw x[n,hw,c]  and its adjoint  y[n,hw,c] = scale * g[n,c]     (tf.reduce_mean(axis=[1,2]), model.py:72)
__global__ void sum_hw_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int HW, int C, float scale, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / C;
    const int c = (int)(i - n * C);
    float acc = 0.f;
    for (int p = 0; p < HW; p++) acc += bf2f(x[(n * HW + p) * C + c]);
    y[i] = f2bf(acc * scale);
  }
}
__global__ void bcast_hw_kernel(const bf16* __restrict__ g, bf16* __restrict__ y, int HW, int C, float scale, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long n = i / ((long)HW * C);
    const int c = (int)(i % C);
    y[i] = f2bf(bf2f(g[n * C + c]) * scale);
  }
}
extern "C" int gank_sum_hw(const void* x, void* y, int N, int HW, int C, float scale, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && HW > 0 && C > 0, "sum_hw: bad arguments");
  const long total = (long)N * C;
  hipLaunchKernelGGL(sum_hw_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, HW, C, scale, total);
  GANK_LAUNCH_OK("sum_hw");
  return 0;
}
extern "C" int gank_bcast_hw(const void* g, void* y, int N, int HW, int C, float scale, void* stream) {
  GANK_REQUIRE(g && y && N > 0 && HW > 0 && C > 0, "bcast_hw: bad arguments");
  const long total = (long)N * HW * C;
  long blocks = (total + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(bcast_hw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, (bf16*)y, HW, C, scale, total);
  GANK_LAUNCH_OK("bcast_hw");
  return 0;
}
