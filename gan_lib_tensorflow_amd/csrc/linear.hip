// Small dense layers (common/ops/linear.py:161-180: tf.matmul + bias_add) on fp32 master weights.
// The SNGAN critic has two of them per forward -- D.Embedding_y [128,300]x[300,128] and D.Output [128,128]x[128,1]
// (gan_cifar_resnet.py:296-304) -- 10 MFLOP and 33 kFLOP: launch-latency problems, not MFMA problems.  Through the
// implicit-GEMM engine each cost 16-38 us (bf16 operand preparation, K-packed gather with per-element index
// arithmetic, 2-4 workgroups on 256 CUs).  Here: no operand preparation, weights read as stored, one thread per
// output, the reduction axis pipelined by unrolling; every call is a few microseconds.
// (The generator's 128 -> 16384 input layer stays on the MFMA engine.)
#include "gank_common.h"

// All kernels keep the dependent-load chain short: every wave issues its weight/activation loads in batches of 16
// independent requests, wave-uniform operands are fetched once per 64 values with a lane-indexed load and broadcast
// with v_readlane, and the reduction axis is split over the 4 waves of a block (summed through LDS).
__device__ __forceinline__ float lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

// y[m][c] = sum_k x[m][k] w[k][c] + b[c];  lanes over c (coalesced w, y), RB rows per thread reuse each w load
__device__ __forceinline__ void store_out(bf16* p, float v) { *p = f2bf(v); }
__device__ __forceinline__ void store_out(float* p, float v) { *p = v; }

template <int RB, typename OUT>
__global__ __launch_bounds__(256) void linear_fwd_wide_kernel(const bf16* __restrict__ x, const float* __restrict__ w,
                                                              const float* __restrict__ b, OUT* __restrict__ y, int M, int K, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int m0 = blockIdx.y * RB;
  const int cc = c < C ? c : C - 1;
  const int kq = (K + 3) / 4, k0 = wv * kq, k1 = min(K, k0 + kq);
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) acc[r] = 0.f;
  for (int kc = k0; kc < k1; kc += 64) {
    float xv[RB];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int m = m0 + r < M ? m0 + r : M - 1;
      xv[r] = (kc + lane < k1) ? bf2f(x[(long)m * K + kc + lane]) : 0.f;
    }
    // 64 reduction steps in 4 batches of 16 independent weight loads (row index clamped: lanes past k1 hold x = 0)
#pragma unroll
    for (int jb = 0; jb < 64; jb += 16) {
      if (kc + jb >= k1) break;
      float wt[16];
#pragma unroll
      for (int u = 0; u < 16; u++) wt[u] = w[(long)min(kc + jb + u, K - 1) * C + cc];
#pragma unroll
      for (int u = 0; u < 16; u++)
#pragma unroll
        for (int r = 0; r < RB; r++) acc[r] += lane_bcast(xv[r], jb + u) * wt[u];
    }
  }
  __shared__ float red[4][RB][64];
#pragma unroll
  for (int r = 0; r < RB; r++) red[wv][r][lane] = acc[r];
  __syncthreads();
  if (wv == 0 && c < C) {
    const float bv = b ? b[c] : 0.f;
#pragma unroll
    for (int r = 0; r < RB; r++)
      if (m0 + r < M) store_out(&y[(long)(m0 + r) * C + c], red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane] + bv);
  }
}

// narrow outputs (C < 64): one wave per (m, c), lanes over k
__global__ void linear_fwd_narrow_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                         bf16* __restrict__ y, int M, int K, int C) {
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= M * C) return;
  const int m = o / C, c = o - m * C;
  float s = 0.f;
  for (int k = threadIdx.x & 63; k < K; k += 64) s += bf2f(x[(long)m * K + k]) * w[(long)k * C + c];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) y[o] = f2bf(s + (b ? b[c] : 0.f));
}

// dx[m][k] = sum_c dy[m][c] w[k][c]: lanes over c (coalesced weight rows), a wave owns KB weight rows x RB batch rows
// and reduces across its lanes
template <int KB, int RB>
__global__ __launch_bounds__(256) void linear_bwd_data_kernel(const bf16* __restrict__ dy, const float* __restrict__ w,
                                                              bf16* __restrict__ dx, int M, int K, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k0 = (blockIdx.x * 4 + wv) * KB;
  const int m0 = blockIdx.y * RB;
  if (k0 >= K) return;
  float acc[KB][RB];
#pragma unroll
  for (int i = 0; i < KB; i++)
#pragma unroll
    for (int r = 0; r < RB; r++) acc[i][r] = 0.f;
  for (int c0 = 0; c0 < C; c0 += 64) {
    const int c = c0 + lane;
    float g[RB], wr[KB];
#pragma unroll
    for (int r = 0; r < RB; r++) g[r] = (c < C && m0 + r < M) ? bf2f(dy[(long)(m0 + r) * C + c]) : 0.f;
#pragma unroll
    for (int i = 0; i < KB; i++) wr[i] = (c < C && k0 + i < K) ? w[(long)(k0 + i) * C + c] : 0.f;
#pragma unroll
    for (int i = 0; i < KB; i++)
#pragma unroll
      for (int r = 0; r < RB; r++) acc[i][r] += g[r] * wr[i];
  }
#pragma unroll
  for (int i = 0; i < KB; i++)
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const float t = wave_sum(acc[i][r]);
      if (lane == 0 && k0 + i < K && m0 + r < M) dx[(long)(m0 + r) * K + k0 + i] = f2bf(t);
    }
}

// dw[k][c] += sum_m x[m][k] dy[m][c]; lanes over c, KB weight rows per thread, the batch axis split over the 4 waves;
// block row 0 also owns dbias
template <int KB>
__global__ __launch_bounds__(256) void linear_bwd_weight_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                float* __restrict__ dw, float* __restrict__ dbias, int M, int K, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + lane;
  const int cc = c < C ? c : C - 1;
  const int k0 = blockIdx.y * KB;
  __shared__ float xs[256][KB];
  __shared__ float red[4][KB + 1][64];
  float acc[KB], bs = 0.f;
#pragma unroll
  for (int r = 0; r < KB; r++) acc[r] = 0.f;
  for (int mc = 0; mc < M; mc += 256) {
    __syncthreads();
    {
      const int m = mc + threadIdx.x;
#pragma unroll
      for (int r = 0; r < KB; r++) xs[threadIdx.x][r] = (m < M && k0 + r < K && dw) ? bf2f(x[(long)m * K + k0 + r]) : 0.f;
    }
    __syncthreads();
    const int mw0 = wv * 64;
#pragma unroll
    for (int jb = 0; jb < 64; jb += 16) {
      if (mc + mw0 + jb >= M) break;
      float g[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int m = mc + mw0 + jb + u;
        g[u] = bf2f(dy[(long)min(m, M - 1) * C + cc]);
        if (m >= M) g[u] = 0.f;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) {
        bs += g[u];
#pragma unroll
        for (int r = 0; r < KB; r++) acc[r] += xs[mw0 + jb + u][r] * g[u];
      }
    }
  }
#pragma unroll
  for (int r = 0; r < KB; r++) red[wv][r][lane] = acc[r];
  red[wv][KB][lane] = bs;
  __syncthreads();
  if (wv == 0 && c < C) {
    if (dw) {
#pragma unroll
      for (int r = 0; r < KB; r++)
        if (k0 + r < K) dw[(long)(k0 + r) * C + c] += red[0][r][lane] + red[1][r][lane] + red[2][r][lane] + red[3][r][lane];
    }
    if (dbias && blockIdx.y == 0) dbias[c] += red[0][KB][lane] + red[1][KB][lane] + red[2][KB][lane] + red[3][KB][lane];
  }
}

// narrow C: dw[k][c] += sum_m x[m][k] dy[m][c] with lanes over k (x rows are read coalesced), batch axis over 4 waves
__global__ __launch_bounds__(256) void linear_bwd_weight_narrow_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy,
                                                                       float* __restrict__ dw, float* __restrict__ dbias, int M, int K, int C) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int k = blockIdx.x * 64 + lane;
  const int kk = k < K ? k : K - 1;
  const int c = blockIdx.y;
  const int mq = (M + 3) / 4, m0 = wv * mq, m1 = min(M, m0 + mq);
  float s = 0.f, bs = 0.f;
  for (int mc = m0; mc < m1; mc += 64) {
    const float gv = (mc + lane < m1) ? bf2f(dy[(long)(mc + lane) * C + c]) : 0.f;
#pragma unroll
    for (int jb = 0; jb < 64; jb += 16) {
      if (mc + jb >= m1) break;
      float xr[16];
#pragma unroll
      for (int u = 0; u < 16; u++) xr[u] = dw ? bf2f(x[(long)min(mc + jb + u, M - 1) * K + kk]) : 0.f;
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const float g = lane_bcast(gv, jb + u);     // 0 past m1
        bs += g;
        s += xr[u] * g;
      }
    }
  }
  __shared__ float red[4][2][64];
  red[wv][0][lane] = s;
  red[wv][1][lane] = bs;
  __syncthreads();
  if (wv == 0) {
    if (dw && k < K) dw[(long)k * C + c] += red[0][0][lane] + red[1][0][lane] + red[2][0][lane] + red[3][0][lane];
    if (dbias && blockIdx.x == 0 && lane == 0) dbias[c] += red[0][1][0] + red[1][1][0] + red[2][1][0] + red[3][1][0];
  }
}

extern "C" int gank_linear_fwd(const void* x, const float* w, const float* bias, void* y, int M, int K, int C, void* stream) {
  GANK_REQUIRE(x && w && y && M > 0 && K > 0 && C > 0, "linear_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (C >= 64)
    hipLaunchKernelGGL((linear_fwd_wide_kernel<2, bf16>), dim3(cdiv(C, 64), cdiv(M, 2)), dim3(256), 0, s, (const bf16*)x, w, bias, (bf16*)y, M, K, C);
  else
    hipLaunchKernelGGL(linear_fwd_narrow_kernel, dim3(cdiv(M * C, 4)), dim3(256), 0, s, (const bf16*)x, w, bias, (bf16*)y, M, K, C);
  GANK_LAUNCH_OK("linear_fwd");
  return 0;
}

extern "C" int gank_linear_fwd_f32out(const void* x, const float* w, const float* bias, float* y, int M, int K, int C, void* stream) {
  GANK_REQUIRE(x && w && y && M > 0 && K > 0 && C > 0, "linear_fwd_f32out: bad arguments");
  hipLaunchKernelGGL((linear_fwd_wide_kernel<2, float>), dim3(cdiv(C, 64), cdiv(M, 2)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, w, bias, y, M, K, C);
  GANK_LAUNCH_OK("linear_fwd_f32out");
  return 0;
}

extern "C" int gank_linear_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, int M, int K,
                               int C, void* stream) {
  GANK_REQUIRE(dy && M > 0 && K > 0 && C > 0, "linear_bwd: bad arguments");
  GANK_REQUIRE(!dx || w, "linear_bwd: dx needs w");
  GANK_REQUIRE(!dw || x, "linear_bwd: dw needs x");
  hipStream_t s = (hipStream_t)stream;
  if (dx)
    hipLaunchKernelGGL((linear_bwd_data_kernel<4, 4>), dim3(cdiv(K, 16), cdiv(M, 4)), dim3(256), 0, s, (const bf16*)dy, w, (bf16*)dx, M, K, C);
  if (dw || dbias) {
    const int Kw = dw ? K : 1;     // dbias alone: one block row, x is never dereferenced
    if (C >= 64)
      hipLaunchKernelGGL(linear_bwd_weight_kernel<4>, dim3(cdiv(C, 64), cdiv(Kw, 4)), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, dw, dbias, M, Kw, C);
    else
      hipLaunchKernelGGL(linear_bwd_weight_narrow_kernel, dim3(cdiv(Kw, 64), C), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, dw, dbias, M, Kw, C);
  }
  GANK_LAUNCH_OK("linear_bwd");
  return 0;
}
