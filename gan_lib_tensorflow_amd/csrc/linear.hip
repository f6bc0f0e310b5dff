// Small dense layers (common/ops/linear.py:161-180: tf.matmul + bias_add) on fp32 master weights.
// The SNGAN critic has two of them per forward -- D.Embedding_y [128,300]x[300,128] and D.Output [128,128]x[128,1]
// (gan_cifar_resnet.py:296-304) -- 10 MFLOP and 33 kFLOP: launch-latency problems, not MFMA problems.  Through the
// implicit-GEMM engine each cost 16-38 us (bf16 operand preparation, K-packed gather with per-element index
// arithmetic, 2-4 workgroups on 256 CUs).  Here: no operand preparation, weights read as stored, one thread per
// output, the reduction axis pipelined by unrolling; every call is a few microseconds.
// (The generator's 128 -> 16384 input layer stays on the MFMA engine.)
#include "gank_common.h"

// y[m][c] = sum_k x[m][k] w[k][c] + b[c];  lanes over c (coalesced w, y), RB rows per thread to reuse each w load
template <int RB>
__global__ void linear_fwd_wide_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                       bf16* __restrict__ y, int M, int K, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int m0 = blockIdx.y * RB;
  if (c >= C) return;
  float acc[RB];
#pragma unroll
  for (int r = 0; r < RB; r++) acc[r] = 0.f;
#pragma unroll 4
  for (int k = 0; k < K; k++) {
    const float wv = w[(long)k * C + c];
#pragma unroll
    for (int r = 0; r < RB; r++) {
      const int m = m0 + r < M ? m0 + r : M - 1;
      acc[r] += bf2f(x[(long)m * K + k]) * wv;      // wave-uniform address: scalar/broadcast load
    }
  }
  const float bv = b ? b[c] : 0.f;
#pragma unroll
  for (int r = 0; r < RB; r++)
    if (m0 + r < M) y[(long)(m0 + r) * C + c] = f2bf(acc[r] + bv);
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

// dx[m][k] = sum_c dy[m][c] w[k][c];  lanes over k (coalesced dx), each lane walks its own weight row
__global__ void linear_bwd_data_kernel(const bf16* __restrict__ dy, const float* __restrict__ w, bf16* __restrict__ dx,
                                       int M, int K, int C) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int m = blockIdx.y;
  if (k >= K) return;
  float s = 0.f;
#pragma unroll 4
  for (int c = 0; c < C; c++) s += bf2f(dy[(long)m * C + c]) * w[(long)k * C + c];
  dx[(long)m * K + k] = f2bf(s);
}

// dw[k][c] += sum_m x[m][k] dy[m][c]; lanes over c, KB weight rows per thread; block row 0 also owns dbias
template <int KB>
__global__ void linear_bwd_weight_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw,
                                         float* __restrict__ dbias, int M, int K, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int k0 = blockIdx.y * KB;
  if (c >= C) return;
  float acc[KB], bs = 0.f;
#pragma unroll
  for (int r = 0; r < KB; r++) acc[r] = 0.f;
#pragma unroll 4
  for (int m = 0; m < M; m++) {
    const float g = bf2f(dy[(long)m * C + c]);
    bs += g;
#pragma unroll
    for (int r = 0; r < KB; r++) {
      const int k = k0 + r < K ? k0 + r : K - 1;
      acc[r] += bf2f(x[(long)m * K + k]) * g;
    }
  }
  if (dw) {
#pragma unroll
    for (int r = 0; r < KB; r++)
      if (k0 + r < K) dw[(long)(k0 + r) * C + c] += acc[r];
  }
  if (dbias && blockIdx.y == 0) dbias[c] += bs;
}

// narrow C: dw[k][c] += sum_m x[m][k] dy[m][c] with lanes over k (x rows are read coalesced)
__global__ void linear_bwd_weight_narrow_kernel(const bf16* __restrict__ x, const bf16* __restrict__ dy, float* __restrict__ dw,
                                                float* __restrict__ dbias, int M, int K, int C) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (k >= K) return;
  float s = 0.f, bs = 0.f;
#pragma unroll 4
  for (int m = 0; m < M; m++) {
    const float g = bf2f(dy[(long)m * C + c]);
    bs += g;
    s += bf2f(x[(long)m * K + k]) * g;
  }
  if (dw) dw[(long)k * C + c] += s;
  if (dbias && k == 0) dbias[c] += bs;
}

extern "C" int gank_linear_fwd(const void* x, const float* w, const float* bias, void* y, int M, int K, int C, void* stream) {
  GANK_REQUIRE(x && w && y && M > 0 && K > 0 && C > 0, "linear_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  if (C >= 64)
    hipLaunchKernelGGL(linear_fwd_wide_kernel<4>, dim3(cdiv(C, 64), cdiv(M, 4)), dim3(64), 0, s, (const bf16*)x, w, bias, (bf16*)y, M, K, C);
  else
    hipLaunchKernelGGL(linear_fwd_narrow_kernel, dim3(cdiv(M * C, 4)), dim3(256), 0, s, (const bf16*)x, w, bias, (bf16*)y, M, K, C);
  GANK_LAUNCH_OK("linear_fwd");
  return 0;
}

extern "C" int gank_linear_bwd(const void* dy, const void* x, const float* w, void* dx, float* dw, float* dbias, int M, int K,
                               int C, void* stream) {
  GANK_REQUIRE(dy && M > 0 && K > 0 && C > 0, "linear_bwd: bad arguments");
  GANK_REQUIRE(!dx || w, "linear_bwd: dx needs w");
  GANK_REQUIRE(!dw || x, "linear_bwd: dw needs x");
  hipStream_t s = (hipStream_t)stream;
  if (dx) hipLaunchKernelGGL(linear_bwd_data_kernel, dim3(cdiv(K, 64), M), dim3(64), 0, s, (const bf16*)dy, w, (bf16*)dx, M, K, C);
  if (dw || dbias) {
    const bf16* xx = dw ? (const bf16*)x : (const bf16*)dy;   // dbias only: x is not dereferenced meaningfully (dw == null)
    if (C >= 64)
      hipLaunchKernelGGL(linear_bwd_weight_kernel<4>, dim3(cdiv(C, 64), dw ? cdiv(K, 4) : 1), dim3(64), 0, s, xx, (const bf16*)dy, dw, dbias, M, dw ? K : 1, C);
    else
      hipLaunchKernelGGL(linear_bwd_weight_narrow_kernel, dim3(dw ? cdiv(K, 64) : 1, C), dim3(64), 0, s, xx, (const bf16*)dy, dw, dbias, M, dw ? K : 1, C);
  }
  GANK_LAUNCH_OK("linear_bwd");
  return 0;
}
