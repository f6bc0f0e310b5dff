// Spectral normalisation (common/ops/sn.py:15-69): one power-iteration step, sigma, W/sigma, and the
// FULL backward through the iteration (sn.py has no stop_gradient), batched over up to 16 weights per
// launch.  The descriptor table travels BY VALUE in the kernel arguments (no device table, no H2D
// copy), so the call is hipGraph-capturable with per-capture pointers.
//
//   a = W u, n=|a|, v = a/(n+eps);  b = W^T v, m=|b|, u' = b/(m+eps);  sigma = b . u'  (= v W u'^T)
//   W_bar = W / sigma
//   dL/dW = G/sigma - (<G,W>/sigma^2) (s v b^T + g_a u^T),  s=(m+2eps)/(m+eps)^2, g_v = s W b,
//   g_a = g_v/(n+eps) - a (a.g_v)/(n (n+eps)^2)
// These are latency-bound wavefront reductions (1.7 M weights in total): one wave per weight row
// for the row dots, 64-row chunks for the column sums, one block per weight for the norms.
#include "gank_common.h"

#define SN_MAX 16
#define SN_EPS 1e-12f

struct SnTable {
  gank_sn_desc d[SN_MAX];
  int count;
};

// The table sits in the kernel arguments: a counted loop over it is one dependent scalar load per entry (12 of them in front
// of every wave's first vector load).  Unrolled over SN_MAX with the offsets increasing, the index is a count of entries at or
// below `row`, and the scalar loads issue back to back.
__device__ __forceinline__ int sn_find_row(const SnTable& t, int row, int& local) {
  int w = 0;
#pragma unroll
  for (int i = 1; i < SN_MAX; i++) w += (i < t.count && row >= t.d[i].row_offset) ? 1 : 0;
  local = row - t.d[w].row_offset;
  return w;
}
__device__ __forceinline__ int sn_find_chunk(const SnTable& t, int chunk, int& local) {
  int w = 0;
#pragma unroll
  for (int i = 1; i < SN_MAX; i++) w += (i < t.count && chunk >= t.d[i].chunk_offset) ? 1 : 0;
  local = chunk - t.d[w].chunk_offset;
  return w;
}

// k1: a[k] = sum_c W[k,c] u[c]           (one wave per row)
__global__ void sn_rowdot_u_kernel(SnTable t, int total_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int k;
  const int wi = sn_find_row(t, row, k);
  const gank_sn_desc& d = t.d[wi];
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int c = lane; c < d.C; c += 64) s += d.W[(long)k * d.C + c] * d.u_in[c];
  s = wave_sum(s);
  if (lane == 0) d.a[k] = s;
  if (k == 0 && d.u_snap)       // the weight's first row also keeps u_in for the backward pass
    for (int c = lane; c < d.C; c += 64) d.u_snap[c] = d.u_in[c];
}

// k2: n = |a|; v = a/(n+eps) for this chunk's rows; bpart[chunk][c] = sum_{k in chunk} W[k,c] v[k]
__global__ void sn_colpart_kernel(SnTable t) {
  int ch;
  const int wi = sn_find_chunk(t, blockIdx.x, ch);
  const gank_sn_desc& d = t.d[wi];
  __shared__ float red[16];
  __shared__ float vs[64];
  float ss = 0.f;
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) { const float x = d.a[k]; ss += x * x; }
  const float n = sqrtf(block_sum(ss, red));
  const int k0 = ch * 64;
  const int kn = min(64, d.K - k0);
  if (threadIdx.x < 64) {
    float v = 0.f;
    if ((int)threadIdx.x < kn) { v = d.a[k0 + threadIdx.x] / (n + SN_EPS); d.v[k0 + threadIdx.x] = v; }
    vs[threadIdx.x] = v;
  }
  if (ch == 0 && threadIdx.x == 0) d.scal[1] = n;
  __syncthreads();
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    float s = 0.f;
    // 64 rows in batches of 16 independent loads (rows past kn re-read the last row against vs = 0): a plain loop with
    // a run-time bound issued one load per L2 round trip
#pragma unroll
    for (int kb = 0; kb < 64; kb += 16) {
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; u++) w[u] = d.W[(long)(k0 + min(kb + u, kn - 1)) * d.C + c];
#pragma unroll
      for (int u = 0; u < 16; u++) s += w[u] * vs[kb + u];
    }
    d.bpart[(long)ch * d.C + c] = s;
  }
}

// k3: b = sum_chunks bpart; m=|b|; u' = b/(m+eps); sigma = b.u'; s   (one block per weight)
__global__ void sn_finalize_kernel(SnTable t) {
  const gank_sn_desc& d = t.d[blockIdx.x];
  __shared__ float red[16];
  const int nch = (d.K + 63) / 64;
  float ss = 0.f;
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    float s = 0.f;
    for (int jb = 0; jb < nch; jb += 8) {            // batches of 8 independent loads
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; u++) t[u] = d.bpart[(long)min(jb + u, nch - 1) * d.C + c];
#pragma unroll
      for (int u = 0; u < 8; u++) s += (jb + u < nch) ? t[u] : 0.f;
    }
    d.b[c] = s;
    ss += s * s;
  }
  const float m = sqrtf(block_sum(ss, red));
  float dot = 0.f;
  for (int c = threadIdx.x; c < d.C; c += blockDim.x) {
    const float bb = d.b[c];
    const float un = bb / (m + SN_EPS);
    d.u_out[c] = un;
    dot += bb * un;
  }
  const float sigma = block_sum(dot, red);
  if (threadIdx.x == 0) {
    d.scal[0] = sigma;
    d.scal[2] = m;
    d.scal[3] = (m + 2.f * SN_EPS) / ((m + SN_EPS) * (m + SN_EPS));
  }
}

// k4: W_bar = W / sigma                 (one wave per row)
__global__ void sn_scale_kernel(SnTable t, int total_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int k;
  const int wi = sn_find_row(t, row, k);
  const gank_sn_desc& d = t.d[wi];
  const float sigma = d.scal[0];
  for (int c = threadIdx.x & 63; c < d.C; c += 64) d.W_bar[(long)k * d.C + c] = d.W[(long)k * d.C + c] / sigma;
}

// b1: rowdot[k] = sum_c G[k,c] W[k,c];  gv[k] = s * sum_c W[k,c] b[c]   (gv stored in ga)
__global__ void sn_bwd_rows_kernel(SnTable t, int total_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int k;
  const int wi = sn_find_row(t, row, k);
  const gank_sn_desc& d = t.d[wi];
  const int lane = threadIdx.x & 63;
  float gw = 0.f, wb = 0.f;
  for (int c = lane; c < d.C; c += 64) {
    const float w = d.W[(long)k * d.C + c];
    gw += d.dW_bar[(long)k * d.C + c] * w;
    wb += w * d.b[c];
  }
  gw = wave_sum(gw);
  wb = wave_sum(wb);
  if (lane == 0) { d.rowdot[k] = gw; d.ga[k] = d.scal[3] * wb; }
}

// b2: GW = sum rowdot; agv = a.gv; ga = gv/(n+eps) - a agv/(n (n+eps)^2)   (one block per weight)
__global__ void sn_bwd_scalars_kernel(SnTable t) {
  const gank_sn_desc& d = t.d[blockIdx.x];
  __shared__ float red[16];
  float s1 = 0.f, s2 = 0.f;
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) { s1 += d.rowdot[k]; s2 += d.a[k] * d.ga[k]; }
  const float GW = block_sum(s1, red);
  const float agv = block_sum(s2, red);
  const float n = d.scal[1];
  const float c1 = 1.f / (n + SN_EPS), c2 = agv / (n * (n + SN_EPS) * (n + SN_EPS));
  for (int k = threadIdx.x; k < d.K; k += blockDim.x) d.ga[k] = d.ga[k] * c1 - d.a[k] * c2;
  if (threadIdx.x == 0) { d.scal[4] = GW; d.scal[5] = agv; }
}

// b3: dW += G/sigma - (GW/sigma^2) (s v[k] b[c] + ga[k] u[c])   (one wave per row)
__global__ void sn_bwd_apply_kernel(SnTable t, int total_rows) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= total_rows) return;
  int k;
  const int wi = sn_find_row(t, row, k);
  const gank_sn_desc& d = t.d[wi];
  const float sigma = d.scal[0], s = d.scal[3];
  const float coef = d.scal[4] / (sigma * sigma);
  const float sv = s * d.v[k], gak = d.ga[k];
  const float* uin = d.u_snap ? d.u_snap : d.u_in;
  for (int c = threadIdx.x & 63; c < d.C; c += 64) {
    const long i = (long)k * d.C + c;
    d.dW[i] += d.dW_bar[i] / sigma - coef * (sv * d.b[c] + gak * uin[c]);
  }
}

static int sn_fill(SnTable& t, const gank_sn_desc* table, int count, int& rows, int& chunks, bool bwd) {
  rows = 0; chunks = 0;
  t.count = count;
  for (int i = 0; i < count; i++) {
    t.d[i] = table[i];
    GANK_REQUIRE(t.d[i].K > 0 && t.d[i].C > 0, "sn: weight %d has bad shape", i);
    GANK_REQUIRE(t.d[i].W && t.d[i].u_in && t.d[i].a && t.d[i].b && t.d[i].v && t.d[i].scal, "sn: weight %d has null pointers", i);
    if (bwd) GANK_REQUIRE(t.d[i].dW_bar && t.d[i].dW && t.d[i].rowdot && t.d[i].ga, "sn bwd: weight %d has null pointers", i);
    else GANK_REQUIRE(t.d[i].u_out && t.d[i].W_bar && t.d[i].bpart, "sn fwd: weight %d has null pointers", i);
    t.d[i].row_offset = rows;
    t.d[i].chunk_offset = chunks;
    rows += t.d[i].K;
    chunks += (t.d[i].K + 63) / 64;
  }
  return 0;
}

extern "C" int gank_sn_power_iter_fwd(const gank_sn_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0, "sn fwd: empty table");
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < count; base += SN_MAX) {
    SnTable t;
    int rows, chunks;
    const int n = count - base < SN_MAX ? count - base : SN_MAX;
    if (sn_fill(t, table + base, n, rows, chunks, false)) return 1;
    hipLaunchKernelGGL(sn_rowdot_u_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, t, rows);
    hipLaunchKernelGGL(sn_colpart_kernel, dim3(chunks), dim3(256), 0, s, t);
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(n), dim3(256), 0, s, t);
    hipLaunchKernelGGL(sn_scale_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, t, rows);
    GANK_LAUNCH_OK("sn_power_iter_fwd");
  }
  return 0;
}

extern "C" int gank_sn_power_iter_bwd(const gank_sn_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0, "sn bwd: empty table");
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < count; base += SN_MAX) {
    SnTable t;
    int rows, chunks;
    const int n = count - base < SN_MAX ? count - base : SN_MAX;
    if (sn_fill(t, table + base, n, rows, chunks, true)) return 1;
    hipLaunchKernelGGL(sn_bwd_rows_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, t, rows);
    hipLaunchKernelGGL(sn_bwd_scalars_kernel, dim3(n), dim3(256), 0, s, t);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(cdiv(rows, 4)), dim3(256), 0, s, t, rows);
    GANK_LAUNCH_OK("sn_power_iter_bwd");
  }
  return 0;
}
