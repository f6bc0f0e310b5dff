// Spectral normalisation (common/ops/sn.py:15-69): one power-iteration step, sigma, W/sigma, and the
// FULL backward through the iteration (sn.py has no stop_gradient), batched over up to 16 weights per
// launch.  The descriptor table travels BY VALUE in the kernel arguments (no device table, no H2D
// copy), so the call is hipGraph-capturable with per-capture pointers.
//
//   a = W u, n=|a|, v = a/(n+eps);  b = W^T v, m=|b|, u' = b/(m+eps);  sigma = b . u'  (= v W u'^T)
//   W_bar = W / sigma
//   dL/dW = G/sigma - (<G,W>/sigma^2) (s v b^T + g_a u^T),  s=(m+2eps)/(m+eps)^2, g_v = s W b,
//   g_a = g_v/(n+eps) - a (a.g_v)/(n (n+eps)^2)
//
// Round 3: TWO launches each way instead of 4 + 3 (+ 1 for the MFMA operand copies).  These are latency-bound
// reductions over 1.7 M weights: every launch is a dependent 5-9-us step of the critic update, which runs 5x per
// iteration, so the structure is chosen for the fewest dependent passes, not for bandwidth.
//   forward A: a block owns a 64-row chunk of one weight and keeps it IN REGISTERS: the row dots a = W u, then the
//              chunk's share of the UN-normalised column sums W^T a (b = W^T a / (n+eps): n is not known yet) and of
//              |a|^2 -- one pass over W instead of two.  The chunk blocks of a weight draw tickets; the block that
//              draws the last one reduces the partial sums (n, b, m, u', sigma, s): no finalize launch.
//   forward B: W_bar = W / sigma, the backward pass's per-row terms (v, and g_a in closed form: a.g_v = s (n+eps) m^2
//              because W^T a = (n+eps) b), AND the bf16 MFMA operand copies of W / sigma (prep_weights.h: the blocks of
//              gank_conv2d_prep_weights_batched with the division folded into their loads), AND optionally the
//              per-label table of a small dense layer on an embedding (the critic's label branch) -- what took a
//              scale launch, a preparation launch, an embedding launch and a dense-layer launch.
//   backward 1: per-chunk partial sums of <G, W>.
//   backward 2: every block sums the (<= a few dozen) partials of its weight itself, then applies the gradient.
#include "gank_common.h"
#include "prep_weights.h"
#include "feed.h"

#define SN_MAX 16
#define SN_EPS 1e-12f
// rows per block: forward A keeps a chunk of SN_ROWS rows in registers and leaves one set of partial sums per chunk;
// forward B and the backward launches are element-wise over the weight (plus a row dot) and run on finer pieces of SN_FINE
// rows -- 4x the blocks in flight for the same bytes (175 blocks of 64 rows left most CUs with one latency chain each)
#ifndef SN_ROWS
#define SN_ROWS 32
#endif
#ifndef SN_FINE
#define SN_FINE 16
#endif

struct SnTable {
  gank_sn_desc d[SN_MAX];
  int count;
};

// Tickets of forward A: one word per table entry, zero between launches (the last arriver of a weight resets its word).
// The slots rotate per host call, so launches that overlap on different streams do not share words unless more than
// SN_TICKET_GROUPS of them are in flight.
#define SN_TICKET_GROUPS 32
__device__ unsigned sn_tickets[SN_TICKET_GROUPS * SN_MAX];
static std::atomic<unsigned> sn_ticket_group{0};

// The table sits in the kernel arguments: a counted loop over it is one dependent scalar load per entry.  Unrolled over
// SN_MAX with the offsets increasing, the index is a count of entries at or below `chunk`, and the loads issue back to back.
__device__ __forceinline__ int sn_find_chunk(const SnTable& t, int chunk, int& local) {
  int w = 0;
#pragma unroll
  for (int i = 1; i < SN_MAX; i++) w += (i < t.count && chunk >= t.d[i].chunk_offset) ? 1 : 0;
  local = chunk - t.d[w].chunk_offset;
  return w;
}

__device__ __forceinline__ int sn_find_fine(const SnTable& t, int piece, int& local) {
  int w = 0;
#pragma unroll
  for (int i = 1; i < SN_MAX; i++) w += (i < t.count && piece >= t.d[i].row_offset) ? 1 : 0;      // row_offset: first fine piece of the weight
  local = piece - t.d[w].row_offset;
  return w;
}

// Register-tile form: C a power of two in [4, 256] -- a chunk is 64*C contiguous floats, thread t takes the f32x4 pieces
// (i*256 + t) of it, so its 4 columns are the same for every piece ((4t) mod C) and a row is C/4 consecutive lanes.
__device__ __forceinline__ bool sn_pow2_c(int C) { return C >= 4 && C <= 256 && (C & (C - 1)) == 0; }
__device__ __forceinline__ bool sn_al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// write-through (sc1) store: the value is in memory, visible to every XCD, once the storing wave's vmcnt has drained
__device__ __forceinline__ void sn_store_wt(float* p, float x) {
  __hip_atomic_store(p, x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// sum over the G = C/4 lanes that hold one row (all of them get the result)
__device__ __forceinline__ float sn_row_sum(float p, int G) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1)
    if (o < G) p += __shfl_xor(p, o, 64);
  return p;
}

// The last chunk block of a weight (ticket) finishes the power iteration from the published partial sums: n = |a|, b = W^T a / (n + eps),
// m = |b|, u' -> u_out, sigma and the scalars of the backward pass.  Shared by forward A and the fused optimiser tail below, whose
// u' goes to a staging buffer instead of over u.
__device__ __forceinline__ void sn_fwd_a_finish(const gank_sn_desc& d, float* u_out, int nch, float* part, float* red, int tid) {
  const int C = d.C;
  const float* n2p = d.bpart + (long)nch * C;
  float s = 0.f;
  for (int j = tid; j < nch; j += 256) s += n2p[j];
  const float n = sqrtf(block_sum(s, red));
  float ss = 0.f;
  if (sn_pow2_c(C) && sn_al16(d.bpart)) {
    // all 256 threads: thread t sums the f32x4 piece (t mod C/4) of the chunks j = t / (C/4), + 1024/C, ... -- every load of
    // the block in flight at once (up to 36 chunks x 1 KB: one L2 round trip instead of five); then the column-group
    // reduction of the main body
    const int G = C >> 2, lc = __builtin_ctz(C), reps = 1024 >> lc;
    const int g = tid & (G - 1), j0 = tid >> (lc - 2);
    f32x4 acc4 = {0.f, 0.f, 0.f, 0.f};
    for (int jb = j0; jb < nch; jb += 16 * reps) {
      f32x4 tt[16];
#pragma unroll
      for (int u = 0; u < 16; u++) {
        const int j = jb + u * reps;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        tt[u] = j < nch ? *reinterpret_cast<const f32x4*>(d.bpart + (long)j * C + 4 * g) : z;
      }
#pragma unroll
      for (int u = 0; u < 16; u++) acc4 += tt[u];
    }
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; e++) part[tid * 4 + e] = acc4[e];
    __syncthreads();
    if (tid < C) {
      const int gg = tid >> 2, e = tid & 3;
      float su = 0.f;
      for (int j = 0; j < reps; j++) su += part[(j * G + gg) * 4 + e];
      const float bb = su / (n + SN_EPS);
      d.b[tid] = bb;
      ss = bb * bb;
    }
  } else {
    for (int c = tid; c < C; c += 256) {
      float su = 0.f;
      for (int jb = 0; jb < nch; jb += 8) {            // batches of 8 independent loads
        float tt[8];
#pragma unroll
        for (int u = 0; u < 8; u++) tt[u] = d.bpart[(long)min(jb + u, nch - 1) * C + c];
#pragma unroll
        for (int u = 0; u < 8; u++) su += (jb + u < nch) ? tt[u] : 0.f;
      }
      const float bb = su / (n + SN_EPS);
      d.b[c] = bb;
      ss += bb * bb;
    }
  }
  const float m2 = block_sum(ss, red);
  const float m = sqrtf(m2);
  float dot = 0.f;
  for (int c = tid; c < C; c += 256) {
    const float bb = d.b[c];                  // this thread's own store
    const float un = bb / (m + SN_EPS);
    u_out[c] = un;
    dot += bb * un;
  }
  const float sigma = block_sum(dot, red);
  if (tid == 0) {
    const float sc = (m + 2.f * SN_EPS) / ((m + SN_EPS) * (m + SN_EPS));
    d.scal[0] = sigma;
    d.scal[1] = n;
    d.scal[2] = m;
    d.scal[3] = sc;
    d.scal[5] = sc * (n + SN_EPS) * m2;       // a . g_v = s a^T W b = s (W^T a) . b = s (n+eps) |b|^2
  }
}

// ---------------------------------------------------------------------------------------------- forward A
__global__ __launch_bounds__(256) void sn_fwd_a_kernel(SnTable t, unsigned* __restrict__ tickets) {
  __shared__ float sm[1024 + SN_ROWS + 16 + 4];
  float* part = sm;
  float* as = sm + 1024;
  float* red = sm + 1024 + SN_ROWS;
  int* flag = reinterpret_cast<int*>(sm + 1024 + SN_ROWS + 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int ch;
  const int wi = sn_find_chunk(t, blockIdx.x, ch);
  const gank_sn_desc& d = t.d[wi];
  const int K = d.K, C = d.C, k0 = ch * SN_ROWS, kn = min(SN_ROWS, K - k0), nch = (K + SN_ROWS - 1) / SN_ROWS;
  const float* __restrict__ W = d.W;
  float* bp = d.bpart + (long)ch * C;
  if (ch == 0 && d.u_snap)       // the weight's first chunk also keeps u_in for the backward pass (u_out may alias u_in)
    for (int c = tid; c < C; c += 256) d.u_snap[c] = d.u_in[c];
  float n2 = 0.f;
  if (sn_pow2_c(C) && sn_al16(W)) {
    const int lc = __builtin_ctz(C), G = C >> 2;
    const int L = max(1, (SN_ROWS * C) >> 10);
    const int col = (tid * 4) & (C - 1);
    float u4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) u4[e] = d.u_in[col + e];
    f32x4 w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (i < L) {
        const int flat = (i * 256 + tid) * 4, row = flat >> lc;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        w[i] = row < kn ? *reinterpret_cast<const f32x4*>(W + (long)k0 * C + flat) : z;
      }
    }
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 16; i++) {
      if (i < L) {
        const int row = ((i * 256 + tid) * 4) >> lc;
        float p = ((w[i][0] * u4[0] + w[i][1] * u4[1]) + w[i][2] * u4[2]) + w[i][3] * u4[3];
        p = sn_row_sum(p, G);
        if (col == 0 && row < kn) { d.a[k0 + row] = p; n2 += p * p; }
#pragma unroll
        for (int e = 0; e < 4; e++) s4[e] += p * w[i][e];      // rows past kn hold zeros
      }
    }
#pragma unroll
    for (int e = 0; e < 4; e++) part[tid * 4 + e] = s4[e];
    __syncthreads();
    if (tid < C) {
      const int g = tid >> 2, e = tid & 3, reps = 1024 >> lc;       // threads per column group
      float s = 0.f;
      for (int j = 0; j < reps; j++) s += part[(j * G + g) * 4 + e];
      sn_store_wt(bp + tid, s);
    }
  } else {
    for (int r = wave; r < kn; r += 4) {
      float s = 0.f;
      for (int c = lane; c < C; c += 64) s += W[(long)(k0 + r) * C + c] * d.u_in[c];
      s = wave_sum(s);
      if (lane == 0) { as[r] = s; d.a[k0 + r] = s; n2 += s * s; }
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
      float s = 0.f;
      for (int r = 0; r < kn; r++) s += as[r] * W[(long)(k0 + r) * C + c];
      sn_store_wt(bp + c, s);
    }
  }
  n2 = block_sum(n2, red);
  if (tid == 0) sn_store_wt(d.bpart + (long)nch * C + ch, n2);

  // ---- publish this chunk's partial sums, draw a ticket; the last chunk of the weight finishes the iteration ----
  // The partial sums another block reads were stored WRITE-THROUGH (sc1: sn_store_wt), so no release fence (an L2 write-back,
  // 1.7-6.5 us on the critical path of every block) is needed: every storing wave drains its stores, the workgroup meets,
  // one lane draws the ticket; the last arriver takes ONE agent-scope acquire and reads with plain loads.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave
  __syncthreads();
  if (tid == 0) {
    int last = 1;
    if (nch > 1) {
      const unsigned tk = __hip_atomic_fetch_add(tickets + wi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      last = tk == (unsigned)(nch - 1);
      if (last) {
        __hip_atomic_store(tickets + wi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // zero again for the next launch
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    *flag = last;
  }
  __syncthreads();
  if (!*flag) return;

  sn_fwd_a_finish(d, d.u_out, nch, part, red, tid);
}

// ---------------------------------------------------------------------------------------------- forward B
struct SnScaleEntry {
  const float* W;
  float* W_bar;
  const float* scal;
  const float* a;
  const float* b;
  float* v;
  float* ga;
  int K, C, chunk_offset, pad_;
};
struct SnScaleTable {
  SnScaleEntry d[SN_MAX];
  int count, nchunks;
};
struct SnPrepExtra {
  const float* sigma[PREP_MAX];     // divisor of prep entry i (the scal word of its weight)
  int prep_blocks;
  int dbg;                          // TUNING builds (GANK_SNB_DBG), timing only: 1 = scale blocks return, 2 = operand blocks return, 4 = label rows return
};
// the per-label rows of a small dense layer on an embedding table: out[l] = bf16( bf16(table[l]) (W / sigma) + bias )
// Consumer side of the fused optimiser tail (below): the power iteration of this forward pass was already run when the weights
// were updated, its u' waits in a staging buffer.  u.assign(u_final) (sn.py:55-56) then is a flat copy over the concatenated u
// vectors of the table, with the snapshot the backward pass reads taken first: u_snap <- u, u <- u_next (total floats).
struct SnAdopt {
  float* u;
  float* u_snap;
  const float* u_next;
  int total, blocks;
};
struct SnLabelDense {
  const float* table;   // [V, D] fp32
  const float* W;       // [D, Cout] fp32 master weight
  const float* sigma;   // its spectral norm (scal word), or NULL
  const float* bias;    // [Cout] or NULL
  bf16* out;            // [V, Cout]
  int V, D, Cout;
};

__device__ __forceinline__ int sn_find_chunk_b(const SnScaleTable& t, int chunk, int& local) {
  int w = 0;
#pragma unroll
  for (int i = 1; i < SN_MAX; i++) w += (i < t.count && chunk >= t.d[i].chunk_offset) ? 1 : 0;
  local = chunk - t.d[w].chunk_offset;
  return w;
}

// One (label row, 64-column group) of the dense layer.  The arithmetic is linear_fwd_wide_kernel's (linear.hip), order of
// additions included: the reduction axis in 4 contiguous quarters, one per wave, each summed in ascending k, the quarters
// added 0..3, then the bias -- with sigma = NULL (W already normalised) the row a sample's embedding used to get from
// embedding_fwd + linear_fwd is reproduced bit for bit; with a sigma the sum over the MASTER weight is divided once (one
// division per output instead of one per weight: 96 divisions per lane were 2-4 us of this block's chain).  Loads as there: the wave's x values by ONE lane-indexed load per 64 k (broadcast with v_readlane), the weights in
// batches of 16 independent loads (a plain k loop is one L2 round trip per step: 150 of them took 50 us).
__device__ __forceinline__ float sn_lane_bcast(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ void sn_label_row(const SnLabelDense& q, int l, int cg, float* red /* [4][64] */) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float sg = q.sigma ? q.sigma[0] : 1.f;
  const int K = q.D, C = q.Cout;
  const int kq = (K + 3) / 4, k0 = wv * kq, k1 = min(K, k0 + kq);
  const int c = cg * 64 + lane, cc = c < C ? c : C - 1;
  float acc = 0.f;
  if (kq <= 96) {
    // the wave's whole quarter requested at once (<= 96 weights per lane in flight), then the additions in ascending k
    const float xa = (k0 + lane < k1) ? bf2f(f2bf(q.table[(long)l * K + k0 + lane])) : 0.f;
    const float xb = (k0 + 64 + lane < k1) ? bf2f(f2bf(q.table[(long)l * K + k0 + 64 + lane])) : 0.f;
    float wt[96];
#pragma unroll
    for (int u = 0; u < 96; u++) wt[u] = q.W[(long)min(k0 + u, K - 1) * C + cc];
#pragma unroll
    for (int u = 0; u < 96; u++) {
      // linear_fwd_wide_kernel adds a (zero) term for every step of a started batch of 16: k0 + u < ceil16(k1 - k0 - 64 [u >= 64]) ...
      const int kc = u < 64 ? k0 : k0 + 64, jb = (u & 63) & ~15;
      if (kc < k1 && kc + jb < k1) acc += sn_lane_bcast(u < 64 ? xa : xb, u & 63) * wt[u];
    }
  } else {
    for (int kc = k0; kc < k1; kc += 64) {
      const float xv = (kc + lane < k1) ? bf2f(f2bf(q.table[(long)l * K + kc + lane])) : 0.f;
#pragma unroll
      for (int jb = 0; jb < 64; jb += 16) {
        if (kc + jb >= k1) break;
        float wt[16];
#pragma unroll
        for (int u = 0; u < 16; u++) wt[u] = q.W[(long)min(kc + jb + u, K - 1) * C + cc];
#pragma unroll
        for (int u = 0; u < 16; u++) acc += sn_lane_bcast(xv, jb + u) * wt[u];
      }
    }
  }
  red[wv * 64 + lane] = acc;
  __syncthreads();
  if (wv == 0 && c < C)
    q.out[(long)l * C + c] = f2bf((red[lane] + red[64 + lane] + red[128 + lane] + red[192 + lane]) / sg + (q.bias ? q.bias[c] : 0.f));
}

__device__ __forceinline__ int ld_blocks(const SnLabelDense& ld) { return ld.table ? ld.V * ((ld.Cout + 63) >> 6) : 0; }

// fd (blocks > 0): the critic's input feed of this update (feed.h) as the last block range of the launch -- it reads and writes
// nothing the other ranges touch, and every consumer of either comes after the launch.
__global__ __launch_bounds__(256) void sn_fwd_b_kernel(SnScaleTable st, PrepTable pt, SnPrepExtra px, SnLabelDense ld, SnAdopt ad, CriticFeedArgs fd) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef GANK_TUNING
  if ((px.dbg & 1) && (int)blockIdx.x < st.nchunks) return;
  if ((px.dbg & 2) && (int)blockIdx.x >= st.nchunks && (int)blockIdx.x - st.nchunks < px.prep_blocks) return;
  if ((px.dbg & 4) && (int)blockIdx.x - st.nchunks >= px.prep_blocks && (int)blockIdx.x - st.nchunks < px.prep_blocks + ld_blocks(ld)) return;
#endif
  if ((int)blockIdx.x >= st.nchunks) {
    const int pb = blockIdx.x - st.nchunks;
    if (pb >= px.prep_blocks + ld_blocks(ld) + ad.blocks) {
      critic_feed_block(fd, pb - (px.prep_blocks + ld_blocks(ld) + ad.blocks));
      return;
    }
    if (pb >= px.prep_blocks + ld_blocks(ld)) {      // u_snap <- u, u <- u_next (nothing else in this launch reads u)
      const int i = (pb - px.prep_blocks - ld_blocks(ld)) * 256 + tid;
      if (i < ad.total) {
        const float uo = ad.u[i];
        if (ad.u_snap) ad.u_snap[i] = uo;
        ad.u[i] = ad.u_next[i];
      }
      return;
    }
    if (pb < px.prep_blocks) {            // bf16 MFMA operand copies of W / sigma
      const int e = prep_batch_entry(pt, pb);
      const float* sp = px.sigma[0];
#pragma unroll
      for (int i = 1; i < PREP_MAX; i++) sp = e == i ? px.sigma[i] : sp;
      prep_batch_block<true>(pt, e, pb, sp[0]);
    } else {
      __shared__ float lred[256];
      const int lb = pb - px.prep_blocks, ncg = (ld.Cout + 63) >> 6;
      sn_label_row(ld, lb / ncg, lb % ncg, lred);
    }
    return;
  }
  int ch;
  const int wi = sn_find_chunk_b(st, blockIdx.x, ch);
  const SnScaleEntry& d = st.d[wi];
  const int K = d.K, C = d.C, k0 = ch * SN_FINE, kn = min(SN_FINE, K - k0);
  const float sigma = d.scal[0], n = d.scal[1], sc = d.scal[3], agv = d.scal[5];
  const float c1 = 1.f / (n + SN_EPS), c2 = agv / (n * (n + SN_EPS) * (n + SN_EPS));
  const float* __restrict__ W = d.W;
  if (sn_pow2_c(C) && sn_al16(W) && sn_al16(d.W_bar)) {
    const int lc = __builtin_ctz(C), G = C >> 2;
    const int L = max(1, (SN_FINE * C) >> 10);
    const int col = (tid * 4) & (C - 1);
    float b4[4];
#pragma unroll
    for (int e = 0; e < 4; e++) b4[e] = d.b[col + e];
    constexpr int LMAX = (SN_FINE * 256) >> 10 > 0 ? (SN_FINE * 256) >> 10 : 1;
    f32x4 w[LMAX];
    float ak[LMAX];
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
      if (i < L) {
        const int flat = (i * 256 + tid) * 4, row = flat >> lc;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        w[i] = row < kn ? *reinterpret_cast<const f32x4*>(W + (long)k0 * C + flat) : z;
        ak[i] = row < kn ? d.a[k0 + row] : 0.f;       // every lane of the row, up front: inside the loop below it was 16 dependent round trips
      }
    }
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
      if (i < L) {
        const int flat = (i * 256 + tid) * 4, row = flat >> lc;
        if (row < kn) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; e++) o[e] = w[i][e] / sigma;
          *reinterpret_cast<f32x4*>(d.W_bar + (long)k0 * C + flat) = o;
        }
        float p = ((w[i][0] * b4[0] + w[i][1] * b4[1]) + w[i][2] * b4[2]) + w[i][3] * b4[3];
        p = sn_row_sum(p, G);
        if (col == 0 && row < kn) {
          d.v[k0 + row] = ak[i] / (n + SN_EPS);
          d.ga[k0 + row] = (sc * p) * c1 - ak[i] * c2;
        }
      }
    }
  } else {
    for (int r = wave; r < kn; r += 4) {
      const long ro = (long)(k0 + r) * C;
      float p = 0.f;
      for (int c = lane; c < C; c += 64) {
        const float x = W[ro + c];
        d.W_bar[ro + c] = x / sigma;
        p += x * d.b[c];
      }
      p = wave_sum(p);
      if (lane == 0) {
        const float ak = d.a[k0 + r];
        d.v[k0 + r] = ak / (n + SN_EPS);
        d.ga[k0 + r] = (sc * p) * c1 - ak * c2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- backward
// b1: gwpart[piece] = sum over the piece (SN_FINE rows) of G * W            (kept behind the forward's partials in bpart)
__global__ __launch_bounds__(256) void sn_bwd_gw_kernel(SnTable t) {
  __shared__ float red[16];
  const int tid = threadIdx.x;
  int ch;
  const int wi = sn_find_fine(t, blockIdx.x, ch);
  const gank_sn_desc& d = t.d[wi];
  const int K = d.K, C = d.C, k0 = ch * SN_FINE, kn = min(SN_FINE, K - k0), nch = (K + SN_ROWS - 1) / SN_ROWS;
  const float* __restrict__ W = d.W + (long)k0 * C;
  const float* __restrict__ G = d.dW_bar + (long)k0 * C;
  const int total = kn * C;
  float s = 0.f;
  if ((C & 3) == 0 && sn_al16(d.W) && sn_al16(d.dW_bar)) {
    constexpr int NB = 4;
    f32x4 w[NB], g[NB];
    const int n4 = total >> 2;
    for (int base = 0; base < n4; base += NB * 256) {
#pragma unroll
      for (int i = 0; i < NB; i++) {
        const int q = base + i * 256 + tid;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        w[i] = q < n4 ? reinterpret_cast<const f32x4*>(W)[q] : z;
        g[i] = q < n4 ? reinterpret_cast<const f32x4*>(G)[q] : z;
      }
#pragma unroll
      for (int i = 0; i < NB; i++) s += ((w[i][0] * g[i][0] + w[i][1] * g[i][1]) + w[i][2] * g[i][2]) + w[i][3] * g[i][3];
    }
  } else {
    for (int q = tid; q < total; q += 256) s += W[q] * G[q];
  }
  s = block_sum(s, red);
  if (tid == 0) d.bpart[(long)nch * C + nch + ch] = s;
}

// b2: <G,W> = sum of the weight's partials (every block for itself: a few dozen floats);
//     dW += G/sigma - (<G,W>/sigma^2) (s v[k] b[c] + ga[k] u[c])
__global__ __launch_bounds__(256) void sn_bwd_apply_kernel(SnTable t) {
  __shared__ float red[16];
  const int tid = threadIdx.x;
  int ch;
  const int wi = sn_find_fine(t, blockIdx.x, ch);
  const gank_sn_desc& d = t.d[wi];
  const int K = d.K, C = d.C, k0 = ch * SN_FINE, kn = min(SN_FINE, K - k0), nch = (K + SN_ROWS - 1) / SN_ROWS;
  const int nfine = (K + SN_FINE - 1) / SN_FINE;
  const float* gwp = d.bpart + (long)nch * C + nch;
  const float* uin = d.u_snap ? d.u_snap : d.u_in;
  const float* __restrict__ G = d.dW_bar + (long)k0 * C;
  float* __restrict__ dW = d.dW + (long)k0 * C;
  const bool fast = sn_pow2_c(C) && sn_al16(d.dW_bar) && sn_al16(d.dW);
  constexpr int LMAX = (SN_FINE * 256) >> 10 > 0 ? (SN_FINE * 256) >> 10 : 1;
  const int lc = fast ? __builtin_ctz(C) : 0;
  const int L = max(1, (SN_FINE * C) >> 10);
  const int col = fast ? (tid * 4) & (C - 1) : 0;
  // every load of the block is requested before the first wait: the piece of G and dW, the row terms, b and u, and the
  // <G,W> partials of the weight (each block sums them itself: a few dozen to a few hundred floats)
  f32x4 g[LMAX], o[LMAX];
  float sv[LMAX], gak[LMAX];
  float b4[4] = {0.f, 0.f, 0.f, 0.f}, u4[4] = {0.f, 0.f, 0.f, 0.f};
  if (fast) {
#pragma unroll
    for (int e = 0; e < 4; e++) { b4[e] = d.b[col + e]; u4[e] = uin[col + e]; }
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
      if (i < L) {
        const int flat = (i * 256 + tid) * 4, row = flat >> lc;
        const bool ok = row < kn;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        g[i] = ok ? *reinterpret_cast<const f32x4*>(G + flat) : z;
        o[i] = ok ? *reinterpret_cast<const f32x4*>(dW + flat) : z;
        sv[i] = ok ? d.v[k0 + row] : 0.f;
        gak[i] = ok ? d.ga[k0 + row] : 0.f;
      }
    }
  }
  float s = 0.f;
  for (int j = tid; j < nfine; j += 256) s += gwp[j];
  const float GW = block_sum(s, red);
  const float sigma = d.scal[0], sc = d.scal[3];
  const float coef = GW / (sigma * sigma);
  if (ch == 0 && tid == 0) d.scal[4] = GW;
  if (fast) {
#pragma unroll
    for (int i = 0; i < LMAX; i++) {
      if (i < L) {
        const int flat = (i * 256 + tid) * 4, row = flat >> lc;
        if (row < kn) {
#pragma unroll
          for (int e = 0; e < 4; e++) o[i][e] += g[i][e] / sigma - coef * ((sc * sv[i]) * b4[e] + gak[i] * u4[e]);
          *reinterpret_cast<f32x4*>(dW + flat) = o[i];
        }
      }
    }
  } else {
    const int total = kn * C;
    for (int q = tid; q < total; q += 256) {
      const int r = q / C, c = q - r * C;
      dW[q] += G[q] / sigma - coef * (sc * d.v[k0 + r] * d.b[c] + d.ga[k0 + r] * uin[c]);
    }
  }
}

// ---------------------------------------------------------------------------------------------- host
static int sn_fill(SnTable& t, const gank_sn_desc* table, int count, int& chunks, int& fine, bool bwd) {
  chunks = 0;
  fine = 0;
  t.count = count;
  for (int i = 0; i < count; i++) {
    t.d[i] = table[i];
    GANK_REQUIRE(t.d[i].K > 0 && t.d[i].C > 0, "sn: weight %d has bad shape", i);
    GANK_REQUIRE(t.d[i].W && t.d[i].u_in && t.d[i].a && t.d[i].b && t.d[i].v && t.d[i].scal && t.d[i].bpart && t.d[i].ga, "sn: weight %d has null pointers", i);
    if (bwd) GANK_REQUIRE(t.d[i].dW_bar && t.d[i].dW, "sn bwd: weight %d has null pointers", i);
    else GANK_REQUIRE(t.d[i].u_out && t.d[i].W_bar, "sn fwd: weight %d has null pointers", i);
    t.d[i].row_offset = fine;          // first fine piece (SN_FINE rows) of the weight
    t.d[i].chunk_offset = chunks;      // first chunk (SN_ROWS rows)
    fine += (t.d[i].K + SN_FINE - 1) / SN_FINE;
    chunks += (t.d[i].K + SN_ROWS - 1) / SN_ROWS;
  }
  return 0;
}

// floats of gank_sn_desc.bpart for a [K, C] weight: per chunk the partial column sums and one |a|^2 partial, per fine piece
// one <G,W> partial; a multiple of 4 (the regions of consecutive weights stay 16-byte aligned)
extern "C" long gank_sn_ws_floats(int K, int C) {
  const long nch = (K + SN_ROWS - 1) / SN_ROWS, nfine = (K + SN_FINE - 1) / SN_FINE;
  return (nch * (C + 1) + nfine + 3) / 4 * 4;
}

// which = 1: forward A only, 2: forward B only (the power iteration's results are in the table's workspaces already: the fused
// optimiser tail or a forward-A-only call put them there), 3: both.  adopt (B): see SnAdopt.
static int sn_forward(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight, int prep_count,
                      const gank_label_dense_desc* label, hipStream_t s, int which = 3, const SnAdopt* adopt = nullptr,
                      const CriticFeedArgs* feed = nullptr) {
  GANK_REQUIRE(table && count > 0, "sn fwd: empty table");
  GANK_REQUIRE(prep_count == 0 || (prep && prep_weight), "sn fwd: preparation entries without their weight indices");
  GANK_REQUIRE(count <= SN_MAX || (prep_count == 0 && !label), "sn fwd: the fused preparation takes at most %d weights per call", SN_MAX);
  GANK_REQUIRE(prep_count <= PREP_MAX, "sn fwd: at most %d preparation entries per call", PREP_MAX);
  // address of the ticket words on the current device, looked up once per device (the first call of a process is an eager
  // one; a lookup inside a stream capture is avoided)
  static std::atomic<unsigned*> ticket_addr[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  unsigned* tickets_base = ticket_addr[dev & 63].load(std::memory_order_relaxed);
  if (!tickets_base) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&tickets_base), HIP_SYMBOL(sn_tickets)) != hipSuccess || !tickets_base)
      return gank_set_error("sn fwd: ticket words not found");
    ticket_addr[dev & 63].store(tickets_base, std::memory_order_relaxed);
  }
  for (int base = 0; base < count; base += SN_MAX) {
    SnTable t;
    int chunks, fine;
    const int n = count - base < SN_MAX ? count - base : SN_MAX;
    if (sn_fill(t, table + base, n, chunks, fine, false)) return 1;
    unsigned* tickets = tickets_base + (sn_ticket_group.fetch_add(1, std::memory_order_relaxed) % SN_TICKET_GROUPS) * SN_MAX;
    if (which & 1) hipLaunchKernelGGL(sn_fwd_a_kernel, dim3(chunks), dim3(256), 0, s, t, tickets);
    if (!(which & 2)) continue;
    SnScaleTable st{};
    st.count = n;
    st.nchunks = fine;
    for (int i = 0; i < n; i++) {
      const gank_sn_desc& d = t.d[i];
      st.d[i] = SnScaleEntry{d.W, d.W_bar, d.scal, d.a, d.b, d.v, d.ga, d.K, d.C, d.row_offset, 0};
    }
    PrepTable pt{};
    SnPrepExtra px{};
    SnLabelDense ld{};
    int label_blocks = 0;
    if (base == 0 && prep_count > 0) {
      const int blocks = prep_table_fill(pt, prep, prep_count, 0);
      if (blocks < 0) return 1;
      px.prep_blocks = blocks;
      for (int i = 0; i < prep_count; i++) {
        GANK_REQUIRE(prep_weight[i] >= 0 && prep_weight[i] < n, "sn fwd: preparation entry %d names weight %d of %d", i, prep_weight[i], n);
        GANK_REQUIRE(prep[i].w == (const float*)table[prep_weight[i]].W, "sn fwd: preparation entry %d must read the MASTER weight of its table entry (the division by sigma happens in the launch)", i);
        px.sigma[i] = t.d[prep_weight[i]].scal;
      }
      for (int i = prep_count; i < PREP_MAX; i++) px.sigma[i] = t.d[0].scal;
    }
    if (base == 0 && label) {
      GANK_REQUIRE(label->table && label->out && label->V > 0 && label->D > 0 && label->weight >= 0 && label->weight < n,
                   "sn fwd: bad label-dense descriptor");
      const gank_sn_desc& d = t.d[label->weight];
      GANK_REQUIRE(d.K == label->D, "sn fwd: label-dense weight %d has %d rows, the table %d columns", label->weight, d.K, label->D);
      ld = SnLabelDense{label->table, d.W, d.scal, label->bias, (bf16*)label->out, label->V, label->D, d.C};
      label_blocks = label->V * ((d.C + 63) / 64);
    }
    SnAdopt ad{};
    if (base == 0 && adopt && adopt->total > 0) {
      ad = *adopt;
      ad.blocks = (ad.total + 255) / 256;
    }
    { static const int dbg_ = gank_tune("GANK_SNB_DBG", 0); px.dbg = dbg_; }
    CriticFeedArgs fd{};
    if (base == 0 && feed) fd = *feed;
    hipLaunchKernelGGL(sn_fwd_b_kernel, dim3(fine + px.prep_blocks + label_blocks + ad.blocks + fd.blocks), dim3(256), 0, s, st, pt, px, ld, ad, fd);
    GANK_LAUNCH_OK("sn_power_iter_fwd");
  }
  return 0;
}

extern "C" int gank_sn_power_iter_fwd(const gank_sn_desc* table, int count, void* stream) {
  return sn_forward(table, count, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
}

extern "C" int gank_sn_power_iter_fwd_prep(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                           int prep_count, const gank_label_dense_desc* label, void* stream) {
  return sn_forward(table, count, prep, prep_weight, prep_count, label, (hipStream_t)stream);
}

extern "C" int gank_sn_power_iter_bwd(const gank_sn_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0, "sn bwd: empty table");
  hipStream_t s = (hipStream_t)stream;
  for (int base = 0; base < count; base += SN_MAX) {
    SnTable t;
    int chunks, fine;
    const int n = count - base < SN_MAX ? count - base : SN_MAX;
    if (sn_fill(t, table + base, n, chunks, fine, true)) return 1;
    hipLaunchKernelGGL(sn_bwd_gw_kernel, dim3(fine), dim3(256), 0, s, t);
    hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(fine), dim3(256), 0, s, t);
    GANK_LAUNCH_OK("sn_power_iter_bwd");
  }
  return 0;
}

// ---------------------------------------------------------------------------------------------- fused optimiser tail (round 5)
// One critic update used to END with  sn_bwd_apply (dW += G/sigma - ...), adam_tf (a pass over the flat buffer) and BEGIN the next one
// with  sn_fwd_a (row dots of the updated weights): three back-to-back passes over the same 6.8 MB, each a dependent 6-17 us step.
// Here a block owns a 32-row chunk of one spectrally normalised weight: it applies the spectral norm's gradient to the chunk
// (the arithmetic of sn_bwd_apply_kernel), takes the TF-Adam step on it (adam_tf_kernel's), clears the gradient slices it has
// consumed (dW and dW_bar), and -- the updated rows still in registers -- computes its rows of a = W_new u and its partial
// column sums of W_new^T a for the NEXT forward pass (sn_fwd_a_kernel's body); the last chunk of a weight (ticket) finishes that
// power iteration into the table's workspaces, u' into a staging buffer (`u_next`: u itself is only advanced when a forward pass
// with update_collection=None consumes it, sn.py:55-56).  The parameters outside the spectrally normalised weights (biases, the
// embedding table) take their Adam step in extra blocks of the same launch.  Bit-identical to the three launches.
struct SnAdamArgs {
  float* p;               // flat parameter buffer [n]; every table entry's W lies inside it
  float* m;
  float* v;
  float* hp;              // {lr, beta1, beta2, eps, grad_scale, decay_on, ticket, -}
  long long* t_state;
  const long long* iteration;
  unsigned long long* health;
  float* g;               // flat gradient buffer [n] (the table's dW are views of it)
  float* u_next[SN_MAX];
  long gap_lo[SN_MAX + 1], gap_hi[SN_MAX + 1];     // [lo, hi) element ranges of the flat buffer outside every table entry
  int gap_block[SN_MAX + 2];                       // first extra block of gap i (1024 elements per block)
  int ngaps, sn_blocks;
  long long* bump;        // optional: a step counter that advances by one when bump_when[0] == 0 (below)
  const int* bump_when;
  int dbg;                // TUNING builds (GANK_SNTAIL_DBG): 1 = no finalize, 2 = return before the publication, 4 = no parameter / slot stores
  int dw_zero;            // every table entry's dW is known to be zero (nothing but this backward pass contributes): neither read nor cleared
};

__device__ __forceinline__ bool sn_adam_elem(float gg, float gs, float b1, float b2, float eps, float lr_t, bool health, float& pp, float& mm, float& vv,
                                             unsigned& bad, unsigned& zero) {
  if (health) {
    bad += (gg - gg != 0.f) ? 1u : 0u;
    zero += gg == 0.f ? 1u : 0u;
    if (gg - gg != 0.f) return false;       // an overflowed element keeps p, m and v (adam_tf_kernel)
  }
  const float gr = gg * gs;
  mm = b1 * mm + (1.f - b1) * gr;
  vv = b2 * vv + (1.f - b2) * gr * gr;
  pp -= lr_t * mm / (sqrtf(vv) + eps);
  return true;
}

__global__ __launch_bounds__(256) void sn_adam_fwd_a_kernel(SnTable t, SnAdamArgs ad, unsigned* __restrict__ tickets) {
  __shared__ float sm[1024 + SN_ROWS + 16 + 4];
  __shared__ float s_lr;
  float* part = sm;
  float* as = sm + 1024;
  float* red = sm + 1024 + SN_ROWS;
  int* flag = reinterpret_cast<int*>(sm + 1024 + SN_ROWS + 16);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // every block derives the step size from the step count BEFORE it takes the optimiser's ticket (at the very end); the block that
  // draws the last one knows that every block has read the count, advances it and resets the ticket (adam_tf_kernel).  The
  // double-precision pow() behind the step size is ~1 us of one lane: a weight chunk's block issues its loads first.
  const float b1 = ad.hp[1], b2 = ad.hp[2], eps = ad.hp[3], gs = ad.hp[4];
  const bool health = ad.health != nullptr;
  unsigned bad = 0u, zero = 0u;

  if ((int)blockIdx.x >= ad.sn_blocks) {
    // ---- plain TF-Adam on a 4096-element piece of a range outside the spectrally normalised weights
    const int eb = blockIdx.x - ad.sn_blocks;
    int gi = 0;
#pragma unroll
    for (int i = 1; i <= SN_MAX; i++) gi += (i < ad.ngaps && eb >= ad.gap_block[i]) ? 1 : 0;
    // (1024 elements per block, every load of a thread requested before the first is used: a 16-iteration load -> store loop was
    //  the launch's longest latency chain)
    const long lo = ad.gap_lo[gi] + (long)(eb - ad.gap_block[gi]) * 1024, hi = min(ad.gap_hi[gi], lo + 1024);
    float pp[4], mm[4], vv[4], gg[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long i = lo + tid + 256 * u;
      const bool ok = i < hi;
      pp[u] = ok ? ad.p[i] : 0.f; mm[u] = ok ? ad.m[i] : 0.f; vv[u] = ok ? ad.v[i] : 0.f; gg[u] = ok ? ad.g[i] : 0.f;
    }
    if (tid == 0) s_lr = adam_lr_t(ad.hp, ad.t_state, ad.iteration);
    __syncthreads();
    const float lr_t = s_lr;
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const long i = lo + tid + 256 * u;
      if (i < hi) {
        ad.g[i] = 0.f;
        if (sn_adam_elem(gg[u], gs, b1, b2, eps, lr_t, health, pp[u], mm[u], vv[u], bad, zero)) { ad.p[i] = pp[u]; ad.m[i] = mm[u]; ad.v[i] = vv[u]; }
      }
    }
  } else {
    int ch;
    const int wi = sn_find_chunk(t, blockIdx.x, ch);
    const gank_sn_desc& d = t.d[wi];
    float* u_next = ad.u_next[0];
#pragma unroll
    for (int i = 1; i < SN_MAX; i++) u_next = wi == i ? ad.u_next[i] : u_next;
    const int K = d.K, C = d.C, k0 = ch * SN_ROWS, kn = min(SN_ROWS, K - k0), nch = (K + SN_ROWS - 1) / SN_ROWS;
    const int nfine = (K + SN_FINE - 1) / SN_FINE;
    float* __restrict__ W = const_cast<float*>(d.W) + (long)k0 * C;
    float* __restrict__ G = const_cast<float*>(d.dW_bar) + (long)k0 * C;
    float* __restrict__ dW = d.dW + (long)k0 * C;
    const long off = (d.W - ad.p) + (long)k0 * C;
    float* __restrict__ M = ad.m + off;
    float* __restrict__ V = ad.v + off;
    const float* gwp = d.bpart + (long)nch * C + nch;
    const float* uold = d.u_snap ? d.u_snap : d.u_in;       // the u the finished forward pass read
    const float* ucur = d.u_in;                             // the u the NEXT forward pass reads
    float* bp = d.bpart + (long)ch * C;
    float n2 = 0.f;
    const bool fast = sn_pow2_c(C) && sn_al16(d.W) && sn_al16(d.dW_bar) && sn_al16(d.dW) && sn_al16(M) && sn_al16(V);
    const int lc = fast ? __builtin_ctz(C) : 0, Gq = C >> 2;
    const int L = max(1, (SN_ROWS * C) >> 10);       // <= 8 pieces per thread (C <= 256)
    f32x4 w[8], mm[8], vv[8];                        // the chunk, its Adam slots: loaded, updated, stored behind the ticket
    if (fast) {
      const int col = (tid * 4) & (C - 1);
      f32x4 g[8], o[8];
      float sv[8], gak[8];
      float b4[4], u4[4], c4[4];
#pragma unroll
      for (int e = 0; e < 4; e++) { b4[e] = d.b[col + e]; u4[e] = uold[col + e]; c4[e] = ucur[col + e]; }
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < L) {
          const int flat = (i * 256 + tid) * 4, row = flat >> lc;
          const bool ok = row < kn;
          const f32x4 z = {0.f, 0.f, 0.f, 0.f};
          w[i] = ok ? *reinterpret_cast<const f32x4*>(W + flat) : z;
          g[i] = ok ? *reinterpret_cast<const f32x4*>(G + flat) : z;
          o[i] = (ok && !ad.dw_zero) ? *reinterpret_cast<const f32x4*>(dW + flat) : z;
          mm[i] = ok ? *reinterpret_cast<const f32x4*>(M + flat) : z;
          vv[i] = ok ? *reinterpret_cast<const f32x4*>(V + flat) : z;
          sv[i] = ok ? d.v[k0 + row] : 0.f;
          gak[i] = ok ? d.ga[k0 + row] : 0.f;
        }
      }
      float s = 0.f;
      for (int j = tid; j < nfine; j += 256) s += gwp[j];
      if (tid == 0) s_lr = adam_lr_t(ad.hp, ad.t_state, ad.iteration);       // (visible behind the barriers of the sum below)
      const float GW = block_sum(s, red);
      const float lr_t = s_lr;
      const float sigma = d.scal[0], sc = d.scal[3];
      const float coef = GW / (sigma * sigma);
      if (ch == 0 && tid == 0) d.scal[4] = GW;
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < L) {
          const int flat = (i * 256 + tid) * 4, row = flat >> lc;
          if (row < kn) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
              o[i][e] += g[i][e] / sigma - coef * ((sc * sv[i]) * b4[e] + gak[i] * u4[e]);       // sn_bwd_apply_kernel
              float pp = w[i][e], m1 = mm[i][e], v1 = vv[i][e];
              sn_adam_elem(o[i][e], gs, b1, b2, eps, lr_t, health, pp, m1, v1, bad, zero);
              w[i][e] = pp; mm[i][e] = m1; vv[i][e] = v1;
            }
          }
        }
      }
#ifdef GANK_TUNING
      if (ad.dbg & 2) { if (w[0][0] == 123.456f) W[0] = 1.f; return; }
#endif
      // (the updated chunk stays in registers: its stores are issued BEHIND the ticket below, so that the publication of the
      //  partial sums does not wait for 80 KB of parameter / slot / cleared-gradient stores to drain)
      // ---- forward A of the next pass on the updated rows (sn_fwd_a_kernel's register-tile body)
      float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < L) {
          const int row = ((i * 256 + tid) * 4) >> lc;
          float pq = ((w[i][0] * c4[0] + w[i][1] * c4[1]) + w[i][2] * c4[2]) + w[i][3] * c4[3];
          pq = sn_row_sum(pq, Gq);
          if (col == 0 && row < kn) { d.a[k0 + row] = pq; n2 += pq * pq; }
#pragma unroll
          for (int e = 0; e < 4; e++) s4[e] += pq * w[i][e];      // rows past kn hold zeros
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < 4; e++) part[tid * 4 + e] = s4[e];
      __syncthreads();
      if (tid < C) {
        const int gq = tid >> 2, e = tid & 3, reps = 1024 >> lc;
        float su = 0.f;
        for (int j = 0; j < reps; j++) su += part[(j * Gq + gq) * 4 + e];
        sn_store_wt(bp + tid, su);
      }
    } else {
      // ---- any other shape (C = 1 of the critic's last dense layer, 256-wide layers, unaligned slices): element-wise apply + Adam,
      // then sn_fwd_a_kernel's generic body on the values just written (the storing thread re-reads its own elements; the column
      // pass reads rows other lanes wrote: ordered by the barrier, through L2 -- stores write through the L1)
      float s = 0.f;
      for (int j = tid; j < nfine; j += 256) s += gwp[j];
      if (tid == 0) s_lr = adam_lr_t(ad.hp, ad.t_state, ad.iteration);
      const float GW = block_sum(s, red);
      const float lr_t = s_lr;
      const float sigma = d.scal[0], sc = d.scal[3];
      const float coef = GW / (sigma * sigma);
      if (ch == 0 && tid == 0) d.scal[4] = GW;
      const int total = kn * C;
      for (int q = tid; q < total; q += 256) {
        const int r = q / C, c = q - r * C;
        const float gg = (ad.dw_zero ? 0.f : dW[q]) + (G[q] / sigma - coef * (sc * d.v[k0 + r] * d.b[c] + d.ga[k0 + r] * uold[c]));
        float pp = W[q], m1 = M[q], v1 = V[q];
        if (sn_adam_elem(gg, gs, b1, b2, eps, lr_t, health, pp, m1, v1, bad, zero)) {
          __hip_atomic_store(W + q, pp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          M[q] = m1; V[q] = v1;
        }
        if (!ad.dw_zero) dW[q] = 0.f;
        G[q] = 0.f;
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      for (int r = wave; r < kn; r += 4) {
        float sd = 0.f;
        for (int c = lane; c < C; c += 64) sd += __hip_atomic_load(W + (long)r * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * ucur[c];
        sd = wave_sum(sd);
        if (lane == 0) { as[r] = sd; d.a[k0 + r] = sd; n2 += sd * sd; }
      }
      __syncthreads();
      for (int c = tid; c < C; c += 256) {
        float su = 0.f;
        for (int r = 0; r < kn; r++) su += as[r] * __hip_atomic_load(W + (long)r * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sn_store_wt(bp + c, su);
      }
    }
    n2 = block_sum(n2, red);
    if (tid == 0) sn_store_wt(d.bpart + (long)nch * C + ch, n2);
    // publish, ticket, finish (as sn_fwd_a_kernel)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
      int last = 1;
      if (nch > 1) {
        const unsigned tk = __hip_atomic_fetch_add(tickets + wi, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = tk == (unsigned)(nch - 1);
        if (last) {
          __hip_atomic_store(tickets + wi, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      }
      *flag = last;
    }
#ifdef GANK_TUNING
    if (!(ad.dbg & 4))
#endif
    if (fast) {          // the updated chunk, its slots, the cleared gradient slices: beside the ticket's round trip
      const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; i++) {
        if (i < L) {
          const int flat = (i * 256 + tid) * 4, row = flat >> lc;
          if (row < kn) {
            *reinterpret_cast<f32x4*>(W + flat) = w[i];
            *reinterpret_cast<f32x4*>(M + flat) = mm[i];
            *reinterpret_cast<f32x4*>(V + flat) = vv[i];
            if (!ad.dw_zero) *reinterpret_cast<f32x4*>(dW + flat) = z4;
            *reinterpret_cast<f32x4*>(G + flat) = z4;
          }
        }
      }
    }
    __syncthreads();
#ifdef GANK_TUNING
    if (!(ad.dbg & 1))
#endif
    if (*flag) sn_fwd_a_finish(d, u_next, nch, part, red, tid);
  }
  if (health) {           // one pair of atomics per wave that saw anything (adam_tf_kernel)
    const unsigned long long b64 = (unsigned long long)wave_sum((float)bad);
    const unsigned long long z64 = (unsigned long long)wave_sum((float)zero);
    if ((threadIdx.x & 63) == 0) {
      if (b64) atomicAdd(ad.health, b64);
      if (z64) atomicAdd(ad.health + 1, z64);
    }
  }
  // the optimiser's ticket LAST (the step count was read into s_lr at the top, through LDS: the load has completed): ~360 same-address
  // atomics whose return a block would otherwise wait for in front of its first barrier
  if (tid == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(ad.hp + 6);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;
      ad.t_state[0] += 1;
      // the train loop's iteration counter (the `_iteration` feed of the LR decay, gan_cifar_resnet.py:320,454-459) advances behind
      // the LAST critic update of an iteration: every block of this launch has derived its step size from the old value by now
      if (ad.bump && ad.bump_when[0] == 0) ad.bump[0] += 1;
    }
  }
}

static unsigned* sn_ticket_base() {
  static std::atomic<unsigned*> ticket_addr[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  unsigned* base = ticket_addr[dev & 63].load(std::memory_order_relaxed);
  if (!base) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(sn_tickets)) != hipSuccess || !base) return nullptr;
    ticket_addr[dev & 63].store(base, std::memory_order_relaxed);
  }
  return base;
}

extern "C" int gank_sn_power_iter_fwd_a(const gank_sn_desc* table, int count, void* stream) {
  return sn_forward(table, count, nullptr, nullptr, 0, nullptr, (hipStream_t)stream, 1);
}

extern "C" int gank_sn_power_iter_fwd_b_prep(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                             int prep_count, const gank_label_dense_desc* label, float* u_flat, float* u_snap_flat,
                                             const float* u_next_flat, int u_total, void* stream) {
  GANK_REQUIRE(count <= SN_MAX, "sn fwd b: at most %d weights per call", SN_MAX);
  GANK_REQUIRE(u_total == 0 || (u_flat && u_next_flat), "sn fwd b: adopting u' needs the flat u and staging buffers");
  SnAdopt ad{};
  ad.u = u_flat; ad.u_snap = u_snap_flat; ad.u_next = u_next_flat; ad.total = u_total;
  return sn_forward(table, count, prep, prep_weight, prep_count, label, (hipStream_t)stream, 2, &ad);
}

extern "C" int gank_sn_power_iter_fwd_b_prep_feed(const gank_sn_desc* table, int count, const gank_prep_desc* prep, const int* prep_weight,
                                                  int prep_count, const gank_label_dense_desc* label, float* u_flat, float* u_snap_flat,
                                                  const float* u_next_flat, int u_total, const gank_critic_feed_desc* feed, void* stream) {
  GANK_REQUIRE(count <= SN_MAX, "sn fwd b: at most %d weights per call", SN_MAX);
  GANK_REQUIRE(u_total == 0 || (u_flat && u_next_flat), "sn fwd b: adopting u' needs the flat u and staging buffers");
  GANK_REQUIRE(feed && feed->real_all && feed->labels_all && feed->fake_all && feed->both && feed->labels2 && feed->slot && feed->rng_state &&
               feed->done_counter && feed->B > 0 && feed->n_slots > 0, "sn fwd b: bad feed descriptor");
  SnAdopt ad{};
  ad.u = u_flat; ad.u_snap = u_snap_flat; ad.u_next = u_next_flat; ad.total = u_total;
  const CriticFeedArgs fd{feed->real_all, feed->labels_all, (const bf16*)feed->fake_all, (bf16*)feed->both, feed->labels2, feed->slot,
                          (unsigned long long*)feed->rng_state, feed->done_counter, feed->B, feed->n_slots, critic_feed_blocks(feed->B)};
  return sn_forward(table, count, prep, prep_weight, prep_count, label, (hipStream_t)stream, 2, &ad, &fd);
}

extern "C" int gank_sn_power_iter_bwd_gw(const gank_sn_desc* table, int count, void* stream) {
  GANK_REQUIRE(table && count > 0 && count <= SN_MAX, "sn bwd gw: 1..%d weights", SN_MAX);
  SnTable t;
  int chunks, fine;
  if (sn_fill(t, table, count, chunks, fine, true)) return 1;
  hipLaunchKernelGGL(sn_bwd_gw_kernel, dim3(fine), dim3(256), 0, (hipStream_t)stream, t);
  GANK_LAUNCH_OK("sn_power_iter_bwd_gw");
  return 0;
}

extern "C" int gank_sn_adam_fwd_a(const gank_sn_desc* table, int count, float* const* u_next, float* p, float* g, float* m, float* v, long n,
                                  float* hp, int64_t* t_state, const int64_t* iteration, uint64_t* health, int flags, int64_t* bump,
                                  const int32_t* bump_when_zero, void* stream) {
  GANK_REQUIRE(table && count > 0 && count <= SN_MAX && u_next && p && g && m && v && hp && t_state && n > 0, "sn_adam_fwd_a: bad arguments");
  SnTable t;
  int chunks, fine;
  if (sn_fill(t, table, count, chunks, fine, true)) return 1;
  SnAdamArgs ad{};
  ad.p = p; ad.g = g; ad.m = m; ad.v = v; ad.hp = hp; ad.t_state = (long long*)t_state; ad.iteration = (const long long*)iteration;
  ad.health = (unsigned long long*)health;
  ad.sn_blocks = chunks;
  ad.dw_zero = (flags & 1) ? 1 : 0;
  GANK_REQUIRE(!bump || bump_when_zero, "sn_adam_fwd_a: a counter to advance needs its condition word");
  ad.bump = (long long*)bump; ad.bump_when = bump_when_zero;
  { static const int dbg_ = gank_tune("GANK_SNTAIL_DBG", 0); ad.dbg = dbg_; }
  // the table's weights, in buffer order, must be disjoint views of [p, p + n) whose gradient views sit at the same offsets of g
  int order[SN_MAX];
  for (int i = 0; i < count; i++) order[i] = i;
  for (int i = 1; i < count; i++)
    for (int j = i; j > 0 && t.d[order[j]].W < t.d[order[j - 1]].W; j--) { const int x = order[j]; order[j] = order[j - 1]; order[j - 1] = x; }
  long pos = 0;
  int blocks = 0;
  for (int i = 0; i <= count; i++) {
    long lo = n, len = 0;
    if (i < count) {
      const gank_sn_desc& d = t.d[order[i]];
      GANK_REQUIRE(u_next[order[i]] && d.C <= 256, "sn_adam_fwd_a: weight %d: no staging buffer, or more than 256 columns", order[i]);
      lo = d.W - p; len = (long)d.K * d.C;
      GANK_REQUIRE(lo >= pos && lo + len <= n, "sn_adam_fwd_a: weight %d is not a disjoint view of the flat parameter buffer", order[i]);
      GANK_REQUIRE(d.dW == g + lo, "sn_adam_fwd_a: weight %d: its gradient is not the view of the flat gradient buffer at the weight's offset", order[i]);
      ad.u_next[order[i]] = u_next[order[i]];
    }
    if (lo > pos) {
      ad.gap_lo[ad.ngaps] = pos; ad.gap_hi[ad.ngaps] = lo; ad.gap_block[ad.ngaps] = blocks;
      blocks += (int)((lo - pos + 1023) / 1024);
      ad.ngaps++;
    }
    pos = lo + len;
  }
  ad.gap_block[ad.ngaps] = blocks;
  unsigned* tickets_base = sn_ticket_base();
  if (!tickets_base) return gank_set_error("sn_adam_fwd_a: ticket words not found");
  unsigned* tickets = tickets_base + (sn_ticket_group.fetch_add(1, std::memory_order_relaxed) % SN_TICKET_GROUPS) * SN_MAX;
  hipLaunchKernelGGL(sn_adam_fwd_a_kernel, dim3(chunks + blocks), dim3(256), 0, (hipStream_t)stream, t, ad, tickets);
  GANK_LAUNCH_OK("sn_adam_fwd_a");
  return 0;
}

// The label table on its own (a caller without a batched spectral norm around it): out[l] = bf16(bf16(table[l]) (W / sigma) + bias)
__global__ __launch_bounds__(256) void label_dense_table_kernel(SnLabelDense q) {
  __shared__ float lred[256];
  const int ncg = (q.Cout + 63) >> 6;
  sn_label_row(q, blockIdx.x / ncg, blockIdx.x % ncg, lred);
}
extern "C" int gank_label_dense_table(const float* table, const float* W, const float* sigma, const float* bias, void* out,
                                      int V, int D, int Cout, void* stream) {
  GANK_REQUIRE(table && W && out && V > 0 && D > 0 && Cout > 0, "label_dense_table: bad arguments");
  SnLabelDense q{table, W, sigma, bias, (bf16*)out, V, D, Cout};
  hipLaunchKernelGGL(label_dense_table_kernel, dim3(V * ((Cout + 63) / 64)), dim3(256), 0, (hipStream_t)stream, q);
  GANK_LAUNCH_OK("label_dense_table");
  return 0;
}
