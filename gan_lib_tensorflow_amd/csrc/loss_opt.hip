// Losses, TF-style Adam, the input pipeline and the graph-safe counter-based RNG.
#include "gank_common.h"
#include "feed.h"

// ------------------------------------------------------------------------------------------------
// hinge / softmax-xent losses: loss value + d loss / d logits in one single-block launch
// ------------------------------------------------------------------------------------------------
// Every loss kernel writes d loss / d logits twice: bf16 (`dl`, what the backward pass of the layer below consumes when
// the loss itself is differentiated, upstream gradient 1) and fp32 (`dl32`, optional: the exact values, scaled by the
// upstream gradient and rounded ONCE by gank_loss_grad_scale when the loss enters a weighted sum).
__global__ void hinge_d_kernel(const bf16* __restrict__ l, float* __restrict__ loss, bf16* __restrict__ dl, float* __restrict__ dl32, int n, int n_real) {
  __shared__ float red[16];
  const int n_fake = n - n_real;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = bf2f(l[i]);
    float d;
    if (i < n_real) {  // mean(relu(1 - real))   gan_cifar_resnet.py:379
      const float t = 1.f - v;
      acc += fmaxf(t, 0.f) / (float)n_real;
      d = t > 0.f ? -1.f / (float)n_real : 0.f;
    } else {           // mean(relu(1 + fake))   gan_cifar_resnet.py:380
      const float t = 1.f + v;
      acc += fmaxf(t, 0.f) / (float)n_fake;
      d = t > 0.f ? 1.f / (float)n_fake : 0.f;
    }
    dl[i] = f2bf(d);
    if (dl32) dl32[i] = d;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = tot;
}

__global__ void hinge_g_kernel(const bf16* __restrict__ l, float* __restrict__ loss, bf16* __restrict__ dl, float* __restrict__ dl32, int n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    acc += bf2f(l[i]);
    dl[i] = f2bf(-1.f / (float)n);
    if (dl32) dl32[i] = -1.f / (float)n;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = -tot / (float)n;  // -mean(disc_fake)   gan_cifar_resnet.py:492
}

__global__ void softmax_xent_kernel(const bf16* __restrict__ lg, const int* __restrict__ labels, float* __restrict__ loss,
                                    bf16* __restrict__ dl, float* __restrict__ dl32, int n, int classes) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float mx = -3.0e38f;
    for (int c = 0; c < classes; c++) mx = fmaxf(mx, bf2f(lg[(long)i * classes + c]));
    float se = 0.f;
    for (int c = 0; c < classes; c++) se += expf(bf2f(lg[(long)i * classes + c]) - mx);
    const float lse = logf(se);
    const int lb = labels[i];
    for (int c = 0; c < classes; c++) {
      const float z = bf2f(lg[(long)i * classes + c]) - mx - lse;
      if (c == lb) acc -= z / (float)n;
      const float d = (expf(z) - (c == lb ? 1.f : 0.f)) / (float)n;
      dl[(long)i * classes + c] = f2bf(d);
      if (dl32) dl32[(long)i * classes + c] = d;
    }
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = tot;
}

// WGAN critic loss  -mean(real) + mean(fake)   (misc.py:328-331; the 'WGAN-GP' branch, :337-352, is the same pair: its
// penalty term is added by the caller -- gank_gp_loss).  Generator side: gank_hinge_g_loss (-mean(fake), :335,:352).
__global__ void wgan_d_kernel(const bf16* __restrict__ l, float* __restrict__ loss, bf16* __restrict__ dl, float* __restrict__ dl32, int n, int n_real) {
  __shared__ float red[16];
  const int n_fake = n - n_real;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = bf2f(l[i]);
    const float d = i < n_real ? -1.f / (float)n_real : 1.f / (float)n_fake;
    acc += v * d;
    dl[i] = f2bf(d);
    if (dl32) dl32[i] = d;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = tot;
}
// The remaining branches of get_loss (common/misc.py:353-394): least squares and the sigmoid cross-entropy / minimax family.
//   kind 0 LSGAN critic     (mean((1 - real)^2) + mean(fake^2)) / 2                           :355-357
//   kind 1 LSGAN generator  mean((1 - fake)^2) / 2                                            :360
//   kind 2 critic of CGAN / Modified_MiniMax / MiniMax:  mean(-log sigmoid(real)) + mean(-log(1 - sigmoid(fake)))
//          = mean(softplus(-real)) + mean(softplus(fake))  (sigmoid_cross_entropy_with_logits, labels 1 / 0)   :362-368, :376-378, :385-387
//   kind 3 generator of CGAN / Modified_MiniMax:  mean(-log sigmoid(fake)) = mean(softplus(-fake))     :370-372, :381
//   kind 4 generator of MiniMax:  mean(log(1 - sigmoid(fake))) = -mean(softplus(fake))                 :390
// softplus(x) = max(x, 0) + log1p(exp(-|x|)) (TensorFlow's stable form); d softplus / dx = sigmoid(x).
__device__ __forceinline__ float gank_softplus(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }
__device__ __forceinline__ float gank_sigmoid(float x) { return 1.f / (1.f + expf(-x)); }
__global__ void gan_pointwise_loss_kernel(const bf16* __restrict__ l, float* __restrict__ loss, bf16* __restrict__ dl, float* __restrict__ dl32, int n, int n_real, int kind) {
  __shared__ float red[16];
  const int n_fake = n - n_real;
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float v = bf2f(l[i]);
    float t, d;
    if (kind == 0) {
      if (i < n_real) { t = 0.5f * (1.f - v) * (1.f - v) / (float)n_real; d = -(1.f - v) / (float)n_real; }
      else { t = 0.5f * v * v / (float)n_fake; d = v / (float)n_fake; }
    } else if (kind == 1) {
      t = 0.5f * (1.f - v) * (1.f - v) / (float)n; d = -(1.f - v) / (float)n;
    } else if (kind == 2) {
      if (i < n_real) { t = gank_softplus(-v) / (float)n_real; d = -gank_sigmoid(-v) / (float)n_real; }
      else { t = gank_softplus(v) / (float)n_fake; d = gank_sigmoid(v) / (float)n_fake; }
    } else if (kind == 3) {
      t = gank_softplus(-v) / (float)n; d = -gank_sigmoid(-v) / (float)n;
    } else if (kind == 4) {
      t = -gank_softplus(v) / (float)n; d = -gank_sigmoid(v) / (float)n;
    } else if (kind == 5) {            // SOFT_PLUS 'Goodfellow' critic: -softplus(log sigmoid(real)), -softplus(log(1 - sigmoid(fake)))
      if (i < n_real) { const float u = -gank_softplus(-v); t = -gank_softplus(u) / (float)n_real; d = -gank_sigmoid(u) * gank_sigmoid(-v) / (float)n_real; }
      else { const float u = -gank_softplus(v); t = -gank_softplus(u) / (float)n_fake; d = gank_sigmoid(u) * gank_sigmoid(v) / (float)n_fake; }
    } else if (kind == 6) {            // SOFT_PLUS 'Goodfellow' generator: softplus(-log sigmoid(fake))
      const float u = gank_softplus(-v);
      t = gank_softplus(u) / (float)n; d = -gank_sigmoid(u) * gank_sigmoid(-v) / (float)n;
    } else {                           // kind 7, SOFT_PLUS 'HINGE' critic: softplus(-min(0, -1 + real)), softplus(-min(0, -1 - fake))
      if (i < n_real) { const float m = fminf(0.f, v - 1.f); t = gank_softplus(-m) / (float)n_real; d = v < 1.f ? -gank_sigmoid(-m) / (float)n_real : 0.f; }
      else { const float m = fminf(0.f, -1.f - v); t = gank_softplus(-m) / (float)n_fake; d = v > -1.f ? gank_sigmoid(-m) / (float)n_fake : 0.f; }
    }
    acc += t;
    dl[i] = f2bf(d);
    if (dl32) dl32[i] = d;
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) loss[0] = tot;
}
extern "C" int gank_gan_pointwise_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, int kind, void* stream) {
  GANK_REQUIRE(logits && loss && dlogits && n > 0 && kind >= 0 && kind <= 7, "gan_pointwise_loss: bad arguments");
  GANK_REQUIRE((kind != 0 && kind != 2 && kind != 5 && kind != 7) || (n_real > 0 && n_real < n), "gan_pointwise_loss: a critic loss needs 0 < n_real < n");
  hipLaunchKernelGGL(gan_pointwise_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, loss, (bf16*)dlogits, dlogits_f32, n, n_real, kind);
  GANK_LAUNCH_OK("gan_pointwise_loss");
  return 0;
}
extern "C" int gank_wgan_d_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, void* stream) {
  GANK_REQUIRE(logits && loss && dlogits && n > 0 && n_real > 0 && n_real < n, "wgan_d_loss: bad arguments");
  hipLaunchKernelGGL(wgan_d_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, loss, (bf16*)dlogits, dlogits_f32, n, n_real);
  GANK_LAUNCH_OK("wgan_d_loss");
  return 0;
}
extern "C" int gank_hinge_d_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, int n_real, void* stream) {
  GANK_REQUIRE(logits && loss && dlogits && n > 0 && n_real > 0 && n_real < n, "hinge_d_loss: bad arguments");
  hipLaunchKernelGGL(hinge_d_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, loss, (bf16*)dlogits, dlogits_f32, n, n_real);
  GANK_LAUNCH_OK("hinge_d_loss");
  return 0;
}
extern "C" int gank_hinge_g_loss(const void* logits, float* loss, void* dlogits, float* dlogits_f32, int n, void* stream) {
  GANK_REQUIRE(logits && loss && dlogits && n > 0, "hinge_g_loss: bad arguments");
  hipLaunchKernelGGL(hinge_g_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, loss, (bf16*)dlogits, dlogits_f32, n);
  GANK_LAUNCH_OK("hinge_g_loss");
  return 0;
}
extern "C" int gank_softmax_xent(const void* logits, const int32_t* labels, float* loss, void* dlogits, float* dlogits_f32, int n, int classes, void* stream) {
  GANK_REQUIRE(logits && labels && loss && dlogits && n > 0 && classes > 0, "softmax_xent: bad arguments");
  hipLaunchKernelGGL(softmax_xent_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const bf16*)logits, labels, loss, (bf16*)dlogits, dlogits_f32, n, classes);
  GANK_LAUNCH_OK("softmax_xent");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Critic head in ONE launch: logits = x w + b (D.Output, gan_cifar_resnet.py:303-304), the hinge loss of them (:379-381 /
// :492) and everything the backward pass needs from this layer -- d loss / d x, and (w_grad / b_grad given) the weight and
// bias gradients ACCUMULATED.  Unfused this is four latency-sized launches (linear fwd, loss, linear bwd data, linear bwd
// weight) in every critic pass.  The arithmetic is that of the unfused path: the logit is rounded to bf16 before the loss
// reads it, d loss / d logit is rounded to bf16 before the two products.  One block of 1024 threads; M * K is a few
// thousand multiply-adds.  mode 0: hinge_d (first n_real rows are real), mode 1: hinge_g.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void critic_head_hinge_kernel(const bf16* __restrict__ x, const float* __restrict__ w, const float* __restrict__ b,
                                                                bf16* __restrict__ logits, float* __restrict__ loss, bf16* __restrict__ dx,
                                                                float* __restrict__ w_grad, float* __restrict__ b_grad, int M, int K, int n_real, int mode, float loss_scale) {
  extern __shared__ float sm[];          // dl[M] | w[K] | red[32] | part[8][K]
  float* s_dl = sm;
  float* s_w = sm + M;
  float* red = s_w + K;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6;
  for (int k = tid; k < K; k += blockDim.x) s_w[k] = w[k];
  __syncthreads();
  const float bias = b ? b[0] : 0.f;
  const int n_fake = M - n_real;
  float acc = 0.f;
  for (int mb = wv; mb < M; mb += 8 * nw) {          // one wave per row, 8 rows in flight (a row at a time was 8 dependent
    float tt[8];                                      // load -> reduce -> store chains in a row: most of this kernel's time)
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const int m = mb + u * nw;
      float t = 0.f;
      if (m < M)
        for (int k = lane; k < K; k += 64) t += bf2f(x[(long)m * K + k]) * s_w[k];
      tt[u] = t;
    }
#pragma unroll
    for (int u = 0; u < 8; u++) tt[u] = wave_sum(tt[u]);
#pragma unroll
    for (int u = 0; u < 8; u++) {
    const int m = mb + u * nw;
    const float t = tt[u];
    if (lane == 0 && m < M) {
      const bf16 lg = f2bf(t + bias);
      logits[m] = lg;
      const float v = bf2f(lg);
      float d, l;
      if (mode == 1) { l = -v / (float)M; d = -1.f / (float)M; }
      else if (m < n_real) { const float u = 1.f - v; l = fmaxf(u, 0.f) / (float)n_real; d = u > 0.f ? -1.f / (float)n_real : 0.f; }
      else { const float u = 1.f + v; l = fmaxf(u, 0.f) / (float)n_fake; d = u > 0.f ? 1.f / (float)n_fake : 0.f; }
      s_dl[m] = bf2f(f2bf(d * loss_scale));      // loss_scale: a power of two (static loss scaling of the fp16 build), undone by the optimiser's grad_scale
      acc += l;
    }
    }
  }
  if (lane != 0) acc = 0.f;
  __syncthreads();
  // block sum of the per-row loss terms (lanes 0 of every wave hold them)
  acc = wave_sum(acc);
  if (lane == 0) red[wv] = acc;
  __syncthreads();
  if (tid == 0) {
    float t = 0.f;
    for (int i = 0; i < nw; i++) t += red[i];
    loss[0] = t;
  }
  // d loss / d x = dl[m] * w[k]
  if (dx)
    for (long i = tid; i < (long)M * K; i += blockDim.x) {
      const int m = (int)(i / K), k = (int)(i - (long)m * K);
      dx[i] = f2bf(s_dl[m] * s_w[k]);
    }
  // w_grad[k] += sum_m x[m][k] dl[m]: the rows in 8 slices per column (a thread per (slice, k): 128 dependent loads in one
  // thread were 10 of this kernel's 16 us), partial sums meet in LDS in slice order: deterministic;  b_grad += sum_m dl[m]
  if (w_grad) {
    __syncthreads();                         // red[] is reused below (K <= 128 * ... guarded by the host: 8 * K floats of LDS)
    float* part = red + 32;                  // [8][K]
    const int nsl = 8, per = (M + nsl - 1) / nsl;
    for (int i = tid; i < nsl * K; i += blockDim.x) {
      const int sl = i / K, k = i - sl * K;
      const int m0 = sl * per, m1 = min(M, m0 + per);
      float t = 0.f;
      for (int m = m0; m < m1; m++) t += bf2f(x[(long)m * K + k]) * s_dl[m];
      part[i] = t;
    }
    __syncthreads();
    for (int k = tid; k < K; k += blockDim.x) {
      float t = 0.f;
#pragma unroll
      for (int sl = 0; sl < 8; sl++) t += part[sl * K + k];
      w_grad[k] += t;
    }
  }
  if (b_grad && tid < 64) {
    float t = 0.f;
    for (int m = tid; m < M; m += 64) t += s_dl[m];
    t = wave_sum(t);
    if (tid == 0) b_grad[0] += t;
  }
}
extern "C" int gank_critic_head_hinge_scaled(const void* x, const float* w, const float* b, void* logits, float* loss, void* dx, float* w_grad,
                                             float* b_grad, int M, int K, int n_real, int mode, float loss_scale, void* stream) {
  GANK_REQUIRE(x && w && logits && loss && M > 0 && K > 0 && (mode == 0 || mode == 1), "critic_head_hinge: bad arguments");
  GANK_REQUIRE(mode == 1 || (n_real > 0 && n_real < M), "critic_head_hinge: n_real must split the batch");
  GANK_REQUIRE(loss_scale > 0.f, "critic_head_hinge: loss_scale must be positive");
  GANK_REQUIRE((size_t)(M + 9 * K + 32) * sizeof(float) <= 60000, "critic_head_hinge: M + 9 K too large for one block");
  hipLaunchKernelGGL(critic_head_hinge_kernel, dim3(1), dim3(1024), (size_t)(M + 9 * K + 32) * sizeof(float), (hipStream_t)stream, (const bf16*)x, w, b,
                     (bf16*)logits, loss, (bf16*)dx, w_grad, b_grad, M, K, n_real, mode, loss_scale);
  GANK_LAUNCH_OK("critic_head_hinge");
  return 0;
}
extern "C" int gank_critic_head_hinge(const void* x, const float* w, const float* b, void* logits, float* loss, void* dx, float* w_grad,
                                      float* b_grad, int M, int K, int n_real, int mode, void* stream) {
  return gank_critic_head_hinge_scaled(x, w, b, logits, loss, dx, w_grad, b_grad, M, K, n_real, mode, 1.f, stream);
}

// d(total)/d(logits) = g[0] * d(loss)/d(logits) for a loss that enters a weighted sum (gen_cost + ACGAN_SCALE_G * xent,
// gan_cifar_resnet.py:476; ACGAN/train.py:119-121): product in fp32, one rounding to bf16
__global__ void loss_grad_scale_kernel(const float* __restrict__ dl32, const float* __restrict__ g, bf16* __restrict__ out, long n) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i < n) out[i] = f2bf(dl32[i] * g[0]);
}
extern "C" int gank_loss_grad_scale(const float* dlogits_f32, const float* g, void* dlogits, long n, void* stream) {
  GANK_REQUIRE(dlogits_f32 && g && dlogits && n > 0, "loss_grad_scale: bad arguments");
  hipLaunchKernelGGL(loss_grad_scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dlogits_f32, g, (bf16*)dlogits, n);
  GANK_LAUNCH_OK("loss_grad_scale");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer over a flat fp32 buffer   (gan_cifar_resnet.py:521-526), step state on device
// hp = {lr, beta1, beta2, eps, grad_scale, decay_on};  t_state[0] = steps taken so far;
// iteration[0] = training iteration fed as `_iteration` (:320, :454-459).
//   decay = iteration < 50000 ? max(0, 1 - iteration/100000) : 0.5          (if decay_on)
//   lr_t  = lr * decay * sqrt(1 - beta2^t) / (1 - beta1^t),  t = t_state[0] + 1
//   m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; p -= lr_t m / (sqrt(v) + eps)
// ------------------------------------------------------------------------------------------------
// (adam_lr_t: gank_common.h -- shared with the fused spectral-norm / optimiser launch of sn.hip)

// One launch per update: the step count advances and (zero_n > 0) the gradient buffer is cleared here too.  Every block
// derives lr_t from t_state[0] BEFORE it takes a ticket (the value goes through LDS, so the load has completed); the block that
// draws the last ticket therefore knows every block has read the count, advances it and resets the ticket (hp[6], as an integer).
// zero_n floats of g (>= n: the critic's buffer carries the spectral-norm scratch half behind the gradients) are cleared after
// they have been consumed, so the next backward pass accumulates into zeros without a fill launch of its own.
__global__ void adam_tf_kernel(float* __restrict__ p, float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                               float* __restrict__ hp, long long* __restrict__ t_state,
                               const long long* __restrict__ iteration, long n, long zero_n, unsigned long long* __restrict__ health) {
  __shared__ float s_lr;
  if (threadIdx.x == 0) s_lr = adam_lr_t(hp, t_state, iteration);
  __syncthreads();
  const float lr_t = s_lr, b1 = hp[1], b2 = hp[2], eps = hp[3], gs = hp[4];
  if (threadIdx.x == 0) {
    unsigned* ticket = reinterpret_cast<unsigned*>(hp + 6);
    if (atomicAdd(ticket, 1u) == gridDim.x - 1) {
      *ticket = 0u;
      t_state[0] += 1;
    }
  }
  const long n4 = n >> 2;
  const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
  unsigned bad = 0u, zero = 0u;        // health (optional): gradients that are not finite (loss-scale overflow) / exactly zero
  // (two elements per trip with all eight loads up front measured the same: 46.4 against 45.7 us for 137 MB)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 pp = reinterpret_cast<f32x4*>(p)[i];
    const f32x4 gg = reinterpret_cast<const f32x4*>(g)[i];
    if (health) {
#pragma unroll
      for (int e = 0; e < 4; e++) { bad += (gg[e] - gg[e] != 0.f) ? 1u : 0u; zero += gg[e] == 0.f ? 1u : 0u; }
    }
    f32x4 mm = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float gr = gg[e] * gs;
      // loss-scaled runs (health != nullptr): an element whose gradient overflowed keeps p, m and v -- one fp16 overflow must
      // not write NaN into the optimiser state for good (the captured graph would keep replaying on it); it is counted above
      if (health && (gg[e] - gg[e] != 0.f)) continue;
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      pp[e] -= lr_t * mm[e] / (sqrtf(vv[e]) + eps);
    }
    reinterpret_cast<f32x4*>(p)[i] = pp;
    reinterpret_cast<f32x4*>(m)[i] = mm;
    reinterpret_cast<f32x4*>(v)[i] = vv;
    if (zero_n > 0) reinterpret_cast<f32x4*>(g)[i] = z4;
  }
  if (blockIdx.x == 0)
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
      const float gr = g[i] * gs;
      if (zero_n > 0) g[i] = 0.f;
      if (health) {
        bad += (gr - gr != 0.f) ? 1u : 0u; zero += gr == 0.f ? 1u : 0u;
        if (gr - gr != 0.f) continue;
      }
      const float mm = b1 * m[i] + (1.f - b1) * gr;
      const float vv = b2 * v[i] + (1.f - b2) * gr * gr;
      m[i] = mm; v[i] = vv;
      p[i] -= lr_t * mm / (sqrtf(vv) + eps);
    }
  // the tail behind the gradients (n is a multiple of 4 whenever zero_n > n: checked by the host)
  for (long i = n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < (zero_n >> 2); i += (long)gridDim.x * blockDim.x)
    reinterpret_cast<f32x4*>(g)[i] = z4;
  if (health) {           // one pair of atomics per wave that saw anything
    const unsigned long long b64 = (unsigned long long)wave_sum((float)bad);
    const unsigned long long z64 = (unsigned long long)wave_sum((float)zero);
    if ((threadIdx.x & 63) == 0) {
      if (b64) atomicAdd(health, b64);
      if (z64) atomicAdd(health + 1, z64);
    }
  }
}

__global__ void counter_add_kernel(long long* c, long long inc) { c[0] += inc; }

extern "C" int gank_adam_tf_health(float* p, float* g, float* m, float* v, float* hp, int64_t* t_state,
                                   const int64_t* iteration, long n, long zero_n, uint64_t* health, void* stream) {
  GANK_REQUIRE(p && g && m && v && hp && t_state && n > 0, "adam_tf: bad arguments");
  GANK_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adam_tf: buffers must be 16-byte aligned");
  GANK_REQUIRE(zero_n == 0 || zero_n == n || (zero_n > n && n % 4 == 0 && zero_n % 4 == 0), "adam_tf: zero_n must be 0, n, or a multiple of 4 beyond an n that is one");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 512) blocks = 512;      // two per CU: every block draws a ticket from ONE address, and same-address atomics serialise (1662 blocks: +16 us)
  if (blocks < 1) blocks = 1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(adam_tf_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, hp, (long long*)t_state,
                     (const long long*)iteration, n, zero_n, (unsigned long long*)health);
  GANK_LAUNCH_OK("adam_tf");
  return 0;
}

extern "C" int gank_adam_tf(float* p, float* g, float* m, float* v, float* hp, int64_t* t_state,
                            const int64_t* iteration, long n, long zero_n, void* stream) {
  return gank_adam_tf_health(p, g, m, v, hp, t_state, iteration, n, zero_n, nullptr, stream);
}

extern "C" int gank_counter_add(int64_t* counter, int64_t inc, void* stream) {
  GANK_REQUIRE(counter, "counter_add: null pointer");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (long long*)counter, (long long)inc);
  GANK_LAUNCH_OK("counter_add");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based RNG; state = {seed, offset} in device memory, advanced on the device
// ------------------------------------------------------------------------------------------------
__global__ void rng_advance_kernel(unsigned long long* state, unsigned long long inc) { state[1] += inc; }

// The generators advance the stream offset THEMSELVES: every workgroup has read {seed, offset} (their use feeds the address-free
// arithmetic below, and the barrier in front of the ticket orders the loads) before it draws a ticket; the last one to arrive
// rewrites the offset and resets the word -- no one-thread rng_advance launch behind every draw (3 per iteration).  The ticket words
// rotate per host call, so draws in flight on different streams do not share one.
__device__ unsigned rng_tickets[32];
static std::atomic<unsigned> rng_ticket_next{0};
static unsigned* rng_ticket_word() {
  static std::atomic<unsigned*> base_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  unsigned* base = base_of[dev & 63].load(std::memory_order_relaxed);
  if (!base) {
    if (hipGetSymbolAddress(reinterpret_cast<void**>(&base), HIP_SYMBOL(rng_tickets)) != hipSuccess || !base) return nullptr;
    base_of[dev & 63].store(base, std::memory_order_relaxed);
  }
  return base + (rng_ticket_next.fetch_add(1, std::memory_order_relaxed) & 31);
}
__device__ __forceinline__ void rng_finish(unsigned long long* state, unsigned long long off, unsigned* done) {
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(done, 1u) == gridDim.x - 1) {
    *done = 0u;
    state[1] = off + 1;
  }
}

__global__ void rng_normal_kernel(bf16* __restrict__ y, long n, unsigned long long* __restrict__ state, unsigned* __restrict__ done) {
  const unsigned long long seed = state[0], off = state[1];
  const long n4 = (n + 3) >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u4 r = philox4x32_10((unsigned long long)i, off, seed);
    const float u1 = 1.f - u01(r.x), u2 = u01(r.y), u3 = 1.f - u01(r.z), u4_ = u01(r.w);  // (0,1]
    const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
    const float z[4] = {ra * cosf(6.2831853f * u2), ra * sinf(6.2831853f * u2), rb * cosf(6.2831853f * u4_), rb * sinf(6.2831853f * u4_)};
    for (int e = 0; e < 4; e++)
      if (i * 4 + e < n) y[i * 4 + e] = f2bf(z[e]);
  }
  rng_finish(state, off, done);
}

__global__ void rng_labels_kernel(int* __restrict__ y, long n, int n_labels, unsigned long long* __restrict__ state, unsigned* __restrict__ done) {
  const unsigned long long seed = state[0], off = state[1];
  const long n4 = (n + 3) >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u4 r = philox4x32_10((unsigned long long)i, off, seed);
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 4; e++)
      if (i * 4 + e < n) {  // tf.cast(tf.random_uniform([n]) * 10, tf.int32)   gan_cifar_resnet.py:467
        int lb = (int)(u01(v[e]) * (float)n_labels);
        y[i * 4 + e] = lb >= n_labels ? n_labels - 1 : lb;
      }
  }
  rng_finish(state, off, done);
}

// What a generator pass needs before its first layer, in ONE launch: (optional) the fake labels (rng_labels_kernel's draw at the
// stream's offset), the noise (rng_normal_kernel's draw at the NEXT offset -- the offset after labels, or the current one without
// them) and a zero fill (the pass's statistics arena) as three block ranges; the last block to finish advances the offset by the
// number of draws.  Bit-identical to the separate launches.
__global__ void generator_feed_kernel(int* __restrict__ labels, long n_lab, int n_labels, bf16* __restrict__ y, long n, float* __restrict__ zero, long zero_n,
                                      unsigned long long* __restrict__ state, unsigned* __restrict__ done, int bl, int bn) {
  const unsigned long long seed = state[0], off0 = state[1];
  const int b = blockIdx.x;
  if (b < bl) {
    const long n4 = (n_lab + 3) >> 2;
    for (long i = b * (long)blockDim.x + threadIdx.x; i < n4; i += (long)bl * blockDim.x) {
      const u4 r = philox4x32_10((unsigned long long)i, off0, seed);
      const unsigned v[4] = {r.x, r.y, r.z, r.w};
      for (int e = 0; e < 4; e++)
        if (i * 4 + e < n_lab) {
          int lb = (int)(u01(v[e]) * (float)n_labels);
          labels[i * 4 + e] = lb >= n_labels ? n_labels - 1 : lb;
        }
    }
  } else if (b < bl + bn) {
    const unsigned long long off = off0 + (bl > 0 ? 1 : 0);
    const long n4 = (n + 3) >> 2;
    for (long i = (b - bl) * (long)blockDim.x + threadIdx.x; i < n4; i += (long)bn * blockDim.x) {
      const u4 r = philox4x32_10((unsigned long long)i, off, seed);
      const float u1 = 1.f - u01(r.x), u2 = u01(r.y), u3 = 1.f - u01(r.z), u4_ = u01(r.w);  // (0,1]
      const float ra = sqrtf(-2.f * logf(u1)), rb = sqrtf(-2.f * logf(u3));
      const float z[4] = {ra * cosf(6.2831853f * u2), ra * sinf(6.2831853f * u2), rb * cosf(6.2831853f * u4_), rb * sinf(6.2831853f * u4_)};
      for (int e = 0; e < 4; e++)
        if (i * 4 + e < n) y[i * 4 + e] = f2bf(z[e]);
    }
  } else {
    const int bz = gridDim.x - bl - bn;
    const long z4 = zero_n >> 2;
    const f32x4 zz = {0.f, 0.f, 0.f, 0.f};
    for (long i = (b - bl - bn) * (long)blockDim.x + threadIdx.x; i < z4; i += (long)bz * blockDim.x) reinterpret_cast<f32x4*>(zero)[i] = zz;
    if (b == bl + bn)
      for (long i = (z4 << 2) + threadIdx.x; i < zero_n; i += blockDim.x) zero[i] = 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(done, 1u) == gridDim.x - 1) {
    *done = 0u;
    state[1] = off0 + (bl > 0 ? 2 : 1);
  }
}

// uint8 CHW-planar [B,3072] -> bf16 HWC [B,32,32,3]:  2*(x/256 - .5) + U[0,1/128)   (gan_cifar_resnet.py:334-337)
__global__ void preprocess_kernel(const unsigned char* __restrict__ data, bf16* __restrict__ y, int B, const unsigned long long* __restrict__ state) {
  const unsigned long long seed = state[0], off = state[1];
  const long n = (long)B * 3072, n4 = n >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u4 r = philox4x32_10((unsigned long long)i, off, seed);
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 4; e++) {
      const long o = i * 4 + e;            // output index in HWC order
      const int b = (int)(o / 3072), rem = (int)(o - (long)b * 3072);
      const int c = rem % 3, hw = rem / 3;
      const float px = (float)data[(long)b * 3072 + c * 1024 + hw];
      y[o] = f2bf(2.f * (px / 256.f - .5f) + u01(v[e]) * (1.f / 128.f));
    }
  }
}

static int rng_advance(unsigned long long* state, unsigned long long inc, hipStream_t s) {
  hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, s, state, inc);
  GANK_LAUNCH_OK("rng_advance");
  return 0;
}

static inline dim3 rgrid(long n4) {
  long g = (n4 + 255) / 256;
  if (g > 2048) g = 2048;
  if (g < 1) g = 1;
  return dim3((unsigned)g);
}

extern "C" int gank_rng_normal_bf16(void* y, long n, uint64_t* rng_state, void* stream) {
  GANK_REQUIRE(y && rng_state && n > 0, "rng_normal: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  unsigned* done = rng_ticket_word();
  if (!done) return gank_set_error("rng_normal: ticket words not found");
  hipLaunchKernelGGL(rng_normal_kernel, rgrid((n + 3) / 4), dim3(256), 0, s, (bf16*)y, n, (unsigned long long*)rng_state, done);
  GANK_LAUNCH_OK("rng_normal");
  return 0;
}
__global__ void rng_uniform_kernel(float* __restrict__ y, long n, const unsigned long long* __restrict__ state) {
  const unsigned long long seed = state[0], off = state[1];
  const long n4 = (n + 3) >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const u4 r = philox4x32_10((unsigned long long)i, off, seed);
    const unsigned v[4] = {r.x, r.y, r.z, r.w};
    for (int e = 0; e < 4; e++)
      if (i * 4 + e < n) y[i * 4 + e] = u01(v[e]);      // tf.random_uniform(minval=0, maxval=1)   ACGAN/train.py:99
  }
}
extern "C" int gank_rng_uniform_f32(float* y, long n, uint64_t* rng_state, void* stream) {
  GANK_REQUIRE(y && rng_state && n > 0, "rng_uniform: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(rng_uniform_kernel, rgrid((n + 3) / 4), dim3(256), 0, s, y, n, (const unsigned long long*)rng_state);
  GANK_LAUNCH_OK("rng_uniform");
  return rng_advance((unsigned long long*)rng_state, 1, s);
}
extern "C" int gank_rng_labels(int32_t* y, long n, int n_labels, uint64_t* rng_state, void* stream) {
  GANK_REQUIRE(y && rng_state && n > 0 && n_labels > 0, "rng_labels: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  unsigned* done = rng_ticket_word();
  if (!done) return gank_set_error("rng_labels: ticket words not found");
  hipLaunchKernelGGL(rng_labels_kernel, rgrid((n + 3) / 4), dim3(256), 0, s, y, n, n_labels, (unsigned long long*)rng_state, done);
  GANK_LAUNCH_OK("rng_labels");
  return 0;
}
// labels (int32 [n_lab], or NULL with n_lab = 0: no label draw) ~ gank_rng_labels, noise (bf16 [n]) ~ gank_rng_normal_bf16 drawn behind
// them, zero_buf (fp32 [zero_n], or NULL with 0) cleared: the three launches in front of a generator pass (gan_cifar_resnet.py:240,467)
// as one, bit-identical outputs and stream offset
extern "C" int gank_generator_feed(int32_t* labels, long n_lab, int n_labels, void* noise, long n, float* zero_buf, long zero_n, uint64_t* rng_state,
                                   void* stream) {
  GANK_REQUIRE(noise && n > 0 && rng_state && (n_lab == 0 || (labels && n_labels > 0)) && (zero_n == 0 || zero_buf) && n_lab >= 0 && zero_n >= 0,
               "generator_feed: bad arguments");
  GANK_REQUIRE(zero_n == 0 || ((uintptr_t)zero_buf & 15) == 0, "generator_feed: the zero buffer must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  unsigned* done = rng_ticket_word();
  if (!done) return gank_set_error("generator_feed: ticket words not found");
  const int bl = n_lab > 0 ? (int)rgrid((n_lab + 3) / 4).x : 0, bn = (int)rgrid((n + 3) / 4).x;
  int bz = zero_n > 0 ? (int)((zero_n / 4 + 1023) / 1024) : 0;      // four 16-byte stores per thread
  if (bz > 1024) bz = 1024;
  if (zero_n > 0 && bz < 1) bz = 1;
  hipLaunchKernelGGL(generator_feed_kernel, dim3(bl + bn + bz), dim3(256), 0, s, labels, n_lab, n_labels, (bf16*)noise, n, zero_buf, zero_n,
                     (unsigned long long*)rng_state, done, bl, bn);
  GANK_LAUNCH_OK("generator_feed");
  return 0;
}
extern "C" int gank_preprocess_real(const uint8_t* data, void* y, uint64_t* rng_state, int B, void* stream) {
  GANK_REQUIRE(data && y && rng_state && B > 0, "preprocess_real: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(preprocess_kernel, rgrid((long)B * 768), dim3(256), 0, s, data, (bf16*)y, B, (const unsigned long long*)rng_state);
  GANK_LAUNCH_OK("preprocess_real");
  return rng_advance((unsigned long long*)rng_state, 1, s);
}

// ---- the critic's feed for one update, in ONE launch (gan_cifar_resnet.py:334-338,361,616-620) ---------------
// both[0:B] = 2*(real/256 - .5) + U[0,1/128) for slot *slot of the iteration's real batches (CHW rows -> HWC),
// both[B:2B] = the generator output kept for that slot, labels2 = the slot's labels twice; then the slot counter
// and the RNG offset advance.  Replaces 3 staging copies, the preprocess launch, its RNG advance and two concats.
// The counters are advanced by the LAST workgroup to finish (every workgroup has read them by then).
__global__ __launch_bounds__(256) void critic_feed_kernel(CriticFeedArgs a) { critic_feed_block(a, blockIdx.x); }

extern "C" int gank_critic_feed(const uint8_t* real_all, const int32_t* labels_all, const void* fake_all, void* both,
                                int32_t* labels2, int32_t* slot, uint64_t* rng_state, uint32_t* done_counter, int B, int n_slots,
                                void* stream) {
  GANK_REQUIRE(real_all && labels_all && fake_all && both && labels2 && slot && rng_state && done_counter && B > 0 && n_slots > 0,
               "critic_feed: bad arguments");
  CriticFeedArgs a{real_all, labels_all, (const bf16*)fake_all, (bf16*)both, labels2, slot, (unsigned long long*)rng_state, done_counter, B, n_slots,
                   critic_feed_blocks(B)};
  hipLaunchKernelGGL(critic_feed_kernel, dim3(a.blocks), dim3(256), 0, (hipStream_t)stream, a);
  GANK_LAUNCH_OK("critic_feed");
  return 0;
}

// ------------------------------------------------------------------------------------------------
// debug: dump what ds_read_b64_tr_b16 returns for LDS image lds[i] = i (16-bit), lane l reading at
// byte address 8*l.  out[l*4+e] = element e delivered to lane l.
// ------------------------------------------------------------------------------------------------
__global__ void tr_probe_kernel(int* out) {
  __shared__ __attribute__((aligned(16))) short img[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) img[i] = (short)i;
  __syncthreads();
  const s16x4 t = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(img + threadIdx.x * 4));
  for (int e = 0; e < 4; e++) out[threadIdx.x * 4 + e] = t[e];
}
extern "C" int gank_debug_tr_probe(int32_t* out, void* stream) {
  GANK_REQUIRE(out, "tr_probe: null pointer");
  hipLaunchKernelGGL(tr_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
  GANK_LAUNCH_OK("tr_probe");
  return 0;
}
