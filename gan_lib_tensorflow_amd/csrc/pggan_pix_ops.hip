// Small HBM / latency bound operators of the PGGAN (BASELINE.json config 4) and Pix2Pix (config 5) paths:
//   axpby           y = alpha*a + beta*b                      fade-in blend (PGGAN/model_nvidia.py:116,206), scaling
//   minibatch_std   (PGGAN/model_nvidia.py:20-29)             forward and backward
//   resize_bilinear tf.image.resize_images (PGGAN/train.py:88-92; Pix2Pix/train.py) legacy semantics (no half-pixel offset)
//   concat_c        tf.concat(axis=3) of two NHWC tensors      U-Net skip connections (Pix2Pix/networks.py:470-520) + split
//   dropout         tf.nn.dropout(keep_prob) from the device RNG (networks.py:480-500), mask kept for the backward pass
//   abs_diff_mean   mean |a - b| and its gradient              L1 loss (Pix2Pix/train.py:510-512)
#include "gank_common.h"

// grid of a GRID-STRIDE kernel: capped.  Kernels that take ONE element per thread and return (`if (i >= n) return;`) must be
// launched with g_all -- capped, they silently left everything behind the first 4096 x 256 elements unwritten (found by the
// batch-16 Pix2Pix generator test: NaN garbage in every tensor of more than 8.4 M elements that passed through them)
static inline dim3 g_all(long n) {
  long b = (n + 255) / 256;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}
static inline dim3 g1(long n, long cap = 4096) {
  long b = (n + 255) / 256;
  if (b > cap) b = cap;
  if (b < 1) b = 1;
  return dim3((unsigned)b);
}

__global__ void axpby_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, float alpha, float beta, bf16* __restrict__ y, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f2bf(alpha * bf2f(a[i]) + (b ? beta * bf2f(b[i]) : 0.f));
}
extern "C" int gank_axpby_bf16(const void* a, const void* b, float alpha, float beta, void* y, long n, void* stream) {
  GANK_REQUIRE(a && y && n > 0, "axpby: bad arguments");
  hipLaunchKernelGGL(axpby_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, alpha, beta, (bf16*)y, n);
  GANK_LAUNCH_OK("axpby");
  return 0;
}

// the same with the weight read from device memory (a captured train step replays with the fade-in weight of the day):
// mode 0: y = (1 - alpha) a + alpha b;  mode 1: y = (1 - alpha) a;  mode 2: y = alpha a      (the two gradients of mode 0)
__global__ void blend_dev_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, const float* __restrict__ alpha, bf16* __restrict__ y, long n, int mode) {
  const float al = alpha[0];
  const float wa = mode == 2 ? al : 1.f - al, wb = mode == 0 ? al : 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = f2bf(wa * bf2f(a[i]) + (mode == 0 ? wb * bf2f(b[i]) : 0.f));
}
extern "C" int gank_blend_dev(const void* a, const void* b, const float* alpha, void* y, long n, int mode, void* stream) {
  GANK_REQUIRE(a && alpha && y && n > 0 && mode >= 0 && mode <= 2 && (mode != 0 || b), "blend_dev: bad arguments");
  hipLaunchKernelGGL(blend_dev_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, alpha, (bf16*)y, n, mode);
  GANK_LAUNCH_OK("blend_dev");
  return 0;
}

// ---- minibatch_std: y = concat(x, s), s = mean over (h,w,c) of sqrt(var_batch(x) + 1e-8) --------------------------------
// ws: fp32 [R + 1] (R = HW*C): per-position sqrt(v + eps), then the scalar at ws[R]
__global__ void mbstd_pos_kernel(const bf16* __restrict__ x, float* __restrict__ ws, int B, long R) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= R) return;
  float m = 0.f;
  for (int b = 0; b < B; b++) m += bf2f(x[b * R + i]);
  m /= (float)B;
  float v = 0.f;
  for (int b = 0; b < B; b++) { const float d = bf2f(x[b * R + i]) - m; v += d * d; }
  ws[i] = sqrtf(v / (float)B + 1e-8f);
}
__global__ __launch_bounds__(256) void mbstd_mean_kernel(float* __restrict__ ws, long R) {
  __shared__ float red[16];
  float acc = 0.f;
  for (long i = threadIdx.x; i < R; i += blockDim.x) acc += ws[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) ws[R] = tot / (float)R;
}
__global__ void mbstd_concat_kernel(const bf16* __restrict__ x, const float* __restrict__ ws, bf16* __restrict__ y, long pixels, int C, long R) {
  const long total = pixels * (C + 1);
  const bf16 s = f2bf(ws[R]);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / (C + 1);
    const int c = (int)(i - p * (C + 1));
    y[i] = c < C ? x[p * C + c] : s;
  }
}
extern "C" int gank_minibatch_std_fwd(const void* x, void* y, float* ws, int B, int HW, int C, void* stream) {
  GANK_REQUIRE(x && y && ws && B > 0 && HW > 0 && C > 0, "minibatch_std_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long R = (long)HW * C;
  hipLaunchKernelGGL(mbstd_pos_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, (const bf16*)x, ws, B, R);
  hipLaunchKernelGGL(mbstd_mean_kernel, dim3(1), dim3(256), 0, s, ws, R);
  hipLaunchKernelGGL(mbstd_concat_kernel, g1((long)B * HW * (C + 1)), dim3(256), 0, s, (const bf16*)x, ws, (bf16*)y, (long)B * HW, C, R);
  GANK_LAUNCH_OK("minibatch_std_fwd");
  return 0;
}
// dx[b,i] = dy[b,i(:C)] + ds/(R*B) * (x[b,i] - m_i) / sqrt(v_i + eps),   ds = sum over (b,hw) of dy[b,hw,C]
__global__ __launch_bounds__(256) void mbstd_ds_kernel(const bf16* __restrict__ dy, float* __restrict__ ws, long pixels, int C, long R) {
  __shared__ float red[16];
  float acc = 0.f;
  for (long p = threadIdx.x; p < pixels; p += blockDim.x) acc += bf2f(dy[p * (C + 1) + C]);
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) ws[R + 1] = tot;
}
__global__ void mbstd_bwd_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ x, const float* __restrict__ ws, bf16* __restrict__ dx,
                                 int B, int HW, int C, long R) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= R) return;
  float m = 0.f;
  for (int b = 0; b < B; b++) m += bf2f(x[b * R + i]);
  m /= (float)B;
  const float k = ws[R + 1] / ((float)R * (float)B) / ws[i];
  const long hw = i / C;
  const int c = (int)(i - hw * C);
  for (int b = 0; b < B; b++)
    dx[b * R + i] = f2bf(bf2f(dy[((long)b * HW + hw) * (C + 1) + c]) + k * (bf2f(x[b * R + i]) - m));
}
extern "C" int gank_minibatch_std_bwd(const void* dy, const void* x, float* ws, void* dx, int B, int HW, int C, void* stream) {
  GANK_REQUIRE(dy && x && ws && dx && B > 0 && HW > 0 && C > 0, "minibatch_std_bwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long R = (long)HW * C;
  hipLaunchKernelGGL(mbstd_ds_kernel, dim3(1), dim3(256), 0, s, (const bf16*)dy, ws, (long)B * HW, C, R);
  hipLaunchKernelGGL(mbstd_bwd_kernel, dim3((unsigned)((R + 255) / 256)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)x, ws, (bf16*)dx, B, HW, C, R);
  GANK_LAUNCH_OK("minibatch_std_bwd");
  return 0;
}

// ---- tf.image.resize_images(method=BILINEAR, align_corners=False), TF1 legacy sampling: src = dst * (in / out) --------
__global__ void resize_bilinear_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo, int C) {
  const long total = (long)N * Ho * Wo * C;
  const float sy = (float)Hi / (float)Ho, sx = (float)Wi / (float)Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long t = i / C;
    const int ox = (int)(t % Wo); t /= Wo;
    const int oy = (int)(t % Ho);
    const int n = (int)(t / Ho);
    const float fy = oy * sy, fx = ox * sx;
    const int y0 = (int)floorf(fy), x0 = (int)floorf(fx);
    const int y1 = y0 + 1 < Hi ? y0 + 1 : Hi - 1, x1 = x0 + 1 < Wi ? x0 + 1 : Wi - 1;
    const float wy = fy - y0, wx = fx - x0;
    const bf16* base = x + (long)n * Hi * Wi * C + c;
    const float v00 = bf2f(base[((long)y0 * Wi + x0) * C]), v01 = bf2f(base[((long)y0 * Wi + x1) * C]);
    const float v10 = bf2f(base[((long)y1 * Wi + x0) * C]), v11 = bf2f(base[((long)y1 * Wi + x1) * C]);
    const float top = v00 + (v01 - v00) * wx, bot = v10 + (v11 - v10) * wx;
    y[i] = f2bf(top + (bot - top) * wy);
  }
}
extern "C" int gank_resize_bilinear(const void* x, void* y, int N, int Hi, int Wi, int Ho, int Wo, int C, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && Hi > 0 && Wi > 0 && Ho > 0 && Wo > 0 && C > 0, "resize_bilinear: bad arguments");
  hipLaunchKernelGGL(resize_bilinear_kernel, g1((long)N * Ho * Wo * C), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, N, Hi, Wi, Ho, Wo, C);
  GANK_LAUNCH_OK("resize_bilinear");
  return 0;
}

// ---- channel concat / split of NHWC tensors --------------------------------------------------------------------------
__global__ void concat_c_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, bf16* __restrict__ y, long pixels, int Ca, int Cb) {
  const int C = Ca + Cb;
  const long total = pixels * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    y[i] = c < Ca ? a[p * Ca + c] : b[p * Cb + (c - Ca)];
  }
}
__global__ void split_c_kernel(const bf16* __restrict__ y, bf16* __restrict__ a, bf16* __restrict__ b, long pixels, int Ca, int Cb) {
  const int C = Ca + Cb;
  const long total = pixels * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long p = i / C;
    const int c = (int)(i - p * C);
    if (c < Ca) { if (a) a[p * Ca + c] = y[i]; }
    else if (b) b[p * Cb + (c - Ca)] = y[i];
  }
}
// the same, 16 bytes per lane, when both channel counts are multiples of 8 (the U-Net skips: 34 us -> HBM speed)
__global__ void concat_c8_kernel(const u32x4* __restrict__ a, const u32x4* __restrict__ b, u32x4* __restrict__ y, long total8, int Ga, int Gb) {
  const int G = Ga + Gb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const long p = i / G;
    const int g = (int)(i - p * G);
    y[i] = g < Ga ? a[p * Ga + g] : b[p * Gb + (g - Ga)];
  }
}
__global__ void split_c8_kernel(const u32x4* __restrict__ y, u32x4* __restrict__ a, u32x4* __restrict__ b, long total8, int Ga, int Gb) {
  const int G = Ga + Gb;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const long p = i / G;
    const int g = (int)(i - p * G);
    if (g < Ga) { if (a) a[p * Ga + g] = y[i]; }
    else if (b) b[p * Gb + (g - Ga)] = y[i];
  }
}
extern "C" int gank_concat_channels(const void* a, const void* b, void* y, long pixels, int Ca, int Cb, void* stream) {
  GANK_REQUIRE(a && b && y && pixels > 0 && Ca > 0 && Cb > 0, "concat_channels: bad arguments");
  if (Ca % 8 == 0 && Cb % 8 == 0) {
    const long total8 = pixels * ((Ca + Cb) / 8);
    hipLaunchKernelGGL(concat_c8_kernel, g1(total8), dim3(256), 0, (hipStream_t)stream, (const u32x4*)a, (const u32x4*)b, (u32x4*)y, total8, Ca / 8, Cb / 8);
    GANK_LAUNCH_OK("concat_channels");
    return 0;
  }
  hipLaunchKernelGGL(concat_c_kernel, g1(pixels * (Ca + Cb)), dim3(256), 0, (hipStream_t)stream, (const bf16*)a, (const bf16*)b, (bf16*)y, pixels, Ca, Cb);
  GANK_LAUNCH_OK("concat_channels");
  return 0;
}
extern "C" int gank_split_channels(const void* y, void* a, void* b, long pixels, int Ca, int Cb, void* stream) {
  GANK_REQUIRE(y && (a || b) && pixels > 0 && Ca > 0 && Cb > 0, "split_channels: bad arguments");
  if (Ca % 8 == 0 && Cb % 8 == 0) {
    const long total8 = pixels * ((Ca + Cb) / 8);
    hipLaunchKernelGGL(split_c8_kernel, g1(total8), dim3(256), 0, (hipStream_t)stream, (const u32x4*)y, (u32x4*)a, (u32x4*)b, total8, Ca / 8, Cb / 8);
    GANK_LAUNCH_OK("split_channels");
    return 0;
  }
  hipLaunchKernelGGL(split_c_kernel, g1(pixels * (Ca + Cb)), dim3(256), 0, (hipStream_t)stream, (const bf16*)y, (bf16*)a, (bf16*)b, pixels, Ca, Cb);
  GANK_LAUNCH_OK("split_channels");
  return 0;
}

// ---- L1 loss: mean |a - b|, gradient wrt a ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void l1_part_kernel(const bf16* __restrict__ a, const bf16* __restrict__ b, float* __restrict__ part, float* __restrict__ dl32, long n) {
  __shared__ float red[16];
  float acc = 0.f;
  const float inv = 1.f / (float)n;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float d = bf2f(a[i]) - bf2f(b[i]);
    acc += fabsf(d);
    dl32[i] = d > 0.f ? inv : (d < 0.f ? -inv : 0.f);
  }
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = tot * inv;
}
__global__ __launch_bounds__(256) void sum_parts_kernel(const float* __restrict__ part, float* __restrict__ out, int n) {
  __shared__ float red[16];
  float acc = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += part[i];
  const float tot = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = tot;
}
extern "C" int gank_l1_loss(const void* a, const void* b, float* loss, float* dl32, float* ws, long n, void* stream) {
  GANK_REQUIRE(a && b && loss && dl32 && ws && n > 0, "l1_loss: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid = g1(n, 1024);
  hipLaunchKernelGGL(l1_part_kernel, grid, dim3(256), 0, s, (const bf16*)a, (const bf16*)b, ws, dl32, n);
  hipLaunchKernelGGL(sum_parts_kernel, dim3(1), dim3(256), 0, s, ws, loss, (int)grid.x);
  GANK_LAUNCH_OK("l1_loss");
  return 0;
}

// ---- dropout: y = x * mask / keep, mask ~ Bernoulli(keep) from Philox (same generator as loss_opt.hip) ---------------------
__device__ __forceinline__ unsigned philox_word(unsigned long long ctr, unsigned long long off, unsigned long long seed) {
  unsigned c0 = (unsigned)ctr, c1 = (unsigned)(ctr >> 32), c2 = (unsigned)off, c3 = (unsigned)(off >> 32);
  unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; r++) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}
__global__ void dropout_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, unsigned char* __restrict__ mask, long n, float keep,
                               const unsigned long long* __restrict__ state) {
  const unsigned long long seed = state[0], off = state[1];
  const float inv = 1.f / keep;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float u = (float)(philox_word((unsigned long long)i, off, seed) >> 8) * (1.0f / 16777216.0f);
    const unsigned char m = u < keep ? 1 : 0;
    mask[i] = m;
    y[i] = f2bf(m ? bf2f(x[i]) * inv : 0.f);
  }
}
__global__ void dropout_advance_kernel(unsigned long long* state) { state[1] += 1; }
__global__ void dropout_bwd_kernel(const bf16* __restrict__ dy, const unsigned char* __restrict__ mask, bf16* __restrict__ dx, long n, float keep) {
  const float inv = 1.f / keep;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = f2bf(mask[i] ? bf2f(dy[i]) * inv : 0.f);
}
extern "C" int gank_dropout_fwd(const void* x, void* y, uint8_t* mask, long n, float keep, uint64_t* rng_state, void* stream) {
  GANK_REQUIRE(x && y && mask && rng_state && n > 0 && keep > 0.f && keep <= 1.f, "dropout_fwd: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dropout_kernel, g1(n), dim3(256), 0, s, (const bf16*)x, (bf16*)y, mask, n, keep, (const unsigned long long*)rng_state);
  hipLaunchKernelGGL(dropout_advance_kernel, dim3(1), dim3(1), 0, s, (unsigned long long*)rng_state);
  GANK_LAUNCH_OK("dropout_fwd");
  return 0;
}
extern "C" int gank_dropout_bwd(const void* dy, const uint8_t* mask, void* dx, long n, float keep, void* stream) {
  GANK_REQUIRE(dy && mask && dx && n > 0 && keep > 0.f, "dropout_bwd: bad arguments");
  hipLaunchKernelGGL(dropout_bwd_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, (const bf16*)dy, mask, (bf16*)dx, n, keep);
  GANK_LAUNCH_OK("dropout_bwd");
  return 0;
}


// ------------------------------------------------------------------------------------------------
// Inception-v3 classifier of the Inception-score harness (common/inception/inception_score.py:29-47 runs tfgan's frozen
// 2015 Inception graph): the two operators it needs beside convolutions -- general 2-D pooling and the branch concat.
//   pool2d: max or average over a k x k window, stride 1 or 2, `pad` leading rows / columns (TF SAME for odd k: (k-1)/2;
//           VALID: 0); the average divides by the number of IN-IMAGE elements (tf.nn.avg_pool with SAME excludes the padding).
//   Both write into a channel slice [c_off, c_off + C) of a wider output [.., Cy] (tf.concat(axis=3) of a block's branches
//   without a concat pass); relu_to_channels is the conv's ReLU placed there.
// ------------------------------------------------------------------------------------------------
__global__ void pool2d_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long total8, int H, int W, int C, int Ho, int Wo, int k, int stride,
                              int pad, int mode, int Cy, int c_off) {
  const int cg = C >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    long p = i / cg;
    const int ow = (int)(p % Wo); p /= Wo;
    const int oh = (int)(p % Ho);
    const int n = (int)(p / Ho);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = mode == 0 ? -3.0e38f : 0.f;
    int cnt = 0;
    for (int dh = 0; dh < k; dh++) {
      const int ih = oh * stride + dh - pad;
      if ((unsigned)ih >= (unsigned)H) continue;
      for (int dw = 0; dw < k; dw++) {
        const int iw = ow * stride + dw - pad;
        if ((unsigned)iw >= (unsigned)W) continue;
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + (((long)n * H + ih) * W + iw) * C + g * 8);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = mode == 0 ? fmaxf(acc[e], bf2f(v[e])) : acc[e] + bf2f(v[e]);
        cnt++;
      }
    }
    bf16x8 o;
#pragma unroll
    for (int e = 0; e < 8; e++) o[e] = f2bf(mode == 0 ? acc[e] : acc[e] / (float)max(cnt, 1));
    *reinterpret_cast<bf16x8*>(y + (((long)n * Ho + oh) * Wo + ow) * Cy + c_off + g * 8) = o;
  }
}
extern "C" int gank_pool2d(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad, int mode, int Cy, int c_off,
                           void* stream) {
  GANK_REQUIRE(x && y && N > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0 && k >= 1 && (stride == 1 || stride == 2) && pad >= 0 && pad < k && (mode == 0 || mode == 1),
               "pool2d: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && Cy % 8 == 0 && c_off % 8 == 0 && c_off + C <= Cy, "pool2d: channels %d into [%d, %d) of %d: multiples of 8 inside the output", C, c_off, c_off + C, Cy);
  GANK_REQUIRE((Ho - 1) * stride - pad < H && (Wo - 1) * stride - pad < W, "pool2d: output %dx%d reaches past the %dx%d input", Ho, Wo, H, W);
  const long total8 = (long)N * Ho * Wo * (C / 8);
  hipLaunchKernelGGL(pool2d_kernel, g1(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, total8, H, W, C, Ho, Wo, k, stride, pad, mode, Cy, c_off);
  GANK_LAUNCH_OK("pool2d");
  return 0;
}

__global__ void relu_to_channels_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long total8, int C, int Cy, int c_off) {
  const int cg = C >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long p = i / cg;
    const u32x4 v = relu_bf16x8(*reinterpret_cast<const u32x4*>(x + p * C + g * 8));
    *reinterpret_cast<u32x4*>(y + p * Cy + c_off + g * 8) = v;
  }
}
extern "C" int gank_relu_to_channels(const void* x, void* y, long pixels, int C, int Cy, int c_off, void* stream) {
  GANK_REQUIRE(x && y && pixels > 0 && C > 0, "relu_to_channels: bad arguments");
  GANK_REQUIRE(C % 8 == 0 && Cy % 8 == 0 && c_off % 8 == 0 && c_off + C <= Cy, "relu_to_channels: channels %d into [%d, %d) of %d: multiples of 8 inside the output", C, c_off, c_off + C, Cy);
  const long total8 = pixels * (C / 8);
  hipLaunchKernelGGL(relu_to_channels_kernel, g1(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, total8, C, Cy, c_off);
  GANK_LAUNCH_OK("relu_to_channels");
  return 0;
}

// im2col of a narrow-channel image (Pix2Pix: the 4x4 stride-2 first layers on 3- and 6-channel inputs, networks.py:335-342, :474-486):
// y[n, oy, ox, tap * Cin + c] = x[n, oy * stride - pad + ky, ox * stride - pad + kx, c]  (zero outside the image and for columns
// >= k * k * Cin).  With it the filter gradient of such a layer is the filter gradient of a 1x1 conv with Kpad input channels --
// an MFMA kernel at HBM speed instead of the scalar-gather kernel (0.8 ms for 1.6 GFLOP at 512 x 512, batch 4).
__global__ void im2col_narrow_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, long total8, int Hin, int Win, int Cin, int Ho, int Wo,
                                     int ks, int stride, int pad, int Kpad) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int g8 = Kpad >> 3;
  const long p = i / g8;
  const int k0 = (int)(i - p * g8) * 8;
  const int ox = (int)(p % Wo);
  const long t = p / Wo;
  const int oy = (int)(t % Ho);
  const long n = t / Ho;
  const int ktot = ks * ks * Cin;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int k = k0 + e;
    float v = 0.f;
    if (k < ktot) {
      const int tap = k / Cin, c = k - tap * Cin;
      const int iy = oy * stride - pad + tap / ks, ix = ox * stride - pad + tap % ks;
      if ((unsigned)iy < (unsigned)Hin && (unsigned)ix < (unsigned)Win) v = bf2f(x[((n * Hin + iy) * Win + ix) * Cin + c]);
    }
    o[e] = f2bf(v);
  }
  *reinterpret_cast<bf16x8*>(y + i * 8) = o;
}
extern "C" int gank_im2col_narrow(const void* x, void* y, int N, int Hin, int Win, int Cin, int Ho, int Wo, int ksize, int stride, int pad,
                                  int Kpad, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && Hin > 0 && Win > 0 && Cin > 0 && Ho > 0 && Wo > 0, "im2col_narrow: bad arguments");
  GANK_REQUIRE(ksize >= 1 && ksize <= 7 && stride >= 1 && pad >= 0 && Kpad % 8 == 0 && Kpad >= ksize * ksize * Cin,
               "im2col_narrow: k=%d stride=%d pad=%d Kpad=%d (needs Kpad %% 8 == 0 and >= k*k*Cin = %d)", ksize, stride, pad, Kpad, ksize * ksize * Cin);
  const long total8 = (long)N * Ho * Wo * (Kpad / 8);
  hipLaunchKernelGGL(im2col_narrow_kernel, g_all(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, total8, Hin, Win, Cin, Ho, Wo,
                     ksize, stride, pad, Kpad);
  GANK_LAUNCH_OK("im2col_narrow");
  return 0;
}

// A conv with very few output channels behind a 2x nearest-neighbour upsample (Pix2Pix decoder_1: relu -> upsample -> 4x4 SAME ->
// 3 channels -> tanh at 512 x 512, networks.py:424-452) as a 1x1 conv at LOW resolution plus a tap gather:
//   Z[n, q, t * Cout + co] = sum_ci relu(x[n, q, ci]) W[t, ci, co]            (an MFMA 1x1 conv with k*k*Cout <= Zc outputs)
//   y[n, p, co] = tanh(b[co] + sum_t Z[n, (p + d(t)) >> 1, t * Cout + co]),   d(t) = (ky - pad, kx - pad), taps outside the image skipped
// instead of a k*k-tap conv whose 3 outputs are padded to a 32-row MFMA tile (1.3 ms at batch 4: 10x the padding, 4x the
// upsampled taps).  The backward pass gathers the same way: col[n, q, t * Cout + co] = sum of g[n, p, co] over the <= 4 pixels p
// with (p + d(t)) >> 1 == q; input and filter gradient are then the 1x1 conv's own (gank_conv2d_dgrad / _wgrad on col).
__global__ void tap_gather_up2_kernel(const bf16* __restrict__ Z, const float* __restrict__ bias, bf16* __restrict__ y, long pixels,
                                      int h, int w, int ks, int pad, int Cout, int Zc, int tanh_out) {
  const long p = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (p >= pixels) return;
  const int W2 = 2 * w, H2 = 2 * h;
  const int px = (int)(p % W2);
  const long t0 = p / W2;
  const int py = (int)(t0 % H2);
  const long n = t0 / H2;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < ks; ky++) {
    const int iy = py + ky - pad;
    if ((unsigned)iy >= (unsigned)H2) continue;
    for (int kx = 0; kx < ks; kx++) {
      const int ix = px + kx - pad;
      if ((unsigned)ix >= (unsigned)W2) continue;
      const bf16* z = Z + ((n * h + (iy >> 1)) * w + (ix >> 1)) * Zc + (ky * ks + kx) * Cout;
      for (int co = 0; co < Cout; co++) acc[co] += bf2f(z[co]);
    }
  }
  for (int co = 0; co < Cout; co++) {
    float v = acc[co] + (bias ? bias[co] : 0.f);
    if (tanh_out) v = tanhf(v);
    y[p * Cout + co] = f2bf(v);
  }
}

__global__ void tap_scatter_up2_kernel(const bf16* __restrict__ g, bf16* __restrict__ col, long total8, int h, int w, int ks, int pad,
                                       int Cout, int Zc) {
  const long i = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int g8 = Zc >> 3;
  const long q = i / g8;
  const int j0 = (int)(i - q * g8) * 8;
  const int qx = (int)(q % w);
  const long t0 = q / w;
  const int qy = (int)(t0 % h);
  const long n = t0 / h;
  const int W2 = 2 * w, H2 = 2 * h;
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; e++) {
    const int j = j0 + e;
    float v = 0.f;
    if (j < ks * ks * Cout) {
      const int t = j / Cout, co = j - t * Cout;
      const int ky = t / ks, kx = t - ky * ks;
#pragma unroll
      for (int a = 0; a < 2; a++) {
        const int py = 2 * qy + a - (ky - pad);
        if ((unsigned)py >= (unsigned)H2) continue;
#pragma unroll
        for (int b = 0; b < 2; b++) {
          const int px = 2 * qx + b - (kx - pad);
          if ((unsigned)px >= (unsigned)W2) continue;
          v += bf2f(g[((n * H2 + py) * W2 + px) * Cout + co]);
        }
      }
    }
    o[e] = f2bf(v);
  }
  *reinterpret_cast<bf16x8*>(col + i * 8) = o;
}

extern "C" int gank_tap_gather_up2(const void* Z, const float* bias, void* y, int N, int h, int w, int ksize, int pad, int Cout, int Zc,
                                   int tanh_out, void* stream) {
  GANK_REQUIRE(Z && y && N > 0 && h > 0 && w > 0, "tap_gather_up2: bad arguments");
  GANK_REQUIRE(ksize >= 1 && ksize <= 7 && pad >= 0 && Cout >= 1 && Cout <= 4 && ksize * ksize * Cout <= Zc,
               "tap_gather_up2: k=%d Cout=%d needs Cout <= 4 and k*k*Cout <= Zc=%d", ksize, Cout, Zc);
  const long pixels = (long)N * 4 * h * w;
  hipLaunchKernelGGL(tap_gather_up2_kernel, g_all(pixels), dim3(256), 0, (hipStream_t)stream, (const bf16*)Z, bias, (bf16*)y, pixels, h, w, ksize, pad,
                     Cout, Zc, tanh_out);
  GANK_LAUNCH_OK("tap_gather_up2");
  return 0;
}
extern "C" int gank_tap_scatter_up2(const void* g, void* col, int N, int h, int w, int ksize, int pad, int Cout, int Zc, void* stream) {
  GANK_REQUIRE(g && col && N > 0 && h > 0 && w > 0, "tap_scatter_up2: bad arguments");
  GANK_REQUIRE(ksize >= 1 && ksize <= 7 && pad >= 0 && Cout >= 1 && Cout <= 4 && ksize * ksize * Cout <= Zc && Zc % 8 == 0,
               "tap_scatter_up2: k=%d Cout=%d needs Cout <= 4, k*k*Cout <= Zc=%d and Zc %% 8 == 0", ksize, Cout, Zc);
  const long total8 = (long)N * h * w * (Zc / 8);
  hipLaunchKernelGGL(tap_scatter_up2_kernel, g_all(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)g, (bf16*)col, total8, h, w, ksize, pad, Cout, Zc);
  GANK_LAUNCH_OK("tap_scatter_up2");
  return 0;
}

// depth_to_space / space_to_depth with block size 2 in the channel order (a, b, c): y[n, 2i+a, 2j+b, c] = x[n, i, j, (2a+b) C + c]
// -- the interleave behind a phase-stacked conv (functional.conv2d_general: NN-upsample + 4x4 as ONE 3x3 conv at low resolution
// with 4 C output channels) and its adjoint.  16 bytes per lane, C % 8 == 0.
template <bool FWD>
__global__ void d2s2_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, long total8, int h, int w, int C) {
  const long i8 = blockIdx.x * (long)blockDim.x + threadIdx.x;
  if (i8 >= total8) return;
  const int cg = C >> 3;
  // index over the HIGH-resolution tensor [n, 2h, 2w, C/8]
  const int g = (int)(i8 % cg);
  long t = i8 / cg;
  const int X = (int)(t % (2 * w)); t /= 2 * w;
  const int Y = (int)(t % (2 * h));
  const long n = t / (2 * h);
  const long lo = (((n * h + (Y >> 1)) * w + (X >> 1)) * 4 + (Y & 1) * 2 + (X & 1)) * cg + g;
  if (FWD) reinterpret_cast<u32x4*>(dst)[i8] = reinterpret_cast<const u32x4*>(src)[lo];
  else reinterpret_cast<u32x4*>(dst)[lo] = reinterpret_cast<const u32x4*>(src)[i8];
}
extern "C" int gank_depth_to_space2(const void* x, void* y, int N, int h, int w, int C, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, "depth_to_space2: bad arguments (C %% 8 == 0)");
  const long total8 = (long)N * 4 * h * w * (C / 8);
  hipLaunchKernelGGL(d2s2_kernel<true>, g_all(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)x, (bf16*)y, total8, h, w, C);
  GANK_LAUNCH_OK("depth_to_space2");
  return 0;
}
extern "C" int gank_space_to_depth2(const void* y, void* x, int N, int h, int w, int C, void* stream) {
  GANK_REQUIRE(x && y && N > 0 && h > 0 && w > 0 && C > 0 && C % 8 == 0, "space_to_depth2: bad arguments (C %% 8 == 0)");
  const long total8 = (long)N * 4 * h * w * (C / 8);
  hipLaunchKernelGGL(d2s2_kernel<false>, g_all(total8), dim3(256), 0, (hipStream_t)stream, (const bf16*)y, (bf16*)x, total8, h, w, C);
  GANK_LAUNCH_OK("space_to_depth2");
  return 0;
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-side transforms of the Pix2Pix / PGGAN routes (fp32, a few MB at most) that used to run as torch.einsum / pad /
// repeat / add_ on the product path: every FLOP of a train step is now a kernel of this library.
// ------------------------------------------------------------------------------------------------------------------
// phase-stacked filter of "NN-upsample + 4x4 SAME conv" (functional.conv2d_general): output row 2i + a reads the low-resolution
// rows  a = 0: i-1 (ky 0), i (ky 1 + ky 2), i+1 (ky 3);  a = 1: i (ky 0 + ky 1), i+1 (ky 2 + ky 3)  -- likewise in x.
//   w3[u][v][ci][(a, b, co)] = sum_{ky in T(a,u), kx in T(b,v)} w4[ky][kx][ci][co]
// T(a,u) as a bit mask over ky:
__device__ __forceinline__ int ps4_taps(int a, int u) {
  return a == 0 ? (u == 0 ? 1 : (u == 1 ? 6 : 8)) : (u == 0 ? 0 : (u == 1 ? 3 : 12));
}
__device__ __forceinline__ int ps4_row(int a, int k) {     // the u with k in T(a,u)
  return a == 0 ? (k == 0 ? 0 : (k == 3 ? 2 : 1)) : (k < 2 ? 1 : 2);
}
__global__ void phase_stack4_fwd_kernel(const float* __restrict__ w4, float* __restrict__ w3, int Cin, int Cout, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    long t = i / Cout;
    const int b = (int)(t & 1), a = (int)((t >> 1) & 1); t >>= 2;
    const int ci = (int)(t % Cin); t /= Cin;
    const int v = (int)(t % 3), u = (int)(t / 3);
    const int my = ps4_taps(a, u), mx = ps4_taps(b, v);
    float sacc = 0.f;
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
      for (int kx = 0; kx < 4; kx++)
        if (((my >> ky) & 1) && ((mx >> kx) & 1)) sacc += w4[((long)(ky * 4 + kx) * Cin + ci) * Cout + co];
    w3[i] = sacc;
  }
}
// adjoint, ACCUMULATED: dw4[ky][kx][ci][co] += sum_{a,b} g3[u(a,ky)][u(b,kx)][ci][(a, b, co)]
__global__ void phase_stack4_bwd_kernel(const float* __restrict__ g3, float* __restrict__ dw4, int Cin, int Cout, long total) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int co = (int)(i % Cout);
    long t = i / Cout;
    const int ci = (int)(t % Cin); t /= Cin;
    const int kx = (int)(t & 3), ky = (int)(t >> 2);
    float sacc = 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
      for (int b = 0; b < 2; b++) {
        const int u = ps4_row(a, ky), v = ps4_row(b, kx);
        sacc += g3[((((long)(u * 3 + v) * Cin + ci) * 2 + a) * 2 + b) * Cout + co];
      }
    dw4[i] += sacc;
  }
}
extern "C" int gank_phase_stack4(const float* w4, float* w3, int Cin, int Cout, int adjoint, void* stream) {
  GANK_REQUIRE(w4 && w3 && Cin > 0 && Cout > 0, "phase_stack4: bad arguments");
  if (!adjoint) {
    const long total = 9L * Cin * 4 * Cout;
    hipLaunchKernelGGL(phase_stack4_fwd_kernel, g1(total), dim3(256), 0, (hipStream_t)stream, w4, w3, Cin, Cout, total);
  } else {       // w3 holds the stacked gradient, w4 the 4x4 filter's gradient (accumulated)
    const long total = 16L * Cin * Cout;
    hipLaunchKernelGGL(phase_stack4_bwd_kernel, g1(total), dim3(256), 0, (hipStream_t)stream, (const float*)w3, const_cast<float*>(w4), Cin, Cout, total);
  }
  GANK_LAUNCH_OK("phase_stack4");
  return 0;
}

// rows of `w_in` values -> rows of `w_out >= w_in` values, zeros behind (adjoint = 0), or the first w_in values of every wide row
// back into the narrow one (adjoint = 1: fp32 ACCUMULATES, 16-bit overwrites).  Serves: zero input channels behind a filter
// ([k*k][Cin*Cout] -> [k*k][Cp*Cout]) and behind an activation ([pixels][C] -> [pixels][Cp]).
template <typename T, bool ADJ>
__global__ void pad_rows_kernel(const T* __restrict__ src, T* __restrict__ dst, long rows, int w_in, int w_out) {
  const long total = ADJ ? rows * w_in : rows * (long)w_out;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    if (ADJ) {
      const long r = i / w_in;
      const int c = (int)(i - r * w_in);
      if constexpr (sizeof(T) == 4) dst[i] += src[r * w_out + c];
      else dst[i] = src[r * w_out + c];
    } else {
      const long r = i / w_out;
      const int c = (int)(i - r * w_out);
      dst[i] = c < w_in ? src[r * w_in + c] : T(0);
    }
  }
}
extern "C" int gank_pad_rows(const void* src, void* dst, long rows, int w_in, int w_out, int elem_bytes, int adjoint, void* stream) {
  GANK_REQUIRE(src && dst && rows > 0 && w_in > 0 && w_out >= w_in && (elem_bytes == 2 || elem_bytes == 4), "pad_rows: bad arguments");
  hipStream_t s = (hipStream_t)stream;
  const long total = adjoint ? rows * w_in : rows * (long)w_out;
  if (elem_bytes == 4) {
    if (adjoint) hipLaunchKernelGGL((pad_rows_kernel<float, true>), g1(total), dim3(256), 0, s, (const float*)src, (float*)dst, rows, w_in, w_out);
    else hipLaunchKernelGGL((pad_rows_kernel<float, false>), g1(total), dim3(256), 0, s, (const float*)src, (float*)dst, rows, w_in, w_out);
  } else {
    if (adjoint) hipLaunchKernelGGL((pad_rows_kernel<unsigned short, true>), g1(total), dim3(256), 0, s, (const unsigned short*)src, (unsigned short*)dst, rows, w_in, w_out);
    else hipLaunchKernelGGL((pad_rows_kernel<unsigned short, false>), g1(total), dim3(256), 0, s, (const unsigned short*)src, (unsigned short*)dst, rows, w_in, w_out);
  }
  GANK_LAUNCH_OK("pad_rows");
  return 0;
}

// b [n] -> [reps][n] (adjoint = 0), or b[c] += sum_j g[j][c] (adjoint = 1): the bias of a phase-stacked conv and its gradient
__global__ void tile_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, int reps, int n, int adjoint) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < (adjoint ? n : reps * n); i += gridDim.x * blockDim.x) {
    if (adjoint) {
      float sacc = 0.f;
      for (int j = 0; j < reps; j++) sacc += src[j * n + i];
      dst[i] += sacc;
    } else {
      dst[i] = src[i % n];
    }
  }
}
extern "C" int gank_tile_rows(const float* src, float* dst, int reps, int n, int adjoint, void* stream) {
  GANK_REQUIRE(src && dst && reps > 0 && n > 0, "tile_rows: bad arguments");
  hipLaunchKernelGGL(tile_rows_kernel, g1(adjoint ? n : (long)reps * n), dim3(256), 0, (hipStream_t)stream, src, dst, reps, n, adjoint);
  GANK_LAUNCH_OK("tile_rows");
  return 0;
}

// filter [k*k][Cin][Cout] (Cout <= 4) <-> the 1x1 operand of the few-output upsample conv, wz [Cin][Zc] with column t*Cout + co
// (zeros behind k*k*Cout); adjoint = 1: dw[t][ci][co] += gz[ci][t*Cout + co]
__global__ void fewout_pack_kernel(const float* __restrict__ src, float* __restrict__ dst, int taps, int Cin, int Cout, int Zc, int adjoint) {
  const int total = adjoint ? taps * Cin * Cout : Cin * Zc;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
    if (adjoint) {
      const int co = i % Cout, ci = (i / Cout) % Cin, t = i / (Cout * Cin);
      dst[i] += src[ci * Zc + t * Cout + co];
    } else {
      const int col = i % Zc, ci = i / Zc;
      const int t = col / Cout, co = col - t * Cout;
      dst[i] = t < taps ? src[(t * Cin + ci) * Cout + co] : 0.f;
    }
  }
}
extern "C" int gank_fewout_pack(const float* src, float* dst, int ksize, int Cin, int Cout, int Zc, int adjoint, void* stream) {
  GANK_REQUIRE(src && dst && ksize >= 1 && ksize <= 7 && Cin > 0 && Cout >= 1 && Cout <= 4 && ksize * ksize * Cout <= Zc, "fewout_pack: bad arguments");
  const int taps = ksize * ksize;
  hipLaunchKernelGGL(fewout_pack_kernel, g1(adjoint ? (long)taps * Cin * Cout : (long)Cin * Zc), dim3(256), 0, (hipStream_t)stream, src, dst, taps, Cin, Cout, Zc, adjoint);
  GANK_LAUNCH_OK("fewout_pack");
  return 0;
}

// zero fill (fp32 scratch of a backward pass: a kernel of this library instead of a framework fill)
__global__ void zero_f32_kernel(float* __restrict__ p, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = 0.f;
}
extern "C" int gank_zero_f32(float* p, long n, void* stream) {
  GANK_REQUIRE(p && n > 0, "zero_f32: bad arguments");
  hipLaunchKernelGGL(zero_f32_kernel, g1(n), dim3(256), 0, (hipStream_t)stream, p, n);
  GANK_LAUNCH_OK("zero_f32");
  return 0;
}
