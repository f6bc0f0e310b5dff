// Shared device/host helpers for libgank (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <atomic>
#include <string>

#include "../../include/gank.h"

// The 16-bit element type of activations, activation gradients and MFMA operands.  The library is built twice from the
// same sources: libgank.so with bfloat16 (default) and libgank_f16.so with IEEE half (-DGANK_ACT_F16: the same kernels on
// v_mfma_f32_32x32x16_f16, fp32 accumulation either way).  The type keeps the name `bf16` in the sources: every conversion
// goes through bf2f / f2bf, every matrix instruction through GANK_MFMA32, and nothing else depends on the encoding (the
// packed relu is a signed 16-bit max: the sign bit sits in the same place; LDS transposes and DMA move raw 16-bit lanes).
#ifdef GANK_ACT_F16
typedef _Float16 bf16;
typedef _Float16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 bf16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 bf16x2 __attribute__((ext_vector_type(2)));
#define GANK_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define GANK_ACT_DTYPE 1
#else
typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
#define GANK_MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#define GANK_ACT_DTYPE 0
#endif
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

int gank_set_error(const char* fmt, ...);

#define GANK_REQUIRE(cond, ...)                       \
  do {                                                \
    if (!(cond)) return gank_set_error(__VA_ARGS__);  \
  } while (0)

#define GANK_LAUNCH_OK(name)                                                               \
  do {                                                                                     \
    hipError_t e__ = hipGetLastError();                                                    \
    /* hipErrorNotReady is the sticky residue of the host's own hipEventQuery polling, not a launch error */ \
    if (e__ != hipSuccess && e__ != hipErrorNotReady)                                      \
      return gank_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));        \
  } while (0)

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (kernel, device): the attribute is per device, so a
// process that drives a second GPU must set it there too (a process-wide flag left the 144 KB kernels unlaunchable
// on every device but the first).
#define GANK_MAX_DYNAMIC_LDS(kern, bytes, name)                                                               \
  do {                                                                                                        \
    static std::atomic<unsigned long long> done__{0};          /* bit d = set on device d (d < 64) */         \
    int dev__ = 0;                                                                                            \
    (void)hipGetDevice(&dev__);                                                                               \
    const unsigned long long bit__ = 1ull << (dev__ & 63);                                                    \
    if (!(done__.load(std::memory_order_relaxed) & bit__)) {                                                  \
      hipError_t e__ = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                               \
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes));         \
      if (e__ != hipSuccess) return gank_set_error("%s: hipFuncSetAttribute: %s", name, hipGetErrorString(e__)); \
      done__.fetch_or(bit__, std::memory_order_relaxed);                                                      \
    }                                                                                                         \
  } while (0)

static inline std::string gank_format(const char* fmt, ...) {
  char buf[160];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  return std::string(buf);
}

// Tuning constants of the launch dispatchers (split targets, prefetch depths, kernel on/off switches).  A production build
// bakes the defaults in: no environment lookups on the product path.  An experiment build (GANK_EXTRA_FLAGS=-DGANK_TUNING
// GANK_LIB_NAME=libgank_tune.so python -m gan_lib_tensorflow_amd.build) reads each one from the environment once, for the A/B
// sweeps under scratch/.
static inline int gank_tune(const char* name, int dflt) {
#ifdef GANK_TUNING
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
#else
  (void)name;
  return dflt;
#endif
}
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int roundup(int a, int b) { return (a + b - 1) / b * b; }
static inline int log2_or_neg(int v) { int s = 0; while ((1 << s) < v) s++; return ((1 << s) == v) ? s : -1; }

// profiling hooks (api.hip)
void gank_prof_begin(int family, double flops, hipStream_t s, double bytes = 0.0);   // bytes: algorithmic operand + result bytes
void gank_prof_end(int family, hipStream_t s);
void gank_prof_tag(int family, const char* kernel_name);   // names the kernel of the open record (static string)

#ifdef __HIPCC__
__device__ __forceinline__ float bf2f(bf16 x) { return (float)x; }
__device__ __forceinline__ bf16 f2bf(float x) { return (bf16)x; }

// relu on 8 packed bf16: as signed 16-bit ints, negative floats are negative ints (v_pk_max_i16)
__device__ __forceinline__ u32x4 relu_bf16x8(u32x4 v) {
  s16x8 s = __builtin_bit_cast(s16x8, v);
  s16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  s = __builtin_elementwise_max(s, z);
  return __builtin_bit_cast(u32x4, s);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; i++) t += red[i];
  return t;
}

// tf.train.AdamOptimizer's step size (loss_opt.hip: adam_tf_kernel; sn.hip: the fused spectral-norm tail): hp = {lr, beta1,
// beta2, eps, grad_scale, decay_on}; t_state[0] = steps taken so far; iteration[0] = the `_iteration` feed of the LR decay
// (SNGAN/gan_cifar_resnet.py:454-459).
__device__ __forceinline__ float adam_lr_t(const float* hp, const long long* t_state, const long long* iteration) {
  const double t = (double)(t_state[0] + 1);
  double lr = hp[0];
  if (hp[5] != 0.f && iteration) {
    const double it = (double)iteration[0];
    lr *= (it < 50000.0) ? fmax(0.0, 1.0 - it / 100000.0) : 0.5;
  }
  return (float)(lr * sqrt(1.0 - pow((double)hp[2], t)) / (1.0 - pow((double)hp[1], t)));
}

// m -> (n, oh, ow) over an [N,H,W] pixel grid; shifts when H*W and W are powers of two (sw/shw >= 0)
__device__ __forceinline__ void pix_decomp(int m, int H, int W, int shw, int sw, int& n, int& oh, int& ow) {
  if (shw >= 0 && sw >= 0) {
    n = m >> shw;
    const int rem = m & ((1 << shw) - 1);
    oh = rem >> sw;
    ow = rem & ((1 << sw) - 1);
  } else {
    const int hw = H * W;
    n = m / hw;
    const int rem = m - n * hw;
    oh = rem / W;
    ow = rem - oh * W;
  }
}

// XCD-aware bijective block remap (8 XCDs; consecutive logical ids share an XCD's L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
  const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return base + (bid >> 3);
}
// Widening of a 32x32 accumulator tile's pieces.  An accumulator quad g of lane (r, h) is channels 8g+4h..+3 of pixel r;
// v_permlane32_swap trades quad 2q+1 of the h = 0 half-wave for quad 2q of the h = 1 half-wave, after which the lane owns 8
// CONSECUTIVE channels 16q + 8h .. +7 (v[0..7]): 16-byte instead of 8-byte stores / mask / residual loads.  All 64 lanes (or
// both lanes r and r + 32 of every pixel) must be active.
__device__ __forceinline__ void acc_widen(const f32x16& acc, int q, float scale, float (&v)[8]) {
#pragma unroll
  for (int e = 0; e < 4; e++) {
    // inline asm, not __builtin_amdgcn_permlane32_swap: hipcc (ROCm 7.2) dropped the builtin's second result
    // (the +4 half came out as a copy of the first).  s_nop 1 = the 2 wait states a VALU write needs before the swap reads it.
    float lo = acc[8 * q + e] * scale, hi = acc[8 * q + 4 + e] * scale;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
    v[e] = lo;
    v[4 + e] = hi;
  }
}

#endif
