// micro-benchmark: filter gradients of the critic's 3-channel-input layers (D.Block.1.Conv1 3x3 @ 32x32, Shortcut 1x1 @ 16x16), 128 samples
#include "stubs.h"
extern "C" int gank_colsum_bf16(const void*, float*, long, int, float, void*) { return 0; }
#include "../../gan_lib_tensorflow_amd/csrc/conv_wgrad.hip"

int main() {
  const int N = 128, C = 128;
  bf16 *x0, *dy0, *x1, *dy1; float *dw0, *db0, *dw1, *db1;
  CK(hipMalloc(&x0, (size_t)N * 32 * 32 * 3 * 2)); CK(hipMalloc(&dy0, (size_t)N * 32 * 32 * C * 2));
  CK(hipMalloc(&x1, (size_t)N * 16 * 16 * 3 * 2)); CK(hipMalloc(&dy1, (size_t)N * 16 * 16 * C * 2));
  CK(hipMalloc(&dw0, 27 * C * 4)); CK(hipMalloc(&db0, C * 4)); CK(hipMalloc(&dw1, 3 * C * 4)); CK(hipMalloc(&db1, C * 4));
  CK(hipMemset(x0, 0x3c, (size_t)N * 32 * 32 * 3 * 2)); CK(hipMemset(dy0, 0x3c, (size_t)N * 32 * 32 * C * 2));
  CK(hipMemset(x1, 0x3c, (size_t)N * 16 * 16 * 3 * 2)); CK(hipMemset(dy1, 0x3c, (size_t)N * 16 * 16 * C * 2));
  CK(hipMemset(dw0, 0, 27 * C * 4)); CK(hipMemset(db0, 0, C * 4)); CK(hipMemset(dw1, 0, 3 * C * 4)); CK(hipMemset(db1, 0, C * 4));
  float* big; CK(hipMalloc(&big, 512u << 20));      // flushes the caches between timed launches where asked
  auto flush = [&] { CK(hipMemsetAsync(big, 1, 512u << 20, 0)); };
  auto conv1 = [&] { gank_conv2d_wgrad(x0, dy0, dw0, db0, nullptr, 0, N, 32, 32, 3, C, 3, 0, 1.f, 0); };
  auto shortc = [&] { gank_conv2d_wgrad(x1, dy1, dw1, db1, nullptr, 0, N, 16, 16, 3, C, 1, 0, 1.f, 0); };
  auto pair = [&] { gank_conv2d_wgrad_narrow_pair(x0, dy0, dw0, db0, N, 32, 32, C, 3, x1, dy1, dw1, db1, N, 16, 16, C, 1, 1.f, 0); };
  for (const char* mode : {"stream", "packed"}) {
    setenv("GANK_WGRAD_STREAM", mode[0] == 's' ? "1" : "0", 1);
    printf("[%s] conv1 3x3 32x32      %.1f us\n", mode, time_us(conv1));
    printf("[%s] shortcut 1x1 16x16   %.1f us\n", mode, time_us(shortc));
    break;     // the knob is read once per process: run the binary twice (GANK_WGRAD_STREAM=0) for the packed kernel
  }
  printf("pair                      %.1f us\n", time_us(pair));
  // cold: caches flushed before every launch (one launch per timing, median of 9)
  std::vector<float> t;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 9; i++) { flush(); CK(hipEventRecord(a)); pair(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f); }
  std::sort(t.begin(), t.end()); printf("pair, caches flushed      %.1f us (event pair around one launch)\n", t[4]);
  t.clear();
  for (int i = 0; i < 9; i++) { flush(); CK(hipEventRecord(a)); conv1(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(ms * 1e3f); }
  std::sort(t.begin(), t.end()); printf("conv1, caches flushed     %.1f us\n", t[4]);
  return 0;
}
