// stubs that let a single csrc/*.hip be compiled into a stand-alone micro-benchmark (scratch/micro/*.hip include the source text)
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
int gank_set_error(const char* fmt, ...) { va_list ap; va_start(ap, fmt); vfprintf(stderr, fmt, ap); va_end(ap); fprintf(stderr, "\n"); return 1; }
void gank_prof_begin(int, double, hipStream_t, double) {}
void gank_prof_end(int, hipStream_t) {}
void gank_prof_tag(int, const char*) {}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
template <class F> static float time_us(F f, int warm = 5, int reps = 50) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < warm; i++) f();
  CK(hipDeviceSynchronize());
  std::vector<float> t;
  for (int r = 0; r < 5; r++) {
    CK(hipEventRecord(a)); for (int i = 0; i < reps; i++) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); t.push_back(1e3f * ms / reps);
  }
  std::sort(t.begin(), t.end()); return t[2];
}
