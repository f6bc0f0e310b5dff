// micro-benchmark: gank_label_dense_bwd + concat_label_bwd at the critic's shapes, with phase stamps (s_memtime) in a copy of the kernel
#include "stubs.h"
#include "../../gan_lib_tensorflow_amd/csrc/elementwise.hip"

int main() {
  const int N = 128, V = 10, D = 300, C2 = 128, HW = 256, C1 = 128;
  float *de, *table, *W, *dW, *db, *dt; int* lab; bf16 *dy, *da;
  CK(hipMalloc(&de, N * C2 * 4)); CK(hipMalloc(&table, V * D * 4)); CK(hipMalloc(&W, D * C2 * 4)); CK(hipMalloc(&dW, D * C2 * 4));
  CK(hipMalloc(&db, C2 * 4)); CK(hipMalloc(&dt, V * D * 4)); CK(hipMalloc(&lab, N * 4)); 
  CK(hipMalloc(&dy, (size_t)N * HW * (C1 + C2) * 2)); CK(hipMalloc(&da, (size_t)N * HW * C1 * 2));
  CK(hipMemset(de, 0, N * C2 * 4)); CK(hipMemset(table, 0, V * D * 4)); CK(hipMemset(W, 0, D * C2 * 4)); CK(hipMemset(dW, 0, D * C2 * 4));
  CK(hipMemset(db, 0, C2 * 4)); CK(hipMemset(dt, 0, V * D * 4)); CK(hipMemset(dy, 0, (size_t)N * HW * (C1 + C2) * 2));
  std::vector<int> h(N); for (int i = 0; i < N; i++) h[i] = i % V;
  CK(hipMemcpy(lab, h.data(), N * 4, hipMemcpyHostToDevice));
  printf("label_dense_bwd        %.1f us\n", time_us([&] { gank_label_dense_bwd(de, lab, table, W, dW, db, dt, N, V, D, C2, 0); }));
  printf("concat_label_bwd       %.1f us\n", time_us([&] { gank_concat_label_bwd(dy, da, de, N, HW, C1, C2, 0); }));
  printf("pair                   %.1f us\n", time_us([&] { gank_concat_label_bwd(dy, da, de, N, HW, C1, C2, 0); gank_label_dense_bwd(de, lab, table, W, dW, db, dt, N, V, D, C2, 0); }));
  return 0;
}
