// micro-benchmark: the batched spectral norm of the SNGAN critic (12 weights), forward A / B and backward 1 / 2 separately
#include "stubs.h"
#include "../../gan_lib_tensorflow_amd/csrc/sn.hip"

int main() {
  const int KC[12][2] = {{3, 128}, {27, 128}, {1152, 128}, {300, 128}, {256, 128}, {2304, 256}, {2304, 128}, {1152, 128}, {1152, 128}, {1152, 128}, {1152, 128}, {128, 1}};
  const int kind[12] = {0, 0, 5, -1, 0, 0, 5, 4, 4, 4, 4, -1};
  const int ks[12] = {1, 3, 3, 1, 1, 3, 3, 3, 3, 3, 3, 1};
  gank_sn_desc t[12];
  gank_prep_desc pd[12]; int pw[12]; int np = 0;
  for (int i = 0; i < 12; i++) {
    const int K = KC[i][0], C = KC[i][1], nch = (K + 63) / 64;
    gank_sn_desc& d = t[i];
    float *W, *u, *uo, *v, *Wb, *sc, *a, *b, *bp, *G, *dW, *ga, *us;
    CK(hipMalloc(&W, K * C * 4)); CK(hipMalloc(&u, C * 4 + 16)); CK(hipMalloc(&uo, C * 4 + 16)); CK(hipMalloc(&v, K * 4)); CK(hipMalloc(&Wb, K * C * 4));
    CK(hipMalloc(&sc, 64)); CK(hipMalloc(&a, K * 4)); CK(hipMalloc(&b, C * 4 + 16)); CK(hipMalloc(&bp, gank_sn_ws_floats(K, C) * 4 + 16)); CK(hipMalloc(&G, K * C * 4));
    CK(hipMalloc(&dW, K * C * 4)); CK(hipMalloc(&ga, K * 4)); CK(hipMalloc(&us, C * 4 + 16));
    std::vector<float> h(K * C); for (auto& x : h) x = 0.05f * ((rand() % 2001) / 1000.f - 1.f);
    CK(hipMemcpy(W, h.data(), K * C * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(G, h.data(), K * C * 4, hipMemcpyHostToDevice));
    std::vector<float> hu(C, 1.f); CK(hipMemcpy(u, hu.data(), C * 4, hipMemcpyHostToDevice)); CK(hipMemset(dW, 0, K * C * 4));
    d = gank_sn_desc{W, u, uo, v, Wb, sc, a, b, bp, G, dW, nullptr, ga, us, K, C, 0, 0};
    if (kind[i] >= 0) {
      const int k = ks[i], cin = K / (k * k), cout = C;
      void *wf, *wd; CK(hipMalloc(&wf, (size_t)32 * K * C + 65536)); CK(hipMalloc(&wd, (size_t)32 * K * C + 65536));
      pd[np] = gank_prep_desc{W, wf, wd, k, cin, cout, kind[i]}; pw[np] = i; np++;
    }
  }
  float *tab, *bias; void* lout; CK(hipMalloc(&tab, 3000 * 4)); CK(hipMalloc(&bias, 512)); CK(hipMalloc(&lout, 10 * 128 * 2));
  CK(hipMemset(tab, 0, 3000 * 4)); CK(hipMemset(bias, 0, 512));
  gank_label_dense_desc ld{tab, bias, lout, 10, 300, 3};
  printf("fwd (A + B, prep, label)   %.1f us\n", time_us([&] { gank_sn_power_iter_fwd_prep(t, 12, pd, pw, np, &ld, 0); }));
  printf("fwd (A + B, prep)          %.1f us\n", time_us([&] { gank_sn_power_iter_fwd_prep(t, 12, pd, pw, np, nullptr, 0); }));
  printf("fwd (A + B only)           %.1f us\n", time_us([&] { gank_sn_power_iter_fwd(t, 12, 0); }));
  printf("bwd (1 + 2)                %.1f us\n", time_us([&] { gank_sn_power_iter_bwd(t, 12, 0); }));
  // A alone / B alone
  SnTable st; int chunks, fine; sn_fill(st, t, 12, chunks, fine, false);
  unsigned* tk; CK(hipGetSymbolAddress((void**)&tk, HIP_SYMBOL(sn_tickets)));
  printf("A alone                    %.1f us   (%d chunks)\n", time_us([&] { hipLaunchKernelGGL(sn_fwd_a_kernel, dim3(chunks), dim3(256), 0, 0, st, tk); }), chunks);
  SnTable sb; sn_fill(sb, t, 12, chunks, fine, true);
  printf("bwd 1 alone                %.1f us\n", time_us([&] { hipLaunchKernelGGL(sn_bwd_gw_kernel, dim3(fine), dim3(256), 0, 0, sb); }));
  printf("bwd 2 alone                %.1f us\n", time_us([&] { hipLaunchKernelGGL(sn_bwd_apply_kernel, dim3(fine), dim3(256), 0, 0, sb); }));
  auto empty = [&] { hipLaunchKernelGGL(label_dense_table_kernel, dim3(0 + 1), dim3(64), 0, 0, SnLabelDense{tab, tab, nullptr, nullptr, (bf16*)lout, 1, 1, 1}); };
  printf("tiny kernel                %.1f us\n", time_us(empty));
  return 0;
}
