// Times the fused discriminator stack (gantts_amd/csrc/dstack_f32.hip.h) at the cfg2 shapes: D step (32768 rows, 3 x 256) and the
// generator step's adversarial pass (16384 rows).  -DDS_ABL=<mask> compiles pieces out (see the header).  Random operands, Philox
// dropout 0.5 (the production epilogue).  Prints us per launch and TFLOP/s of the products it contains.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../gantts_amd/csrc/dstack_f32.hip.h"
using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static float* dev_rand(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * ((float)(s >> 8) / 8388608.f - 1.f); }
  float* d; CK(hipMalloc(&d, n * 4)); CK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice)); return d;
}
int main(int argc, char** argv) {
  const int HD = 256, L = argc > 1 ? atoi(argv[1]) : 3, N = argc > 2 ? atoi(argv[2]) : 16384, K0 = 483, Da = 58, col0 = 425;
  const int reps = 200;
  StepScalars hsc; memset(&hsc, 0, sizeof(hsc)); hsc.tv = 12000.f; hsc.inv_tv = 1.f / 12000.f;
  StepScalars* sc; CK(hipMalloc(&sc, sizeof(hsc))); CK(hipMemcpy(sc, &hsc, sizeof(hsc), hipMemcpyHostToDevice));
  for (int mode = 0; mode < 2; ++mode) {
    const int rows = mode == 0 ? 2 * N : N;
    DStackArgs a; memset(&a, 0, sizeof(a));
    a.mode = mode; a.L = L; a.rows = rows; a.n_real = N;
    a.H0 = dev_rand((size_t)rows * HD, 1.f, 1);
    for (int l = 0; l < L; ++l) {
      a.W[l] = dev_rand((size_t)HD * HD, 0.06f, 10 + l); a.b[l] = dev_rand(HD, 0.05f, 20 + l);
      DropoutSpec d; memset(&d, 0, sizeof(d)); d.mode = DROP_PHILOX; d.p = 0.5f; d.scale = 2.f; d.thresh = 0x8000u; d.key0 = 77u + l; d.key1 = 99u;
      a.drop[l] = d;
      float* h; CK(hipMalloc(&h, (size_t)rows * HD * 4)); a.Hout[l] = h;
    }
    a.w_last = dev_rand(HD, 0.06f, 3); a.b_last = dev_rand(1, 0.05f, 4);
    std::vector<float> hm(N, 1.f); float* m; CK(hipMalloc(&m, N * 4)); CK(hipMemcpy(m, hm.data(), N * 4, hipMemcpyHostToDevice));
    a.mask = m; a.n_mask = N; a.eps = 1e-20f; a.sc = sc; a.want_grad = 1;
    CK(hipMalloc(&a.dZtop, (size_t)rows * HD * 4)); CK(hipMalloc(&a.Dout, (size_t)rows * 4));
    const int grid = (rows + DS_R - 1) / DS_R;
    CK(hipMalloc(&a.hp, grid * sizeof(HeadPartials))); CK(hipMalloc(&a.dw_partial, (size_t)grid * HD * 4));
    a.W0 = dev_rand((size_t)HD * K0, 0.05f, 5); a.ldw0 = K0; a.col0 = col0; a.Da = Da;
    CK(hipMalloc(&a.gadv, (size_t)rows * Da * 4)); a.ld_gadv = Da;
    const size_t lds = dstack_lds_bytes<256>();
    CK(hipFuncSetAttribute((const void*)dstack_kernel<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int occ = -1; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)dstack_kernel<256>, DS_THREADS, lds));
    unsigned long long* dbg; CK(hipMalloc(&dbg, (size_t)grid * 16 * 8)); CK(hipMemset(dbg, 0, (size_t)grid * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // clock spin-up: the device idles at a few hundred MHz and takes tens of milliseconds of load to reach its working clocks
    for (int w = 0; w < (getenv("DS_NOSPIN") ? 5 : 4000); ++w) hipLaunchKernelGGL((dstack_kernel<256>), dim3(grid), dim3(DS_THREADS), lds, 0, a);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((dstack_kernel<256>), dim3(grid), dim3(DS_THREADS), lds, 0, a);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps;
    if (getenv("DS_DBG")) {     // one more launch with phase stamps
      a.dbg = dbg;
      hipLaunchKernelGGL((dstack_kernel<256>), dim3(grid), dim3(DS_THREADS), lds, 0, a);
      CK(hipDeviceSynchronize());
      std::vector<unsigned long long> h((size_t)grid * 16);
      CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int b = 0; b < grid; ++b) { if (h[b * 16] < t0) t0 = h[b * 16]; for (int i = 0; i < 16; ++i) if (h[b * 16 + i] > t1) t1 = h[b * 16 + i]; }
      printf("  kernel span by stamps: %.1f us\n", (t1 - t0) * 0.01);
      const int show[4] = {0, grid / 3, grid / 2 + 1, grid - 1};
      for (int q = 0; q < 4; ++q) {
        const int b = show[q];
        printf("  wg %4d start +%.1f us; phase durations (us):", b, (h[b * 16] - t0) * 0.01);
        for (int i = 1; i < 16 && h[b * 16 + i]; ++i) printf(" %.2f", (h[b * 16 + i] - h[b * 16 + i - 1]) * 0.01);
        printf("\n");
      }
      a.dbg = nullptr;
    }
    double flop = 2.0 * rows * HD * HD * (L - 1) * (mode == 1 ? 2 : 1) + (mode == 1 ? 2.0 * rows * HD * Da : 0.0);
    printf("DS_ABL=%d mode %s rows %d L %d: %.1f us per launch, %.1f TFLOP/s (%.2f of 157.3), lds %zu B, grid %d, occupancy %d WG/CU\n", DS_ABL, mode ? "G_ADV " : "D_STEP", rows, L, us,
           flop / us * 1e-6, flop / us * 1e-6 / 157.3, lds, grid, occ);
  }
  return 0;
}
