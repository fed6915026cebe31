// Harness (no torch, GPU box only): f32 MFMA GEMM of gemm_f32.hip.h on the cfg2 shapes over tile shapes and K depths.
//   * tile sweep: NT / NN at 128x128, 64x128, 128x64, 64x64, TN at 128x128 / 128x64 (slab count as the engine picks it)
//   * K sweep (NT): time vs K at fixed M, N -> slope (steady-state K loop) and intercept (launch + prologue + epilogue)
// usage: gemm_tile_sweep [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
#include "../gantts_amd/csrc/gemm_f32.hip.h"
#include "gemm_dma_variant.hip.h"
using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static float* dfill(size_t n, float scale, unsigned s) {
  std::vector<float> h(n);
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f) * scale; }
  float* p; CK(hipMalloc((void**)&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
static int g_reps = 30, g_target = 512, g_bm = 128;
template <int KIND, int BM, int BN, int BKT = 32>
static double run(GemmArgs g, int nslab) {
  const size_t lds = gemm_lds_bytes<KIND, BM, BN, PREC_F32, BKT>();
  auto kern = gemm_f32_kernel<KIND, BM, BN, true, true, PREC_F32, BKT>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + BM - 1) / BM; g.n_tiles_n = (g.N + BN - 1) / BN;
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / g_reps;
}
template <int KIND>
static double run_dma(GemmArgs g) {
  const size_t lds = gemm_dma_lds_bytes();
  auto kern = gemm_dma_kernel<KIND>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + 63) / 64; g.n_tiles_n = (g.N + 63) / 64;
  const int grid = g.n_tiles_m * g.n_tiles_n;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < g_reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / g_reps;
}
// max |difference| of the DMA kernel against the register-staged 64x64 kernel on the same arguments, relative to max |C|
template <int KIND>
static double check_dma(GemmArgs g, size_t n_out) {
  std::vector<float> ref(n_out), got(n_out);
  g.n_tiles_m = (g.M + 63) / 64; g.n_tiles_n = (g.N + 63) / 64;
  const int grid = g.n_tiles_m * g.n_tiles_n;
  const size_t l0 = gemm_lds_bytes<KIND, 64, 64>();
  CK(hipFuncSetAttribute((const void*)gemm_f32_kernel<KIND, 64, 64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l0));
  CK(hipMemset(g.C, 0, n_out * 4));
  hipLaunchKernelGGL((gemm_f32_kernel<KIND, 64, 64, true, true>), dim3(grid), dim3(256), l0, 0, g);
  CK(hipMemcpy(ref.data(), g.C, n_out * 4, hipMemcpyDeviceToHost));
  CK(hipFuncSetAttribute((const void*)gemm_dma_kernel<KIND>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)gemm_dma_lds_bytes()));
  CK(hipMemset(g.C, 0, n_out * 4));
  hipLaunchKernelGGL(gemm_dma_kernel<KIND>, dim3(grid), dim3(256), gemm_dma_lds_bytes(), 0, g);
  CK(hipMemcpy(got.data(), g.C, n_out * 4, hipMemcpyDeviceToHost));
  double mx = 0, sc = 0;
  for (size_t i = 0; i < n_out; ++i) { mx = std::max(mx, (double)fabsf(ref[i] - got[i])); sc = std::max(sc, (double)fabsf(ref[i])); }
  return mx / (sc > 0 ? sc : 1);
}
struct Bufs { float *A, *B, *C, *H, *bias; };
static GemmArgs make(int kind, int M, int N, int K, const Bufs& b, bool philox, int* nslab, int bn_for_slabs) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.C = b.C; g.ldc = N; g.wide_store = 1;
  g.drop.mode = philox ? DROP_PHILOX : DROP_NONE; g.drop.p = 0.5f; g.drop.scale = 2.f; g.drop.thresh = 32768; g.drop.key0 = 123; g.drop.key1 = 456;
  *nslab = 1;
  if (kind == GEMM_NT) { g.A = b.A; g.lda = K; g.B = b.B; g.ldb = K; g.bias = b.bias; g.act = ACT_LEAKY_DROPOUT; }
  else if (kind == GEMM_NN) { g.A = b.A; g.lda = K; g.B = b.B; g.ldb = N; g.act = ACT_LEAKY_DROPOUT; g.H = b.H; g.ldh = N; }
  else {
    const int tiles = ((M + g_bm - 1) / g_bm) * ((N + bn_for_slabs - 1) / bn_for_slabs);
    int ns = g_target / tiles; if (ns < 1) ns = 1;
    int kc = (((K + ns - 1) / ns) + 31) / 32 * 32; ns = (K + kc - 1) / kc;
    g.A = b.A; g.lda = M; g.B = b.B; g.ldb = N; *nslab = ns; g.k_chunk = kc; g.slab_stride = (long)M * N; g.drop.mode = DROP_NONE; g.act = ACT_NONE;
  }
  return g;
}
int main(int argc, char** argv) {
  g_reps = argc > 1 ? atoi(argv[1]) : 30;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  const size_t big = (size_t)32768 * 2048;
  Bufs b; b.A = dfill(big, 1.f, 1); b.B = dfill(big, 0.05f, 2); CK(hipMalloc((void**)&b.C, big * 4 + 64)); b.H = dfill(big, 1.f, 3); b.bias = dfill(4096, 0.1f, 4);
  struct Shape { const char* name; int kind, M, N, K; };
  const Shape shapes[] = {
    {"fwd  G 16384x512x512", GEMM_NT, 16384, 512, 512}, {"fwd  D 32768x256x256", GEMM_NT, 32768, 256, 256}, {"fwd  D 16384x256x256", GEMM_NT, 16384, 256, 256},
    {"bwdX G 16384x512x512", GEMM_NN, 16384, 512, 512}, {"bwdX D 32768x256x256", GEMM_NN, 32768, 256, 256}, {"bwdX D 16384x256x256", GEMM_NN, 16384, 256, 256},
    {"bwdW G 512x512x16384", GEMM_TN, 512, 512, 16384}, {"bwdW D 256x256x32768", GEMM_TN, 256, 256, 32768},
  };
  const bool quick = argc > 2;          // quick: 64x64 tiles only, every shape twice (A/B of build variants, e.g. -DGT_EPI_STORE=n)
  if (quick) {
    for (int rep = 0; rep < 2; ++rep)
      for (const Shape& sh : shapes) {
        int ns; GemmArgs g = make(sh.kind, sh.M, sh.N, sh.K, b, true, &ns, 64);
        g_bm = 64;
        if (sh.kind == GEMM_TN) g = make(sh.kind, sh.M, sh.N, sh.K, b, false, &ns, 64);
        const double fl = 2.0 * sh.M * sh.N * sh.K;
        const double us = sh.kind == GEMM_NT ? run<GEMM_NT, 64, 64>(g, 1) : sh.kind == GEMM_NN ? run<GEMM_NN, 64, 64>(g, 1) : run<GEMM_TN, 64, 64>(g, ns);
        printf("   %s 64x64: %7.1f us %6.1f TF\n", sh.name, us, fl / us / 1e6);
      }
    for (int K : {64, 256, 512, 2048}) {
      int ns; GemmArgs g = make(GEMM_NT, 16384, 512, K, b, true, &ns, 64);
      printf("   NT 16384x512 K %4d: %7.1f us\n", K, run<GEMM_NT, 64, 64>(g, 1));
    }
    return 0;
  }
  for (const Shape& sh : shapes) {
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    printf("%s  ideal %.1f us\n", sh.name, fl / 157.3e12 * 1e6);
    for (int philox : {1, 0}) {
      if (sh.kind == GEMM_TN && philox) continue;
      int ns; double us;
#define REP(BM, BN, call) { g_bm = BM; GemmArgs g = make(sh.kind, sh.M, sh.N, sh.K, b, philox, &ns, BN); us = call; \
        printf("   %3dx%-3d philox %d slabs %2d: %7.1f us %6.1f TF\n", BM, BN, philox, ns, us, fl / us / 1e6); }
      if (sh.kind == GEMM_NT) {
        REP(128, 128, (run<GEMM_NT, 128, 128>(g, 1))) REP(64, 128, (run<GEMM_NT, 64, 128>(g, 1)))
        REP(128, 64, (run<GEMM_NT, 128, 64>(g, 1))) REP(64, 64, (run<GEMM_NT, 64, 64>(g, 1)))
        printf("   K depth 16:\n");
        REP(128, 128, (run<GEMM_NT, 128, 128, 16>(g, 1))) REP(64, 128, (run<GEMM_NT, 64, 128, 16>(g, 1))) REP(64, 64, (run<GEMM_NT, 64, 64, 16>(g, 1)))
      } else if (sh.kind == GEMM_NN) {
        REP(128, 128, (run<GEMM_NN, 128, 128>(g, 1))) REP(64, 128, (run<GEMM_NN, 64, 128>(g, 1)))
        REP(128, 64, (run<GEMM_NN, 128, 64>(g, 1))) REP(64, 64, (run<GEMM_NN, 64, 64>(g, 1)))
        printf("   K depth 16:\n");
        REP(128, 128, (run<GEMM_NN, 128, 128, 16>(g, 1))) REP(64, 128, (run<GEMM_NN, 64, 128, 16>(g, 1))) REP(64, 64, (run<GEMM_NN, 64, 64, 16>(g, 1)))
      } else {
        REP(128, 128, (run<GEMM_TN, 128, 128>(g, ns))) REP(128, 64, (run<GEMM_TN, 128, 64>(g, ns)))
        printf("   K depth 16:\n");
        REP(64, 64, (run<GEMM_TN, 64, 64, 16>(g, ns))) REP(128, 128, (run<GEMM_TN, 128, 128, 16>(g, ns)))
        for (int target : {512, 1024, 2048}) {     // workgroups in the launch: 64-row tiles need 4x fewer slabs for the same count
          g_target = target;
          REP(64, 64, (run<GEMM_TN, 64, 64>(g, ns))) REP(64, 128, (run<GEMM_TN, 64, 128>(g, ns))) REP(128, 128, (run<GEMM_TN, 128, 128>(g, ns)))
        }
        g_target = 512;
      }
#undef REP
    }
  }
  printf("LDS-DMA 64x64 kernel vs the register-staged 64x64 kernel (philox on)\n");
  for (const Shape& sh : shapes) {
    if (sh.kind == GEMM_TN) continue;
    int ns;
    for (int M : {sh.M, sh.M - 37}) {       // a ragged row count too
      GemmArgs g = make(sh.kind, M, sh.N, sh.K, b, true, &ns, 64);
      const double fl = 2.0 * M * sh.N * sh.K;
      double t0, t1, err;
      if (sh.kind == GEMM_NT) { t0 = run<GEMM_NT, 64, 64>(g, 1); t1 = run_dma<GEMM_NT>(g); err = check_dma<GEMM_NT>(g, (size_t)M * sh.N); }
      else { t0 = run<GEMM_NN, 64, 64>(g, 1); t1 = run_dma<GEMM_NN>(g); err = check_dma<GEMM_NN>(g, (size_t)M * sh.N); }
      printf("   %s M=%d: staged %7.1f us %6.1f TF   dma %7.1f us %6.1f TF   max rel diff %.2e\n", sh.name, M, t0, fl / t0 / 1e6, t1, fl / t1 / 1e6, err);
    }
  }
  printf("K sweep, NT, M=16384 N=512, philox on\n");
  for (int K : {64, 128, 256, 512, 1024, 2048}) {
    int ns; GemmArgs g = make(GEMM_NT, 16384, 512, K, b, true, &ns, 128);
    const double a = run<GEMM_NT, 128, 128>(g, 1), c = run<GEMM_NT, 64, 128>(g, 1), d = run<GEMM_NT, 64, 64>(g, 1);
    const double id = 2.0 * 16384 * 512 * K / 157.3e12 * 1e6;
    printf("   K %4d ideal %6.1f us: 128x128 %7.1f us   64x128 %7.1f us   64x64 %7.1f us\n", K, id, a, c, d);
  }
  printf("K sweep, NT, M=32768 N=256, philox on\n");
  for (int K : {64, 128, 256, 512, 1024}) {
    int ns; GemmArgs g = make(GEMM_NT, 32768, 256, K, b, true, &ns, 128);
    const double a = run<GEMM_NT, 128, 128>(g, 1), c = run<GEMM_NT, 64, 128>(g, 1), d = run<GEMM_NT, 64, 64>(g, 1);
    const double id = 2.0 * 32768 * 256 * K / 157.3e12 * 1e6;
    printf("   K %4d ideal %6.1f us: 128x128 %7.1f us   64x128 %7.1f us   64x64 %7.1f us\n", K, id, a, c, d);
  }
  return 0;
}
