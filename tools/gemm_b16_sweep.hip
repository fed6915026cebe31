// Tile-shape sweep of the bf16-storage product kernel (gemm_bf16s.hip.h) on the large shapes of the SRU step (no torch):
//   forward U = x W (32768 x 3072 x 1024), backward-data (32768 x 1024 x 3072), weight gradient (1024 x 3072 over 32768 frames)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_b16_sweep.hip -o tools/bin/gemm_b16_sweep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../gantts_amd/csrc/gemm_bf16s.hip.h"
#include "experiments/gemm_b16_big.hip.h"
using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static __bf16* dfill(size_t n, unsigned seed) {
  std::vector<unsigned short> h(n);
  for (auto& v : h) { seed = seed * 1664525u + 1013904223u; const float f = ((seed >> 8) & 0xffff) / 65536.f - 0.5f; unsigned u; memcpy(&u, &f, 4); v = (unsigned short)(u >> 16); }
  __bf16* p; CK(hipMalloc((void**)&p, n * 2)); CK(hipMemcpy(p, h.data(), n * 2, hipMemcpyHostToDevice));
  return p;
}
template <int BM, int BN, int EPI, int PF = 1>
static float run(GemmB16Args g, int nslab, hipStream_t s, int reps) {
  const size_t lds = gemm_b16_lds_bytes<BM, BN>();
  const void* k = (const void*)gemm_b16_kernel<BM, BN, EPI, B16_A_NONE, PF>;
  CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + BM - 1) / BM; g.n_tiles_n = (g.N + BN - 1) / BN;
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((gemm_b16_kernel<BM, BN, EPI, B16_A_NONE, PF>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, GEMM_THREADS, lds));
  printf("    %3d x %3d pf%d  grid %6d  %d wg/cu  %8.1f us  %7.1f TFLOP/s\n", BM, BN, PF, grid, occ, best * 1e3, 2.0 * g.M * g.N * g.K / best / 1e9);
  return best;
}
template <int BM, int BN, int EPI, int NS, int WGM = 2, int WGN = 2>
static float run_dma(GemmB16Args g, int nslab, hipStream_t s, int reps) {
  const size_t lds = gemm_b16_dma_lds_bytes<BM, BN, NS>();
  const void* k = (const void*)gemm_b16_dma_kernel<BM, BN, EPI, B16_A_NONE, NS, WGM, WGN>;
  CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + BM - 1) / BM; g.n_tiles_n = (g.N + BN - 1) / BN;
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((gemm_b16_dma_kernel<BM, BN, EPI, B16_A_NONE, NS, WGM, WGN>), dim3(grid), dim3(64 * WGM * WGN), lds, s, g);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  int occ = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k, 64 * WGM * WGN, lds));
  printf("    %3d x %3d dma ring %d, %d x %d waves  grid %6d  %d wg/cu  %8.1f us  %7.1f TFLOP/s\n", BM, BN, NS, WGM, WGN, grid, occ, best * 1e3, 2.0 * g.M * g.N * g.K / best / 1e9);
  return best;
}
template <int EPI>
static float run_big(GemmB16Args g, int nslab, hipStream_t s, int reps) {
  const size_t lds = gemm_b16_big_lds_bytes();
  const void* k = (const void*)gemm_b16_big_kernel<EPI, B16_A_NONE>;
  CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + 255) / 256; g.n_tiles_n = (g.N + 255) / 256;
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int r = 0; r < reps + 1; ++r) {
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((gemm_b16_big_kernel<EPI, B16_A_NONE>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (r > 0) best = std::min(best, ms);
  }
  CK(hipGetLastError());
  printf("    256 x 256 big (ring 4 x 32 k, pipelined fragments)  grid %6d  %8.1f us  %7.1f TFLOP/s\n", grid, best * 1e3, 2.0 * g.M * g.N * g.K / best / 1e9);
  return best;
}
static double checksum(const float* d, size_t n) {
  std::vector<float> h(n); CK(hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost));
  double s = 0; for (size_t i = 0; i < n; i += 97) s += h[i];
  return s;
}
int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  const int M = 32768, N = 3072, K = 1024;
  __bf16* X = dfill((size_t)M * K, 1); __bf16* W = dfill((size_t)N * K, 2);      // forward: A = x [M][K], B = WT [N][K]
  __bf16* DU = dfill((size_t)M * N, 3); __bf16* W2 = dfill((size_t)K * N, 4);    // backward-data: A = dU [M][N], B = W [K][N]
  __bf16* XT = dfill((size_t)K * M, 5); __bf16* DUT = dfill((size_t)N * M, 6);   // weight gradient: A = xT [K][M], B = dUT [N][M]
  float* C; CK(hipMalloc((void**)&C, (size_t)M * N * 4 * 2));
  GemmB16Args g; memset(&g, 0, sizeof(g)); g.drop.mode = DROP_NONE; g.drop.scale = 1.f;
#define CSUM printf("      checksum %.9e\n", checksum(C, (size_t)g.M * g.N));
  // every tile shape / loader on one product: register-staged 64 and 128 tiles (1 / 2 stages in flight), LDS-DMA rings with 4- and
  // 8-wave workgroups, the 256 x 256 tile, the experiment of tools/experiments/gemm_b16_big.hip.h
#define VARIANTS(EPI) { run<64, 64, EPI, 1>(g, 1, s, 3); CSUM run<128, 128, EPI, 1>(g, 1, s, 3); CSUM run<128, 128, EPI, 2>(g, 1, s, 3); CSUM \
    run_dma<128, 128, EPI, 2>(g, 1, s, 3); CSUM run_dma<128, 128, EPI, 3>(g, 1, s, 3); CSUM run_dma<128, 128, EPI, 2, 2, 4>(g, 1, s, 3); CSUM \
    run_dma<256, 256, EPI, 2>(g, 1, s, 3); CSUM run_dma<256, 256, EPI, 2, 2, 4>(g, 1, s, 3); CSUM run_big<EPI>(g, 1, s, 3); CSUM }
  printf("forward 32768 x 3072 x 1024 (float32 result)\n");
  g.A = X; g.lda = K; g.B = W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.C = C; g.ldc = N; g.epi = B16_FWD; g.act = ACT_NONE;
  VARIANTS(B16_FWD)
  printf("  ... with the L2-aware band order (band_c = 4)\n");
  g.band_c = 4; run_dma<128, 128, B16_FWD, 2>(g, 1, s, 3); run_dma<256, 256, B16_FWD, 2, 2, 4>(g, 1, s, 3); g.band_c = 0;
  printf("backward-data 32768 x 1024 x 3072 (float32 result)\n");
  g.A = DU; g.lda = N; g.B = W2; g.ldb = N; g.M = M; g.N = K; g.K = N; g.C = C; g.ldc = K; g.epi = B16_BWD_DATA;
  VARIANTS(B16_BWD_DATA)
  printf("weight gradient 1024 x 3072 over 32768 frames, by slab count\n");
  g.A = XT; g.lda = M; g.B = DUT; g.ldb = M; g.M = K; g.N = N; g.K = M; g.C = C; g.ldc = N; g.epi = B16_SLAB; g.slab_stride = (long)K * N;
  for (int ns : {1, 2, 4, 5, 8, 16}) {
    g.k_chunk = ((M / ns + 63) / 64) * 64;
    printf("  %d slabs (k_chunk %d)\n", ns, g.k_chunk);
    run<64, 64, B16_SLAB, 1>(g, ns, s, 3); run<128, 128, B16_SLAB, 1>(g, ns, s, 3); run_dma<128, 128, B16_SLAB, 2, 2, 4>(g, ns, s, 3); run_dma<256, 256, B16_SLAB, 2, 2, 4>(g, ns, s, 3);
  }
  printf("weight gradient 512 x 2048 over 32768 frames (cfg3 W_ih), by slab count\n");
  g.M = 512; g.N = 2048; g.ldc = 2048; g.slab_stride = 512L * 2048;
  for (int ns : {2, 4, 8, 16}) {
    g.k_chunk = ((M / ns + 63) / 64) * 64;
    printf("  %d slabs\n", ns);
    run<64, 64, B16_SLAB, 1>(g, ns, s, 3); run<128, 128, B16_SLAB, 1>(g, ns, s, 3); run_dma<128, 128, B16_SLAB, 2, 2, 4>(g, ns, s, 3);
  }
  g.k_chunk = 0;
  printf("cfg2 forward 16384 x 512 x 512\n");
  g.A = X; g.lda = 512; g.B = W; g.ldb = 512; g.M = 16384; g.N = 512; g.K = 512; g.C = C; g.ldc = 512; g.epi = B16_FWD;
  run<64, 64, B16_FWD, 1>(g, 1, s, 5); run<128, 128, B16_FWD, 1>(g, 1, s, 5); run_dma<128, 128, B16_FWD, 2>(g, 1, s, 5); run_dma<128, 128, B16_FWD, 2, 2, 4>(g, 1, s, 5);
  run_dma<64, 64, B16_FWD, 2>(g, 1, s, 5); run_dma<256, 256, B16_FWD, 2, 2, 4>(g, 1, s, 5);
  // a float32 product emulated by three bf16 pieces per operand and six piece products IS a bf16 product over
  // the K-concatenated piece images [A0|A0|A1|A0|A1|A2] . [B0|B1|B0|B2|B1|B0]^T, K' = 6 K: the cfg2 hidden layer as 16384 x 512 x 3072.
  // (A dedicated kernel would fetch three pieces instead of six images -- the MFMA count is the same.)  Gate: <= 45 us, the f32 launch is 76.
  printf("bf16x6 gate: 16384 x 512 x 3072 (six piece products of the f32 product 16384 x 512 x 512; the f32 MFMA launch takes 76 us)\n");
  g.A = DU; g.lda = 3072; g.B = W; g.ldb = 3072; g.M = 16384; g.N = 512; g.K = 3072; g.C = C; g.ldc = 512; g.epi = B16_FWD;
  run<64, 64, B16_FWD, 1>(g, 1, s, 5); run<128, 128, B16_FWD, 1>(g, 1, s, 5); run<128, 128, B16_FWD, 2>(g, 1, s, 5); run_dma<128, 128, B16_FWD, 2>(g, 1, s, 5);
  run_dma<128, 128, B16_FWD, 3>(g, 1, s, 5); run_dma<128, 128, B16_FWD, 2, 2, 4>(g, 1, s, 5); run_dma<256, 256, B16_FWD, 2, 2, 4>(g, 1, s, 5); run_big<B16_FWD>(g, 1, s, 5);
  g.K = 512; g.lda = 512; g.ldb = 512; g.A = X;
  printf("cfg3 X-projection 32768 x 2048 x 512\n");
  g.M = 32768; g.N = 2048; g.ldc = 2048;
  run<128, 128, B16_FWD, 1>(g, 1, s, 3); run_dma<128, 128, B16_FWD, 2>(g, 1, s, 3); run_dma<128, 128, B16_FWD, 2, 2, 4>(g, 1, s, 3); run_dma<256, 256, B16_FWD, 2, 2, 4>(g, 1, s, 3);
#ifdef GT_B16_CLK_DBG
  {   // shader clock under this load: per-workgroup clock64() / wall_clock64() deltas around the K loop
    unsigned long long* dbg; CK(hipMalloc((void**)&dbg, 16 * 8192));
    printf("forward 32768 x 3072 x 1024, 128 x 128 dma ring 2, clock probe\n");
    g.A = X; g.lda = K; g.B = W; g.ldb = K; g.M = M; g.N = N; g.K = K; g.C = C; g.ldc = N; g.epi = B16_FWD;
    CK(hipMemset(dbg, 0, 16 * 8192));
    g.rowsum_slab = reinterpret_cast<float*>(dbg);
    run_dma<128, 128, B16_FWD, 2>(g, 1, s, 1);
    std::vector<unsigned long long> h(2 * 6144); CK(hipMemcpy(h.data(), dbg, 16 * 6144, hipMemcpyDeviceToHost));
    double mhz_min = 1e9, mhz_max = 0, mhz_sum = 0, us_sum = 0; int n = 0;
    for (int i = 0; i < 6144; ++i) if (h[2 * i + 1]) { const double mhz = 100.0 * h[2 * i] / h[2 * i + 1]; mhz_min = std::min(mhz_min, mhz); mhz_max = std::max(mhz_max, mhz); mhz_sum += mhz; us_sum += h[2 * i + 1] / 100.0; ++n; }
    printf("    %d workgroups: shader clock %.0f .. %.0f MHz, mean %.0f; K loop of a tile %.1f us on average (16 stages)\n", n, mhz_min, mhz_max, mhz_sum / n, us_sum / n);
    g.rowsum_slab = nullptr;
  }
#endif
  return 0;
}
