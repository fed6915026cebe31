// Ablation harness for the f32 MFMA GEMM (GPU box only): hipcc -DGT_ABLATE_... tools/gemm_ablate.hip
// Prints kernel time / TFLOP/s for the NT 128x128 kernel at M=16384,N=512,K=512 (wrong results when ablated).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include "../gantts_amd/csrc/gemm_f32.hip.h"
using namespace gt;
#ifndef VEC
#define VEC true
#endif
int main(int argc, char** argv) {
  const int M = 16384, N = 512, K = argc > 1 ? atoi(argv[1]) : 512;
  float *A, *B, *Cc, *bias;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&Cc, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
  hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(B, 0x3c, (size_t)N * K * 4); hipMemset(bias, 0, N * 4);
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = A; g.lda = K; g.B = B; g.ldb = K; g.C = Cc; g.ldc = N; g.M = M; g.N = N; g.K = K; g.bias = bias; g.act = ACT_NONE;
  g.drop.mode = DROP_NONE; g.drop.scale = 1.f; g.wide_store = getenv("NOWIDE") ? 0 : 1; g.n_tiles_m = M / 128; g.n_tiles_n = N / 128;
  const size_t lds = gemm_lds_bytes<GEMM_NT, 128, 128>();
  hipFuncSetAttribute((const void*)gemm_f32_kernel<GEMM_NT, 128, 128, VEC, VEC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int grid_div = 1; grid_div <= 2; grid_div *= 2) {
    const int grid = g.n_tiles_m * g.n_tiles_n / grid_div;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gemm_f32_kernel<GEMM_NT, 128, 128, VEC, VEC>), dim3(grid), dim3(256), lds, 0, g);
    hipEventRecord(e0);
    const int it = 20;
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL((gemm_f32_kernel<GEMM_NT, 128, 128, VEC, VEC>), dim3(grid), dim3(256), lds, 0, g);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it;
    printf("K=%d grid %4d: %.1f us/launch  %.1f TFLOP/s (of tiles launched)\n", K, grid, us, 2.0 * M * N * K / grid_div / (us * 1e-6) / 1e12);
  }
  return 0;
}
