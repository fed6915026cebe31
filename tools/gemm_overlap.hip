// Experiment (GPU box only): does splitting a GEMM's rows over S streams (kernels desynchronised, so one's
// epilogue overlaps another's main loop) beat one launch?  NT 128x128, M=16384, N=512, K arg.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../gantts_amd/csrc/gemm_f32.hip.h"
using namespace gt;
int main(int argc, char** argv) {
  const int M = 16384, N = argc > 2 ? atoi(argv[2]) : 512, K = argc > 1 ? atoi(argv[1]) : 512;
  float *A, *B, *Cc, *bias;
  hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4); hipMalloc(&Cc, (size_t)M * N * 4); hipMalloc(&bias, N * 4);
  hipMemset(A, 0x3c, (size_t)M * K * 4); hipMemset(B, 0x3c, (size_t)N * K * 4); hipMemset(bias, 0, N * 4);
  const size_t lds = gemm_lds_bytes<GEMM_NT, 128, 128>();
  auto kern = gemm_f32_kernel<GEMM_NT, 128, 128, true, true>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipStream_t st[8];
  for (int i = 0; i < 8; ++i) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
  hipEvent_t e0, e1, ej[8]; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 8; ++i) hipEventCreateWithFlags(&ej[i], hipEventDisableTiming);
  for (int parts : {1, 2, 4, 8}) {
    for (int act : {0, 1}) {
      const int rows = M / parts;
      auto launch_all = [&]() {
        for (int p = 0; p < parts; ++p) {
          GemmArgs g; memset(&g, 0, sizeof(g));
          g.A = A + (size_t)p * rows * K; g.lda = K; g.B = B; g.ldb = K; g.C = Cc + (size_t)p * rows * N; g.ldc = N;
          g.M = rows; g.N = N; g.K = K; g.bias = bias; g.act = act ? ACT_LEAKY_DROPOUT : ACT_NONE;
          g.drop.mode = act ? DROP_PHILOX : DROP_NONE; g.drop.scale = 2.f; g.drop.p = 0.5f; g.drop.thresh = 0x8000u;
          g.drop.key0 = 123; g.drop.key1 = 456;
          g.wide_store = getenv("NOWIDE") ? 0 : 1; g.n_tiles_m = rows / 128; g.n_tiles_n = N / 128;
          hipLaunchKernelGGL(kern, dim3(g.n_tiles_m * g.n_tiles_n), dim3(256), lds, st[p], g);
        }
      };
      // chain of `it` dependent "layers": each layer = `parts` launches on their own streams (stream order = dependency)
      for (int i = 0; i < 3; ++i) launch_all();
      hipDeviceSynchronize();
      hipEventRecord(e0, st[0]);
      for (int p = 1; p < parts; ++p) hipStreamWaitEvent(st[p], e0, 0);
      const int it = 20;
      for (int i = 0; i < it; ++i) launch_all();
      for (int p = 1; p < parts; ++p) { hipEventRecord(ej[p], st[p]); hipStreamWaitEvent(st[0], ej[p], 0); }
      hipEventRecord(e1, st[0]); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / it;
      printf("K=%d N=%d act=%d parts=%d: %.1f us/layer  %.1f TFLOP/s\n", K, N, act, parts, us, 2.0 * M * N * K / (us * 1e-6) / 1e12);
    }
  }
  return 0;
}
