// Microbenchmark of the discriminator panel-chain kernels (GPU box only).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/chain_bench.hip -o /tmp/chain_bench && /tmp/chain_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../gantts_amd/csrc/chain_kernels.hip.h"
using namespace gt;
#ifndef CHAIN_LAUNCH
#define CHAIN_LAUNCH(FWD, a, lds) hipLaunchKernelGGL((FWD ? chain_fwd_kernel<2> : chain_bwd_kernel<2>), dim3((unsigned)((a.rows + CH_ROWS - 1) / CH_ROWS)), dim3(CH_THREADS), lds, 0, a)
#endif
int main(int argc, char** argv) {
  const int H = 256, K0 = 58, K0p = 64;
  float *W0, *W, *bias, *P, *A0, *act[3], *dz[3];
  hipMalloc(&W0, H * K0p * 4); hipMalloc(&W, (size_t)H * H * 4); hipMalloc(&bias, H * 4);
  hipMemset(W0, 0x3c, H * K0p * 4); hipMemset(W, 0x3c, (size_t)H * H * 4); hipMemset(bias, 0, H * 4);
  const long maxrows = 32768;
  hipMalloc(&P, maxrows * H * 4); hipMemset(P, 0, maxrows * H * 4);
  hipMalloc(&A0, maxrows * 60 * 4); hipMemset(A0, 0x3c, maxrows * 60 * 4);
  for (int i = 0; i < 3; ++i) { hipMalloc(&act[i], maxrows * H * 4); hipMalloc(&dz[i], maxrows * H * 4); hipMemset(act[i], 0x3c, maxrows * H * 4); hipMemset(dz[i], 0x3c, maxrows * H * 4); }
  hipFuncSetAttribute((const void*)chain_fwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  hipFuncSetAttribute((const void*)chain_bwd_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  long long* dbg; hipMalloc(&dbg, 1024 * 16 * 8); hipMemset(dbg, 0, 1024 * 16 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (long rows : {16384L, 32768L}) {
    for (int fwd = 1; fwd >= 0; --fwd) {
      ChainArgs a; memset(&a, 0, sizeof(a));
      a.rows = rows; a.H = H; a.dbg = dbg;
      DropoutSpec d; memset(&d, 0, sizeof(d)); d.mode = getenv("NODROP") ? DROP_NONE : DROP_PHILOX; d.p = 0.5f; d.scale = 2.f; d.thresh = 0x8000u; d.key0 = 1; d.key1 = 2;
      double flops;
      if (fwd) {
        a.K0 = K0; a.K0p = K0p; a.A0 = A0; a.lda0 = 60; a.P = P; a.ldp = H; a.p_mod = 16384; a.n_stages = 3;
        for (int l = 0; l < 3; ++l) { a.st[l].W = l ? W : W0; a.st[l].ldw = l ? H : K0p; a.st[l].bias = bias; a.st[l].out = act[l]; a.st[l].ldo = H; a.st[l].act = getenv("NOACT") ? ACT_NONE : ACT_LEAKY_DROPOUT; a.st[l].drop = d; }
        flops = 2.0 * rows * H * (K0p + 2.0 * H);
      } else {
        a.A0 = dz[2]; a.lda0 = H; a.n_stages = 2;
        for (int l = 0; l < 2; ++l) { a.st[l].W = W; a.st[l].ldw = H; a.st[l].Hact = act[l]; a.st[l].ldh = H; a.st[l].out = dz[l]; a.st[l].ldo = H; a.st[l].act = getenv("NOACT") ? ACT_NONE : ACT_LEAKY_DROPOUT; a.st[l].drop = d; }
        flops = 2.0 * rows * H * (2.0 * H);
      }
      const size_t lds = chain_lds_bytes(H, fwd ? K0p : 0, fwd != 0);
      for (int i = 0; i < 3; ++i) CHAIN_LAUNCH(fwd, a, lds);
      hipEventRecord(e0);
      const int it = 20;
      for (int i = 0; i < it; ++i) CHAIN_LAUNCH(fwd, a, lds);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double us = ms * 1e3 / it;
#ifdef CH_DEBUG_TIMING
      { long long h[1024 * 16]; hipMemcpy(h, dbg, sizeof(h), hipMemcpyDeviceToHost);
        for (int wg : {0, 255, (int)(rows / 64) - 1}) { printf("  wg %4d stamps (cycles since first):", wg); for (int i = 1; i < 16 && h[wg * 16 + i]; ++i) printf(" %lld", h[wg * 16 + i] - h[wg * 16]); printf("   | start offset vs wg0 %lld\n", h[wg * 16] - h[0]); } }
#endif
      printf("%s rows=%ld: %.1f us  %.1f TFLOP/s  (%s)\n", fwd ? "fwd(3 stages)" : "bwd(2 stages)", rows, us, flops / (us * 1e-6) / 1e12, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
