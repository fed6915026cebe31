// A/B harness (no torch): f32 MFMA GEMM launches of the cfg2 step with the start stagger of gemm_f32.hip.h
// (who-is-late modes x delays) on random operands.   usage: gemm_stagger_bench [reps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../gantts_amd/csrc/gemm_f32.hip.h"
using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static float* dfill(size_t n, float scale, unsigned s) {
  std::vector<float> h(n);
  const bool half_zero = getenv("GT_SB_HALF_ZERO") != nullptr;     // like post-dropout activations: every other value exactly 0
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f) * scale; if (half_zero && (s & 0x10000u)) v = 0.f; }
  float* p; CK(hipMalloc((void**)&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
__global__ void census_kernel(unsigned* keys, unsigned* hw) { if (threadIdx.x == 0) { keys[blockIdx.x] = cu_key(); hw[blockIdx.x] = hw_id_reg(); } }

template <int KIND, int BM, int BN>
static double run(GemmArgs g, int nslab, int reps, unsigned* ticket) {
  const size_t lds = gemm_lds_bytes<KIND, BM, BN>();
  auto kern = gemm_f32_kernel<KIND, BM, BN, true, true>;
  CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  g.n_tiles_m = (g.M + BM - 1) / BM; g.n_tiles_n = (g.N + BN - 1) / BN;
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, g);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  return ms * 1e3 / reps;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 30;
  // census: what does a 512-workgroup launch look like to cu_key()?
  {
    unsigned *dk, *dh; CK(hipMalloc((void**)&dk, 1024 * 4)); CK(hipMalloc((void**)&dh, 1024 * 4));
    hipLaunchKernelGGL(census_kernel, dim3(512), dim3(256), 60000, 0, dk, dh);
    std::vector<unsigned> k(512), h(512);
    CK(hipMemcpy(k.data(), dk, 512 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h.data(), dh, 512 * 4, hipMemcpyDeviceToHost));
    std::vector<int> cnt(2048, 0);
    for (auto v : k) cnt[v & 2047]++;
    int distinct = 0, twos = 0; for (int c : cnt) { if (c) ++distinct; if (c == 2) ++twos; }
    printf("census: 512 workgroups (60 KB LDS each) -> %d distinct CU keys, %d keys with exactly 2\n", distinct, twos);
    printf("  first blocks: "); for (int b = 0; b < 12; ++b) printf("[b%d key %03x hw %08x] ", b, k[b], h[b]); printf("\n");
    printf("  blocks 256..: "); for (int b = 256; b < 262; ++b) printf("[b%d key %03x hw %08x] ", b, k[b], h[b]); printf("\n");
    int same = 0; for (int b = 0; b < 256; ++b) if (k[b] == k[b + 256]) ++same;
    printf("  blocks b and b+256 on the same CU: %d of 256;  wave-slot parity of b<256: ", same);
    int odd = 0; for (int b = 0; b < 256; ++b) odd += h[b] & 1; printf("%d odd, of b>=256: ", odd);
    odd = 0; for (int b = 256; b < 512; ++b) odd += h[b] & 1; printf("%d odd\n", odd);
  }
  unsigned* ticket; CK(hipMalloc((void**)&ticket, 2048 * 4)); CK(hipMemset(ticket, 0, 2048 * 4));
  struct Shape { const char* name; int kind, M, N, K, bm; };
  const Shape shapes[] = {
    {"fwd  G  16384x512x512  (128x128)", GEMM_NT, 16384, 512, 512, 128},
    {"fwd  D  32768x256x256  (128x128)", GEMM_NT, 32768, 256, 256, 128},
    {"fwd  D  16384x256x256  ( 64x128)", GEMM_NT, 16384, 256, 256, 64},
    {"bwdX G  16384x512x512  (128x128)", GEMM_NN, 16384, 512, 512, 128},
    {"bwdW G  512x512x16384  (128x128, 32 slabs)", GEMM_TN, 512, 512, 16384, 128},
  };
  for (const Shape& sh : shapes) {
    const bool tn = sh.kind == GEMM_TN;
    float* A = dfill((size_t)(tn ? sh.K * sh.M : sh.M * sh.K), 1.f, 1);
    float* B = dfill((size_t)sh.N * sh.K, 0.05f, 2);
    float* Cc; CK(hipMalloc((void**)&Cc, (size_t)(tn ? 32 : 1) * sh.M * sh.N * 4 + 64));
    float* Hh = dfill((size_t)sh.M * sh.N, 1.f, 3);
    float* bias = dfill(sh.N, 0.1f, 4);
    GemmArgs g; memset(&g, 0, sizeof(g));
    g.M = sh.M; g.N = sh.N; g.K = sh.K; g.C = Cc; g.ldc = sh.N; g.wide_store = 1;
    g.drop.mode = DROP_PHILOX; g.drop.p = 0.5f; g.drop.scale = 2.f; g.drop.thresh = 32768; g.drop.key0 = 123; g.drop.key1 = 456;
    int nslab = 1;
    if (sh.kind == GEMM_NT) { g.A = A; g.lda = sh.K; g.B = B; g.ldb = sh.K; g.bias = bias; g.act = ACT_LEAKY_DROPOUT; }
    else if (sh.kind == GEMM_NN) { g.A = A; g.lda = sh.K; g.B = B; g.ldb = sh.N; g.act = ACT_LEAKY_DROPOUT; g.H = Hh; g.ldh = sh.N; }
    else { g.A = A; g.lda = sh.M; g.B = B; g.ldb = sh.N; nslab = 32; g.k_chunk = sh.K / 32; g.slab_stride = (long)sh.M * sh.N; g.drop.mode = DROP_NONE; g.act = ACT_NONE; }
    const double flops = 2.0 * sh.M * sh.N * sh.K;
    printf("%s   ideal %.1f us\n", sh.name, flops / 157.3e12 * 1e6);
    unsigned* dbg; CK(hipMalloc((void**)&dbg, 4096 * 4));
    for (int mode : {0, 2}) {
      for (int us10 : {1, 2, 3, 5, 10, 20}) {           // delay in units of 0.1 us
        if (mode == 0 && us10 != 1) continue;
        g.stagger_mode = mode; g.stagger_ticket = ticket; g.stagger_dbg = dbg;
        g.stagger_ticks = mode == 0 ? 0 : us10 * 10;             // 100 MHz ticks
        double us;
        if (sh.kind == GEMM_NT) us = sh.bm == 128 ? run<GEMM_NT, 128, 128>(g, 1, reps, ticket) : run<GEMM_NT, 64, 128>(g, 1, reps, ticket);
        else if (sh.kind == GEMM_NN) us = run<GEMM_NN, 128, 128>(g, 1, reps, ticket);
        else us = run<GEMM_TN, 128, 128>(g, nslab, reps, ticket);
        // who was late in the LAST launch: CUs with exactly one late workgroup of their two
        std::vector<unsigned> hd(4096); CK(hipMemcpy(hd.data(), dbg, 4096 * 4, hipMemcpyDeviceToHost));
        const int grid = ((sh.M + sh.bm - 1) / sh.bm) * ((sh.N + 127) / 128) * nslab;
        std::vector<int> tot(2048, 0), lat(2048, 0);
        for (int b = 0; b < grid && b < 4096; ++b) { tot[hd[b] & 2047]++; lat[hd[b] & 2047] += (hd[b] >> 16) & 1; }
        int cus = 0, one = 0; for (int k = 0; k < 2048; ++k) if (tot[k]) { ++cus; if (tot[k] == 2 && lat[k] == 1) ++one; }
        printf("   mode %d delay %5.1f us: %7.1f us  %6.1f TFLOP/s   (%d CUs, %d with one early + one late)\n", mode, mode ? us10 / 10.0 : 0.0, us,
               flops / (us * 1e-6) / 1e12, cus, mode ? one : 0);
      }
    }
    hipFree(dbg);
    hipFree(A); hipFree(B); hipFree(Cc); hipFree(Hh); hipFree(bias);
  }
  // ---- mixed sequence: the five launches one after the other (as inside a training step), repeated
  {
    struct Buf { float *A, *B, *C, *H, *bias; };
    std::vector<Buf> bufs; std::vector<GemmArgs> gs; std::vector<int> slabs;
    for (const Shape& sh : shapes) {
      const bool tn = sh.kind == GEMM_TN;
      Buf b; b.A = dfill((size_t)(tn ? sh.K * sh.M : sh.M * sh.K), 1.f, 1); b.B = dfill((size_t)sh.N * sh.K, 0.05f, 2);
      CK(hipMalloc((void**)&b.C, (size_t)(tn ? 32 : 1) * sh.M * sh.N * 4 + 64)); b.H = dfill((size_t)sh.M * sh.N, 1.f, 3); b.bias = dfill(sh.N, 0.1f, 4);
      GemmArgs g; memset(&g, 0, sizeof(g));
      g.M = sh.M; g.N = sh.N; g.K = sh.K; g.C = b.C; g.ldc = sh.N; g.wide_store = 1;
      g.drop.mode = DROP_PHILOX; g.drop.p = 0.5f; g.drop.scale = 2.f; g.drop.thresh = 32768; g.drop.key0 = 123; g.drop.key1 = 456;
      int nslab = 1;
      if (sh.kind == GEMM_NT) { g.A = b.A; g.lda = sh.K; g.B = b.B; g.ldb = sh.K; g.bias = b.bias; g.act = ACT_LEAKY_DROPOUT; }
      else if (sh.kind == GEMM_NN) { g.A = b.A; g.lda = sh.K; g.B = b.B; g.ldb = sh.N; g.act = ACT_LEAKY_DROPOUT; g.H = b.H; g.ldh = sh.N; }
      else { g.A = b.A; g.lda = sh.M; g.B = b.B; g.ldb = sh.N; nslab = 32; g.k_chunk = sh.K / 32; g.slab_stride = (long)sh.M * sh.N; g.drop.mode = DROP_NONE; g.act = ACT_NONE; }
      bufs.push_back(b); gs.push_back(g); slabs.push_back(nslab);
    }
    auto launch_one = [&](int i, int mode, int ticks) {
      GemmArgs g = gs[i]; g.stagger_mode = mode; g.stagger_ticks = ticks; g.stagger_ticket = ticket;
      const Shape& sh = shapes[i];
#define L1(KIND, BM) { const size_t lds = gemm_lds_bytes<KIND, BM, 128>(); auto kern = gemm_f32_kernel<KIND, BM, 128, true, true>; \
        g.n_tiles_m = (g.M + BM - 1) / BM; g.n_tiles_n = (g.N + 127) / 128; \
        hipLaunchKernelGGL(kern, dim3(g.n_tiles_m * g.n_tiles_n * slabs[i]), dim3(256), lds, 0, g); }
      if (sh.kind == GEMM_NT) { if (sh.bm == 128) L1(GEMM_NT, 128) else L1(GEMM_NT, 64) }
      else if (sh.kind == GEMM_NN) L1(GEMM_NN, 128)
      else L1(GEMM_TN, 128)
#undef L1
    };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    double fl = 0; for (const Shape& sh : shapes) fl += 2.0 * sh.M * sh.N * sh.K;
    for (int mode : {0, 2, 3}) for (int ticks : {50, 100, 200}) {
      if (mode == 0 && ticks != 50) continue;
      for (int it = 0; it < 3; ++it) for (int i = 0; i < 5; ++i) launch_one(i, mode, mode ? ticks : 0);
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < reps; ++it) for (int i = 0; i < 5; ++i) launch_one(i, mode, mode ? ticks : 0);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mixed sequence of the 5 launches: mode %d delay %.1f us: %.1f us per sequence  %.1f TFLOP/s\n", mode, mode ? ticks / 100.0 : 0.0,
             ms * 1e3 / reps, fl / (ms * 1e-3 / reps) / 1e12);
    }
  }
  return 0;
}
