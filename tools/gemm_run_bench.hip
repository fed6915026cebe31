// Harness (no torch, GPU box only): the streamed tile runs of gantts_amd/csrc/gemm_f32_run.hip.h against the per-tile launches of
// gemm_f32.hip.h on the cfg2 shapes (+ ragged / K-tail shapes): results compared BIT FOR BIT, us per launch of both, and (RUN_DBG=1)
// per-workgroup wall-clock stamps of the streamed launch (start, first K stage done, last K stage done, end; tiles walked).
// usage: gemm_run_bench [reps]          env: RUN_DBG=1 stamps, RUN_GRID=<workgroups> (default 4 x CUs)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/gemm_run_bench.hip -o tools/bin/gemm_run_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>
#include <math.h>
#include "experiments/gemm_f32_run.hip.h"
using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
static float* dfill(size_t n, float scale, unsigned s) {
  std::vector<float> h(n);
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = (((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f) * scale; }
  float* p; CK(hipMalloc((void**)&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice)); return p;
}
static int g_reps = 30, g_grid = 1024, g_dbg = 0, g_wpe = 4;
static unsigned int* g_queues;
static RunDbg g_rd = {nullptr, nullptr, nullptr};

template <typename F>
static double time_us(F launch) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) launch();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < g_reps; ++i) launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipGetLastError());
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return ms * 1e3 / g_reps;
}
static void spin_up() {      // clocks: ~0.3 s of launches before anything is timed
  float* a = dfill((size_t)4096 * 512, 1.f, 9); float* b = dfill((size_t)512 * 512, 0.05f, 8); float* c; CK(hipMalloc((void**)&c, (size_t)4096 * 512 * 4));
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.A = a; g.lda = 512; g.B = b; g.ldb = 512; g.C = c; g.ldc = 512; g.M = 4096; g.N = 512; g.K = 512; g.wide_store = 1; g.n_tiles_m = 64; g.n_tiles_n = 8;
  const size_t lds = gemm_lds_bytes<GEMM_NT, 64, 64>();
  CK(hipFuncSetAttribute((const void*)gemm_f32_kernel<GEMM_NT, 64, 64, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int i = 0; i < 15000; ++i) hipLaunchKernelGGL((gemm_f32_kernel<GEMM_NT, 64, 64, true, true>), dim3(512), dim3(256), lds, 0, g);
  CK(hipDeviceSynchronize());
}
static void dump_stamps(const char* what, int grid) {
  if (!g_dbg) return;
  std::vector<unsigned long long> st((size_t)grid * 4); std::vector<unsigned> tl(grid);
  CK(hipMemcpy(st.data(), g_rd.stamps, st.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(tl.data(), g_rd.tiles, tl.size() * 4, hipMemcpyDeviceToHost));
  unsigned long long t0 = ~0ull, t1 = 0;
  for (int b = 0; b < grid; ++b) { t0 = std::min(t0, st[4 * b]); t1 = std::max(t1, st[4 * b + 3]); }
  std::vector<double> s0, s1, s2, s3; unsigned tmin = ~0u, tmax = 0; unsigned long tsum = 0;
  for (int b = 0; b < grid; ++b) {
    s0.push_back((st[4 * b] - t0) * 0.01);
    if (st[4 * b + 1]) s1.push_back((st[4 * b + 1] - t0) * 0.01);
    if (st[4 * b + 2]) s2.push_back((st[4 * b + 2] - t0) * 0.01);
    s3.push_back((st[4 * b + 3] - t0) * 0.01);
    tmin = std::min(tmin, tl[b]); tmax = std::max(tmax, tl[b]); tsum += tl[b];
  }
  auto pr = [](const char* n, std::vector<double>& v) { if (v.empty()) return; std::sort(v.begin(), v.end());
    printf("      %-22s min %6.1f  p10 %6.1f  median %6.1f  p90 %6.1f  max %6.1f us\n", n, v.front(), v[v.size() / 10], v[v.size() / 2], v[v.size() * 9 / 10], v.back()); };
  printf("    stamps of %s: span %.1f us, tiles per workgroup %u .. %u (sum %lu)\n", what, (t1 - t0) * 0.01, tmin, tmax, tsum);
  pr("workgroup start", s0); pr("first K stage done", s1); pr("last K stage done (ph0)", s2); pr("workgroup end", s3);
  std::vector<unsigned long long> tr(8 * 64);
  CK(hipMemcpy(tr.data(), g_rd.trace, tr.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemset(g_rd.trace, 0, 8 * 64 * 8));
  for (int i = 0; i < 8; ++i) {
    if (!tr[64 * i]) continue;
    printf("      stage ends of a workgroup (us after the launch's first start; then durations):");
    printf(" %.1f |", (tr[64 * i] - t0) * 0.01);
    for (int j = 1; j < 64 && tr[64 * i + j]; ++j) printf(" %.1f", (tr[64 * i + j] - tr[64 * i + j - 1]) * 0.01);
    printf("\n");
  }
}
static void set_tiles(GemmArgs& g) { g.n_tiles_m = (g.M + 63) / 64; g.n_tiles_n = (g.N + 63) / 64; }
static int run_grid(int items) { return std::min(g_grid, (items + 7) / 8 * 8); }

template <int KIND, int AMODE>
static void launch_old(GemmArgs g, int nslab) {
  const size_t lds = gemm_lds_bytes<KIND, 64, 64>();
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_f32_kernel<KIND, 64, 64, true, true, PREC_F32, 32, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  hipLaunchKernelGGL((gemm_f32_kernel<KIND, 64, 64, true, true, PREC_F32, 32, AMODE>), dim3(g.n_tiles_m * g.n_tiles_n * nslab), dim3(256), lds, 0, g);
}
template <int KIND, int AMODE>
static void launch_new(GemmArgs g, int nslab) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_run_kernel<KIND, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes()));
               CK(hipFuncSetAttribute((const void*)gemm_run_kernel<KIND, AMODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes())); once = true; }
  const int items = g.n_tiles_m * g.n_tiles_n * nslab;
  static bool once3 = false;
  if (g_wpe == 3 && !once3) { CK(hipFuncSetAttribute((const void*)gemm_run_kernel<KIND, AMODE, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes())); once3 = true; }
  if (g_dbg) hipLaunchKernelGGL((gemm_run_kernel<KIND, AMODE, true>), dim3(run_grid(items)), dim3(256), run_lds_bytes(), 0, g, items, g_queues, g_rd);
  else if (g_wpe == 3) hipLaunchKernelGGL((gemm_run_kernel<KIND, AMODE, false, 3>), dim3(run_grid(items)), dim3(256), run_lds_bytes(), 0, g, items, g_queues, g_rd);
  else hipLaunchKernelGGL((gemm_run_kernel<KIND, AMODE>), dim3(run_grid(items)), dim3(256), run_lds_bytes(), 0, g, items, g_queues, g_rd);
}
template <int AMODE>
static void launch_pair_old(GemmArgs nn, GemmArgs tn, int nslab) {
  const size_t lds = std::max(gemm_lds_bytes<GEMM_NN, 64, 64>(), gemm_lds_bytes<GEMM_TN, 64, 64>());
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_pair_kernel<PREC_F32, AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); once = true; }
  const int n1 = nn.n_tiles_m * nn.n_tiles_n, n2 = tn.n_tiles_m * tn.n_tiles_n * nslab;
  hipLaunchKernelGGL((gemm_pair_kernel<PREC_F32, AMODE>), dim3(n1 + n2), dim3(256), lds, 0, nn, tn, n1, 1);
}
template <int AMODE>
static void launch_pair_new(GemmArgs nn, GemmArgs tn, int nslab) {
  static bool once = false;
  if (!once) { CK(hipFuncSetAttribute((const void*)gemm_run_pair_kernel<AMODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes()));
               CK(hipFuncSetAttribute((const void*)gemm_run_pair_kernel<AMODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes())); once = true; }
  const int n1 = nn.n_tiles_m * nn.n_tiles_n, n2 = tn.n_tiles_m * tn.n_tiles_n * nslab;
  static bool once3 = false;
  if (g_wpe == 3 && !once3) { CK(hipFuncSetAttribute((const void*)gemm_run_pair_kernel<AMODE, false, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)run_lds_bytes())); once3 = true; }
  if (g_dbg) hipLaunchKernelGGL((gemm_run_pair_kernel<AMODE, true>), dim3(run_grid(n1 + n2)), dim3(256), run_lds_bytes(), 0, nn, tn, n1, n2, g_queues, g_rd);
  else if (g_wpe == 3) hipLaunchKernelGGL((gemm_run_pair_kernel<AMODE, false, 3>), dim3(run_grid(n1 + n2)), dim3(256), run_lds_bytes(), 0, nn, tn, n1, n2, g_queues, g_rd);
  else hipLaunchKernelGGL((gemm_run_pair_kernel<AMODE>), dim3(run_grid(n1 + n2)), dim3(256), run_lds_bytes(), 0, nn, tn, n1, n2, g_queues, g_rd);
}

struct Bufs { float *A, *B, *C, *C2, *H, *bias, *X, *S, *S2, *cs, *cs2; };
// compare two device buffers bit for bit; returns the number of differing words
static size_t diff_words(const float* a, const float* b, size_t n) {
  std::vector<unsigned> x(n), y(n);
  CK(hipMemcpy(x.data(), a, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b, n * 4, hipMemcpyDeviceToHost));
  size_t d = 0; for (size_t i = 0; i < n; ++i) d += x[i] != y[i] && ((x[i] | y[i]) & 0x7fffffffu) != 0u;     // (bit for bit, except that -0 == +0: a dropped
  return d;                                                                                                    //  element of a ragged backward-data tile is v * 0 in the per-tile kernel)
}
static GemmArgs mk(int kind, int M, int N, int K, int lda, int ldb, int ldc, const Bufs& b, bool philox, float* C) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.A = b.A; g.lda = lda; g.B = b.B; g.ldb = ldb; g.C = C; g.ldc = ldc; g.wide_store = (ldc % 4 == 0) ? 1 : 0;
  g.drop.mode = philox ? DROP_PHILOX : DROP_NONE; g.drop.p = 0.5f; g.drop.scale = 2.f; g.drop.thresh = 32768; g.drop.key0 = 123; g.drop.key1 = 456;
  g.act = philox ? ACT_LEAKY_DROPOUT : ACT_NONE;
  if (kind == GEMM_NT) g.bias = b.bias;
  if (kind == GEMM_NN && philox) { g.H = b.H; g.ldh = ldc; }
  set_tiles(g);
  return g;
}
static GemmArgs mk_tn(int M, int N, int K, int lda, int ldb, int target, const Bufs& b, float* slabs, float* cs, int* nslab) {
  GemmArgs g; memset(&g, 0, sizeof(g));
  g.M = M; g.N = N; g.K = K; g.A = b.A; g.lda = lda; g.B = b.X; g.ldb = ldb; g.C = slabs; g.ldc = N; g.drop.mode = DROP_NONE; g.drop.scale = 1.f; g.act = ACT_NONE;
  set_tiles(g);
  const int tiles = g.n_tiles_m * g.n_tiles_n;
  int ns = std::max(1, target / tiles); ns = std::min(ns, target > 512 ? K / 32 : (K + 255) / 256);
  const int kc = (((K + ns - 1) / ns) + 31) / 32 * 32; ns = (K + kc - 1) / kc;
  g.k_chunk = kc; g.slab_stride = (long)M * N; g.colsum_slab = cs; *nslab = ns;
  return g;
}

int main(int argc, char** argv) {
  g_reps = argc > 1 ? atoi(argv[1]) : 30;
  g_dbg = getenv("RUN_DBG") ? atoi(getenv("RUN_DBG")) : 0;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  g_wpe = getenv("RUN_WPE") ? atoi(getenv("RUN_WPE")) : 4;
  g_grid = getenv("RUN_GRID") ? atoi(getenv("RUN_GRID")) : prop.multiProcessorCount * g_wpe / 8 * 8;
  printf("device %s, %d CUs, streamed grid %d, reps %d\n", prop.name, prop.multiProcessorCount, g_grid, g_reps);
  CK(hipMalloc((void**)&g_queues, RUN_Q_WORDS * 4)); CK(hipMemset(g_queues, 0, RUN_Q_WORDS * 4));
  if (g_dbg) { CK(hipMalloc((void**)&g_rd.stamps, (size_t)g_grid * 4 * 8)); CK(hipMalloc((void**)&g_rd.tiles, (size_t)g_grid * 4)); CK(hipMalloc((void**)&g_rd.trace, 8 * 64 * 8)); CK(hipMemset(g_rd.trace, 0, 8 * 64 * 8)); }
  const size_t big = (size_t)32768 * 520;
  Bufs b; b.A = dfill(big, 1.f, 1); b.B = dfill((size_t)1024 * 1024, 0.05f, 2); b.H = dfill(big, 1.f, 3); b.bias = dfill(4096, 0.1f, 4); b.X = dfill(big, 1.f, 5);
  CK(hipMalloc((void**)&b.C, big * 4)); CK(hipMalloc((void**)&b.C2, big * 4));
  const size_t slab_words = (size_t)16 * 512 * 512;
  CK(hipMalloc((void**)&b.S, slab_words * 4)); CK(hipMalloc((void**)&b.S2, slab_words * 4)); CK(hipMalloc((void**)&b.cs, 16 * 512 * 4)); CK(hipMalloc((void**)&b.cs2, 16 * 512 * 4));
  spin_up();

  struct Shape { const char* name; int kind, M, N, K, lda, ldb, ldc; bool philox; };
  const Shape shapes[] = {
    {"fwd  G 16384x512x512 philox", GEMM_NT, 16384, 512, 512, 512, 512, 512, true},
    {"fwd  G1 16384x512x448 philox (pitch 452, ldb 449: unaligned rows)", GEMM_NT, 16384, 512, 448, 452, 449, 512, true},
    {"fwd  Glast 16384x187x512 none (ldc 188)", GEMM_NT, 16384, 187, 512, 512, 512, 188, false},
    {"fwd  D 32768x256x256 philox", GEMM_NT, 32768, 256, 256, 256, 256, 256, true},
    {"bwdX G 16384x512x512 philox", GEMM_NN, 16384, 512, 512, 512, 512, 512, true},
    {"bwdX 16384x512x192 philox (lda 196)", GEMM_NN, 16384, 512, 192, 196, 512, 512, true},
    {"bwdX D 32768x256x256 philox", GEMM_NN, 32768, 256, 256, 256, 256, 256, true},
    {"bwdX 16384x512x512 none", GEMM_NN, 16384, 512, 512, 512, 512, 512, false},
    {"fwd  ragged 1000x200x96 philox", GEMM_NT, 1000, 200, 96, 96, 96, 200, true},
    {"bwdX ragged 1000x200x64 philox", GEMM_NN, 1000, 200, 64, 64, 200, 200, true},
  };
  for (const Shape& sh : shapes) {
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    GemmArgs g0 = mk(sh.kind, sh.M, sh.N, sh.K, sh.lda, sh.ldb, sh.ldc, b, sh.philox, b.C), g1 = g0; g1.C = b.C2;
    const size_t words = (size_t)sh.M * sh.ldc;
    CK(hipMemset(b.C, 0xff, words * 4)); CK(hipMemset(b.C2, 0xff, words * 4));
    double t_old, t_new;
#define BOTH(KIND, AM) { launch_old<KIND, AM>(g0, 1); launch_new<KIND, AM>(g1, 1); CK(hipDeviceSynchronize()); \
      t_old = time_us([&] { launch_old<KIND, AM>(g0, 1); }); t_new = time_us([&] { launch_new<KIND, AM>(g1, 1); }); }
    if (sh.kind == GEMM_NT) { if (sh.philox) BOTH(GEMM_NT, GEMM_A_LEAKY_PHILOX) else BOTH(GEMM_NT, GEMM_A_NONE) }
    else { if (sh.philox) BOTH(GEMM_NN, GEMM_A_LEAKY_PHILOX) else BOTH(GEMM_NN, GEMM_A_NONE) }
    const size_t d = diff_words(b.C, b.C2, words);
    printf("%-44s per-tile %7.1f us %6.1f TF | streamed %7.1f us %6.1f TF | differing words %zu of %zu %s\n", sh.name, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, d, words, d ? "MISMATCH" : "bit-identical");
    if (d) {      // where, and what do the two say?
      std::vector<float> x(words), y(words);
      CK(hipMemcpy(x.data(), b.C, words * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), b.C2, words * 4, hipMemcpyDeviceToHost));
      int shown = 0; int mmin = 1 << 30, mmax = -1, nmin = 1 << 30, nmax = -1;
      for (size_t i = 0; i < words; ++i) if (memcmp(&x[i], &y[i], 4) && (x[i] != 0.f || y[i] != 0.f)) {
        const int m = (int)(i / sh.ldc), n = (int)(i % sh.ldc);
        mmin = std::min(mmin, m); mmax = std::max(mmax, m); nmin = std::min(nmin, n); nmax = std::max(nmax, n);
        if (shown++ < 6) printf("      [%d][%d]: per-tile %.9g streamed %.9g\n", m, n, x[i], y[i]);
      }
      printf("      mismatches in rows %d..%d, columns %d..%d\n", mmin, mmax, nmin, nmax);
    }
    dump_stamps(sh.name, run_grid(g1.n_tiles_m * g1.n_tiles_n));
  }
  // weight gradients alone
  struct TShape { const char* name; int M, N, K, lda, ldb; };
  const TShape tshapes[] = { {"bwdW G 512x512x16384", 512, 512, 16384, 512, 512}, {"bwdW D 256x256x32768", 256, 256, 32768, 256, 256},
                             {"bwdW Glast 187x512x16384 (lda 188)", 187, 512, 16384, 188, 512}, {"bwdW G1 512x425x16384 (ldb 428)", 512, 425, 16384, 512, 428},
                             {"bwdW ragged 200x72x1024", 200, 72, 1024, 200, 72} };
  for (const TShape& sh : tshapes) {
    int ns;
    GemmArgs g0 = mk_tn(sh.M, sh.N, sh.K, sh.lda, sh.ldb, strstr(sh.name, "one-stage") ? 4096 : 512, b, b.S, b.cs, &ns), g1 = g0; g1.C = b.S2; g1.colsum_slab = b.cs2;
    const size_t words = (size_t)ns * sh.M * sh.N, cw = (size_t)ns * sh.M;
    CK(hipMemset(b.S, 0xff, words * 4)); CK(hipMemset(b.S2, 0xff, words * 4)); CK(hipMemset(b.cs, 0xff, cw * 4)); CK(hipMemset(b.cs2, 0xff, cw * 4));
    launch_old<GEMM_TN, GEMM_A_RUNTIME>(g0, ns); launch_new<GEMM_TN, GEMM_A_NONE>(g1, ns); CK(hipDeviceSynchronize());
    const double t_old = time_us([&] { launch_old<GEMM_TN, GEMM_A_RUNTIME>(g0, ns); }), t_new = time_us([&] { launch_new<GEMM_TN, GEMM_A_NONE>(g1, ns); });
    const size_t d = diff_words(b.S, b.S2, words) + diff_words(b.cs, b.cs2, cw);
    const double fl = 2.0 * sh.M * sh.N * sh.K;
    printf("%-44s per-tile %7.1f us %6.1f TF | streamed %7.1f us %6.1f TF | slabs %d, differing words %zu %s\n", sh.name, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, ns, d, d ? "MISMATCH" : "bit-identical");
    dump_stamps(sh.name, run_grid(g1.n_tiles_m * g1.n_tiles_n * ns));
  }
  // pair launches: backward-data + weight gradient of one layer
  struct PShape { const char* name; int rows, out, in, lddz; bool philox; };
  const PShape pshapes[] = { {"pair G hidden: 16384 rows, 512 -> 512", 16384, 512, 512, 512, true}, {"pair D hidden: 32768 rows, 256 -> 256", 32768, 256, 256, 256, true},
                             {"pair ragged: 16384 rows, 500 -> 192 (pitch 192)", 16384, 192, 500, 192, true} };
  for (const PShape& sh : pshapes) {
    // dX[rows][in] = dZ[rows][out] . W[out][in] (.) f'(H);  dW[out][in] = dZ^T . X[rows][in]
    int ns;
    GemmArgs nn0 = mk(GEMM_NN, sh.rows, sh.in, sh.out, sh.lddz, sh.in, sh.in, b, sh.philox, b.C), nn1 = nn0; nn1.C = b.C2;
    GemmArgs tn0 = mk_tn(sh.out, sh.in, sh.rows, sh.lddz, sh.in, 512, b, b.S, b.cs, &ns), tn1 = tn0; tn1.C = b.S2; tn1.colsum_slab = b.cs2;
    const size_t words = (size_t)sh.rows * sh.in, sw = (size_t)ns * sh.out * sh.in, cw = (size_t)ns * sh.out;
    CK(hipMemset(b.C, 0xff, words * 4)); CK(hipMemset(b.C2, 0xff, words * 4)); CK(hipMemset(b.S, 0xff, sw * 4)); CK(hipMemset(b.S2, 0xff, sw * 4));
    CK(hipMemset(b.cs, 0xff, cw * 4)); CK(hipMemset(b.cs2, 0xff, cw * 4));
    launch_pair_old<GEMM_A_LEAKY_PHILOX>(nn0, tn0, ns); launch_pair_new<GEMM_A_LEAKY_PHILOX>(nn1, tn1, ns); CK(hipDeviceSynchronize());
    const double t_old = time_us([&] { launch_pair_old<GEMM_A_LEAKY_PHILOX>(nn0, tn0, ns); }), t_new = time_us([&] { launch_pair_new<GEMM_A_LEAKY_PHILOX>(nn1, tn1, ns); });
    const size_t d = diff_words(b.C, b.C2, words) + diff_words(b.S, b.S2, sw) + diff_words(b.cs, b.cs2, cw);
    const double fl = 4.0 * sh.rows * sh.in * sh.out;
    printf("%-44s per-tile %7.1f us %6.1f TF | streamed %7.1f us %6.1f TF | slabs %d, differing words %zu %s\n", sh.name, t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6, ns, d, d ? "MISMATCH" : "bit-identical");
    dump_stamps(sh.name, run_grid(nn1.n_tiles_m * nn1.n_tiles_n + tn1.n_tiles_m * tn1.n_tiles_n * ns));
  }
  // a de-phased sequence, as in the step: the five hot launches back to back, per-tile vs streamed
  {
    int ns1, ns2;
    GemmArgs f = mk(GEMM_NT, 16384, 512, 512, 512, 512, 512, b, true, b.C);
    GemmArgs nn = mk(GEMM_NN, 16384, 512, 512, 512, 512, 512, b, true, b.C2), tn = mk_tn(512, 512, 16384, 512, 512, 512, b, b.S, b.cs, &ns1);
    GemmArgs nd = mk(GEMM_NN, 32768, 256, 256, 256, 256, 256, b, true, b.C2), td = mk_tn(256, 256, 32768, 256, 256, 512, b, b.S2, b.cs2, &ns2);
    const double fl = 2.0 * 2 * 16384.0 * 512 * 512 + 2 * 4.0 * 16384 * 512 * 512 + 4.0 * 32768 * 256 * 256;
    const double t_old = time_us([&] { launch_old<GEMM_NT, GEMM_A_LEAKY_PHILOX>(f, 1); launch_old<GEMM_NT, GEMM_A_LEAKY_PHILOX>(f, 1);
                                       launch_pair_old<GEMM_A_LEAKY_PHILOX>(nd, td, ns2); launch_pair_old<GEMM_A_LEAKY_PHILOX>(nn, tn, ns1); launch_pair_old<GEMM_A_LEAKY_PHILOX>(nn, tn, ns1); });
    const double t_new = time_us([&] { launch_new<GEMM_NT, GEMM_A_LEAKY_PHILOX>(f, 1); launch_new<GEMM_NT, GEMM_A_LEAKY_PHILOX>(f, 1);
                                       launch_pair_new<GEMM_A_LEAKY_PHILOX>(nd, td, ns2); launch_pair_new<GEMM_A_LEAKY_PHILOX>(nn, tn, ns1); launch_pair_new<GEMM_A_LEAKY_PHILOX>(nn, tn, ns1); });
    printf("sequence fwd, fwd, pair D, pair G, pair G:     per-tile %7.1f us %6.1f TF | streamed %7.1f us %6.1f TF\n", t_old, fl / t_old / 1e6, t_new, fl / t_new / 1e6);
  }
  unsigned q[RUN_Q_WORDS]; CK(hipMemcpy(q, g_queues, sizeof(q), hipMemcpyDeviceToHost));
  unsigned nz = 0; for (unsigned v : q) nz += v != 0;
  printf("queue words left non-zero after the last launch: %u (must be 0)\n", nz);
  return 0;
}
