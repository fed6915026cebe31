// libgantts_hip.so -- host orchestration + C ABI (include/gantts_hip.h) of the MI355X GAN step.
//
// The engine keeps the reference's step semantics (train.py:245-320) while restructuring the
// work for the hardware:
//   * D(real) and D(fake) of the D step run as ONE 2N-row pass (same weights), so the frame x
//     weight GEMMs see 32768 rows at the headline config;
//   * G is back-propagated ONCE per step: dloss_d/dy_hat_static (old D weights, the reference's
//     un-detached "leak", train.py:265,274) is stashed and summed with dloss_g/dy_hat_static at
//     y_hat_static before MLPG^T and the G backward (backward is linear in the upstream gradient);
//   * the G-step D pass computes no weight gradients (the reference's are discarded by the next
//     zero_grad, train.py:538-539);
//   * every reduction is two-stage with a fixed order => run-to-run bit-reproducible.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/gantts_hip.h"
#include "frame_kernels.hip.h"
#include "gemm_f32.hip.h"
#include "gemm_bf16s.hip.h"
#include "lstm_kernels.hip.h"
#include "lstm_seq_kernels.hip.h"
#include "sru_kernels.hip.h"

using namespace gt;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(GT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != GT_OK) return _r; \
  } while (0)
#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

extern "C" const char* gt_last_error(void) { return g_err; }
extern "C" const char* gt_version(void) { return "gantts_hip 0.1 (gfx950, f32 MFMA)"; }

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: remember the largest value set per
// (kernel, device), so that a process driving several GPUs raises the limit on each of them.
#include <map>
#include <mutex>
static int ensure_dyn_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  size_t& cur = done[std::make_pair(kernel, dev)];
  if (bytes > cur) {
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur = bytes;
  }
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// optional per-launch timing of the GEMM family (HIP events on the launch stream); used by
// bench.py for the live roofline figure.  Off by default: zero overhead on the normal path.
// ------------------------------------------------------------------------------------------
struct GemmProfiler {
  bool on = false;
  struct Rec { int kind, bn; double flops, bytes; hipEvent_t e0, e1; };
  double last_bytes[GT_PROFILE_SLOTS] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // algorithmic bytes per slot of the last gt_profile_read
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
  }
};
static GemmProfiler g_prof;

extern "C" int gt_profile_enable(int on) {
  g_prof.on = on != 0;
  return GT_OK;
}
// Drains the recorded launches into per-variant totals.  variant = kind*2 + (bn==128): 6 slots, then slot 6 = layer-chain
// launches of forward products, slot 7 = layer-chain launches of backward-data products (gemm_chain.hip.h), slot 8 = pair
// launches (backward-data + weight gradient of one layer, gemm_pair_kernel).
// out_ms[v] = summed kernel time, out_flops[v] = summed algorithmic 2*M*N*K, out_count[v] = launches.
extern "C" int gt_profile_read(double* out_ms, double* out_flops, int64_t* out_count) {
  for (int v = 0; v < GT_PROFILE_SLOTS; ++v) { out_ms[v] = 0; out_flops[v] = 0; out_count[v] = 0; g_prof.last_bytes[v] = 0; }
  for (auto& r : g_prof.recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail(GT_ERR_HIP, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return fail(GT_ERR_HIP, "hipEventElapsedTime failed");
    const int v = r.kind >= 3 ? 3 + r.kind : r.kind * 2 + (r.bn == 128 ? 1 : 0);     // kind 3 / 4: chain of NT / NN products, 5: pair
    out_ms[v] += ms; out_flops[v] += r.flops; out_count[v] += 1; g_prof.last_bytes[v] += r.bytes;
    g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1);
  }
  g_prof.recs.clear();
  return GT_OK;
}
// Algorithmic HBM bytes (every operand once, the result once, fp32) of the launches the last gt_profile_read drained.
extern "C" int gt_profile_bytes(double* out_bytes) {
  for (int v = 0; v < GT_PROFILE_SLOTS; ++v) out_bytes[v] = g_prof.last_bytes[v];
  return GT_OK;
}
static double gemm_algorithmic_bytes(int kind, const GemmArgs& g) {
  double b = 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N);
  if (kind == GEMM_NN && g.act != ACT_NONE && g.H) b += 4.0 * (double)g.M * g.N;     // the producer's stored activation
  return b;
}

// ------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------
static int gemm_cu_count() {
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  hipDeviceProp_t prop;
  const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  cus[dev] = n;
  return n;
}
// per-device ticket counters of the start stagger (mode 3): the two workgroups of a CU draw consecutive tickets
static unsigned int* gemm_stagger_tickets() {
  static std::map<int, unsigned int*> bufs;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  auto it = bufs.find(dev);
  if (it != bufs.end()) return it->second;
  unsigned int* p = nullptr;
  if (hipMalloc((void**)&p, 2048 * sizeof(unsigned int)) != hipSuccess || hipMemset(p, 0, 2048 * sizeof(unsigned int)) != hipSuccess) p = nullptr;
  bufs[dev] = p;
  return p;
}
// products of the GEMM family: f32 MFMA (default) or bf16 MFMA with f32 accumulation (GT_OPT_MATMUL_BF16; set per engine
// entry point for the launches it issues on this thread)
static thread_local int tl_gemm_prec = PREC_F32;

template <int KIND, int BM, int BN, bool VA, bool VB, int PREC, int AM>
static int launch_gemm_impl(GemmArgs g, int nslab, hipStream_t s) {
  const size_t lds = gemm_lds_bytes<KIND, BM, BN, PREC>();
  CHK(ensure_dyn_lds((const void*)gemm_f32_kernel<KIND, BM, BN, VA, VB, PREC, 32, AM>, lds));
  g.n_tiles_m = cdiv(g.M, BM);
  g.n_tiles_n = cdiv(g.N, BN);
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  if (grid <= 0) return GT_OK;
  // Start stagger (gemm_f32.hip.h), OFF by default: in isolation (tools/gemm_stagger_bench.hip, dense random operands,
  // the five big launches of a step back to back) letting one of the two workgroups of a CU start 0.5 .. 2 us late is
  // worth 10-15 % (372 -> 325 us per sequence); inside the training step it measured 0.0 % in every mode,
  // so it stays a measurement switch: GT_GEMM_STAGGER_TICKS (10 ns units), GT_GEMM_STAGGER_MODE.
  static const int stagger_ticks = getenv("GT_GEMM_STAGGER_TICKS") ? atoi(getenv("GT_GEMM_STAGGER_TICKS")) : 0;
  static const int stagger_mode = getenv("GT_GEMM_STAGGER_MODE") ? atoi(getenv("GT_GEMM_STAGGER_MODE")) : 3;
  if (stagger_ticks > 0 && grid > gemm_cu_count()) {
    g.stagger_ticks = stagger_ticks; g.stagger_mode = stagger_mode;
    if (stagger_mode == 3 && !(g.stagger_ticket = gemm_stagger_tickets())) g.stagger_ticks = 0;
  }
  GemmProfiler::Rec rec;
  if (g_prof.on) {
    rec.kind = KIND; rec.bn = BN; rec.flops = 2.0 * g.M * g.N * g.K; rec.bytes = gemm_algorithmic_bytes(KIND, g);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_f32_kernel<KIND, BM, BN, VA, VB, PREC, 32, AM>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.on) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
template <int KIND, int BM, int BN, bool VA, bool VB, int PREC>
static int launch_gemm_t(const GemmArgs& g, int nslab, hipStream_t s) {
  // the hot shape of the float32 step (64 x 64 tiles, 16-byte loadable operands) has its two common epilogue flavours
  // compiled in: no activation, LeakyReLU + Philox dropout (gemm_f32.hip.h: GemmAmode); everything else decides at run time
  if constexpr (KIND != GEMM_TN && BM == 64 && BN == 64 && VA && VB && PREC == PREC_F32) {
    if (g.act == ACT_NONE) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_NONE>(g, nslab, s);
    if (g.act == ACT_LEAKY_DROPOUT && g.drop.mode == DROP_PHILOX) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_LEAKY_PHILOX>(g, nslab, s);
  }
  return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_RUNTIME>(g, nslab, s);
}
static int pick_bn(int N) { return (cdiv(N, 64) * 64 < cdiv(N, 128) * 128) ? 64 : 128; }

template <int KIND, int BM, int BN>
static int launch_gemm_v(const GemmArgs& g_in, int nslab, hipStream_t s) {
  GemmArgs g = g_in;
  g.wide_store = KIND != GEMM_TN && (g.ldc % 4 == 0) && (((uintptr_t)g.C) % 16 == 0) &&
                 (KIND != GEMM_NN || g.act == ACT_NONE || ((g.ldh % 4 == 0) && (((uintptr_t)g.H) % 16 == 0)));
  const bool va = (g.lda % 4 == 0) && (((uintptr_t)g.A) % 16 == 0);
  const bool vb = (g.ldb % 4 == 0) && (((uintptr_t)g.B) % 16 == 0);
  if (tl_gemm_prec == PREC_BF16) {
    if (va && vb) return launch_gemm_t<KIND, BM, BN, true, true, PREC_BF16>(g, nslab, s);
    if (va) return launch_gemm_t<KIND, BM, BN, true, false, PREC_BF16>(g, nslab, s);
    if (vb) return launch_gemm_t<KIND, BM, BN, false, true, PREC_BF16>(g, nslab, s);
    return launch_gemm_t<KIND, BM, BN, false, false, PREC_BF16>(g, nslab, s);
  }
  if (va && vb) return launch_gemm_t<KIND, BM, BN, true, true, PREC_F32>(g, nslab, s);
  if (va) return launch_gemm_t<KIND, BM, BN, true, false, PREC_F32>(g, nslab, s);
  if (vb) return launch_gemm_t<KIND, BM, BN, false, true, PREC_F32>(g, nslab, s);
  return launch_gemm_t<KIND, BM, BN, false, false, PREC_F32>(g, nslab, s);
}

// Tile choice (measured with tools/gemm_tile_sweep.hip on the cfg2 shapes, 16-byte-loadable operands): 64x64 tiles beat
// 128x128 / 64x128 / 128x64 on every forward / backward-data shape of the step (16384x512x512: 77 vs 80 us; 32768x256x256:
// 44.7 vs 47.9; 16384x256x256: 25.1 vs 27.4 / 31.5) -- four tiles per CU drain their epilogues under each other's K loops,
// and the 2x operand re-reads come out of the XCD's L2.  Operands that need the 4-byte loader keep the larger tiles
// (their loader is the bottleneck, and a tile re-read costs 4x the instructions).
static int gemm_tile_mode() {   // measurement switch: GT_GEMM_TILES=big restores the residency model for every launch
  static const int m = [] { const char* v = getenv("GT_GEMM_TILES"); return v && !strcmp(v, "big") ? 1 : 0; }();
  return m;
}
static bool gemm_vec_ok(const float* p, int ld) { return (ld % 4 == 0) && (((uintptr_t)p) % 16 == 0); }
static bool gemm_small_tiles_ok() { return gemm_tile_mode() == 0; }   // f32 and bf16 products alike (bf16: cfg2 1.30 -> 1.21 ms, SRU 32.0 -> 28.2 ms)

struct Scratch {  // growable device buffer
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return GT_OK;
    if (p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(p)); p = nullptr; bytes = 0; }
    const size_t cap = need + need / 8;
    HIPCHK(hipMalloc(&p, cap));
    bytes = cap;
    return GT_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  template <typename T> T* as() const { return (T*)p; }
};

static void gemm_set_wide_store(int kind, GemmArgs& g) {
  // 16-byte accesses need a 16-byte aligned base and a row pitch that is a multiple of 4 floats
  g.wide_store = kind != GEMM_TN && (g.ldc % 4 == 0) && (((uintptr_t)g.C) % 16 == 0) &&
                 (kind != GEMM_NN || g.act == ACT_NONE || ((g.ldh % 4 == 0) && (((uintptr_t)g.H) % 16 == 0)));
}
static int launch_gemm(int kind, const GemmArgs& g, int nslab, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail(GT_ERR_INVALID, "empty GEMM");
  const int bn = pick_bn(g.N);
  const bool vec = gemm_vec_ok(g.A, g.lda) && gemm_vec_ok(g.B, g.ldb);
  if (kind != GEMM_TN && vec && g.M > 64 && gemm_small_tiles_ok())
    return kind == GEMM_NT ? launch_gemm_v<GEMM_NT, 64, 64>(g, 1, s) : launch_gemm_v<GEMM_NN, 64, 64>(g, 1, s);
  // Otherwise tile height by a residency model: 128-row tiles keep 2 workgroups per CU resident (512 at once),
  // 64-row tiles 3-4 (LDS-limited: 768 with 128 columns, 1024 with 64) at ~0.55x the work each.
  // Cost = resident rounds x work per tile; e.g. 384 tiles (187-wide output) or 1024 tiles (2N x 483)
  // finish sooner as 64-row tiles, exactly 512 tiles do not.
  bool small = false;
  if (kind != GEMM_TN && g.M > 64) {
    const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, bn), t64 = (long)cdiv(g.M, 64) * cdiv(g.N, bn);
    const double c128 = (double)cdiv(t128, 512), c64 = 0.55 * (double)cdiv(t64, bn == 64 ? 1024 : 768);
    small = c64 < c128;
  }
  switch (kind) {
    case GEMM_NT:
      if (small) return bn == 64 ? launch_gemm_v<GEMM_NT, 64, 64>(g, 1, s) : launch_gemm_v<GEMM_NT, 64, 128>(g, 1, s);
      return bn == 64 ? launch_gemm_v<GEMM_NT, 128, 64>(g, 1, s) : launch_gemm_v<GEMM_NT, 128, 128>(g, 1, s);
    case GEMM_NN:
      if (small) return bn == 64 ? launch_gemm_v<GEMM_NN, 64, 64>(g, 1, s) : launch_gemm_v<GEMM_NN, 64, 128>(g, 1, s);
      return bn == 64 ? launch_gemm_v<GEMM_NN, 128, 64>(g, 1, s) : launch_gemm_v<GEMM_NN, 128, 128>(g, 1, s);
    default:
      if (g.n_tiles_m == 64) return launch_gemm_v<GEMM_TN, 64, 64>(g, nslab, s);     // linear_backward_weight's choice (tile height in n_tiles_m)
      return bn == 64 ? launch_gemm_v<GEMM_TN, 128, 64>(g, nslab, s) : launch_gemm_v<GEMM_TN, 128, 128>(g, nslab, s);
  }
}

static DropoutSpec no_drop() {
  DropoutSpec d;
  memset(&d, 0, sizeof(d));
  d.mode = DROP_NONE;
  d.scale = 1.f;
  return d;
}

// Y = act(X W^T + b)
static int linear_forward(const float* X, int ldx, const float* W, int ldw, const float* b, float* Y, int ldy,
                          long rows, int in, int out, int act, const DropoutSpec& drop, hipStream_t s) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
  g.M = (int)rows; g.N = out; g.K = in; g.bias = b; g.act = act; g.drop = drop;
  return launch_gemm(GEMM_NT, g, 1, s);
}
// dX = (dZ W[:, col0:col0+ncols]) (.) f'(H)
static GemmArgs backward_data_args(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                                   long rows, int out, int ncols, int act_prev, const float* H, int ldh, const DropoutSpec& drop) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dZ; g.lda = lddz; g.B = W + col0; g.ldb = ldw; g.C = dX; g.ldc = lddx;
  g.M = (int)rows; g.N = ncols; g.K = out; g.act = act_prev; g.H = H; g.ldh = ldh; g.drop = drop;
  return g;
}
static int linear_backward_data(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                                long rows, int out, int ncols, int act_prev, const float* H, int ldh,
                                const DropoutSpec& drop, hipStream_t s) {
  return launch_gemm(GEMM_NN, backward_data_args(dZ, lddz, W, ldw, col0, dX, lddx, rows, out, ncols, act_prev, H, ldh, drop), 1, s);
}
// Pair launch (gemm_pair_kernel): the same layer's backward-data product rides in the weight gradient's launch when both
// run on 64x64 tiles with 16-byte loadable operands.  GT_GEMM_PAIR=0 keeps them apart (measurement switch).
static bool gemm_pair_enabled() {
  static const bool on = [] { const char* v = getenv("GT_GEMM_PAIR"); return !(v && v[0] == '0'); }();
  return on;
}
static bool gemm_pair_ok(const GemmArgs& nn) {
  return gemm_pair_enabled() && gemm_small_tiles_ok() && nn.M > 64 && gemm_vec_ok(nn.A, nn.lda) && gemm_vec_ok(nn.B, nn.ldb);
}
static int launch_gemm_pair(const GemmArgs& nn_in, const GemmArgs& tn_in, int nslab, hipStream_t s) {
  GemmArgs nn = nn_in, tn = tn_in;
  gemm_set_wide_store(GEMM_NN, nn);
  tn.wide_store = 0;
  nn.n_tiles_m = cdiv(nn.M, 64); nn.n_tiles_n = cdiv(nn.N, 64);
  tn.n_tiles_m = cdiv(tn.M, 64); tn.n_tiles_n = cdiv(tn.N, 64);
  const int n1 = nn.n_tiles_m * nn.n_tiles_n, n2 = tn.n_tiles_m * tn.n_tiles_n * nslab;
  const bool bf16 = tl_gemm_prec == PREC_BF16;
  const size_t lds = bf16 ? std::max(gemm_lds_bytes<GEMM_NN, 64, 64, PREC_BF16>(), gemm_lds_bytes<GEMM_TN, 64, 64, PREC_BF16>())
                          : std::max(gemm_lds_bytes<GEMM_NN, 64, 64>(), gemm_lds_bytes<GEMM_TN, 64, 64>());
  const int am = bf16 ? GEMM_A_RUNTIME : (nn.act == ACT_NONE ? GEMM_A_NONE : ((nn.act == ACT_LEAKY_DROPOUT && nn.drop.mode == DROP_PHILOX) ? GEMM_A_LEAKY_PHILOX : GEMM_A_RUNTIME));
  const void* kern = bf16 ? (const void*)gemm_pair_kernel<PREC_BF16> : (am == GEMM_A_NONE ? (const void*)gemm_pair_kernel<PREC_F32, GEMM_A_NONE> :
                     (am == GEMM_A_LEAKY_PHILOX ? (const void*)gemm_pair_kernel<PREC_F32, GEMM_A_LEAKY_PHILOX> : (const void*)gemm_pair_kernel<PREC_F32>));
  CHK(ensure_dyn_lds(kern, lds));
  GemmProfiler::Rec rec;
  if (g_prof.on) {
    rec.kind = 5; rec.bn = 64; rec.flops = 2.0 * nn.M * nn.N * nn.K + 2.0 * tn.M * tn.N * tn.K;
    rec.bytes = gemm_algorithmic_bytes(GEMM_NN, nn) + gemm_algorithmic_bytes(GEMM_TN, tn);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  // GT_PAIR_ORDER (default 1): weight-gradient workgroups first (longest work first); 0 = backward-data tiles first
  static const int tn_first = getenv("GT_PAIR_ORDER") ? atoi(getenv("GT_PAIR_ORDER")) : 1;   // measured: 108.2 -> 104.4 us per pair launch, cfg2 step 1.523 -> 1.499 ms
  if (bf16) hipLaunchKernelGGL(gemm_pair_kernel<PREC_BF16>, dim3(n1 + n2), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else if (am == GEMM_A_NONE) hipLaunchKernelGGL((gemm_pair_kernel<PREC_F32, GEMM_A_NONE>), dim3(n1 + n2), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else if (am == GEMM_A_LEAKY_PHILOX) hipLaunchKernelGGL((gemm_pair_kernel<PREC_F32, GEMM_A_LEAKY_PHILOX>), dim3(n1 + n2), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else      hipLaunchKernelGGL(gemm_pair_kernel<PREC_F32>, dim3(n1 + n2), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  LAUNCH_CHECK();
  if (g_prof.on) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}


// dW (+)= dZ^T X ; db (+)= colsum(dZ)   -- split over the frame dimension, fixed-order combine.
// The bias gradient rides along in the weight-gradient kernel (column sums of its A operand).
// Deferred combines (SlabDefer): the partial slabs of every layer go to their own piece of a pool and the combine is
// only RECORDED; slab_defer_flush() runs all recorded combines in one launch.  Used by the fused single-GPU step, where
// nothing reads a weight gradient between a network's backward pass and its optimizer step.
struct SlabDefer {
  Scratch pool;
  size_t used = 0;
  SlabJobs jobs;
  int blocks = 0;
  bool active = false;
  SlabDefer() { jobs.n = 0; jobs.pad_ = 0; }
};
static int slab_defer_flush(SlabDefer& d, hipStream_t s) {
  if (d.jobs.n > 0) {
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(d.blocks), dim3(256), 0, s, d.jobs);
    LAUNCH_CHECK();
  }
  d.jobs.n = 0; d.blocks = 0; d.used = 0;
  return GT_OK;
}
// `ride_along` (optional): the backward-data product of the same layer; if it can share the weight gradient's launch it
// does and *rode is set, otherwise the caller launches it itself.
static int linear_backward_weight(const float* dZ, int lddz, const float* X, int ldx, long rows, int out, int in,
                                  float* dW, float* db, bool accumulate, Scratch& slabs, Scratch& colp, hipStream_t s,
                                  SlabDefer* defer = nullptr, const GemmArgs* ride_along = nullptr, bool* rode = nullptr) {
  if (rode) *rode = false;
  if (dW) {
    // 64x64 tiles when both operands take 16-byte loads: the same workgroup count with 4x fewer partial slabs (less slab
    // traffic in the product's epilogue and in the combine: 512x512 over 16384 frames 8 slabs instead of 32)
    const bool t64 = gemm_vec_ok(dZ, lddz) && gemm_vec_ok(X, ldx) && gemm_small_tiles_ok();
    const int bn = t64 ? 64 : pick_bn(in);
    const int tiles = cdiv(out, t64 ? 64 : 128) * cdiv(in, bn);
    static const int slab_wgs = getenv("GT_TN_WGS") ? atoi(getenv("GT_TN_WGS")) : 512;   // measurement switch
    int nslab = std::max(1, slab_wgs / tiles);   // <= 2 workgroups per CU x 256 CUs: one resident round
    const int max_slab = (int)((rows + 255) / 256);
    if (nslab > max_slab) nslab = max_slab;
    if (nslab < 1) nslab = 1;
    int k_chunk = cdiv(cdiv(rows, nslab), GEMM_BK) * GEMM_BK;
    nslab = cdiv(rows, k_chunk);
    const long slab_stride = (long)out * in;
    const size_t need = (((size_t)nslab * slab_stride + (size_t)nslab * out) * sizeof(float) + 255) & ~(size_t)255;
    const bool can4 = slab_stride % 4 == 0 && ((uintptr_t)dW) % 16 == 0;
    float* slab_base = nullptr;
    if (defer && defer->active && accumulate) { CHK(slab_defer_flush(*defer, s)); defer = nullptr; }   // never two combines of one dW in a launch
    if (defer && defer->active && can4) {
      if (defer->jobs.n == SLAB_MAX_JOBS || defer->used + need > defer->pool.bytes) {
        CHK(slab_defer_flush(*defer, s));                       // recorded combines first, then (maybe) a larger pool
        if (need > defer->pool.bytes) CHK(defer->pool.ensure(std::max(need * 4, (size_t)64 << 20)));
      }
      slab_base = (float*)((char*)defer->pool.p + defer->used);
      defer->used += need;
    } else {
      defer = nullptr;
      CHK(slabs.ensure(need));
      slab_base = slabs.as<float>();
    }
    float* bias_slabs = slab_base + (size_t)nslab * slab_stride;
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dZ; g.lda = lddz; g.B = X; g.ldb = ldx; g.C = slab_base; g.ldc = in;
    g.M = out; g.N = in; g.K = (int)rows; g.k_chunk = k_chunk; g.slab_stride = slab_stride;
    g.colsum_slab = db ? bias_slabs : nullptr;
    g.drop = no_drop();
    g.n_tiles_m = t64 ? 64 : 128;        // tile height request (launch_gemm_t overwrites the field with the tile count)
    if (t64 && ride_along && rode && gemm_pair_ok(*ride_along)) {
      CHK(launch_gemm_pair(*ride_along, g, nslab, s));
      *rode = true;
    } else {
      CHK(launch_gemm(GEMM_TN, g, nslab, s));
    }
    if (can4) {
      const int main_blocks = cdiv(slab_stride / 4, 256);
      const int bias_blocks = db ? cdiv(out, 256) : 0;
      if (defer) {
        SlabJob& J = defer->jobs.j[defer->jobs.n++];
        J.slabs = slab_base; J.slab_stride = slab_stride; J.n4 = slab_stride / 4; J.out = dW; J.bslabs = bias_slabs; J.bout = db;
        J.nslab = nslab; J.accumulate = accumulate ? 1 : 0; J.nb = out; J.main_blocks = main_blocks; J.block0 = defer->blocks; J.pad_ = 0;
        defer->blocks += main_blocks + bias_blocks;
        return GT_OK;
      }
      hipLaunchKernelGGL(slab_reduce4_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, s, slab_base, slab_stride, nslab,
                         slab_stride / 4, dW, accumulate ? 1 : 0, (const float*)bias_slabs, out, db, main_blocks);
      LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(slab_stride, 256)), dim3(256), 0, s, slabs.as<float>(),
                         slab_stride, nslab, slab_stride, dW, accumulate ? 1 : 0);
      LAUNCH_CHECK();
      if (db) {
        hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(out, 64)), dim3(1024), 0, s, bias_slabs, (long)out, nslab, out, db,
                           accumulate ? 1 : 0);
        LAUNCH_CHECK();
      }
    }
  } else if (db) {
    const int rows_per_blk = 128;
    const int nblk = cdiv(rows, rows_per_blk);
    CHK(colp.ensure((size_t)nblk * out * sizeof(float)));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, cdiv(out, 64)), dim3(256), 0, s, dZ, lddz, rows, out, rows_per_blk,
                       colp.as<float>());
    LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(out, 256)), dim3(256), 0, s, colp.as<float>(), nblk, out, db,
                       accumulate ? 1 : 0);
    LAUNCH_CHECK();
  }
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// bf16-storage products (gemm_bf16s.hip.h; GT_OPT_MATMUL_BF16)
// ------------------------------------------------------------------------------------------
static inline int pad8(long n) { return (int)((n + 7) & ~7L); }
template <int BM, int BN, int EPI, int AMODE>
static int launch_gemm_b16_t(GemmB16Args g, int nslab, hipStream_t s) {
  const size_t lds = gemm_b16_lds_bytes<BM, BN>();
  CHK(ensure_dyn_lds((const void*)gemm_b16_kernel<BM, BN, EPI, AMODE>, lds));
  g.n_tiles_m = cdiv(g.M, BM);
  g.n_tiles_n = cdiv(g.N, BN);
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  if (grid <= 0) return GT_OK;
  GemmProfiler::Rec rec;
  if (g_prof.on) {
    rec.kind = g.epi; rec.bn = BN; rec.flops = 2.0 * g.M * g.N * g.K;
    rec.bytes = 2.0 * ((double)g.M * g.K + (double)g.K * g.N) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N +
                (g.CbT ? 2.0 : 0.0) * g.M * g.N + ((g.epi == B16_BWD_DATA && g.act != ACT_NONE) ? 2.0 * g.M * g.N : 0.0);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_b16_kernel<BM, BN, EPI, AMODE>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.on) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
// the LDS-DMA forms (gemm_bf16s.hip.h: gemm_b16_tile_dma, ring of 2 stages), 8-wave workgroups (2 x 4): T = 128: 128 x 128
// tile, two workgroups per CU; T = 256: 256 x 256 tile, one workgroup per CU -- half the operand bytes per flop.  More waves pull
// more operand bytes per CU (tools/gemm_b16_sweep: 32768 x 3072 x 1024 forward, 128 x 128 with 4 waves 309 us, with 8 waves
// 296 us, 256 x 256 with 4 waves 301 us, with 8 waves 248 us; backward-data 259 -> 208 us; 32768 x 2048 x 512: 136 / 119 / 105 us)
template <int EPI, int AMODE, int T>
static int launch_gemm_b16_dma(GemmB16Args g, int nslab, hipStream_t s) {
  constexpr int WGN = 4;
  const size_t lds = gemm_b16_dma_lds_bytes<T, T, 2>();
  CHK(ensure_dyn_lds((const void*)gemm_b16_dma_kernel<T, T, EPI, AMODE, 2, 2, WGN>, lds));
  g.n_tiles_m = cdiv(g.M, T);
  g.n_tiles_n = cdiv(g.N, T);
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  if (grid <= 0) return GT_OK;
  GemmProfiler::Rec rec;
  if (g_prof.on) {
    rec.kind = g.epi; rec.bn = T; rec.flops = 2.0 * g.M * g.N * g.K;
    rec.bytes = 2.0 * ((double)g.M * g.K + (double)g.K * g.N) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N +
                (g.CbT ? 2.0 : 0.0) * g.M * g.N + ((g.epi == B16_BWD_DATA && g.act != ACT_NONE) ? 2.0 * g.M * g.N : 0.0);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_b16_dma_kernel<T, T, EPI, AMODE, 2, 2, WGN>), dim3(grid), dim3(64 * 2 * WGN), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.on) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
// tile: 0 = chosen here from the shape; 64 / 128 = the caller's choice (the weight gradient sizes its slabs for a tile)
static int launch_gemm_b16(const GemmB16Args& g, int nslab, hipStream_t s, int tile = 0) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail(GT_ERR_INVALID, "empty GEMM");
  if ((g.lda & 7) || (g.ldb & 7) || (((uintptr_t)g.A) & 15) || (((uintptr_t)g.B) & 15))
    return fail(GT_ERR_INVALID, "bf16 product: operands must be 16-byte aligned with a row pitch that is a multiple of 8");
  if (g.CbT && ((g.ldcbt & 3) || (((uintptr_t)g.CbT) & 7))) return fail(GT_ERR_INVALID, "bf16 product: transposed result must be 8-byte aligned");
  // 128 x 128 tiles once they still give every CU two workgroups (one resident round), else 64 x 64 (four per CU)
  const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * nslab;
  static const int force_tiles = getenv("GT_B16_TILES") ? atoi(getenv("GT_B16_TILES")) : 0;      // measurement switch: 64 / 128
  const bool big = tile ? tile >= 128
                        : (force_tiles == 64 ? false : (g.epi != B16_SLAB && g.M >= 128 && g.N >= 128 && (force_tiles == 128 || t128 >= 2L * gemm_cu_count())));
  // operand stages by LDS-DMA when no element of a stage needs masking and no row sums ride in the loader
  static const bool dma_on = !(getenv("GT_B16_DMA") && getenv("GT_B16_DMA")[0] == '0');
  const bool dma = dma_on && big && g.K % 64 == 0 && (g.epi != B16_SLAB || (g.k_chunk % 64 == 0 && !g.rowsum_slab));
  // 256 x 256 tiles when they fill whole rounds of CUs (one workgroup per CU) to within 15 %
  const long t256 = (long)cdiv(g.M, 256) * cdiv(g.N, 256) * nslab, cus = gemm_cu_count();
  const bool huge = dma && (tile ? tile == 256 : (force_tiles == 256 || (force_tiles == 0 && g.epi != B16_SLAB && g.M >= 256 && g.N >= 256 && t256 >= cus &&
                                                   (double)(cdiv(t256, cus) * cus - t256) <= 0.15 * (double)(cdiv(t256, cus) * cus))));
  // the epilogue flavour is a template parameter of the kernel (gemm_bf16s.hip.h: GemmB16Amode)
  int amode = B16_A_NONE;
  if (g.epi != B16_SLAB) {
    if (g.act == ACT_SIGMOID) amode = B16_A_SIGMOID;
    else if (g.act == ACT_LEAKY_DROPOUT) amode = g.drop.mode == DROP_PHILOX ? B16_A_LEAKY_PHILOX : (g.drop.mode == DROP_BUFFER ? B16_A_LEAKY_BUFFER : B16_A_LEAKY);
  }
#define GT_B16_CASE(E, A) if (g.epi == E && amode == A) \
    return huge ? launch_gemm_b16_dma<E, A, 256>(g, nslab, s) \
                : (dma ? launch_gemm_b16_dma<E, A, 128>(g, nslab, s) : (big ? launch_gemm_b16_t<128, 128, E, A>(g, nslab, s) : launch_gemm_b16_t<64, 64, E, A>(g, nslab, s)));
  GT_B16_CASE(B16_FWD, B16_A_NONE) GT_B16_CASE(B16_FWD, B16_A_LEAKY_PHILOX) GT_B16_CASE(B16_FWD, B16_A_LEAKY_BUFFER)
  GT_B16_CASE(B16_FWD, B16_A_LEAKY) GT_B16_CASE(B16_FWD, B16_A_SIGMOID)
  GT_B16_CASE(B16_BWD_DATA, B16_A_NONE) GT_B16_CASE(B16_BWD_DATA, B16_A_LEAKY_PHILOX) GT_B16_CASE(B16_BWD_DATA, B16_A_LEAKY_BUFFER)
  GT_B16_CASE(B16_BWD_DATA, B16_A_LEAKY) GT_B16_CASE(B16_BWD_DATA, B16_A_SIGMOID)
#undef GT_B16_CASE
  if (g.epi == B16_SLAB)
    return huge ? launch_gemm_b16_dma<B16_SLAB, B16_A_NONE, 256>(g, nslab, s)
                : (dma ? launch_gemm_b16_dma<B16_SLAB, B16_A_NONE, 128>(g, nslab, s)
                       : (big ? launch_gemm_b16_t<128, 128, B16_SLAB, B16_A_NONE>(g, nslab, s) : launch_gemm_b16_t<64, 64, B16_SLAB, B16_A_NONE>(g, nslab, s)));
  return fail(GT_ERR_INVALID, "bf16 product: unknown epilogue");
}
static GemmB16Args b16_args() {
  GemmB16Args g;
  memset(&g, 0, sizeof(g));
  g.drop.mode = DROP_NONE; g.drop.scale = 1.f;
  return g;
}
// [rows][ld_in] float32 / bf16  ->  bf16 [rows][ldo] and / or its transpose [cols][ldt] (+ per-column sums -> colsum, the
// bias gradient of a dZ that no product wrote)
template <typename TIN>
static int cast_transpose(const TIN* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
                          float* colsum, bool colsum_accumulate, Scratch* colp, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return GT_OK;
  const int gx = cdiv(rows, 64);
  float* part = nullptr;
  if (colsum) { CHK(colp->ensure((size_t)gx * cols * sizeof(float))); part = colp->as<float>(); }
  hipLaunchKernelGGL((cast_transpose_kernel<TIN>), dim3(gx, cdiv(cols, 64)), dim3(256), 0, s, in, ld_in, rows, cols, out, ldo, outT, ldt, part);
  LAUNCH_CHECK();
  if (colsum) {
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, (const float*)part, gx, cols, colsum, colsum_accumulate ? 1 : 0);
    LAUNCH_CHECK();
  }
  return GT_OK;
}
// dW (+)= dZT . XT^T over the frame dimension (K = rows), db (+)= row sums of dZT; split into float32 slabs, fixed-order combine
// defer (optional, fused single-GPU step): the slabs go to the network's pool and the combine is only recorded; all
// recorded combines of a network run as ONE launch in front of its optimizer step (slab_defer_flush).
static int weight_grad_b16(const __bf16* dZT, long lddzt, const __bf16* XT, long ldxt, long rows, int out, int in, float* dW, float* db,
                           bool accumulate, Scratch& slabs, hipStream_t s, SlabDefer* defer = nullptr) {
  // 128 x 128 tiles (two workgroups per CU) once the matrix has at least 16 of them, with the slab count that fills whole
  // rounds of 2 x CUs workgroups best (r = 1 .. 3 rounds; tools/gemm_b16_sweep: 1024 x 3072 over 32 768 frames 450 us with
  // 64 x 64 tiles -> 246 us with 8 slabs of 128 x 128; 512 x 2048: 124 -> 76 us); 64 x 64 (four per CU) below that
  static const int force_wg_tile = getenv("GT_B16_WG_TILE") ? atoi(getenv("GT_B16_WG_TILE")) : 0;     // measurement switch: 64 / 128
  const int t128 = cdiv(out, 128) * cdiv(in, 128), t256 = cdiv(out, 256) * cdiv(in, 256);
  const bool big = force_wg_tile ? force_wg_tile >= 128 : (out >= 128 && in >= 128 && t128 >= 16);
  // 256 x 256 tiles (one 8-wave workgroup per CU, LDS-DMA only: no bias gradient, whole 64-frame stages) for the largest matrices
  const bool huge = big && !db && rows % 64 == 0 && (force_wg_tile ? force_wg_tile == 256 : (out >= 512 && in >= 512 && t256 >= 32));
  int nslab;
  if (big) {
    const int slots = huge ? gemm_cu_count() : 2 * gemm_cu_count(), tl = huge ? t256 : t128;
    double best = 2.0;
    nslab = 1;
    for (int r = 1; r <= 3; ++r) {
      const int ns = std::max(1, slots * r / tl);
      const double waste = 1.0 - (double)tl * ns / ((double)slots * cdiv((long)tl * ns, slots)) + 0.05 * (r - 1);   // every round has its own epilogues
      if (waste < best - 1e-9) { best = waste; nslab = ns; }
    }
  } else {
    nslab = std::max(1, 1024 / (cdiv(out, 64) * cdiv(in, 64)));       // four 64 x 64 workgroups per CU
  }
  nslab = std::min<long>(nslab, std::max<long>(1, rows / 512));
  const int k_chunk = cdiv(cdiv(rows, nslab), B16_BK) * B16_BK;
  nslab = cdiv(rows, k_chunk);
  const long slab_stride = (long)out * in;
  const size_t need = (((size_t)nslab * slab_stride + (size_t)nslab * out) * sizeof(float) + 255) & ~(size_t)255;
  const bool can4 = slab_stride % 4 == 0 && ((uintptr_t)dW) % 16 == 0;
  float* slab_base = nullptr;
  if (defer && defer->active && accumulate) { CHK(slab_defer_flush(*defer, s)); defer = nullptr; }
  if (defer && defer->active && can4) {
    if (defer->jobs.n == SLAB_MAX_JOBS || defer->used + need > defer->pool.bytes) {
      CHK(slab_defer_flush(*defer, s));
      if (need > defer->pool.bytes) CHK(defer->pool.ensure(std::max(need * 4, (size_t)64 << 20)));
    }
    slab_base = (float*)((char*)defer->pool.p + defer->used);
    defer->used += need;
  } else {
    defer = nullptr;
    CHK(slabs.ensure(need));
    slab_base = slabs.as<float>();
  }
  float* bias_slabs = slab_base + (size_t)nslab * slab_stride;
  GemmB16Args g = b16_args();
  g.A = dZT; g.lda = (int)lddzt; g.B = XT; g.ldb = (int)ldxt; g.M = out; g.N = in; g.K = (int)rows;
  g.C = slab_base; g.ldc = in; g.epi = B16_SLAB; g.k_chunk = k_chunk; g.slab_stride = slab_stride;
  g.rowsum_slab = db ? bias_slabs : nullptr;
  CHK(launch_gemm_b16(g, nslab, s, huge ? 256 : (big ? 128 : 64)));
  if (can4) {
    const int main_blocks = cdiv(slab_stride / 4, 256), bias_blocks = db ? cdiv(out, 256) : 0;
    if (defer) {
      SlabJob& J = defer->jobs.j[defer->jobs.n++];
      J.slabs = slab_base; J.slab_stride = slab_stride; J.n4 = slab_stride / 4; J.out = dW; J.bslabs = bias_slabs; J.bout = db;
      J.nslab = nslab; J.accumulate = accumulate ? 1 : 0; J.nb = out; J.main_blocks = main_blocks; J.block0 = defer->blocks; J.pad_ = 0;
      defer->blocks += main_blocks + bias_blocks;
      return GT_OK;
    }
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, s, slab_base, slab_stride, nslab, slab_stride / 4,
                       dW, accumulate ? 1 : 0, (const float*)bias_slabs, out, db, main_blocks);
    LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(slab_stride, 256)), dim3(256), 0, s, slab_base, slab_stride, nslab, slab_stride, dW,
                       accumulate ? 1 : 0);
    LAUNCH_CHECK();
    if (db) {
      hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(out, 64)), dim3(1024), 0, s, bias_slabs, (long)out, nslab, out, db, accumulate ? 1 : 0);
      LAUNCH_CHECK();
    }
  }
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// engine state
// ------------------------------------------------------------------------------------------
struct Lin { float *W, *b, *dW, *db; int in, out; };
// a tensor kept as bf16 in both orientations: rm [rows][ld], tr [cols][ldt]   (gemm_bf16s.hip.h)
struct B16Img {
  Scratch rm, tr;
  int ld = 0; long ldt = 0;
  __bf16* r() { return rm.as<__bf16>(); }
  __bf16* t() { return tr.as<__bf16>(); }
  int ensure(long rows, int cols, bool want_t) {
    ld = pad8(cols); ldt = pad8(rows);
    CHK(rm.ensure((size_t)rows * ld * 2 + 64));
    if (want_t) CHK(tr.ensure((size_t)cols * ldt * 2 + 64));
    return GT_OK;
  }
  void release() { rm.release(); tr.release(); }
};
// bf16 shadows of one nn.Linear weight (out, in): w [out][ldw] feeds the forward product, wt [in][ldwt] backward-data
struct LinShadow { Scratch w, wt; int ldw = 0, ldwt = 0; };
static inline bool is_i2o(int arch) { return arch == GT_ARCH_IN2OUT || arch == GT_ARCH_IN2OUT_RNN; }
static inline bool has_lstm_body(int arch) { return arch == GT_ARCH_LSTM || arch == GT_ARCH_IN2OUT_RNN; }
struct LstmDirP { float *Wih, *Whh, *bih, *bhh, *dWih, *dWhh, *dbih, *dbhh; };
struct LstmLayerP { int in; LstmDirP d[2]; };
struct SruLayerP { int in, k; float *W, *b, *dW, *db; };

struct Net {
  bool bound = false;
  gt_model_desc d;
  std::vector<Lin> hidden;
  Lin last, gate;
  std::vector<LstmLayerP> lstm;   // GT_ARCH_LSTM; `last` is hidden2out
  std::vector<SruLayerP> sru;     // GT_ARCH_SRU;  `last` is hidden2out
  bool training = true;
  bool grads_dirty = false;      // false after zero_grad: next backward overwrites instead of accumulating
  bool has_opt = false;
  gt_optim_desc od;
  long step = 0;
  // injected dropout masks [pass][layer]
  const float* inj[3][16];
  Net() { memset(inj, 0, sizeof(inj)); }
};

// Banded images of the MLPG matrices the caller has passed so far, one per (R pointer, T): batches of a corpus come in
// many padded lengths and the caller (gantts_amd.paramgen / the reference's per-batch R) keeps one R per T, so after the
// first sight of a T there is no extraction kernel, no D2H copy and no host synchronisation on the step path.
// Contract (include/gantts_hip.h): R is immutable while cached; gt_invalidate_mlpg_cache() after rewriting / freeing it.
struct MlpgBand {
  const float* R = nullptr;
  int T = 0, kb = 0;
  uint64_t last_use = 0;
  Scratch band;
};
struct MlpgCache {
  static constexpr size_t MAX_ENTRIES = 256;
  std::vector<MlpgBand*> entries;
  MlpgBand* cur = nullptr;
  uint64_t tick = 0;
  Scratch tmp;                      // per-offset maxima of a new R (persistent: no hipFree on the step path)
  void clear() { for (auto* b : entries) { b->band.release(); delete b; } entries.clear(); cur = nullptr; }
};

struct gt_engine {
  gt_stream_config cfg;
  Net net[2];
  uint64_t seed = 0x5DEECE66DULL;
  uint64_t step_counter = 0;
  float tv_override = -1.f;
  const double* tv_dev = nullptr;   // device-resident global normaliser (data parallel: no host round trip)
  // derived stream maps (device)
  int Dout_cfg = 0, Ds = 0, Da = 0;
  std::vector<int> h_scol, h_sstride, h_adv_cols, h_adv_inv;
  int *d_scol = nullptr, *d_sstride = nullptr, *d_adv_cols = nullptr, *d_adv_inv = nullptr;
  // In2Out uses a single dynamic stream of width out_dim
  int *d_scol_i2o = nullptr, *d_sstride_i2o = nullptr; int i2o_ds = 0;
  MlpgCache mlpg;
  // workspace
  std::vector<Scratch> g_act, d_act;       // hidden activations
  Scratch dcat, dzA, dzB, leak, gadv, gs, gy, slabs, colp, partial, headp, headw, dmask, tx, gx, dgx, dtz, dout;
  Scratch scal;                            // StepScalars + StepResults
  StepResults* h_res = nullptr;            // pinned
  StepResults* h_res_dev = nullptr;        // the same page as the kernels see it: the fused calls' finalisation writes the
                                           // scalars straight into host memory (no device -> host copy launch behind it)
  // per-step state
  int B = 0, T = 0; long N = 0;
  const float* last_x = nullptr; const float* last_yhat = nullptr; const float* last_yhs = nullptr;
  bool g_pass_valid = false, leak_pending = false, fake_cat_valid = false;
  const float* fake_cat_x = nullptr; const float* fake_cat_yhs = nullptr;
  bool d_begin_done = false, g_begin_done = false, g_has_adv = false, g_used_mlpg = false;
  const float* tv_mask = nullptr; long tv_n = 0; float tv_ovr = 0.f;   // sum(mask) already on the device for this step
  // early results (single-GPU fused entry points): the step scalars are final right after the loss
  // kernels, long before backward + optimizer finish; they are copied out then, and the call returns
  // as soon as THAT copy has landed, leaving the rest of the step queued on the stream.
  bool early = false, early_done = false;
  hipEvent_t ev_res = nullptr;
  // deferred results of the split-phase calls (out == NULL): own pinned copy + event per role, fetched by gt_*_result
  StepResults* h_def[2] = {nullptr, nullptr}; hipEvent_t ev_def[2] = {nullptr, nullptr}; bool def_pending[2] = {false, false};
  std::vector<DropoutSpec> g_specs, d_specs;   // dropout sites of the stashed passes
  // recurrent generator workspace (per layer) and the lengths of the current batch
  std::vector<Scratch> l_xproj, l_gates, l_cst, l_out, l_outd;   // l_outd: inter-layer-dropped outputs
  Scratch i2o_gout;                                              // In2OutRNNHighwayNet: hidden2out output G(x)
  Scratch l_state, l_dout, l_hshift;
  Scratch l_xch;                                   // persistent recurrence: exchange granules
  struct GtComm* comm = nullptr;                   // gt_comm_init: RCCL communicator + comm stream (data parallel)
  int dp_rank = 0, dp_world = 1;                   // this engine's shard of the minibatch (gt_comm_init / gt_set_shard): sequence b here
                                                   // is sequence dp_rank + dp_world * b of the whole minibatch (round-robin dealing)
  int chk_B = 0, chk_T = 0;                        // (B, T) of the entry point that is running (check_common)
  std::vector<std::pair<long, long>> comm_done[2]; // per role: gradient ranges (offset, count) already handed to RCCL this step
  std::vector<std::pair<long, long>> comm_pending[2];   // final on the step stream, not handed over yet (merged into few messages)
  Scratch comm_tv;                                 // device double: global valid-frame count
  bool tv_inflight = false;                        // its all-reduce has been issued for the current mask
  // GT_OPT_MATMUL_BF16 on MLP stacks: bf16 images (both orientations) of everything that only feeds products
  std::vector<B16Img> g_actb, d_actb;              // hidden activations
  B16Img xin_b, dcat_b, gy_b, dz_b[2], fwd_b;      // G's input, D's [x | adv] image (2N rows), dloss/dy_hat, dZ ping-pong, gt_model_forward's input
  std::vector<B16Img> l_in_b;                      // recurrent generator: image of every layer's input (+ the top output, last entry)
  std::vector<B16Img> s_in_b;                      // SRU generator: image of every layer's (dropped) input (+ the top output, last entry)
  B16Img s_du_b;                                   // SRU: dU of the current layer, both orientations
  std::vector<LinShadow> ssh;                      // SRU: per layer W (n_in, ncols*k) as w [n_in][..] and wt [ncols*k][n_in]; last entry: hidden2out
  std::vector<B16Img> l_dg_b;                      // per layer: dG image (both orientations; per layer because the side stream reads it late)
  B16Img l_hs_b;                                   // h that entered each frame (transposed)
  hipStream_t side = nullptr;                      // recurrent generator: weight-gradient products run beside the next layer's recurrence
  hipEvent_t ev_side_go = nullptr, ev_side_done = nullptr;
  Scratch slabs_side, colp_side;
  std::vector<LinShadow> lsh;                      // per LSTM layer: W_ih of all directions stacked [dirs*4H][in]; last entry: hidden2out
  bool dcat_b_ok = false;                          // dcat_b's generated half holds [x | adv(y_hat_static)] of the tensors below
  const float* dcat_b_x = nullptr; const float* dcat_b_yhs = nullptr;
  std::vector<LinShadow> wsh[2];                   // per role: bf16 shadows of the hidden layers' weights, then of the last layer's
  SlabDefer sdefer[2];                             // per role: deferred weight-gradient combines of the fused step
  Scratch w0pad[2];                                // per role: first hidden layer's weight with a 16-byte row pitch (stack_forward)
  unsigned int* h_fault_dev = nullptr;             // device view of h_fault[1]: the optimizer kernel mirrors a raised fault word
  unsigned int* d_fault = nullptr;                 // device fault word of the persistent kernels (0 = ok)
  unsigned int* h_fault = nullptr;                 // pinned mirror, refreshed behind every persistent launch
  bool lstm_persistent = getenv("GT_LSTM_STEPS") == nullptr;   // GT_OPT_LSTM_PERSISTENT
  int lstm_fwd_upc = 0;                            // 0 = automatic
  bool lstm_xcd_local = getenv("GT_LSTM_NO_XCD_LOCAL") == nullptr;   // GT_OPT_LSTM_XCD_LOCAL
  bool matmul_bf16 = false;                                          // GT_OPT_MATMUL_BF16
  // sequence lengths travel on the step stream through a small ring (pinned host slot -> device slot): the kernels of
  // the previous step, still queued when the next batch's lengths arrive, keep reading THEIR slot
  static constexpr int LEN_RING = 4;
  int* len_host[LEN_RING] = {nullptr, nullptr, nullptr, nullptr};
  Scratch len_dev[LEN_RING];
  hipEvent_t len_ev[LEN_RING] = {nullptr, nullptr, nullptr, nullptr};
  int len_cap = 0, len_slot = -1;
  int* d_lengths() { return len_slot < 0 ? nullptr : len_dev[len_slot].as<int>(); }
  std::vector<Scratch> s_wt;                                 // SRU float32 mode: transposed copies of the layers' W
  std::vector<Scratch> s_u, s_h, s_c, s_xdrop, s_xmask;     // SRU per-layer stashes (s_xmask: input-dropout multipliers [B][n_in])
  Scratch s_du, s_dx, s_dbias;
  std::vector<int> h_lengths;
  StepScalars* sc() { return scal.as<StepScalars>(); }
  StepResults* res() { return (StepResults*)((char*)scal.p + 256); }
};

static int upload_ints(const std::vector<int>& v, int** dptr) {
  if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
  if (v.empty()) return GT_OK;
  HIPCHK(hipMalloc((void**)dptr, v.size() * sizeof(int)));
  HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return GT_OK;
}

extern "C" int gt_engine_create(const gt_stream_config* cfg, gt_engine** out) {
  if (!cfg || !out) return fail(GT_ERR_INVALID, "null argument");
  if (cfg->n_streams < 1 || cfg->n_streams > GT_MAX_STREAMS) return fail(GT_ERR_INVALID, "n_streams out of range");
  if (cfg->num_windows < 1 || cfg->num_windows > MLPG_MAXW) return fail(GT_ERR_INVALID, "num_windows must be in [1,%d]", MLPG_MAXW);
  gt_engine* e = new gt_engine();
  e->cfg = *cfg;
  // static layout: get_static_stream_sizes (multistream.py:46-53) + per-column source map
  int col = 0, scol_out = 0;
  std::vector<int> static_start, static_size;
  for (int s = 0; s < cfg->n_streams; ++s) {
    const int sz = cfg->stream_sizes[s];
    const bool dyn = cfg->has_dynamic_features[s] != 0;
    const int ss = dyn ? sz / cfg->num_windows : sz;
    static_start.push_back(scol_out);
    static_size.push_back(ss);
    for (int c = 0; c < ss; ++c) {
      e->h_scol.push_back(col + c);
      e->h_sstride.push_back(dyn ? ss : 0);
    }
    col += sz;
    scol_out += ss;
  }
  e->Dout_cfg = col;
  e->Ds = scol_out;
  // adversarial columns: select_streams on the static layout, then drop the first n (train.py:232-242)
  if (cfg->adversarial_streams[0] < 0) {
    for (int c = 0; c < e->Ds; ++c) e->h_adv_cols.push_back(c);
  } else {
    for (int s = 0; s < cfg->n_streams; ++s)
      if (cfg->adversarial_streams[s])
        for (int c = 0; c < static_size[s]; ++c) e->h_adv_cols.push_back(static_start[s] + c);
    if (cfg->mask_nth_mgc_for_adv_loss > 0) {
      if ((size_t)cfg->mask_nth_mgc_for_adv_loss >= e->h_adv_cols.size()) { delete e; return fail(GT_ERR_INVALID, "mask_nth_mgc_for_adv_loss too large"); }
      e->h_adv_cols.erase(e->h_adv_cols.begin(), e->h_adv_cols.begin() + cfg->mask_nth_mgc_for_adv_loss);
    }
  }
  e->Da = (int)e->h_adv_cols.size();
  e->h_adv_inv.assign(e->Ds, -1);
  for (int j = 0; j < e->Da; ++j) e->h_adv_inv[e->h_adv_cols[j]] = j;
  int r;
  if ((r = upload_ints(e->h_scol, &e->d_scol)) || (r = upload_ints(e->h_sstride, &e->d_sstride)) ||
      (r = upload_ints(e->h_adv_cols, &e->d_adv_cols)) || (r = upload_ints(e->h_adv_inv, &e->d_adv_inv))) { delete e; return r; }
  if ((r = e->scal.ensure(1024))) { delete e; return r; }
  if (hipMemset(e->scal.p, 0, 1024) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipMemset failed"); }
  if (hipHostMalloc((void**)&e->h_res, sizeof(StepResults)) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipHostMalloc failed"); }
  if (hipHostGetDevicePointer((void**)&e->h_res_dev, e->h_res, 0) != hipSuccess) { (void)hipGetLastError(); e->h_res_dev = nullptr; }
  if (hipMalloc((void**)&e->d_fault, 64) != hipSuccess || hipMemset(e->d_fault, 0, 64) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipMalloc failed"); }
  if (hipHostMalloc((void**)&e->h_fault, 64) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipHostMalloc failed"); }
  for (int i = 0; i < 16; ++i) e->h_fault[i] = 0;   // [0] copy of the device word, [1] its mirror by the optimizer kernel, [2 + role] skipped steps
  if (hipHostGetDevicePointer((void**)&e->h_fault_dev, e->h_fault, 0) != hipSuccess) { (void)hipGetLastError(); e->h_fault_dev = nullptr; }
  else e->h_fault_dev += 1;
  *out = e;
  return GT_OK;
}

extern "C" void gt_engine_destroy(gt_engine* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  for (auto& s : e->g_act) s.release();
  for (auto& s : e->d_act) s.release();
  for (auto* v : {&e->l_xproj, &e->l_gates, &e->l_cst, &e->l_out, &e->l_outd}) for (auto& s : *v) s.release();
  e->i2o_gout.release();
  (void)gt_comm_destroy(e);
  e->comm_tv.release();
  e->l_state.release(); e->l_dout.release(); e->l_hshift.release(); e->l_xch.release();
  if (e->d_fault) (void)hipFree(e->d_fault);
  if (e->h_fault) (void)hipHostFree(e->h_fault);
  for (int i = 0; i < gt_engine::LEN_RING; ++i) {
    e->len_dev[i].release();
    if (e->len_host[i]) (void)hipHostFree(e->len_host[i]);
    if (e->len_ev[i]) (void)hipEventDestroy(e->len_ev[i]);
  }
  for (auto* v : {&e->s_u, &e->s_h, &e->s_c, &e->s_xdrop, &e->s_xmask, &e->s_wt}) for (auto& s : *v) s.release();
  e->s_du.release(); e->s_dx.release(); e->s_dbias.release();
  Scratch* all[] = {&e->dcat, &e->dzA, &e->dzB, &e->leak, &e->gadv, &e->gs, &e->gy, &e->slabs, &e->colp, &e->partial,
                    &e->headp, &e->headw, &e->dmask, &e->tx, &e->gx, &e->dgx, &e->dtz, &e->dout, &e->scal, &e->mlpg.tmp};
  for (auto* s : all) s->release();
  e->w0pad[0].release(); e->w0pad[1].release();
  for (auto* v : {&e->g_actb, &e->d_actb}) for (auto& b : *v) b.release();
  e->xin_b.release(); e->dcat_b.release(); e->gy_b.release(); e->dz_b[0].release(); e->dz_b[1].release(); e->fwd_b.release();
  for (int r = 0; r < 2; ++r) for (auto& w : e->wsh[r]) { w.w.release(); w.wt.release(); }
  for (auto& b : e->l_in_b) b.release();
  for (auto& b : e->s_in_b) b.release();
  e->s_du_b.release();
  for (auto& w : e->ssh) { w.w.release(); w.wt.release(); }
  for (auto& b : e->l_dg_b) b.release();
  e->l_hs_b.release(); e->slabs_side.release(); e->colp_side.release();
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->ev_side_go) (void)hipEventDestroy(e->ev_side_go);
  if (e->ev_side_done) (void)hipEventDestroy(e->ev_side_done);
  for (auto& w : e->lsh) { w.w.release(); w.wt.release(); }
  e->sdefer[0].pool.release(); e->sdefer[1].pool.release();
  e->mlpg.clear();
  int* ints[] = {e->d_scol, e->d_sstride, e->d_adv_cols, e->d_adv_inv, e->d_scol_i2o, e->d_sstride_i2o};
  for (int* p : ints) if (p) (void)hipFree(p);
  if (e->h_res) (void)hipHostFree(e->h_res);
  if (e->ev_res) (void)hipEventDestroy(e->ev_res);
  for (int r = 0; r < 2; ++r) { if (e->h_def[r]) (void)hipHostFree(e->h_def[r]); if (e->ev_def[r]) (void)hipEventDestroy(e->ev_def[r]); }
  delete e;
}

static long expected_params(const gt_model_desc& d) {
  long n = 0;
  if (has_lstm_body(d.arch)) {
    const int H = d.hidden_dim, dirs = d.bidirectional ? 2 : 1;
    if (d.arch == GT_ARCH_IN2OUT_RNN) n += (long)d.static_dim * d.static_dim + d.static_dim;
    for (int l = 0; l < d.num_hidden; ++l) {
      const int in = l == 0 ? d.in_dim : H * dirs;
      n += (long)dirs * (4L * H * in + 4L * H * H + 8L * H);
    }
    return n + (long)d.out_dim * H * dirs + d.out_dim;
  }
  if (d.arch == GT_ARCH_SRU) {
    const int ncols = d.hidden_dim * (d.bidirectional ? 2 : 1);
    for (int l = 0; l < d.num_hidden; ++l) {
      const int in = l == 0 ? d.in_dim : ncols;
      n += (long)in * ncols * (in == ncols ? 3 : 4) + 2L * ncols;
    }
    return n + (long)d.out_dim * ncols + d.out_dim;
  }
  if (d.arch == GT_ARCH_IN2OUT) n += (long)d.static_dim * d.static_dim + d.static_dim;
  int in = d.in_dim;
  for (int l = 0; l < d.num_hidden; ++l) { n += (long)d.hidden_dim * in + d.hidden_dim; in = d.hidden_dim; }
  n += (long)d.out_dim * in + d.out_dim;
  return n;
}

extern "C" int gt_bind_model(gt_engine* e, int role, const gt_model_desc* desc) {
  if (!e || !desc || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  if (desc->arch < GT_ARCH_MLP || desc->arch > GT_ARCH_IN2OUT_RNN)
    return fail(GT_ERR_INVALID, "unsupported arch %d", desc->arch);
  if (desc->num_hidden < 1 || desc->num_hidden > 16) return fail(GT_ERR_INVALID, "num_hidden must be in [1,16]");
  if (desc->dropout < 0.f || desc->dropout >= 1.f) return fail(GT_ERR_INVALID, "dropout must be in [0,1)");
  if (!desc->params) return fail(GT_ERR_INVALID, "params is null");
  if (desc->n_params != expected_params(*desc))
    return fail(GT_ERR_INVALID, "n_params %ld does not match the architecture (%ld)", (long)desc->n_params, expected_params(*desc));
  if (desc->arch == GT_ARCH_LSTM && desc->hidden_dim < 1) return fail(GT_ERR_INVALID, "hidden_dim must be positive");
  if (role == GT_ROLE_D && (desc->arch != GT_ARCH_MLP || desc->out_dim != 1 || !desc->last_sigmoid))
    return fail(GT_ERR_INVALID, "discriminator must be MLP(out_dim=1, last_sigmoid=True) (hparams.py:56-64,230-239)");
  Net& n = e->net[role];
  n.d = *desc;
  n.hidden.clear();
  float* p = desc->params;
  float* g = desc->grads;
  auto take = [&](int out, int in) {
    Lin l;
    l.in = in; l.out = out;
    l.W = p; l.dW = g; p += (long)out * in; if (g) g += (long)out * in;
    l.b = p; l.db = g; p += out; if (g) g += out;
    return l;
  };
  n.lstm.clear();
  n.sru.clear();
  if (has_lstm_body(desc->arch)) {
    const int H = desc->hidden_dim, dirs = desc->bidirectional ? 2 : 1;
    if (desc->arch == GT_ARCH_IN2OUT_RNN) {
      if (desc->in_dim != desc->out_dim)
        return fail(GT_ERR_DIM, "In2OutRNNHighwayNet returns its input as y_hat (models.py:118): in_dim must equal out_dim");
      n.gate = take(desc->static_dim, desc->static_dim);
    }
    auto adv = [&](float*& wp, float*& gp, long cnt) { wp = p; gp = g; p += cnt; if (g) g += cnt; };
    for (int l = 0; l < desc->num_hidden; ++l) {
      LstmLayerP L;
      memset(&L, 0, sizeof(L));
      L.in = l == 0 ? desc->in_dim : H * dirs;
      for (int dd = 0; dd < dirs; ++dd) {
        adv(L.d[dd].Wih, L.d[dd].dWih, 4L * H * L.in);
        adv(L.d[dd].Whh, L.d[dd].dWhh, 4L * H * H);
        adv(L.d[dd].bih, L.d[dd].dbih, 4L * H);
        adv(L.d[dd].bhh, L.d[dd].dbhh, 4L * H);
      }
      n.lstm.push_back(L);
    }
    n.last = take(desc->out_dim, H * dirs);
    e->l_xproj.resize(desc->num_hidden); e->l_gates.resize(desc->num_hidden);
    e->l_cst.resize(desc->num_hidden); e->l_out.resize(desc->num_hidden); e->l_outd.resize(desc->num_hidden);
  } else if (desc->arch == GT_ARCH_SRU) {
    if (desc->rnn_dropout < 0.f || desc->rnn_dropout >= 1.f) return fail(GT_ERR_INVALID, "rnn_dropout must be in [0,1)");
    if (desc->num_hidden > 8) return fail(GT_ERR_INVALID, "SRURNN: at most 8 layers (two dropout sites per layer)");
    const int ncols = desc->hidden_dim * (desc->bidirectional ? 2 : 1);
    n.sru.clear();
    for (int l = 0; l < desc->num_hidden; ++l) {
      SruLayerP L;
      L.in = l == 0 ? desc->in_dim : ncols;
      L.k = L.in == ncols ? 3 : 4;
      const long nw = (long)L.in * ncols * L.k;
      L.W = p; L.dW = g; p += nw; if (g) g += nw;
      L.b = p; L.db = g; p += 2 * ncols; if (g) g += 2 * ncols;
      n.sru.push_back(L);
    }
    n.last = take(desc->out_dim, ncols);
    e->s_u.resize(desc->num_hidden); e->s_h.resize(desc->num_hidden);
    e->s_c.resize(desc->num_hidden); e->s_xdrop.resize(desc->num_hidden); e->s_xmask.resize(desc->num_hidden); e->s_wt.resize(desc->num_hidden);
  } else {
    if (desc->arch == GT_ARCH_IN2OUT) n.gate = take(desc->static_dim, desc->static_dim);
    int in = desc->in_dim;
    for (int l = 0; l < desc->num_hidden; ++l) { n.hidden.push_back(take(desc->hidden_dim, in)); in = desc->hidden_dim; }
    n.last = take(desc->out_dim, in);
  }
  n.bound = true;
  n.grads_dirty = false;
  auto& acts = role == GT_ROLE_G ? e->g_act : e->d_act;
  acts.resize(desc->num_hidden);
  if (role == GT_ROLE_G && is_i2o(desc->arch)) {
    // single dynamic stream of width out_dim (models.py:66,115)
    const int sd = desc->out_dim / e->cfg.num_windows;
    std::vector<int> sc(sd), ss(sd, sd);
    for (int c = 0; c < sd; ++c) sc[c] = c;
    e->i2o_ds = sd;
    CHK(upload_ints(sc, &e->d_scol_i2o));
    CHK(upload_ints(ss, &e->d_sstride_i2o));
    if (sd != desc->static_dim) return fail(GT_ERR_DIM, "In2OutHighwayNet: out_dim/num_windows (%d) != static_dim (%d)", sd, desc->static_dim);
  }
  if (role == GT_ROLE_G) e->g_pass_valid = false;
  return GT_OK;
}

extern "C" int gt_bind_optimizer(gt_engine* e, int role, const gt_optim_desc* od) {
  if (!e || !od || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  Net& n = e->net[role];
  if (!n.bound) return fail(GT_ERR_STATE, "bind the model before its optimizer");
  if (!n.d.grads) return fail(GT_ERR_INVALID, "model was bound without a grads buffer");
  if (od->kind != GT_OPT_ADAGRAD && od->kind != GT_OPT_ADAM) return fail(GT_ERR_INVALID, "unknown optimizer kind");
  if (!od->state0 || (od->kind == GT_OPT_ADAM && !od->state1)) return fail(GT_ERR_INVALID, "optimizer state buffer is null");
  n.od = *od;
  n.step = od->step;
  n.has_opt = true;
  return GT_OK;
}
extern "C" int gt_set_training(gt_engine* e, int role, int training) {
  if (!e || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].training = training != 0;
  return GT_OK;
}
extern "C" int gt_set_lr(gt_engine* e, int role, float lr) {
  if (!e || role < 0 || role > 1 || !e->net[role].has_opt) return fail(GT_ERR_INVALID, "no optimizer bound");
  e->net[role].od.lr = lr;
  return GT_OK;
}
extern "C" int gt_get_optimizer_step(gt_engine* e, int role, int64_t* step) {
  if (!e || role < 0 || role > 1 || !step) return fail(GT_ERR_INVALID, "bad argument");
  *step = e->net[role].step;
  return GT_OK;
}
extern "C" int gt_set_seed(gt_engine* e, uint64_t seed) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->seed = seed;
  e->step_counter = 0;
  return GT_OK;
}
extern "C" int gt_set_dropout_mask(gt_engine* e, int role, int pass, int layer, const float* mask) {
  if (!e || role < 0 || role > 1 || pass < 0 || pass > 2 || layer < 0 || layer > 15) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].inj[pass][layer] = mask;
  return GT_OK;
}
extern "C" int gt_set_option(gt_engine* e, int option, int value) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  switch (option) {
    case GT_OPT_LSTM_PERSISTENT: e->lstm_persistent = value != 0; return GT_OK;
    case GT_OPT_LSTM_FWD_UNITS: e->lstm_fwd_upc = value; return GT_OK;
    case GT_OPT_LSTM_XCD_LOCAL: e->lstm_xcd_local = value != 0; return GT_OK;
    case GT_OPT_MATMUL_BF16:
      // the storage precision belongs to a PASS: buffers of a stashed forward pass (bf16 images vs float32 stashes) are not
      // interchangeable, so a change drops whatever is stashed -- the next update_* then asks for a fresh apply_generator
      // instead of back-propagating through buffers the forward never filled
      if (e->matmul_bf16 != (value != 0)) {
        e->g_pass_valid = false; e->fake_cat_valid = false; e->dcat_b_ok = false; e->leak_pending = false;
        e->d_begin_done = false; e->g_begin_done = false;
      }
      e->matmul_bf16 = value != 0;
      return GT_OK;
  }
  return fail(GT_ERR_INVALID, "unknown option %d", option);
}
extern "C" int gt_set_loss_normalizer(gt_engine* e, float tv) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->tv_override = tv;
  e->tv_dev = nullptr;
  return GT_OK;
}
extern "C" int gt_set_loss_normalizer_device(gt_engine* e, const double* tv_dev) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->tv_dev = tv_dev;
  e->tv_mask = nullptr; e->tv_inflight = false;          // re-read on the next step function
  return GT_OK;
}
extern "C" int gt_set_lengths(gt_engine* e, const int64_t* lengths_host, int B, void* stream) {
  if (!e || !lengths_host || B < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  e->h_lengths.resize(B);
  for (int b = 0; b < B; ++b) {
    if (lengths_host[b] < 0 || lengths_host[b] > 0x3fffffff) return fail(GT_ERR_INVALID, "length out of range");
    e->h_lengths[b] = (int)lengths_host[b];
  }
  if (B > e->len_cap) {          // (re)allocate the pinned slots; device slots grow on demand
    HIPCHK(hipStreamSynchronize(s));
    const int cap = std::max(64, B + B / 2);
    for (int i = 0; i < gt_engine::LEN_RING; ++i) {
      if (e->len_ev[i]) HIPCHK(hipEventSynchronize(e->len_ev[i]));
      if (e->len_host[i]) { HIPCHK(hipHostFree(e->len_host[i])); e->len_host[i] = nullptr; }
      HIPCHK(hipHostMalloc((void**)&e->len_host[i], (size_t)cap * sizeof(int)));
      if (!e->len_ev[i]) HIPCHK(hipEventCreateWithFlags(&e->len_ev[i], hipEventDisableTiming));
    }
    e->len_cap = cap;
  }
  const int slot = (e->len_slot + 1) % gt_engine::LEN_RING;
  HIPCHK(hipEventSynchronize(e->len_ev[slot]));          // the copy that last used this pinned slot (4 batches ago)
  memcpy(e->len_host[slot], e->h_lengths.data(), (size_t)B * sizeof(int));
  CHK(e->len_dev[slot].ensure((size_t)e->len_cap * sizeof(int)));
  HIPCHK(hipMemcpyAsync(e->len_dev[slot].p, e->len_host[slot], (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(e->len_ev[slot], s));
  e->len_slot = slot;
  return GT_OK;
}

extern "C" int gt_zero_grad(gt_engine* e, int role) {
  if (!e || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].grads_dirty = false;   // lazily: the next backward overwrites
  { SlabDefer& sd = e->sdefer[role]; sd.active = false; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; }   // nothing recorded survives a zero_grad
  e->tv_mask = nullptr; e->tv_inflight = false;
  if (role == GT_ROLE_G) e->leak_pending = false;
  return GT_OK;
}
extern "C" int gt_scalar_buffer(gt_engine* e, double** dev_ptr, int* n) {
  if (!e || !dev_ptr || !n) return fail(GT_ERR_INVALID, "bad argument");
  *dev_ptr = &e->sc()->s_real;
  *n = 7;   // D step: s_real, s_fake, n_real_ok, n_fake_ok | G step: s_adv, s_mge, s_mse
  return GT_OK;
}

// Philox dropout site (role, pass, layer) of engine step `step`: the keep decision of element (row, col) is
// philox_keep(key0, key1, thresh, row, col) (gemm_f32.hip.h) -- a function of the site and the element only, not of
// the tiling of whichever kernel applies it.  The keep probability is quantised to 16 bits (thresh = round(p * 2^16)):
// exact for p = k / 65536 (0.5, 0.25, ...), otherwise |P(keep) - (1-p)| <= 2^-17 while the survivors are scaled by
// the nominal 1/(1-p) like nn.Dropout does.
// Data parallel ("Dropout/noise RNG keyed by global sequence index so DP=k reproduces DP=1"): with world > 1 the
// site's row groups are mapped to the groups the same frames have in the one-process minibatch (DropoutSpec::dp_*,
// philox_group in gemm_f32.hip.h), so a world-k run draws exactly the masks a world-1 run draws for the whole minibatch
// (the reference draws ONE mask over the whole minibatch: models.py:139, train.py:538-585).  `half_rows`: rows of one half of
// a [real | generated] two-half pass, 0 for a single block.  The map needs whole 16-row groups per sequence (T % 16 == 0);
// for other T the rank is folded into the key instead (independent masks per rank: valid dropout, not world-1's bits).
static DropoutSpec philox_site_spec(gt_engine* e, int role, int pass, int layer, uint64_t step, float p, long half_rows = 0) {
  DropoutSpec d = no_drop();
  d.p = p;
  d.scale = 1.f / (1.f - p);
  d.mode = DROP_PHILOX;
  const double th = (double)p * 65536.0 + 0.5;
  d.thresh = th >= 65535.0 ? 65535u : (uint32_t)th;
  const uint64_t site = step * 64ULL + (uint64_t)(role * 32 + pass * 16 + layer);
  uint32_t rk = 0;
  if (e->dp_world > 1) {
    const int T = e->chk_T, B = e->chk_B;
    const long groups = ((long)B * T / 16) * e->dp_world * 2;
    if (T > 0 && T % 16 == 0 && groups < (1L << 21) && (half_rows == 0 || half_rows == (long)B * T)) {
      d.dp_t16 = (uint32_t)(T / 16);
      d.dp_inv_t16 = 1.f / (float)d.dp_t16;
      d.dp_nl16 = half_rows ? (uint32_t)(half_rows / 16) : 0xffffffffu;
      d.dp_half = half_rows ? (uint32_t)((long)B * e->dp_world * (T / 16) - half_rows / 16) : 0u;
      d.dp_add = (uint32_t)e->dp_rank * d.dp_t16;
      d.dp_mul = (uint32_t)(e->dp_world - 1) * d.dp_t16;
    } else {
      rk = (uint32_t)e->dp_rank;
    }
  }
  d.key0 = (uint32_t)(e->seed ^ (site * 0x9E3779B97F4A7C15ULL)) ^ (rk * 0x85EBCA6Bu);
  d.key1 = (uint32_t)((e->seed >> 32) ^ (site >> 7) ^ 0xA5A5A5A5u) + (uint32_t)site + rk * 0xC2B2AE35u;
  return d;
}

// dropout spec of (role, pass, layer).  rows_off: first row of `pass` inside the stacked mask buffer.
static DropoutSpec drop_spec(gt_engine* e, int role, int pass, int layer, const float* stacked_mask, int ld, long half_rows = 0) {
  Net& n = e->net[role];
  DropoutSpec d = no_drop();
  if (!n.training || n.d.dropout <= 0.f) return d;
  if (stacked_mask) {
    d.p = n.d.dropout;
    d.scale = 1.f / (1.f - n.d.dropout);
    d.mode = DROP_BUFFER; d.mask = stacked_mask; d.ld_mask = ld;
    return d;
  }
  return philox_site_spec(e, role, pass, layer, e->step_counter, n.d.dropout, half_rows);
}

// Parity hook: the 0/1 keep mask the engine's Philox stream assigns to dropout site (role, pass, layer) of the step
// that starts `steps_ahead` apply_generator calls from now (1 = the next one), for a (rows, cols) activation.
extern "C" int gt_op_philox_mask(gt_engine* e, int role, int pass, int layer, int64_t steps_ahead, float p, int64_t rows, int cols,
                                 float* mask, void* stream) {
  if (!e || !mask || role < 0 || role > 1 || pass < 0 || pass > 2 || layer < 0 || layer > 15 || rows < 1 || cols < 1 ||
      steps_ahead < 0 || !(p > 0.f && p < 1.f) || rows > 0x7fffffffL)
    return fail(GT_ERR_INVALID, "bad argument");
  // (data parallel: the row-group map of the last entry point's (B, T); a 2*B*T-row request is the D step's two-half pass)
  const DropoutSpec d = philox_site_spec(e, role, pass, layer, e->step_counter + (uint64_t)steps_ahead, p,
                                         rows == 2L * e->chk_B * e->chk_T ? (long)e->chk_B * e->chk_T : 0L);
  hipLaunchKernelGGL(philox_mask_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, d, rows, cols, mask);
  LAUNCH_CHECK();
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// MLPG band cache
// ------------------------------------------------------------------------------------------
static int ensure_band(gt_engine* e, const float* R, int T, hipStream_t s) {
  MlpgCache& m = e->mlpg;
  ++m.tick;
  for (auto* b : m.entries)
    if (b->R == R && b->T == T) { b->last_use = m.tick; m.cur = b; return GT_OK; }
  const int nW = e->cfg.num_windows;
  // first sight of this (R, T): per-offset maxima -> host, pick the smallest half-width whose outside is negligible
  CHK(m.tmp.ensure((size_t)(2 * T - 1) * sizeof(float)));
  hipLaunchKernelGGL(mlpg_offset_max_kernel, dim3(2 * T - 1), dim3(256), 0, s, R, T, nW, m.tmp.as<float>());
  LAUNCH_CHECK();
  std::vector<float> off(2 * T - 1);
  HIPCHK(hipMemcpyAsync(off.data(), m.tmp.p, off.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  float peak = 0.f;
  for (float v : off) peak = fmaxf(peak, v);
  if (!(peak > 0.f) || !isfinite(peak)) return fail(GT_ERR_INVALID, "MLPG matrix R is empty or not finite");
  int kb = 0;
  for (int o = -(T - 1); o <= T - 1; ++o)
    if (off[o + T - 1] > 1e-9f * peak) kb = std::max(kb, abs(o));
  if (kb > 63 || (kb > 48 && kb > T / 4))
    return fail(GT_ERR_INVALID, "MLPG matrix R is not banded (half-width %d of T=%d): only window sets whose "
                "R = (W^T W)^-1 W^T decays (hparams.py:22-26) are supported", kb, T);
  MlpgBand* b = nullptr;
  if (m.entries.size() >= MlpgCache::MAX_ENTRIES) {      // recycle the least recently used entry
    size_t lru = 0;
    for (size_t i = 1; i < m.entries.size(); ++i) if (m.entries[i]->last_use < m.entries[lru]->last_use) lru = i;
    b = m.entries[lru];
    HIPCHK(hipStreamSynchronize(s));                      // its band may still be read by queued kernels
  } else {
    b = new MlpgBand();
    m.entries.push_back(b);
  }
  const int nb = 2 * kb + 1;
  b->R = nullptr;
  CHK(b->band.ensure((size_t)T * nW * nb * sizeof(float)));
  hipLaunchKernelGGL(mlpg_extract_band_kernel, dim3(cdiv((long)T * nW * nb, 256)), dim3(256), 0, s, R, T, nW, kb, b->band.as<float>());
  LAUNCH_CHECK();
  b->R = R; b->T = T; b->kb = kb; b->last_use = m.tick;
  m.cur = b;
  return GT_OK;
}
extern "C" int gt_invalidate_mlpg_cache(gt_engine* e) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipDeviceSynchronize());
  e->mlpg.clear();
  return GT_OK;
}

static int mlpg_forward(gt_engine* e, const float* y, int ldy, const int* scol, const int* sstride, int Ds,
                        float* ys, int ldys, int B, int T, hipStream_t s) {
  const int nW = e->cfg.num_windows, kb = e->mlpg.cur->kb;
  const size_t lds = ((size_t)(MLPG_TT + 2 * kb) * nW * MLPG_CC + (size_t)MLPG_TT * nW * (2 * kb + 1 + 2 * MLPG_PAD)) * sizeof(float);
  dim3 grid(B * cdiv(T, MLPG_TT), cdiv(Ds, MLPG_CC));
  static const int fpl = getenv("GT_MLPG_FPL") ? atoi(getenv("GT_MLPG_FPL")) : 2;   // frames per lane of the compute phase: 2 measured best (4: 24.7 us, 2: 21.8, 1: 26.6)
#define GT_MLPG_FWD(F) { CHK(ensure_dyn_lds((const void*)mlpg_forward_kernel<F>, lds)); \
    hipLaunchKernelGGL(mlpg_forward_kernel<F>, grid, dim3(MLPG_THREADS), lds, s, y, ldy, e->mlpg.cur->band.as<float>(), kb, nW, scol, sstride, Ds, ys, ldys, B, T); }
  if (fpl == 1) GT_MLPG_FWD(1) else if (fpl == 2) GT_MLPG_FWD(2) else GT_MLPG_FWD(4)
#undef GT_MLPG_FWD
  LAUNCH_CHECK();
  return GT_OK;
}
static int mlpg_backward(gt_engine* e, const float* gs, int ldgs, const int* scol, const int* sstride, int Ds,
                         float* gy, int ldgy, int B, int T, float mse_w, const float* yhat, const float* ytgt, int ldt,
                         const float* mask, hipStream_t s) {
  const int nW = e->cfg.num_windows, kb = e->mlpg.cur->kb;
  const size_t lds = ((size_t)(MLPG_TT + 2 * kb) * MLPG_CC + (size_t)(MLPG_TT + 2 * kb) * nW * (2 * kb + 1 + 2 * MLPG_PAD)) * sizeof(float);
  dim3 grid(B * cdiv(T, MLPG_TT), cdiv(Ds, MLPG_CC));
  static const int fpl = getenv("GT_MLPG_FPL") ? atoi(getenv("GT_MLPG_FPL")) : 2;   // frames per lane of the compute phase: 2 measured best (4: 24.7 us, 2: 21.8, 1: 26.6)
#define GT_MLPG_BWD(F) { CHK(ensure_dyn_lds((const void*)mlpg_backward_kernel<F>, lds)); \
    hipLaunchKernelGGL(mlpg_backward_kernel<F>, grid, dim3(MLPG_THREADS), lds, s, gs, ldgs, e->mlpg.cur->band.as<float>(), kb, nW, scol, sstride, Ds, \
                       gy, ldgy, B, T, mse_w, yhat, ytgt, ldt, mask, e->sc()); }
  if (fpl == 1) GT_MLPG_BWD(1) else if (fpl == 2) GT_MLPG_BWD(2) else GT_MLPG_BWD(4)
#undef GT_MLPG_BWD
  LAUNCH_CHECK();
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// network passes
// ------------------------------------------------------------------------------------------
// stacked injected-mask buffer for one layer of a D pass group (real rows then fake rows)
static int stage_injected(gt_engine* e, int role, int layer, const int* passes, int npass, long rows_each, int width,
                          const float** out, hipStream_t s) {
  Net& n = e->net[role];
  *out = nullptr;
  if (!n.training || n.d.dropout <= 0.f) return GT_OK;
  bool any = false, all = true;
  for (int i = 0; i < npass; ++i) { any |= n.inj[passes[i]][layer] != nullptr; all &= n.inj[passes[i]][layer] != nullptr; }
  if (!any) return GT_OK;
  if (!all) return fail(GT_ERR_INVALID, "injected dropout masks must be given for every pass of a step or for none");
  if (npass == 1) { *out = n.inj[passes[0]][layer]; return GT_OK; }
  const size_t per_layer = (size_t)npass * rows_each * width * sizeof(float);
  CHK(e->dmask.ensure(per_layer * n.d.num_hidden));
  float* base = (float*)((char*)e->dmask.p + per_layer * layer);
  for (int i = 0; i < npass; ++i)
    HIPCHK(hipMemcpyAsync(base + (size_t)i * rows_each * width, n.inj[passes[i]][layer], (size_t)rows_each * width * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
  *out = base;
  return GT_OK;
}

// hidden stack forward: in -> acts[0..L-1]; returns specs used (for backward)
static int stack_forward(gt_engine* e, int role, const float* in, int ld_in, long rows, std::vector<Scratch>& acts,
                         const int* passes, int npass, long rows_each, std::vector<DropoutSpec>& specs, hipStream_t s) {
  Net& n = e->net[role];
  specs.resize(n.hidden.size());
  const float* cur = in;
  int ld = ld_in;
  for (size_t l = 0; l < n.hidden.size(); ++l) {
    const Lin& L = n.hidden[l];
    CHK(acts[l].ensure((size_t)rows * L.out * sizeof(float)));
    const float* inj = nullptr;
    CHK(stage_injected(e, role, (int)l, passes, npass, rows_each, L.out, &inj, s));
    specs[l] = drop_spec(e, role, passes[0], (int)l, inj, L.out, npass == 2 ? rows_each : 0);
    const float* W = L.W;
    int ldw = L.in;
    if (l == 0 && (L.in & 3) && gemm_vec_ok(cur, ld) && tl_gemm_prec == PREC_F32) {
      // The input image takes 16-byte loads (the discriminator's [x | adv] image, pitch 484) but the weight rows (483
      // floats) do not: multiply by a copy of W with its row pitch rounded up to 4 floats, re-made from the caller's
      // parameter buffer before every pass (it may have been stepped, loaded or broadcast since).
      ldw = (L.in + 3) & ~3;
      CHK(e->w0pad[role].ensure((size_t)L.out * ldw * sizeof(float)));
      hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv((long)L.out * ldw, 256)), dim3(256), 0, s, L.W, L.in, L.out, e->w0pad[role].as<float>(), ldw);
      LAUNCH_CHECK();
      W = e->w0pad[role].as<float>();
    }
    CHK(linear_forward(cur, ld, W, ldw, L.b, acts[l].as<float>(), L.out, rows, L.in, L.out, ACT_LEAKY_DROPOUT, specs[l], s));
    cur = acts[l].as<float>();
    ld = L.out;
  }
  return GT_OK;
}

static int comm_grads_ready(gt_engine* e, int role, const float* lo, long count, hipStream_t compute);
static int comm_flush(gt_engine* e, int role, hipStream_t compute);
// measurement switches of the data-parallel schedule
static bool comm_d_one_message() { static const bool v = !(getenv("GT_COMM_D_ONE_MSG") && getenv("GT_COMM_D_ONE_MSG")[0] == '0'); return v; }
static bool comm_early_g() { static const bool v = !(getenv("GT_COMM_EARLY_G") && getenv("GT_COMM_EARLY_G")[0] == '0'); return v; }
static bool comm_group() { static const bool v = getenv("GT_COMM_GROUP") && getenv("GT_COMM_GROUP")[0] == '1'; return v; }
// Measured with one rank and forced collectives (bench.py --force-dp, plain step 1.465 ms): round-2 schedule 1.577 ms; the
// discriminator's gradient as ONE message 1.561 (kept); additionally ncclGroupStart/End around a step's closing messages
// 1.571 (off); generator loss sums sent with the closing messages instead of early 1.604 (off: the host then waits for the
// whole step before it can enqueue the next one).

// hidden stack backward.  dz_top: gradient w.r.t. the pre-activation of the TOP hidden layer
// (already multiplied by f'), in buffer `cur` (rows x hidden).  Produces dW/db (if want_w) and,
// optionally, dX[:, col0:col0+ncols] of the stack input for rows [row0, row0+nrows).
static int stack_backward(gt_engine* e, int role, const float* in, int ld_in, long rows, std::vector<Scratch>& acts,
                          const std::vector<DropoutSpec>& specs, float* cur, float* other, bool want_w,
                          float* dX, int lddx, int col0, int ncols, long row0, long nrows, hipStream_t s) {
  Net& n = e->net[role];
  const int L = (int)n.hidden.size();
  {
    // per-layer launches: each layer's weight gradient right behind the product that made its dZ (still warm in L2 / MALL)
    for (int l = L - 1; l >= 0; --l) {
      const Lin& Lr = n.hidden[l];
      const float* Xin = l > 0 ? acts[l - 1].as<float>() : in;
      const int ldx = l > 0 ? n.hidden[l - 1].out : ld_in;
      bool rode = false;
      GemmArgs nn;
      if (l > 0) nn = backward_data_args(cur, Lr.out, Lr.W, Lr.in, 0, other, Lr.in, rows, Lr.out, Lr.in, ACT_LEAKY_DROPOUT,
                                         acts[l - 1].as<float>(), Lr.in, specs[l - 1]);
      if (want_w) {
        CHK(linear_backward_weight(cur, Lr.out, Xin, ldx, rows, Lr.out, Lr.in, Lr.dW, Lr.db, n.grads_dirty, e->slabs, e->colp, s, &e->sdefer[role],
                                   l > 0 ? &nn : nullptr, &rode));
        CHK(comm_grads_ready(e, role, Lr.dW, (long)Lr.out * Lr.in + Lr.out, s));
        // generator: all layers above the first leave as one message under the first layer's backward; the discriminator's
        // whole gradient (1 MB) is ONE message at the end of its backward pass (a second launch costs more than it hides)
        if (l == 1 && (role == GT_ROLE_G || !comm_d_one_message())) CHK(comm_flush(e, role, s));
      }
      if (l > 0) {
        if (!rode) CHK(launch_gemm(GEMM_NN, nn, 1, s));
        std::swap(cur, other);
      } else if (dX) {
        CHK(linear_backward_data(cur + row0 * Lr.out, Lr.out, Lr.W, Lr.in, col0, dX, lddx, nrows, Lr.out, ncols, ACT_NONE, nullptr, 0,
                                 no_drop(), s));
      }
    }
    return GT_OK;
  }
}

// ------------------------------------------------------------------------------------------
// bf16-storage MLP stacks (GT_OPT_MATMUL_BF16; gemm_bf16s.hip.h): activations, dZ, the input images and weight shadows
// live in HBM as bf16, in both orientations; every product is the k-contiguous form.
// ------------------------------------------------------------------------------------------
static bool use_b16(const gt_engine* e, int role) {
  const Net& n = e->net[role];
  if (!e->matmul_bf16 || !n.bound || n.d.arch != GT_ARCH_MLP || (n.d.hidden_dim & 7)) return false;
  return true;
}
// re-made from the caller's float32 parameters before every pass (they may have been stepped, loaded or broadcast since)
static int refresh_shadows(gt_engine* e, int role, bool with_last, hipStream_t s) {
  Net& n = e->net[role];
  auto& sh = e->wsh[role];
  sh.resize(n.hidden.size() + 1);
  CastJobs jobs;
  jobs.n = 0; jobs.pad_ = 0;
  int blocks = 0;
  auto flush = [&]() -> int {
    if (jobs.n > 0) {
      hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(blocks), dim3(256), 0, s, jobs);
      LAUNCH_CHECK();
    }
    jobs.n = 0; blocks = 0;
    return GT_OK;
  };
  for (size_t l = 0; l <= n.hidden.size(); ++l) {     // all layers of the network in ONE launch
    if (l == n.hidden.size() && !with_last) break;
    const Lin& L = l < n.hidden.size() ? n.hidden[l] : n.last;
    LinShadow& w = sh[l];
    w.ldw = pad8(L.in); w.ldwt = pad8(L.out);
    CHK(w.w.ensure((size_t)L.out * w.ldw * 2 + 64));
    CHK(w.wt.ensure((size_t)L.in * w.ldwt * 2 + 64));
    if (jobs.n == CAST_MAX_JOBS) CHK(flush());
    CastJob& J = jobs.j[jobs.n++];
    J.in = L.W; J.ldi = L.in; J.rows = L.out; J.cols = L.in; J.out = w.w.as<__bf16>(); J.ldo = w.ldw; J.outT = w.wt.as<__bf16>(); J.ldt = w.ldwt;
    J.gy = cdiv(L.in, 64); J.block0 = blocks; J.pad_ = 0;
    blocks += cdiv(L.out, 64) * J.gy;
  }
  return flush();
}
// in_b [rows][ld_in] bf16 -> acts[l] (bf16, + transposed twin when want_t: the weight gradients read it)
static int stack_forward_b16(gt_engine* e, int role, const __bf16* in_b, int ld_in, long rows, std::vector<B16Img>& acts,
                             const int* passes, int npass, long rows_each, std::vector<DropoutSpec>& specs, bool want_t, hipStream_t s) {
  Net& n = e->net[role];
  specs.resize(n.hidden.size());
  acts.resize(n.hidden.size());
  const __bf16* cur = in_b;
  int ld = ld_in;
  for (size_t l = 0; l < n.hidden.size(); ++l) {
    const Lin& L = n.hidden[l];
    CHK(acts[l].ensure(rows, L.out, want_t));
    const float* inj = nullptr;
    CHK(stage_injected(e, role, (int)l, passes, npass, rows_each, L.out, &inj, s));
    specs[l] = drop_spec(e, role, passes[0], (int)l, inj, L.out, npass == 2 ? rows_each : 0);
    GemmB16Args g = b16_args();
    g.A = cur; g.lda = ld; g.B = e->wsh[role][l].w.as<__bf16>(); g.ldb = e->wsh[role][l].ldw;
    g.M = (int)rows; g.N = L.out; g.K = L.in; g.bias = L.b; g.epi = B16_FWD; g.act = ACT_LEAKY_DROPOUT; g.drop = specs[l];
    g.Cb = acts[l].r(); g.ldcb = acts[l].ld;
    if (want_t) { g.CbT = acts[l].t(); g.ldcbt = (int)acts[l].ldt; }
    CHK(launch_gemm_b16(g, 1, s));
    cur = acts[l].r();
    ld = acts[l].ld;
  }
  return GT_OK;
}
static int comm_grads_ready(gt_engine* e, int role, const float* lo, long count, hipStream_t compute);
static int comm_flush(gt_engine* e, int role, hipStream_t compute);
// dz[cur]: gradient w.r.t. the pre-activation of the TOP hidden layer (both orientations when want_w).  in_t: transposed
// image of the stack input [in][rows8] (weight gradient of layer 0).  dX (float32, optional): d loss / d input columns
// [col0, col0 + ncols) for rows [row0, row0 + nrows).
static int stack_backward_b16(gt_engine* e, int role, const __bf16* in_t, long ld_int, long rows, std::vector<B16Img>& acts,
                              const std::vector<DropoutSpec>& specs, int cur, bool want_w, float* dX, int lddx, int col0, int ncols,
                              long row0, long nrows, hipStream_t s) {
  Net& n = e->net[role];
  const int L = (int)n.hidden.size();
  for (int l = L - 1; l >= 0; --l) {
    const Lin& Lr = n.hidden[l];
    B16Img& dz = e->dz_b[cur];
    if (want_w) {
      const __bf16* XT = l > 0 ? acts[l - 1].t() : in_t;
      const long ldxt = l > 0 ? acts[l - 1].ldt : ld_int;
      CHK(weight_grad_b16(dz.t(), dz.ldt, XT, ldxt, rows, Lr.out, Lr.in, Lr.dW, Lr.db, n.grads_dirty, e->slabs, s, &e->sdefer[role]));
      CHK(comm_grads_ready(e, role, Lr.dW, (long)Lr.out * Lr.in + Lr.out, s));
      if (l == 1 && (role == GT_ROLE_G || !comm_d_one_message())) CHK(comm_flush(e, role, s));
    }
    if (l > 0) {
      B16Img& nx = e->dz_b[cur ^ 1];
      CHK(nx.ensure(rows, Lr.in, want_w));
      GemmB16Args g = b16_args();
      g.A = dz.r(); g.lda = dz.ld; g.B = e->wsh[role][l].wt.as<__bf16>(); g.ldb = e->wsh[role][l].ldwt;
      g.M = (int)rows; g.N = Lr.in; g.K = Lr.out; g.epi = B16_BWD_DATA; g.act = ACT_LEAKY_DROPOUT;
      g.H = acts[l - 1].r(); g.ldh = acts[l - 1].ld; g.drop = specs[l - 1];
      g.Cb = nx.r(); g.ldcb = nx.ld;
      if (want_w) { g.CbT = nx.t(); g.ldcbt = (int)nx.ldt; }
      CHK(launch_gemm_b16(g, 1, s));
      cur ^= 1;
    } else if (dX) {
      GemmB16Args g = b16_args();
      g.A = dz.r() + row0 * dz.ld; g.lda = dz.ld;
      g.B = e->wsh[role][0].wt.as<__bf16>() + (long)col0 * e->wsh[role][0].ldwt; g.ldb = e->wsh[role][0].ldwt;
      g.M = (int)nrows; g.N = ncols; g.K = Lr.out; g.epi = B16_BWD_DATA; g.act = ACT_NONE;
      g.C = dX; g.ldc = lddx;
      CHK(launch_gemm_b16(g, 1, s));
    }
  }
  return GT_OK;
}

static int cond_dim(gt_engine* e);

// ------------------------------------------------------------------------------------------
// data-parallel communicator (gt_comm_init / gt_comm_destroy).
// One process per GPU; every rank holds the full G / D and a shard of the minibatch (whole sequences).  With a
// communicator attached, the fused step functions are data-parallel by themselves: the valid-frame count, every
// network's gradient and the additive loss sums are summed over the ranks with RCCL (the ROCm build of NCCL, xGMI
// between the GPUs of a node) on a separate HIP stream, each gradient bucket (one layer) as soon as its weight-gradient
// reduction has finished -- i.e. UNDER the rest of the backward pass -- and clip-norm + optimizer run on the reduced
// gradient, so all replicas take bit-identical steps.  RCCL is bound at run time (dlopen), the library has no link
// dependency on it; a host in any language drives this through the C ABI.
// ------------------------------------------------------------------------------------------
#include <dlfcn.h>
struct GtNcclId { char internal[GT_COMM_ID_BYTES]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(GtNcclId*) = nullptr;
  int (*CommInitRank)(void**, int, GtNcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  // GT_RCCL_LIB=<path>: bind that library instead (tests/fake_rccl.cpp -- a shared-memory test double that lets two
  // processes on ONE GPU run a world-2 communicator; RCCL itself refuses two ranks on one device).  Otherwise prefer
  // the copy that is already in the process (PyTorch ships its own librccl), then the ROCm one.
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  const char* forced = getenv("GT_RCCL_LIB");
  if (forced && forced[0]) {
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "gantts_hip: GT_RCCL_LIB=%s could not be loaded: %s\n", forced, dlerror()); return nullptr; }
  }
  if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) return nullptr;
#define GT_SYM(field, name) *(void**)(&api.field) = dlsym(h, name)
  GT_SYM(GetUniqueId, "ncclGetUniqueId"); GT_SYM(CommInitRank, "ncclCommInitRank"); GT_SYM(CommDestroy, "ncclCommDestroy");
  GT_SYM(AllReduce, "ncclAllReduce"); GT_SYM(GroupStart, "ncclGroupStart"); GT_SYM(GroupEnd, "ncclGroupEnd");
  GT_SYM(GetErrorString, "ncclGetErrorString");
#undef GT_SYM
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd) return nullptr;
  api.lib = h;
  return &api;
}
#define NCCLCHK(expr)                                                                                      \
  do {                                                                                                     \
    int _r = (expr);                                                                                       \
    if (_r != 0) return fail(GT_ERR_HIP, "%s failed: %s", #expr, rccl_api()->GetErrorString ? rccl_api()->GetErrorString(_r) : "?"); \
  } while (0)
enum { GT_NCCL_SUM = 0, GT_NCCL_FLOAT = 7, GT_NCCL_DOUBLE = 8 };

struct GtComm {
  void* comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int next_ev = 0;
  hipEvent_t ev_done = nullptr;
};

// A sum over ONE rank is the identity: a single-rank communicator takes the plain path (no collective, no second
// stream).  GT_COMM_FORCE_COLLECTIVES=1 issues every call anyway -- the launch / cross-stream cost of the schedule can
// then be measured on one GPU (bench.py --force-dp).
static inline bool comm_on(const gt_engine* e);

extern "C" int gt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(GT_ERR_INVALID, "null argument");
  RcclApi* api = rccl_api();
  if (!api) return fail(GT_ERR_HIP, "RCCL (librccl.so) could not be loaded");
  NCCLCHK(api->GetUniqueId((GtNcclId*)id_out));
  return GT_OK;
}
extern "C" int gt_comm_destroy(gt_engine* e) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  if (!e->comm) return GT_OK;
  (void)hipDeviceSynchronize();
  GtComm* c = e->comm;
  if (c->comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
  for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  e->comm = nullptr;
  e->dp_rank = 0; e->dp_world = 1;
  return GT_OK;
}
extern "C" int gt_comm_init(gt_engine* e, int rank, int world, const void* id) {
  if (!e || !id || world < 1 || rank < 0 || rank >= world) return fail(GT_ERR_INVALID, "bad argument");
  RcclApi* api = rccl_api();
  if (!api) return fail(GT_ERR_HIP, "RCCL (librccl.so) could not be loaded");
  CHK(gt_comm_destroy(e));
  GtComm* c = new GtComm();
  c->rank = rank; c->world = world;
  e->comm = c;
  e->dp_rank = rank; e->dp_world = world;
  GtNcclId nid;
  memcpy(&nid, id, sizeof(nid));
  int r = api->CommInitRank(&c->comm, world, nid, rank);
  if (r != 0) { c->comm = nullptr; (void)gt_comm_destroy(e); return fail(GT_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "?"); }
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  for (auto& ev : c->ev) HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
  HIPCHK(hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming));
  CHK(e->comm_tv.ensure(64));
  return GT_OK;
}
// The engine's shard of the minibatch without a communicator (hosts that all-reduce themselves between the split-phase
// calls, gantts_amd/parallel.py): sequence b of this engine is sequence rank + world * b of the whole minibatch.  Keys the
// dropout streams globally, so that every rank draws its own rows of the ONE mask a single process would draw.
extern "C" int gt_set_shard(gt_engine* e, int rank, int world) {
  if (!e || world < 1 || rank < 0 || rank >= world) return fail(GT_ERR_INVALID, "bad argument");
  if (e->comm && (e->comm->rank != rank || e->comm->world != world))
    return fail(GT_ERR_STATE, "gt_set_shard(%d, %d) contradicts the attached communicator (%d, %d)", rank, world, e->comm->rank, e->comm->world);
  e->dp_rank = rank; e->dp_world = world;
  return GT_OK;
}
extern "C" int gt_comm_info(gt_engine* e, int* rank, int* world) {
  if (!e || !rank || !world) return fail(GT_ERR_INVALID, "null argument");
  *rank = e->comm ? e->comm->rank : 0;
  *world = e->comm ? e->comm->world : 1;
  return GT_OK;
}

static inline bool comm_on(const gt_engine* e) {
  static const bool force = getenv("GT_COMM_FORCE_COLLECTIVES") != nullptr;
  return e->comm != nullptr && (e->comm->world > 1 || force);
}
// all-reduce(sum) of buf[0..count) in place on the communicator's stream, ordered after everything queued on `compute`
static int comm_allreduce_after(gt_engine* e, void* buf, size_t count, int dtype, hipStream_t compute) {
  GtComm* c = e->comm;
  hipEvent_t ev = c->ev[c->next_ev];
  c->next_ev = (c->next_ev + 1) % 8;
  HIPCHK(hipEventRecord(ev, compute));
  HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
  NCCLCHK(rccl_api()->AllReduce(buf, buf, count, dtype, GT_NCCL_SUM, c->comm, c->stream));
  return GT_OK;
}
// `compute` continues only after everything handed to the communicator so far has finished
static int comm_join(gt_engine* e, hipStream_t compute) {
  GtComm* c = e->comm;
  HIPCHK(hipEventRecord(c->ev_done, c->stream));
  HIPCHK(hipStreamWaitEvent(compute, c->ev_done, 0));
  return GT_OK;
}
// the gradient of [lo, lo + count) of `role` is final on `compute`.  Ranges are collected and handed to RCCL by
// comm_flush in as few messages as their adjacency allows (a small all-reduce is pure latency: the layers above the
// first one leave together, under the first layer's backward; only the first layer's message is exposed).
static int comm_grads_ready(gt_engine* e, int role, const float* lo, long count, hipStream_t compute) {
  if (!comm_on(e) || !lo || count <= 0) return GT_OK;
  Net& n = e->net[role];
  const long off = lo - n.d.grads;
  if (off < 0 || off + count > n.d.n_params) return fail(GT_ERR_STATE, "gradient bucket outside the bound buffer");
  e->comm_pending[role].push_back(std::make_pair(off, count));
  return GT_OK;
}
static int comm_flush(gt_engine* e, int role, hipStream_t compute) {
  if (!comm_on(e)) return GT_OK;
  auto& pend = e->comm_pending[role];
  if (pend.empty()) return GT_OK;
  Net& n = e->net[role];
  std::sort(pend.begin(), pend.end());
  size_t i = 0;
  while (i < pend.size()) {
    long lo = pend[i].first, hi = lo + pend[i].second;
    size_t j = i + 1;
    while (j < pend.size() && pend[j].first <= hi) { hi = std::max(hi, pend[j].first + pend[j].second); ++j; }
    CHK(comm_allreduce_after(e, n.d.grads + lo, (size_t)(hi - lo), GT_NCCL_FLOAT, compute));
    e->comm_done[role].push_back(std::make_pair(lo, hi - lo));
    i = j;
  }
  pend.clear();
  return GT_OK;
}
// end of a backward pass: whatever part of the flat gradient no bucket covered, plus the step's additive loss sums
// (`n_sums` doubles at `sums`), then `compute` waits for the communicator
static int comm_finish_step(gt_engine* e, int role, bool grads, double* sums, int n_sums, hipStream_t compute) {
  if (!comm_on(e)) return GT_OK;
  Net& n = e->net[role];
  const bool grp = comm_group() && rccl_api();
  if (grp) NCCLCHK(rccl_api()->GroupStart());       // the closing messages of a step (rest of the gradient + loss sums): one launch
  if (grads) {
    CHK(comm_flush(e, role, compute));
    auto& done = e->comm_done[role];
    std::sort(done.begin(), done.end());
    long pos = 0;
    for (size_t i = 0; i <= done.size(); ++i) {
      const long next = i < done.size() ? done[i].first : (long)n.d.n_params;
      if (next > pos) CHK(comm_allreduce_after(e, n.d.grads + pos, (size_t)(next - pos), GT_NCCL_FLOAT, compute));
      if (i < done.size()) pos = std::max(pos, done[i].first + done[i].second);
    }
  }
  e->comm_done[role].clear();
  e->comm_pending[role].clear();
  if (sums && n_sums > 0) CHK(comm_allreduce_after(e, sums, (size_t)n_sums, GT_NCCL_DOUBLE, compute));
  if (grp) NCCLCHK(rccl_api()->GroupEnd());
  return comm_join(e, compute);
}

// Data-parallel early results: the step's loss sums are final on `compute` here, long before its backward pass is.
// They are summed over the ranks, finalised and copied to the host ON THE COMMUNICATOR'S STREAM, so the fused call can
// return as soon as that copy lands while the backward pass, its gradient buckets and the optimizer keep going.
static int post_early_results(gt_engine* e, hipStream_t s);
static int comm_early_results(gt_engine* e, int role, double* sums, int n_sums, float adv_w, float mse_w, float mge_w, hipStream_t compute) {
  GtComm* c = e->comm;
  CHK(comm_allreduce_after(e, sums, (size_t)n_sums, GT_NCCL_DOUBLE, compute));
  StepResults* target = e->h_res_dev ? e->h_res_dev : e->res();     // see post_early_results
  if (role == GT_ROLE_D) hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, c->stream, e->sc(), target, 1);
  else hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, c->stream, e->sc(), target, adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0, 1,
                          (const double*)nullptr, 0, (const double*)nullptr, 0);
  LAUNCH_CHECK();
  return post_early_results(e, c->stream);
}

// tv = sum(mask) (or the data-parallel override) -> device scalars; once per (step, mask).  With a communicator the
// count is the GLOBAL one: losses are normalised by the valid frames of the whole minibatch (train.py:258, seqloss.py:43).
// Two halves so that the all-reduce of the count runs under the forward pass that precedes its first use:
// ensure_tv_begin where the mask is first seen, ensure_tv right before the first kernel that reads the normaliser.
static int ensure_tv_begin(gt_engine* e, const float* mask, long N, hipStream_t s) {
  if (e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override) return GT_OK;
  if (comm_on(e) && !e->tv_dev && !(e->tv_override > 0.f) && !e->tv_inflight) {
    hipLaunchKernelGGL(mask_total_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, e->comm_tv.as<double>());
    LAUNCH_CHECK();
    CHK(comm_allreduce_after(e, e->comm_tv.p, 1, GT_NCCL_DOUBLE, s));
    e->tv_inflight = true;
  }
  return GT_OK;
}
static int ensure_tv(gt_engine* e, const float* mask, long N, hipStream_t s) {
  if (e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override) return GT_OK;
  const double* tv_dev = e->tv_dev;
  if (comm_on(e) && !tv_dev && !(e->tv_override > 0.f)) {
    CHK(ensure_tv_begin(e, mask, N, s));
    CHK(comm_join(e, s));
    e->tv_inflight = false;
    tv_dev = e->comm_tv.as<double>();
  }
  hipLaunchKernelGGL(mask_sum_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, e->tv_override, tv_dev, e->sc());
  LAUNCH_CHECK();
  e->tv_mask = mask; e->tv_n = N; e->tv_ovr = e->tv_override;
  return GT_OK;
}

static int check_common(gt_engine* e, int B, int T) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  tl_gemm_prec = e->matmul_bf16 ? PREC_BF16 : PREC_F32;       // every step / forward entry point passes through here
  if (B < 1 || T < 1) return fail(GT_ERR_INVALID, "B and T must be positive");
  if ((long)B * T > 0x3fffffffL) return fail(GT_ERR_INVALID, "B*T too large");
  e->chk_B = B; e->chk_T = T;
  return GT_OK;
}


// ------------------------------------------------------------------------------------------
// recurrent generator (GT_ARCH_LSTM): forward / backward of the LSTM stack
// ------------------------------------------------------------------------------------------
static int lstm_check_lengths(gt_engine* e, int B, int T) {
  if ((int)e->h_lengths.size() != B)
    return fail(GT_ERR_STATE, "recurrent generator: call with lengths (gt_set_lengths) for this batch of %d sequences "
                "(reference models.py:204-210 packs the batch by `lengths`)", B);
  for (int b = 0; b < B; ++b)
    if (e->h_lengths[b] > T) return fail(GT_ERR_INVALID, "length %d exceeds the padded length %d", e->h_lengths[b], T);
  return GT_OK;
}

static int lstm_launch_steps(gt_engine* e, const Net& G, int layer, int B, int T, bool backward, const float* dout, hipStream_t s) {
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  const int Bpad = cdiv(B, 32) * 32;
  const size_t st = (size_t)dirs * Bpad * H;              // floats per state array
  CHK(e->l_state.ensure(5 * st * sizeof(float)));          // h0,h1,c0,c1 (ping-pong) + dc
  float* base = e->l_state.as<float>();
  HIPCHK(hipMemsetAsync(base, 0, 5 * st * sizeof(float), s));
  CHK(ensure_dyn_lds((const void*)lstm_fwd_step_kernel, lstm_lds_bytes()));
  CHK(ensure_dyn_lds((const void*)lstm_bwd_step_kernel, lstm_lds_bytes()));
  LstmStepArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.Bpad = Bpad;
  a.lengths = e->d_lengths();
  for (int d = 0; d < dirs; ++d) { a.Whh[d] = G.lstm[layer].d[d].Whh; a.bih[d] = G.lstm[layer].d[d].bih; a.bhh[d] = G.lstm[layer].d[d].bhh; }
  a.xproj = e->l_xproj[layer].as<float>();
  a.gates = e->l_gates[layer].as<float>();
  a.cst = e->l_cst[layer].as<float>();
  a.out = e->l_out[layer].as<float>();
  a.dout = dout;
  a.dc_state = base + 4 * st;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    if (!backward) {
      const int cur = step & 1;
      a.h_prev = base + (size_t)cur * st;       a.c_prev = base + (2 + (size_t)cur) * st;
      a.h_next = base + (size_t)(cur ^ 1) * st; a.c_next = base + (2 + (size_t)(cur ^ 1)) * st;
      hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(cdiv(H, 8), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
    } else {
      hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3(cdiv(H, 32), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
    }
  }
  LAUNCH_CHECK();
  return GT_OK;
}

// ---- persistent recurrence (lstm_seq_kernels.hip.h): one launch per layer and pass ----
// A persistent launch that gave up (a peer workgroup never published: LSTM_FAULT_*) leaves garbage behind.  The fault
// word is mirrored to the host behind every such launch without waiting; every later entry point looks at the mirror
// first, gt_check_faults() synchronises and looks.
static int fault_seen(gt_engine* e) {
  const unsigned int f = e->h_fault ? (e->h_fault[0] | e->h_fault[1]) : 0u;
  if (f & 0xffu)
    return fail(GT_ERR_HIP, "persistent LSTM kernel fault %u: a workgroup timed out waiting for its peers (results of that "
                "step are invalid and its optimizer updates were skipped; gt_clear_faults() re-arms the engine, "
                "GT_OPT_LSTM_PERSISTENT=0 / GT_LSTM_STEPS=1 selects the per-step kernels)", f);
  if (f) return fail(GT_ERR_HIP, "device fault word 0x%x: results of that step are invalid", f);
  return GT_OK;
}
// After a fault: parameters, gradients and optimizer state were left untouched by every optimizer launch that saw the
// raised word (optim_step_kernel returns before its first write and counts the skipped step in pinned memory).  This
// call waits for the stream, takes the skipped steps back out of the host-side step counters, and clears the word, so
// that the engine is usable again (typically after gt_set_option(GT_OPT_LSTM_PERSISTENT, 0)).
extern "C" int gt_clear_faults(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemset(e->d_fault, 0, 64));
  for (int r = 0; r < 2; ++r) { e->net[r].step -= (long)e->h_fault[2 + r]; if (e->net[r].step < 0) e->net[r].step = 0; }
  for (int i = 0; i < 4; ++i) e->h_fault[i] = 0;
  e->g_pass_valid = false; e->leak_pending = false; e->fake_cat_valid = false;
  e->d_begin_done = e->g_begin_done = false; e->early_done = false;
  return GT_OK;
}
extern "C" int gt_check_faults(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return fault_seen(e);
}
// Co-resident workgroups per XCD: all workgroups of a launch spin on each other, so the whole grid must be resident
// at once.  The occupancy API may over-report by one block per CU (MI355X_MICROARCH.md, residency): keep that margin
// above one per CU.  The grid is laid out per XCD (seq_group_of), so the bound is per XCD as well.
static int seq_xcds(int* nxcd, int* cus_per_xcd) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  static std::map<int, int> cus;
  if (!cus.count(dev)) { hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, dev)); cus[dev] = prop.multiProcessorCount; }
  *nxcd = cus[dev] % 8 == 0 && cus[dev] >= 64 ? 8 : 1;      // MI355X: 8 XCDs x 32 CUs
  *cus_per_xcd = cus[dev] / *nxcd;
  return GT_OK;
}
template <typename K>
static int launch_seq(K kern, size_t lds, LstmSeqArgs& a, hipStream_t s, bool* launched, int block = 256) {
  CHK(ensure_dyn_lds((const void*)kern, lds));
  int per_cu = 0, nxcd = 1, cpx = 1;
  CHK(seq_xcds(&nxcd, &cpx));
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, block, lds));
  if (per_cu > 1) per_cu -= 1;
  per_cu = std::min(per_cu, 4);
  const int ngroups = a.dirs * a.nbt;
  const int rounds = cdiv(ngroups, nxcd);                   // groups that share one XCD
  if ((long)a.ncu * rounds > (long)cpx * per_cu) { *launched = false; return GT_OK; }
  a.nxcd = nxcd;
  hipLaunchKernelGGL(kern, dim3(nxcd * a.ncu * rounds), dim3(block), lds, s, a);
  LAUNCH_CHECK();
  *launched = true;
  return GT_OK;
}
// forward: loader waves + the fast gate functions (lstm_seq_kernels.hip.h: 2.15 -> 1.59 us per step in bf16, 2.54 -> 1.92
// in f32 on a cfg3 layer; tools/lstm_sched_bench keeps the round-2 variant as the A/B reference)
template <int HP, int UPC>
static int launch_fwd_seq(LstmSeqArgs& a, int bt, bool bf16, hipStream_t s, bool* launched) {
  if (bf16)
    return bt == 8 ? launch_seq(lstm_fwd_seq_kernel<HP, UPC, 8, PREC_BF16, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 8, true))
                   : launch_seq(lstm_fwd_seq_kernel<HP, UPC, 16, PREC_BF16, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 16, true));
  return bt == 8 ? launch_seq(lstm_fwd_seq_kernel<HP, UPC, 8, PREC_F32, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 8, true))
                 : launch_seq(lstm_fwd_seq_kernel<HP, UPC, 16, PREC_F32, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 16, true));
}
// backward: loader waves in both precisions; the tagged exchange where dG travels as bf16 (in f32 it doubles the exchange
// volume and loses).  cfg3 layer, us per step: bf16 2.38 -> 1.85, f32 3.13 -> 2.76 (tools/lstm_sched_bench).
template <int HP>
static int launch_bwd_seq(LstmSeqArgs& a, int bt, bool bf16, hipStream_t s, bool* launched) {
  if (bf16)
    return bt == 8 ? launch_seq(lstm_bwd_seq_kernel<HP, 8, PREC_BF16, true, true>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(8, true))
                   : launch_seq(lstm_bwd_seq_kernel<HP, 16, PREC_BF16, true, true>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(16, true));
  return bt == 8 ? launch_seq(lstm_bwd_seq_kernel<HP, 8, PREC_F32, true, false>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(8, true))
                 : launch_seq(lstm_bwd_seq_kernel<HP, 16, PREC_F32, true, false>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(16, true));
}

// Runs one layer's recurrence (forward, or backward when `backward`) as ONE persistent launch when the shape fits
// (H <= 512, grid co-resident); *launched = false leaves the work to the per-step kernels.
static int lstm_launch_seq(gt_engine* e, const Net& G, int layer, int B, int T, bool backward, const float* dout, hipStream_t s,
                           bool* launched) {
  *launched = false;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  if (!e->lstm_persistent || H > 512 || T < 2) return GT_OK;
  const int HP = H <= 256 ? 256 : 512;
  int nxcd = 1, cpx = 1;
  CHK(seq_xcds(&nxcd, &cpx));
  // batch tile: 16 sequences per group (full MFMA rows) once that already gives every XCD a group; else 8, which
  // halves the exchange volume of a group and spreads the recurrences over more XCDs (their L2s bound the exchange)
  const int bt = dirs * cdiv(B, 16) >= nxcd ? 16 : 8;
  LstmSeqArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.nbt = cdiv(B, bt);
  a.lengths = e->d_lengths();
  for (int d = 0; d < dirs; ++d) { a.Whh[d] = G.lstm[layer].d[d].Whh; a.bih[d] = G.lstm[layer].d[d].bih; a.bhh[d] = G.lstm[layer].d[d].bhh; }
  a.xproj = e->l_xproj[layer].as<float>();
  a.gates = e->l_gates[layer].as<float>();
  a.cst = e->l_cst[layer].as<float>();
  a.out = e->l_out[layer].as<float>();
  a.dout = dout;
  a.fault = e->d_fault;
  a.timeout_ticks = 200000000ULL;          // 2 s at 100 MHz: far beyond any real wait, far below the watchdog
  const int ngroups = dirs * a.nbt;
  const size_t xch_n = (size_t)ngroups * (backward ? lstm_bwd_xch_u64(HP) : lstm_fwd_xch_u64(HP)), chk_n = (size_t)ngroups * 256;
  CHK(e->l_xch.ensure((xch_n + chk_n) * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(e->l_xch.p, 0, (xch_n + chk_n) * sizeof(unsigned long long), s));   // no tag / flag of an earlier launch survives
  a.xch = e->l_xch.as<unsigned long long>();
  a.xcc_chk = a.xch + xch_n;              // [group][256]
  a.allow_xcd_local = e->lstm_xcd_local ? 1 : 0;
  if (backward) {
    a.ncu = cdiv(H, 16);
    CHK(HP == 256 ? launch_bwd_seq<256>(a, bt, e->matmul_bf16, s, launched) : launch_bwd_seq<512>(a, bt, e->matmul_bf16, s, launched));
    if (*launched) HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    return GT_OK;
  }
  // forward: 8 hidden units per workgroup (32 workgroups per group at H = 256: one XCD's worth), else 16
  int upc = e->lstm_fwd_upc;
  if (upc != 8 && upc != 16) upc = 8;
  for (; upc <= 16 && !*launched; upc *= 2) {
    a.ncu = cdiv(H, upc);
    if (upc == 8) CHK(HP == 256 ? (launch_fwd_seq<256, 8>(a, bt, e->matmul_bf16, s, launched)) : (launch_fwd_seq<512, 8>(a, bt, e->matmul_bf16, s, launched)));
    else          CHK(HP == 256 ? (launch_fwd_seq<256, 16>(a, bt, e->matmul_bf16, s, launched)) : (launch_fwd_seq<512, 16>(a, bt, e->matmul_bf16, s, launched)));
  }
  if (*launched) HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  return GT_OK;
}

static bool lstm_b16(const gt_engine* e) { return e->matmul_bf16 && (e->net[GT_ROLE_G].d.hidden_dim & 7) == 0; }
// x (N, in_dim) -> y_hat (N, out_dim); stashes X-projections / gates / cell states / layer outputs
static int lstm_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  CHK(lstm_check_lengths(e, B, T));
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  const float* in = x;
  int ld_in = G.d.in_dim;
  // GT_OPT_MATMUL_BF16: the layer inputs go through bf16 images (both orientations: the weight gradients read the
  // transposed one) and W_ih of all directions is one stacked bf16 shadow -- the X-projection of a layer is ONE product
  const bool b16 = lstm_b16(e);
  const bool want_t = G.d.grads != nullptr;
  const int Lc_ = G.d.num_hidden;
  if (b16) {
    e->l_in_b.resize(Lc_ + 1); e->lsh.resize(Lc_ + 1);
    for (int l = 0; l <= Lc_; ++l) {
      LinShadow& w = e->lsh[l];
      if (l == Lc_) {
        w.ldw = pad8(G.last.in); w.ldwt = pad8(G.last.out);
        CHK(w.w.ensure((size_t)G.last.out * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)G.last.in * w.ldwt * 2 + 64));
        CHK(cast_transpose<float>(G.last.W, G.last.in, G.last.out, G.last.in, w.w.as<__bf16>(), w.ldw, w.wt.as<__bf16>(), w.ldwt, nullptr, false, &e->colp, s));
        break;
      }
      const LstmLayerP& L = G.lstm[l];
      w.ldw = pad8(L.in); w.ldwt = pad8(dirs * 4 * H);
      CHK(w.w.ensure((size_t)dirs * 4 * H * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)L.in * w.ldwt * 2 + 64));
      for (int d = 0; d < dirs; ++d)
        CHK(cast_transpose<float>(L.d[d].Wih, L.in, 4 * H, L.in, w.w.as<__bf16>() + (size_t)d * 4 * H * w.ldw, w.ldw,
                                  w.wt.as<__bf16>() + (size_t)d * 4 * H, w.ldwt, nullptr, false, &e->colp, s));
    }
  }
  for (int l = 0; l < G.d.num_hidden; ++l) {
    const LstmLayerP& L = G.lstm[l];
    CHK(e->l_xproj[l].ensure((size_t)N * dirs * 4 * H * sizeof(float)));
    CHK(e->l_gates[l].ensure((size_t)N * dirs * 4 * H * sizeof(float)));
    CHK(e->l_cst[l].ensure((size_t)N * dirs * H * sizeof(float)));
    CHK(e->l_out[l].ensure((size_t)N * dirs * H * sizeof(float)));
    if (b16) {
      B16Img& I = e->l_in_b[l];
      CHK(I.ensure(N, L.in, want_t));
      CHK(cast_transpose<float>(in, ld_in, N, L.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
      GemmB16Args g = b16_args();
      g.A = I.r(); g.lda = I.ld; g.B = e->lsh[l].w.as<__bf16>(); g.ldb = e->lsh[l].ldw;
      g.M = (int)N; g.N = dirs * 4 * H; g.K = L.in; g.epi = B16_FWD; g.act = ACT_NONE;
      g.C = e->l_xproj[l].as<float>(); g.ldc = dirs * 4 * H;
      CHK(launch_gemm_b16(g, 1, s));
    } else {
      for (int d = 0; d < dirs; ++d)   // Xp[:, d*4H:(d+1)*4H] = X W_ih^T (biases are added in the step kernel)
        CHK(linear_forward(in, ld_in, L.d[d].Wih, L.in, nullptr, e->l_xproj[l].as<float>() + (size_t)d * 4 * H, dirs * 4 * H, N, L.in,
                           4 * H, ACT_NONE, no_drop(), s));
    }
    bool seq = false;
    CHK(lstm_launch_seq(e, G, l, B, T, false, nullptr, s, &seq));
    if (!seq) CHK(lstm_launch_steps(e, G, l, B, T, false, nullptr, s));
    in = e->l_out[l].as<float>();
    ld_in = dirs * H;
    if (G.training && G.d.dropout > 0.f && l + 1 < G.d.num_hidden) {
      // nn.LSTM(dropout=p): dropout on the outputs of every layer but the last (training only)
      CHK(e->l_outd[l].ensure((size_t)N * dirs * H * sizeof(float)));
      const DropoutSpec ds = drop_spec(e, GT_ROLE_G, 0, l, G.inj[0][l], dirs * H);
      hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(N * dirs * H, 256)), dim3(256), 0, s, in, e->l_outd[l].as<float>(), N,
                         dirs * H, ds);
      LAUNCH_CHECK();
      in = e->l_outd[l].as<float>();
    }
  }
  if (b16) {
    B16Img& I = e->l_in_b[Lc_];
    CHK(I.ensure(N, G.last.in, want_t));
    CHK(cast_transpose<float>(in, ld_in, N, G.last.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
    GemmB16Args g = b16_args();
    g.A = I.r(); g.lda = I.ld; g.B = e->lsh[Lc_].w.as<__bf16>(); g.ldb = e->lsh[Lc_].ldw;
    g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
    g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  return linear_forward(in, ld_in, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in, G.last.out,
                        G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

// gy (N, out_dim) = dL/dy_hat -> parameter gradients of hidden2out and of every LSTM layer
static int lstm_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, Do = G.d.out_dim, Lc = G.d.num_hidden;
  const bool acc = G.grads_dirty;
  CHK(e->l_dout.ensure((size_t)2 * N * dirs * H * sizeof(float)));
  CHK(e->l_hshift.ensure((size_t)N * H * sizeof(float)));
  float* dout = e->l_dout.as<float>();                       // gradient w.r.t. the current layer's output
  float* dout_other = dout + (size_t)N * dirs * H;
  const bool b16 = lstm_b16(e) && (int)e->l_in_b.size() == Lc + 1 && (int)e->lsh.size() == Lc + 1;
  if (b16) {
    // hidden2out through the bf16 images: gy -> (gy, gyT); dW = gyT . topT^T, d out_top = gy . W_lastT^T
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose<float>(gy, Do, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->l_in_b[Lc];
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, dirs * H, G.last.dW, G.last.db, acc, e->slabs, s));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * dirs * H + Do, s));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = e->lsh[Lc].wt.as<__bf16>(); g.ldb = e->lsh[Lc].ldwt;
    g.M = (int)N; g.N = dirs * H; g.K = Do; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dout; g.ldc = dirs * H;
    CHK(launch_gemm_b16(g, 1, s));
  } else {
  // hidden2out: dW = gy^T out_top, db, d out_top = gy W
  CHK(linear_backward_weight(gy, Do, e->l_out[Lc - 1].as<float>(), dirs * H, N, Do, dirs * H, G.last.dW, G.last.db, acc, e->slabs,
                             e->colp, s));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * dirs * H + Do, s));
  CHK(linear_backward_data(gy, Do, G.last.W, G.last.in, 0, dout, dirs * H, N, Do, dirs * H, ACT_NONE, nullptr, 0, no_drop(), s));
  }
  // Side stream (GT_LSTM_SIDE=1; OFF by default): a layer's weight-gradient products (dW_ih, dW_hh, the shifts and combines:
  // ~3 ms of a cfg3 step) depend on its dG only, and nothing on the way to the layer below depends on them, so they can
  // run beside the persistent recurrence of the layer below, whose workgroups leave the matrix pipes idle: the step stream
  // carries recurrence -> d(input) product -> next recurrence and joins the side stream before clip-norm + optimizer.
  // Built, correct (the at-size cfg3 parity test passes with it) and MEASURED NOT TO PAY: cfg3 fp32 25.74 ms with it vs
  // 25.68 ms without, bf16 19.04 vs 18.14 ms -- the recurrence is bound by its L2 hand-offs, and the products' operand
  // traffic through the same L2s slows every one of its 1024 steps by about what the overlap hides.
  static const bool side_on = getenv("GT_LSTM_SIDE") && getenv("GT_LSTM_SIDE")[0] == '1';
  hipStream_t ws = s;
  if (side_on) {
    if (!e->side) {
      HIPCHK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
      HIPCHK(hipEventCreateWithFlags(&e->ev_side_go, hipEventDisableTiming));
      HIPCHK(hipEventCreateWithFlags(&e->ev_side_done, hipEventDisableTiming));
    }
    ws = e->side;
  }
  Scratch& wsl = side_on ? e->slabs_side : e->slabs;
  Scratch& wcp = side_on ? e->colp_side : e->colp;
  if (b16) e->l_dg_b.resize(Lc);
  for (int l = Lc - 1; l >= 0; --l) {
    const LstmLayerP& L = G.lstm[l];
    bool seq = false;
    CHK(lstm_launch_seq(e, G, l, B, T, true, dout, s, &seq));  // dG overwrites l_xproj[l]
    if (!seq) CHK(lstm_launch_steps(e, G, l, B, T, true, dout, s));
    const float* dG = e->l_xproj[l].as<float>();
    const bool dropped_in = l > 0 && G.training && G.d.dropout > 0.f;
    if (b16) {
      B16Img& DG0 = e->l_dg_b[l];
      CHK(DG0.ensure(N, dirs * 4 * H, true));
      CHK(cast_transpose<float>(dG, dirs * 4 * H, N, dirs * 4 * H, DG0.r(), DG0.ld, DG0.t(), DG0.ldt, nullptr, false, &e->colp, s));
    }
    if (side_on) { HIPCHK(hipEventRecord(e->ev_side_go, s)); HIPCHK(hipStreamWaitEvent(ws, e->ev_side_go, 0)); }
    if (b16) {
      // dG -> bf16 image in both orientations (one pass), then every product of this layer reads bf16:
      // dW_ih_d = dGT_d . inT^T (+ db from the loader), dW_hh_d = dGT_d . hshiftT^T, d in = dG . W_ihT^T (all directions in ONE product)
      B16Img& DG = e->l_dg_b[l];
      B16Img& I = e->l_in_b[l];
      if (l > 0) {      // the step stream's part first: d(layer input), all directions in ONE product
        GemmB16Args g = b16_args();
        g.A = DG.r(); g.lda = DG.ld; g.B = e->lsh[l].wt.as<__bf16>(); g.ldb = e->lsh[l].ldwt;
        g.M = (int)N; g.N = L.in; g.K = dirs * 4 * H; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dout_other; g.ldc = L.in;
        CHK(launch_gemm_b16(g, 1, s));
        if (dropped_in) {
          const DropoutSpec ds = drop_spec(e, GT_ROLE_G, 0, l - 1, G.inj[0][l - 1], dirs * H);
          hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(N * dirs * H, 256)), dim3(256), 0, s, dout_other, dout_other, N, dirs * H, ds);
          LAUNCH_CHECK();
        }
        std::swap(dout, dout_other);
      }
      for (int d = 0; d < dirs; ++d) {
        const __bf16* dgt = DG.t() + (size_t)d * 4 * H * DG.ldt;
        CHK(weight_grad_b16(dgt, DG.ldt, I.t(), I.ldt, N, 4 * H, L.in, L.d[d].dWih, L.d[d].dbih, acc, wsl, ws));
        hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(4 * H, 64)), dim3(1024), 0, ws, L.d[d].dbih, (long)4 * H, 1, 4 * H,
                           L.d[d].dbhh, 0);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(lstm_shift_kernel, dim3(cdiv(N * H, 256)), dim3(256), 0, ws, e->l_out[l].as<float>(), dirs * H, d, H, B, T,
                           e->d_lengths(), e->l_hshift.as<float>());
        LAUNCH_CHECK();
        CHK(e->l_hs_b.ensure(N, H, true));
        CHK(cast_transpose<float>(e->l_hshift.as<float>(), H, N, H, (__bf16*)nullptr, 0, e->l_hs_b.t(), e->l_hs_b.ldt, nullptr, false, &wcp, ws));
        CHK(weight_grad_b16(dgt, DG.ldt, e->l_hs_b.t(), e->l_hs_b.ldt, N, 4 * H, H, L.d[d].dWhh, nullptr, acc, wsl, ws));
      }
      CHK(comm_grads_ready(e, GT_ROLE_G, L.d[0].dWih, (long)dirs * (4L * H * L.in + 4L * H * H + 8L * H), ws));
      CHK(comm_flush(e, GT_ROLE_G, ws));
      continue;
    }
    const float* Xl = l == 0 ? x : (dropped_in ? e->l_outd[l - 1].as<float>() : e->l_out[l - 1].as<float>());
    const int ldx = l == 0 ? G.d.in_dim : dirs * H;
    float* const dx_dst = dout_other;
    for (int d = 0; d < dirs; ++d) {
      const float* dGd = dG + (size_t)d * 4 * H;
      // dW_ih = dG_d^T X, db_ih = colsum(dG_d) (= db_hh)
      CHK(linear_backward_weight(dGd, dirs * 4 * H, Xl, ldx, N, 4 * H, L.in, L.d[d].dWih, L.d[d].dbih, acc, wsl, wcp, ws));
      // bias_ih and bias_hh always receive the same gradient: keep them equal by copy
      hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(4 * H, 64)), dim3(1024), 0, ws, L.d[d].dbih, (long)4 * H, 1, 4 * H,
                         L.d[d].dbhh, 0);
      LAUNCH_CHECK();
      // dW_hh = dG_d^T H_shift (h that entered each frame)
      hipLaunchKernelGGL(lstm_shift_kernel, dim3(cdiv(N * H, 256)), dim3(256), 0, ws, e->l_out[l].as<float>(), dirs * H, d, H, B, T,
                         e->d_lengths(), e->l_hshift.as<float>());
      LAUNCH_CHECK();
      CHK(linear_backward_weight(dGd, dirs * 4 * H, e->l_hshift.as<float>(), H, N, 4 * H, H, L.d[d].dWhh, nullptr, acc, wsl,
                                 wcp, ws));
    }
    // this layer's parameters (both directions: W_ih, W_hh, b_ih, b_hh each) are one contiguous bucket
    CHK(comm_grads_ready(e, GT_ROLE_G, L.d[0].dWih, (long)dirs * (4L * H * L.in + 4L * H * H + 8L * H), ws));
    CHK(comm_flush(e, GT_ROLE_G, ws));              // with hidden2out above it: under the recurrence of the layer below
    (void)dx_dst;
    if (l > 0) {   // gradient w.r.t. the layer below's output: sum over directions of dG_d W_ih_d
      for (int d = 0; d < dirs; ++d) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dG + (size_t)d * 4 * H; g.lda = dirs * 4 * H; g.B = L.d[d].Wih; g.ldb = L.in; g.C = dout_other; g.ldc = L.in;
        g.M = (int)N; g.N = L.in; g.K = 4 * H; g.act = ACT_NONE; g.accumulate = d > 0 ? 1 : 0; g.drop = no_drop();
        CHK(launch_gemm(GEMM_NN, g, 1, s));
      }
      if (dropped_in) {   // through the inter-layer dropout of layer l-1 (same Philox site as the forward)
        const DropoutSpec ds = drop_spec(e, GT_ROLE_G, 0, l - 1, G.inj[0][l - 1], dirs * H);
        hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(N * dirs * H, 256)), dim3(256), 0, s, dout_other, dout_other, N, dirs * H, ds);
        LAUNCH_CHECK();
      }
      std::swap(dout, dout_other);
    }
  }
  if (side_on) { HIPCHK(hipEventRecord(e->ev_side_done, ws)); HIPCHK(hipStreamWaitEvent(s, e->ev_side_done, 0)); }
  return GT_OK;
}


// ------------------------------------------------------------------------------------------
// recurrent generator (GT_ARCH_SRU)
// ------------------------------------------------------------------------------------------
static void sru_keys(gt_engine* e, int layer, int which, uint32_t* k0, uint32_t* k1) {
  // data parallel: the masks are per (sequence, column); the kernels count sequences globally (SruArgs::seq_mul / seq_add), so a
  // world-k run draws the whole minibatch's masks of a world-1 run -- no rank in the key
  const uint64_t site = e->step_counter * 64ULL + 40 + (uint64_t)(layer * 2 + which);
  *k0 = (uint32_t)(e->seed ^ (site * 0x9E3779B97F4A7C15ULL));
  *k1 = (uint32_t)((e->seed >> 32) ^ (site >> 7) ^ 0x5A5A5A5Au) + (uint32_t)site;
}
static uint32_t drop_thresh(float p) {
  const double th = (double)p * 4294967296.0;
  return th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
}

static SruArgs sru_args(gt_engine* e, const Net& G, int l, int B, int T, const float* in, int ld_in) {
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs;
  const SruLayerP& L = G.sru[l];
  SruArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.k = L.k; a.act = G.d.use_relu ? SRU_RELU : SRU_TANH;
  a.U = e->s_u[l].as<float>(); a.ldu = ncols * L.k;
  a.x = in; a.ldx = ld_in;
  a.bias = L.b;
  a.h = e->s_h[l].as<float>(); a.c = e->s_c[l].as<float>();
  a.seq_mul = e->dp_world; a.seq_add = e->dp_rank;
  if (G.training && G.d.dropout > 0.f && l + 1 < G.d.num_hidden) {   // the last layer has dropout 0 (SRU.__init__)
    a.use_mask = 1; a.keep_scale = 1.f / (1.f - G.d.dropout); a.thresh = drop_thresh(G.d.dropout);
    sru_keys(e, l, 1, &a.key0, &a.key1);
    a.mask_buf = G.inj[0][2 * l + 1];                                // gt_set_dropout_mask(G, 0, 2*l + 1): [B][ncols]
  }
  return a;
}

// the scans with loader waves (sru_kernels.hip.h); GT_SRU_LW=0 selects the one-wave kernels (A/B reference, bit-identical results)
// (read at every launch: the A/B test flips it between two steps of one process)
static bool sru_loader_waves() { const char* v = getenv("GT_SRU_LW"); return !(v && v[0] == '0'); }
static bool sru_b16(const gt_engine* e) { return e->matmul_bf16 && (e->net[GT_ROLE_G].d.hidden_dim & 7) == 0; }
static int sru_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs;
  const float* in = x;
  int ld_in = G.d.in_dim;
  // GT_OPT_MATMUL_BF16: the (dropped) layer inputs go through bf16 images in both orientations, W through bf16 shadows in
  // both orientations: U = xin . WT^T, dW = xinT . dUT^T, d in = dU . W^T are all the k-contiguous bf16 product
  const bool b16 = sru_b16(e);
  const bool want_t = G.d.grads != nullptr;
  const int Lc_ = G.d.num_hidden;
  if (b16) {
    e->s_in_b.resize(Lc_ + 1); e->ssh.resize(Lc_ + 1);
    for (int l = 0; l <= Lc_; ++l) {
      LinShadow& w = e->ssh[l];
      const float* W = l < Lc_ ? G.sru[l].W : G.last.W;
      const int rows = l < Lc_ ? G.sru[l].in : G.last.out, cols = l < Lc_ ? ncols * G.sru[l].k : G.last.in;
      w.ldw = pad8(cols); w.ldwt = pad8(rows);
      CHK(w.w.ensure((size_t)rows * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)cols * w.ldwt * 2 + 64));
      CHK(cast_transpose<float>(W, cols, rows, cols, w.w.as<__bf16>(), w.ldw, w.wt.as<__bf16>(), w.ldwt, nullptr, false, &e->colp, s));
    }
  }
  for (int l = 0; l < G.d.num_hidden; ++l) {
    const SruLayerP& L = G.sru[l];
    CHK(e->s_u[l].ensure((size_t)N * ncols * L.k * sizeof(float)));
    CHK(e->s_h[l].ensure((size_t)N * ncols * sizeof(float)));
    CHK(e->s_c[l].ensure((size_t)N * ncols * sizeof(float)));
    const float* xin = in;
    int ld_xin = ld_in;
    const bool rdrop = G.training && G.d.rnn_dropout > 0.f;
    if (rdrop) {      // variational input dropout, mask shared over time: the multipliers of this step, [B][n_in]
      CHK(e->s_xmask[l].ensure((size_t)B * L.in * sizeof(float)));
      uint32_t k0, k1;
      sru_keys(e, l, 0, &k0, &k1);
      hipLaunchKernelGGL(sru_input_mask_kernel, dim3(cdiv((long)B * L.in, 256)), dim3(256), 0, s, e->s_xmask[l].as<float>(), B, L.in,
                         1.f / (1.f - G.d.rnn_dropout), drop_thresh(G.d.rnn_dropout), k0, k1,
                         (const float*)G.inj[0][2 * l], e->dp_world, e->dp_rank);     // gt_set_dropout_mask(G, 0, 2*l): [B][n_in]
      LAUNCH_CHECK();
    }
    if (rdrop && !b16) {      // float32 products read a dropped float32 copy
      CHK(e->s_xdrop[l].ensure((size_t)N * L.in * sizeof(float)));
      hipLaunchKernelGGL(sru_input_dropout_kernel, dim3(cdiv(N * L.in, 256)), dim3(256), 0, s, in, ld_in, e->s_xdrop[l].as<float>(),
                         L.in, B, T, L.in, (const float*)e->s_xmask[l].as<float>());
      LAUNCH_CHECK();
      xin = e->s_xdrop[l].as<float>();
      ld_xin = L.in;
    }
    if (b16) {
      B16Img& I = e->s_in_b[l];
      CHK(I.ensure(N, L.in, want_t));
      if (rdrop) {            // bf16 products: dropout rides in the cast, the dropped input exists as bf16 images only
        const SeqDropSrc src{in, ld_in, e->s_xmask[l].as<float>(), T, L.in};
        hipLaunchKernelGGL(seqdrop_cast_transpose_kernel, dim3(cdiv(N, 64), cdiv(L.in, 64)), dim3(256), 0, s, src, N, L.in, I.r(), I.ld,
                           want_t ? I.t() : (__bf16*)nullptr, I.ldt);
        LAUNCH_CHECK();
      } else {
        CHK(cast_transpose<float>(xin, ld_xin, N, L.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
      }
      GemmB16Args g = b16_args();
      g.A = I.r(); g.lda = I.ld; g.B = e->ssh[l].wt.as<__bf16>(); g.ldb = e->ssh[l].ldwt;     // WT [ncols*k][n_in]: k = n_in contiguous
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.epi = B16_FWD; g.act = ACT_NONE; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      CHK(launch_gemm_b16(g, 1, s));
    } else if ((L.in & 3) == 0 && N >= 4096) {
      // U = xin W with W (n_in, ncols*k): the k-contiguous (NT) product runs at 146 TFLOP/s on these shapes, the n-contiguous (NN)
      // one at 114 (profiles/r03_sru_fp32_summary.md: 1.41 vs 1.80 ms per layer) -- multiply by a transposed copy of W, re-made
      // from the caller's parameter buffer before every pass (12 MB, ~10 us)
      CHK(e->s_wt[l].ensure((size_t)ncols * L.k * L.in * sizeof(float)));
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(cdiv(ncols * L.k, 32), cdiv(L.in, 32)), dim3(256), 0, s, L.W, L.in, ncols * L.k, ncols * L.k,
                         e->s_wt[l].as<float>(), L.in);
      LAUNCH_CHECK();
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = xin; g.lda = ld_xin; g.B = e->s_wt[l].as<float>(); g.ldb = L.in; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.act = ACT_NONE; g.drop = no_drop();
      CHK(launch_gemm(GEMM_NT, g, 1, s));
    } else {  // U = xin W   (W is (n_in, ncols*k): n-contiguous rows -> NN orientation)
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = xin; g.lda = ld_xin; g.B = L.W; g.ldb = ncols * L.k; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.act = ACT_NONE; g.drop = no_drop();
      CHK(launch_gemm(GEMM_NN, g, 1, s));
    }
    SruArgs a = sru_args(e, G, l, B, T, in, ld_in);
    if (sru_loader_waves()) {
      CHK(ensure_dyn_lds((const void*)sru_fwd_lw_kernel, sru_fwd_lw_lds()));
      hipLaunchKernelGGL(sru_fwd_lw_kernel, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_fwd_lw_lds(), s, a);
    } else {
      hipLaunchKernelGGL(sru_fwd_kernel, dim3(cdiv((long)B * ncols, SRU_THREADS)), dim3(SRU_THREADS), 0, s, a);
    }
    LAUNCH_CHECK();
    in = e->s_h[l].as<float>();
    ld_in = ncols;
  }
  if (b16) {
    B16Img& I = e->s_in_b[Lc_];
    CHK(I.ensure(N, G.last.in, want_t));
    CHK(cast_transpose<float>(in, ld_in, N, G.last.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
    GemmB16Args g = b16_args();
    g.A = I.r(); g.lda = I.ld; g.B = e->ssh[Lc_].w.as<__bf16>(); g.ldb = e->ssh[Lc_].ldw;    // hidden2out.weight (out, ncols): k = ncols contiguous
    g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
    g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  return linear_forward(in, ld_in, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in, G.last.out,
                        G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

static int sru_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs, Do = G.d.out_dim, Lc = G.d.num_hidden;
  const bool acc = G.grads_dirty;
  int kmax = 3, inmax = ncols;
  for (auto& L : G.sru) { kmax = std::max(kmax, L.k); inmax = std::max(inmax, L.in); }
  CHK(e->l_dout.ensure((size_t)2 * N * std::max(ncols, inmax) * sizeof(float)));
  CHK(e->s_du.ensure((size_t)N * ncols * kmax * sizeof(float)));
  CHK(e->s_dx.ensure((size_t)2 * N * ncols * sizeof(float)));     // highway gradients of two consecutive layers (read by the layer underneath)
  CHK(e->s_dbias.ensure((size_t)B * 2 * ncols * sizeof(float)));
  float* dh = e->l_dout.as<float>();
  float* dh_other = dh + (size_t)N * std::max(ncols, inmax);
  const bool b16 = sru_b16(e) && (int)e->s_in_b.size() == Lc + 1 && (int)e->ssh.size() == Lc + 1;
  if (b16) {
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose<float>(gy, Do, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->s_in_b[Lc];
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, ncols, G.last.dW, G.last.db, acc, e->slabs, s));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * ncols + Do, s));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = e->ssh[Lc].wt.as<__bf16>(); g.ldb = e->ssh[Lc].ldwt;   // hidden2out.weight^T [ncols][Do]
    g.M = (int)N; g.N = ncols; g.K = Do; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dh; g.ldc = ncols;
    CHK(launch_gemm_b16(g, 1, s));
  } else {
  CHK(linear_backward_weight(gy, Do, e->s_h[Lc - 1].as<float>(), ncols, N, Do, ncols, G.last.dW, G.last.db, acc, e->slabs, e->colp, s));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * ncols + Do, s));
  CHK(linear_backward_data(gy, Do, G.last.W, G.last.in, 0, dh, ncols, N, Do, ncols, ACT_NONE, nullptr, 0, no_drop(), s));
  }
  for (int l = Lc - 1; l >= 0; --l) {
    const SruLayerP& L = G.sru[l];
    const float* in = l == 0 ? x : e->s_h[l - 1].as<float>();
    const int ld_in = l == 0 ? G.d.in_dim : ncols;
    const bool rdrop = G.training && G.d.rnn_dropout > 0.f;
    SruArgs a = sru_args(e, G, l, B, T, in, ld_in);
    a.dh = dh; a.dU = e->s_du.as<float>();
    // k == 3: the highway gradient goes straight to the layer input.  Without input dropout it is
    // written into the next dh buffer and the GEMM below accumulates onto it.
    auto dx_of = [&](int layer) { return e->s_dx.as<float>() + (size_t)(layer & 1) * N * ncols; };
    float* dx_res = L.k == 3 ? (rdrop ? dx_of(l) : dh_other) : nullptr;
    a.dx = dx_res; a.lddx = ncols;
    if (rdrop && l + 1 < Lc) {      // dh is the raw dU.W^T of the layer above: its input dropout and highway gradient are applied by the scan
      if (G.sru[l + 1].in != ncols || !e->s_xmask[l + 1].p)
        return fail(GT_ERR_STATE, "SRU backward: layer %d's input-dropout table is missing or not %d wide", l + 1, ncols);
      a.up_mul = e->s_xmask[l + 1].as<float>();
      a.up_add = G.sru[l + 1].k == 3 ? dx_of(l + 1) : nullptr;
      a.ld_up_add = ncols;
    }
    a.dbias_part = e->s_dbias.as<float>();
    // bf16 storage with loader waves, whole blocks of 8 frames, whole workgroups inside one sequence and one direction: dU leaves
    // the scan as the bf16 images the two products read (no float32 dU, no cast pass)
    const bool du_b16 = b16 && sru_loader_waves() && T % 8 == 0 && H % 64 == 0;
    if (du_b16) {
      B16Img& DU = e->s_du_b;
      CHK(DU.ensure(N, ncols * L.k, true));
      a.dU = nullptr; a.dU_b = DU.r(); a.ld_dub = DU.ld; a.dU_bt = DU.t(); a.ld_dubt = DU.ldt;
      CHK(ensure_dyn_lds((const void*)sru_bwd_lw_kernel<true>, sru_bwd_lw_lds()));
      hipLaunchKernelGGL(sru_bwd_lw_kernel<true>, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_bwd_lw_lds(), s, a);
    } else if (sru_loader_waves()) {
      CHK(ensure_dyn_lds((const void*)sru_bwd_lw_kernel<false>, sru_bwd_lw_lds()));
      hipLaunchKernelGGL(sru_bwd_lw_kernel<false>, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_bwd_lw_lds(), s, a);
    } else {
      hipLaunchKernelGGL(sru_bwd_kernel, dim3(cdiv((long)B * ncols, SRU_THREADS)), dim3(SRU_THREADS), 0, s, a);
    }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(2 * ncols, 64)), dim3(1024), 0, s, e->s_dbias.as<float>(), (long)2 * ncols, B,
                       2 * ncols, L.db, acc ? 1 : 0);
    LAUNCH_CHECK();
    const float* xin = rdrop ? e->s_xdrop[l].as<float>() : in;
    const int ld_xin = rdrop ? L.in : ld_in;
    if (b16) {
      // dU -> bf16 image in both orientations (one pass); dW = xinT . dUT^T over the frames, d in = dU . W^T
      B16Img& DU = e->s_du_b;
      if (!du_b16) {
        CHK(DU.ensure(N, ncols * L.k, true));
        CHK(cast_transpose<float>(e->s_du.as<float>(), ncols * L.k, N, ncols * L.k, DU.r(), DU.ld, DU.t(), DU.ldt, nullptr, false, &e->colp, s));
      }
      B16Img& I = e->s_in_b[l];
      CHK(weight_grad_b16(I.t(), I.ldt, DU.t(), DU.ldt, N, L.in, ncols * L.k, L.dW, nullptr, acc, e->slabs, s));
    } else {
    // dW = xin^T dU   (TN: A = xin is m-contiguous over n_in, B = dU)
    CHK(linear_backward_weight(xin, ld_xin, e->s_du.as<float>(), ncols * L.k, N, L.in, ncols * L.k, L.dW, nullptr, acc, e->slabs,
                               e->colp, s));
    }
    CHK(comm_grads_ready(e, GT_ROLE_G, L.dW, (long)L.in * ncols * L.k + 2L * ncols, s));
    if (l > 0) CHK(comm_flush(e, GT_ROLE_G, s));
    if (l > 0) {
      if (b16) {
        GemmB16Args g = b16_args();
        g.A = e->s_du_b.r(); g.lda = e->s_du_b.ld; g.B = e->ssh[l].w.as<__bf16>(); g.ldb = e->ssh[l].ldw;    // W [n_in][ncols*k]: k contiguous
        g.M = (int)N; g.N = L.in; g.K = ncols * L.k; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dh_other; g.ldc = L.in;
        g.accumulate = (L.k == 3 && !rdrop) ? 1 : 0;
        CHK(launch_gemm_b16(g, 1, s));
      } else {
      // d in = (dU W^T) (.) mask_in + highway term     (NT: B[n = i][k = c] = W[i*ldw + c])
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->s_du.as<float>(); g.lda = ncols * L.k; g.B = L.W; g.ldb = ncols * L.k; g.C = dh_other; g.ldc = L.in;
      g.M = (int)N; g.N = L.in; g.K = ncols * L.k; g.act = ACT_NONE; g.drop = no_drop();
      g.accumulate = (L.k == 3 && !rdrop) ? 1 : 0;
      CHK(launch_gemm(GEMM_NT, g, 1, s));
      }
      std::swap(dh, dh_other);         // (with input dropout: finished by the scan of layer l - 1, SruArgs::up_mul / up_add)
    }
    // l == 0: no gradient with respect to the network input is produced on this path (nothing upstream of the generator
    // takes one: x is data, train.py:542).  NOTE for anything that wants to read `dh` between layers: with rnn_dropout it
    // is the RAW dU.W^T -- the input-dropout mask and the k = 3 highway term are applied by the next scan's loads.
  }
  return GT_OK;
}

static int generator_forward(gt_engine* e, const float* x, const float* R, int B, int T, float* y_hat, float* y_hat_static,
                             bool stash, hipStream_t s, std::vector<DropoutSpec>& specs) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int pass0[1] = {0};
  const float* gsrc = y_hat;            // what MLPG is applied to
  if (G.d.arch == GT_ARCH_LSTM) {
    CHK(lstm_forward(e, x, B, T, y_hat, s));
  } else if (G.d.arch == GT_ARCH_IN2OUT_RNN) {
    // G(x) = hidden2out(LSTM(x)) stays internal; the model returns its INPUT as y_hat (models.py:118)
    CHK(e->i2o_gout.ensure((size_t)N * G.d.out_dim * sizeof(float)));
    CHK(lstm_forward(e, x, B, T, e->i2o_gout.as<float>(), s));
    hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * G.d.in_dim, 256)), dim3(256), 0, s, x, G.d.in_dim, 0, (const int*)nullptr,
                       y_hat, G.d.in_dim, 0, (int)N, G.d.in_dim);
    LAUNCH_CHECK();
    gsrc = e->i2o_gout.as<float>();
  } else if (G.d.arch == GT_ARCH_SRU) {
    CHK(sru_forward(e, x, B, T, y_hat, s));
  } else {
    if (use_b16(e, GT_ROLE_G)) {
      const bool want_t = stash && G.d.grads != nullptr;
      CHK(e->xin_b.ensure(N, G.d.in_dim, want_t));
      CHK(cast_transpose<float>(x, G.d.in_dim, N, G.d.in_dim, e->xin_b.r(), e->xin_b.ld, want_t ? e->xin_b.t() : (__bf16*)nullptr, e->xin_b.ldt,
                                nullptr, false, &e->colp, s));
      CHK(refresh_shadows(e, GT_ROLE_G, true, s));
      CHK(stack_forward_b16(e, GT_ROLE_G, e->xin_b.r(), e->xin_b.ld, N, e->g_actb, pass0, 1, N, specs, want_t, s));
      const LinShadow& ws = e->wsh[GT_ROLE_G][G.hidden.size()];
      GemmB16Args g = b16_args();
      g.A = e->g_actb.back().r(); g.lda = e->g_actb.back().ld; g.B = ws.w.as<__bf16>(); g.ldb = ws.ldw;
      g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
      g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
      CHK(launch_gemm_b16(g, 1, s));
    } else {
      CHK(stack_forward(e, GT_ROLE_G, x, G.d.in_dim, N, e->g_act, pass0, 1, N, specs, s));
      const Lin& Lh = G.hidden.back();
      CHK(linear_forward(e->g_act.back().as<float>(), Lh.out, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in,
                         G.last.out, G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s));
    }
  }
  if (is_i2o(G.d.arch)) {
    if (!R) return fail(GT_ERR_INVALID, "In2OutHighwayNet needs the MLPG matrix R (models.py:54)");
    const int sd = G.d.static_dim;
    CHK(ensure_band(e, R, T, s));
    CHK(e->tx.ensure((size_t)N * sd * sizeof(float)));
    CHK(e->gx.ensure((size_t)N * sd * sizeof(float)));
    // T(x) = sigmoid(T x_static), x_static = x[:, :, :static_dim]   (models.py:57-60)
    CHK(linear_forward(x, G.d.in_dim, G.gate.W, sd, G.gate.b, e->tx.as<float>(), sd, N, sd, sd, ACT_SIGMOID, no_drop(), s));
    CHK(mlpg_forward(e, gsrc, G.d.out_dim, e->d_scol_i2o, e->d_sstride_i2o, sd, e->gx.as<float>(), sd, B, T, s));
    if (stash) e->g_used_mlpg = true;
    hipLaunchKernelGGL(highway_forward_kernel, dim3(cdiv(N * sd, 256)), dim3(256), 0, s, x, G.d.in_dim, e->tx.as<float>(), sd,
                       e->gx.as<float>(), sd, y_hat_static, sd, N, sd);
    LAUNCH_CHECK();
  } else {
    if (G.d.out_dim != e->Dout_cfg)
      return fail(GT_ERR_DIM, "You probably have specified wrong dimention params.");  // multistream.py:93-94
    if (R) {
      CHK(ensure_band(e, R, T, s));
      CHK(mlpg_forward(e, y_hat, G.d.out_dim, e->d_scol, e->d_sstride, e->Ds, y_hat_static, e->Ds, B, T, s));
      if (stash) e->g_used_mlpg = true;
    } else {
      if (e->Ds != G.d.out_dim) return fail(GT_ERR_INVALID, "R is None but the stream config has dynamic features");
      if (stash) e->g_used_mlpg = false;
      // R is None: num_windows = 1, every stream passes through (multistream.py:88-89,119-120)
      hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * G.d.out_dim, 256)), dim3(256), 0, s, y_hat, G.d.out_dim, 0,
                         (const int*)nullptr, y_hat_static, G.d.out_dim, 0, (int)N, G.d.out_dim);
      LAUNCH_CHECK();
    }
  }
  return GT_OK;
}

extern "C" int gt_apply_generator(gt_engine* e, const float* x, const float* R, int B, int T, float* y_hat,
                                  float* y_hat_static, void* stream) {
  CHK(check_common(e, B, T));
  Net& G = e->net[GT_ROLE_G];
  if (!G.bound) return fail(GT_ERR_STATE, "generator not bound");
  if (!x || !y_hat || !y_hat_static) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(fault_seen(e));
  e->step_counter++;
  e->B = B; e->T = T; e->N = (long)B * T;
  e->g_pass_valid = false;
  e->fake_cat_valid = false; e->dcat_b_ok = false;
  e->tv_mask = nullptr; e->tv_inflight = false;             // a new batch: the mask contents may have changed
  CHK(generator_forward(e, x, R, B, T, y_hat, y_hat_static, true, s, e->g_specs));
  e->last_x = x; e->last_yhat = y_hat; e->last_yhs = y_hat_static;
  e->g_pass_valid = true;
  return GT_OK;
}

// width of the conditioning input x fed to D (train.py:254-256); derived from the bound D when not configured
static int cond_dim(gt_engine* e) {
  if (!e->cfg.discriminator_linguistic_condition) return 0;
  if (e->cfg.cond_dim > 0) return e->cfg.cond_dim;
  return e->net[GT_ROLE_D].bound ? e->net[GT_ROLE_D].d.in_dim - e->Da : 0;
}
static int d_in_dim(gt_engine* e) { return e->Da + cond_dim(e); }

// rows [row0, row0+N) of dcat <- [x | feats[:, adv_cols]]
static int build_cat(gt_engine* e, const float* x, const float* feats, int ld_feats, long row0, long N, int ldc, hipStream_t s) {
  float* dst = e->dcat.as<float>() + row0 * ldc;
  int off = 0;
  if (e->cfg.discriminator_linguistic_condition) {
    if (!x) return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
    const int cd = cond_dim(e);
    if (cd <= 0) return fail(GT_ERR_DIM, "discriminator in_dim too small for linguistic conditioning");
    hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * cd, 256)), dim3(256), 0, s, x, cd, 0, (const int*)nullptr, dst, ldc, 0,
                       (int)N, cd);
    LAUNCH_CHECK();
    off = cd;
  }
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * e->Da, 256)), dim3(256), 0, s, feats, ld_feats, 0, e->d_adv_cols, dst, ldc,
                     off, (int)N, e->Da);
  LAUNCH_CHECK();
  return GT_OK;
}

// H: the top hidden activation, float32 [n_rows][K] or (h_ld > 0) its bf16 image with row pitch h_ld
static int run_head(gt_engine* e, int mode, const void* H, int K, long n_rows, long n_real, const float* mask, long n_mask,
                    float eps, bool want_grad, float* dH, const DropoutSpec& spec, bool want_w, hipStream_t s,
                    StepResults* early_res = nullptr, int h_ld = 0, B16Img* dz_img = nullptr, bool dz_t = false) {
  Net& D = e->net[GT_ROLE_D];
  const int nblk = (int)std::min<long>(1024, (n_rows + 31) / 32);
  CHK(e->headp.ensure((size_t)nblk * sizeof(HeadPartials)));
  CHK(e->headw.ensure((size_t)nblk * K * sizeof(float)));
  CHK(e->dout.ensure((size_t)n_rows * sizeof(float)));
  const size_t lds = (size_t)4 * K * sizeof(float);
#define GT_HEAD_LAUNCH(KP_)                                                                                              \
  if (h_ld > 0)                                                                                                          \
    hipLaunchKernelGGL((d_head_kernel<KP_, __bf16, true>), dim3(nblk), dim3(256), lds, s, (const __bf16*)H, h_ld, K, D.last.W, D.last.b, mask, (int)n_mask, \
                       (int)n_real, (int)n_rows, mode, eps, e->dout.as<float>(), dz_img ? (float*)nullptr : dH, K, want_grad ? 1 : 0, spec, 1, e->sc(), \
                       e->headp.as<HeadPartials>(), e->headw.as<float>(), dz_img ? dz_img->r() : (__bf16*)nullptr, dz_img ? dz_img->ld : 0,  \
                       (dz_img && dz_t) ? dz_img->t() : (__bf16*)nullptr, dz_img ? dz_img->ldt : 0L);                    \
  else                                                                                                                   \
    hipLaunchKernelGGL((d_head_kernel<KP_, float, false>), dim3(nblk), dim3(256), lds, s, (const float*)H, K, K, D.last.W, D.last.b, mask, (int)n_mask,  \
                       (int)n_real, (int)n_rows, mode, eps, e->dout.as<float>(), dH, K, want_grad ? 1 : 0, spec, 1, e->sc(), \
                       e->headp.as<HeadPartials>(), e->headw.as<float>())
  if (K <= 128) { GT_HEAD_LAUNCH(2); }
  else if (K <= 256) { GT_HEAD_LAUNCH(4); }
  else if (K <= 512) { GT_HEAD_LAUNCH(8); }
  else if (K <= 1024) { GT_HEAD_LAUNCH(16); }
  else return fail(GT_ERR_INVALID, "discriminator hidden_dim > 1024 is not supported by the fused head kernel");
#undef GT_HEAD_LAUNCH
  LAUNCH_CHECK();
  const bool w = want_grad && want_w;
  hipLaunchKernelGGL(d_head_finalize_kernel, dim3(cdiv(K, 64)), dim3(1024), 0, s, e->headp.as<HeadPartials>(), e->headw.as<float>(),
                     nblk, K, mode, e->sc(), w ? D.last.dW : (float*)nullptr, w ? D.last.db : (float*)nullptr, D.grads_dirty ? 1 : 0,
                     early_res);
  LAUNCH_CHECK();
  return GT_OK;
}

static int optimizer_step(gt_engine* e, int role, double* norm2_out, hipStream_t s) {
  Net& n = e->net[role];
  if (!n.has_opt) return fail(GT_ERR_STATE, "phase == \"train\" but no optimizer is bound for role %d", role);
  const long np = n.d.n_params;
  const int nblk = (int)std::min<long>(512, cdiv(np, RED_THREADS * 4));
  CHK(e->partial.ensure(4096 * sizeof(double)));
  double* part = e->partial.as<double>() + 2048;
  CHK(slab_defer_flush(e->sdefer[role], s));            // the fused step's recorded weight-gradient combines, one launch
  e->sdefer[role].active = false;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, n.d.grads, np, part);
  LAUNCH_CHECK();
  n.step += 1;
  OptimSpec o;
  o.kind = n.od.kind; o.lr = n.od.lr; o.weight_decay = n.od.weight_decay; o.eps = n.od.eps; o.lr_decay = n.od.lr_decay;
  o.beta1 = n.od.beta1; o.beta2 = n.od.beta2; o.step = n.step; o.max_norm = n.od.max_grad_norm;
  const int grid = (int)std::min<long>(1024, cdiv(np, RED_THREADS));
  hipLaunchKernelGGL(optim_step_kernel, dim3(grid), dim3(RED_THREADS), 0, s, n.d.params, n.d.grads, n.od.state0, n.od.state1, np,
                     part, nblk, norm2_out, o, (const unsigned int*)e->d_fault, e->h_fault_dev,
                     e->h_fault_dev ? e->h_fault_dev + 1 + role : (unsigned int*)nullptr);
  LAUNCH_CHECK();
  return GT_OK;
}

static int fetch_results(gt_engine* e, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(e->h_res, e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return GT_OK;
}
static int post_deferred_results(gt_engine* e, int role, hipStream_t s) {
  if (!e->h_def[role]) HIPCHK(hipHostMalloc((void**)&e->h_def[role], sizeof(StepResults)));
  if (!e->ev_def[role]) HIPCHK(hipEventCreateWithFlags(&e->ev_def[role], hipEventDisableTiming));
  HIPCHK(hipMemcpyAsync(e->h_def[role], e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(e->ev_def[role], s));
  e->def_pending[role] = true;
  return GT_OK;
}
static StepResults* early_res_target(gt_engine* e) { return e->h_res_dev ? e->h_res_dev : e->res(); }
static int post_early_results(gt_engine* e, hipStream_t s) {
  if (!e->ev_res) HIPCHK(hipEventCreateWithFlags(&e->ev_res, hipEventDisableTiming));
  if (!e->h_res_dev) HIPCHK(hipMemcpyAsync(e->h_res, e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(e->ev_res, s));
  e->early_done = true;
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// update_discriminator
// ------------------------------------------------------------------------------------------
extern "C" int gt_update_discriminator_begin(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                                             const float* mask, int B, int T, int train, float eps, void* stream) {
  CHK(check_common(e, B, T));
  Net& D = e->net[GT_ROLE_D];
  if (!D.bound) return fail(GT_ERR_STATE, "discriminator not bound");
  if (!y_static || !y_hat_static || !mask) return fail(GT_ERR_INVALID, "null tensor");
  if (D.d.in_dim != d_in_dim(e))
    return fail(GT_ERR_DIM, "discriminator in_dim %d != adversarial input width %d (train.py:760-768)", D.d.in_dim, d_in_dim(e));
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const int K0 = D.d.in_dim, ldc = (K0 + 3) & ~3;
  const bool tr = train != 0;
  if (comm_on(e) && tr && D.grads_dirty)
    return fail(GT_ERR_STATE, "data-parallel step: optimizer_d.zero_grad() must precede update_discriminator (the gradient buckets are summed over the ranks in place)");
  { SlabDefer& sd = e->sdefer[GT_ROLE_D]; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = e->early && !comm_on(e) && tr && D.has_opt; }
  CHK(ensure_tv_begin(e, mask, N, s));        // data parallel: the global count travels under the D forward pass
  const int passes[2] = {0, 1};
  // the [x | adv] image of both halves: real rows, then generated rows
  const bool b16 = use_b16(e, GT_ROLE_D);
  if (b16) {
    // bf16 storage: the image is written ONCE, as bf16, in both orientations (no float32 image at all)
    if (e->cfg.discriminator_linguistic_condition && (!x || cond_dim(e) <= 0))
      return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
    CHK(e->dcat_b.ensure(2 * N, K0, tr));
    CatSrc src;
    src.x = x; src.cd = cond_dim(e); src.fa = y_static; src.fb = y_hat_static; src.ldf = e->Ds; src.idx = e->d_adv_cols; src.N = N; src.row_off = 0;
    hipLaunchKernelGGL(cat_cast_transpose_kernel, dim3(cdiv(2 * N, 64), cdiv(K0, 64)), dim3(256), 0, s, src, 2 * N, K0, e->dcat_b.r(), e->dcat_b.ld,
                       tr ? e->dcat_b.t() : (__bf16*)nullptr, e->dcat_b.ldt);
    LAUNCH_CHECK();
    e->dcat_b_ok = true;
    e->fake_cat_valid = false;                 // the float32 image was not built
  } else {
  CHK(e->dcat.ensure((size_t)2 * N * ldc * sizeof(float)));
  if (e->cfg.discriminator_linguistic_condition && x && cond_dim(e) > 0) {
    hipLaunchKernelGGL(build_cat2_kernel, dim3(cdiv(N * K0, 256)), dim3(256), 0, s, x, cond_dim(e), y_static, y_hat_static, e->Ds,
                       e->d_adv_cols, e->Da, e->dcat.as<float>(), ldc, N);
    LAUNCH_CHECK();
  } else {
    CHK(build_cat(e, x, y_static, e->Ds, 0, N, ldc, s));
    CHK(build_cat(e, x, y_hat_static, e->Ds, N, N, ldc, s));
  }
  e->fake_cat_valid = true; e->fake_cat_x = x; e->fake_cat_yhs = y_hat_static;
  }
  e->dcat_b_x = x; e->dcat_b_yhs = y_hat_static;
  if (b16) {
    CHK(refresh_shadows(e, GT_ROLE_D, false, s));
    CHK(stack_forward_b16(e, GT_ROLE_D, e->dcat_b.r(), e->dcat_b.ld, 2 * N, e->d_actb, passes, 2, N, e->d_specs, tr, s));
  } else {
    CHK(stack_forward(e, GT_ROLE_D, e->dcat.as<float>(), ldc, 2 * N, e->d_act, passes, 2, N, e->d_specs, s));
  }
  const int H = D.d.hidden_dim;
  if (tr && !D.d.grads) return fail(GT_ERR_STATE, "phase == \"train\" but the discriminator was bound without grads");
  CHK(e->dzA.ensure((size_t)2 * N * std::max(H, 1) * sizeof(float)));
  CHK(e->dzB.ensure((size_t)2 * N * std::max(H, 1) * sizeof(float)));
  // fused call: losses and counts are final after the head (the gradient norm is not: reported as 0), so the head's
  // reduction kernel also writes the result struct and the scalars start their way to the host right behind it
  const bool plain_early = e->early && !comm_on(e), comm_early = e->early && comm_on(e);
  CHK(ensure_tv(e, mask, N, s));
  if (b16 && tr) CHK(e->dz_b[0].ensure(2 * N, H, true));
  CHK(run_head(e, HEAD_D_STEP, b16 ? (const void*)e->d_actb.back().r() : (const void*)e->d_act.back().as<float>(), H, 2 * N, N, mask, N, eps, tr,
               e->dzA.as<float>(), e->d_specs.back(), true, s, plain_early ? early_res_target(e) : nullptr, b16 ? e->d_actb.back().ld : 0,
               (b16 && tr) ? &e->dz_b[0] : nullptr, true));
  e->early_done = false;
  if (plain_early) CHK(post_early_results(e, s));
  if (comm_early) CHK(comm_early_results(e, GT_ROLE_D, &e->sc()->s_real, 4, 0.f, 0.f, 0.f, s));
  if (tr) {
    CHK(comm_grads_ready(e, GT_ROLE_D, D.last.dW, (long)D.last.in * D.last.out + D.last.out, s));
    // keep dloss_d/dy_hat_static only when y_hat_static is the tensor apply_generator produced
    // (the autograd graph in the reference, train.py:265) and a generator with grads exists
    Net& G = e->net[GT_ROLE_G];
    const bool want_leak = G.bound && G.d.grads && e->g_pass_valid && y_hat_static == e->last_yhs && e->N == N;
    if (want_leak && e->leak_pending)
      return fail(GT_ERR_STATE, "update_discriminator called twice without optimizer_g.zero_grad() (train.py:538)");
    float* leak = nullptr;
    if (want_leak) { CHK(e->leak.ensure((size_t)N * e->Da * sizeof(float))); leak = e->leak.as<float>(); }
    const int col0 = cond_dim(e);
    if (b16) {   // the head wrote its seed gradient as the top dZ image, both orientations
      CHK(stack_backward_b16(e, GT_ROLE_D, e->dcat_b.t(), e->dcat_b.ldt, 2 * N, e->d_actb, e->d_specs, 0, true, leak, e->Da, col0, e->Da, N, N, s));
    } else {
      CHK(stack_backward(e, GT_ROLE_D, e->dcat.as<float>(), ldc, 2 * N, e->d_act, e->d_specs, e->dzA.as<float>(),
                         e->dzB.as<float>(), true, leak, e->Da, col0, e->Da, N, N, s));
    }
    D.grads_dirty = true;
    if (want_leak) e->leak_pending = true;
  }
  // data parallel: the rest of D's gradient + the four loss / count sums, then the step stream waits for the communicator
  CHK(comm_finish_step(e, GT_ROLE_D, tr, &e->sc()->s_real, comm_early ? 0 : 4, s));
  e->d_begin_done = true;
  return GT_OK;
}

static void fill_d_result(const StepResults* h, gt_d_result* out) {
  out->loss_d = h->loss_d; out->loss_fake_d = h->loss_fake_d; out->loss_real_d = h->loss_real_d;
  out->real_correct_count = h->real_correct; out->fake_correct_count = h->fake_correct;
  out->grad_norm = h->gnorm_d;
}
static void fill_g_result(const StepResults* h, gt_g_result* out) {
  out->loss_mse = h->loss_mse; out->loss_mge = h->loss_mge; out->loss_adv = h->loss_adv;
  out->loss_g = h->loss_g; out->grad_norm = h->gnorm_g;
}

extern "C" int gt_update_discriminator_end(gt_engine* e, int train, gt_d_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null argument");
  if (!e->d_begin_done) return fail(GT_ERR_STATE, "gt_update_discriminator_end without _begin");
  hipStream_t s = (hipStream_t)stream;
  e->d_begin_done = false;
  if (!out) {   // deferred: enqueue everything, synchronise nothing; gt_update_discriminator_result collects
    if (e->early_done) return fail(GT_ERR_STATE, "deferred results are a split-phase feature");
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), train ? 0 : 1);
    LAUNCH_CHECK();
    return post_deferred_results(e, GT_ROLE_D, s);
  }
  if (e->early_done) {
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    HIPCHK(hipEventSynchronize(e->ev_res));       // only the scalar copy; backward + step stay queued
    e->early_done = false;
  } else {
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), train ? 0 : 1);
    LAUNCH_CHECK();
    CHK(fetch_results(e, s));
  }
  fill_d_result(e->h_res, out);
  return GT_OK;
}
extern "C" int gt_update_discriminator_result(gt_engine* e, gt_d_result* out) {
  if (!e || !out) return fail(GT_ERR_INVALID, "null argument");
  if (!e->def_pending[GT_ROLE_D]) return fail(GT_ERR_STATE, "no deferred discriminator result pending");
  HIPCHK(hipEventSynchronize(e->ev_def[GT_ROLE_D]));
  e->def_pending[GT_ROLE_D] = false;
  fill_d_result(e->h_def[GT_ROLE_D], out);
  return GT_OK;
}

extern "C" int gt_update_discriminator(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                                       const float* mask, int B, int T, int train, float eps, gt_d_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->early = true;
  int r = gt_update_discriminator_begin(e, x, y_static, y_hat_static, mask, B, T, train, eps, stream);
  e->early = false;
  if (r != GT_OK) { e->early_done = false; return r; }
  return gt_update_discriminator_end(e, train, out, stream);
}

// ------------------------------------------------------------------------------------------
// update_generator
// ------------------------------------------------------------------------------------------
// e->partial: [0, 1024) MGE partials, [1024, 2048) MSE partials, [2048, ...) the optimizer's squared-norm partials.
// deferred_blocks != null: the per-block partial sums stay in the MSE region and *deferred_blocks says how many -- the
// caller folds their reduction into a later launch (finalize_g_kernel) instead of paying a launch for it here.
static int sum_sqerr(gt_engine* e, const float* a, int lda, const float* b, int ldb, const float* mask, long rows, int D,
                     double* out, float* g, int ldg, float gscale, hipStream_t s, int* deferred_blocks = nullptr) {
  const int nblk = (int)std::min<long>(1024, cdiv(rows * D, RED_THREADS * 4));
  CHK(e->partial.ensure(4096 * sizeof(double)));
  double* part = e->partial.as<double>() + 1024;
  hipLaunchKernelGGL(masked_sqerr_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, a, lda, b, ldb, mask, rows, D, part, g, ldg, gscale, e->sc());
  LAUNCH_CHECK();
  if (deferred_blocks) { *deferred_blocks = nblk; return GT_OK; }
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, part, nblk, out);
  LAUNCH_CHECK();
  return GT_OK;
}

// backward of G from the gradient at y_hat_static (gs) [+ masked-MSE term at y_hat]
static int generator_backward(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* mask,
                              float mse_w, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = e->N;
  const int B = e->B, T = e->T, Do = G.d.out_dim;
  // dloss/dy_hat is an engine buffer: for the MLP stacks its row pitch is rounded up to 4 floats, so that the last layer's
  // two backward products (K = out_dim = 187 for the acoustic model) take the 16-byte loader and the 64x64 tiles; the
  // pad column is never read as data (K tail / row clamp of the GEMM loader)
  const bool mlp_body = !(has_lstm_body(G.d.arch) || G.d.arch == GT_ARCH_SRU);
  const int ldgy = mlp_body && (is_i2o(G.d.arch) || e->g_used_mlpg) ? (Do + 3) & ~3 : Do;
  CHK(e->gy.ensure((size_t)N * ldgy * sizeof(float)));
  float* gy = e->gy.as<float>();
  const float* gs = e->gs.as<float>();
  if (is_i2o(G.d.arch)) {
    const int sd = G.d.static_dim;
    if (G.d.arch == GT_ARCH_IN2OUT_RNN) mse_w = 0.f;   // y_hat is the input x there: the MSE term has no path into G
    CHK(e->dgx.ensure((size_t)N * sd * sizeof(float)));
    CHK(e->dtz.ensure((size_t)N * sd * sizeof(float)));
    hipLaunchKernelGGL(highway_backward_kernel, dim3(cdiv(N * sd, 256)), dim3(256), 0, s, gs, sd, e->tx.as<float>(), sd,
                       e->gx.as<float>(), sd, e->dgx.as<float>(), sd, e->dtz.as<float>(), sd, N, sd);
    LAUNCH_CHECK();
    CHK(linear_backward_weight(e->dtz.as<float>(), sd, x, G.d.in_dim, N, sd, sd, G.gate.dW, G.gate.db, G.grads_dirty, e->slabs,
                               e->colp, s, &e->sdefer[GT_ROLE_G]));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.gate.dW, (long)sd * sd + sd, s));
    CHK(mlpg_backward(e, e->dgx.as<float>(), sd, e->d_scol_i2o, e->d_sstride_i2o, sd, gy, ldgy, B, T, mse_w, y_hat, y, Do, mask, s));
  } else if (e->g_used_mlpg) {
    CHK(mlpg_backward(e, gs, e->Ds, e->d_scol, e->d_sstride, e->Ds, gy, ldgy, B, T, mse_w, y_hat, y, Do, mask, s));
  } else {
    // no parameter generation: y_hat_static == y_hat, the gradient passes straight through,
    // plus the masked-MSE gradient (which also yields loss_mse's sum)
    if (mse_w != 0.f) CHK(sum_sqerr(e, y_hat, Do, y, Do, mask, N, Do, &e->sc()->s_mse, gy, Do, mse_w, s));
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(N * Do, 256)), dim3(256), 0, s, gs, 0L, 1, N * Do, gy, mse_w != 0.f ? 1 : 0);
    LAUNCH_CHECK();
  }
  if (has_lstm_body(G.d.arch) || G.d.arch == GT_ARCH_SRU) {
    CHK(has_lstm_body(G.d.arch) ? lstm_backward(e, x, gy, B, T, s) : sru_backward(e, x, gy, B, T, s));
    G.grads_dirty = true;
    return GT_OK;
  }
  if (use_b16(e, GT_ROLE_G)) {
    // dloss/dy_hat -> bf16 image (both orientations); last_linear: dW = gyT . H_topT^T, dZ_top = (gy . W_lastT^T) (.) f'(H_top)
    const int H = G.d.hidden_dim;
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose<float>(gy, ldgy, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->g_actb.back();
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, G.last.in, G.last.dW, G.last.db, G.grads_dirty, e->slabs, s,
                        &e->sdefer[GT_ROLE_G]));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * G.last.in + Do, s));
    const LinShadow& ws = e->wsh[GT_ROLE_G][G.hidden.size()];
    CHK(e->dz_b[0].ensure(N, H, true));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = ws.wt.as<__bf16>(); g.ldb = ws.ldwt; g.M = (int)N; g.N = H; g.K = Do;
    g.epi = B16_BWD_DATA; g.act = ACT_LEAKY_DROPOUT; g.H = top.r(); g.ldh = top.ld; g.drop = e->g_specs.back();
    g.Cb = e->dz_b[0].r(); g.ldcb = e->dz_b[0].ld; g.CbT = e->dz_b[0].t(); g.ldcbt = (int)e->dz_b[0].ldt;
    CHK(launch_gemm_b16(g, 1, s));
    CHK(stack_backward_b16(e, GT_ROLE_G, e->xin_b.t(), e->xin_b.ldt, N, e->g_actb, e->g_specs, 0, true, nullptr, 0, 0, 0, 0, 0, s));
    G.grads_dirty = true;
    return GT_OK;
  }
  // last_linear: dW = gy^T H_top, db ; dZ_top = (gy W_last) (.) f'(H_top)
  const Lin& Lt = G.hidden.back();
  const int H = G.d.hidden_dim;
  CHK(e->dzA.ensure((size_t)2 * N * H * sizeof(float)));
  CHK(e->dzB.ensure((size_t)2 * N * H * sizeof(float)));
  const GemmArgs nn_last = backward_data_args(gy, ldgy, G.last.W, G.last.in, 0, e->dzA.as<float>(), H, N, Do, H, ACT_LEAKY_DROPOUT,
                                              e->g_act.back().as<float>(), H, e->g_specs.back());
  bool rode = false;
  CHK(linear_backward_weight(gy, ldgy, e->g_act.back().as<float>(), Lt.out, N, Do, G.last.in, G.last.dW, G.last.db, G.grads_dirty,
                             e->slabs, e->colp, s, &e->sdefer[GT_ROLE_G], &nn_last, &rode));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * G.last.in + Do, s));
  if (!rode) CHK(launch_gemm(GEMM_NN, nn_last, 1, s));
  // The first layer's weight gradient reads G's input as its frame operand.  When the discriminator's input image of
  // this step holds the very same x (linguistic conditioning on the generator's own input, no noise channels), the x
  // columns of its rows are a bit-exact copy with a 16-byte row pitch: use it, and the product takes the 16-byte loader.
  const float* xin = x;
  int ldxin = G.d.in_dim;
  if (e->fake_cat_valid && e->fake_cat_x == x && e->cfg.discriminator_linguistic_condition && cond_dim(e) == G.d.in_dim &&
      e->dcat.p && (G.d.in_dim & 3)) {
    ldxin = (d_in_dim(e) + 3) & ~3;
    xin = e->dcat.as<float>() + N * ldxin;      // the generated half: the one that is valid whenever fake_cat_valid is
  }
  CHK(stack_backward(e, GT_ROLE_G, xin, ldxin, N, e->g_act, e->g_specs, e->dzA.as<float>(), e->dzB.as<float>(), true,
                     nullptr, 0, 0, 0, 0, 0, s));
  G.grads_dirty = true;
  return GT_OK;
}

extern "C" int gt_update_generator_begin(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* y_static,
                                         const float* y_hat_static, float adv_w, const float* mask, int B, int T, int train,
                                         float mse_w, float mge_w, float eps, void* stream) {
  CHK(check_common(e, B, T));
  Net& G = e->net[GT_ROLE_G];
  Net& D = e->net[GT_ROLE_D];
  if (!G.bound) return fail(GT_ERR_STATE, "generator not bound");
  if (!y || !y_hat || !y_static || !y_hat_static || !mask) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const bool tr = train != 0;
  if (tr) {
    if (!e->g_pass_valid || y_hat != e->last_yhat || y_hat_static != e->last_yhs || N != e->N)
      return fail(GT_ERR_STATE, "update_generator(phase=\"train\") needs the y_hat / y_hat_static returned by the last apply_generator");
    if (!G.d.grads) return fail(GT_ERR_STATE, "phase == \"train\" but the generator was bound without grads");
    if (G.d.last_sigmoid) return fail(GT_ERR_INVALID, "training a generator with last_sigmoid=True is not supported");
  }
  const int Do = G.d.out_dim;
  const int Ds = is_i2o(G.d.arch) ? G.d.static_dim : e->Ds;
  if (comm_on(e) && tr && G.grads_dirty)
    return fail(GT_ERR_STATE, "data-parallel step: optimizer_g.zero_grad() must precede update_generator");
  { SlabDefer& sd = e->sdefer[GT_ROLE_G]; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = e->early && !comm_on(e) && tr && G.has_opt; }
  CHK(ensure_tv(e, mask, N, s));
  // loss_mse (always reported, train.py:294); its gradient is fused into the MLPG^T kernel
  const bool direct = !is_i2o(G.d.arch) && !e->g_used_mlpg;
  // (the fused single-GPU call reduces the MSE partials inside its finalisation launch: see early_now below)
  const bool early_fold = e->early && !comm_on(e) && !(tr && direct && mse_w != 0.f);
  int mse_blocks = 0;
  if (!(tr && direct && mse_w != 0.f))
    CHK(sum_sqerr(e, y_hat, Do, y, Do, mask, N, Do, &e->sc()->s_mse, nullptr, 0, 0.f, s, early_fold ? &mse_blocks : nullptr));
  // adversarial term with the CURRENT (already updated) D weights and a fresh dropout mask (train.py:297-308)
  e->g_has_adv = adv_w > 0.f;
  float* gadv = nullptr;
  if (adv_w > 0.f) {
    if (!D.bound) return fail(GT_ERR_STATE, "adv_w > 0 but no discriminator bound");
    if (D.d.in_dim != d_in_dim(e)) return fail(GT_ERR_DIM, "discriminator in_dim mismatch");
    const int K0 = D.d.in_dim, ldc = (K0 + 3) & ~3;
    const int passes[1] = {2};
    const bool b16 = use_b16(e, GT_ROLE_D);
    const float* cat = nullptr;
    if (!b16) {
      CHK(e->dcat.ensure((size_t)2 * N * ldc * sizeof(float)));
      if (!(e->fake_cat_valid && e->fake_cat_x == x && e->fake_cat_yhs == y_hat_static)) {
        CHK(build_cat(e, x, y_hat_static, Ds, N, N, ldc, s));
        e->fake_cat_valid = true; e->fake_cat_x = x; e->fake_cat_yhs = y_hat_static;
      }
      cat = e->dcat.as<float>() + N * ldc;
    }
    if (b16) {   // the generated half of the bf16 image: rows N .. 2N, kept from the D step of the same batch or built here
      CHK(e->dcat_b.ensure(2 * N, K0, false));
      if (!(e->dcat_b_ok && e->dcat_b_x == x && e->dcat_b_yhs == y_hat_static)) {
        if (e->cfg.discriminator_linguistic_condition && (!x || cond_dim(e) <= 0))
          return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
        CatSrc src;
        src.x = x; src.cd = cond_dim(e); src.fa = y_hat_static; src.fb = y_hat_static; src.ldf = Ds; src.idx = e->d_adv_cols; src.N = N; src.row_off = N;
        hipLaunchKernelGGL(cat_cast_transpose_kernel, dim3(cdiv(N, 64), cdiv(K0, 64)), dim3(256), 0, s, src, N, K0,
                           e->dcat_b.r() + N * e->dcat_b.ld, e->dcat_b.ld, (__bf16*)nullptr, 0L);
        LAUNCH_CHECK();
        e->dcat_b_ok = true; e->dcat_b_x = x; e->dcat_b_yhs = y_hat_static;
      }
      CHK(refresh_shadows(e, GT_ROLE_D, false, s));        // D has just been stepped (train.py:276 before :307)
      CHK(stack_forward_b16(e, GT_ROLE_D, e->dcat_b.r() + N * e->dcat_b.ld, e->dcat_b.ld, N, e->d_actb, passes, 1, N, e->d_specs, false, s));
    } else {
      CHK(stack_forward(e, GT_ROLE_D, cat, ldc, N, e->d_act, passes, 1, N, e->d_specs, s));
    }
    const int H = D.d.hidden_dim;
    CHK(e->dzA.ensure((size_t)2 * N * H * sizeof(float)));
    CHK(e->dzB.ensure((size_t)2 * N * H * sizeof(float)));
    if (b16 && tr) CHK(e->dz_b[0].ensure(N, H, false));
    CHK(run_head(e, HEAD_G_ADV, b16 ? (const void*)e->d_actb.back().r() : (const void*)e->d_act.back().as<float>(), H, N, N, mask, N, eps, tr,
                 e->dzA.as<float>(), e->d_specs.back(), false, s, nullptr, b16 ? e->d_actb.back().ld : 0, (b16 && tr) ? &e->dz_b[0] : nullptr, false));
    if (tr) {
      CHK(e->gadv.ensure((size_t)N * e->Da * sizeof(float)));
      gadv = e->gadv.as<float>();
      const int col0 = cond_dim(e);
      if (b16) {
        CHK(stack_backward_b16(e, GT_ROLE_D, nullptr, 0, N, e->d_actb, e->d_specs, 0, false, gadv, e->Da, col0, e->Da, 0, N, s));
      } else {
        CHK(stack_backward(e, GT_ROLE_D, cat, ldc, N, e->d_act, e->d_specs, e->dzA.as<float>(), e->dzB.as<float>(), false, gadv,
                           e->Da, col0, e->Da, 0, N, s));
      }
    }
  }
  // MGE loss + gradient assembly at y_hat_static
  const bool early_ok = e->early && !(tr && direct && mse_w != 0.f);
  const bool early_now = early_ok && !comm_on(e), comm_early = early_ok && comm_on(e) && comm_early_g();
  int mge_blocks = 0;
  {
    const int nblk = (int)std::min<long>(1024, cdiv(N * Ds, RED_THREADS * 4));
    CHK(e->partial.ensure(4096 * sizeof(double)));
    float* gs = nullptr;
    if (tr) { CHK(e->gs.ensure((size_t)N * Ds * sizeof(float))); gs = e->gs.as<float>(); }
    const float* leak = (tr && e->leak_pending) ? e->leak.as<float>() : nullptr;
    hipLaunchKernelGGL(static_grad_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, y_hat_static, Ds, y_static, Ds, mask, N, Ds, mge_w,
                       e->d_adv_inv, leak, e->Da, gadv, e->Da, adv_w, gs, Ds, e->partial.as<double>(), e->sc());
    LAUNCH_CHECK();
    mge_blocks = nblk;
    if (!early_now) {   // the split-phase (data-parallel) caller all-reduces the sum itself: it must exist now
      hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, e->partial.as<double>(), nblk, &e->sc()->s_mge);
      LAUNCH_CHECK();
    }
  }
  e->early_done = false;
  if (early_now) {   // all four losses are final here; the MGE partials are reduced inside the finalisation launch
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(256), 0, s, e->sc(), early_res_target(e), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0, 1,
                       (const double*)e->partial.as<double>(), mge_blocks,
                       mse_blocks ? (const double*)(e->partial.as<double>() + 1024) : (const double*)nullptr, mse_blocks);
    LAUNCH_CHECK();
    CHK(post_early_results(e, s));
  }
  if (comm_early) CHK(comm_early_results(e, GT_ROLE_G, &e->sc()->s_adv, 3, adv_w, mse_w, mge_w, s));
  if (tr) {
    CHK(generator_backward(e, e->last_x, y, y_hat, mask, mse_w, s));  // G's own input (cat(x, z), train.py:542)
    e->leak_pending = false;
  }
  CHK(comm_finish_step(e, GT_ROLE_G, tr, &e->sc()->s_adv, comm_early ? 0 : 3, s));
  e->g_begin_done = true;
  return GT_OK;
}

extern "C" int gt_update_generator_end(gt_engine* e, int train, float adv_w, float mse_w, float mge_w, gt_g_result* out,
                                       void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null argument");
  if (!e->g_begin_done) return fail(GT_ERR_STATE, "gt_update_generator_end without _begin");
  hipStream_t s = (hipStream_t)stream;
  e->g_begin_done = false;
  if (!out) {
    if (e->early_done) return fail(GT_ERR_STATE, "deferred results are a split-phase feature");
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0,
                       train ? 0 : 1, (const double*)nullptr, 0, (const double*)nullptr, 0);
    LAUNCH_CHECK();
    return post_deferred_results(e, GT_ROLE_G, s);
  }
  if (e->early_done) {
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    HIPCHK(hipEventSynchronize(e->ev_res));
    e->early_done = false;
  } else {
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0,
                       train ? 0 : 1, (const double*)nullptr, 0, (const double*)nullptr, 0);
    LAUNCH_CHECK();
    CHK(fetch_results(e, s));
  }
  fill_g_result(e->h_res, out);
  return fault_seen(e);
}
extern "C" int gt_update_generator_result(gt_engine* e, gt_g_result* out) {
  if (!e || !out) return fail(GT_ERR_INVALID, "null argument");
  if (!e->def_pending[GT_ROLE_G]) return fail(GT_ERR_STATE, "no deferred generator result pending");
  HIPCHK(hipEventSynchronize(e->ev_def[GT_ROLE_G]));
  e->def_pending[GT_ROLE_G] = false;
  fill_g_result(e->h_def[GT_ROLE_G], out);
  return GT_OK;
}

extern "C" int gt_update_generator(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* y_static,
                                   const float* y_hat_static, float adv_w, const float* mask, int B, int T, int train,
                                   float mse_w, float mge_w, float eps, gt_g_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->early = true;
  int r = gt_update_generator_begin(e, x, y, y_hat, y_static, y_hat_static, adv_w, mask, B, T, train, mse_w, mge_w, eps, stream);
  e->early = false;
  if (r != GT_OK) { e->early_done = false; return r; }
  return gt_update_generator_end(e, train, adv_w, mse_w, mge_w, out, stream);
}

extern "C" int gt_flush_generator_grads(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  Net& G = e->net[GT_ROLE_G];
  if (!G.bound || !G.d.grads || !e->g_pass_valid) return fail(GT_ERR_STATE, "no generator pass to back-propagate");
  hipStream_t s = (hipStream_t)stream;
  tl_gemm_prec = e->matmul_bf16 ? PREC_BF16 : PREC_F32;      // this entry point launches GEMMs without check_common
  CHK(fault_seen(e));
  { SlabDefer& sd = e->sdefer[GT_ROLE_G]; sd.active = false; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; }   // combines run in place here
  const long N = e->N;
  const int Ds = is_i2o(G.d.arch) ? G.d.static_dim : e->Ds;
  CHK(e->gs.ensure((size_t)N * Ds * sizeof(float)));
  HIPCHK(hipMemsetAsync(e->gs.p, 0, (size_t)N * Ds * sizeof(float), s));
  if (e->leak_pending) {
    // scatter leak[:, j] -> gs[:, adv_cols[j]]  (gather with swapped roles: one column at a time is fine here)
    std::vector<int>& cols = e->h_adv_cols;
    for (int j = 0; j < e->Da; ++j) {
      hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, e->leak.as<float>(), e->Da, j, (const int*)nullptr,
                         e->gs.as<float>(), Ds, cols[j], (int)N, 1);
    }
    LAUNCH_CHECK();
  }
  CHK(generator_backward(e, e->last_x, e->last_yhat, e->last_yhat, (const float*)nullptr, 0.f, s));
  e->leak_pending = false;
  HIPCHK(hipStreamSynchronize(s));
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// plain forward
// ------------------------------------------------------------------------------------------
extern "C" int gt_model_forward(gt_engine* e, int role, const float* x, const float* R, int B, int T, float* out, float* out2,
                                void* stream) {
  CHK(check_common(e, B, T));
  if (role < 0 || role > 1 || !e->net[role].bound) return fail(GT_ERR_STATE, "model not bound");
  if (!x || !out) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  Net& n = e->net[role];
  const long N = (long)B * T;
  e->step_counter++;
  std::vector<DropoutSpec> specs;
  if (role == GT_ROLE_G && is_i2o(n.d.arch)) {
    if (!out2) return fail(GT_ERR_INVALID, "In2OutHighwayNet forward returns two tensors");
    e->g_pass_valid = false;
    return generator_forward(e, x, R, B, T, out, out2, false, s, specs);
  }
  if (n.d.arch == GT_ARCH_LSTM || n.d.arch == GT_ARCH_SRU) {
    if (role != GT_ROLE_G) return fail(GT_ERR_INVALID, "recurrent networks are supported in the generator slot only");
    e->g_pass_valid = false;
    return n.d.arch == GT_ARCH_LSTM ? lstm_forward(e, x, B, T, out, s) : sru_forward(e, x, B, T, out, s);
  }
  const int pass0[1] = {0};
  auto& acts = role == GT_ROLE_G ? e->g_act : e->d_act;
  if (role == GT_ROLE_G) e->g_pass_valid = false;
  if (use_b16(e, role)) {
    auto& actb = role == GT_ROLE_G ? e->g_actb : e->d_actb;
    CHK(e->fwd_b.ensure(N, n.d.in_dim, false));
    CHK(cast_transpose<float>(x, n.d.in_dim, N, n.d.in_dim, e->fwd_b.r(), e->fwd_b.ld, (__bf16*)nullptr, 0, nullptr, false, &e->colp, s));
    CHK(refresh_shadows(e, role, true, s));
    CHK(stack_forward_b16(e, role, e->fwd_b.r(), e->fwd_b.ld, N, actb, pass0, 1, N, specs, false, s));
    const LinShadow& ws = e->wsh[role][n.hidden.size()];
    GemmB16Args g = b16_args();
    g.A = actb.back().r(); g.lda = actb.back().ld; g.B = ws.w.as<__bf16>(); g.ldb = ws.ldw;
    g.M = (int)N; g.N = n.last.out; g.K = n.last.in; g.bias = n.last.b; g.epi = B16_FWD;
    g.act = n.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = out; g.ldc = n.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  CHK(stack_forward(e, role, x, n.d.in_dim, N, acts, pass0, 1, N, specs, s));
  return linear_forward(acts.back().as<float>(), n.hidden.back().out, n.last.W, n.last.in, n.last.b, out, n.d.out_dim, N, n.last.in,
                        n.last.out, n.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

// ------------------------------------------------------------------------------------------
// stand-alone operators
// ------------------------------------------------------------------------------------------
extern "C" int gt_op_sequence_mask(const int64_t* lengths, int B, int T, float* mask, void* stream) {
  if (!lengths || !mask || B < 1 || T < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipLaunchKernelGGL(sequence_mask_kernel, dim3(cdiv((long)B * T, 256)), dim3(256), 0, (hipStream_t)stream, (const long*)lengths, B, T, mask);
  LAUNCH_CHECK();
  return GT_OK;
}

extern "C" int gt_op_masked_mse(const float* input, const float* target, const float* mask, int B, int T, int D, float* loss_out,
                                float* grad_input, void* stream) {
  if (!input || !target) return fail(GT_ERR_INVALID, "null tensor");
  if (!mask) return fail(GT_ERR_INVALID, "Should provide either lengths or mask");  // seqloss.py:33-34
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  static thread_local Scratch tls_ws;     // grow-only, no per-call hipMalloc/hipFree (both synchronise the device)
  CHK(tls_ws.ensure(1024 + 1024 * sizeof(double)));
  void* ws = tls_ws.p;
  StepScalars* sc = (StepScalars*)ws;
  double* part = (double*)((char*)ws + 1024);
  hipLaunchKernelGGL(mask_sum_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, -1.f, (const double*)nullptr, sc);
  const int nblk = (int)std::min<long>(1000, cdiv(N * D, RED_THREADS * 4));
  hipLaunchKernelGGL(masked_sqerr_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, input, D, target, D, mask, N, D, part, grad_input, D,
                     1.f, sc);
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, part, nblk, &sc->s_mse);
  StepScalars h;
  hipError_t err = hipMemcpyAsync(&h, sc, sizeof(h), hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);
  if (err != hipSuccess) return fail(GT_ERR_HIP, "masked_mse: %s", hipGetErrorString(err));
  if (loss_out) *loss_out = (float)h.s_mse / h.tv;
  return GT_OK;
}

extern "C" int gt_compute_distortions(const float* y_static, const float* y_hat_static, int Ds, const void* stat_mean,
                                      const void* stat_std, int stats_f64, const int32_t* col_stat_host,
                                      const int32_t* col_role_host, int vuv_col, const int64_t* lengths_host, int B, int T,
                                      gt_distortion_sums* out, void* stream) {
  if (!y_static || !y_hat_static || !stat_mean || !stat_std || !col_stat_host || !col_role_host || !out)
    return fail(GT_ERR_INVALID, "null argument");
  if (Ds < 1 || B < 1 || T < 1 || vuv_col >= Ds) return fail(GT_ERR_DIM, "bad sizes: Ds=%d B=%d T=%d vuv_col=%d", Ds, B, T, vuv_col);
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const int nblk = (int)std::min<long>(1024, cdiv(N, 4));
  std::vector<int> host(2 * Ds + B);
  for (int c = 0; c < Ds; ++c) {
    if (col_stat_host[c] < 0) return fail(GT_ERR_INVALID, "negative statistics index");
    host[c] = col_stat_host[c];
    host[Ds + c] = col_role_host[c];
  }
  for (int b = 0; b < B; ++b) {
    const int64_t n = lengths_host ? lengths_host[b] : T;
    if (n < 0 || n > T) return fail(GT_ERR_INVALID, "length %lld outside [0, T=%d]", (long long)n, T);
    host[2 * Ds + b] = (int)n;
  }
  // grow-only workspace shared by all calls of this thread: the function runs once per training step
  // (train.py:588-595) -- a hipMalloc/hipFree pair per call would synchronise the device every step
  static thread_local Scratch tls_ws;
  const size_t off_part = ((host.size() * sizeof(int) + 255) / 256) * 256;
  const size_t off_out = off_part + (size_t)nblk * DIST_NSUM * sizeof(double);
  CHK(tls_ws.ensure(off_out + DIST_NSUM * sizeof(double)));
  void* ws = tls_ws.p;
  int* d_int = (int*)ws;
  double* part = (double*)((char*)ws + off_part);
  double* d_out = (double*)((char*)ws + off_out);
  hipError_t err = hipMemcpyAsync(d_int, host.data(), host.size() * sizeof(int), hipMemcpyHostToDevice, s);
  if (err == hipSuccess) {
    if (stats_f64)
      hipLaunchKernelGGL(distortion_kernel<double>, dim3(nblk), dim3(256), 0, s, y_static, y_hat_static, Ds, (const double*)stat_mean,
                         (const double*)stat_std, d_int, d_int + Ds, vuv_col, d_int + 2 * Ds, B, T, part);
    else
      hipLaunchKernelGGL(distortion_kernel<float>, dim3(nblk), dim3(256), 0, s, y_static, y_hat_static, Ds, (const float*)stat_mean,
                         (const float*)stat_std, d_int, d_int + Ds, vuv_col, d_int + 2 * Ds, B, T, part);
    hipLaunchKernelGGL(distortion_finalize_kernel, dim3(1), dim3(64 * DIST_NSUM), 0, s, part, nblk, d_out);
    err = hipGetLastError();
  }
  double h[DIST_NSUM];
  if (err == hipSuccess) err = hipMemcpyAsync(h, d_out, sizeof(h), hipMemcpyDeviceToHost, s);
  if (err == hipSuccess) err = hipStreamSynchronize(s);   // also keeps `host` alive until the H2D is done
  if (err != hipSuccess) return fail(GT_ERR_HIP, "compute_distortions: %s", hipGetErrorString(err));
  out->s_mcd = h[0]; out->s_bap = h[1]; out->s_f0 = h[2]; out->n_voiced = h[3];
  out->n_vuv_err = h[4]; out->s_mse = h[5]; out->n_frames = h[6];
  return GT_OK;
}

extern "C" int gt_op_gather_cols(const float* in, int ld_in, const int32_t* idx, int n_idx, float* out, int ld_out,
                                 int out_col_offset, int64_t rows, void* stream) {
  if (!in || !out || n_idx < 0 || rows < 0) return fail(GT_ERR_INVALID, "bad argument");
  if (rows == 0 || n_idx == 0) return GT_OK;
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(rows * n_idx, 256)), dim3(256), 0, (hipStream_t)stream, in, ld_in, 0, idx, out,
                     ld_out, out_col_offset, (int)rows, n_idx);
  LAUNCH_CHECK();
  return GT_OK;
}

extern "C" int gt_op_mlpg_forward(gt_engine* e, const float* y, const float* R, int B, int T, float* y_static, void* stream) {
  CHK(check_common(e, B, T));
  if (!y || !R || !y_static) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(ensure_band(e, R, T, s));
  return mlpg_forward(e, y, e->Dout_cfg, e->d_scol, e->d_sstride, e->Ds, y_static, e->Ds, B, T, s);
}
extern "C" int gt_op_mlpg_backward(gt_engine* e, const float* g_static, const float* R, int B, int T, float* g_y, void* stream) {
  CHK(check_common(e, B, T));
  if (!g_static || !R || !g_y) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(ensure_band(e, R, T, s));
  return mlpg_backward(e, g_static, e->Ds, e->d_scol, e->d_sstride, e->Ds, g_y, e->Dout_cfg, B, T, 0.f, nullptr, nullptr, 0, nullptr, s);
}

static DropoutSpec buffer_spec(const float* keep_mask, float p, int ld) {
  DropoutSpec d = no_drop();
  if (keep_mask && p > 0.f) { d.mode = DROP_BUFFER; d.mask = keep_mask; d.ld_mask = ld; d.p = p; d.scale = 1.f / (1.f - p); }
  return d;
}

extern "C" int gt_op_linear_forward(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy, int64_t rows,
                                    int in_dim, int out_dim, int act, const float* keep_mask, float p, void* stream) {
  if (!X || !W || !Y || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  if (act < 0 || act > 2) return fail(GT_ERR_INVALID, "unknown activation");
  tl_gemm_prec = PREC_F32;
  return linear_forward(X, ldx, W, in_dim, bias, Y, ldy, rows, in_dim, out_dim, act, buffer_spec(keep_mask, p, out_dim), (hipStream_t)stream);
}

extern "C" int gt_op_linear_backward(const float* dY, int lddy, const float* X, int ldx, const float* W, int64_t rows, int in_dim,
                                     int out_dim, float* dX, int lddx, const float* H_prev, int act_prev,
                                     const float* keep_mask_prev, float p_prev, float* dW, float* db, void* stream) {
  if (!dY || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  tl_gemm_prec = PREC_F32;
  if (dX) {
    if (!W) return fail(GT_ERR_INVALID, "dX requested without W");
    if (act_prev != ACT_NONE && !H_prev) return fail(GT_ERR_INVALID, "activation derivative requested without H_prev");
    CHK(linear_backward_data(dY, lddy, W, in_dim, 0, dX, lddx, rows, out_dim, in_dim, act_prev, H_prev, in_dim,
                             buffer_spec(keep_mask_prev, p_prev, in_dim), s));
  }
  if (dW || db) {
    if (dW && !X) return fail(GT_ERR_INVALID, "dW requested without X");
    Scratch slabs, colp;
    int r = linear_backward_weight(dY, lddy, X, ldx, rows, out_dim, in_dim, dW, db, false, slabs, colp, s);
    hipError_t err = hipStreamSynchronize(s);
    slabs.release(); colp.release();
    if (r) return r;
    if (err != hipSuccess) return fail(GT_ERR_HIP, "linear_backward: %s", hipGetErrorString(err));
  }
  return GT_OK;
}

// nn.Linear forward / backward through the bf16-STORAGE products (gemm_bf16s.hip.h): operands are cast to bfloat16 images
// (both orientations) exactly as the engine keeps them with GT_OPT_MATMUL_BF16, results come back as float32.  Parity
// hook: against float64 arithmetic on the bf16-rounded operands the results agree to float32 accumulation error.
extern "C" int gt_op_linear_bf16(const float* X, const float* W, const float* bias, int64_t rows, int in_dim, int out_dim, int act,
                                 const float* keep_mask, float p, float* Y, const float* dY, const float* H_prev, int act_prev,
                                 const float* keep_mask_prev, float p_prev, float* dX, float* dW, float* db,
                                 float* Y_image, float* YT_image, void* stream) {
  if (!X || !W || rows < 1 || in_dim < 1 || out_dim < 1) return fail(GT_ERR_INVALID, "bad argument");
  if (act < 0 || act > 2 || act_prev < 0 || act_prev > 2) return fail(GT_ERR_INVALID, "unknown activation");
  hipStream_t s = (hipStream_t)stream;
  const int in8 = pad8(in_dim), out8 = pad8(out_dim);
  const long rows8 = pad8(rows);
  Scratch xb, xbt, wb, wbt, yb, ybt, dyb, dybt, hb, slabs, colp;
  int r = GT_OK;
  auto body = [&]() -> int {
    CHK(xb.ensure((size_t)rows * in8 * 2)); CHK(xbt.ensure((size_t)in_dim * rows8 * 2));
    CHK(wb.ensure((size_t)out_dim * in8 * 2)); CHK(wbt.ensure((size_t)in_dim * out8 * 2));
    CHK(cast_transpose<float>(X, in_dim, rows, in_dim, xb.as<__bf16>(), in8, xbt.as<__bf16>(), rows8, nullptr, false, &colp, s));
    CHK(cast_transpose<float>(W, in_dim, out_dim, in_dim, wb.as<__bf16>(), in8, wbt.as<__bf16>(), out8, nullptr, false, &colp, s));
    if (Y) {
      CHK(yb.ensure((size_t)rows * out8 * 2)); CHK(ybt.ensure((size_t)out_dim * rows8 * 2));
      GemmB16Args g = b16_args();
      g.A = xb.as<__bf16>(); g.lda = in8; g.B = wb.as<__bf16>(); g.ldb = in8; g.M = (int)rows; g.N = out_dim; g.K = in_dim;
      g.C = Y; g.ldc = out_dim; g.Cb = yb.as<__bf16>(); g.ldcb = out8; g.CbT = ybt.as<__bf16>(); g.ldcbt = (int)rows8;
      g.bias = bias; g.epi = B16_FWD; g.act = act; g.drop = buffer_spec(keep_mask, p, out_dim);
      CHK(launch_gemm_b16(g, 1, s));
      // the two bf16 images the same epilogue wrote ([frame][out] and [out][frame]), widened to float32 for inspection
      if (Y_image) {
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(cdiv(rows * out_dim, 256)), dim3(256), 0, s, (const __bf16*)yb.as<__bf16>(), (long)out8, rows, out_dim, Y_image);
        LAUNCH_CHECK();
      }
      if (YT_image) {
        hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(cdiv(rows * out_dim, 256)), dim3(256), 0, s, (const __bf16*)ybt.as<__bf16>(), rows8, (long)out_dim, (int)rows, YT_image);
        LAUNCH_CHECK();
      }
    }
    if (dY) {
      CHK(dyb.ensure((size_t)rows * out8 * 2)); CHK(dybt.ensure((size_t)out_dim * rows8 * 2));
      CHK(cast_transpose<float>(dY, out_dim, rows, out_dim, dyb.as<__bf16>(), out8, dybt.as<__bf16>(), rows8, nullptr, false, &colp, s));
      if (dX) {
        GemmB16Args g = b16_args();
        g.A = dyb.as<__bf16>(); g.lda = out8; g.B = wbt.as<__bf16>(); g.ldb = out8; g.M = (int)rows; g.N = in_dim; g.K = out_dim;
        g.C = dX; g.ldc = in_dim; g.epi = B16_BWD_DATA; g.act = ACT_NONE;
        if (H_prev && act_prev != ACT_NONE) {
          CHK(hb.ensure((size_t)rows * in8 * 2));
          CHK(cast_transpose<float>(H_prev, in_dim, rows, in_dim, hb.as<__bf16>(), in8, nullptr, 0, nullptr, false, &colp, s));
          g.act = act_prev; g.H = hb.as<__bf16>(); g.ldh = in8; g.drop = buffer_spec(keep_mask_prev, p_prev, in_dim);
        }
        CHK(launch_gemm_b16(g, 1, s));
      }
      if (dW) CHK(weight_grad_b16(dybt.as<__bf16>(), rows8, xbt.as<__bf16>(), rows8, rows, out_dim, in_dim, dW, db, false, slabs, s));
    }
    return GT_OK;
  };
  r = body();
  const hipError_t err = hipStreamSynchronize(s);
  for (Scratch* q : {&xb, &xbt, &wb, &wbt, &yb, &ybt, &dyb, &dybt, &hb, &slabs, &colp}) q->release();
  if (r) return r;
  if (err != hipSuccess) return fail(GT_ERR_HIP, "linear_bf16: %s", hipGetErrorString(err));
  return GT_OK;
}
