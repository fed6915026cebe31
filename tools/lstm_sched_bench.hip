// Schedule / geometry sweep of the persistent LSTM recurrence kernels (no torch): every (precision, batch tile, units per
// workgroup, EARLY, FM) variant of lstm_seq_kernels.hip.h on the same random cfg3 layer; results are compared with the
// round-2 kernel (EARLY = FM = false) of the same precision and geometry: EARLY alone must be bit-identical (it changes
// the order of memory operations, never a value), FM within a few 1e-7.  -DGT_LSTM_SEQ_ABLATE=bits times the kernels with
// the stash stores (1), the gate functions (2), the stash loads (4) removed.  The variants that were measured and dropped
// (gate threads as extra waves, deeper request queues, tagged backward exchange) live in tools/experiments/.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DGT_LSTM_SEQ_ABLATE=bits] tools/lstm_sched_bench.hip -o tools/bin/lstm_sched_bench
//   usage: lstm_sched_bench [B T H dirs reps]     (default 32 1024 256 2 3; H <= 256)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "../gantts_amd/csrc/lstm_seq_kernels.hip.h"

using namespace gt;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static float frand(unsigned& s) { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f * 2.f - 1.f; }
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc((void**)&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
static float* dfill(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  for (auto& v : h) v = frand(seed) * scale;
  float* p = dalloc<float>(n);
  CK(hipMemcpy(p, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return p;
}
static size_t ndiff(const float* a, const float* b, size_t n) {
  std::vector<unsigned> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  size_t c = 0;
  for (size_t i = 0; i < n; ++i) c += ha[i] != hb[i];
  return c;
}
static double maxabs(const float* a, const float* b, size_t n) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * 4, hipMemcpyDeviceToHost));
  double m = 0;
  for (size_t i = 0; i < n; ++i) { if (!std::isfinite(ha[i])) return 1e30; m = std::max(m, (double)fabsf(ha[i] - hb[i])); }
  return m;
}
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
struct Bufs { float *xproj, *gates, *cst, *out; };

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 256;
  const int dirs = argc > 4 ? atoi(argv[4]) : 2, reps = argc > 5 ? atoi(argv[5]) : 3;
  if (H > 256) { printf("H <= 256 only\n"); return 1; }
  const long N = (long)B * T;
  printf("LSTM layer B=%d T=%d H=%d dirs=%d  ablate=%d\n", B, T, H, dirs, (int)LSTM_ABL);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const float k = 1.f / sqrtf((float)H);
  float *Whh[2], *bih[2], *bhh[2];
  for (int d = 0; d < 2; ++d) { Whh[d] = dfill((size_t)4 * H * H, k, 11 + d); bih[d] = dfill(4 * H, k, 21 + d); bhh[d] = dfill(4 * H, k, 31 + d); }
  float* xp0 = dfill((size_t)N * dirs * 4 * H, 1.0f, 5);
  float* dout = dfill((size_t)N * dirs * H, 0.1f, 6);
  std::vector<int> hl(B);
  unsigned sd = 77;
  for (int b = 0; b < B; ++b) { hl[b] = T / 2 + (int)((frand(sd) * 0.5f + 0.5f) * (T - T / 2)); hl[b] = std::min(std::max(hl[b], 1), T); }
  hl[0] = T;
  int* lengths = dalloc<int>(B);
  CK(hipMemcpy(lengths, hl.data(), B * sizeof(int), hipMemcpyHostToDevice));
  Bufs R, P;
  for (Bufs* q : {&R, &P}) {
    q->xproj = dalloc<float>((size_t)N * dirs * 4 * H); q->gates = dalloc<float>((size_t)N * dirs * 4 * H);
    q->cst = dalloc<float>((size_t)N * dirs * H); q->out = dalloc<float>((size_t)N * dirs * H);
  }
  unsigned int* fault = dalloc<unsigned int>(16);
  constexpr int HP = 256;
  const size_t xch_n = (size_t)dirs * cdiv(B, 8) * lstm_bwd_xch_u64(HP), chk_n = (size_t)dirs * cdiv(B, 8) * 256;
  unsigned long long* xch = dalloc<unsigned long long>(xch_n + chk_n);
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  auto launch = [&](const void* kern, size_t lds, Bufs& q, int bt, int ncu, int block) -> bool {
    LstmSeqArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.nbt = cdiv(B, bt); a.lengths = lengths;
    for (int d = 0; d < dirs; ++d) { a.Whh[d] = Whh[d]; a.bih[d] = bih[d]; a.bhh[d] = bhh[d]; }
    a.xproj = q.xproj; a.gates = q.gates; a.cst = q.cst; a.out = q.out; a.dout = dout;
    a.xch = xch; a.fault = fault; a.timeout_ticks = 50000000ULL;      // 0.5 s
    CK(hipMemsetAsync(xch, 0, (xch_n + chk_n) * sizeof(unsigned long long), s));
    a.xcc_chk = xch + xch_n; a.nxcd = 8; a.allow_xcd_local = 1; a.ncu = ncu;
    const int rounds = cdiv(dirs * a.nbt, a.nxcd);
    CK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, block, lds));
    if (a.ncu * rounds > per * (prop.multiProcessorCount / 8)) { printf("  (grid exceeds residency)\n"); return false; }
    void* args[] = {&a};
    CK(hipLaunchKernel(kern, dim3(a.nxcd * a.ncu * rounds), dim3(block), args, lds, s));
    return true;
  };
  auto timed = [&](auto fn) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, s)); fn(); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = std::min(best, ms);
    }
    return best;
  };
  auto reset_xp = [&](Bufs& q) { CK(hipMemcpyAsync(q.xproj, xp0, (size_t)N * dirs * 4 * H * sizeof(float), hipMemcpyDeviceToDevice, s)); };
  auto faultw = [&] { unsigned f; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost)); return f; };

  printf("forward   (us per time step; 'diff' = words that differ from the SCHED 0 result of the same geometry)\n");
#define FWD(PREC, BT, UPC, EARLY, FM, LW) { \
    const bool ref = !EARLY && !FM && !LW; \
    Bufs& q = ref ? R : P; reset_xp(q); \
    const void* kern = (const void*)lstm_fwd_seq_kernel<HP, UPC, BT, PREC, EARLY, FM, LW>; \
    bool ok = true; \
    const float ms = timed([&] { ok = launch(kern, lstm_fwd_seq_lds<HP, UPC>(), q, BT, cdiv(H, UPC), lstm_fwd_block(UPC, BT, LW)); }); \
    if (ok) { \
      size_t nd = ref ? 0 : ndiff(P.gates, R.gates, (size_t)N * dirs * 4 * H) + ndiff(P.out, R.out, (size_t)N * dirs * H) + ndiff(P.cst, R.cst, (size_t)N * dirs * H); \
      const double md = ref ? 0 : std::max(maxabs(P.gates, R.gates, (size_t)N * dirs * 4 * H), std::max(maxabs(P.out, R.out, (size_t)N * dirs * H), maxabs(P.cst, R.cst, (size_t)N * dirs * H))); \
      printf("  %s bt%-2d upc%-2d early%d fm%d lw%d  %8.3f ms = %6.3f us/step   diff %zu max|d| %.2e  fault %u\n", PREC == PREC_BF16 ? "bf16" : "f32 ", BT, UPC, (int)EARLY, (int)FM, (int)LW, ms, 1e3 * ms / T, nd, md, faultw()); \
    } }
#define FWD3(PREC, BT, UPC) FWD(PREC, BT, UPC, false, false, false) FWD(PREC, BT, UPC, true, false, false) FWD(PREC, BT, UPC, false, false, true) FWD(PREC, BT, UPC, true, true, false) FWD(PREC, BT, UPC, false, true, true)
  FWD3(PREC_BF16, 8, 8) FWD3(PREC_BF16, 8, 16) FWD3(PREC_BF16, 16, 8) FWD3(PREC_BF16, 16, 16)
  FWD3(PREC_F32, 8, 8) FWD3(PREC_F32, 8, 16) FWD3(PREC_F32, 16, 8) FWD3(PREC_F32, 16, 16)

  printf("backward\n");
  // stashes of one forward pass (f32, bt8, upc8, sched 0) feed every backward variant
  reset_xp(R); launch((const void*)lstm_fwd_seq_kernel<HP, 8, 8, PREC_F32, false, false, false>, lstm_fwd_seq_lds<HP, 8>(), R, 8, cdiv(H, 8), 256);
  CK(hipStreamSynchronize(s));
  CK(hipMemcpy(P.gates, R.gates, (size_t)N * dirs * 4 * H * sizeof(float), hipMemcpyDeviceToDevice));
  CK(hipMemcpy(P.cst, R.cst, (size_t)N * dirs * H * sizeof(float), hipMemcpyDeviceToDevice));
  CK(hipMemcpy(P.out, R.out, (size_t)N * dirs * H * sizeof(float), hipMemcpyDeviceToDevice));
#define BWD(PREC, BT, LW, TAG) { \
    const bool ref = !LW && !TAG; \
    Bufs& q = ref ? R : P; \
    const void* kern = (const void*)lstm_bwd_seq_kernel<HP, BT, PREC, LW, TAG>; \
    bool ok = true; \
    const float ms = timed([&] { ok = launch(kern, lstm_bwd_seq_lds<HP>(), q, BT, cdiv(H, 16), lstm_bwd_block(BT, LW)); }); \
    if (ok) { \
      const size_t nd = ref ? 0 : ndiff(P.xproj, R.xproj, (size_t)N * dirs * 4 * H); \
      printf("  %s bt%-2d loader-waves %d tagged %d  %8.3f ms = %6.3f us/step  diff %zu  fault %u\n", PREC == PREC_BF16 ? "bf16" : "f32 ", BT, (int)LW, (int)TAG, ms, 1e3 * ms / T, nd, faultw()); } }
#define BWD4(PREC, BT) BWD(PREC, BT, false, false) BWD(PREC, BT, true, false) BWD(PREC, BT, false, true) BWD(PREC, BT, true, true)
  BWD4(PREC_BF16, 8) BWD4(PREC_BF16, 16) BWD4(PREC_F32, 8) BWD4(PREC_F32, 16)
  return 0;
}
