// A/B harness of the LSTM recurrence kernels (no torch): per-step launches (lstm_kernels.hip.h) vs the persistent
// kernels (lstm_seq_kernels.hip.h) on the same random layer -- max |difference| of every stash / gradient buffer and
// time per layer pass.   usage: lstm_seq_bench [B T H dirs reps]      (default: cfg3 layer, 32 1024 256 2 3)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/lstm_seq_bench.hip -o tools/bin/lstm_seq_bench
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
static double maxdiff(const float* a, const float* b, size_t n, double* scale) {
  std::vector<float> ha(n), hb(n);
  CK(hipMemcpy(ha.data(), a, n * sizeof(float), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hb.data(), b, n * sizeof(float), hipMemcpyDeviceToHost));
  double m = 0, s = 0;
  for (size_t i = 0; i < n; ++i) {
    if (!std::isfinite(ha[i]) || !std::isfinite(hb[i])) return 1e30;
    m = std::max(m, (double)fabsf(ha[i] - hb[i])); s = std::max(s, (double)fabsf(hb[i]));
  }
  *scale = s;
  return m;
}
static int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

struct Bufs { float *xproj, *gates, *cst, *out; };

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 32, T = argc > 2 ? atoi(argv[2]) : 1024, H = argc > 3 ? atoi(argv[3]) : 256;
  const int dirs = argc > 4 ? atoi(argv[4]) : 2, reps = argc > 5 ? atoi(argv[5]) : 3;
  const long N = (long)B * T;
  const int Bpad = cdiv(B, 32) * 32;
  printf("LSTM layer B=%d T=%d H=%d dirs=%d\n", B, T, H, dirs);
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.name, prop.multiProcessorCount);
  // parameters and inputs
  const float k = 1.f / sqrtf((float)H);
  float *Whh[2], *bih[2], *bhh[2];
  for (int d = 0; d < 2; ++d) { Whh[d] = dfill((size_t)4 * H * H, k, 11 + d); bih[d] = dfill(4 * H, k, 21 + d); bhh[d] = dfill(4 * H, k, 31 + d); }
  float* xp0 = dfill((size_t)N * dirs * 4 * H, 1.0f, 5);          // X-projection (kept pristine)
  float* dout = dfill((size_t)N * dirs * H, 0.1f, 6);
  std::vector<int> hl(B);
  unsigned sd = 77;
  for (int b = 0; b < B; ++b) { hl[b] = T / 2 + (int)((frand(sd) * 0.5f + 0.5f) * (T - T / 2)); if (hl[b] > T) hl[b] = T; if (hl[b] < 1) hl[b] = 1; }
  hl[0] = T;
  if (B > 2) hl[B - 1] = 1;
  int* lengths = dalloc<int>(B);
  CK(hipMemcpy(lengths, hl.data(), B * sizeof(int), hipMemcpyHostToDevice));
  Bufs R, P;   // reference (per-step) and persistent
  for (Bufs* q : {&R, &P}) {
    q->xproj = dalloc<float>((size_t)N * dirs * 4 * H); q->gates = dalloc<float>((size_t)N * dirs * 4 * H);
    q->cst = dalloc<float>((size_t)N * dirs * H); q->out = dalloc<float>((size_t)N * dirs * H);
  }
  const size_t st = (size_t)dirs * Bpad * H;
  float* state = dalloc<float>(5 * st);
  unsigned int* fault = dalloc<unsigned int>(16);
  const int HP = H <= 256 ? 256 : 512;
  const size_t xch_n = (size_t)dirs * cdiv(B, 8) * lstm_bwd_xch_u64(HP);       // the larger of the two layouts, 8-sequence tiles
  const size_t chk_n = (size_t)dirs * cdiv(B, 8) * 256;
  unsigned long long* xch = dalloc<unsigned long long>(xch_n + chk_n);
  int xcd_local = 1, bt = 16;
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipFuncSetAttribute((const void*)lstm_fwd_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lstm_lds_bytes()));
  CK(hipFuncSetAttribute((const void*)lstm_bwd_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lstm_lds_bytes()));

  auto steps = [&](Bufs& q, bool backward) {
    CK(hipMemsetAsync(state, 0, 5 * st * sizeof(float), s));
    LstmStepArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.Bpad = Bpad; a.lengths = lengths;
    for (int d = 0; d < dirs; ++d) { a.Whh[d] = Whh[d]; a.bih[d] = bih[d]; a.bhh[d] = bhh[d]; }
    a.xproj = q.xproj; a.gates = q.gates; a.cst = q.cst; a.out = q.out; a.dout = dout; a.dc_state = state + 4 * st;
    for (int step = 0; step < T; ++step) {
      a.step = step;
      if (!backward) {
        const int cur = step & 1;
        a.h_prev = state + (size_t)cur * st; a.c_prev = state + (2 + (size_t)cur) * st;
        a.h_next = state + (size_t)(cur ^ 1) * st; a.c_next = state + (2 + (size_t)(cur ^ 1)) * st;
        hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(cdiv(H, 8), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
      } else {
        hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3(cdiv(H, 32), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
      }
    }
    CK(hipGetLastError());
  };
  auto seq = [&](Bufs& q, bool backward, int upc) -> bool {
    LstmSeqArgs a; memset(&a, 0, sizeof(a));
    a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.nbt = cdiv(B, bt); a.lengths = lengths;
    for (int d = 0; d < dirs; ++d) { a.Whh[d] = Whh[d]; a.bih[d] = bih[d]; a.bhh[d] = bhh[d]; }
    a.xproj = q.xproj; a.gates = q.gates; a.cst = q.cst; a.out = q.out; a.dout = dout;
    a.xch = xch; a.fault = fault; a.timeout_ticks = 100000000ULL;     // 1 s
    CK(hipMemsetAsync(xch, 0, (xch_n + chk_n) * sizeof(unsigned long long), s));
    a.xcc_chk = xch + xch_n; a.nxcd = 8; a.allow_xcd_local = xcd_local; a.dbg_protocol = fault + 4;
    a.ncu = backward ? cdiv(H, 16) : cdiv(H, upc);
    const int rounds = cdiv(dirs * a.nbt, a.nxcd);
    const int grid = a.nxcd * a.ncu * rounds;
    size_t lds; const void* kern;
#define PICK(KERN, LDS) { kern = (const void*)KERN; lds = LDS; CK(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
      int per = 0; CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, KERN, 256, lds)); \
      if (a.ncu * rounds > per * (prop.multiProcessorCount / 8)) { printf("  %d workgroups per XCD exceed residency %d x %d\n", a.ncu * rounds, per, prop.multiProcessorCount / 8); return false; } \
      hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), lds, s, a); }
#define PICK_BT(K8, K16, LDS) { if (bt == 8) PICK(K8, LDS) else PICK(K16, LDS) }
    if (backward) { if (HP == 256) PICK_BT((lstm_bwd_seq_kernel<256, 8>), (lstm_bwd_seq_kernel<256, 16>), (lstm_bwd_seq_lds<256>())) else PICK_BT((lstm_bwd_seq_kernel<512, 8>), (lstm_bwd_seq_kernel<512, 16>), (lstm_bwd_seq_lds<512>())) }
    else if (upc == 4) { if (HP == 256) PICK_BT((lstm_fwd_seq_kernel<256, 4, 8>), (lstm_fwd_seq_kernel<256, 4, 16>), (lstm_fwd_seq_lds<256, 4>())) else PICK_BT((lstm_fwd_seq_kernel<512, 4, 8>), (lstm_fwd_seq_kernel<512, 4, 16>), (lstm_fwd_seq_lds<512, 4>())) }
    else if (upc == 8) { if (HP == 256) PICK_BT((lstm_fwd_seq_kernel<256, 8, 8>), (lstm_fwd_seq_kernel<256, 8, 16>), (lstm_fwd_seq_lds<256, 8>())) else PICK_BT((lstm_fwd_seq_kernel<512, 8, 8>), (lstm_fwd_seq_kernel<512, 8, 16>), (lstm_fwd_seq_lds<512, 8>())) }
    else { if (HP == 256) PICK_BT((lstm_fwd_seq_kernel<256, 16, 8>), (lstm_fwd_seq_kernel<256, 16, 16>), (lstm_fwd_seq_lds<256, 16>())) else PICK_BT((lstm_fwd_seq_kernel<512, 16, 8>), (lstm_fwd_seq_kernel<512, 16, 16>), (lstm_fwd_seq_lds<512, 16>())) }
#undef PICK_BT
#undef PICK
    CK(hipGetLastError());
    return true;
  };
  auto timed = [&](const char* what, auto fn) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, s));
      fn();
      CK(hipEventRecord(e1, s));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
    }
    printf("  %-34s %9.3f ms  = %7.2f us/step\n", what, best, 1e3 * best / T);
    return best;
  };
  auto reset_xp = [&](Bufs& q) { CK(hipMemcpyAsync(q.xproj, xp0, (size_t)N * dirs * 4 * H * sizeof(float), hipMemcpyDeviceToDevice, s)); };
  auto report = [&](const char* what) {
    double sc;
    double dg = maxdiff(P.gates, R.gates, (size_t)N * dirs * 4 * H, &sc); printf("    %s gates  max|d| %.3e (scale %.2e)\n", what, dg, sc);
    double dc = maxdiff(P.cst, R.cst, (size_t)N * dirs * H, &sc); printf("    %s c      max|d| %.3e (scale %.2e)\n", what, dc, sc);
    double dh = maxdiff(P.out, R.out, (size_t)N * dirs * H, &sc); printf("    %s h      max|d| %.3e (scale %.2e)\n", what, dh, sc);
    unsigned f, pr[8]; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(pr, fault + 4, 32, hipMemcpyDeviceToHost));
    printf("    protocol per group (1 xcd-local, 2 agent): %u %u %u %u\n", pr[0], pr[1], pr[2], pr[3]);
    printf("    fault word %u  %s\n", f, (dg < 2e-4 && dc < 2e-3 && dh < 2e-4 && f == 0) ? "OK" : "MISMATCH");
  };

  // ---- forward
  printf("forward\n");
  reset_xp(R);
  timed("per-step launches", [&] { steps(R, false); });
  for (int proto = 0; proto < 2; ++proto)
  for (int btv : {16, 8})
  for (int upc : {4, 8, 16}) {
    xcd_local = proto; bt = btv;
    if (proto == 0 && btv == 8) continue;
    reset_xp(P);
    CK(hipMemsetAsync(P.gates, 0xff, (size_t)N * dirs * 4 * H * sizeof(float), s));
    char nm[64]; snprintf(nm, sizeof(nm), "persist. %s bt%d %d units/wg", proto ? "xcd-local" : "agent", bt, upc);
    bool ok = true;
    timed(nm, [&] { ok = seq(P, false, upc); });
    if (ok) report("fwd");
  }
  // ---- backward (stashes of the reference forward are shared: copy them so both sides differentiate the same values)
  printf("backward\n");
  CK(hipMemcpy(P.gates, R.gates, (size_t)N * dirs * 4 * H * sizeof(float), hipMemcpyDeviceToDevice));
  CK(hipMemcpy(P.cst, R.cst, (size_t)N * dirs * H * sizeof(float), hipMemcpyDeviceToDevice));
  CK(hipMemcpy(P.out, R.out, (size_t)N * dirs * H * sizeof(float), hipMemcpyDeviceToDevice));
  timed("per-step launches", [&] { steps(R, true); });
  for (int proto = 0; proto < 2; ++proto)
  for (int btv : {16, 8}) {
  xcd_local = proto; bt = btv;
  bool okb = true;
  char nmb[64]; snprintf(nmb, sizeof(nmb), "persistent %s bt%d", proto ? "xcd-local" : "agent", bt);
  timed(nmb, [&] { okb = seq(P, true, 0); });
  if (okb) {
    double sc;
    double dd = maxdiff(P.xproj, R.xproj, (size_t)N * dirs * 4 * H, &sc);
    unsigned f; CK(hipMemcpy(&f, fault, 4, hipMemcpyDeviceToHost));
    printf("    bwd dG     max|d| %.3e (scale %.2e)  fault word %u  %s\n", dd, sc, f, (dd < 1e-4 * std::max(1.0, sc) && f == 0) ? "OK" : "MISMATCH");
  }
  }
  return 0;
}
