// Ablation harness for the LSTM forward step kernel (GPU box only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../gantts_amd/csrc/lstm_kernels.hip.h"
using namespace gt;
int main() {
  const int B = 32, T = 1024, H = 256, dirs = 2;
  const long N = (long)B * T;
  float *xproj, *gates, *cst, *out, *state, *W, *bias; int* len;
  hipMalloc(&xproj, N * dirs * 4 * H * 4); hipMalloc(&gates, N * dirs * 4 * H * 4); hipMalloc(&cst, N * dirs * H * 4);
  hipMalloc(&out, N * dirs * H * 4); hipMalloc(&state, 5L * dirs * 32 * H * 4); hipMalloc(&W, 4L * H * H * 4); hipMalloc(&bias, 4 * H * 4);
  hipMalloc(&len, B * 4);
  hipMemset(xproj, 0, N * dirs * 4 * H * 4); hipMemset(state, 0, 5L * dirs * 32 * H * 4); hipMemset(W, 0, 4L * H * H * 4); hipMemset(bias, 0, 4 * H * 4);
  { std::vector<float> h(4L * H * H); for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.12f;
    hipMemcpy(W, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> xp(N * dirs * 4 * H); for (auto& v : xp) v = (rand() / (float)RAND_MAX - 0.5f) * 2.f;
    hipMemcpy(xproj, xp.data(), xp.size() * 4, hipMemcpyHostToDevice); }
  std::vector<int> hl(B, T); hipMemcpy(len, hl.data(), B * 4, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)lstm_fwd_step_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lstm_lds_bytes());
  LstmStepArgs a; memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.Bpad = 32; a.lengths = len;
  for (int d = 0; d < 2; ++d) { a.Whh[d] = W; a.bih[d] = bias; a.bhh[d] = bias; }
  a.xproj = xproj; a.gates = gates; a.cst = cst; a.out = out;
  const size_t st = (size_t)dirs * 32 * H;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 40; ++rep) {
    hipEventRecord(e0);
    for (int s = 0; s < T; ++s) {
      a.step = s; const int cur = s & 1;
      a.h_prev = state + cur * st; a.c_prev = state + (2 + cur) * st; a.h_next = state + (cur ^ 1) * st; a.c_next = state + (2 + (cur ^ 1)) * st;
      hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(H / 8, dirs, 1), dim3(256), lstm_lds_bytes(), 0, a);
    }
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep % 5 == 0 || rep == 39) printf("rep %d fwd step: %.2f us per launch (%s)\n", rep, ms * 1e3 / T, hipGetErrorString(hipGetLastError()));
  }
  return 0;
}
