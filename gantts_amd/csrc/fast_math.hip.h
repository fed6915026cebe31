// Fast gate functions shared by the recurrent kernels (lstm_seq_kernels.hip.h forward kernel, sru_kernels.hip.h scans).
#pragma once
#include <hip/hip_runtime.h>

namespace gt {

// Gate non-linearities (one v_exp_f32 + one v_rcp_f32 each instead of the library's expf / tanhf, which sit on
// the per-step critical path of one wave): e^x = 2^(x log2 e) with the product's rounding error folded back in (a few
// ulp); tanh by its odd Taylor polynomial below 0.3 (no cancellation) and by (1 - e^-2|x|) / (1 + e^-2|x|) above.
// Against the library functions on a cfg3 LSTM layer (T = 1024 steps of feedback): max |difference| 3.6e-7 over gates, c, h.
// Saturation: the argument is clamped to [-87.3, 88.7] (exp2's finite, normal range), so e is never inf / denormal-zero
// and the correction term never meets inf * 0: fast_sigmoid(-100) = 0-ish (3e-39 flushes to 0 after the rcp), fast_sigmoid(+100) = 1,
// fast_tanh(+-100) = +-1 -- like the library functions -- instead of NaN (exp2(t >= 128) = inf, then fmaf(inf, r*LN2 <= 0, inf)); NaN in, NaN out.
__device__ __forceinline__ float fast_exp(float x) {
  const float L2E_HI = 1.44269502162933349609375f, L2E_LO = 1.925963033500011e-8f, LN2 = 0.6931471805599453f;
  x = x < -87.3f ? -87.3f : (x > 88.7f ? 88.7f : x);      // selects, not min/max: a NaN argument stays NaN
  const float t = x * L2E_HI;
  float r = fmaf(x, L2E_HI, -t);
  r = fmaf(x, L2E_LO, r);
  const float e = __builtin_amdgcn_exp2f(t);
  return fmaf(e, r * LN2, e);
}
__device__ __forceinline__ float fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.f + fast_exp(-x)); }
__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x), x2 = x * x;
  const float p = x * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, -1382.f / 155925.f, 62.f / 2835.f), -17.f / 315.f), 2.f / 15.f), -1.f / 3.f), 1.f);
  const float t = fast_exp(-2.f * ax);
  const float q = copysignf((1.f - t) * __builtin_amdgcn_rcpf(1.f + t), x);
  return ax < 0.3f ? p : q;
}
}  // namespace gt
