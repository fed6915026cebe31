// libgantts_hip.so -- float32 MFMA family, weight-gradient slabs dW = dZ^T.X (GEMM_TN): kernel instantiations
#include "gemm_f32_launch.hip.h"

int launch_gemm_tn(const GemmArgs& g, int bm, int bn, int nslab, hipStream_t s) {
  if (bm == 64) return launch_gemm_v<GEMM_TN, 64, 64>(g, nslab, s);
  return bn == 64 ? launch_gemm_v<GEMM_TN, 128, 64>(g, nslab, s) : launch_gemm_v<GEMM_TN, 128, 128>(g, nslab, s);
}
