// libgantts_hip.so -- float32 MFMA family, backward-data products dX = (dZ.W) (.) f'(H) (GEMM_NN): kernel instantiations
#include "gemm_f32_launch.hip.h"

int launch_gemm_nn(const GemmArgs& g, int bm, int bn, hipStream_t s) {
  if (bm == 64) return bn == 64 ? launch_gemm_v<GEMM_NN, 64, 64>(g, 1, s) : launch_gemm_v<GEMM_NN, 64, 128>(g, 1, s);
  return bn == 64 ? launch_gemm_v<GEMM_NN, 128, 64>(g, 1, s) : launch_gemm_v<GEMM_NN, 128, 128>(g, 1, s);
}
