// libgantts_hip.so -- float32 MFMA family, pair launches (a layer's backward-data product + weight gradient in one launch, gemm_pair_kernel)
#include "gemm_f32_launch.hip.h"

static void gemm_set_wide_store(int kind, GemmArgs& g) {
  // 16-byte accesses need a 16-byte aligned base and a row pitch that is a multiple of 4 floats
  g.wide_store = kind != GEMM_TN && (g.ldc % 4 == 0) && (((uintptr_t)g.C) % 16 == 0) &&
                 (kind != GEMM_NN || g.act == ACT_NONE || ((g.ldh % 4 == 0) && (((uintptr_t)g.H) % 16 == 0)));
}
int launch_gemm_pair(const GemmArgs& nn_in, const GemmArgs& tn_in, int nslab, hipStream_t s) {
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
