// libgantts_hip.so -- float32 MFMA family, pair launches (a layer's backward-data product + weight gradient in one launch, gemm_pair_kernel)
#include "gemm_f32_launch.hip.h"

int launch_gemm_pair(const GemmArgs& nn_in, const GemmArgs& tn_in, int nslab, hipStream_t s) {
  GemmArgs nn = nn_in, tn = tn_in;
  nn.wide_store = gemm_wide_store_ok(GEMM_NN, nn) ? 1 : 0;
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
  if (g_prof.wants(5)) {
    rec.kind = 5; rec.bn = 64; rec.am = -1; rec.flops = 2.0 * nn.M * nn.N * nn.K + 2.0 * tn.M * tn.N * tn.K;
    rec.bytes = gemm_algorithmic_bytes(GEMM_NN, nn) + gemm_algorithmic_bytes(GEMM_TN, tn);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  // GT_PAIR_ORDER (default 1): weight-gradient workgroups first (longest work first); 0 = backward-data tiles first
  const int tn_first = gt_tuning().pair_order;   // measured: 108.2 -> 104.4 us per pair launch, cfg2 step 1.523 -> 1.499 ms
  const int grid = n1 + n2;
  if (bf16) hipLaunchKernelGGL(gemm_pair_kernel<PREC_BF16>, dim3(grid), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else if (am == GEMM_A_NONE) hipLaunchKernelGGL((gemm_pair_kernel<PREC_F32, GEMM_A_NONE>), dim3(grid), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else if (am == GEMM_A_LEAKY_PHILOX) hipLaunchKernelGGL((gemm_pair_kernel<PREC_F32, GEMM_A_LEAKY_PHILOX>), dim3(grid), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  else      hipLaunchKernelGGL(gemm_pair_kernel<PREC_F32>, dim3(grid), dim3(GEMM_THREADS), lds, s, nn, tn, n1, tn_first);
  LAUNCH_CHECK();
  if (g_prof.wants(5)) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}

// the two weight-gradient products of a split first layer in one launch (gemm_tn_pair_kernel); float32, 64 x 64 tiles
// rider (optional): a backward-data product without activation derivative whose B operand takes the 4-byte loader (K-contiguous A, 16-byte
// loadable): its 64 x 64 tiles are the launch's last workgroups.  *rode tells the caller whether it was taken.
int launch_gemm_tn_pair(const GemmArgs& g1_in, const GemmArgs& g2_in, int nslab1, int nslab2, hipStream_t s, const GemmArgs* rider, bool* rode) {
  GemmArgs g1 = g1_in, g2 = g2_in, g3;
  memset(&g3, 0, sizeof(g3));
  int n3 = 0;
  if (rode) *rode = false;
  if (rider && rider->act == ACT_NONE && !rider->accumulate && rider->M > 0 && gemm_vec_ok(rider->A, rider->lda, true) && tl_gemm_prec == PREC_F32 &&
      gemm_small_tiles_ok()) {
    g3 = *rider;
    g3.wide_store = gemm_wide_store_ok(GEMM_NN, g3) ? 1 : 0;
    g3.n_tiles_m = cdiv(g3.M, 64); g3.n_tiles_n = cdiv(g3.N, 64);
    n3 = g3.n_tiles_m * g3.n_tiles_n;
    if (rode) *rode = true;
  }
  g1.wide_store = g2.wide_store = 0;
  g1.n_tiles_m = cdiv(g1.M, 64); g1.n_tiles_n = cdiv(g1.N, 64);
  g2.n_tiles_m = cdiv(g2.M, 64); g2.n_tiles_n = cdiv(g2.N, 64);
  const int n1 = g1.n_tiles_m * g1.n_tiles_n * nslab1, n2 = g2.n_tiles_m * g2.n_tiles_n * nslab2;
  const size_t lds = std::max(gemm_lds_bytes<GEMM_TN, 64, 64>(), gemm_lds_bytes<GEMM_NN, 64, 64>());
  CHK(ensure_dyn_lds((const void*)gemm_tn_pair_kernel<PREC_F32>, lds));
  GemmProfiler::Rec rec;
  if (g_prof.wants(6)) {
    rec.kind = 6; rec.bn = 64; rec.am = -1; rec.flops = 2.0 * g1.M * g1.N * g1.K + 2.0 * g2.M * g2.N * g2.K + (n3 ? 2.0 * g3.M * g3.N * g3.K : 0.0);
    rec.bytes = gemm_algorithmic_bytes(GEMM_TN, g1) + 4.0 * (double)g1.M * g1.K + gemm_algorithmic_bytes(GEMM_TN, g2) + (n3 ? gemm_algorithmic_bytes(GEMM_NN, g3) : 0.0);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL(gemm_tn_pair_kernel<PREC_F32>, dim3(n1 + n2 + n3), dim3(GEMM_THREADS), lds, s, g1, g2, n1, g3, n3);
  LAUNCH_CHECK();
  if (g_prof.wants(6)) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
