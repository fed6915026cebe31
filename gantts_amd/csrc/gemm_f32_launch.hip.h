// Launchers of the float32 MFMA GEMM family (gemm_f32.hip.h), included by the translation units that instantiate its kernels
// (eng_gemm_f32_nt / _nn / _tn / _pair.hip: one operand orientation each, so that they compile in parallel).
#pragma once
#include "engine_internal.hip.h"

static double gemm_algorithmic_bytes(int kind, const GemmArgs& g) {
  double b = 4.0 * ((double)g.M * g.K + (double)g.K * g.N + (double)g.M * g.N);
  if (kind == GEMM_NN && g.act != ACT_NONE && g.H) b += 4.0 * (double)g.M * g.N;     // the producer's stored activation
  return b;
}
// (r6: the hot 64 x 64 forward launches with HALF the LDS stage (BKT = 16: 58 VGPRs, 18 KB of LDS, eight workgroups per CU instead of four) measured
//  77.6 vs 77.1 us: the K stage is not waiting for latency.  BKT stays 32.)
template <int KIND, int BM, int BN, bool VA, bool VB, int PREC, int AM>
static int launch_gemm_impl(GemmArgs g, int nslab, hipStream_t s) {
  const size_t lds = gemm_lds_bytes<KIND, BM, BN, PREC>();
  CHK(ensure_dyn_lds((const void*)gemm_f32_kernel<KIND, BM, BN, VA, VB, PREC, 32, AM>, lds));
  g.n_tiles_m = cdiv(g.M, BM);
  g.n_tiles_n = cdiv(g.N, BN);
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  if (grid <= 0) return GT_OK;
  GemmProfiler::Rec rec;
  if (g_prof.wants(KIND)) {
    rec.kind = KIND; rec.bn = BN; rec.am = AM; rec.flops = 2.0 * g.M * g.N * g.K; rec.bytes = gemm_algorithmic_bytes(KIND, g);
    if (g.K_seg > 0) {      // + the second segment over all result rows; the result (and the second segment's A rows) once per half
      const double halves = g.dual_rows > 0 ? 2.0 : 1.0;
      rec.flops += 2.0 * halves * g.M * g.N * g.K_seg;
      rec.bytes += 4.0 * (halves * g.M * g.K_seg + (double)g.K_seg * g.N + (halves - 1.0) * g.M * g.N);
    }
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_f32_kernel<KIND, BM, BN, VA, VB, PREC, 32, AM>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.wants(KIND)) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
template <int KIND, int BM, int BN, bool VA, bool VB, int PREC>
static int launch_gemm_t(const GemmArgs& g, int nslab, hipStream_t s) {
  // the hot shape of the float32 step (64 x 64 tiles, 16-byte loadable operands) has its two common epilogue flavours
  // compiled in: no activation, LeakyReLU + Philox dropout (gemm_f32.hip.h: GemmAmode); everything else decides at run time
  if (g.K_seg > 0) {      // two-segment forward product (+ two halves): the compiled LeakyReLU + Philox flavour on 64 x 64 tiles only
    if constexpr (KIND == GEMM_NT && BM == 64 && BN == 64 && VA && VB && PREC == PREC_F32) {
      if (g.act == ACT_LEAKY_DROPOUT && g.drop.mode == DROP_PHILOX && g.addm == nullptr)
        return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_LEAKY_PHILOX_SEG>(g, nslab, s);
    }
    return fail(GT_ERR_INVALID, "segmented forward product: float32, 64 x 64 tiles, 16-byte loadable operands, LeakyReLU + Philox only");
  }
  if constexpr (KIND != GEMM_TN && BM == 64 && BN == 64 && VA && VB && PREC == PREC_F32) {
    if (g.addm == nullptr) {
      if (g.act == ACT_NONE) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_NONE>(g, nslab, s);
      if (g.act == ACT_LEAKY_DROPOUT && g.drop.mode == DROP_PHILOX) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_LEAKY_PHILOX>(g, nslab, s);
    } else if constexpr (KIND == GEMM_NT) {
      if (g.act == ACT_LEAKY_DROPOUT && g.drop.mode == DROP_PHILOX) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_LEAKY_PHILOX_ADDM>(g, nslab, s);
    }
  }
  if constexpr (KIND == GEMM_TN) {
    if (g.A2 != nullptr) {      // summed A operand: the 64 x 64 float32 form with 16-byte loadable operands only
      if constexpr (BM == 64 && BN == 64 && VA && VB && PREC == PREC_F32) return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_TN_SUM2>(g, nslab, s);
      else return fail(GT_ERR_INVALID, "weight gradient with a summed operand needs 64 x 64 tiles and 16-byte loadable operands");
    }
  } else if (g.A2 != nullptr) {
    return fail(GT_ERR_INVALID, "summed A operand: weight gradients only");
  }
  if (g.addm != nullptr && KIND != GEMM_NT) return fail(GT_ERR_INVALID, "added matrix: forward products only");
  return launch_gemm_impl<KIND, BM, BN, VA, VB, PREC, GEMM_A_RUNTIME>(g, nslab, s);
}
template <int KIND, int BM, int BN>
static int launch_gemm_v(const GemmArgs& g_in, int nslab, hipStream_t s) {
  GemmArgs g = g_in;
  g.wide_store = gemm_wide_store_ok(KIND, g) ? 1 : 0;
  const bool va = gemm_vec_ok(g.A, g.lda, KIND != GEMM_TN);
  const bool vb = gemm_vec_ok(g.B, g.ldb, KIND == GEMM_NT);
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

