// libgantts_hip.so -- dispatch of the bf16-storage product family (gemm_bf16s.hip.h; GT_OPT_MATMUL_BF16)
#include "engine_internal.hip.h"

using namespace gt;
// ------------------------------------------------------------------------------------------
// bf16-storage products (gemm_bf16s.hip.h; GT_OPT_MATMUL_BF16)
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int EPI, int AMODE>
static int launch_gemm_b16_t(GemmB16Args g, int nslab, hipStream_t s) {
  const size_t lds = gemm_b16_lds_bytes<BM, BN>();
  CHK(ensure_dyn_lds((const void*)gemm_b16_kernel<BM, BN, EPI, AMODE>, lds));
  g.n_tiles_m = cdiv(g.M, BM);
  g.n_tiles_n = cdiv(g.N, BN);
  const int grid = g.n_tiles_m * g.n_tiles_n * nslab;
  if (grid <= 0) return GT_OK;
  GemmProfiler::Rec rec;
  if (g_prof.wants(g.epi)) {
    rec.kind = g.epi; rec.bn = BN; rec.am = -1; rec.flops = 2.0 * g.M * g.N * g.K;
    rec.bytes = 2.0 * ((double)g.M * g.K + (double)g.K * g.N) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N +
                (g.CbT ? 2.0 : 0.0) * g.M * g.N + ((g.epi == B16_BWD_DATA && g.act != ACT_NONE) ? 2.0 * g.M * g.N : 0.0);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_b16_kernel<BM, BN, EPI, AMODE>), dim3(grid), dim3(GEMM_THREADS), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.wants(g.epi)) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
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
  if (g_prof.wants(g.epi)) {
    rec.kind = g.epi; rec.bn = T; rec.am = -1; rec.flops = 2.0 * g.M * g.N * g.K;
    rec.bytes = 2.0 * ((double)g.M * g.K + (double)g.K * g.N) + (g.C ? 4.0 : 0.0) * g.M * g.N + (g.Cb ? 2.0 : 0.0) * g.M * g.N +
                (g.CbT ? 2.0 : 0.0) * g.M * g.N + ((g.epi == B16_BWD_DATA && g.act != ACT_NONE) ? 2.0 * g.M * g.N : 0.0);
    rec.e0 = g_prof.get(); rec.e1 = g_prof.get();
    HIPCHK(hipEventRecord(rec.e0, s));
  }
  hipLaunchKernelGGL((gemm_b16_dma_kernel<T, T, EPI, AMODE, 2, 2, WGN>), dim3(grid), dim3(64 * 2 * WGN), lds, s, g);
  LAUNCH_CHECK();
  if (g_prof.wants(g.epi)) { HIPCHK(hipEventRecord(rec.e1, s)); g_prof.recs.push_back(rec); }
  return GT_OK;
}
// tile: 0 = chosen here from the shape; 64 / 128 = the caller's choice (the weight gradient sizes its slabs for a tile)
int launch_gemm_b16(const GemmB16Args& g, int nslab, hipStream_t s, int tile) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail(GT_ERR_INVALID, "empty GEMM");
  if ((g.lda & 7) || (g.ldb & 7) || (((uintptr_t)g.A) & 15) || (((uintptr_t)g.B) & 15))
    return fail(GT_ERR_INVALID, "bf16 product: operands must be 16-byte aligned with a row pitch that is a multiple of 8");
  if (g.CbT && ((g.ldcbt & 3) || (((uintptr_t)g.CbT) & 7))) return fail(GT_ERR_INVALID, "bf16 product: transposed result must be 8-byte aligned");
  // 128 x 128 tiles once they still give every CU two workgroups (one resident round), else 64 x 64 (four per CU)
  const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, 128) * nslab;
  const int force_tiles = gt_tuning().b16_tiles;      // measurement knob: 64 / 128 / 256
  const bool big = tile ? tile >= 128
                        : (force_tiles == 64 ? false : (g.epi != B16_SLAB && g.M >= 128 && g.N >= 128 && (force_tiles == 128 || t128 >= 2L * gemm_cu_count())));
  // operand stages by LDS-DMA when no element of a stage needs masking and no row sums ride in the loader
  const bool dma_on = gt_tuning().b16_dma != 0;
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
GemmB16Args b16_args() {
  GemmB16Args g;
  memset(&g, 0, sizeof(g));
  g.drop.mode = DROP_NONE; g.drop.scale = 1.f;
  return g;
}
// [rows][ld_in] float32 / bf16  ->  bf16 [rows][ldo] and / or its transpose [cols][ldt] (+ per-column sums -> colsum, the
// bias gradient of a dZ that no product wrote)
template <typename TIN>
static int cast_transpose_t(const TIN* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
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
int cast_transpose(const float* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
                   float* colsum, bool colsum_accumulate, Scratch* colp, hipStream_t s) {
  return cast_transpose_t<float>(in, ld_in, rows, cols, out, ldo, outT, ldt, colsum, colsum_accumulate, colp, s);
}
int cast_transpose(const __bf16* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
                   float* colsum, bool colsum_accumulate, Scratch* colp, hipStream_t s) {
  return cast_transpose_t<__bf16>(in, ld_in, rows, cols, out, ldo, outT, ldt, colsum, colsum_accumulate, colp, s);
}
// dW (+)= dZT . XT^T over the frame dimension (K = rows), db (+)= row sums of dZT; split into float32 slabs, fixed-order combine
// defer (optional, fused single-GPU step): the slabs go to the network's pool and the combine is only recorded; all
// recorded combines of a network run as ONE launch in front of its optimizer step (slab_defer_flush).
int weight_grad_b16(const __bf16* dZT, long lddzt, const __bf16* XT, long ldxt, long rows, int out, int in, float* dW, float* db,
                           bool accumulate, Scratch& slabs, hipStream_t s, SlabDefer* defer) {
  // 128 x 128 tiles (two workgroups per CU) once the matrix has at least 16 of them, with the slab count that fills whole
  // rounds of 2 x CUs workgroups best (r = 1 .. 3 rounds; tools/gemm_b16_sweep: 1024 x 3072 over 32 768 frames 450 us with
  // 64 x 64 tiles -> 246 us with 8 slabs of 128 x 128; 512 x 2048: 124 -> 76 us); 64 x 64 (four per CU) below that
  const int force_wg_tile = gt_tuning().b16_wg_tile;     // measurement knob: 64 / 128 / 256
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

