// libgantts_hip.so -- dispatch of the float32 MFMA GEMM family (gemm_f32.hip.h): tile choice, pair launches, weight-gradient slabs
#include "engine_internal.hip.h"

using namespace gt;
// the kernels are instantiated per operand orientation in their own translation units (eng_gemm_f32_{nt,nn,tn,pair}.hip)
int launch_gemm_nt(const GemmArgs& g, int bm, int bn, hipStream_t s);
int launch_gemm_nn(const GemmArgs& g, int bm, int bn, hipStream_t s);
int launch_gemm_tn(const GemmArgs& g, int bm, int bn, int nslab, hipStream_t s);
int launch_gemm_pair(const GemmArgs& nn_in, const GemmArgs& tn_in, int nslab, hipStream_t s);
int launch_gemm_tn_pair(const GemmArgs& g1_in, const GemmArgs& g2_in, int nslab1, int nslab2, hipStream_t s, const GemmArgs* rider, bool* rode);
// ------------------------------------------------------------------------------------------
// GEMM dispatch
// ------------------------------------------------------------------------------------------
int gemm_cu_count() {
  static std::map<int, int> cus;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  auto it = cus.find(dev);
  if (it != cus.end()) return it->second;
  hipDeviceProp_t prop;
  const int n = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
  cus[dev] = n;
  return n;
}
// products of the GEMM family: f32 MFMA (default) or bf16 MFMA with f32 accumulation (GT_OPT_MATMUL_BF16; set per engine
// entry point for the launches it issues on this thread)
thread_local int tl_gemm_prec = PREC_F32;

static int pick_bn(int N) { return (cdiv(N, 64) * 64 < cdiv(N, 128) * 128) ? 64 : 128; }

// Tile choice (measured with tools/gemm_tile_sweep.hip on the cfg2 shapes, 16-byte-loadable operands): 64x64 tiles beat
// 128x128 / 64x128 / 128x64 on every forward / backward-data shape of the step (16384x512x512: 77 vs 80 us; 32768x256x256:
// 44.7 vs 47.9; 16384x256x256: 25.1 vs 27.4 / 31.5) -- four tiles per CU drain their epilogues under each other's K loops,
// and the 2x operand re-reads come out of the XCD's L2.  Operands that need the 4-byte loader keep the larger tiles
// (their loader is the bottleneck, and a tile re-read costs 4x the instructions).
static int gemm_tile_mode() { return gt_tuning().gemm_tiles_big; }   // measurement knob: GT_GEMM_TILES=big restores the residency model for every launch
static bool gemm_unaligned_ok() { return gt_tuning().gemm_unaligned != 0; }   // measurement knob
bool gemm_vec_ok(const float* p, int ld, bool k_contiguous) {
  if (k_contiguous && gemm_unaligned_ok()) return true;
  return (ld % 4 == 0) && (((uintptr_t)p) % 16 == 0);
}
// NT / NN results (and the producer's activation the NN epilogue reads) leave / arrive row-wise, 16 bytes per lane
bool gemm_wide_store_ok(int kind, const GemmArgs& g) {
  if (kind == GEMM_TN) return false;
  if (gemm_unaligned_ok()) return true;
  return (g.ldc % 4 == 0) && (((uintptr_t)g.C) % 16 == 0) &&
         (kind != GEMM_NN || g.act == ACT_NONE || ((g.ldh % 4 == 0) && (((uintptr_t)g.H) % 16 == 0)));
}
bool gemm_small_tiles_ok() { return gemm_tile_mode() == 0; }   // f32 and bf16 products alike (bf16: cfg2 1.30 -> 1.21 ms, SRU 32.0 -> 28.2 ms)


int launch_gemm(int kind, const GemmArgs& g, int nslab, hipStream_t s) {
  if (g.M <= 0 || g.N <= 0 || g.K <= 0) return fail(GT_ERR_INVALID, "empty GEMM");
  const int bn = pick_bn(g.N);
  const bool vec = gemm_vec_ok(g.A, g.lda, kind != GEMM_TN) && gemm_vec_ok(g.B, g.ldb, kind == GEMM_NT);
  if (kind != GEMM_TN && vec && g.M > 64 && gemm_small_tiles_ok())
    return kind == GEMM_NT ? launch_gemm_nt(g, 64, 64, s) : launch_gemm_nn(g, 64, 64, s);
  // Otherwise tile height by a residency model: 128-row tiles keep 2 workgroups per CU resident (512 at once),
  // 64-row tiles 3-4 (LDS-limited: 768 with 128 columns, 1024 with 64) at ~0.55x the work each.
  // Cost = resident rounds x work per tile; e.g. 384 tiles (187-wide output) or 1024 tiles (2N x 483)
  // finish sooner as 64-row tiles, exactly 512 tiles do not.
  bool small = false;
  if (kind != GEMM_TN && g.M > 64) {
    const long t128 = (long)cdiv(g.M, 128) * cdiv(g.N, bn), t64 = (long)cdiv(g.M, 64) * cdiv(g.N, bn);
    const double c128 = (double)cdiv(t128, 512), c64 = 0.55 * (double)cdiv(t64, bn == 64 ? 1024 : 768);
    small = c64 < c128;
  }
  switch (kind) {
    case GEMM_NT:
      if (small) return bn == 64 ? launch_gemm_nt(g, 64, 64, s) : launch_gemm_nt(g, 64, 128, s);
      return bn == 64 ? launch_gemm_nt(g, 128, 64, s) : launch_gemm_nt(g, 128, 128, s);
    case GEMM_NN:
      if (small) return bn == 64 ? launch_gemm_nn(g, 64, 64, s) : launch_gemm_nn(g, 64, 128, s);
      return bn == 64 ? launch_gemm_nn(g, 128, 64, s) : launch_gemm_nn(g, 128, 128, s);
    default:
      if (g.n_tiles_m == 64) return launch_gemm_tn(g, 64, 64, nslab, s);     // linear_backward_weight's choice (tile height in n_tiles_m)
      return bn == 64 ? launch_gemm_tn(g, 128, 64, nslab, s) : launch_gemm_tn(g, 128, 128, nslab, s);
  }
}

DropoutSpec no_drop() {
  DropoutSpec d;
  memset(&d, 0, sizeof(d));
  d.mode = DROP_NONE;
  d.scale = 1.f;
  return d;
}

// Y = act(X W^T + b)
int linear_forward(const float* X, int ldx, const float* W, int ldw, const float* b, float* Y, int ldy,
                          long rows, int in, int out, int act, const DropoutSpec& drop, hipStream_t s) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = X; g.lda = ldx; g.B = W; g.ldb = ldw; g.C = Y; g.ldc = ldy;
  g.M = (int)rows; g.N = out; g.K = in; g.bias = b; g.act = act; g.drop = drop;
  return launch_gemm(GEMM_NT, g, 1, s);
}
// dX = (dZ W[:, col0:col0+ncols]) (.) f'(H)
GemmArgs backward_data_args(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                                   long rows, int out, int ncols, int act_prev, const float* H, int ldh, const DropoutSpec& drop) {
  GemmArgs g;
  memset(&g, 0, sizeof(g));
  g.A = dZ; g.lda = lddz; g.B = W + col0; g.ldb = ldw; g.C = dX; g.ldc = lddx;
  g.M = (int)rows; g.N = ncols; g.K = out; g.act = act_prev; g.H = H; g.ldh = ldh; g.drop = drop;
  return g;
}
int linear_backward_data(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                                long rows, int out, int ncols, int act_prev, const float* H, int ldh,
                                const DropoutSpec& drop, hipStream_t s) {
  return launch_gemm(GEMM_NN, backward_data_args(dZ, lddz, W, ldw, col0, dX, lddx, rows, out, ncols, act_prev, H, ldh, drop), 1, s);
}
// Pair launch (gemm_pair_kernel): the same layer's backward-data product rides in the weight gradient's launch when both
// run on 64x64 tiles with 16-byte loadable operands.  GT_GEMM_PAIR=0 keeps them apart (measurement switch).
static bool gemm_pair_enabled() { return gt_tuning().gemm_pair != 0; }
static bool gemm_pair_ok(const GemmArgs& nn) {
  return gemm_pair_enabled() && gemm_small_tiles_ok() && nn.M > 64 && gemm_vec_ok(nn.A, nn.lda, true) && gemm_vec_ok(nn.B, nn.ldb);
}


// dW (+)= dZ^T X ; db (+)= colsum(dZ)   -- split over the frame dimension, fixed-order combine.
// The bias gradient rides along in the weight-gradient kernel (column sums of its A operand).
// Deferred combines (SlabDefer): the partial slabs of every layer go to their own piece of a pool and the combine is
// only RECORDED; slab_defer_flush() runs all recorded combines in one launch.  Used by the fused single-GPU step, where
// nothing reads a weight gradient between a network's backward pass and its optimizer step.
int slab_defer_flush(SlabDefer& d, hipStream_t s) {
  if (d.jobs.n > 0) {
    hipLaunchKernelGGL(slab_reduce_multi_kernel, dim3(d.blocks), dim3(256), 0, s, d.jobs);
    LAUNCH_CHECK();
  }
  d.jobs.n = 0; d.blocks = 0; d.used = 0;
  return GT_OK;
}
// `ride_along` (optional): the backward-data product of the same layer; if it can share the weight gradient's launch it
// does and *rode is set, otherwise the caller launches it itself.
int linear_backward_weight(const float* dZ, int lddz, const float* X, int ldx, long rows, int out, int in,
                                  float* dW, float* db, bool accumulate, Scratch& slabs, Scratch& colp, hipStream_t s,
                                  SlabDefer* defer, const GemmArgs* ride_along, bool* rode) {
  if (rode) *rode = false;
  if (dW) {
    // 64x64 tiles when both operands take 16-byte loads: the same workgroup count with 4x fewer partial slabs (less slab
    // traffic in the product's epilogue and in the combine: 512x512 over 16384 frames 8 slabs instead of 32)
    const bool t64 = gemm_vec_ok(dZ, lddz) && gemm_vec_ok(X, ldx) && gemm_small_tiles_ok();
    const int bn = t64 ? 64 : pick_bn(in);
    const int tiles = cdiv(out, t64 ? 64 : 128) * cdiv(in, bn);
    const int slab_wgs = gt_tuning().tn_wgs;   // measurement knob
    int nslab = std::max(1, slab_wgs / tiles);   // <= 2 workgroups per CU x 256 CUs: one resident round
    const int max_slab = (int)((rows + 255) / 256);
    if (nslab > max_slab) nslab = max_slab;
    if (nslab < 1) nslab = 1;
    int k_chunk = cdiv(cdiv(rows, nslab), GEMM_BK) * GEMM_BK;
    nslab = cdiv(rows, k_chunk);
    const long slab_stride = (long)out * in;
    const size_t need = (((size_t)nslab * slab_stride + (size_t)nslab * out) * sizeof(float) + 255) & ~(size_t)255;
    const bool can4 = slab_stride % 4 == 0 && ((uintptr_t)dW) % 16 == 0;
    float* slab_base = nullptr;
    if (defer && defer->active && accumulate) { CHK(slab_defer_flush(*defer, s)); defer = nullptr; }   // never two combines of one dW in a launch
    if (defer && defer->active && can4) {
      if (defer->jobs.n == SLAB_MAX_JOBS || defer->used + need > defer->pool.bytes) {
        CHK(slab_defer_flush(*defer, s));                       // recorded combines first, then (maybe) a larger pool
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
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = dZ; g.lda = lddz; g.B = X; g.ldb = ldx; g.C = slab_base; g.ldc = in;
    g.M = out; g.N = in; g.K = (int)rows; g.k_chunk = k_chunk; g.slab_stride = slab_stride;
    g.colsum_slab = db ? bias_slabs : nullptr;
    g.drop = no_drop();
    g.n_tiles_m = t64 ? 64 : 128;        // tile height request (launch_gemm_t overwrites the field with the tile count)
    if (t64 && ride_along && rode && gemm_pair_ok(*ride_along)) {
      CHK(launch_gemm_pair(*ride_along, g, nslab, s));
      *rode = true;
    } else {
      CHK(launch_gemm(GEMM_TN, g, nslab, s));
    }
    if (can4) {
      const int main_blocks = cdiv(slab_stride / 4, 256);
      const int bias_blocks = db ? cdiv(out, 256) : 0;
      if (defer) {
        SlabJob& J = defer->jobs.j[defer->jobs.n++];
        J.slabs = slab_base; J.slab_stride = slab_stride; J.n4 = slab_stride / 4; J.out = dW; J.bslabs = bias_slabs; J.bout = db;
        J.nslab = nslab; J.accumulate = accumulate ? 1 : 0; J.nb = out; J.main_blocks = main_blocks; J.block0 = defer->blocks; J.pad_ = 0;
        defer->blocks += main_blocks + bias_blocks;
        return GT_OK;
      }
      hipLaunchKernelGGL(slab_reduce4_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, s, slab_base, slab_stride, nslab,
                         slab_stride / 4, dW, accumulate ? 1 : 0, (const float*)bias_slabs, out, db, main_blocks);
      LAUNCH_CHECK();
    } else {
      hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(slab_stride, 256)), dim3(256), 0, s, slabs.as<float>(),
                         slab_stride, nslab, slab_stride, dW, accumulate ? 1 : 0);
      LAUNCH_CHECK();
      if (db) {
        hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(out, 64)), dim3(1024), 0, s, bias_slabs, (long)out, nslab, out, db,
                           accumulate ? 1 : 0);
        LAUNCH_CHECK();
      }
    }
  } else if (db) {
    const int rows_per_blk = 128;
    const int nblk = cdiv(rows, rows_per_blk);
    CHK(colp.ensure((size_t)nblk * out * sizeof(float)));
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(nblk, cdiv(out, 64)), dim3(256), 0, s, dZ, lddz, rows, out, rows_per_blk,
                       colp.as<float>());
    LAUNCH_CHECK();
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3(cdiv(out, 256)), dim3(256), 0, s, colp.as<float>(), nblk, out, db,
                       accumulate ? 1 : 0);
    LAUNCH_CHECK();
  }
  return GT_OK;
}


// Weight gradient of a SPLIT first layer (conditioned discriminator: input [x | adv], the same x under the real and the
// generated rows):  dW[:, :cd] = (dZ[:wrap] + dZ[wrap:])^T . x   over `wrap` frames (the loader sums the two halves),
//                   dW[:, cd:] = dZ^T . adv                         over all `rows` frames,      db = column sums of dZ.
// Both products write column blocks of one slab set [nslab][out][cd + Da] in ONE launch; the combine is the usual one.
// xp: x with a 16-byte row pitch (n-contiguous operand), adv: [rows][ld_adv] with a 16-byte pitch.
int linear_backward_weight_split(const float* dZ, int lddz, long rows, long wrap, const float* xp, int ldxp, int cd,
                                 const float* adv, int ld_adv, int Da, int out, float* dW, float* db, bool accumulate,
                                 Scratch& slabs, hipStream_t s, SlabDefer* defer, const GemmArgs* rider, bool* rode) {
  if (rode) *rode = false;
  if (rows != wrap && rows != 2 * wrap) return fail(GT_ERR_INVALID, "split weight gradient: rows must be one or two halves");
  if (!gemm_vec_ok(dZ, lddz) || !gemm_vec_ok(xp, ldxp) || !gemm_vec_ok(adv, ld_adv) || tl_gemm_prec != PREC_F32)
    return fail(GT_ERR_INVALID, "split weight gradient: operands must be 16-byte loadable (float32 products)");
  const int in = cd + Da;
  // Slab count: the adversarial block has cdiv(Da, 64) tile columns against cdiv(cd, 64) of the x block but twice the frames, so
  // its workgroups are the long ones (rows / nslab frames each): enough slabs that one of them is about as long as a quarter of
  // the launch (4 workgroups per CU), dispatched FIRST (gemm_tn_pair_kernel), the x block's workgroups back-fill behind them.
  const int tiles = cdiv(out, 64) * (cdiv(cd, 64) + cdiv(Da, 64));      // both blocks' tiles: one resident round of 4 workgroups per CU
  int nslab = std::max(1, gt_tuning().tn_split_wgs / tiles);
  nslab = std::min<long>(nslab, std::max<long>(1, wrap / 256));
  const int kc1 = cdiv(cdiv(wrap, nslab), GEMM_BK) * GEMM_BK;
  const int ns1 = cdiv(wrap, kc1);
  const int kc2 = cdiv(cdiv(rows, ns1), GEMM_BK) * GEMM_BK;
  const int ns2 = cdiv(rows, kc2);                      // <= ns1
  nslab = ns1;
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
  if (ns2 < ns1)      // (tiny batches) the adversarial block and the bias sums of the trailing slabs are not written by any workgroup
    HIPCHK(hipMemsetAsync(slab_base, 0, need, s));
  GemmArgs g1, g2;
  memset(&g1, 0, sizeof(g1));
  g1.A = dZ; g1.lda = lddz; g1.A2 = rows == 2 * wrap ? dZ + wrap * (long)lddz : nullptr;
  g1.B = xp; g1.ldb = ldxp; g1.C = slab_base; g1.ldc = in;
  g1.M = out; g1.N = cd; g1.K = (int)wrap; g1.k_chunk = kc1; g1.slab_stride = slab_stride; g1.drop = no_drop();
  g2 = g1;
  g2.A2 = nullptr; g2.B = adv; g2.ldb = ld_adv; g2.C = slab_base + cd; g2.N = Da; g2.K = (int)rows; g2.k_chunk = kc2;
  g2.colsum_slab = db ? bias_slabs : nullptr;
  if (g1.A2) {
    CHK(launch_gemm_tn_pair(g1, g2, ns1, ns2, s, rider, rode));
  } else {             // one half only: two plain weight-gradient launches
    g1.n_tiles_m = 64; g2.n_tiles_m = 64;
    CHK(launch_gemm(GEMM_TN, g1, ns1, s));
    CHK(launch_gemm(GEMM_TN, g2, ns2, s));
  }
  if (can4) {
    const int main_blocks = cdiv(slab_stride / 4, 256), bias_blocks = db ? cdiv(out, 256) : 0;
    if (defer) {
      SlabJob& J = defer->jobs.j[defer->jobs.n++];
      J.slabs = slab_base; J.slab_stride = slab_stride; J.n4 = slab_stride / 4; J.out = dW; J.bslabs = bias_slabs; J.bout = db;
      J.nslab = nslab; J.accumulate = accumulate ? 1 : 0; J.nb = out; J.main_blocks = main_blocks; J.block0 = defer->blocks; J.pad_ = 0;
      defer->blocks += main_blocks + bias_blocks;
      return GT_OK;
    }
    hipLaunchKernelGGL(slab_reduce4_kernel, dim3(main_blocks + bias_blocks), dim3(256), 0, s, slab_base, slab_stride, nslab,
                       slab_stride / 4, dW, accumulate ? 1 : 0, (const float*)bias_slabs, out, db, main_blocks);
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
