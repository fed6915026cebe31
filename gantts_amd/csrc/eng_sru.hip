// libgantts_hip.so -- recurrent generator (GT_ARCH_SRU)
#include "engine_internal.hip.h"
#include "sru_cs_kernels.hip.h"
#include "sru_kernels.hip.h"
using namespace gt;
// ------------------------------------------------------------------------------------------
// recurrent generator (GT_ARCH_SRU)
// ------------------------------------------------------------------------------------------
static void sru_keys(gt_engine* e, int layer, int which, uint32_t* k0, uint32_t* k1) {
  // data parallel: the masks are per (sequence, column); the kernels count sequences globally (SruArgs::seq_mul / seq_add), so a
  // world-k run draws the whole minibatch's masks of a world-1 run -- no rank in the key
  const uint64_t site = e->step_counter * 64ULL + 40 + (uint64_t)(layer * 2 + which);
  *k0 = (uint32_t)(e->seed ^ (site * 0x9E3779B97F4A7C15ULL));
  *k1 = (uint32_t)((e->seed >> 32) ^ (site >> 7) ^ 0x5A5A5A5Au) + (uint32_t)site;
}
static uint32_t drop_thresh(float p) {
  const double th = (double)p * 4294967296.0;
  return th >= 4294967295.0 ? 4294967295u : (uint32_t)th;
}

static SruArgs sru_args(gt_engine* e, const Net& G, int l, int B, int T, const float* in, int ld_in) {
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs;
  const SruLayerP& L = G.sru[l];
  SruArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.k = L.k; a.act = G.d.use_relu ? SRU_RELU : SRU_TANH;
  a.U = e->s_u[l].as<float>(); a.ldu = ncols * L.k;
  a.x = in; a.ldx = ld_in;
  a.bias = L.b;
  a.h = e->s_h[l].as<float>(); a.c = e->s_c[l].as<float>();
  a.seq_mul = e->dp_world; a.seq_add = e->dp_rank;
  if (G.training && G.d.dropout > 0.f && l + 1 < G.d.num_hidden) {   // the last layer has dropout 0 (SRU.__init__)
    a.use_mask = 1; a.keep_scale = 1.f / (1.f - G.d.dropout); a.thresh = drop_thresh(G.d.dropout);
    sru_keys(e, l, 1, &a.key0, &a.key1);
    a.mask_buf = G.inj[0][2 * l + 1];                                // gt_set_dropout_mask(G, 0, 2*l + 1): [B][ncols]
  }
  return a;
}

// the scans with loader waves (sru_kernels.hip.h); GT_SRU_LW=0 selects the one-wave kernels (A/B reference, bit-identical results)
// (read at every launch: the A/B test flips it between two steps of one process)
static bool sru_loader_waves() { return gt_tuning().sru_lw != 0; }
// GT_SRU_LW=2 (default): the cooperative block scans (sru_cs_kernels.hip.h): every wave of a workgroup loads AND walks eight frames
// of a block, the waves' composites are combined through LDS.  Eight waves per 64 columns up to two workgroups per CU (cfg4's B = 16, the
// hparams-default generator's B = 32), four beyond.
static bool sru_coop() { return gt_tuning().sru_lw >= 2; }
static int sru_coop_waves(long B, int ncols) {
  const int forced = gt_tuning().sru_cs_waves;      // (tests: both instantiations on every shape)
  if (forced == 4 || forced == 8) return forced;
  return cdiv(B * ncols, 64) <= 2 * gemm_cu_count() ? 8 : 4;      // (measured: cfg4 B = 16 7.39 vs 7.85 ms; hparams-default generator B = 32, T = 1024 8.07 vs 8.32 ms)
}
// Measured and dropped (round 4, gpurun_out/r4k): 32 columns per workgroup (twice the recurrence waves per CU, half of every wave
// idle) for the shapes that give fewer than three 64-column workgroups per CU -- cfg4 (B = 16, T = 2048) 11.36 vs 10.84 ms, the
// hparams-default generator at B = 32 9.49 vs 9.32 ms: the scan is not bound by one wave's per-frame latency.
static bool sru_b16(const gt_engine* e) { return e->matmul_bf16 && (e->net[GT_ROLE_G].d.hidden_dim & 7) == 0; }
int sru_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs;
  const float* in = x;
  int ld_in = G.d.in_dim;
  // GT_OPT_MATMUL_BF16: the (dropped) layer inputs go through bf16 images in both orientations, W through bf16 shadows in
  // both orientations: U = xin . WT^T, dW = xinT . dUT^T, d in = dU . W^T are all the k-contiguous bf16 product
  const bool b16 = sru_b16(e);
  const bool want_t = G.d.grads != nullptr;
  const int Lc_ = G.d.num_hidden;
  if (b16) {
    e->s_in_b.resize(Lc_ + 1); e->ssh.resize(Lc_ + 1);
    for (int l = 0; l <= Lc_; ++l) {
      LinShadow& w = e->ssh[l];
      const float* W = l < Lc_ ? G.sru[l].W : G.last.W;
      const int rows = l < Lc_ ? G.sru[l].in : G.last.out, cols = l < Lc_ ? ncols * G.sru[l].k : G.last.in;
      w.ldw = pad8(cols); w.ldwt = pad8(rows);
      CHK(w.w.ensure((size_t)rows * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)cols * w.ldwt * 2 + 64));
      CHK(cast_transpose(W, cols, rows, cols, w.w.as<__bf16>(), w.ldw, w.wt.as<__bf16>(), w.ldwt, nullptr, false, &e->colp, s));
    }
  }
  // bf16 storage + cooperative scans: layer l's scan writes the bf16 input images of the product behind it (layer l + 1's U product, or
  // hidden2out) itself -- with that layer's variational input dropout applied -- so the multiplier tables of ALL layers are drawn first
  const bool rdrop_all = G.training && G.d.rnn_dropout > 0.f;
  const bool fold = b16 && sru_coop() && T % 8 == 0 && H % 64 == 0 && ((long)B * ncols) % 64 == 0;
  if (rdrop_all) {
    for (int l = 0; l < G.d.num_hidden; ++l) {
      const SruLayerP& L = G.sru[l];
      CHK(e->s_xmask[l].ensure((size_t)B * L.in * sizeof(float)));
      uint32_t k0, k1;
      sru_keys(e, l, 0, &k0, &k1);
      hipLaunchKernelGGL(sru_input_mask_kernel, dim3(cdiv((long)B * L.in, 256)), dim3(256), 0, s, e->s_xmask[l].as<float>(), B, L.in,
                         1.f / (1.f - G.d.rnn_dropout), drop_thresh(G.d.rnn_dropout), k0, k1,
                         (const float*)G.inj[0][2 * l], e->dp_world, e->dp_rank);     // gt_set_dropout_mask(G, 0, 2*l): [B][n_in]
      LAUNCH_CHECK();
    }
  }
  bool img_ready = false;       // the current layer's input images were written by the scan underneath
  for (int l = 0; l < G.d.num_hidden; ++l) {
    const SruLayerP& L = G.sru[l];
    CHK(e->s_u[l].ensure((size_t)N * ncols * L.k * sizeof(float)));
    CHK(e->s_h[l].ensure((size_t)N * ncols * sizeof(float)));
    CHK(e->s_c[l].ensure((size_t)N * ncols * sizeof(float)));
    const float* xin = in;
    int ld_xin = ld_in;
    const bool rdrop = rdrop_all;      // (variational input dropout, mask shared over time: the multipliers [B][n_in] were drawn above)
    if (rdrop && !b16) {      // float32 products read a dropped float32 copy
      CHK(e->s_xdrop[l].ensure((size_t)N * L.in * sizeof(float)));
      hipLaunchKernelGGL(sru_input_dropout_kernel, dim3(cdiv(N * L.in, 256)), dim3(256), 0, s, in, ld_in, e->s_xdrop[l].as<float>(),
                         L.in, B, T, L.in, (const float*)e->s_xmask[l].as<float>());
      LAUNCH_CHECK();
      xin = e->s_xdrop[l].as<float>();
      ld_xin = L.in;
    }
    if (b16) {
      B16Img& I = e->s_in_b[l];
      CHK(I.ensure(N, L.in, want_t));
      if (img_ready) {
        // written by the scan of the layer underneath (SruArgs::nx_*): no cast pass
      } else if (rdrop) {            // bf16 products: dropout rides in the cast, the dropped input exists as bf16 images only
        const SeqDropSrc src{in, ld_in, e->s_xmask[l].as<float>(), T, L.in};
        hipLaunchKernelGGL(seqdrop_cast_transpose_kernel, dim3(cdiv(N, 64), cdiv(L.in, 64)), dim3(256), 0, s, src, N, L.in, I.r(), I.ld,
                           want_t ? I.t() : (__bf16*)nullptr, I.ldt);
        LAUNCH_CHECK();
      } else {
        CHK(cast_transpose(xin, ld_xin, N, L.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
      }
      GemmB16Args g = b16_args();
      g.A = I.r(); g.lda = I.ld; g.B = e->ssh[l].wt.as<__bf16>(); g.ldb = e->ssh[l].ldwt;     // WT [ncols*k][n_in]: k = n_in contiguous
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.epi = B16_FWD; g.act = ACT_NONE; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      CHK(launch_gemm_b16(g, 1, s));
    } else if ((L.in & 3) == 0 && N >= 4096) {
      // U = xin W with W (n_in, ncols*k): the k-contiguous (NT) product runs at 146 TFLOP/s on these shapes, the n-contiguous (NN)
      // one at 114 (profiles/r03_sru_fp32_summary.md: 1.41 vs 1.80 ms per layer) -- multiply by a transposed copy of W, re-made
      // from the caller's parameter buffer before every pass (12 MB, ~10 us)
      CHK(e->s_wt[l].ensure((size_t)ncols * L.k * L.in * sizeof(float)));
      hipLaunchKernelGGL(transpose_f32_kernel, dim3(cdiv(ncols * L.k, 32), cdiv(L.in, 32)), dim3(256), 0, s, L.W, L.in, ncols * L.k, ncols * L.k,
                         e->s_wt[l].as<float>(), L.in);
      LAUNCH_CHECK();
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = xin; g.lda = ld_xin; g.B = e->s_wt[l].as<float>(); g.ldb = L.in; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.act = ACT_NONE; g.drop = no_drop();
      CHK(launch_gemm(GEMM_NT, g, 1, s));
    } else {  // U = xin W   (W is (n_in, ncols*k): n-contiguous rows -> NN orientation)
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = xin; g.lda = ld_xin; g.B = L.W; g.ldb = ncols * L.k; g.C = e->s_u[l].as<float>(); g.ldc = ncols * L.k;
      g.M = (int)N; g.N = ncols * L.k; g.K = L.in; g.act = ACT_NONE; g.drop = no_drop();
      CHK(launch_gemm(GEMM_NN, g, 1, s));
    }
    SruArgs a = sru_args(e, G, l, B, T, in, ld_in);
    img_ready = false;
    if (fold) {       // the images of the product behind this layer: layer l + 1's input (its dropout applied), or hidden2out's
      const int nl = l + 1;
      const int n_in_next = nl < Lc_ ? G.sru[nl].in : G.last.in;
      if (n_in_next == ncols) {
        B16Img& NI = e->s_in_b[nl];
        CHK(NI.ensure(N, ncols, want_t));
        a.nx_b = NI.r(); a.ld_nxb = NI.ld; a.nx_bt = want_t ? NI.t() : (__bf16*)nullptr; a.ld_nxbt = NI.ldt;
        a.nx_mul = (nl < Lc_ && rdrop_all) ? e->s_xmask[nl].as<float>() : (const float*)nullptr;
        img_ready = true;
      }
    }
    if (sru_coop()) {
      const int grid = cdiv((long)B * ncols, 64);
      const bool w8 = sru_coop_waves(B, ncols) == 8;
      if (img_ready) {
        if (w8) { CHK(ensure_dyn_lds((const void*)sru_fwd_cs_kernel<8, true>, sru_fwd_cs_lds<8>(true)));
                  hipLaunchKernelGGL((sru_fwd_cs_kernel<8, true>), dim3(grid), dim3(512), sru_fwd_cs_lds<8>(true), s, a); }
        else hipLaunchKernelGGL((sru_fwd_cs_kernel<4, true>), dim3(grid), dim3(256), sru_fwd_cs_lds<4>(true), s, a);
      } else if (w8) hipLaunchKernelGGL((sru_fwd_cs_kernel<8, false>), dim3(grid), dim3(512), sru_fwd_cs_lds<8>(), s, a);
      else hipLaunchKernelGGL((sru_fwd_cs_kernel<4, false>), dim3(grid), dim3(256), sru_fwd_cs_lds<4>(), s, a);
    } else if (sru_loader_waves()) {
      CHK(ensure_dyn_lds((const void*)sru_fwd_lw_kernel, sru_fwd_lw_lds()));
      hipLaunchKernelGGL(sru_fwd_lw_kernel, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_fwd_lw_lds(), s, a);
    } else {
      hipLaunchKernelGGL(sru_fwd_kernel, dim3(cdiv((long)B * ncols, SRU_THREADS)), dim3(SRU_THREADS), 0, s, a);
    }
    LAUNCH_CHECK();
    in = e->s_h[l].as<float>();
    ld_in = ncols;
  }
  if (b16) {
    B16Img& I = e->s_in_b[Lc_];
    CHK(I.ensure(N, G.last.in, want_t));
    if (!img_ready)      // (else: written by the last layer's scan)
      CHK(cast_transpose(in, ld_in, N, G.last.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
    GemmB16Args g = b16_args();
    g.A = I.r(); g.lda = I.ld; g.B = e->ssh[Lc_].w.as<__bf16>(); g.ldb = e->ssh[Lc_].ldw;    // hidden2out.weight (out, ncols): k = ncols contiguous
    g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
    g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  return linear_forward(in, ld_in, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in, G.last.out,
                        G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

int sru_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, ncols = H * dirs, Do = G.d.out_dim, Lc = G.d.num_hidden;
  const bool acc = G.grads_dirty;
  int kmax = 3, inmax = ncols;
  for (auto& L : G.sru) { kmax = std::max(kmax, L.k); inmax = std::max(inmax, L.in); }
  CHK(e->l_dout.ensure((size_t)2 * N * std::max(ncols, inmax) * sizeof(float)));
  CHK(e->s_du.ensure((size_t)N * ncols * kmax * sizeof(float)));
  CHK(e->s_dx.ensure((size_t)2 * N * ncols * sizeof(float)));     // highway gradients of two consecutive layers (read by the layer underneath)
  CHK(e->s_dbias.ensure((size_t)B * 2 * ncols * sizeof(float)));
  float* dh = e->l_dout.as<float>();
  float* dh_other = dh + (size_t)N * std::max(ncols, inmax);
  const bool b16 = sru_b16(e) && (int)e->s_in_b.size() == Lc + 1 && (int)e->ssh.size() == Lc + 1;
  if (b16) {
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose(gy, Do, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->s_in_b[Lc];
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, ncols, G.last.dW, G.last.db, acc, e->slabs, s));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * ncols + Do, s));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = e->ssh[Lc].wt.as<__bf16>(); g.ldb = e->ssh[Lc].ldwt;   // hidden2out.weight^T [ncols][Do]
    g.M = (int)N; g.N = ncols; g.K = Do; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dh; g.ldc = ncols;
    CHK(launch_gemm_b16(g, 1, s));
  } else {
  CHK(linear_backward_weight(gy, Do, e->s_h[Lc - 1].as<float>(), ncols, N, Do, ncols, G.last.dW, G.last.db, acc, e->slabs, e->colp, s));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * ncols + Do, s));
  CHK(linear_backward_data(gy, Do, G.last.W, G.last.in, 0, dh, ncols, N, Do, ncols, ACT_NONE, nullptr, 0, no_drop(), s));
  }
  for (int l = Lc - 1; l >= 0; --l) {
    const SruLayerP& L = G.sru[l];
    const float* in = l == 0 ? x : e->s_h[l - 1].as<float>();
    const int ld_in = l == 0 ? G.d.in_dim : ncols;
    const bool rdrop = G.training && G.d.rnn_dropout > 0.f;
    SruArgs a = sru_args(e, G, l, B, T, in, ld_in);
    a.dh = dh; a.dU = e->s_du.as<float>();
    // k == 3: the highway gradient goes straight to the layer input.  Without input dropout it is
    // written into the next dh buffer and the GEMM below accumulates onto it.
    auto dx_of = [&](int layer) { return e->s_dx.as<float>() + (size_t)(layer & 1) * N * ncols; };
    float* dx_res = L.k == 3 ? (rdrop ? dx_of(l) : dh_other) : nullptr;
    a.dx = dx_res; a.lddx = ncols;
    if (rdrop && l + 1 < Lc) {      // dh is the raw dU.W^T of the layer above: its input dropout and highway gradient are applied by the scan
      if (G.sru[l + 1].in != ncols || !e->s_xmask[l + 1].p)
        return fail(GT_ERR_STATE, "SRU backward: layer %d's input-dropout table is missing or not %d wide", l + 1, ncols);
      a.up_mul = e->s_xmask[l + 1].as<float>();
      a.up_add = G.sru[l + 1].k == 3 ? dx_of(l + 1) : nullptr;
      a.ld_up_add = ncols;
    }
    a.dbias_part = e->s_dbias.as<float>();
    // bf16 storage with loader waves, whole blocks of 8 frames, whole workgroups inside one sequence and one direction: dU leaves
    // the scan as the bf16 images the two products read (no float32 dU, no cast pass)
    const bool du_b16 = b16 && sru_loader_waves() && T % 8 == 0 && H % 64 == 0;
    if (du_b16) {
      B16Img& DU = e->s_du_b;
      CHK(DU.ensure(N, ncols * L.k, true));
      a.dU = nullptr; a.dU_b = DU.r(); a.ld_dub = DU.ld; a.dU_bt = DU.t(); a.ld_dubt = DU.ldt;
    }
    if (sru_coop()) {
      const int grid = cdiv((long)B * ncols, 64);
      const bool w8 = sru_coop_waves(B, ncols) == 8;
#define GT_SRU_CS_LAUNCH(NW_, B16_)                                                                                           \
      do {                                                                                                                     \
        CHK(ensure_dyn_lds((const void*)sru_bwd_cs_kernel<NW_, B16_>, sru_bwd_cs_lds<NW_>(B16_)));                             \
        hipLaunchKernelGGL((sru_bwd_cs_kernel<NW_, B16_>), dim3(grid), dim3(64 * NW_), sru_bwd_cs_lds<NW_>(B16_), s, a);        \
      } while (0)
      if (du_b16) { if (w8) GT_SRU_CS_LAUNCH(8, true); else GT_SRU_CS_LAUNCH(4, true); }
      else { if (w8) GT_SRU_CS_LAUNCH(8, false); else GT_SRU_CS_LAUNCH(4, false); }
#undef GT_SRU_CS_LAUNCH
    } else if (du_b16) {
      CHK(ensure_dyn_lds((const void*)sru_bwd_lw_kernel<true>, sru_bwd_lw_lds()));
      hipLaunchKernelGGL(sru_bwd_lw_kernel<true>, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_bwd_lw_lds(), s, a);
    } else if (sru_loader_waves()) {
      CHK(ensure_dyn_lds((const void*)sru_bwd_lw_kernel<false>, sru_bwd_lw_lds()));
      hipLaunchKernelGGL(sru_bwd_lw_kernel<false>, dim3(cdiv((long)B * ncols, 64)), dim3(SRU_LW_THREADS), sru_bwd_lw_lds(), s, a);
    } else {
      hipLaunchKernelGGL(sru_bwd_kernel, dim3(cdiv((long)B * ncols, SRU_THREADS)), dim3(SRU_THREADS), 0, s, a);
    }
    LAUNCH_CHECK();
    hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(2 * ncols, 64)), dim3(1024), 0, s, e->s_dbias.as<float>(), (long)2 * ncols, B,
                       2 * ncols, L.db, acc ? 1 : 0);
    LAUNCH_CHECK();
    const float* xin = rdrop ? e->s_xdrop[l].as<float>() : in;
    const int ld_xin = rdrop ? L.in : ld_in;
    if (b16) {
      // dU -> bf16 image in both orientations (one pass); dW = xinT . dUT^T over the frames, d in = dU . W^T
      B16Img& DU = e->s_du_b;
      if (!du_b16) {
        CHK(DU.ensure(N, ncols * L.k, true));
        CHK(cast_transpose(e->s_du.as<float>(), ncols * L.k, N, ncols * L.k, DU.r(), DU.ld, DU.t(), DU.ldt, nullptr, false, &e->colp, s));
      }
      B16Img& I = e->s_in_b[l];
      CHK(weight_grad_b16(I.t(), I.ldt, DU.t(), DU.ldt, N, L.in, ncols * L.k, L.dW, nullptr, acc, e->slabs, s));
    } else {
    // dW = xin^T dU   (TN: A = xin is m-contiguous over n_in, B = dU)
    CHK(linear_backward_weight(xin, ld_xin, e->s_du.as<float>(), ncols * L.k, N, L.in, ncols * L.k, L.dW, nullptr, acc, e->slabs,
                               e->colp, s));
    }
    CHK(comm_grads_ready(e, GT_ROLE_G, L.dW, (long)L.in * ncols * L.k + 2L * ncols, s));
    if (l > 0) CHK(comm_flush(e, GT_ROLE_G, s));
    if (l > 0) {
      if (b16) {
        GemmB16Args g = b16_args();
        g.A = e->s_du_b.r(); g.lda = e->s_du_b.ld; g.B = e->ssh[l].w.as<__bf16>(); g.ldb = e->ssh[l].ldw;    // W [n_in][ncols*k]: k contiguous
        g.M = (int)N; g.N = L.in; g.K = ncols * L.k; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dh_other; g.ldc = L.in;
        g.accumulate = (L.k == 3 && !rdrop) ? 1 : 0;
        CHK(launch_gemm_b16(g, 1, s));
      } else {
      // d in = (dU W^T) (.) mask_in + highway term     (NT: B[n = i][k = c] = W[i*ldw + c])
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = e->s_du.as<float>(); g.lda = ncols * L.k; g.B = L.W; g.ldb = ncols * L.k; g.C = dh_other; g.ldc = L.in;
      g.M = (int)N; g.N = L.in; g.K = ncols * L.k; g.act = ACT_NONE; g.drop = no_drop();
      g.accumulate = (L.k == 3 && !rdrop) ? 1 : 0;
      CHK(launch_gemm(GEMM_NT, g, 1, s));
      }
      std::swap(dh, dh_other);         // (with input dropout: finished by the scan of layer l - 1, SruArgs::up_mul / up_add)
    }
    // l == 0: no gradient with respect to the network input is produced on this path (nothing upstream of the generator
    // takes one: x is data, train.py:542).  NOTE for anything that wants to read `dh` between layers: with rnn_dropout it
    // is the RAW dU.W^T -- the input-dropout mask and the k = 3 highway term are applied by the next scan's loads.
  }
  return GT_OK;
}

