// libgantts_hip.so -- the G+D step: MLPG, MLP stacks, apply_generator / update_discriminator / update_generator (reference train.py:245-355), plain forward
#include "engine_internal.hip.h"
#include "dstack_f32.hip.h"

using namespace gt;
// ------------------------------------------------------------------------------------------
// MLPG band cache
// ------------------------------------------------------------------------------------------
int ensure_band(gt_engine* e, const float* R, int T, hipStream_t s) {
  MlpgCache& m = e->mlpg;
  ++m.tick;
  for (auto* b : m.entries)
    if (b->R == R && b->T == T) { b->last_use = m.tick; m.cur = b; return GT_OK; }
  const int nW = e->cfg.num_windows;
  // first sight of this (R, T): per-offset maxima -> host, pick the smallest half-width whose outside is negligible
  CHK(m.tmp.ensure((size_t)(2 * T - 1) * sizeof(float)));
  hipLaunchKernelGGL(mlpg_offset_max_kernel, dim3(2 * T - 1), dim3(256), 0, s, R, T, nW, m.tmp.as<float>());
  LAUNCH_CHECK();
  std::vector<float> off(2 * T - 1);
  HIPCHK(hipMemcpyAsync(off.data(), m.tmp.p, off.size() * sizeof(float), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  float peak = 0.f;
  for (float v : off) peak = fmaxf(peak, v);
  if (!(peak > 0.f) || !isfinite(peak)) return fail(GT_ERR_INVALID, "MLPG matrix R is empty or not finite");
  int kb = 0;
  for (int o = -(T - 1); o <= T - 1; ++o)
    if (off[o + T - 1] > 1e-9f * peak) kb = std::max(kb, abs(o));
  if (kb > 63 || (kb > 48 && kb > T / 4))
    return fail(GT_ERR_INVALID, "MLPG matrix R is not banded (half-width %d of T=%d): only window sets whose "
                "R = (W^T W)^-1 W^T decays (hparams.py:22-26) are supported", kb, T);
  MlpgBand* b = nullptr;
  if (m.entries.size() >= MlpgCache::MAX_ENTRIES) {      // recycle the least recently used entry
    size_t lru = 0;
    for (size_t i = 1; i < m.entries.size(); ++i) if (m.entries[i]->last_use < m.entries[lru]->last_use) lru = i;
    b = m.entries[lru];
    HIPCHK(hipStreamSynchronize(s));                      // its band may still be read by queued kernels
  } else {
    b = new MlpgBand();
    m.entries.push_back(b);
  }
  const int nb = 2 * kb + 1;
  b->R = nullptr;
  CHK(b->band.ensure((size_t)T * nW * nb * sizeof(float)));
  hipLaunchKernelGGL(mlpg_extract_band_kernel, dim3(cdiv((long)T * nW * nb, 256)), dim3(256), 0, s, R, T, nW, kb, b->band.as<float>());
  LAUNCH_CHECK();
  b->R = R; b->T = T; b->kb = kb; b->last_use = m.tick;
  m.cur = b;
  return GT_OK;
}
extern "C" int gt_invalidate_mlpg_cache(gt_engine* e) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipDeviceSynchronize());
  e->mlpg.clear();
  return GT_OK;
}

// output frames per workgroup of the MLPG kernels: 32; 16-frame tiles (gt_set_tuning("mlpg_tt", 16) / "mlpg_small16"): twice the workgroups for
// batches whose 32-frame tiles leave CUs empty (a rank's share of a strong-scaling run: B * ceil(T / 32) = 64 workgroups at 4 sequences of
// 512 frames), at (16 + 2 kb) / 16 staged rows per output frame.  (64-frame tiles -- half the halo re-reads, one workgroup per CU instead of two --
// measured no gain in round 4, 1.404 / 1.398 vs 1.393 / 1.398 ms, and left the library in round 6.)
static int mlpg_tile_frames(gt_engine* e, int B, int T) {
  (void)e;
  const int tt = gt_tuning().mlpg_tt;
  if (tt == 16) return 16;
  if (tt == 0 && gt_tuning().mlpg_small16 && (long)B * cdiv(T, 32) * 2 <= gemm_cu_count()) return 16;
  return 32;
}
int mlpg_forward(gt_engine* e, const float* y, int ldy, const int* scol, const int* sstride, int Ds,
                 float* ys, int ldys, int B, int T, hipStream_t s) {
  const int nW = e->cfg.num_windows, kb = e->mlpg.cur->kb;
  auto lds_of = [&](int tt) { return ((size_t)(tt + 2 * kb) * nW * MLPG_CC + (size_t)tt * nW * (2 * kb + 1 + 2 * MLPG_PAD)) * sizeof(float); };
  const int tt = mlpg_tile_frames(e, B, T);
  const size_t lds = lds_of(tt);
  dim3 grid(B * cdiv(T, tt), cdiv(Ds, MLPG_CC));
  const int fpl = gt_tuning().mlpg_fpl;   // frames per lane of the compute phase: 2 measured best (round 4: 4: 24.7 us, 2: 21.8, 1: 26.6; round 5, unrolled tap loops: forward 18.5 / 17.2 / 18.3, backward 24.9 / 19.5 / 20.6)
#define GT_MLPG_FWD(F, TTV) { CHK(ensure_dyn_lds((const void*)mlpg_forward_kernel<F, TTV>, lds)); \
    hipLaunchKernelGGL((mlpg_forward_kernel<F, TTV>), grid, dim3(MLPG_THREADS), lds, s, y, ldy, e->mlpg.cur->band.as<float>(), kb, nW, scol, sstride, Ds, ys, ldys, B, T); }
  if (tt == 16) GT_MLPG_FWD(2, 16)
  else { if (fpl == 1) GT_MLPG_FWD(1, 32) else if (fpl == 2) GT_MLPG_FWD(2, 32) else GT_MLPG_FWD(4, 32) }
#undef GT_MLPG_FWD
  LAUNCH_CHECK();
  return GT_OK;
}
int mlpg_backward(gt_engine* e, const float* gs, int ldgs, const int* scol, const int* sstride, int Ds,
                  float* gy, int ldgy, int B, int T, float mse_w, const float* yhat, const float* ytgt, int ldt,
                  const float* mask, hipStream_t s) {
  const int nW = e->cfg.num_windows, kb = e->mlpg.cur->kb;
  auto lds_of = [&](int tt) { return ((size_t)(tt + 2 * kb) * MLPG_CC + (size_t)(tt + 2 * kb) * nW * (2 * kb + 1 + 2 * MLPG_PAD)) * sizeof(float); };
  const int tt = mlpg_tile_frames(e, B, T);
  const size_t lds = lds_of(tt);
  dim3 grid(B * cdiv(T, tt), cdiv(Ds, MLPG_CC));
  const int fpl = gt_tuning().mlpg_fpl;
#define GT_MLPG_BWD(F, TTV) { CHK(ensure_dyn_lds((const void*)mlpg_backward_kernel<F, TTV>, lds)); \
    hipLaunchKernelGGL((mlpg_backward_kernel<F, TTV>), grid, dim3(MLPG_THREADS), lds, s, gs, ldgs, e->mlpg.cur->band.as<float>(), kb, nW, scol, sstride, Ds, \
                       gy, ldgy, B, T, mse_w, yhat, ytgt, ldt, mask, e->sc()); }
  if (tt == 16) GT_MLPG_BWD(2, 16)
  else { if (fpl == 1) GT_MLPG_BWD(1, 32) else if (fpl == 2) GT_MLPG_BWD(2, 32) else GT_MLPG_BWD(4, 32) }
#undef GT_MLPG_BWD
  LAUNCH_CHECK();
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// network passes
// ------------------------------------------------------------------------------------------
// stacked injected-mask buffer for one layer of a D pass group (real rows then fake rows)
static int stage_injected(gt_engine* e, int role, int layer, const int* passes, int npass, long rows_each, int width,
                          const float** out, hipStream_t s) {
  Net& n = e->net[role];
  *out = nullptr;
  if (!n.training || n.d.dropout <= 0.f) return GT_OK;
  bool any = false, all = true;
  for (int i = 0; i < npass; ++i) { any |= n.inj[passes[i]][layer] != nullptr; all &= n.inj[passes[i]][layer] != nullptr; }
  if (!any) return GT_OK;
  if (!all) return fail(GT_ERR_INVALID, "injected dropout masks must be given for every pass of a step or for none");
  if (npass == 1) { *out = n.inj[passes[0]][layer]; return GT_OK; }
  const size_t per_layer = (size_t)npass * rows_each * width * sizeof(float);
  CHK(e->dmask.ensure(per_layer * n.d.num_hidden));
  float* base = (float*)((char*)e->dmask.p + per_layer * layer);
  for (int i = 0; i < npass; ++i)
    HIPCHK(hipMemcpyAsync(base + (size_t)i * rows_each * width, n.inj[passes[i]][layer], (size_t)rows_each * width * sizeof(float),
                          hipMemcpyDeviceToDevice, s));
  *out = base;
  return GT_OK;
}

// SPLIT first layer of the conditioned discriminator (float32 path).  D sees [x | adv] (train.py:254-256) for the real AND the
// generated rows of a D step with the SAME x, so layer 0 is evaluated as
//     P = x . W[:, :cd]^T + b                      once, over the `wrap` rows of x             (no activation)
//     H = act(adv . W[:, cd:]^T + P[r mod wrap])   over all rows of the pass                   (K = Da = 58 columns)
// instead of one product over a concatenated [2N][cd + Da] image: no image of x is built (109 MB / step at cfg2), the
// x product of a D step is done once instead of twice (- 44 % of the layer's flops), and the weight gradient contracts
// (dZ_real + dZ_generated) with x over N frames (linear_backward_weight_split).  Same sums up to float32 association.
struct FirstSplit {
  const float* x; int ldx; int cd;     // conditioning input: `wrap` rows, any pitch (k-contiguous operand)
  const float* xp; int ldxp;           // the same rows with a 16-byte pitch (weight gradient), or null (forward only)
  const float* adv; int ld_adv;        // [rows][ld_adv]: the pass's adversarial columns, 16-byte pitch
  long wrap;
};
// a caller tensor's rows with a 16-byte pitch: the tensor itself when it already has one, else a copy made once per step
static int pitched_rows(gt_engine* e, int slot, const float* p, int ld, int cols, long rows, const float** out, int* ldo, hipStream_t s) {
  if (gemm_vec_ok(p, ld)) { *out = p; *ldo = ld; return GT_OK; }
  gt_engine::Pitched& P = e->pitched[slot];
  const int ldp = (cols + 3) & ~3;
  if (!(P.src == p && P.ld == ld && P.cols == cols && P.rows == rows && P.step == e->step_counter)) {
    CHK(P.buf.ensure((size_t)rows * ldp * sizeof(float)));
    hipLaunchKernelGGL(repitch_kernel, dim3(cdiv(rows * (ldp / 4), 256)), dim3(256), 0, s, p, ld, cols, rows, P.buf.as<float>(), ldp);
    LAUNCH_CHECK();
    P.src = p; P.ld = ld; P.cols = cols; P.rows = rows; P.step = e->step_counter;
  }
  *out = P.buf.as<float>(); *ldo = ldp;
  return GT_OK;
}

// hidden stack forward: in -> acts[0..L-1]; returns specs used (for backward)
static int stack_forward(gt_engine* e, int role, const float* in, int ld_in, long rows, std::vector<Scratch>& acts,
                         const int* passes, int npass, long rows_each, std::vector<DropoutSpec>& specs, hipStream_t s,
                         const FirstSplit* fs = nullptr, int max_layers = 1 << 30 /* launch only the first max_layers layers (the dropout
                         sites and the stash buffers of ALL layers are still set up: the fused stack launch takes over from there) */) {
  Net& n = e->net[role];
  specs.resize(n.hidden.size());
  const float* cur = in;
  int ld = ld_in;
  for (size_t l = 0; l < n.hidden.size(); ++l) {
    const Lin& L = n.hidden[l];
    CHK(acts[l].ensure((size_t)rows * L.out * sizeof(float)));
    const float* inj = nullptr;
    CHK(stage_injected(e, role, (int)l, passes, npass, rows_each, L.out, &inj, s));
    specs[l] = drop_spec(e, role, passes[0], (int)l, inj, L.out, npass == 2 ? rows_each : 0);
    if ((int)l >= max_layers) continue;
    if (l == 0 && fs && specs[l].mode == DROP_PHILOX && gt_tuning().split_fused && gemm_vec_ok(fs->x, fs->ldx, true) &&
        gemm_vec_ok(L.W, L.in, true) && gemm_vec_ok(fs->adv, fs->ld_adv, true) && fs->wrap > 64 && gemm_small_tiles_ok() &&
        tl_gemm_prec == PREC_F32 && (rows == fs->wrap || rows == 2 * fs->wrap)) {
      // ONE launch (GEMM_A_LEAKY_PHILOX_SEG): every tile multiplies its rows of x by W[:, :cd]^T once, then -- per half of the
      // pass -- continues the same accumulators over the adversarial columns and runs that half's epilogue: no P buffer
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = fs->x; g.lda = fs->ldx; g.B = L.W; g.ldb = L.in; g.C = acts[l].as<float>(); g.ldc = L.out;
      g.M = (int)fs->wrap; g.N = L.out; g.K = fs->cd; g.bias = L.b; g.act = ACT_LEAKY_DROPOUT; g.drop = specs[l];
      g.A_seg = fs->adv; g.lda_seg = fs->ld_adv; g.B_seg = L.W + fs->cd; g.K_seg = L.in - fs->cd;
      g.dual_rows = rows == 2 * fs->wrap ? (int)fs->wrap : 0;
      CHK(launch_gemm(GEMM_NT, g, 1, s));
      cur = acts[l].as<float>();
      ld = L.out;
      continue;
    }
    if (l == 0 && fs) {
      const int Da = L.in - fs->cd;
      CHK(e->d_pre.ensure((size_t)fs->wrap * L.out * sizeof(float)));
      CHK(linear_forward(fs->x, fs->ldx, L.W, L.in, L.b, e->d_pre.as<float>(), L.out, fs->wrap, fs->cd, L.out, ACT_NONE, no_drop(), s));
      GemmArgs g;
      memset(&g, 0, sizeof(g));
      g.A = fs->adv; g.lda = fs->ld_adv; g.B = L.W + fs->cd; g.ldb = L.in; g.C = acts[l].as<float>(); g.ldc = L.out;
      g.M = (int)rows; g.N = L.out; g.K = Da; g.act = ACT_LEAKY_DROPOUT; g.drop = specs[l];
      g.addm = e->d_pre.as<float>(); g.ld_addm = L.out; g.addm_wrap = (int)fs->wrap;
      CHK(launch_gemm(GEMM_NT, g, 1, s));
      cur = acts[l].as<float>();
      ld = L.out;
      continue;
    }
    const float* W = L.W;
    int ldw = L.in;
    if (l == 0 && (L.in & 3) && gemm_vec_ok(cur, ld) && !gemm_vec_ok(L.W, L.in, true) && tl_gemm_prec == PREC_F32) {
      // The input image takes 16-byte loads (the discriminator's [x | adv] image, pitch 484) but the weight rows (483
      // floats) do not: multiply by a copy of W with its row pitch rounded up to 4 floats, re-made from the caller's
      // parameter buffer before every pass (it may have been stepped, loaded or broadcast since).
      ldw = (L.in + 3) & ~3;
      CHK(e->w0pad[role].ensure((size_t)L.out * ldw * sizeof(float)));
      hipLaunchKernelGGL(pad_rows_kernel, dim3(cdiv((long)L.out * ldw, 256)), dim3(256), 0, s, L.W, L.in, L.out, e->w0pad[role].as<float>(), ldw);
      LAUNCH_CHECK();
      W = e->w0pad[role].as<float>();
    }
    CHK(linear_forward(cur, ld, W, ldw, L.b, acts[l].as<float>(), L.out, rows, L.in, L.out, ACT_LEAKY_DROPOUT, specs[l], s));
    cur = acts[l].as<float>();
    ld = L.out;
  }
  return GT_OK;
}

// Measured with one rank and forced collectives (bench.py --force-dp, plain step 1.465 ms): round-2 schedule 1.577 ms; the
// discriminator's gradient as ONE message 1.561 (kept); additionally ncclGroupStart/End around a step's closing messages
// 1.571 (off); generator loss sums sent with the closing messages instead of early 1.604 (off: the host then waits for the
// whole step before it can enqueue the next one).

// hidden stack backward.  dz_top: gradient w.r.t. the pre-activation of the TOP hidden layer
// (already multiplied by f'), in buffer `cur` (rows x hidden).  Produces dW/db (if want_w) and,
// optionally, dX[:, col0:col0+ncols] of the stack input for rows [row0, row0+nrows).
static int stack_backward(gt_engine* e, int role, const float* in, int ld_in, long rows, std::vector<Scratch>& acts,
                          const std::vector<DropoutSpec>& specs, float* cur, float* other, bool want_w,
                          float* dX, int lddx, int col0, int ncols, long row0, long nrows, hipStream_t s, const FirstSplit* fs = nullptr) {
  Net& n = e->net[role];
  const int L = (int)n.hidden.size();
  {
    // per-layer launches: each layer's weight gradient right behind the product that made its dZ (still warm in L2 / MALL)
    for (int l = L - 1; l >= 0; --l) {
      const Lin& Lr = n.hidden[l];
      const float* Xin = l > 0 ? acts[l - 1].as<float>() : in;
      const int ldx = l > 0 ? n.hidden[l - 1].out : ld_in;
      bool rode = false;
      GemmArgs nn;
      if (l > 0) nn = backward_data_args(cur, Lr.out, Lr.W, Lr.in, 0, other, Lr.in, rows, Lr.out, Lr.in, ACT_LEAKY_DROPOUT,
                                         acts[l - 1].as<float>(), Lr.in, specs[l - 1]);
      if (want_w && l == 0 && fs) {
        // the kept gradient w.r.t. the generated rows' adversarial columns (no activation derivative) rides in the same launch
        GemmArgs leak;
        if (dX) leak = backward_data_args(cur + row0 * Lr.out, Lr.out, Lr.W, Lr.in, col0, dX, lddx, nrows, Lr.out, ncols, ACT_NONE, nullptr, 0, no_drop());
        CHK(linear_backward_weight_split(cur, Lr.out, rows, fs->wrap, fs->xp, fs->ldxp, fs->cd, fs->adv, fs->ld_adv, Lr.in - fs->cd, Lr.out,
                                         Lr.dW, Lr.db, n.grads_dirty, e->slabs, s, &e->sdefer[role], dX && gt_tuning().leak_rider ? &leak : nullptr, &rode));
        CHK(comm_grads_ready(e, role, Lr.dW, (long)Lr.out * Lr.in + Lr.out, s));
      } else if (want_w) {
        CHK(linear_backward_weight(cur, Lr.out, Xin, ldx, rows, Lr.out, Lr.in, Lr.dW, Lr.db, n.grads_dirty, e->slabs, e->colp, s, &e->sdefer[role],
                                   l > 0 ? &nn : nullptr, &rode));
        CHK(comm_grads_ready(e, role, Lr.dW, (long)Lr.out * Lr.in + Lr.out, s));
        // generator: all layers above the first leave as one message under the first layer's backward; the discriminator's
        // whole gradient (1 MB) is ONE message at the end of its backward pass (a second launch costs more than it hides)
        if (l == 1 && (role == GT_ROLE_G || !e->opt_comm_d_one_msg)) CHK(comm_flush(e, role, s));
      }
      if (l > 0) {
        if (!rode) CHK(launch_gemm(GEMM_NN, nn, 1, s));
        std::swap(cur, other);
      } else if (dX && !rode) {
        CHK(linear_backward_data(cur + row0 * Lr.out, Lr.out, Lr.W, Lr.in, col0, dX, lddx, nrows, Lr.out, ncols, ACT_NONE, nullptr, 0,
                                 no_drop(), s));
      }
    }
    return GT_OK;
  }
}

// ------------------------------------------------------------------------------------------
// bf16-storage MLP stacks (GT_OPT_MATMUL_BF16; gemm_bf16s.hip.h): activations, dZ, the input images and weight shadows
// live in HBM as bf16, in both orientations; every product is the k-contiguous form.
// ------------------------------------------------------------------------------------------
static bool use_b16(const gt_engine* e, int role) {
  const Net& n = e->net[role];
  if (!e->matmul_bf16 || !n.bound || n.d.arch != GT_ARCH_MLP || (n.d.hidden_dim & 7)) return false;
  return true;
}
// re-made from the caller's float32 parameters before every pass (they may have been stepped, loaded or broadcast since)
static int refresh_shadows(gt_engine* e, int role, bool with_last, hipStream_t s) {
  Net& n = e->net[role];
  auto& sh = e->wsh[role];
  sh.resize(n.hidden.size() + 1);
  CastJobs jobs;
  jobs.n = 0; jobs.pad_ = 0;
  int blocks = 0;
  auto flush = [&]() -> int {
    if (jobs.n > 0) {
      hipLaunchKernelGGL(cast_transpose_multi_kernel, dim3(blocks), dim3(256), 0, s, jobs);
      LAUNCH_CHECK();
    }
    jobs.n = 0; blocks = 0;
    return GT_OK;
  };
  for (size_t l = 0; l <= n.hidden.size(); ++l) {     // all layers of the network in ONE launch
    if (l == n.hidden.size() && !with_last) break;
    const Lin& L = l < n.hidden.size() ? n.hidden[l] : n.last;
    LinShadow& w = sh[l];
    w.ldw = pad8(L.in); w.ldwt = pad8(L.out);
    CHK(w.w.ensure((size_t)L.out * w.ldw * 2 + 64));
    CHK(w.wt.ensure((size_t)L.in * w.ldwt * 2 + 64));
    if (jobs.n == CAST_MAX_JOBS) CHK(flush());
    CastJob& J = jobs.j[jobs.n++];
    J.in = L.W; J.ldi = L.in; J.rows = L.out; J.cols = L.in; J.out = w.w.as<__bf16>(); J.ldo = w.ldw; J.outT = w.wt.as<__bf16>(); J.ldt = w.ldwt;
    J.gy = cdiv(L.in, 64); J.block0 = blocks; J.pad_ = 0;
    blocks += cdiv(L.out, 64) * J.gy;
  }
  return flush();
}
// in_b [rows][ld_in] bf16 -> acts[l] (bf16, + transposed twin when want_t: the weight gradients read it)
static int stack_forward_b16(gt_engine* e, int role, const __bf16* in_b, int ld_in, long rows, std::vector<B16Img>& acts,
                             const int* passes, int npass, long rows_each, std::vector<DropoutSpec>& specs, bool want_t, hipStream_t s) {
  Net& n = e->net[role];
  specs.resize(n.hidden.size());
  acts.resize(n.hidden.size());
  const __bf16* cur = in_b;
  int ld = ld_in;
  for (size_t l = 0; l < n.hidden.size(); ++l) {
    const Lin& L = n.hidden[l];
    CHK(acts[l].ensure(rows, L.out, want_t));
    const float* inj = nullptr;
    CHK(stage_injected(e, role, (int)l, passes, npass, rows_each, L.out, &inj, s));
    specs[l] = drop_spec(e, role, passes[0], (int)l, inj, L.out, npass == 2 ? rows_each : 0);
    GemmB16Args g = b16_args();
    g.A = cur; g.lda = ld; g.B = e->wsh[role][l].w.as<__bf16>(); g.ldb = e->wsh[role][l].ldw;
    g.M = (int)rows; g.N = L.out; g.K = L.in; g.bias = L.b; g.epi = B16_FWD; g.act = ACT_LEAKY_DROPOUT; g.drop = specs[l];
    g.Cb = acts[l].r(); g.ldcb = acts[l].ld;
    if (want_t) { g.CbT = acts[l].t(); g.ldcbt = (int)acts[l].ldt; }
    CHK(launch_gemm_b16(g, 1, s));
    cur = acts[l].r();
    ld = acts[l].ld;
  }
  return GT_OK;
}
// dz[cur]: gradient w.r.t. the pre-activation of the TOP hidden layer (both orientations when want_w).  in_t: transposed
// image of the stack input [in][rows8] (weight gradient of layer 0).  dX (float32, optional): d loss / d input columns
// [col0, col0 + ncols) for rows [row0, row0 + nrows).
static int stack_backward_b16(gt_engine* e, int role, const __bf16* in_t, long ld_int, long rows, std::vector<B16Img>& acts,
                              const std::vector<DropoutSpec>& specs, int cur, bool want_w, float* dX, int lddx, int col0, int ncols,
                              long row0, long nrows, hipStream_t s) {
  Net& n = e->net[role];
  const int L = (int)n.hidden.size();
  for (int l = L - 1; l >= 0; --l) {
    const Lin& Lr = n.hidden[l];
    B16Img& dz = e->dz_b[cur];
    if (want_w) {
      const __bf16* XT = l > 0 ? acts[l - 1].t() : in_t;
      const long ldxt = l > 0 ? acts[l - 1].ldt : ld_int;
      CHK(weight_grad_b16(dz.t(), dz.ldt, XT, ldxt, rows, Lr.out, Lr.in, Lr.dW, Lr.db, n.grads_dirty, e->slabs, s, &e->sdefer[role]));
      CHK(comm_grads_ready(e, role, Lr.dW, (long)Lr.out * Lr.in + Lr.out, s));
      if (l == 1 && (role == GT_ROLE_G || !e->opt_comm_d_one_msg)) CHK(comm_flush(e, role, s));
    }
    if (l > 0) {
      B16Img& nx = e->dz_b[cur ^ 1];
      CHK(nx.ensure(rows, Lr.in, want_w));
      GemmB16Args g = b16_args();
      g.A = dz.r(); g.lda = dz.ld; g.B = e->wsh[role][l].wt.as<__bf16>(); g.ldb = e->wsh[role][l].ldwt;
      g.M = (int)rows; g.N = Lr.in; g.K = Lr.out; g.epi = B16_BWD_DATA; g.act = ACT_LEAKY_DROPOUT;
      g.H = acts[l - 1].r(); g.ldh = acts[l - 1].ld; g.drop = specs[l - 1];
      g.Cb = nx.r(); g.ldcb = nx.ld;
      if (want_w) { g.CbT = nx.t(); g.ldcbt = (int)nx.ldt; }
      CHK(launch_gemm_b16(g, 1, s));
      cur ^= 1;
    } else if (dX) {
      GemmB16Args g = b16_args();
      g.A = dz.r() + row0 * dz.ld; g.lda = dz.ld;
      g.B = e->wsh[role][0].wt.as<__bf16>() + (long)col0 * e->wsh[role][0].ldwt; g.ldb = e->wsh[role][0].ldwt;
      g.M = (int)nrows; g.N = ncols; g.K = Lr.out; g.epi = B16_BWD_DATA; g.act = ACT_NONE;
      g.C = dX; g.ldc = lddx;
      CHK(launch_gemm_b16(g, 1, s));
    }
  }
  return GT_OK;
}


// row pitch of the generator's input / of the conditioning x (gt_set_x_pitch; dense by default)
static int gx_pitch(const gt_engine* e) { return (e->ld_gx > 0 && !e->gx_dense_on) ? e->ld_gx : e->net[GT_ROLE_G].d.in_dim; }
static int cx_pitch(gt_engine* e) { return e->ld_cx > 0 ? e->ld_cx : cond_dim(e); }
// the generator input as the network can read it: pitched rows (gt_set_x_pitch) stay where they are for the float32 MLP generator;
// every other generator reads a dense copy made here (the copy is also what its backward pass reads: e->last_x)
static int dense_gx(gt_engine* e, const float** x, long N, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  e->gx_dense_on = false;
  if (e->ld_gx <= 0 || e->ld_gx == G.d.in_dim) return GT_OK;
  if (e->ld_gx < G.d.in_dim) return fail(GT_ERR_INVALID, "gt_set_x_pitch: pitch %d for %d generator input columns", e->ld_gx, G.d.in_dim);
  if (G.d.arch == GT_ARCH_MLP && !use_b16(e, GT_ROLE_G)) return GT_OK;
  CHK(e->gx_dense.ensure((size_t)N * G.d.in_dim * sizeof(float)));
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * G.d.in_dim, 256)), dim3(256), 0, s, *x, e->ld_gx, 0, (const int*)nullptr,
                     e->gx_dense.as<float>(), G.d.in_dim, 0, (int)N, G.d.in_dim);
  LAUNCH_CHECK();
  *x = e->gx_dense.as<float>();
  e->gx_dense_on = true;
  return GT_OK;
}
// the conditioning x of the discriminator passes that cannot read pitched rows (no split first layer): a dense copy, once per step
static int dense_cx(gt_engine* e, const float** x, long N, hipStream_t s) {
  if (!*x || !e->cfg.discriminator_linguistic_condition || e->ld_cx <= 0 || e->ld_cx == cond_dim(e)) return GT_OK;
  const int cd = cond_dim(e);
  if (e->ld_cx < cd) return fail(GT_ERR_INVALID, "gt_set_x_pitch: pitch %d for %d conditioning columns", e->ld_cx, cd);
  // (the copy is keyed on everything it depends on -- ADVICE r5: the pointer and the step alone served a stale or too small copy to a caller that
  //  refilled x in place or changed B*T between two update_* calls of one step)
  if (!(e->cxd_src == *x && e->cxd_step == e->step_counter && e->cxd_rows == N && e->cxd_ld == e->ld_cx && e->cxd_cols == cd && e->cx_dense.p)) {
    CHK(e->cx_dense.ensure((size_t)N * cd * sizeof(float)));
    hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * cd, 256)), dim3(256), 0, s, *x, e->ld_cx, 0, (const int*)nullptr,
                       e->cx_dense.as<float>(), cd, 0, (int)N, cd);
    LAUNCH_CHECK();
    e->cxd_src = *x; e->cxd_step = e->step_counter; e->cxd_rows = N; e->cxd_ld = e->ld_cx; e->cxd_cols = cd;
  }
  *x = e->cx_dense.as<float>();
  return GT_OK;
}

static int generator_forward(gt_engine* e, const float* x, const float* R, int B, int T, float* y_hat, float* y_hat_static,
                             bool stash, hipStream_t s, std::vector<DropoutSpec>& specs) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  CHK(dense_gx(e, &x, N, s));
  const int pass0[1] = {0};
  const float* gsrc = y_hat;            // what MLPG is applied to
  if (G.d.arch == GT_ARCH_LSTM) {
    CHK(lstm_forward(e, x, B, T, y_hat, s));
  } else if (G.d.arch == GT_ARCH_IN2OUT_RNN) {
    // G(x) = hidden2out(LSTM(x)) stays internal; the model returns its INPUT as y_hat (models.py:118)
    CHK(e->i2o_gout.ensure((size_t)N * G.d.out_dim * sizeof(float)));
    CHK(lstm_forward(e, x, B, T, e->i2o_gout.as<float>(), s));
    hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * G.d.in_dim, 256)), dim3(256), 0, s, x, G.d.in_dim, 0, (const int*)nullptr,
                       y_hat, G.d.in_dim, 0, (int)N, G.d.in_dim);
    LAUNCH_CHECK();
    gsrc = e->i2o_gout.as<float>();
  } else if (G.d.arch == GT_ARCH_SRU) {
    CHK(sru_forward(e, x, B, T, y_hat, s));
  } else {
    if (use_b16(e, GT_ROLE_G)) {
      const bool want_t = stash && G.d.grads != nullptr;
      CHK(e->xin_b.ensure(N, G.d.in_dim, want_t));
      CHK(cast_transpose(x, G.d.in_dim, N, G.d.in_dim, e->xin_b.r(), e->xin_b.ld, want_t ? e->xin_b.t() : (__bf16*)nullptr, e->xin_b.ldt,
                                nullptr, false, &e->colp, s));
      CHK(refresh_shadows(e, GT_ROLE_G, true, s));
      CHK(stack_forward_b16(e, GT_ROLE_G, e->xin_b.r(), e->xin_b.ld, N, e->g_actb, pass0, 1, N, specs, want_t, s));
      const LinShadow& ws = e->wsh[GT_ROLE_G][G.hidden.size()];
      GemmB16Args g = b16_args();
      g.A = e->g_actb.back().r(); g.lda = e->g_actb.back().ld; g.B = ws.w.as<__bf16>(); g.ldb = ws.ldw;
      g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
      g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
      CHK(launch_gemm_b16(g, 1, s));
    } else {
      CHK(stack_forward(e, GT_ROLE_G, x, gx_pitch(e), N, e->g_act, pass0, 1, N, specs, s));
      const Lin& Lh = G.hidden.back();
      CHK(linear_forward(e->g_act.back().as<float>(), Lh.out, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in,
                         G.last.out, G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s));
    }
  }
  if (is_i2o(G.d.arch)) {
    if (!R) return fail(GT_ERR_INVALID, "In2OutHighwayNet needs the MLPG matrix R (models.py:54)");
    const int sd = G.d.static_dim;
    CHK(ensure_band(e, R, T, s));
    CHK(e->tx.ensure((size_t)N * sd * sizeof(float)));
    CHK(e->gx.ensure((size_t)N * sd * sizeof(float)));
    // T(x) = sigmoid(T x_static), x_static = x[:, :, :static_dim]   (models.py:57-60)
    CHK(linear_forward(x, G.d.in_dim, G.gate.W, sd, G.gate.b, e->tx.as<float>(), sd, N, sd, sd, ACT_SIGMOID, no_drop(), s));
    CHK(mlpg_forward(e, gsrc, G.d.out_dim, e->d_scol_i2o, e->d_sstride_i2o, sd, e->gx.as<float>(), sd, B, T, s));
    if (stash) e->g_used_mlpg = true;
    hipLaunchKernelGGL(highway_forward_kernel, dim3(cdiv(N * sd, 256)), dim3(256), 0, s, x, G.d.in_dim, e->tx.as<float>(), sd,
                       e->gx.as<float>(), sd, y_hat_static, sd, N, sd);
    LAUNCH_CHECK();
  } else {
    if (G.d.out_dim != e->Dout_cfg)
      return fail(GT_ERR_DIM, "You probably have specified wrong dimention params.");  // multistream.py:93-94
    if (R) {
      CHK(ensure_band(e, R, T, s));
      CHK(mlpg_forward(e, y_hat, G.d.out_dim, e->d_scol, e->d_sstride, e->Ds, y_hat_static, e->Ds, B, T, s));
      if (stash) e->g_used_mlpg = true;
    } else {
      if (e->Ds != G.d.out_dim) return fail(GT_ERR_INVALID, "R is None but the stream config has dynamic features");
      if (stash) e->g_used_mlpg = false;
      // R is None: num_windows = 1, every stream passes through (multistream.py:88-89,119-120)
      hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * G.d.out_dim, 256)), dim3(256), 0, s, y_hat, G.d.out_dim, 0,
                         (const int*)nullptr, y_hat_static, G.d.out_dim, 0, (int)N, G.d.out_dim);
      LAUNCH_CHECK();
    }
  }
  return GT_OK;
}

extern "C" int gt_apply_generator(gt_engine* e, const float* x, const float* R, int B, int T, float* y_hat,
                                  float* y_hat_static, void* stream) {
  CHK(check_common(e, B, T));
  Net& G = e->net[GT_ROLE_G];
  if (!G.bound) return fail(GT_ERR_STATE, "generator not bound");
  if (!x || !y_hat || !y_hat_static) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  CHK(fault_seen(e));
  e->step_counter++;
  e->B = B; e->T = T; e->N = (long)B * T;
  e->g_pass_valid = false;
  e->fake_cat_valid = false; e->dcat_b_ok = false; e->adv2_fake_ok = false; e->cxd_src = nullptr;
  e->tv_mask = nullptr; e->tv_inflight = false;             // a new batch: the mask contents may have changed
  CHK(generator_forward(e, x, R, B, T, y_hat, y_hat_static, true, s, e->g_specs));
  e->last_x = e->gx_dense_on ? e->gx_dense.as<float>() : x;      // (what the backward pass reads: the dense copy when one was made)
  e->last_yhat = y_hat; e->last_yhs = y_hat_static;
  e->g_pass_valid = true;
  return GT_OK;
}

// width of the conditioning input x fed to D (train.py:254-256); derived from the bound D when not configured
int cond_dim(gt_engine* e) {
  if (!e->cfg.discriminator_linguistic_condition) return 0;
  if (e->cfg.cond_dim > 0) return e->cfg.cond_dim;
  return e->net[GT_ROLE_D].bound ? e->net[GT_ROLE_D].d.in_dim - e->Da : 0;
}
static int d_in_dim(gt_engine* e) { return e->Da + cond_dim(e); }

// rows [row0, row0+N) of dcat <- [x | feats[:, adv_cols]]
static int build_cat(gt_engine* e, const float* x, const float* feats, int ld_feats, long row0, long N, int ldc, hipStream_t s) {
  float* dst = e->dcat.as<float>() + row0 * ldc;
  int off = 0;
  if (e->cfg.discriminator_linguistic_condition) {
    if (!x) return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
    const int cd = cond_dim(e);
    if (cd <= 0) return fail(GT_ERR_DIM, "discriminator in_dim too small for linguistic conditioning");
    hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * cd, 256)), dim3(256), 0, s, x, cd, 0, (const int*)nullptr, dst, ldc, 0,
                       (int)N, cd);
    LAUNCH_CHECK();
    off = cd;
  }
  hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * e->Da, 256)), dim3(256), 0, s, feats, ld_feats, 0, e->d_adv_cols, dst, ldc,
                     off, (int)N, e->Da);
  LAUNCH_CHECK();
  return GT_OK;
}

// the head's per-workgroup partials (e->headp, e->headw: nblk of them) -> the step's scalars and d last_linear
static int head_finalize(gt_engine* e, int mode, int nblk, int K, bool w, hipStream_t s, StepResults* early_res, int* defer_scalars, unsigned ticket) {
  Net& D = e->net[GT_ROLE_D];
  if (defer_scalars && !w) { *defer_scalars = nblk; return GT_OK; }
  const int cgw = (w && nblk >= 512) ? 16 : 64;       // many rows of partials: narrower column groups, more workgroups (frame_kernels.hip.h)
  const int n_dw = cdiv(K, cgw), extra = cgw == 16 ? 1 : 0;       // (with many partials the scalars get a workgroup of their own, beside the dw ones)
  hipLaunchKernelGGL(d_head_finalize_kernel, dim3(n_dw + extra), dim3(1024), 0, s, e->headp.as<HeadPartials>(), e->headw.as<float>(),
                     nblk, K, mode, e->sc(), w ? D.last.dW : (float*)nullptr, w ? D.last.db : (float*)nullptr, D.grads_dirty ? 1 : 0,
                     early_res, ticket ? e->ticket_dev() : (unsigned*)nullptr, ticket, cgw, extra ? n_dw : 0);
  LAUNCH_CHECK();
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// fused discriminator stack (dstack_f32.hip.h): layers 1 .. L-1 + head (+ the generator step's backward-data chain) in one launch
// ------------------------------------------------------------------------------------------
// rows: frames of the pass.  A panel is walked through all layers by ONE workgroup (about 60 us of latency at 3 x 256): the fused
// launch pays when the pass has at least one panel per CU; below that (per-rank batches of a few sequences) the per-layer launches,
// which spread every layer over all CUs, are faster (b = 4: 0.413 vs 0.417 ms per step).  GT_OPT_FUSED_DSTACK = 2 forces the fused
// path at any size (tests).
static bool d_fused_ok(gt_engine* e, bool b16, long rows) {
  Net& D = e->net[GT_ROLE_D];
  if (e->opt_fused_dstack < 2 && dstack_panels(rows) < gemm_cu_count()) return false;
  if (!e->opt_fused_dstack || b16 || D.d.arch != GT_ARCH_MLP || tl_gemm_prec != PREC_F32 || !dstack_hidden_ok(D.d.hidden_dim)) return false;
  if (D.hidden.empty() || (int)D.hidden.size() > DS_MAXL || D.last.out != 1 || e->Da < 1 || e->Da > 64) return false;
  for (size_t l = 0; l < D.hidden.size(); ++l)
    if (D.hidden[l].out != D.d.hidden_dim || (l > 0 && D.hidden[l].in != D.d.hidden_dim)) return false;
  return true;
}
// mode DSTACK_D_STEP: rows = 2N (natural then generated), writes the stashes e->d_act[1 ..], the seed gradient e->dzA and the head's
// partials, then finalises them (as run_head does).  mode DSTACK_G_ADV: rows = N, writes gadv (want_grad) and the partials.
static int run_dstack(gt_engine* e, int mode, long rows, long n_real, const float* mask, long n_mask, float eps, bool want_grad,
                      float* gadv, hipStream_t s, StepResults* early_res, int* defer_scalars, const double* tv_dev, unsigned ticket, bool unit_tv) {
  Net& D = e->net[GT_ROLE_D];
  const int H = D.d.hidden_dim, L = (int)D.hidden.size();
  const int nblk = dstack_panels(rows);
  CHK(e->headp.ensure((size_t)nblk * sizeof(HeadPartials)));
  CHK(e->dout.ensure((size_t)rows * sizeof(float)));
  DStackArgs a;
  memset(&a, 0, sizeof(a));
  a.mode = mode; a.L = L; a.rows = (int)rows; a.n_real = (int)n_real;
  a.H0 = e->d_act[0].as<float>();
  for (int l = 0; l < L; ++l) {
    a.W[l] = D.hidden[l].W; a.b[l] = D.hidden[l].b; a.drop[l] = e->d_specs[l];
    if (a.drop[l].mode == DROP_BUFFER && a.drop[l].ld_mask != H) return fail(GT_ERR_INVALID, "injected dropout mask pitch");
  }
  a.w_last = D.last.W; a.b_last = D.last.b; a.mask = mask; a.n_mask = (int)n_mask; a.eps = eps; a.unit_tv = unit_tv ? 1 : 0; a.tv_dev = tv_dev;
  a.sc = e->sc(); a.want_grad = want_grad ? 1 : 0; a.Dout = e->dout.as<float>(); a.hp = e->headp.as<HeadPartials>();
  if (mode == DSTACK_D_STEP) {
    for (int l = 1; l + 1 < L; ++l) a.Hout[l] = e->d_act[l].as<float>();      // (the top layer's activation only feeds the head, which is fused: no stash)
    a.dZtop = e->dzA.as<float>();
    if (want_grad) { CHK(e->headw.ensure((size_t)nblk * H * sizeof(float))); a.dw_partial = e->headw.as<float>(); }
  } else {
    a.W0 = D.hidden[0].W; a.ldw0 = D.hidden[0].in; a.col0 = cond_dim(e); a.Da = e->Da; a.gadv = gadv; a.ld_gadv = e->Da;
    if (want_grad && !gadv) return fail(GT_ERR_INVALID, "fused discriminator stack: no gradient buffer");
  }
  CHK(launch_dstack(a, H, s));
  return head_finalize(e, mode == DSTACK_D_STEP ? HEAD_D_STEP : HEAD_G_ADV, nblk, H, mode == DSTACK_D_STEP && want_grad, s, early_res, defer_scalars, ticket);
}

// H: the top hidden activation, float32 [n_rows][K] or (h_ld > 0) its bf16 image with row pitch h_ld
static int run_head(gt_engine* e, int mode, const void* H, int K, long n_rows, long n_real, const float* mask, long n_mask,
                    float eps, bool want_grad, float* dH, const DropoutSpec& spec, bool want_w, hipStream_t s,
                    StepResults* early_res = nullptr, int h_ld = 0, B16Img* dz_img = nullptr, bool dz_t = false,
                    int* defer_scalars = nullptr /* HEAD_G_ADV without weight gradients: the caller reduces the partials; <- their count */,
                    const double* tv_dev = nullptr /* the valid-frame count when it is not in the step's scalars yet */,
                    unsigned ticket = 0 /* early_res in host memory: the ticket that announces it */,
                    bool unit_tv = false /* seed the backward pass of the UNNORMALISED loss (GT_OPT_COMM_TV_IN_SUMS) */,
                    bool has_act = true /* H is LeakyReLU + dropout of a pre-activation (MLP); false: a recurrent stack's output */) {
  Net& D = e->net[GT_ROLE_D];
  const int nblk = (int)std::min<long>(1024, (n_rows + 31) / 32);
  CHK(e->headp.ensure((size_t)nblk * sizeof(HeadPartials)));
  CHK(e->headw.ensure((size_t)nblk * K * sizeof(float)));
  CHK(e->dout.ensure((size_t)n_rows * sizeof(float)));
  const size_t lds = (size_t)4 * K * sizeof(float);
#define GT_HEAD_LAUNCH(KP_)                                                                                              \
  if (h_ld > 0)                                                                                                          \
    hipLaunchKernelGGL((d_head_kernel<KP_, __bf16, true>), dim3(nblk), dim3(256), lds, s, (const __bf16*)H, h_ld, K, D.last.W, D.last.b, mask, (int)n_mask, \
                       (int)n_real, (int)n_rows, mode, eps, e->dout.as<float>(), dz_img ? (float*)nullptr : dH, K, want_grad ? 1 : 0, spec, has_act ? 1 : 0, e->sc(), \
                       e->headp.as<HeadPartials>(), e->headw.as<float>(), dz_img ? dz_img->r() : (__bf16*)nullptr, dz_img ? dz_img->ld : 0,  \
                       (dz_img && dz_t) ? dz_img->t() : (__bf16*)nullptr, dz_img ? dz_img->ldt : 0L, tv_dev, unit_tv ? 1 : 0);            \
  else if ((KP_) % 4 == 0 && gt_tuning().head_vec)                                                                       \
    hipLaunchKernelGGL((d_head_kernel<((KP_) % 4 == 0 ? (KP_) : 4), float, false, true>), dim3(nblk), dim3(256), lds, s, (const float*)H, K, K, D.last.W, D.last.b, mask, (int)n_mask,  \
                       (int)n_real, (int)n_rows, mode, eps, e->dout.as<float>(), dH, K, want_grad ? 1 : 0, spec, has_act ? 1 : 0, e->sc(), \
                       e->headp.as<HeadPartials>(), e->headw.as<float>(), (__bf16*)nullptr, 0, (__bf16*)nullptr, 0L, tv_dev, unit_tv ? 1 : 0); \
  else                                                                                                                   \
    hipLaunchKernelGGL((d_head_kernel<KP_, float, false>), dim3(nblk), dim3(256), lds, s, (const float*)H, K, K, D.last.W, D.last.b, mask, (int)n_mask,  \
                       (int)n_real, (int)n_rows, mode, eps, e->dout.as<float>(), dH, K, want_grad ? 1 : 0, spec, has_act ? 1 : 0, e->sc(), \
                       e->headp.as<HeadPartials>(), e->headw.as<float>(), (__bf16*)nullptr, 0, (__bf16*)nullptr, 0L, tv_dev, unit_tv ? 1 : 0)
  if (K <= 128) { GT_HEAD_LAUNCH(2); }
  else if (K <= 256) { GT_HEAD_LAUNCH(4); }
  else if (K <= 512) { GT_HEAD_LAUNCH(8); }
  else if (K <= 1024) { GT_HEAD_LAUNCH(16); }
  else return fail(GT_ERR_INVALID, "discriminator hidden_dim > 1024 is not supported by the fused head kernel");
#undef GT_HEAD_LAUNCH
  LAUNCH_CHECK();
  return head_finalize(e, mode, nblk, K, want_grad && want_w, s, early_res, defer_scalars, ticket);
}

static int optimizer_step(gt_engine* e, int role, double* norm2_out, hipStream_t s) {
  Net& n = e->net[role];
  if (!n.has_opt) return fail(GT_ERR_STATE, "phase == \"train\" but no optimizer is bound for role %d", role);
  const long np = n.d.n_params;
  CHK(e->partial.ensure(4096 * sizeof(double)));
  double* part = e->partial.as<double>() + 2048;
  OptimSpec o;
  o.kind = n.od.kind; o.lr = n.od.lr; o.weight_decay = n.od.weight_decay; o.eps = n.od.eps; o.lr_decay = n.od.lr_decay;
  o.beta1 = n.od.beta1; o.beta2 = n.od.beta2; o.step = n.step + 1; o.max_norm = n.od.max_grad_norm;
  unsigned int* skipped = e->h_fault_dev ? e->h_fault_dev + 1 + role : (unsigned int*)nullptr;
  // the discriminator's gradient of a GT_OPT_COMM_TV_IN_SUMS step is that of the unnormalised loss: x 1 / Tv in the update kernel
  const float* gscale = (role == GT_ROLE_D && e->d_unnorm) ? &e->sc()->inv_tv : (const float*)nullptr;
  if (role == GT_ROLE_D) e->d_unnorm = false;
  SlabDefer& sd = e->sdefer[role];
  if (sd.active && sd.jobs.n > 0) {
    // The fused step recorded this network's weight-gradient combines.  They write disjoint ranges of the flat gradient; whatever
    // they do not cover (the discriminator's last layer, written by the head's reduction) still counts for the norm and is
    // stepped: `rest`.  Default: combines + squared norm in ONE launch (slab_reduce_norm_kernel), then clip + step;
    // GT_OPT_FUSED_OPTIMIZER: all of it in one launch behind a device-wide barrier (measured slower).
    std::vector<std::pair<long, long>> cov;
    bool ok = true;
    for (int q = 0; q < sd.jobs.n && ok; ++q) {
      const SlabJob& J = sd.jobs.j[q];
      const long o0 = J.out - n.d.grads;
      if (o0 < 0 || o0 + J.n4 * 4 > np) ok = false;
      cov.push_back(std::make_pair(o0, J.n4 * 4));
      if (J.bout && J.main_blocks < (q + 1 < sd.jobs.n ? sd.jobs.j[q + 1].block0 : sd.blocks) - J.block0) {
        const long b0 = J.bout - n.d.grads;
        if (b0 < 0 || b0 + J.nb > np) ok = false;
        cov.push_back(std::make_pair(b0, (long)J.nb));
      }
    }
    std::sort(cov.begin(), cov.end());
    OptimRest rest;
    memset(&rest, 0, sizeof(rest));
    long pos = 0, rest_total = 0;
    for (size_t i = 0; i <= cov.size() && ok; ++i) {
      const long next = i < cov.size() ? cov[i].first : np;
      if (next < pos) { ok = false; break; }                       // overlapping jobs: not this path
      if (next > pos) {
        if (rest.n_rest == OPTIM_REST_MAX) { ok = false; break; }
        rest.off[rest.n_rest] = pos; rest.n[rest.n_rest] = next - pos; ++rest.n_rest;
        rest_total += next - pos;
      }
      if (i < cov.size()) pos = cov[i].first + cov[i].second;
    }
    if (ok && e->opt_fused_optimizer && !gscale) {
      if (!e->opt_bar.p) { CHK(e->opt_bar.ensure(64)); HIPCHK(hipMemsetAsync(e->opt_bar.p, 0, 64, s)); e->opt_bar_count = 0; }
      const int grid = std::min(4 * gemm_cu_count(), std::max(sd.blocks, 64));
      e->opt_bar_count += (unsigned long long)grid;
      n.step += 1;
      hipLaunchKernelGGL(optim_fused_kernel, dim3(grid), dim3(256), 0, s, sd.jobs, sd.blocks, rest, n.d.params, n.d.grads, n.od.state0, n.od.state1,
                         part, e->opt_bar.as<unsigned long long>(), e->opt_bar_count, 200000000ULL /* 2 s of 100 MHz ticks */, norm2_out, o,
                         e->d_fault, e->h_fault_dev, skipped);
      LAUNCH_CHECK();
      sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = false;
      return GT_OK;
    }
    const int rest_blocks = rest_total > 0 ? (int)std::min<long>(64, cdiv(rest_total, 256)) : 0;
    if (ok && sd.blocks + rest_blocks <= 2048) {
      hipLaunchKernelGGL(slab_reduce_norm_kernel, dim3(sd.blocks + rest_blocks), dim3(256), 0, s, sd.jobs, sd.blocks, rest, n.d.grads, part);
      LAUNCH_CHECK();
      const int n_partial = sd.blocks + rest_blocks;
      sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = false;
      n.step += 1;
      const int grid = (int)std::min<long>(1024, cdiv(np, RED_THREADS));
      hipLaunchKernelGGL(optim_step_kernel, dim3(grid), dim3(RED_THREADS), 0, s, n.d.params, n.d.grads, n.od.state0, n.od.state1, np,
                         part, n_partial, norm2_out, o, (const unsigned int*)e->d_fault, e->h_fault_dev, skipped, gscale);
      LAUNCH_CHECK();
      return GT_OK;
    }
  }
  const int nblk = (int)std::min<long>(512, cdiv(np, RED_THREADS * 4));
  CHK(slab_defer_flush(e->sdefer[role], s));            // the fused step's recorded weight-gradient combines, one launch
  e->sdefer[role].active = false;
  hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, n.d.grads, np, part);
  LAUNCH_CHECK();
  n.step += 1;
  const int grid = (int)std::min<long>(1024, cdiv(np, RED_THREADS));
  hipLaunchKernelGGL(optim_step_kernel, dim3(grid), dim3(RED_THREADS), 0, s, n.d.params, n.d.grads, n.od.state0, n.od.state1, np,
                     part, nblk, norm2_out, o, (const unsigned int*)e->d_fault, e->h_fault_dev, skipped, gscale);
  LAUNCH_CHECK();
  return GT_OK;
}

static int fetch_results(gt_engine* e, hipStream_t s) {
  HIPCHK(hipMemcpyAsync(e->h_res, e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return GT_OK;
}
static int post_deferred_results(gt_engine* e, int role, hipStream_t s) {
  if (!e->h_def[role]) HIPCHK(hipHostMalloc((void**)&e->h_def[role], sizeof(StepResults)));
  if (!e->ev_def[role]) HIPCHK(hipEventCreateWithFlags(&e->ev_def[role], hipEventDisableTiming));
  HIPCHK(hipMemcpyAsync(e->h_def[role], e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(e->ev_def[role], s));
  e->def_pending[role] = true;
  return GT_OK;
}
static StepResults* early_res_target(gt_engine* e) { return e->h_res_dev ? e->h_res_dev : e->res(); }
// a ticket for results that the NEXT finalising launch writes into the host page (0: not available -- the event is used)
static unsigned take_ticket(gt_engine* e) {
  if (!e->opt_poll_results || !e->h_res_dev) return 0;
  if (++e->ticket_next == 0) ++e->ticket_next;
  return e->ticket_next;
}
static int wait_early_results(gt_engine* e) {
  if (e->ticket_wait) {
    volatile unsigned* t = e->ticket_host();
    const unsigned want = e->ticket_wait;
    e->ticket_wait = 0;
    unsigned long spins = 0;
    std::chrono::steady_clock::time_point t0;
    while (__atomic_load_n((const unsigned*)t, __ATOMIC_ACQUIRE) != want) {
      __builtin_ia32_pause();
      if ((++spins & 0xfffff) == 0) {      // a kernel that died never writes the ticket: give up after a minute
        if (spins == 0x100000) t0 = std::chrono::steady_clock::now();
        else if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 60.0) {
          HIPCHK(hipDeviceSynchronize());
          return fail(GT_ERR_HIP, "the step's results never arrived in host memory");
        }
      }
    }
    return GT_OK;
  }
  HIPCHK(hipEventSynchronize(e->ev_res));
  return GT_OK;
}
int post_early_results(gt_engine* e, hipStream_t s, unsigned ticket) {
  if (ticket) { e->ticket_wait = ticket; e->early_done = true; return GT_OK; }
  e->ticket_wait = 0;
  if (!e->ev_res) HIPCHK(hipEventCreateWithFlags(&e->ev_res, hipEventDisableTiming));
  if (!e->h_res_dev) HIPCHK(hipMemcpyAsync(e->h_res, e->res(), sizeof(StepResults), hipMemcpyDeviceToHost, s));
  HIPCHK(hipEventRecord(e->ev_res, s));
  e->early_done = true;
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// update_discriminator
// ------------------------------------------------------------------------------------------
// Small HBM-bound kernels that nothing in front of them depends on run on the engine's side stream, UNDER the matrix
// products of the step stream (which leave the memory system mostly idle): the valid-frame count of the D step (needed
// only by the head, a whole forward pass later) and the reported MSE loss of the G step (train.py:294; needed only by the
// step's finalisation).  side_fork: the side stream starts behind everything queued on `s` so far; side_join: `s` continues
// behind the side stream.  Fused single-GPU calls only.  MEASURED SLOWER (cfg2 1.426 vs 1.413 ms: two event hand-offs per use cost
// more than the 17 us of kernels they hide), so GT_OPT_SIDE_OVERLAP is off by default.
static int side_fork(gt_engine* e, hipStream_t s) {
  if (!e->side) {
    HIPCHK(hipStreamCreateWithFlags(&e->side, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&e->ev_side_go, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&e->ev_side_done, hipEventDisableTiming));
  }
  HIPCHK(hipEventRecord(e->ev_side_go, s));
  HIPCHK(hipStreamWaitEvent(e->side, e->ev_side_go, 0));
  return GT_OK;
}
static int side_join(gt_engine* e, hipStream_t s) {
  HIPCHK(hipEventRecord(e->ev_side_done, e->side));
  HIPCHK(hipStreamWaitEvent(s, e->ev_side_done, 0));
  return GT_OK;
}

// the split first layer (FirstSplit) applies to the conditioned discriminator on the float32 path
static bool d_split_ok(gt_engine* e, const float* x, bool b16) {
  return e->opt_split_first && !b16 && e->net[GT_ROLE_D].d.arch == GT_ARCH_MLP && tl_gemm_prec == PREC_F32 && e->cfg.discriminator_linguistic_condition && x && cond_dim(e) > 0 &&
         e->Da > 0 && gemm_small_tiles_ok() && (e->net[GT_ROLE_D].d.hidden_dim & 3) == 0;     // (dZ as a 16-byte loadable operand)
}

extern "C" int gt_update_discriminator_begin(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                                             const float* mask, int B, int T, int train, float eps, void* stream) {
  CHK(check_common(e, B, T));
  Net& D = e->net[GT_ROLE_D];
  if (!D.bound) return fail(GT_ERR_STATE, "discriminator not bound");
  if (!y_static || !y_hat_static || !mask) return fail(GT_ERR_INVALID, "null tensor");
  if (D.d.in_dim != d_in_dim(e))
    return fail(GT_ERR_DIM, "discriminator in_dim %d != adversarial input width %d (train.py:760-768)", D.d.in_dim, d_in_dim(e));
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const int K0 = D.d.in_dim, ldc = (K0 + 3) & ~3;
  const bool tr = train != 0;
  if (comm_on(e) && tr && D.grads_dirty)
    return fail(GT_ERR_STATE, "data-parallel step: optimizer_d.zero_grad() must precede update_discriminator (the gradient buckets are summed over the ranks in place)");
  { SlabDefer& sd = e->sdefer[GT_ROLE_D]; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = e->early && tr && D.has_opt; }
  // data parallel: the global count travels under the D forward pass.  With the split first layer its local term is summed by a
  // rider of the gather launch below (tv_ride_dp), else by a launch of its own right here.
  const bool tv_known = e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override;
  const bool dp_count = e->early && e->opt_launch_riders && comm_on(e) && !tv_known && !e->tv_dev && !(e->tv_override > 0.f) && !e->tv_inflight;
  // GT_OPT_COMM_TV_IN_SUMS: no collective for the count at all -- it leaves with the loss sums (engine_internal.hip.h)
  const bool unnorm = dp_count && e->opt_comm_tv_in_sums && tr && D.has_opt && D.d.grads;
  e->d_unnorm = unnorm;
  const bool split_gather = !use_b16(e, GT_ROLE_D) && d_split_ok(e, x, false);
  const bool tv_ride_dp = dp_count && !unnorm && split_gather;
  if (!tv_ride_dp && !unnorm) CHK(ensure_tv_begin(e, mask, N, s));
  if (unnorm && !split_gather) {
    hipLaunchKernelGGL(mask_total_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, &e->sc()->tv_sum);
    LAUNCH_CHECK();
  }
  const bool tv_side = e->early && !comm_on(e) && e->opt_side_overlap && !(e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override);
  if (tv_side) { CHK(side_fork(e, s)); CHK(ensure_tv(e, mask, N, e->side)); }      // single GPU: the count is summed under the D forward pass
  const int passes[2] = {0, 1};
  // the [x | adv] image of both halves: real rows, then generated rows
  const bool b16 = use_b16(e, GT_ROLE_D);
  const bool split = d_split_ok(e, x, b16);
  if (!split) CHK(dense_cx(e, &x, N, s));        // pitched rows are read in place by the split first layer only
  FirstSplit fs;
  memset(&fs, 0, sizeof(fs));
  if (b16) {
    // bf16 storage: the image is written ONCE, as bf16, in both orientations (no float32 image at all)
    if (e->cfg.discriminator_linguistic_condition && (!x || cond_dim(e) <= 0))
      return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
    CHK(e->dcat_b.ensure(2 * N, K0, tr));
    CatSrc src;
    src.x = x; src.cd = cond_dim(e); src.fa = y_static; src.fb = y_hat_static; src.ldf = e->Ds; src.idx = e->d_adv_cols; src.N = N; src.row_off = 0;
    hipLaunchKernelGGL(cat_cast_transpose_kernel, dim3(cdiv(2 * N, 64), cdiv(K0, 64)), dim3(256), 0, s, src, 2 * N, K0, e->dcat_b.r(), e->dcat_b.ld,
                       tr ? e->dcat_b.t() : (__bf16*)nullptr, e->dcat_b.ldt);
    LAUNCH_CHECK();
    e->dcat_b_ok = true;
    e->fake_cat_valid = false;                 // the float32 image was not built
  } else {
  if (split) {
    // split first layer (FirstSplit): only the adversarial columns of the two halves are gathered (58 of 483 columns at cfg2)
    e->ld_adv2 = (e->Da + 3) & ~3;
    CHK(e->adv2.ensure((size_t)2 * N * e->ld_adv2 * sizeof(float)));
    // (the valid-frame count rides in this launch when it is not known yet: fused single-GPU call)
    const bool tv_ride = e->early && e->opt_launch_riders && !comm_on(e) && !tv_side && !tv_known;
    const bool any_ride = tv_ride || tv_ride_dp || unnorm;
    hipLaunchKernelGGL(build_adv_kernel, dim3(cdiv(2 * N * (e->ld_adv2 / 4), 256) + (any_ride ? 1 : 0)), dim3(256), 0, s, y_static, y_hat_static,
                       e->Ds, e->d_adv_cols, e->Da, e->adv2.as<float>(), e->ld_adv2, N, 2 * N, any_ride ? mask : (const float*)nullptr,
                       (int)N, e->tv_override, e->sc(), unnorm ? &e->sc()->tv_sum : tv_ride_dp ? e->comm_tv.as<double>() : (double*)nullptr);
    LAUNCH_CHECK();
    if (tv_ride) { e->tv_mask = mask; e->tv_n = N; e->tv_ovr = e->tv_override; }
    if (tv_ride_dp) CHK(comm_tv_sent(e, s));          // all-reduce of the count on the communicator's stream, joined in front of the head
    e->adv2_fake_ok = true; e->adv2_yhs = y_hat_static;
    e->fake_cat_valid = false;
    fs.x = x; fs.ldx = cx_pitch(e); fs.cd = cond_dim(e); fs.adv = e->adv2.as<float>(); fs.ld_adv = e->ld_adv2; fs.wrap = N;
    fs.xp = nullptr; fs.ldxp = 0;
    if (tr) CHK(pitched_rows(e, 0, x, fs.ldx, fs.cd, N, &fs.xp, &fs.ldxp, s));
  } else {
  CHK(e->dcat.ensure((size_t)2 * N * ldc * sizeof(float)));
  if (e->cfg.discriminator_linguistic_condition && x && cond_dim(e) > 0) {
    hipLaunchKernelGGL(build_cat2_kernel, dim3(cdiv(N * K0, 256)), dim3(256), 0, s, x, cond_dim(e), y_static, y_hat_static, e->Ds,
                       e->d_adv_cols, e->Da, e->dcat.as<float>(), ldc, N);
    LAUNCH_CHECK();
  } else {
    CHK(build_cat(e, x, y_static, e->Ds, 0, N, ldc, s));
    CHK(build_cat(e, x, y_hat_static, e->Ds, N, N, ldc, s));
  }
  e->fake_cat_valid = true; e->fake_cat_x = x; e->fake_cat_yhs = y_hat_static;
  e->adv2_fake_ok = false;
  }
  }
  e->dcat_b_x = x; e->dcat_b_yhs = y_hat_static;
  // a recurrent discriminator (LSTMRNN in the discriminator slot, train.py:773-774): the natural and the generated sequences run as ONE
  // batch of 2B sequences through its stack (lengths twice), the fused head reads the top layer's output (hidden2out is its weight)
  const bool d_rec = has_lstm_body(D.d.arch);
  const bool fused = !d_rec && d_fused_ok(e, b16, 2 * N);        // layers 1 .. L-1 + the head as ONE launch (dstack_f32.hip.h)
  const float* rec_top = nullptr;
  int rec_ld = 0;
  if (d_rec) {
    CHK(lstm_check_lengths(e, B, T));
    CHK(lstm_stack_forward(e, GT_ROLE_D, e->dcat.as<float>(), ldc, 2 * B, T, passes, 2, s, &rec_top, &rec_ld));
  } else if (b16) {
    CHK(refresh_shadows(e, GT_ROLE_D, false, s));
    CHK(stack_forward_b16(e, GT_ROLE_D, e->dcat_b.r(), e->dcat_b.ld, 2 * N, e->d_actb, passes, 2, N, e->d_specs, tr, s));
  } else {
    CHK(stack_forward(e, GT_ROLE_D, split ? nullptr : e->dcat.as<float>(), ldc, 2 * N, e->d_act, passes, 2, N, e->d_specs, s, split ? &fs : nullptr,
                      fused ? 1 : 1 << 30));
  }
  const int H = d_rec ? rec_ld : D.d.hidden_dim;
  if (d_rec) CHK(e->dl_dout.ensure((size_t)2 * 2 * N * H * sizeof(float)));
  if (tr && !D.d.grads) return fail(GT_ERR_STATE, "phase == \"train\" but the discriminator was bound without grads");
  CHK(e->dzA.ensure((size_t)2 * N * std::max(H, 1) * sizeof(float)));
  CHK(e->dzB.ensure((size_t)2 * N * std::max(H, 1) * sizeof(float)));
  // fused call: losses and counts are final after the head (the gradient norm is not: reported as 0), so the head's
  // reduction kernel also writes the result struct and the scalars start their way to the host right behind it
  const bool plain_early = e->early && !comm_on(e), comm_early = e->early && comm_on(e);
  if (tv_side) CHK(side_join(e, s));
  // data parallel + riders: the head reads the all-reduced count where the collective left it (and files it in the step's scalars):
  // no conversion launch between the join and the head
  const unsigned d_ticket = plain_early ? take_ticket(e) : 0;
  const double* head_tv = nullptr;
  if (unnorm) {      // the normaliser arrives with the sums (finalize_d_kernel on the communicator's stream files it); valid from the join on
    e->tv_mask = mask; e->tv_n = N; e->tv_ovr = e->tv_override;
  } else if (comm_on(e) && e->opt_launch_riders && e->tv_inflight && !tv_known) {
    CHK(comm_tv_join(e, s));
    head_tv = e->comm_tv.as<double>();
    e->tv_mask = mask; e->tv_n = N; e->tv_ovr = e->tv_override;
  } else CHK(ensure_tv(e, mask, N, s));
  if (b16 && tr) CHK(e->dz_b[0].ensure(2 * N, H, true));
  if (d_rec)
    CHK(run_head(e, HEAD_D_STEP, (const void*)rec_top, H, 2 * N, N, mask, N, eps, tr, e->dl_dout.as<float>(), no_drop(), true, s,
                 plain_early ? early_res_target(e) : nullptr, 0, nullptr, true, nullptr, head_tv, d_ticket, unnorm, false));
  else if (fused)
    CHK(run_dstack(e, DSTACK_D_STEP, 2 * N, N, mask, N, eps, tr, nullptr, s, plain_early ? early_res_target(e) : nullptr, nullptr, head_tv, d_ticket, unnorm));
  else
  CHK(run_head(e, HEAD_D_STEP, b16 ? (const void*)e->d_actb.back().r() : (const void*)e->d_act.back().as<float>(), H, 2 * N, N, mask, N, eps, tr,
               e->dzA.as<float>(), e->d_specs.back(), true, s, plain_early ? early_res_target(e) : nullptr, b16 ? e->d_actb.back().ld : 0,
               (b16 && tr) ? &e->dz_b[0] : nullptr, true, nullptr, head_tv, d_ticket, unnorm));
  e->early_done = false;
  if (plain_early) CHK(post_early_results(e, s, d_ticket));
  if (comm_early) CHK(comm_early_results(e, GT_ROLE_D, unnorm ? &e->sc()->tv_sum : &e->sc()->s_real, unnorm ? 5 : 4, 0.f, 0.f, 0.f, s));
  if (tr) {
    CHK(comm_grads_ready(e, GT_ROLE_D, D.last.dW, (long)D.last.in * D.last.out + D.last.out, s));
    // keep dloss_d/dy_hat_static only when y_hat_static is the tensor apply_generator produced
    // (the autograd graph in the reference, train.py:265) and a generator with grads exists
    Net& G = e->net[GT_ROLE_G];
    const bool want_leak = G.bound && G.d.grads && e->g_pass_valid && y_hat_static == e->last_yhs && e->N == N;
    if (want_leak && e->leak_pending)
      return fail(GT_ERR_STATE, "update_discriminator called twice without optimizer_g.zero_grad() (train.py:538)");
    float* leak = nullptr;
    if (want_leak) { CHK(e->leak.ensure((size_t)N * e->Da * sizeof(float))); leak = e->leak.as<float>(); }
    const int col0 = cond_dim(e);
    if (d_rec) {   // through the recurrent stack: weight gradients, and the gradient w.r.t. the [x | adv] rows when the generator wants it
      float* dx0 = nullptr;
      if (leak) { CHK(e->d_dx0.ensure((size_t)2 * N * K0 * sizeof(float))); dx0 = e->d_dx0.as<float>(); }
      CHK(lstm_stack_backward(e, GT_ROLE_D, e->dcat.as<float>(), ldc, 2 * B, T, passes, 2, true, dx0, s));
      if (leak) {   // the generated rows' adversarial columns (train.py:265, 274)
        hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * e->Da, 256)), dim3(256), 0, s, dx0 + N * K0, K0, col0, (const int*)nullptr, leak, e->Da, 0,
                           (int)N, e->Da);
        LAUNCH_CHECK();
      }
    } else if (b16) {   // the head wrote its seed gradient as the top dZ image, both orientations
      CHK(stack_backward_b16(e, GT_ROLE_D, e->dcat_b.t(), e->dcat_b.ldt, 2 * N, e->d_actb, e->d_specs, 0, true, leak, e->Da, col0, e->Da, N, N, s));
    } else {
      CHK(stack_backward(e, GT_ROLE_D, split ? nullptr : e->dcat.as<float>(), ldc, 2 * N, e->d_act, e->d_specs, e->dzA.as<float>(),
                         e->dzB.as<float>(), true, leak, e->Da, col0, e->Da, N, N, s, split ? &fs : nullptr));
    }
    D.grads_dirty = true;
    if (want_leak) { e->leak_pending = true; e->leak_unnorm = unnorm; }
  }
  // data parallel: the rest of D's gradient + the four loss / count sums, then the step stream waits for the communicator
  CHK(comm_finish_step(e, GT_ROLE_D, tr, &e->sc()->s_real, comm_early ? 0 : 4, s));
  e->d_begin_done = true;
  return GT_OK;
}

static void fill_d_result(const StepResults* h, gt_d_result* out) {
  out->loss_d = h->loss_d; out->loss_fake_d = h->loss_fake_d; out->loss_real_d = h->loss_real_d;
  out->real_correct_count = h->real_correct; out->fake_correct_count = h->fake_correct;
  out->grad_norm = h->gnorm_d;
}
static void fill_g_result(const StepResults* h, gt_g_result* out) {
  out->loss_mse = h->loss_mse; out->loss_mge = h->loss_mge; out->loss_adv = h->loss_adv;
  out->loss_g = h->loss_g; out->grad_norm = h->gnorm_g;
}

extern "C" int gt_update_discriminator_end(gt_engine* e, int train, gt_d_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null argument");
  if (!e->d_begin_done) return fail(GT_ERR_STATE, "gt_update_discriminator_end without _begin");
  hipStream_t s = (hipStream_t)stream;
  e->d_begin_done = false;
  if (!out) {   // deferred: enqueue everything, synchronise nothing; gt_update_discriminator_result collects
    if (e->early_done) return fail(GT_ERR_STATE, "deferred results are a split-phase feature");
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), train ? 0 : 1);
    LAUNCH_CHECK();
    return post_deferred_results(e, GT_ROLE_D, s);
  }
  if (e->early_done) {
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    CHK(wait_early_results(e));                   // only the scalars; backward + step stay queued
    e->early_done = false;
  } else {
    if (train) CHK(optimizer_step(e, GT_ROLE_D, &e->sc()->gnorm2_d, s));
    hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), train ? 0 : 1);
    LAUNCH_CHECK();
    CHK(fetch_results(e, s));
  }
  fill_d_result(e->h_res, out);
  return GT_OK;
}
extern "C" int gt_update_discriminator_result(gt_engine* e, gt_d_result* out) {
  if (!e || !out) return fail(GT_ERR_INVALID, "null argument");
  if (!e->def_pending[GT_ROLE_D]) return fail(GT_ERR_STATE, "no deferred discriminator result pending");
  HIPCHK(hipEventSynchronize(e->ev_def[GT_ROLE_D]));
  e->def_pending[GT_ROLE_D] = false;
  fill_d_result(e->h_def[GT_ROLE_D], out);
  return GT_OK;
}

extern "C" int gt_update_discriminator(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                                       const float* mask, int B, int T, int train, float eps, gt_d_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->early = true;
  int r = gt_update_discriminator_begin(e, x, y_static, y_hat_static, mask, B, T, train, eps, stream);
  e->early = false;
  if (r != GT_OK) { e->early_done = false; e->d_unnorm = false; e->ticket_wait = 0; return r; }
  return gt_update_discriminator_end(e, train, out, stream);
}

// ------------------------------------------------------------------------------------------
// update_generator
// ------------------------------------------------------------------------------------------
// e->partial: [0, 1024) MGE partials, [1024, 2048) MSE partials, [2048, ...) the optimizer's squared-norm partials.
// deferred_blocks != null: the per-block partial sums stay in the MSE region and *deferred_blocks says how many -- the
// caller folds their reduction into a later launch (finalize_g_kernel) instead of paying a launch for it here.
static int sum_sqerr(gt_engine* e, const float* a, int lda, const float* b, int ldb, const float* mask, long rows, int D,
                     double* out, float* g, int ldg, float gscale, hipStream_t s, int* deferred_blocks = nullptr) {
  const int nblk = (int)std::min<long>(1024, cdiv(rows * D, RED_THREADS * 4));
  CHK(e->partial.ensure(4096 * sizeof(double)));
  double* part = e->partial.as<double>() + 1024;
  hipLaunchKernelGGL(masked_sqerr_kernel, dim3(nblk), dim3(RED_THREADS), 0, s, a, lda, b, ldb, mask, rows, D, part, g, ldg, gscale, e->sc());
  LAUNCH_CHECK();
  if (deferred_blocks) { *deferred_blocks = nblk; return GT_OK; }
  hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, part, nblk, out);
  LAUNCH_CHECK();
  return GT_OK;
}

// backward of G from the gradient at y_hat_static (gs) [+ masked-MSE term at y_hat]
static int generator_backward(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* mask,
                              float mse_w, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = e->N;
  const int B = e->B, T = e->T, Do = G.d.out_dim;
  // dloss/dy_hat is an engine buffer: for the MLP stacks its row pitch is rounded up to 4 floats, so that the last layer's
  // two backward products (K = out_dim = 187 for the acoustic model) take the 16-byte loader and the 64x64 tiles; the
  // pad column is never read as data (K tail / row clamp of the GEMM loader)
  const bool mlp_body = !(has_lstm_body(G.d.arch) || G.d.arch == GT_ARCH_SRU);
  const int ldgy = mlp_body && (is_i2o(G.d.arch) || e->g_used_mlpg) ? (Do + 3) & ~3 : Do;
  CHK(e->gy.ensure((size_t)N * ldgy * sizeof(float)));
  float* gy = e->gy.as<float>();
  const float* gs = e->gs.as<float>();
  if (is_i2o(G.d.arch)) {
    const int sd = G.d.static_dim;
    if (G.d.arch == GT_ARCH_IN2OUT_RNN) mse_w = 0.f;   // y_hat is the input x there: the MSE term has no path into G
    CHK(e->dgx.ensure((size_t)N * sd * sizeof(float)));
    CHK(e->dtz.ensure((size_t)N * sd * sizeof(float)));
    hipLaunchKernelGGL(highway_backward_kernel, dim3(cdiv(N * sd, 256)), dim3(256), 0, s, gs, sd, e->tx.as<float>(), sd,
                       e->gx.as<float>(), sd, e->dgx.as<float>(), sd, e->dtz.as<float>(), sd, N, sd);
    LAUNCH_CHECK();
    CHK(linear_backward_weight(e->dtz.as<float>(), sd, x, G.d.in_dim, N, sd, sd, G.gate.dW, G.gate.db, G.grads_dirty, e->slabs,
                               e->colp, s, &e->sdefer[GT_ROLE_G]));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.gate.dW, (long)sd * sd + sd, s));
    CHK(mlpg_backward(e, e->dgx.as<float>(), sd, e->d_scol_i2o, e->d_sstride_i2o, sd, gy, ldgy, B, T, mse_w, y_hat, y, Do, mask, s));
  } else if (e->g_used_mlpg) {
    CHK(mlpg_backward(e, gs, e->Ds, e->d_scol, e->d_sstride, e->Ds, gy, ldgy, B, T, mse_w, y_hat, y, Do, mask, s));
  } else {
    // no parameter generation: y_hat_static == y_hat, the gradient passes straight through,
    // plus the masked-MSE gradient (which also yields loss_mse's sum)
    if (mse_w != 0.f) CHK(sum_sqerr(e, y_hat, Do, y, Do, mask, N, Do, &e->sc()->s_mse, gy, Do, mse_w, s));
    hipLaunchKernelGGL(slab_reduce_kernel, dim3(cdiv(N * Do, 256)), dim3(256), 0, s, gs, 0L, 1, N * Do, gy, mse_w != 0.f ? 1 : 0);
    LAUNCH_CHECK();
  }
  if (G.d.last_sigmoid && !is_i2o(G.d.arch)) {      // y_hat = sigmoid(last layer) (models.py:141, 167, 190, 213): through s (1 - s)
    hipLaunchKernelGGL(sigmoid_grad_kernel, dim3(cdiv(N * Do, 256)), dim3(256), 0, s, gy, ldgy, y_hat, Do, N, Do);
    LAUNCH_CHECK();
  }
  if (has_lstm_body(G.d.arch) || G.d.arch == GT_ARCH_SRU) {
    CHK(has_lstm_body(G.d.arch) ? lstm_backward(e, x, gy, B, T, s) : sru_backward(e, x, gy, B, T, s));
    G.grads_dirty = true;
    return GT_OK;
  }
  if (use_b16(e, GT_ROLE_G)) {
    // dloss/dy_hat -> bf16 image (both orientations); last_linear: dW = gyT . H_topT^T, dZ_top = (gy . W_lastT^T) (.) f'(H_top)
    const int H = G.d.hidden_dim;
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose(gy, ldgy, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->g_actb.back();
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, G.last.in, G.last.dW, G.last.db, G.grads_dirty, e->slabs, s,
                        &e->sdefer[GT_ROLE_G]));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * G.last.in + Do, s));
    const LinShadow& ws = e->wsh[GT_ROLE_G][G.hidden.size()];
    CHK(e->dz_b[0].ensure(N, H, true));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = ws.wt.as<__bf16>(); g.ldb = ws.ldwt; g.M = (int)N; g.N = H; g.K = Do;
    g.epi = B16_BWD_DATA; g.act = ACT_LEAKY_DROPOUT; g.H = top.r(); g.ldh = top.ld; g.drop = e->g_specs.back();
    g.Cb = e->dz_b[0].r(); g.ldcb = e->dz_b[0].ld; g.CbT = e->dz_b[0].t(); g.ldcbt = (int)e->dz_b[0].ldt;
    CHK(launch_gemm_b16(g, 1, s));
    CHK(stack_backward_b16(e, GT_ROLE_G, e->xin_b.t(), e->xin_b.ldt, N, e->g_actb, e->g_specs, 0, true, nullptr, 0, 0, 0, 0, 0, s));
    G.grads_dirty = true;
    return GT_OK;
  }
  // last_linear: dW = gy^T H_top, db ; dZ_top = (gy W_last) (.) f'(H_top)
  const Lin& Lt = G.hidden.back();
  const int H = G.d.hidden_dim;
  CHK(e->dzA.ensure((size_t)2 * N * H * sizeof(float)));
  CHK(e->dzB.ensure((size_t)2 * N * H * sizeof(float)));
  const GemmArgs nn_last = backward_data_args(gy, ldgy, G.last.W, G.last.in, 0, e->dzA.as<float>(), H, N, Do, H, ACT_LEAKY_DROPOUT,
                                              e->g_act.back().as<float>(), H, e->g_specs.back());
  bool rode = false;
  CHK(linear_backward_weight(gy, ldgy, e->g_act.back().as<float>(), Lt.out, N, Do, G.last.in, G.last.dW, G.last.db, G.grads_dirty,
                             e->slabs, e->colp, s, &e->sdefer[GT_ROLE_G], &nn_last, &rode));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * G.last.in + Do, s));
  if (!rode) CHK(launch_gemm(GEMM_NN, nn_last, 1, s));
  // The first layer's weight gradient reads G's input as its frame operand.  When the discriminator's input image of
  // this step holds the very same x (linguistic conditioning on the generator's own input, no noise channels), the x
  // columns of its rows are a bit-exact copy with a 16-byte row pitch: use it, and the product takes the 16-byte loader.
  // The first layer's weight gradient reads G's input as its n-contiguous frame operand: with a 16-byte pitch it takes the
  // 16-byte loader and the 64 x 64 tiles.  The caller's tensor when it has one, else the discriminator's input image of this step
  // when that holds the very same x (bit-exact copy), else a pitched copy made once per step (shared with the split first layer
  // of D when it conditions on the same tensor).
  const float* xin = x;
  int ldxin = gx_pitch(e);
  if (!gemm_vec_ok(x, ldxin) && tl_gemm_prec == PREC_F32) {
    if (e->fake_cat_valid && e->fake_cat_x == x && e->cfg.discriminator_linguistic_condition && cond_dim(e) == G.d.in_dim && e->dcat.p) {
      ldxin = (d_in_dim(e) + 3) & ~3;
      xin = e->dcat.as<float>() + N * ldxin;      // the generated half: the one that is valid whenever fake_cat_valid is
    } else {
      const int slot = (e->pitched[0].src == x && e->pitched[0].ld == gx_pitch(e) && e->pitched[0].cols == G.d.in_dim && e->pitched[0].step == e->step_counter) ? 0 : 1;
      CHK(pitched_rows(e, slot, x, gx_pitch(e), G.d.in_dim, N, &xin, &ldxin, s));
    }
  }
  CHK(stack_backward(e, GT_ROLE_G, xin, ldxin, N, e->g_act, e->g_specs, e->dzA.as<float>(), e->dzB.as<float>(), true,
                     nullptr, 0, 0, 0, 0, 0, s));
  G.grads_dirty = true;
  return GT_OK;
}

extern "C" int gt_update_generator_begin(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* y_static,
                                         const float* y_hat_static, float adv_w, const float* mask, int B, int T, int train,
                                         float mse_w, float mge_w, float eps, void* stream) {
  CHK(check_common(e, B, T));
  Net& G = e->net[GT_ROLE_G];
  Net& D = e->net[GT_ROLE_D];
  if (!G.bound) return fail(GT_ERR_STATE, "generator not bound");
  if (!y || !y_hat || !y_static || !y_hat_static || !mask) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  const long N = (long)B * T;
  const bool tr = train != 0;
  if (tr) {
    if (!e->g_pass_valid || y_hat != e->last_yhat || y_hat_static != e->last_yhs || N != e->N)
      return fail(GT_ERR_STATE, "update_generator(phase=\"train\") needs the y_hat / y_hat_static returned by the last apply_generator");
    if (!G.d.grads) return fail(GT_ERR_STATE, "phase == \"train\" but the generator was bound without grads");
  }
  const int Do = G.d.out_dim;
  const int Ds = is_i2o(G.d.arch) ? G.d.static_dim : e->Ds;
  if (comm_on(e) && tr && G.grads_dirty)
    return fail(GT_ERR_STATE, "data-parallel step: optimizer_g.zero_grad() must precede update_generator");
  { SlabDefer& sd = e->sdefer[GT_ROLE_G]; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; sd.active = e->early && tr && G.has_opt; }
  CHK(ensure_tv(e, mask, N, s));
  // loss_mse (always reported, train.py:294); its gradient is fused into the MLPG^T kernel
  const bool direct = !is_i2o(G.d.arch) && !e->g_used_mlpg;
  // (the fused single-GPU call reduces the MSE partials inside its finalisation launch: see early_now below)
  const bool early_fold = e->early && !comm_on(e) && !(tr && direct && mse_w != 0.f);
  int mse_blocks = 0, mge_pre_blocks = 0;
  const bool mse_side = early_fold && e->opt_side_overlap && adv_w > 0.f;      // (under the D pass of the adversarial term)
  // launch riders (fused single-GPU call): both reported sums of squares in one launch here; the head's scalar reduction and the
  // step's finalisation as one extra workgroup of the gradient-assembly launch -- four launches become two
  const bool riders = early_fold && e->opt_launch_riders && !mse_side;
  // data parallel: the same launches, the rider then only files the three sums for the collective (nothing is reported from it)
  const bool riders_dp = e->early && comm_on(e) && e->opt_launch_riders && !(tr && direct && mse_w != 0.f);
  if (riders || riders_dp) {
    mse_blocks = (int)std::min<long>(1024, cdiv(N * Do, RED_THREADS * 4));
    mge_pre_blocks = (int)std::min<long>(1024, cdiv(N * Ds, RED_THREADS * 4));
    CHK(e->partial.ensure(4096 * sizeof(double)));
    hipLaunchKernelGGL(g_losses_kernel, dim3(mse_blocks + mge_pre_blocks), dim3(RED_THREADS), 0, s, y_hat, Do, y, Do, Do, mse_blocks,
                       e->partial.as<double>() + 1024, y_hat_static, Ds, y_static, Ds, Ds, e->partial.as<double>(), mask, N);
    LAUNCH_CHECK();
  } else if (!(tr && direct && mse_w != 0.f)) {
    if (mse_side) CHK(side_fork(e, s));
    CHK(sum_sqerr(e, y_hat, Do, y, Do, mask, N, Do, &e->sc()->s_mse, nullptr, 0, 0.f, mse_side ? e->side : s, early_fold ? &mse_blocks : nullptr));
  }
  // adversarial term with the CURRENT (already updated) D weights and a fresh dropout mask (train.py:297-308)
  e->g_has_adv = adv_w > 0.f;
  float* gadv = nullptr;
  int head_blocks = 0;
  if (adv_w > 0.f) {
    if (!D.bound) return fail(GT_ERR_STATE, "adv_w > 0 but no discriminator bound");
    if (D.d.in_dim != d_in_dim(e)) return fail(GT_ERR_DIM, "discriminator in_dim mismatch");
    const int K0 = D.d.in_dim, ldc = (K0 + 3) & ~3;
    const int passes[1] = {2};
    const bool b16 = use_b16(e, GT_ROLE_D);
    const float* cat = nullptr;
    const bool split = d_split_ok(e, x, b16);
    if (!split) CHK(dense_cx(e, &x, N, s));
    FirstSplit fs;
    memset(&fs, 0, sizeof(fs));
    if (split) {     // the generated rows' adversarial columns: kept from the D step of the same batch, or gathered here
      e->ld_adv2 = (e->Da + 3) & ~3;
      CHK(e->adv2.ensure((size_t)2 * N * e->ld_adv2 * sizeof(float)));
      float* fake = e->adv2.as<float>() + N * e->ld_adv2;
      if (!(e->adv2_fake_ok && e->adv2_yhs == y_hat_static)) {
        hipLaunchKernelGGL(build_adv_kernel, dim3(cdiv(N * (e->ld_adv2 / 4), 256)), dim3(256), 0, s, y_hat_static, y_hat_static, Ds, e->d_adv_cols, e->Da,
                           fake, e->ld_adv2, N, N, (const float*)nullptr, 0, 0.f, (StepScalars*)nullptr, (double*)nullptr);
        LAUNCH_CHECK();
        e->adv2_fake_ok = true; e->adv2_yhs = y_hat_static;
      }
      fs.x = x; fs.ldx = cx_pitch(e); fs.cd = cond_dim(e); fs.adv = fake; fs.ld_adv = e->ld_adv2; fs.wrap = N;
    } else if (!b16) {
      CHK(e->dcat.ensure((size_t)2 * N * ldc * sizeof(float)));
      if (!(e->fake_cat_valid && e->fake_cat_x == x && e->fake_cat_yhs == y_hat_static)) {
        CHK(build_cat(e, x, y_hat_static, Ds, N, N, ldc, s));
        e->fake_cat_valid = true; e->fake_cat_x = x; e->fake_cat_yhs = y_hat_static;
      }
      cat = e->dcat.as<float>() + N * ldc;
    }
    const bool d_rec = has_lstm_body(D.d.arch);
    // layers 1 .. L-1, the head and the backward-data chain down to the adversarial columns as ONE launch (dstack_f32.hip.h)
    const bool fused = !d_rec && d_fused_ok(e, b16, N);
    const float* rec_top = nullptr;
    int rec_ld = 0;
    if (d_rec) {
      CHK(lstm_check_lengths(e, B, T));
      CHK(lstm_stack_forward(e, GT_ROLE_D, cat, ldc, B, T, passes, 1, s, &rec_top, &rec_ld));
    } else
    if (b16) {   // the generated half of the bf16 image: rows N .. 2N, kept from the D step of the same batch or built here
      CHK(e->dcat_b.ensure(2 * N, K0, false));
      if (!(e->dcat_b_ok && e->dcat_b_x == x && e->dcat_b_yhs == y_hat_static)) {
        if (e->cfg.discriminator_linguistic_condition && (!x || cond_dim(e) <= 0))
          return fail(GT_ERR_INVALID, "discriminator_linguistic_condition is set but x is null");
        CatSrc src;
        src.x = x; src.cd = cond_dim(e); src.fa = y_hat_static; src.fb = y_hat_static; src.ldf = Ds; src.idx = e->d_adv_cols; src.N = N; src.row_off = N;
        hipLaunchKernelGGL(cat_cast_transpose_kernel, dim3(cdiv(N, 64), cdiv(K0, 64)), dim3(256), 0, s, src, N, K0,
                           e->dcat_b.r() + N * e->dcat_b.ld, e->dcat_b.ld, (__bf16*)nullptr, 0L);
        LAUNCH_CHECK();
        e->dcat_b_ok = true; e->dcat_b_x = x; e->dcat_b_yhs = y_hat_static;
      }
      CHK(refresh_shadows(e, GT_ROLE_D, false, s));        // D has just been stepped (train.py:276 before :307)
      CHK(stack_forward_b16(e, GT_ROLE_D, e->dcat_b.r() + N * e->dcat_b.ld, e->dcat_b.ld, N, e->d_actb, passes, 1, N, e->d_specs, false, s));
    } else {
      CHK(stack_forward(e, GT_ROLE_D, cat, ldc, N, e->d_act, passes, 1, N, e->d_specs, s, split ? &fs : nullptr, fused ? 1 : 1 << 30));
    }
    const int H = d_rec ? rec_ld : D.d.hidden_dim;
    CHK(e->dzA.ensure((size_t)2 * N * H * sizeof(float)));
    CHK(e->dzB.ensure((size_t)2 * N * H * sizeof(float)));
    if (b16 && tr) CHK(e->dz_b[0].ensure(N, H, false));
    if (fused) {
      if (tr) { CHK(e->gadv.ensure((size_t)N * e->Da * sizeof(float))); gadv = e->gadv.as<float>(); }
      CHK(run_dstack(e, DSTACK_G_ADV, N, N, mask, N, eps, tr, gadv, s, nullptr, riders || riders_dp ? &head_blocks : nullptr, nullptr, 0, false));
    } else
    if (d_rec) {
      CHK(e->dl_dout.ensure((size_t)2 * N * H * sizeof(float)));
      CHK(run_head(e, HEAD_G_ADV, (const void*)rec_top, H, N, N, mask, N, eps, tr, e->dl_dout.as<float>(), no_drop(), false, s, nullptr, 0, nullptr,
                   false, riders || riders_dp ? &head_blocks : nullptr, nullptr, 0, false, false));
    } else
    CHK(run_head(e, HEAD_G_ADV, b16 ? (const void*)e->d_actb.back().r() : (const void*)e->d_act.back().as<float>(), H, N, N, mask, N, eps, tr,
                 e->dzA.as<float>(), e->d_specs.back(), false, s, nullptr, b16 ? e->d_actb.back().ld : 0, (b16 && tr) ? &e->dz_b[0] : nullptr, false,
                 riders || riders_dp ? &head_blocks : nullptr));
    if (tr && !fused) {
      CHK(e->gadv.ensure((size_t)N * e->Da * sizeof(float)));
      gadv = e->gadv.as<float>();
      const int col0 = cond_dim(e);
      if (d_rec) {   // back through the recurrent stack to the generated rows' adversarial columns; no weight gradients (train.py:307-308)
        CHK(e->d_dx0.ensure((size_t)2 * N * K0 * sizeof(float)));
        float* dx0 = e->d_dx0.as<float>();
        CHK(lstm_stack_backward(e, GT_ROLE_D, cat, ldc, B, T, passes, 1, false, dx0, s));
        hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N * e->Da, 256)), dim3(256), 0, s, dx0, K0, col0, (const int*)nullptr, gadv, e->Da, 0, (int)N,
                           e->Da);
        LAUNCH_CHECK();
      } else if (b16) {
        CHK(stack_backward_b16(e, GT_ROLE_D, nullptr, 0, N, e->d_actb, e->d_specs, 0, false, gadv, e->Da, col0, e->Da, 0, N, s));
      } else {
        CHK(stack_backward(e, GT_ROLE_D, cat, ldc, N, e->d_act, e->d_specs, e->dzA.as<float>(), e->dzB.as<float>(), false, gadv,
                           e->Da, col0, e->Da, 0, N, s));
      }
    }
  }
  // MGE loss + gradient assembly at y_hat_static
  const bool early_ok = e->early && !(tr && direct && mse_w != 0.f);
  const bool early_now = early_ok && !comm_on(e), comm_early = early_ok && comm_on(e) && e->opt_comm_early_g;
  int mge_blocks = 0;
  e->early_done = false;
  {
    const int nblk = (int)std::min<long>(1024, cdiv(N * Ds, RED_THREADS * 4));
    CHK(e->partial.ensure(4096 * sizeof(double)));
    float* gs = nullptr;
    if (tr) { CHK(e->gs.ensure((size_t)N * Ds * sizeof(float))); gs = e->gs.as<float>(); }
    const float* leak = (tr && e->leak_pending) ? e->leak.as<float>() : nullptr;
    GFinalize fin;
    memset(&fin, 0, sizeof(fin));
    const bool rid = riders || riders_dp;
    if (rid) {       // (riders implies early_now: the fused single-GPU call; riders_dp: sums only)
      fin.on = 1; fin.sc = e->sc(); fin.out = riders ? early_res_target(e) : (StepResults*)nullptr; fin.adv_w = adv_w; fin.mse_w = mse_w; fin.mge_w = mge_w;
      fin.has_adv = e->g_has_adv ? 1 : 0;
      fin.part_mge = e->partial.as<double>(); fin.n_mge = mge_pre_blocks;
      fin.part_mse = e->partial.as<double>() + 1024; fin.n_mse = mse_blocks;
      fin.hp = head_blocks ? e->headp.as<HeadPartials>() : (const HeadPartials*)nullptr; fin.n_hp = head_blocks;
      if (riders) { fin.ticket_value = take_ticket(e); fin.ticket = fin.ticket_value ? e->ticket_dev() : (unsigned*)nullptr; }
    }
    if (tr || !rid)
      hipLaunchKernelGGL(static_grad_kernel, dim3(nblk + (rid ? 1 : 0)), dim3(RED_THREADS), 0, s, y_hat_static, Ds,
                         y_static, Ds, mask, N, Ds, mge_w, e->d_adv_inv, leak, e->Da, gadv, e->Da, adv_w, gs, Ds,
                         rid ? (double*)nullptr : e->partial.as<double>(), e->sc(), fin, leak && e->leak_unnorm ? 1 : 0);
    else      // phase != "train": no gradient to assemble, the finalisation alone
      hipLaunchKernelGGL(finalize_g_rider_kernel, dim3(1), dim3(RED_THREADS), 0, s, fin);
    LAUNCH_CHECK();
    mge_blocks = nblk;
    if (riders) CHK(post_early_results(e, s, fin.ticket_value));
    else if (!early_now && !riders_dp) {   // the split-phase (data-parallel) caller all-reduces the sum itself: it must exist now
      hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, e->partial.as<double>(), nblk, &e->sc()->s_mge);
      LAUNCH_CHECK();
    }
  }
  if (mse_side && mse_blocks) CHK(side_join(e, s));
  if (early_now && !riders) {   // all four losses are final here; the MGE partials are reduced inside the finalisation launch
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(256), 0, s, e->sc(), early_res_target(e), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0, 1,
                       (const double*)e->partial.as<double>(), mge_blocks,
                       mse_blocks ? (const double*)(e->partial.as<double>() + 1024) : (const double*)nullptr, mse_blocks);
    LAUNCH_CHECK();
    CHK(post_early_results(e, s));
  }
  if (comm_early) CHK(comm_early_results(e, GT_ROLE_G, &e->sc()->s_adv, 3, adv_w, mse_w, mge_w, s));
  if (tr) {
    CHK(generator_backward(e, e->last_x, y, y_hat, mask, mse_w, s));  // G's own input (cat(x, z), train.py:542)
    e->leak_pending = false; e->leak_unnorm = false;
  }
  CHK(comm_finish_step(e, GT_ROLE_G, tr, &e->sc()->s_adv, comm_early ? 0 : 3, s));
  e->g_begin_done = true;
  return GT_OK;
}

extern "C" int gt_update_generator_end(gt_engine* e, int train, float adv_w, float mse_w, float mge_w, gt_g_result* out,
                                       void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null argument");
  if (!e->g_begin_done) return fail(GT_ERR_STATE, "gt_update_generator_end without _begin");
  hipStream_t s = (hipStream_t)stream;
  e->g_begin_done = false;
  if (!out) {
    if (e->early_done) return fail(GT_ERR_STATE, "deferred results are a split-phase feature");
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0,
                       train ? 0 : 1, (const double*)nullptr, 0, (const double*)nullptr, 0);
    LAUNCH_CHECK();
    return post_deferred_results(e, GT_ROLE_G, s);
  }
  if (e->early_done) {
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    CHK(wait_early_results(e));
    e->early_done = false;
  } else {
    if (train) CHK(optimizer_step(e, GT_ROLE_G, &e->sc()->gnorm2_g, s));
    hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, s, e->sc(), e->res(), adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0,
                       train ? 0 : 1, (const double*)nullptr, 0, (const double*)nullptr, 0);
    LAUNCH_CHECK();
    CHK(fetch_results(e, s));
  }
  fill_g_result(e->h_res, out);
  return fault_seen(e);
}
extern "C" int gt_update_generator_result(gt_engine* e, gt_g_result* out) {
  if (!e || !out) return fail(GT_ERR_INVALID, "null argument");
  if (!e->def_pending[GT_ROLE_G]) return fail(GT_ERR_STATE, "no deferred generator result pending");
  HIPCHK(hipEventSynchronize(e->ev_def[GT_ROLE_G]));
  e->def_pending[GT_ROLE_G] = false;
  fill_g_result(e->h_def[GT_ROLE_G], out);
  return GT_OK;
}

extern "C" int gt_update_generator(gt_engine* e, const float* x, const float* y, const float* y_hat, const float* y_static,
                                   const float* y_hat_static, float adv_w, const float* mask, int B, int T, int train,
                                   float mse_w, float mge_w, float eps, gt_g_result* out, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->early = true;
  int r = gt_update_generator_begin(e, x, y, y_hat, y_static, y_hat_static, adv_w, mask, B, T, train, mse_w, mge_w, eps, stream);
  e->early = false;
  if (r != GT_OK) { e->early_done = false; return r; }
  return gt_update_generator_end(e, train, adv_w, mse_w, mge_w, out, stream);
}

extern "C" int gt_flush_generator_grads(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  Net& G = e->net[GT_ROLE_G];
  if (!G.bound || !G.d.grads || !e->g_pass_valid) return fail(GT_ERR_STATE, "no generator pass to back-propagate");
  hipStream_t s = (hipStream_t)stream;
  tl_gemm_prec = e->matmul_bf16 ? PREC_BF16 : PREC_F32;      // this entry point launches GEMMs without check_common
  CHK(fault_seen(e));
  { SlabDefer& sd = e->sdefer[GT_ROLE_G]; sd.active = false; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; }   // combines run in place here
  const long N = e->N;
  const int Ds = is_i2o(G.d.arch) ? G.d.static_dim : e->Ds;
  CHK(e->gs.ensure((size_t)N * Ds * sizeof(float)));
  HIPCHK(hipMemsetAsync(e->gs.p, 0, (size_t)N * Ds * sizeof(float), s));
  if (e->leak_pending) {
    // scatter leak[:, j] -> gs[:, adv_cols[j]]  (gather with swapped roles: one column at a time is fine here)
    std::vector<int>& cols = e->h_adv_cols;
    for (int j = 0; j < e->Da; ++j) {
      hipLaunchKernelGGL(gather_cols_kernel, dim3(cdiv(N, 256)), dim3(256), 0, s, e->leak.as<float>(), e->Da, j, (const int*)nullptr,
                         e->gs.as<float>(), Ds, cols[j], (int)N, 1);
    }
    if (e->leak_unnorm) hipLaunchKernelGGL(scale_by_inv_tv_kernel, dim3(cdiv(N * Ds, 256)), dim3(256), 0, s, e->gs.as<float>(), N * Ds, e->sc());
    LAUNCH_CHECK();
  }
  CHK(generator_backward(e, e->last_x, e->last_yhat, e->last_yhat, (const float*)nullptr, 0.f, s));
  e->leak_pending = false; e->leak_unnorm = false;
  HIPCHK(hipStreamSynchronize(s));
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// plain forward
// ------------------------------------------------------------------------------------------
extern "C" int gt_model_forward(gt_engine* e, int role, const float* x, const float* R, int B, int T, float* out, float* out2,
                                void* stream) {
  CHK(check_common(e, B, T));
  if (role < 0 || role > 1 || !e->net[role].bound) return fail(GT_ERR_STATE, "model not bound");
  if (!x || !out) return fail(GT_ERR_INVALID, "null tensor");
  hipStream_t s = (hipStream_t)stream;
  Net& n = e->net[role];
  const long N = (long)B * T;
  e->step_counter++;
  std::vector<DropoutSpec> specs;
  if (role == GT_ROLE_G && is_i2o(n.d.arch)) {
    if (!out2) return fail(GT_ERR_INVALID, "In2OutHighwayNet forward returns two tensors");
    e->g_pass_valid = false;
    return generator_forward(e, x, R, B, T, out, out2, false, s, specs);
  }
  const int pass0d[1] = {0};
  if (n.d.arch == GT_ARCH_LSTM || n.d.arch == GT_ARCH_SRU) {
    if (role != GT_ROLE_G) {
      if (n.d.arch != GT_ARCH_LSTM) return fail(GT_ERR_INVALID, "the discriminator slot takes MLP and LSTMRNN networks");
      // the recurrent discriminator's plain forward: its stack, then hidden2out (+ sigmoid)
      CHK(lstm_check_lengths(e, B, T));
      const float* top = nullptr;
      int ld = 0;
      CHK(lstm_stack_forward(e, role, x, n.d.in_dim, B, T, pass0d, 1, s, &top, &ld));
      return linear_forward(top, ld, n.last.W, n.last.in, n.last.b, out, n.d.out_dim, N, n.last.in, n.last.out,
                            n.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
    }
    e->g_pass_valid = false;
    return n.d.arch == GT_ARCH_LSTM ? lstm_forward(e, x, B, T, out, s) : sru_forward(e, x, B, T, out, s);
  }
  const int pass0[1] = {0};
  auto& acts = role == GT_ROLE_G ? e->g_act : e->d_act;
  if (role == GT_ROLE_G) e->g_pass_valid = false;
  if (use_b16(e, role)) {
    auto& actb = role == GT_ROLE_G ? e->g_actb : e->d_actb;
    CHK(e->fwd_b.ensure(N, n.d.in_dim, false));
    CHK(cast_transpose(x, n.d.in_dim, N, n.d.in_dim, e->fwd_b.r(), e->fwd_b.ld, (__bf16*)nullptr, 0, nullptr, false, &e->colp, s));
    CHK(refresh_shadows(e, role, true, s));
    CHK(stack_forward_b16(e, role, e->fwd_b.r(), e->fwd_b.ld, N, actb, pass0, 1, N, specs, false, s));
    const LinShadow& ws = e->wsh[role][n.hidden.size()];
    GemmB16Args g = b16_args();
    g.A = actb.back().r(); g.lda = actb.back().ld; g.B = ws.w.as<__bf16>(); g.ldb = ws.ldw;
    g.M = (int)N; g.N = n.last.out; g.K = n.last.in; g.bias = n.last.b; g.epi = B16_FWD;
    g.act = n.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = out; g.ldc = n.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  CHK(stack_forward(e, role, x, n.d.in_dim, N, acts, pass0, 1, N, specs, s));      // (plain forward: dense rows)
  return linear_forward(acts.back().as<float>(), n.hidden.back().out, n.last.W, n.last.in, n.last.b, out, n.d.out_dim, N, n.last.in,
                        n.last.out, n.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

