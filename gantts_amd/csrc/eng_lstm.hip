// libgantts_hip.so -- recurrent generator (GT_ARCH_LSTM / GT_ARCH_IN2OUT_RNN): persistent and per-step LSTM kernels
#include "engine_internal.hip.h"
#include "lstm_kernels.hip.h"
#include "lstm_seq_kernels.hip.h"
using namespace gt;
// ------------------------------------------------------------------------------------------
// recurrent generator (GT_ARCH_LSTM): forward / backward of the LSTM stack
// ------------------------------------------------------------------------------------------
// Per-role stashes of a recurrent network: the generator's must survive the discriminator passes of a step (its backward runs last),
// so a recurrent discriminator has its own set
struct LstmBufs {
  std::vector<Scratch>& xproj; std::vector<Scratch>& gates; std::vector<Scratch>& cst; std::vector<Scratch>& out; std::vector<Scratch>& outd;
  Scratch& dout; Scratch& hshift;
};
static LstmBufs lstm_bufs(gt_engine* e, int role) {
  if (role == GT_ROLE_G) return LstmBufs{e->l_xproj, e->l_gates, e->l_cst, e->l_out, e->l_outd, e->l_dout, e->l_hshift};
  return LstmBufs{e->dl_xproj, e->dl_gates, e->dl_cst, e->dl_out, e->dl_outd, e->dl_dout, e->dl_hshift};
}
int lstm_check_lengths(gt_engine* e, int B, int T) {
  if ((int)e->h_lengths.size() != B)
    return fail(GT_ERR_STATE, "recurrent network: call with lengths (gt_set_lengths) for this batch of %d sequences "
                "(reference models.py:204-210 packs the batch by `lengths`)", B);
  for (int b = 0; b < B; ++b)
    if (e->h_lengths[b] > T) return fail(GT_ERR_INVALID, "length %d exceeds the padded length %d", e->h_lengths[b], T);
  return GT_OK;
}

static int lstm_launch_steps(gt_engine* e, const Net& G, const LstmBufs& W, int layer, int B, int T, bool backward, const float* dout, hipStream_t s) {
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  const int Bpad = cdiv(B, 32) * 32;
  const size_t st = (size_t)dirs * Bpad * H;              // floats per state array
  CHK(e->l_state.ensure(5 * st * sizeof(float)));          // h0,h1,c0,c1 (ping-pong) + dc
  float* base = e->l_state.as<float>();
  HIPCHK(hipMemsetAsync(base, 0, 5 * st * sizeof(float), s));
  CHK(ensure_dyn_lds((const void*)lstm_fwd_step_kernel, lstm_lds_bytes()));
  CHK(ensure_dyn_lds((const void*)lstm_bwd_step_kernel, lstm_lds_bytes()));
  LstmStepArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.Bpad = Bpad;
  a.lengths = e->d_lengths();
  for (int d = 0; d < dirs; ++d) { a.Whh[d] = G.lstm[layer].d[d].Whh; a.bih[d] = G.lstm[layer].d[d].bih; a.bhh[d] = G.lstm[layer].d[d].bhh; }
  a.xproj = W.xproj[layer].as<float>();
  a.gates = W.gates[layer].as<float>();
  a.cst = W.cst[layer].as<float>();
  a.out = W.out[layer].as<float>();
  a.dout = dout;
  a.dc_state = base + 4 * st;
  for (int step = 0; step < T; ++step) {
    a.step = step;
    if (!backward) {
      const int cur = step & 1;
      a.h_prev = base + (size_t)cur * st;       a.c_prev = base + (2 + (size_t)cur) * st;
      a.h_next = base + (size_t)(cur ^ 1) * st; a.c_next = base + (2 + (size_t)(cur ^ 1)) * st;
      hipLaunchKernelGGL(lstm_fwd_step_kernel, dim3(cdiv(H, 8), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
    } else {
      hipLaunchKernelGGL(lstm_bwd_step_kernel, dim3(cdiv(H, 32), dirs, cdiv(B, 32)), dim3(256), lstm_lds_bytes(), s, a);
    }
  }
  LAUNCH_CHECK();
  return GT_OK;
}

// ---- persistent recurrence (lstm_seq_kernels.hip.h): one launch per layer and pass ----
// Co-resident workgroups per XCD: all workgroups of a launch spin on each other, so the whole grid must be resident
// at once.  The occupancy API may over-report by one block per CU (MI355X_MICROARCH.md, residency): keep that margin
// above one per CU.  The grid is laid out per XCD (seq_group_of), so the bound is per XCD as well.
static int seq_xcds(int* nxcd, int* cus_per_xcd) {
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  static std::map<int, int> cus;
  if (!cus.count(dev)) { hipDeviceProp_t prop; HIPCHK(hipGetDeviceProperties(&prop, dev)); cus[dev] = prop.multiProcessorCount; }
  *nxcd = cus[dev] % 8 == 0 && cus[dev] >= 64 ? 8 : 1;      // MI355X: 8 XCDs x 32 CUs
  *cus_per_xcd = cus[dev] / *nxcd;
  return GT_OK;
}
template <typename K>
static int launch_seq(K kern, size_t lds, LstmSeqArgs& a, hipStream_t s, bool* launched, int block = 256) {
  CHK(ensure_dyn_lds((const void*)kern, lds));
  int per_cu = 0, nxcd = 1, cpx = 1;
  CHK(seq_xcds(&nxcd, &cpx));
  HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, block, lds));
  if (per_cu > 1) per_cu -= 1;
  per_cu = std::min(per_cu, 4);
  const int ngroups = a.dirs * a.nbt;
  const int rounds = cdiv(ngroups, nxcd);                   // groups that share one XCD
  if ((long)a.ncu * rounds > (long)cpx * per_cu) { *launched = false; return GT_OK; }
  a.nxcd = nxcd;
  hipLaunchKernelGGL(kern, dim3(nxcd * a.ncu * rounds), dim3(block), lds, s, a);
  LAUNCH_CHECK();
  *launched = true;
  return GT_OK;
}
// forward: loader waves + the fast gate functions (lstm_seq_kernels.hip.h: 2.15 -> 1.59 us per step in bf16, 2.54 -> 1.92
// in f32 on a cfg3 layer; tools/lstm_sched_bench keeps the round-2 variant as the A/B reference)
template <int HP, int UPC>
static int launch_fwd_seq(LstmSeqArgs& a, int bt, bool bf16, hipStream_t s, bool* launched) {
  if (bf16)
    return bt == 8 ? launch_seq(lstm_fwd_seq_kernel<HP, UPC, 8, PREC_BF16, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 8, true))
                   : launch_seq(lstm_fwd_seq_kernel<HP, UPC, 16, PREC_BF16, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 16, true));
  return bt == 8 ? launch_seq(lstm_fwd_seq_kernel<HP, UPC, 8, PREC_F32, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 8, true))
                 : launch_seq(lstm_fwd_seq_kernel<HP, UPC, 16, PREC_F32, false, true, true>, lstm_fwd_seq_lds<HP, UPC>(), a, s, launched, lstm_fwd_block(UPC, 16, true));
}
// backward: loader waves in both precisions; the tagged exchange where dG travels as bf16 (in f32 it doubles the exchange
// volume and loses).  cfg3 layer, us per step: bf16 2.38 -> 1.85, f32 3.13 -> 2.76 (tools/lstm_sched_bench).
template <int HP>
static int launch_bwd_seq(LstmSeqArgs& a, int bt, bool bf16, hipStream_t s, bool* launched) {
  if (bf16)
    return bt == 8 ? launch_seq(lstm_bwd_seq_kernel<HP, 8, PREC_BF16, true, true>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(8, true))
                   : launch_seq(lstm_bwd_seq_kernel<HP, 16, PREC_BF16, true, true>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(16, true));
  return bt == 8 ? launch_seq(lstm_bwd_seq_kernel<HP, 8, PREC_F32, true, false>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(8, true))
                 : launch_seq(lstm_bwd_seq_kernel<HP, 16, PREC_F32, true, false>, lstm_bwd_seq_lds<HP>(), a, s, launched, lstm_bwd_block(16, true));
}

// Runs one layer's recurrence (forward, or backward when `backward`) as ONE persistent launch when the shape fits
// (H <= 512, grid co-resident); *launched = false leaves the work to the per-step kernels.
static int lstm_launch_seq(gt_engine* e, const Net& G, const LstmBufs& W, bool bf16, int layer, int B, int T, bool backward, const float* dout,
                           hipStream_t s, bool* launched) {
  *launched = false;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  if (!e->lstm_persistent || H > 512 || T < 2) return GT_OK;
  const int HP = H <= 256 ? 256 : 512;
  int nxcd = 1, cpx = 1;
  CHK(seq_xcds(&nxcd, &cpx));
  // batch tile: 16 sequences per group (full MFMA rows) once that already gives every XCD a group; else 8, which
  // halves the exchange volume of a group and spreads the recurrences over more XCDs (their L2s bound the exchange)
  const int bt = dirs * cdiv(B, 16) >= nxcd ? 16 : 8;
  LstmSeqArgs a;
  memset(&a, 0, sizeof(a));
  a.B = B; a.T = T; a.H = H; a.dirs = dirs; a.nbt = cdiv(B, bt);
  a.lengths = e->d_lengths();
  for (int d = 0; d < dirs; ++d) { a.Whh[d] = G.lstm[layer].d[d].Whh; a.bih[d] = G.lstm[layer].d[d].bih; a.bhh[d] = G.lstm[layer].d[d].bhh; }
  a.xproj = W.xproj[layer].as<float>();
  a.gates = W.gates[layer].as<float>();
  a.cst = W.cst[layer].as<float>();
  a.out = W.out[layer].as<float>();
  a.dout = dout;
  a.fault = e->d_fault;
  a.timeout_ticks = 200000000ULL;          // 2 s at 100 MHz: far beyond any real wait, far below the watchdog
  const int ngroups = dirs * a.nbt;
  const size_t xch_n = (size_t)ngroups * (backward ? lstm_bwd_xch_u64(HP) : lstm_fwd_xch_u64(HP)), chk_n = (size_t)ngroups * 256;
  CHK(e->l_xch.ensure((xch_n + chk_n) * sizeof(unsigned long long)));
  HIPCHK(hipMemsetAsync(e->l_xch.p, 0, (xch_n + chk_n) * sizeof(unsigned long long), s));   // no tag / flag of an earlier launch survives
  a.xch = e->l_xch.as<unsigned long long>();
  a.xcc_chk = a.xch + xch_n;              // [group][256]
  a.allow_xcd_local = e->lstm_xcd_local ? 1 : 0;
  if (backward) {
    a.ncu = cdiv(H, 16);
    CHK(HP == 256 ? launch_bwd_seq<256>(a, bt, bf16, s, launched) : launch_bwd_seq<512>(a, bt, bf16, s, launched));
    if (*launched) HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    return GT_OK;
  }
  // forward: 8 hidden units per workgroup (32 workgroups per group at H = 256: one XCD's worth), else 16
  int upc = e->lstm_fwd_upc;
  if (upc != 8 && upc != 16) upc = 8;
  for (; upc <= 16 && !*launched; upc *= 2) {
    a.ncu = cdiv(H, upc);
    if (upc == 8) CHK(HP == 256 ? (launch_fwd_seq<256, 8>(a, bt, bf16, s, launched)) : (launch_fwd_seq<512, 8>(a, bt, bf16, s, launched)));
    else          CHK(HP == 256 ? (launch_fwd_seq<256, 16>(a, bt, bf16, s, launched)) : (launch_fwd_seq<512, 16>(a, bt, bf16, s, launched)));
  }
  if (*launched) HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, s));
  return GT_OK;
}

static bool lstm_b16(const gt_engine* e) { return e->matmul_bf16 && (e->net[GT_ROLE_G].d.hidden_dim & 7) == 0; }
// The LSTM stack of network `role` over nseq sequences (rows = nseq * T of x, row pitch ld_x): X-projections, recurrences, inter-layer
// dropout; stashes X-projections / gates / cell states / layer outputs in the role's buffers.  passes / npass: the dropout sites of the
// row groups (the D step runs the natural and the generated sequences as ONE batch of 2B: two groups of nseq / 2 sequences).
// *top / *ld_top: the top layer's (dropped, if applicable) output.
int lstm_stack_forward(gt_engine* e, int role, const float* x, int ld_x, int nseq, int T, const int* passes, int npass, hipStream_t s,
                       const float** top, int* ld_top) {
  Net& G = e->net[role];
  LstmBufs W = lstm_bufs(e, role);
  const int B = nseq;
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1;
  const float* in = x;
  int ld_in = ld_x;
  // GT_OPT_MATMUL_BF16: the layer inputs go through bf16 images (both orientations: the weight gradients read the
  // transposed one) and W_ih of all directions is one stacked bf16 shadow -- the X-projection of a layer is ONE product
  const bool b16 = role == GT_ROLE_G && lstm_b16(e);
  const bool want_t = G.d.grads != nullptr;
  const int Lc_ = G.d.num_hidden;
  if (b16) {
    e->l_in_b.resize(Lc_ + 1); e->lsh.resize(Lc_ + 1);
    for (int l = 0; l <= Lc_; ++l) {
      LinShadow& w = e->lsh[l];
      if (l == Lc_) {
        w.ldw = pad8(G.last.in); w.ldwt = pad8(G.last.out);
        CHK(w.w.ensure((size_t)G.last.out * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)G.last.in * w.ldwt * 2 + 64));
        CHK(cast_transpose(G.last.W, G.last.in, G.last.out, G.last.in, w.w.as<__bf16>(), w.ldw, w.wt.as<__bf16>(), w.ldwt, nullptr, false, &e->colp, s));
        break;
      }
      const LstmLayerP& L = G.lstm[l];
      w.ldw = pad8(L.in); w.ldwt = pad8(dirs * 4 * H);
      CHK(w.w.ensure((size_t)dirs * 4 * H * w.ldw * 2 + 64)); CHK(w.wt.ensure((size_t)L.in * w.ldwt * 2 + 64));
      for (int d = 0; d < dirs; ++d)
        CHK(cast_transpose(L.d[d].Wih, L.in, 4 * H, L.in, w.w.as<__bf16>() + (size_t)d * 4 * H * w.ldw, w.ldw,
                                  w.wt.as<__bf16>() + (size_t)d * 4 * H, w.ldwt, nullptr, false, &e->colp, s));
    }
  }
  for (int l = 0; l < G.d.num_hidden; ++l) {
    const LstmLayerP& L = G.lstm[l];
    CHK(W.xproj[l].ensure((size_t)N * dirs * 4 * H * sizeof(float)));
    CHK(W.gates[l].ensure((size_t)N * dirs * 4 * H * sizeof(float)));
    CHK(W.cst[l].ensure((size_t)N * dirs * H * sizeof(float)));
    CHK(W.out[l].ensure((size_t)N * dirs * H * sizeof(float)));
    if (b16) {
      B16Img& I = e->l_in_b[l];
      CHK(I.ensure(N, L.in, want_t));
      CHK(cast_transpose(in, ld_in, N, L.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
      GemmB16Args g = b16_args();
      g.A = I.r(); g.lda = I.ld; g.B = e->lsh[l].w.as<__bf16>(); g.ldb = e->lsh[l].ldw;
      g.M = (int)N; g.N = dirs * 4 * H; g.K = L.in; g.epi = B16_FWD; g.act = ACT_NONE;
      g.C = W.xproj[l].as<float>(); g.ldc = dirs * 4 * H;
      CHK(launch_gemm_b16(g, 1, s));
    } else {
      for (int d = 0; d < dirs; ++d)   // Xp[:, d*4H:(d+1)*4H] = X W_ih^T (biases are added in the step kernel)
        CHK(linear_forward(in, ld_in, L.d[d].Wih, L.in, nullptr, W.xproj[l].as<float>() + (size_t)d * 4 * H, dirs * 4 * H, N, L.in,
                           4 * H, ACT_NONE, no_drop(), s));
    }
    bool seq = false;
    CHK(lstm_launch_seq(e, G, W, role == GT_ROLE_G && e->matmul_bf16, l, B, T, false, nullptr, s, &seq));
    if (!seq) CHK(lstm_launch_steps(e, G, W, l, B, T, false, nullptr, s));
    in = W.out[l].as<float>();
    ld_in = dirs * H;
    if (G.training && G.d.dropout > 0.f && l + 1 < G.d.num_hidden) {
      // nn.LSTM(dropout=p): dropout on the outputs of every layer but the last (training only)
      CHK(W.outd[l].ensure((size_t)N * dirs * H * sizeof(float)));
      const long Ng = N / npass;                        // rows of one pass (its own dropout site)
      for (int q = 0; q < npass; ++q) {
        const DropoutSpec ds = drop_spec(e, role, passes[q], l, G.inj[passes[q]][l], dirs * H);
        hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(Ng * dirs * H, 256)), dim3(256), 0, s, in + q * Ng * dirs * H,
                           W.outd[l].as<float>() + q * Ng * dirs * H, Ng, dirs * H, ds);
      }
      LAUNCH_CHECK();
      in = W.outd[l].as<float>();
    }
  }
  *top = in; *ld_top = ld_in;
  return GT_OK;
}

// x (N, in_dim) -> y_hat (N, out_dim): the generator's stack + hidden2out
int lstm_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  CHK(lstm_check_lengths(e, B, T));
  const long N = (long)B * T;
  const int Lc_ = G.d.num_hidden;
  const bool b16 = lstm_b16(e);
  const bool want_t = G.d.grads != nullptr;
  const int passes[1] = {0};
  const float* in = nullptr;
  int ld_in = 0;
  CHK(lstm_stack_forward(e, GT_ROLE_G, x, G.d.in_dim, B, T, passes, 1, s, &in, &ld_in));
  if (b16) {
    B16Img& I = e->l_in_b[Lc_];
    CHK(I.ensure(N, G.last.in, want_t));
    CHK(cast_transpose(in, ld_in, N, G.last.in, I.r(), I.ld, want_t ? I.t() : (__bf16*)nullptr, I.ldt, nullptr, false, &e->colp, s));
    GemmB16Args g = b16_args();
    g.A = I.r(); g.lda = I.ld; g.B = e->lsh[Lc_].w.as<__bf16>(); g.ldb = e->lsh[Lc_].ldw;
    g.M = (int)N; g.N = G.last.out; g.K = G.last.in; g.bias = G.last.b; g.epi = B16_FWD;
    g.act = G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE; g.C = y_hat; g.ldc = G.d.out_dim;
    return launch_gemm_b16(g, 1, s);
  }
  return linear_forward(in, ld_in, G.last.W, G.last.in, G.last.b, y_hat, G.d.out_dim, N, G.last.in, G.last.out,
                        G.d.last_sigmoid ? ACT_SIGMOID : ACT_NONE, no_drop(), s);
}

// gy (N, out_dim) = dL/dy_hat -> parameter gradients of hidden2out and of every LSTM layer
int lstm_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s) {
  Net& G = e->net[GT_ROLE_G];
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, Do = G.d.out_dim, Lc = G.d.num_hidden;
  const bool acc = G.grads_dirty;
  CHK(e->l_dout.ensure((size_t)2 * N * dirs * H * sizeof(float)));
  float* dout = e->l_dout.as<float>();                       // gradient w.r.t. the top layer's output
  const bool b16 = lstm_b16(e) && (int)e->l_in_b.size() == Lc + 1 && (int)e->lsh.size() == Lc + 1;
  if (b16) {
    // hidden2out through the bf16 images: gy -> (gy, gyT); dW = gyT . topT^T, d out_top = gy . W_lastT^T
    CHK(e->gy_b.ensure(N, Do, true));
    CHK(cast_transpose(gy, Do, N, Do, e->gy_b.r(), e->gy_b.ld, e->gy_b.t(), e->gy_b.ldt, nullptr, false, &e->colp, s));
    B16Img& top = e->l_in_b[Lc];
    CHK(weight_grad_b16(e->gy_b.t(), e->gy_b.ldt, top.t(), top.ldt, N, Do, dirs * H, G.last.dW, G.last.db, acc, e->slabs, s));
    CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * dirs * H + Do, s));
    GemmB16Args g = b16_args();
    g.A = e->gy_b.r(); g.lda = e->gy_b.ld; g.B = e->lsh[Lc].wt.as<__bf16>(); g.ldb = e->lsh[Lc].ldwt;
    g.M = (int)N; g.N = dirs * H; g.K = Do; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dout; g.ldc = dirs * H;
    CHK(launch_gemm_b16(g, 1, s));
  } else {
  // hidden2out: dW = gy^T out_top, db, d out_top = gy W
  CHK(linear_backward_weight(gy, Do, e->l_out[Lc - 1].as<float>(), dirs * H, N, Do, dirs * H, G.last.dW, G.last.db, acc, e->slabs,
                             e->colp, s));
  CHK(comm_grads_ready(e, GT_ROLE_G, G.last.dW, (long)Do * dirs * H + Do, s));
  CHK(linear_backward_data(gy, Do, G.last.W, G.last.in, 0, dout, dirs * H, N, Do, dirs * H, ACT_NONE, nullptr, 0, no_drop(), s));
  }
  const int passes[1] = {0};
  return lstm_stack_backward(e, GT_ROLE_G, x, G.d.in_dim, B, T, passes, 1, true, nullptr, s);
}

// From the gradient w.r.t. the top layer's output (first half of the role's `dout` buffer, row pitch dirs * H) down through the
// stack: weight gradients of every layer (want_w), and the gradient w.r.t. the stack's input when dx0 != null ([rows][in_dim], dense)
// -- what a recurrent discriminator hands back to the generator.
int lstm_stack_backward(gt_engine* e, int role, const float* x, int ld_x, int nseq, int T, const int* passes, int npass, bool want_w,
                        float* dx0, hipStream_t s) {
  Net& G = e->net[role];
  LstmBufs W = lstm_bufs(e, role);
  const int B = nseq;
  const long N = (long)B * T;
  const int H = G.d.hidden_dim, dirs = G.d.bidirectional ? 2 : 1, Lc = G.d.num_hidden;
  const bool acc = G.grads_dirty;
  CHK(W.dout.ensure((size_t)2 * N * dirs * H * sizeof(float)));
  CHK(W.hshift.ensure((size_t)N * H * sizeof(float)));
  float* dout = W.dout.as<float>();                          // gradient w.r.t. the current layer's output
  float* dout_other = dout + (size_t)N * dirs * H;
  const bool b16 = role == GT_ROLE_G && lstm_b16(e) && (int)e->l_in_b.size() == Lc + 1 && (int)e->lsh.size() == Lc + 1;
  // (A side stream for a layer's weight-gradient products -- beside the recurrence of the layer below -- was built in round 3, measured not to
  //  pay (cfg3 fp32 25.74 vs 25.68 ms, bf16 19.04 vs 18.14: the products' operand traffic slows every step of the recurrence by what the
  //  overlap hides) and left the library in round 6.)
  hipStream_t ws = s;
  Scratch& wsl = e->slabs;
  Scratch& wcp = e->colp;
  if (b16) e->l_dg_b.resize(Lc);
  for (int l = Lc - 1; l >= 0; --l) {
    const LstmLayerP& L = G.lstm[l];
    bool seq = false;
    CHK(lstm_launch_seq(e, G, W, role == GT_ROLE_G && e->matmul_bf16, l, B, T, true, dout, s, &seq));  // dG overwrites xproj[l]
    if (!seq) CHK(lstm_launch_steps(e, G, W, l, B, T, true, dout, s));
    const float* dG = W.xproj[l].as<float>();
    const bool dropped_in = l > 0 && G.training && G.d.dropout > 0.f;
    if (b16) {
      B16Img& DG0 = e->l_dg_b[l];
      CHK(DG0.ensure(N, dirs * 4 * H, true));
      CHK(cast_transpose(dG, dirs * 4 * H, N, dirs * 4 * H, DG0.r(), DG0.ld, DG0.t(), DG0.ldt, nullptr, false, &e->colp, s));
    }
    if (b16) {
      // dG -> bf16 image in both orientations (one pass), then every product of this layer reads bf16:
      // dW_ih_d = dGT_d . inT^T (+ db from the loader), dW_hh_d = dGT_d . hshiftT^T, d in = dG . W_ihT^T (all directions in ONE product)
      B16Img& DG = e->l_dg_b[l];
      B16Img& I = e->l_in_b[l];
      if (l > 0) {      // the step stream's part first: d(layer input), all directions in ONE product
        GemmB16Args g = b16_args();
        g.A = DG.r(); g.lda = DG.ld; g.B = e->lsh[l].wt.as<__bf16>(); g.ldb = e->lsh[l].ldwt;
        g.M = (int)N; g.N = L.in; g.K = dirs * 4 * H; g.epi = B16_BWD_DATA; g.act = ACT_NONE; g.C = dout_other; g.ldc = L.in;
        CHK(launch_gemm_b16(g, 1, s));
        if (dropped_in) {
          const long Ng = N / npass;
          for (int q = 0; q < npass; ++q) {
            const DropoutSpec ds = drop_spec(e, role, passes[q], l - 1, G.inj[passes[q]][l - 1], dirs * H);
            hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(Ng * dirs * H, 256)), dim3(256), 0, s, dout_other + q * Ng * dirs * H,
                               dout_other + q * Ng * dirs * H, Ng, dirs * H, ds);
          }
          LAUNCH_CHECK();
        }
        std::swap(dout, dout_other);
      }
      for (int d = 0; d < dirs; ++d) {
        const __bf16* dgt = DG.t() + (size_t)d * 4 * H * DG.ldt;
        CHK(weight_grad_b16(dgt, DG.ldt, I.t(), I.ldt, N, 4 * H, L.in, L.d[d].dWih, L.d[d].dbih, acc, wsl, ws));
        hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(4 * H, 64)), dim3(1024), 0, ws, L.d[d].dbih, (long)4 * H, 1, 4 * H,
                           L.d[d].dbhh, 0);
        LAUNCH_CHECK();
        hipLaunchKernelGGL(lstm_shift_kernel, dim3(cdiv(N * H, 256)), dim3(256), 0, ws, W.out[l].as<float>(), dirs * H, d, H, B, T,
                           e->d_lengths(), W.hshift.as<float>());
        LAUNCH_CHECK();
        CHK(e->l_hs_b.ensure(N, H, true));
        CHK(cast_transpose(W.hshift.as<float>(), H, N, H, (__bf16*)nullptr, 0, e->l_hs_b.t(), e->l_hs_b.ldt, nullptr, false, &wcp, ws));
        CHK(weight_grad_b16(dgt, DG.ldt, e->l_hs_b.t(), e->l_hs_b.ldt, N, 4 * H, H, L.d[d].dWhh, nullptr, acc, wsl, ws));
      }
      CHK(comm_grads_ready(e, role, L.d[0].dWih, (long)dirs * (4L * H * L.in + 4L * H * H + 8L * H), ws));
      CHK(comm_flush(e, role, ws));
      continue;
    }
    const float* Xl = l == 0 ? x : (dropped_in ? W.outd[l - 1].as<float>() : W.out[l - 1].as<float>());
    const int ldx = l == 0 ? ld_x : dirs * H;
    for (int d = 0; d < dirs && want_w; ++d) {
      const float* dGd = dG + (size_t)d * 4 * H;
      // dW_ih = dG_d^T X, db_ih = colsum(dG_d) (= db_hh)
      CHK(linear_backward_weight(dGd, dirs * 4 * H, Xl, ldx, N, 4 * H, L.in, L.d[d].dWih, L.d[d].dbih, acc, wsl, wcp, ws));
      // bias_ih and bias_hh always receive the same gradient: keep them equal by copy
      hipLaunchKernelGGL(slab_reduce_small_kernel, dim3(cdiv(4 * H, 64)), dim3(1024), 0, ws, L.d[d].dbih, (long)4 * H, 1, 4 * H,
                         L.d[d].dbhh, 0);
      LAUNCH_CHECK();
      // dW_hh = dG_d^T H_shift (h that entered each frame)
      hipLaunchKernelGGL(lstm_shift_kernel, dim3(cdiv(N * H, 256)), dim3(256), 0, ws, W.out[l].as<float>(), dirs * H, d, H, B, T,
                         e->d_lengths(), W.hshift.as<float>());
      LAUNCH_CHECK();
      CHK(linear_backward_weight(dGd, dirs * 4 * H, W.hshift.as<float>(), H, N, 4 * H, H, L.d[d].dWhh, nullptr, acc, wsl,
                                 wcp, ws));
    }
    // this layer's parameters (both directions: W_ih, W_hh, b_ih, b_hh each) are one contiguous bucket
    if (want_w) {
      CHK(comm_grads_ready(e, role, L.d[0].dWih, (long)dirs * (4L * H * L.in + 4L * H * H + 8L * H), ws));
      if (role == GT_ROLE_G || !e->opt_comm_d_one_msg) CHK(comm_flush(e, role, ws));              // with hidden2out above it: under the recurrence of the layer below
    }
    if (l > 0 || dx0) {   // gradient w.r.t. the layer below's output (or the stack's input): sum over directions of dG_d W_ih_d
      float* const dst = l > 0 ? dout_other : dx0;
      for (int d = 0; d < dirs; ++d) {
        GemmArgs g;
        memset(&g, 0, sizeof(g));
        g.A = dG + (size_t)d * 4 * H; g.lda = dirs * 4 * H; g.B = L.d[d].Wih; g.ldb = L.in; g.C = dst; g.ldc = L.in;
        g.M = (int)N; g.N = L.in; g.K = 4 * H; g.act = ACT_NONE; g.accumulate = d > 0 ? 1 : 0; g.drop = no_drop();
        CHK(launch_gemm(GEMM_NN, g, 1, s));
      }
      if (dropped_in) {   // through the inter-layer dropout of layer l-1 (same Philox sites as the forward)
        const long Ng = N / npass;
        for (int q = 0; q < npass; ++q) {
          const DropoutSpec ds = drop_spec(e, role, passes[q], l - 1, G.inj[passes[q]][l - 1], dirs * H);
          hipLaunchKernelGGL(dropout_apply_kernel, dim3(cdiv(Ng * dirs * H, 256)), dim3(256), 0, s, dout_other + q * Ng * dirs * H,
                           dout_other + q * Ng * dirs * H, Ng, dirs * H, ds);
      }
        LAUNCH_CHECK();
      }
      if (l > 0) std::swap(dout, dout_other);
    }
  }
  return GT_OK;
}


