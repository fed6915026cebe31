// libgantts_hip.so -- errors, launch profiler, engine life cycle, model / optimizer binding, options, dropout sites, fault word (C ABI: include/gantts_hip.h)
#include "engine_internal.hip.h"

using namespace gt;
// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}
extern "C" const char* gt_last_error(void) { return g_err; }
extern "C" const char* gt_version(void) { return "gantts_hip 0.1 (gfx950, f32 MFMA)"; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: remember the largest value set per
// (kernel, device), so that a process driving several GPUs raises the limit on each of them.
int ensure_dyn_lds(const void* kernel, size_t bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, size_t> done;
  int dev = 0;
  HIPCHK(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(mu);
  size_t& cur = done[std::make_pair(kernel, dev)];
  if (bytes > cur) {
    HIPCHK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    cur = bytes;
  }
  return GT_OK;
}

// ------------------------------------------------------------------------------------------
// optional per-launch timing of the GEMM family (HIP events on the launch stream); used by
// bench.py for the live roofline figure.  Off by default: zero overhead on the normal path.
// ------------------------------------------------------------------------------------------
GemmProfiler g_prof;

extern "C" int gt_profile_enable(int on) {
  g_prof.on = on != 0;
  g_prof.only_kind = on >= 2 ? on - 2 : -1;
  return GT_OK;
}
// Drains the recorded launches into per-kernel totals (slot layout: include/gantts_hip.h).
// out_ms[v] = summed kernel time, out_flops[v] = summed algorithmic 2*M*N*K, out_count[v] = launches.
extern "C" int gt_profile_read(double* out_ms, double* out_flops, int64_t* out_count) {
  for (int v = 0; v < GT_PROFILE_SLOTS; ++v) { out_ms[v] = 0; out_flops[v] = 0; out_count[v] = 0; g_prof.last_bytes[v] = 0; }
  for (auto& r : g_prof.recs) {
    if (hipEventSynchronize(r.e1) != hipSuccess) return fail(GT_ERR_HIP, "hipEventSynchronize failed");
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return fail(GT_ERR_HIP, "hipEventElapsedTime failed");
    int v = r.kind * 2 + (r.bn == 128 ? 1 : 0);
    if (r.kind == 5) v = 8;                                   // pair launch
    else if (r.kind == 6) v = 12;                             // weight-gradient pair of a split first layer
    else if (r.kind == 7) v = 14;                             // fused discriminator stack (dstack_f32.hip.h)
    else if (r.kind == GEMM_NT && r.am == GEMM_A_NONE) v = 6;
    else if (r.kind == GEMM_NT && r.am == GEMM_A_LEAKY_PHILOX) v = 7;
    else if (r.kind == GEMM_NT && r.am == GEMM_A_LEAKY_PHILOX_ADDM) v = 9;
    else if (r.kind == GEMM_NT && r.am == GEMM_A_LEAKY_PHILOX_SEG) v = 13;
    else if (r.kind == GEMM_NN && r.am == GEMM_A_NONE) v = 10;
    else if (r.kind == GEMM_NN && r.am == GEMM_A_LEAKY_PHILOX) v = 11;
    if (v < 0 || v >= GT_PROFILE_SLOTS) v = GT_PROFILE_SLOTS - 1;
    out_ms[v] += ms; out_flops[v] += r.flops; out_count[v] += 1; g_prof.last_bytes[v] += r.bytes;
    g_prof.pool.push_back(r.e0); g_prof.pool.push_back(r.e1);
  }
  g_prof.recs.clear();
  return GT_OK;
}
// Algorithmic HBM bytes (every operand once, the result once, fp32) of the launches the last gt_profile_read drained.
extern "C" int gt_profile_bytes(double* out_bytes) {
  for (int v = 0; v < GT_PROFILE_SLOTS; ++v) out_bytes[v] = g_prof.last_bytes[v];
  return GT_OK;
}

static int upload_ints(const std::vector<int>& v, int** dptr) {
  if (*dptr) { (void)hipFree(*dptr); *dptr = nullptr; }
  if (v.empty()) return GT_OK;
  HIPCHK(hipMalloc((void**)dptr, v.size() * sizeof(int)));
  HIPCHK(hipMemcpy(*dptr, v.data(), v.size() * sizeof(int), hipMemcpyHostToDevice));
  return GT_OK;
}

extern "C" int gt_engine_create(const gt_stream_config* cfg, gt_engine** out) {
  if (!cfg || !out) return fail(GT_ERR_INVALID, "null argument");
  if (cfg->n_streams < 1 || cfg->n_streams > GT_MAX_STREAMS) return fail(GT_ERR_INVALID, "n_streams out of range");
  if (cfg->num_windows < 1 || cfg->num_windows > MLPG_MAXW) return fail(GT_ERR_INVALID, "num_windows must be in [1,%d]", MLPG_MAXW);
  gt_engine* e = new gt_engine();
  e->cfg = *cfg;
  // static layout: get_static_stream_sizes (multistream.py:46-53) + per-column source map
  int col = 0, scol_out = 0;
  std::vector<int> static_start, static_size;
  for (int s = 0; s < cfg->n_streams; ++s) {
    const int sz = cfg->stream_sizes[s];
    const bool dyn = cfg->has_dynamic_features[s] != 0;
    const int ss = dyn ? sz / cfg->num_windows : sz;
    static_start.push_back(scol_out);
    static_size.push_back(ss);
    for (int c = 0; c < ss; ++c) {
      e->h_scol.push_back(col + c);
      e->h_sstride.push_back(dyn ? ss : 0);
    }
    col += sz;
    scol_out += ss;
  }
  e->Dout_cfg = col;
  e->Ds = scol_out;
  // adversarial columns: select_streams on the static layout, then drop the first n (train.py:232-242)
  if (cfg->adversarial_streams[0] < 0) {
    for (int c = 0; c < e->Ds; ++c) e->h_adv_cols.push_back(c);
  } else {
    for (int s = 0; s < cfg->n_streams; ++s)
      if (cfg->adversarial_streams[s])
        for (int c = 0; c < static_size[s]; ++c) e->h_adv_cols.push_back(static_start[s] + c);
    if (cfg->mask_nth_mgc_for_adv_loss > 0) {
      if ((size_t)cfg->mask_nth_mgc_for_adv_loss >= e->h_adv_cols.size()) { delete e; return fail(GT_ERR_INVALID, "mask_nth_mgc_for_adv_loss too large"); }
      e->h_adv_cols.erase(e->h_adv_cols.begin(), e->h_adv_cols.begin() + cfg->mask_nth_mgc_for_adv_loss);
    }
  }
  e->Da = (int)e->h_adv_cols.size();
  e->h_adv_inv.assign(e->Ds, -1);
  for (int j = 0; j < e->Da; ++j) e->h_adv_inv[e->h_adv_cols[j]] = j;
  int r;
  if ((r = upload_ints(e->h_scol, &e->d_scol)) || (r = upload_ints(e->h_sstride, &e->d_sstride)) ||
      (r = upload_ints(e->h_adv_cols, &e->d_adv_cols)) || (r = upload_ints(e->h_adv_inv, &e->d_adv_inv))) { delete e; return r; }
  if ((r = e->scal.ensure(1024))) { delete e; return r; }
  if (hipMemset(e->scal.p, 0, 1024) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipMemset failed"); }
  static_assert(sizeof(StepResults) <= 64, "the result ticket sits 64 bytes behind the results");
  if (hipHostMalloc((void**)&e->h_res, 128) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipHostMalloc failed"); }
  memset(e->h_res, 0, 128);
  if (hipHostGetDevicePointer((void**)&e->h_res_dev, e->h_res, 0) != hipSuccess) { (void)hipGetLastError(); e->h_res_dev = nullptr; }
  if (!env_flag("GT_RES_HOSTMAP", true)) e->h_res_dev = nullptr;      // (measurement: results through a device -> host copy instead)
  if (hipMalloc((void**)&e->d_fault, 64) != hipSuccess || hipMemset(e->d_fault, 0, 64) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipMalloc failed"); }
  if (hipHostMalloc((void**)&e->h_fault, 64) != hipSuccess) { delete e; return fail(GT_ERR_HIP, "hipHostMalloc failed"); }
  for (int i = 0; i < 16; ++i) e->h_fault[i] = 0;   // [0] copy of the device word, [1] its mirror by the optimizer kernel, [2 + role] skipped steps
  if (hipHostGetDevicePointer((void**)&e->h_fault_dev, e->h_fault, 0) != hipSuccess) { (void)hipGetLastError(); e->h_fault_dev = nullptr; }
  else e->h_fault_dev += 1;
  *out = e;
  return GT_OK;
}

extern "C" void gt_engine_destroy(gt_engine* e) {
  if (!e) return;
  (void)hipDeviceSynchronize();
  for (auto& s : e->g_act) s.release();
  for (auto& s : e->d_act) s.release();
  for (auto* v : {&e->l_xproj, &e->l_gates, &e->l_cst, &e->l_out, &e->l_outd, &e->dl_xproj, &e->dl_gates, &e->dl_cst, &e->dl_out, &e->dl_outd})
    for (auto& s : *v) s.release();
  e->dl_dout.release(); e->dl_hshift.release(); e->d_dx0.release();
  e->i2o_gout.release();
  (void)gt_comm_destroy(e);
  e->comm_tv.release();
  e->l_state.release(); e->l_dout.release(); e->l_hshift.release(); e->l_xch.release();
  if (e->d_fault) (void)hipFree(e->d_fault);
  if (e->h_fault) (void)hipHostFree(e->h_fault);
  for (int i = 0; i < gt_engine::LEN_RING; ++i) {
    e->len_dev[i].release();
    if (e->len_host[i]) (void)hipHostFree(e->len_host[i]);
    if (e->len_ev[i]) (void)hipEventDestroy(e->len_ev[i]);
  }
  for (auto* v : {&e->s_u, &e->s_h, &e->s_c, &e->s_xdrop, &e->s_xmask, &e->s_wt}) for (auto& s : *v) s.release();
  e->s_du.release(); e->s_dx.release(); e->s_dbias.release();
  Scratch* all[] = {&e->dcat, &e->dzA, &e->dzB, &e->leak, &e->gadv, &e->gs, &e->gy, &e->slabs, &e->colp, &e->partial,
                    &e->headp, &e->headw, &e->gx_dense, &e->cx_dense, &e->dmask, &e->tx, &e->gx, &e->dgx, &e->dtz, &e->dout, &e->scal, &e->mlpg.tmp};
  for (auto* s : all) s->release();
  e->w0pad[0].release(); e->w0pad[1].release();
  e->opt_bar.release(); e->d_pre.release(); e->adv2.release(); e->pitched[0].buf.release(); e->pitched[1].buf.release();
  for (auto* v : {&e->g_actb, &e->d_actb}) for (auto& b : *v) b.release();
  e->xin_b.release(); e->dcat_b.release(); e->gy_b.release(); e->dz_b[0].release(); e->dz_b[1].release(); e->fwd_b.release();
  for (int r = 0; r < 2; ++r) for (auto& w : e->wsh[r]) { w.w.release(); w.wt.release(); }
  for (auto& b : e->l_in_b) b.release();
  for (auto& b : e->s_in_b) b.release();
  e->s_du_b.release();
  for (auto& w : e->ssh) { w.w.release(); w.wt.release(); }
  for (auto& b : e->l_dg_b) b.release();
  e->l_hs_b.release();
  if (e->side) (void)hipStreamDestroy(e->side);
  if (e->ev_side_go) (void)hipEventDestroy(e->ev_side_go);
  if (e->ev_side_done) (void)hipEventDestroy(e->ev_side_done);
  for (auto& w : e->lsh) { w.w.release(); w.wt.release(); }
  e->sdefer[0].pool.release(); e->sdefer[1].pool.release();
  e->mlpg.clear();
  int* ints[] = {e->d_scol, e->d_sstride, e->d_adv_cols, e->d_adv_inv, e->d_scol_i2o, e->d_sstride_i2o};
  for (int* p : ints) if (p) (void)hipFree(p);
  if (e->h_res) (void)hipHostFree(e->h_res);
  if (e->ev_res) (void)hipEventDestroy(e->ev_res);
  for (int r = 0; r < 2; ++r) { if (e->h_def[r]) (void)hipHostFree(e->h_def[r]); if (e->ev_def[r]) (void)hipEventDestroy(e->ev_def[r]); }
  delete e;
}

static long expected_params(const gt_model_desc& d) {
  long n = 0;
  if (has_lstm_body(d.arch)) {
    const int H = d.hidden_dim, dirs = d.bidirectional ? 2 : 1;
    if (d.arch == GT_ARCH_IN2OUT_RNN) n += (long)d.static_dim * d.static_dim + d.static_dim;
    for (int l = 0; l < d.num_hidden; ++l) {
      const int in = l == 0 ? d.in_dim : H * dirs;
      n += (long)dirs * (4L * H * in + 4L * H * H + 8L * H);
    }
    return n + (long)d.out_dim * H * dirs + d.out_dim;
  }
  if (d.arch == GT_ARCH_SRU) {
    const int ncols = d.hidden_dim * (d.bidirectional ? 2 : 1);
    for (int l = 0; l < d.num_hidden; ++l) {
      const int in = l == 0 ? d.in_dim : ncols;
      n += (long)in * ncols * (in == ncols ? 3 : 4) + 2L * ncols;
    }
    return n + (long)d.out_dim * ncols + d.out_dim;
  }
  if (d.arch == GT_ARCH_IN2OUT) n += (long)d.static_dim * d.static_dim + d.static_dim;
  int in = d.in_dim;
  for (int l = 0; l < d.num_hidden; ++l) { n += (long)d.hidden_dim * in + d.hidden_dim; in = d.hidden_dim; }
  n += (long)d.out_dim * in + d.out_dim;
  return n;
}

extern "C" int gt_bind_model(gt_engine* e, int role, const gt_model_desc* desc) {
  if (!e || !desc || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  if (desc->arch < GT_ARCH_MLP || desc->arch > GT_ARCH_IN2OUT_RNN)
    return fail(GT_ERR_INVALID, "unsupported arch %d", desc->arch);
  if (desc->num_hidden < 1 || desc->num_hidden > 16) return fail(GT_ERR_INVALID, "num_hidden must be in [1,16]");
  if (desc->dropout < 0.f || desc->dropout >= 1.f) return fail(GT_ERR_INVALID, "dropout must be in [0,1)");
  if (!desc->params) return fail(GT_ERR_INVALID, "params is null");
  if (desc->n_params != expected_params(*desc))
    return fail(GT_ERR_INVALID, "n_params %ld does not match the architecture (%ld)", (long)desc->n_params, expected_params(*desc));
  if (desc->arch == GT_ARCH_LSTM && desc->hidden_dim < 1) return fail(GT_ERR_INVALID, "hidden_dim must be positive");
  if (role == GT_ROLE_D && ((desc->arch != GT_ARCH_MLP && desc->arch != GT_ARCH_LSTM) || desc->out_dim != 1 || !desc->last_sigmoid))
    return fail(GT_ERR_INVALID, "discriminator must be MLP or LSTMRNN with out_dim=1, last_sigmoid=True (hparams.py:56-64,230-239; train.py:773-774)");
  if (role == GT_ROLE_D && desc->arch == GT_ARCH_LSTM && desc->hidden_dim * (desc->bidirectional ? 2 : 1) > 1024)
    return fail(GT_ERR_INVALID, "recurrent discriminator: hidden_dim x directions > 1024 is not supported by the fused head kernel");
  Net& n = e->net[role];
  n.d = *desc;
  n.hidden.clear();
  float* p = desc->params;
  float* g = desc->grads;
  auto take = [&](int out, int in) {
    Lin l;
    l.in = in; l.out = out;
    l.W = p; l.dW = g; p += (long)out * in; if (g) g += (long)out * in;
    l.b = p; l.db = g; p += out; if (g) g += out;
    return l;
  };
  n.lstm.clear();
  n.sru.clear();
  if (has_lstm_body(desc->arch)) {
    const int H = desc->hidden_dim, dirs = desc->bidirectional ? 2 : 1;
    if (desc->arch == GT_ARCH_IN2OUT_RNN) {
      if (desc->in_dim != desc->out_dim)
        return fail(GT_ERR_DIM, "In2OutRNNHighwayNet returns its input as y_hat (models.py:118): in_dim must equal out_dim");
      n.gate = take(desc->static_dim, desc->static_dim);
    }
    auto adv = [&](float*& wp, float*& gp, long cnt) { wp = p; gp = g; p += cnt; if (g) g += cnt; };
    for (int l = 0; l < desc->num_hidden; ++l) {
      LstmLayerP L;
      memset(&L, 0, sizeof(L));
      L.in = l == 0 ? desc->in_dim : H * dirs;
      for (int dd = 0; dd < dirs; ++dd) {
        adv(L.d[dd].Wih, L.d[dd].dWih, 4L * H * L.in);
        adv(L.d[dd].Whh, L.d[dd].dWhh, 4L * H * H);
        adv(L.d[dd].bih, L.d[dd].dbih, 4L * H);
        adv(L.d[dd].bhh, L.d[dd].dbhh, 4L * H);
      }
      n.lstm.push_back(L);
    }
    n.last = take(desc->out_dim, H * dirs);
    if (role == GT_ROLE_G) {
      e->l_xproj.resize(desc->num_hidden); e->l_gates.resize(desc->num_hidden);
      e->l_cst.resize(desc->num_hidden); e->l_out.resize(desc->num_hidden); e->l_outd.resize(desc->num_hidden);
    } else {
      e->dl_xproj.resize(desc->num_hidden); e->dl_gates.resize(desc->num_hidden);
      e->dl_cst.resize(desc->num_hidden); e->dl_out.resize(desc->num_hidden); e->dl_outd.resize(desc->num_hidden);
    }
  } else if (desc->arch == GT_ARCH_SRU) {
    if (desc->rnn_dropout < 0.f || desc->rnn_dropout >= 1.f) return fail(GT_ERR_INVALID, "rnn_dropout must be in [0,1)");
    if (desc->num_hidden > 8) return fail(GT_ERR_INVALID, "SRURNN: at most 8 layers (two dropout sites per layer)");
    const int ncols = desc->hidden_dim * (desc->bidirectional ? 2 : 1);
    n.sru.clear();
    for (int l = 0; l < desc->num_hidden; ++l) {
      SruLayerP L;
      L.in = l == 0 ? desc->in_dim : ncols;
      L.k = L.in == ncols ? 3 : 4;
      const long nw = (long)L.in * ncols * L.k;
      L.W = p; L.dW = g; p += nw; if (g) g += nw;
      L.b = p; L.db = g; p += 2 * ncols; if (g) g += 2 * ncols;
      n.sru.push_back(L);
    }
    n.last = take(desc->out_dim, ncols);
    e->s_u.resize(desc->num_hidden); e->s_h.resize(desc->num_hidden);
    e->s_c.resize(desc->num_hidden); e->s_xdrop.resize(desc->num_hidden); e->s_xmask.resize(desc->num_hidden); e->s_wt.resize(desc->num_hidden);
  } else {
    if (desc->arch == GT_ARCH_IN2OUT) n.gate = take(desc->static_dim, desc->static_dim);
    int in = desc->in_dim;
    for (int l = 0; l < desc->num_hidden; ++l) { n.hidden.push_back(take(desc->hidden_dim, in)); in = desc->hidden_dim; }
    n.last = take(desc->out_dim, in);
  }
  n.bound = true;
  n.grads_dirty = false;
  auto& acts = role == GT_ROLE_G ? e->g_act : e->d_act;
  acts.resize(desc->num_hidden);
  if (role == GT_ROLE_G && is_i2o(desc->arch)) {
    // single dynamic stream of width out_dim (models.py:66,115)
    const int sd = desc->out_dim / e->cfg.num_windows;
    std::vector<int> sc(sd), ss(sd, sd);
    for (int c = 0; c < sd; ++c) sc[c] = c;
    e->i2o_ds = sd;
    CHK(upload_ints(sc, &e->d_scol_i2o));
    CHK(upload_ints(ss, &e->d_sstride_i2o));
    if (sd != desc->static_dim) return fail(GT_ERR_DIM, "In2OutHighwayNet: out_dim/num_windows (%d) != static_dim (%d)", sd, desc->static_dim);
  }
  if (role == GT_ROLE_G) e->g_pass_valid = false;
  return GT_OK;
}

extern "C" int gt_bind_optimizer(gt_engine* e, int role, const gt_optim_desc* od) {
  if (!e || !od || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  Net& n = e->net[role];
  if (!n.bound) return fail(GT_ERR_STATE, "bind the model before its optimizer");
  if (!n.d.grads) return fail(GT_ERR_INVALID, "model was bound without a grads buffer");
  if (od->kind != GT_OPT_ADAGRAD && od->kind != GT_OPT_ADAM) return fail(GT_ERR_INVALID, "unknown optimizer kind");
  if (!od->state0 || (od->kind == GT_OPT_ADAM && !od->state1)) return fail(GT_ERR_INVALID, "optimizer state buffer is null");
  n.od = *od;
  n.step = od->step;
  n.has_opt = true;
  return GT_OK;
}
extern "C" int gt_set_training(gt_engine* e, int role, int training) {
  if (!e || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].training = training != 0;
  return GT_OK;
}
extern "C" int gt_set_lr(gt_engine* e, int role, float lr) {
  if (!e || role < 0 || role > 1 || !e->net[role].has_opt) return fail(GT_ERR_INVALID, "no optimizer bound");
  e->net[role].od.lr = lr;
  return GT_OK;
}
extern "C" int gt_get_optimizer_step(gt_engine* e, int role, int64_t* step) {
  if (!e || role < 0 || role > 1 || !step) return fail(GT_ERR_INVALID, "bad argument");
  *step = e->net[role].step;
  return GT_OK;
}
extern "C" int gt_set_seed(gt_engine* e, uint64_t seed) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->seed = seed;
  e->step_counter = 0;
  return GT_OK;
}
extern "C" int gt_set_dropout_mask(gt_engine* e, int role, int pass, int layer, const float* mask) {
  if (!e || role < 0 || role > 1 || pass < 0 || pass > 2 || layer < 0 || layer > 15) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].inj[pass][layer] = mask;
  return GT_OK;
}
extern "C" int gt_set_option(gt_engine* e, int option, int value) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  switch (option) {
    case GT_OPT_LSTM_PERSISTENT: e->lstm_persistent = value != 0; return GT_OK;
    case GT_OPT_LSTM_FWD_UNITS: e->lstm_fwd_upc = value; return GT_OK;
    case GT_OPT_LSTM_XCD_LOCAL: e->lstm_xcd_local = value != 0; return GT_OK;
    case GT_OPT_SPLIT_FIRST_LAYER:
      if (e->opt_split_first != (value != 0)) { e->fake_cat_valid = false; e->adv2_fake_ok = false; }
      e->opt_split_first = value != 0;
      return GT_OK;
    case GT_OPT_FUSED_OPTIMIZER: e->opt_fused_optimizer = value != 0; return GT_OK;
    case GT_OPT_SIDE_OVERLAP: e->opt_side_overlap = value != 0; return GT_OK;
    case GT_OPT_LSTM_SIDE: return value ? fail(GT_ERR_INVALID, "GT_OPT_LSTM_SIDE: removed in round 6 (measured: no gain); the value must be 0") : GT_OK;
    case GT_OPT_COMM_D_ONE_MSG: e->opt_comm_d_one_msg = value != 0; return GT_OK;
    case GT_OPT_COMM_EARLY_G: e->opt_comm_early_g = value != 0; return GT_OK;
    case GT_OPT_COMM_GROUP: e->opt_comm_group = value != 0; return GT_OK;
    case GT_OPT_COMM_FORCE: e->opt_comm_force = value != 0; return GT_OK;
    case GT_OPT_COMM_CLOSE_INLINE: e->opt_comm_close_inline = value != 0; return GT_OK;
    case GT_OPT_COMM_IPC: e->opt_comm_ipc = value != 0; return GT_OK;
    case GT_OPT_FUSED_DSTACK: e->opt_fused_dstack = value < 0 ? 0 : value > 2 ? 2 : value; return GT_OK;
    case GT_OPT_COMM_TV_IN_SUMS: e->opt_comm_tv_in_sums = value != 0; return GT_OK;
    case GT_OPT_POLL_RESULTS: e->opt_poll_results = value != 0; return GT_OK;
    case GT_OPT_LAUNCH_RIDERS: e->opt_launch_riders = value != 0; return GT_OK;
    case GT_OPT_MATMUL_BF16:
      // the storage precision belongs to a PASS: buffers of a stashed forward pass (bf16 images vs float32 stashes) are not
      // interchangeable, so a change drops whatever is stashed -- the next update_* then asks for a fresh apply_generator
      // instead of back-propagating through buffers the forward never filled
      if (e->matmul_bf16 != (value != 0)) {
        e->g_pass_valid = false; e->fake_cat_valid = false; e->dcat_b_ok = false; e->adv2_fake_ok = false; e->leak_pending = false; e->cxd_src = nullptr;
        e->d_begin_done = false; e->g_begin_done = false;
      }
      e->matmul_bf16 = value != 0;
      return GT_OK;
  }
  return fail(GT_ERR_INVALID, "unknown option %d", option);
}
GtTuning& gt_tuning() {
  static GtTuning t = [] {
    GtTuning v;
    auto geti = [](const char* n, int d) { const char* s = getenv(n); return s && s[0] ? atoi(s) : d; };
    v.gemm_pair = geti("GT_GEMM_PAIR", v.gemm_pair); v.pair_order = geti("GT_PAIR_ORDER", v.pair_order);
    { const char* s = getenv("GT_GEMM_TILES"); v.gemm_tiles_big = s && !strcmp(s, "big") ? 1 : 0; }
    v.gemm_unaligned = geti("GT_GEMM_UNALIGNED", v.gemm_unaligned); v.tn_wgs = geti("GT_TN_WGS", v.tn_wgs); v.tn_split_wgs = geti("GT_TN_SPLIT_WGS", v.tn_split_wgs); v.split_fused = geti("GT_SPLIT_FUSED", v.split_fused);
    v.b16_tiles = geti("GT_B16_TILES", v.b16_tiles); v.b16_wg_tile = geti("GT_B16_WG_TILE", v.b16_wg_tile); v.b16_dma = geti("GT_B16_DMA", v.b16_dma);
    v.mlpg_fpl = geti("GT_MLPG_FPL", v.mlpg_fpl); v.mlpg_tt = geti("GT_MLPG_TT", v.mlpg_tt); v.sru_lw = geti("GT_SRU_LW", v.sru_lw); v.sru_cs_waves = geti("GT_SRU_CS_WAVES", v.sru_cs_waves); v.leak_rider = geti("GT_LEAK_RIDER", v.leak_rider);
    v.head_vec = geti("GT_HEAD_VEC", v.head_vec);
    v.mlpg_small16 = geti("GT_MLPG_SMALL16", v.mlpg_small16);
    return v;
  }();
  return t;
}
extern "C" int gt_set_tuning(const char* name, int value) {
  if (!name) return fail(GT_ERR_INVALID, "null name");
  GtTuning& t = gt_tuning();
  struct { const char* n; int* p; } tab[] = {
      {"gemm_pair", &t.gemm_pair}, {"pair_order", &t.pair_order}, {"gemm_tiles_big", &t.gemm_tiles_big}, {"gemm_unaligned", &t.gemm_unaligned},
      {"tn_wgs", &t.tn_wgs}, {"tn_split_wgs", &t.tn_split_wgs}, {"split_fused", &t.split_fused}, {"b16_tiles", &t.b16_tiles},
      {"b16_wg_tile", &t.b16_wg_tile}, {"b16_dma", &t.b16_dma}, {"mlpg_fpl", &t.mlpg_fpl}, {"mlpg_tt", &t.mlpg_tt}, {"sru_lw", &t.sru_lw}, {"sru_cs_waves", &t.sru_cs_waves}, {"leak_rider", &t.leak_rider}, {"head_vec", &t.head_vec}, {"mlpg_small16", &t.mlpg_small16}};
  for (auto& e : tab) if (!strcmp(e.n, name)) { *e.p = value; return GT_OK; }
  return fail(GT_ERR_INVALID, "unknown tuning knob '%s'", name);
}
extern "C" int gt_set_x_pitch(gt_engine* e, int ld_generator_input, int ld_condition) {
  if (!e || ld_generator_input < 0 || ld_condition < 0) return fail(GT_ERR_INVALID, "bad argument");
  e->ld_gx = ld_generator_input; e->ld_cx = ld_condition;
  return GT_OK;
}
extern "C" int gt_set_loss_normalizer(gt_engine* e, float tv) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->tv_override = tv;
  e->tv_dev = nullptr;
  return GT_OK;
}
extern "C" int gt_set_loss_normalizer_device(gt_engine* e, const double* tv_dev) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  e->tv_dev = tv_dev;
  e->tv_mask = nullptr; e->tv_inflight = false;          // re-read on the next step function
  return GT_OK;
}
extern "C" int gt_set_lengths(gt_engine* e, const int64_t* lengths_host, int B, void* stream) {
  if (!e || !lengths_host || B < 1) return fail(GT_ERR_INVALID, "bad argument");
  hipStream_t s = (hipStream_t)stream;
  e->h_lengths.resize(B);
  for (int b = 0; b < B; ++b) {
    if (lengths_host[b] < 0 || lengths_host[b] > 0x3fffffff) return fail(GT_ERR_INVALID, "length out of range");
    e->h_lengths[b] = (int)lengths_host[b];
  }
  // the device copy holds the lengths TWICE ([len | len]): a recurrent discriminator runs the natural and the generated sequences of a
  // D step as one batch of 2B sequences
  if (2 * B > e->len_cap) {          // (re)allocate the pinned slots; device slots grow on demand
    HIPCHK(hipStreamSynchronize(s));
    const int cap = std::max(64, 2 * B + B);
    for (int i = 0; i < gt_engine::LEN_RING; ++i) {
      if (e->len_ev[i]) HIPCHK(hipEventSynchronize(e->len_ev[i]));
      if (e->len_host[i]) { HIPCHK(hipHostFree(e->len_host[i])); e->len_host[i] = nullptr; }
      HIPCHK(hipHostMalloc((void**)&e->len_host[i], (size_t)cap * sizeof(int)));
      if (!e->len_ev[i]) HIPCHK(hipEventCreateWithFlags(&e->len_ev[i], hipEventDisableTiming));
    }
    e->len_cap = cap;
  }
  const int slot = (e->len_slot + 1) % gt_engine::LEN_RING;
  HIPCHK(hipEventSynchronize(e->len_ev[slot]));          // the copy that last used this pinned slot (4 batches ago)
  memcpy(e->len_host[slot], e->h_lengths.data(), (size_t)B * sizeof(int));
  memcpy(e->len_host[slot] + B, e->h_lengths.data(), (size_t)B * sizeof(int));
  CHK(e->len_dev[slot].ensure((size_t)e->len_cap * sizeof(int)));
  HIPCHK(hipMemcpyAsync(e->len_dev[slot].p, e->len_host[slot], (size_t)2 * B * sizeof(int), hipMemcpyHostToDevice, s));
  HIPCHK(hipEventRecord(e->len_ev[slot], s));
  e->len_slot = slot;
  return GT_OK;
}

extern "C" int gt_zero_grad(gt_engine* e, int role) {
  if (!e || role < 0 || role > 1) return fail(GT_ERR_INVALID, "bad argument");
  e->net[role].grads_dirty = false;   // lazily: the next backward overwrites
  { SlabDefer& sd = e->sdefer[role]; sd.active = false; sd.jobs.n = 0; sd.blocks = 0; sd.used = 0; }   // nothing recorded survives a zero_grad
  e->tv_mask = nullptr; e->tv_inflight = false;
  if (role == GT_ROLE_G) { e->leak_pending = false; e->leak_unnorm = false; }
  return GT_OK;
}
extern "C" int gt_scalar_buffer(gt_engine* e, double** dev_ptr, int* n) {
  if (!e || !dev_ptr || !n) return fail(GT_ERR_INVALID, "bad argument");
  *dev_ptr = &e->sc()->s_real;
  *n = 7;   // D step: s_real, s_fake, n_real_ok, n_fake_ok | G step: s_adv, s_mge, s_mse
  return GT_OK;
}

// Philox dropout site (role, pass, layer) of engine step `step`: the keep decision of element (row, col) is
// philox_keep(key0, key1, thresh, row, col) (gemm_f32.hip.h) -- a function of the site and the element only, not of
// the tiling of whichever kernel applies it.  The keep probability is quantised to 16 bits (thresh = round(p * 2^16)):
// exact for p = k / 65536 (0.5, 0.25, ...), otherwise |P(keep) - (1-p)| <= 2^-17 while the survivors are scaled by
// the nominal 1/(1-p) like nn.Dropout does.
// Data parallel ("Dropout/noise RNG keyed by global sequence index so DP=k reproduces DP=1"): with world > 1 the
// site's row groups are mapped to the groups the same frames have in the one-process minibatch (DropoutSpec::dp_*,
// philox_group in gemm_f32.hip.h), so a world-k run draws exactly the masks a world-1 run draws for the whole minibatch
// (the reference draws ONE mask over the whole minibatch: models.py:139, train.py:538-585).  `half_rows`: rows of one half of
// a [real | generated] two-half pass, 0 for a single block.  The map needs whole 16-row groups per sequence (T % 16 == 0);
// for other T the rank is folded into the key instead (independent masks per rank: valid dropout, not world-1's bits).
DropoutSpec philox_site_spec(gt_engine* e, int role, int pass, int layer, uint64_t step, float p, long half_rows) {
  DropoutSpec d = no_drop();
  d.p = p;
  d.scale = 1.f / (1.f - p);
  d.mode = DROP_PHILOX;
  const double th = (double)p * 65536.0 + 0.5;
  d.thresh = th >= 65535.0 ? 65535u : (uint32_t)th;
  const uint64_t site = step * 64ULL + (uint64_t)(role * 32 + pass * 16 + layer);
  uint32_t rk = 0;
  if (e->dp_world > 1) {
    const int T = e->chk_T, B = e->chk_B;
    const long groups = ((long)B * T / 16) * e->dp_world * 2;
    if (T > 0 && T % 16 == 0 && groups < (1L << 21) && (half_rows == 0 || half_rows == (long)B * T)) {
      d.dp_t16 = (uint32_t)(T / 16);
      d.dp_inv_t16 = 1.f / (float)d.dp_t16;
      d.dp_nl16 = half_rows ? (uint32_t)(half_rows / 16) : 0xffffffffu;
      d.dp_half = half_rows ? (uint32_t)((long)B * e->dp_world * (T / 16) - half_rows / 16) : 0u;
      d.dp_add = (uint32_t)e->dp_rank * d.dp_t16;
      d.dp_mul = (uint32_t)(e->dp_world - 1) * d.dp_t16;
    } else {
      rk = (uint32_t)e->dp_rank;
    }
  }
  d.key0 = (uint32_t)(e->seed ^ (site * 0x9E3779B97F4A7C15ULL)) ^ (rk * 0x85EBCA6Bu);
  d.key1 = (uint32_t)((e->seed >> 32) ^ (site >> 7) ^ 0xA5A5A5A5u) + (uint32_t)site + rk * 0xC2B2AE35u;
  return d;
}

// dropout spec of (role, pass, layer).  rows_off: first row of `pass` inside the stacked mask buffer.
DropoutSpec drop_spec(gt_engine* e, int role, int pass, int layer, const float* stacked_mask, int ld, long half_rows) {
  Net& n = e->net[role];
  DropoutSpec d = no_drop();
  if (!n.training || n.d.dropout <= 0.f) return d;
  if (stacked_mask) {
    d.p = n.d.dropout;
    d.scale = 1.f / (1.f - n.d.dropout);
    d.mode = DROP_BUFFER; d.mask = stacked_mask; d.ld_mask = ld;
    return d;
  }
  return philox_site_spec(e, role, pass, layer, e->step_counter, n.d.dropout, half_rows);
}

// Parity hook: the 0/1 keep mask the engine's Philox stream assigns to dropout site (role, pass, layer) of the step
// that starts `steps_ahead` apply_generator calls from now (1 = the next one), for a (rows, cols) activation.
extern "C" int gt_op_philox_mask(gt_engine* e, int role, int pass, int layer, int64_t steps_ahead, float p, int64_t rows, int cols,
                                 float* mask, void* stream) {
  if (!e || !mask || role < 0 || role > 1 || pass < 0 || pass > 2 || layer < 0 || layer > 15 || rows < 1 || cols < 1 ||
      steps_ahead < 0 || !(p > 0.f && p < 1.f) || rows > 0x7fffffffL)
    return fail(GT_ERR_INVALID, "bad argument");
  // (data parallel: the row-group map of the last entry point's (B, T); a 2*B*T-row request is the D step's two-half pass)
  const DropoutSpec d = philox_site_spec(e, role, pass, layer, e->step_counter + (uint64_t)steps_ahead, p,
                                         rows == 2L * e->chk_B * e->chk_T ? (long)e->chk_B * e->chk_T : 0L);
  hipLaunchKernelGGL(philox_mask_kernel, dim3(cdiv(rows * cols, 256)), dim3(256), 0, (hipStream_t)stream, d, rows, cols, mask);
  LAUNCH_CHECK();
  return GT_OK;
}

// A persistent launch that gave up (a peer workgroup never published: LSTM_FAULT_*) leaves garbage behind.  The fault
// word is mirrored to the host behind every such launch without waiting; every later entry point looks at the mirror
// first, gt_check_faults() synchronises and looks.
int fault_seen(gt_engine* e) {
  const unsigned int f = e->h_fault ? (e->h_fault[0] | e->h_fault[1]) : 0u;
  if (f & 0xffu)
    return fail(GT_ERR_HIP, "persistent LSTM kernel fault %u: a workgroup timed out waiting for its peers (results of that "
                "step are invalid and its optimizer updates were skipped; gt_clear_faults() re-arms the engine, "
                "GT_OPT_LSTM_PERSISTENT=0 / GT_LSTM_STEPS=1 selects the per-step kernels)", f);
  if (f) return fail(GT_ERR_HIP, "device fault word 0x%x: results of that step are invalid", f);
  return GT_OK;
}
// After a fault: parameters, gradients and optimizer state were left untouched by every optimizer launch that saw the
// raised word (optim_step_kernel returns before its first write and counts the skipped step in pinned memory).  This
// call waits for the stream, takes the skipped steps back out of the host-side step counters, and clears the word, so
// that the engine is usable again (typically after gt_set_option(GT_OPT_LSTM_PERSISTENT, 0)).
extern "C" int gt_clear_faults(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemset(e->d_fault, 0, 64));
  if (e->opt_bar.p) { HIPCHK(hipMemset(e->opt_bar.p, 0, 64)); e->opt_bar_count = 0; }      // the barrier counter restarts with the re-armed engine
  for (int r = 0; r < 2; ++r) { e->net[r].step -= (long)e->h_fault[2 + r]; if (e->net[r].step < 0) e->net[r].step = 0; }
  for (int i = 0; i < 4; ++i) e->h_fault[i] = 0;
  e->g_pass_valid = false; e->leak_pending = false; e->fake_cat_valid = false; e->adv2_fake_ok = false; e->cxd_src = nullptr;
  e->d_begin_done = e->g_begin_done = false; e->early_done = false;
  return GT_OK;
}
extern "C" int gt_check_faults(gt_engine* e, void* stream) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  HIPCHK(hipMemcpyAsync(e->h_fault, e->d_fault, sizeof(unsigned int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIPCHK(hipStreamSynchronize((hipStream_t)stream));
  return fault_seen(e);
}
int check_common(gt_engine* e, int B, int T) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  tl_gemm_prec = e->matmul_bf16 ? PREC_BF16 : PREC_F32;       // every step / forward entry point passes through here
  if (B < 1 || T < 1) return fail(GT_ERR_INVALID, "B and T must be positive");
  if ((long)B * T > 0x3fffffffL) return fail(GT_ERR_INVALID, "B*T too large");
  e->chk_B = B; e->chk_T = T;
  return GT_OK;
}
