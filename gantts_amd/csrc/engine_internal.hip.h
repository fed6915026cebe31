// libgantts_hip.so -- internal header shared by the engine's translation units (eng_*.hip).
//
// The engine keeps the reference's step semantics (train.py:245-320) while restructuring the
// work for the hardware:
//   * D(real) and D(fake) of the D step run as ONE 2N-row pass (same weights), so the frame x
//     weight GEMMs see 32768 rows at the headline config;
//   * G is back-propagated ONCE per step: dloss_d/dy_hat_static (old D weights, the reference's
//     un-detached "leak", train.py:265,274) is stashed and summed with dloss_g/dy_hat_static at
//     y_hat_static before MLPG^T and the G backward (backward is linear in the upstream gradient);
//   * the G-step D pass computes no weight gradients (the reference's are discarded by the next
//     zero_grad, train.py:538-539);
//   * every reduction is two-stage with a fixed order => run-to-run bit-reproducible.
//
// Translation units: eng_core (life cycle, binding, options, dropout sites, faults), eng_gemm_f32 / eng_gemm_b16 (dispatch
// of the two product families), eng_step (the G+D step and the MLP stacks), eng_lstm / eng_sru (recurrent generators),
// eng_comm (data-parallel communicator), eng_ops (stand-alone operators).  Kernels live in the *.hip.h headers; the
// non-template ones have internal linkage (GT_KERNEL) so that several units may include them.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#pragma GCC visibility push(default)
#include "../../include/gantts_hip.h"
#pragma GCC visibility pop
#include "frame_kernels.hip.h"
#include "gemm_f32.hip.h"
#include "gemm_bf16s.hip.h"

using namespace gt;

// ------------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------------
int fail(int code, const char* fmt, ...);
#define HIPCHK(expr)                                                                         \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(GT_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
  } while (0)
#define CHK(expr)            \
  do {                       \
    int _r = (expr);         \
    if (_r != GT_OK) return _r; \
  } while (0)
#define LAUNCH_CHECK() HIPCHK(hipGetLastError())

static inline bool env_flag(const char* name, bool dflt) { const char* v = getenv(name); return v && v[0] ? v[0] != '0' : dflt; }
static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }
static inline int pad8(long n) { return (int)((n + 7) & ~7L); }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE function attribute: remembered per (kernel, device)
int ensure_dyn_lds(const void* kernel, size_t bytes);
int gemm_cu_count();

// ------------------------------------------------------------------------------------------
// Dispatch knobs of the product families and frame kernels: ONE process-wide table, read once from the environment when the
// library is first used and settable afterwards by name (gt_set_tuning; tools/ harnesses and A/B runs).  Every default is the
// measured best; none selects different arithmetic.  What changes a STEP's schedule or a network's path is
// per-engine state instead (gt_engine::opt_*, gt_set_option): two engines of one process may differ.
// ------------------------------------------------------------------------------------------
struct GtTuning {
  int gemm_pair = 1;          // GT_GEMM_PAIR      backward-data + weight gradient of a layer in one launch
  int pair_order = 1;         // GT_PAIR_ORDER     pair launch: weight-gradient workgroups first
  int gemm_tiles_big = 0;     // GT_GEMM_TILES=big residency model (128-wide tiles) for every float32 launch
  int gemm_unaligned = 1;     // GT_GEMM_UNALIGNED 16-byte loads of k-contiguous operands at 4-byte aligned addresses
  int tn_wgs = 512;           // GT_TN_WGS         workgroups of a weight-gradient launch
  int split_fused = 1;        // GT_SPLIT_FUSED    split first layer's forward as one two-segment launch (0: x product + K = 58 pass)
  int tn_split_wgs = 1024;    // GT_TN_SPLIT_WGS   ... of the split first layer's two-block launch
  int b16_tiles = 0;          // GT_B16_TILES      bf16-storage products: force 64 / 128 / 256 tiles (0 = by shape)
  int b16_wg_tile = 0;        // GT_B16_WG_TILE    ... their weight gradients
  int b16_dma = 1;            // GT_B16_DMA        LDS-DMA operand stages
  int mlpg_fpl = 2;           // GT_MLPG_FPL       frames per lane of the MLPG compute phase
  int head_vec = 0;           // GT_HEAD_VEC       discriminator head: 16-byte accesses (lane <-> four consecutive hidden units); measured -3 us
                              //                   per step, NOT the default: it sums the row's dot product in another order, and one oracle-only
                              //                   at-size case (a cold-Adagrad update, lr * g / |g|) then lands 1.2x outside its 1e-4
  int mlpg_small16 = 0;       // GT_MLPG_SMALL16   16-frame MLPG tiles when 32-frame tiles would fill at most half the CUs (mlpg_tt == 0)
  int mlpg_tt = 0;            // GT_MLPG_TT        output frames per MLPG workgroup (0 = by shape, 16, 32)
  int leak_rider = 1;         // GT_LEAK_RIDER     D step: the kept dloss_d / dy_hat_static product rides in the split first layer's weight-gradient launch
  int sru_cs_waves = 0;       // GT_SRU_CS_WAVES   waves per 64 columns of the cooperative SRU scans: 0 = by shape (8 where B x ncols / 64 <= CUs, else 4), 4, 8
  int sru_lw = 2;             // GT_SRU_LW         2: cooperative block scans (sru_cs_kernels.hip.h); 1: loader-wave scans; 0: one-wave kernels (1 == 0 bit for bit)
};
GtTuning& gt_tuning();

// optional per-launch timing of the product families (HIP events on the launch stream); bench.py's live roofline figure
struct GemmProfiler {
  bool on = false;
  int only_kind = -1;       // >= 0: only launches of this kind are instrumented (5 = the pair launches): gt_profile_enable(2 + kind)
  bool wants(int kind) const { return on && (only_kind < 0 || only_kind == kind); }
  struct Rec { int kind, bn, am; double flops, bytes; hipEvent_t e0, e1; };   // am: GemmAmode of a float32 kernel (-1: run-time flavour)
  double last_bytes[GT_PROFILE_SLOTS] = {};   // algorithmic bytes per slot of the last gt_profile_read
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;
  hipEvent_t get() {
    if (!pool.empty()) { hipEvent_t e = pool.back(); pool.pop_back(); return e; }
    hipEvent_t e; (void)hipEventCreate(&e); return e;
  }
};
extern GemmProfiler g_prof;

struct Scratch {  // growable device buffer
  void* p = nullptr;
  size_t bytes = 0;
  int ensure(size_t need) {
    if (need <= bytes) return GT_OK;
    if (p) { HIPCHK(hipDeviceSynchronize()); HIPCHK(hipFree(p)); p = nullptr; bytes = 0; }
    const size_t cap = need + need / 8;
    HIPCHK(hipMalloc(&p, cap));
    bytes = cap;
    return GT_OK;
  }
  void release() { if (p) { (void)hipFree(p); p = nullptr; bytes = 0; } }
  template <typename T> T* as() const { return (T*)p; }
};

// Deferred weight-gradient combines (eng_gemm_f32.hip: linear_backward_weight): the partial slabs of every layer go to their
// own piece of a pool and the combine is only RECORDED; slab_defer_flush() runs all recorded combines in one launch.
struct SlabDefer {
  Scratch pool;
  size_t used = 0;
  SlabJobs jobs;
  int blocks = 0;
  bool active = false;
  SlabDefer() { jobs.n = 0; jobs.pad_ = 0; }
};

// ------------------------------------------------------------------------------------------
// engine state
// ------------------------------------------------------------------------------------------
struct Lin { float *W, *b, *dW, *db; int in, out; };
// a tensor kept as bf16 in both orientations: rm [rows][ld], tr [cols][ldt]   (gemm_bf16s.hip.h)
struct B16Img {
  Scratch rm, tr;
  int ld = 0; long ldt = 0;
  __bf16* r() { return rm.as<__bf16>(); }
  __bf16* t() { return tr.as<__bf16>(); }
  int ensure(long rows, int cols, bool want_t) {
    ld = pad8(cols); ldt = pad8(rows);
    CHK(rm.ensure((size_t)rows * ld * 2 + 64));
    if (want_t) CHK(tr.ensure((size_t)cols * ldt * 2 + 64));
    return GT_OK;
  }
  void release() { rm.release(); tr.release(); }
};
// bf16 shadows of one nn.Linear weight (out, in): w [out][ldw] feeds the forward product, wt [in][ldwt] backward-data
struct LinShadow { Scratch w, wt; int ldw = 0, ldwt = 0; };
static inline bool is_i2o(int arch) { return arch == GT_ARCH_IN2OUT || arch == GT_ARCH_IN2OUT_RNN; }
static inline bool has_lstm_body(int arch) { return arch == GT_ARCH_LSTM || arch == GT_ARCH_IN2OUT_RNN; }
struct LstmDirP { float *Wih, *Whh, *bih, *bhh, *dWih, *dWhh, *dbih, *dbhh; };
struct LstmLayerP { int in; LstmDirP d[2]; };
struct SruLayerP { int in, k; float *W, *b, *dW, *db; };

struct Net {
  bool bound = false;
  gt_model_desc d;
  std::vector<Lin> hidden;
  Lin last, gate;
  std::vector<LstmLayerP> lstm;   // GT_ARCH_LSTM; `last` is hidden2out
  std::vector<SruLayerP> sru;     // GT_ARCH_SRU;  `last` is hidden2out
  bool training = true;
  bool grads_dirty = false;      // false after zero_grad: next backward overwrites instead of accumulating
  bool has_opt = false;
  gt_optim_desc od;
  long step = 0;
  // injected dropout masks [pass][layer]
  const float* inj[3][16];
  Net() { memset(inj, 0, sizeof(inj)); }
};

// Banded images of the MLPG matrices the caller has passed so far, one per (R pointer, T): batches of a corpus come in
// many padded lengths and the caller (gantts_amd.paramgen / the reference's per-batch R) keeps one R per T, so after the
// first sight of a T there is no extraction kernel, no D2H copy and no host synchronisation on the step path.
// Contract (include/gantts_hip.h): R is immutable while cached; gt_invalidate_mlpg_cache() after rewriting / freeing it.
struct MlpgBand {
  const float* R = nullptr;
  int T = 0, kb = 0;
  uint64_t last_use = 0;
  Scratch band;
};
struct MlpgCache {
  static constexpr size_t MAX_ENTRIES = 256;
  std::vector<MlpgBand*> entries;
  MlpgBand* cur = nullptr;
  uint64_t tick = 0;
  Scratch tmp;                      // per-offset maxima of a new R (persistent: no hipFree on the step path)
  void clear() { for (auto* b : entries) { b->band.release(); delete b; } entries.clear(); cur = nullptr; }
};

struct gt_engine {
  gt_stream_config cfg;
  Net net[2];
  uint64_t seed = 0x5DEECE66DULL;
  uint64_t step_counter = 0;
  float tv_override = -1.f;
  const double* tv_dev = nullptr;   // device-resident global normaliser (data parallel: no host round trip)
  // derived stream maps (device)
  int Dout_cfg = 0, Ds = 0, Da = 0;
  std::vector<int> h_scol, h_sstride, h_adv_cols, h_adv_inv;
  int *d_scol = nullptr, *d_sstride = nullptr, *d_adv_cols = nullptr, *d_adv_inv = nullptr;
  // In2Out uses a single dynamic stream of width out_dim
  int *d_scol_i2o = nullptr, *d_sstride_i2o = nullptr; int i2o_ds = 0;
  MlpgCache mlpg;
  // workspace
  std::vector<Scratch> g_act, d_act;       // hidden activations
  Scratch dcat, dzA, dzB, leak, gadv, gs, gy, slabs, colp, partial, headp, headw, dmask, tx, gx, dgx, dtz, dout;
  Scratch scal;                            // StepScalars + StepResults
  StepResults* h_res = nullptr;            // pinned
  StepResults* h_res_dev = nullptr;        // the same page as the kernels see it: the fused calls' finalisation writes the
                                           // scalars straight into host memory (no device -> host copy launch behind it)
  // per-step state
  int B = 0, T = 0; long N = 0;
  const float* last_x = nullptr; const float* last_yhat = nullptr; const float* last_yhs = nullptr;
  bool g_pass_valid = false, leak_pending = false, fake_cat_valid = false;
  const float* fake_cat_x = nullptr; const float* fake_cat_yhs = nullptr;
  bool d_begin_done = false, g_begin_done = false, g_has_adv = false, g_used_mlpg = false;
  const float* tv_mask = nullptr; long tv_n = 0; float tv_ovr = 0.f;   // sum(mask) already on the device for this step
  // early results (single-GPU fused entry points): the step scalars are final right after the loss
  // kernels, long before backward + optimizer finish; they are copied out then, and the call returns
  // as soon as THAT copy has landed, leaving the rest of the step queued on the stream.
  bool early = false, early_done = false;
  hipEvent_t ev_res = nullptr;
  // GT_OPT_POLL_RESULTS: instead of an event behind the finalisation (its record is a system-scope release in the MIDDLE of the step:
  // ~6 us of idle stream, twice per step), the finalising thread writes a ticket number behind the results in the host page and
  // the call polls for it.  ticket_* : last number handed out / the one the pending early results carry (0 = the event is used)
  unsigned ticket_next = 0, ticket_wait = 0;
  unsigned* ticket_dev() { return h_res_dev ? (unsigned*)((char*)h_res_dev + 64) : nullptr; }
  volatile unsigned* ticket_host() { return (volatile unsigned*)((char*)h_res + 64); }
  // deferred results of the split-phase calls (out == NULL): own pinned copy + event per role, fetched by gt_*_result
  StepResults* h_def[2] = {nullptr, nullptr}; hipEvent_t ev_def[2] = {nullptr, nullptr}; bool def_pending[2] = {false, false};
  std::vector<DropoutSpec> g_specs, d_specs;   // dropout sites of the stashed passes
  // recurrent generator workspace (per layer) and the lengths of the current batch
  std::vector<Scratch> l_xproj, l_gates, l_cst, l_out, l_outd;   // l_outd: inter-layer-dropped outputs
  Scratch i2o_gout;                                              // In2OutRNNHighwayNet: hidden2out output G(x)
  Scratch l_state, l_dout, l_hshift;
  // a recurrent DISCRIMINATOR (train.py:773-774 allows any model class): its own stashes (the generator's must survive the
  // discriminator passes of a step), the gradient w.r.t. its [x | adv] input
  std::vector<Scratch> dl_xproj, dl_gates, dl_cst, dl_out, dl_outd;
  Scratch dl_dout, dl_hshift, d_dx0;
  Scratch l_xch;                                   // persistent recurrence: exchange granules
  struct GtComm* comm = nullptr;                   // gt_comm_init: RCCL communicator + comm stream (data parallel)
  struct GtIpc* ipc = nullptr;                     // gt_comm_ipc_*: interprocess arenas of the two-shot all-reduce (eng_ipc.hip)
  bool opt_comm_ipc = env_flag("GT_COMM_IPC", true);      // GT_OPT_COMM_IPC: messages that fit the arena's slots take that path once the arenas are attached
  int dp_rank = 0, dp_world = 1;                   // this engine's shard of the minibatch (gt_comm_init / gt_set_shard): sequence b here
                                                   // is sequence dp_rank + dp_world * b of the whole minibatch (round-robin dealing)
  int chk_B = 0, chk_T = 0;                        // (B, T) of the entry point that is running (check_common)
  std::vector<std::pair<long, long>> comm_done[2]; // per role: gradient ranges (offset, count) already handed to RCCL this step
  std::vector<std::pair<long, long>> comm_pending[2];   // final on the step stream, not handed over yet (merged into few messages)
  Scratch comm_tv;                                 // device double: global valid-frame count
  bool tv_inflight = false;                        // its all-reduce has been issued for the current mask
  // GT_OPT_MATMUL_BF16 on MLP stacks: bf16 images (both orientations) of everything that only feeds products
  std::vector<B16Img> g_actb, d_actb;              // hidden activations
  B16Img xin_b, dcat_b, gy_b, dz_b[2], fwd_b;      // G's input, D's [x | adv] image (2N rows), dloss/dy_hat, dZ ping-pong, gt_model_forward's input
  std::vector<B16Img> l_in_b;                      // recurrent generator: image of every layer's input (+ the top output, last entry)
  std::vector<B16Img> s_in_b;                      // SRU generator: image of every layer's (dropped) input (+ the top output, last entry)
  B16Img s_du_b;                                   // SRU: dU of the current layer, both orientations
  std::vector<LinShadow> ssh;                      // SRU: per layer W (n_in, ncols*k) as w [n_in][..] and wt [ncols*k][n_in]; last entry: hidden2out
  std::vector<B16Img> l_dg_b;                      // per layer: dG image (both orientations; per layer because the side stream reads it late)
  B16Img l_hs_b;                                   // h that entered each frame (transposed)
  hipStream_t side = nullptr;                      // recurrent generator: weight-gradient products run beside the next layer's recurrence
  hipEvent_t ev_side_go = nullptr, ev_side_done = nullptr;
  std::vector<LinShadow> lsh;                      // per LSTM layer: W_ih of all directions stacked [dirs*4H][in]; last entry: hidden2out
  bool dcat_b_ok = false;                          // dcat_b's generated half holds [x | adv(y_hat_static)] of the tensors below
  const float* dcat_b_x = nullptr; const float* dcat_b_yhs = nullptr;
  std::vector<LinShadow> wsh[2];                   // per role: bf16 shadows of the hidden layers' weights, then of the last layer's
  SlabDefer sdefer[2];                             // per role: deferred weight-gradient combines of the fused step
  Scratch d_pre;                                   // split first layer of the conditioned D: P = x . W[:, :cd]^T + b  (eng_step.hip: FirstSplit)
  Scratch adv2; int ld_adv2 = 0;                   // ... and the adversarial columns of a pass's rows, [real | generated], 16-byte pitch
  bool adv2_fake_ok = false; const float* adv2_yhs = nullptr;   // its generated half holds adv(y_hat_static) of this tensor
  struct Pitched { Scratch buf; const float* src = nullptr; int ld = 0, cols = 0; long rows = 0; uint64_t step = ~0ULL; };
  Pitched pitched[2];                              // 16-byte-pitch copies of caller tensors (slot 0: D's x, 1: G's input), once per step
  // GT_OPT_SPLIT_FIRST_LAYER / GT_OPT_FUSED_OPTIMIZER (per engine; the environment only provides the default at creation)
  // GT_OPT_FUSED_DSTACK: the float32 MLP discriminator's layers above the first one + the head (+, in the generator step, the
  // backward-data chain down to the adversarial columns) as ONE launch per pass (dstack_f32.hip.h)
  int opt_fused_dstack = getenv("GT_FUSED_DSTACK") ? atoi(getenv("GT_FUSED_DSTACK")) : 1;      // 0 off, 1 when the pass has >= one panel per CU, 2 always
  bool opt_split_first = env_flag("GT_D_SPLIT", true);
  bool opt_fused_optimizer = env_flag("GT_OPT_FUSED", false);      // measured slower: off
  bool opt_side_overlap = env_flag("GT_SIDE_OVERLAP", false);      // GT_OPT_SIDE_OVERLAP: tv / MSE kernels on the side stream (measured slower: off)
  // data-parallel schedule (GT_OPT_COMM_*): D's gradient as one message, G's loss sums early, grouped closing messages,
  // collectives issued even with one rank (bench.py --force-dp, tests)
  bool opt_comm_d_one_msg = env_flag("GT_COMM_D_ONE_MSG", true), opt_comm_early_g = env_flag("GT_COMM_EARLY_G", true),
       opt_comm_group = env_flag("GT_COMM_GROUP", false), opt_comm_force = getenv("GT_COMM_FORCE_COLLECTIVES") != nullptr;
  bool opt_comm_close_inline = env_flag("GT_COMM_CLOSE_INLINE", true);   // GT_OPT_COMM_CLOSE_INLINE: a step's closing messages on the step stream itself
  // GT_OPT_COMM_TV_IN_SUMS: the data-parallel D step does not all-reduce the valid-frame count ahead of the head; the head seeds the
  // backward pass of the UNNORMALISED loss, the local count leaves with the four loss / count sums (one message instead of two), and
  // 1 / Tv is applied where the gradient is consumed: by the optimizer kernel (which writes the normalised, clipped gradient back) and
  // by the generator step's gradient assembly for the kept dloss_d/dy_hat_static.  d_unnorm / leak_unnorm: in that state right now.
  bool opt_comm_tv_in_sums = env_flag("GT_COMM_TV_IN_SUMS", true);
  bool d_unnorm = false, leak_unnorm = false;
  bool opt_poll_results = env_flag("GT_POLL_RESULTS", true);      // round 4: no gain (1.380 / 1.384 vs 1.381 / 1.375 ms); round 5, the step 60 us shorter: 1.3125 / 1.312 vs
                                                                  // 1.319 / 1.315 / 1.3175 / 1.3206 ms, b = 4 0.400 vs 0.406 -- the two 5.8 us holes behind the finalising launches now show
  bool opt_launch_riders = env_flag("GT_LAUNCH_RIDERS", true);     // GT_OPT_LAUNCH_RIDERS: small reductions as extra workgroups of neighbouring launches
  int ld_gx = 0, ld_cx = 0;                        // gt_set_x_pitch: row pitch of the generator input / the conditioning x (0 = dense)
  // Pitched rows are read in place by the float32 MLP generator and by the split first layer of the conditioned float32 MLP
  // discriminator.  Every other network gets a DENSE copy made by the engine, once per step (dense_gx / dense_cx in eng_step.hip):
  // gt_set_x_pitch never makes a step fail (DevicePrefetcher(pitch_x=True) with a recurrent discriminator did).
  Scratch gx_dense, cx_dense;
  bool gx_dense_on = false;                        // this step's generator input is the dense copy: gx_pitch() == in_dim
  const float* cxd_src = nullptr; uint64_t cxd_step = ~0ull; long cxd_rows = 0; int cxd_ld = 0, cxd_cols = 0;     // what cx_dense holds
  Scratch opt_bar; unsigned long long opt_bar_count = 0;   // arrival counter of optim_fused_kernel's device-wide barrier (monotonic across launches)
  Scratch w0pad[2];                                // per role: first hidden layer's weight with a 16-byte row pitch (stack_forward)
  unsigned int* h_fault_dev = nullptr;             // device view of h_fault[1]: the optimizer kernel mirrors a raised fault word
  unsigned int* d_fault = nullptr;                 // device fault word of the persistent kernels (0 = ok)
  unsigned int* h_fault = nullptr;                 // pinned mirror, refreshed behind every persistent launch
  bool lstm_persistent = getenv("GT_LSTM_STEPS") == nullptr;   // GT_OPT_LSTM_PERSISTENT (the environment only provides the default at creation)
  int lstm_fwd_upc = 0;                            // 0 = automatic
  bool lstm_xcd_local = getenv("GT_LSTM_NO_XCD_LOCAL") == nullptr;   // GT_OPT_LSTM_XCD_LOCAL
  bool matmul_bf16 = false;                                          // GT_OPT_MATMUL_BF16
  // sequence lengths travel on the step stream through a small ring (pinned host slot -> device slot): the kernels of
  // the previous step, still queued when the next batch's lengths arrive, keep reading THEIR slot
  static constexpr int LEN_RING = 4;
  int* len_host[LEN_RING] = {nullptr, nullptr, nullptr, nullptr};
  Scratch len_dev[LEN_RING];
  hipEvent_t len_ev[LEN_RING] = {nullptr, nullptr, nullptr, nullptr};
  int len_cap = 0, len_slot = -1;
  int* d_lengths() { return len_slot < 0 ? nullptr : len_dev[len_slot].as<int>(); }
  std::vector<Scratch> s_wt;                                 // SRU float32 mode: transposed copies of the layers' W
  std::vector<Scratch> s_u, s_h, s_c, s_xdrop, s_xmask;     // SRU per-layer stashes (s_xmask: input-dropout multipliers [B][n_in])
  Scratch s_du, s_dx, s_dbias;
  std::vector<int> h_lengths;
  StepScalars* sc() { return scal.as<StepScalars>(); }
  StepResults* res() { return (StepResults*)((char*)scal.p + 256); }
};


// ------------------------------------------------------------------------------------------
// eng_core.hip
// ------------------------------------------------------------------------------------------
int check_common(gt_engine* e, int B, int T);
int fault_seen(gt_engine* e);
gt::DropoutSpec philox_site_spec(gt_engine* e, int role, int pass, int layer, uint64_t step, float p, long half_rows = 0);
gt::DropoutSpec drop_spec(gt_engine* e, int role, int pass, int layer, const float* stacked_mask, int ld, long half_rows = 0);

// ------------------------------------------------------------------------------------------
// eng_gemm_f32.hip -- products of the float32 family (PREC_F32) or bf16 products on float32 storage (PREC_BF16), chosen per
// engine entry point for the launches it issues on this thread
// ------------------------------------------------------------------------------------------
extern thread_local int tl_gemm_prec;
// may this operand be loaded 16 bytes per lane?  k-contiguous operands (rows of X / W in the forward product, rows of dZ in
// backward-data): any float pointer and pitch (unaligned 16-byte loads, gemm_f32.hip.h: ld4u); operands whose contiguous
// direction is m / n: 16-byte aligned base and a pitch that is a multiple of 4 floats (GT_GEMM_UNALIGNED=0: that rule for all)
bool gemm_vec_ok(const float* p, int ld, bool k_contiguous = false);
bool gemm_wide_store_ok(int kind, const gt::GemmArgs& g);
bool gemm_small_tiles_ok();
// fused discriminator stack (eng_dstack.hip, dstack_f32.hip.h)
namespace gt { struct DStackArgs; }
bool dstack_hidden_ok(int hidden_dim);
int dstack_panels(long rows);
int launch_dstack(const gt::DStackArgs& a, int hidden_dim, hipStream_t s);
gt::DropoutSpec no_drop();
int launch_gemm(int kind, const gt::GemmArgs& g, int nslab, hipStream_t s);
int linear_forward(const float* X, int ldx, const float* W, int ldw, const float* b, float* Y, int ldy,
                   long rows, int in, int out, int act, const gt::DropoutSpec& drop, hipStream_t s);
gt::GemmArgs backward_data_args(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                                long rows, int out, int ncols, int act_prev, const float* H, int ldh, const gt::DropoutSpec& drop);
int linear_backward_data(const float* dZ, int lddz, const float* W, int ldw, int col0, float* dX, int lddx,
                         long rows, int out, int ncols, int act_prev, const float* H, int ldh, const gt::DropoutSpec& drop, hipStream_t s);
int slab_defer_flush(SlabDefer& d, hipStream_t s);
int linear_backward_weight(const float* dZ, int lddz, const float* X, int ldx, long rows, int out, int in,
                           float* dW, float* db, bool accumulate, Scratch& slabs, Scratch& colp, hipStream_t s,
                           SlabDefer* defer = nullptr, const gt::GemmArgs* ride_along = nullptr, bool* rode = nullptr);

int linear_backward_weight_split(const float* dZ, int lddz, long rows, long wrap, const float* xp, int ldxp, int cd,
                                 const float* adv, int ld_adv, int Da, int out, float* dW, float* db, bool accumulate,
                                 Scratch& slabs, hipStream_t s, SlabDefer* defer, const GemmArgs* rider = nullptr, bool* rode = nullptr);

// ------------------------------------------------------------------------------------------
// eng_gemm_b16.hip
// ------------------------------------------------------------------------------------------
gt::GemmB16Args b16_args();
int launch_gemm_b16(const gt::GemmB16Args& g, int nslab, hipStream_t s, int tile = 0);
// [rows][ld_in] float32 / bf16 -> bf16 [rows][ldo] and / or its transpose [cols][ldt] (+ per-column sums -> colsum)
int cast_transpose(const float* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
                   float* colsum, bool colsum_accumulate, Scratch* colp, hipStream_t s);
int cast_transpose(const __bf16* in, int ld_in, long rows, int cols, __bf16* out, int ldo, __bf16* outT, long ldt,
                   float* colsum, bool colsum_accumulate, Scratch* colp, hipStream_t s);
int weight_grad_b16(const __bf16* dZT, long lddzt, const __bf16* XT, long ldxt, long rows, int out, int in, float* dW, float* db,
                    bool accumulate, Scratch& slabs, hipStream_t s, SlabDefer* defer = nullptr);

// ------------------------------------------------------------------------------------------
// eng_step.hip
// ------------------------------------------------------------------------------------------
int ensure_band(gt_engine* e, const float* R, int T, hipStream_t s);
int mlpg_forward(gt_engine* e, const float* y, int ldy, const int* scol, const int* sstride, int Ds,
                 float* ys, int ldys, int B, int T, hipStream_t s);
int mlpg_backward(gt_engine* e, const float* gs, int ldgs, const int* scol, const int* sstride, int Ds,
                  float* gy, int ldgy, int B, int T, float mse_w, const float* yhat, const float* ytgt, int ldt,
                  const float* mask, hipStream_t s);
int post_early_results(gt_engine* e, hipStream_t s, unsigned ticket = 0);
int cond_dim(gt_engine* e);

// ------------------------------------------------------------------------------------------
// eng_comm.hip
// ------------------------------------------------------------------------------------------
struct GtComm {
  void* comm = nullptr;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;
  hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  int next_ev = 0;
  hipEvent_t ev_done = nullptr;
  // schedule trace (gt_comm_trace, a measurement): every message and every join of the step stream bracketed by timed events
  struct TraceRec { int kind, inl; double bytes; hipEvent_t e0, e1; bool closed; };     // kind 0 message, 1 join; closed: e1 was recorded
  bool trace = false;
  long trace_dropped = 0;                          // messages / joins that found the trace full
  std::vector<TraceRec> trec;
};
bool comm_on(const gt_engine* e);
int comm_grads_ready(gt_engine* e, int role, const float* lo, long count, hipStream_t compute);
int comm_flush(gt_engine* e, int role, hipStream_t compute, bool closing = false);
int comm_tv_sent(gt_engine* e, hipStream_t s);
// eng_ipc.hip
void ipc_destroy(gt_engine* e);
bool ipc_usable(const gt_engine* e, size_t bytes);
int ipc_allreduce(gt_engine* e, void* buf, size_t count, bool dtype_double, hipStream_t s);
int comm_tv_join(gt_engine* e, hipStream_t s);
int comm_finish_step(gt_engine* e, int role, bool grads, double* sums, int n_sums, hipStream_t compute);
int comm_early_results(gt_engine* e, int role, double* sums, int n_sums, float adv_w, float mse_w, float mge_w, hipStream_t compute);
int ensure_tv_begin(gt_engine* e, const float* mask, long N, hipStream_t s);
int ensure_tv(gt_engine* e, const float* mask, long N, hipStream_t s);
// (the data-parallel schedule switches are per-engine state: gt_engine::opt_comm_*)

// ------------------------------------------------------------------------------------------
// eng_lstm.hip / eng_sru.hip
// ------------------------------------------------------------------------------------------
int lstm_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s);
int lstm_check_lengths(gt_engine* e, int B, int T);
int lstm_stack_forward(gt_engine* e, int role, const float* x, int ld_x, int nseq, int T, const int* passes, int npass, hipStream_t s,
                       const float** top, int* ld_top);
int lstm_stack_backward(gt_engine* e, int role, const float* x, int ld_x, int nseq, int T, const int* passes, int npass, bool want_w,
                        float* dx0, hipStream_t s);
int lstm_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s);
int sru_forward(gt_engine* e, const float* x, int B, int T, float* y_hat, hipStream_t s);
int sru_backward(gt_engine* e, const float* x, const float* gy, int B, int T, hipStream_t s);
