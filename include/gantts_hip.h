/* gantts_hip.h -- C ABI of the MI355X-native GAN training engine (libgantts_hip.so).
 *
 * Drop-in boundary for the G+D adversarial training step of r9y9/gantts.  Every entry point
 * cites the reference interface (file:line in the upstream tree) it stands in for; the Python
 * host package `gantts_amd` binds these with ctypes and re-exposes the reference's own Python
 * signatures.
 *
 * Conventions
 *  - plain C: opaque handle, raw device pointers (HIP, float32 unless noted), sizes; no C++/torch types.
 *  - every function returns 0 on success, non-zero on error; gt_last_error() gives the message
 *    (thread-local).  HIP errors are mapped to GT_ERR_HIP; the library never aborts.
 *  - all frame tensors are contiguous (B,T,D) row-major, feature dimension fastest
 *    (reference train.py:145-159).  A "row" is one frame; rows = B*T.
 *  - the caller owns inputs, outputs, parameters, gradients and optimizer state (so checkpoints
 *    stay torch.save-compatible, reference train.py:162-171); the engine borrows the pointers for
 *    the duration of a call and owns only its workspace (activation stash, slabs, scalars).
 *  - calls on one handle are not re-entrant; kernels are enqueued on the `stream` argument
 *    (a hipStream_t, may be NULL for the default stream); functions returning host scalars wait only
 *    for those scalars (an event after the loss kernels): the backward pass and the optimizer step of
 *    the same call may still be executing on the stream when the call returns -- stream order keeps
 *    every later call (and any reader on the same stream) correct.
 */
#ifndef GANTTS_HIP_H_
#define GANTTS_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GT_OK 0
#define GT_ERR_INVALID 1      /* bad argument / unsupported configuration */
#define GT_ERR_HIP 2          /* HIP runtime error */
#define GT_ERR_STATE 3        /* call order violated (e.g. update_* before apply_generator) */
#define GT_ERR_DIM 4          /* "You probably have specified wrong dimention params." (multistream.py:93-94) */

#define GT_ROLE_G 0
#define GT_ROLE_D 1

#define GT_ARCH_MLP 0         /* gantts/models.py:121-141 */
#define GT_ARCH_IN2OUT 1      /* gantts/models.py:21-69  (In2OutHighwayNet) */
#define GT_ARCH_LSTM 2        /* gantts/models.py:170-213 (LSTMRNN, GRURNN: nn.LSTM + hidden2out) */
#define GT_ARCH_SRU 3         /* gantts/models.py:144-167 (SRURNN: third-party SRU cells + hidden2out) */
#define GT_ARCH_IN2OUT_RNN 4  /* gantts/models.py:72-118 (In2OutRNNHighwayNet: T gate | lstm.* | hidden2out.*) */

#define GT_OPT_ADAGRAD 0      /* torch.optim.Adagrad, train.py:796-799 with hparams.py:48-52,223-227 */
#define GT_OPT_ADAM 1         /* torch.optim.Adam,    hparams.py:125-130 */

#define GT_MAX_STREAMS 8

typedef struct gt_engine gt_engine;

/* hp.* fields read on the step path (train.py:61, 233-241, 248, 254, 299, 304, 352-353). */
typedef struct {
  int32_t n_streams;
  int32_t stream_sizes[GT_MAX_STREAMS];          /* hp.stream_sizes (static+delta widths)        */
  int32_t has_dynamic_features[GT_MAX_STREAMS];  /* hp.has_dynamic_features                       */
  int32_t num_windows;                           /* len(hp.windows)                               */
  int32_t adversarial_streams[GT_MAX_STREAMS];   /* hp.adversarial_streams; [0] = -1 means None   */
  int32_t mask_nth_mgc_for_adv_loss;             /* hp.mask_nth_mgc_for_adv_loss                  */
  int32_t discriminator_linguistic_condition;    /* hp.discriminator_linguistic_condition         */
  int32_t cond_dim;                              /* width of x fed to D when conditioned (x, not cat(x,z): train.py:254-256) */
} gt_stream_config;

/* One network.  `params`/`grads` are flat float32 device buffers laid out in state_dict order:
 *   MLP:    layers.0.weight (hidden x in) | layers.0.bias | ... | last_linear.weight | last_linear.bias
 *   IN2OUT: T.weight (sd x sd) | T.bias | H.0.weight | H.0.bias | ... | last_linear.weight | last_linear.bias
 *   LSTM:   for layer k, for direction d (forward, then reverse when bidirectional):
 *             weight_ih (4H x in_k) | weight_hh (4H x H) | bias_ih (4H) | bias_hh (4H)     (gate order i,f,g,o)
 *           then hidden2out.weight (out x H*dirs) | hidden2out.bias        (torch.nn.LSTM parameter order)
 *   SRU:    for layer i: rnn_lst.i.weight (n_in x ncols*k, ncols = H*dirs, k = 3 if n_in == ncols else 4,
 *           column j owns entries j*k..j*k+k-1) | rnn_lst.i.bias (2*ncols = b_f | b_r);
 *           then hidden2out.weight (out x ncols) | hidden2out.bias
 * Linear / LSTM weights are (out, in) row-major exactly as torch stores them; SRU weights are (in, out). */
typedef struct {
  int32_t arch;
  int32_t in_dim, out_dim, num_hidden, hidden_dim;
  int32_t static_dim;        /* IN2OUT only */
  float dropout;
  int32_t last_sigmoid;
  int32_t bidirectional;     /* LSTM, SRU */
  int32_t use_relu;          /* SRU: g = relu (1) or tanh (0), models.py:152-154 */
  float rnn_dropout;         /* SRU: variational input dropout (training) */
  int32_t reserved_;
  float* params;
  float* grads;
  int64_t n_params;
} gt_model_desc;

typedef struct {
  int32_t kind;              /* GT_OPT_* */
  float lr, weight_decay, eps;
  float lr_decay;            /* Adagrad */
  float beta1, beta2;        /* Adam */
  float max_grad_norm;       /* clip_grad_norm_ threshold, 1.0 in train.py:275,317 */
  int64_t step;              /* number of steps already taken (restored from a checkpoint) */
  float* state0;             /* Adagrad "sum" / Adam "exp_avg"     (flat, same layout as params) */
  float* state1;             /* Adam "exp_avg_sq" (NULL for Adagrad) */
} gt_optim_desc;

typedef struct {             /* return values of update_discriminator, train.py:278-279 (same order) */
  float loss_d, loss_fake_d, loss_real_d, real_correct_count, fake_correct_count;
  float grad_norm;           /* pre-clip ||grad D||_2 from the split-phase gt_update_discriminator_end; 0 from the fused call,
                              * which returns before the backward pass has finished */
} gt_d_result;

typedef struct {             /* return values of update_generator, train.py:320 (same order) */
  float loss_mse, loss_mge, loss_adv, loss_g;
  float grad_norm;           /* pre-clip ||grad G||_2 of the accumulated gradient (split-phase form only, see above) */
} gt_g_result;

/* ---- lifecycle ------------------------------------------------------------------------ */
const char* gt_last_error(void);
const char* gt_version(void);
int gt_engine_create(const gt_stream_config* cfg, gt_engine** out);
void gt_engine_destroy(gt_engine* e);

/* getattr(gantts.models, name)(**params) + .cuda()            (train.py:773-793).  GT_ROLE_G: every architecture; GT_ROLE_D: GT_ARCH_MLP
 * or GT_ARCH_LSTM (an LSTMRNN scoring frames), out_dim 1, last_sigmoid set; other combinations are rejected (GT_ERR_INVALID). */
int gt_bind_model(gt_engine* e, int role, const gt_model_desc* desc);
/* getattr(optim, hp.optimizer_*)(model.parameters(), **params) (train.py:796-799) */
int gt_bind_optimizer(gt_engine* e, int role, const gt_optim_desc* desc);
/* model.train() / model.eval()                                 (train.py:481-486) */
int gt_set_training(gt_engine* e, int role, int training);
/* exp_lr_scheduler writes param_group["lr"]                     (train.py:323-333) */
int gt_set_lr(gt_engine* e, int role, float lr);
int gt_get_optimizer_step(gt_engine* e, int role, int64_t* step);
int gt_set_seed(gt_engine* e, uint64_t seed);
/* Parity hook: use the caller's 0/1 float mask ((rows, hidden) contiguous, device) instead of the
 * Philox stream for dropout site `layer` of forward pass `pass` of `role`; NULL restores Philox.
 * passes: G: 0 = apply_generator.  D: 0 = real rows of the D step, 1 = generated rows of the
 * D step, 2 = generated rows of the G step (the order nn.Dropout is consumed in train.py:261-307).
 * SRURNN (models.py:152-154: rnn_dropout / dropout of the SRU cell are VARIATIONAL masks, one per sequence and shared
 * over time): `layer` = 2*l selects the input mask of SRU layer l, shape (B, n_in_l); 2*l + 1 its output mask, (B, H*dirs). */
int gt_set_dropout_mask(gt_engine* e, int role, int pass, int layer, const float* mask);
/* Parity hook for the production (Philox) dropout path, nn.Dropout inside gantts/models.py:132-139: writes the 0/1 keep
 * mask ((rows, cols) contiguous, device) that the engine's counter-based stream assigns to dropout site (role, pass,
 * layer) of the step that starts `steps_ahead` gt_apply_generator calls from now (1 = the next step).  The D step runs
 * its real and generated rows as ONE 2N-row pass: its site is pass 0 with rows [0,N) = real, [N,2N) = generated; pass 2
 * is the N-row D pass of the G step.  Feeding these masks to the reference (patched nn.Dropout) reproduces the step. */
int gt_op_philox_mask(gt_engine* e, int role, int pass, int layer, int64_t steps_ahead, float p, int64_t rows, int cols,
                      float* mask, void* stream);

/* Sequence lengths of the NEXT batch (host int64 array, as the `lengths` list the reference passes to
 * model(x, lengths=lengths), train.py:344; models.py:204-210 pack_padded_sequence).  Needed by the
 * recurrent networks only -- a recurrent generator, or an LSTMRNN in the discriminator slot (train.py:262, 268, 307) --; MLP ignores
 * lengths like the reference.  Unlike pack_padded_sequence the
 * batch need not be sorted.  The array is copied before the call returns; it reaches the device IN STREAM ORDER on
 * `stream` (the stream of the following step functions) through a ring of pinned slots, so the still-queued kernels of
 * the previous step keep reading their own lengths and the host never waits for the GPU here. */
int gt_set_lengths(gt_engine* e, const int64_t* lengths_host, int B, void* stream);
/* The engine keeps a banded image of every MLPG matrix R it has been given, keyed by (R pointer, T) (the reference
 * rebuilds and uploads R every batch, train.py:511-513; callers of this library keep one device R per padded length).
 * R must not be rewritten in place or freed-and-reused while cached: call this first (drops all cached bands). */
int gt_invalidate_mlpg_cache(gt_engine* e);

/* ---- hot path -------------------------------------------------------------------------- */
/* optimizer.zero_grad()                                         (train.py:538-539) */
int gt_zero_grad(gt_engine* e, int role);

/* apply_generator(model_g, x, R, lengths) -> (y_hat, y_hat_static)   (train.py:336-355)
 * x (B,T,in_dim of G); R dense (T, num_windows*T) from unit_variance_mlpg_matrix or NULL
 * (train.py:510-515); outputs y_hat (B,T,out_dim), y_hat_static (B,T,static width). */
int gt_apply_generator(gt_engine* e, const float* x, const float* R, int B, int T,
                       float* y_hat, float* y_hat_static, void* stream);

/* update_discriminator(model_d, optimizer_d, x, y_static, y_hat_static, lengths, mask, phase, eps)
 * (train.py:245-279).  x is the conditioning input (cond_dim wide, may be NULL when not
 * conditioned); mask (B,T) float; train != 0 <=> phase == "train".  When y_hat_static is the
 * buffer the last gt_apply_generator wrote (the autograd graph of the reference), the D-loss
 * gradient w.r.t. it is kept and added to G's gradient by gt_update_generator (train.py:265,274,316). */
int gt_update_discriminator(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                            const float* mask, int B, int T, int train, float eps,
                            gt_d_result* out, void* stream);

/* update_generator(model_g, model_d, optimizer_g, x, y, y_hat, y_static, y_hat_static, adv_w,
 *                  lengths, mask, phase, mse_w, mge_w, eps)      (train.py:282-320) */
int gt_update_generator(gt_engine* e, const float* x, const float* y, const float* y_hat,
                        const float* y_static, const float* y_hat_static, float adv_w,
                        const float* mask, int B, int T, int train, float mse_w, float mge_w, float eps,
                        gt_g_result* out, void* stream);

/* Split-phase forms for data parallelism (one process per GPU): *_begin runs forward+backward and
 * leaves the LOCAL gradient sums in the bound grads buffer and the local loss sums in
 * gt_scalar_buffer(); the host all-reduces both (RCCL), then *_end clips, steps and finalises.
 * tv_global > 0 overrides sum(mask) as the loss normaliser (global valid-frame count). */
int gt_set_loss_normalizer(gt_engine* e, float tv_global);
/* same, from a device double the caller keeps alive (e.g. the all-reduced sum(mask)): no host round trip; the
 * value is read by the next step functions in stream order.  NULL returns to the local / host normaliser. */
int gt_set_loss_normalizer_device(gt_engine* e, const double* tv_global_dev);
/* engine switches that do not change results beyond fp32 summation order. */
/* GT_OPT_LSTM_PERSISTENT (default 1): recurrent generators (models.py:170-213) run each layer's time loop as ONE
 * persistent launch (W_hh slices resident in registers, h exchanged between workgroups through tagged granules);
 * 0 = one launch per time step.  GT_OPT_LSTM_FWD_UNITS: hidden units per workgroup of the forward persistent kernel
 * (8 or 16; 0 = automatic). */
#define GT_OPT_LSTM_PERSISTENT 2
#define GT_OPT_LSTM_FWD_UNITS 3
/* GT_OPT_LSTM_XCD_LOCAL (default 1): a group of workgroups that verifies at kernel start that it runs on ONE XCD
 * exchanges through that XCD's L2 (workgroup-scope stores) instead of write-through stores; 0 = always write-through. */
#define GT_OPT_LSTM_XCD_LOCAL 4
/* GT_OPT_MATMUL_BF16 (default 0): mixed precision for the frame x weight products of BOTH networks (BASELINE.json
 * configs[2]): operands are rounded to bfloat16 inside the GEMM and multiplied on the bf16 matrix cores with float32
 * accumulation; parameters ("master weights"), optimizer state, activations in memory, the recurrent state and every
 * reduction stay float32.  Results differ from the float32 path at the 1e-2 relative level (tests/test_gpu_parity.py). */
#define GT_OPT_MATMUL_BF16 5
/* GT_OPT_SPLIT_FIRST_LAYER (default 1): the conditioned float32 discriminator evaluates its first layer as x . W_x^T (once per
 * D step, shared by the real and the generated rows) + adv . W_adv^T instead of one product over a concatenated [x | adv] image
 * (train.py:254-256); 0 = the concatenated image.  Same sums up to float32 association. */
#define GT_OPT_SPLIT_FIRST_LAYER 6
/* GT_OPT_FUSED_OPTIMIZER (default 0): the fused single-GPU step closes a network's update with ONE launch (weight-gradient
 * combines + squared norm + clip + optimizer step behind a device-wide barrier) instead of three.  Same arithmetic; measured
 * SLOWER on MI355X (the barrier's wait costs more than two launch edges: cfg2 1.453 vs 1.411 ms), so it is off by default. */
#define GT_OPT_FUSED_OPTIMIZER 7
/* Schedule switches (defaults are the measured best; the GT_* environment variables of the same names only provide the default at
 * engine creation): GT_OPT_SIDE_OVERLAP (0; measured slower) small memory-bound kernels on a side stream under the products; GT_OPT_LSTM_SIDE
 * (removed in round 6: LSTM weight gradients beside the next layer's recurrence measured no gain; only the value 0 is accepted); GT_OPT_COMM_D_ONE_MSG (1) / _EARLY_G (1) / _GROUP (0) data-parallel
 * message schedule; GT_OPT_COMM_FORCE (0) issue the collectives with one rank as well (bench.py --force-dp, tests);
 * GT_OPT_LAUNCH_RIDERS (1) the fused single-GPU step's small reductions (valid-frame count, the head's scalars in the generator
 * step, the generator step's finalisation) ride as an extra workgroup of a neighbouring launch instead of launches of their own;
 * GT_OPT_COMM_CLOSE_INLINE (1) a data-parallel step's closing messages are issued on the step's own stream (no event hand-off on
 * the critical path); GT_OPT_POLL_RESULTS (1; round 5: -6 us per cfg2 step, the event's two 5.8 us holes in the kernel trace) the fused single-GPU calls learn that their scalars have landed in host memory from
 * a ticket the finalising kernel writes behind them, not from an event recorded in the middle of the step;
 * GT_OPT_COMM_TV_IN_SUMS (1) the data-parallel discriminator step sends its valid-frame count WITH its loss sums (five collectives
 * per G+D step instead of six): the backward pass runs on the unnormalised loss, 1 / Tv is applied by the optimizer kernel (the
 * gradient it writes back is the normalised, clipped one) and by the generator step where it adds the kept gradient;
 * GT_OPT_COMM_IPC (1) see gt_comm_ipc_* below. */
#define GT_OPT_SIDE_OVERLAP 8
#define GT_OPT_LSTM_SIDE 9
#define GT_OPT_COMM_D_ONE_MSG 10
#define GT_OPT_COMM_EARLY_G 11
#define GT_OPT_COMM_GROUP 12
#define GT_OPT_COMM_FORCE 13
#define GT_OPT_LAUNCH_RIDERS 14
#define GT_OPT_COMM_CLOSE_INLINE 15
#define GT_OPT_POLL_RESULTS 16
#define GT_OPT_COMM_TV_IN_SUMS 17
#define GT_OPT_COMM_IPC 18
/* GT_OPT_FUSED_DSTACK (default 1): the float32 MLP discriminator (gantts/models.py:121-141; hidden_dim 128 or 256, <= 4 hidden
 * layers) runs its hidden layers above the first one, last_linear + sigmoid + the BCE terms (train.py:261-271) and -- in the
 * generator step (train.py:307-308) -- the whole backward-data chain down to the adversarial input columns as ONE launch per pass,
 * panel of 32 frames by panel, activations resident in LDS.  1 (default): when the pass has at least one panel per CU (a panel is
 * walked by ONE workgroup: small per-rank batches are faster as per-layer launches); 2: always; 0: one launch per layer + the head
 * kernel.  Same sums up to float32 association. */
#define GT_OPT_FUSED_DSTACK 19
int gt_set_option(gt_engine* e, int option, int value);
/* Process-wide dispatch knobs of the kernels (tile shapes, pair launches, loader variants ...: measurement switches of the tools/
 * harnesses and A/B runs; none selects different arithmetic).  Names: gemm_pair, pair_order, gemm_tiles_big, gemm_unaligned, tn_wgs, tn_split_wgs, split_fused,
 * b16_tiles, b16_wg_tile, b16_dma, mlpg_fpl, mlpg_tt, sru_lw, head_vec, mlpg_small16; the environment variables GT_<NAME>
 * provide the initial values. */
int gt_set_tuning(const char* name, int value);
/* Row pitch (in floats) of the input tensors `x` of the step functions, like the `lda` of a BLAS call: ld_generator_input for the
 * x of gt_apply_generator (train.py:542: cat(x, z) or x), ld_condition for the conditioning x of gt_update_discriminator /
 * gt_update_generator (train.py:254-256).  0 (default) = dense rows (pitch = width).  A pitch that is a multiple of 4 floats on a
 * 16-byte aligned tensor lets every product read the caller's rows with 16-byte loads -- a batch pipeline that stages batches
 * anyway (gantts_amd.data.DevicePrefetcher(pitch_x=True)) provides it for free; with dense 425-wide rows the engine makes the
 * 16-byte-pitch copy its weight-gradient products need itself, once per step.  Supported where it pays: float32 MLP generators
 * and the conditioned float32 discriminator; other paths reject a non-dense pitch. */
int gt_set_x_pitch(gt_engine* e, int ld_generator_input, int ld_condition);
/* The persistent recurrence kernels bound every inter-workgroup wait by a wall-clock timeout and raise a device fault
 * word instead of hanging.  The step functions report a fault they have seen (GT_ERR_HIP) at their next entry;
 * this call synchronises `stream` and reports the current state. */
int gt_check_faults(gt_engine* e, void* stream);
/* A raised fault word makes every optimizer launch behind it a no-op (parameters, gradients and optimizer state of the
 * faulted step stay as they were; optimizer.step() of train.py:276,318 is simply not taken).  This call synchronises,
 * takes the skipped steps back out of the step counters, clears the word and resets the per-step call state, so the
 * engine can be used again -- e.g. after gt_set_option(e, GT_OPT_LSTM_PERSISTENT, 0).  The next call must be
 * gt_apply_generator.  The gradient norm of a skipped update is reported as NaN.
 * DATA PARALLEL: the fault word is local to a rank -- a timeout on one rank makes only that rank skip its update while its peers
 * apply the all-reduced gradient, and the faulted rank stops posting collectives at its next step entry.  Under a communicator a
 * fault is therefore FATAL for the job: tear the ranks down and restart from the last checkpoint (gt_clear_faults re-arms one
 * engine, it does not re-synchronise replicas). */
int gt_clear_faults(gt_engine* e, void* stream);
/* ---- data-parallel communicator (one process per GPU; RCCL == NCCL on ROCm, bound at run time) ----
 * The reference has no multi-device code; the step being sharded is train.py:538-585.  Every rank holds the
 * full G / D and whole sequences of the minibatch.  With a communicator attached the FUSED step functions above are
 * data-parallel by themselves: the valid-frame count, each network's gradient (one bucket per layer, handed to RCCL as
 * soon as that layer's weight gradient is final, i.e. overlapped with the rest of the backward pass) and the additive
 * loss / count sums are summed over the ranks; clip-norm + optimizer then run on the reduced gradient, so all replicas
 * take bit-identical steps and every rank returns the GLOBAL scalars.  zero_grad must precede each update_* call.
 * gt_comm_unique_id: rank 0 fills `id_out` (GT_COMM_ID_BYTES) and ships it to the other ranks by any means;
 * gt_comm_init: collective over all ranks (ncclCommInitRank); gt_comm_destroy detaches (also done by gt_engine_destroy). */
#define GT_COMM_ID_BYTES 128
int gt_comm_unique_id(void* id_out);
int gt_comm_init(gt_engine* e, int rank, int world, const void* id);
int gt_comm_destroy(gt_engine* e);
int gt_comm_info(gt_engine* e, int* rank, int* world);
/* Schedule trace of the data-parallel step (a measurement, bench.py --comm-trace): while it is on, every message of the step and every
 * wait of the step stream for the communicator's stream is bracketed by timed events.  gt_comm_trace(e, 1) clears and starts,
 * gt_comm_trace(e, 0) stops; gt_comm_trace_read synchronises the device and fills `out` with up to max_records records of five doubles
 * {kind (0 message, 1 wait of the step stream), bytes, issued on the step stream itself (closing message) 0/1, start, end} -- times in
 * microseconds after the first record's start -- and sets *n_records (max_records == 0: the number of records held).  The reference
 * has no counterpart: train.py:538-585 is one process. */
int gt_comm_trace(gt_engine* e, int enable);
int gt_comm_trace_read(gt_engine* e, double* out, int max_records, int* n_records);
/* The small-message collective of SURVEY 8(e): a full-mesh TWO-SHOT all-reduce over hipIpc peer buffers (gantts_amd/csrc/eng_ipc.hip),
 * for every message of the step that fits an 8 MB slot -- at cfg2 all of them.  Every rank exports an arena in its own HBM
 * (gt_comm_ipc_export fills `handle_out`, GT_IPC_HANDLE_BYTES), the handles travel to all ranks by any means (like the unique id),
 * gt_comm_ipc_attach maps the peers' arenas (`handles`: world x GT_IPC_HANDLE_BYTES, in rank order; the own entry is ignored).  From
 * then on (GT_OPT_COMM_IPC, default 1) such messages are reduced by three launches of the engine's own -- publish, reduce my 1/W
 * chunk in rank order and push it to every rank, collect -- instead of an RCCL call: two link latencies per message instead of a
 * ring's 2 (W - 1), bit-identical results on all replicas; larger messages and everything before the attach use RCCL.  A
 * communicator (gt_comm_init) must be attached as well: it provides the stream, the fallback and the rank / world the arenas must
 * agree with.  Cross-device waits are bounded by a wall-clock timeout that raises the fault word (gt_check_faults).
 * gt_comm_ipc_messages: how many messages have taken this path (tests). */
#define GT_IPC_HANDLE_BYTES 64
#define GT_IPC_MAX_WORLD 8
int gt_comm_ipc_export(gt_engine* e, void* handle_out);
int gt_comm_ipc_attach(gt_engine* e, int rank, int world, const void* handles);
int gt_comm_ipc_messages(gt_engine* e, long long* n);
/* The shard this engine holds, for hosts that all-reduce themselves between the split-phase calls (no communicator):
 * sequence b of this engine is sequence rank + world * b of the whole minibatch (round-robin dealing).
 * gt_comm_init implies it.  It keys the dropout streams by GLOBAL frame / sequence: with T % 16 == 0 a world-k run draws, for
 * its rows, exactly the bits a one-process run draws for the whole minibatch (the reference draws one mask over the whole
 * minibatch: gantts/models.py:139 inside train.py:538-585), so data-parallel runs reproduce the single-GPU run; for other T
 * the MLP sites fall back to independent per-rank streams (the SRU's per-sequence masks are global for every T). */
int gt_set_shard(gt_engine* e, int rank, int world);

int gt_update_discriminator_begin(gt_engine* e, const float* x, const float* y_static, const float* y_hat_static,
                                  const float* mask, int B, int T, int train, float eps, void* stream);
int gt_update_discriminator_end(gt_engine* e, int train, gt_d_result* out, void* stream);
int gt_update_generator_begin(gt_engine* e, const float* x, const float* y, const float* y_hat,
                              const float* y_static, const float* y_hat_static, float adv_w,
                              const float* mask, int B, int T, int train, float mse_w, float mge_w, float eps,
                              void* stream);
int gt_update_generator_end(gt_engine* e, int train, float adv_w, float mse_w, float mge_w,
                            gt_g_result* out, void* stream);
/* Deferred results: the *_end calls accept out == NULL -- optimizer step, the scalars' finalisation and their
 * D2H copy are enqueued, nothing is synchronised; these two block on just that copy.  A data-parallel host
 * uses the gap to enqueue the next phase, so the GPU never waits for the host. */
int gt_update_discriminator_result(gt_engine* e, gt_d_result* out);
int gt_update_generator_result(gt_engine* e, gt_g_result* out);
/* device pointer + length (in doubles) of the additive loss/count sums of the current step */
int gt_scalar_buffer(gt_engine* e, double** dev_ptr, int* n_doubles);

/* model(x, lengths=lengths) -- plain forward of a bound network (inference / reference-D spoofing
 * rate, train.py:549-558; evaluation_tts.py:167,221).  For IN2OUT `out2` receives y_hat_static and
 * R must be given; for MLP out2/R are ignored. */
int gt_model_forward(gt_engine* e, int role, const float* x, const float* R, int B, int T,
                     float* out, float* out2, void* stream);

/* Materialise G's pending gradient (the D-loss leak) into G's grads buffer without stepping --
 * parity/introspection helper mirroring `p.grad` after update_discriminator in the reference. */
int gt_flush_generator_grads(gt_engine* e, void* stream);

/* ---- stand-alone operators (same kernels as the engine; used by the host-side mirrors of
 *      gantts.seqloss / gantts.multistream and by the parity tests) ------------------------ */
/* sequence_mask(lengths, max_len)                                (gantts/seqloss.py:9-20); lengths int64 device */
int gt_op_sequence_mask(const int64_t* lengths, int B, int T, float* mask, void* stream);
/* MaskedMSELoss()(input, target, mask=mask)                      (gantts/seqloss.py:27-43); returns host scalar;
 * grad_input (optional) receives d loss / d input */
int gt_op_masked_mse(const float* input, const float* target, const float* mask, int B, int T, int D,
                     float* loss_out, float* grad_input, void* stream);
/* Device-side collate: padding (train.py:139-159 `_pad_2d` / collate_fn) and the descending length sort of the batch
 * (train.py:494-501) without a padded host copy.  `ragged`: the batch's utterances un-padded, back to back, [total][D] (device);
 * start[b] / len[b] (device int64, B entries): first frame and frame count of the utterance that becomes OUTPUT sequence b (the
 * host passes them in sorted order); out: (B, T, D) on a row pitch of ld_out floats (>= D; pad columns and frames t >= len[b]
 * are zero, as `_pad_2d` leaves them).  Bit-exact copies. */
int gt_op_pad_sequences(const float* ragged, int D, const int64_t* start, const int64_t* len, int B, int T, float* out, int ld_out,
                        void* stream);
/* out[:, j] = in[:, idx[j]]  (select_streams / get_static_features / get_selected_static_stream:
 * gantts/multistream.py:33-79, train.py:232-242); idx int32 device array */
int gt_op_gather_cols(const float* in, int ld_in, const int32_t* idx, int n_idx, float* out, int ld_out,
                      int out_col_offset, int64_t rows, void* stream);
/* compute_distortions(y_static, y_hat_static, Y_data_mean, Y_data_std, lengths)  (train.py:399-432, with
 * split_streams / inv_scale :358-396 and nnmnkwii.metrics underneath) as one masked reduction over the valid
 * frames.  col_role[c] (host, Ds entries): 0 mel-cepstrum column of "mcd", 1 column of "bap_mcd", 2 lf0,
 * 3 vuv, 4 plain squared error ("dur_rmse"), -1 ignored; col_stat[c] (host): index of column c's
 * statistics inside stat_mean / stat_std (device; float, or double when stats_f64).  vuv_col = -1 if none.
 * lengths_host may be NULL (every sequence has T frames).  The caller forms the reference's ratios. */
typedef struct gt_distortion_sums {
  double s_mcd;      /* sum over valid frames of ||mgc - mgc_hat||_2 (inverse-scaled)            */
  double s_bap;      /* same for the bap columns                                                 */
  double s_f0;       /* sum of (exp(lf0) - exp(lf0_hat))^2 over frames voiced in both            */
  double n_voiced;   /* frames voiced in both (vuv binarised with > 0.5, train.py:375-377)        */
  double n_vuv_err;  /* frames whose binary vuv decisions differ                                 */
  double s_mse;      /* sum of squared errors of the role-4 columns                              */
  double n_frames;   /* valid frames                                                              */
} gt_distortion_sums;
int gt_compute_distortions(const float* y_static, const float* y_hat_static, int Ds, const void* stat_mean,
                           const void* stat_std, int stats_f64, const int32_t* col_stat_host,
                           const int32_t* col_role_host, int vuv_col, const int64_t* lengths_host, int B, int T,
                           gt_distortion_sums* out, void* stream);
/* multi_stream_mlpg(inputs, R, stream_sizes, has_dynamic_features) (gantts/multistream.py:82-123) and its
 * transpose (autograd backward).  Uses the engine's stream config. */
int gt_op_mlpg_forward(gt_engine* e, const float* y, const float* R, int B, int T, float* y_static, void* stream);
int gt_op_mlpg_backward(gt_engine* e, const float* g_static, const float* R, int B, int T, float* g_y, void* stream);
/* nn.Linear forward / backward with the fused activation used by the models
 * (act: 0 none, 1 LeakyReLU(0.01) [+dropout mask], 2 sigmoid).  Y (rows,out), X (rows,in), W (out,in). */
int gt_op_linear_forward(const float* X, int ldx, const float* W, const float* bias, float* Y, int ldy,
                         int64_t rows, int in_dim, int out_dim, int act, const float* keep_mask, float p,
                         void* stream);
/* dX = (dY . W) [* f'(H_prev)],  dW = dY^T . X,  db = colsum(dY); any output may be NULL.
 * workspace is managed internally (hipMallocAsync on `stream`). */
int gt_op_linear_backward(const float* dY, int lddy, const float* X, int ldx, const float* W,
                          int64_t rows, int in_dim, int out_dim,
                          float* dX, int lddx, const float* H_prev, int act_prev, const float* keep_mask_prev, float p_prev,
                          float* dW, float* db, void* stream);

/* nn.Linear forward / backward through the bf16-STORAGE products of GT_OPT_MATMUL_BF16 (gemm_bf16s.hip.h): X, W, dY (and
 * H_prev) are cast to bfloat16 images exactly as the engine keeps them, multiplied on the bf16 matrix cores with float32
 * accumulation, and the results returned as float32.  Same argument meaning as gt_op_linear_forward / _backward; any of
 * Y / dY / dX / dW / db may be NULL.  Y_image / YT_image (optional, (rows,out) / (out,rows) float32): the bf16 result
 * image and its transposed twin as the forward epilogue wrote them.  Parity hook (tests/test_gpu_parity.py). */
int gt_op_linear_bf16(const float* X, const float* W, const float* bias, int64_t rows, int in_dim, int out_dim, int act,
                      const float* keep_mask, float p, float* Y, const float* dY, const float* H_prev, int act_prev,
                      const float* keep_mask_prev, float p_prev, float* dX, float* dW, float* db,
                      float* Y_image, float* YT_image, void* stream);

/* ---- measurement (bench.py): HIP-event timing of every GEMM launch on its own stream --------
 * One slot per KERNEL (template instantiation family), so that the figures line up with a rocprofv3 kernel trace:
 *   0..5  = kind*2 + (tile N == 128), kind: 0 forward (X W^T), 1 backward-data (dZ W), 2 backward-weight (dZ^T X) -- the
 *           instantiations whose epilogue flavour is decided at run time, and every bf16-storage product;
 *   6, 7  = 64x64 forward kernels with a compiled-in epilogue: none / LeakyReLU + Philox dropout;
 *   8     = pair launches (one layer's backward-data product and weight gradient in one launch);
 *   9     = 64x64 forward, LeakyReLU + Philox dropout on (product + added matrix): the split first layer of the conditioned D;
 *   10,11 = 64x64 backward-data kernels with a compiled-in epilogue: none / LeakyReLU + Philox;
 *   12    = the two weight-gradient products of a split first layer in one launch;
 *   13    = the split first layer's forward in one launch (two K segments, two result halves);
 *   14    = the fused discriminator stack (GT_OPT_FUSED_DSTACK: layers 1..L-1 + head [+ backward-data chain] of one pass);  15 unused.
 * flops are algorithmic 2*M*N*K of the unpadded problems.  The three arrays hold GT_PROFILE_SLOTS entries.
 * gt_profile_enable(0) off, (1) every product launch, (2 + k) only launches of kind k (5 = the pair launches: what bench.py samples
 * inside its timed region -- two events per launch are not free). */
#define GT_PROFILE_SLOTS 16
int gt_profile_enable(int on);
int gt_profile_read(double* ms_per_slot, double* flops_per_slot, int64_t* launches_per_slot);
/* algorithmic HBM bytes (each operand once + the result once, fp32) of the launches the last gt_profile_read summed */
int gt_profile_bytes(double* bytes_per_slot);

#ifdef __cplusplus
}
#endif
#endif /* GANTTS_HIP_H_ */
