// libgantts_hip.so -- data-parallel communicator (RCCL by dlopen), gradient buckets, global valid-frame count
#include "engine_internal.hip.h"

using namespace gt;
// ------------------------------------------------------------------------------------------
// data-parallel communicator (gt_comm_init / gt_comm_destroy).
// One process per GPU; every rank holds the full G / D and a shard of the minibatch (whole sequences).  With a
// communicator attached, the fused step functions are data-parallel by themselves: the valid-frame count, every
// network's gradient and the additive loss sums are summed over the ranks with RCCL (the ROCm build of NCCL, xGMI
// between the GPUs of a node) on a separate HIP stream, each gradient bucket (one layer) as soon as its weight-gradient
// reduction has finished -- i.e. UNDER the rest of the backward pass -- and clip-norm + optimizer run on the reduced
// gradient, so all replicas take bit-identical steps.  RCCL is bound at run time (dlopen), the library has no link
// dependency on it; a host in any language drives this through the C ABI.
// ------------------------------------------------------------------------------------------
#include <dlfcn.h>
struct GtNcclId { char internal[GT_COMM_ID_BYTES]; };
struct RcclApi {
  void* lib = nullptr;
  int (*GetUniqueId)(GtNcclId*) = nullptr;
  int (*CommInitRank)(void**, int, GtNcclId, int) = nullptr;
  int (*CommDestroy)(void*) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, void*, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
};
static RcclApi* rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.lib ? &api : nullptr;
  tried = true;
  // GT_RCCL_LIB=<path>: bind that library instead (tests/fake_rccl.cpp -- a shared-memory test double that lets two
  // processes on ONE GPU run a world-2 communicator; RCCL itself refuses two ranks on one device).  Otherwise prefer
  // the copy that is already in the process (PyTorch ships its own librccl), then the ROCm one.
  const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void* h = nullptr;
  const char* forced = getenv("GT_RCCL_LIB");
  if (forced && forced[0]) {
    h = dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "gantts_hip: GT_RCCL_LIB=%s could not be loaded: %s\n", forced, dlerror()); return nullptr; }
  }
  if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
  if (!h) for (const char* n : names) if ((h = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
  if (!h) return nullptr;
#define GT_SYM(field, name) *(void**)(&api.field) = dlsym(h, name)
  GT_SYM(GetUniqueId, "ncclGetUniqueId"); GT_SYM(CommInitRank, "ncclCommInitRank"); GT_SYM(CommDestroy, "ncclCommDestroy");
  GT_SYM(AllReduce, "ncclAllReduce"); GT_SYM(GroupStart, "ncclGroupStart"); GT_SYM(GroupEnd, "ncclGroupEnd");
  GT_SYM(GetErrorString, "ncclGetErrorString");
#undef GT_SYM
  if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.GroupStart || !api.GroupEnd) return nullptr;
  api.lib = h;
  return &api;
}
#define NCCLCHK(expr)                                                                                      \
  do {                                                                                                     \
    int _r = (expr);                                                                                       \
    if (_r != 0) return fail(GT_ERR_HIP, "%s failed: %s", #expr, rccl_api()->GetErrorString ? rccl_api()->GetErrorString(_r) : "?"); \
  } while (0)
enum { GT_NCCL_SUM = 0, GT_NCCL_FLOAT = 7, GT_NCCL_DOUBLE = 8 };


// A sum over ONE rank is the identity: a single-rank communicator takes the plain path (no collective, no second
// stream).  GT_COMM_FORCE_COLLECTIVES=1 issues every call anyway -- the launch / cross-stream cost of the schedule can
// then be measured on one GPU (bench.py --force-dp).

static void trace_clear(GtComm* c);
extern "C" int gt_comm_unique_id(void* id_out) {
  if (!id_out) return fail(GT_ERR_INVALID, "null argument");
  RcclApi* api = rccl_api();
  if (!api) return fail(GT_ERR_HIP, "RCCL (librccl.so) could not be loaded");
  NCCLCHK(api->GetUniqueId((GtNcclId*)id_out));
  return GT_OK;
}
extern "C" int gt_comm_destroy(gt_engine* e) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  ipc_destroy(e);
  if (!e->comm) return GT_OK;
  (void)hipDeviceSynchronize();
  GtComm* c = e->comm;
  if (c->comm && rccl_api()) (void)rccl_api()->CommDestroy(c->comm);
  for (auto& ev : c->ev) if (ev) (void)hipEventDestroy(ev);
  if (c->ev_done) (void)hipEventDestroy(c->ev_done);
  trace_clear(c);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
  e->comm = nullptr;
  e->dp_rank = 0; e->dp_world = 1;
  return GT_OK;
}
extern "C" int gt_comm_init(gt_engine* e, int rank, int world, const void* id) {
  if (!e || !id || world < 1 || rank < 0 || rank >= world) return fail(GT_ERR_INVALID, "bad argument");
  RcclApi* api = rccl_api();
  if (!api) return fail(GT_ERR_HIP, "RCCL (librccl.so) could not be loaded");
  CHK(gt_comm_destroy(e));
  GtComm* c = new GtComm();
  c->rank = rank; c->world = world;
  e->comm = c;
  e->dp_rank = rank; e->dp_world = world;
  GtNcclId nid;
  memcpy(&nid, id, sizeof(nid));
  int r = api->CommInitRank(&c->comm, world, nid, rank);
  if (r != 0) { c->comm = nullptr; (void)gt_comm_destroy(e); return fail(GT_ERR_HIP, "ncclCommInitRank failed: %s", api->GetErrorString ? api->GetErrorString(r) : "?"); }
  HIPCHK(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  // stream-to-stream hand-offs on ONE device: a device-scope release is enough (the collective's own kernels fence what leaves the
  // device), and a record then does not flush to system scope in the middle of the step
  const unsigned evf = hipEventDisableTiming | (env_flag("GT_COMM_EVENT_DEVICE_SCOPE", true) ? hipEventReleaseToDevice : 0u);
  for (auto& ev : c->ev) HIPCHK(hipEventCreateWithFlags(&ev, evf));
  HIPCHK(hipEventCreateWithFlags(&c->ev_done, evf));
  CHK(e->comm_tv.ensure(64));
  return GT_OK;
}
// The engine's shard of the minibatch without a communicator (hosts that all-reduce themselves between the split-phase
// calls, gantts_amd/parallel.py): sequence b of this engine is sequence rank + world * b of the whole minibatch.  Keys the
// dropout streams globally, so that every rank draws its own rows of the ONE mask a single process would draw.
extern "C" int gt_set_shard(gt_engine* e, int rank, int world) {
  if (!e || world < 1 || rank < 0 || rank >= world) return fail(GT_ERR_INVALID, "bad argument");
  if (e->comm && (e->comm->rank != rank || e->comm->world != world))
    return fail(GT_ERR_STATE, "gt_set_shard(%d, %d) contradicts the attached communicator (%d, %d)", rank, world, e->comm->rank, e->comm->world);
  e->dp_rank = rank; e->dp_world = world;
  return GT_OK;
}
extern "C" int gt_comm_info(gt_engine* e, int* rank, int* world) {
  if (!e || !rank || !world) return fail(GT_ERR_INVALID, "null argument");
  *rank = e->comm ? e->comm->rank : 0;
  *world = e->comm ? e->comm->world : 1;
  return GT_OK;
}

// Schedule trace: while it is on, comm_allreduce_after brackets every message with two timed events on the stream that carries it
// (the first one behind the hand-off wait: the message may start there) and comm_join brackets the step stream's wait, so a run tells
// how many messages a step sends, how long each one holds the communicator's stream and how long the step stream stands still for them.
static void trace_clear(GtComm* c) {
  for (auto& r : c->trec) { if (r.e0) (void)hipEventDestroy(r.e0); if (r.e1) (void)hipEventDestroy(r.e1); }
  c->trec.clear();
  c->trace_dropped = 0;
}
static GtComm::TraceRec* trace_open(GtComm* c, int kind, bool inl, double bytes, hipStream_t on) {
  if (!c->trace) return nullptr;
  if (c->trec.size() >= 65536) { ++c->trace_dropped; return nullptr; }      // (reported by gt_comm_trace_read)
  GtComm::TraceRec r;
  r.kind = kind; r.inl = inl ? 1 : 0; r.bytes = bytes; r.e0 = r.e1 = nullptr; r.closed = false;
  if (hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess || hipEventRecord(r.e0, on) != hipSuccess) {
    if (r.e0) (void)hipEventDestroy(r.e0);
    if (r.e1) (void)hipEventDestroy(r.e1);
    return nullptr;
  }
  c->trec.push_back(r);
  return &c->trec.back();
}
extern "C" int gt_comm_trace(gt_engine* e, int enable) {
  if (!e) return fail(GT_ERR_INVALID, "null engine");
  if (!e->comm) return fail(GT_ERR_STATE, "gt_comm_trace without a communicator");
  HIPCHK(hipDeviceSynchronize());
  if (enable) trace_clear(e->comm);
  e->comm->trace = enable != 0;
  return GT_OK;
}
extern "C" int gt_comm_trace_read(gt_engine* e, double* out, int max_records, int* n_records) {
  if (!e || !n_records || (max_records > 0 && !out)) return fail(GT_ERR_INVALID, "null argument");
  if (!e->comm) return fail(GT_ERR_STATE, "gt_comm_trace_read without a communicator");
  HIPCHK(hipDeviceSynchronize());
  GtComm* c = e->comm;
  if (c->trace_dropped > 0)
    return fail(GT_ERR_STATE, "schedule trace full (65536 records): %ld messages / joins were not recorded -- trace fewer steps", c->trace_dropped);
  int n = 0, n_closed = 0;
  for (size_t i = 0; i < c->trec.size(); ++i) n_closed += c->trec[i].closed ? 1 : 0;
  for (size_t i = 0; i < c->trec.size() && n < max_records; ++i) {
    const GtComm::TraceRec& r = c->trec[i];
    if (!r.closed) continue;                    // the message failed between its two events (its error was reported then)
    float t0 = 0.f, t1 = 0.f;
    HIPCHK(hipEventElapsedTime(&t0, c->trec[0].e0, r.e0));
    HIPCHK(hipEventElapsedTime(&t1, c->trec[0].e0, r.e1));
    double* o = out + 5 * (size_t)n++;
    o[0] = r.kind; o[1] = r.bytes; o[2] = r.inl; o[3] = 1e3 * (double)t0; o[4] = 1e3 * (double)t1;
  }
  *n_records = max_records > 0 ? n : n_closed;
  return GT_OK;
}

bool comm_on(const gt_engine* e) {
  return e->comm != nullptr && (e->comm->world > 1 || e->opt_comm_force);
}
// all-reduce(sum) of buf[0..count) in place on the communicator's stream, ordered after everything queued on `compute`
// on_compute: the collective is issued on `compute` itself -- for a step's CLOSING messages, which nothing can overlap: no event
// hand-off to the communicator's stream and back on the critical path (RCCL orders successive calls on one communicator)
static int comm_allreduce_after(gt_engine* e, void* buf, size_t count, int dtype, hipStream_t compute, bool on_compute = false) {
  GtComm* c = e->comm;
  // messages that fit the interprocess arena's slots take the full-mesh two-shot path (eng_ipc.hip) once the arenas are attached
  const bool dbl = dtype == GT_NCCL_DOUBLE;
  const bool ipc = ipc_usable(e, count * (dbl ? sizeof(double) : sizeof(float)));
  const double bytes = (double)count * (dbl ? sizeof(double) : sizeof(float));
  hipStream_t on = compute;
  if (!on_compute) {
    hipEvent_t ev = c->ev[c->next_ev];
    c->next_ev = (c->next_ev + 1) % 8;
    HIPCHK(hipEventRecord(ev, compute));
    HIPCHK(hipStreamWaitEvent(c->stream, ev, 0));
    on = c->stream;
  }
  GtComm::TraceRec* tr = trace_open(c, 0, on_compute, bytes, on);
  const long tr_i = tr ? (long)c->trec.size() - 1 : -1;       // (a record that is not closed below -- the message failed -- is skipped by the reader)
  if (ipc) CHK(ipc_allreduce(e, buf, count, dbl, on));
  else NCCLCHK(rccl_api()->AllReduce(buf, buf, count, dtype, GT_NCCL_SUM, c->comm, on));
  if (tr_i >= 0) { HIPCHK(hipEventRecord(c->trec[tr_i].e1, on)); c->trec[tr_i].closed = true; }
  return GT_OK;
}
// `compute` continues only after everything handed to the communicator so far has finished
static int comm_join(gt_engine* e, hipStream_t compute) {
  GtComm* c = e->comm;
  HIPCHK(hipEventRecord(c->ev_done, c->stream));
  GtComm::TraceRec* tr = trace_open(c, 1, false, 0.0, compute);
  const long tr_i = tr ? (long)c->trec.size() - 1 : -1;
  HIPCHK(hipStreamWaitEvent(compute, c->ev_done, 0));
  if (tr_i >= 0) { HIPCHK(hipEventRecord(c->trec[tr_i].e1, compute)); c->trec[tr_i].closed = true; }
  return GT_OK;
}
// the gradient of [lo, lo + count) of `role` is final on `compute`.  Ranges are collected and handed to RCCL by
// comm_flush in as few messages as their adjacency allows (a small all-reduce is pure latency: the layers above the
// first one leave together, under the first layer's backward; only the first layer's message is exposed).
int comm_grads_ready(gt_engine* e, int role, const float* lo, long count, hipStream_t compute) {
  if (!comm_on(e) || !lo || count <= 0) return GT_OK;
  Net& n = e->net[role];
  const long off = lo - n.d.grads;
  if (off < 0 || off + count > n.d.n_params) return fail(GT_ERR_STATE, "gradient bucket outside the bound buffer");
  e->comm_pending[role].push_back(std::make_pair(off, count));
  return GT_OK;
}
int comm_flush(gt_engine* e, int role, hipStream_t compute, bool closing) {
  if (!comm_on(e)) return GT_OK;
  // the fused step only RECORDS its weight-gradient combines (SlabDefer): the ranges about to leave must be final, so the recorded
  // combines run now, as ONE launch per message instead of one per layer
  CHK(slab_defer_flush(e->sdefer[role], compute));
  auto& pend = e->comm_pending[role];
  if (pend.empty()) return GT_OK;
  Net& n = e->net[role];
  std::sort(pend.begin(), pend.end());
  size_t i = 0;
  while (i < pend.size()) {
    long lo = pend[i].first, hi = lo + pend[i].second;
    size_t j = i + 1;
    while (j < pend.size() && pend[j].first <= hi) { hi = std::max(hi, pend[j].first + pend[j].second); ++j; }
    CHK(comm_allreduce_after(e, n.d.grads + lo, (size_t)(hi - lo), GT_NCCL_FLOAT, compute, closing));
    e->comm_done[role].push_back(std::make_pair(lo, hi - lo));
    i = j;
  }
  pend.clear();
  return GT_OK;
}
// end of a backward pass: whatever part of the flat gradient no bucket covered, plus the step's additive loss sums
// (`n_sums` doubles at `sums`), then `compute` waits for the communicator
int comm_finish_step(gt_engine* e, int role, bool grads, double* sums, int n_sums, hipStream_t compute) {
  if (!comm_on(e)) return GT_OK;
  Net& n = e->net[role];
  const bool grp = e->opt_comm_group && rccl_api();
  const bool inl = e->opt_comm_close_inline;
  // closing messages on the step's own stream: whatever the communicator's stream still carries (earlier buckets of this step)
  // must be ordered in front of them for the final join below to cover everything; RCCL serialises the calls of a communicator
  // in issue order, and the step stream also waits for the communicator's stream explicitly before the first inline call
  if (inl) CHK(comm_join(e, compute));
  if (grp) NCCLCHK(rccl_api()->GroupStart());       // the closing messages of a step (rest of the gradient + loss sums): one launch
  if (grads) {
    CHK(comm_flush(e, role, compute, inl));
    auto& done = e->comm_done[role];
    std::sort(done.begin(), done.end());
    long pos = 0;
    for (size_t i = 0; i <= done.size(); ++i) {
      const long next = i < done.size() ? done[i].first : (long)n.d.n_params;
      if (next > pos) CHK(comm_allreduce_after(e, n.d.grads + pos, (size_t)(next - pos), GT_NCCL_FLOAT, compute, inl));
      if (i < done.size()) pos = std::max(pos, done[i].first + done[i].second);
    }
  }
  e->comm_done[role].clear();
  e->comm_pending[role].clear();
  if (sums && n_sums > 0) CHK(comm_allreduce_after(e, sums, (size_t)n_sums, GT_NCCL_DOUBLE, compute, inl));
  if (grp) NCCLCHK(rccl_api()->GroupEnd());
  return inl ? GT_OK : comm_join(e, compute);
}

// Data-parallel early results: the step's loss sums are final on `compute` here, long before its backward pass is.
// They are summed over the ranks, finalised and copied to the host ON THE COMMUNICATOR'S STREAM, so the fused call can
// return as soon as that copy lands while the backward pass, its gradient buckets and the optimizer keep going.
int comm_early_results(gt_engine* e, int role, double* sums, int n_sums, float adv_w, float mse_w, float mge_w, hipStream_t compute) {
  GtComm* c = e->comm;
  CHK(comm_allreduce_after(e, sums, (size_t)n_sums, GT_NCCL_DOUBLE, compute));
  StepResults* target = e->h_res_dev ? e->h_res_dev : e->res();     // see post_early_results
  if (role == GT_ROLE_D) hipLaunchKernelGGL(finalize_d_kernel, dim3(1), dim3(1), 0, c->stream, e->sc(), target, 1, e->d_unnorm ? 1 : 0);
  else hipLaunchKernelGGL(finalize_g_kernel, dim3(1), dim3(1), 0, c->stream, e->sc(), target, adv_w, mse_w, mge_w, e->g_has_adv ? 1 : 0, 1,
                          (const double*)nullptr, 0, (const double*)nullptr, 0);
  LAUNCH_CHECK();
  return post_early_results(e, c->stream);
}

// tv = sum(mask) (or the data-parallel override) -> device scalars; once per (step, mask).  With a communicator the
// count is the GLOBAL one: losses are normalised by the valid frames of the whole minibatch (train.py:258, seqloss.py:43).
// Two halves so that the all-reduce of the count runs under the forward pass that precedes its first use:
// ensure_tv_begin where the mask is first seen, ensure_tv right before the first kernel that reads the normaliser.
int ensure_tv_begin(gt_engine* e, const float* mask, long N, hipStream_t s) {
  if (e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override) return GT_OK;
  if (comm_on(e) && !e->tv_dev && !(e->tv_override > 0.f) && !e->tv_inflight) {
    hipLaunchKernelGGL(mask_total_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, e->comm_tv.as<double>());
    LAUNCH_CHECK();
    CHK(comm_allreduce_after(e, e->comm_tv.p, 1, GT_NCCL_DOUBLE, s));
    e->tv_inflight = true;
  }
  return GT_OK;
}
// riders (eng_step.hip): the local count was written to comm_tv by a rider of another launch / is consumed by the head directly
int comm_tv_sent(gt_engine* e, hipStream_t s) {
  CHK(comm_allreduce_after(e, e->comm_tv.p, 1, GT_NCCL_DOUBLE, s));
  e->tv_inflight = true;
  return GT_OK;
}
int comm_tv_join(gt_engine* e, hipStream_t s) {
  CHK(comm_join(e, s));
  e->tv_inflight = false;
  return GT_OK;
}
int ensure_tv(gt_engine* e, const float* mask, long N, hipStream_t s) {
  if (e->tv_mask == mask && e->tv_n == N && e->tv_ovr == e->tv_override) return GT_OK;
  const double* tv_dev = e->tv_dev;
  if (comm_on(e) && !tv_dev && !(e->tv_override > 0.f)) {
    CHK(ensure_tv_begin(e, mask, N, s));
    CHK(comm_join(e, s));
    e->tv_inflight = false;
    tv_dev = e->comm_tv.as<double>();
  }
  hipLaunchKernelGGL(mask_sum_kernel, dim3(1), dim3(1024), 0, s, mask, (int)N, e->tv_override, tv_dev, e->sc());
  LAUNCH_CHECK();
  e->tv_mask = mask; e->tv_n = N; e->tv_ovr = e->tv_override;
  return GT_OK;
}

