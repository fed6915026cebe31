// libgantts_hip.so -- the small-message collective of SURVEY 8(e): a full-mesh TWO-SHOT all-reduce over hipIpc peer buffers.
// See engine_internal.hip.h; C ABI in include/gantts_hip.h (gt_comm_ipc_*).
//
// Why: the step's messages are small (cfg2: 3.4 MB of generator gradient, 1 MB of discriminator gradient, five and three doubles
// of sums) and xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring / tree collective pays 2 (W - 1) link latencies per
// message, a full mesh pays two.  Every rank owns an ARENA in its own HBM -- [flags | in slot | out slot] -- exported with
// hipIpcGetMemHandle and mapped by all peers.  One all-reduce of n elements is three launches on the caller's stream:
//   publish : buf -> my in slot
//   reduce  : tell every peer "my input of message #seq is complete" (a store into THEIR flags), wait until all of them have told me,
//             then reduce MY 1/W chunk -- reading that chunk from every rank's in slot, summed in rank order, so every element of
//             the result is computed by exactly one rank and all replicas receive the same bits -- and PUSH the sums into every
//             rank's out slot (reduce-scatter and all-gather of a two-shot all-reduce in one pass over the chunk);
//   collect : tell every peer "my chunk has landed in your out slot", wait for all of them, out slot -> buf.
// The launch boundaries give the intra-device ordering (all workgroups of `publish` are done before `reduce` announces it), the
// two cross-device waits are spins on flags in the spinning rank's OWN memory, bounded by a wall-clock timeout that raises the
// engine's fault word instead of hanging (include/gantts_hip.h: a fault under a communicator is fatal for the job).
// Slot reuse is safe without further handshakes: a rank leaves `collect` of message k only when every peer has finished `reduce`
// of message k (nobody still reads its in slot), and peers push message k + 1 into its out slot only after it has announced
// message k + 1, which its stream does after `collect` of message k.
// Messages above the slot size, and everything when no arena is attached, go to RCCL (eng_comm.hip).
#include "engine_internal.hip.h"

namespace gt {

constexpr size_t IPC_FLAGS_BYTES = 4096;
constexpr size_t IPC_SLOT_BYTES = (size_t)8 << 20;
constexpr size_t IPC_ARENA_BYTES = IPC_FLAGS_BYTES + 2 * IPC_SLOT_BYTES;
constexpr unsigned IPC_FAULT_TIMEOUT = 0x200u;
constexpr unsigned long long IPC_TIMEOUT_TICKS = 1000000000ULL;      // 10 s of the 100 MHz wall clock (two processes on ONE GPU may be time-sliced)

struct IpcFlags {
  unsigned arrive[GT_IPC_MAX_WORLD];      // arrive[p] = seq: rank p's input of message #seq is complete (written by rank p)
  unsigned pad0[16 - GT_IPC_MAX_WORLD];
  unsigned done[GT_IPC_MAX_WORLD];        // done[p] = seq: rank p's chunk of message #seq has landed in MY out slot (written by rank p)
};
struct IpcPeers { char* arena[GT_IPC_MAX_WORLD]; };

__device__ __forceinline__ IpcFlags* ipc_flags(char* arena) { return reinterpret_cast<IpcFlags*>(arena); }
template <typename T> __device__ __forceinline__ T* ipc_in(char* arena) { return reinterpret_cast<T*>(arena + IPC_FLAGS_BYTES); }
template <typename T> __device__ __forceinline__ T* ipc_out(char* arena) { return reinterpret_cast<T*>(arena + IPC_FLAGS_BYTES + IPC_SLOT_BYTES); }

// lanes 0 .. world-1 of every workgroup: spin until flag[lane] has reached `seq` (sequence numbers wrap: signed distance)
__device__ __forceinline__ void ipc_wait_all(const unsigned* flag, int world, unsigned seq, unsigned int* fault) {
  if ((int)threadIdx.x < world) {
    const unsigned long long t0 = wall_clock64();
    while ((int)(__hip_atomic_load(flag + threadIdx.x, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) - seq) < 0) {
      __builtin_amdgcn_s_sleep(8);
      // an IPC timeout raised by another workgroup of this message ends the wait at once; a FOREIGN fault bit (an earlier,
      // uncleared recurrence timeout: the step's updates are skipped anyway) does not -- the peers are still coming, and
      // giving up early would reduce unsynchronised data and mis-report the fault as an interprocess one
      if ((__hip_atomic_load(fault, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & IPC_FAULT_TIMEOUT) != 0u) break;
      if (wall_clock64() - t0 > IPC_TIMEOUT_TICKS) { atomicOr(fault, IPC_FAULT_TIMEOUT); break; }
    }
  }
  __syncthreads();
}

template <typename T>
static __global__ __launch_bounds__(256) void ipc_publish_kernel(const T* __restrict__ buf, char* arena, long n) {
  T* in = ipc_in<T>(arena);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) in[i] = buf[i];
  __threadfence_system();
}

template <typename T>
static __global__ __launch_bounds__(256) void ipc_reduce_kernel(const IpcPeers P, int rank, int world, long n, unsigned seq, unsigned int* fault) {
  if (blockIdx.x == 0 && (int)threadIdx.x < world)
    __hip_atomic_store(&ipc_flags(P.arena[threadIdx.x])->arrive[rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  ipc_wait_all(ipc_flags(P.arena[rank])->arrive, world, seq, fault);
  const long chunk = (n + world - 1) / world;
  const long c0 = (long)rank * chunk, c1 = min(n, c0 + chunk);
  for (long i = c0 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < c1; i += (long)gridDim.x * blockDim.x) {
    T s = __builtin_nontemporal_load(ipc_in<T>(P.arena[0]) + i);
    for (int p = 1; p < world; ++p) s += __builtin_nontemporal_load(ipc_in<T>(P.arena[p]) + i);      // rank order: the same bits on every replica
    for (int p = 0; p < world; ++p) __builtin_nontemporal_store(s, ipc_out<T>(P.arena[p]) + i);
  }
  __threadfence_system();
}

template <typename T>
static __global__ __launch_bounds__(256) void ipc_collect_kernel(const IpcPeers P, int rank, int world, T* __restrict__ buf, long n, unsigned seq,
                                                                 unsigned int* fault) {
  if (blockIdx.x == 0 && (int)threadIdx.x < world)
    __hip_atomic_store(&ipc_flags(P.arena[threadIdx.x])->done[rank], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  ipc_wait_all(ipc_flags(P.arena[rank])->done, world, seq, fault);
  const T* out = ipc_out<T>(P.arena[rank]);
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) buf[i] = __builtin_nontemporal_load(out + i);
}

}  // namespace gt
using namespace gt;

struct GtIpc {
  char* arena = nullptr;
  char* peer[GT_IPC_MAX_WORLD] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};     // peer[rank] == arena
  int rank = 0, world = 1;
  unsigned seq = 0;
  long long messages = 0;
  bool attached = false;
  bool fine_grained = false;      // the arena is fine-grained device memory: peers' stores / loads are coherent INSIDE a running kernel
};

void ipc_destroy(gt_engine* e) {
  GtIpc* c = e->ipc;
  if (!c) return;
  (void)hipDeviceSynchronize();
  for (int p = 0; p < GT_IPC_MAX_WORLD; ++p)
    if (c->peer[p] && c->peer[p] != c->arena) (void)hipIpcCloseMemHandle(c->peer[p]);
  if (c->arena) (void)hipFree(c->arena);
  delete c;
  e->ipc = nullptr;
}

bool ipc_usable(const gt_engine* e, size_t bytes) {
  return e->ipc && e->ipc->attached && e->opt_comm_ipc && bytes <= IPC_SLOT_BYTES;
}

template <typename T>
static int ipc_allreduce_t(gt_engine* e, T* buf, long n, hipStream_t s) {
  GtIpc* c = e->ipc;
  IpcPeers P;
  for (int p = 0; p < GT_IPC_MAX_WORLD; ++p) P.arena[p] = c->peer[p];
  if (++c->seq == 0) ++c->seq;
  const int grid = (int)std::max<long>(1, std::min<long>(256, cdiv(n, 256 * 4)));
  const long chunk = (n + c->world - 1) / c->world;
  const int grid_c = (int)std::max<long>(1, std::min<long>(256, cdiv(chunk, 256 * 2)));
  hipLaunchKernelGGL(ipc_publish_kernel<T>, dim3(grid), dim3(256), 0, s, (const T*)buf, c->arena, n);
  hipLaunchKernelGGL(ipc_reduce_kernel<T>, dim3(grid_c), dim3(256), 0, s, P, c->rank, c->world, n, c->seq, e->d_fault);
  hipLaunchKernelGGL(ipc_collect_kernel<T>, dim3(grid), dim3(256), 0, s, P, c->rank, c->world, buf, n, c->seq, e->d_fault);
  LAUNCH_CHECK();
  c->messages += 1;
  return GT_OK;
}
// all-reduce(sum) of buf[0 .. count) in place on `s`; dtype_double: the elements are doubles (the loss / count sums), else floats
int ipc_allreduce(gt_engine* e, void* buf, size_t count, bool dtype_double, hipStream_t s) {
  return dtype_double ? ipc_allreduce_t<double>(e, (double*)buf, (long)count, s) : ipc_allreduce_t<float>(e, (float*)buf, (long)count, s);
}

extern "C" int gt_comm_ipc_export(gt_engine* e, void* handle_out) {
  if (!e || !handle_out) return fail(GT_ERR_INVALID, "null argument");
  static_assert(sizeof(hipIpcMemHandle_t) <= GT_IPC_HANDLE_BYTES, "hipIpcMemHandle_t does not fit GT_IPC_HANDLE_BYTES");
  static_assert(sizeof(IpcFlags) <= IPC_FLAGS_BYTES, "flags block");
  ipc_destroy(e);
  GtIpc* c = new GtIpc();
  e->ipc = c;
  // The two-shot protocol needs visibility INSIDE running kernels: peers store flags into this rank's arena while its reduce /
  // collect kernels spin on them, and read its in slot mid-kernel.  Only fine-grained device memory gives that across devices
  // (coarse-grained allocations are coherent at kernel boundaries only: a reader on another device may be served a stale line from
  // its own L2).  So the export FAILS where the runtime does not offer fine-grained memory -- RCCL then carries every message --
  // instead of silently falling back to hipMalloc.  GT_IPC_ALLOW_COARSE=1 permits the coarse arena for the one setting
  // where it is coherent: all ranks on ONE device (one L2 -- the two-process test of tests/test_gpu_comm2.py).
  void* p = nullptr;
  if (hipExtMallocWithFlags(&p, IPC_ARENA_BYTES, hipDeviceMallocFinegrained) == hipSuccess) {
    c->fine_grained = true;
  } else {
    (void)hipGetLastError();
    p = nullptr;
    const char* allow = getenv("GT_IPC_ALLOW_COARSE");
    if (!(allow && atoi(allow) == 1)) {
      ipc_destroy(e);
      return fail(GT_ERR_HIP, "gt_comm_ipc_export: no fine-grained device memory (hipExtMallocWithFlags(hipDeviceMallocFinegrained) failed): the two-shot "
                              "all-reduce needs in-kernel cross-device coherence; RCCL stays the collective (GT_IPC_ALLOW_COARSE=1: ranks sharing ONE device only)");
    }
    if (hipMalloc(&p, IPC_ARENA_BYTES) != hipSuccess) { ipc_destroy(e); return fail(GT_ERR_HIP, "the interprocess arena could not be allocated"); }
  }
  c->arena = (char*)p;
  if (hipMemset(c->arena, 0, IPC_FLAGS_BYTES) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { ipc_destroy(e); return fail(GT_ERR_HIP, "hipMemset failed"); }
  hipIpcMemHandle_t h;
  hipError_t r = hipIpcGetMemHandle(&h, c->arena);
  if (r != hipSuccess) {
    (void)hipGetLastError();
    ipc_destroy(e);
    return fail(GT_ERR_HIP, "hipIpcGetMemHandle failed: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 is needed where only dmabuf IPC is supported)", hipGetErrorString(r));
  }
  memset(handle_out, 0, GT_IPC_HANDLE_BYTES);
  memcpy(handle_out, &h, sizeof(h));
  return GT_OK;
}

extern "C" int gt_comm_ipc_attach(gt_engine* e, int rank, int world, const void* handles) {
  if (!e || !handles) return fail(GT_ERR_INVALID, "null argument");
  if (world < 1 || world > GT_IPC_MAX_WORLD || rank < 0 || rank >= world) return fail(GT_ERR_INVALID, "bad rank / world (at most %d ranks)", GT_IPC_MAX_WORLD);
  if (!e->ipc || !e->ipc->arena) return fail(GT_ERR_STATE, "gt_comm_ipc_attach without gt_comm_ipc_export");
  if (e->comm && (e->comm->rank != rank || e->comm->world != world))
    return fail(GT_ERR_STATE, "gt_comm_ipc_attach(%d, %d) contradicts the attached communicator (%d, %d)", rank, world, e->comm->rank, e->comm->world);
  GtIpc* c = e->ipc;
  if (c->attached) return fail(GT_ERR_STATE, "the interprocess arenas are attached already");
  c->rank = rank; c->world = world;
  for (int p = 0; p < world; ++p) {
    if (p == rank) { c->peer[p] = c->arena; continue; }
    hipIpcMemHandle_t h;
    memcpy(&h, (const char*)handles + (size_t)p * GT_IPC_HANDLE_BYTES, sizeof(h));
    void* q = nullptr;
    hipError_t r = hipIpcOpenMemHandle(&q, h, hipIpcMemLazyEnablePeerAccess);
    if (r != hipSuccess) {
      (void)hipGetLastError();
      // leave nothing half-mapped: a retry would open the earlier peers' handles a second time
      for (int q2 = 0; q2 < p; ++q2)
        if (c->peer[q2] && c->peer[q2] != c->arena) (void)hipIpcCloseMemHandle(c->peer[q2]);
      for (int q2 = 0; q2 < GT_IPC_MAX_WORLD; ++q2) c->peer[q2] = nullptr;
      return fail(GT_ERR_HIP, "hipIpcOpenMemHandle of rank %d's arena failed: %s", p, hipGetErrorString(r));
    }
    c->peer[p] = (char*)q;
  }
  c->attached = true;
  return GT_OK;
}

extern "C" int gt_comm_ipc_messages(gt_engine* e, long long* n) {
  if (!e || !n) return fail(GT_ERR_INVALID, "null argument");
  *n = e->ipc ? e->ipc->messages : 0;
  return GT_OK;
}
