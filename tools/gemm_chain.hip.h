// Layer chains in ONE launch (gfx950 only): consecutive frame x weight products of a network whose operand A is the
// previous product's output (the hidden stack of gantts/models.py:129-141 forward, and its backward-data pass), run by
// persistent workgroups that pull 64x64 tiles from per-XCD queues.
//
// Why: a per-layer launch of these K = 256 .. 512 products pays ~14 us of launch edge (prologue, epilogue drain, arrival
// skew) for 45 .. 65 us of matrix work (tools/gemm_tile_sweep.hip: the K sweep's intercept).  In a chain the only edges
// left are the first prologue and the last epilogue: a tile of layer l+1 starts as soon as the 64-row panel it reads is
// complete in layer l, while other panels of layer l are still being computed.
//
// Placement and visibility: row panel p (64 frames) belongs to XCD p % nxcd in EVERY op of the chain.  A workgroup reads
// its XCC_ID register and serves only that XCD's queue, so the tiles that produce a panel and the tiles that consume it
// run on one XCD and meet in its L2 (coherent inside the XCD; the per-CU L1 is write-through and never holds a line of a
// panel before that panel's flag was seen, because no address is read before it is written inside the launch).  Producer:
// epilogue stores, s_waitcnt vmcnt(0) (all stores acknowledged by L2), workgroup barrier, one agent-scope atomic
// increment of done[op][panel].  Consumer: agent-scope (sc1) polls of that counter by one lane, with s_sleep, bounded by
// a wall-clock timeout that raises the engine's fault word instead of hanging.
//
// Liveness does not depend on how many workgroups are resident: a tile only waits for tiles that sit EARLIER in the same
// queue, i.e. were already taken by a running workgroup.  If an XCD received no workgroup at all its queue would stay
// unserved: the last workgroup to leave checks every queue head and raises the fault word (never a silent wrong result).
#pragma once
#include "gemm_f32.hip.h"

namespace gt {

constexpr int CHAIN_MAX_OPS = 6;
constexpr int CHAIN_BM = 64, CHAIN_BN = 64;

struct ChainOp {
  GemmArgs g;           // n_tiles_m / n_tiles_n filled for 64x64 tiles
  int kind;             // GEMM_NT or GEMM_NN
  int dep;              // op whose row panel must be complete before a tile of this op starts, or -1
  int done_base;        // offset of this op's per-panel counters inside ctl (after the queue heads)
  int pad_;
};
struct ChainArgs {
  int n_ops, nxcd;
  int ctl_words, pad_;
  unsigned int* ctl;    // [nxcd] queue heads, [1] exit counter, [...] done counters; all zero at launch, and left zeroed by
                        // the last workgroup of the launch for the next one
  unsigned int* fault;  // engine fault word (device)
  unsigned long long timeout_ticks;   // 100 MHz wall-clock ticks
  unsigned long long* dbg;            // diagnosis only (GT_CHAIN_DBG): per workgroup {tiles, dequeue, wait, tile, publish, total} ticks
  ChainOp op[CHAIN_MAX_OPS];
};
constexpr unsigned CHAIN_FAULT_TIMEOUT = 0x100u, CHAIN_FAULT_UNSERVED = 0x200u;

__device__ __forceinline__ int chain_items_of(const ChainOp& o, int x, int nxcd) {
  const int ntm = o.g.n_tiles_m;
  return ntm > x ? ((ntm - x + nxcd - 1) / nxcd) * o.g.n_tiles_n : 0;
}

__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_chain_kernel(const ChainArgs c) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int sh_item;
  const int tid = threadIdx.x;
  const int x = (int)(xcc_id_reg() & 0xfu) % c.nxcd;
  unsigned int* heads = c.ctl;
  unsigned int* done = c.ctl + c.nxcd + 1;
  unsigned long long d_n = 0, d_deq = 0, d_wait = 0, d_tile = 0, d_pub = 0;
  const unsigned long long d_t0 = c.dbg ? wall_clock64() : 0ull;
  for (;;) {
    __syncthreads();                       // the previous tile's LDS (and sh_item) is dead from here on
    unsigned long long ta = c.dbg ? wall_clock64() : 0ull;
    if (tid == 0) sh_item = (int)__hip_atomic_fetch_add(heads + x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int j = __builtin_amdgcn_readfirstlane(sh_item);    // uniform: op descriptors are read with scalar loads
    int o = 0;
    for (; o < c.n_ops; ++o) {
      const int cnt = chain_items_of(c.op[o], x, c.nxcd);
      if (j < cnt) break;
      j -= cnt;
    }
    if (o >= c.n_ops) break;
    const ChainOp& op = c.op[o];
    const int ntn = op.g.n_tiles_n;
    const int tile_m = x + c.nxcd * (j / ntn), tile_n = j % ntn;
    unsigned long long tb = c.dbg ? wall_clock64() : 0ull;
    if (op.dep >= 0) {
      if (tid == 0) {
        const unsigned int* flag = done + c.op[op.dep].done_base + tile_m;
        const unsigned int need = (unsigned)c.op[op.dep].g.n_tiles_n;
        const unsigned long long t0 = wall_clock64();
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < need) {
          __builtin_amdgcn_s_sleep(4);
          if (wall_clock64() - t0 > c.timeout_ticks) { atomicOr(c.fault, CHAIN_FAULT_TIMEOUT); break; }
        }
      }
      __syncthreads();
    }
    unsigned long long tc = c.dbg ? wall_clock64() : 0ull;
    if (op.kind == GEMM_NT) gemm_tile<GEMM_NT, CHAIN_BM, CHAIN_BN, true, true, PREC_F32>(op.g, 0, tile_m, tile_n, smem);
    else gemm_tile<GEMM_NN, CHAIN_BM, CHAIN_BN, true, true, PREC_F32>(op.g, 0, tile_m, tile_n, smem);
    // publish: every store of this tile acknowledged by the L2, then one increment of the panel's counter
    unsigned long long td = c.dbg ? wall_clock64() : 0ull;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(done + op.done_base + tile_m, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (c.dbg) { const unsigned long long te = wall_clock64(); d_n += 1; d_deq += tb - ta; d_wait += tc - tb; d_tile += td - tc; d_pub += te - td; }
  }
  if (c.dbg && tid == 0) {
    unsigned long long* o = c.dbg + 8 * blockIdx.x;
    o[0] = d_n; o[1] = d_deq; o[2] = d_wait; o[3] = d_tile; o[4] = d_pub; o[5] = wall_clock64() - d_t0; o[6] = (unsigned long long)x;
  }
  // the last workgroup out checks that every queue was drained (an XCD without a workgroup would leave its tiles undone)
  // and leaves the control block zeroed for the next launch
  __syncthreads();
  if (tid == 0) sh_item = __hip_atomic_fetch_add(c.ctl + c.nxcd, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
  __syncthreads();
  if (sh_item) {
    if (tid < c.nxcd) {
      int total = 0;
      for (int o = 0; o < c.n_ops; ++o) total += chain_items_of(c.op[o], tid, c.nxcd);
      if (__hip_atomic_load(heads + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)total) atomicOr(c.fault, CHAIN_FAULT_UNSERVED);
    }
    __syncthreads();
    for (int i = tid; i < c.ctl_words; i += GEMM_THREADS) __hip_atomic_store(c.ctl + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// XCC_ID of every workgroup of a small launch: the chain launch is only used on parts whose workgroups report the
// ids 0 .. nxcd-1 (engine.hip: chain_topology_ok)
__global__ void chain_census_kernel(unsigned int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = xcc_id_reg() & 0xfu;
}

}  // namespace gt
