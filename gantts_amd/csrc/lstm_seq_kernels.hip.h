// Persistent recurrence kernels of LSTMRNN / GRURNN / In2OutRNNHighwayNet (reference gantts/models.py:170-213:
// pack_padded_sequence -> nn.LSTM -> pad_packed_sequence) for gfx950: ONE launch walks all T time steps of one
// layer (both directions, every batch tile) instead of one launch per step (lstm_kernels.hip.h, kept as the
// fallback for shapes that do not fit and as the A/B reference).
//
// Decomposition.  A *group* is one (direction, batch tile of BT = 16 or 8 sequences): an independent recurrence
// (the launcher picks BT = 8 when that gives every XCD a group of its own: the exchange is bound by the L2 of the
// XCD a group lives on, and 8-sequence tiles halve its volume at the price of half-empty MFMA rows).  Its gate
// columns are split over `ncu` workgroups (one per CU).  Every workgroup keeps ITS slice of W_hh in registers for
// the whole launch (MFMA B operand, v_mfma_f32_16x16x4_f32: M = 16 sequences), keeps the cell state / running
// dL/dc of its hidden units in registers, and per time step
//   forward : gathers h_{t-1} of the whole group (16 x H) -> LDS, multiplies by its W_hh slice, applies the gate
//             non-linearities for its UPC hidden units, stashes gates / c / h, publishes its slice of h_t;
//   backward: gathers dG_{t+1} (16 x 4H) -> LDS, dh_t = dOut_t + dG_{t+1} . W_hh for its 16 hidden units,
//             gate derivatives, writes dG_t (over the X-projection storage) and publishes it.  The 4x wider payload
//             goes out untagged (16-byte {dgi, dgf, dgg, dgo} per (unit, sequence)) behind one flag per producing
//             wave (G16, R1: payload stores, drain, flag; consumer: flag, then L1-bypassing 16-byte loads).
// The exchange between workgroups uses the placement-independent granule protocol of the CDNA4 guide (G16, R2):
// the data IS the flag -- 8-byte {value, tag = step + 1} granules written by ONE relaxed agent-scope (sc1,
// write-through) store and swept with relaxed agent-scope loads (L1 bypass) until every tag matches; two buffers
// alternate by step parity (a producer can only be one step ahead of the slowest consumer, because publishing step
// s+1 needs all of step s).  All workgroups must be co-resident (the launcher checks the grid against the device's
// capacity); every wait is bounded by a wall-clock timeout that raises a fault word instead of hanging the GPU.
//
// XCD-local fast path.  The write-through hand-off costs ~3 us per step (store to the fabric, reload from it).  The
// workgroup -> group map therefore puts the workgroups of one group on ONE XCD under the observed dispatch rule
// (workgroup b runs on XCD b % 8), and at kernel start every group CHECKS it: each workgroup publishes its XCC_ID
// through the placement-independent protocol, all members read all ids.  Only if they agree does the group switch
// its stores to workgroup-scope ones (sc0: the line stays in that XCD's L2, which is the coherence point of all its
// CUs; the sweeps keep bypassing L1), a hand-off at L2-hit latency.  A group that is NOT co-located keeps the
// agent-scope protocol: placement changes speed, never results.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hip.h"
#include "lstm_kernels.hip.h"
#include "fast_math.hip.h"

namespace gt {

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned int gu32;

enum { LSTM_FAULT_TIMEOUT_FWD = 1, LSTM_FAULT_TIMEOUT_BWD = 2 };
// Timing ablations, set by tools/lstm_sched_bench only (results are then wrong on purpose): bit 0 no stash stores,
// bit 1 gate non-linearities replaced by one multiply, bit 2 no stash / X-projection loads inside the time loop.
#ifndef GT_LSTM_SEQ_ABLATE
#define GT_LSTM_SEQ_ABLATE 0
#endif
constexpr int LSTM_ABL = GT_LSTM_SEQ_ABLATE;

struct LstmSeqArgs {
  int B, T, H, dirs;
  int nbt;                        // batch tiles of 16 sequences
  int ncu;                        // workgroups per group
  const int* lengths;             // [B] device
  const float* Whh[2];            // (4H, H) row-major, per direction
  const float* bih[2];
  const float* bhh[2];
  float* xproj;                   // [N][dirs*4H]  forward: X W_ih^T (in); backward: dG (out), same storage
  float* gates;                   // [N][dirs*4H]  post-activation i,f,g,o
  float* cst;                     // [N][dirs*H]
  float* out;                     // [N][dirs*H]   h_t (zero beyond the length)
  const float* dout;              // [N][dirs*H]   upstream gradient (backward)
  unsigned long long* xch;        // granules: [group][2][16 * KX]  (KX = HP forward, 4*HP backward), zeroed per launch
  unsigned long long* xcc_chk;    // co-location check: [group][256] granules {XCC_ID, tag 1}, zeroed per launch (ncu <= 256)
  int nxcd;                       // XCDs the launcher spread the groups over (grid = nxcd * ncu * ceil(ngroups / nxcd))
  int allow_xcd_local;            // 0: always the agent-scope protocol
  unsigned int* dbg_protocol;     // optional [ngroups]: 1 = the group ran XCD-local, 2 = agent scope
  unsigned int* fault;            // device fault word (0 = ok)
  unsigned long long timeout_ticks;   // wall_clock64 ticks (100 MHz) a single wait may take
};

__device__ __forceinline__ unsigned long long xch_load(const unsigned long long* p) {
  return __hip_atomic_load((const gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void xch_store(unsigned long long* p, float v, unsigned tag, bool xcd_local) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
  if (xcd_local) __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else           __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned xcc_id_of_wave() { return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xfu; }   // HW_REG_XCC_ID
__device__ __forceinline__ unsigned fault_load(const unsigned int* p) {
  return __hip_atomic_load((const gu32*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool FM> __device__ __forceinline__ float gate_sigmoid(float x) { return FM ? fast_sigmoid(x) : sigmoidf_(x); }
template <bool FM> __device__ __forceinline__ float gate_tanh(float x) { return FM ? fast_tanh(x) : tanhf(x); }

// LDS image of the MFMA A operand (16 rows x K): element (k, m) lives where ONE ds_read_b128 of lane
// (m = lane & 15, kq = lane >> 4) returns the operands of four consecutive 16x16x4 MFMAs (k = 16*kb + 4*j + kq).
__device__ __forceinline__ int a_img_idx(int k, int m) { return (((k >> 4) * 64 + (k & 3) * 16 + m) << 2) + ((k >> 2) & 3); }

// wave-uniform bookkeeping of a spin loop: true = give up (a peer raised the fault word, or this wait timed out)
__device__ __forceinline__ bool spin_expired(const LstmSeqArgs& a, unsigned& spins, unsigned long long& t_start, unsigned fault_code) {
  if ((++spins & 63u) != 0u) return false;
  const unsigned long long now = wall_clock64();
  if (t_start == 0) t_start = now;
  if (fault_load(a.fault) != 0u) return true;
  if (now - t_start > a.timeout_ticks) {
    if ((threadIdx.x & 63) == 0) atomicCAS(a.fault, 0u, fault_code);
    return true;
  }
  return false;
}

// Forward exchange, consumer side.  Granule (k, m) of `src` lives at k*BT + m; thread -> m = tid % BT and the NG
// columns k = tid / BT + (256 / BT) * j.  With 16 granules per thread, first ONE of them is polled (a different one per
// workgroup) until it carries `epoch` -- the producers publish within a fraction of a microsecond of each other, and
// polling the whole image from every workgroup is what saturates the XCD's L2 -- then whole sweeps, started at a
// workgroup-dependent column so that the workgroups of a group do not walk the L2 channels in lock step.  Values go
// to the LDS A image.
// bf16 variant of the A image (PREC_BF16: the recurrent products run on v_mfma_f32_16x16x32_bf16, 8 consecutive k per
// lane): row-major [m][kp] bf16 with kp = K + 8 -- the 16 rows of a ds_read_b128 lane group start in 16 different
// 16-byte slots of the 256-byte bank row (row pitch = 16 bytes mod 256 for K a multiple of 128).
__device__ __forceinline__ int a_imgh_idx(int k, int m, int kp) { return m * kp + k; }

template <int NG, int BT, int PREC, int KPH>
__device__ __forceinline__ bool sweep_to_lds(const unsigned long long* __restrict__ src, unsigned epoch, int kvalid, float* sA,
                                             const LstmSeqArgs& a, int rot, unsigned fault_code) {
  constexpr int KS_ = 256 / BT;        // column stride between a thread's granules
  const int tid = threadIdx.x;
  const int m = tid % BT, k0 = tid / BT;
  unsigned spins = 0;
  unsigned long long t_start = 0;
  if (NG > 8) {      // measured (tools/lstm_seq_bench): the extra round trip pays only for the 16-granule sweeps
    const int ks = k0 + KS_ * (rot % NG);
    if (ks < kvalid) {
      const unsigned long long* sp = src + (size_t)ks * BT + m;
      for (;;) {
        const bool ok = (unsigned)(xch_load(sp) >> 32) == epoch;
        if (__all(ok)) break;
        if (spin_expired(a, spins, t_start, fault_code)) return false;
        __builtin_amdgcn_s_sleep(1);
      }
    }
  }
  unsigned long long v[NG];
  int kk[NG];
#pragma unroll
  for (int j = 0; j < NG; ++j) { const int jr = j + rot; kk[j] = k0 + KS_ * (jr % NG); }
  for (;;) {
    bool ok = true;
#pragma unroll
    for (int j = 0; j < NG; ++j) v[j] = xch_load(src + (size_t)(kk[j] < kvalid ? kk[j] : 0) * BT + m);
#pragma unroll
    for (int j = 0; j < NG; ++j) ok &= (kk[j] >= kvalid) || ((unsigned)(v[j] >> 32) == epoch);
    if (__all(ok)) break;
    if (spin_expired(a, spins, t_start, fault_code)) return false;
    __builtin_amdgcn_s_sleep(1);
  }
#pragma unroll
  for (int j = 0; j < NG; ++j)
    if (kk[j] < kvalid) {
      const float val = __uint_as_float((unsigned)v[j]);
      if (PREC == PREC_BF16) reinterpret_cast<__bf16*>(sA)[a_imgh_idx(kk[j], m, KPH)] = (__bf16)val;
      else sA[a_img_idx(kk[j], m)] = val;
    }
  return true;
}

// workgroup -> (group, cu): the workgroups of one group share bid % nxcd (one XCD under the observed dispatch rule)
__device__ __forceinline__ bool seq_group_of(const LstmSeqArgs& a, int* group, int* cu) {
  const int xs = blockIdx.x % a.nxcd, idx = blockIdx.x / a.nxcd;
  *group = xs + a.nxcd * (idx / a.ncu);
  *cu = idx % a.ncu;
  return *group < a.dirs * a.nbt;
}
// true iff every workgroup of this group runs on the same XCD (decided identically by all of them)
__device__ __forceinline__ bool seq_colocated(const LstmSeqArgs& a, int group, int cu, float* scratch, unsigned fault_code) {
  if (!a.allow_xcd_local) return false;
  unsigned long long* chk = a.xcc_chk + (size_t)group * 256;
  const unsigned mine = xcc_id_of_wave();
  if (threadIdx.x == 0) xch_store(chk + cu, __uint_as_float(mine), 1u, false);
  int* flag = reinterpret_cast<int*>(scratch);
  if (threadIdx.x < 64) {
    bool same = true, alive = true;
    unsigned spins = 0;
    unsigned long long t_start = 0;
    for (int i = threadIdx.x; i < a.ncu && alive; i += 64) {
      unsigned long long v;
      for (;;) {
        v = xch_load(chk + i);
        if ((unsigned)(v >> 32) == 1u) break;
        if ((++spins & 63u) == 0u) {
          const unsigned long long now = wall_clock64();
          if (t_start == 0) t_start = now;
          if (fault_load(a.fault) != 0u || now - t_start > a.timeout_ticks) { atomicCAS(a.fault, 0u, fault_code); alive = false; break; }
        }
        __builtin_amdgcn_s_sleep(2);
      }
      same &= alive && (unsigned)v == mine;
    }
    const bool all_same = __all(same && alive);
    if (threadIdx.x == 0) flag[0] = all_same ? 1 : 0;
  }
  __syncthreads();
  const bool r = flag[0] != 0;
  __syncthreads();
  return r;
}

template <int HP, int UPC> constexpr size_t lstm_fwd_seq_lds() {
  return (size_t)(HP * 16 + (16 / UPC) * 16 * 4 * UPC + 2 * 4 * 256) * sizeof(float);     // A image + KS x 16 x NC partials + the loader waves' hand-off
}
template <int HP> constexpr size_t lstm_bwd_seq_lds() { return (size_t)(4 * HP * 16 + 4 * 16 * 16 + 2 * 6 * 256) * sizeof(float); }   // + the loader waves' stash hand-off
// exchange area of one group, in 8-byte units (the launcher sizes and zeroes it): forward 2 granule images;
// backward 2 images of 16-byte chunks + 128 flag words
constexpr size_t lstm_fwd_xch_u64(int HP) { return (size_t)2 * 16 * HP; }
constexpr size_t lstm_bwd_xch_u64(int HP) { return (size_t)2 * 16 * HP * 4 + 64; }   // sized for the tagged f32 form: 32 B per (unit, sequence)

// ------------------------------------------------------------------------------------------
// forward.  grid = nxcd * ncu * ceil(ngroups / nxcd) workgroups of 256 (seq_group_of); the workgroup
// owns hidden units [cu*UPC, cu*UPC + UPC), i.e. NC = 4*UPC gate columns c = gate*UPC + unit.
// wave w: N tile w % NT of 16 columns, K part w / NT of HP/KS rows (NT = NC/16, KS = 4/NT).
// ------------------------------------------------------------------------------------------
// EARLY: the next step's X-projection is requested right behind the sweep's barrier instead of behind this step's
// stores (a wave's vector-memory results come back in issue order: requested late, the HBM-latency loads sit in front
// of the next sweep's polls in the gate wave's queue).  FM: the fast gate functions above.  Measured on a cfg3 layer
// (tools/lstm_sched_bench, us per time step, bf16 / f32): 2.15 / 2.53 -> EARLY 1.94 / 2.07 -> EARLY + FM 1.81 / 1.99.
// Also measured and NOT kept: gate threads as extra waves, requests 2-4 steps ahead, sweeps by the
// non-gate waves only, delayed stash stores.
// LW: the request is issued by extra loader waves (block = 256 + 64 * ceil(BT*UPC/64), loader thread i serves gate thread
// i) behind the sweep's barrier and handed over through LDS, as in the backward kernel below.
constexpr int lstm_fwd_block(int UPC, int BT, bool LW) { return LW ? 256 + 64 * ((BT * UPC + 63) / 64) : 256; }
template <int HP, int UPC, int BT, int PREC = PREC_F32, bool EARLY = false, bool FM = false, bool LW = false>
__global__ __launch_bounds__(lstm_fwd_block(UPC, BT, LW)) void lstm_fwd_seq_kernel(const LstmSeqArgs a) {
  static_assert(!(LW && EARLY), "one request order");
  constexpr int NTH = lstm_fwd_block(UPC, BT, LW);
  constexpr int NC = 4 * UPC, NT = NC / 16, KS = 4 / NT, KW = HP / KS, WR = KW / 4, NG = HP * BT / 256;
  constexpr int KPH = HP + 8, WRH = KW / 32;     // PREC_BF16: image pitch, MFMAs (= 8-bf16 weight fragments) per wave
  static_assert(UPC == 4 || UPC == 8 || UPC == 16, "UPC");
  static_assert(BT == 8 || BT == 16, "BT");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;                       // [HP x 16] permuted
  float* red = sm + HP * 16;            // [KS][16][NC]
  float* sxi = red + KS * 16 * NC;      // LW: [2][4][256] X-projection of the next step, by gate thread
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const bool loader = LW && tid >= 256;
  int group, cu;
  if (!seq_group_of(a, &group, &cu)) return;
  const int d = group % a.dirs, bt = group / a.dirs;
  const int H = a.H, T = a.T, B = a.B;
  const int u0 = cu * UPC;
  const int kvalid = a.ncu * UPC;       // hidden units that have a producer (>= H)
  const int ld4 = a.dirs * 4 * H, ld1 = a.dirs * H;

  // ---- this wave's W_hh slice -> registers (B operand: lane holds W[k = .. + kq][n = lane & 15])
  const int tile = wave % NT, kpart = wave / NT;
  const int n = lane & 15, kq = lane >> 4;
  float wreg[PREC == PREC_BF16 ? 1 : WR];
  bf16x8 wregh[PREC == PREC_BF16 ? WRH : 1];
  if (!loader) {
    const int c = tile * 16 + n, gate = c / UPC, uu = c % UPC;
    const bool wok = u0 + uu < H;
    const float* Wrow = a.Whh[d] + (long)(gate * H + min(u0 + uu, H - 1)) * H;
    if (PREC == PREC_BF16) {       // lane: column n, k = kpart*KW + 32*i + 8*kq + 0..7
#pragma unroll
      for (int i = 0; i < WRH; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = kpart * KW + 32 * i + 8 * kq + q;
          wregh[i][q] = (__bf16)((wok && k < H) ? Wrow[min(k, H - 1)] : 0.f);
        }
    } else {
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const int k = kpart * KW + 16 * (i >> 2) + 4 * (i & 3) + kq;
        wreg[i] = (wok && k < H) ? Wrow[min(k, H - 1)] : 0.f;
      }
    }
  }
  // ---- gate stage: thread (sequence gb, unit gu) for tid < BT*UPC
  const bool gthread = tid < BT * UPC;
  const int ct = loader ? tid - 256 : tid;        // the (sequence, unit) pair a gate thread owns / a loader thread serves
  const int gb = ct % BT, gu = (ct / BT) % UPC;
  const int bg = bt * BT + gb, bgc = min(bg, B - 1);
  const int j = u0 + gu, jc = min(j, H - 1);
  const bool store_ok = gthread && bg < B && j < H;
  const int len = (gthread && bg < B) ? a.lengths[bgc] : 0;
  float bias[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) bias[g] = a.bih[d][g * H + jc] + a.bhh[d][g * H + jc];
  float c_state = 0.f;
  unsigned long long* xb = a.xch + (size_t)group * lstm_fwd_xch_u64(HP);

  const bool xcd_local = seq_colocated(a, group, cu, red, LSTM_FAULT_TIMEOUT_FWD);
  if (fault_load(a.fault) != 0u) return;
  if (a.dbg_protocol && cu == 0 && tid == 0) a.dbg_protocol[group] = xcd_local ? 1u : 2u;
  for (int i = tid; i < HP * 16; i += NTH) sA[i] = 0.f;     // rows >= BT and columns without a producer stay zero for good
  __syncthreads();

  auto row_of = [&](int s) { return (long)bgc * T + (d == 0 ? s : T - 1 - s); };
  if (loader) {       // two barriers per step s >= 1, as everybody else
    for (int s = 0; s < T; ++s) {
      if (s > 0) __syncthreads();
      const bool have = s + 1 < T && ct < BT * UPC && !(LSTM_ABL & 4);
      float z[4] = {0.f, 0.f, 0.f, 0.f};
      if (have) {
        const long row = row_of(s + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) z[g] = a.xproj[row * ld4 + d * 4 * H + g * H + jc];
      }
      if (s > 0) __syncthreads();
      if (have) {
        float* q = sxi + ((s + 1) & 1) * 4 * 256 + ct;
#pragma unroll
        for (int g = 0; g < 4; ++g) q[g * 256] = z[g];
      }
    }
    return;
  }
  float xin[4];
  {
    const long row = row_of(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) xin[g] = a.xproj[row * ld4 + d * 4 * H + g * H + jc];
  }
  float xnext[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? s : T - 1 - s;
    if (s > 0) {
      if (!sweep_to_lds<NG, BT, PREC, KPH>(xb + (size_t)((s - 1) & 1) * (BT * HP), (unsigned)s, kvalid, sA, a, cu, LSTM_FAULT_TIMEOUT_FWD)) return;
      __syncthreads();
    }
    if (EARLY && gthread && s + 1 < T && !(LSTM_ABL & 4)) {
      const long row = row_of(s + 1);
#pragma unroll
      for (int g = 0; g < 4; ++g) xnext[g] = a.xproj[row * ld4 + d * 4 * H + g * H + jc];
    }
    if (s > 0) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (PREC == PREC_BF16) {
        const __bf16* ah = reinterpret_cast<const __bf16*>(sA) + n * KPH + kpart * KW + 8 * kq;     // row m = lane & 15
#pragma unroll
        for (int i = 0; i < WRH; ++i)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(ah + 32 * i), wregh[i], acc, 0, 0, 0);
      } else {
        const float* ap = sA + (((kpart * (KW / 16)) * 64 + kq * 16 + n) << 2);
#pragma unroll
        for (int kb = 0; kb < KW / 16; ++kb) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + kb * 256);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], wreg[kb * 4 + q], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(kpart * 16 + kq * 4 + r) * NC + tile * 16 + n] = acc[r];   // C: row = kq*4 + r, col = n
      __syncthreads();
    }
    if (gthread) {
      if (LW && s > 0 && !(LSTM_ABL & 4)) {          // written by the loader waves before this step's first barrier
        const float* q = sxi + (s & 1) * 4 * 256 + tid;
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = q[g * 256];
      }
      float pre[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float sum = 0.f;
        if (s > 0) {
#pragma unroll
          for (int kp = 0; kp < KS; ++kp) sum += red[(kp * 16 + gb) * NC + g * UPC + gu];
        }
        pre[g] = sum + (xin[g] + bias[g]);
      }
      const bool active = t < len && j < H;         // padding units of the last workgroup publish exact zeros
      float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, h = 0.f;
      if (active && (LSTM_ABL & 2)) {
        ig = 0.1f * pre[0]; fg = 0.1f * pre[1]; gg = 0.1f * pre[2]; og = 0.1f * pre[3];
        c = fg * c_state + ig * gg;
        h = og * c;
      } else if (active) {
        ig = gate_sigmoid<FM>(pre[0]); fg = gate_sigmoid<FM>(pre[1]); gg = gate_tanh<FM>(pre[2]); og = gate_sigmoid<FM>(pre[3]);
        c = fg * c_state + ig * gg;
        h = og * gate_tanh<FM>(c);
      }
      c_state = c;                       // state is held at zero while inactive
      if (s + 1 < T) xch_store(xb + (size_t)(s & 1) * (BT * HP) + (size_t)j * BT + gb, h, (unsigned)(s + 1), xcd_local);
      if (store_ok && !(LSTM_ABL & 1)) {
        const long row = (long)bg * T + t;
        *reinterpret_cast<f32x4*>(a.gates + lstm_gate_idx(row, ld4, d, H, j)) = f32x4{ig, fg, gg, og};
        a.cst[row * ld1 + d * H + j] = c;
        a.out[row * ld1 + d * H + j] = h;
      }
      if (EARLY) {
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = xnext[g];
      } else if (!LW && s + 1 < T && !(LSTM_ABL & 4)) {     // next step's X-projection: in flight while the group exchanges h_t
        const long row = row_of(s + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) xin[g] = a.xproj[row * ld4 + d * 4 * H + g * H + jc];
      }
    }
  }
}

// 16-byte loads that bypass L1 (sc1), N at a time, completed inside the statement (hipcc does not count asm loads)
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void load8_sc1(const u32x4* const (&p)[8], u32x4 (&v)[8]) {
  asm volatile(
      "global_load_dwordx4 %0, %8, off sc1\n\t"
      "global_load_dwordx4 %1, %9, off sc1\n\t"
      "global_load_dwordx4 %2, %10, off sc1\n\t"
      "global_load_dwordx4 %3, %11, off sc1\n\t"
      "global_load_dwordx4 %4, %12, off sc1\n\t"
      "global_load_dwordx4 %5, %13, off sc1\n\t"
      "global_load_dwordx4 %6, %14, off sc1\n\t"
      "global_load_dwordx4 %7, %15, off sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3]), "=&v"(v[4]), "=&v"(v[5]), "=&v"(v[6]), "=&v"(v[7])
      : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]), "v"(p[4]), "v"(p[5]), "v"(p[6]), "v"(p[7])
      : "memory");
}
__device__ __forceinline__ void flag_store(unsigned int* p, unsigned v, bool xcd_local) {
  if (xcd_local) __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else           __hip_atomic_store((gu32*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void payload_store(float* p, float v, bool xcd_local) {
  flag_store(reinterpret_cast<unsigned int*>(p), __float_as_uint(v), xcd_local);
}

// ------------------------------------------------------------------------------------------
// backward.  grid as forward with ncu = ceil(H/16): the workgroup owns 16 hidden units (one MFMA N tile) of dh;
// K = 4*HP gate columns (k = gate*HP + unit), wave w multiplies gate w's block.  Direction 0 walks t = T-1..0,
// direction 1 walks t = 0..T-1.  Thread (sequence tid % BT, unit tid / BT < 16) owns one (b, u) pair.
// Exchange image of one step: 16-byte chunk (u, b) = {dgi, dgf, dgg, dgo} at u*BT + b; flag word of producing wave
// w of workgroup c at c*NW + w, monotonic (= steps published).
// ------------------------------------------------------------------------------------------
// LW (loader waves): the stash of step s+1 is requested by EXTRA waves (block = 256 + 16*BT threads, loader thread i
// serves gate thread i) right behind step s's sweep barrier and handed over through LDS.  The vector L1 returns a CU's
// loads in issue order: requested at the end of a step (round 2), the HBM-latency stash loads sit in front of the next
// step's flag polls of EVERY wave of the CU (measured: 0.5 us of a 2.4 us step, whichever wave issues them, however far
// ahead); requested behind the sweep they come back while the CU has nothing urgent in flight.  The gate waves cannot issue
// them there themselves: their drain in front of the flag store would wait for them.
// TAG: the exchange in self-validating form, as in the forward kernel -- every 8-byte half of a 16-byte chunk is a
// {data, tag = step + 1} granule written by one store: PREC_BF16 one chunk per (unit, sequence) {bf16 dgi | dgf << 16, tag,
// bf16 dgg | dgo << 16, tag} (the consumer rounds dG to bf16 on its way into LDS anyway: rounding at the producer gives the
// same operand), PREC_F32 two chunks {dgi, tag, dgf, tag} {dgg, tag, dgo, tag}.  No flag, no drain in front of a flag, no
// flag poll in front of the payload loads: one sweep of 16-byte loads, repeated until every tag matches.
constexpr int lstm_bwd_block(int BT, bool LW) { return LW ? 256 + 16 * BT : 256; }
__device__ __forceinline__ void xch_store_u32(unsigned long long* p, unsigned v, unsigned tag, bool xcd_local) {
  const unsigned long long g = ((unsigned long long)tag << 32) | (unsigned long long)v;
  if (xcd_local) __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else           __hip_atomic_store((gu64*)p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned bf16_bits(float x) { return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)x); }
template <int HP, int BT, int PREC = PREC_F32, bool LW = false, bool TAG = false>
__global__ __launch_bounds__(lstm_bwd_block(BT, LW)) void lstm_bwd_seq_kernel(const LstmSeqArgs a) {
  constexpr int NTH = lstm_bwd_block(BT, LW);
  constexpr int CPC = PREC == PREC_BF16 ? 1 : 2;   // TAG: 16-byte chunks per (unit, sequence)
  constexpr int NCT = HP * BT * CPC / 256;         // TAG: chunks per thread and step
  constexpr int WR = HP / 4;
  constexpr int KPH = 4 * HP + 8, WRH = HP / 32;  // PREC_BF16: pitch of the [16][4*HP] bf16 image, MFMAs per wave
  constexpr int NW = 16 * BT / 64;                // producing waves per workgroup
  constexpr int NCH = HP * BT / 256;              // 16-byte chunks per thread and step
  static_assert(BT == 8 || BT == 16, "BT");
  static_assert(NCH % 8 == 0, "chunks are loaded 8 at a time");
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;                       // [4*HP x 16] permuted
  float* red = sm + 4 * HP * 16;        // [4][16][16]
  float* sst = red + 4 * 16 * 16;       // LW: [2][6][256] stash of the next step, by gate thread
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  const bool loader = LW && tid >= 256;
  int group, cu;
  if (!seq_group_of(a, &group, &cu)) return;
  const int d = group % a.dirs, bt = group / a.dirs;
  const int H = a.H, T = a.T, B = a.B;
  const int u0 = cu * 16;
  const int uvalid = a.ncu * 16;        // units that have a producer
  const int ld4 = a.dirs * 4 * H, ld1 = a.dirs * H;
  const int n = lane & 15, kq = lane >> 4;

  // B operand: W_hh[(gate = wave)*H + k][u0 + n], k = 16*(i>>2) + 4*(i&3) + kq
  float wreg[PREC == PREC_BF16 ? 1 : WR];
  bf16x8 wregh[PREC == PREC_BF16 ? WRH : 1];
  if (!loader) {
    const bool wok = u0 + n < H;
    const float* Wc = a.Whh[d] + (long)wave * H * H + min(u0 + n, H - 1);
    if (PREC == PREC_BF16) {       // lane: output unit n, k = 32*i + 8*kq + 0..7 (rows of gate `wave`'s block of W_hh)
#pragma unroll
      for (int i = 0; i < WRH; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int k = 32 * i + 8 * kq + q;
          wregh[i][q] = (__bf16)((wok && k < H) ? Wc[(long)min(k, H - 1) * H] : 0.f);
        }
    } else {
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        const int k = 16 * (i >> 2) + 4 * (i & 3) + kq;
        wreg[i] = (wok && k < H) ? Wc[(long)min(k, H - 1) * H] : 0.f;
      }
    }
  }
  const bool gthread = tid < 16 * BT;
  const int ct = loader ? tid - 256 : tid;        // the (sequence, unit) pair a gate thread owns / a loader thread serves
  const int gb = ct % BT, gu = (ct / BT) & 15;
  const int bg = bt * BT + gb, bgc = min(bg, B - 1);
  const int j = u0 + gu, jc = min(j, H - 1);
  const bool store_ok = gthread && bg < B && j < H;
  const int len = (gthread && bg < B) ? a.lengths[bgc] : 0;
  float dcs = 0.f;
  unsigned long long* xb = a.xch + (size_t)group * lstm_bwd_xch_u64(HP);
  u32x4* xdata = reinterpret_cast<u32x4*>(xb);                                   // [2][HP*BT] chunks
  unsigned int* xflag = reinterpret_cast<unsigned int*>(xb + (size_t)2 * 16 * HP * 2);   // [ncu * NW] <= 128
  const int nflags = a.ncu * NW;

  const bool xcd_local = seq_colocated(a, group, cu, red, LSTM_FAULT_TIMEOUT_BWD);
  if (fault_load(a.fault) != 0u) return;
  if (a.dbg_protocol && cu == 0 && tid == 0) a.dbg_protocol[group] = xcd_local ? 1u : 2u;
  for (int i = tid; i < 4 * HP * 16; i += NTH) sA[i] = 0.f;
  __syncthreads();

  // stash of one (b, u, t): requested one step ahead of its use.  Three requests per step: the 16-byte gate record, the
  // upstream gradient, and the cell state ENTERING the frame -- which is the cell state OF the frame the walk visits
  // next, so the state of the frame itself is carried over from the previous step instead of being read again.
  struct Stash { float dout, ig, fg, gg, og, cp; };
  auto load_stash = [&](int s) {
    Stash z;
    const int t = d == 0 ? T - 1 - s : s;
    const long row = (long)bgc * T + t;
    z.dout = a.dout[row * ld1 + d * H + jc];
    const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gates + lstm_gate_idx(row, ld4, d, H, jc));
    z.ig = gv[0]; z.fg = gv[1]; z.gg = gv[2]; z.og = gv[3];
    const long rowp = d == 0 ? (t > 0 ? row - 1 : row) : (t + 1 < T ? row + 1 : row);   // clamped; validity checked at use
    z.cp = a.cst[rowp * ld1 + d * H + jc];
    return z;
  };
  if (loader) {       // two barriers per step s >= 1, as everybody else
    for (int s = 0; s < T; ++s) {
      if (s > 0) __syncthreads();
      const bool have = s + 1 < T && !(LSTM_ABL & 4);
      Stash z = {};
      if (have) z = load_stash(s + 1);
      if (s > 0) __syncthreads();
      if (have) {
        float* q = sst + ((s + 1) & 1) * 6 * 256 + ct;
        q[0 * 256] = z.dout; q[1 * 256] = z.ig; q[2 * 256] = z.fg; q[3 * 256] = z.gg; q[4 * 256] = z.og; q[5 * 256] = z.cp;
      }
    }
    return;
  }
  Stash st = load_stash(0);
  float c_here = a.cst[((long)bgc * T + (d == 0 ? T - 1 : 0)) * ld1 + d * H + jc];     // cell state of the first frame of the walk
  for (int s = 0; s < T; ++s) {
    const int t = d == 0 ? T - 1 - s : s;
    if (s > 0 && TAG) {
      const u32x4* src = xdata + (size_t)((s - 1) & 1) * (HP * BT * CPC);
      unsigned spins = 0;
      unsigned long long t_start = 0;
#pragma unroll
      for (int c0 = 0; c0 < NCT; c0 += 8) {
        const u32x4* p[8];
        u32x4 v[8];
        int cc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int jr = c0 + q + cu;                      // workgroup-dependent start: spread the L2 channels
          cc[q] = tid + 256 * (jr % NCT);
          const int u = cc[q] / (BT * CPC);
          p[q] = src + (u < uvalid ? cc[q] : 0);
        }
        for (;;) {
          load8_sc1(p, v);
          bool ok = true;
#pragma unroll
          for (int q = 0; q < 8; ++q) ok &= cc[q] / (BT * CPC) >= uvalid || (v[q][1] == (unsigned)s && v[q][3] == (unsigned)s);
          if (__all(ok)) break;
          if (spin_expired(a, spins, t_start, LSTM_FAULT_TIMEOUT_BWD)) return;
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int cell = cc[q] / CPC, part = cc[q] % CPC, u = cell / BT, b = cell % BT;
          if (u < uvalid) {
            if (PREC == PREC_BF16) {
              unsigned short* ah = reinterpret_cast<unsigned short*>(sA) + b * KPH + u;
              ah[0 * HP] = (unsigned short)(v[q][0] & 0xffffu); ah[1 * HP] = (unsigned short)(v[q][0] >> 16);
              ah[2 * HP] = (unsigned short)(v[q][2] & 0xffffu); ah[3 * HP] = (unsigned short)(v[q][2] >> 16);
            } else {
              const int ai = a_img_idx(u, b);
              sA[(2 * part) * HP * 16 + ai] = __uint_as_float(v[q][0]);
              sA[(2 * part + 1) * HP * 16 + ai] = __uint_as_float(v[q][2]);
            }
          }
        }
      }
    } else if (s > 0) {
      {   // every wave waits for every producing wave of the group to have published step s-1
        unsigned spins = 0;
        unsigned long long t_start = 0;
        for (;;) {
          bool ok = true;
          for (int i = lane; i < nflags; i += 64) ok &= fault_load(xflag + i) >= (unsigned)s;
          if (__all(ok)) break;
          if (spin_expired(a, spins, t_start, LSTM_FAULT_TIMEOUT_BWD)) return;
          __builtin_amdgcn_s_sleep(1);
        }
      }
      const u32x4* src = xdata + (size_t)((s - 1) & 1) * (HP * BT);
#pragma unroll
      for (int c0 = 0; c0 < NCH; c0 += 8) {
        const u32x4* p[8];
        u32x4 v[8];
        int cc[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int jr = c0 + q + cu;                      // workgroup-dependent start: spread the L2 channels
          cc[q] = tid + 256 * (jr % NCH);
          const int u = cc[q] / BT;
          p[q] = src + (u < uvalid ? cc[q] : 0);
        }
        load8_sc1(p, v);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int u = cc[q] / BT, b = cc[q] % BT;
          if (u < uvalid) {
            if (PREC == PREC_BF16) {
              __bf16* ah = reinterpret_cast<__bf16*>(sA) + b * KPH + u;
#pragma unroll
              for (int g = 0; g < 4; ++g) ah[g * HP] = (__bf16)__uint_as_float(v[q][g]);
            } else {
              const int ai = a_img_idx(u, b);
#pragma unroll
              for (int g = 0; g < 4; ++g) sA[g * HP * 16 + ai] = __uint_as_float(v[q][g]);
            }
          }
        }
      }
    }
    if (s > 0) {
      __syncthreads();
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
      if (PREC == PREC_BF16) {
        const __bf16* ah = reinterpret_cast<const __bf16*>(sA) + n * KPH + wave * HP + 8 * kq;
#pragma unroll
        for (int i = 0; i < WRH; ++i)
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(ah + 32 * i), wregh[i], acc, 0, 0, 0);
      } else {
        const float* ap = sA + wave * HP * 16 + ((kq * 16 + n) << 2);
#pragma unroll
        for (int kb = 0; kb < HP / 16; ++kb) {
          const f32x4 av = *reinterpret_cast<const f32x4*>(ap + kb * 256);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[q], wreg[kb * 4 + q], acc, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) red[(wave * 16 + kq * 4 + r) * 16 + n] = acc[r];
      __syncthreads();
    }
    if (gthread) {
      if (LW && s > 0 && !(LSTM_ABL & 4)) {          // written by the loader waves before this step's first barrier
        const float* q = sst + (s & 1) * 6 * 256 + tid;
        st.dout = q[0 * 256]; st.ig = q[1 * 256]; st.fg = q[2 * 256]; st.gg = q[3 * 256]; st.og = q[4 * 256]; st.cp = q[5 * 256];
      }
      const bool active = t < len && j < H;
      float dgi = 0.f, dgf = 0.f, dgg = 0.f, dgo = 0.f, dcn = 0.f;
      if (active) {
        float dh = st.dout;
        if (s > 0) dh += (red[(0 * 16 + gb) * 16 + gu] + red[(1 * 16 + gb) * 16 + gu]) + (red[(2 * 16 + gb) * 16 + gu] + red[(3 * 16 + gb) * 16 + gu]);
        float cp = 0.f;                                        // cell state entering this frame
        if (d == 0) { if (t > 0) cp = st.cp; }
        else        { if (t + 1 < len) cp = st.cp; }
        const float tc = (LSTM_ABL & 2) ? 0.1f * c_here : tanhf(c_here);
        const float dc = dcs + dh * st.og * (1.f - tc * tc);
        dgo = dh * tc * (st.og * (1.f - st.og));
        dgi = dc * st.gg * (st.ig * (1.f - st.ig));
        dgf = dc * cp * (st.fg * (1.f - st.fg));
        dgg = dc * st.ig * (1.f - st.gg * st.gg);
        dcn = dc * st.fg;
      }
      dcs = dcn;
      if (TAG && s + 1 < T) {
        unsigned long long* dst = xb + ((size_t)(s & 1) * (HP * BT * CPC) + ((size_t)j * BT + gb) * CPC) * 2;
        if (PREC == PREC_BF16) {
          xch_store_u32(dst + 0, bf16_bits(dgi) | (bf16_bits(dgf) << 16), (unsigned)(s + 1), xcd_local);
          xch_store_u32(dst + 1, bf16_bits(dgg) | (bf16_bits(dgo) << 16), (unsigned)(s + 1), xcd_local);
        } else {
          xch_store(dst + 0, dgi, (unsigned)(s + 1), xcd_local);
          xch_store(dst + 1, dgf, (unsigned)(s + 1), xcd_local);
          xch_store(dst + 2, dgg, (unsigned)(s + 1), xcd_local);
          xch_store(dst + 3, dgo, (unsigned)(s + 1), xcd_local);
        }
      } else if (s + 1 < T) {
        float* dst = reinterpret_cast<float*>(xdata + (size_t)(s & 1) * (HP * BT) + (size_t)j * BT + gb);
        payload_store(dst + 0, dgi, xcd_local);
        payload_store(dst + 1, dgf, xcd_local);
        payload_store(dst + 2, dgg, xcd_local);
        payload_store(dst + 3, dgo, xcd_local);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's payload is in L2 (or written through) ...
        if (lane == 0) flag_store(xflag + cu * NW + wave, (unsigned)(s + 1), xcd_local);   // ... before its flag says so
      }
      if (store_ok && !(LSTM_ABL & 1)) {
        const long row = (long)bg * T + t;
        a.xproj[row * ld4 + d * 4 * H + 0 * H + j] = dgi;      // dG overwrites the X-projection storage
        a.xproj[row * ld4 + d * 4 * H + 1 * H + j] = dgf;
        a.xproj[row * ld4 + d * 4 * H + 2 * H + j] = dgg;
        a.xproj[row * ld4 + d * 4 * H + 3 * H + j] = dgo;
      }
      c_here = st.cp;                    // (at the ends of the walk rowp is clamped: the value is then not used)
      if (!LW && s + 1 < T && !(LSTM_ABL & 4)) st = load_stash(s + 1);
    }
  }
}

}  // namespace gt
