// Row-panel-resident MLP chains for the discriminator (reference gantts/models.py:121-141 forward,
// autograd backward-data; called from train.py:245-320).
//
// The per-layer GEMM launches of a narrow MLP (hidden 256, K = 256: 8 K-tiles per workgroup) spend
// ~40 % of their time in prologue / epilogue / launch ramp, and every layer boundary is a round trip
// of the (rows x H) activation through HBM with all workgroups bursting their stores at once.  Here a
// workgroup owns TWO 32-row PANELS (64 consecutive rows) for the WHOLE chain of layers:
//
//   * the panels' activation (64 x H fp32) lives in LDS in the k-major image the next layer's MFMA A
//     operand is read from (As[k][m], pitch 65: conflict-free for the C-layout scatter of the epilogue
//     and for the operand fetch);
//   * only the weights stream: global/L2 -> registers (one 32-k chunk ahead) -> a double-buffered LDS
//     stage shared by all 8 waves, ONE barrier per chunk of 32 MFMAs per wave;
//   * each layer's result leaves for HBM (the backward pass needs it) as row-wise 16 B stores that
//     overlap the NEXT layer's MFMAs -- no epilogue burst, no prologue, one launch per chain;
//   * the epilogue's inputs (Philox keep bits, stored activations for f') are produced in the shadow
//     of the layer's first MFMA chunks.
//
// 8 waves per workgroup: wave = 4 * panel + column group; a wave owns 32 rows x H/4 columns
// (TN_ = H/128 tiles of v_mfma_f32_32x32x2_f32).  LDS at H = 256: 65 KB panels + 2 x 33 KB weight stage.
//
// Forward  (chain_fwd_kernel):  stage 0: Z = P[r % p_mod] + A0[r, 0:K0] . W0[:, 0:K0]^T   (P = the part of
//            the first layer that is shared by the real and the fake half of the discriminator batch:
//            x . W1[:, :Din]^T + b1, computed once per step by the big GEMM kernel);
//            stage s >= 1: Z = H_{s-1} . W_s^T + b_s;  H_s = dropout(leaky(Z)), stored.
// Backward (chain_bwd_kernel):  dZ_{s-1} = (dZ_s . W_s) (.) f'(H_{s-1}), stored when the weight gradient
//            needs it (D step), not stored in the G step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "gemm_f32.hip.h"

namespace gt {

constexpr int CH_PANELS = 2;                 // 32-row panels per workgroup
constexpr int CH_ROWS = 32 * CH_PANELS;      // rows per workgroup
constexpr int CH_KC = 32;                    // weight chunk depth (k rows per LDS stage)
constexpr int CH_AP = CH_ROWS + 1;           // panel pitch
constexpr int CH_THREADS = 256 * CH_PANELS;
constexpr int CH_MAXS = 6;                   // stages per launch

struct ChainStage {
  const float* W; int ldw;     // fwd: W[n*ldw + k];  bwd: W[k*ldw + n]   (k = contraction index); 16-byte aligned rows
  const float* bias;           // fwd: [H] or null
  float* out; int ldo;         // stage result (rows x H, 16-byte aligned, ldo % 4 == 0), or null
  const float* Hact; int ldh;  // bwd: stored forward activation whose f' multiplies the result
  int act;                     // Act
  DropoutSpec drop;
};

struct ChainArgs {
  long rows; int H;
  int K0, K0p;                 // fwd: columns of A0; stage 0 contracts over K0p >= max(K0, 2*CH_KC), a multiple of CH_KC
                               // (W0 zero-padded); stages >= 1 (and all bwd stages) contract over H
  const float* A0; int lda0;   // fwd: stage-0 A operand (rows x K0);  bwd: dZ entering the chain (rows x H)
  const float* P; int ldp; long p_mod;   // fwd: stage-0 accumulator init P[r % p_mod][n] (bias included), or null
  int n_stages;
  ChainStage st[CH_MAXS];
  long long* dbg;              // CH_DEBUG_TIMING builds only: per-workgroup clock stamps (tools/chain_bench.hip)
};

static inline int chain_k0p(int K0) { const int k = ((K0 + CH_KC - 1) / CH_KC) * CH_KC; return k < 2 * CH_KC ? 2 * CH_KC : k; }
template <bool FWD> constexpr int chain_bp(int H) { return FWD ? H + 1 : H + 4; }   // weight-stage pitch
static inline size_t chain_lds_bytes(int H, int K0p, bool fwd) {
  const int kmax = H > K0p ? H : K0p;
  return ((size_t)kmax * CH_AP + (size_t)2 * CH_KC * (fwd ? chain_bp<true>(H) : chain_bp<false>(H))) * sizeof(float);
}

// one weight chunk (CH_KC k-rows x H columns): global -> registers (16 B per lane, whole 128-byte lines per
// 8 lanes) -> LDS stage Bs[k][n].  FWD: W is n-major (k contiguous): scattered ds_write_b32, conflict-free with
// the odd pitch H+1;  BWD: W is k-major (n contiguous): ds_write_b128 rows, pitch H+4.
template <int TN_, bool FWD>
struct ChainWLoad {
  static constexpr int H = TN_ * 128, BP = chain_bp<FWD>(H);
  static constexpr int NU = CH_KC * H / 4 / CH_THREADS;   // 16-byte units per thread
  f32x4 r[NU];
  __device__ __forceinline__ void load(const float* __restrict__ W, uint32_t ldw, int k0, int tid) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const uint32_t e = (uint32_t)tid + (uint32_t)u * CH_THREADS;
      uint32_t off;
      if (FWD) { const uint32_t kq = e & 7u, n = e >> 3; off = n * ldw + (uint32_t)k0 + kq * 4u; }
      else     { const uint32_t nq = e % (H / 4), kk = e / (H / 4); off = ((uint32_t)k0 + kk) * ldw + nq * 4u; }
      r[u] = *reinterpret_cast<const f32x4*>(W + off);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ Bs, int tid) const {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int e = tid + u * CH_THREADS;
      if (FWD) {
        const int kq = e & 7, n = e >> 3;
#pragma unroll
        for (int c = 0; c < 4; ++c) Bs[(kq * 4 + c) * BP + n] = r[u][c];
      } else {
        const int nq = e % (H / 4), kk = e / (H / 4);
        *reinterpret_cast<f32x4*>(Bs + kk * BP + nq * 4) = r[u];
      }
    }
  }
};

// Epilogue inputs of one stage, produced in the SHADOW of the stage's first four MFMA chunks instead of
// after them: the dropout keep bits (Philox is ~100 VALU instructions per call -- issued between MFMAs they
// are free, issued in the epilogue every wave of the workgroup pays them with an idle matrix pipe) and, for
// the backward chain, the stored forward activations at the accumulator positions (global loads whose
// latency would otherwise be exposed).
template <int TN_, bool WANT_H>
struct ChainSide {
  uint32_t keep[TN_];             // bit r of keep[j] <-> acc[j][r]   (all ones when there is no Philox dropout)
  float h[WANT_H ? TN_ : 1][16];  // forward activation at (row of acc index r, column of tile j)
  bool philox; uint32_t key0, key1, thresh;
  uint32_t grp;                   // (m0 >> 4): 16-row group of the panel's first row
  uint32_t ncol0;                 // column of tile 0 for this lane
  uint32_t half;
  const float* Hp; uint32_t ldh, nvalid;   // panel base of the activation, pitch, valid rows
  __device__ __forceinline__ void init() {
#pragma unroll
    for (int j = 0; j < TN_; ++j) keep[j] = 0xffffffffu;
  }
  template <int C>
  __device__ __forceinline__ void run() {
    constexpr int NCALL = 2 * TN_, CPC = (NCALL + 3) / 4;
#pragma unroll
    for (int i = 0; i < CPC; ++i) {
      constexpr int dummy = 0; (void)dummy;
      const int idx = C * CPC + i;
      if (idx < NCALL && philox) {
        const int j = idx >> 1, c2 = idx & 1;
        uint32_t r[4];
        philox4x32_10(2u * (grp + (uint32_t)c2) + half, ncol0 + (uint32_t)j * 32u, key0, key1, r);
        uint32_t bits = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) bits |= (philox_piece(r, p) >= thresh ? 1u : 0u) << p;
        keep[j] = c2 == 0 ? ((keep[j] & ~0xffu) | bits) : ((keep[j] & ~0xff00u) | (bits << 8));
        if (c2 == 1) keep[j] &= 0xffffu;
      }
    }
    if (WANT_H) {
#pragma unroll
      for (int e = 0; e < 4 * TN_; ++e) {
        const int idx = C * 4 * TN_ + e, j = idx / 16, rr = idx % 16;
        uint32_t row = (uint32_t)(8 * (rr >> 2) + (rr & 3)) + 4u * half;
        row = row < nvalid ? row : nvalid - 1u;
        h[WANT_H ? j : 0][rr] = Hp[row * ldh + ncol0 + (uint32_t)j * 32u];
      }
    }
  }
};

// acc += panel(32 x K) . W, K % CH_KC == 0, K >= 2 * CH_KC.  Software pipeline per chunk c: the registers
// holding chunk c+1 go to the OTHER LDS buffer, the global loads of chunk c+2 are issued, then the 16 k-pairs of
// chunk c run from the current buffer with the operand fragments of group g+1 requested before the MFMAs of
// group g (4 k-pairs per group); one barrier closes the chunk.  The caller's barrier orders the panel writes
// before this call; the final chunk's barrier orders every wave's panel reads before the caller overwrites it.
template <int TN_, bool FWD, typename Side>
__device__ __forceinline__ void chain_gemm(f32x16 (&acc)[TN_], const float* __restrict__ As, float* __restrict__ Bs,
                                           const float* __restrict__ W, int ldw, int K, int tid, int pw, int cw, int l31, int half,
                                           Side& side) {
  constexpr int H = TN_ * 128, BP = chain_bp<FWD>(H), CW = H / 4, NG = CH_KC / 8;   // NG groups of 4 k-pairs
  const int nchunks = K / CH_KC;
  ChainWLoad<TN_, FWD> wl;
  wl.load(W, (uint32_t)ldw, 0, tid);
  wl.store(Bs, tid);
  wl.load(W, (uint32_t)ldw, CH_KC, tid);      // nchunks >= 2
  __syncthreads();
  auto chunk = [&](int c, auto CC) {
    constexpr int C = decltype(CC)::value;     // 0 / 1: side work slices {0,1} / {2,3} ride in this chunk; -1: none
    const float* bcur = Bs + (c & 1) * CH_KC * BP + half * BP + cw * CW + l31;
    float* bnxt = Bs + ((c & 1) ^ 1) * CH_KC * BP;
    const float* as = As + (c * CH_KC + half) * CH_AP + pw * 32 + l31;
    float fa[2][4], fb[2][4][TN_];
    auto frags = [&](int g, int slot) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fa[slot][t] = as[2 * (4 * g + t) * CH_AP];
#pragma unroll
        for (int j = 0; j < TN_; ++j) fb[slot][t][j] = bcur[2 * (4 * g + t) * BP + j * 32];
      }
    };
    frags(0, 0);                               // first: what the first MFMAs wait for
    __builtin_amdgcn_sched_barrier(0);
#ifndef CH_ABL_NOSTORE
    if (c + 1 < nchunks) wl.store(bnxt, tid);  // then the traffic nobody waits for in this chunk
#endif
#ifndef CH_ABL_NOLOAD
    if (c + 2 < nchunks) wl.load(W, (uint32_t)ldw, (c + 2) * CH_KC, tid);
#endif
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (g + 1 < NG) frags(g + 1, (g + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < TN_; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g & 1][t], fb[g & 1][t][j], acc[j], 0, 0, 0);
      if constexpr (C >= 0) {
        if (g == 0) side.template run<2 * C>();
        if (g == 2) side.template run<2 * C + 1>();
        if (g == 0 || g == 2) {
          // interleave the side work with this group's and the next group's MFMAs
#pragma unroll
          for (int q = 0; q < 4 * TN_; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 14, 0);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef CH_ABL_NOBAR
    __syncthreads();
#endif
  };
  chunk(0, std::integral_constant<int, 0>{});
  chunk(1, std::integral_constant<int, 1>{});
  for (int c = 2; c < nchunks; ++c) chunk(c, std::integral_constant<int, -1>{});
}

// panels (k-major LDS image, columns [0, H)) -> global rows, 16 B per lane
template <int TN_>
__device__ __forceinline__ void chain_store_panel(const float* __restrict__ As, float* __restrict__ out, int ldo, long m0, long rows,
                                                  int tid) {
  constexpr int H = TN_ * 128, Q = H / 4;
#pragma unroll
  for (int i = 0; i < CH_ROWS * Q / CH_THREADS; ++i) {
    const int idx = tid + i * CH_THREADS;
    const int row = idx / Q, c4 = (idx % Q) * 4;
    f32x4 v;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = As[(c4 + c) * CH_AP + row];
    if (m0 + row < rows) *reinterpret_cast<f32x4*>(out + (m0 + row) * ldo + c4) = v;
  }
}

template <int TN_>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_fwd_kernel(const ChainArgs a) {
  constexpr int H = TN_ * 128, CW = H / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K0p = a.K0p;
  const int kmax = H > K0p ? H : K0p;
  float* As = smem;
  float* Bs = smem + (size_t)kmax * CH_AP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int pw = wave >> 2, cw = wave & 3;
  const long m0 = (long)blockIdx.x * CH_ROWS;                 // first row of the workgroup
  const long pm0 = m0 + 32 * pw;                              // first row of this wave's panel
  const uint32_t wvalid = (uint32_t)(a.rows - m0 < CH_ROWS ? a.rows - m0 : CH_ROWS);   // valid rows of the workgroup (>= 1)

  // stage-0 A operand: 64 rows x K0 -> As[k][m], zero beyond K0, rows clamped
  for (int e = tid; e < CH_ROWS * K0p; e += CH_THREADS) {
    const int k = e % K0p, m = e / K0p;
    const uint32_t mm = (uint32_t)m < wvalid ? (uint32_t)m : wvalid - 1u;
    As[k * CH_AP + m] = k < a.K0 ? (a.A0 + m0 * a.lda0)[mm * (uint32_t)a.lda0 + (uint32_t)k] : 0.f;
  }
  // rows of this wave's panel that exist (0 if the panel lies entirely beyond the matrix: its results are never stored)
  const uint32_t nvalid = pm0 >= a.rows ? 0u : (uint32_t)(a.rows - pm0 < 32 ? a.rows - pm0 : 32);

#ifdef CH_DEBUG_TIMING
  int dbg_i = 0;
#define CH_STAMP() do { if (a.dbg && tid == 0) a.dbg[blockIdx.x * 16 + dbg_i] = clock64(); ++dbg_i; } while (0)
#else
#define CH_STAMP() do {} while (0)
#endif
  CH_STAMP();
  f32x16 acc[TN_];
  const long pbase = a.P ? pm0 % a.p_mod : 0;
#pragma unroll
  for (int j = 0; j < TN_; ++j) {
    const int n = cw * CW + j * 32 + l31;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v = 0.f;
        if (a.P) {
          long r = pbase + 8 * q + 4 * half + s;      // (pm0 + row) % p_mod without a per-element division
          r = r >= a.p_mod ? r - a.p_mod : r;
          r = (uint32_t)(8 * q + 4 * half + s) < nvalid ? r : 0;
          v = a.P[(uint32_t)r * (uint32_t)a.ldp + (uint32_t)n];       // p_mod * ldp < 2^31 (launcher)
        }
        acc[j][q * 4 + s] = v;
      }
  }
  const uint32_t nclamp = nvalid ? nvalid : 1u;

  for (int st = 0; st < a.n_stages; ++st) {
    const ChainStage& S = a.st[st];
    ChainSide<TN_, false> side;
    side.init();
    side.philox = S.act == ACT_LEAKY_DROPOUT && S.drop.mode == DROP_PHILOX;
    side.key0 = S.drop.key0; side.key1 = S.drop.key1; side.thresh = S.drop.thresh;
    side.grp = (uint32_t)(pm0 >> 4); side.ncol0 = (uint32_t)(cw * CW + l31); side.half = (uint32_t)half;
    side.Hp = nullptr; side.ldh = 0; side.nvalid = nclamp;
    __syncthreads();      // the panel (stage 0: A0 image; later: the previous layer's output) is complete
    CH_STAMP();
    chain_gemm<TN_, true>(acc, As, Bs, S.W, S.ldw, st == 0 ? K0p : H, tid, pw, cw, l31, half, side);
    CH_STAMP();
    // epilogue: bias, LeakyReLU, dropout factor; mode checks are hoisted out of the element loops (one uniform
    // branch per stage instead of three per element)
    const bool use_bias = S.bias != nullptr && !(st == 0 && a.P);
    const float slope = S.act == ACT_LEAKY_DROPOUT ? 0.01f : 1.f;
    const float scale = S.act == ACT_LEAKY_DROPOUT && S.drop.mode != DROP_NONE ? S.drop.scale : 1.f;
    float* wpanel = As + pw * 32 + 4 * half;
    if (S.act == ACT_LEAKY_DROPOUT && S.drop.mode == DROP_BUFFER) {
      const float* mk = S.drop.mask + (nvalid ? pm0 : 0) * S.drop.ld_mask;
      const uint32_t ldm = (uint32_t)S.drop.ld_mask;
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = cw * CW + j * 32 + l31;
        const float bias = use_bias ? S.bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = (uint32_t)(8 * (r >> 2) + (r & 3)) + 4u * (uint32_t)half;
          const uint32_t rr = row < nclamp ? row : nclamp - 1u;
          float v = acc[j][r] + bias;
          v = v > 0.f ? v : v * slope;
          wpanel[n * CH_AP + 8 * (r >> 2) + (r & 3)] = mk[rr * ldm + (uint32_t)n] != 0.f ? v * scale : 0.f;
          acc[j][r] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = cw * CW + j * 32 + l31;
        const float bias = use_bias ? S.bias[n] : 0.f;
        const uint32_t kb = side.keep[j];       // all ones unless Philox dropout
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[j][r] + bias;
          v = v > 0.f ? v : v * slope;
          wpanel[n * CH_AP + 8 * (r >> 2) + (r & 3)] = ((kb >> r) & 1u) ? v * scale : 0.f;
          acc[j][r] = 0.f;
        }
      }
    }
    __syncthreads();
    CH_STAMP();
#ifndef CH_ABL_NOOUT
    if (S.out) chain_store_panel<TN_>(As, S.out, S.ldo, m0, a.rows, tid);
#endif
    CH_STAMP();
  }
}

template <int TN_>
__global__ __launch_bounds__(CH_THREADS, 2) void chain_bwd_kernel(const ChainArgs a) {
  constexpr int H = TN_ * 128, CW = H / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + (size_t)H * CH_AP;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int pw = wave >> 2, cw = wave & 3;
  const long m0 = (long)blockIdx.x * CH_ROWS;
  const long pm0 = m0 + 32 * pw;
  const uint32_t wvalid = (uint32_t)(a.rows - m0 < CH_ROWS ? a.rows - m0 : CH_ROWS);
  const uint32_t nvalid = pm0 >= a.rows ? 0u : (uint32_t)(a.rows - pm0 < 32 ? a.rows - pm0 : 32);
  const uint32_t nclamp = nvalid ? nvalid : 1u;
  const long pr = nvalid ? pm0 : 0;        // a panel beyond the matrix reads row 0 (results never stored)

  // dZ entering the chain: 64 rows x H, 16 B per lane, -> As[n][m]
  for (int idx = tid; idx < CH_ROWS * (H / 4); idx += CH_THREADS) {
    const int row = idx / (H / 4), c4 = (idx % (H / 4)) * 4;
    const uint32_t rr = (uint32_t)row < wvalid ? (uint32_t)row : wvalid - 1u;
    const f32x4 v = *reinterpret_cast<const f32x4*>((a.A0 + m0 * a.lda0) + rr * (uint32_t)a.lda0 + (uint32_t)c4);
#pragma unroll
    for (int c = 0; c < 4; ++c) As[(c4 + c) * CH_AP + row] = v[c];
  }

#ifdef CH_DEBUG_TIMING
  int dbg_i = 0;
#define CH_STAMP() do { if (a.dbg && tid == 0) a.dbg[blockIdx.x * 16 + dbg_i] = clock64(); ++dbg_i; } while (0)
#else
#define CH_STAMP() do {} while (0)
#endif
  CH_STAMP();
  f32x16 acc[TN_];
#pragma unroll
  for (int j = 0; j < TN_; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  for (int st = 0; st < a.n_stages; ++st) {
    const ChainStage& S = a.st[st];
    ChainSide<TN_, true> side;
    side.init();
    side.philox = S.act == ACT_LEAKY_DROPOUT && S.drop.mode == DROP_PHILOX;
    side.key0 = S.drop.key0; side.key1 = S.drop.key1; side.thresh = S.drop.thresh;
    side.grp = (uint32_t)(pm0 >> 4); side.ncol0 = (uint32_t)(cw * CW + l31); side.half = (uint32_t)half;
    side.Hp = S.Hact + pr * S.ldh; side.ldh = (uint32_t)S.ldh; side.nvalid = nclamp;
    __syncthreads();
    CH_STAMP();
    chain_gemm<TN_, false>(acc, As, Bs, S.W, S.ldw, H, tid, pw, cw, l31, half, side);
    CH_STAMP();
    // epilogue: multiply by f'(H) = keep ? (h > 0 ? scale : 0.01 * scale) : 0
    const bool has_act = S.act == ACT_LEAKY_DROPOUT;
    const float scale = has_act && S.drop.mode != DROP_NONE ? S.drop.scale : 1.f;
    float* wpanel = As + pw * 32 + 4 * half;
    if (has_act && S.drop.mode == DROP_BUFFER) {
      const float* mk = S.drop.mask + pr * S.drop.ld_mask;
      const uint32_t ldm = (uint32_t)S.drop.ld_mask;
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = cw * CW + j * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t row = (uint32_t)(8 * (r >> 2) + (r & 3)) + 4u * (uint32_t)half;
          const uint32_t rr = row < nclamp ? row : nclamp - 1u;
          const float f = mk[rr * ldm + (uint32_t)n] != 0.f ? (side.h[j][r] > 0.f ? scale : 0.01f * scale) : 0.f;
          wpanel[n * CH_AP + 8 * (r >> 2) + (r & 3)] = acc[j][r] * f;
          acc[j][r] = 0.f;
        }
      }
    } else {
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = cw * CW + j * 32 + l31;
        const uint32_t kb = side.keep[j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float f = 1.f;
          if (has_act) f = ((kb >> r) & 1u) ? (side.h[j][r] > 0.f ? scale : 0.01f * scale) : 0.f;
          wpanel[n * CH_AP + 8 * (r >> 2) + (r & 3)] = acc[j][r] * f;
          acc[j][r] = 0.f;
        }
      }
    }
    __syncthreads();
    CH_STAMP();
#ifndef CH_ABL_NOOUT
    if (S.out) chain_store_panel<TN_>(As, S.out, S.ldo, m0, a.rows, tid);
#endif
    CH_STAMP();
  }
}

}  // namespace gt
