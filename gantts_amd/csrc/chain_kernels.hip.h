// Row-panel-resident MLP chains for the discriminator (reference gantts/models.py:121-141 forward,
// autograd backward-data; called from train.py:245-320).
//
// The per-layer GEMM launches of a narrow MLP (hidden 256, K = 256: 8 K-tiles per workgroup) spend
// ~40 % of their time in prologue / epilogue / launch ramp, and every layer boundary is a round trip
// of the (rows x H) activation through HBM with all workgroups bursting their stores at once.  Here a
// workgroup owns a PANEL of 32 consecutive rows for the WHOLE chain of layers:
//
//   * the panel's activation (32 x H fp32) lives in LDS in the k-major image the next layer's MFMA A
//     operand is read from (As[k][m], pitch 33: conflict-free for the C-layout scatter of the epilogue
//     and for the operand fetch);
//   * only the weights stream, global/L2 -> registers, straight into the MFMA B operand one 16-k chunk ahead
//     (every workgroup streams every W: <= 256 KB per layer, L2-resident) -- no barrier inside a layer;
//   * each layer's result leaves for HBM (it is needed by the backward pass) as row-wise 16 B stores
//     that overlap the NEXT layer's MFMAs -- no epilogue burst, no prologue, one launch per chain.
//
// 4 waves per workgroup, wave w owns output columns [w*H/4, (w+1)*H/4): 32 x (H/4) accumulators as
// TN_ = H/128 tiles of v_mfma_f32_32x32x2_f32.  LDS = H*33*4 bytes (34 KB at H = 256), 3 workgroups =
// 12 waves per CU (register-limited).
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

constexpr int CH_ROWS = 32;            // panel rows
constexpr int CH_KC = 16;              // weight chunk depth (k rows per LDS stage)
constexpr int CH_AP = CH_ROWS + 1;     // panel pitch
constexpr int CH_THREADS = 256;
constexpr int CH_MAXS = 6;             // stages per launch

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
  int K0, K0p;                 // fwd: columns of A0; stage 0 contracts over K0p >= max(K0, 4*CH_KC), a multiple of CH_KC
                               // (W0 zero-padded); stages >= 1 (and all bwd stages) contract over H
  const float* A0; int lda0;   // fwd: stage-0 A operand (rows x K0);  bwd: dZ entering the chain (rows x H)
  const float* P; int ldp; long p_mod;   // fwd: stage-0 accumulator init P[r % p_mod][n] (bias included), or null
  int n_stages;
  ChainStage st[CH_MAXS];
};

static inline int chain_k0p(int K0) { const int k = ((K0 + CH_KC - 1) / CH_KC) * CH_KC; return k < 4 * CH_KC ? 4 * CH_KC : k; }
static inline size_t chain_lds_bytes(int H, int K0p) {
  const int kmax = H > K0p ? H : K0p;
  return (size_t)kmax * CH_AP * sizeof(float);
}

// The MFMA B operand (weights) goes global/L2 -> registers directly, one chunk (CH_KC k-values) ahead: no LDS
// stage and NO barrier inside a layer's K loop -- the four waves of a workgroup only meet at layer boundaries.
// A 32x32x2 MFMA contracts over two k values, lanes 0-31 supplying one and lanes 32-63 the other; WHICH two
// is free as long as A and B agree, so within a chunk MFMA t (0..7) uses k = 8*(t>>2) + 4*half + (t&3):
// a lane's four consecutive MFMAs then read four CONSECUTIVE k -> one 16-byte load per lane for the forward
// orientation (W[n][k], k contiguous), no element fetched twice.
template <int TN_, bool FWD>
struct ChainWFrag {
  float b[TN_][CH_KC / 2];       // b[j][t]: B fragment of MFMA t, column tile j
  __device__ __forceinline__ void load(const float* __restrict__ W, uint32_t ldw, uint32_t k0, uint32_t ncol0, uint32_t half) {
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const uint32_t n = ncol0 + (uint32_t)j * 32u;
      if (FWD) {
#pragma unroll
        for (int i = 0; i < CH_KC / 8; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(W + n * ldw + k0 + 8u * i + 4u * half);
#pragma unroll
          for (int c = 0; c < 4; ++c) b[j][4 * i + c] = v[c];
        }
      } else {
#pragma unroll
        for (int t = 0; t < CH_KC / 2; ++t) b[j][t] = W[(k0 + 8u * (t >> 2) + 4u * half + (uint32_t)(t & 3)) * ldw + n];
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

// acc += panel(32 x K) . W, K % (2 * CH_KC) == 0, K >= 4 * CH_KC.  Barrier-free; the caller orders the panel
// writes before this call and the panel overwrite after it.
template <int TN_, bool FWD, typename Side>
__device__ __forceinline__ void chain_gemm(f32x16 (&acc)[TN_], const float* __restrict__ As, const float* __restrict__ W, int ldw,
                                           int K, int wave, int l31, int half, Side& side) {
  constexpr int H = TN_ * 128, CW = H / 4;
  const int nchunks = K / CH_KC;
  const uint32_t ncol0 = (uint32_t)(wave * CW + l31);
  ChainWFrag<TN_, FWD> w0, w1;
  w0.load(W, (uint32_t)ldw, 0u, ncol0, (uint32_t)half);
  auto chunk = [&](int c, const ChainWFrag<TN_, FWD>& cur, ChainWFrag<TN_, FWD>& nxt, auto CC) {
    constexpr int C = decltype(CC)::value;
#ifndef CH_ABL_NOLOAD
    if (c + 1 < nchunks) nxt.load(W, (uint32_t)ldw, (uint32_t)(c + 1) * CH_KC, ncol0, (uint32_t)half);
#endif
    const float* as = As + (c * CH_KC + 4 * half) * CH_AP + l31;
    float fa[CH_KC / 2];
#pragma unroll
    for (int t = 0; t < CH_KC / 2; ++t) fa[t] = as[(8 * (t >> 2) + (t & 3)) * CH_AP];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int t = 0; t < CH_KC / 2; ++t)
#pragma unroll
      for (int j = 0; j < TN_; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t], cur.b[j][t], acc[j], 0, 0, 0);
    if constexpr (C >= 0) {
      side.template run<C>();
      // interleave: after every MFMA a slice of the side work (VALU, and the activation loads of the backward chain)
#pragma unroll
      for (int g = 0; g < (CH_KC / 2) * TN_; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 12, 0);
      }
    }
  };
  chunk(0, w0, w1, std::integral_constant<int, 0>{});
  chunk(1, w1, w0, std::integral_constant<int, 1>{});
  chunk(2, w0, w1, std::integral_constant<int, 2>{});
  chunk(3, w1, w0, std::integral_constant<int, 3>{});
  for (int c = 4; c < nchunks; c += 2) {
    chunk(c, w0, w1, std::integral_constant<int, -1>{});
    chunk(c + 1, w1, w0, std::integral_constant<int, -1>{});
  }
}

// panel (k-major LDS image, columns [0, H)) -> global rows, 16 B per lane
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
__global__ __launch_bounds__(CH_THREADS, 3) void chain_fwd_kernel(const ChainArgs a) {
  constexpr int H = TN_ * 128, CW = H / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int K0p = a.K0p;
  const int kmax = H > K0p ? H : K0p;
  float* As = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const long m0 = (long)blockIdx.x * CH_ROWS;
  const uint32_t nvalid = (uint32_t)(a.rows - m0 < CH_ROWS ? a.rows - m0 : CH_ROWS);   // valid rows of this panel (>= 1)

  // stage-0 A operand: 32 rows x K0 -> As[k][m], zero beyond K0, rows clamped
  for (int e = tid; e < CH_ROWS * K0p; e += CH_THREADS) {
    const int k = e % K0p, m = e / K0p;
    const uint32_t mm = (uint32_t)m < nvalid ? (uint32_t)m : nvalid - 1u;
    As[k * CH_AP + m] = k < a.K0 ? (a.A0 + m0 * a.lda0)[mm * (uint32_t)a.lda0 + (uint32_t)k] : 0.f;
  }

  f32x16 acc[TN_];
  const long pbase = a.P ? m0 % a.p_mod : 0;
#pragma unroll
  for (int j = 0; j < TN_; ++j) {
    const int n = wave * CW + j * 32 + l31;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        float v = 0.f;
        if (a.P) {
          long r = pbase + 8 * q + 4 * half + s;      // (m0 + row) % p_mod without a per-element division
          r = r >= a.p_mod ? r - a.p_mod : r;
          r = (uint32_t)(8 * q + 4 * half + s) < nvalid ? r : 0;
          v = a.P[(uint32_t)r * (uint32_t)a.ldp + (uint32_t)n];       // p_mod * ldp < 2^31 (launcher)
        }
        acc[j][q * 4 + s] = v;
      }
  }

  for (int st = 0; st < a.n_stages; ++st) {
    const ChainStage& S = a.st[st];
    ChainSide<TN_, false> side;
    side.init();
    side.philox = S.act == ACT_LEAKY_DROPOUT && S.drop.mode == DROP_PHILOX;
    side.key0 = S.drop.key0; side.key1 = S.drop.key1; side.thresh = S.drop.thresh;
    side.grp = (uint32_t)(m0 >> 4); side.ncol0 = (uint32_t)(wave * CW + l31); side.half = (uint32_t)half;
    side.Hp = nullptr; side.ldh = 0; side.nvalid = nvalid;
    __syncthreads();      // the panel (stage 0: A0 image; later: the previous layer's output) is complete
    chain_gemm<TN_, true>(acc, As, S.W, S.ldw, st == 0 ? K0p : H, wave, l31, half, side);
    __syncthreads();      // every wave is done reading the old panel
    const bool use_bias = S.bias != nullptr && !(st == 0 && a.P);
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int n = wave * CW + j * 32 + l31;
      const float bias = use_bias ? S.bias[n] : 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float v = acc[j][q * 4 + s] + bias;
          if (S.act == ACT_LEAKY_DROPOUT) {
            v = leaky(v);
            if (S.drop.mode == DROP_PHILOX) v = ((side.keep[j] >> (q * 4 + s)) & 1u) ? v * S.drop.scale : 0.f;
            else if (S.drop.mode == DROP_BUFFER) {
              const uint32_t rr = (uint32_t)(8 * q + 4 * half + s) < nvalid ? (uint32_t)(8 * q + 4 * half + s) : nvalid - 1u;
              v = (S.drop.mask + m0 * S.drop.ld_mask)[rr * (uint32_t)S.drop.ld_mask + (uint32_t)n] != 0.f ? v * S.drop.scale : 0.f;
            }
          } else if (S.act == ACT_SIGMOID) {
            v = 1.f / (1.f + expf(-v));
          }
          As[n * CH_AP + 8 * q + 4 * half + s] = v;
          acc[j][q * 4 + s] = 0.f;
        }
      }
    }
    __syncthreads();
#ifndef CH_ABL_NOOUT
    if (S.out) chain_store_panel<TN_>(As, S.out, S.ldo, m0, a.rows, tid);
#endif
  }
}

template <int TN_>
__global__ __launch_bounds__(CH_THREADS, 3) void chain_bwd_kernel(const ChainArgs a) {
  constexpr int H = TN_ * 128, CW = H / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const long m0 = (long)blockIdx.x * CH_ROWS;
  const uint32_t nvalid = (uint32_t)(a.rows - m0 < CH_ROWS ? a.rows - m0 : CH_ROWS);

  // dZ entering the chain: 32 rows x H, 16 B per lane, -> As[n][m]
  for (int idx = tid; idx < CH_ROWS * (H / 4); idx += CH_THREADS) {
    const int row = idx / (H / 4), c4 = (idx % (H / 4)) * 4;
    const uint32_t rr = (uint32_t)row < nvalid ? (uint32_t)row : nvalid - 1u;
    const f32x4 v = *reinterpret_cast<const f32x4*>((a.A0 + m0 * a.lda0) + rr * (uint32_t)a.lda0 + (uint32_t)c4);
#pragma unroll
    for (int c = 0; c < 4; ++c) As[(c4 + c) * CH_AP + row] = v[c];
  }

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
    side.grp = (uint32_t)(m0 >> 4); side.ncol0 = (uint32_t)(wave * CW + l31); side.half = (uint32_t)half;
    side.Hp = S.Hact + m0 * S.ldh; side.ldh = (uint32_t)S.ldh; side.nvalid = nvalid;
    __syncthreads();
    chain_gemm<TN_, false>(acc, As, S.W, S.ldw, H, wave, l31, half, side);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int n = wave * CW + j * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          float v = acc[j][q * 4 + s];
          const float h = side.h[j][q * 4 + s];
          if (S.act == ACT_LEAKY_DROPOUT) {
            bool keep = true;
            float scale = 1.f;
            if (S.drop.mode == DROP_PHILOX) { keep = ((side.keep[j] >> (q * 4 + s)) & 1u) != 0u; scale = S.drop.scale; }
            else if (S.drop.mode == DROP_BUFFER) {
              const uint32_t rr = (uint32_t)(8 * q + 4 * half + s) < nvalid ? (uint32_t)(8 * q + 4 * half + s) : nvalid - 1u;
              keep = (S.drop.mask + m0 * S.drop.ld_mask)[rr * (uint32_t)S.drop.ld_mask + (uint32_t)n] != 0.f; scale = S.drop.scale;
            }
            v *= leaky_drop_grad(h, keep, scale);
          } else if (S.act == ACT_SIGMOID) {
            v *= h * (1.f - h);
          }
          As[n * CH_AP + 8 * q + 4 * half + s] = v;
          acc[j][q * 4 + s] = 0.f;
        }
      }
    }
    __syncthreads();
#ifndef CH_ABL_NOOUT
    if (S.out) chain_store_panel<TN_>(As, S.out, S.ldo, m0, a.rows, tid);
#endif
  }
}

}  // namespace gt
