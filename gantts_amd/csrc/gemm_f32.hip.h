// fp32 MFMA GEMM family for the frame x weight products of the G/D networks (gfx950 only).
//
// One compute core (v_mfma_f32_32x32x2_f32, exact f32 == an fmaf chain), three operand
// orientations, runtime-selected fused epilogues:
//
//   NT  Y  = act(X . W^T + b)          forward of nn.Linear      (reference gantts/models.py:129-141)
//   NN  dX = (dZ . W) (.) f'(H)        backward-data, activation derivative of the producer fused
//   TN  dW = dZ^T . X  (split over the frame dimension, deterministic partial slabs)
//
// Tile: BM x BN per 256-thread workgroup (4 waves, 2x2), each wave (BM/2)x(BN/2) in 32x32 MFMA
// tiles; K step 32.  LDS image is k-major  As[k][m] / Bs[k][n]  with leading dimension BM+1
// (== 1 mod 32): the MFMA operand fetch `As[k][m0 + lane%32]` is a conflict-free ds_read_b32 and
// both loader orientations write conflict-free (k-contiguous sources scatter with bank = (k+m)%32,
// m-contiguous sources write consecutive banks).  Double-buffered LDS, next tile's global loads
// are issued before the MFMA block of the current tile (register staging), one barrier per K step.
// f32 MFMA is 64 cycles per instruction per SIMD, so the loop is matrix-pipe bound by a wide
// margin: per K step a wave issues 64 MFMAs (4096 cycles) against 64 ds_read_b32 + 32 global
// dword loads + 32 ds_write_b32.  Dword (4 B) global loads are used on purpose: the frame
// matrices have row strides of 425 / 483 / 187 / 63 floats, i.e. rows are not 16-byte aligned.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gt {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;

enum GemmKind { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum Act { ACT_NONE = 0, ACT_LEAKY_DROPOUT = 1, ACT_SIGMOID = 2 };
enum DropMode { DROP_NONE = 0, DROP_PHILOX = 1, DROP_BUFFER = 2 };

struct DropoutSpec {
  int mode;           // DropMode
  float p;            // drop probability
  float scale;        // 1/(1-p)
  uint32_t thresh;    // keep iff philox word >= thresh  (thresh = p * 2^32)
  uint32_t key0, key1;  // philox key: (seed, site/step)
  const float* mask;  // DROP_BUFFER: [rows][ld_mask] 0/1 floats
  int ld_mask;
};

// ---- Philox4x32-10, counter = (row>>2, col, 0, 0): word j decides row (row&~3)+j ------------
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
  uint32_t c2 = 0x243F6A88u, c3 = 0x85A308D3u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

__device__ __forceinline__ float leaky(float z) { return z > 0.f ? z : 0.01f * z; }

// derivative factor of h = dropout(leaky(z)) w.r.t. z, recovered from the stored h and keep bit.
// (in-place LeakyReLU in the reference: backward uses the output's sign, gantts/models.py:132)
__device__ __forceinline__ float leaky_drop_grad(float h, bool keep, float scale) {
  return keep ? (h > 0.f ? scale : 0.01f * scale) : 0.f;
}

struct GemmArgs {
  // C[M][N] = sum_k A(m,k) * B(k,n)
  const float* A; int lda;   // NT/NN: A[m*lda + k];  TN: A[k*lda + m]
  const float* B; int ldb;   // NT: B[n*ldb + k];     NN/TN: B[k*ldb + n]
  float* C; int ldc;         // TN: slab s at C + s*slab_stride
  int M, N, K;
  // epilogue
  const float* bias;         // NT: [N] or null
  int act;                   // Act
  const float* H; int ldh;   // NN: producer's stored activation (for f'), or null
  DropoutSpec drop;
  // TN split
  int k_chunk;               // rows of K per slab (multiple of GEMM_BK)
  long slab_stride;
  float* colsum_slab;        // TN: per-slab column sums of A (bias gradient), [nslab][M], or null
  int n_tiles_m, n_tiles_n;
};

template <int KIND, int BM, int BN>
__global__ __launch_bounds__(GEMM_THREADS, 2) void gemm_f32_kernel(const GemmArgs g) {
  constexpr int LDM = BM + 1, LDN = BN + 1;
  constexpr int WM = BM / 2, WN = BN / 2;   // wave tile
  constexpr int TM = WM / 32, TN_ = WN / 32;  // MFMA tiles per wave
  constexpr int A_PER_THR = BM * GEMM_BK / GEMM_THREADS;
  constexpr int B_PER_THR = BN * GEMM_BK / GEMM_THREADS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;                          // [2][BK][LDM]
  float* Bs = smem + 2 * GEMM_BK * LDM;      // [2][BK][LDN]

  // XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each
  // XCD a contiguous run of tiles (neighbouring tiles share the weight panel in that XCD's L2).
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tiles_mn = g.n_tiles_m * g.n_tiles_n;
  const int slab = bid / tiles_mn;
  const int t = bid - slab * tiles_mn;
  // n fastest: workgroups sharing an M panel (the big frame matrix) run back to back
  const int tile_m = t / g.n_tiles_n, tile_n = t - tile_m * g.n_tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  int k_begin = 0, k_end = g.K;
  if (KIND == GEMM_TN) {
    k_begin = slab * g.k_chunk;
    k_end = min(g.K, k_begin + g.k_chunk);
  }

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  float ra[A_PER_THR], rb[B_PER_THR];

  // Per-thread element offsets of this thread's A_PER_THR / B_PER_THR operand elements inside
  // the K-tile starting at k = 0, computed ONCE.  Rows/columns outside the matrix are clamped to
  // the last valid one (their products land in accumulator rows/columns the epilogue never
  // stores), so the steady-state loads are unconditional: no exec-mask branches, no per-load
  // address arithmetic (base pointer advances by one K-tile per iteration).  Only the K tail is
  // zero-filled (select after an in-bounds load).
  uint32_t offA[A_PER_THR], offB[B_PER_THR];
  // k index inside the tile of element i (k-contiguous: tid%32 for every i; otherwise tid/BM + i*(256/BM))
  auto kkA = [&](int i) { return KIND == GEMM_TN ? (tid + i * GEMM_THREADS) / BM : tid % GEMM_BK; };
  auto kkB = [&](int i) { return KIND == GEMM_NT ? tid % GEMM_BK : (tid + i * GEMM_THREADS) / BN; };
#pragma unroll
  for (int i = 0; i < A_PER_THR; ++i) {
    const int e = tid + i * GEMM_THREADS;
    if (KIND == GEMM_TN) {  // m-contiguous: A[k*lda + m]
      const int mm = e % BM, kk = e / BM;
      offA[i] = (uint32_t)kk * (uint32_t)g.lda + (uint32_t)min(m0 + mm, g.M - 1);
    } else {                // k-contiguous: A[m*lda + k]
      const int kk = e % GEMM_BK, mm = e / GEMM_BK;
      offA[i] = (uint32_t)min(m0 + mm, g.M - 1) * (uint32_t)g.lda + (uint32_t)kk;
    }
  }
#pragma unroll
  for (int i = 0; i < B_PER_THR; ++i) {
    const int e = tid + i * GEMM_THREADS;
    if (KIND == GEMM_NT) {  // k-contiguous: B[n*ldb + k]
      const int kk = e % GEMM_BK, nn = e / GEMM_BK;
      offB[i] = (uint32_t)min(n0 + nn, g.N - 1) * (uint32_t)g.ldb + (uint32_t)kk;
    } else {                // n-contiguous: B[k*ldb + n]
      const int nn = e % BN, kk = e / BN;
      offB[i] = (uint32_t)kk * (uint32_t)g.ldb + (uint32_t)min(n0 + nn, g.N - 1);
    }
  }
  const long stepA = (KIND == GEMM_TN) ? (long)GEMM_BK * g.lda : (long)GEMM_BK;
  const long stepB = (KIND == GEMM_NT) ? (long)GEMM_BK : (long)GEMM_BK * g.ldb;
  const float* pA = g.A + (KIND == GEMM_TN ? (long)k_begin * g.lda : (long)k_begin);
  const float* pB = g.B + (KIND == GEMM_NT ? (long)k_begin : (long)k_begin * g.ldb);

  auto load_tile = [&](int k0, bool tail) {
    if (!tail) {
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) ra[i] = pA[offA[i]];
#pragma unroll
      for (int i = 0; i < B_PER_THR; ++i) rb[i] = pB[offB[i]];
    } else {
      // K tail: clamp k to the last valid index (keeps the address in bounds), then zero
      const int krem = k_end - k0;   // valid k in this tile: [0, krem)
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) {
        const int back = max(0, kkA(i) - (krem - 1));
        const float v = pA[offA[i] - (uint32_t)back * (KIND == GEMM_TN ? (uint32_t)g.lda : 1u)];
        ra[i] = kkA(i) < krem ? v : 0.f;
      }
#pragma unroll
      for (int i = 0; i < B_PER_THR; ++i) {
        const int back = max(0, kkB(i) - (krem - 1));
        const float v = pB[offB[i] - (uint32_t)back * (KIND == GEMM_NT ? 1u : (uint32_t)g.ldb)];
        rb[i] = kkB(i) < krem ? v : 0.f;
      }
    }
    pA += stepA;
    pB += stepB;
  };
  // TN: column sums of the A operand (dZ) fall out of the loader for free -- element i of this
  // thread always has the same column m = tid % BM (GEMM_THREADS % BM == 0).
  float csum = 0.f;
  const bool want_csum = KIND == GEMM_TN && g.colsum_slab != nullptr && tile_n == 0;
  auto store_tile = [&](int buf) {
    if (KIND == GEMM_TN && want_csum) {
#pragma unroll
      for (int i = 0; i < A_PER_THR; ++i) csum += ra[i];
    }
    float* as = As + buf * GEMM_BK * LDM;
    float* bs = Bs + buf * GEMM_BK * LDN;
#pragma unroll
    for (int i = 0; i < A_PER_THR; ++i) {
      const int e = tid + i * GEMM_THREADS;
      int mm, kk;
      if (KIND == GEMM_TN) { mm = e % BM; kk = e / BM; } else { kk = e % GEMM_BK; mm = e / GEMM_BK; }
      as[kk * LDM + mm] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < B_PER_THR; ++i) {
      const int e = tid + i * GEMM_THREADS;
      int nn, kk;
      if (KIND == GEMM_NT) { kk = e % GEMM_BK; nn = e / GEMM_BK; } else { nn = e % BN; kk = e / BN; }
      bs[kk * LDN + nn] = rb[i];
    }
  };

  f32x16 acc[TM][TN_];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;
  const bool has_tail = ((k_end - k_begin) % GEMM_BK) != 0;
  if (nk > 0) {
    load_tile(k_begin, has_tail && nk == 1);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) load_tile(k_begin + (kt + 1) * GEMM_BK, has_tail && kt + 2 == nk);
    const float* as = As + buf * GEMM_BK * LDM + wm * WM + l31;
    const float* bs = Bs + buf * GEMM_BK * LDN + wn * WN + l31;
#pragma unroll
    for (int kk = 0; kk < GEMM_BK; kk += 2) {
      float a[TM], b[TN_];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = as[(kk + half) * LDM + i * 32];
#pragma unroll
      for (int j = 0; j < TN_; ++j) b[j] = bs[(kk + half) * LDN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) store_tile(buf ^ 1);
    __syncthreads();
  }

  if (KIND == GEMM_TN && want_csum) {
    // all waves are past the last barrier of the K loop; reuse the LDS as scratch
    smem[tid] = csum;
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      float tot = 0.f;
#pragma unroll
      for (int j = 0; j < GEMM_THREADS / BM; ++j) tot += smem[tid + j * BM];
      g.colsum_slab[(long)slab * g.M + m0 + tid] = tot;
    }
  }

  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C = g.C + (KIND == GEMM_TN ? (long)slab * g.slab_stride : 0L);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int n = n0 + wn * WN + j * 32 + l31;
      const bool n_ok = n < g.N;
      float bias = 0.f;
      if (KIND == GEMM_NT && g.bias && n_ok) bias = g.bias[n];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mrow = m0 + wm * WM + i * 32 + 8 * q + 4 * half;  // rows mrow..mrow+3
        uint32_t rnd[4];
        const bool philox = (KIND != GEMM_TN) && g.act == ACT_LEAKY_DROPOUT && g.drop.mode == DROP_PHILOX;
        if (philox) philox4x32_10((uint32_t)(mrow >> 2), (uint32_t)n, g.drop.key0, g.drop.key1, rnd);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int m = mrow + s;
          if (!n_ok || m >= g.M) continue;
          float v = acc[i][j][q * 4 + s];
          if (KIND == GEMM_NT) {
            v += bias;
            if (g.act == ACT_LEAKY_DROPOUT) {
              v = leaky(v);
              if (g.drop.mode == DROP_PHILOX) v = rnd[s] >= g.drop.thresh ? v * g.drop.scale : 0.f;
              else if (g.drop.mode == DROP_BUFFER)
                v = g.drop.mask[(long)m * g.drop.ld_mask + n] != 0.f ? v * g.drop.scale : 0.f;
            } else if (g.act == ACT_SIGMOID) {
              v = 1.f / (1.f + __expf(-v));
            }
          } else if (KIND == GEMM_NN) {
            if (g.act == ACT_LEAKY_DROPOUT) {
              const float h = g.H[(long)m * g.ldh + n];
              bool keep = true;
              float scale = 1.f;
              if (g.drop.mode == DROP_PHILOX) { keep = rnd[s] >= g.drop.thresh; scale = g.drop.scale; }
              else if (g.drop.mode == DROP_BUFFER) {
                keep = g.drop.mask[(long)m * g.drop.ld_mask + n] != 0.f; scale = g.drop.scale;
              }
              v *= leaky_drop_grad(h, keep, scale);
            } else if (g.act == ACT_SIGMOID) {
              const float h = g.H[(long)m * g.ldh + n];
              v *= h * (1.f - h);
            }
          }
          C[(long)m * g.ldc + n] = v;
        }
      }
    }
  }
}

template <int BM, int BN>
constexpr size_t gemm_lds_bytes() { return (size_t)2 * GEMM_BK * ((BM + 1) + (BN + 1)) * sizeof(float); }

}  // namespace gt
