// bf16-STORAGE GEMM family (BASELINE.json configs[2]: "bf16"; SURVEY 8(d) cfg3: bf16 storage / fp32 accumulate + fp32
// master weights and optimizer state).  gfx950 only.
//
// With GT_OPT_MATMUL_BF16 the tensors that only feed matrix products live in HBM as bfloat16 -- hidden activations, the
// gradients w.r.t. pre-activations (dZ), the network inputs' images and bf16 "shadow" copies of the weights, re-made from
// the float32 masters before every pass -- so a product moves half the bytes of the float32 path and its loader neither
// converts nor waits for 4-byte words.  ONE product form serves forward, backward-data and weight gradient:
//
//      C[m][n] = sum_k A[m][k] * B[n][k]            both operands bf16, k CONTIGUOUS in memory
//
//   forward         Y  [frame][out]  = X  [frame][in]   . W  [out][in]          epilogue: bias, LeakyReLU, dropout / sigmoid
//   backward-data   dX [frame][in]   = dZ [frame][out]  . WT [in][out]          epilogue: (.) f'(H) of the producer
//   weight gradient dW [out][in]     = dZT[out][frame]  . XT [in][frame]        split over frames into float32 slabs
//
// which is why every such tensor is kept in BOTH orientations ([frame][feature] and [feature][frame]): the transposed
// copy of a product's result is written by the same epilogue -- in the 32x32 MFMA C layout a lane owns 4 consecutive rows
// of one column, i.e. 8 contiguous bytes of the transposed image -- and tensors that do not come out of a product
// (inputs, the head's seed gradient, MLPG's gradient, the recurrence's dG) go through cast_transpose_kernel once.
//
// Arithmetic: v_mfma_f32_32x32x16_bf16, float32 accumulation; bias / activation / dropout / f' in float32 on the
// accumulators; results rounded to bf16 (RNE) once, on their way out.  Reductions, losses, gradients of the parameters
// (slabs), parameters and optimizer state stay float32.
#pragma once
#include <type_traits>
#include "gemm_f32.hip.h"

namespace gt {

enum GemmB16Epi { B16_FWD = 0, B16_BWD_DATA = 1, B16_SLAB = 2 };

struct GemmB16Args {
  const __bf16* A; int lda;      // [M][lda]; lda % 8 == 0, lda >= K rounded up to 8 (pad contents are never used)
  const __bf16* B; int ldb;      // [N][ldb]; same rules
  int M, N, K;
  float* C; int ldc;             // float32 result [M][ldc] (or null); B16_SLAB: slab s at C + s * slab_stride
  __bf16* Cb; int ldcb;          // bf16 result [M][ldcb] (or null)
  __bf16* CbT; int ldcbt;        // transposed bf16 result [N][ldcbt] (or null); ldcbt % 4 == 0
  const float* bias;             // B16_FWD: [N] or null
  int epi;                       // GemmB16Epi
  int act;                       // Act (B16_FWD: applied; B16_BWD_DATA: the PRODUCER's activation whose derivative multiplies)
  const __bf16* H; int ldh;      // B16_BWD_DATA with act != NONE: the producer's stored activation [M][ldh]
  int accumulate;                // float32 C += result (sum over LSTM directions)
  DropoutSpec drop;
  int k_chunk;                   // B16_SLAB: k per slab (multiple of 64)
  long slab_stride;
  float* rowsum_slab;            // B16_SLAB: per-slab row sums of A (= bias gradient when A = dZT), [nslab][M], or null
  int n_tiles_m, n_tiles_n;
  int band_c;                    // > 0: L2-aware tile order (gemm_b16_tile_of): the 8 XCDs form a (8 / band_c) x band_c grid over the tiles
};

// Timing ablations of the K loop, set by tools/gemm_b16_sweep only (results are then wrong on purpose): bit 0 no operand loads
// after the first stage, bit 1 no LDS deposit, bit 2 no MFMA, bit 3 no fragment reads.
#ifndef GT_B16_ABLATE
#define GT_B16_ABLATE 0
#endif
constexpr int B16_ABL = GT_B16_ABLATE;
// Workgroup -> (slab, tile).  Default: every XCD takes a contiguous run of tiles in n-fastest order (gemm_xcd_order).  band_c > 0
// (one slab, n_tiles_n % band_c == 0, n_tiles_m % (8 / band_c) == 0): XCD x owns the tile rows of row group x / band_c and the tile
// columns of column band x % band_c, and walks its columns fastest -- its band of B stays in that XCD's L2 for the whole
// launch and an A row panel is used band-width times back to back, instead of every tile row re-reading all of B through the
// fabric when B is larger than one L2 (4 MB).
__device__ __forceinline__ void gemm_b16_tile_of(const GemmB16Args& g, int* slab, int* tile_m, int* tile_n) {
  if (g.band_c > 0) {
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const int R = 8 / g.band_c, rows = g.n_tiles_m / R, cols = g.n_tiles_n / g.band_c;
    *slab = 0;
    *tile_m = (xcd / g.band_c) * rows + idx / cols;
    *tile_n = (xcd % g.band_c) * cols + idx % cols;
    return;
  }
  const int bid = gemm_xcd_order(blockIdx.x, gridDim.x);
  const int tiles_mn = g.n_tiles_m * g.n_tiles_n;
  *slab = bid / tiles_mn;
  const int t = bid - *slab * tiles_mn;
  *tile_m = t / g.n_tiles_n;
  *tile_n = t - *tile_m * g.n_tiles_n;
}
constexpr int B16_BK = 64;        // k depth of one LDS stage
constexpr int B16_KP = 72;        // LDS row pitch in bf16 (144 B: the 16 rows of a ds_read_b128 lane group start in 16 different 16-byte slots)

// (r6: every MFMA group of these K loops issued at wave priority 3 (s_setprio; what pays in the float32 family, gemm_f32.hip.h: GT_KLOOP_PRIO) measured
//  nothing on the 128 x 128 tiles (306.5 vs 308.8 us forward, 250.9 vs 251.9 backward-data) and a loss on the 256 x 256 eight-wave tile (257 vs 250,
//  220 vs 210): these K loops have no VALU-heavy neighbour to win against.  profiles/r06_b16_prio_sweep.txt)
template <int BM, int BN>
constexpr size_t gemm_b16_lds_bytes() { return (size_t)2 * (BM + BN) * B16_KP * 2; }

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
  const __bf16 a = (__bf16)lo, b = (__bf16)hi;
  return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// Epilogue flavour, COMPILE-TIME (the epilogue is fully unrolled over the tile's 16 .. 64 accumulators per lane: with the
// flavour decided at run time per element the 128 x 128 kernel was 18 000 instructions, 17 000 of them epilogue).
enum GemmB16Amode { B16_A_NONE = 0, B16_A_LEAKY_PHILOX = 1, B16_A_LEAKY_BUFFER = 2, B16_A_LEAKY = 3, B16_A_SIGMOID = 4 };

// One BM x BN tile.  256 threads = 2 x 2 waves, each wave (BM/2) x (BN/2) in 32x32 MFMA tiles.
// EPI (GemmB16Epi) and AMODE (GemmB16Amode) restate g.epi and (g.act, g.drop.mode) at compile time (launch_gemm_b16 dispatches).
// Epilogue of one BM x BN tile whose accumulators are in the 2 x 2-wave layout of gemm_b16_tile / gemm_b16_tile_dma.
// (WGM x WGN waves per workgroup, 2 x 2 unless stated: the 8-wave 256 x 256 tile of gemm_b16_tile_dma uses 2 x 4)
template <int BM, int BN, int EPI, int AMODE, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void gemm_b16_epilogue(const GemmB16Args& g, const int slab, const int m0, const int n0,
                                                  f32x16 (&acc)[BM / (32 * WGM)][BN / (32 * WGN)], __bf16* smem) {
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN_ = WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wm = wave / WGN, wn = wave % WGN;
  // ---- epilogue, in the MFMA C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
  float* C = g.C ? g.C + (EPI == B16_SLAB ? (long)slab * g.slab_stride : 0L) : nullptr;
  constexpr bool philox = EPI != B16_SLAB && AMODE == B16_A_LEAKY_PHILOX;
  // Full tiles with 16-byte-addressable results leave through a wave-private 32 x 32 float32 staging tile in LDS (the
  // K loop's image is dead: its last iteration ended with a barrier): every result image is then written as 16-byte
  // stores of contiguous runs -- 8 bf16 along the columns for Cb, 8 bf16 along the rows for the transposed twin, 4 floats
  // for C -- instead of one 2- / 4-byte store per element (80 -> 8 store instructions per lane of a 128 x 128 tile).
  const bool staged = m0 + BM <= g.M && n0 + BN <= g.N &&
                      (!g.Cb || ((g.ldcb & 7) == 0 && (((uintptr_t)g.Cb) & 15) == 0)) &&
                      (!g.CbT || ((g.ldcbt & 7) == 0 && (((uintptr_t)g.CbT) & 15) == 0)) &&
                      (!C || ((g.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0));
  float* stg = reinterpret_cast<float*>(smem) + wave * (32 * 33);
  static_assert((size_t)WGM * WGN * 32 * 33 * sizeof(float) <= gemm_b16_lds_bytes<BM, BN>(), "epilogue staging exceeds the LDS image");
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int nc = min(n0 + wn * WN + j * 32 + l31, g.N - 1);
      const float bias = (EPI == B16_FWD && g.bias) ? g.bias[nc] : 0.f;
      uint32_t rnd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mrow = m0 + wm * WM + i * 32 + 8 * q + 4 * half;     // rows mrow .. mrow + 3
        if (philox && (q & 1) == 0)
          philox4x32_10(2u * philox_group(g.drop, (uint32_t)mrow >> 4) + (uint32_t)half, (uint32_t)nc, g.drop.key0, g.drop.key1, rnd);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const long mc = min(mrow + s, g.M - 1);
          float x = acc[i][j][q * 4 + s];
          if (EPI == B16_FWD) {
            x += bias;
            if (AMODE == B16_A_LEAKY_PHILOX || AMODE == B16_A_LEAKY_BUFFER || AMODE == B16_A_LEAKY) {
              x = leaky(x);
              if (AMODE == B16_A_LEAKY_PHILOX) x = philox_piece(rnd, 4 * (q & 1) + s) >= g.drop.thresh ? x * g.drop.scale : 0.f;
              else if (AMODE == B16_A_LEAKY_BUFFER) x = g.drop.mask[mc * g.drop.ld_mask + nc] != 0.f ? x * g.drop.scale : 0.f;
            } else if (AMODE == B16_A_SIGMOID) {
              x = 1.f / (1.f + expf(-x));
            }
          } else if (EPI == B16_BWD_DATA && AMODE != B16_A_NONE) {
            const float h = (float)g.H[mc * g.ldh + nc];
            if (AMODE == B16_A_SIGMOID) {
              x *= h * (1.f - h);
            } else {
              bool keep = true;
              float scale = 1.f;
              if (AMODE == B16_A_LEAKY_PHILOX) { keep = philox_piece(rnd, 4 * (q & 1) + s) >= g.drop.thresh; scale = g.drop.scale; }
              else if (AMODE == B16_A_LEAKY_BUFFER) { keep = g.drop.mask[mc * g.drop.ld_mask + nc] != 0.f; scale = g.drop.scale; }
              x *= leaky_drop_grad(h, keep, scale);
            }
          }
          stg[(8 * q + 4 * half + s) * 33 + l31] = x;      // every tile leaves through the staging tile (edge tiles: element-wise below)
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      const long mb = m0 + wm * WM + i * 32;
      const int nb = n0 + wn * WN + j * 32;
      if (!staged) {        // edge tile / unaligned result: guarded element-wise stores, a ROLLED loop (16 elements per lane)
#pragma nounroll
        for (int t = 0; t < 16; ++t) {
          const int e = lane + 64 * t, row = e >> 5, col = e & 31;
          const long m = mb + row;
          const int nn = nb + col;
          if (m < g.M && nn < g.N) {
            const float x = stg[row * 33 + col];
            if (C) {
              float* dst = C + m * g.ldc + nn;
              *dst = g.accumulate ? *dst + x : x;
            }
            if (g.Cb) g.Cb[m * g.ldcb + nn] = (__bf16)x;
            if (g.CbT) g.CbT[(long)nn * g.ldcbt + m] = (__bf16)x;
          }
        }
      } else {
        if (g.Cb) {          // chunk c = lane + 64 t: row c / 4, columns (c % 4) * 8 .. + 7
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int c = lane + 64 * t, row = c >> 2, cc = (c & 3) * 8;
            const float* sp = stg + row * 33 + cc;
            uint4 w;
            w.x = pack_bf16x2(sp[0], sp[1]); w.y = pack_bf16x2(sp[2], sp[3]); w.z = pack_bf16x2(sp[4], sp[5]); w.w = pack_bf16x2(sp[6], sp[7]);
            *reinterpret_cast<uint4*>(g.Cb + (mb + row) * g.ldcb + nb + cc) = w;
          }
        }
        if (g.CbT) {         // chunk c: column c / 4 of the tile = row of the transposed image, rows (c % 4) * 8 .. + 7 of the tile
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const int c = lane + 64 * t, col = c >> 2, rc = (c & 3) * 8;
            const float* sp = stg + rc * 33 + col;
            uint4 w;
            w.x = pack_bf16x2(sp[0], sp[33]); w.y = pack_bf16x2(sp[66], sp[99]); w.z = pack_bf16x2(sp[132], sp[165]); w.w = pack_bf16x2(sp[198], sp[231]);
            *reinterpret_cast<uint4*>(g.CbT + (long)(nb + col) * g.ldcbt + mb + rc) = w;
          }
        }
        if (C) {             // chunk c = lane + 64 t: row c / 8, columns (c % 8) * 4 .. + 3
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int c = lane + 64 * t, row = c >> 3, cc = (c & 7) * 4;
            const float* sp = stg + row * 33 + cc;
            f32x4 w = {sp[0], sp[1], sp[2], sp[3]};
            float* dst = C + (mb + row) * g.ldc + nb + cc;
            if (g.accumulate) {
              const f32x4 o = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
              for (int e = 0; e < 4; ++e) w[e] += o[e];
            }
            *reinterpret_cast<f32x4*>(dst) = w;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// PF: stages of operand loads in flight in registers (1: stage t+1 is requested when the multiplication of stage t starts;
// 2: stage t+2 is -- one more stage of global-load latency covered, 16 .. 64 more registers).
template <int BM, int BN, int EPI, int AMODE, int PF = 1>
__device__ __forceinline__ void gemm_b16_tile(const GemmB16Args& g, const int slab, const int tile_m, const int tile_n, __bf16* smem) {
  static_assert(PF == 1 || PF == 2, "PF");
  constexpr int WM = BM / 2, WN = BN / 2, TM = WM / 32, TN_ = WN / 32;
  constexpr int UA = BM * (B16_BK / 8) / GEMM_THREADS, UB = BN * (B16_BK / 8) / GEMM_THREADS;   // 16-byte chunks per thread per stage
  static_assert(UA >= 1 && UB >= 1, "tile too small for 256 threads");
  __bf16* Ah = smem;                                  // [2][BM][KP]
  __bf16* Bh = smem + 2 * BM * B16_KP;                // [2][BN][KP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int k_begin = 0, k_end = g.K;
  if (EPI == B16_SLAB) { k_begin = slab * g.k_chunk; k_end = min(g.K, k_begin + g.k_chunk); }

  // chunk u of this thread: row (tid + u*256) / 8 of the tile, k offset ((tid + u*256) % 8) * 8 inside the stage
  const uint4* pa[UA];
  const uint4* pb[UB];
  int kca[UA], kcb[UB];
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    const int c = tid + u * GEMM_THREADS, row = min(m0 + c / 8, g.M - 1);
    kca[u] = (c % 8) * 8;
    pa[u] = reinterpret_cast<const uint4*>(g.A + (long)row * g.lda + k_begin + kca[u]);
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    const int c = tid + u * GEMM_THREADS, row = min(n0 + c / 8, g.N - 1);
    kcb[u] = (c % 8) * 8;
    pb[u] = reinterpret_cast<const uint4*>(g.B + (long)row * g.ldb + k_begin + kcb[u]);
  }
  uint4 ra[PF][UA], rb[PF][UB];
  const bool want_rs = EPI == B16_SLAB && g.rowsum_slab != nullptr && tile_n == 0;
  float rsum[UA];
#pragma unroll
  for (int u = 0; u < UA; ++u) rsum[u] = 0.f;

  // a chunk whose 8 k values are not all below krem (k left in this slab from the stage's first k): masked element-wise;
  // a chunk that starts at or beyond krem is not loaded at all (its address may lie outside the row)
  auto load_chunk = [&](const uint4* p, int kc, int krem) -> uint4 {
    uint4 v = {0u, 0u, 0u, 0u};
    if (kc < krem) {
      v = *p;
      if (kc + 8 > krem) {
        unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (kc + e >= krem) w[e >> 1] &= (e & 1) ? 0x0000ffffu : 0xffff0000u;
        v = make_uint4(w[0], w[1], w[2], w[3]);
      }
    }
    return v;
  };
  auto request = [&](int t, auto SET) {       // global -> register set SET, stage t
    constexpr int S = decltype(SET)::value;
    if ((B16_ABL & 1) && t > 1) return;
    const int krem = k_end - (k_begin + t * B16_BK);
    if (krem >= B16_BK) {
#pragma unroll
      for (int u = 0; u < UA; ++u) ra[S][u] = pa[u][0];
#pragma unroll
      for (int u = 0; u < UB; ++u) rb[S][u] = pb[u][0];
    } else {
#pragma unroll
      for (int u = 0; u < UA; ++u) ra[S][u] = load_chunk(pa[u], kca[u], krem);
#pragma unroll
      for (int u = 0; u < UB; ++u) rb[S][u] = load_chunk(pb[u], kcb[u], krem);
    }
#pragma unroll
    for (int u = 0; u < UA; ++u) pa[u] += B16_BK / 8;
#pragma unroll
    for (int u = 0; u < UB; ++u) pb[u] += B16_BK / 8;
  };
  auto deposit = [&](int buf, auto SET) {     // register set SET -> LDS buffer
    constexpr int S = decltype(SET)::value;
    if ((B16_ABL & 2) && buf >= 0) return;
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      const int c = tid + u * GEMM_THREADS;
      *reinterpret_cast<uint4*>(Ah + (buf * BM + c / 8) * B16_KP + kca[u]) = ra[S][u];
      if (want_rs) {
        const unsigned w[4] = {ra[S][u].x, ra[S][u].y, ra[S][u].z, ra[S][u].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) rsum[u] += __uint_as_float(w[e] << 16) + __uint_as_float(w[e] & 0xffff0000u);
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int c = tid + u * GEMM_THREADS;
      *reinterpret_cast<uint4*>(Bh + (buf * BN + c / 8) * B16_KP + kcb[u]) = rb[S][u];
    }
  };

  f32x16 acc[TM][TN_];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (k_end - k_begin + B16_BK - 1) / B16_BK;
  using S0 = std::integral_constant<int, 0>;
  using S1 = std::integral_constant<int, PF - 1>;
  if (nk > 0) { request(0, S0{}); if (PF == 2 && nk > 1) request(1, S1{}); deposit(0, S0{}); }
  __syncthreads();
  auto multiply = [&](int buf) {
    const __bf16* ah = Ah + (buf * BM + wm * WM + l31) * B16_KP + 8 * half;
    const __bf16* bh = Bh + (buf * BN + wn * WN + l31) * B16_KP + 8 * half;
#pragma unroll
    for (int kk = 0; kk < B16_BK / 16; ++kk) {
      bf16x8 fa[TM], fb[TN_];
      if (B16_ABL & 8) {      // fragments from registers that the compiler cannot fold
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = __builtin_bit_cast(bf16x8, ra[0][0]);
#pragma unroll
        for (int j = 0; j < TN_; ++j) fb[j] = __builtin_bit_cast(bf16x8, rb[0][0]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * B16_KP + kk * 16);
#pragma unroll
        for (int j = 0; j < TN_; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * B16_KP + kk * 16);
      }
      if (B16_ABL & 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN_; ++j) acc[i][j][kk] += (float)fa[i][kk] * (float)fb[j][kk];
      } else {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    }
  };
  if (PF == 1) {
    for (int t = 0; t < nk; ++t) {
      if (t + 1 < nk) request(t + 1, S0{});
      multiply(t & 1);
      if (t + 1 < nk) deposit((t + 1) & 1, S0{});
      __syncthreads();
    }
  } else {      // stage t lives in register set t & 1 until it is deposited; the loop is unrolled by two so that the sets are static
    for (int t = 0; t < nk; t += 2) {
      if (t + 2 < nk) request(t + 2, S0{});          // set 0 was deposited (stage t) before this iteration
      multiply(0);
      if (t + 1 < nk) deposit(1, S1{});
      __syncthreads();
      if (t + 1 < nk) {
        if (t + 3 < nk) request(t + 3, S1{});
        multiply(1);
        if (t + 2 < nk) deposit(0, S0{});
        __syncthreads();
      }
    }
  }

  if (want_rs) {      // the 8 lanes that share a row hold its 8 k-chunks: sum them, lane 0 of the group writes
#pragma unroll
    for (int u = 0; u < UA; ++u) {
      float v = rsum[u];
      v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64);
      const int row = m0 + (tid + u * GEMM_THREADS) / 8;
      if ((tid & 7) == 0 && row < g.M) g.rowsum_slab[(long)slab * g.M + row] = v;
    }
  }

  gemm_b16_epilogue<BM, BN, EPI, AMODE>(g, slab, m0, n0, acc, smem);
}

template <int BM, int BN, int EPI, int AMODE, int PF = 1>
__global__ __launch_bounds__(GEMM_THREADS, (BM == 64 && BN == 64) ? 4 : (BM * BN >= 256 * 256 ? 1 : 2)) void gemm_b16_kernel(const GemmB16Args g) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  int slab, tile_m, tile_n;
  gemm_b16_tile_of(g, &slab, &tile_m, &tile_n);
  gemm_b16_tile<BM, BN, EPI, AMODE, PF>(g, slab, tile_m, tile_n, reinterpret_cast<__bf16*>(smem_f));
}

// ------------------------------------------------------------------------------------------
// LDS-DMA form of the same tile, for products whose K (per slab) is a multiple of 64 and that need no row sums: operand
// stages go global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write, no wait for the data in the
// issue stream of the multiplying waves; lane l's 16 bytes land at base + 16 l, tools/lds_dma_probe.hip).  A stage is
// [rows][64 k] bf16 = 128-byte rows WITHOUT padding; the 16-byte chunks of a row are XOR-swizzled by (row >> 1) & 7 -- every
// lane chooses which global chunk it fetches, so the swizzle is free -- which makes the fragment reads (16 rows, one chunk
// each, per ds_read_b128 pass) hit 16 different bank quads.  NS stages form a ring: stage t + NS - 1 is requested at the
// top of iteration t into the buffer iteration t - 1 read; the end-of-iteration wait leaves the younger requests in flight
// (vmcnt counts them in order).  Measured against the register-staged loop (tools/gemm_b16_sweep): DESIGN.md 3.6.
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int NS>
constexpr size_t gemm_b16_dma_lds_bytes() { return (size_t)NS * (BM + BN) * 64 * 2; }

template <int BM, int BN, int EPI, int AMODE, int NS, int WGM = 2, int WGN = 2>
__device__ __forceinline__ void gemm_b16_tile_dma(const GemmB16Args& g, const int slab, const int tile_m, const int tile_n, __bf16* smem) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  constexpr int NW = WGM * WGN;                        // waves per workgroup
  constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN_ = WN / 32;
  constexpr int NA = BM / (8 * NW), NB = BN / (8 * NW);  // 1-KiB blocks (8 rows) per wave and stage
  static_assert(NA >= 1 && NB >= 1 && BM % (8 * NW) == 0 && BN % (8 * NW) == 0, "blocks per wave");
  constexpr int STAGE = (BM + BN) * 64;                // bf16 elements of one stage
  static_assert(NS >= 2 && NS <= 4, "ring depth");
  static_assert((size_t)NW * 32 * 33 * sizeof(float) <= gemm_b16_dma_lds_bytes<BM, BN, NS>(), "epilogue staging exceeds the LDS image");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wm = wave / WGN, wn = wave % WGN;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int k_begin = 0, k_end = g.K;
  if (EPI == B16_SLAB) { k_begin = slab * g.k_chunk; k_end = min(g.K, k_begin + g.k_chunk); }
  // block b of an operand = rows 8 b .. 8 b + 7 of the tile; this wave issues blocks wave, wave + NW, ...; lane -> row 8 b + (lane >> 3),
  // LDS slot lane & 7, which holds the row's chunk (lane & 7) ^ ((row >> 1) & 7)
  const __bf16* srcA[NA];
  const __bf16* srcB[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave + NW * i) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    srcA[i] = g.A + (long)min(m0 + r, g.M - 1) * g.lda + k_begin + 8 * c;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (wave + NW * i) * 8 + (lane >> 3), c = (lane & 7) ^ ((r >> 1) & 7);
    srcB[i] = g.B + (long)min(n0 + r, g.N - 1) * g.ldb + k_begin + 8 * c;
  }
  auto issue = [&](int buf) {
    __bf16* As = smem + buf * STAGE;
    __bf16* Bs = As + BM * 64;
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(As + (wave + NW * i) * 512), 16, 0, 0);
      srcA[i] += 64;
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(Bs + (wave + NW * i) * 512), 16, 0, 0);
      srcB[i] += 64;
    }
  };
  f32x16 acc[TM][TN_];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (k_end - k_begin) / 64;
#ifdef GT_B16_CLK_DBG
  const unsigned long long dbg_c0 = clock64(), dbg_w0 = wall_clock64();
#endif
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < nk) issue(p);
  // stage 0 has landed when at most the (NS - 2) younger stages' requests of this wave are outstanding
  if (NS == 2 || nk < 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if (NS == 3 || nk < 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory");
  __syncthreads();
  const int arow = wm * WM + l31, brow = wn * WN + l31;
  const int fsw = (l31 >> 1) & 7;                      // (row >> 1) & 7 of every fragment row this lane reads (tiles start at multiples of 32)
  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    int nbuf = buf + NS - 1; if (nbuf >= NS) nbuf -= NS;
    const bool more = t + NS - 1 < nk;
    if (more) issue(nbuf);
    const __bf16* ah = smem + buf * STAGE + arow * 64;
    const __bf16* bh = smem + buf * STAGE + BM * 64 + brow * 64;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int off = 8 * ((2 * kk + half) ^ fsw);
      bf16x8 fa[TM], fb[TN_];
#pragma unroll
      for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * 64 + off);
#pragma unroll
      for (int j = 0; j < TN_; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * 64 + off);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    // stage t + 1 (requested NS - 1 iterations ago) has landed for this wave once only the younger requests are outstanding
    if (NS == 2 || !more) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NA + NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (NA + NB)) : "memory");
    __syncthreads();                                   // ... for everybody, and everybody is done reading stage t
    ++buf; if (buf >= NS) buf = 0;
  }
#ifdef GT_B16_CLK_DBG     // tools/gemm_b16_sweep: shader cycles and 100 MHz wall ticks of this workgroup's K loop
  if (g.rowsum_slab && tid == 0) {
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    unsigned long long* d = reinterpret_cast<unsigned long long*>(g.rowsum_slab) + 2 * (size_t)blockIdx.x;
    d[0] = c1 - dbg_c0; d[1] = w1 - dbg_w0;
  }
#endif
  gemm_b16_epilogue<BM, BN, EPI, AMODE, WGM, WGN>(g, slab, m0, n0, acc, smem);
}

constexpr int gemm_b16_dma_wgs(int BM, int BN, int NS) { return 2 * NS * (BM + BN) * 128 <= 160 * 1024 ? 2 : 1; }   // workgroups per CU the LDS ring allows
template <int BM, int BN, int EPI, int AMODE, int NS, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN, gemm_b16_dma_wgs(BM, BN, NS)) void gemm_b16_dma_kernel(const GemmB16Args g) {
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  int slab, tile_m, tile_n;
  gemm_b16_tile_of(g, &slab, &tile_m, &tile_n);
  gemm_b16_tile_dma<BM, BN, EPI, AMODE, NS, WGM, WGN>(g, slab, tile_m, tile_n, reinterpret_cast<__bf16*>(smem_f));
}

// in [rows][ldi] (float32 or bf16)  ->  out [rows][ldo] bf16 (optional)  and  outT [cols][ldt] bf16 (optional), plus
// per-block column sums of the float32 values (optional: colsum_part [gridDim.x][cols], the bias gradient of a dZ that
// did not come out of a product).  64 x 64 tiles through LDS.  Full tiles leave as 16-byte stores in both orientations
// (8 bf16 along the columns for `out`, 8 along the rows for `outT`: 128 contiguous bytes per 8 lanes) and arrive as
// 16-byte loads when the source allows (float32, pitch % 4 == 0, 16-byte aligned base); pads of out / outT are not written.
// One 64 x 64 tile (tile row bx, tile column by).  SRC(r, c) -> float supplies elements that are not read as 16-byte
// vectors; `vin` (float32 source, pitch % 4 == 0, aligned) enables the vector read of full tiles.
template <typename SRC>
__device__ __forceinline__ void cast_transpose_tile(float (*tile)[65], const SRC& src, const float* vin, int ldvin, long rows, int cols,
                                                    __bf16* __restrict__ out, int ldo, __bf16* __restrict__ outT, long ldt,
                                                    float* __restrict__ colsum_part, int bx, int by) {
  const long r0 = (long)bx * 64;
  const int c0 = by * 64;
  const int tid = threadIdx.x;
  const bool full = r0 + 64 <= rows && c0 + 64 <= cols;
  if (vin && full) {         // 16 float4 per row: thread -> (row = tid / 16 + 16 p, 4 columns)
    const int cq = (tid & 15) * 4;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int rr = (tid >> 4) + 16 * p;
      const f32x4 v = *reinterpret_cast<const f32x4*>(vin + (r0 + rr) * ldvin + c0 + cq);
#pragma unroll
      for (int e = 0; e < 4; ++e) tile[rr][cq + e] = v[e];
    }
  } else {                   // one wave per row, 16 rows in flight per thread
    const int tx = tid & 63, ty = tid >> 6;
    float v[16];
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      const long r = r0 + ty + 4 * p;
      const int c = c0 + tx;
      v[p] = (r < rows && c < cols) ? src(r, c) : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 16; ++p) tile[ty + 4 * p][tx] = v[p];
  }
  __syncthreads();
  if (full && out && (ldo & 7) == 0 && (((uintptr_t)out) & 15) == 0) {          // row-major: thread -> (row = tid / 8 + 32 p, 8 columns) = one 16-byte store
    const int cc = (tid & 7) * 8;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int rr = (tid >> 3) + 32 * p;
      uint4 w;
      w.x = pack_bf16x2(tile[rr][cc], tile[rr][cc + 1]); w.y = pack_bf16x2(tile[rr][cc + 2], tile[rr][cc + 3]);
      w.z = pack_bf16x2(tile[rr][cc + 4], tile[rr][cc + 5]); w.w = pack_bf16x2(tile[rr][cc + 6], tile[rr][cc + 7]);
      *reinterpret_cast<uint4*>(out + (r0 + rr) * ldo + c0 + cc) = w;
    }
  } else if (out) {
    const int tx = tid & 63, ty = tid >> 6;
    for (int rr = ty; rr < 64; rr += 4) {
      const long r = r0 + rr;
      const int c = c0 + tx;
      if (r < rows && c < cols) out[r * ldo + c] = (__bf16)tile[rr][tx];
    }
  }
  if (full && outT && (ldt & 7) == 0 && (((uintptr_t)outT) & 15) == 0) {        // transposed: thread -> (column = tid / 8 + 32 p, 8 rows) = one 16-byte store
    const int rc = (tid & 7) * 8;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int cc = (tid >> 3) + 32 * p;
      uint4 w;
      w.x = pack_bf16x2(tile[rc][cc], tile[rc + 1][cc]); w.y = pack_bf16x2(tile[rc + 2][cc], tile[rc + 3][cc]);
      w.z = pack_bf16x2(tile[rc + 4][cc], tile[rc + 5][cc]); w.w = pack_bf16x2(tile[rc + 6][cc], tile[rc + 7][cc]);
      *reinterpret_cast<uint4*>(outT + (long)(c0 + cc) * ldt + r0 + rc) = w;
    }
  } else if (outT) {
    const int tx = tid & 63, ty = tid >> 6;
    for (int cc = ty; cc < 64; cc += 4) {
      const int c = c0 + cc;
      const long r = r0 + tx;
      if (c < cols && r < rows) outT[(long)c * ldt + r] = (__bf16)tile[tx][cc];
    }
  }
  if (colsum_part && tid < 64 && c0 + tid < cols) {
    float sum = 0.f;
#pragma unroll 8
    for (int rr = 0; rr < 64; ++rr) sum += tile[rr][tid];
    colsum_part[(long)bx * cols + c0 + tid] = sum;
  }
}
template <typename TIN>
struct PlainSrc {
  const TIN* in; int ldi;
  __device__ __forceinline__ float operator()(long r, int c) const { return (float)in[r * ldi + c]; }
};
template <typename TIN>
__global__ __launch_bounds__(256) void cast_transpose_kernel(const TIN* __restrict__ in, int ldi, long rows, int cols,
                                                             __bf16* __restrict__ out, int ldo,
                                                             __bf16* __restrict__ outT, long ldt,
                                                             float* __restrict__ colsum_part) {
  __shared__ float tile[64][65];
  const PlainSrc<TIN> src{in, ldi};
  const float* vin = (sizeof(TIN) == 4 && (ldi & 3) == 0 && ((((uintptr_t)in) & 15) == 0)) ? reinterpret_cast<const float*>(in) : nullptr;
  cast_transpose_tile(tile, src, vin, ldi, rows, cols, out, ldo, outT, ldt, colsum_part, (int)blockIdx.x, (int)blockIdx.y);
}
// Variational (per-sequence) input dropout fused into the cast: element (r, c) = in[r][c] * mul[r / T][c] -- the dropped
// layer input of the SRU (sru_kernels.hip.h) exists as bf16 images only, never as a float32 copy.
struct SeqDropSrc {
  const float* in; int ldi; const float* mul; int T, n;
  __device__ __forceinline__ float operator()(long r, int c) const { return in[r * ldi + c] * mul[(r / T) * n + c]; }
};
static __global__ __launch_bounds__(256) void seqdrop_cast_transpose_kernel(const SeqDropSrc src, long rows, int cols, __bf16* __restrict__ out, int ldo,
                                                                     __bf16* __restrict__ outT, long ldt) {
  __shared__ float tile[64][65];
  cast_transpose_tile(tile, src, nullptr, 0, rows, cols, out, ldo, outT, ldt, nullptr, (int)blockIdx.x, (int)blockIdx.y);
}
// Several float32 matrices in ONE launch (the weight shadows of a network: one job per nn.Linear).
constexpr int CAST_MAX_JOBS = 8;
struct CastJob { const float* in; __bf16* out; __bf16* outT; long rows, ldt; int ldi, cols, ldo, gy, block0, pad_; };
struct CastJobs { int n, pad_; CastJob j[CAST_MAX_JOBS]; };
static __global__ __launch_bounds__(256) void cast_transpose_multi_kernel(const CastJobs jobs) {
  __shared__ float tile[64][65];
  int q = 0;
  while (q + 1 < jobs.n && (int)blockIdx.x >= jobs.j[q + 1].block0) ++q;
  const CastJob& J = jobs.j[q];
  const int b = (int)blockIdx.x - J.block0;
  const PlainSrc<float> src{J.in, J.ldi};
  const float* vin = ((J.ldi & 3) == 0 && ((((uintptr_t)J.in) & 15) == 0)) ? J.in : nullptr;
  cast_transpose_tile(tile, src, vin, J.ldi, J.rows, J.cols, J.out, J.ldo, J.outT, J.ldt, nullptr, b / J.gy, b % J.gy);
}
// The discriminator's input image [x | feats[:, idx]] (train.py:254-256) written straight as bf16, both orientations, for
// `nhalf` stacked halves: virtual row r of half h (rows [h*N, (h+1)*N) of the image) = [x[r - h*N] | f_h[r - h*N][idx]],
// f_0 = fa (natural frames), f_1 = fb (generated frames).  row_off: first image row written (0, or N for the generated half alone).
struct CatSrc {
  const float* x; int cd; const float* fa; const float* fb; int ldf; const int* idx; long N, row_off;
  __device__ __forceinline__ float operator()(long r, int c) const {
    const long g = r + row_off;
    const long rr = g >= N ? g - N : g;
    if (c < cd) return x[rr * cd + c];
    return (g >= N ? fb : fa)[rr * ldf + idx[c - cd]];
  }
};
static __global__ __launch_bounds__(256) void cat_cast_transpose_kernel(const CatSrc src, long rows, int cols, __bf16* __restrict__ out, int ldo,
                                                                 __bf16* __restrict__ outT, long ldt) {
  __shared__ float tile[64][65];
  cast_transpose_tile(tile, src, nullptr, 0, rows, cols, out, ldo, outT, ldt, nullptr, (int)blockIdx.x, (int)blockIdx.y);
}

// out[r][c] (contiguous float32) = in[r][c] of a bf16 image with row pitch ld (inspection / parity hooks)
static __global__ void bf16_to_f32_kernel(const __bf16* __restrict__ in, long ld, long rows, int cols, float* __restrict__ out) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const long r = e / cols;
  const int c = (int)(e - r * cols);
  out[e] = (float)in[r * ld + c];
}

}  // namespace gt
