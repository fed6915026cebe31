// fp32 MFMA GEMM family for the frame x weight products of the G/D networks (gfx950 only).
//
// One compute core (v_mfma_f32_32x32x2_f32, exact f32 == an fmaf chain), three operand
// orientations, runtime-selected fused epilogues:
//
//   NT  Y  = act(X . W^T + b)          forward of nn.Linear      (reference gantts/models.py:129-141)
//   NN  dX = (dZ . W) (.) f'(H)        backward-data, activation derivative of the producer fused
//   TN  dW = dZ^T . X  (+ column sums of dZ = bias gradient), split over the frame dimension into
//                       deterministic partial slabs
//
// Tile: BM x BN (64 or 128 each) per 256-thread workgroup (4 waves, 2x2), each wave (BM/2)x(BN/2) in
// 32x32 MFMA tiles; K step 32.  The engine runs 64x64 tiles (4 workgroups per CU) whenever both operands take
// 16-byte loads and 128-wide tiles (2 per CU) otherwise (eng_gemm_f32.hip: launch_gemm; measurements in DESIGN.md 3.1).
// gemm_tile() is the tile body, gemm_store_tile() its epilogue; gemm_f32_kernel launches one product,
// gemm_pair_kernel a layer's backward-data product + weight gradient, gemm_chain.hip.h a stack of layers.
// LDS image is k-major  As[k][m] / Bs[k][n]; the MFMA
// operand fetch `As[k][m0 + lane%32]` is a conflict-free ds_read_b32 for any pitch.  The pitch is
// chosen per operand orientation so that the loader's LDS writes are conflict-free too:
//   * k-contiguous source (rows of X / W): pitch BM+1 (== 1 mod 32): a lane holding k..k+3 of one
//     row scatters 4 ds_write_b32 with bank = (k + m) % 32;
//   * m-contiguous source (rows of dZ / H / W for the transposed products): pitch BM+4, a lane
//     holding m..m+3 of one k row issues one aligned ds_write_b128.
// Global loads are 16 B per lane when the operand's row pitch and base are 16-byte aligned
// (hidden activations, most weights), 4 B per lane otherwise (row pitches 425 / 483 / 187 / 63 of
// the reference's (B,T,D) tensors are not multiples of 4 floats).
//
// Schedule: f32 MFMA is 64 cycles per instruction per SIMD and issue is in-order, so everything
// else is slotted into the MFMA shadow (see the K loop): next group's LDS fragments, slices of
// the next K-tile's global loads, and slices of its LDS writes with counted vmcnt waits.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace gt {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// 16-byte global accesses at addresses that are only 4-byte aligned (row pitches 425 / 483 / 187 of the reference's tensors):
// legal on gfx950 (the compiler emits global_load_dwordx4 / global_store_dwordx4 for an align-4 aggregate; measured with
// tools/unaligned_probe.hip), so k-contiguous operands and result rows take the vector path whatever their pitch.
struct __attribute__((packed, aligned(4))) F4U { float v[4]; };
__device__ __forceinline__ f32x4 ld4u(const float* p) {
  const F4U t = *reinterpret_cast<const F4U*>(p);
  f32x4 r; r[0] = t.v[0]; r[1] = t.v[1]; r[2] = t.v[2]; r[3] = t.v[3];
  return r;
}
__device__ __forceinline__ void st4u(float* p, const f32x4& r) {
  F4U t; t.v[0] = r[0]; t.v[1] = r[1]; t.v[2] = r[2]; t.v[3] = r[3];
  *reinterpret_cast<F4U*>(p) = t;
}

// PREC: arithmetic of the products.  PREC_F32: v_mfma_f32_32x32x2_f32 on f32 operands (exact f32, the default).
// PREC_BF16 (mixed precision, BASELINE.json configs[2]): operands are rounded to bf16 (RNE) on their way into LDS and
// multiplied by v_mfma_f32_32x32x16_bf16 with f32 accumulation -- 16x the matrix rate; memory layout, epilogues and
// the f32 master copies of weights / activations / gradients are unchanged.
enum GemmPrec { PREC_F32 = 0, PREC_BF16 = 1 };
constexpr int GEMM_KP = 40;     // PREC_BF16: LDS row pitch in bf16 (BK + 8 = 80 bytes: the 16 rows of a ds_read_b128 lane group
                                // start in 16 different 16-byte slots of the 256-byte bank row)

constexpr int GEMM_BK = 32;
constexpr int GEMM_THREADS = 256;

enum GemmKind { GEMM_NT = 0, GEMM_NN = 1, GEMM_TN = 2 };
enum Act { ACT_NONE = 0, ACT_LEAKY_DROPOUT = 1, ACT_SIGMOID = 2 };
enum DropMode { DROP_NONE = 0, DROP_PHILOX = 1, DROP_BUFFER = 2 };

struct DropoutSpec {
  int mode;           // DropMode
  float p;            // drop probability
  float scale;        // 1/(1-p)
  uint32_t thresh;    // keep iff 16-bit philox piece >= thresh  (thresh = p * 2^16)
  uint32_t key0, key1;  // philox key: (seed, site/step)
  const float* mask;  // DROP_BUFFER: [rows][ld_mask] 0/1 floats
  int ld_mask;
  // Data parallel ("RNG keyed by global sequence index so DP=k reproduces DP=1"; world > 1, T % 16 == 0): the
  // 16-row group g of this rank's LOCAL rows holds frames t0..t0+15 of local sequence b of half h ([real | generated] pass);
  // its keep bits are those of the group the same frames have in the one-process minibatch, where local sequence b of
  // rank r is sequence r + world * b (round-robin dealing):  philox_group().  dp_t16 == 0: identity (one rank).
  uint32_t dp_t16;    // T / 16
  uint32_t dp_nl16;   // 16-row groups per half of a two-half pass (local), 0xffffffff for a single block of rows
  uint32_t dp_half;   // added to the groups of the second half: B_global * T / 16 - dp_nl16
  uint32_t dp_add;    // rank * T / 16
  uint32_t dp_mul;    // (world - 1) * T / 16 per local sequence
  float dp_inv_t16;   // 1 / dp_t16
};
__device__ __forceinline__ uint32_t philox_group(const DropoutSpec& d, uint32_t g) {
  if (d.dp_t16 == 0u) return g;                   // kernel-argument uniform
  const uint32_t h = g >= d.dp_nl16 ? 1u : 0u;
  const uint32_t gg = g - h * d.dp_nl16;
  const uint32_t b = (uint32_t)(((float)gg + 0.5f) * d.dp_inv_t16);   // exact for gg < 2^21 (the engine checks)
  return g + h * d.dp_half + d.dp_add + b * d.dp_mul;
}

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

// Dropout keep bits from 16-bit pieces: ONE Philox call decides 8 elements of a column -- the rows that
// share (row >> 4, (row >> 2) & 1), i.e. exactly the 8 rows of a 16-row group that one lane of the
// 32x32 MFMA C layout owns (rows 8q + 4*half + s, q in {0,1}).  Layout-independent definition:
//   counter = (2*(row>>4) + ((row>>2)&1), col),  piece = 4*((row>>3)&1) + (row&3),  keep iff piece >= p*2^16
__device__ __forceinline__ uint32_t philox_piece(const uint32_t r[4], int piece) {
  return (r[piece >> 1] >> ((piece & 1) * 16)) & 0xffffu;
}
__device__ __forceinline__ bool philox_keep(uint32_t key0, uint32_t key1, uint32_t thresh, int row, int col) {
  uint32_t r[4];
  philox4x32_10((uint32_t)(2 * (row >> 4) + ((row >> 2) & 1)), (uint32_t)col, key0, key1, r);
  return philox_piece(r, 4 * ((row >> 3) & 1) + (row & 3)) >= thresh;
}

// the same decision for a dropout site (applies the data-parallel row-group map)
__device__ __forceinline__ bool philox_keep_spec(const DropoutSpec& d, int row, int col) {
  uint32_t r[4];
  philox4x32_10(2u * philox_group(d, (uint32_t)row >> 4) + (((uint32_t)row >> 2) & 1u), (uint32_t)col, d.key0, d.key1, r);
  return philox_piece(r, 4 * ((row >> 3) & 1) + (row & 3)) >= d.thresh;
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
  int accumulate;            // NT/NN: C += result (sum over LSTM directions / SRU highway term)
  // NT: a matrix added to the product before the activation: z = A.B^T + bias + addm[m mod addm_wrap][n].  The conditioned
  // discriminator's first layer sees [x | adv] for the real AND the generated rows with the same x: x . W_x^T is computed once
  // (N rows) and added to adv . W_adv^T of both halves (2N rows, addm_wrap = N).   (reference train.py:254-256)
  const float* addm; int ld_addm; int addm_wrap;
  // TN: the A operand is the element-wise sum A + A2 (same pitch): dW_x = (dZ_real + dZ_generated)^T . x over N frames
  const float* A2;
  // NT, GEMM_A_LEAKY_PHILOX_SEG: a SECOND K segment on top of the first (K_seg > 0): z = A . B^T + A_seg . B_seg^T + bias, both
  // operands k-contiguous, B_seg on B's pitch (the columns of W behind the first K).  dual_rows > 0: the first segment's rows
  // (M = dual_rows of them) serve TWO halves of the result -- rows [0, dual_rows) and [dual_rows, 2 dual_rows), each with its
  // own rows of A_seg: the first segment is multiplied once, the second segment and the epilogue run once per half.
  // (The split first layer of the conditioned discriminator in one launch: x . W_x^T + adv . W_adv^T, eng_step.hip: FirstSplit.)
  const float* A_seg; int lda_seg; const float* B_seg; int K_seg; int dual_rows;
  int wide_store;            // NT/NN: C (and H) 16-byte aligned with pitch % 4 == 0 -> full tiles are written
                             // row-wise with 16 B stores through an LDS transpose (set by the launcher)
  DropoutSpec drop;
  // TN split
  int k_chunk;               // rows of K per slab (multiple of GEMM_BK)
  long slab_stride;
  float* colsum_slab;        // TN: per-slab column sums of A (bias gradient), [nslab][M], or null
  int n_tiles_m, n_tiles_n;
};

// HW_REG_HW_ID (id 4) and HW_REG_XCC_ID (id 20), all 32 bits: s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id
__device__ __forceinline__ unsigned hw_id_reg() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ unsigned xcc_id_reg() { return __builtin_amdgcn_s_getreg((31 << 11) | 20); }
__device__ __forceinline__ unsigned cu_key() {   // CU_ID [11:8], SH_ID [12], SE_ID [15:13] of HW_ID + XCC_ID [3:0]
  const unsigned h = hw_id_reg();
  return ((xcc_id_reg() & 0xfu) << 7) | (((h >> 13) & 0x7u) << 5) | (((h >> 12) & 0x1u) << 4) | ((h >> 8) & 0xfu);
}
// LDS pitches per orientation (floats)
template <int KIND, int BM> constexpr int gemm_ldm() { return KIND == GEMM_TN ? BM + 4 : BM + 1; }
template <int KIND, int BN> constexpr int gemm_ldn() { return KIND == GEMM_NT ? BN + 1 : BN + 4; }
// (r6: a k-contiguous operand kept as ROWS in LDS, its fragments of four MFMA groups fetched with one ds_read_b128 -- a quarter of the LDS read
//  instructions, bit-identical, measured slower: forward launch 75.6 -> 78.5 us; tools/experiments/gemm_f32_lds_rows.patch)
template <int KIND, int BM, int BN, int PREC = PREC_F32, int BKT = GEMM_BK>
constexpr size_t gemm_lds_bytes() {
  if (PREC == PREC_F32) {     // the operand image, but never less than the epilogue staging (4 waves x 32 x (BN/2 + 4) floats)
    size_t img = (size_t)2 * BKT * (gemm_ldm<KIND, BM>() + gemm_ldn<KIND, BN>()) * sizeof(float), stg = (size_t)4 * 32 * (BN / 2 + 4) * 4;
    return img > stg ? img : stg;
  }
  // bf16 image [2][BM + BN rows][GEMM_KP], but never less than what the epilogue staging (4 waves x 32 x (BN/2 + 4) floats)
  // and the column-sum scratch (256 x 4 floats) reuse it for
  size_t img = (size_t)2 * (BM + BN) * GEMM_KP * 2, stg = (size_t)4 * 32 * (BN / 2 + 4) * 4, cs = (size_t)GEMM_THREADS * 4 * 4;
  return img > stg ? (img > cs ? img : cs) : (stg > cs ? stg : cs);
}

// GT_KLOOP_PRIO (compile-time): wave issue priority (s_setprio).  Co-resident workgroups are in different phases: while one runs its epilogue (the two
// Philox calls of a wave tile are ~ 2200 VALU cycles) the others are in their K loops, and an MFMA that becomes ready while the VALU port is taken by
// another wave's instruction waits for it -- a few cycles per MFMA, the "+ 8" of the 72 cycles per MFMA of DESIGN 3.1.  1 = K loops at priority 2,
// epilogues at 0; 2 = K loops at 1, every MFMA group issued at 3, epilogues at 0; 0 = everything at the default priority.
// Measured (r6, cfg2 ms/step, alternating builds, profiles/r06_kloop_prio_ab.txt): 0: 1.2889 1.2905 1.2945   1: 1.2794 1.2800 1.2807 1.2811
// 2: 1.2797 1.2799 1.2848; forward launch 76.2 -> 74.2 us, pair launch 103.8 -> 101.9 us with 2.  2 it is.
#ifndef GT_KLOOP_PRIO
#define GT_KLOOP_PRIO 2
#endif
__device__ __forceinline__ void gemm_kloop_prio(bool on) {
#if GT_KLOOP_PRIO
  if (on) __builtin_amdgcn_s_setprio(GT_KLOOP_PRIO == 2 ? 1 : 2); else __builtin_amdgcn_s_setprio(0);
#endif
}

// Result stores of the epilogue.  GT_EPI_STORE (compile-time, measurement): 0 = plain stores (the result stays dirty in the
// XCD's L2 and is written back at the end of the kernel), 1 = non-temporal, 2 = write-through (sc0 sc1).  Measured: 1 / 2
// shorten an isolated launch by 1-2 us (tools/gemm_tile_sweep.hip quick) but lengthen the training step (1.555 vs 1.540
// ms): the next layer reads this result, and finds less of it in L2 / MALL.  0 it is.
#ifndef GT_EPI_STORE
#define GT_EPI_STORE 0
#endif
__device__ __forceinline__ void epi_store4(float* dst, const f32x4& v) {
#if GT_EPI_STORE == 1
  __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
#elif GT_EPI_STORE == 2
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#else
  st4u(dst, v);
#endif
}
__device__ __forceinline__ void epi_store1(float* dst, float v) {
#if GT_EPI_STORE == 1
  __builtin_nontemporal_store(v, dst);
#elif GT_EPI_STORE == 2
  asm volatile("global_store_dword %0, %1, off sc0 sc1" ::"v"(dst), "v"(v) : "memory");
#else
  *dst = v;
#endif
}

// ---- epilogue of one BM x BN tile whose accumulators are in the MFMA C layout (col = lane&31,
// row = (r&3) + 8*(r>>2) + 4*(lane>>5)): bias / activation / dropout / f' of the producer, stores.
// LDS (`smem`, at least gemm_lds_bytes) must be free of K-loop readers on entry for the wide path: it starts with a barrier.
// AMODE: epilogue flavour restated at COMPILE time for the hot instantiations (-1 = decided at run time from g.act /
// g.drop.mode).  The epilogue is unrolled over the tile's accumulators; with run-time flavours every element carries the
// branches of all of them (the 64 x 64 forward kernel: ~2500 of its 3100 instructions are prologue + epilogue).
enum GemmAmode { GEMM_A_RUNTIME = -1, GEMM_A_NONE = 0, GEMM_A_LEAKY_PHILOX = 1,
                 GEMM_A_LEAKY_PHILOX_ADDM = 2,   // NT: LeakyReLU + Philox dropout on (product + bias + addm)
                 GEMM_A_TN_SUM2 = 3,             // TN: loader sums A + A2
                 GEMM_A_LEAKY_PHILOX_SEG = 4 };  // NT: LeakyReLU + Philox dropout behind a two-segment K loop (A_seg / B_seg)
template <int KIND, int BM, int BN, int PREC, int BKT, int AMODE = GEMM_A_RUNTIME>
__device__ __forceinline__ void gemm_store_tile(const GemmArgs& g, const int slab, const int m0, const int n0,
                                                f32x16 (&acc)[BM / 64][BN / 64], float* smem, const int m_lim) {
  constexpr bool RT = AMODE == GEMM_A_RUNTIME || AMODE == GEMM_A_TN_SUM2;
  const int g_act = RT ? g.act : (AMODE == GEMM_A_NONE ? (int)ACT_NONE : (int)ACT_LEAKY_DROPOUT);
  const int g_dmode = RT ? g.drop.mode : (AMODE == GEMM_A_NONE ? (int)DROP_NONE : (int)DROP_PHILOX);
  const bool has_addm = KIND == GEMM_NT && (AMODE == GEMM_A_LEAKY_PHILOX_ADDM || (RT && g.addm != nullptr));
  constexpr int WM = BM / 2, WN = BN / 2;
  constexpr int TM = WM / 32, TN_ = WN / 32;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  // ---- epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  float* C = g.C + (KIND == GEMM_TN ? (long)slab * g.slab_stride : 0L);
  const bool full_tile = m0 + BM <= m_lim && n0 + BN <= g.N;   // workgroup-uniform: no per-element guards (m_lim: row limit of this result block)
  const bool philox = (KIND != GEMM_TN) && g_act == ACT_LEAKY_DROPOUT && g_dmode == DROP_PHILOX;

  if (KIND != GEMM_TN && full_tile && g.wide_store) {
    // Wide path: everything that is keyed by the MFMA layout (bias column, the 4 Philox words of rows
    // mrow..mrow+3) is applied in the C layout, the 32 x WN strip is transposed through a wave-private
    // LDS region, and the strip leaves row-wise: 16 B per lane, WN*4-byte contiguous row segments
    // (4x fewer store instructions, full-line writes).  H (NN: f' of the producer) is read the same way.
    constexpr int EP = WN + 4;                    // pitch (floats), keeps 16 B alignment
    static_assert((size_t)4 * 32 * EP * sizeof(float) <= gemm_lds_bytes<KIND, BM, BN, PREC, BKT>(), "epilogue staging exceeds the LDS image");
    constexpr int LPR = WN / 4;                   // lanes per row
    constexpr int RPI = 64 / LPR;                 // rows per store instruction
    __syncthreads();                              // the K loop's LDS image is dead from here on
    float* stg = smem + wave * (32 * EP);
    const int srow = lane / LPR, sc4 = (lane % LPR) * 4;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int j = 0; j < TN_; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
        float bias = 0.f;
        if (KIND == GEMM_NT && g.bias) bias = g.bias[n];
        uint32_t rnd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int mrow = m0 + wm * WM + i * 32 + 8 * q + 4 * half;
          if (philox && (q & 1) == 0)
            philox4x32_10(2u * philox_group(g.drop, (uint32_t)mrow >> 4) + (uint32_t)half, (uint32_t)n, g.drop.key0, g.drop.key1, rnd);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const bool keep_px = philox_piece(rnd, 4 * (q & 1) + s4) >= g.drop.thresh;
            float v = acc[i][j][q * 4 + s4];
            if (KIND == GEMM_NT) {
              v += bias;
              if (has_addm) {
                const int mr = mrow + s4;
                v += g.addm[(long)(mr >= g.addm_wrap ? mr - g.addm_wrap : mr) * g.ld_addm + n];
              }
              if (g_act == ACT_LEAKY_DROPOUT) {
                v = leaky(v);
                if (g_dmode == DROP_PHILOX) v = keep_px ? v * g.drop.scale : 0.f;
                else if (g_dmode == DROP_BUFFER)
                  v = g.drop.mask[(long)(mrow + s4) * g.drop.ld_mask + n] != 0.f ? v * g.drop.scale : 0.f;
              } else if (g_act == ACT_SIGMOID) {
                v = 1.f / (1.f + expf(-v));
              }
            } else if (g_act == ACT_LEAKY_DROPOUT) {   // NN: keep bit * scale here, sign factor row-wise below
              if (g_dmode == DROP_PHILOX) v = keep_px ? v * g.drop.scale : 0.f;
              else if (g_dmode == DROP_BUFFER)
                v = g.drop.mask[(long)(mrow + s4) * g.drop.ld_mask + n] != 0.f ? v * g.drop.scale : 0.f;
            }
            stg[(8 * q + 4 * half + s4) * EP + j * 32 + l31] = v;
          }
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int it = 0; it < 32 / RPI; ++it) {
        const int row = it * RPI + srow;
        f32x4 v = *reinterpret_cast<const f32x4*>(stg + row * EP + sc4);
        const long m = m0 + wm * WM + i * 32 + row;
        const int n = n0 + wn * WN + sc4;
        if (KIND == GEMM_NN && g_act != ACT_NONE) {
          const f32x4 h = ld4u(g.H + m * g.ldh + n);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            v[c] *= g_act == ACT_SIGMOID ? h[c] * (1.f - h[c]) : (h[c] > 0.f ? 1.f : 0.01f);
        }
        float* dst = C + m * g.ldc + n;
        if (g.accumulate) {
          const f32x4 o = ld4u(dst);
#pragma unroll
          for (int c = 0; c < 4; ++c) v[c] += o[c];
        }
        epi_store4(dst, v);
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN_; ++j) {
      const int n = n0 + wn * WN + j * 32 + l31;
      const bool n_ok = full_tile || n < g.N;
      float bias = 0.f;
      if (KIND == GEMM_NT && g.bias && n_ok) bias = g.bias[n];
      uint32_t rnd[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int mrow = m0 + wm * WM + i * 32 + 8 * q + 4 * half;  // rows mrow..mrow+3
        if (philox && (q & 1) == 0)
          philox4x32_10(2u * philox_group(g.drop, (uint32_t)mrow >> 4) + (uint32_t)half, (uint32_t)n, g.drop.key0, g.drop.key1, rnd);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const int m = mrow + s;
          const bool keep_px = philox_piece(rnd, 4 * (q & 1) + s) >= g.drop.thresh;
          if (!full_tile && (!n_ok || m >= m_lim)) continue;
          float v = acc[i][j][q * 4 + s];
          if (KIND == GEMM_NT) {
            v += bias;
            if (has_addm) v += g.addm[(long)(m >= g.addm_wrap ? m - g.addm_wrap : m) * g.ld_addm + n];
            if (g_act == ACT_LEAKY_DROPOUT) {
              v = leaky(v);
              if (g_dmode == DROP_PHILOX) v = keep_px ? v * g.drop.scale : 0.f;
              else if (g_dmode == DROP_BUFFER)
                v = g.drop.mask[(long)m * g.drop.ld_mask + n] != 0.f ? v * g.drop.scale : 0.f;
            } else if (g_act == ACT_SIGMOID) {
              v = 1.f / (1.f + expf(-v));
            }
          } else if (KIND == GEMM_NN) {
            if (g_act == ACT_LEAKY_DROPOUT) {
              const float h = g.H[(long)m * g.ldh + n];
              bool keep = true;
              float scale = 1.f;
              if (g_dmode == DROP_PHILOX) { keep = keep_px; scale = g.drop.scale; }
              else if (g_dmode == DROP_BUFFER) {
                keep = g.drop.mask[(long)m * g.drop.ld_mask + n] != 0.f; scale = g.drop.scale;
              }
              v *= leaky_drop_grad(h, keep, scale);
            } else if (g_act == ACT_SIGMOID) {
              const float h = g.H[(long)m * g.ldh + n];
              v *= h * (1.f - h);
            }
          }
          if (KIND != GEMM_TN && g.accumulate) v += C[(long)m * g.ldc + n];
          epi_store1(C + (long)m * g.ldc + n, v);
        }
      }
    }
  }
}

// One BM x BN output tile (slab `slab` of the frame split for TN) of the product described by g, by the 256 threads of a
// workgroup; `smem` = gemm_lds_bytes<KIND, BM, BN, PREC>() bytes of LDS, free on entry (callers that run several tiles in
// one workgroup put a barrier between them).
// VA / VB: operand is loaded 16 B per lane (requires 16-byte aligned base and pitch % 4 == 0)
// BKT: K depth of one LDS stage (f32 only: 32, or 16 = half the LDS image, twice the workgroups per CU)
template <int KIND, int BM, int BN, bool VA, bool VB, int PREC = PREC_F32, int BKT = 32, int AMODE = GEMM_A_RUNTIME>
__device__ __forceinline__ void gemm_tile(const GemmArgs& g, const int slab, const int tile_m, const int tile_n, float* smem) {
  static_assert(BKT == 32 || (BKT == 16 && PREC == PREC_F32), "K depth of an LDS stage");
  constexpr int GEMM_BK = BKT;                  // shadows the namespace constant inside this function
  constexpr int LDM = gemm_ldm<KIND, BM>(), LDN = gemm_ldn<KIND, BN>();
  constexpr int WM = BM / 2, WN = BN / 2;       // wave tile
  constexpr int TM = WM / 32, TN_ = WN / 32;    // MFMA tiles per wave
  constexpr bool A_KC = KIND != GEMM_TN;        // A is k-contiguous in memory
  constexpr bool B_KC = KIND == GEMM_NT;        // B is k-contiguous in memory
  constexpr int VWA = VA ? 4 : 1, VWB = VB ? 4 : 1;
  constexpr int UA = BM * GEMM_BK / GEMM_THREADS / VWA;   // load units (instructions) per thread per K-tile
  constexpr int UB = BN * GEMM_BK / GEMM_THREADS / VWB;
  float* As = smem;                          // [2][BK][LDM]
  float* Bs = smem + 2 * GEMM_BK * LDM;      // [2][BK][LDN]
  // PREC_BF16: row-major bf16 images, k contiguous: Ah[2][BM][GEMM_KP], Bh[2][BN][GEMM_KP]
  __bf16* Ah = reinterpret_cast<__bf16*>(smem);
  __bf16* Bh = Ah + 2 * BM * GEMM_KP;
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

  // ---- loader geometry.  Unit u of this thread covers VW consecutive elements along the
  // operand's contiguous direction.  (kk, xx) = position inside the K-tile: k index, m/n index.
  auto a_pos = [&](int u, int& kk, int& mm) {
    const int e = tid + u * GEMM_THREADS;
    if (A_KC) { kk = (e % (GEMM_BK / VWA)) * VWA; mm = e / (GEMM_BK / VWA); }
    else      { mm = (e % (BM / VWA)) * VWA;      kk = e / (BM / VWA); }
  };
  auto b_pos = [&](int u, int& kk, int& nn) {
    const int e = tid + u * GEMM_THREADS;
    if (B_KC) { kk = (e % (GEMM_BK / VWB)) * VWB; nn = e / (GEMM_BK / VWB); }
    else      { nn = (e % (BN / VWB)) * VWB;      kk = e / (BN / VWB); }
  };
  // Element offsets inside the K-tile at k = 0, computed ONCE.  Rows/columns outside the matrix
  // are clamped to the last valid position (their products land in accumulator rows/columns the
  // epilogue never stores), so steady-state loads are unconditional: no exec-mask branches and no
  // per-load address arithmetic (the base pointer advances by one K-tile per iteration).  Only
  // the K tail is zero-filled (select after an in-bounds load).
  uint32_t offA[UA], offB[UB];
  const int mclamp = A_KC ? g.M - 1 : ((g.M - 1) / VWA) * VWA;   // last valid row / vector start
  const int nclamp = B_KC ? g.N - 1 : ((g.N - 1) / VWB) * VWB;
#pragma unroll
  for (int u = 0; u < UA; ++u) {
    int kk, mm; a_pos(u, kk, mm);
    const int m = min(m0 + mm, mclamp);
    offA[u] = A_KC ? (uint32_t)m * (uint32_t)g.lda + (uint32_t)kk : (uint32_t)kk * (uint32_t)g.lda + (uint32_t)m;
  }
#pragma unroll
  for (int u = 0; u < UB; ++u) {
    int kk, nn; b_pos(u, kk, nn);
    const int n = min(n0 + nn, nclamp);
    offB[u] = B_KC ? (uint32_t)n * (uint32_t)g.ldb + (uint32_t)kk : (uint32_t)kk * (uint32_t)g.ldb + (uint32_t)n;
  }
  const long stepA = A_KC ? (long)GEMM_BK : (long)GEMM_BK * g.lda;
  const long stepB = B_KC ? (long)GEMM_BK : (long)GEMM_BK * g.ldb;
  const float* pA = g.A + (A_KC ? (long)k_begin : (long)k_begin * g.lda);
  const long a2_delta = (KIND == GEMM_TN && AMODE == GEMM_A_TN_SUM2) ? (long)(g.A2 - g.A) : 0L;     // A2 as an offset from the moving A pointer
  const float* pB = g.B + (B_KC ? (long)k_begin : (long)k_begin * g.ldb);

  float ra[UA * VWA], rb[UB * VWB];

  // One unit of the next K-tile: global -> registers.  krem = number of valid k in that tile.
  auto load_a = [&](auto& ra, int u, bool tail, int krem) {
    int kk, mm; a_pos(u, kk, mm);
    uint32_t off = offA[u];
    if (tail && A_KC && VA) {
      // k-contiguous vector unit in the K tail: element-wise, clamped to the last valid k -- nothing is read beyond the row's
      // K elements (the pitch need not leave room behind them), the rest of the unit is zero
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = kk + c;
        const float v = pA[(long)off + (min(k, krem - 1) - kk)];
        ra[u * 4 + c] = k < krem ? v : 0.f;
      }
      return;
    }
    if (tail) {  // keep the address inside the matrix: step back to the last valid k
      const int back = max(0, kk - (krem - 1));
      off -= (uint32_t)back * (A_KC ? 1u : (uint32_t)g.lda);
    }
    if (VA) {
      f32x4 v = A_KC ? ld4u(pA + off) : *reinterpret_cast<const f32x4*>(pA + off);
      if (KIND == GEMM_TN && AMODE == GEMM_A_TN_SUM2) v += *reinterpret_cast<const f32x4*>(pA + a2_delta + off);
#pragma unroll
      for (int c = 0; c < 4; ++c) ra[u * 4 + c] = v[c];
    } else {
      ra[u] = pA[off];
      if (KIND == GEMM_TN && AMODE == GEMM_A_TN_SUM2) ra[u] += pA[a2_delta + off];
    }
    if (tail) {
#pragma unroll
      for (int c = 0; c < VWA; ++c) {
        const int k = A_KC ? kk + c : kk;
        if (k >= krem) ra[u * VWA + c] = 0.f;
      }
    }
  };
  auto load_b = [&](auto& rb, int u, bool tail, int krem) {
    int kk, nn; b_pos(u, kk, nn);
    uint32_t off = offB[u];
    if (tail && B_KC && VB) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = kk + c;
        const float v = pB[(long)off + (min(k, krem - 1) - kk)];
        rb[u * 4 + c] = k < krem ? v : 0.f;
      }
      return;
    }
    if (tail) {
      const int back = max(0, kk - (krem - 1));
      off -= (uint32_t)back * (B_KC ? 1u : (uint32_t)g.ldb);
    }
    if (VB) {
      const f32x4 v = B_KC ? ld4u(pB + off) : *reinterpret_cast<const f32x4*>(pB + off);
#pragma unroll
      for (int c = 0; c < 4; ++c) rb[u * 4 + c] = v[c];
    } else {
      rb[u] = pB[off];
    }
    if (tail) {
#pragma unroll
      for (int c = 0; c < VWB; ++c) {
        const int k = B_KC ? kk + c : kk;
        if (k >= krem) rb[u * VWB + c] = 0.f;
      }
    }
  };
  // TN: column sums of the A operand (dZ) fall out of the loader for free -- every unit of this
  // thread covers the same VWA columns (GEMM_THREADS is a multiple of BM / VWA).
  float csum[VWA];
#pragma unroll
  for (int c = 0; c < VWA; ++c) csum[c] = 0.f;
  const bool want_csum = KIND == GEMM_TN && g.colsum_slab != nullptr && tile_n == 0;
  // registers -> LDS
  auto store_a = [&](auto& ra, int u, float* as) {
    int kk, mm; a_pos(u, kk, mm);
    if (PREC == PREC_BF16) {
      __bf16* ah = reinterpret_cast<__bf16*>(as);      // the caller passes the buffer base reinterpreted
      if (A_KC && VA) {
        bf16x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (__bf16)ra[u * 4 + c];
        *reinterpret_cast<bf16x4*>(ah + mm * GEMM_KP + kk) = v;
      } else if (A_KC) {
        ah[mm * GEMM_KP + kk] = (__bf16)ra[u];
      } else {
#pragma unroll
        for (int c = 0; c < VWA; ++c) ah[(mm + c) * GEMM_KP + kk] = (__bf16)ra[u * VWA + c];
      }
      if (KIND == GEMM_TN && want_csum) {
#pragma unroll
        for (int c = 0; c < VWA; ++c) csum[c] += ra[u * VWA + c];
      }
      return;
    }
    if (A_KC) {
#pragma unroll
      for (int c = 0; c < VWA; ++c) as[(kk + c) * LDM + mm] = ra[u * VWA + c];
    } else if (VA) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ra[u * 4 + c];
      *reinterpret_cast<f32x4*>(as + kk * LDM + mm) = v;
    } else {
      as[kk * LDM + mm] = ra[u];
    }
    if (KIND == GEMM_TN && want_csum) {
#pragma unroll
      for (int c = 0; c < VWA; ++c) csum[c] += ra[u * VWA + c];
    }
  };
  auto store_b = [&](auto& rb, int u, float* bs) {
    int kk, nn; b_pos(u, kk, nn);
    if (PREC == PREC_BF16) {
      __bf16* bh = reinterpret_cast<__bf16*>(bs);
      if (B_KC && VB) {
        bf16x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = (__bf16)rb[u * 4 + c];
        *reinterpret_cast<bf16x4*>(bh + nn * GEMM_KP + kk) = v;
      } else if (B_KC) {
        bh[nn * GEMM_KP + kk] = (__bf16)rb[u];
      } else {
#pragma unroll
        for (int c = 0; c < VWB; ++c) bh[(nn + c) * GEMM_KP + kk] = (__bf16)rb[u * VWB + c];
      }
      return;
    }
    if (B_KC) {
#pragma unroll
      for (int c = 0; c < VWB; ++c) bs[(kk + c) * LDN + nn] = rb[u * VWB + c];
    } else if (VB) {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = rb[u * 4 + c];
      *reinterpret_cast<f32x4*>(bs + kk * LDN + nn) = v;
    } else {
      bs[kk * LDN + nn] = rb[u];
    }
  };

  f32x16 acc[TM][TN_];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN_; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // (mutable: the segmented forward product runs prologue + K loop once per segment)
  int nk = (k_end - k_begin + GEMM_BK - 1) / GEMM_BK;
  bool has_tail = ((k_end - k_begin) % GEMM_BK) != 0;
  auto prologue = [&]() {
  if (nk > 0) {   // prologue: tile 0 -> LDS buffer 0
    const bool tail0 = has_tail && nk == 1;
    const int krem0 = k_end - k_begin;
#pragma unroll
    for (int u = 0; u < UA; ++u) load_a(ra, u, tail0, krem0);
#pragma unroll
    for (int u = 0; u < UB; ++u) load_b(rb, u, tail0, krem0);
    pA += stepA; pB += stepB;
#pragma unroll
    for (int u = 0; u < UA; ++u) store_a(ra, u, PREC == PREC_BF16 ? reinterpret_cast<float*>(Ah) : As);
#pragma unroll
    for (int u = 0; u < UB; ++u) store_b(rb, u, PREC == PREC_BF16 ? reinterpret_cast<float*>(Bh) : Bs);
  }
  __syncthreads();
  };

  // K loop.  A K-tile is NG = 16 MFMA groups (one k-pair each, TM*TN_ MFMAs of 64 cycles).  The
  // matrix pipe is the bottleneck resource, so everything else is slotted into its shadow:
  //   * group g first issues the LDS fragment reads of group g+1 (register double buffer), so the
  //     lgkmcnt wait in front of the next group's MFMAs is already satisfied;
  //   * groups 0..NG/2-1 issue the NEXT tile's global loads, a slice per group,
  //   * groups NG/2..NG-1 write those registers to the other LDS buffer (the loads have had
  //     >= NG/2 groups, i.e. > 2000 cycles, to land: counted vmcnt waits, never 0 mid-loop).
  // sched_barrier pins the group boundaries so the compiler keeps this interleave.  The tile
  // body is instantiated three times (steady state / next tile is the K tail / last tile) so
  // `prefetch` and `tail` are compile-time inside the MFMA stream: no branches there.
  constexpr int NG = GEMM_BK / 2, NH = NG / 2;
  // (r6: the two Philox calls of a wave tile drawn in the MFMA shadows of the LAST K stage, handed to the epilogue as a 16-bit mask -- built,
  //  bit-identical, measured: cfg2 1.3021 / 1.3057 vs 1.3019 ms, pair launches 104.2 vs 104.2 us.  The epilogue's VALU burst already runs under
  //  the co-resident workgroups' K loops; removed again.)
  auto k_tile = [&](int kt, auto PF, auto TL) {
    constexpr bool prefetch = decltype(PF)::value, tail = decltype(TL)::value;
    const int buf = kt & 1;
    const int krem = k_end - (k_begin + (kt + 1) * GEMM_BK);
    const float* as = As + buf * GEMM_BK * LDM + wm * WM + l31 + half * LDM;      // the lane's fragment of group gi is row 2 gi + half
    const float* bs = Bs + buf * GEMM_BK * LDN + wn * WN + l31 + half * LDN;
    float* as_w = As + (buf ^ 1) * GEMM_BK * LDM;
    float* bs_w = Bs + (buf ^ 1) * GEMM_BK * LDN;
    float a_cur[TM], b_cur[TN_], a_nxt[TM], b_nxt[TN_];
#pragma unroll
    for (int i = 0; i < TM; ++i) a_cur[i] = as[i * 32];
#pragma unroll
    for (int j = 0; j < TN_; ++j) b_cur[j] = bs[j * 32];
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
#ifdef GT_ABLATE_NO_LDSREAD
      if (gi + 1 < NG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) { a_nxt[i] = a_cur[i]; asm volatile("" : "+v"(a_nxt[i])); }
#pragma unroll
        for (int j = 0; j < TN_; ++j) { b_nxt[j] = b_cur[j]; asm volatile("" : "+v"(b_nxt[j])); }
      }
#else
      if (gi + 1 < NG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_nxt[i] = as[(2 * gi + 2) * LDM + i * 32];
#pragma unroll
        for (int j = 0; j < TN_; ++j) b_nxt[j] = bs[(2 * gi + 2) * LDN + j * 32];
      }
#endif
#ifdef GT_ABLATE_NO_GLOBAL
      if (false) {
#else
      if (prefetch) {
#endif
        // unit u is loaded in group (u*NH)/U and stored in group NH + (u*NH)/U
#pragma unroll
        for (int u = 0; u < UA; ++u) {
          if ((u * NH) / UA == gi) load_a(ra, u, tail, krem);
#ifdef GT_ABLATE_NO_LDSWRITE
          if (NH + (u * NH) / UA == gi) { for (int c = 0; c < VWA; ++c) asm volatile("" ::"v"(ra[u * VWA + c])); }
#else
          if (NH + (u * NH) / UA == gi) store_a(ra, u, as_w);
#endif
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
          if ((u * NH) / UB + (UB < NH ? 1 : 0) == gi) load_b(rb, u, tail, krem);
#ifdef GT_ABLATE_NO_LDSWRITE
          if (NH + (u * NH) / UB + (UB < NH ? 1 : 0) == gi) { for (int c = 0; c < VWB; ++c) asm volatile("" ::"v"(rb[u * VWB + c])); }
#else
          if (NH + (u * NH) / UB + (UB < NH ? 1 : 0) == gi) store_b(rb, u, bs_w);
#endif
        }
      }
#if GT_KLOOP_PRIO == 2
      __builtin_amdgcn_s_setprio(3);
#endif
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[i], b_cur[j], acc[i][j], 0, 0, 0);
#if GT_KLOOP_PRIO == 2
      __builtin_amdgcn_s_setprio(1);
#endif
      if (gi + 1 < NG) {
#pragma unroll
        for (int i = 0; i < TM; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
        for (int j = 0; j < TN_; ++j) b_cur[j] = b_nxt[j];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (prefetch) { pA += stepA; pB += stepB; }
#ifndef GT_ABLATE_NO_BARRIER
    __syncthreads();
#endif
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  auto k_loop = [&]() {
  if (PREC == PREC_BF16) {
    // 8 MFMAs of 32 cycles per K-tile: here the loader, not the matrix pipe, sets the pace (the launch is bound by the
    // f32 operands it streams and converts).  Per tile: request tile t+1 (global -> registers), multiply tile t, deposit
    // tile t+1 into the other LDS buffer, one barrier; the load latency is covered by the three workgroups a CU holds.
    // (A second register set requesting tile t+2 was measured SLOWER -- 246 VGPRs, one workgroup less per CU:
    // cfg2 step 1.30 -> 1.54 ms.)
    auto mma_tile = [&](int buf) {
      const __bf16* ah = Ah + (buf * BM + wm * WM + l31) * GEMM_KP + 8 * half;
      const __bf16* bh = Bh + (buf * BN + wn * WN + l31) * GEMM_KP + 8 * half;
#pragma unroll
      for (int kk = 0; kk < GEMM_BK / 16; ++kk) {
        bf16x8 fa[TM], fb[TN_];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * GEMM_KP + kk * 16);
#pragma unroll
        for (int j = 0; j < TN_; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * GEMM_KP + kk * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN_; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      }
    };
    auto request = [&](auto& Ra, auto& Rb, int t) {        // global loads of tile t (pA / pB point at it)
      const bool tl = has_tail && t == nk - 1;
      const int krem = k_end - (k_begin + t * GEMM_BK);
#pragma unroll
      for (int u = 0; u < UA; ++u) load_a(Ra, u, tl, krem);
#pragma unroll
      for (int u = 0; u < UB; ++u) load_b(Rb, u, tl, krem);
      pA += stepA; pB += stepB;
    };
    auto deposit = [&](auto& Ra, auto& Rb, int buf) {      // registers -> LDS buffer `buf`
#pragma unroll
      for (int u = 0; u < UA; ++u) store_a(Ra, u, reinterpret_cast<float*>(Ah + buf * BM * GEMM_KP));
#pragma unroll
      for (int u = 0; u < UB; ++u) store_b(Rb, u, reinterpret_cast<float*>(Bh + buf * BN * GEMM_KP));
    };
    for (int t = 0; t < nk; ++t) {          // tile 0 is in LDS buffer 0 (prologue above)
      if (t + 1 < nk) request(ra, rb, t + 1);
      mma_tile(t & 1);
      if (t + 1 < nk) deposit(ra, rb, (t + 1) & 1);
      __syncthreads();
    }
  } else {
  int kt = 0;
  for (; kt + 2 < nk; ++kt) k_tile(kt, T_{}, F_{});                 // next tile is a full one
  if (kt + 2 == nk) {                                               // next tile is the last one
    if (has_tail) k_tile(kt, T_{}, T_{}); else k_tile(kt, T_{}, F_{});
    ++kt;
  }
  if (kt + 1 == nk) k_tile(kt, F_{}, F_{});                         // last tile: nothing to prefetch
  }
  };
  gemm_kloop_prio(true);
  prologue();
  k_loop();
  gemm_kloop_prio(false);

  if (KIND == GEMM_TN && want_csum) {
    // all waves are past the last barrier of the K loop; reuse the LDS as scratch.
    // threads tid, tid + BM/VWA, ... own the same VWA columns starting at (tid % (BM/VWA)) * VWA
    constexpr int OWN = BM / VWA;
#pragma unroll
    for (int c = 0; c < VWA; ++c) smem[tid * VWA + c] = csum[c];
    __syncthreads();
    if (tid < BM && m0 + tid < g.M) {
      const int own = tid / VWA, c = tid % VWA;
      float tot = 0.f;
#pragma unroll
      for (int j = 0; j < GEMM_THREADS / OWN; ++j) tot += smem[(own + j * OWN) * VWA + c];
      g.colsum_slab[(long)slab * g.M + m0 + tid] = tot;
    }
  }

#ifdef GT_ABLATE_NO_EPILOGUE
  {
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN_; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) v += acc[i][j][r];
    g.C[(long)(m0 + wm * WM + l31) * g.ldc + n0 + wn * WN + half] = v;
    return;
  }
#endif
  if constexpr (KIND == GEMM_NT && AMODE == GEMM_A_LEAKY_PHILOX_SEG) {
    // second K segment (per half), then the epilogue of that half
    const int nh = g.dual_rows > 0 ? 2 : 1;
    f32x16 acc0[TM][TN_];
    if (nh > 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN_; ++j) acc0[i][j] = acc[i][j];
    }
    for (int h = 0; h < nh; ++h) {
      if (h > 0) {
        __syncthreads();                       // the first half's epilogue staging is dead before the next prologue writes LDS
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN_; ++j) acc[i][j] = acc0[i][j];
      }
#pragma unroll
      for (int u = 0; u < UA; ++u) {
        int kk, mm; a_pos(u, kk, mm);
        offA[u] = (uint32_t)min(m0 + mm, mclamp) * (uint32_t)g.lda_seg + (uint32_t)kk;
      }
      pA = g.A_seg + (long)h * g.dual_rows * g.lda_seg;
      pB = g.B_seg;                            // same rows n0.. and pitch as B: offB stands
      k_begin = 0; k_end = g.K_seg;
      nk = (k_end + GEMM_BK - 1) / GEMM_BK;
      has_tail = (k_end % GEMM_BK) != 0;
      gemm_kloop_prio(true);
      prologue();
      k_loop();
      gemm_kloop_prio(false);
      gemm_store_tile<KIND, BM, BN, PREC, BKT, AMODE>(g, slab, m0 + h * g.dual_rows, n0, acc, smem, nh > 1 ? (h + 1) * g.dual_rows : g.M);
    }
    return;
  }
  gemm_store_tile<KIND, BM, BN, PREC, BKT, AMODE>(g, slab, m0, n0, acc, smem, g.M);
}

// XCD-aware tile order: consecutive workgroup ids round-robin over the 8 XCDs, so give each XCD a contiguous run of
// tiles (neighbouring tiles share the weight panel in that XCD's L2).
__device__ __forceinline__ int gemm_xcd_order(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// One launch = TWO independent products of a layer's backward pass: its backward-data product (g1: dX = (dZ . W) (.) f')
// and its weight gradient (g2: partial slabs of dZ^T . X), both on 64x64 tiles with 16-byte loadable operands.  The
// first n1 workgroups run g1's tiles, the rest g2's: one launch edge (ramp, first-tile latency, drain) instead of two,
// and the second product's tiles start while the first one's last tiles finish.
// tn_first: the weight-gradient workgroups (K = frames / slabs: several times the work of a backward-data tile) take the
// FIRST block ids, i.e. are dispatched first -- longest work first, the short tiles back-fill behind them (eng_gemm_f32_pair.hip).
template <int PREC, int AMODE = GEMM_A_RUNTIME>
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_pair_kernel(const GemmArgs g1, const GemmArgs g2, const int n1, const int tn_first) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int bid = blockIdx.x;
  const int n2 = (int)gridDim.x - n1;
  const bool is_nn = tn_first ? bid >= n2 : bid < n1;
  if (is_nn) {
    bid = gemm_xcd_order(tn_first ? bid - n2 : bid, n1);
    const int tile_m = bid / g1.n_tiles_n, tile_n = bid - tile_m * g1.n_tiles_n;
    gemm_tile<GEMM_NN, 64, 64, true, true, PREC, 32, AMODE>(g1, 0, tile_m, tile_n, smem);
  } else {
    bid = gemm_xcd_order(tn_first ? bid : bid - n1, n2);
    const int tiles_mn = g2.n_tiles_m * g2.n_tiles_n;
    const int slab = bid / tiles_mn, t = bid - slab * tiles_mn;
    const int tile_m = t / g2.n_tiles_n, tile_n = t - tile_m * g2.n_tiles_n;
    gemm_tile<GEMM_TN, 64, 64, true, true, PREC>(g2, slab, tile_m, tile_n, smem);
  }
}

// One launch = the two weight-gradient products of a SPLIT first layer (eng_step.hip: FirstSplit): g1 = the x columns, A operand
// summed over the two halves of the pass (GEMM_A_TN_SUM2), g2 = the adversarial columns over all rows; both write column blocks
// of the same slab set.  g2's workgroups (twice the frames each) take the FIRST block ids: longest work first.
// n3 > 0: a THIRD, independent product rides in the launch -- the last n3 workgroups run the 64 x 64 tiles of backward-data product g3 whose B operand
// takes the 4-byte loader and which has no activation derivative (the kept dloss_d / dy_hat_static of the D step: dZ_0[generated rows] . W_0[:, adv
// columns], train.py:265, 274 -- it reads the same dZ_0 as the two weight gradients and used to be a launch of its own at its floor).
template <int PREC>
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_tn_pair_kernel(const GemmArgs g1, const GemmArgs g2, const int n1, const GemmArgs g3, const int n3) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  int bid = blockIdx.x;
  if (bid >= (int)gridDim.x - n3) {
    bid -= (int)gridDim.x - n3;
    gemm_tile<GEMM_NN, 64, 64, true, false, PREC, 32, GEMM_A_NONE>(g3, 0, bid / g3.n_tiles_n, bid % g3.n_tiles_n, smem);
    return;
  }
  const int n2 = (int)gridDim.x - n3 - n1;
  if (bid >= n2) {      // g2's (long) workgroups take the first block ids
    bid = gemm_xcd_order(bid - n2, n1);
    const int tiles_mn = g1.n_tiles_m * g1.n_tiles_n;
    const int slab = bid / tiles_mn, t = bid - slab * tiles_mn;
    gemm_tile<GEMM_TN, 64, 64, true, true, PREC, 32, GEMM_A_TN_SUM2>(g1, slab, t / g1.n_tiles_n, t % g1.n_tiles_n, smem);
  } else {
    bid = gemm_xcd_order(bid, n2);
    const int tiles_mn = g2.n_tiles_m * g2.n_tiles_n;
    const int slab = bid / tiles_mn, t = bid - slab * tiles_mn;
    gemm_tile<GEMM_TN, 64, 64, true, true, PREC>(g2, slab, t / g2.n_tiles_n, t % g2.n_tiles_n, smem);
  }
}

// One launch = one product: workgroup -> (slab, tile_m, tile_n).
// (second launch bound = workgroups per CU the register allocation must leave room for)
#ifndef GT_SEG_WGS
#define GT_SEG_WGS 4
#endif
template <int BM, int BN, int AMODE> constexpr int gemm_min_wgs() { return (BM == 64 && BN == 64 && AMODE == GEMM_A_LEAKY_PHILOX_SEG) ? GT_SEG_WGS : 2; }
template <int KIND, int BM, int BN, bool VA, bool VB, int PREC = PREC_F32, int BKT = 32, int AMODE = GEMM_A_RUNTIME>
__global__ __launch_bounds__(GEMM_THREADS, (gemm_min_wgs<BM, BN, AMODE>())) void gemm_f32_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int bid = gemm_xcd_order(blockIdx.x, gridDim.x);
  const int tiles_mn = g.n_tiles_m * g.n_tiles_n;
  const int slab = bid / tiles_mn;
  const int t = bid - slab * tiles_mn;
  // n fastest: workgroups sharing an M panel (the big frame matrix) run back to back
  const int tile_m = t / g.n_tiles_n, tile_n = t - tile_m * g.n_tiles_n;
  gemm_tile<KIND, BM, BN, VA, VB, PREC, BKT, AMODE>(g, slab, tile_m, tile_n, smem);
}

}  // namespace gt
