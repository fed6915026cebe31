// Fused discriminator stack (float32, gfx950): the hidden layers ABOVE the first one, the head, and -- for the generator
// step's adversarial term -- the whole backward-data chain down to the adversarial input columns, as ONE launch per pass.
//
// Reference: gantts/models.py:121-141 (MLP: Dropout(LeakyReLU(Linear)) x L, last_linear, sigmoid), train.py:261-271 (D step:
// BCE terms of D(real) / D(fake), correct counts), train.py:307-308 (G step: loss_adv = -mean log D(G(x)) through the
// already-updated D).  What it replaces in the per-layer schedule (eng_step.hip: stack_forward / run_head / stack_backward):
//   D step  : L-1 forward products + d_head                                          -> 1 launch (mode DSTACK_D_STEP)
//   G step  : L-1 forward products + d_head + L-1 backward-data products + the
//             58-column product that yields dloss_adv / dy_hat_static[:, adv]        -> 1 launch (mode DSTACK_G_ADV)
// and the HBM round trips between them: in the G step no activation and no dZ of the discriminator leaves the CU at all.
//
// Decomposition.  The networks are per-frame, so a PANEL of 32 frames is independent of every other: one 256-thread
// workgroup walks its panel through all layers.  Wave w owns output columns [w HD/4, (w+1) HD/4): for HD = 256 two independent
// 32 x 32 accumulators, so consecutive MFMAs of a wave never depend on each other.  The panel's activation (and, on the way back,
// its dZ) lives in LDS as the MFMA A operand, row-major Act[m][k] with pitch HD + 4 (row m -> bank 4 m: conflict-free ds_read_b128
// over the 16-lane groups); 45 KB of LDS per workgroup at HD = 256.
//
// The k pair that MFMA group g of a 16-deep stage multiplies is {16 t + g, 16 t + 8 + g} (lanes 0-31 / 32-63) instead of
// {2 g, 2 g + 1}: any permutation of k is legal as long as A and B agree, and with this one a lane's eight k of a stage are
// CONTIGUOUS -- two ds_read_b128 per stage for A (256 B/clk, the rate one wave per SIMD already reaches; the first version read
// k-major images with ds_read_b32 and spent more time in the LDS than in the matrix pipe), and for B exactly what one lane can load
// from the weight matrix itself (below).  Act being row-major, the D step's stashes (H_l, dZ_top) leave as fully coalesced 16-byte stores.
//
// The activation derivative of every layer is kept as a 2-bit code per element (dropped / kept & negative / kept & positive) in
// registers of the thread that owns the element in the MFMA C layout -- the backward products have the same output tiling, so
// the owner of dZ[m][n] is the owner of H[m][n]: nothing is re-read and no Philox bit is regenerated.  Dropout bits: the
// layout-independent definition of gemm_f32.hip.h (philox_keep: one call per 8 rows of a column), or an injected mask buffer
// (parity tests), or none (eval).  Every sum has a fixed order: run-to-run bit-reproducible.
#pragma once
#include "frame_kernels.hip.h"
#include "gemm_f32.hip.h"

namespace gt {

#ifndef DS_ABL                    // tools/dstack_bench.hip: pieces compiled out (1 Philox calls, 2 weight loads, 4 MFMAs, 16 global stores)
#define DS_ABL 0
#endif
constexpr int DS_R = 32;          // frames per panel (one MFMA M tile)
constexpr int DS_BK = 16;         // K depth of one weight stage
constexpr int DS_THREADS = 256;
constexpr int DS_MAXL = 4;        // hidden layers (derivative codes: 2 registers per layer at HD = 256)

enum DStackMode { DSTACK_D_STEP = 0, DSTACK_G_ADV = 1 };

struct DStackArgs {
  int mode;                       // DStackMode
  int L;                          // hidden layers (1 .. DS_MAXL); layer 0's OUTPUT is this kernel's input
  int rows;                       // frames of the pass (D step: 2N, natural then generated; G step: N)
  int n_real;                     // D step: rows [0, n_real) are natural frames
  const float* H0;                // [rows][HD]: act(layer 0), dropout applied (eng_step.hip: the split first layer's launch)
  const float* W[DS_MAXL];        // layer l >= 1: [HD][HD] (out, in), row-major
  const float* b[DS_MAXL];
  DropoutSpec drop[DS_MAXL];      // dropout site of layer l (0 .. L-1)
  const float* w_last;            // last_linear: [HD], bias [1]
  const float* b_last;
  const float* mask; int n_mask;  // valid-frame mask, row r -> mask[r % n_mask]
  float eps;
  int unit_tv;                    // seed the backward pass of the UNNORMALISED loss (GT_OPT_COMM_TV_IN_SUMS)
  const double* tv_dev;           // data parallel: the all-reduced valid-frame count when it is not in *sc yet
  const StepScalars* sc;
  int want_grad;
  // outputs
  float* Hout[DS_MAXL];           // D step: activation stash of layers 1 .. L-1, [rows][HD] (the weight gradients' operand)
  float* dZtop;                   // D step: gradient at the top hidden layer's pre-activation, [rows][HD]
  float* Dout;                    // D(x) per row, or null
  HeadPartials* hp;               // [grid]
  float* dw_partial;              // D step: [grid][HD] partial gradient of last_linear.weight
  // G step: down to the adversarial columns of the first layer's input
  const float* W0; int ldw0;      // first layer's weight [HD][ldw0]
  int col0, Da;                   // its adversarial columns [col0, col0 + Da), Da <= 64
  float* gadv; int ld_gadv;       // [rows][ld_gadv]: dloss_adv / d(adversarial input columns)
  unsigned long long* dbg;        // tools/dstack_bench.hip: [grid][16] wall-clock stamps (100 MHz) of the phases, or null
};

template <int HD> constexpr int dstack_ap() { return HD + 4; }       // Act row pitch (floats)
constexpr int DS_XCH = 2 * 64 * 17;                                  // exchange area of the last product's K halves
constexpr int DS_BP = DS_BK + 4;                                     // row pitch of a wave's private weight stage (floats)
template <int HD> constexpr int dstack_priv() {                      // floats of the four private stages (the exchange area aliases them)
  return 4 * (HD / 4) * DS_BP > DS_XCH ? 4 * (HD / 4) * DS_BP : DS_XCH;
}
template <int HD> constexpr size_t dstack_lds_bytes() {
  return (size_t)(DS_R * dstack_ap<HD>() + dstack_priv<HD>() + HD + 64 + 2 * 32 * 5) * sizeof(float);
}

__device__ __forceinline__ float dstack_fprime(uint32_t code, float scale) {      // 0: dropped, 1: kept & h <= 0, 2: kept & h > 0
  return code == 0u ? 0.f : (code == 1u ? 0.01f * scale : scale);
}

template <int HD>
__global__ __launch_bounds__(DS_THREADS, 2) void dstack_kernel(const DStackArgs a) {
  static_assert(HD == 128 || HD == 256, "hidden widths the fused discriminator stack is instantiated for");
  constexpr int NT = HD / 128;                 // 32-column MFMA tiles per wave
  constexpr int WN = HD / 4;                   // output columns per wave
  constexpr int AP = dstack_ap<HD>();
  constexpr int NS = HD / DS_BK;               // stages of a product over K = HD
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Act = smem;                                               // [32][AP]
  float* priv = Act + DS_R * AP;                                   // [4 waves][HD / 4 rows][DS_BP]: a wave's weight stage (forward products)
  float* xch = priv;                                               // [2 tiles][64 lanes][17] (last product: the stages are idle then)
  float* wl = priv + dstack_priv<HD>();                            // [HD] last_linear.weight
  float* seeds = wl + HD;                                          // [32] dz per row (+ [32] spare)
  double* red = reinterpret_cast<double*>(seeds + 64);             // [32][5]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int r0 = blockIdx.x * DS_R;
  const int L = a.L;
  const bool g_mode = a.mode == DSTACK_G_ADV;
  const bool want_grad = a.want_grad != 0;
  int dbg_n = 0;
  auto stamp = [&]() { if (a.dbg && tid == 0 && dbg_n < 16) a.dbg[(long)blockIdx.x * 16 + dbg_n++] = wall_clock64(); };
  stamp();
  uint32_t code[DS_MAXL][NT];   // derivative codes of this thread's elements, 16 x 2 bits per tile (layer loops are unrolled: static indices)
#pragma unroll
  for (int l = 0; l < DS_MAXL; ++l)
#pragma unroll
    for (int j = 0; j < NT; ++j) code[l][j] = 0u;

  // ---- the weight operand goes global -> REGISTERS, in the MFMA fragment layout (version 3) -----------------------------------------------
  // Versions 1-2 staged the weight through an LDS ring shared by the workgroup's four waves: one barrier per 16-deep stage.  Two
  // such workgroups share a CU's four matrix pipes, and barrier-coupled waves convoy: where pipe arbitration favours workgroup A on
  // one SIMD and B on another, each workgroup waits at its barrier for the wave the other one is starving (measured: one workgroup per
  // CU 73.6 us, two 114 us for twice the work; start stagger, deeper prefetch, wider LDS accesses changed nothing).  A lane needs,
  // per stage and 32-column tile, eight consecutive k of ONE weight row (forward) or eight rows of one weight column (backward):
  // it loads exactly those -- 2 x 16 bytes (32-byte sectors, fully used) or 8 x 4 bytes (coalesced across the lanes).  No LDS
  // traffic for B, no barrier inside a product: the eight waves of a CU are independent between layer boundaries.  L2 -> CU bytes
  // are what the ring moved.  The two 16-byte pieces of a stage's fragment are requested separately, each as soon as the registers
  // it lands in are free: 1.5 stages (>= 1500 cycles) ahead of its first use (phase stamps: with one stage of run-ahead a product
  // took 10.0 us alone on a CU against 6.8 us of matrix time -- 64 distinct rows per load instruction are slow to arrive).
  f32x4 bf[2][NT][2];           // [register set][tile][piece: k 0-3 / 4-7 of the lane's eight]
  // backward: B(k = n_out, n = k_in) = W[n_out][k_in]; lane (n, half) <- W[16 t + 8 half + 4 piece + i][n], i = 0 .. 3: 4-byte loads,
  // lanes <-> consecutive columns (two 128-byte lines per instruction), straight into the fragment registers
  auto load_bwd = [&](f32x4 (&b)[NT][2], const float* W, int t, int piece) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* src = W + (long)(t * DS_BK + 8 * half + 4 * piece) * HD + wave * WN + j * 32 + l31;
#pragma unroll
      for (int i = 0; i < 4; ++i) b[j][piece][i] = src[(long)i * HD];
    }
  };
  // forward: B(k, n) = W[n][k], k-contiguous in memory.  A lane's fragment (8 consecutive k of ONE row) loaded directly is 64 distinct
  // cache lines per instruction -- the texture addresser works through them one per clock, and with 8-12 waves per CU it became the
  // bottleneck (version 3: 117 -> 93 us without the weight loads).  So the wave loads ITS 64 rows x 16 k block coalesced (lane ->
  // row 16 i + lane / 4, 16-byte chunk lane % 4: 16 lines per instruction), parks it in a WAVE-PRIVATE LDS stage (pitch 20: the
  // fragment read-back is a conflict-free ds_read_b128) and reads its fragments back: the transposition through the LDS the ring of
  // versions 1-2 did, but no other wave ever touches the stage -- LDS operations of one wave execute in order, so there is nothing to
  // synchronise.
  constexpr int UF = WN / 16;   // 16-byte units per lane and stage (HD = 256: 4)
  f32x4 rg[UF];
  float* pst = priv + wave * WN * DS_BP;
  auto issue_fwd = [&](const float* W, int t) {
#pragma unroll
    for (int u = 0; u < UF; ++u) rg[u] = ld4u(W + (long)(wave * WN + 16 * u + (lane >> 2)) * HD + t * DS_BK + 4 * (lane & 3));
  };
  auto commit_fwd = [&]() {
#pragma unroll
    for (int u = 0; u < UF; ++u) *reinterpret_cast<f32x4*>(pst + (16 * u + (lane >> 2)) * DS_BP + 4 * (lane & 3)) = rg[u];
  };
  auto frags_fwd = [&](f32x4 (&b)[NT][2]) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float* src = pst + (j * 32 + l31) * DS_BP + 8 * half;
      b[j][0] = *reinterpret_cast<const f32x4*>(src); b[j][1] = *reinterpret_cast<const f32x4*>(src + 4);
    }
  };

  // ---- one product over K = HD: acc[j] (32 rows x 32 columns each) += Act . W -------------------------------------------------------------
  // No barrier inside a product.  The A fragments (Act, LDS, two ds_read_b128 per stage) are read four MFMA groups ahead.
  // Backward: on entry bf[0] holds stage 0 and bf[1] piece 0 of stage 1 (requested under the caller's epilogue); stage t on set
  // c = t & 1 requests piece 1 of stage t + 1 (set c ^ 1) at group 0 and piece 0 of stage t + 2 (set c, its piece-0 registers are
  // consumed by then) at group 4: every piece has 1.5 stages to arrive.
  // Forward: on entry rg holds stage 0 (requested under the caller's epilogue): commit, read back, request stage 1.  Stage t: at
  // group 4 commit stage t + 1 (requested seven groups earlier), at group 5 request stage t + 2, at group 6 read the fragments of
  // stage t + 1 into the other register set.
  f32x16 acc[NT];
  auto prefetch_first = [&](const float* W, bool fwd) {
    if (fwd) issue_fwd(W, 0);
    else { load_bwd(bf[0], W, 0, 0); load_bwd(bf[0], W, 0, 1); load_bwd(bf[1], W, 1, 0); }
  };
  auto product = [&](const float* W, auto FWD_) {
    constexpr bool FWD = decltype(FWD_)::value;
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    f32x4 fa[2];
    const float* arow = Act + l31 * AP + 8 * half;
    // (r6: s_setprio 2 for the product / 0 for the epilogues, as the 64 x 64 product kernels do (GT_KLOOP_PRIO): 104.6 -> 104.3 / 107.4 / 105.1 us, not kept)
    fa[0] = *reinterpret_cast<const f32x4*>(arow);
    if (FWD) { commit_fwd(); frags_fwd(bf[0]); if (!(DS_ABL & 2)) issue_fwd(W, 1); }
    auto stage = [&](int t, f32x4 (&cur)[NT][2], f32x4 (&nxt)[NT][2]) {
#pragma unroll
      for (int gi = 0; gi < 8; ++gi) {
        if (gi == 0) {
          fa[1] = *reinterpret_cast<const f32x4*>(arow + t * DS_BK + 4);
          if (!FWD && t + 1 < NS && !(DS_ABL & 2)) load_bwd(nxt, W, t + 1, 1);
        }
        if (gi == 4) {
          if (t + 1 < NS) fa[0] = *reinterpret_cast<const f32x4*>(arow + (t + 1) * DS_BK);
          if (!FWD && t + 2 < NS && !(DS_ABL & 2)) load_bwd(cur, W, t + 2, 0);
          if (FWD && t + 1 < NS) commit_fwd();
        }
        if (FWD && gi == 5 && t + 2 < NS && !(DS_ABL & 2)) issue_fwd(W, t + 2);
        if (FWD && gi == 6 && t + 1 < NS) frags_fwd(nxt);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (DS_ABL & 4) { acc[j][0] += fa[gi >> 2][gi & 3] * cur[j][gi >> 2][gi & 3]; continue; }
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[gi >> 2][gi & 3], cur[j][gi >> 2][gi & 3], acc[j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
#pragma unroll 1
    for (int t = 0; t < NS; t += 2) {
      stage(t, bf[0], bf[1]);
      stage(t + 1, bf[1], bf[0]);
    }
    __syncthreads();            // every wave is done reading Act: the caller's epilogue may overwrite it
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  // Philox keep bits of the 8 rows 16 q_pair + 8 q' + 4 half + s (q' in {0, 1}, s in 0..3) of column n of this panel: bit 4 q' + s
  auto keep_bits8 = [&](const DropoutSpec& d, int q_pair, int n) -> uint32_t {
    uint32_t rnd[4];
    philox4x32_10(2u * philox_group(d, (uint32_t)(r0 >> 4) + (uint32_t)q_pair) + (uint32_t)half, (uint32_t)n, d.key0, d.key1, rnd);
    uint32_t bits = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) bits |= (philox_piece(rnd, p) >= d.thresh ? 1u : 0u) << p;
    return bits;
  };
  // the panel's rows of Act -> a [rows][HD] tensor: 16-byte chunks, consecutive lanes consecutive chunks (fully coalesced)
  auto store_panel = [&](float* dst) {
    if (DS_ABL & 16) return;
#pragma unroll
    for (int u = 0; u < HD / 32; ++u) {
      const int e = tid + u * DS_THREADS;
      const int m = e / (HD / 4), c4 = e % (HD / 4);
      if (r0 + m < a.rows) st4u(dst + (long)(r0 + m) * HD + 4 * c4, *reinterpret_cast<const f32x4*>(Act + m * AP + 4 * c4));
    }
  };

  // ---- prologue: the panel's H0 -> Act, last_linear.weight -> LDS; everything the later phases need from global memory is requested
  // here, so that its latency hides under the products: this thread's bias values, its row's mask value, the head's bias --------------------
#pragma unroll
  for (int u = 0; u < HD / 32; ++u) {
    const int e = tid + u * DS_THREADS;
    const int m = e / (HD / 4), c4 = e % (HD / 4);
    const int row = min(r0 + m, a.rows - 1);                   // clamped: rows past the pass are computed and never used
    *reinterpret_cast<f32x4*>(Act + m * AP + 4 * c4) = ld4u(a.H0 + (long)row * HD + 4 * c4);
  }
  if (L > 1) prefetch_first(a.W[1], true);
  for (int k = tid; k < HD; k += DS_THREADS) wl[k] = a.w_last[k];
  float bias[DS_MAXL][NT];
#pragma unroll
  for (int l = 1; l < DS_MAXL; ++l)
#pragma unroll
    for (int j = 0; j < NT; ++j) bias[l][j] = l < L ? a.b[l][wave * WN + j * 32 + l31] : 0.f;
  const float b_last = a.b_last[0];
  const int head_row = r0 + wave * 8 + (lane & 7);
  const float head_mask = head_row < a.rows ? a.mask[head_row % a.n_mask] : 0.f;
  const float inv_tv = a.unit_tv ? 1.0f : a.tv_dev ? 1.0f / (float)*a.tv_dev : a.sc->inv_tv;
  if (a.tv_dev && blockIdx.x == 0 && tid == 0) {
    StepScalars* scw = const_cast<StepScalars*>(a.sc);
    scw->tv = (float)*a.tv_dev; scw->inv_tv = 1.0f / (float)*a.tv_dev;
  }
  __syncthreads();
  stamp();

  // ---- derivative codes of layer 0 from its stored output: the G step's backward chain ends at layer 0; with ONE hidden layer the seed
  // gradient of either mode sits right on top of it ----------------------------------------------------------------------------------
  if (want_grad && (g_mode || L == 1)) {
    const DropoutSpec& d = a.drop[0];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WN + j * 32 + l31;
      const float* acol = Act + 4 * half * AP + n;      // element (8 q + 4 half + s, n) = acol[(8 q + s) AP]: constant offsets
      uint32_t cw = 0, bits = 0xffu;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (d.mode == DROP_PHILOX && (q & 1) == 0) bits = keep_bits8(d, q >> 1, n);      // one call decides the rows of q and q + 1
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int m = 8 * q + 4 * half + s4;
          bool keep = ((bits >> (4 * (q & 1) + s4)) & 1u) != 0u;
          if (d.mode == DROP_BUFFER) keep = d.mask[(long)min(r0 + m, a.rows - 1) * d.ld_mask + n] != 0.f;
          else if (d.mode == DROP_NONE) keep = true;
          cw |= (keep ? (acol[(8 * q + s4) * AP] > 0.f ? 2u : 1u) : 0u) << (2 * (4 * q + s4));
        }
      }
      code[0][j] = cw;
    }
  }

  // ---- forward through the hidden layers 1 .. L-1 (unrolled: the layer index is static in every access to the arguments) ---------------------
#pragma unroll
  for (int l = 1; l < DS_MAXL; ++l) {
    if (l < L) {
      product(a.W[l], T_{});
      stamp();
      // the next product's first fragments travel under this epilogue
      const bool next_fwd = l + 1 < L, next_bwd = !next_fwd && g_mode && want_grad;
      if (next_fwd) prefetch_first(a.W[l + 1 < DS_MAXL ? l + 1 : l], true);
      else if (next_bwd) prefetch_first(a.W[l], false);
      const DropoutSpec& d = a.drop[l];
      const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = wave * WN + j * 32 + l31;
        float* acol = Act + 4 * half * AP + n;
        uint32_t cw = 0, bits = 0xffu;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (!(DS_ABL & 1) && d.mode == DROP_PHILOX && (q & 1) == 0) bits = keep_bits8(d, q >> 1, n);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4) {
            const int m = 8 * q + 4 * half + s4;
            float v = leaky(acc[j][4 * q + s4] + bias[l][j]);
            bool keep = ((bits >> (4 * (q & 1) + s4)) & 1u) != 0u;
            if (d.mode == DROP_BUFFER) keep = d.mask[(long)min(r0 + m, a.rows - 1) * d.ld_mask + n] != 0.f;
            else if (d.mode == DROP_NONE) keep = true;
            v = keep ? v * scale : 0.f;
            cw |= (keep ? (v > 0.f ? 2u : 1u) : 0u) << (2 * (4 * q + s4));
            acol[(8 * q + s4) * AP] = v;          // (every reader of Act is past the product's closing barrier)
          }
        }
        code[l][j] = cw;
      }
      __syncthreads();
      stamp();
      if (!g_mode && a.Hout[l]) store_panel(a.Hout[l]);      // D step: the stash the weight gradients read (Act stays as it is until the next epilogue)
    }
  }

  // ---- head: z = <h, w> + b, D = sigmoid(z), BCE terms, seed dz per row ------------------------------------------------------------
  // The row's dot product is summed exactly as d_head_kernel sums it (lane <-> units lane + 64 j as an fmaf chain, then the xor
  // butterfly over the wave): the two paths then differ by the activations' rounding only.  Wave w takes rows 8 w .. 8 w + 7; lane i
  // of the wave does row 8 w + i's scalar work (all eight exp / log sequences run side by side).
  {
    float zrow = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* hrow = Act + (wave * 8 + i) * AP + lane;
      float part = 0.f;
#pragma unroll
      for (int jj = 0; jj < HD / 64; ++jj) part = fmaf(hrow[64 * jj], wl[lane + 64 * jj], part);
      const float zi = wave_sum(part) + b_last;
      if (lane == i) zrow = zi;
    }
    if (lane < 8) {
      const int m = wave * 8 + lane, row = r0 + m;
      const bool valid = row < a.rows;
      const float z = zrow;
      const float D = 1.f / (1.f + expf(-z));
      const float mk = head_mask;
      const bool is_real = g_mode || row < a.n_real;
      double s_real = 0, s_fake = 0, n_rok = 0, n_fok = 0;
      float dD;
      if (is_real) {
        s_real = (double)(logf(D + a.eps) * mk);
        if (!g_mode) n_rok = (D > 0.5f ? 1.0 : 0.0) * (double)mk;
        dD = -mk * inv_tv / (D + a.eps);
      } else {
        const float om = (1.f - D) + a.eps;
        s_fake = (double)(logf(om) * mk);
        n_fok = (D < 0.5f ? 1.0 : 0.0) * (double)mk;
        dD = mk * inv_tv / om;
      }
      const float dz = (valid && want_grad) ? dD * ((1.f - D) * D) : 0.f;
      if (!valid) { s_real = 0; s_fake = 0; n_rok = 0; n_fok = 0; }
      seeds[m] = dz;
      if (a.Dout && valid) a.Dout[row] = D;
      red[0 * 32 + m] = s_real; red[1 * 32 + m] = s_fake; red[2 * 32 + m] = n_rok; red[3 * 32 + m] = n_fok; red[4 * 32 + m] = (double)dz;
    }
  }
  __syncthreads();
  if (tid < 5) {                // one thread per sum, rows in a fixed order
    double v = 0.0;
#pragma unroll 8
    for (int m = 0; m < DS_R; ++m) v += red[tid * 32 + m];
    double* dst = &a.hp[blockIdx.x].s_real;      // HeadPartials: five consecutive doubles
    dst[tid] = v;
  }
  stamp();
  if (!want_grad) return;
  if (!g_mode && a.dw_partial && tid < HD) {      // d last_linear.weight: sum over the panel's rows of dz[m] h[m][k], k = tid
    float s = 0.f;
#pragma unroll 8
    for (int m = 0; m < DS_R; ++m) s = fmaf(seeds[m], Act[m * AP + tid], s);
    a.dw_partial[(long)blockIdx.x * HD + tid] = s;
  }
  __syncthreads();                                 // (the readers of Act above, before the seed image overwrites it)

  // ---- seed gradient at the top layer's pre-activation: dZ[m][n] = dz[m] w[n] f'(h[m][n]) -------------------------------------------
#pragma unroll
  for (int lt = 0; lt < DS_MAXL; ++lt) {
    if (lt == L - 1) {
      const DropoutSpec& d = a.drop[lt];
      const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = wave * WN + j * 32 + l31;
        const uint32_t cw = code[lt][j];
        const float wn = wl[n];
        float* acol = Act + 4 * half * AP + n;
        const float* sd = seeds + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 dz4 = *reinterpret_cast<const f32x4*>(sd + 8 * q);
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acol[(8 * q + s4) * AP] = dz4[s4] * wn * dstack_fprime((cw >> (2 * (4 * q + s4))) & 3u, scale);
        }
      }
    }
  }
  __syncthreads();
  if (!g_mode) { store_panel(a.dZtop); stamp(); return; }
  stamp();

  // ---- G step: backward-data chain, dZ_{l-1} = (dZ_l . W_l) (.) f'(h_{l-1}) -------------------------------------------------------------
  // adversarial columns of the first layer: B(k = unit, n = j) = W0[unit][col0 + j]; wave -> (column tile, K half); a lane loads its
  // column's units (4-byte loads, coalesced; col0 = 425 at cfg2: no alignment to speak of), FOUR 8-deep pieces ahead
  const int tile = wave & 1, kh = wave >> 1;
  const int jl = tile * 32 + l31;
  const bool jok = jl < a.Da;
  f32x4 wb[4];                  // ring of four 4-unit pieces
  auto load_adv = [&](f32x4& b, int pc /* piece: units kh HD/2 + 8 (pc >> 1) + ... */) {
    // piece pc covers units base + 16 (pc >> 1) + 8 half + 4 (pc & 1) .. + 3   (lanes 0-31 / 32-63 take k and k + 8 of a 16-deep stage)
    const float* src = a.W0 + (long)(kh * (HD / 2) + 16 * (pc >> 1) + 8 * half + 4 * (pc & 1)) * a.ldw0 + a.col0 + (jok ? jl : 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = jok ? src[(long)i * a.ldw0] : 0.f;
  };
#pragma unroll
  for (int l = DS_MAXL - 1; l >= 1; --l) {
    if (l < L) {
      product(a.W[l], F_{});
      stamp();
      if (l > 1) prefetch_first(a.W[l - 1], false);
      else {
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) load_adv(wb[pc], pc);
      }
      const DropoutSpec& d = a.drop[l - 1];
      const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int n = wave * WN + j * 32 + l31;
        const uint32_t cw = code[l - 1][j];
        float* acol = Act + 4 * half * AP + n;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int s4 = 0; s4 < 4; ++s4)
            acol[(8 * q + s4) * AP] = acc[j][4 * q + s4] * dstack_fprime((cw >> (2 * (4 * q + s4))) & 3u, scale);
      }
      __syncthreads();
    }
  }
  if (L == 1) {
#pragma unroll
    for (int pc = 0; pc < 4; ++pc) load_adv(wb[pc], pc);
  }

  // ---- gadv = dZ_0 . W0[:, col0 : col0 + Da]: two 32-column tiles, K split over the wave pairs ------------------------------------------
  {
    f32x16 ga;
#pragma unroll
    for (int r = 0; r < 16; ++r) ga[r] = 0.f;
    const float* arow = Act + l31 * AP + kh * (HD / 2) + 8 * half;
    constexpr int NP = HD / 2 / 8;                 // 4-unit pieces of this wave's K half (per lane half): 16 at HD = 256
#pragma unroll 1
    for (int p0 = 0; p0 < NP; p0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int pc = p0 + u;
        const f32x4 av = *reinterpret_cast<const f32x4*>(arow + 16 * (pc >> 1) + 4 * (pc & 1));
        const f32x4 bv = wb[u];
        if (pc + 4 < NP) load_adv(wb[u], pc + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) ga = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[i], ga, 0, 0, 0);
      }
    }
    // combine the two K halves in a fixed order (lower half + upper half) through LDS
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(tile * 64 + lane) * 17 + r] = ga[r];
    }
    __syncthreads();
    if (kh == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int row = r0 + 8 * q + 4 * half + s4;
          const float v = ga[4 * q + s4] + xch[(tile * 64 + lane) * 17 + 4 * q + s4];
          if (jok && row < a.rows) a.gadv[(long)row * a.ld_gadv + jl] = v;
        }
    }
  }
  stamp();
}

}  // namespace gt
