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
// workgroup walks its panel through all layers.  The panel's activation (32 x HD floats) lives in LDS as the MFMA A operand
// (k-major image Act[k][m], pitch 36: conflict-free ds_read_b32 fragments, ds_write_b128 from the C layout); the layer's weight
// streams through a two-stage LDS ring in 16-deep K steps (HD x 16 floats per stage) -- 16 flop per byte fetched from L2, the
// ratio of the 64 x 64 tiles of gemm_f32.hip.h, with the A operand costing nothing.  Wave w owns output columns
// [w HD/4, (w+1) HD/4): for HD = 256 two independent 32 x 32 accumulators, so consecutive MFMAs of a wave never depend on each
// other.  ~79 KB of LDS per workgroup (HD = 256): two workgroups per CU, de-phased, each one's epilogues run under the other's K
// loops; a pass of N = 16384 frames is 512 panels = exactly one resident round of 256 CUs x 2.
//
// The activation derivative of every layer is kept as a 2-bit code per element (dropped / kept & negative / kept & positive)
// in LDS, indexed by the thread that owns the element in the MFMA C layout -- the backward products have the same output
// tiling, so the owner of dZ[m][n] is the owner of H[m][n]: nothing is re-read and no Philox bit is regenerated.
// Dropout bits: the layout-independent definition of gemm_f32.hip.h (philox_keep: one call per 8 rows of a column), or an
// injected mask buffer (parity tests), or none (eval).  Every sum has a fixed order: run-to-run bit-reproducible.
#pragma once
#include "frame_kernels.hip.h"
#include "gemm_f32.hip.h"

namespace gt {

constexpr int DS_R = 32;          // frames per panel (one MFMA M tile)
constexpr int DS_BK = 16;         // K depth of one weight stage
constexpr int DS_THREADS = 256;
constexpr int DS_MAXL = 4;        // hidden layers the derivative codes have LDS for
constexpr int DS_AP = DS_R + 4;   // Act row pitch (floats): 16-byte aligned rows

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
};

template <int HD> constexpr int dstack_ldnb() { return HD + 4; }
template <int HD> constexpr size_t dstack_lds_bytes() {
  return (size_t)(HD * DS_AP + 2 * DS_BK * dstack_ldnb<HD>() + DS_MAXL * (HD / 128) * DS_THREADS + HD + 64 + 2 * 32 * 5) * sizeof(float);
}

__device__ __forceinline__ float dstack_fprime(uint32_t code, float scale) {      // 0: dropped, 1: kept & h <= 0, 2: kept & h > 0
  return code == 0u ? 0.f : (code == 1u ? 0.01f * scale : scale);
}

template <int HD>
__global__ __launch_bounds__(DS_THREADS, 2) void dstack_kernel(const DStackArgs a) {
  static_assert(HD == 128 || HD == 256, "hidden widths the fused discriminator stack is instantiated for");
  constexpr int NT = HD / 128;                 // 32-column MFMA tiles per wave
  constexpr int WN = HD / 4;                   // output columns per wave
  constexpr int AP = DS_AP;
  constexpr int LDNF = HD + 1;                 // weight stage pitch, forward (k-contiguous source: 4-way scatter, bank (k + n) % 32)
  constexpr int LDNB = dstack_ldnb<HD>();      // backward (n-contiguous source: ds_write_b128)
  constexpr int BUFS = DS_BK * LDNB;           // floats per stage
  constexpr int UW = HD / 64;                  // 16-byte load units per thread and stage
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Act = smem;                                               // [HD][AP]
  float* Bs = Act + HD * AP;                                       // [2][DS_BK][LDNB]
  uint32_t* codes = reinterpret_cast<uint32_t*>(Bs + 2 * BUFS);    // [DS_MAXL][NT][256]
  float* wl = reinterpret_cast<float*>(codes + DS_MAXL * NT * DS_THREADS);   // [HD] last_linear.weight
  float* seeds = wl + HD;                                          // [32] dz per row (+ [32] spare)
  double* red = reinterpret_cast<double*>(seeds + 64);             // [32][5]

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const int r0 = blockIdx.x * DS_R;
  const int L = a.L;
  const bool g_mode = a.mode == DSTACK_G_ADV;
  const bool want_grad = a.want_grad != 0;

  f32x4 rg[UW];                 // staged weight units (global -> registers -> LDS)

  // ---- weight stages ---------------------------------------------------------------------------------------------------------
  // forward: B(k, n) = W[n][k]; stage t holds k in [16 t, 16 t + 16).  Unit u: 4 consecutive k of row n.
  auto issue_fwd = [&](const float* W, int t) {
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = tid + u * DS_THREADS;
      rg[u] = ld4u(W + (long)(e >> 2) * HD + t * DS_BK + (e & 3) * 4);
    }
  };
  auto commit_fwd = [&](float* bs) {
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = tid + u * DS_THREADS;
      const int kk = (e & 3) * 4, nn = e >> 2;
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[(kk + c) * LDNF + nn] = rg[u][c];
    }
  };
  // backward: B(k = n_out, n = k_in) = W[n_out][k_in]; stage t holds weight rows [16 t, 16 t + 16).  Unit u: 4 consecutive columns.
  auto issue_bwd = [&](const float* W, int t) {
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = tid + u * DS_THREADS;
      rg[u] = ld4u(W + (long)(t * DS_BK + e / (HD / 4)) * HD + (e % (HD / 4)) * 4);
    }
  };
  auto commit_bwd = [&](float* bs) {
#pragma unroll
    for (int u = 0; u < UW; ++u) {
      const int e = tid + u * DS_THREADS;
      *reinterpret_cast<f32x4*>(bs + (e / (HD / 4)) * LDNB + (e % (HD / 4)) * 4) = rg[u];
    }
  };
  // adversarial columns of the first layer: B(k = unit, n = j) = W0[unit][col0 + j], 64 columns (zero beyond Da); 4-byte loads
  // (col0 = 425 at cfg2: no alignment to speak of).  Unit c of thread: row 16 t + 4 c + tid / 64, column tid % 64.
  auto issue_adv = [&](int t) {
    const int j = tid & 63;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int k = t * DS_BK + 4 * c + (tid >> 6);
      rg[0][c] = j < a.Da ? a.W0[(long)k * a.ldw0 + a.col0 + j] : 0.f;
    }
  };
  auto commit_adv = [&](float* bs) {
    const int j = tid & 63;
#pragma unroll
    for (int c = 0; c < 4; ++c) bs[(4 * c + (tid >> 6)) * LDNB + j] = rg[0][c];
  };

  // ---- one product over K = HD: acc[j] (32 rows x 32 columns each) += Act(k-major A image) . stage ring ------------------------
  // FWD selects the stage pitch and the loader; the first stage is in ring slot 0 on entry (committed + barrier by the caller).
  f32x16 acc[NT];
  auto product = [&](const float* W, auto FWD_) {
    constexpr bool FWD = decltype(FWD_)::value;
    constexpr int LDN = FWD ? LDNF : LDNB;
    constexpr int NS = HD / DS_BK;             // stages
    constexpr int NG = DS_BK / 2;              // MFMA groups (one k pair each) per stage
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    for (int t = 0; t < NS; ++t) {
      const float* as = Act + (t * DS_BK + half) * AP + l31;
      const float* bs = Bs + (t & 1) * BUFS + half * LDN + wave * WN + l31;
      float* bw = Bs + ((t + 1) & 1) * BUFS;
      const bool more = t + 1 < NS;
      float a_cur = as[0], a_nxt = 0.f, b_cur[NT], b_nxt[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) { b_cur[j] = bs[j * 32]; b_nxt[j] = 0.f; }
#pragma unroll
      for (int gi = 0; gi < NG; ++gi) {
        if (gi + 1 < NG) {
          a_nxt = as[(2 * gi + 2) * AP];
#pragma unroll
          for (int j = 0; j < NT; ++j) b_nxt[j] = bs[(2 * gi + 2) * LDN + j * 32];
        }
        if (gi == 0 && more) { if (FWD) issue_fwd(W, t + 1); else issue_bwd(W, t + 1); }
        if (gi == NG / 2 + 1 && more) { if (FWD) commit_fwd(bw); else commit_bwd(bw); }
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
        a_cur = a_nxt;
#pragma unroll
        for (int j = 0; j < NT; ++j) b_cur[j] = b_nxt[j];
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;

  // keep decision + derivative code of element (row r0 + m, column n) of layer l's output h
  //   returns the stored activation (0 when dropped) and sets code
  auto keep_bits8 = [&](const DropoutSpec& d, int q_pair, int n) -> uint32_t {   // Philox: the 8 rows 16 (q_pair) + 8 q' + 4 half + s of column n
    uint32_t rnd[4];
    philox4x32_10(2u * philox_group(d, (uint32_t)(r0 >> 4) + (uint32_t)q_pair) + (uint32_t)half, (uint32_t)n, d.key0, d.key1, rnd);
    uint32_t bits = 0;
#pragma unroll
    for (int p = 0; p < 8; ++p) bits |= (philox_piece(rnd, p) >= d.thresh ? 1u : 0u) << p;
    return bits;
  };

  // ---- prologue: last_linear.weight -> LDS, the panel's H0 -> Act, first weight stage -> ring slot 0 -----------------------------
  for (int k = tid; k < HD; k += DS_THREADS) wl[k] = a.w_last[k];
  {
    // lanes <-> (16 rows, 2 consecutive 16-byte pieces): 32-byte sectors fully used, LDS scatter bank (16 k4 + 4 c + m) % 32 distinct
    const int m = (lane & 15) + 16 * (wave & 1);
    const int row = min(r0 + m, a.rows - 1);                  // clamped: rows past the pass are computed and never used
    const float* src = a.H0 + (long)row * HD;
#pragma unroll
    for (int u = 0; u < HD / 32; ++u) {
      const int k4 = (lane >> 4) + 4 * (wave >> 1) + 8 * u;
      const f32x4 v = ld4u(src + 4 * k4);
#pragma unroll
      for (int c = 0; c < 4; ++c) Act[(4 * k4 + c) * AP + m] = v[c];
    }
  }
  if (L > 1) { issue_fwd(a.W[1], 0); commit_fwd(Bs); }
  else if (g_mode && want_grad) { issue_adv(0); commit_adv(Bs); }
  __syncthreads();

  // ---- G step: derivative codes of layer 0 from its stored output ------------------------------------------------------------
  if (g_mode && want_grad) {
    const DropoutSpec& d = a.drop[0];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WN + j * 32 + l31;
      uint32_t cw = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t bits = 0xffu;
        if (d.mode == DROP_PHILOX && (q & 1) == 0) bits = keep_bits8(d, q >> 1, n);
        const uint32_t kb = d.mode == DROP_PHILOX ? 0u : 0xffu;   // (placeholder, overwritten below for Philox)
        (void)kb;
        static uint32_t dummy = 0; (void)dummy;
        // Philox bits of the pair (q even, q odd) come from ONE call: keep them across the two iterations
        if ((q & 1) == 0) cw = (cw & 0x00ffffffu) | (bits << 24);      // park the 8 bits in the word's top byte
        const uint32_t b8 = d.mode == DROP_PHILOX ? (cw >> 24) : 0xffu;
        const f32x4 h4 = *reinterpret_cast<const f32x4*>(Act + n * AP + 8 * q + 4 * half);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          bool keep = ((b8 >> (4 * (q & 1) + s4)) & 1u) != 0u;
          if (d.mode == DROP_BUFFER) {
            const int row = min(r0 + 8 * q + 4 * half + s4, a.rows - 1);
            keep = d.mask[(long)row * d.ld_mask + n] != 0.f;
          }
          const uint32_t code = keep ? (h4[s4] > 0.f ? 2u : 1u) : 0u;
          // 16 elements x 2 bits: only the low 24 bits are free while the top byte parks Philox bits, so pack into a second word
          // for q >= 3?  No: element index e = 4 q + s4 < 16 -> bit 2 e < 32; the parked byte is dropped after q = 3 (see below).
          if (q < 3) cw = (cw & ~(3u << (2 * (4 * q + s4)))) | (code << (2 * (4 * q + s4)));
          else {
            // q == 3 writes bits 24 .. 31: the parked Philox byte was consumed for this element already (b8 is a copy)
            cw = (cw & ~(3u << (2 * (4 * q + s4)))) | (code << (2 * (4 * q + s4)));
          }
        }
      }
      codes[(0 * NT + j) * DS_THREADS + tid] = cw;
    }
  }

  // ---- forward through the hidden layers 1 .. L-1 -------------------------------------------------------------------------------
  for (int l = 1; l < L; ++l) {
    product(a.W[l], T_{});
    // the next product's first stage travels under this epilogue
    const bool next_fwd = l + 1 < L, next_bwd = !next_fwd && g_mode && want_grad;
    if (next_fwd) issue_fwd(a.W[l + 1], 0);
    else if (next_bwd) issue_bwd(a.W[l], 0);
    const DropoutSpec& d = a.drop[l];
    const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WN + j * 32 + l31;
      const float bias = a.b[l][n];
      uint32_t cw = 0, bits = 0xffu;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (d.mode == DROP_PHILOX && (q & 1) == 0) bits = keep_bits8(d, q >> 1, n);
        f32x4 h4;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          float v = leaky(acc[j][4 * q + s4] + bias);
          bool keep = ((bits >> (4 * (q & 1) + s4)) & 1u) != 0u;
          const int row = r0 + 8 * q + 4 * half + s4;
          if (d.mode == DROP_BUFFER) keep = d.mask[(long)min(row, a.rows - 1) * d.ld_mask + n] != 0.f;
          else if (d.mode == DROP_NONE) keep = true;
          v = keep ? v * scale : 0.f;
          cw |= (keep ? (v > 0.f ? 2u : 1u) : 0u) << (2 * (4 * q + s4));
          h4[s4] = v;
          if (!g_mode && a.Hout[l] && row < a.rows) a.Hout[l][(long)row * HD + n] = v;
        }
        *reinterpret_cast<f32x4*>(Act + n * AP + 8 * q + 4 * half) = h4;     // (all K-loop readers of Act are past the product's last barrier)
      }
      codes[(l * NT + j) * DS_THREADS + tid] = cw;
    }
    if (next_fwd) commit_fwd(Bs);
    else if (next_bwd) commit_bwd(Bs);
    __syncthreads();
  }

  // ---- head: z = <h, w> + b, D = sigmoid(z), BCE terms, seed dz per row ------------------------------------------------------------
  const float inv_tv = a.unit_tv ? 1.0f : a.tv_dev ? 1.0f / (float)*a.tv_dev : a.sc->inv_tv;
  if (a.tv_dev && blockIdx.x == 0 && tid == 0) {
    StepScalars* scw = const_cast<StepScalars*>(a.sc);
    scw->tv = (float)*a.tv_dev; scw->inv_tv = 1.0f / (float)*a.tv_dev;
  }
  {
    const int m = tid >> 3, p = tid & 7;
    float part = 0.f;
#pragma unroll 8
    for (int i = 0; i < HD / 8; ++i) part = fmaf(Act[(p + 8 * i) * AP + m], wl[p + 8 * i], part);
    part += __shfl_xor(part, 1);
    part += __shfl_xor(part, 2);
    part += __shfl_xor(part, 4);
    if (p == 0) {
      const int row = r0 + m;
      const bool valid = row < a.rows;
      const float z = part + a.b_last[0];
      const float D = 1.f / (1.f + expf(-z));
      const float mk = valid ? a.mask[row % a.n_mask] : 0.f;
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
      red[m * 5 + 0] = s_real; red[m * 5 + 1] = s_fake; red[m * 5 + 2] = n_rok; red[m * 5 + 3] = n_fok; red[m * 5 + 4] = (double)dz;
    }
  }
  __syncthreads();
  if (tid == 0) {
    HeadPartials pp;
    double v[5] = {0, 0, 0, 0, 0};
    for (int m = 0; m < DS_R; ++m)
#pragma unroll
      for (int c = 0; c < 5; ++c) v[c] += red[m * 5 + c];
    pp.s_real = v[0]; pp.s_fake = v[1]; pp.n_real_ok = v[2]; pp.n_fake_ok = v[3]; pp.db = v[4];
    a.hp[blockIdx.x] = pp;
  }
  if (!want_grad) return;
  if (!g_mode && a.dw_partial && tid < HD) {      // d last_linear.weight: sum over the panel's rows of dz[m] h[m][k], k = tid
    float s = 0.f;
#pragma unroll 8
    for (int m = 0; m < DS_R; ++m) s = fmaf(seeds[m], Act[tid * AP + m], s);
    a.dw_partial[(long)blockIdx.x * HD + tid] = s;
  }
  if (HD < DS_THREADS) __syncthreads();            // (dw readers of Act, before the seed image overwrites it below: only when tid >= HD exist)
  else __syncthreads();

  // ---- seed gradient at the top layer's pre-activation: dZ[m][n] = dz[m] w[n] f'(h[m][n]) -------------------------------------------
  {
    const DropoutSpec& d = a.drop[L - 1];
    const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WN + j * 32 + l31;
      const uint32_t cw = codes[((L - 1) * NT + j) * DS_THREADS + tid];
      const float wn = wl[n];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 dz4 = *reinterpret_cast<const f32x4*>(seeds + 8 * q + 4 * half);
        f32x4 o;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          o[s4] = dz4[s4] * wn * dstack_fprime((cw >> (2 * (4 * q + s4))) & 3u, scale);
          const int row = r0 + 8 * q + 4 * half + s4;
          if (!g_mode && row < a.rows) a.dZtop[(long)row * HD + n] = o[s4];
        }
        if (g_mode) *reinterpret_cast<f32x4*>(Act + n * AP + 8 * q + 4 * half) = o;
      }
    }
  }
  if (!g_mode) return;
  __syncthreads();

  // ---- G step: backward-data chain, dZ_{l-1} = (dZ_l . W_l) (.) f'(h_{l-1}) -------------------------------------------------------------
  for (int l = L - 1; l >= 1; --l) {
    product(a.W[l], F_{});
    if (l > 1) issue_bwd(a.W[l - 1], 0); else issue_adv(0);
    const DropoutSpec& d = a.drop[l - 1];
    const float scale = d.mode == DROP_NONE ? 1.f : d.scale;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = wave * WN + j * 32 + l31;
      const uint32_t cw = codes[((l - 1) * NT + j) * DS_THREADS + tid];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        f32x4 o;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) o[s4] = acc[j][4 * q + s4] * dstack_fprime((cw >> (2 * (4 * q + s4))) & 3u, scale);
        *reinterpret_cast<f32x4*>(Act + n * AP + 8 * q + 4 * half) = o;
      }
    }
    if (l > 1) commit_bwd(Bs); else commit_adv(Bs);
    __syncthreads();
  }

  // ---- gadv = dZ_0 . W0[:, col0 : col0 + Da]: two 32-column tiles, K split over the wave pairs ----------------------------------------------
  {
    constexpr int NS = HD / DS_BK;
    f32x16 ga;
#pragma unroll
    for (int r = 0; r < 16; ++r) ga[r] = 0.f;
    const int tile = wave & 1, kh = wave >> 1;           // wave -> (column tile, half of each stage's 16 k)
    for (int t = 0; t < NS; ++t) {
      const float* as = Act + (t * DS_BK + 8 * kh + half) * AP + l31;
      const float* bs = Bs + (t & 1) * BUFS + (8 * kh + half) * LDNB + tile * 32 + l31;
      const bool more = t + 1 < NS;
      if (more) issue_adv(t + 1);
#pragma unroll
      for (int gi = 0; gi < 4; ++gi) ga = __builtin_amdgcn_mfma_f32_32x32x2f32(as[2 * gi * AP], bs[2 * gi * LDNB], ga, 0, 0, 0);
      if (more) commit_adv(Bs + ((t + 1) & 1) * BUFS);
      __syncthreads();
    }
    // combine the two K halves in a fixed order (lower half + upper half) through LDS (the ring is free now)
    float* xch = Bs;                                     // [2 tiles][64 lanes][16]
    if (kh == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[(tile * 64 + lane) * 17 + r] = ga[r];
    }
    __syncthreads();
    if (kh == 0) {
      const int jcol = tile * 32 + l31;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const int row = r0 + 8 * q + 4 * half + s4;
          const float v = ga[4 * q + s4] + xch[(tile * 64 + lane) * 17 + 4 * q + s4];
          if (jcol < a.Da && row < a.rows) a.gadv[(long)row * a.ld_gadv + jcol] = v;
        }
    }
  }
}

}  // namespace gt
