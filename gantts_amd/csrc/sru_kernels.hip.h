// SRU recurrence of SRURNN (reference gantts/models.py:144-167 -> third-party `cuda_functional.SRU`,
// github.com/taolei87/sru 2017 layout, NOT vendored in the reference: restated from the published
// recurrence, Lei et al. 2017 arXiv:1709.02755; parity unpinned, see oracle/gantts_oracle.py).
//
//   U = x W  (one f32 MFMA GEMM per layer, gemm_f32.hip.h; column j owns U[.., j*k .. j*k+k-1])
//   f = sigmoid(u1 + b_f[j]),  r = sigmoid(u2 + b_r[j])
//   c_t = (c_{t-1} - u0) f + u0
//   h_t = (g(c_t) mask_h - x') r + x'          x' = x_t[j] (k == 3) or u3 (k == 4)
//
// The scan is sequential in time per (sequence, column) and embarrassingly parallel across them:
// one lane per column, lanes <-> consecutive columns (coalesced rows of U), time unrolled by 4 so
// that the U / x loads of the next frames (which do not depend on the carried state) are in flight
// while the current frame is computed -> HBM-streaming bound.  Columns j >= H of a bidirectional
// layer walk time backwards.  Sequence lengths are ignored, exactly like the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hip.h"

namespace gt {

enum SruAct { SRU_ID = 0, SRU_TANH = 1, SRU_RELU = 2 };

struct SruArgs {
  int B, T, H, dirs, k, act;
  const float* U; int ldu;        // [N][ncols*k], row = b*T + t
  const float* x; int ldx;        // layer input (highway term when k == 3)
  const float* bias;              // [2*ncols] = b_f | b_r
  float* h;                       // [N][ncols]
  float* c;                       // [N][ncols] cell-state stash
  // backward
  const float* dh;                // [N][ncols]
  float* dU;                      // [N][ncols*k]
  float* dx; int lddx;            // k == 3: highway gradient d/dx' -> [N][ncols]
  float* dbias_part;              // [B][2*ncols]
  // variational output dropout (one mask per (sequence, column), shared over time)
  int use_mask; float keep_scale; uint32_t thresh, key0, key1;
  const float* mask_buf;          // parity hook: injected 0/1 keep mask [B][ncols] instead of the Philox stream
  // backward only: this layer's output is the NEXT layer's input, and that layer's variational input dropout (+ its k == 3
  // highway gradient) is applied here, where the gradient is read: dh = g * up_mul[b][col] + up_add.  The multiplier is
  // constant per lane (one (sequence, column) pair per lane).
  const float* up_mul;            // [B][ncols] multipliers {0, 1/(1-p)} of the next layer's input dropout, or null
  const float* up_add; int ld_up_add;   // [N][ncols] highway gradient of the next layer (k == 3), or null
};

__device__ __forceinline__ float sru_act(float c, int act) { return act == SRU_RELU ? fmaxf(c, 0.f) : (act == SRU_TANH ? tanhf(c) : c); }
__device__ __forceinline__ float sru_dact(float c, float val, int act) {
  return act == SRU_RELU ? (c > 0.f ? 1.f : 0.f) : (act == SRU_TANH ? 1.f - val * val : 1.f);
}
__device__ __forceinline__ float sru_mask(const SruArgs& a, int b, int col) {
  if (!a.use_mask) return 1.f;
  if (a.mask_buf) return a.mask_buf[(long)b * (a.H * a.dirs) + col] != 0.f ? a.keep_scale : 0.f;
  uint32_t r[4];
  philox4x32_10((uint32_t)b, (uint32_t)col, a.key0, a.key1, r);
  return r[0] >= a.thresh ? a.keep_scale : 0.f;
}

// The scan has only B * ncols independent lanes (32 768 at B = 32, 6x512 bidirectional): its HBM rate is set by the bytes
// each lane keeps in flight: SRU_UNROLL_F / _B frames of loads per lane (under the 63 the vmcnt counter can track) and
// 64-lane workgroups, so that the 512 waves spread over all 256 CUs instead of 128.
constexpr int SRU_UNROLL_F = 12;      // forward: 4 loads per frame -> 48 in flight
constexpr int SRU_UNROLL_B = 8;       // backward: 6 loads per frame (7 with the next layer's highway gradient) -> 48 / 56 in flight
// (measured per layer at B = 32, T = 1024, 6x512 bidirectional: 4 frames x 256-lane workgroups 430 / 648 us forward /
//  backward; 8 frames x 64 lanes 320 / 428 us; 12 frames forward 284 us; dwordx3 / dwordx4 loads of a frame's k values are
//  SLOWER: 352 / 737 us)
constexpr int SRU_THREADS = 64;

// grid = ceil(B*ncols / SRU_THREADS)
__global__ __launch_bounds__(SRU_THREADS) void sru_fwd_kernel(const SruArgs a) {
  const int ncols = a.H * a.dirs;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)a.B * ncols) return;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;                 // reverse direction
  const int T = a.T, k = a.k;
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  float* hb = a.h + (long)b * T * ncols + col;
  float* cb = a.c + (long)b * T * ncols + col;
  float c = 0.f;
  for (int t0 = 0; t0 < T; t0 += SRU_UNROLL_F) {
    float u0[SRU_UNROLL_F], u1[SRU_UNROLL_F], u2[SRU_UNROLL_F], xp[SRU_UNROLL_F];
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_F; ++q) {      // loads of the next frames: independent of c
      const int tt = min(t0 + q, T - 1);
      const int t = flip ? T - 1 - tt : tt;
      const float* u = Ub + (long)t * a.ldu;
      u0[q] = u[0]; u1[q] = u[1]; u2[q] = u[2];
      xp[q] = k == 3 ? xb[(long)t * a.ldx] : u[3];
    }
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_F; ++q) {
      const int tt = t0 + q;
      if (tt < T) {                               // predicated, not a break: the frame loop stays fully unrolled (registers)
        const int t = flip ? T - 1 - tt : tt;
        const float f = 1.f / (1.f + expf(-(u1[q] + bf)));
        const float r = 1.f / (1.f + expf(-(u2[q] + br)));
        c = (c - u0[q]) * f + u0[q];
        const float val = sru_act(c, a.act) * mk;
        hb[(long)t * ncols] = (val - xp[q]) * r + xp[q];
        cb[(long)t * ncols] = c;
      }
    }
  }
}

// same thread mapping, time walked in the reverse of the forward order
__global__ __launch_bounds__(SRU_THREADS) void sru_bwd_kernel(const SruArgs a) {
  const int ncols = a.H * a.dirs;
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)a.B * ncols) return;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;
  const int T = a.T, k = a.k;
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  const float* cb = a.c + (long)b * T * ncols + col;
  const float* dhb = a.dh + (long)b * T * ncols + col;
  const float up_mul = a.up_mul ? a.up_mul[(long)b * ncols + col] : 1.f;
  const float* upb = a.up_add ? a.up_add + (long)b * T * a.ld_up_add + col : nullptr;
  float* dUb = a.dU + (long)b * T * a.ldu + (long)col * k;
  float* dxb = a.dx ? a.dx + (long)b * T * a.lddx + col : nullptr;
  float dc = 0.f, dbf = 0.f, dbr = 0.f;
  // c_{tt-1} of a frame is c_tt of the frame the walk visits next: the cell states are read once -- cc[q + 1] is frame
  // q's predecessor, and the first state of the NEXT block of frames is requested with this block (cc[SRU_UNROLL_B])
  float c_first = cb[(long)(flip ? 0 : T - 1) * ncols];
  for (int s0 = 0; s0 < T; s0 += SRU_UNROLL_B) {
    float u0[SRU_UNROLL_B], u1[SRU_UNROLL_B], u2[SRU_UNROLL_B], xp[SRU_UNROLL_B], cc[SRU_UNROLL_B + 1], dh[SRU_UNROLL_B];
    cc[0] = c_first;
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_B; ++q) {
      const int tt = max(T - 1 - (s0 + q), 0);          // forward-order index, descending
      const int t = flip ? T - 1 - tt : tt;
      const int tp = flip ? t + 1 : t - 1;              // frame of c_{tt-1}
      const float* u = Ub + (long)t * a.ldu;
      u0[q] = u[0]; u1[q] = u[1]; u2[q] = u[2];
      xp[q] = k == 3 ? xb[(long)t * a.ldx] : u[3];
      cc[q + 1] = tt > 0 ? cb[(long)min(max(tp, 0), T - 1) * ncols] : 0.f;
      dh[q] = dhb[(long)t * ncols] * up_mul + (upb ? upb[(long)t * a.ld_up_add] : 0.f);
    }
    c_first = cc[SRU_UNROLL_B];
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_B; ++q) {
      const int tt = T - 1 - (s0 + q);
      if (tt < 0) continue;                       // (the tail of the last block of frames)
      const int t = flip ? T - 1 - tt : tt;
      const float f = 1.f / (1.f + expf(-(u1[q] + bf)));
      const float r = 1.f / (1.f + expf(-(u2[q] + br)));
      const float val = sru_act(cc[q], a.act);
      const float dr = dh[q] * (val * mk - xp[q]);
      const float dxp = dh[q] * (1.f - r);
      const float dct = dc + dh[q] * r * mk * sru_dact(cc[q], val, a.act);
      const float du0 = dct * (1.f - f);
      const float df = dct * (cc[q + 1] - u0[q]);
      dc = dct * f;
      const float du1 = df * f * (1.f - f), du2 = dr * r * (1.f - r);
      float* du = dUb + (long)t * a.ldu;
      du[0] = du0; du[1] = du1; du[2] = du2;
      if (k == 3) dxb[(long)t * a.lddx] = dxp; else du[3] = dxp;
      dbf += du1; dbr += du2;
    }
  }
  a.dbias_part[(long)b * 2 * ncols + col] = dbf;
  a.dbias_part[(long)b * 2 * ncols + ncols + col] = dbr;
}

// The variational input-dropout mask of a layer as multipliers {0, 1/(1-p)}, [B][n]: drawn once per step (Philox, or the
// injected 0/1 mask of the parity hook) and read by the forward pass (dropout kernel below, or the fused dropout + bf16
// cast of GT_OPT_MATMUL_BF16) and by the backward scan of the layer underneath (SruArgs::up_mul).
__global__ void sru_input_mask_kernel(float* __restrict__ mul, int B, int n, float keep_scale, uint32_t thresh, uint32_t key0, uint32_t key1,
                                      const float* __restrict__ inj) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * n) return;
  bool keep;
  if (inj) keep = inj[e] != 0.f;
  else {
    uint32_t r[4];
    philox4x32_10((uint32_t)(e / n), (uint32_t)(e % n), key0, key1, r);
    keep = r[0] >= thresh;
  }
  mul[e] = keep ? keep_scale : 0.f;
}

// variational input dropout for the float32 products: y[row][i] = x[row][i] * mul[b][i]    (mask shared over time)
__global__ void sru_input_dropout_kernel(const float* __restrict__ x, int ldx, float* __restrict__ y, int ldy, int B, int T, int n,
                                         const float* __restrict__ mul) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * T * n) return;
  const int i = (int)(e % n);
  const long row = e / n;
  y[row * ldy + i] = x[row * ldx + i] * mul[(row / T) * n + i];
}

}  // namespace gt
