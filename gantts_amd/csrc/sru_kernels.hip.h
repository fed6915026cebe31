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
#include "fast_math.hip.h"

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
  int seq_mul, seq_add;           // data parallel: local sequence b is sequence seq_add + seq_mul * b of the whole minibatch (1, 0 on one rank)
  const float* mask_buf;          // parity hook: injected 0/1 keep mask [B][ncols] instead of the Philox stream
  // backward only: this layer's output is the NEXT layer's input, and that layer's variational input dropout (+ its k == 3
  // highway gradient) is applied here, where the gradient is read: dh = g * up_mul[b][col] + up_add.  The multiplier is
  // constant per lane (one (sequence, column) pair per lane).
  // backward, GT_OPT_MATMUL_BF16 with loader waves (T % 8 == 0, H % 64 == 0, B * ncols % 64 == 0): dU leaves the scan as the two
  // bf16 images the products read (row-major [N][ld_dub], transposed [ncols*k][ld_dubt]) instead of float32 + a cast pass
  __bf16* dU_b; int ld_dub;
  __bf16* dU_bt; long ld_dubt;
  // forward, GT_OPT_MATMUL_BF16 with the cooperative scans (T % 8 == 0, H % 64 == 0, B * ncols % 64 == 0): the scan writes the bf16
  // images of the NEXT product's input (the next SRU layer's dropped input, or hidden2out's input) itself -- row-major [N][ld_nxb] and,
  // when the backward pass will want it, transposed [ncols][ld_nxbt] -- value h * nx_mul[b][col] rounded as the cast pass rounds it
  __bf16* nx_b; int ld_nxb;
  __bf16* nx_bt; long ld_nxbt;
  const float* nx_mul;            // [B][ncols] multipliers of the next layer's variational input dropout, or null (1)
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
  philox4x32_10((uint32_t)(a.seq_add + a.seq_mul * b), (uint32_t)col, a.key0, a.key1, r);
  return r[0] >= a.thresh ? a.keep_scale : 0.f;
}

// One frame of the recurrence / of its adjoint: ONE definition used by the one-wave and the loader-wave kernels (written with
// explicit fmaf / __fmul_rn so that the two kernels cannot end up with different contractions).
struct SruFwdOut { float c, h; };
// The two sigmoids are fast_sigmoid (fast_math.hip.h: v_exp_f32 + v_rcp_f32, a few ulp): with one or two waves per SIMD the
// scan is bound by the instruction count of a frame, and the library expf was most of it.
// The gates (f, r) of a block of frames do not depend on the carried state: the kernels evaluate them for the whole block first
// (independent transcendental chains the scheduler can interleave), then walk the short dependent chain.
__device__ __forceinline__ SruFwdOut sru_fwd_frame(float u0, float f, float r, float xp, float c_in, float mk, int act) {
  SruFwdOut o;
  o.c = fmaf(c_in - u0, f, u0);
  const float val = __fmul_rn(sru_act(o.c, act), mk);
  o.h = fmaf(val - xp, r, xp);
  return o;
}
struct SruBwdOut { float du0, du1, du2, dxp, dc; };
__device__ __forceinline__ SruBwdOut sru_bwd_frame(float u0, float f, float r, float xp, float c_here, float c_prev, float dh, float dc_in,
                                                   float mk, int act) {
  const float val = sru_act(c_here, act);
  const float dr = __fmul_rn(dh, fmaf(val, mk, -xp));
  SruBwdOut o;
  o.dxp = __fmul_rn(dh, 1.f - r);
  const float dct = fmaf(__fmul_rn(__fmul_rn(dh, r), mk), sru_dact(c_here, val, act), dc_in);
  o.du0 = __fmul_rn(dct, 1.f - f);
  const float df = __fmul_rn(dct, c_prev - u0);
  o.dc = __fmul_rn(dct, f);
  o.du1 = __fmul_rn(__fmul_rn(df, f), 1.f - f);
  o.du2 = __fmul_rn(__fmul_rn(dr, r), 1.f - r);
  return o;
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
    for (int q = 0; q < SRU_UNROLL_F; ++q) { u1[q] = fast_sigmoid(u1[q] + bf); u2[q] = fast_sigmoid(u2[q] + br); }     // f, r
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_F; ++q) {
      const int tt = t0 + q;
      if (tt < T) {                               // predicated, not a break: the frame loop stays fully unrolled (registers)
        const int t = flip ? T - 1 - tt : tt;
        const SruFwdOut o = sru_fwd_frame(u0[q], u1[q], u2[q], xp[q], c, mk, a.act);
        c = o.c;
        hb[(long)t * ncols] = o.h;
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
      dh[q] = fmaf(dhb[(long)t * ncols], up_mul, upb ? upb[(long)t * a.ld_up_add] : 0.f);
    }
    c_first = cc[SRU_UNROLL_B];
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_B; ++q) { u1[q] = fast_sigmoid(u1[q] + bf); u2[q] = fast_sigmoid(u2[q] + br); }     // f, r
#pragma unroll
    for (int q = 0; q < SRU_UNROLL_B; ++q) {
      const int tt = T - 1 - (s0 + q);
      if (tt < 0) continue;                       // (the tail of the last block of frames)
      const int t = flip ? T - 1 - tt : tt;
      const SruBwdOut o = sru_bwd_frame(u0[q], u1[q], u2[q], xp[q], cc[q], cc[q + 1], dh[q], dc, mk, a.act);
      dc = o.dc;
      float* du = dUb + (long)t * a.ldu;
      du[0] = o.du0; du[1] = o.du1; du[2] = o.du2;
      if (k == 3) dxb[(long)t * a.lddx] = o.dxp; else du[3] = o.dxp;
      dbf += o.du1; dbr += o.du2;
    }
  }
  a.dbias_part[(long)b * 2 * ncols + col] = dbf;
  a.dbias_part[(long)b * 2 * ncols + ncols + col] = dbr;
}

// ------------------------------------------------------------------------------------------
// Loader-wave form of the two scans.  The scan has B * ncols independent lanes and nothing more (two waves per CU at
// B = 32, 6 x 512 bidirectional): its HBM rate is the bytes those lanes keep in flight over the load latency, ~3 TB/s with
// 48 / 56 loads per lane.  Here a workgroup is 64 lanes' worth of columns handled by FOUR waves: wave 0 does the recurrence,
// waves 1-3 only load -- loader l fetches frame blocks l, l+3, l+6, ... (FB frames each) into registers and hands them to
// wave 0 through a 3-slot LDS ring, so three blocks per column are in flight instead of one.  Same arithmetic in the same
// order as sru_fwd_kernel / sru_bwd_kernel: results are bit-identical.  One s_barrier per block; the barrier is a bare
// s_barrier behind lgkmcnt(0) (the ring is LDS; __syncthreads() would also drain the loaders' global loads, i.e. undo the
// run-ahead).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sru_ring_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int SRU_LW_THREADS = 256;
constexpr int SRU_LW_FBF = 12, SRU_LW_FBB = 8;      // frames per block, forward / backward
constexpr size_t sru_fwd_lw_lds() { return (size_t)3 * SRU_LW_FBF * 4 * 64 * sizeof(float); }
constexpr size_t sru_bwd_lw_lds() { return (size_t)(3 * SRU_LW_FBB * 7 * 64 + 2 * SRU_LW_FBB * 4 * 64) * sizeof(float); }   // in ring + (B16OUT) out ring

// grid = ceil(B*ncols / 64) workgroups of 256
__global__ __launch_bounds__(SRU_LW_THREADS) void sru_fwd_lw_kernel(const SruArgs a) {
  constexpr int FB = SRU_LW_FBF;
  extern __shared__ __attribute__((aligned(16))) float ring[];      // [3][FB][4][64]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncols = a.H * a.dirs;
  const long gid0 = (long)blockIdx.x * 64 + lane;
  const bool valid = gid0 < (long)a.B * ncols;
  const long gid = valid ? gid0 : 0;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;
  const int T = a.T, k = a.k;
  const int nblk = (T + FB - 1) / FB;
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  if (wave > 0) {
    const int l = wave - 1;
    float v[FB][4];
    auto request = [&](int blk) {
#pragma unroll
      for (int q = 0; q < FB; ++q) {
        const int tt = min(blk * FB + q, T - 1);
        const int t = flip ? T - 1 - tt : tt;
        const float* u = Ub + (long)t * a.ldu;
        v[q][0] = u[0]; v[q][1] = u[1]; v[q][2] = u[2];
        v[q][3] = k == 3 ? xb[(long)t * a.ldx] : u[3];
      }
    };
    auto deposit = [&](int slot) {
      float* r = ring + (size_t)slot * FB * 4 * 64 + lane;
#pragma unroll
      for (int q = 0; q < FB; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) r[(q * 4 + j) * 64] = v[q][j];
    };
    if (l < nblk) request(l);
    if (l == 0) { deposit(0); if (3 < nblk) request(3); }
    for (int i = 0; i < nblk; ++i) {
      sru_ring_barrier();
      const int kb = i + 1;                 // the block the recurrence reads next
      if (kb < nblk && kb % 3 == l) { deposit(l); if (kb + 3 < nblk) request(kb + 3); }
    }
    return;
  }
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  float* hb = a.h + (long)b * T * ncols + col;
  float* cb = a.c + (long)b * T * ncols + col;
  float c = 0.f;
  for (int i = 0; i < nblk; ++i) {
    sru_ring_barrier();
    const float* r = ring + (size_t)(i % 3) * FB * 4 * 64 + lane;
    float fg[FB], rg[FB];
#pragma unroll
    for (int q = 0; q < FB; ++q) { fg[q] = fast_sigmoid(r[(q * 4 + 1) * 64] + bf); rg[q] = fast_sigmoid(r[(q * 4 + 2) * 64] + br); }
#pragma unroll
    for (int q = 0; q < FB; ++q) {
      const int tt = i * FB + q;
      if (tt < T) {
        const int t = flip ? T - 1 - tt : tt;
        const float u0 = r[(q * 4 + 0) * 64], xp = r[(q * 4 + 3) * 64];
        const SruFwdOut o = sru_fwd_frame(u0, fg[q], rg[q], xp, c, mk, a.act);
        c = o.c;
        if (valid) {
          hb[(long)t * ncols] = o.h;
          cb[(long)t * ncols] = c;
        }
      }
    }
  }
}

__device__ __forceinline__ unsigned sru_pack_bf16x2(float lo, float hi) {
  return (unsigned)__builtin_bit_cast(unsigned short, (__bf16)lo) | ((unsigned)__builtin_bit_cast(unsigned short, (__bf16)hi) << 16);
}
// B16OUT: dU as bf16 images.  The recurrence wave leaves a block's dU (8 frames x k values x 64 columns, float32) in a 2-slot LDS
// ring; during the next block the three loader waves -- which have issue slots to spare -- round it to bf16 and write both images
// with 16-byte stores: one chunk of the row-major image (8 consecutive gate columns of one frame) and one of the transposed image
// (8 consecutive frames of one gate column) per lane.  The recurrence wave issues no dU store at all.
template <bool B16OUT>
__global__ __launch_bounds__(SRU_LW_THREADS) void sru_bwd_lw_kernel(const SruArgs a) {
  constexpr int FB = SRU_LW_FBB;
  static_assert(FB == 8, "a transposed-image chunk is 8 frames");
  extern __shared__ __attribute__((aligned(16))) float ring[];      // [3][FB][7][64]: u0, u1, u2, x', c of the predecessor frame, dh, highway gradient
  float* oring = ring + 3 * FB * 7 * 64;                             // B16OUT: [2][FB][4][64] dU of the block just walked
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncols = a.H * a.dirs;
  const long gid0 = (long)blockIdx.x * 64 + lane;
  const bool valid = gid0 < (long)a.B * ncols;
  const long gid = valid ? gid0 : 0;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;
  const int T = a.T, k = a.k;
  const int nblk = (T + FB - 1) / FB;
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  const float* cb = a.c + (long)b * T * ncols + col;
  const float* dhb = a.dh + (long)b * T * ncols + col;
  const float* upb = a.up_add ? a.up_add + (long)b * T * a.ld_up_add + col : nullptr;
  if (wave > 0) {
    const int l = wave - 1;
    float v[FB][7];
    auto request = [&](int blk) {
#pragma unroll
      for (int q = 0; q < FB; ++q) {
        const int tt = max(T - 1 - (blk * FB + q), 0);     // forward-order index, descending
        const int t = flip ? T - 1 - tt : tt;
        const int tp = flip ? t + 1 : t - 1;               // frame of c_{tt-1}
        const float* u = Ub + (long)t * a.ldu;
        v[q][0] = u[0]; v[q][1] = u[1]; v[q][2] = u[2];
        v[q][3] = k == 3 ? xb[(long)t * a.ldx] : u[3];
        v[q][4] = tt > 0 ? cb[(long)min(max(tp, 0), T - 1) * ncols] : 0.f;
        v[q][5] = dhb[(long)t * ncols];
        v[q][6] = upb ? upb[(long)t * a.ld_up_add] : 0.f;
      }
    };
    auto deposit = [&](int slot) {
      float* r = ring + (size_t)slot * FB * 7 * 64 + lane;
#pragma unroll
      for (int q = 0; q < FB; ++q)
#pragma unroll
        for (int j = 0; j < 7; ++j) r[(q * 7 + j) * 64] = v[q][j];
    };
    // B16OUT: block j's dU -> bf16 images.  The workgroup's 64 columns are one sequence's (B * ncols % 64 == 0 and ncols % 64 == 0)
    // and one direction's (H % 64 == 0): col0 = first column, frames t_lo .. t_lo + 7 ascending = walk order (flip) or reversed
    const int col0 = (int)(((long)blockIdx.x * 64) % ncols);
    auto store_block = [&](int j) {
      const float* o = oring + (size_t)(j & 1) * FB * 4 * 64;
      const int t_lo = flip ? FB * j : T - FB * (j + 1);
      const long row0 = (long)b * T + t_lo;
      const int nchunk = FB * 8 * k;                       // 16-byte chunks of either image: 8 frames x (64 k / 8), resp. 64 k gate columns
      for (int c = l * 64 + lane; c < nchunk; c += 192) {
        {   // row-major: frame f (ascending), chunk cc of its 64 k values
          const int f = c / (8 * k), cc = c % (8 * k), q = flip ? f : FB - 1 - f;
          float v[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) { const int idx = 8 * cc + e; v[e] = o[(q * 4 + idx % k) * 64 + idx / k]; }
          uint4 w;
          w.x = sru_pack_bf16x2(v[0], v[1]); w.y = sru_pack_bf16x2(v[2], v[3]); w.z = sru_pack_bf16x2(v[4], v[5]); w.w = sru_pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<uint4*>(a.dU_b + (row0 + f) * a.ld_dub + (long)col0 * k + 8 * cc) = w;
        }
        {   // transposed: gate column gc = c (local column c / k, value c % k), its 8 frames ascending
          const int lc = c / k, jv = c % k;
          float v[8];
#pragma unroll
          for (int f = 0; f < 8; ++f) v[f] = o[((flip ? f : FB - 1 - f) * 4 + jv) * 64 + lc];
          uint4 w;
          w.x = sru_pack_bf16x2(v[0], v[1]); w.y = sru_pack_bf16x2(v[2], v[3]); w.z = sru_pack_bf16x2(v[4], v[5]); w.w = sru_pack_bf16x2(v[6], v[7]);
          *reinterpret_cast<uint4*>(a.dU_bt + ((long)(col0 + lc) * k + jv) * a.ld_dubt + row0) = w;
        }
      }
    };
    if (l < nblk) request(l);
    if (l == 0) { deposit(0); if (3 < nblk) request(3); }
    for (int i = 0; i < nblk; ++i) {
      sru_ring_barrier();
      if (B16OUT && i > 0) store_block(i - 1);
      const int kb = i + 1;
      if (kb < nblk && kb % 3 == l) { deposit(l); if (kb + 3 < nblk) request(kb + 3); }
    }
    if (B16OUT) { sru_ring_barrier(); store_block(nblk - 1); }
    return;
  }
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  const float up_mul = a.up_mul ? a.up_mul[(long)b * ncols + col] : 1.f;
  float* dUb = B16OUT ? nullptr : a.dU + (long)b * T * a.ldu + (long)col * k;
  float* dxb = a.dx ? a.dx + (long)b * T * a.lddx + col : nullptr;
  float dc = 0.f, dbf = 0.f, dbr = 0.f;
  float c_here = cb[(long)(flip ? 0 : T - 1) * ncols];      // cell state of the first frame of the walk
  for (int i = 0; i < nblk; ++i) {
    sru_ring_barrier();
    const float* r = ring + (size_t)(i % 3) * FB * 7 * 64 + lane;
    float fg[FB], rg[FB];
#pragma unroll
    for (int q = 0; q < FB; ++q) { fg[q] = fast_sigmoid(r[(q * 7 + 1) * 64] + bf); rg[q] = fast_sigmoid(r[(q * 7 + 2) * 64] + br); }
#pragma unroll
    for (int q = 0; q < FB; ++q) {
      const int tt = T - 1 - (i * FB + q);
      if (tt < 0) continue;
      const int t = flip ? T - 1 - tt : tt;
      const float u0 = r[(q * 7 + 0) * 64], xp = r[(q * 7 + 3) * 64];
      const float c_prev = r[(q * 7 + 4) * 64];
      const float dh = fmaf(r[(q * 7 + 5) * 64], up_mul, r[(q * 7 + 6) * 64]);
      const SruBwdOut o = sru_bwd_frame(u0, fg[q], rg[q], xp, c_here, c_prev, dh, dc, mk, a.act);
      dc = o.dc;
      if (B16OUT) {
        float* od = oring + ((size_t)(i & 1) * FB + q) * 4 * 64 + lane;
        od[0 * 64] = o.du0; od[1 * 64] = o.du1; od[2 * 64] = o.du2; od[3 * 64] = o.dxp;
        if (k == 3) dxb[(long)t * a.lddx] = o.dxp;
      } else if (valid) {
        float* du = dUb + (long)t * a.ldu;
        du[0] = o.du0; du[1] = o.du1; du[2] = o.du2;
        if (k == 3) dxb[(long)t * a.lddx] = o.dxp; else du[3] = o.dxp;
      }
      dbf += o.du1; dbr += o.du2;
      c_here = c_prev;
    }
  }
  if (B16OUT) sru_ring_barrier();        // the last block's dU is in LDS: the loader waves write it out
  if (valid) {
    a.dbias_part[(long)b * 2 * ncols + col] = dbf;
    a.dbias_part[(long)b * 2 * ncols + ncols + col] = dbr;
  }
}

// The variational input-dropout mask of a layer as multipliers {0, 1/(1-p)}, [B][n]: drawn once per step (Philox, or the
// injected 0/1 mask of the parity hook) and read by the forward pass (dropout kernel below, or the fused dropout + bf16
// cast of GT_OPT_MATMUL_BF16) and by the backward scan of the layer underneath (SruArgs::up_mul).
__global__ void sru_input_mask_kernel(float* __restrict__ mul, int B, int n, float keep_scale, uint32_t thresh, uint32_t key0, uint32_t key1,
                                      const float* __restrict__ inj, int seq_mul, int seq_add) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= B * n) return;
  bool keep;
  if (inj) keep = inj[e] != 0.f;
  else {
    uint32_t r[4];
    philox4x32_10((uint32_t)(seq_add + seq_mul * (e / n)), (uint32_t)(e % n), key0, key1, r);
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
