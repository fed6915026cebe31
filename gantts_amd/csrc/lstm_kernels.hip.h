// Recurrent cells of LSTMRNN / GRURNN (reference gantts/models.py:170-213: nn.LSTM over
// pack_padded_sequence, batch_first, optional bidirectional) for gfx950.
//
// Decomposition per layer (both directions together):
//   1. X-projection for ALL frames at once:  Xp = X . W_ih^T          -> f32 MFMA GEMM (gemm_f32.hip.h)
//   2. the recurrence, one launch per time step: gates = Xp[t] + b_ih + b_hh + h_{t-1} . W_hh^T,
//      i,f,o = sigmoid, g = tanh, c_t = f c_{t-1} + i g, h_t = o tanh(c_t)        (this file)
//   3. backward recurrence, one launch per time step (reverse order): dh_t = dOut_t + dG_{t+1} . W_hh,
//      gate derivatives -> dG_t                                                      (this file)
//   4. weight gradients for ALL frames at once: dW_ih = dG^T X, dW_hh = dG^T H_shift, db = colsum(dG),
//      dX = dG . W_ih                                                               -> GEMMs again
// Packed-sequence semantics: sequence b is active at frame t iff t < length[b]; outputs beyond the
// length are zero (pad_packed_sequence) and the reverse direction starts from zero state at each
// sequence's own last valid frame (state is held at zero while inactive).
//
// Step kernels: workgroup = (tile of hidden units, direction, 32 sequences); the per-step product
// [32 x K] . [K x 32] runs on v_mfma_f32_32x32x2_f32 with K split over the 4 waves and staged
// through LDS in chunks of 256 (k-major, pitch 33: conflict-free scatter and fragment reads).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hip.h"

namespace gt {

constexpr int LSTM_KC = 256;      // K chunk staged in LDS
constexpr int LSTM_P = 33;        // LDS pitch
constexpr size_t lstm_lds_bytes() { return (size_t)(2 * LSTM_KC * LSTM_P + 4 * 32 * LSTM_P) * sizeof(float); }

struct LstmStepArgs {
  int B, T, H, dirs;
  int step;                       // launch index s = 0..T-1
  const int* lengths;             // [B] device
  // per direction (index 0 forward, 1 reverse)
  const float* Whh[2];            // (4H, H) row-major
  const float* bih[2];
  const float* bhh[2];
  // frame-major buffers, row = b*T + t
  float* xproj;                   // [N][dirs*4H]  forward: X W_ih^T (in) ; backward: dG (out), same storage
  float* gates;                   // [N][dirs*4H]  post-activation i,f,g,o (stash)
  float* cst;                     // [N][dirs*H]   cell state (stash)
  float* out;                     // [N][dirs*H]   h_t (zero beyond the length)
  const float* dout;              // [N][dirs*H]   upstream gradient w.r.t. out (backward)
  // running state, [dirs][Bpad][H]
  const float* h_prev; const float* c_prev;
  float* h_next; float* c_next;
  float* dc_state;                // backward: running dL/dc, [dirs][Bpad][H]
  int Bpad;
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// Gate stash (post-activation i, f, g, o; read back by the backward recurrence only -- no GEMM sees it): the four gates
// of one (frame, direction, unit) are ONE 16-byte record, [N][dirs][H][4].  (X-projections / dG keep the gate-major
// [N][dirs][4][H] order of W_ih's rows: those are GEMM operands.)
__device__ __forceinline__ long lstm_gate_idx(long row, int ld4, int d, int H, int j) { return row * ld4 + (long)d * 4 * H + 4 * j; }

// Stage a 32-row, k-contiguous panel (row r at src + roff(r), valid iff rvalid(r)) transposed into
// LDS [k][row] (pitch 33): branch-free (clamped address + select), 8 loads in flight per lane,
// 16 B per lane when `vec` (pitch and base 16-byte aligned), else 4 B.
template <typename RowOff, typename RowOk>
__device__ __forceinline__ void lstm_stage_kc(float* sdst, const float* __restrict__ src, int kc, int kmax, bool vec,
                                              RowOff roff, RowOk rvalid) {
  const int tid = threadIdx.x;
  if (vec) {
    constexpr int PER = 32 * LSTM_KC / 4 / 256;            // float4 per thread = 8
    f32x4 v[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 256;
      const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
      const int kg = min(kc + k4, kmax - 4);
      v[q] = *reinterpret_cast<const f32x4*>(src + roff(r) + kg);
    }
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int e = tid + q * 256;
      const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
      const bool ok = rvalid(r) && kc + k4 < kmax;
#pragma unroll
      for (int c = 0; c < 4; ++c) sdst[(k4 + c) * LSTM_P + r] = ok ? v[q][c] : 0.f;
    }
  } else {
    for (int e0 = tid; e0 < 32 * LSTM_KC; e0 += 8 * 256) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * 256;
        const int k = e % LSTM_KC, r = e / LSTM_KC;
        v[q] = src[roff(r) + min(kc + k, kmax - 1)];
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int e = e0 + q * 256;
        const int k = e % LSTM_KC, r = e / LSTM_KC;
        sdst[k * LSTM_P + r] = (rvalid(r) && kc + k < kmax) ? v[q] : 0.f;
      }
    }
  }
}

// acc += sA^T-fragment x sB-fragment over this wave's quarter of the staged chunk
__device__ __forceinline__ void lstm_mma_chunk(f32x16& acc, const float* sA, const float* sB, int wave, int l31, int half) {
  const int k0 = wave * (LSTM_KC / 4);
#pragma unroll 8
  for (int kk = 0; kk < LSTM_KC / 4; kk += 2) {
    const float a = sA[(k0 + kk + half) * LSTM_P + l31];
    const float b = sB[(k0 + kk + half) * LSTM_P + l31];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  }
}

// cross-wave reduction of the 4 partial 32x32 tiles: red[w][row][col]
__device__ __forceinline__ void lstm_store_partial(const f32x16& acc, float* red, int wave, int l31, int half) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * half;
    red[(wave * 32 + row) * LSTM_P + l31] = acc[r];
  }
}

// ------------------------------------------------------------------------------------------
// forward step.  grid = (ceil(H/8), dirs, ceil(B/32)); tile columns c = gate*8 + unit.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_fwd_step_kernel(const LstmStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;                               // h_{t-1}^T  [KC][P]  (k, batch)
  float* sB = sm + LSTM_KC * LSTM_P;            // W_hh tile  [KC][P]  (k, gate*8+unit)
  float* red = sm + 2 * LSTM_KC * LSTM_P;       // [4][32][P]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int H = a.H, T = a.T, d = blockIdx.y;
  const int u0 = blockIdx.x * 8, b0 = blockIdx.z * 32;
  const int t = d == 0 ? a.step : T - 1 - a.step;
  const float* W = a.Whh[d];
  const float* hp = a.h_prev + ((long)d * a.Bpad + b0) * H;

  const bool vec = (H % 4) == 0;     // rows of h / W_hh are then 16-byte aligned
  // The gate stage's inputs (X-projection, biases, c_{t-1}, the length) do not depend on the matrix product:
  // request them first, so that their latency is covered by the staging + MFMA phase instead of following it.
  const int u = tid & 7, bl = tid >> 3;            // one (sequence, unit) pair per thread: u fastest
  const int j = u0 + u, b = b0 + bl;
  const bool mine = j < H && b < a.B;
  const int jc = min(j, H - 1), bc = min(b, a.B - 1);
  const long row = (long)bc * T + t;
  const int ld4 = a.dirs * 4 * H, ld1 = a.dirs * H;
  const long sidx = ((long)d * a.Bpad + bc) * H + jc;
  float xin[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) xin[g] = a.xproj[row * ld4 + d * 4 * H + g * H + jc] + (a.bih[d][g * H + jc] + a.bhh[d][g * H + jc]);
  const float c_in = a.c_prev[sidx];
  const bool active = t < a.lengths[bc];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  for (int kc = 0; kc < H; kc += LSTM_KC) {
    // stage h_{t-1}[b][kc..] and W_hh[gate*H + u0+u][kc..], zero beyond H / unit range
    // (the state arrays are padded to 32 sequences and zero there, so every row is readable)
#ifndef GT_ABLATE_LSTM_NO_STAGE
    auto offA = [&](int r) { return (long)r * H; };
    auto okA = [&](int r) { return true; };
    auto offB = [&](int r) { return ((long)(r >> 3) * H + min(u0 + (r & 7), H - 1)) * H; };
    auto okB = [&](int r) { return u0 + (r & 7) < H; };
    if (vec) {
      // both panels' loads are issued before either panel's LDS stores: one memory latency per chunk, not two
      constexpr int PER = 32 * LSTM_KC / 4 / 256;
      f32x4 va[PER], vb[PER];
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int e = tid + q * 256;
        const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
        const int kg = min(kc + k4, H - 4);
        va[q] = *reinterpret_cast<const f32x4*>(hp + offA(r) + kg);
        vb[q] = *reinterpret_cast<const f32x4*>(W + offB(r) + kg);
      }
#pragma unroll
      for (int q = 0; q < PER; ++q) {
        const int e = tid + q * 256;
        const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
        const bool kin = kc + k4 < H;
        const bool bok = kin && okB(r);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          sA[(k4 + c) * LSTM_P + r] = kin ? va[q][c] : 0.f;
          sB[(k4 + c) * LSTM_P + r] = bok ? vb[q][c] : 0.f;
        }
      }
    } else {
      lstm_stage_kc(sA, hp, kc, H, false, offA, okA);
      lstm_stage_kc(sB, W, kc, H, false, offB, okB);
    }
#endif
    __syncthreads();
#ifndef GT_ABLATE_LSTM_NO_MMA
    lstm_mma_chunk(acc, sA, sB, wave, l31, half);
#endif
    __syncthreads();
  }
  lstm_store_partial(acc, red, wave, l31, half);
  __syncthreads();
#ifdef GT_ABLATE_LSTM_NO_GATES
  if (tid == 0) a.out[blockIdx.x] = red[3];
  return;
#endif
  if (!mine) return;
  float pre[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = g * 8 + u;
    float s = (red[(0 * 32 + bl) * LSTM_P + c] + red[(1 * 32 + bl) * LSTM_P + c]) +
              (red[(2 * 32 + bl) * LSTM_P + c] + red[(3 * 32 + bl) * LSTM_P + c]);
    pre[g] = s + xin[g];
  }
  float ig = 0.f, fg = 0.f, gg = 0.f, og = 0.f, c = 0.f, h = 0.f;
  if (active) {
    ig = sigmoidf_(pre[0]); fg = sigmoidf_(pre[1]); gg = tanhf(pre[2]); og = sigmoidf_(pre[3]);
    c = fg * c_in + ig * gg;
    h = og * tanhf(c);
  }
  *reinterpret_cast<f32x4*>(a.gates + lstm_gate_idx(row, ld4, d, H, j)) = f32x4{ig, fg, gg, og};
  a.cst[row * ld1 + d * H + j] = c;
  a.out[row * ld1 + d * H + j] = h;      // zero beyond the length (pad_packed_sequence)
  a.h_next[sidx] = h;                    // state is held at zero while inactive
  a.c_next[sidx] = c;
}

// ------------------------------------------------------------------------------------------
// backward step.  grid = (ceil(H/32), dirs, ceil(B/32)).  Direction 0 walks t = T-1..0,
// direction 1 walks t = 0..T-1 (the reverse of each direction's forward order).
//   dh = dOut[t] + dG[t_next] . W_hh      (t_next = the step processed by the previous launch)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void lstm_bwd_step_kernel(const LstmStepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sA = sm;                               // dG_prev^T  [KC][P]  (k = gate column, batch)
  float* sB = sm + LSTM_KC * LSTM_P;            // W_hh       [KC][P]  (k, unit)
  float* red = sm + 2 * LSTM_KC * LSTM_P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int H = a.H, T = a.T, d = blockIdx.y;
  const int u0 = blockIdx.x * 32, b0 = blockIdx.z * 32;
  const int t = d == 0 ? T - 1 - a.step : a.step;
  const int tprev = d == 0 ? t + 1 : t - 1;      // frame handled by the previous launch
  const int ld4 = a.dirs * 4 * H, ld1 = a.dirs * H;
  const float* W = a.Whh[d];
  const int K = 4 * H;

  const bool vec = (H % 4) == 0;
  // inputs of the gate-derivative stage (independent of the matrix product): requested up front
  const int pu = tid & 31, pj = min(u0 + pu, H - 1);
  float p_dout[4], p_g[4][4], p_c[4], p_cp[4], p_dcs[4];
  int p_len[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bc = min(b0 + (tid >> 5) + 8 * i, a.B - 1);
    const long row = (long)bc * T + t;
    p_len[i] = a.lengths[bc];
    p_dout[i] = a.dout[row * ld1 + d * H + pj];
    {
      const f32x4 gv = *reinterpret_cast<const f32x4*>(a.gates + lstm_gate_idx(row, ld4, d, H, pj));
#pragma unroll
      for (int g = 0; g < 4; ++g) p_g[i][g] = gv[g];
    }
    p_c[i] = a.cst[row * ld1 + d * H + pj];
    const long rowp = d == 0 ? (t > 0 ? row - 1 : row) : (t + 1 < T ? row + 1 : row);     // clamped; validity checked at use
    p_cp[i] = a.cst[rowp * ld1 + d * H + pj];
    p_dcs[i] = a.dc_state[((long)d * a.Bpad + bc) * H + pj];
  }
  if (a.step > 0) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const float* dGp = a.xproj + (long)tprev * ld4 + d * 4 * H;       // dG of the previous launch's frame
    auto arow = [&](int r) { return (long)min(b0 + r, a.B - 1) * T * ld4; };
    if (vec) {
      // software pipeline over the K chunks: the loads of chunk c+1 (A: dG_prev rows, 16 B/lane;
      // B: W_hh rows of 32 units, 4 B/lane) are in flight while chunk c runs on the matrix pipe
      f32x4 va[8];
      float vb[32];
      auto load_ab = [&](int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = tid + q * 256;
          const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
          va[q] = *reinterpret_cast<const f32x4*>(dGp + arow(r) + min(kc + k4, K - 4));
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int e = tid + q * 256;
          const int n = e & 31, k = e >> 5;
          vb[q] = W[(long)min(kc + k, K - 1) * H + min(u0 + n, H - 1)];
        }
      };
      auto store_ab = [&](int kc) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int e = tid + q * 256;
          const int k4 = (e % (LSTM_KC / 4)) * 4, r = e / (LSTM_KC / 4);
          const bool ok = b0 + r < a.B && kc + k4 < K;
#pragma unroll
          for (int c = 0; c < 4; ++c) sA[(k4 + c) * LSTM_P + r] = ok ? va[q][c] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int e = tid + q * 256;
          const int n = e & 31, k = e >> 5;
          sB[k * LSTM_P + n] = (kc + k < K && u0 + n < H) ? vb[q] : 0.f;
        }
      };
      load_ab(0);
      for (int kc = 0; kc < K; kc += LSTM_KC) {
        store_ab(kc);
        __syncthreads();
        if (kc + LSTM_KC < K) load_ab(kc + LSTM_KC);
        lstm_mma_chunk(acc, sA, sB, wave, l31, half);
        __syncthreads();
      }
    } else {
      for (int kc = 0; kc < K; kc += LSTM_KC) {
        lstm_stage_kc(sA, dGp, kc, K, false, arow, [&](int r) { return b0 + r < a.B; });
        for (int e = tid; e < 32 * LSTM_KC; e += 256) {
          const int n = e & 31, k = e >> 5;
          sB[k * LSTM_P + n] = (kc + k < K && u0 + n < H) ? W[(long)(kc + k) * H + u0 + n] : 0.f;
        }
        __syncthreads();
        lstm_mma_chunk(acc, sA, sB, wave, l31, half);
        __syncthreads();
      }
    }
    lstm_store_partial(acc, red, wave, l31, half);
    __syncthreads();
  }
  // 32 sequences x 32 units per workgroup, 4 pairs per thread, unit fastest
  const int u = tid & 31;
  const int j = u0 + u;
  if (j >= H) return;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int bl = (tid >> 5) + 8 * i;
    const int b = b0 + bl;
    if (b >= a.B) continue;
    const long row = (long)b * T + t;
    const int len = p_len[i];
    const bool active = t < len;
    float dgi = 0.f, dgf = 0.f, dgg = 0.f, dgo = 0.f, dcn = 0.f;
    if (active) {
      float dh = p_dout[i];
      if (a.step > 0)
        dh += (red[(0 * 32 + bl) * LSTM_P + u] + red[(1 * 32 + bl) * LSTM_P + u]) +
              (red[(2 * 32 + bl) * LSTM_P + u] + red[(3 * 32 + bl) * LSTM_P + u]);
      const float ig = p_g[i][0], fg = p_g[i][1], gg = p_g[i][2], og = p_g[i][3];
      const float c = p_c[i];
      float cp = 0.f;                                        // cell state entering this frame
      if (d == 0) { if (t > 0) cp = p_cp[i]; }
      else        { if (t + 1 < len) cp = p_cp[i]; }
      const float tc = tanhf(c);
      const float dc = p_dcs[i] + dh * og * (1.f - tc * tc);
      dgo = dh * tc * (og * (1.f - og));
      dgi = dc * gg * (ig * (1.f - ig));
      dgf = dc * cp * (fg * (1.f - fg));
      dgg = dc * ig * (1.f - gg * gg);
      dcn = dc * fg;
    }
    a.xproj[row * ld4 + d * 4 * H + 0 * H + j] = dgi;      // dG overwrites the X-projection storage
    a.xproj[row * ld4 + d * 4 * H + 1 * H + j] = dgf;
    a.xproj[row * ld4 + d * 4 * H + 2 * H + j] = dgg;
    a.xproj[row * ld4 + d * 4 * H + 3 * H + j] = dgo;
    a.dc_state[((long)d * a.Bpad + b) * H + j] = dcn;
  }
}

// h_shift[row][j] = out[b][t -/+ 1][d*H + j] (the h that entered frame t in direction d), 0 at the start
__global__ void lstm_shift_kernel(const float* __restrict__ out, int ld1, int d, int H, int B, int T,
                                  const int* __restrict__ lengths, float* __restrict__ hs) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * T * H) return;
  const int j = (int)(e % H);
  const long row = e / H;
  const int t = (int)(row % T), b = (int)(row / T);
  float v = 0.f;
  if (d == 0) { if (t > 0) v = out[(row - 1) * ld1 + j]; }
  else        { if (t + 1 < lengths[b]) v = out[(row + 1) * ld1 + H + j]; }
  hs[row * H + j] = v;
}

}  // namespace gt
