// Cooperative block scans of the SRU recurrence (reference gantts/models.py:144-167 -> `cuda_functional.SRU`, un-vendored: restated
// recurrence, parity unpinned -- see sru_kernels.hip.h / oracle/gantts_oracle.py).
//
// The loader-wave scans of sru_kernels.hip.h give a workgroup's 64 columns ONE wave that walks the recurrence (three more only load):
// at B = 16 (BASELINE.json configs[3]: B x ncols = 16 384 lanes = one workgroup per CU) that is one wave per CU doing all the
// arithmetic, ~440 cycles per frame, and the scans ran at 2.1 / 2.9 TB/s (0.27 / 0.36 of the HBM peak).  The cell is LINEAR in the
// carried state,
//     forward :  c_t      = f_t c_{t-1} + (1 - f_t) u0_t
//     backward:  dc_{t-1} = f_t (g_t + dc_t),         g_t = dh_t r_t mask act'(c_t)
// so a block of NW x 8 frames is walked by NW waves AT ONCE: wave w loads its own 8 frames (every wave keeps its own loads in
// flight: NW x the bytes per column), scans them from a zero state while recording the prefix products of f, publishes its
// composite (prod f, end state) through LDS -- one barrier per block --, composes the composites of the waves in front of it with
// the state carried from the previous block (NW - 1 fused multiply-adds, every wave redundantly, so the carry never travels), and
// corrects its eight frames by (prefix product) x (incoming state).  Outputs are written by the wave that owns the frames.
// Same recurrence, another association of the products: equal to the sequential scans to rounding (tests: oracle cases at 1e-4, and
// the two forms against each other).  Frames past T act as the identity (f = 1, g = 0).
#pragma once
#include "sru_kernels.hip.h"

namespace gt {

constexpr int SRU_CS_S = 8;       // frames per wave and block

template <int NW> constexpr size_t sru_fwd_cs_lds(bool nxout = false) { return (size_t)(2 * NW * 2 * 64 + (nxout ? NW * SRU_CS_S * 64 : 0)) * sizeof(float); }
template <int NW> constexpr size_t sru_bwd_cs_lds(bool b16out) {
  return (size_t)(2 * NW * 2 * 64 + NW * 2 * 64 + (b16out ? NW * SRU_CS_S * 4 * 64 : 0)) * sizeof(float);
}

// grid = ceil(B * ncols / 64) workgroups of 64 * NW threads
// NXOUT: also write the bf16 images of the next product's input (SruArgs::nx_*): the transposed image straight from the registers (a
// lane's eight frames of its column are 16 contiguous bytes), the row-major image through a wave-private LDS stage -- what
// seqdrop_cast_transpose_kernel / cast_transpose_kernel did in a pass of their own (cfg4: 0.43 ms per step).
template <int NW, bool NXOUT = false>
__global__ __launch_bounds__(64 * NW) void sru_fwd_cs_kernel(const SruArgs a) {
  constexpr int S = SRU_CS_S, FBT = NW * S;
  extern __shared__ __attribute__((aligned(16))) float lds[];      // comp[2][NW][2][64]: (prod f, end state) of wave w, by block parity
  float* ost = lds + 2 * NW * 2 * 64 + (size_t)(threadIdx.x >> 6) * S * 64;      // NXOUT: this wave's [S][64] stage
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncols = a.H * a.dirs;
  const long gid0 = (long)blockIdx.x * 64 + lane;
  const bool valid = gid0 < (long)a.B * ncols;
  const long gid = valid ? gid0 : 0;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;
  const int T = a.T, k = a.k;
  const int nblk = (T + FBT - 1) / FBT;
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  float* hb = a.h + (long)b * T * ncols + col;
  float* cb = a.c + (long)b * T * ncols + col;
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  const float nmul = (NXOUT && a.nx_mul) ? a.nx_mul[(long)b * ncols + col] : 1.f;
  const int col0 = (int)(((long)blockIdx.x * 64) % ncols);           // NXOUT: the workgroup's first column (one sequence, one direction)
  float carry = 0.f;
  float v0[S][4], v1[S][4];
  auto request = [&](float (&v)[S][4], int blk) {
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int tt = min(blk * FBT + wave * S + q, T - 1);
      const int t = flip ? T - 1 - tt : tt;
      const float* u = Ub + (long)t * a.ldu;
      v[q][0] = u[0]; v[q][1] = u[1]; v[q][2] = u[2];
      v[q][3] = k == 3 ? xb[(long)t * a.ldx] : u[3];
    }
  };
  auto block = [&](int i, float (&v)[S][4]) {
    float f[S], r[S], cl[S], P[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const bool in = i * FBT + wave * S + q < T;
      f[q] = in ? fast_sigmoid(v[q][1] + bf) : 1.f;              // (a frame past T is the identity of the recurrence)
      r[q] = fast_sigmoid(v[q][2] + br);
    }
    float c = 0.f, p = 1.f;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      c = fmaf(c - v[q][0], f[q], v[q][0]);
      p = __fmul_rn(p, f[q]);
      cl[q] = c; P[q] = p;
    }
    float* comp = lds + (size_t)(i & 1) * NW * 2 * 64;
    comp[(wave * 2 + 0) * 64 + lane] = p;
    comp[(wave * 2 + 1) * 64 + lane] = c;
    sru_ring_barrier();                                            // LDS only: the run-ahead global loads stay in flight
    float cin = carry, all = carry;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float pw = comp[(w * 2 + 0) * 64 + lane], cw = comp[(w * 2 + 1) * 64 + lane];
      all = fmaf(pw, all, cw);
      if (w + 1 == wave) cin = all;                                // state behind waves 0 .. wave - 1
    }
    if (wave == 0) cin = carry;
    carry = all;
    float hq[S];
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int tt = i * FBT + wave * S + q;
      hq[q] = 0.f;
      if (tt < T && valid) {
        const int t = flip ? T - 1 - tt : tt;
        const float cq = fmaf(P[q], cin, cl[q]);
        const float val = __fmul_rn(sru_act(cq, a.act), mk);
        hq[q] = fmaf(val - v[q][3], r[q], v[q][3]);
        hb[(long)t * ncols] = hq[q];
        cb[(long)t * ncols] = cq;
      }
    }
    if (NXOUT && i * FBT + wave * S < T) {      // (T % 8 == 0: the wave's eight frames are all inside the sequence)
      const int s0 = i * FBT + wave * S;
      const int t_lo = flip ? T - S - s0 : s0;                     // forward walk: tt = s0 + q; t = tt, or T - 1 - tt for the reverse direction
      const long row0 = (long)b * T + t_lo;
      float e[S];
#pragma unroll
      for (int fr = 0; fr < S; ++fr) e[fr] = __fmul_rn(hq[flip ? S - 1 - fr : fr], nmul);      // ascending in t
      if (a.nx_bt) {
        uint4 w;
        w.x = sru_pack_bf16x2(e[0], e[1]); w.y = sru_pack_bf16x2(e[2], e[3]); w.z = sru_pack_bf16x2(e[4], e[5]); w.w = sru_pack_bf16x2(e[6], e[7]);
        *reinterpret_cast<uint4*>(a.nx_bt + (long)col * a.ld_nxbt + row0) = w;
      }
#pragma unroll
      for (int fr = 0; fr < S; ++fr) ost[fr * 64 + lane] = e[fr];
      // (the wave's own LDS operations execute in order: no barrier between its writes above and its reads below)
      const int fr = lane >> 3, cc = lane & 7;                    // 64 chunks of 8 columns: one per lane
      const float* o = ost + fr * 64 + 8 * cc;
      uint4 w;
      w.x = sru_pack_bf16x2(o[0], o[1]); w.y = sru_pack_bf16x2(o[2], o[3]); w.z = sru_pack_bf16x2(o[4], o[5]); w.w = sru_pack_bf16x2(o[6], o[7]);
      *reinterpret_cast<uint4*>(a.nx_b + (row0 + fr) * a.ld_nxb + col0 + 8 * cc) = w;
    }
  };
  request(v0, 0);
  for (int i = 0; i < nblk; i += 2) {
    if (i + 1 < nblk) request(v1, i + 1);
    block(i, v0);
    if (i + 1 < nblk) {
      if (i + 2 < nblk) request(v0, i + 2);
      block(i + 1, v1);
    }
  }
}

// Backward.  Walk position s = 0 .. T-1 visits forward-order index tt = T - 1 - s.  Per frame: u0, u1, u2, x', the cell state of the
// predecessor frame, dh (+ the next layer's input-dropout multiplier and highway gradient); the cell state of a wave's FIRST frame
// is one more load per block.  B16OUT: dU leaves as the two bf16 images the products read -- the transposed image straight from the
// registers (a lane's eight frames of one gate column are 16 contiguous bytes), the row-major image through a wave-private LDS
// stage (the lanes' values of one frame are interleaved k by k).  Needs T % 8 == 0, H % 64 == 0, B * ncols % 64 == 0 (checked by
// the launcher) exactly like sru_bwd_lw_kernel<true>.
template <int NW, bool B16OUT>
__global__ __launch_bounds__(64 * NW) void sru_bwd_cs_kernel(const SruArgs a) {
  constexpr int S = SRU_CS_S, FBT = NW * S;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* dbs = lds + 2 * NW * 2 * 64;                                // [NW][2][64] bias-gradient partials of the waves
  float* ost = dbs + NW * 2 * 64 + (size_t)(threadIdx.x >> 6) * S * 4 * 64;   // B16OUT: this wave's [S][4][64] dU stage
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncols = a.H * a.dirs;
  const long gid0 = (long)blockIdx.x * 64 + lane;
  const bool valid = gid0 < (long)a.B * ncols;
  const long gid = valid ? gid0 : 0;
  const int col = (int)(gid % ncols), b = (int)(gid / ncols);
  const bool flip = col >= a.H;
  const int T = a.T, k = a.k;
  const int nblk = (T + FBT - 1) / FBT;
  const float* Ub = a.U + (long)b * T * a.ldu + (long)col * k;
  const float* xb = a.x + (long)b * T * a.ldx + col;
  const float* cb = a.c + (long)b * T * ncols + col;
  const float* dhb = a.dh + (long)b * T * ncols + col;
  const float* upb = a.up_add ? a.up_add + (long)b * T * a.ld_up_add + col : nullptr;
  const float bf = a.bias[col], br = a.bias[ncols + col];
  const float mk = sru_mask(a, b, col);
  const float up_mul = a.up_mul ? a.up_mul[(long)b * ncols + col] : 1.f;
  float* dUb = B16OUT ? nullptr : a.dU + (long)b * T * a.ldu + (long)col * k;
  float* dxb = a.dx ? a.dx + (long)b * T * a.lddx + col : nullptr;
  const int col0 = (int)(((long)blockIdx.x * 64) % ncols);           // B16OUT: the workgroup's first column (one sequence, one direction)
  float carry = 0.f, dbf = 0.f, dbr = 0.f;
  float v0[S][7], v1[S][7], cf0, cf1;
  auto request = [&](float (&v)[S][7], float& cfirst, int blk) {
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int tt = max(T - 1 - (blk * FBT + wave * S + q), 0);
      const int t = flip ? T - 1 - tt : tt;
      const int tp = flip ? t + 1 : t - 1;                           // frame of c_{tt-1}
      const float* u = Ub + (long)t * a.ldu;
      v[q][0] = u[0]; v[q][1] = u[1]; v[q][2] = u[2];
      v[q][3] = k == 3 ? xb[(long)t * a.ldx] : u[3];
      v[q][4] = tt > 0 ? cb[(long)min(max(tp, 0), T - 1) * ncols] : 0.f;
      v[q][5] = dhb[(long)t * ncols];
      v[q][6] = upb ? upb[(long)t * a.ld_up_add] : 0.f;
      if (q == 0) cfirst = cb[(long)t * ncols];
    }
  };
  auto block = [&](int i, float (&v)[S][7], float cfirst) {
    float f[S], r[S], dl[S], Q[S], gm[S], da[S];
    float c_here = cfirst;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const bool in = i * FBT + wave * S + q < T;
      f[q] = in ? fast_sigmoid(v[q][1] + bf) : 1.f;
      r[q] = fast_sigmoid(v[q][2] + br);
      const float dh = in ? fmaf(v[q][5], up_mul, v[q][6]) : 0.f;
      v[q][5] = dh;
      const float val = sru_act(c_here, a.act);
      gm[q] = __fmul_rn(__fmul_rn(dh, r[q]), mk);
      da[q] = sru_dact(c_here, val, a.act);
      v[q][6] = val;                                                 // (the slot of the highway gradient is free now)
      c_here = v[q][4];
    }
    // local scan from a zero incoming gradient: dl[q] = dct of frame q, Q[q] = product of f over the frames in front of it
    float dc = 0.f, p = 1.f;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      Q[q] = p;
      dl[q] = fmaf(gm[q], da[q], dc);
      dc = __fmul_rn(dl[q], f[q]);
      p = __fmul_rn(p, f[q]);
    }
    float* comp = lds + (size_t)(i & 1) * NW * 2 * 64;
    comp[(wave * 2 + 0) * 64 + lane] = p;
    comp[(wave * 2 + 1) * 64 + lane] = dc;
    sru_ring_barrier();
    float din = carry, all = carry;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const float pw = comp[(w * 2 + 0) * 64 + lane], dw = comp[(w * 2 + 1) * 64 + lane];
      all = fmaf(pw, all, dw);
      if (w + 1 == wave) din = all;
    }
    if (wave == 0) din = carry;
    carry = all;
    float o0[S], o1[S], o2[S], o3[S];
    c_here = cfirst;
#pragma unroll
    for (int q = 0; q < S; ++q) {
      const int tt = T - 1 - (i * FBT + wave * S + q);
      const float dct = fmaf(Q[q], din, dl[q]);
      const float u0 = v[q][0], xp = v[q][3], c_prev = v[q][4], dh = v[q][5], val = v[q][6];
      const float dr = __fmul_rn(dh, fmaf(val, mk, -xp));
      o3[q] = __fmul_rn(dh, 1.f - r[q]);
      o0[q] = __fmul_rn(dct, 1.f - f[q]);
      const float df = __fmul_rn(dct, c_prev - u0);
      o1[q] = __fmul_rn(__fmul_rn(df, f[q]), 1.f - f[q]);
      o2[q] = __fmul_rn(__fmul_rn(dr, r[q]), 1.f - r[q]);
      if (tt >= 0) { dbf += o1[q]; dbr += o2[q]; }
      else { o0[q] = 0.f; o1[q] = 0.f; o2[q] = 0.f; o3[q] = 0.f; }
      c_here = c_prev;
      if (!B16OUT && tt >= 0 && valid) {
        const int t = flip ? T - 1 - tt : tt;
        float* du = dUb + (long)t * a.ldu;
        du[0] = o0[q]; du[1] = o1[q]; du[2] = o2[q];
        if (k == 3) dxb[(long)t * a.lddx] = o3[q]; else du[3] = o3[q];
      }
    }
    if (B16OUT && i * FBT + wave * S < T) {      // (T % 8 == 0: a wave's eight frames are all inside the sequence or all past it)
      // the wave's eight frames: walk positions s0 .. s0 + 7, i.e. t = t_lo .. t_lo + 7 ascending with q (flip) or descending
      const int s0 = i * FBT + wave * S;
      const int t_lo = flip ? s0 : T - S - s0;
      const long row0 = (long)b * T + t_lo;
      if (k == 3) {
#pragma unroll
        for (int q = 0; q < S; ++q) dxb[(long)(flip ? t_lo + q : t_lo + S - 1 - q) * a.lddx] = o3[q];
      }
      // transposed image: gate column (col, jv), eight frames ascending in t = 16 contiguous bytes
#pragma unroll
      for (int jv = 0; jv < 4; ++jv) {
        if (jv < k) {
          float e[S];
#pragma unroll
          for (int fr = 0; fr < S; ++fr) { const int q = flip ? fr : S - 1 - fr; e[fr] = jv == 0 ? o0[q] : jv == 1 ? o1[q] : jv == 2 ? o2[q] : o3[q]; }
          uint4 w;
          w.x = sru_pack_bf16x2(e[0], e[1]); w.y = sru_pack_bf16x2(e[2], e[3]); w.z = sru_pack_bf16x2(e[4], e[5]); w.w = sru_pack_bf16x2(e[6], e[7]);
          *reinterpret_cast<uint4*>(a.dU_bt + ((long)col * k + jv) * a.ld_dubt + row0) = w;
        }
      }
      // row-major image through the wave's LDS stage: [frame ascending][jv][lane]
#pragma unroll
      for (int fr = 0; fr < S; ++fr) {
        const int q = flip ? fr : S - 1 - fr;
        ost[(fr * 4 + 0) * 64 + lane] = o0[q]; ost[(fr * 4 + 1) * 64 + lane] = o1[q]; ost[(fr * 4 + 2) * 64 + lane] = o2[q]; ost[(fr * 4 + 3) * 64 + lane] = o3[q];
      }
      // (the wave's own LDS operations execute in order: no barrier between its writes above and its reads below)
      const int nchunk = S * 8 * k;                       // 16-byte chunks: 8 frames x (64 k values / 8)
      for (int c = lane; c < nchunk; c += 64) {
        const int fr = c / (8 * k), cc = c % (8 * k);
        float e[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) { const int idx = 8 * cc + x; e[x] = ost[(fr * 4 + idx % k) * 64 + idx / k]; }
        uint4 w;
        w.x = sru_pack_bf16x2(e[0], e[1]); w.y = sru_pack_bf16x2(e[2], e[3]); w.z = sru_pack_bf16x2(e[4], e[5]); w.w = sru_pack_bf16x2(e[6], e[7]);
        *reinterpret_cast<uint4*>(a.dU_b + (row0 + fr) * a.ld_dub + (long)col0 * k + 8 * cc) = w;
      }
    }
  };
  request(v0, cf0, 0);
  for (int i = 0; i < nblk; i += 2) {
    if (i + 1 < nblk) request(v1, cf1, i + 1);
    block(i, v0, cf0);
    if (i + 1 < nblk) {
      if (i + 2 < nblk) request(v0, cf0, i + 2);
      block(i + 1, v1, cf1);
    }
  }
  // bias gradients: the waves' partial sums in wave order
  dbs[(wave * 2 + 0) * 64 + lane] = dbf;
  dbs[(wave * 2 + 1) * 64 + lane] = dbr;
  __syncthreads();
  if (wave == 0 && valid) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { s0 += dbs[(w * 2 + 0) * 64 + lane]; s1 += dbs[(w * 2 + 1) * 64 + lane]; }
    a.dbias_part[(long)b * 2 * ncols + col] = s0;
    a.dbias_part[(long)b * 2 * ncols + ncols + col] = s1;
  }
}

}  // namespace gt
