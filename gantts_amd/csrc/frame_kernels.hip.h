// HBM-bound per-frame kernels of the G+D step (gfx950): stream gathers, banded MLPG,
// masked losses, the discriminator's sigmoid/BCE head, gradient assembly, fused
// clip-norm + Adagrad/Adam.  All reductions are two-stage and deterministic (no float atomics).
//
// Layout everywhere: frames are rows, features are the fastest dimension ((B,T,D) contiguous,
// reference train.py:145-159), so lane <-> feature gives coalesced rows.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "gemm_f32.hip.h"

namespace gt {

constexpr int RED_THREADS = 256;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block-wide sum (256 threads), result valid in thread 0
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  float r = 0.f;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}
__device__ __forceinline__ double block_sum_d(double v, double* sh) {
  v = wave_sum_d(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) for (int i = 0; i < (int)(blockDim.x >> 6); ++i) r += sh[i];
  return r;
}

__device__ __forceinline__ bool dropout_keep(const DropoutSpec& d, int row, int col) {
  if (d.mode == DROP_PHILOX) return philox_keep_spec(d, row, col);
  if (d.mode == DROP_BUFFER) return d.mask[(long)row * d.ld_mask + col] != 0.f;
  return true;
}

// parity hook (gt_op_philox_mask): materialises the keep bits of one Philox dropout site from the layout-independent
// definition philox_keep(row, col), one element per thread
static __global__ void philox_mask_kernel(const DropoutSpec d, long rows, int cols, float* __restrict__ mask) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const int r = (int)(e / cols), c = (int)(e - (long)r * cols);
  mask[e] = philox_keep_spec(d, r, c) ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------
// device scalars of one step (engine workspace).  Sums are kept in double so the 9 reported
// scalars do not depend on the reduction tree.
// ---------------------------------------------------------------------------------------
struct StepScalars {
  float tv;          // sum(mask)  (valid frames, reference train.py:258,286)
  float inv_tv;
  double tv_sum;     // data parallel, GT_OPT_COMM_TV_IN_SUMS: the local valid-frame count, summed over the ranks TOGETHER with the four
                     // doubles behind it (one message); tv / inv_tv are derived from it when that message has landed
  // additive sums (all-reduced across ranks in data-parallel runs): D step [0..3], G step [4..6]
  double s_real, s_fake;          // sum(log(..)*mask)
  double n_real_ok, n_fake_ok;    // correct counts
  double s_adv;                   // sum(log(D(fake))*mask) of the G step
  double s_mge, s_mse;            // sum of squared masked differences
  double gnorm2_d, gnorm2_g;      // squared grad norms (pre-clip)
};
// results of one update_* call, written by a single thread and copied D2H once
struct StepResults {
  float loss_d, loss_fake_d, loss_real_d, real_correct, fake_correct;   // train.py:278-279 order
  float loss_mse, loss_mge, loss_adv, loss_g;                           // train.py:320 order
  float gnorm_d, gnorm_g, tv;
};
// the results above are in host-visible memory: publish them to a polling host thread (call from the ONE thread that wrote them)
__device__ __forceinline__ void publish_ticket(unsigned* ticket, unsigned value) {
  if (!ticket) return;
  __threadfence_system();
  __hip_atomic_store(ticket, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// sum(mask[0..n)) over one workgroup (valid in thread 0); 8 independent loads in flight per lane (one workgroup is latency-bound)
__device__ __forceinline__ double mask_total_body(const float* __restrict__ mask, int n, double* sh /* [16] */) {
  float p[8];      // 0/1 values: float partial sums of <= 2^24 terms are exact
#pragma unroll
  for (int u = 0; u < 8; ++u) p[u] = 0.f;
  const int bd = blockDim.x;
  int i = threadIdx.x;
  if ((reinterpret_cast<uintptr_t>(mask) & 15) == 0) {      // 16-byte loads, 8 in flight per lane: 32 values per lane and round trip
    const f32x4* m4 = reinterpret_cast<const f32x4*>(mask);
    const int n4 = n >> 2;
    int j = threadIdx.x;
    for (; j + 7 * bd < n4; j += 8 * bd) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = m4[j + u * bd];
#pragma unroll
      for (int u = 0; u < 8; ++u) p[u] += (v[u][0] + v[u][1]) + (v[u][2] + v[u][3]);
    }
    for (; j < n4; j += bd) { const f32x4 v = m4[j]; p[0] += (v[0] + v[1]) + (v[2] + v[3]); }
    i = 4 * n4 + threadIdx.x;       // the tail elements
  }
  for (; i + 7 * bd < n; i += 8 * bd) {
#pragma unroll
    for (int u = 0; u < 8; ++u) p[u] += mask[i + u * bd];
  }
  for (; i < n; i += bd) p[0] += mask[i];
  const double v = (((double)p[0] + (double)p[1]) + ((double)p[2] + (double)p[3])) + (((double)p[4] + (double)p[5]) + ((double)p[6] + (double)p[7]));
  return block_sum_d(v, sh);
}
// tv = sum(mask[0..n)) ; single workgroup (n = B*T is small); any workgroup size that is a multiple of 64
__device__ __forceinline__ void mask_sum_body(const float* __restrict__ mask, int n, float tv_override, const double* __restrict__ tv_dev,
                                              StepScalars* sc, double* sh /* [16] */) {
  const double tot = mask_total_body(mask, n, sh);
  if (threadIdx.x == 0) {
    const float tv = tv_dev ? (float)*tv_dev : (tv_override > 0.f ? tv_override : (float)tot);
    sc->tv = tv;
    sc->inv_tv = 1.0f / tv;
  }
}
static __global__ void mask_sum_kernel(const float* __restrict__ mask, int n, float tv_override, const double* __restrict__ tv_dev,
                                StepScalars* sc) {
  __shared__ double sh[16];
  mask_sum_body(mask, n, tv_override, tv_dev, sc, sh);
}

// out[0] = sum(mask[0..n)) as a double (data parallel: the local term of the global valid-frame count)
static __global__ void mask_total_kernel(const float* __restrict__ mask, int n, double* __restrict__ out) {
  __shared__ double sh[16];
  const double tot = mask_total_body(mask, n, sh);
  if (threadIdx.x == 0) out[0] = tot;
}

// mask[b][t] = t < len[b]   (reference gantts/seqloss.py:9-20)
static __global__ void sequence_mask_kernel(const long* __restrict__ lengths, int B, int T, float* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * T) return;
  const int b = i / T, t = i - b * T;
  mask[i] = (long)t < lengths[b] ? 1.f : 0.f;
}

// out[r*ldo + ooff + j] = in[r*ldi + (idx ? idx[j] : ioff + j)]   -- bit-exact column gather
// (reference gantts/multistream.py:33-79, train.py:232-242,254-256)
static __global__ void gather_cols_kernel(const float* __restrict__ in, int ldi, int ioff, const int* __restrict__ idx,
                                   float* __restrict__ out, int ldo, int ooff, int rows, int nj) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)rows * nj) return;
  int r, j;
  if ((long)rows * nj < (1L << 31)) { r = (int)((unsigned)e / (unsigned)nj); j = (int)((unsigned)e - (unsigned)r * (unsigned)nj); }
  else { r = (int)(e / nj); j = (int)(e - (long)r * nj); }
  const int c = idx ? idx[j] : ioff + j;
  out[(long)r * ldo + ooff + j] = in[(long)r * ldi + c];
}

// D-step input image in one pass: rows [0,N) = [x | fa[:, idx]], rows [N,2N) = [x | fb[:, idx]] (train.py:254-256 for
// the real and the generated half).  x is read once and written twice; one launch instead of four gathers.
static __global__ void build_cat2_kernel(const float* __restrict__ x, int cd, const float* __restrict__ fa, const float* __restrict__ fb,
                                  int ldf, const int* __restrict__ idx, int na, float* __restrict__ out, int ldo, long N) {
  const int w = cd + na;
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= N * w) return;
  long r;
  if (N * w < (1L << 31)) r = (long)((unsigned)e / (unsigned)w); else r = e / w;
  const int c = (int)(e - r * w);
  if (c < cd) {
    const float v = x[r * cd + c];
    out[r * ldo + c] = v;
    out[(N + r) * ldo + c] = v;
  } else {
    const int j = idx[c - cd];
    out[r * ldo + c] = fa[r * ldf + j];
    out[(N + r) * ldo + c] = fb[r * ldf + j];
  }
}

// out[c][r] = in[r][c]  (float32, rows x cols -> cols x rows), 32 x 32 tiles through LDS.  grid = (ceil(cols/32), ceil(rows/32)), 256 threads
static __global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ in, int rows, int cols, int ldi, float* __restrict__ out, int ldo) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    tile[ty + 8 * i][tx] = (r < rows && c < cols) ? in[(long)r * ldi + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = c0 + ty + 8 * i, r = r0 + tx;
    if (c < cols && r < rows) out[(long)c * ldo + r] = tile[tx][ty + 8 * i];
  }
}

// out[r][0..ldo) = in[r][0..cols), zero in the pad columns (row pitch rounded up for 16-byte loads)
// Device-side collate (train.py:139-159 `_pad_2d` + collate_fn, and the length sort of train.py:494-501): the utterances of a
// batch arrive un-padded, back to back ([total frames][D]); output sequence b is the utterance that starts at frame start[b]
// and has len[b] frames (the host gives them in the sorted order), zero-padded to T frames on a row pitch of ld floats.
static __global__ void pad_sequences_kernel(const float* __restrict__ ragged, int D, const long* __restrict__ start, const long* __restrict__ len,
                                            int B, int T, float* __restrict__ out, int ld) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (long)B * T * ld) return;
  const long row = e / ld;
  const int c = (int)(e - row * ld);
  const int b = (int)(row / T), t = (int)(row - (long)b * T);
  out[e] = (c < D && t < len[b]) ? ragged[(start[b] + t) * D + c] : 0.f;
}
// generator with last_sigmoid=True (models.py:141, 213): y_hat = sigmoid(z), so the gradient at y_hat becomes the gradient at z
// in place: g[r][c] *= y[r][c] (1 - y[r][c])
static __global__ void sigmoid_grad_kernel(float* __restrict__ g, int ldg, const float* __restrict__ y, int ldy, long rows, int cols) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const long r = e / cols;
  const int c = (int)(e - r * cols);
  const float s = y[r * ldy + c];
  g[r * ldg + c] *= s * (1.f - s);
}
// The adversarial columns of a pass's rows as one image with a 16-byte row pitch (the split first layer of the conditioned
// discriminator, eng_step.hip): out[r][j] = fa[r][idx[j]] for r < split, fb[r - split][idx[j]] otherwise; pad columns are 0.
// Bit-exact copies (train.py:232-242 select_streams on the static features).
// Rider (tv_mask != null): one extra workgroup at the end of the grid sums the frame mask into the step's normaliser
// (mask_sum_kernel's work: nothing in this launch reads it, the head a whole forward pass later does) -- one launch less per step.
static __global__ void build_adv_kernel(const float* __restrict__ fa, const float* __restrict__ fb, int ldf, const int* __restrict__ idx,
                                        int na, float* __restrict__ out, int ldo, long split, long rows,
                                        const float* __restrict__ tv_mask, int tv_n, float tv_override, StepScalars* sc,
                                        double* __restrict__ tv_total /* data parallel: the local term of the global count instead */) {
  if (tv_mask && blockIdx.x == gridDim.x - 1) {
    __shared__ double sh[16];
    if (tv_total) {
      const double tot = mask_total_body(tv_mask, tv_n, sh);
      if (threadIdx.x == 0) tv_total[0] = tot;
    } else mask_sum_body(tv_mask, tv_n, tv_override, nullptr, sc, sh);
    return;
  }
  // one thread per FOUR output columns (ldo % 4 == 0, `out` 16-byte aligned: a hipMalloc'd image): four independent gathers in flight and one
  // 16-byte store, a quarter of the threads and of the 64-bit divisions (round 5: one element per thread took 10.8 us for 17 MB)
  // the column map goes through the LDS once per workgroup (na <= 256; wider maps read it from memory): the gathers then depend on no
  // other memory access
  __shared__ int sidx[256];
  const bool map_lds = na <= 256;
  if (map_lds) {
    for (int k = threadIdx.x; k < na; k += blockDim.x) sidx[k] = idx[k];
    __syncthreads();
  }
  const long e4 = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int g4 = ldo >> 2;
  if (e4 >= rows * g4) return;
  const long r = e4 / g4;
  const int c = (int)(e4 - r * g4) * 4;
  const float* src = r < split ? fa + r * ldf : fb + (r - split) * ldf;
  int ix[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) ix[k] = map_lds ? sidx[min(c + k, na - 1)] : idx[min(c + k, na - 1)];
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = src[ix[k]];
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = c + k < na ? v[k] : 0.f;
  *reinterpret_cast<f32x4*>(out + r * ldo + c) = v;
}
// [rows][cols] with row pitch ld_in -> the same rows with row pitch ldo (multiple of 4 floats, pad columns 0): the operand image of
// products that read a caller tensor along its rows' direction (n-contiguous operand of a weight gradient)
static __global__ void repitch_kernel(const float* __restrict__ in, int ld_in, int cols, long rows, float* __restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // one 4-float group per thread
  const int g4 = ldo >> 2;
  if (i >= rows * g4) return;
  const long r = i / g4;
  const int c = (int)(i - r * g4) * 4;
  const float* src = in + r * ld_in + c;
  f32x4 v;
#pragma unroll
  for (int k = 0; k < 4; ++k) v[k] = c + k < cols ? src[k] : 0.f;
  *reinterpret_cast<f32x4*>(out + r * ldo + c) = v;
}
static __global__ void pad_rows_kernel(const float* __restrict__ in, int cols, int rows, float* __restrict__ out, int ldo) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)rows * ldo) return;
  const int r = (int)(i / ldo), c = (int)(i - (long)r * ldo);
  out[i] = c < cols ? in[(long)r * cols + c] : 0.f;
}

// ---------------------------------------------------------------------------------------
// MLPG.  The reference multiplies by a dense (T x nW*T) matrix R (nnmnkwii
// unit_variance_mlpg, call sites gantts/multistream.py:120, models.py:66).  R is numerically
// banded; the engine extracts band[t][w][j] = R[t][w*T + t + j - kb] from the caller's dense R
// and verifies that everything outside the band is negligible before using the O(T*kb) form.
// ---------------------------------------------------------------------------------------
// per-offset max |R[t][w*T + t + o]| , o in [-(T-1), T-1]  ->  offmax[o + T - 1]
static __global__ void mlpg_offset_max_kernel(const float* __restrict__ R, int T, int nW, float* __restrict__ offmax) {
  const int o = blockIdx.x - (T - 1);
  __shared__ float sh[16];
  float mx = 0.f;
  for (int i = threadIdx.x; i < T * nW; i += blockDim.x) {
    const int w = i / T, t = i - w * T;
    const int tt = t + o;
    if (tt >= 0 && tt < T) mx = fmaxf(mx, fabsf(R[(long)t * nW * T + (long)w * T + tt]));
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) mx = fmaxf(mx, __shfl_xor(mx, s, 64));
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) sh[wv] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) mx = fmaxf(mx, sh[i]);
    offmax[blockIdx.x] = mx;
  }
}

static __global__ void mlpg_extract_band_kernel(const float* __restrict__ R, int T, int nW, int kb, float* __restrict__ band) {
  const int nb = 2 * kb + 1;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= T * nW * nb) return;
  const int j = i % nb, w = (i / nb) % nW, t = i / (nb * nW);
  const int tt = t + j - kb;
  band[i] = (tt >= 0 && tt < T) ? R[(long)t * nW * T + (long)w * T + tt] : 0.f;
}

// static-column map: for static column c, scol[c] = column of its static component in the full
// (static+delta) layout, sstride[c] = stream's static width (distance between window blocks),
// 0 for a stream without dynamic features (pass-through copy, bit-exact).
constexpr int MLPG_TT = 32;      // output frames per workgroup
constexpr int MLPG_CC = 64;      // static columns per workgroup

// y_static[b][t][c] = sum_w sum_j band[t][w][j] * y[b][t+j-kb][scol[c] + w*sstride[c]]
//
// Workgroup = (sequence b, MLPG_TT = 32 output frames, MLPG_CC = 64 static columns).  LDS holds the
// (T,D) tile with its +-kb halo, [(TT+2kb)][nW][CC], and the TT band rows, zero-padded by
// MLPG_PAD taps on both sides, [TT][nW][nb+2*PAD].  Each lane owns a 2-column x 4-frame register
// block: one ds_read_b64 of data feeds 8 FMAs, the 4 coefficients are wave-broadcast reads
// (0.6 LDS instructions per FMA instead of 2 for the one-output-per-lane form).
constexpr int MLPG_PAD = 3;      // = frames per lane - 1
constexpr int MLPG_MAXW = 4;
constexpr int MLPG_THREADS = 1024; // all 16 waves stage the tile and band rows, waves 0-3 compute

template <int FPL, int TT = MLPG_TT>   // FPL: frames per lane of the compute phase: TT / FPL frame groups x 32 column pairs = TT * 32 / FPL compute threads;
                                       // TT: output frames per workgroup (32, or 64: half the halo re-reads, one workgroup per CU)
__global__ __launch_bounds__(MLPG_THREADS) void mlpg_forward_kernel(
    const float* __restrict__ y, int ldy, const float* __restrict__ band, int kb, int nW,
    const int* __restrict__ scol, const int* __restrict__ sstride, int Ds,
    float* __restrict__ ys, int ldys, int B, int T) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * kb + 1, nbp = nb + 2 * MLPG_PAD;
  const int tiles_t = (T + TT - 1) / TT;
  const int b = blockIdx.x / tiles_t, t0 = (blockIdx.x % tiles_t) * TT;
  const int c0 = blockIdx.y * MLPG_CC;
  const int nc = min(MLPG_CC, Ds - c0);
  const int rows = TT + 2 * kb;
  float* sb = sm + rows * nW * MLPG_CC;            // [TT][nW][nbp]
  const float* yb = y + (long)b * T * ldy;
  // Staging is one wave per LDS row (64 lanes = the 64 columns of a data row / the taps of a band row), rows strided over
  // the 16 waves, (row, window) advanced incrementally: no integer division per element, 8 independent loads in flight.
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = MLPG_THREADS / 64;
  // The first batch of band rows (taps jp = lane of rows wv, wv + 16, ...) is requested FIRST and parked in registers: it depends on
  // nothing, so it travels with the column-map loads and the first batch of tile rows instead of being a fourth dependent round trip
  // behind them (round 5).
  float vb0[8];
  {
    const long src_rows = (long)T * nW;
    const int jc = min(max(lane - MLPG_PAD, 0), nb - 1);
#pragma unroll
    for (int q = 0; q < 8; ++q) vb0[q] = band[min((long)t0 * nW + wv + q * nwv, src_rows - 1) * nb + jc];
  }
  {  // data tile: LDS row rw = r * nW + w holds frame t0 - kb + r, window w
    const bool c_ok = lane < nc;
    const int my_col = c_ok ? scol[c0 + lane] : 0, my_st = c_ok ? sstride[c0 + lane] : 0;
    const int nrw = rows * nW;
    const int dr = nwv / nW, dw = nwv % nW;
    int r = wv / nW, w = wv % nW;
    for (int rw0 = wv; rw0 < nrw; rw0 += 8 * nwv) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = t0 - kb + r;
        const int tc = min(max(t, 0), T - 1);
        const float x = yb[(long)tc * ldy + my_col + (my_st > 0 ? w * my_st : 0)];
        v[q] = (c_ok && t >= 0 && t < T && (my_st > 0 || w == 0)) ? x : 0.f;
        r += dr; w += dw;
        if (w >= nW) { w -= nW; ++r; }
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int rw = rw0 + q * nwv;
        if (rw < nrw) sm[rw * MLPG_CC + lane] = v[q];
      }
    }
  }
  {  // band rows: LDS row tw = tl * nW + w  <-  band row t0 * nW + tw (the band is [t][w][nb], so rows are consecutive)
    const int nrow = TT * nW;
    const long src_rows = (long)T * nW;
    if (lane < nbp) {       // the batch requested up front
      const int j = lane - MLPG_PAD;
      const bool j_ok = j >= 0 && j < nb;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int tw = wv + q * nwv;
        if (tw < nrow) sb[tw * nbp + lane] = (j_ok && (long)t0 * nW + tw < src_rows) ? vb0[q] : 0.f;
      }
    }
    for (int jp = lane; jp < nbp; jp += 64) {      // one pass for half-widths up to 28 (nbp <= 64 taps)
      const int j = jp - MLPG_PAD;
      const bool j_ok = j >= 0 && j < nb;
      const int jc = min(max(j, 0), nb - 1);
      for (int tw0 = jp == lane ? wv + 8 * nwv : wv; tw0 < nrow; tw0 += 8 * nwv) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const long row = (long)t0 * nW + tw0 + q * nwv;
          const float x = band[min(row, src_rows - 1) * nb + jc];
          v[q] = (j_ok && row < src_rows) ? x : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int tw = tw0 + q * nwv;
          if (tw < nrow) sb[tw * nbp + jp] = v[q];
        }
      }
    }
  }
  __syncthreads();
  static_assert(FPL >= 1 && FPL <= MLPG_PAD + 1, "the band rows are padded for at most MLPG_PAD + 1 frames per lane");
  if (threadIdx.x >= TT * 32 / FPL) return;  // every wave stages (memory-level parallelism); 16 / FPL of them compute
  const int cp = threadIdx.x & 31, fg = threadIdx.x >> 5;     // column pair, frame group (FPL frames)
  const int tl0 = fg * FPL;
  float acc[FPL][2];
#pragma unroll
  for (int i = 0; i < FPL; ++i) acc[i][0] = acc[i][1] = 0.f;
  // The tap loop runs on a counter that is the same in every lane (the staged row is tl0 + j0) and is unrolled by eight: the LDS reads of
  // eight taps are in flight before the first FMA waits (round 5: written over r = tl0 .. the compiler kept a lane-dependent loop with one
  // s_waitcnt lgkmcnt(0) per tap; 20.3 -> 18.9 us at cfg2).  Same products in the same order.
  const int ntap = (FPL - 1) + nb;
  for (int w = 0; w < nW; ++w) {
    const float* dcol = sm + (tl0 * nW + w) * MLPG_CC + 2 * cp;      // + j0*nW*CC
    const float* cf = sb + (tl0 * nW + w) * nbp + MLPG_PAD;           // + i*nW*nbp + (j0 - i)
#pragma unroll 8
    for (int j0 = 0; j0 < ntap; ++j0) {
      const float2 d = *reinterpret_cast<const float2*>(dcol + j0 * nW * MLPG_CC);
#pragma unroll
      for (int i = 0; i < FPL; ++i) {
        const float cfi = cf[i * nW * nbp + j0 - i];
        acc[i][0] = fmaf(cfi, d.x, acc[i][0]);
        acc[i][1] = fmaf(cfi, d.y, acc[i][1]);
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = 2 * cp + q;
    if (c >= nc) continue;
    const bool pass = sstride[c0 + c] == 0;
#pragma unroll
    for (int i = 0; i < FPL; ++i) {
      const int t = t0 + tl0 + i;
      if (t >= T) continue;
      const float out = pass ? sm[((tl0 + i + kb) * nW + 0) * MLPG_CC + c] : acc[i][q];   // pass-through: bit-exact copy
      ys[((long)b * T + t) * ldys + c0 + c] = out;
    }
  }
}

// transpose of the above:  gy[b][t'][scol[c]+w*st] = sum_t band[t][w][t'-t+kb] * gs[b][t][c]
// plus the masked-MSE gradient in the static+delta domain when mse_w != 0:
//   gy += mse_w * 2 * (yhat*m - y*m) * m / Tv        (reference gantts/seqloss.py:41-43)
// LDS: gs tile [(TT+2kb)][CC] + the band rows of the same frames, padded, [(TT+2kb)][nW][nb+2*PAD].
// Lane = 2 columns x 4 frames x all windows: one ds_read_b64 of gs feeds 8*nW FMAs.
template <int FPL, int TT = MLPG_TT>
__global__ __launch_bounds__(MLPG_THREADS) __attribute__((amdgpu_waves_per_eu(FPL <= 2 ? 8 : 4)))      // two 16-wave workgroups per CU: <= 64 VGPRs AND <= 96 SGPRs
void mlpg_backward_kernel(
    const float* __restrict__ gs, int ldgs, const float* __restrict__ band, int kb, int nW,
    const int* __restrict__ scol, const int* __restrict__ sstride, int Ds,
    float* __restrict__ gy, int ldgy, int B, int T,
    float mse_w, const float* __restrict__ yhat, const float* __restrict__ ytgt, int ldt,
    const float* __restrict__ mask, const StepScalars* __restrict__ sc) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int nb = 2 * kb + 1, nbp = nb + 2 * MLPG_PAD;
  const int tiles_t = (T + TT - 1) / TT;
  const int b = blockIdx.x / tiles_t, t0 = (blockIdx.x % tiles_t) * TT;
  const int c0 = blockIdx.y * MLPG_CC;
  const int nc = min(MLPG_CC, Ds - c0);
  const int rows = TT + 2 * kb;
  float* sb = sm + rows * MLPG_CC;                 // [rows][nW][nbp]; frames outside [0,T) are zero
  const float* gb = gs + (long)b * T * ldgs;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = MLPG_THREADS / 64;
  float vb0[16];            // the first TWO batches of band rows (all of them at TT = 32: (TT + 2 kb) nW / 16 <= 16 rows per wave), requested first
  {
    const long src_rows = (long)T * nW, base = (long)(t0 - kb) * nW;
    const int jc = min(max(lane - MLPG_PAD, 0), nb - 1);
#pragma unroll
    for (int q = 0; q < 16; ++q) vb0[q] = band[min(max(base + wv + q * nwv, 0L), src_rows - 1) * nb + jc];
  }
  {  // gradient tile: one wave per frame row (see the forward kernel's staging)
    const bool c_ok = lane < nc;
    const int ccl = c_ok ? c0 + lane : c0;
    for (int r0 = wv; r0 < rows; r0 += 8 * nwv) {
      float v[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int t = t0 - kb + r0 + q * nwv;
        const int tc = min(max(t, 0), T - 1);
        const float x = gb[(long)tc * ldgs + ccl];
        v[q] = (c_ok && t >= 0 && t < T) ? x : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int r = r0 + q * nwv;
        if (r < rows) sm[r * MLPG_CC + lane] = v[q];
      }
    }
  }
  {  // band rows of the staged frames: LDS row rw = r * nW + w  <-  band row (t0 - kb) * nW + rw; frames outside [0,T) are zero
    const int nrow = rows * nW;
    const long src_rows = (long)T * nW, base = (long)(t0 - kb) * nW;
    if (lane < nbp) {       // the batch requested up front
      const int j = lane - MLPG_PAD;
      const bool j_ok = j >= 0 && j < nb;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int rw = wv + q * nwv;
        const long row = base + rw;
        if (rw < nrow) sb[rw * nbp + lane] = (j_ok && row >= 0 && row < src_rows) ? vb0[q] : 0.f;
      }
    }
    for (int jp = lane; jp < nbp; jp += 64) {
      const int j = jp - MLPG_PAD;
      const bool j_ok = j >= 0 && j < nb;
      const int jc = min(max(j, 0), nb - 1);
      for (int rw0 = jp == lane ? wv + 16 * nwv : wv; rw0 < nrow; rw0 += 8 * nwv) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const long row = base + rw0 + q * nwv;
          const float x = band[min(max(row, 0L), src_rows - 1) * nb + jc];
          v[q] = (j_ok && row >= 0 && row < src_rows) ? x : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const int rw = rw0 + q * nwv;
          if (rw < nrow) sb[rw * nbp + jp] = v[q];
        }
      }
    }
  }
  __syncthreads();
  static_assert(FPL >= 1 && FPL <= MLPG_PAD + 1, "the band rows are padded for at most MLPG_PAD + 1 frames per lane");
  if (threadIdx.x >= TT * 32 / FPL) return;
  const int cp = threadIdx.x & 31, fg = threadIdx.x >> 5;
  const int tl0 = fg * FPL;
  float acc[MLPG_MAXW][FPL][2];
#pragma unroll
  for (int w = 0; w < MLPG_MAXW; ++w)
#pragma unroll
    for (int i = 0; i < FPL; ++i) acc[w][i][0] = acc[w][i][1] = 0.f;
  // staged row r holds frame t = t0 - kb + r; it reaches output frame tl (t' = t0 + tl) with
  // q = r - tl in [0, nb) through the coefficient band[t][w][nb - 1 - q]
  const int ntap = (FPL - 1) + nb;       // (uniform counter + unroll: see the forward kernel; 21.9 -> 21.1 us)
  const float* drow = sm + tl0 * MLPG_CC + 2 * cp;
  const float* cfrow = sb + (long)tl0 * nW * nbp + MLPG_PAD + (nb - 1);
#pragma unroll 4
  for (int j0 = 0; j0 < ntap; ++j0) {
    const float2 d = *reinterpret_cast<const float2*>(drow + j0 * MLPG_CC);
    const float* cf = cfrow + (long)j0 * nW * nbp - j0;   // + w*nbp + i
#pragma unroll
    for (int w = 0; w < MLPG_MAXW; ++w) {
      if (w >= nW) break;
#pragma unroll
      for (int i = 0; i < FPL; ++i) {
        const float cfi = cf[w * nbp + i];
        acc[w][i][0] = fmaf(cfi, d.x, acc[w][i][0]);
        acc[w][i][1] = fmaf(cfi, d.y, acc[w][i][1]);
      }
    }
  }
  const float msk_scale = mse_w != 0.f ? 2.f * mse_w * sc->inv_tv : 0.f;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = 2 * cp + q;
    if (c >= nc) continue;
    const int col0 = scol[c0 + c], st = sstride[c0 + c];
#pragma unroll
    for (int i = 0; i < FPL; ++i) {
      const int tp = t0 + tl0 + i;
      if (tp >= T) continue;
      const long row = (long)b * T + tp;
      const float m = msk_scale != 0.f ? mask[row] : 0.f;
#pragma unroll
      for (int w = 0; w < MLPG_MAXW; ++w) {
        if (w >= nW || (st == 0 && w > 0)) break;
        float out = st == 0 ? sm[(tl0 + i + kb) * MLPG_CC + c] : acc[w][i][q];
        const int col = col0 + w * st;
        if (msk_scale != 0.f) out += msk_scale * (yhat[row * ldt + col] * m - ytgt[row * ldt + col] * m) * m;
        gy[row * ldgy + col] = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------
// masked squared error: partial[blk] = sum_{rows of blk} sum_d (a*m - b*m)^2   (double)
// optional gradient out: g[r][d] = gscale * 2 * (a*m - b*m) * m * inv_tv
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ double masked_sqerr_body(
    const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
    const float* __restrict__ mask, long rows, int D, float* __restrict__ g, int ldg, float gs, int blk, int nblk) {
  double acc = 0.0;
  const long total = rows * D;
  // (row, column) advance incrementally with the grid stride: no 64-bit division per element
  const long stride = (long)nblk * blockDim.x;
  const long sr = stride / D;
  const int sd = (int)(stride - sr * D);
  long e = (long)blk * blockDim.x + threadIdx.x;
  long r = e / D;
  int d = (int)(e - r * D);
  // FOUR grid strides per trip (round 5): the twelve loads of four elements are in flight together -- a thread of the cfg2 launch walks
  // ~12 elements, and one dependent memory round trip per element made the kernel latency-bound (g_losses 11.3 us for 33 MB).  The
  // squares are added in the order the one-element loop adds them.
  // (the last trip is predicated, not a one-element tail loop: a thread with 11 elements takes three round trips like one with 12, not five)
  for (; e < total; e += 4 * stride) {
    long rr[4];
    int dd[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      rr[u] = r; dd[u] = d; ok[u] = e + u * stride < total;
      r += sr; d += sd;
      if (d >= D) { d -= D; ++r; }
    }
    float m[4], av[4], bv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      m[u] = ok[u] ? mask[rr[u]] : 0.f; av[u] = ok[u] ? a[rr[u] * lda + dd[u]] : 0.f; bv[u] = ok[u] ? b[rr[u] * ldb + dd[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      const float diff = av[u] * m[u] - bv[u] * m[u];
      acc += (double)diff * (double)diff;
      if (g) g[rr[u] * ldg + dd[u]] = gs * diff * m[u];
    }
  }
  return acc;
}
static __global__ __launch_bounds__(RED_THREADS) void masked_sqerr_kernel(
    const float* __restrict__ a, int lda, const float* __restrict__ b, int ldb,
    const float* __restrict__ mask, long rows, int D, double* __restrict__ partial,
    float* __restrict__ g, int ldg, float gscale, const StepScalars* __restrict__ sc) {
  __shared__ double sh[16];
  const float gs = g ? 2.f * gscale * sc->inv_tv : 0.f;
  const double acc = masked_sqerr_body(a, lda, b, ldb, mask, rows, D, g, ldg, gs, blockIdx.x, gridDim.x);
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}
// The two reported sums of squares of a generator step in ONE launch (train.py:291-294): workgroups [0, n1) the masked squared error
// of (a1, b1) -> partial1 (loss_mse over y_hat), workgroups [n1, gridDim) that of (a2, b2) -> partial2 (loss_mge over the static
// features).  Both only need the batch, so the launch goes first in the step and nothing waits for it.
static __global__ __launch_bounds__(RED_THREADS) void g_losses_kernel(
    const float* __restrict__ a1, int lda1, const float* __restrict__ b1, int ldb1, int D1, int n1, double* __restrict__ partial1,
    const float* __restrict__ a2, int lda2, const float* __restrict__ b2, int ldb2, int D2, double* __restrict__ partial2,
    const float* __restrict__ mask, long rows) {
  __shared__ double sh[16];
  const bool first = (int)blockIdx.x < n1;
  const int blk = first ? blockIdx.x : blockIdx.x - n1, nblk = first ? n1 : gridDim.x - n1;
  const double acc = first ? masked_sqerr_body(a1, lda1, b1, ldb1, mask, rows, D1, nullptr, 0, 0.f, blk, nblk)
                           : masked_sqerr_body(a2, lda2, b2, ldb2, mask, rows, D2, nullptr, 0, 0.f, blk, nblk);
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) (first ? partial1 : partial2)[blk] = tot;
}

// ---------------------------------------------------------------------------------------
// Discriminator head: last_linear (out_dim 1) + sigmoid + BCE terms + backward seed, fused.
// rows [0, n_real) are "natural" frames, rows [n_real, n_rows) are generated frames
// (reference train.py:261-271 for the D step; :307-308 for the adversarial term of the G step
//  where every row is a generated frame scored against the "natural" label).
// One wave per row, lane <-> hidden unit (coalesced), 4 rows per workgroup iteration.
//   z = b + <h, w>;  D = sigmoid(z)
//   real:  loss -= log(D + eps) * m / Tv          dD = -m / Tv / (D + eps)
//   fake:  loss -= log(1 - D + eps) * m / Tv      dD = +m / Tv / ((1 - D) + eps)
//   dz = dD * D * (1 - D);  dH[r][k] = dz * w[k] * f'(h[r][k]);  dw[k] += dz * h[r][k];  db += dz
// ---------------------------------------------------------------------------------------
enum HeadMode { HEAD_D_STEP = 0, HEAD_G_ADV = 1 };

struct HeadPartials {   // one per workgroup; summed in fixed order afterwards
  double s_real, s_fake, n_real_ok, n_fake_ok, db;
};

// KP = hidden units per lane (K <= 64*KP): the row, the weight vector and the dw accumulators live
// in registers; the next row is prefetched while the current one goes through the reduction chain.
// TH: storage type of the activation (float, or __bf16 with GT_OPT_MATMUL_BF16: the bf16 image the last hidden layer wrote)
// VEC (float32 storage, KP % 4 == 0): lane <-> FOUR consecutive hidden units (k = 4 lane + j, + 256 per further group of four): the
// row and the seed gradient move as 16-byte accesses instead of four 4-byte ones per lane.  Philox bits are keyed by the unit's index,
// not by the lane that holds it: same masks.
template <int KP, typename TH = float, bool B16OUT = false, bool VEC = false>
__global__ __launch_bounds__(256) void d_head_kernel(
    const TH* __restrict__ H, int ldh, int K, const float* __restrict__ w, const float* __restrict__ bias,
    const float* __restrict__ mask, int n_mask, int n_real, int n_rows, int mode, float eps,
    float* __restrict__ Dout, float* __restrict__ dH, int lddh, int want_grad, DropoutSpec drop,
    int has_act, const StepScalars* __restrict__ sc,
    HeadPartials* __restrict__ hp, float* __restrict__ dw_partial /* [grid][K] */,
    __bf16* __restrict__ dHb = nullptr, int lddhb = 0,        // bf16 image of dH (GT_OPT_MATMUL_BF16), instead of / beside dH
    __bf16* __restrict__ dHbT = nullptr, long lddhbt = 0,     // and its transposed twin [K][rows]: 4 consecutive rows per 8-byte store
    const double* __restrict__ tv_dev = nullptr,              // data parallel: the all-reduced valid-frame count, not yet in *sc (workgroup 0 puts it there)
    int unit_tv = 0) {                                        // GT_OPT_COMM_TV_IN_SUMS: the count is not known yet -- unnormalised losses' gradient (the optimizer applies 1 / Tv)
  extern __shared__ __attribute__((aligned(16))) float smf[];   // [4][K] dw staging
  __shared__ double shd[5][4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const float inv_tv = unit_tv ? 1.0f : tv_dev ? 1.0f / (float)*tv_dev : sc->inv_tv;
  if (tv_dev && blockIdx.x == 0 && threadIdx.x == 0) {
    StepScalars* scw = const_cast<StepScalars*>(sc);
    scw->tv = (float)*tv_dev; scw->inv_tv = 1.0f / (float)*tv_dev;
  }
  const float b0 = bias[0];
  double s_real = 0, s_fake = 0, n_rok = 0, n_fok = 0, dbs = 0;
  float wreg[KP], dwacc[KP];
  int kidx[KP], kof[KP];
#pragma unroll
  for (int j = 0; j < KP; ++j) {
    const int k = VEC ? 4 * lane + (j & 3) + 256 * (j >> 2) : lane + 64 * j;
    kof[j] = k;
    kidx[j] = min(k, K - 1);                       // clamped: loads stay in bounds, extra lanes use w = 0
    wreg[j] = k < K ? w[k] : 0.f;
    dwacc[j] = 0.f;
  }
  // A wave owns ITEMS of 8 rows: the rows of a 16-row group that share one Philox call per column
  // (rows 16g + 8q + 4h + s, q in {0,1}, s in 0..3; gemm_f32.hip.h: philox_keep) -- one call per lane and
  // column slot instead of one per element.  All 8 rows of an item are requested up front.
  const int n_items = ((n_rows + 15) / 16) * 2;
  for (int item = blockIdx.x * 4 + wv; item < n_items; item += gridDim.x * 4) {
    const int g = item >> 1, h = item & 1;
    float hrow[8][KP];
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) {
      const int r = min(16 * g + 8 * (ri >> 2) + 4 * h + (ri & 3), n_rows - 1);
      if constexpr (VEC) {
#pragma unroll
        for (int jj = 0; jj < KP / 4; ++jj) {
          if (kof[4 * jj] + 3 < K) {
            const f32x4 v = ld4u(reinterpret_cast<const float*>(H) + (long)r * ldh + kof[4 * jj]);
#pragma unroll
            for (int q = 0; q < 4; ++q) hrow[ri][4 * jj + q] = v[q];
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) hrow[ri][4 * jj + q] = (float)H[(long)r * ldh + kidx[4 * jj + q]];
          }
        }
      } else {
#pragma unroll
        for (int j = 0; j < KP; ++j) hrow[ri][j] = (float)H[(long)r * ldh + kidx[j]];
      }
    }
    float outv[B16OUT ? 8 : 1][KP];      // B16OUT: the item's dH values, kept for the transposed 8-byte stores below
#pragma unroll
    for (int ri = 0; ri < (B16OUT ? 8 : 1); ++ri)
#pragma unroll
      for (int j = 0; j < KP; ++j) outv[ri][j] = 0.f;
    uint32_t keepb[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
      keepb[j] = 0xffu;
      if (want_grad && has_act && drop.mode == DROP_PHILOX) {
        uint32_t rnd[4];
        philox4x32_10(2u * philox_group(drop, (uint32_t)g) + (uint32_t)h, (uint32_t)kidx[j], drop.key0, drop.key1, rnd);
        uint32_t bits = 0;
#pragma unroll
        for (int p = 0; p < 8; ++p) bits |= (philox_piece(rnd, p) >= drop.thresh ? 1u : 0u) << p;
        keepb[j] = bits;
      }
    }
#pragma unroll
    for (int ri = 0; ri < 8; ++ri) {
      const int r = 16 * g + 8 * (ri >> 2) + 4 * h + (ri & 3);
      if (r >= n_rows) continue;                   // wave-uniform
      float part = 0.f;
#pragma unroll
      for (int j = 0; j < KP; ++j) part = fmaf(hrow[ri][j], wreg[j], part);
      const float z = wave_sum(part) + b0;
      const float D = 1.f / (1.f + expf(-z));
      const float m = mask[r % n_mask];
      const bool is_real = (mode == HEAD_G_ADV) || (r < n_real);
      float dD;
      if (is_real) {
        const float l = logf(D + eps) * m;
        if (lane == 0) s_real += (double)l;
        if (lane == 0 && mode == HEAD_D_STEP) n_rok += (D > 0.5f ? 1.0 : 0.0) * (double)m;
        dD = -m * inv_tv / (D + eps);
      } else {
        const float om = (1.f - D) + eps;
        const float l = logf(om) * m;
        if (lane == 0) { s_fake += (double)l; n_fok += (D < 0.5f ? 1.0 : 0.0) * (double)m; }
        dD = m * inv_tv / om;
      }
      if (lane == 0 && Dout) Dout[r] = D;
      if (want_grad) {
        const float dz = dD * ((1.f - D) * D);
        if (lane == 0) dbs += (double)dz;
        float vals[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
          const int k = kof[j];
          const float hv = hrow[ri][j];
          dwacc[j] = fmaf(dz, hv, dwacc[j]);
          float f = 1.f;
          if (has_act) {
            bool keep = ((keepb[j] >> ri) & 1u) != 0u;
            if (drop.mode == DROP_BUFFER) keep = drop.mask[(long)r * drop.ld_mask + kidx[j]] != 0.f;
            f = leaky_drop_grad(hv, keep, drop.mode == DROP_NONE ? 1.f : drop.scale);
          }
          const float val = dz * wreg[j] * f;
          vals[j] = val;
          if (B16OUT) outv[ri][j] = val;
          if (!VEC && k < K) {
            if (dH) dH[(long)r * lddh + k] = val;
            if (B16OUT && dHb) dHb[(long)r * lddhb + k] = (__bf16)val;
          }
        }
        if constexpr (VEC) {
          if (dH) {
#pragma unroll
            for (int jj = 0; jj < KP / 4; ++jj) {
              float* dst = dH + (long)r * lddh + kof[4 * jj];
              if (kof[4 * jj] + 3 < K) {
                f32x4 v;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = vals[4 * jj + q];
                st4u(dst, v);
              } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) if (kof[4 * jj + q] < K) dst[q] = vals[4 * jj + q];
              }
            }
          }
        }
      }
    }
    if (B16OUT && want_grad && dHbT) {     // the item's rows are two runs of 4 consecutive rows: 16g + 4h + {0..3} and 16g + 8 + 4h + {0..3}
#pragma unroll
      for (int j = 0; j < KP; ++j) {
        const int k = kof[j];
        if (k >= K) continue;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int rs = 16 * g + 8 * q + 4 * h;
          __bf16* dst = dHbT + (long)k * lddhbt + rs;
          if (rs + 3 < n_rows) {
            bf16x4 pk;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) pk[s4] = (__bf16)outv[B16OUT ? 4 * q + s4 : 0][j];
            *reinterpret_cast<bf16x4*>(dst) = pk;
          } else {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) if (rs + s4 < n_rows) dst[s4] = (__bf16)outv[B16OUT ? 4 * q + s4 : 0][j];
          }
        }
      }
    }
  }
  // reduce the scalar partials over the 4 waves
  if (lane == 0) { shd[0][wv] = s_real; shd[1][wv] = s_fake; shd[2][wv] = n_rok; shd[3][wv] = n_fok; shd[4][wv] = dbs; }
  if (want_grad && dw_partial) {
#pragma unroll
    for (int j = 0; j < KP; ++j) { const int k = kof[j]; if (k < K) smf[wv * K + k] = dwacc[j]; }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    HeadPartials p;
    p.s_real = shd[0][0] + shd[0][1] + shd[0][2] + shd[0][3];
    p.s_fake = shd[1][0] + shd[1][1] + shd[1][2] + shd[1][3];
    p.n_real_ok = shd[2][0] + shd[2][1] + shd[2][2] + shd[2][3];
    p.n_fake_ok = shd[3][0] + shd[3][1] + shd[3][2] + shd[3][3];
    p.db = shd[4][0] + shd[4][1] + shd[4][2] + shd[4][3];
    hp[blockIdx.x] = p;
  }
  if (want_grad && dw_partial) {
    for (int k = threadIdx.x; k < K; k += blockDim.x)
      dw_partial[(long)blockIdx.x * K + k] = (smf[k] + smf[K + k]) + (smf[2 * K + k] + smf[3 * K + k]);
  }
}

// finalize head: sums HeadPartials in a fixed order into StepScalars, writes dw/db of last_linear.
// grid = ceil(K/64) workgroups of 1024 threads (16 row-parts x 64 columns); workgroup 0 also
// reduces the scalar partials.
static __global__ __launch_bounds__(1024) void d_head_finalize_kernel(const HeadPartials* __restrict__ hp, const float* __restrict__ dw_partial,
                                                               int nblk, int K, int mode, StepScalars* sc,
                                                               float* __restrict__ dw, float* __restrict__ db, int accumulate,
                                                               StepResults* early_res /* D step: also finalize_d (gnorm 0) */,
                                                               unsigned* ticket = nullptr, unsigned ticket_value = 0,
                                                               int cgw = 64 /* columns per workgroup: 64, or 16 for many partials (grid = ceil(K / cgw)) */,
                                                               int scal_blk = 0 /* the workgroup that sums the scalars: 0, or one extra workgroup behind the dw ones */) {
  __shared__ float shw[1024];                      // [1024 / cgw row parts][cgw columns]
  __shared__ double shd[16];
  // cgw = 16 (round 5; the fused stack leaves 1024 rows of partials at cfg2): 64 row parts x 16 columns per workgroup -- a thread's sixteen
  // loads cover ITS share of 1024 rows in one round trip and the 1 MB is read by 16 CUs instead of four (four dependent trips each)
  const int nparts = (int)blockDim.x / cgw;
  const int kl = threadIdx.x % cgw, part = threadIdx.x / cgw;
  const int k = blockIdx.x * cgw + kl;
  if (dw) {
    float sa[16];                                  // 16 independent loads in flight per thread (latency-bound otherwise)
#pragma unroll
    for (int u = 0; u < 16; ++u) sa[u] = 0.f;
    if (k < K) {
      int i = part;
      for (; i + 15 * nparts < nblk; i += 16 * nparts) {
#pragma unroll
        for (int u = 0; u < 16; ++u) sa[u] += dw_partial[(long)(i + nparts * u) * K + k];
      }
      for (; i < nblk; i += nparts) sa[0] += dw_partial[(long)i * K + k];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) sa[u] += sa[u + 8];
#pragma unroll
    for (int u = 0; u < 4; ++u) sa[u] += sa[u + 4];
    shw[part * cgw + kl] = (sa[0] + sa[1]) + (sa[2] + sa[3]);
    __syncthreads();
    if (part == 0 && k < K) {
      float tot = 0.f;
      for (int q = 0; q < nparts; ++q) tot += shw[q * cgw + kl];
      dw[k] = accumulate ? dw[k] + tot : tot;
    }
  }
  if ((int)blockIdx.x == scal_blk) {
    double v[5] = {0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblk; i += blockDim.x) {
      v[0] += hp[i].s_real; v[1] += hp[i].s_fake; v[2] += hp[i].n_real_ok; v[3] += hp[i].n_fake_ok; v[4] += hp[i].db;
    }
    double r[5];
    for (int q = 0; q < 5; ++q) r[q] = block_sum_d(v[q], shd);
    if (threadIdx.x == 0) {
      if (mode == HEAD_D_STEP) {
        sc->s_real = r[0]; sc->s_fake = r[1]; sc->n_real_ok = r[2]; sc->n_fake_ok = r[3];
        if (early_res) {       // same arithmetic as finalize_d_kernel, one launch less on the way to the host
          const float T = sc->tv;
          const float lr = -((float)r[0]) / T, lf = -((float)r[1]) / T;
          early_res->loss_real_d = lr; early_res->loss_fake_d = lf; early_res->loss_d = lr + lf;
          early_res->real_correct = (float)r[2]; early_res->fake_correct = (float)r[3];
          early_res->gnorm_d = 0.f; early_res->tv = T;
          publish_ticket(ticket, ticket_value);
        }
      } else sc->s_adv = r[0];
      if (db) db[0] = accumulate ? db[0] + (float)r[4] : (float)r[4];
    }
  }
}


// ---------------------------------------------------------------------------------------
// generator-side gradient assembly at y_hat_static (the "linearity trick"):
//   gs[n][c] = mge_w * 2 * (yhs*m - ys*m) * m / Tv                         (train.py:291,314)
//            + leak[n][j]          (dloss_d/dy_hat_static, OLD D weights;   train.py:265,274)
//            + adv_w * gadv[n][j]  (dloss_adv/dy_hat_static, NEW D weights; train.py:307-308,314)
// for c = adv_cols[j]; also produces the MGE loss partials.
// ---------------------------------------------------------------------------------------
// Rider of static_grad_kernel (fused single-GPU generator step): the step's scalar finalisation -- what d_head_finalize_kernel and
// finalize_g_kernel would do in two more launches -- runs as one extra workgroup at the end of the grid.  Everything it reads
// was produced by EARLIER launches (the head's per-workgroup partials, g_losses_kernel's partial sums), nothing by this one.
struct GFinalize {
  int on;
  StepScalars* sc;
  StepResults* out;
  float adv_w, mse_w, mge_w;
  int has_adv;
  const double* part_mge; int n_mge;
  const double* part_mse; int n_mse;
  const HeadPartials* hp; int n_hp;
  unsigned* ticket; unsigned ticket_value;
};
__device__ __forceinline__ void finalize_g_body(StepScalars* sc, StepResults* out, float adv_w, float mse_w, float mge_w, int has_adv,
                                                int zero_gnorm, const double* __restrict__ part_mge, int n_mge,
                                                const double* __restrict__ part_mse, int n_mse,
                                                const HeadPartials* __restrict__ hp, int n_hp, double* shp /* [16] */,
                                                unsigned* ticket = nullptr, unsigned ticket_value = 0) {
  if (part_mge || part_mse || hp) {
    if (part_mge) {
      double v = 0.0;
      for (int i = threadIdx.x; i < n_mge; i += blockDim.x) v += __builtin_nontemporal_load(part_mge + i);
      const double t = block_sum_d(v, shp);
      if (threadIdx.x == 0) sc->s_mge = t;
    }
    if (part_mse) {
      double v = 0.0;
      for (int i = threadIdx.x; i < n_mse; i += blockDim.x) v += __builtin_nontemporal_load(part_mse + i);
      const double t = block_sum_d(v, shp);
      if (threadIdx.x == 0) sc->s_mse = t;
    }
    if (hp) {          // the adversarial term's sum of log D(G(x)) over the head's workgroups (d_head_finalize_kernel, HEAD_G_ADV)
      double v = 0.0;
      for (int i = threadIdx.x; i < n_hp; i += blockDim.x) v += hp[i].s_real;
      const double t = block_sum_d(v, shp);
      if (threadIdx.x == 0) sc->s_adv = t;
    }
    __syncthreads();
  }
  if (threadIdx.x || !out) return;       // out == null (data parallel): the sums only -- they are all-reduced before anything is reported
  const float T = sc->tv;
  const float mse = (float)sc->s_mse / T, mge = (float)sc->s_mge / T;
  const float adv = has_adv ? -((float)sc->s_adv) / T : 0.f;
  out->loss_mse = mse; out->loss_mge = mge; out->loss_adv = adv;
  out->loss_g = (mse_w * mse + mge_w * mge) + adv_w * adv;
  out->gnorm_g = zero_gnorm ? 0.f : (float)sqrt(sc->gnorm2_g); out->tv = T;
  publish_ticket(ticket, ticket_value);
}

// partial == null: no sum of squares (g_losses_kernel produced it)
static __global__ __launch_bounds__(RED_THREADS) void static_grad_kernel(
    const float* __restrict__ yhs, int ld1, const float* __restrict__ ys, int ld2,
    const float* __restrict__ mask, long rows, int Ds, float mge_w,
    const int* __restrict__ adv_inv /* [Ds] -> j or -1 */, const float* __restrict__ leak, int ldl,
    const float* __restrict__ gadv, int lda, float adv_w,
    float* __restrict__ gs, int ldg, double* __restrict__ partial, StepScalars* __restrict__ sc, const GFinalize fin,
    int leak_unnorm /* the kept dloss_d/dy_hat_static is that of the UNNORMALISED loss (GT_OPT_COMM_TV_IN_SUMS): x 1 / Tv here */) {
  __shared__ double sh[16];
  int nblk = gridDim.x;
  const int blk = blockIdx.x;
  if (fin.on) {
    if (blockIdx.x == gridDim.x - 1) {
      finalize_g_body(fin.sc, fin.out, fin.adv_w, fin.mse_w, fin.mge_w, fin.has_adv, 1, fin.part_mge, fin.n_mge, fin.part_mse, fin.n_mse,
                      fin.hp, fin.n_hp, sh, fin.ticket, fin.ticket_value);
      return;
    }
    --nblk;
  }
  double acc = 0.0;
  const long total = rows * Ds;
  const float sc2 = 2.f * mge_w * sc->inv_tv;
  const float leak_s = leak_unnorm ? sc->inv_tv : 1.f;
  const long stride = (long)nblk * blockDim.x;
  const long sr = stride / Ds;
  const int sd = (int)(stride - sr * Ds);
  long e = (long)blk * blockDim.x + threadIdx.x;
  long r = e / Ds;
  int c = (int)(e - r * Ds);
  // FOUR grid strides per trip (a thread of the cfg2 launch walks four elements): their loads -- mask, the two features, the column map,
  // then the two kept gradients -- are in flight together instead of one dependent round trip per element (round 5)
  for (; e + 3 * stride < total; e += 4 * stride) {
    long rr[4];
    int cc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (c >= Ds) { c -= Ds; ++r; }
      rr[u] = r; cc[u] = c;
      r += sr; c += sd;
    }
    float m[4], yv[4], tv[4], lk[4], ga[4];
    int jj[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      m[u] = mask[rr[u]]; yv[u] = yhs[rr[u] * ld1 + cc[u]]; tv[u] = ys[rr[u] * ld2 + cc[u]];
      jj[u] = (gs && adv_inv) ? adv_inv[cc[u]] : -1;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lk[u] = (jj[u] >= 0 && leak) ? leak[rr[u] * ldl + jj[u]] : 0.f;
      ga[u] = (jj[u] >= 0 && gadv) ? gadv[rr[u] * lda + jj[u]] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const float diff = yv[u] * m[u] - tv[u] * m[u];
      if (partial) acc += (double)diff * (double)diff;
      if (gs) {
        float v = sc2 * diff * m[u];
        if (jj[u] >= 0) {
          if (leak) v += leak_s * lk[u];
          if (gadv) v += adv_w * ga[u];
        }
        gs[rr[u] * ldg + cc[u]] = v;
      }
    }
  }
  for (; e < total; e += stride, r += sr, c += sd) {
    if (c >= Ds) { c -= Ds; ++r; }
    const float m = mask[r];
    const float diff = yhs[r * ld1 + c] * m - ys[r * ld2 + c] * m;
    if (partial) acc += (double)diff * (double)diff;
    if (gs) {
      float v = sc2 * diff * m;
      const int j = adv_inv ? adv_inv[c] : -1;
      if (j >= 0) {
        if (leak) v += leak_s * leak[r * ldl + j];
        if (gadv) v += adv_w * gadv[r * lda + j];
      }
      gs[r * ldg + c] = v;
    }
  }
  if (partial) {
    const double tot = block_sum_d(acc, sh);
    if (threadIdx.x == 0) partial[blk] = tot;
  }
}

static __global__ __launch_bounds__(RED_THREADS) void finalize_g_rider_kernel(const GFinalize fin) {
  __shared__ double sh[16];
  finalize_g_body(fin.sc, fin.out, fin.adv_w, fin.mse_w, fin.mge_w, fin.has_adv, 1, fin.part_mge, fin.n_mge, fin.part_mse, fin.n_mse,
                  fin.hp, fin.n_hp, sh, fin.ticket, fin.ticket_value);
}

// g[i] *= 1 / Tv  (the rare consumers of an unnormalised kept gradient outside the fused step)
static __global__ void scale_by_inv_tv_kernel(float* __restrict__ g, long n, const StepScalars* __restrict__ sc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] *= sc->inv_tv;
}

static __global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int n, double* __restrict__ out) {
  __shared__ double sh[16];
  double v = 0.0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) v += partial[i];
  const double tot = block_sum_d(v, sh);
  if (threadIdx.x == 0) *out = tot;
}

// column sums of a frame matrix (bias gradient when no weight gradient is requested):
// workgroup = 64 columns x rows_per_blk rows, 4 row lanes; partial[blk_r][c] then a fixed-order finalize
static __global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X, int ldx, long rows, int cols,
                                                             int rows_per_blk, float* __restrict__ partial) {
  __shared__ float sh[4][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.y * 64 + cl;
  const long r0 = (long)blockIdx.x * rows_per_blk;
  const long r1 = min(rows, r0 + rows_per_blk);
  float s = 0.f;
  if (c < cols) for (long r = r0 + rl; r < r1; r += 4) s += X[r * ldx + c];
  sh[rl][cl] = s;
  __syncthreads();
  if (rl == 0 && c < cols) partial[(long)blockIdx.x * cols + c] = (sh[0][cl] + sh[1][cl]) + (sh[2][cl] + sh[3][cl]);
}
static __global__ void colsum_finalize_kernel(const float* __restrict__ partial, int nblk, int cols, float* __restrict__ out, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float s = 0.f;
  for (int i = 0; i < nblk; ++i) s += partial[(long)i * cols + c];
  out[c] = accumulate ? out[c] + s : s;
}

// dW = (accumulate ? dW : 0) + sum_s slab[s]   (deterministic split-K combine of the TN GEMM)
// 16 B per lane, 8 slabs in flight per lane; slab_stride and n4*4 must keep 16-byte alignment.
__device__ __forceinline__ void slab_reduce4_body(int blk, const float* __restrict__ slabs, long slab_stride, int nslab, long n4,
                                                  float* __restrict__ out, int accumulate,
                                                  const float* __restrict__ bslabs, int nb, float* __restrict__ bout, int main_blocks) {
  if (blk >= main_blocks) {
    // trailing workgroups: the bias-gradient slabs [nslab][nb] ride along in the same launch
    const int c = (blk - main_blocks) * blockDim.x + threadIdx.x;
    if (c >= nb) return;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int k = 0;
    for (; k + 4 <= nslab; k += 4) {
      s0 += bslabs[(long)k * nb + c];       s1 += bslabs[(long)(k + 1) * nb + c];
      s2 += bslabs[(long)(k + 2) * nb + c]; s3 += bslabs[(long)(k + 3) * nb + c];
    }
    for (; k < nslab; ++k) s0 += bslabs[(long)k * nb + c];
    const float tot = (s0 + s1) + (s2 + s3);
    bout[c] = accumulate ? bout[c] + tot : tot;
    return;
  }
  const long i = (long)blk * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const f32x4* p = reinterpret_cast<const f32x4*>(slabs) + i;
  const long st4 = slab_stride / 4;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  int k = 0;
  for (; k + 8 <= nslab; k += 8) {
    f32x4 v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[(long)(k + q) * st4];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += v[q];
  }
  for (; k < nslab; ++k) s += p[(long)k * st4];
  f32x4* o = reinterpret_cast<f32x4*>(out) + i;
  if (accumulate) s += *o;
  *o = s;
}
// Several combines in ONE launch (the fused single-GPU step defers the combines of a network's layers to just before its
// optimizer step: one launch instead of one per layer).  Same arithmetic per element as slab_reduce4_kernel.
constexpr int SLAB_MAX_JOBS = 8;
struct SlabJob {
  const float* slabs; long slab_stride; long n4; float* out;
  const float* bslabs; float* bout;
  int nslab, accumulate, nb, main_blocks, block0, pad_;
};
struct SlabJobs { int n, pad_; SlabJob j[SLAB_MAX_JOBS]; };
static __global__ __launch_bounds__(256) void slab_reduce_multi_kernel(const SlabJobs jobs) {
  int q = 0;
  while (q + 1 < jobs.n && (int)blockIdx.x >= jobs.j[q + 1].block0) ++q;
  const SlabJob& J = jobs.j[q];
  slab_reduce4_body((int)blockIdx.x - J.block0, J.slabs, J.slab_stride, J.nslab, J.n4, J.out, J.accumulate, J.bslabs, J.nb, J.bout, J.main_blocks);
}
static __global__ __launch_bounds__(256) void slab_reduce4_kernel(const float* __restrict__ slabs, long slab_stride, int nslab, long n4,
                                                           float* __restrict__ out, int accumulate,
                                                           const float* __restrict__ bslabs, int nb, float* __restrict__ bout, int main_blocks) {
  slab_reduce4_body((int)blockIdx.x, slabs, slab_stride, nslab, n4, out, accumulate, bslabs, nb, bout, main_blocks);
}
// small-n variant (bias gradients): 64 columns x 16 slab lanes per workgroup, fixed-order combine
static __global__ __launch_bounds__(1024) void slab_reduce_small_kernel(const float* __restrict__ slabs, long slab_stride, int nslab, int n,
                                                                 float* __restrict__ out, int accumulate) {
  __shared__ float sh[16][64];
  const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float s0 = 0.f, s1 = 0.f;
  if (c < n) {
    int k = part;
    for (; k + 16 < nslab; k += 32) { s0 += slabs[(long)k * slab_stride + c]; s1 += slabs[(long)(k + 16) * slab_stride + c]; }
    for (; k < nslab; k += 16) s0 += slabs[(long)k * slab_stride + c];
  }
  sh[part][cl] = s0 + s1;
  __syncthreads();
  if (part == 0 && c < n) {
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) tot += sh[q][cl];
    out[c] = accumulate ? out[c] + tot : tot;
  }
}
static __global__ void slab_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int nslab, long n,
                                   float* __restrict__ out, int accumulate) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < nslab; ++k) s += slabs[(long)k * slab_stride + i];
  out[i] = accumulate ? out[i] + s : s;
}

// ---------------------------------------------------------------------------------------
// clip_grad_norm_(params, 1.0) + optimizer step, fused over the flat parameter buffer
// (reference train.py:275-276, 317-318; torch.optim.Adagrad / Adam update rules)
// ---------------------------------------------------------------------------------------
static __global__ __launch_bounds__(RED_THREADS) void sqnorm_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ partial) {
  __shared__ double sh[16];
  double acc = 0.0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const double v = (double)g[i];
    acc += v * v;
  }
  const double tot = block_sum_d(acc, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

struct OptimSpec {
  int kind;            // 0 Adagrad, 1 Adam
  float lr, weight_decay, eps;
  float lr_decay;      // Adagrad
  float beta1, beta2;  // Adam
  long step;           // 1-based step count of THIS update
  float max_norm;      // clip threshold (1.0 in the reference); <= 0 disables clipping
};

// one element of clip_grad_norm_ + Adagrad / Adam (torch's update rules; `coef` = the clip coefficient, clr / step_size / bc2_sqrt
// the step's scalar factors).  The clipped gradient is written back (clip_grad_norm_ scales .grad in place).
__device__ __forceinline__ void optim_update(float& p, float& g, float& s0, float* s1, float coef, const OptimSpec& o, float clr, float step_size,
                                             float bc2_sqrt) {
  float gi = g * coef;
  g = gi;
  const float pi = p;
  if (o.weight_decay != 0.f) gi = fmaf(o.weight_decay, pi, gi);
  if (o.kind == 0) {
    const float s = fmaf(gi, gi, s0);
    s0 = s;
    p = pi - clr * (gi / (sqrtf(s) + o.eps));
  } else {
    const float m = o.beta1 * s0 + (1.f - o.beta1) * gi;
    const float v = o.beta2 * *s1 + (1.f - o.beta2) * gi * gi;
    s0 = m; *s1 = v;
    const float denom = sqrtf(v) / bc2_sqrt + o.eps;
    p = pi - step_size * (m / denom);
  }
}

static __global__ __launch_bounds__(RED_THREADS) void optim_step_kernel(
    float* __restrict__ p, float* __restrict__ g, float* __restrict__ s0, float* __restrict__ s1, long n,
    const double* __restrict__ norm_partial, int n_partial, double* __restrict__ norm2_out, OptimSpec o,
    const unsigned int* __restrict__ fault_dev, unsigned int* fault_host /* pinned, or null */,
    unsigned int* skipped_host /* pinned, or null */,
    const float* __restrict__ gscale = nullptr /* the gradient in g is that of a loss still to be multiplied by *gscale (1 / Tv) */) {
  __shared__ float coef_sh;
  __shared__ double shn[16];
  // A persistent launch of this step that gave up raised the device fault word: its gradients are garbage.  Mirror the
  // word to the host (no copy launch) and leave parameters, gradients and optimizer state untouched; the skipped step
  // is counted so that gt_clear_faults can take it back out of the host's step counter.
  if (fault_dev && *fault_dev) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (fault_host) *fault_host = *fault_dev;
      if (skipped_host) *skipped_host += 1u;
      if (norm2_out) *norm2_out = __longlong_as_double(0x7ff8000000000000LL);   // no update, no norm: NaN, not the previous step's value
    }
    return;
  }
  double part = 0.0;
  {
    const int bd = blockDim.x;
    int i = threadIdx.x;
    for (; i + 3 * bd < n_partial; i += 4 * bd) {      // four partials in flight per thread, added in index order as before
      const double a0 = norm_partial[i], a1 = norm_partial[i + bd], a2 = norm_partial[i + 2 * bd], a3 = norm_partial[i + 3 * bd];
      part += a0; part += a1; part += a2; part += a3;
    }
    for (; i < n_partial; i += bd) part += norm_partial[i];
  }
  double tot = block_sum_d(part, shn);      // same fixed order in every workgroup
  if (threadIdx.x == 0) {
    const float gsc = gscale ? *gscale : 1.f;
    tot *= (double)gsc * (double)gsc;
    if (blockIdx.x == 0 && norm2_out) *norm2_out = tot;
    float coef = 1.f;
    if (o.max_norm > 0.f) {
      const float total_norm = (float)sqrt(tot);
      coef = fminf(o.max_norm / (total_norm + 1e-6f), 1.f);
    }
    coef_sh = coef * gsc;       // (the scaled, clipped gradient is what optim_update writes back)
  }
  __syncthreads();
  const float coef = coef_sh;
  float clr = o.lr, bc2_sqrt = 1.f, step_size = o.lr;
  if (o.kind == 0) {
    clr = o.lr / (1.f + (float)(o.step - 1) * o.lr_decay);
  } else {
    const double bc1 = 1.0 - pow((double)o.beta1, (double)o.step);
    const double bc2 = 1.0 - pow((double)o.beta2, (double)o.step);
    step_size = (float)((double)o.lr / bc1);
    bc2_sqrt = (float)sqrt(bc2);
  }
  // four grid strides per trip: the loads of four elements are in flight together (a thread of the cfg2 generator's launch walks 3-4)
  const long gstride = (long)gridDim.x * blockDim.x;
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i < n; i += 4 * gstride) {      // (the last trip is predicated: no one-element tail)
    float pv[4], gv[4], sv[4], mv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long k = i + u * gstride;
      const bool ok = k < n;
      pv[u] = ok ? p[k] : 0.f; gv[u] = ok ? g[k] : 0.f; sv[u] = ok ? s0[k] : 0.f; mv[u] = (ok && s1) ? s1[k] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long k = i + u * gstride;
      if (k >= n) continue;
      optim_update(pv[u], gv[u], sv[u], s1 ? &mv[u] : nullptr, coef, o, clr, step_size, bc2_sqrt);
      p[k] = pv[u]; g[k] = gv[u]; s0[k] = sv[u];
      if (s1) s1[k] = mv[u];
    }
  }
}

// ---------------------------------------------------------------------------------------
// The fused single-GPU step's closing launch of a network: weight-gradient combines (the recorded slab jobs) + squared norm +
// clip + optimizer step in ONE launch instead of three (slab_reduce_multi, sqnorm_partial, optim_step).  Every workgroup
// owns the same elements in both phases -- it sums their slabs, writes the gradient and keeps its share of the squared norm;
// a device-wide barrier (one arrival counter, monotonically increasing across launches; every wait bounded by a wall-clock
// timeout that raises the engine's fault word and skips the update) makes the total known to all; then it clips and steps
// what it owns.  Only the per-workgroup norm partials cross workgroups -- written, counted and read with RELAXED agent-scope
// atomics, which act at the device's coherence point; no release / acquire fence is needed (and none is used: a release at
// agent scope writes back the XCD's whole dirty L2, measured at +50 us per launch), because no gradient is read through
// another workgroup's cache: a workgroup's own stores precede its own atomic store in program order.  Same arithmetic per element as the three kernels; the norm is the same fixed-order double sum.
// Requires gridDim.x workgroups to be co-resident (<= 4 per CU: the launcher's choice).
// ---------------------------------------------------------------------------------------
constexpr int OPTIM_REST_MAX = 12;
struct OptimRest { long off[OPTIM_REST_MAX]; long n[OPTIM_REST_MAX]; int n_rest, pad_; };   // ranges of the flat gradient that no slab job writes
constexpr unsigned OPT_FAULT_BARRIER = 0x100u;
__device__ __forceinline__ double slab_own_block(int pass, int blk, const SlabJob& J, float* __restrict__ p, float* __restrict__ g,
                                                 float* __restrict__ s0, float* __restrict__ s1, float coef, const OptimSpec& o,
                                                 float clr, float step_size, float bc2_sqrt) {
  double sq = 0.0;
  if (blk >= J.main_blocks) {      // bias-gradient slabs
    const int c = (blk - J.main_blocks) * blockDim.x + threadIdx.x;
    if (c >= J.nb) return 0.0;
    const long e = (J.bout - g) + c;
    if (pass == 0) {
      float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
      int k = 0;
      for (; k + 4 <= J.nslab; k += 4) {
        a0 += J.bslabs[(long)k * J.nb + c];       a1 += J.bslabs[(long)(k + 1) * J.nb + c];
        a2 += J.bslabs[(long)(k + 2) * J.nb + c]; a3 += J.bslabs[(long)(k + 3) * J.nb + c];
      }
      for (; k < J.nslab; ++k) a0 += J.bslabs[(long)k * J.nb + c];
      const float tot = (a0 + a1) + (a2 + a3);
      const float v = J.accumulate ? g[e] + tot : tot;
      g[e] = v;
      sq = (double)v * (double)v;
    } else {
      optim_update(p[e], g[e], s0[e], s1 ? s1 + e : nullptr, coef, o, clr, step_size, bc2_sqrt);
    }
    return sq;
  }
  const long i = (long)blk * blockDim.x + threadIdx.x;
  if (i >= J.n4) return 0.0;
  const long e = (J.out - g) + 4 * i;
  if (pass == 0) {
    const f32x4* q = reinterpret_cast<const f32x4*>(J.slabs) + i;
    const long st4 = J.slab_stride / 4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 8 <= J.nslab; k += 8) {
      f32x4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = q[(long)(k + u) * st4];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; k < J.nslab; ++k) a += q[(long)k * st4];
    f32x4* og = reinterpret_cast<f32x4*>(g + e);
    if (J.accumulate) a += *og;
    *og = a;
#pragma unroll
    for (int c = 0; c < 4; ++c) sq += (double)a[c] * (double)a[c];
  } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) optim_update(p[e + c], g[e + c], s0[e + c], s1 ? s1 + e + c : nullptr, coef, o, clr, step_size, bc2_sqrt);
  }
  return sq;
}
// The fused single-GPU step's combine launch of a network: the recorded weight-gradient combines (slab_reduce_multi_kernel's work)
// AND the squared norm of the whole flat gradient (sqnorm_partial_kernel's work) in one launch -- every workgroup squares what it
// has just written (or, for the trailing `rest_blocks` workgroups, what no job writes: OptimRest) and leaves one double per
// workgroup for optim_step_kernel's fixed-order sum.  One launch and one pass over the gradient less per network and step.
static __global__ __launch_bounds__(256) void slab_reduce_norm_kernel(const SlabJobs jobs, const int n_job_blocks, const OptimRest rest,
                                                                      float* __restrict__ g, double* __restrict__ norm_part) {
  __shared__ double shn[16];
  double acc = 0.0;
  if ((int)blockIdx.x < n_job_blocks) {
    int q = 0;
    while (q + 1 < jobs.n && (int)blockIdx.x >= jobs.j[q + 1].block0) ++q;
    OptimSpec none;
    none.kind = 0; none.lr = none.weight_decay = none.eps = none.lr_decay = none.beta1 = none.beta2 = 0.f; none.step = 1; none.max_norm = 0.f;
    acc = slab_own_block(0, (int)blockIdx.x - jobs.j[q].block0, jobs.j[q], nullptr, g, nullptr, nullptr, 1.f, none, 0.f, 0.f, 1.f);
  } else {
    const long nrb = (long)gridDim.x - n_job_blocks, rb = (long)blockIdx.x - n_job_blocks;
    for (int r = 0; r < rest.n_rest; ++r)
      for (long i = rb * blockDim.x + threadIdx.x; i < rest.n[r]; i += nrb * blockDim.x) {
        const double v = (double)g[rest.off[r] + i];
        acc += v * v;
      }
  }
  const double part = block_sum_d(acc, shn);
  if (threadIdx.x == 0) norm_part[blockIdx.x] = part;
}

static __global__ __launch_bounds__(256) void optim_fused_kernel(
    const SlabJobs jobs, const int n_job_blocks, const OptimRest rest, float* __restrict__ p, float* __restrict__ g, float* __restrict__ s0,
    float* __restrict__ s1, double* norm_part /* [gridDim.x] */, unsigned long long* bar, const unsigned long long bar_target,
    const unsigned long long timeout_ticks, double* __restrict__ norm2_out, OptimSpec o, unsigned int* fault_dev, unsigned int* fault_host,
    unsigned int* skipped_host) {
  __shared__ double shn[16];
  __shared__ float coef_sh;
  __shared__ int ok_sh;
  if (*fault_dev) {       // a persistent launch of this step gave up: no combine, no update (optim_step_kernel's rule); uniform over the grid
    if (threadIdx.x == 0) __hip_atomic_fetch_add(bar, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // the counter advances by the grid per launch
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      if (fault_host) *fault_host = *fault_dev;
      if (skipped_host) *skipped_host += 1u;
      if (norm2_out) *norm2_out = __longlong_as_double(0x7ff8000000000000LL);
    }
    return;
  }
  float clr = o.lr, bc2_sqrt = 1.f, step_size = o.lr;
  if (o.kind == 0) {
    clr = o.lr / (1.f + (float)(o.step - 1) * o.lr_decay);
  } else {
    const double bc1 = 1.0 - pow((double)o.beta1, (double)o.step);
    const double bc2 = 1.0 - pow((double)o.beta2, (double)o.step);
    step_size = (float)((double)o.lr / bc1);
    bc2_sqrt = (float)sqrt(bc2);
  }
  float coef = 1.f;
  for (int pass = 0; pass < 2; ++pass) {
    double acc = 0.0;
    int q = 0;
    for (int vb = blockIdx.x; vb < n_job_blocks; vb += gridDim.x) {
      while (q + 1 < jobs.n && vb >= jobs.j[q + 1].block0) ++q;
      acc += slab_own_block(pass, vb - jobs.j[q].block0, jobs.j[q], p, g, s0, s1, coef, o, clr, step_size, bc2_sqrt);
    }
    for (int r = 0; r < rest.n_rest; ++r)
      for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < rest.n[r]; i += (long)gridDim.x * blockDim.x) {
        const long e = rest.off[r] + i;
        if (pass == 0) { const double v = (double)g[e]; acc += v * v; }
        else optim_update(p[e], g[e], s0[e], s1 ? s1 + e : nullptr, coef, o, clr, step_size, bc2_sqrt);
      }
    if (pass == 1) break;
    const double part = block_sum_d(acc, shn);
    if (threadIdx.x == 0) {
      __hip_atomic_store(norm_part + blockIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // s_waitcnt: the partial is acknowledged before the arrival is counted (no L2 write-back)
      __hip_atomic_fetch_add(bar, 1ULL, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t0 = wall_clock64();
      int ok = 1;
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < bar_target) {
        __builtin_amdgcn_s_sleep(4);
        if (wall_clock64() - t0 > timeout_ticks) { ok = 0; break; }
      }
      if (!ok) atomicOr(fault_dev, OPT_FAULT_BARRIER);
      ok_sh = ok;
    }
    __syncthreads();
    if (!ok_sh) return;           // nobody steps: a workgroup that never arrived keeps everyone below the target
    double tot = 0.0;
    for (int i = threadIdx.x; i < (int)gridDim.x; i += blockDim.x)
      tot += __hip_atomic_load(norm_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    tot = block_sum_d(tot, shn);     // the same fixed order in every workgroup
    if (threadIdx.x == 0) {
      if (blockIdx.x == 0 && norm2_out) *norm2_out = tot;
      float c = 1.f;
      if (o.max_norm > 0.f) c = fminf(o.max_norm / ((float)sqrt(tot) + 1e-6f), 1.f);
      coef_sh = c;
    }
    __syncthreads();
    coef = coef_sh;
  }
}

// out[r][c] = in[r][c] * keep(r, c) / (1 - p)   (nn.LSTM inter-layer dropout and its backward; in == out allowed)
static __global__ void dropout_apply_kernel(const float* __restrict__ in, float* __restrict__ out, long rows, int cols, DropoutSpec d) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * cols) return;
  const long r = e / cols;
  const int c = (int)(e - r * cols);
  out[e] = dropout_keep(d, (int)r, c) ? in[e] * d.scale : 0.f;
}

// ---------------------------------------------------------------------------------------
// Distortion metrics of the training loop (reference train.py:358-432: inv_scale + split_streams +
// nnmnkwii.metrics.{melcd, lf0_mean_squared_error, vuv_error, mean_squared_error}) as ONE masked
// reduction over the valid frames of (y_static, y_hat_static): 2*Ds*4 B/frame of HBM reads, nothing
// written but 7 doubles.  One wave per frame, lanes <-> columns.  Column roles and the index of each
// column's statistics (static+dynamic domain, train.py:361-372) come from the host.
//   S = arithmetic type of the inverse scaling (float for f32 statistics, double for f64 ones: torch
//   promotes f32 features * 1-D f64 statistics to f64).  The vuv column is always f32: its statistics
//   are 0-dim tensors (train.py:373), which do not promote.  mul and add are rounded separately
//   (no fma contraction) so that the vuv > 0.5 binarisation is bit-exact.
// ---------------------------------------------------------------------------------------
enum DistRole { DIST_MCD = 0, DIST_BAP = 1, DIST_LF0 = 2, DIST_VUV = 3, DIST_MSE = 4 };
constexpr int DIST_NSUM = 7;   // s_mcd, s_bap, s_f0, n_voiced, n_vuv_err, s_mse, n_frames

__device__ __forceinline__ float inv_scale_rn(float x, float s, float m) { return __fadd_rn(__fmul_rn(x, s), m); }
__device__ __forceinline__ double inv_scale_rn(double x, double s, double m) { return __dadd_rn(__dmul_rn(x, s), m); }
__device__ __forceinline__ float dist_exp(float v) { return expf(v); }
__device__ __forceinline__ double dist_exp(double v) { return exp(v); }

template <typename S>
__global__ __launch_bounds__(256) void distortion_kernel(const float* __restrict__ y, const float* __restrict__ yh, int Ds,
                                                         const S* __restrict__ mean, const S* __restrict__ stdv,
                                                         const int* __restrict__ col_stat, const int* __restrict__ col_role,
                                                         int vuv_col, const int* __restrict__ lengths, int B, int T,
                                                         double* __restrict__ partials) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const long N = (long)B * T;
  double acc[DIST_NSUM];
#pragma unroll
  for (int i = 0; i < DIST_NSUM; ++i) acc[i] = 0.0;
  float vs = 0.f, vm = 0.f;
  if (vuv_col >= 0) { vs = (float)stdv[col_stat[vuv_col]]; vm = (float)mean[col_stat[vuv_col]]; }
  for (long f = (long)blockIdx.x * nwave + wave; f < N; f += (long)gridDim.x * nwave) {
    const int b = (int)(f / T), t = (int)(f - (long)b * T);
    if (t >= lengths[b]) continue;                       // wave-uniform
    const float* yr = y + f * Ds;
    const float* hr = yh + f * Ds;
    bool va = false, vb = false;
    if (vuv_col >= 0) {
      va = inv_scale_rn(yr[vuv_col], vs, vm) > 0.5f;     // train.py:375-377
      vb = inv_scale_rn(hr[vuv_col], vs, vm) > 0.5f;
    }
    double q_mcd = 0.0, q_bap = 0.0, q_f0 = 0.0, q_mse = 0.0;
    for (int c = lane; c < Ds; c += 64) {
      const int role = col_role[c];
      if (role < 0 || role == DIST_VUV) continue;
      const int si = col_stat[c];
      S a = inv_scale_rn((S)yr[c], stdv[si], mean[si]);
      S h = inv_scale_rn((S)hr[c], stdv[si], mean[si]);
      if (role == DIST_LF0) {
        if (!(va && vb)) continue;
        a = dist_exp(a); h = dist_exp(h);                // linear_domain=True (train.py:407)
      }
      const S z = a - h;
      const double zz = (double)(z * z);
      if (role == DIST_MCD) q_mcd += zz; else if (role == DIST_BAP) q_bap += zz;
      else if (role == DIST_LF0) q_f0 += zz; else q_mse += zz;
    }
    q_mcd = wave_sum_d(q_mcd); q_bap = wave_sum_d(q_bap); q_f0 = wave_sum_d(q_f0); q_mse = wave_sum_d(q_mse);
    acc[0] += sqrt(q_mcd); acc[1] += sqrt(q_bap); acc[2] += q_f0;
    acc[3] += (va && vb) ? 1.0 : 0.0; acc[4] += (va != vb) ? 1.0 : 0.0;
    acc[5] += q_mse; acc[6] += 1.0;
  }
  __shared__ double sh[4][DIST_NSUM];
  if (lane == 0)
    for (int i = 0; i < DIST_NSUM; ++i) sh[wave][i] = acc[i];
  __syncthreads();
  if (threadIdx.x < DIST_NSUM) {
    double r = 0.0;
    for (int w = 0; w < nwave; ++w) r += sh[w][threadIdx.x];
    partials[(long)blockIdx.x * DIST_NSUM + threadIdx.x] = r;
  }
}

// out[i] = sum over blocks of partials[blk][i], fixed order; grid 1 x 64*DIST_NSUM threads (one wave per sum)
static __global__ void distortion_finalize_kernel(const double* __restrict__ partials, int nblk, double* __restrict__ out) {
  const int lane = threadIdx.x & 63, i = threadIdx.x >> 6;
  double r = 0.0;
  for (int k = lane; k < nblk; k += 64) r += partials[(long)k * DIST_NSUM + i];
  r = wave_sum_d(r);
  if (lane == 0) out[i] = r;
}

// ---------------------------------------------------------------------------------------
// In2OutHighwayNet combine (reference gantts/models.py:57-69):
//   y_hat_static = x_static + Tx * Gx      ;  backward: dGx = g*Tx, dTz = g*Gx*Tx*(1-Tx)
// ---------------------------------------------------------------------------------------
static __global__ void highway_forward_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ Tx, int ldt,
                                       const float* __restrict__ Gx, int ldg, float* __restrict__ out, int ldo,
                                       long rows, int sd) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * sd) return;
  const long r = e / sd; const int c = (int)(e - r * sd);
  out[r * ldo + c] = x[r * ldx + c] + Tx[r * ldt + c] * Gx[r * ldg + c];
}
static __global__ void highway_backward_kernel(const float* __restrict__ g, int ldgr, const float* __restrict__ Tx, int ldt,
                                        const float* __restrict__ Gx, int ldg, float* __restrict__ dGx, int ld1,
                                        float* __restrict__ dTz, int ld2, long rows, int sd) {
  const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= rows * sd) return;
  const long r = e / sd; const int c = (int)(e - r * sd);
  const float gv = g[r * ldgr + c], t = Tx[r * ldt + c];
  dGx[r * ld1 + c] = gv * t;
  dTz[r * ld2 + c] = gv * Gx[r * ldg + c] * ((1.f - t) * t);
}

// ---------------------------------------------------------------------------------------
// results of one update_* call, written by a single thread and copied D2H once
// ---------------------------------------------------------------------------------------
// zero_gnorm: the gradient norm is not known / not applicable at this point (early results, phase != "train"): report 0
// tv_from_sum: the valid-frame count arrived with the sums (StepScalars::tv_sum): file it first
static __global__ void finalize_d_kernel(StepScalars* sc, StepResults* out, int zero_gnorm, int tv_from_sum = 0) {
  if (threadIdx.x || blockIdx.x) return;
  if (tv_from_sum) { sc->tv = (float)sc->tv_sum; sc->inv_tv = 1.0f / (float)sc->tv_sum; }
  const float T = sc->tv;
  const float lr = -((float)sc->s_real) / T, lf = -((float)sc->s_fake) / T;
  out->loss_real_d = lr; out->loss_fake_d = lf; out->loss_d = lr + lf;
  out->real_correct = (float)sc->n_real_ok; out->fake_correct = (float)sc->n_fake_ok;
  out->gnorm_d = zero_gnorm ? 0.f : (float)sqrt(sc->gnorm2_d); out->tv = T;
}
// part_mge / part_mse (optional): per-block partial sums that have not been reduced into sc yet -- the fused
// (single-GPU) call folds sum_partials_kernel into this launch; launch with 256 threads then, else 1.
static __global__ void finalize_g_kernel(StepScalars* sc, StepResults* out, float adv_w, float mse_w, float mge_w, int has_adv,
                                  int zero_gnorm, const double* __restrict__ part_mge, int n_mge,
                                  const double* __restrict__ part_mse, int n_mse) {
  if (blockIdx.x) return;
  __shared__ double shp[16];
  finalize_g_body(sc, out, adv_w, mse_w, mge_w, has_adv, zero_gnorm, part_mge, n_mge, part_mse, n_mse, nullptr, 0, shp);
}

}  // namespace gt
