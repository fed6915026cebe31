// Experiment (tools only, not part of the engine): LDS-DMA variant of the 64x64 f32 tile.  Measured by
// tools/gemm_tile_sweep.hip: results equal to the register-staged kernel up to the summation order (max rel diff 5e-7),
// time equal within noise (16384x512x512: 77.7 vs 78.1 us; 32768x256x256: 56.1 vs 51.8 us) -- the K loop of the staged
// kernel already runs at ~90 % of the matrix rate the shader clock allows; what a launch loses is its fixed 12 us.
#pragma once
#include "../gantts_amd/csrc/gemm_f32.hip.h"
namespace gt {
// ------------------------------------------------------------------------------------------
// LDS-DMA variant of the 64x64 f32 tile for the forward (NT) and backward-data (NN) products whose operands take 16-byte
// loads and whose K is a multiple of 32.  Operand tiles go global -> LDS directly (global_load_lds_dwordx4: no staging
// registers, no ds_write in the issue stream; lane l's 16 bytes land at base + 16*l, tools/lds_dma_probe.hip), as
// row-major [row][32 k] images (128-byte rows) whose 16-byte chunks are XOR-swizzled by (row >> 1) & 7 -- each lane
// CHOOSES which global chunk it fetches, so the swizzle costs nothing -- and the MFMA fragments are ds_read_b128:
// lane (row, half) reads chunk 2j + half, i.e. the operands of four consecutive MFMAs (k = 8j + 4*half + c), conflict
// free (the 16 lanes of a pass hit 16 different bank quads).  LDS instructions per K-tile and wave: 8 (NT) / 4 + 16 (NN,
// whose B tile is k-major [32 k][64 n], swizzled by ((k >> 2) & 1) * 8, read 4 bytes at a time) instead of 32 reads + 16
// writes.  The summation order over k inside a K-tile differs from gemm_tile (k pairs (c, c+4) instead of (2g, 2g+1));
// both are exact f32 FMA chains.
// ------------------------------------------------------------------------------------------
constexpr size_t gemm_dma_lds_bytes() { return (size_t)4 * 2048 * sizeof(float); }   // 2 stages x (A 8 KB + B 8 KB)

template <int KIND>
__device__ __forceinline__ void gemm_tile_dma(const GemmArgs& g, const int tile_m, const int tile_n, float* smem) {
  static_assert(KIND == GEMM_NT || KIND == GEMM_NN, "forward / backward-data only");
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * 64, n0 = tile_n * 64;
  float* As = smem;                 // [2][64][32]
  float* Bs = smem + 2 * 2048;      // NT: [2][64 (n)][32];  NN: [2][32 (k)][64]
  // this lane's source chunk of each of the two 1-KiB blocks per operand and K-tile (block = 2 * i ... wave-uniform)
  const float* srcA[2];
  const float* srcB[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int blk = i * 4 + wave;
    const int r = blk * 8 + (lane >> 3), sl = lane & 7;
    const int c = sl ^ ((r >> 1) & 7);
    srcA[i] = g.A + (long)min(m0 + r, g.M - 1) * g.lda + 4 * c;
    if (KIND == GEMM_NT) {
      srcB[i] = g.B + (long)min(n0 + r, g.N - 1) * g.ldb + 4 * c;
    } else {
      const int k = blk * 4 + (lane >> 4), sb = lane & 15;
      const int cb = sb ^ (((k >> 2) & 1) * 8);
      srcB[i] = g.B + (long)k * g.ldb + min(n0 + 4 * cb, ((g.N - 1) >> 2) << 2);
    }
  }
  const long stepB = KIND == GEMM_NT ? 32L : 32L * g.ldb;
  auto issue = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int blk = i * 4 + wave;
      __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(As + buf * 2048 + blk * 256), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(Bs + buf * 2048 + blk * 256), 16, 0, 0);
      srcA[i] += 32; srcB[i] += stepB;
    }
  };
  f32x16 acc[1][1];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
  const int arow = wm * 32 + l31, fa = (arow >> 1) & 7;
  const int brow = wn * 32 + l31, fb = (brow >> 1) & 7;       // NT: B row = output column
  const int nk = g.K / 32;
  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < nk; ++t) {
    const int buf = t & 1;
    if (t + 1 < nk) issue(buf ^ 1);
    const float* ar = As + buf * 2048 + arow * 32;
    f32x4 a[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) a[j] = *reinterpret_cast<const f32x4*>(ar + 4 * ((2 * j + half) ^ fa));
    if (KIND == GEMM_NT) {
      const float* br = Bs + buf * 2048 + brow * 32;
      f32x4 b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const f32x4*>(br + 4 * ((2 * j + half) ^ fb));
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][c], b[j][c], acc[0][0], 0, 0, 0);
    } else {
      const float* bb = Bs + buf * 2048 + (brow & 3);
      const int q = brow >> 2;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float b[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = 8 * j + 4 * half + c;                 // (k >> 2) & 1 == half
          b[c] = bb[k * 64 + 4 * (q ^ (half * 8))];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j][c], b[c], acc[0][0], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of the next tile has landed
    __syncthreads();                                      // ... everybody's has, and everybody is done reading this one
  }
  gemm_store_tile<KIND, 64, 64, PREC_F32, 32>(g, 0, m0, n0, acc, smem, g.M);
}

template <int KIND>
__global__ __launch_bounds__(GEMM_THREADS, 4) void gemm_dma_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int nwg = gridDim.x;
  int bid = blockIdx.x;
  {   // XCD-aware tile order, as gemm_f32_kernel
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = bid / g.n_tiles_n, tile_n = bid - tile_m * g.n_tiles_n;
  gemm_tile_dma<KIND>(g, tile_m, tile_n, smem);
}

}  // namespace gt
