// Experiment: 256 x 256 tile of the bf16-storage product with one workgroup per CU -- LDS-DMA operand stages of 32 k (32 KB:
// 64-byte rows, chunks swizzled by (row >> 2) & 3), a ring of 4 stages (three in flight), fragments of the next half-stage
// requested in front of the 16 MFMAs of the current one.  Twice the flops per operand byte of the 128 x 128 tile.
#pragma once
#include "../../gantts_amd/csrc/gemm_bf16s.hip.h"
namespace gt {
constexpr size_t gemm_b16_big_lds_bytes() { return (size_t)4 * 512 * 32 * 2; }     // 4 stages x (256 + 256) rows x 32 k

template <int EPI, int AMODE>
__global__ __launch_bounds__(GEMM_THREADS, 1) void gemm_b16_big_kernel(const GemmB16Args g) {
  typedef const __attribute__((address_space(1))) void* gptr_t;
  typedef __attribute__((address_space(3))) void* lptr_t;
  extern __shared__ __attribute__((aligned(16))) float smem_f[];
  __bf16* smem = reinterpret_cast<__bf16*>(smem_f);
  constexpr int BM = 256, BN = 256, NS = 4, STAGE = 512 * 32;
  int slab, tile_m, tile_n;
  gemm_b16_tile_of(g, &slab, &tile_m, &tile_n);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, half = lane >> 5, wm = wave >> 1, wn = wave & 1;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  int k_begin = 0, k_end = g.K;
  if (EPI == B16_SLAB) { k_begin = slab * g.k_chunk; k_end = min(g.K, k_begin + g.k_chunk); }
  // block b = rows 16 b .. 16 b + 15; this wave issues blocks wave, wave + 4, wave + 8, wave + 12 of each operand
  const __bf16* srcA[4];
  const __bf16* srcB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = (wave + 4 * i) * 16 + (lane >> 2), c = (lane & 3) ^ ((r >> 2) & 3);
    srcA[i] = g.A + (long)min(m0 + r, g.M - 1) * g.lda + k_begin + 8 * c;
    srcB[i] = g.B + (long)min(n0 + r, g.N - 1) * g.ldb + k_begin + 8 * c;
  }
  auto issue = [&](int buf) {
    __bf16* As = smem + buf * STAGE;
    __bf16* Bs = As + 256 * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      __builtin_amdgcn_global_load_lds((gptr_t)srcA[i], (lptr_t)(As + (wave + 4 * i) * 512), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)srcB[i], (lptr_t)(Bs + (wave + 4 * i) * 512), 16, 0, 0);
      srcA[i] += 32; srcB[i] += 32;
    }
  };
  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int nk = (k_end - k_begin) / 32;
#pragma unroll
  for (int p = 0; p < NS - 1; ++p)
    if (p < nk) issue(p);
  if (nk >= 3) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if (nk == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int arow = wm * 128 + l31, brow = wn * 128 + l31;
  const int fsw = (l31 >> 2) & 3;
  bf16x8 fa0[4], fb0[4], fa1[4], fb1[4];
  auto frags = [&](int buf, int kk, bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
    const __bf16* ah = smem + buf * STAGE + arow * 32 + 8 * ((2 * kk + half) ^ fsw);
    const __bf16* bh = smem + buf * STAGE + 256 * 32 + brow * 32 + 8 * ((2 * kk + half) ^ fsw);
#pragma unroll
    for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(ah + i * 32 * 32);
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = *reinterpret_cast<const bf16x8*>(bh + j * 32 * 32);
  };
  auto mma = [&](bf16x8 (&fa)[4], bf16x8 (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
  };
  if (nk > 0) frags(0, 0, fa0, fb0);
  int buf = 0;
  for (int t = 0; t < nk; ++t) {
    int nbuf = buf + NS - 1; if (nbuf >= NS) nbuf -= NS;
    if (t + NS - 1 < nk) issue(nbuf);
    frags(buf, 1, fa1, fb1);
    mma(fa0, fb0);
    const int younger = min(nk - 1, t + NS - 1) - (t + 1);       // stages requested after stage t + 1
    if (younger >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (younger == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    ++buf; if (buf >= NS) buf = 0;
    if (t + 1 < nk) frags(buf, 0, fa0, fb0);
    mma(fa1, fb1);
  }
  __syncthreads();
  gemm_b16_epilogue<BM, BN, EPI, AMODE>(g, slab, m0, n0, acc, smem);
}
}  // namespace gt
