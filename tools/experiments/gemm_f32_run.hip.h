// [r6] Streamed tile runs of the float32 MFMA GEMM family (gfx950 only): the 64 x 64 products of gemm_f32.hip.h as a PERSISTENT grid
// whose workgroups walk a run of tiles with no seam between them ("persistent tile loop WITH the next tile's first
// K-stage loads in flight under the current tile's epilogue").
//
//   * grid = resident slots (4 workgroups of 256 threads per CU); a workgroup's first tile is static (its rank inside its XCD), every
//     further tile is pulled from its XCD's queue (one returning agent-scope atomic per tile, issued a whole tile ahead of its use);
//     an exhausted queue steals from the other XCDs' queues.  Tiles keep the XCD-contiguous order of gemm_xcd_order (neighbouring
//     tiles share operand panels in that XCD's L2).  Nobody ever waits for another workgroup: no residency assumption, no deadlock.
//   * ONE flat K-stage stream per workgroup: the last K stage of tile i requests the FIRST K stage of tile i+1 (same register
//     staging, same LDS double buffer, same counted vmcnt) -- there is no prologue between tiles;
//   * the epilogue of tile i (bias / LeakyReLU / Philox keep bits / f' of the producer / stores) runs out of a second accumulator
//     set, in 16 micro-steps slotted into the MFMA shadows of the first K stage of tile i+1, straight from the MFMA C layout
//     (row segments of 128 B per half wave, no LDS transpose: the LDS belongs to the operand stream);
//   * only the first prologue and the last epilogue of a workgroup are exposed.
// Per element the arithmetic is that of gemm_tile / gemm_store_tile (k ascending in pairs, same epilogue operations in the same order):
// results are BIT-IDENTICAL to the per-tile launches (tests/test_gpu_parity.py, tools/gemm_run_bench.hip).
// Supported: float32, both operands 16-byte loadable, K (and every weight-gradient slab) a multiple of 32 and at least 64, epilogue flavours
// GEMM_A_NONE / GEMM_A_LEAKY_PHILOX, no accumulate / addm / K segments / summed operand (the launchers fall back to gemm_f32_kernel / gemm_pair_kernel otherwise).
#pragma once
#include "../../gantts_amd/csrc/gemm_f32.hip.h"

namespace gt {

// Work queues of one launch (device memory, zero before the launch; the last workgroup to leave zeroes them again):
//   [0..7]  items handed out dynamically from XCD q's queue, phase 0     [8..15] the same, phase 1     [16] workgroups that have left
constexpr int RUN_Q_WORDS = 32;
constexpr int RUN_SCRATCH_FLOATS = GEMM_THREADS * 4;      // column-sum exchange of the weight gradient
constexpr size_t run_lds_bytes() { return (size_t)2 * GEMM_BK * (68 + 68) * sizeof(float) + RUN_SCRATCH_FLOATS * sizeof(float) + 64; }

struct RunDbg {          // optional per-workgroup stamps (100 MHz wall clock), tools/gemm_run_bench.hip
  unsigned long long* stamps;   // [grid][4]: start, first K stage done, last K stage done, end
  unsigned int* tiles;          // [grid]: tiles walked
  unsigned long long* trace;    // [8][64]: stage-end stamps of workgroups 0, 137, 300, 511, 600, 777, 900, 1023 (first 64 stages)
};

// Workgroup barrier for the LDS hand-overs of the stream: this wave's LDS operations have completed, then s_barrier.  NOT __syncthreads():
// its workgroup-scope release also waits for every pending global store / returning atomic of the wave (vmcnt(0)) -- measured (r6 run5
// stamps): the stage that carries an epilogue's 16 stores waited 3-7 us for them at its barrier, and the prologue for the queue atomic.
// Nothing in these kernels passes GLOBAL data between the waves of a workgroup.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ void run_chunk(int n, int q, int& start, int& count) {     // XCD q's contiguous share of n items
  const int b = n >> 3, r = n & 7;
  count = b + (q < r ? 1 : 0);
  start = q * b + min(q, r);
}

// philox_group() without its branch on a launch-uniform (a branch on a loop invariant inside the tile loop invites the optimiser to
// clone the loop): with dp_t16 == 0 every data-parallel field is 0 and the formula returns g.
__device__ __forceinline__ uint32_t philox_group_nb(const DropoutSpec& d, uint32_t g) {
  const uint32_t h = g >= d.dp_nl16 ? 1u : 0u;
  const uint32_t gg = g - h * d.dp_nl16;
  const uint32_t b = (uint32_t)(((float)gg + 0.5f) * d.dp_inv_t16);
  return g + h * d.dp_half + d.dp_add + b * d.dp_mul;
}

__device__ __forceinline__ void philox_round(uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& k0, uint32_t& k1) {
  const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
  const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
  const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
  c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
  k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
}

// One phase of a launch: the workgroup walks tiles of ONE product (orientation KIND, epilogue flavour AMODE) until the phase's queues
// are empty.  smem: run_lds_bytes().  The LDS must be free on entry (a barrier separates two phases).
template <int KIND, int AMODE, bool DBG = false>
__device__ __forceinline__ int gemm_run_phase(const GemmArgs& g, const int n_items, unsigned int* next, float* smem, unsigned long long* stamp, unsigned long long* trace = nullptr) {
  static_assert(AMODE == GEMM_A_NONE || AMODE == GEMM_A_LEAKY_PHILOX, "epilogue flavours of the streamed runs");
  static_assert(KIND != GEMM_TN || AMODE == GEMM_A_NONE, "weight gradients have no activation");
  constexpr int LDM = gemm_ldm<KIND, 64>(), LDN = gemm_ldn<KIND, 64>();
  constexpr bool A_KC = KIND != GEMM_TN;        // A is k-contiguous in memory
  constexpr bool B_KC = KIND == GEMM_NT;        // B is k-contiguous in memory
  constexpr int BK = GEMM_BK, NG = BK / 2, NH = NG / 2;
  constexpr int U = 2;                          // 16-byte load units per thread, operand and K stage (64 x 32 floats / 256 threads / 4)
  constexpr bool PHILOX = AMODE == GEMM_A_LEAKY_PHILOX;
  float* As = smem;                             // [2][BK][LDM]
  float* Bs = smem + 2 * BK * LDM;              // [2][BK][LDN]
  float* scratch = smem + 2 * BK * (68 + 68);
  int* nxt_word = reinterpret_cast<int*>(scratch + RUN_SCRATCH_FLOATS);      // (written before a barrier, read behind it)

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (the wave index as a scalar: everything wave-uniform stays on the scalar unit)
  const int l31 = lane & 31, half = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_mn = g.n_tiles_m * g.n_tiles_n;
  const int S = (int)(gridDim.x >> 3), rank = (int)(blockIdx.x >> 3);

  // ---- work queue (thread 0)
  int fq = (int)(blockIdx.x & 7);               // the queue this workgroup pulls from
  unsigned int ftok = 0;                        // pending atomic's return
  auto fetch_issue = [&]() { if (tid == 0) ftok = __hip_atomic_fetch_add(next + fq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto fetch_resolve = [&]() -> int {           // thread 0: the pulled item; -1 = this XCD's queue is empty
    // (No stealing from the other XCDs' queues: measured -- r6 run4 stamps -- a workgroup that finds its queue empty and probes the seven others,
    //  while a thousand workgroups do the same at the end of the launch, spends 20-40 us in those round trips; the queues are equal by construction.)
    int start, count;
    run_chunk(n_items, fq, start, count);
    const int j = S + (int)ftok;
    return j < count ? start + j : -1;
  };
  auto publish = [&]() { if (tid == 0) *nxt_word = fetch_resolve(); };

  // ---- loader geometry: unit u of this thread covers 4 consecutive elements along the operand's contiguous direction
  auto a_pos = [&](int u, int& kk, int& mm) {
    const int e = tid + u * GEMM_THREADS;
    if (A_KC) { kk = (e % (BK / 4)) * 4; mm = e / (BK / 4); }
    else      { mm = (e % 16) * 4;       kk = e / 16; }
  };
  auto b_pos = [&](int u, int& kk, int& nn) {
    const int e = tid + u * GEMM_THREADS;
    if (B_KC) { kk = (e % (BK / 4)) * 4; nn = e / (BK / 4); }
    else      { nn = (e % 16) * 4;       kk = e / 16; }
  };
  const int mclamp = A_KC ? g.M - 1 : ((g.M - 1) / 4) * 4;
  const int nclamp = B_KC ? g.N - 1 : ((g.N - 1) / 4) * 4;
  const long stepA = A_KC ? (long)BK : (long)BK * g.lda;
  const long stepB = B_KC ? (long)BK : (long)BK * g.ldb;

  // ---- the stream's state
  uint32_t offA[U], offB[U];                    // BYTE offsets of this thread's units from the (uniform) stage pointers: saddr + 32-bit voffset loads
  const float* pA = g.A;                        // next K stage to request
  const float* pB = g.B;
  float ra[U * 4], rb[U * 4];
  float csum[4] = {0.f, 0.f, 0.f, 0.f};         // TN: column sums of the tile whose K stages are being deposited
  bool cs_want = false;                         // ... and whether that tile reports them (tile_n == 0)

  auto item_coords = [&](int it, int& slab, int& m0, int& n0) {
    slab = KIND == GEMM_TN ? it / tiles_mn : 0;
    const int t = it - slab * tiles_mn;
    const int tm = t / g.n_tiles_n;
    m0 = tm * 64; n0 = (t - tm * g.n_tiles_n) * 64;
  };
  auto item_k = [&](int slab, int& kb, int& ke) {
    kb = 0; ke = g.K;
    if (KIND == GEMM_TN) { kb = slab * g.k_chunk; ke = min(g.K, kb + g.k_chunk); }
  };
  auto aim = [&](int m0, int n0, int kb) {      // point the loader at K stage 0 of a tile
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int kk, mm; a_pos(u, kk, mm);
      const int m = min(m0 + mm, mclamp);
      offA[u] = 4u * (A_KC ? (uint32_t)m * (uint32_t)g.lda + (uint32_t)kk : (uint32_t)kk * (uint32_t)g.lda + (uint32_t)m);
      int nn; b_pos(u, kk, nn);
      const int n = min(n0 + nn, nclamp);
      offB[u] = 4u * (B_KC ? (uint32_t)n * (uint32_t)g.ldb + (uint32_t)kk : (uint32_t)kk * (uint32_t)g.ldb + (uint32_t)n);
    }
    pA = g.A + (A_KC ? (long)kb : (long)kb * g.lda);
    pB = g.B + (B_KC ? (long)kb : (long)kb * g.ldb);
  };
  // one unit of the requested K stage: global -> registers (the launchers guarantee K % 32 == 0 for these launches: no K tails)
  auto load_a = [&](int u) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pA) + offA[u]);
    const f32x4 v = A_KC ? ld4u(p) : *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int c = 0; c < 4; ++c) ra[u * 4 + c] = v[c];
  };
  auto load_b = [&](int u) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(pB) + offB[u]);
    const f32x4 v = B_KC ? ld4u(p) : *reinterpret_cast<const f32x4*>(p);
#pragma unroll
    for (int c = 0; c < 4; ++c) rb[u * 4 + c] = v[c];
  };
  auto store_a = [&](int u, float* as) {
    int kk, mm; a_pos(u, kk, mm);
    if (A_KC) {
#pragma unroll
      for (int c = 0; c < 4; ++c) as[(kk + c) * LDM + mm] = ra[u * 4 + c];
    } else {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = ra[u * 4 + c];
      *reinterpret_cast<f32x4*>(as + kk * LDM + mm) = v;
    }
    if (KIND == GEMM_TN && cs_want) {
#pragma unroll
      for (int c = 0; c < 4; ++c) csum[c] += ra[u * 4 + c];
    }
  };
  auto store_b = [&](int u, float* bs) {
    int kk, nn; b_pos(u, kk, nn);
    if (B_KC) {
#pragma unroll
      for (int c = 0; c < 4; ++c) bs[(kk + c) * LDN + nn] = rb[u * 4 + c];
    } else {
      f32x4 v;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = rb[u * 4 + c];
      *reinterpret_cast<f32x4*>(bs + kk * LDN + nn) = v;
    }
  };

  // ---- epilogue of the PREVIOUS tile (accumulators accp), 16 micro-steps; MFMA C layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  f32x16 acc, accp;
  int pv_m0 = 0, pv_n0 = 0, pv_slab = 0;
  bool pv_full = true, pv_cs = false;
  // NN: f' of the producer needs the sign of its stored activation H at this lane's 16 elements.  They are requested during the tile's OWN last
  // K stage (the epilogue registers accp are dead there: every tile is at least two stages deep) and folded into a 16-bit mask at the tile
  // switch: the epilogue stage carries one register for them.
  float hraw[8];
  uint32_t hmask = 0u;
  float e_bias = 0.f;                           // (raw load; selected against has_bias where it is USED: a select at the load would wait for it there)
  const bool has_bias = KIND == GEMM_NT && g.bias != nullptr;
  const float* bias_p = has_bias ? g.bias : g.A;        // (always a valid address: the load is unconditional, its value selected)
  const bool cs_any = KIND == GEMM_TN && g.colsum_slab != nullptr;
  uint32_t pa0 = 0, pa1 = 0, pa2 = 0, pa3 = 0, pak0 = 0, pak1 = 0;      // Philox state: call A (rows 0-15 of the wave tile: q = 0, 1), then call B (rows 16-31)
  // Addressing of the epilogue: a lane's element (q, s) of the finished tile lies at  base(tile, wave) + (8 q + s) pitch  [scalar]  +  (4 half pitch + l31)
  // [one 32-bit byte offset per lane, constant over the phase]: loads / stores with a scalar base and a 32-bit lane offset, no per-element
  // address arithmetic on the vector unit.  Ragged tiles (pv_full == false) take a guarded path.
  const uint32_t lo_c = 4u * (uint32_t)(4 * half * g.ldc + l31);
  const uint32_t lo_h = (KIND == GEMM_NN && PHILOX) ? 4u * (uint32_t)(4 * half * g.ldh + l31) : 0u;
  auto e_row0 = [&]() { return pv_m0 + wm * 32; };             // (scalar) first row / column of this wave's 32 x 32 piece
  auto e_col0 = [&]() { return pv_n0 + wn * 32; };
  auto h_request = [&](int q, int m0, int n0) {                 // full tiles: rows m0 + wm 32 + 8 q + 4 half + s of the tile being multiplied
    const float* hb = g.H + (long)(m0 + wm * 32 + 8 * q) * g.ldh + n0 + wn * 32;
#pragma unroll
    for (int s = 0; s < 4; ++s) hraw[(q & 1) * 4 + s] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(hb + (long)s * g.ldh) + lo_h);
  };
  auto h_fold = [&](int q) {                                    // the signs of q's four values into the mask
#pragma unroll
    for (int s = 0; s < 4; ++s) hmask |= (hraw[(q & 1) * 4 + s] > 0.f ? 1u : 0u) << (q * 4 + s);
  };
  auto e_philox_init = [&](int q, uint32_t& c0, uint32_t& c1, uint32_t& c2, uint32_t& c3, uint32_t& k0, uint32_t& k1) {
    // rows e_row0 + 8 q + 4 half + s: their 16-row group does not depend on half (the scalar part is a multiple of 8)
    c0 = 2u * philox_group_nb(g.drop, (uint32_t)(e_row0() + 8 * q) >> 4) + (uint32_t)half; c1 = (uint32_t)(e_col0() + l31);
    c2 = 0x243F6A88u; c3 = 0x85A308D3u; k0 = g.drop.key0; k1 = g.drop.key1;
  };
  auto e_apply = [&](int q, const uint32_t (&rnd)[4], auto FULL) {       // 4 elements: rows e_row0 + 8 q + 4 half + s of this lane's column
    float v[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      v[s] = accp[q * 4 + s];
      if (KIND == GEMM_NT) {
        v[s] += has_bias ? e_bias : 0.f;
        if (PHILOX) {
          v[s] = leaky(v[s]);
          v[s] = philox_piece(rnd, 4 * (q & 1) + s) >= g.drop.thresh ? v[s] * g.drop.scale : 0.f;
        }
      } else if (KIND == GEMM_NN && PHILOX) {
        v[s] = philox_piece(rnd, 4 * (q & 1) + s) >= g.drop.thresh ? v[s] * g.drop.scale : 0.f;
        bool pos;
        if (decltype(FULL)::value) pos = ((hmask >> (q * 4 + s)) & 1u) != 0u;
        else pos = g.H[(long)min(e_row0() + 8 * q + 4 * half + s, g.M - 1) * g.ldh + min(e_col0() + l31, g.N - 1)] > 0.f;     // (ragged tiles: cold path)
        v[s] *= pos ? 1.f : 0.01f;
      }
    }
    float* cb = g.C + (KIND == GEMM_TN ? (long)pv_slab * g.slab_stride : 0L) + (long)(e_row0() + 8 * q) * g.ldc + e_col0();
    if (decltype(FULL)::value) {
#pragma unroll
      for (int s = 0; s < 4; ++s) epi_store1(reinterpret_cast<float*>(reinterpret_cast<char*>(cb + (long)s * g.ldc) + lo_c), v[s]);
    } else {
      const int n = e_col0() + l31;
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (e_row0() + 8 * q + 4 * half + s < g.M && n < g.N) epi_store1(reinterpret_cast<float*>(reinterpret_cast<char*>(cb + (long)s * g.ldc) + lo_c), v[s]);
    }
  };
  auto e_colsum = [&]() {                        // TN: the finished tile's column sums of A, summed in gemm_tile's order
    if (KIND == GEMM_TN && pv_cs) {
      if (tid < 64 && pv_m0 + tid < g.M) {
        const int own = tid >> 2, c = tid & 3;
        float tot = 0.f;
#pragma unroll
        for (int j = 0; j < GEMM_THREADS / 16; ++j) tot += scratch[(own + j * 16) * 4 + c];
        g.colsum_slab[(long)pv_slab * g.M + pv_m0 + tid] = tot;
      }
    }
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  // FULL (compile time): the finished tile lies inside the matrix -- the flavour that rides in the K stages; ragged tiles are finished
  // synchronously at the tile switch (rare: the last row / column of tiles)
  auto epi_micro = [&](int m, auto FULL) {
    if (PHILOX) {
      // 0: bias, Philox A init; 1-5: its rounds (2 per step); 6, 7: rows 0-15; 8: Philox B init (same registers); 9-13: its rounds; 14, 15: rows 16-31
      if (m == 0) { if (KIND == GEMM_NT) e_bias = bias_p[has_bias ? min(e_col0() + l31, g.N - 1) : 0]; e_philox_init(0, pa0, pa1, pa2, pa3, pak0, pak1); }
      if (m == 8) e_philox_init(2, pa0, pa1, pa2, pa3, pak0, pak1);
      if ((m >= 1 && m <= 5) || (m >= 9 && m <= 13)) { philox_round(pa0, pa1, pa2, pa3, pak0, pak1); philox_round(pa0, pa1, pa2, pa3, pak0, pak1); }
      if (m == 6 || m == 7) { const uint32_t r[4] = {pa0, pa1, pa2, pa3}; e_apply(m - 6, r, FULL); }
      if (m == 14 || m == 15) { const uint32_t r[4] = {pa0, pa1, pa2, pa3}; e_apply(m - 12, r, FULL); }
    } else {
      const uint32_t r[4] = {0u, 0u, 0u, 0u};
      if (m == 0 && KIND == GEMM_NT) e_bias = bias_p[has_bias ? min(e_col0() + l31, g.N - 1) : 0];
      if (m == 2) e_apply(0, r, FULL);
      if (m == 5) e_apply(1, r, FULL);
      if (m == 8) e_apply(2, r, FULL);
      if (m == 11) e_apply(3, r, FULL);
      if (m == 13) e_colsum();
    }
  };

  // ---- one K stage: 16 MFMA groups (one k pair each); everything else rides in their shadows (gemm_tile's schedule): the fragments of the
  // next group, the request of the NEXT stage (groups 0, 1, 4, 5), its deposit into the other LDS buffer (groups 8, 9, 12, 13, counted vmcnt).
  // The next stage is ALWAYS requested (behind the last tile of a workgroup the loader is aimed at a valid stage nobody reads).  Two flavours:
  //   FIRST  stages 0 .. nk-2 of a tile: while `pend`, one micro-step of the PREVIOUS tile's epilogue per group (only stage 0 has pend set);
  //   LAST   the last stage of a tile: NN + Philox requests the tile's own H values and folds their signs into hmask.
  auto k_stage = [&](auto LAST, const int buf, const bool pend, const int m0, const int n0, const bool hreq) {
    constexpr bool last = decltype(LAST)::value;
    const float* as = As + buf * BK * LDM + wm * 32 + l31 + half * LDM;
    const float* bs = Bs + buf * BK * LDN + wn * 32 + l31 + half * LDN;
    float* as_w = As + (buf ^ 1) * BK * LDM;
    float* bs_w = Bs + (buf ^ 1) * BK * LDN;
    float a_cur = as[0], b_cur = bs[0], a_nxt = 0.f, b_nxt = 0.f;
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if (gi + 1 < NG) { a_nxt = as[(2 * gi + 2) * LDM]; b_nxt = bs[(2 * gi + 2) * LDN]; }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if ((u * NH) / U == gi) load_a(u);
        if (NH + (u * NH) / U == gi) store_a(u, as_w);
        if ((u * NH) / U + 1 == gi) load_b(u);
        if (NH + (u * NH) / U + 1 == gi) store_b(u, bs_w);
      }
      if (!last) { if (pend) epi_micro(gi, T_{}); }
      if (last && KIND == GEMM_NN && PHILOX) {      // H of rows 0-15 requested in groups 2, 3, folded in 10, 11, where rows 16-31 are requested (folded at the switch)
        if (hreq) {
          if (gi == 2) { hmask = 0u; h_request(0, m0, n0); }
          if (gi == 3) h_request(1, m0, n0);
          if (gi == 10) { h_fold(0); h_request(2, m0, n0); }
          if (gi == 11) { h_fold(1); h_request(3, m0, n0); }
        }
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur, acc, 0, 0, 0);
      a_cur = a_nxt; b_cur = b_nxt;
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- first tile: static, by rank inside the XCD
  int it;
  {
    int start, count;
    run_chunk(n_items, fq, start, count);
    it = rank < count ? start + rank : -1;
  }
  if (it < 0) {                                 // more slots than tiles in this XCD: ask the queues
    fetch_issue();
    publish();
    lds_barrier();
    it = __builtin_amdgcn_readfirstlane(*nxt_word);
    lds_barrier();
    if (it < 0) return 0;
  }
  int c_slab, c_m0, c_n0, kb, ke;
  item_coords(it, c_slab, c_m0, c_n0);
  item_k(c_slab, kb, ke);
  int nk = (ke - kb) / BK;                      // >= 2 (launcher)
  aim(c_m0, c_n0, kb);
  cs_want = cs_any && c_n0 == 0;
  {                                             // prologue: K stage 0 -> LDS buffer 0 (the only exposed one of this phase)
#pragma unroll
    for (int u = 0; u < U; ++u) { load_a(u); load_b(u); }
    pA += stepA; pB += stepB;
#pragma unroll
    for (int u = 0; u < U; ++u) { store_a(u, As); store_b(u, Bs); }
  }
  fetch_issue();
  lds_barrier();
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int buf = 0, n_tiles = 0, tr_n = 0;
  bool pend = false;
  if (DBG && trace && tid == 0) trace[tr_n++] = wall_clock64();
  float cs_done[4] = {0.f, 0.f, 0.f, 0.f};
#pragma clang loop unroll(disable)
  for (;;) {                                    // one tile per trip
#pragma clang loop unroll(disable)
    for (int kt = 0; kt + 1 < nk; ++kt) {       // stages whose successor belongs to the same tile
      k_stage(F_{}, buf, pend, 0, 0, false);
      pend = false;
      pA += stepA; pB += stepB;
      if (kt == nk - 2) publish();              // the next tile's id, read at the top of the last stage
      lds_barrier();
      buf ^= 1;
      if (DBG && n_tiles == 0 && kt == 0 && tid == 0) stamp[1] = wall_clock64();
      if (DBG && trace && tid == 0 && tr_n < 64) trace[tr_n++] = wall_clock64();
    }
    // last stage: its request is stage 0 of the next tile
    const int nx = __builtin_amdgcn_readfirstlane(*nxt_word);
    const bool cs_this = cs_want;
    const bool c_full = c_m0 + 64 <= g.M && c_n0 + 64 <= g.N;
    int n_slab = 0, n_m0 = 0, n_n0 = 0, n_kb = 0, n_ke = 0;
    if (KIND == GEMM_TN) {
#pragma unroll
      for (int c = 0; c < 4; ++c) { cs_done[c] = csum[c]; csum[c] = 0.f; }
    }
    if (nx >= 0) {
      item_coords(nx, n_slab, n_m0, n_n0);
      item_k(n_slab, n_kb, n_ke);
      aim(n_m0, n_n0, n_kb);
      cs_want = cs_any && n_n0 == 0;
    } else {                                    // nothing follows: request this tile's stage 0 again (valid memory, never read)
      aim(c_m0, c_n0, kb);
      cs_want = false;
    }
    k_stage(T_{}, buf, false, c_m0, c_n0, c_full);
    pA += stepA; pB += stepB;
    if (KIND == GEMM_TN && cs_this) {           // hand the finished tile's column sums to its epilogue (read behind the barrier below)
#pragma unroll
      for (int c = 0; c < 4; ++c) scratch[tid * 4 + c] = cs_done[c];
    }
    lds_barrier();
    buf ^= 1;
    if (DBG && trace && tid == 0 && tr_n < 64) trace[tr_n++] = wall_clock64();
    // tile finished: its accumulators move to the epilogue set; the stream continues with the next tile
    if (KIND == GEMM_NN && PHILOX && c_full) {
      h_fold(2); h_fold(3);
    }
    accp = acc;
    pv_m0 = c_m0; pv_n0 = c_n0; pv_slab = c_slab; pv_cs = cs_this;
    pv_full = c_full;
    pend = true;
    ++n_tiles;
    if (!pv_full) {                             // ragged tile: finished here, guarded
#pragma unroll
      for (int m = 0; m < NG; ++m) { epi_micro(m, F_{}); __builtin_amdgcn_sched_barrier(0); }     // (pinned: hoisting the loads of all 16 steps costs the hot loop its registers)
      pend = false;
    }
    if (nx < 0) break;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    c_slab = n_slab; c_m0 = n_m0; c_n0 = n_n0; kb = n_kb; ke = n_ke;
    nk = (ke - kb) / BK;
    fetch_issue();
  }
  if (DBG && tid == 0) stamp[2] = wall_clock64();
  // the last tile's epilogue is exposed
  if (pend) {
#pragma unroll
    for (int m = 0; m < NG; ++m) { epi_micro(m, T_{}); __builtin_amdgcn_sched_barrier(0); }
  }
  return n_tiles;
}

// workgroup exit: the last one to leave zeroes the launch's queues for the next launch that uses them
__device__ __forceinline__ void gemm_run_leave(unsigned int* queues) {
  if (threadIdx.x == 0) {
    const unsigned int left = __hip_atomic_fetch_add(queues + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i <= 16; ++i) __hip_atomic_store(queues + i, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// One launch = one product (forward / backward-data / weight gradient) as streamed tile runs.
template <int KIND, int AMODE, bool DBG = false, int WPE = 4>
__global__ __launch_bounds__(GEMM_THREADS, WPE) void gemm_run_kernel(const GemmArgs g, const int n_items, unsigned int* queues, const RunDbg dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long* st = DBG ? dbg.stamps + 4 * (size_t)blockIdx.x : nullptr;
  if (DBG && threadIdx.x == 0) { st[0] = wall_clock64(); st[1] = st[2] = 0; }
  unsigned long long* tr = nullptr;
  if (DBG && dbg.trace) {
    const int ids[8] = {0, 137, 300, 511, 600, 777, 900, 1023};
#pragma unroll
    for (int i = 0; i < 8; ++i) if ((int)blockIdx.x == ids[i]) tr = dbg.trace + 64 * i;
  }
  const int n = gemm_run_phase<KIND, AMODE, DBG>(g, n_items, queues, smem, st, tr);
  if (DBG && threadIdx.x == 0) { st[3] = wall_clock64(); dbg.tiles[blockIdx.x] = (unsigned)n; }
  gemm_run_leave(queues);
}

// One launch = a layer's weight gradient (g2: n2 units, phase 0 -- longest work first) and its backward-data product (g1: n1 tiles, phase 1).
template <int AMODE, bool DBG = false, int WPE = 4>
__global__ __launch_bounds__(GEMM_THREADS, WPE) void gemm_run_pair_kernel(const GemmArgs g1, const GemmArgs g2, const int n1, const int n2, unsigned int* queues,
                                                                        const RunDbg dbg) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long* st = DBG ? dbg.stamps + 4 * (size_t)blockIdx.x : nullptr;
  if (DBG && threadIdx.x == 0) { st[0] = wall_clock64(); st[1] = st[2] = 0; }
  int n = gemm_run_phase<GEMM_TN, GEMM_A_NONE, DBG>(g2, n2, queues, smem, st);
  lds_barrier();
  n += gemm_run_phase<GEMM_NN, AMODE, false>(g1, n1, queues + 8, smem, nullptr);
  if (DBG && threadIdx.x == 0) { st[3] = wall_clock64(); dbg.tiles[blockIdx.x] = (unsigned)n; }
  gemm_run_leave(queues);
}

}  // namespace gt
