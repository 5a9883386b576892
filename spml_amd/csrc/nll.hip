// K4/K5: pixel-to-segment NCA negative log-likelihood, forward and backward.
//
// Replaces _calculate_log_likelihood / _one_hot_calculate_log_likelihood
// (segsort/loss.py:15-82, :85-130 of the reference), which materialise five
// [P,M] fp32 temporaries: sim = exp(kappa * E @ Pr^T), two label masks and two
// masked products.  Here the [P,M] similarity only ever exists as 32x32 MFMA
// accumulator tiles:
//
//   forward   one wave owns 32*NB pixels (B operand, fragments resident in
//             registers) and streams every 32-prototype tile (A operand, 1-KB
//             coalesced fragment blocks from L2); epilogue per tile: exp, label
//             predicate, three running per-pixel sums.  No LDS, no barriers.
//   backward  flash-style recompute of the tile; T = s * dnll/ds is split to
//             f16 (hi, lo) in registers and used directly as the B operand of
//             a second MFMA chain (the accumulator layout IS a valid B
//             fragment for a permuted contraction order; the matching A
//             fragments are laid out by the prep kernels):
//               dE^T [d][pixel] += PrT[d][m] * T[m][pixel]      (kernel bwd_de)
//               dPr^T[d][proto] += ET [d][p] * T'[p][proto]     (kernel bwd_dp)
//
// All contractions run on the f16 matrix cores at fp32 accuracy (split-f16 x2,
// see common.hpp).  Prep kernels convert fp32 rows to fragment-major (hi, lo)
// f16 arrays once per call; algorithmic flops fwd 2*P*M*D, bwd ~6*P*M*D.
#include <algorithm>
#include <type_traits>

#include "nll_common.hpp"

namespace spml {
namespace {

// ---------------------------------------------------------------------------
// prep: fp32 rows [R,D] -> fragment-major split-f16.
//  std layout : frag[((tile*KS + ks)*64 + lane)*8 + e] = X[32*tile + (lane&31)][16*ks + 8*(lane>>5) + e]
//  T   layout : frag[(((tile*DT + dt)*2 + s)*64 + lane)*8 + e]
//                 = scale[rho] * X[rho][32*dt + (lane&31)],  rho = 32*tile + tile_row(8*s + e, lane>>5)
// out-of-range rows / channels are written as zero.
// ---------------------------------------------------------------------------
// RAW: the residual is stored unscaled, l = f16(v - h), so that all three product terms of the split
// contraction go into ONE accumulator (nll_fwd2); the caller scales the two operands by 2^3 / 2^-3 so
// that residuals of O(1) elements stay normal f16 numbers and the rest lose at most 2^-25 absolute.
template <bool RAW>
__global__ __launch_bounds__(256) void prep_std(const float* __restrict__ x, int64_t R, int D,
                                                int KS, float scale, _Float16* __restrict__ oh,
                                                _Float16* __restrict__ ol) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // fragment block id
  const int64_t nfrag = ((R + 31) / 32) * KS;
  if (f >= nfrag) return;
  const int lane = threadIdx.x & 63;
  const int64_t tile = f / KS;
  const int ks = (int)(f - tile * KS);
  const int64_t row = 32 * tile + (lane & 31);
  const int k0 = 16 * ks + 8 * (lane >> 5);
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (row < R && k0 + e < D) v = x[(size_t)row * D + k0 + e] * scale;
    _Float16 a, b;
    if constexpr (RAW) {
      a = (_Float16)v;
      b = (_Float16)(v - (float)a);
    } else {
      split_f16(v, a, b);
    }
    h[e] = a; l[e] = b;
  }
  *reinterpret_cast<half8*>(oh + ((size_t)f * 64 + lane) * 8) = h;
  *reinterpret_cast<half8*>(ol + ((size_t)f * 64 + lane) * 8) = l;
}

template <bool RAW>
__global__ __launch_bounds__(256) void prep_T(const float* __restrict__ x, int64_t R, int D,
                                              int DT, const float* __restrict__ rowscale,
                                              const float* __restrict__ gscale, float scale,
                                              _Float16* __restrict__ oh,
                                              _Float16* __restrict__ ol) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nfrag = ((R + 31) / 32) * DT * 2;
  if (f >= nfrag) return;
  const int lane = threadIdx.x & 63;
  const int s = (int)(f & 1);
  const int64_t td = f >> 1;
  const int64_t tile = td / DT;
  const int dt = (int)(td - tile * DT);
  const int d = 32 * dt + (lane & 31);
  const float gs = gscale ? 1.0f / gscale[0] : 1.0f;     // power of two: exact
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t rho = 32 * tile + tile_row(8 * s + e, lane >> 5);
    float v = 0.f;
    if (rho < R && d < D) {
      v = x[(size_t)rho * D + d] * scale;
      if (rowscale) v *= rowscale[rho] * gs;
    }
    _Float16 a, b;
    if constexpr (RAW) {
      v = fminf(fmaxf(v, -65000.f), 65000.f);      // (saturate: rows of pixels whose T scale is far below 2^14)
      a = (_Float16)v;
      b = (_Float16)(v - (float)a);
    } else {
      split_f16(v, a, b);
    }
    h[e] = a; l[e] = b;
  }
  *reinterpret_cast<half8*>(oh + ((size_t)f * 64 + lane) * 8) = h;
  *reinterpret_cast<half8*>(ol + ((size_t)f * 64 + lane) * 8) = l;
}

// gscale[0] = power of two >= max |g|  (so that g/gscale is in [-1,1]); 1 if all zero
__global__ __launch_bounds__(1024) void max_abs_pow2(const float* __restrict__ g, int64_t n,
                                                     float* __restrict__ out) {
  __shared__ float red[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    float p = 1.0f;
    if (m > 0.f && m < INFINITY) {
      int ex;
      frexpf(m, &ex);            // m = f * 2^ex, f in [0.5, 1)
      p = ldexpf(1.0f, ex);
    }
    out[0] = p;
  }
}

// z tile (rows = A rows, cols = B cols) for KS k-steps; A fragments from LDS
// ([KS] hi blocks then [KS] lo blocks of 1 KB)
template <int KS>
__device__ __forceinline__ void zgemm_lds(const unsigned char* a_lds, int lane,
                                          const half8 (&bh)[KS], const half8 (&bl)[KS],
                                          float16v& zh, float16v& zx) {
#pragma unroll
  for (int r = 0; r < 16; ++r) { zh[r] = 0.f; zx[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const half8 a_h = *reinterpret_cast<const half8*>(a_lds + ((size_t)ks * 64 + lane) * 16);
    const half8 a_l = *reinterpret_cast<const half8*>(a_lds + ((size_t)(KS + ks) * 64 + lane) * 16);
    zh = mfma32(a_h, bh[ks], zh);
    zx = mfma32(a_h, bl[ks], zx);
    zx = mfma32(a_l, bh[ks], zx);
  }
}

// z tile with A streamed straight from global (used by the dPr kernel)
template <int KS>
__device__ __forceinline__ void zgemm(const _Float16* __restrict__ ah_g,
                                      const _Float16* __restrict__ al_g, int lane,
                                      const half8 (&bh)[KS], const half8 (&bl)[KS],
                                      float16v& zh, float16v& zx) {
#pragma unroll
  for (int r = 0; r < 16; ++r) { zh[r] = 0.f; zx[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const half8 a_h = *reinterpret_cast<const half8*>(ah_g + ((size_t)ks * 64 + lane) * 8);
    const half8 a_l = *reinterpret_cast<const half8*>(al_g + ((size_t)ks * 64 + lane) * 8);
    zh = mfma32(a_h, bh[ks], zh);
    zx = mfma32(a_h, bl[ks], zx);
    zx = mfma32(a_l, bh[ks], zx);
  }
}

// ------------------------------- forward -----------------------------------
// 256 threads = 4 waves x (32*NB pixels, resident B fragments).  The prototype
// tiles (A operand) stream through a 2-slot LDS ring by direct-to-LDS DMA, shared
// by the 4 waves: the next tile lands while the current one is computed.
template <int KS, int NB, bool TAG, bool C32>
__global__ __launch_bounds__(256) void nll_fwd(NllArgs a) {
  using CodeT = code_t<C32>;
  constexpr int NBLK = 2 * KS + 1;               // hi blocks, lo blocks, 32 row codes (+pad)
  constexpr int SLOT = NBLK * 1024;
  const int DEPTH = a.depth_fwd;                 // LDS ring: 1 tile in use, DEPTH-1 in flight
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t wave = (int64_t)blockIdx.x * 4 + wv;
  const int64_t pt0 = wave * NB;

  half8 bh[NB][KS], bl[NB][KS];
  CodeT pcode[NB];
  int own[NB];
  float s_same[NB], s_diff[NB], s_own[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int64_t pt = min(pt0 + nb, a.n.PT - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[nb][ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      bl[nb][ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    }
    const int64_t p = min(32 * pt + j, a.n.P - 1);
    pcode[nb] = (CodeT)a.px_code[p];
    own[nb] = (int)a.own[p];
    s_same[nb] = 0.f; s_diff[nb] = 0.f; s_own[nb] = 0.f;
  }

  auto stage = [&](int64_t mt, int slot) {
    unsigned char* dst = sm + slot * SLOT;
    for (int b = wv; b < 2 * KS + 1; b += 4) {           // wave-uniform loop
      const void* src;
      if (b < KS) src = a.ph + ((size_t)(mt * KS + b) * 64 + lane) * 8;
      else if (b < 2 * KS) src = a.pl + ((size_t)(mt * KS + (b - KS)) * 64 + lane) * 8;
      else src = a.pr_code_pad + 32 * mt + 2 * min(lane, 15);   // padded copy: always in bounds
      dma_block(src, dst + (size_t)b * 1024);
    }
  };
  // row codes: lane i < 16 fetches the codes of rows (2i, 2i+1) of the tile

  const int my_blocks = (NBLK - wv + 3) / 4;           // DMA instructions this wave issues per tile
  for (int64_t t0 = 0; t0 < DEPTH - 1 && t0 < a.n.MT; ++t0) stage(t0, (int)t0);
  int slot = -1;                                       // ring slot of tile mt (no 64-bit modulo per tile)
  for (int64_t mt = 0; mt < a.n.MT; ++mt) {
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
    // tile mt has landed once only the younger tiles' copies are outstanding
    if (DEPTH > 2) wait_vmcnt(my_blocks * (int)min((int64_t)(DEPTH - 2), a.n.MT - 1 - mt));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                                      // ... for every wave; slot of tile mt-1 is free
    if (mt + DEPTH - 1 < a.n.MT) stage(mt + DEPTH - 1, slot == 0 ? DEPTH - 1 : slot - 1);
    const unsigned char* at = sm + slot * SLOT;
    const int64_t* codes = reinterpret_cast<const int64_t*>(at + 2 * KS * 1024);
    CodeT rc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rc[r] = (CodeT)codes[tile_row(r, half)];
    const bool ragged = 32 * (mt + 1) > a.n.M;           // uniform: last, partial tile
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float16v zh, zx;
      zgemm_lds<KS>(at, lane, bh[nb], bl[nb], zh, zx);
      float sv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) sv[r] = __builtin_amdgcn_exp2f(zh[r] + zx[r] * kSplitInv);
      if (ragged) {
        // row 32*mt + tile_row(r, half) < M  <=>  tile_row(r, 0) < lim (immediates, no per-row adds)
        const int lim = (int)(a.n.M - 32 * mt) - 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) sv[r] = (tile_row(r, 0) < lim) ? sv[r] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float t = code_match<TAG, CodeT>(pcode[nb], rc[r]) ? sv[r] : 0.f;
        s_same[nb] += t;
        s_diff[nb] += sv[r] - t;                         // exact: t is sv[r] or 0
      }
      if (__any((own[nb] >> 5) == (int)mt)) {            // wave-uniform, ~1 tile in 8 at most
        const int own_rel = own[nb] - (int)(32 * mt) - 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) s_own[nb] += (tile_row(r, 0) == own_rel) ? sv[r] : 0.f;
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    // the two lane halves saw different prototype rows of the same pixel
    const float same = s_same[nb] + __shfl_xor(s_same[nb], 32, 64);
    const float diff = s_diff[nb] + __shfl_xor(s_diff[nb], 32, 64);
    const float osim = s_own[nb] + __shfl_xor(s_own[nb], 32, 64);
    const int64_t p = 32 * (pt0 + nb) + j;
    if (half == 0 && pt0 + nb < a.n.PT && p < a.n.P) {
      // loss.py:61-80: pos = sum_same - own (that order); fallback to own if pos <= 0
      const float pos = same - osim;
      const bool fb = (a.mode & SPML_NLL_PLAIN) || !(pos > 0.f);
      const float num = fb ? osim : pos;
      const float den = diff + num;
      a.nll[p] = -logf(num / den);
      float4v st = {num, den, osim, fb ? 1.f : 0.f};
      *reinterpret_cast<float4v*>(a.stats + (size_t)p * 4) = st;
    }
  }
}

// nll_fwd2: prototype tiles per chunk -- a constant, so that the summation order of a pixel's three
// sums depends on M alone (3072 prototypes per chunk: 12 KB of codes + 64 KB of ring = 76 KB, two
// workgroups per CU -- 80 KB each would need the CU's whole LDS and only one fits)

// ------------------------------- forward, v2 -------------------------------
// Round 3.  What bounded nll_fwd / nll_fwd_pipe was not instruction count but waiting: one workgroup
// barrier + LDS-DMA round trip per 32-prototype tile (~1200 cycles of work per wave between barriers),
// and dependent MFMA chains on one accumulator pair.  Here
//   * a wave owns TWO pixel tiles (64 pixels, B fragments resident) and a ring slot holds MTB
//     prototype tiles: 2 * MTB tile products per wave between two barriers (8 for MTB = 4), every A
//     fragment read from LDS feeds two products, half the LDS-DMA bytes per pair;
//   * all three split terms go into ONE accumulator per product (unscaled residuals, prep_std<true>),
//     and the two products of a prototype tile alternate on the matrix pipe, so that no MFMA waits for
//     the write-back of its predecessor (a dependent MFMA that is not issued back-to-back costs ~43
//     extra cycles, MI355X_MICROARCH "per-instruction constants");
//   * the epilogue of tile g runs while the 6 * KS MFMAs of tile g+1 are in the pipe (second accumulator
//     pair), den = sum over ALL prototypes - own (loss.py:61-80 rearranged: neg + pos = all - own), which
//     saves one VALU op per pair;
//   * the prototype range is cut into chunks of kFwd2TilesPerChunk tiles (grid.y) -- a function of M
//     only, so a pixel's result does not depend on which other pixels are in the call -- for
//     >> 512 workgroups and short tails; per-chunk partial sums are combined in chunk order by
//     nll_finalize (deterministic).
// Two workgroups per CU (12 KB of row codes + 2 x MTB x 2 KS KB of ring each, <= 256 VGPRs).
template <int KS, int MTB, bool TAG>
__global__ __launch_bounds__(256, 2) void nll_fwd2(NllArgs a) {
  constexpr int TILE = 2 * KS * 1024;            // hi blocks, lo blocks of one prototype tile
  constexpr int SLOT = MTB * TILE;
  constexpr int CODES = kFwd2TilesPerChunk * 32 * 4;   // the chunk's row codes (low words), resident
  static_assert((MTB & (MTB - 1)) == 0 && MTB >= 2, "MTB: even power of two");
  static_assert(kFwd2TilesPerChunk <= 128, "two 64-bit masks of uniform-tile flags");
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  int* const codes_lds = reinterpret_cast<int*>(sm);
  unsigned char* const ring = sm + CODES;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t pt0 = ((int64_t)blockIdx.x * 4 + wv) * 2;
  const int64_t mt_lo = (int64_t)blockIdx.y * kFwd2TilesPerChunk;
  const int ntile = (int)(min(a.n.MT, mt_lo + kFwd2TilesPerChunk) - mt_lo);
  const int nstage = (ntile + MTB - 1) / MTB;

  auto stage = [&](int st, int slot) {
    unsigned char* dst = ring + slot * SLOT;
    for (int b = wv; b < MTB * 2 * KS; b += 4) {         // wave-uniform loop
      const int t = b / (2 * KS), q = b - t * (2 * KS);
      const int64_t mt = mt_lo + min(st * MTB + t, ntile - 1);   // past the end: a harmless duplicate
      const void* src = q < KS ? a.ph + ((size_t)(mt * KS + q) * 64 + lane) * 8
                               : a.pl + ((size_t)(mt * KS + (q - KS)) * 64 + lane) * 8;
      dma_block(src, dst + (size_t)b * 1024);
    }
  };
  stage(0, 0);
  if (nstage > 1) stage(1, 1);
  // row codes of the whole chunk: stay in LDS for the kernel's lifetime (the ring only carries fragments,
  // so an epilogue may read its codes after the tile's ring slot has been handed to a later stage)
  for (int i = threadIdx.x; i < 32 * ntile; i += 256) codes_lds[i] = (int)a.pr_code_pad[32 * mt_lo + i];

  half8 bh[2][KS], bl[2][KS];
  int pcode[2], own[2];
  float s_same[2], s_all[2], s_own[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int64_t pt = min(pt0 + nb, a.n.PT - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[nb][ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      bl[nb][ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    }
    const int64_t p = min(32 * pt + j, a.n.P - 1);
    pcode[nb] = (int)a.px_code[p];
    own[nb] = (int)a.own[p] - (int)(32 * mt_lo);         // relative to this chunk's first prototype
    s_same[nb] = 0.f; s_all[nb] = 0.f; s_own[nb] = 0.f;
  }

  // stage st lives in ring slot st & 1.  advance(st): stage st has landed for every wave, every wave is
  // done reading the fragments of stage st - 1, whose slot takes stage st + 1
  auto advance = [&](int st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (st + 1 < nstage) stage(st + 1, (st + 1) & 1);
  };
  auto tile_at = [&](int g) -> const unsigned char* {
    return ring + ((g / MTB) & 1) * SLOT + (g & (MTB - 1)) * TILE + (size_t)lane * 16;
  };
  // One prototype tile step, scheduled by hand (the compiler's own schedule serialises the two
  // accumulator chains and parks the epilogue behind them): the 6 * KS MFMAs of tile `gn` -- two
  // products, alternating accumulators zn0 / zn1 -- each followed by its share of the epilogue of tile
  // `gp` (32 similarity values of zp0 / zp1: exp2, total, predicate, positive sum).  A fragments are
  // read one k-step ahead.  sched_barrier(0) pins the order; GEMM / EPI switch the halves off for the
  // pipeline's head and tail, LAST: the tile may be the ragged last one of the call.
  // pa / pl: the k-step-0 fragments of tile gn, read by the PREVIOUS step (or by prefetch() after a ring
  // advance), so that the MFMA burst does not open with an exposed LDS round trip; g_next >= 0: read
  // those of tile g_next on the way out
  half8 pa, pl;
  auto prefetch = [&](int g) {
    const unsigned char* at = tile_at(g);
    pa = *reinterpret_cast<const half8*>(at);
    pl = *reinterpret_cast<const half8*>(at + KS * 1024);
  };
  auto step = [&](int gn, float16v& zn0, float16v& zn1, int gp, const float16v& zp0, const float16v& zp1,
                  int g_next, auto gemm_tag, auto epi_tag, auto last_tag, auto uni_tag) {
    constexpr bool GEMM = decltype(gemm_tag)::value, EPI = decltype(epi_tag)::value;
    constexpr bool LAST = decltype(last_tag)::value;
    // UNI: all 32 prototypes of tile gp carry one code (image-major prototypes of the co-occurrence
    // term: > 90 % of the tiles) -> one predicate per pixel and tile, the epilogue is exp2 + add
    constexpr bool UNI = decltype(uni_tag)::value && !LAST;
    constexpr int NSLOT = 6 * KS;
    int rc[16];
    int lim = 0;
    float same[2] = {s_same[0], s_same[1]}, all[2] = {s_all[0], s_all[1]};
    float tsum[2] = {0.f, 0.f};
    if constexpr (EPI && !UNI) {
#pragma unroll
      for (int r = 0; r < 16; ++r) rc[r] = codes_lds[32 * gp + tile_row(r, half)];
      lim = (int)(a.n.M - 32 * (mt_lo + gp)) - 4 * half;               // rows below lim exist (>= 32: all)
    }
    const unsigned char* at = GEMM ? tile_at(gn) : ring;
    half8 ah, al, ah_n, al_n;
    if constexpr (GEMM) { ah = pa; al = pl; }
    const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto epi_values = [&](int slot) {                                  // slot: compile-time after unrolling
      if constexpr (EPI) {
#pragma unroll
        for (int v = slot * 32 / NSLOT; v < (slot + 1) * 32 / NSLOT; ++v) {
          const int nb = v >> 4, r = v & 15;
          float sv = __builtin_amdgcn_exp2f(nb ? zp1[r] : zp0[r]);
          if constexpr (UNI) {
            tsum[nb] += sv;
          } else {
            if constexpr (LAST) sv = (tile_row(r, 0) < lim) ? sv : 0.f;
            all[nb] += sv;
            same[nb] += code_match<TAG, int>(pcode[nb], rc[r]) ? sv : 0.f;
          }
        }
      }
    };
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (GEMM) {
        if (ks + 1 < KS) {
          ah_n = *reinterpret_cast<const half8*>(at + (ks + 1) * 1024);
          al_n = *reinterpret_cast<const half8*>(at + (KS + ks + 1) * 1024);
        }
        zn0 = mfma32(ah, bh[0][ks], ks == 0 ? zero : zn0);
      }
      epi_values(6 * ks + 0);
      if constexpr (GEMM) zn1 = mfma32(ah, bh[1][ks], ks == 0 ? zero : zn1);
      epi_values(6 * ks + 1);
      if constexpr (GEMM) zn0 = mfma32(ah, bl[0][ks], zn0);
      epi_values(6 * ks + 2);
      if constexpr (GEMM) zn1 = mfma32(ah, bl[1][ks], zn1);
      epi_values(6 * ks + 3);
      if constexpr (GEMM) zn0 = mfma32(al, bh[0][ks], zn0);
      epi_values(6 * ks + 4);
      if constexpr (GEMM) zn1 = mfma32(al, bh[1][ks], zn1);
      epi_values(6 * ks + 5);
      if constexpr (GEMM) { ah = ah_n; al = al_n; }
    }
    if constexpr (EPI && UNI) {
      const int code = codes_lds[32 * gp];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        all[nb] += tsum[nb];
        same[nb] += code_match<TAG, int>(pcode[nb], code) ? tsum[nb] : 0.f;
      }
    }
    if constexpr (GEMM && EPI) {
      // the schedule: one MFMA, then a 1/NSLOT share of the epilogue's VALU instructions
      constexpr int kValu = ((UNI ? 2 : TAG ? 6 : 5) * 32 + 8) / NSLOT + 1;
#pragma unroll
      for (int i = 0; i < NSLOT; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, kValu, 0);
      }
    }
    if constexpr (EPI) {
      // (pins the epilogue in this block: its results are not used before the kernel's end, and the
      // optimiser would otherwise sink it below the branches that follow)
      asm volatile("" : "+v"(same[0]), "+v"(same[1]), "+v"(all[0]), "+v"(all[1]));
      s_same[0] = same[0]; s_same[1] = same[1]; s_all[0] = all[0]; s_all[1] = all[1];
    }
    if (g_next >= 0) prefetch(g_next);
  };
  // the own prototype's similarity, taken from the very accumulator value that went into the sums
  // (rare, wave-uniform branch; kept out of the block above)
  auto own_check = [&](int g, const float16v& z0, const float16v& z1) {
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
      if (__any((own[nb] >> 5) == g)) {
        const float16v& z = nb ? z1 : z0;
        const int own_rel = own[nb] - 32 * g - 4 * half;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float zz = z[r];
          asm volatile("" : "+v"(zz));        // not to be merged with the epilogue's exp2 of the same value
          s_own[nb] += (tile_row(r, 0) == own_rel) ? __builtin_amdgcn_exp2f(zz) : 0.f;
        }
      }
    }
  };
  using Y = std::true_type;
  using N = std::false_type;

  const int my_blocks = (MTB * 2 * KS - wv + 3) / 4;
  if (nstage > 1) wait_vmcnt(my_blocks); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                    // stage 0 and the codes are in place
  // per tile: do all 32 rows carry one code?  Two ballots (tiles 0-63, 64-95) -> wave-uniform bit masks
  unsigned long long uni_lo, uni_hi;
  {
    bool u0 = lane < ntile && 32 * (mt_lo + lane + 1) <= a.n.M;        // a ragged tile: general epilogue
    bool u1 = lane + 64 < ntile && 32 * (mt_lo + lane + 65) <= a.n.M;
    for (int i = 1; i < 32; ++i) {
      u0 &= codes_lds[32 * min(lane, ntile - 1) + i] == codes_lds[32 * min(lane, ntile - 1)];
      u1 &= codes_lds[32 * min(lane + 64, ntile - 1) + i] == codes_lds[32 * min(lane + 64, ntile - 1)];
    }
    uni_lo = __ballot(u0);
    uni_hi = __ballot(u1);
  }
  auto is_uniform = [&](int g) -> bool { return ((g < 64 ? uni_lo >> g : uni_hi >> (g - 64)) & 1ull) != 0; };

  float16v zA0, zA1, zB0, zB1;
  prefetch(0);
  step(0, zA0, zA1, 0, zA0, zA1, 1, Y{}, N{}, N{}, N{});   // head: GEMM of tile 0 only
  int g = 0;
  for (; g + 2 < ntile; g += 2) {                     // A holds tile g; pa / pl hold tile g + 1's first fragments
    own_check(g, zA0, zA1);
    const bool turn = ((g + 2) & (MTB - 1)) == 0;     // tile g + 2 opens the next ring stage
    if (is_uniform(g)) step(g + 1, zB0, zB1, g, zA0, zA1, turn ? -1 : g + 2, Y{}, Y{}, N{}, Y{});
    else step(g + 1, zB0, zB1, g, zA0, zA1, turn ? -1 : g + 2, Y{}, Y{}, N{}, N{});
    own_check(g + 1, zB0, zB1);
    if (turn) { advance((g + 2) / MTB); prefetch(g + 2); }
    if (is_uniform(g + 1)) step(g + 2, zA0, zA1, g + 1, zB0, zB1, g + 3, Y{}, Y{}, N{}, Y{});
    else step(g + 2, zA0, zA1, g + 1, zB0, zB1, g + 3, Y{}, Y{}, N{}, N{});
  }
  own_check(g, zA0, zA1);
  if (g + 1 < ntile) {
    step(g + 1, zB0, zB1, g, zA0, zA1, -1, Y{}, Y{}, N{}, N{});
    own_check(g + 1, zB0, zB1);
    step(0, zA0, zA1, g + 1, zB0, zB1, -1, N{}, Y{}, Y{}, N{});   // tail: epilogue of the last tile only
  } else {
    step(0, zB0, zB1, g, zA0, zA1, -1, N{}, Y{}, Y{}, N{});
  }

#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    // the two lane halves saw different prototype rows of the same pixel
    const float same = s_same[nb] + __shfl_xor(s_same[nb], 32, 64);
    const float all = s_all[nb] + __shfl_xor(s_all[nb], 32, 64);
    const float osim = s_own[nb] + __shfl_xor(s_own[nb], 32, 64);
    const int64_t p = 32 * (pt0 + nb) + j;
    if (half == 0 && pt0 + nb < a.n.PT) {
      const float4v st = {same, all, osim, 0.f};
      *reinterpret_cast<float4v*>(a.partial + ((size_t)blockIdx.y * a.n.PT * 32 + p) * 4) = st;
    }
  }
}

// per pixel: chunk partials (same, all, own) summed in chunk order -> nll, stats (loss.py:61-80)
__global__ __launch_bounds__(256) void nll_finalize(const float* __restrict__ partial, int chunks,
                                                    int64_t P, int64_t P_pad, int plain,
                                                    float* __restrict__ nll, float* __restrict__ stats) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  float same = 0.f, all = 0.f, osim = 0.f;
  for (int c = 0; c < chunks; ++c) {
    const float4v v = *reinterpret_cast<const float4v*>(partial + ((size_t)c * P_pad + p) * 4);
    same += v[0]; all += v[1]; osim += v[2];
  }
  // pos = sum_same - own (that order); fallback to own if pos <= 0; den = neg + num with
  // neg = all - same: all - own without the fallback, all - same + own with it
  const float pos = same - osim;
  const bool fb = plain || !(pos > 0.f);
  const float num = fb ? osim : pos;
  const float den = fb ? (all - same) + osim : all - osim;
  nll[p] = -logf(num / den);
  const float4v st = {num, den, osim, fb ? 1.f : 0.f};
  *reinterpret_cast<float4v*>(stats + (size_t)p * 4) = st;
}

// ---------------------------------------------------------------------------
// backward.  With  s = exp(kappa z),  nll = -log(num) + log(den):
//   d nll / d s_m = w_m = dnum_m * (1/den - 1/num) + [m not same] / den,
//   dnum_m = fallback ? [m == own] : [m same] - [m == own].
// Per pixel this is  wa = fallback ? 0 : (1/den - 1/num)  for same-class prototypes,
// wb = 1/den for the others, and one special value for the own prototype.
// T = s * w (|T| <= 1) is split to f16 in registers; g * kappa is applied outside
// the contraction (dE: per pixel, in the epilogue; dPr: folded into the ET
// fragments by the prep kernel, scaled by a power of two so that they stay in
// f16 range).
// ---------------------------------------------------------------------------
template <bool TAG>
__global__ void coef_kernel(const float* __restrict__ stats, const int64_t* __restrict__ own,
                            const int64_t* __restrict__ px_code, const int64_t* __restrict__ pr_code, int64_t M,
                            int64_t P, int64_t P_pad, PixelCoef* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P_pad) return;
  PixelCoef c{0.f, 0.f, -1, 0.f};
  if (i < P) {
    const float4v st = *reinterpret_cast<const float4v*>(stats + (size_t)i * 4);
    const float inv_num = 1.0f / st[0], inv_den = 1.0f / st[1];
    c.wa = st[3] != 0.f ? 0.f : inv_den - inv_num;
    c.wb = inv_den;
    c.own = (int)own[i];
    const int64_t m = own[i];
    const bool own_same = m >= 0 && m < M && code_match<TAG, int64_t>(px_code[i], pr_code[m]);
    c.tscale = nll_t_scale(st[0], st[1], st[2], st[3] != 0.f, own_same);
  }
  out[i] = c;
}

// kappa g_p (2^14 / tscale_p): row scale of the transposed pixel fragments of nll_dp3.hip (T carries tscale_p)
__global__ void rowscale_ts_kernel(const float* g, float kappa, const PixelCoef* __restrict__ coef, int64_t n, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = g[i] * kappa * (kTScale / coef[i].tscale);
}

// Term of a pixel's OWN prototype, for the v2 / v3 dE kernels, which leave it out of their tiles:
// out[p] = w_own * s_own with s_own = exp2(kappa log2(e) <e_p, pr_own>) (fp64 dot product) and the own
// prototype's weight (see "backward" above): same code -> fallback ? 1/den - 1/num : 0, else
// fallback ? 2/den - 1/num : 1/num.  nll_de_finalize adds out[p] * pr_own to the tile sums.
template <bool TAG>
__global__ void own_term_kernel(const float* __restrict__ stats, const int64_t* __restrict__ own,
                                const int64_t* __restrict__ px_code, const int64_t* __restrict__ pr_code,
                                const float* __restrict__ emb, const float* __restrict__ protos, int64_t P,
                                int64_t P_pad, int64_t M, int D, float kappa_log2e, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P_pad) return;
  float v = 0.f;
  if (i < P) {
    const int64_t m = own[i];
    if (m >= 0 && m < M) {
      const float4v st = *reinterpret_cast<const float4v*>(stats + (size_t)i * 4);
      const float inv_num = 1.0f / st[0], inv_den = 1.0f / st[1];
      const bool fb = st[3] != 0.f;
      const bool same = code_match<TAG, int64_t>(px_code[i], pr_code[m]);
      const float c1 = inv_den - inv_num;
      const float w = same ? (fb ? c1 : 0.f) : (fb ? c1 + inv_den : inv_num);
      if (w != 0.f) {
        double dot = 0.0;
        for (int d = 0; d < D; ++d) dot += (double)emb[(size_t)i * D + d] * (double)protos[(size_t)m * D + d];
        v = w * __builtin_amdgcn_exp2f((float)(dot * (double)kappa_log2e));
      }
    }
  }
  out[i] = v;
}

// weight of the own prototype (rare path)
__device__ __forceinline__ float own_weight(bool same, float wa, float wb) {
  // not fallback (wa != 0 or generic): same -> 0, else 1/num = wb - wa
  // fallback (wa == 0):                same -> c1 = wb - 1/num ... needs 1/num; see caller
  return same ? 0.f : wb - wa;
}

// split 8 floats held in registers into (hi, lo) f16 fragments
__device__ __forceinline__ void split_regs(const float (&t)[16], int s2, half8& th, half8& tl) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = t[8 * s2 + e];
  split8(v, th, tl);
}

// dacc[dt] += A_T(dt, s2) * T   for both k-steps; A fragments in LDS at `at_lds`
// laid out [DT][2][hi|lo] blocks of 1 KB.
// The lo terms carry an exact 2^-11 and are accumulated apart.  For DT <= kLoResident
// the lo accumulators live in registers for the whole tile loop (dlo, folded in once
// by fold_lo at the end); for wider embeddings they are folded in per tile.
constexpr int kLoResident = 3;
template <int DT>
struct LoAcc {
  float16v v[DT <= kLoResident ? DT : 1];
};
template <int DT>
__device__ __forceinline__ void zero_lo(LoAcc<DT>& lo) {
#pragma unroll
  for (int dt = 0; dt < (DT <= kLoResident ? DT : 1); ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) lo.v[dt][r] = 0.f;
}
template <int DT>
__device__ __forceinline__ void fold_lo(const LoAcc<DT>& lo, float16v (&dacc)[DT]) {
  if constexpr (DT <= kLoResident) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[dt][r] += lo.v[dt][r] * kSplitInv;
  }
}

template <int DT>
__device__ __forceinline__ void second_gemm(const unsigned char* at_lds, int lane,
                                            const float (&t)[16], float16v (&dacc)[DT],
                                            LoAcc<DT>& dlo) {
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) {
    half8 th, tl;
    split_regs(t, s2, th, tl);
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const unsigned char* blk = at_lds + (size_t)((dt * 2 + s2) * 2) * 1024;
      const half8 a_h = *reinterpret_cast<const half8*>(blk + (size_t)lane * 16);
      const half8 a_l = *reinterpret_cast<const half8*>(blk + 1024 + (size_t)lane * 16);
      if constexpr (DT <= kLoResident) {
        dlo.v[dt] = mfma32(a_h, tl, dlo.v[dt]);
        dlo.v[dt] = mfma32(a_l, th, dlo.v[dt]);
      } else {
        float16v lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) lo[r] = 0.f;
        lo = mfma32(a_h, tl, lo);
        lo = mfma32(a_l, th, lo);
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[dt][r] += lo[r] * kSplitInv;
      }
      dacc[dt] = mfma32(a_h, th, dacc[dt]);
    }
  }
}

// T tile <-> cache: 64 contiguous bytes per lane
__device__ __forceinline__ void tcache_store(float* base, int64_t tile, int lane, const float (&t)[16]) {
  float4v* p = reinterpret_cast<float4v*>(base + ((size_t)tile * 64 + lane) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4v v = {t[4 * q], t[4 * q + 1], t[4 * q + 2], t[4 * q + 3]};
    p[q] = v;
  }
}
__device__ __forceinline__ void tcache_load(const float* base, int64_t tile, int lane, float (&t)[16]) {
  const float4v* p = reinterpret_cast<const float4v*>(base + ((size_t)tile * 64 + lane) * 16);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4v v = p[q];
    t[4 * q] = v[0]; t[4 * q + 1] = v[1]; t[4 * q + 2] = v[2]; t[4 * q + 3] = v[3];
  }
}

// ------------------------------- backward: dE ------------------------------
// Same streaming structure as the forward: 4 waves x 32 resident pixels, the
// prototype tiles (std fragments + codes + transposed fragments) flow through an
// LDS ring.  dE^T[d][pixel] += PrT[d][m] * T[m][pixel].
// TM: 0 = compute T, 1 = compute and keep it in the cache, 2 = read it from the cache (no similarity GEMM)
template <int KS, int DT, bool TAG, bool C32, int TM = 0>
__global__ __launch_bounds__(256) void nll_bwd_de(NllArgs a) {
  using CodeT = code_t<C32>;
  constexpr int kStd = TM == 2 ? 0 : 2 * KS + 1; // std hi/lo + codes (not needed when T comes from the cache)
  constexpr int NBLK = kStd + 4 * DT;            // ... + T-layout [DT][2][hi|lo]
  constexpr int SLOT = NBLK * 1024;
  const int DEPTH = a.depth;                     // LDS ring slots (2 or 3)
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t pt = min(a.spt0 + (int64_t)blockIdx.x * 4 + wv, a.spt1 - 1);
  const bool active = a.spt0 + (int64_t)blockIdx.x * 4 + wv < a.spt1;

  half8 bh[TM == 2 ? 1 : KS], bl[TM == 2 ? 1 : KS];
  if constexpr (TM != 2) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      bl[ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    }
  }
  const int64_t p = min(32 * pt + j, a.n.P - 1);
  const CodeT pcode = (CodeT)a.px_code[p];
  const PixelCoef cf = a.coef[32 * pt + j];
  const float inv_num = 1.0f / a.stats[(size_t)p * 4];

  float16v dacc[DT];
  LoAcc<DT> dlo;
  zero_lo<DT>(dlo);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[dt][r] = 0.f;

  auto stage = [&](int64_t mt, int slot) {
    unsigned char* dst = sm + slot * SLOT;
    for (int b = wv; b < NBLK; b += 4) {
      const void* src;
      if (b < kStd && b < KS) src = a.ph + ((size_t)(mt * KS + b) * 64 + lane) * 8;
      else if (b < kStd && b < 2 * KS) src = a.pl + ((size_t)(mt * KS + (b - KS)) * 64 + lane) * 8;
      else if (b < kStd) src = a.pr_code_pad + 32 * mt + 2 * min(lane, 15);
      else {
        const int q = b - kStd;                         // (dt*2 + s2)*2 + hi/lo
        const size_t f = ((size_t)mt * a.dt_all + a.dt0) * 2 + (q >> 1);
        src = ((q & 1) ? a.ptl : a.pth) + (f * 64 + lane) * 8;
      }
      dma_block(src, dst + (size_t)b * 1024);
    }
  };
  const int my_blocks = (NBLK - wv + 3) / 4;
  for (int64_t t0 = 0; t0 < DEPTH - 1 && t0 < a.n.MT; ++t0) stage(t0, (int)t0);
  int slot = -1;
  for (int64_t mt = 0; mt < a.n.MT; ++mt) {
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
    if (mt + 1 < a.n.MT && DEPTH > 2) wait_vmcnt(my_blocks * (int)min((int64_t)(DEPTH - 2), a.n.MT - 1 - mt));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (mt + DEPTH - 1 < a.n.MT) stage(mt + DEPTH - 1, slot == 0 ? DEPTH - 1 : slot - 1);
    const unsigned char* at = sm + slot * SLOT;
    const int64_t* codes = reinterpret_cast<const int64_t*>(at + 2 * KS * 1024);
    float t[16];
    if constexpr (TM == 2) {
      tcache_load(a.tcache_de, (pt - a.spt0) * a.n.MT + mt, lane, t);
    } else {
      float16v zh, zx;
      zgemm_lds<KS>(at, lane, bh, bl, zh, zx);
      const bool ragged = 32 * (mt + 1) > a.n.M;
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float s = __builtin_amdgcn_exp2f(zh[r] + zx[r] * kSplitInv);
        const bool same = code_match<TAG, CodeT>(pcode, (CodeT)codes[tile_row(r, half)]);
        t[r] = s * (same ? cf.wa : cf.wb);
      }
      if (__any((cf.own >> 5) == (int)mt)) {               // own prototype in this tile (rare)
        const int own_rel = cf.own - (int)(32 * mt) - 4 * half;
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (tile_row(r, 0) == own_rel) {
            const float s = __builtin_amdgcn_exp2f(zh[r] + zx[r] * kSplitInv);
            const bool same = code_match<TAG, CodeT>(pcode, (CodeT)codes[tile_row(r, half)]);
            const bool fb = cf.wa == 0.f && a.stats[(size_t)p * 4 + 3] != 0.f;
            const float c1 = cf.wb - inv_num;
            const float w = fb ? (c1 + (same ? 0.f : cf.wb)) : (same ? 0.f : inv_num);
            t[r] = s * w;
          }
        }
      }
      if (ragged) {
        const int lim = (int)(a.n.M - 32 * mt) - 4 * half;
  #pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = (tile_row(r, 0) < lim) ? t[r] : 0.f;
      }
      if (TM == 1) tcache_store(a.tcache_de, (pt - a.spt0) * a.n.MT + mt, lane, t);
    }
    second_gemm<DT>(at + kStd * 1024, lane, t, dacc, dlo);
  }
  fold_lo<DT>(dlo, dacc);
  // dE[p][d] = g_p * kappa * acc[d][p]
  const int64_t pp = 32 * pt + j;
  if (active && pp < a.n.P) {
    const float gk = a.d_nll[pp] * a.kappa;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * (a.dt0 + dt) + tile_row(r, half);
        if (d < a.n.D) a.d_emb[(size_t)pp * a.n.D + d] = gk * dacc[dt][r];
      }
  }
}

// ------------------------------- backward: dE, v2 ---------------------------
// The forward v2 structure applied to the embedding gradient (narrow embeddings, 32-bit codes): a wave
// owns two pixel tiles (B fragments of the similarity recompute, resident) and 2 * DT gradient
// accumulators; a ring slot holds MTB prototype tiles (std fragments for the recompute + T-layout
// fragments for the second contraction); all split terms go into one accumulator per product
// (unscaled residuals); 32-prototype tiles that carry one code take one predicate per pixel;
// the prototype range is walked in chunks of kFwd2TilesPerChunk tiles (row codes resident in LDS),
// workgroup (x, y) takes chunks y, y + gridDim.y, ...; per-y partial gradients leave in accumulator
// layout and are summed in y order by nll_de_finalize (deterministic, independent of P).
template <int KS, int DT, int MTB, bool TAG>
__global__ __launch_bounds__(256, 2) void nll_bwd_de2(NllArgs a) {
  constexpr int TSTD = 2 * KS * 1024;            // std hi | lo blocks of one prototype tile
  constexpr int TILE = TSTD + 4 * DT * 1024;     // + T-layout [DT][2][hi|lo]
  constexpr int SLOT = MTB * TILE;
  constexpr int CODES = kFwd2TilesPerChunk * 32 * 4;
  static_assert((MTB & (MTB - 1)) == 0 && MTB >= 2, "MTB: even power of two");
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  int* const codes_lds = reinterpret_cast<int*>(sm);
  unsigned char* const ring = sm + CODES;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t pt0 = ((int64_t)blockIdx.x * 4 + wv) * 2;
  const int nchunk = (int)((a.n.MT + kFwd2TilesPerChunk - 1) / kFwd2TilesPerChunk);

  half8 bh[2][KS], bl[2][KS];
  int pcode[2], own[2];
  float wa[2], wb[2];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    const int64_t pt = min(pt0 + nb, a.n.PT - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[nb][ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      bl[nb][ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    }
    const int64_t p = min(32 * pt + j, a.n.P - 1);
    pcode[nb] = (int)a.px_code[p];
    const PixelCoef cf = a.coef[32 * pt + j];
    wa[nb] = cf.wa * cf.tscale; wb[nb] = cf.wb * cf.tscale; own[nb] = cf.own;     // (tscale: nll_common.hpp; 0 past P)
  }
  float16v dacc[2][DT];
#pragma unroll
  for (int nb = 0; nb < 2; ++nb)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) dacc[nb][dt][r] = 0.f;

  for (int c = blockIdx.y; c < nchunk; c += gridDim.y) {
    const int64_t mt_lo = (int64_t)c * kFwd2TilesPerChunk;
    const int ntile = (int)(min(a.n.MT, mt_lo + kFwd2TilesPerChunk) - mt_lo);
    const int nstage = (ntile + MTB - 1) / MTB;
    __syncthreads();                                   // the previous chunk's codes / ring are no longer read
    auto stage = [&](int st, int slot) {
      unsigned char* dst = ring + slot * SLOT;
      for (int b = wv; b < MTB * (2 * KS + 4 * DT); b += 4) {     // wave-uniform loop
        const int t = b / (2 * KS + 4 * DT), q = b - t * (2 * KS + 4 * DT);
        const int64_t mt = mt_lo + min(st * MTB + t, ntile - 1);  // past the end: a harmless duplicate
        const void* src;
        if (q < KS) src = a.ph + ((size_t)(mt * KS + q) * 64 + lane) * 8;
        else if (q < 2 * KS) src = a.pl + ((size_t)(mt * KS + (q - KS)) * 64 + lane) * 8;
        else {
          const int u = q - 2 * KS;                    // (dt * 2 + s2) * 2 + hi/lo
          src = ((u & 1) ? a.ptl : a.pth) + (((size_t)mt * DT * 2 + (u >> 1)) * 64 + lane) * 8;
        }
        dma_block(src, dst + (size_t)b * 1024);
      }
    };
    stage(0, 0);
    if (nstage > 1) stage(1, 1);
    for (int i = threadIdx.x; i < 32 * ntile; i += 256) codes_lds[i] = (int)a.pr_code_pad[32 * mt_lo + i];
    const int my_blocks = (MTB * (2 * KS + 4 * DT) - wv + 3) / 4;
    if (nstage > 1) wait_vmcnt(my_blocks); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    unsigned long long uni_lo, uni_hi;
    {
      bool u0 = lane < ntile && 32 * (mt_lo + lane + 1) <= a.n.M;
      bool u1 = lane + 64 < ntile && 32 * (mt_lo + lane + 65) <= a.n.M;
      for (int i = 1; i < 32; ++i) {
        u0 &= codes_lds[32 * min(lane, ntile - 1) + i] == codes_lds[32 * min(lane, ntile - 1)];
        u1 &= codes_lds[32 * min(lane + 64, ntile - 1) + i] == codes_lds[32 * min(lane + 64, ntile - 1)];
      }
      uni_lo = __ballot(u0);
      uni_hi = __ballot(u1);
    }
    const int own_c0 = own[0] - (int)(32 * mt_lo), own_c1 = own[1] - (int)(32 * mt_lo);

    auto tile = [&](int g, auto uni_tag) {
      constexpr bool UNI = decltype(uni_tag)::value;
      const unsigned char* at = ring + ((g / MTB) & 1) * SLOT + (g & (MTB - 1)) * TILE + (size_t)lane * 16;
      const float16v zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int code0 = codes_lds[32 * g];
      const bool ragged = mt_lo + g + 1 == a.n.MT && 32 * a.n.MT > a.n.M;   // uniform: the last, partial tile
      const int lim = (int)(a.n.M - 32 * (mt_lo + g)) - 4 * half;
      // one pixel tile at a time (registers): recompute, weights, split, second contraction
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        float16v z;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const half8 ah = *reinterpret_cast<const half8*>(at + ks * 1024);
          const half8 al = *reinterpret_cast<const half8*>(at + (KS + ks) * 1024);
          z = mfma32(ah, bh[nb][ks], ks == 0 ? zero : z);
          z = mfma32(ah, bl[nb][ks], z);
          z = mfma32(al, bh[nb][ks], z);
        }
        float t[16];
        const float wu = code_match<TAG, int>(pcode[nb], code0) ? wa[nb] : wb[nb];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float sv = __builtin_amdgcn_exp2f(z[r]);
          if constexpr (UNI) t[r] = sv * wu;
          else t[r] = sv * (code_match<TAG, int>(pcode[nb], codes_lds[32 * g + tile_row(r, half)]) ? wa[nb] : wb[nb]);
        }
        const int own_c = nb ? own_c1 : own_c0;
        if (__any((own_c >> 5) == g)) {                      // own prototype in this tile (rare): its weight follows
          const int own_rel = own_c - 32 * g - 4 * half;     // another formula and its T may exceed 1 (s_own / num):
#pragma unroll                                               // left out here, added by nll_de_finalize (own_term_kernel)
          for (int r = 0; r < 16; ++r)
            if (tile_row(r, 0) == own_rel) t[r] = 0.f;
        }
        if (ragged) {
#pragma unroll
          for (int r = 0; r < 16; ++r) t[r] = (tile_row(r, 0) < lim) ? t[r] : 0.f;
        }
        // T (|T| <= 1) -> (hi, lo) f16 fragments of the second contraction, residual unscaled
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          half8 th, tl;
          // two values per instruction: hi = rtz(t) (v_cvt_pkrtz), residual t - hi by a mixed-precision
          // fma, lo = rtz(residual): |t - hi - lo| <= 2^-21 |t|
#pragma unroll
          for (int e = 0; e < 8; e += 2) {
            const float t0 = t[8 * s2 + e], t1 = t[8 * s2 + e + 1];
            const half2v hh = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(t0, t1));
            float r0, r1;                                  // t - (float)hi, the f16 half read in place
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh), "v"(t0));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh), "v"(t1));
            const half2v ll = __builtin_bit_cast(half2v, __builtin_amdgcn_cvt_pkrtz(r0, r1));
            th[e] = hh[0]; th[e + 1] = hh[1];
            tl[e] = ll[0]; tl[e + 1] = ll[1];
          }
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const unsigned char* blk = at + TSTD + (size_t)((dt * 2 + s2) * 2) * 1024;
            const half8 qh = *reinterpret_cast<const half8*>(blk);
            const half8 ql = *reinterpret_cast<const half8*>(blk + 1024);
            dacc[nb][dt] = mfma32(qh, th, dacc[nb][dt]);
            dacc[nb][dt] = mfma32(qh, tl, dacc[nb][dt]);
            dacc[nb][dt] = mfma32(ql, th, dacc[nb][dt]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);               // one pixel tile after the other (register budget)
      }
    };

    for (int g = 0; g < ntile; ++g) {
      if (g > 0 && (g & (MTB - 1)) == 0) {             // stage g / MTB: landed for every wave; refill the other slot
        const int st = g / MTB;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        if (st + 1 < nstage) stage(st + 1, (st + 1) & 1);
      }
      if (((g < 64 ? uni_lo >> g : uni_hi >> (g - 64)) & 1ull) != 0) tile(g, std::true_type{});
      else tile(g, std::false_type{});
    }
  }

  // partial gradients in accumulator layout: [y][pt][dt][r][lane]
#pragma unroll
  for (int nb = 0; nb < 2; ++nb) {
    if (pt0 + nb < a.n.PT) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          a.partial_de[((((size_t)blockIdx.y * a.n.PT + pt0 + nb) * DT + dt) * 16 + r) * 64 + lane] = dacc[nb][dt][r];
    }
  }
}

// dE[p][d] = g_p * kappa / scale * sum_y partial[y][pt][dt][r][lane]  (y order; coalesced row writes)
// own_term != NULL (v2 / v3 kernels): + own_term[p] * protos[own[p]] -- the own prototype's term, which those kernels
// leave out of their tiles (partial sums are in units of 8 tscale_p x prototype: factor = kappa / 8, own_scale = 8)
__global__ __launch_bounds__(256) void nll_de_finalize(const float* __restrict__ partial, int ny, int64_t P,
                                                       int64_t PT, int D, int DT, const float* __restrict__ d_nll,
                                                       float factor, float* __restrict__ d_emb,
                                                       const float* __restrict__ own_term,
                                                       const int64_t* __restrict__ own,
                                                       const float* __restrict__ protos, float own_scale,
                                                       const PixelCoef* __restrict__ coef) {
  // The partial sums of a pixel tile are ONE contiguous block of DT * 1024 floats per y in accumulator order
  // [dt][r][half][pixel]: read them as such (16-byte loads; the per-output gather of round 3 touched a 32-byte
  // sector per float: 219 us for 0.4 GB), transpose through LDS, write rows.  Summation over y left to right.
  extern __shared__ float tile[];                    // [32 pixels][DT * 32 + 1]
  const int64_t pt = blockIdx.x;
  const int width = DT * 32, pitch = width + 1;
  const int n4 = DT * 256;                           // float4 units of the block
  for (int u = threadIdx.x; u < n4; u += 256) {
    float4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int y = 0; y < ny; ++y)
      acc += *reinterpret_cast<const float4v*>(partial + (((size_t)y * PT + pt) * DT) * 1024 + 4 * (size_t)u);
    const int e = 4 * u;                             // element (dt, r, hf, i .. i + 3)
    const int dt = e >> 10, r = (e >> 6) & 15, hf = (e >> 5) & 1, i = e & 31;
    const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * hf;      // (tile_row(r, hf))
#pragma unroll
    for (int q = 0; q < 4; ++q) tile[(i + q) * pitch + d] = acc[q];
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 32 * D; o += 256) {
    const int i = o / D, d = o - i * D;
    const int64_t p = 32 * pt + i;
    if (p >= P) continue;
    float acc = tile[i * pitch + d];
    const float ts = coef[p].tscale;                 // the pixel's T scale (a power of two)
    if (own_term) {
      const float ot = own_term[p];
      if (ot != 0.f) acc += ot * own_scale * ts * protos[(size_t)own[p] * D + d];
    }
    d_emb[(size_t)p * D + d] = acc * d_nll[p] * (factor / ts);
  }
}

// ------------------------------- backward: dPr -----------------------------
// grid (ceil(MTg/4), chunks): the 4 waves of a workgroup own 4 consecutive
// prototype tiles (B operand, resident) and share one stream of pixel tiles
// (std fragments, per-pixel coefficients + codes, transposed fragments) through
// the LDS ring.  dPr^T[d][proto] += ET[d][p] * T'[p][proto]; every wave owns its
// accumulators, which leave with one fp32 atomic per element per chunk.
template <int KS, int DT, bool TAG, bool C32, int TM = 0>
__global__ __launch_bounds__(256) void nll_bwd_dp(NllArgs a) {
  using CodeT = code_t<C32>;
  constexpr int kStd = TM == 2 ? 0 : 2 * KS + 2; // std hi/lo, coef, codes (not needed when T comes from the cache)
  constexpr int NBLK = kStd + 4 * DT;            // ... + T-layout blocks
  constexpr int SLOT = NBLK * 1024;
  const int DEPTH = a.depth;                     // LDS ring slots (2 or 3)
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t mt = min((int64_t)blockIdx.x * 4 + wv, a.n.MT - 1);
  const bool active = (int64_t)blockIdx.x * 4 + wv < a.mt_grad;
  const int64_t per = (a.spt1 - a.spt0 + a.chunks - 1) / a.chunks;
  const int64_t pt_lo = a.spt0 + (int64_t)blockIdx.y * per;
  const int64_t pt_hi = min(a.spt1, pt_lo + per);

  half8 bh[TM == 2 ? 1 : KS], bl[TM == 2 ? 1 : KS];
  if constexpr (TM != 2) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[ks] = *reinterpret_cast<const half8*>(a.ph + (((size_t)mt * KS + ks) * 64 + lane) * 8);
      bl[ks] = *reinterpret_cast<const half8*>(a.pl + (((size_t)mt * KS + ks) * 64 + lane) * 8);
    }
  }
  const int col = (int)(32 * mt) + j;                 // prototype of this lane's column
  const bool col_ok = col < a.n.M;
  const CodeT ccode = (CodeT)a.pr_code_pad[col];

  float16v dacc[DT];
  LoAcc<DT> dlo;
  zero_lo<DT>(dlo);
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[dt][r] = 0.f;

  auto stage = [&](int64_t pt, int slot) {
    unsigned char* dst = sm + slot * SLOT;
    for (int b = wv; b < NBLK; b += 4) {
      const void* src;
      if (b < kStd && b < KS) src = a.eh + ((size_t)(pt * KS + b) * 64 + lane) * 8;
      else if (b < kStd && b < 2 * KS) src = a.el + ((size_t)(pt * KS + (b - KS)) * 64 + lane) * 8;
      else if (b < kStd && b == 2 * KS) src = a.coef + 32 * pt + min(lane, 31);          // 32 x 16 B
      else if (b < kStd) src = a.px_code_pad + 32 * pt + 2 * min(lane, 15);
      else {
        const int q = b - kStd;
        const size_t f = ((size_t)pt * a.dt_all + a.dt0) * 2 + (q >> 1);
        src = ((q & 1) ? a.etl : a.eth) + (f * 64 + lane) * 8;
      }
      dma_block(src, dst + (size_t)b * 1024);
    }
  };
  const int my_blocks = (NBLK - wv + 3) / 4;
  const int64_t ntile = pt_hi - pt_lo;
  for (int64_t t0 = 0; t0 < DEPTH - 1 && t0 < ntile; ++t0) stage(pt_lo + t0, (int)t0);
  int slot = -1;
  for (int64_t it = 0; it < ntile; ++it) {
    const int64_t pt = pt_lo + it;
    slot = slot + 1 == DEPTH ? 0 : slot + 1;
    if (it + 1 < ntile && DEPTH > 2) wait_vmcnt(my_blocks * (int)min((int64_t)(DEPTH - 2), ntile - 1 - it));
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (it + DEPTH - 1 < ntile) stage(pt + DEPTH - 1, slot == 0 ? DEPTH - 1 : slot - 1);
    const unsigned char* at = sm + slot * SLOT;
    const PixelCoef* coef = reinterpret_cast<const PixelCoef*>(at + 2 * KS * 1024);
    const int64_t* codes = reinterpret_cast<const int64_t*>(at + (2 * KS + 1) * 1024);
    float t[16];
    if constexpr (TM == 2) {
      tcache_load(a.tcache_dp, mt * (a.spt1 - a.spt0) + (pt - a.spt0), lane, t);
    } else {
      // z'[row = pixel][col = prototype]
      float16v zh, zx;
      zgemm_lds<KS>(at, lane, bh, bl, zh, zx);
      bool own_here = false;
  #pragma unroll
      for (int r = 0; r < 16; ++r) {
        const PixelCoef c = coef[tile_row(r, half)];
        const float s = __builtin_amdgcn_exp2f(zh[r] + zx[r] * kSplitInv);
        const bool same = code_match<TAG, CodeT>((CodeT)codes[tile_row(r, half)], ccode);
        float w = same ? c.wa : c.wb;
        w = c.tscale != 0.f ? w : 0.f;
        own_here |= (c.own == col);
        t[r] = s * w;
      }
      if (__any(own_here)) {                               // rare: a pixel whose own prototype is here
  #pragma unroll
        for (int r = 0; r < 16; ++r) {
          const PixelCoef c = coef[tile_row(r, half)];
          if (c.tscale != 0.f && c.own == col) {
            const int64_t pr = 32 * pt + tile_row(r, half);
            const float4v st = *reinterpret_cast<const float4v*>(a.stats + (size_t)pr * 4);
            const float s = __builtin_amdgcn_exp2f(zh[r] + zx[r] * kSplitInv);
            const bool same = code_match<TAG, CodeT>((CodeT)codes[tile_row(r, half)], ccode);
            const float inv_num = 1.0f / st[0];
            const float w = st[3] != 0.f ? ((c.wb - inv_num) + (same ? 0.f : c.wb))
                                         : (same ? 0.f : inv_num);
            t[r] = s * w;
          }
        }
      }
      if (!col_ok) {
  #pragma unroll
        for (int r = 0; r < 16; ++r) t[r] = 0.f;
      }
      if (TM == 1) tcache_store(a.tcache_dp, mt * (a.spt1 - a.spt0) + (pt - a.spt0), lane, t);
    }
    second_gemm<DT>(at + kStd * 1024, lane, t, dacc, dlo);
  }
  fold_lo<DT>(dlo, dacc);
  if (active && col_ok) {
    const float gs = a.gscale[0], igs = 1.0f / gs;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * (a.dt0 + dt) + tile_row(r, half);
        if (d < a.n.D) dpr_add(a, (size_t)col * a.n.D + d, dacc[dt][r] * gs, igs);
      }
  }
}

__global__ void pad_codes_kernel(const int64_t* in, int64_t n, int64_t n_pad, int64_t* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n_pad) out[i] = i < n ? in[i] : 0;
}

__global__ void rowscale_kernel(const float* g, float kappa, int64_t n, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = g[i] * kappa;
}

// ------------------------------- host --------------------------------------
struct NllWs {
  size_t eh, el, ph, pl, pth, ptl, eth, etl, gscale, rowscale, codes, pxcodes, coef, ownterm, dprows, tde, tdp, partial, partial_de, dp64, total;
};

// Wide embeddings: pixel tiles per strip of the backward.  The kept weight tiles cost strip x M x 4 B per
// cache; a strip is as long as SPML_NLL_TCACHE_MB (default 8192 MB per cache) allows, so that the
// workspace stays bounded when the global prototype count grows with the number of ranks (P x M x 8 B
// would be 130 GB per GPU for BASELINE config 5 on 8 ranks).  A function of (P, M) only.
inline int64_t tcache_strip_tiles(const NllDims& n) {
  int64_t mb = 8192;
  if (const char* e = getenv("SPML_NLL_TCACHE_MB")) { const long long v = atoll(e); if (v > 0) mb = v; }
  int64_t tiles = (mb << 20) / (n.MT * 4096);
  tiles = tiles / 8 * 8;
  if (tiles < 8) tiles = 8;
  return tiles < n.PT ? tiles : n.PT;
}

inline int64_t fwd2_chunks(const NllDims& n) { return (n.MT + kFwd2TilesPerChunk - 1) / kFwd2TilesPerChunk; }
// grid rows of the v2 backward kernels: chunk c is taken by row c mod rows (a function of M alone)
inline int bwd2_rows(const NllDims& n) { return (int)std::min<int64_t>(fwd2_chunks(n), 8); }

NllWs nll_ws(const NllDims& n) {
  NllWs w{};
  size_t o = 0;
  const size_t e_std = (size_t)n.PT * n.KS * 512 * 2, p_std = (size_t)n.MT * n.KS * 512 * 2;
  const size_t e_t = (size_t)n.PT * n.DT * 2 * 512 * 2, p_t = (size_t)n.MT * n.DT * 2 * 512 * 2;
  w.eh = o; o = align_up(o + e_std, 256);
  w.el = o; o = align_up(o + e_std, 256);
  w.ph = o; o = align_up(o + p_std, 256);
  w.pl = o; o = align_up(o + p_std, 256);
  w.pth = o; o = align_up(o + p_t, 256);
  w.ptl = o; o = align_up(o + p_t, 256);
  w.eth = o; o = align_up(o + e_t, 256);
  w.etl = o; o = align_up(o + e_t, 256);
  w.gscale = o; o = align_up(o + 16, 256);
  w.rowscale = o; o = align_up(o + (size_t)n.P * 4, 256);
  w.codes = o; o = align_up(o + (size_t)n.MT * 32 * 8, 256);
  w.pxcodes = o; o = align_up(o + (size_t)n.PT * 32 * 8, 256);
  w.coef = o; o = align_up(o + (size_t)n.PT * 32 * 16, 256);
  w.ownterm = o; o = align_up(o + (size_t)n.PT * 32 * 4, 256);
  w.dprows = o; o = align_up(o + nll_dp3_rows_bytes(n.PT), 256);
  w.tde = w.tdp = 0;
  if (n.KS > 17) {           // several d-chunk launches: the T tiles of the first one are kept (see NllArgs),
    const size_t tiles = (size_t)tcache_strip_tiles(n) * n.MT * 4096;     // one strip of pixel tiles at a time
    w.tde = o; o = align_up(o + tiles, 256);
    w.tdp = o; o = align_up(o + tiles, 256);
  }
  w.partial = 0;
  w.partial_de = 0;
  if (n.KS <= 5) {           // narrow embeddings: per-chunk partial sums of the v2 forward
    w.partial = o; o = align_up(o + (size_t)fwd2_chunks(n) * n.PT * 32 * 16, 256);
  }
  if (n.KS <= 4) {           // ... and the per-y partial gradients of the v2 dE kernel (the same bytes: never both alive)
    const size_t need = (size_t)bwd2_rows(n) * n.PT * n.DT * 4096;
    w.partial_de = w.partial;
    o = align_up(std::max(o, w.partial + need), 256);
  }
  // deterministic mode: the fixed-point prototype gradient, [MT * 32][D] 64-bit sums
  w.dp64 = 0;
  if (deterministic_mode()) { w.dp64 = o; o = align_up(o + (size_t)n.MT * 32 * n.D * 8, 256); }
  w.total = o;
  return w;
}

// fixed-point sums -> d_protos (+=), and the sums back to zero
__global__ void nll_dpr_from_fix(long long* __restrict__ acc, int64_t n, const float* __restrict__ gscale,
                                 float* __restrict__ d_protos) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long q = acc[i];
  if (q != 0) d_protos[i] += (float)((double)q * kDetFixInv) * gscale[0];
}

int ks_bucket(int ks) {
  if (ks <= 2) return 2;
  if (ks <= 3) return 3;
  if (ks <= 4) return 4;
  if (ks <= 5) return 5;
  if (ks <= 9) return 9;
  if (ks <= 17) return 17;
  if (ks <= 33) return 33;      // 272 < D <= 528 (e.g. the 512-d stress configuration)
  return 0;
}

// d-tiles (32 channels) of the backward's second contraction handled per launch.  Up to
// KS = 17 one launch covers all of them; for KS = 33 the LDS ring (2 slots of prototype /
// pixel fragments + the transposed fragments of the chunk) only leaves room for 3, so the
// backward kernels are launched once per chunk of 96 channels (each recomputes the
// similarity tile -- "works on the stress configuration", not tuned for it).
int dt_per_launch(int ks) { return ks <= 17 ? (ks + 1) / 2 : 3; }

void launch_prep_std(const float* x, int64_t R, int D, int KS, float scale, _Float16* h,
                     _Float16* l, hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * KS;
  hipLaunchKernelGGL(prep_std<false>, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, KS,
                     scale, h, l);
}
void launch_prep_raw(const float* x, int64_t R, int D, int KS, float scale, _Float16* h,
                     _Float16* l, hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * KS;
  hipLaunchKernelGGL(prep_std<true>, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, KS,
                     scale, h, l);
}
void launch_prep_T(const float* x, int64_t R, int D, int DT, const float* rowscale,
                   const float* gscale, _Float16* h, _Float16* l, hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * DT * 2;
  hipLaunchKernelGGL(prep_T<false>, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, DT,
                     rowscale, gscale, 1.0f, h, l);
}
void launch_prep_T_raw(const float* x, int64_t R, int D, int DT, const float* rowscale,
                       const float* gscale, float scale, _Float16* h, _Float16* l, hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * DT * 2;
  hipLaunchKernelGGL(prep_T<true>, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, DT,
                     rowscale, gscale, scale, h, l);
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" size_t spml_segsort_nll_workspace_bytes(int64_t P, int64_t M, int D) {
  if (P < 0 || M <= 0 || D <= 0) return 0;
  NllDims n = nll_dims(P, M, D);
  if (ks_bucket(n.KS)) {
    n.KS = ks_bucket(n.KS);
    const int per = dt_per_launch(n.KS);
    n.DT = ((n.KS + 1) / 2 + per - 1) / per * per;      // whole chunks
  }
  return nll_ws(n).total;
}

static int nll_common(bool backward, const float* emb, const int64_t* own,
                      const int64_t* px_code, int64_t P, const float* protos,
                      const int64_t* pr_code, int64_t M, int D, float kappa, int mode,
                      float* nll, float* stats, const float* d_nll, float* d_emb,
                      float* d_protos, int64_t m_grad, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!emb || !own || !px_code || !protos || !pr_code || P < 0 || M <= 0 || D <= 0 || !stats)
    return SPML_ERR_INVALID_ARG;
  if (mode < 0 || mode > (SPML_NLL_TAGSET | SPML_NLL_PLAIN | SPML_NLL_CODE32)) return SPML_ERR_INVALID_ARG;
  if (backward ? (!d_nll || !d_emb || !d_protos) : !nll) return SPML_ERR_INVALID_ARG;
  NllDims n = nll_dims(P, M, D);
  const int ksb = ks_bucket(n.KS);
  if (!ksb) return SPML_ERR_UNSUPPORTED;                    // D <= 528
  n.KS = ksb;
  const int dt_launch = dt_per_launch(ksb);                 // template tile counts
  n.DT = ((ksb + 1) / 2 + dt_launch - 1) / dt_launch * dt_launch;
  const NllWs w = nll_ws(n);
  if (!ws || ws_bytes < w.total) return SPML_ERR_WORKSPACE;
  if (P == 0) return SPML_OK;
  unsigned char* b = static_cast<unsigned char*>(ws);
  NllArgs a{};
  a.n = n;
  _Float16* eh = reinterpret_cast<_Float16*>(b + w.eh);
  _Float16* el = reinterpret_cast<_Float16*>(b + w.el);
  _Float16* ph = reinterpret_cast<_Float16*>(b + w.ph);
  _Float16* pl = reinterpret_cast<_Float16*>(b + w.pl);
  a.eh = eh; a.el = el; a.ph = ph; a.pl = pl;
  a.own = own; a.px_code = px_code; a.pr_code = pr_code;
  a.kappa = kappa;
  a.kappa_log2e = kappa * 1.4426950408889634f;
  a.mode = mode; a.nll = nll; a.stats = stats; a.d_nll = d_nll; a.d_emb = d_emb;
  a.d_protos = d_protos;
  int64_t* codes = reinterpret_cast<int64_t*>(b + w.codes);
  a.pr_code_pad = codes;
  hipLaunchKernelGGL(pad_codes_kernel, dim3((unsigned)((n.MT * 32 + 255) / 256)), dim3(256), 0, s,
                     pr_code, M, n.MT * 32, codes);
  // v2 forward (narrow embeddings, 32-bit codes; SPML_NLL_FWD2=0 selects the round-2 kernels): unscaled
  // residuals, pixels x 2^3 and prototypes x kappa * log2(e) * 2^-3 (exact powers of two)
  const char* env2 = getenv("SPML_NLL_FWD2");
  const bool fwd2 = !backward && n.KS <= 5 && (mode & SPML_NLL_CODE32) && !(env2 && env2[0] == '0');
  if (fwd2) {
    launch_prep_raw(emb, P, D, n.KS, 8.0f, eh, el, s);
    launch_prep_raw(protos, M, D, n.KS, a.kappa_log2e * 0.125f, ph, pl, s);
    a.partial = reinterpret_cast<float*>(b + w.partial);
    const int64_t chunks = fwd2_chunks(n);
    const unsigned groups = (unsigned)((n.PT + 7) / 8);         // 4 waves x 2 pixel tiles
#define SPML_FWD2(KS_, MTB_)                                                                        \
    {                                                                                               \
      constexpr int LDS2 = kFwd2TilesPerChunk * 128 + 2 * MTB_ * 2 * KS_ * 1024;                                         \
      if (mode & SPML_NLL_TAGSET) {                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_fwd2<KS_, MTB_, true>),        \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);                \
        hipLaunchKernelGGL((nll_fwd2<KS_, MTB_, true>), dim3(groups, (unsigned)chunks), dim3(256), LDS2, s, a); \
      } else {                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_fwd2<KS_, MTB_, false>),       \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS2);                \
        hipLaunchKernelGGL((nll_fwd2<KS_, MTB_, false>), dim3(groups, (unsigned)chunks), dim3(256), LDS2, s, a); \
      }                                                                                             \
    }
    switch (n.KS) {
      case 2: SPML_FWD2(2, 4); break;
      case 3: SPML_FWD2(3, 4); break;
      case 4: SPML_FWD2(4, 4); break;
      default: SPML_FWD2(5, 2); break;
    }
#undef SPML_FWD2
    hipLaunchKernelGGL(nll_finalize, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, a.partial, (int)chunks, P,
                       n.PT * 32, (mode & SPML_NLL_PLAIN) ? 1 : 0, nll, stats);
    return launch_status();
  }
  // kappa * log2(e) is folded into the prototype fragments: the MFMA result is the exp2 argument
  launch_prep_std(emb, P, D, n.KS, 1.0f, eh, el, s);
  launch_prep_std(protos, M, D, n.KS, a.kappa_log2e, ph, pl, s);

#define SPML_KS_SWITCH(MACRO)             \
  switch (n.KS) {                         \
    case 2: MACRO(2); break;              \
    case 3: MACRO(3); break;              \
    case 4: MACRO(4); break;              \
    case 5: MACRO(5); break;              \
    case 9: MACRO(9); break;              \
    case 17: MACRO(17); break;            \
    case 33: MACRO(33); break;            \
    default: return SPML_ERR_UNSUPPORTED; \
  }
  if (!backward) {
#define SPML_FWD_ONE(KS_, TAG_, C32_)                                                      \
    if (((mode & SPML_NLL_TAGSET) != 0) == TAG_ && ((mode & SPML_NLL_CODE32) != 0) == C32_) { \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_fwd<KS_, NB, TAG_, C32_>), \
                                hipFuncAttributeMaxDynamicSharedMemorySize, lds_f);        \
      hipLaunchKernelGGL((nll_fwd<KS_, NB, TAG_, C32_>), dim3((unsigned)((waves + 3) / 4)), dim3(256), lds_f, s, a); \
    }
#define SPML_FWD(KS_)                                                                      \
  {                                                                                        \
    constexpr int NB = 1;                                                                  \
    constexpr int SLOT_F = (2 * KS_ + 1) * 1024;                                           \
    /* narrow embeddings: a 2-slot ring (18 KB) lets five workgroups share a CU -- 5-7 % faster than 4 slots */ \
    a.depth_fwd = KS_ <= 5 ? 2 : (4 * SLOT_F <= 160 * 1024 ? 4 : 2);                       \
    if (const char* e_ = getenv("SPML_NLL_DEPTH_FWD")) { const int v_ = atoi(e_); if (v_ >= 2 && v_ <= 4) a.depth_fwd = v_; } \
    const int lds_f = a.depth_fwd * SLOT_F;                                                \
    const int64_t waves = (n.PT + NB - 1) / NB;                                            \
    SPML_FWD_ONE(KS_, true, true) SPML_FWD_ONE(KS_, true, false)                           \
    SPML_FWD_ONE(KS_, false, true) SPML_FWD_ONE(KS_, false, false)                         \
  }
    SPML_KS_SWITCH(SPML_FWD)
#undef SPML_FWD
#undef SPML_FWD_ONE
    return launch_status();
  }

  // ---- backward ----
  _Float16* pth = reinterpret_cast<_Float16*>(b + w.pth);
  _Float16* ptl = reinterpret_cast<_Float16*>(b + w.ptl);
  _Float16* eth = reinterpret_cast<_Float16*>(b + w.eth);
  _Float16* etl = reinterpret_cast<_Float16*>(b + w.etl);
  float* gscale = reinterpret_cast<float*>(b + w.gscale);
  float* rowscale = reinterpret_cast<float*>(b + w.rowscale);
  a.pth = pth; a.ptl = ptl; a.eth = eth; a.etl = etl; a.gscale = gscale;
  // deterministic mode: the prototype gradient is accumulated in fixed point (relative to gscale) and converted
  // once, at the end of the call
  a.d_protos64 = nullptr;
  if (w.dp64) {
    a.d_protos64 = reinterpret_cast<long long*>(b + w.dp64);
    if (hipMemsetAsync(a.d_protos64, 0, (size_t)M * D * 8, s) != hipSuccess) return SPML_ERR_LAUNCH;
  }
  auto finish = [&](int rc) -> int {
    if (rc != SPML_OK) return rc;
    if (a.d_protos64)
      hipLaunchKernelGGL(nll_dpr_from_fix, dim3((unsigned)(((int64_t)M * D + 255) / 256)), dim3(256), 0, s,
                         a.d_protos64, (int64_t)M * D, (const float*)gscale, d_protos);
    return launch_status();
  };
  hipLaunchKernelGGL(max_abs_pow2, dim3(1), dim3(1024), 0, s, d_nll, P, gscale);
  hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, d_nll,
                     kappa, P, rowscale);
  // gscale bounds |g|; kappa is folded into rowscale (kappa * g / gscale stays O(kappa))
  launch_prep_T(protos, M, D, n.DT, nullptr, nullptr, pth, ptl, s);
  // (the pipelined kernels of nll_de3.hip / nll_dp3.hip -- narrow embeddings, 32-bit codes -- prepare their own)
  const auto env_off = [](const char* name) { const char* e = getenv(name); return e && e[0] == '0'; };
  const bool v3 = n.KS == 4 && (mode & SPML_NLL_CODE32) && !env_off("SPML_NLL_BWD2") && !env_off("SPML_NLL_DE3");
  const bool dp3 = v3 && !env_off("SPML_NLL_DP3");
  if (!dp3) launch_prep_T(emb, P, D, n.DT, rowscale, gscale, eth, etl, s);
  int64_t* pxcodes = reinterpret_cast<int64_t*>(b + w.pxcodes);
  PixelCoef* coef = reinterpret_cast<PixelCoef*>(b + w.coef);
  a.px_code_pad = pxcodes;
  a.coef = coef;
  a.tcache_de = w.tde ? reinterpret_cast<float*>(b + w.tde) : nullptr;
  a.tcache_dp = w.tdp ? reinterpret_cast<float*>(b + w.tdp) : nullptr;
  hipLaunchKernelGGL(pad_codes_kernel, dim3((unsigned)((n.PT * 32 + 255) / 256)), dim3(256), 0, s,
                     px_code, P, n.PT * 32, pxcodes);
  if (mode & SPML_NLL_TAGSET)
    hipLaunchKernelGGL(coef_kernel<true>, dim3((unsigned)((n.PT * 32 + 255) / 256)), dim3(256), 0, s, stats, own, px_code,
                       pr_code, M, P, n.PT * 32, coef);
  else
    hipLaunchKernelGGL(coef_kernel<false>, dim3((unsigned)((n.PT * 32 + 255) / 256)), dim3(256), 0, s, stats, own, px_code,
                       pr_code, M, P, n.PT * 32, coef);
  // prototypes [0, m_grad) receive a gradient (the rest, e.g. a detached memory bank, is skipped)
  const int64_t mg = m_grad < 0 || m_grad > M ? M : m_grad;
  a.mt_grad = (mg + 31) / 32;
  const int64_t mgroups = (a.mt_grad + 3) / 4;
  // pixel chunks so that the grid has a few thousand workgroups
  int64_t chunks = mgroups > 0 ? (2048 + mgroups - 1) / mgroups : 1;
  if (chunks > (n.PT + 7) / 8) chunks = (n.PT + 7) / 8;
  if (chunks < 1) chunks = 1;
  if (chunks > 65535) chunks = 65535;
  a.chunks = (int)chunks;
  a.spt0 = 0;
  a.spt1 = n.PT;

  // v2 dE kernel (narrow embeddings, 32-bit codes; SPML_NLL_BWD2=0: round-2 kernels): unscaled residuals,
  // pixels x 2^3, prototypes x kappa * log2(e) * 2^-3, transposed prototypes x 2^3 (folded back below)
  const char* envb = getenv("SPML_NLL_BWD2");
  const bool de2 = n.KS <= 4 && (mode & SPML_NLL_CODE32) && !(envb && envb[0] == '0');
  if (de2) {
    launch_prep_raw(emb, P, D, n.KS, 8.0f, eh, el, s);
    launch_prep_raw(protos, M, D, n.KS, a.kappa_log2e * 0.125f, ph, pl, s);
    launch_prep_T_raw(protos, M, D, n.DT, nullptr, nullptr, 8.0f, pth, ptl, s);
    a.partial_de = reinterpret_cast<float*>(b + w.partial_de);
    const int rows = bwd2_rows(n);
    const unsigned groups = (unsigned)((n.PT + 7) / 8);
#define SPML_DE2(KS_, DT_)                                                                          \
    {                                                                                               \
      constexpr int LDSB = kFwd2TilesPerChunk * 128 + 2 * 2 * (2 * KS_ + 4 * DT_) * 1024;           \
      if (mode & SPML_NLL_TAGSET) {                                                                 \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_de2<KS_, DT_, 2, true>),   \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);                \
        hipLaunchKernelGGL((nll_bwd_de2<KS_, DT_, 2, true>), dim3(groups, (unsigned)rows), dim3(256), LDSB, s, a); \
      } else {                                                                                      \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_de2<KS_, DT_, 2, false>),  \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);                \
        hipLaunchKernelGGL((nll_bwd_de2<KS_, DT_, 2, false>), dim3(groups, (unsigned)rows), dim3(256), LDSB, s, a); \
      }                                                                                             \
    }
    const bool use3 = v3;
    float* own_term = reinterpret_cast<float*>(b + w.ownterm);
    if (mode & SPML_NLL_TAGSET)
      hipLaunchKernelGGL(own_term_kernel<true>, dim3((unsigned)((n.PT * 32 + 255) / 256)), dim3(256), 0, s, stats, own,
                         px_code, pr_code, emb, protos, P, n.PT * 32, M, D, a.kappa_log2e, own_term);
    else
      hipLaunchKernelGGL(own_term_kernel<false>, dim3((unsigned)((n.PT * 32 + 255) / 256)), dim3(256), 0, s, stats, own,
                         px_code, pr_code, emb, protos, P, n.PT * 32, M, D, a.kappa_log2e, own_term);
    if (use3) {
      const int rc3 = nll_launch_bwd_de3(a, rows, s);
      if (rc3 != SPML_OK) return rc3;
    } else {
      switch (n.KS) {
        case 2: SPML_DE2(2, 1); break;
        case 3: SPML_DE2(3, 2); break;
        default: SPML_DE2(4, 2); break;
      }
    }
#undef SPML_DE2
    hipLaunchKernelGGL(nll_de_finalize, dim3((unsigned)n.PT), dim3(256), (size_t)32 * (n.DT * 32 + 1) * sizeof(float), s, a.partial_de, rows, P, n.PT, D, n.DT,
                       d_nll, kappa * 0.125f, d_emb, (const float*)own_term, own, protos, 8.0f, (const PixelCoef*)coef);
    if (dp3) {
      // prototype gradient on the pipelined kernel too: it takes the same std fragments; the transposed pixel
      // fragments with unscaled residuals, x 2^8 (nll_dp3.hip)
      hipLaunchKernelGGL(rowscale_ts_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, d_nll, kappa,
                         (const PixelCoef*)coef, P, rowscale);
      launch_prep_T_raw(emb, P, D, n.DT, rowscale, gscale, 16.0f, eth, etl, s);
      return finish(nll_launch_bwd_dp3(a, own_term, emb, reinterpret_cast<float*>(b + w.dprows), s));
    }
    // the round-2 dPr kernel takes the pre-scaled fragments (a v2 form of it -- two prototype tiles per wave, pixel
    // tiles streamed -- measured slower: 19 vs 15.5 ms for the live third at M = 139 k, profiles/r03_nll.md)
    launch_prep_std(emb, P, D, n.KS, 1.0f, eh, el, s);
    launch_prep_std(protos, M, D, n.KS, a.kappa_log2e, ph, pl, s);
  }
  a.skip_de = de2 ? 1 : 0;

#define SPML_BWD_LAUNCH(KS_, DT_, TAG_, C32_, TM_)                                                    \
  {                                                                                              \
    constexpr int STD_DE = TM_ == 2 ? 0 : 2 * KS_ + 1, STD_DP = TM_ == 2 ? 0 : 2 * KS_ + 2;      \
    constexpr int SLOT_DE = (STD_DE + 4 * DT_) * 1024, SLOT_DP = (STD_DP + 4 * DT_) * 1024;      \
    /* narrow embeddings: 2 slots (36 KB) -> three workgroups of the dE kernel per CU (3-4 % faster); \
       otherwise 3 slots while two workgroups still fit */                                      \
    a.depth = KS_ <= 5 ? 2 : (3 * SLOT_DP <= 80 * 1024 ? 3 : 2);                                 \
    if (const char* e_ = getenv("SPML_NLL_DEPTH_BWD")) { const int v_ = atoi(e_); if (v_ >= 2 && v_ <= 3) a.depth = v_; } \
    if (2 * SLOT_DP > 160 * 1024) return SPML_ERR_UNSUPPORTED;                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_de<KS_, DT_, TAG_, C32_, TM_>),   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, a.depth * SLOT_DE);    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_dp<KS_, DT_, TAG_, C32_, TM_>),   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, a.depth * SLOT_DP);    \
    if (!a.skip_de)                                                                              \
      hipLaunchKernelGGL((nll_bwd_de<KS_, DT_, TAG_, C32_, TM_>), dim3((unsigned)((a.spt1 - a.spt0 + 3) / 4)), \
                         dim3(256), a.depth * SLOT_DE, s, a);                                    \
    if (mgroups > 0)                                                                             \
      hipLaunchKernelGGL((nll_bwd_dp<KS_, DT_, TAG_, C32_, TM_>), dim3((unsigned)mgroups, (unsigned)a.chunks), \
                         dim3(256), a.depth * SLOT_DP, s, a);                                    \
  }
#define SPML_BWD_DT(KS_, DT_, TAG_, C32_)                                                              \
  {                                                                                              \
    a.dt_all = n.DT;                                                                             \
    if constexpr (KS_ > 17) {                                                                    \
      /* wide embeddings: the first launch (3 d-tiles) computes the weight tiles and keeps them, \
         the others read them back and contract 7 d-tiles each (no similarity GEMM, no resident  \
         pixel fragments: the registers go to the accumulators) */                               \
      if (a.tcache_de != nullptr && 32 * DT_ < D) {                                              \
        const int64_t strip = tcache_strip_tiles(n);                                             \
        for (int64_t t0 = 0; t0 < n.PT; t0 += strip) {       /* bounded workspace: strip by strip */ \
          a.spt0 = t0;                                                                           \
          a.spt1 = t0 + strip < n.PT ? t0 + strip : n.PT;                                        \
          int64_t ch = chunks;                                                                   \
          if (ch > (a.spt1 - a.spt0 + 7) / 8) ch = (a.spt1 - a.spt0 + 7) / 8;                    \
          a.chunks = (int)(ch < 1 ? 1 : ch);                                                     \
          a.dt0 = 0;                                                                             \
          SPML_BWD_LAUNCH(KS_, DT_, TAG_, C32_, 1)                                               \
          for (int dt0 = DT_; 32 * dt0 < D; dt0 += 7) {                                          \
            a.dt0 = dt0;                                                                         \
            SPML_BWD_LAUNCH(KS_, 7, TAG_, C32_, 2)                                               \
          }                                                                                      \
        }                                                                                        \
      } else {                                                                                   \
        for (int dt0 = 0; dt0 < n.DT && 32 * dt0 < D; dt0 += DT_) {                              \
          a.dt0 = dt0;                                                                           \
          SPML_BWD_LAUNCH(KS_, DT_, TAG_, C32_, 0)                                               \
        }                                                                                        \
      }                                                                                          \
    } else {                                                                                     \
      for (int dt0 = 0; dt0 < n.DT && 32 * dt0 < D; dt0 += DT_) {    /* one launch per d-chunk */ \
        a.dt0 = dt0;                                                                             \
        SPML_BWD_LAUNCH(KS_, DT_, TAG_, C32_, 0)                                                 \
      }                                                                                          \
    }                                                                                            \
  }
#define SPML_BWD(KS_)                                          \
  {                                                            \
    constexpr int DTM = KS_ <= 17 ? (KS_ + 1) / 2 : 3;         \
    /* SPML_NLL_CODE32 only selects the forward variant: the backward kernels measured 6-12 %  \
       slower with 32-bit predicates (profiles/r02_nll_scaling.md), the 64-bit form is exact for both */ \
    if (mode & SPML_NLL_TAGSET) SPML_BWD_DT(KS_, DTM, true, false)   \
    else SPML_BWD_DT(KS_, DTM, false, false)                   \
  }
  SPML_KS_SWITCH(SPML_BWD)
#undef SPML_BWD
#undef SPML_BWD_DT
#undef SPML_BWD_LAUNCH
#undef SPML_KS_SWITCH
  return finish(SPML_OK);
}

extern "C" int spml_segsort_nll_fwd_f32(const float* emb, const int64_t* own,
                                        const int64_t* px_code, int64_t P, const float* protos,
                                        const int64_t* pr_code, int64_t M, int D, float kappa,
                                        int mode, float* nll, float* stats, void* ws,
                                        size_t ws_bytes, void* stream) {
  return nll_common(false, emb, own, px_code, P, protos, pr_code, M, D, kappa, mode, nll, stats,
                    nullptr, nullptr, nullptr, -1, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int spml_segsort_nll_bwd_f32(const float* emb, const int64_t* own,
                                        const int64_t* px_code, int64_t P, const float* protos,
                                        const int64_t* pr_code, int64_t M, int D, float kappa,
                                        int mode, const float* stats, const float* d_nll,
                                        float* d_emb, float* d_protos, int64_t m_grad, void* ws,
                                        size_t ws_bytes, void* stream) {
  return nll_common(true, emb, own, px_code, P, protos, pr_code, M, D, kappa, mode, nullptr,
                    const_cast<float*>(stats), d_nll, d_emb, d_protos, m_grad, ws, ws_bytes,
                    (hipStream_t)stream);
}
