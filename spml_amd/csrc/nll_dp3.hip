// K4/K5 backward, prototype gradient: the software-pipelined kernel for 64-channel embeddings and
// 32-bit codes (segsort/loss.py:15-130 of the reference; weights: csrc/nll.hip, "backward").
//
// Round 4.  nll_bwd_dp (round 2) spends 2 600 cycles per (pixel tile, prototype tile) product against 768
// of matrix-pipe time: one product per wave between two workgroup barriers, per-pixel coefficients and
// codes fetched from LDS for each of the 16 accumulator rows of every product, the three phases
// (recompute, weights / split, second contraction) one after the other.  This is nll_bwd_de3 (see
// nll_de3.hip for the pipeline, the slot plan and the hand-assigned registers: the register map and its
// helpers, nll_de3_regs.inc, are shared) with the roles of the two operands swapped:
//   * a wave owns FOUR prototype tiles -- their std fragments (B operand of the recompute) and the
//     gradient accumulators dPr^T[d][prototype] stay in the accumulation registers for the wave's life;
//   * pixel tiles stream through the 3-slot LDS ring: std fragments (A operand of the recompute),
//     T-layout fragments of  E * g kappa / gscale  (A operand of the second contraction:
//     dPr^T[d][m] += ET[d][p] T'[p][m]) and one 1-KB block of per-pixel rows (weights, codes, own ids);
//   * the similarity tile has pixels as rows: a weight depends on the ROW (1/den, 1/den - 1/num of the
//     pixel) and on the predicate between the row's and the lane's code.  The 32 pixels of a tile almost
//     always carry one code (pixels are image-major; the co-occurrence term's code is the image's tag
//     set): then the predicate is one compare per product and the row weights are two register sets
//     read once per pixel tile; tiles with mixed codes take a step version with one compare per element;
//   * a pixel's own prototype: z = -inf in a rare wave-uniform branch, its term (own_term_kernel) is
//     added by nll_dp_own_term with one atomic per channel of the few pixels that have one;
//   * T carries the pixel's power-of-two scale (nll_common.hpp, nll_t_scale: 2^14 but for pixels whose own
//     prototype is not of their class) and the transposed pixel fragments 2^4 x (2^14 / that scale) before
//     their unscaled-residual splits, so that every product carries 2^18; it comes out again with gscale in
//     the final atomics.
// Grid: (groups of 16 prototype tiles that receive a gradient) x (chunks of pixel tiles); accumulators
// leave with one fp32 atomic per element and workgroup, as in the round-2 kernel.
#include "nll_common.hpp"

#include <algorithm>

namespace spml {
namespace {

#include "nll_de3_regs.inc"

template <int... I, typename F>
__device__ __forceinline__ void dp3_for_each(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void dp3_static_for(F&& f) {
  dp3_for_each(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

constexpr float kDp3EtScale = 16.0f;     // scale of the transposed pixel fragments (see above)

// Row block of a pixel tile (256 dwords): [0,32) wa * tscale, [32,64) wb * tscale (0 for rows past P),
// [64,96) code (low 32 bits), [96,128) own prototype (-1 past P), [128] 1 if the 32 codes are equal.
__global__ __launch_bounds__(64) void dp3_rows_kernel(const PixelCoef* __restrict__ coef,
                                                      const int64_t* __restrict__ px_code, int64_t P, int64_t PT,
                                                      float* __restrict__ rows) {
  const int64_t pt = (int64_t)blockIdx.x * 2 + (threadIdx.x >> 5);
  if (pt >= PT) return;
  const int j = threadIdx.x & 31;
  const int64_t i = 32 * pt + j;
  float wa = 0.f, wb = 0.f;
  int code = 0, o = -1;
  if (i < P) {
    const PixelCoef c = coef[i];
    wa = c.wa * c.tscale;
    wb = c.wb * c.tscale;
    code = (int)px_code[i];
    o = c.own;
  }
  const int code0 = __shfl(code, threadIdx.x & 32, 64);
  const unsigned long long same = __ballot(code == code0);
  const bool uni = ((same >> (threadIdx.x & 32)) & 0xffffffffull) == 0xffffffffull;
  float* r = rows + (size_t)pt * 256;
  r[j] = wa;
  r[32 + j] = wb;
  reinterpret_cast<int*>(r)[64 + j] = code;
  reinterpret_cast<int*>(r)[96 + j] = o;
  if (j == 0) reinterpret_cast<int*>(r)[128] = uni ? 1 : 0;
}

// dPr[own[p]] += kappa g_p own_term[p] E[p]   for the pixels whose own prototype has a non-zero weight
__global__ __launch_bounds__(256) void dp3_own_term_kernel(const float* __restrict__ own_term,
                                                           const int64_t* __restrict__ own,
                                                           const float* __restrict__ emb,
                                                           const float* __restrict__ d_nll, int64_t P, int D,
                                                           float kappa, int64_t m_grad, float* __restrict__ d_protos,
                                                           long long* __restrict__ d_protos64,
                                                           const float* __restrict__ gscale) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= P) return;
  const float ot = own_term[i];
  const int64_t m = own[i];
  if (ot == 0.f || m < 0 || m >= m_grad) return;
  const float f = ot * kappa * d_nll[i];
  if (d_protos64) {                                // deterministic mode (nll_common.hpp, dpr_add)
    const float igs = 1.0f / gscale[0];
    for (int d = threadIdx.x & 63; d < D; d += 64) det_atomic_add(d_protos64 + (size_t)m * D + d, f * emb[(size_t)i * D + d] * igs);
    return;
  }
  for (int d = threadIdx.x & 63; d < D; d += 64) unsafeAtomicAdd(d_protos + (size_t)m * D + d, f * emb[(size_t)i * D + d]);
}

template <int KS, int DT, bool TAG>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(96))) void nll_bwd_dp3(NllArgs a, const float* rows) {
  static_assert(KS == 4 && DT == 2 && SPML_DE3_NB == 4, "register map and slot plan: tools/gen_nll_de3.py");
  constexpr int NB = 4, MTB = 2, NSLOT = 3;
  constexpr int TSTD = 2 * KS * 1024;            // std hi | lo blocks of one pixel tile
  constexpr int TROW = TSTD + 4 * DT * 1024;     // offset of the row block
  constexpr int TILE = TROW + 1024;
  constexpr int SLOT = MTB * TILE;
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, jl = lane & 31;
  const int64_t mt0 = ((int64_t)blockIdx.x * 4 + wv) * NB;          // this wave's prototype tiles mt0 .. mt0 + 3
  const int64_t per = (a.n.PT + a.chunks - 1) / a.chunks;
  const int64_t pt_lo = (int64_t)blockIdx.y * per;
  const int ntile = (int)(min(a.n.PT, pt_lo + per) - pt_lo);
  if (ntile <= 0) return;
  const int nstage = (ntile + MTB - 1) / MTB;
  float mone = -1.0f, minus_inf = -INFINITY;
  asm("" : "+s"(mone));                          // opaque: the residual stays one v_fma_mix_f32
  const unsigned ring_lds = (unsigned)(size_t)((lptr_t)sm);
  const unsigned dma_off_std = ((unsigned)wv * 64u + (unsigned)lane) * 16u;         // this lane's bytes of block wv of a tile
  const unsigned dma_off_t = ((unsigned)(wv >> 1) * 64u + (unsigned)lane) * 16u;    // ... of T-layout block pair wv >> 1
  const _Float16* const dma_t = (wv & 1) ? a.etl : a.eth;

  de3_claim_registers();                         // accumulators = 0
  int pcode[NB], col[NB];
  dp3_static_for<NB>([&](auto nbc) {
    constexpr int nb = decltype(nbc)::value;
    const int64_t mt = min(mt0 + nb, a.n.MT - 1);
    dp3_static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      de3_ld_frag((nb * KS + ks) * 2, a.ph + (((size_t)mt * KS + ks) * 64 + lane) * 8);
      de3_ld_frag((nb * KS + ks) * 2 + 1, a.pl + (((size_t)mt * KS + ks) * 64 + lane) * 8);
    });
    pcode[nb] = (int)a.pr_code_pad[32 * mt + jl];
    col[nb] = (int)(32 * mt) + jl;               // prototype of this lane's column
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the resident fragments (loads the compiler does not track)

  // LDS-DMA of stage st (two pixel tiles of 17 blocks) into ring slot st % 3: wave wv brings blocks
  // {wv, 4 + wv, 8 + wv, 12 + wv} of either tile, wave 0 the row block as well
  auto stage = [&](int st) {
    const unsigned dst = ring_lds + (unsigned)((st % NSLOT) * SLOT) + (unsigned)wv * 1024u;
#pragma unroll
    for (int t = 0; t < MTB; ++t) {
      const unsigned pt = (unsigned)(pt_lo + min(st * MTB + t, ntile - 1));   // past the end: a harmless duplicate
      const unsigned off_std = dma_off_std + pt * 4096u, off_t = dma_off_t + pt * 4096u;
      const unsigned d = dst + (unsigned)(t * TILE);
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(off_std), "s"(a.eh) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d + 4096u), "v"(off_std), "s"(a.el) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d + 8192u), "v"(off_t), "s"(dma_t) : "memory", "m0");
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" : : "s"(d + 12288u - 2048u), "v"(off_t), "s"(dma_t) : "memory", "m0");   // (the offset also moves the LDS address)
      if (wv == 0) {
        const unsigned off_r = (unsigned)lane * 16u + pt * 1024u;
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d + (unsigned)TROW), "v"(off_r), "s"(rows) : "memory", "m0");
      }
    }
  };
  stage(0);
  if (nstage > 1) stage(1);
  const int my_blocks = MTB * (wv == 0 ? 5 : 4);
  if (nstage > 1) wait_vmcnt(my_blocks); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // LDS byte address (32-bit) of this lane's 16 bytes in block 0 of tile t
  auto tile_addr = [&](int t) -> unsigned {
    return ring_lds + (unsigned)(((t / MTB) % NSLOT) * SLOT + (t % MTB) * TILE) + (unsigned)lane * 16u;
  };
  auto r_mfma = [&](auto ic, auto hfc, auto nbc) {
    constexpr int i = decltype(ic)::value, hf = decltype(hfc)::value, nb = decltype(nbc)::value;
    constexpr int ks = 2 * hf + i / 3, w = i % 3;
    de3_r((((nb * KS + ks) * 3 + w) * 2) + ((hf == 0 && i == 0) ? 1 : 0));
  };
  auto c_mfma = [&](auto ic, auto dtc, auto nbc) {
    constexpr int i = decltype(ic)::value, dt = decltype(dtc)::value, nb = decltype(nbc)::value;
    de3_c(((nb * DT + dt) * 2 + i / 3) * 3 + i % 3);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;

  // pipeline fill: R(0) and the first half of R(1), back to back (dependent chains: the only place)
  de3_zero_pipeline();
  dp3_static_for<8>([&](auto ic) { de3_ld_a(decltype(ic)::value, tile_addr(0)); });
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  dp3_static_for<6>([&](auto ic) { r_mfma(ic, I0{}, I0{}); });
  dp3_static_for<6>([&](auto ic) { r_mfma(ic, I1{}, I0{}); });
  dp3_static_for<6>([&](auto ic) { r_mfma(ic, I0{}, I1{}); });
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // R(0) -> V(0) without MFMA slots in between

  // per-row values of the current pixel tile for this lane's half (rows 8 q + 4 half + 0..3, q = 0..3)
  float wa_r[16], wb_r[16];
  int rc_r[16];
  auto read_rows = [&](unsigned tile_base) {     // compiler-tracked LDS reads (12 x ds_read_b128)
    const float* rw = reinterpret_cast<const float*>(sm + (tile_base - ring_lds) + TROW);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4v va = *reinterpret_cast<const float4v*>(rw + 8 * q + 4 * half);
      const float4v vb = *reinterpret_cast<const float4v*>(rw + 32 + 8 * q + 4 * half);
      const int4 vc = *reinterpret_cast<const int4*>(rw + 64 + 8 * q + 4 * half);
#pragma unroll
      for (int e = 0; e < 4; ++e) { wa_r[4 * q + e] = va[e]; wb_r[4 * q + e] = vb[e]; }
      rc_r[4 * q] = vc.x; rc_r[4 * q + 1] = vc.y; rc_r[4 * q + 2] = vc.z; rc_r[4 * q + 3] = vc.w;
    }
  };

  // one step: 6 groups of [R(k+2) first half | C(k-1) d-tile 0 | R(k+1) second half | C(k-1) d-tile 1],
  // the vector operations of V(k) in slots 2..23 (products k = 4 g + j: pixel tile g x prototype tile j)
  auto step = [&](auto jc, auto uni_tag, bool own_here, unsigned tile_base, unsigned at, unsigned at_next) {
    constexpr int J = decltype(jc)::value;
    constexpr bool UNI = decltype(uni_tag)::value;
    constexpr int NBC = (J + NB - 1) % NB, NBV = J, NBR1 = (J + 1) % NB, NBR2 = (J + 2) % NB;
    constexpr int PAR = NBV & 1;                     // T buffer written by V(k); C(k-1) reads the other one
    // a pixel of this tile has its own prototype among the wave's 128 (rare): that element of z becomes -inf
    if (__builtin_expect(own_here, 0)) {
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");    // z: 16 passes after R's last MFMA
      const int* ro = reinterpret_cast<const int*>(sm + (tile_base - ring_lds) + TROW) + 96;
      dp3_static_for<16>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const int own_row = ro[(r & 3) + 8 * (r >> 2) + 4 * half];
        de3_own_patch(NBV * 16 + r, minus_inf, __ballot(own_row == col[NBV]));
      });
    }
    // weights: uniform pixel tile -> one predicate (this lane's prototype against the tile's code) selects
    // between the two row-weight sets; otherwise one predicate per element
    // (the selection mask of an element: an SGPR pair from a compiler-scheduled compare; one for the whole product
    // when the tile is uniform)
    const unsigned long long su = __ballot(code_match<TAG, int>(rc_r[0], pcode[NBV]));
    float e[2][2], w0, w1;
    auto mask_of = [&](auto rc) -> unsigned long long {
      constexpr int r = decltype(rc)::value;
      if constexpr (UNI) return su;
      else return __ballot(code_match<TAG, int>(rc_r[r], pcode[NBV]));
    };
    // vector operation n (0..79): pair p = values (2p, 2p+1): A = 2 x exp2 (one pair ahead), W = 2 x weight select,
    // B = 2 x multiply, C = hi = rtz pack, D = 2 x (t - hi) by v_fma_mix_f32, E = lo = rtz pack
    auto vop = [&](auto nc) {
      constexpr int n = decltype(nc)::value;
      constexpr int m = n - 2;
      constexpr int p = n < 2 ? 0 : (m < 70 ? m / 10 + (m % 10 < 2 ? 1 : 0) : 7);
      constexpr int k = n < 2 ? n : (m < 70 ? m % 10 : m - 70 + 2);      // 0,1 A  2,3 W  4,5 B  6 C  7,8 D  9 E
      float& e0 = e[p & 1][0];
      float& e1 = e[p & 1][1];
      if constexpr (k == 0) de3_exp(NBV * 16 + 2 * p, e0);
      else if constexpr (k == 1) de3_exp(NBV * 16 + 2 * p + 1, e1);
      else if constexpr (k == 2) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(w0) : "v"(wb_r[2 * p]), "v"(wa_r[2 * p]), "s"(mask_of(std::integral_constant<int, 2 * p>{})));
      else if constexpr (k == 3) asm volatile("v_cndmask_b32 %0, %1, %2, %3" : "=v"(w1) : "v"(wb_r[2 * p + 1]), "v"(wa_r[2 * p + 1]), "s"(mask_of(std::integral_constant<int, 2 * p + 1>{})));
      else if constexpr (k == 4) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e0) : "v"(w0));
      else if constexpr (k == 5) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e1) : "v"(w1));
      else if constexpr (k == 6) de3_pk_hi(PAR * 8 + p, e0, e1);
      else if constexpr (k == 7) de3_mix0(PAR * 8 + p, e0, mone);
      else if constexpr (k == 8) de3_mix1(PAR * 8 + p, e1, mone);
      else de3_pk_lo(PAR * 8 + p, e0, e1);
    };
    dp3_static_for<24>([&](auto ic) {
      constexpr int i = decltype(ic)::value;
      using G = std::integral_constant<int, (i >> 2)>;
      // outstanding LDS reads of the streamed operands, oldest first, at the slot that needs the oldest ones
      // (slot plan of nll_de3.hip; the compiler's own LDS reads can only make the waits stricter)
      if constexpr (J == 1 && i == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      if constexpr (J == 1 && i == 13) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      if constexpr (J == 2 && i == 0) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      if constexpr (J == 2 && i == 12) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      if constexpr (J == 3 && i == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      if constexpr (J == 3 && i == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr ((i & 3) == 0) r_mfma(G{}, I0{}, std::integral_constant<int, NBR2>{});
      else if constexpr ((i & 3) == 1) c_mfma(G{}, I0{}, std::integral_constant<int, NBC>{});
      else if constexpr ((i & 3) == 2) r_mfma(G{}, I1{}, std::integral_constant<int, NBR1>{});
      else c_mfma(G{}, I1{}, std::integral_constant<int, NBC>{});
      if constexpr (i >= 2 && i < 22) {              // 80 operations: four per slot, slots 2..21
        vop(std::integral_constant<int, 4 * (i - 2)>{});
        vop(std::integral_constant<int, 4 * (i - 2) + 1>{});
        vop(std::integral_constant<int, 4 * (i - 2) + 2>{});
        vop(std::integral_constant<int, 4 * (i - 2) + 3>{});
      }
      if constexpr (J == 0 && i == 11) { de3_ld_q(0, at); de3_ld_q(1, at); de3_ld_q(4, at); de3_ld_q(5, at); }   // s2 = 0
      if constexpr (J == 0 && i == 23) { de3_ld_q(2, at); de3_ld_q(3, at); de3_ld_q(6, at); de3_ld_q(7, at); }   // s2 = 1
      if constexpr (J == 1 && i == 8) { de3_ld_a(0, at_next); de3_ld_a(1, at_next); }      // k-step 0
      if constexpr (J == 1 && i == 20) { de3_ld_a(2, at_next); de3_ld_a(3, at_next); }     // k-step 1
      if constexpr (J == 2 && i == 10) { de3_ld_a(4, at_next); de3_ld_a(5, at_next); }     // k-step 2
      if constexpr (J == 2 && i == 22) { de3_ld_a(6, at_next); de3_ld_a(7, at_next); }     // k-step 3
    });
  };

  unsigned tile_base = ring_lds;                     // LDS address of the current tile (block 0), kept incrementally
  int slot = 0;
  const int m_lo = (int)(32 * mt0);                  // the wave's prototypes: [m_lo, m_lo + 128)
  auto tile = [&](int g, auto uni_tag) {
    unsigned next_base = tile_base;                  // (past the last tile the recompute runs on this one and is dropped)
    if (g + 1 < ntile) {
      if ((g + 1) % MTB == 0) {
        slot = slot + 1 == NSLOT ? 0 : slot + 1;
        next_base = ring_lds + (unsigned)(slot * SLOT);
      } else {
        next_base = tile_base + (unsigned)TILE;
      }
    }
    const unsigned at = tile_base + (unsigned)lane * 16u, at_next = next_base + (unsigned)lane * 16u;
    // own prototypes of this tile's pixels among the wave's 128 prototypes?
    const int own_j = reinterpret_cast<const int*>(sm + (tile_base - ring_lds) + TROW)[96 + jl];
    const bool own_here = __any((unsigned)(own_j - m_lo) < 128u);
    step(std::integral_constant<int, 0>{}, uni_tag, own_here, tile_base, at, at_next);
    // the next tile's first read (step 1): if it opens a stage, that stage has to have landed for every wave;
    // the same barrier frees the slot of the stage before the current one
    if ((g + 1) % MTB == 0 && g + 1 < ntile) {
      const int st = (g + 1) / MTB;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      wg_barrier();
      if (st + 1 < nstage) stage(st + 1);
    }
    step(std::integral_constant<int, 1>{}, uni_tag, own_here, tile_base, at, at_next);
    step(std::integral_constant<int, 2>{}, uni_tag, own_here, tile_base, at, at_next);
    step(std::integral_constant<int, 3>{}, uni_tag, own_here, tile_base, at, at_next);
    tile_base = next_base;
  };
  for (int g = 0; g < ntile; ++g) {
    read_rows(tile_base);
    const bool uni = reinterpret_cast<const int*>(sm + (tile_base - ring_lds) + TROW)[128] != 0;
    if (uni) tile(g, std::true_type{});
    else tile(g, std::false_type{});
  }
  dp3_static_for<6>([&](auto ic) {                   // the pipeline's tail: C(last tile, NB - 1)
    c_mfma(ic, I0{}, std::integral_constant<int, NB - 1>{});
    c_mfma(ic, I1{}, std::integral_constant<int, NB - 1>{});
  });

  // the accumulators are read 16 passes after the last MFMA at the earliest
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  const float f = a.gscale[0] / (kTScale * kDp3EtScale);
  if (a.d_protos64) {                                // deterministic mode: fixed point relative to gscale (dpr_add)
    constexpr float fd = 1.0f / (kTScale * kDp3EtScale);
    dp3_static_for<NB * DT * 16>([&](auto ic) {
      constexpr int i = decltype(ic)::value, nb = i / (DT * 16), dt = (i / 16) % DT, r = i % 16;
      const float v = de3_read_acc(i);
      const int d = 32 * dt + tile_row(r, half);
      if (mt0 + nb < a.mt_grad && col[nb] < a.n.M && d < a.n.D)
        det_atomic_add(a.d_protos64 + (size_t)col[nb] * a.n.D + d, v * fd);
    });
    return;
  }
  dp3_static_for<NB * DT * 16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, nb = i / (DT * 16), dt = (i / 16) % DT, r = i % 16;
    const float v = de3_read_acc(i);
    const int d = 32 * dt + tile_row(r, half);
    if (mt0 + nb < a.mt_grad && col[nb] < a.n.M && d < a.n.D)
      unsafeAtomicAdd(a.d_protos + (size_t)col[nb] * a.n.D + d, v * f);
  });
}

}  // namespace

size_t nll_dp3_rows_bytes(int64_t PT) { return (size_t)PT * 1024; }

// rows: workspace of nll_dp3_rows_bytes(PT); own_term: [PT * 32] from own_term_kernel; a.eth / a.etl: T-layout
// fragments of emb * (kappa g / gscale) * 16 * (2^14 / tscale_p) with unscaled residuals; a.coef: coef_kernel; a.eh / el / ph / pl: as for nll_bwd_de3
int nll_launch_bwd_dp3(const NllArgs& a, const float* own_term, const float* emb, float* rows, hipStream_t s) {
  if (a.n.KS != 4 || a.n.DT != 2) return SPML_ERR_UNSUPPORTED;
  if (a.mt_grad <= 0) return SPML_OK;
  hipLaunchKernelGGL(dp3_rows_kernel, dim3((unsigned)((a.n.PT + 1) / 2)), dim3(64), 0, s, a.coef, a.px_code, a.n.P,
                     a.n.PT, rows);
  const unsigned groups = (unsigned)((a.mt_grad + 15) / 16);     // 4 waves x 4 prototype tiles
  // pixel chunks: ~1000 workgroups of at least 64 pixel tiles (the register load + pipeline fill of a workgroup is
  // worth ~4 tiles; 4 000 workgroups of 25 tiles measured 40 % slower at M = 17 k, no faster at 139 k).  One
  // workgroup per CU: the count is chosen so that the last round over the 256 CUs is as full as possible.
  int64_t chunks = 1;
  {
    // (small calls -- the semantic-annotation term sees the labelled pixels only: too few 64-tile chunks to fill the
    // chip: 16-tile chunks then.  The first version of this search started above `most` and fell through to ONE chunk
    // for such calls: 7 workgroups, 2.0 ms in the bench step)
    int64_t most = (a.n.PT + 63) / 64;
    if ((int64_t)groups * most < 512) most = (a.n.PT + 15) / 16;
    const int64_t c0 = (1024 + groups - 1) / groups;
    const int64_t lo = std::max<int64_t>(1, std::min<int64_t>((c0 + 1) / 2, most)), hi = std::max(lo, std::min<int64_t>(2 * c0, most));
    double best = -1.0;
    for (int64_t c = lo; c <= hi; ++c) {
      const int64_t wgs = (int64_t)groups * c, rounds = (wgs + 255) / 256;
      const double fill = (double)wgs / (double)(rounds * 256) - 0.002 * (double)(c > c0 ? c - c0 : c0 - c);
      if (fill > best) { best = fill; chunks = c; }
    }
  }
  if (chunks < 1) chunks = 1;
  if (chunks > 65535) chunks = 65535;
  NllArgs b = a;
  b.chunks = (int)chunks;
  constexpr int LDS3 = 3 * 2 * (2 * 4 + 4 * 2 + 1) * 1024;
  if (a.mode & SPML_NLL_TAGSET) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_dp3<4, 2, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
    hipLaunchKernelGGL((nll_bwd_dp3<4, 2, true>), dim3(groups, (unsigned)chunks), dim3(256), LDS3, s, b, (const float*)rows);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_dp3<4, 2, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
    hipLaunchKernelGGL((nll_bwd_dp3<4, 2, false>), dim3(groups, (unsigned)chunks), dim3(256), LDS3, s, b, (const float*)rows);
  }
  hipLaunchKernelGGL(dp3_own_term_kernel, dim3((unsigned)((a.n.P + 3) / 4)), dim3(256), 0, s, own_term, a.own, emb, a.d_nll,
                     a.n.P, a.n.D, a.kappa, a.mt_grad * 32 < a.n.M ? a.mt_grad * 32 : a.n.M, a.d_protos, a.d_protos64, a.gscale);
  return launch_status();
}

}  // namespace spml
