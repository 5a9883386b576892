// K4/K5 backward, embedding gradient: the software-pipelined kernel for 64-channel embeddings and
// 32-bit codes -- what the predictors' calls of the semantic terms take (segsort/loss.py:15-130 of
// the reference; derivation of the weights: csrc/nll.hip, "backward").  Its own translation unit:
// one 512-register wave per SIMD, registers assigned by hand (nll_de3_regs.inc, written by
// tools/gen_nll_de3.py).
//
// Round 4.  nll_bwd_de2 runs the three phases of a (pixel tile, prototype tile) product -- R: 3 KS
// MFMAs that recompute the similarity tile, V: exp2 / weights / f16 split on the vector ALU, C: 6 DT
// MFMAs of the second contraction -- one after the other, so the matrix pipe idles during V and the
// only overlap comes from the second wave of the SIMD: 1 370 cycles per product against 768 of
// matrix-pipe time.  tools/hw_probes/mfma_valu_overlap.hip shows what the hardware allows: ONE wave
// per SIMD that issues 1 MFMA + 4 VALU in alternation, on alternating accumulators, runs at 34.7
// cycles per MFMA (the VALU is free).
//
// Here: one workgroup of 4 waves (one per SIMD) per CU, a wave owns FOUR pixel tiles, and the
// products k = 4 g + j of its walk over the prototype tiles g are software-pipelined:
//     step k :   C(k-1): T[k-1] -> dacc  |  k-steps {2,3} of R(k+1), {0,1} of R(k+2)  |  V(k): z[k] -> T[k]
// The 24 MFMAs of a step are four INDEPENDENT accumulation chains of six, issued in rotation
// (R(k+2), C d-tile 0, R(k+1), C d-tile 1): a dependent MFMA that does not directly follow its
// predecessor waits for the write-back (~75 cycles; the same step with 12 + 12 dependent MFMAs
// measured 1 280 cycles).  They are the issue slots; the 64 vector operations of V are dealt out
// over slots 2..23, three per slot.  Every instruction of a step is a single-instruction
// `asm volatile` statement: the compiler keeps their order.  Its own scheduler clusters the vector
// work, its own register assignment spills at this size and copies whole operand sets at every
// join of the control flow -- so the registers of the pipeline are assigned BY HAND and named in the
// asm text (map: tools/gen_nll_de3.py); the compiler is confined to v0..v95 (amdgpu_num_vgpr) and has
// no accumulation-register value of its own.  tests/test_cabi_exports.py disassembles the kernel and
// checks that no compiler-generated instruction touches a hand-assigned register.
// What the compiler cannot see through the asm is met by construction: V(k) starts three MFMA issues
// (>= 96 cycles) after the last MFMA of R(k); a transcendental's result is used two instructions later
// at the earliest; T is written a whole step before C reads it; LDS reads are issued right after the
// last MFMA that reads the registers they overwrite and waited for by hand (lgkmcnt).
// A operands are read from LDS once per prototype tile and serve four products (16 ds_read_b128 per
// tile instead of 64).  A tile whose 32 prototypes carry one code (> 90 % of them for the co-occurrence
// term) takes one predicate per pixel tile; the others a step version with 16 row weights.  The own
// prototype of a pixel -- the one element of its row whose weight follows another formula -- is taken
// out of the tile (z = -inf => T = 0, a rare wave-uniform branch) and added by nll_de_finalize from a
// per-pixel coefficient (own_term_kernel).  3-slot LDS ring of 2-tile stages: the barrier before the
// first read of stage s also frees the slot of stage s - 2.  Rows of padded prototypes need no masking:
// their transposed fragments are zero.  The sums are those of nll_bwd_de2, in the same chunk order.
#include "nll_common.hpp"

namespace spml {
namespace {

#include "nll_de3_regs.inc"

// compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1 (the register helpers are
// switches over register names: every key has to be a constant)
template <int... I, typename F>
__device__ __forceinline__ void de3_for_each(std::integer_sequence<int, I...>, F&& f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void de3_static_for(F&& f) {
  de3_for_each(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int KS, int DT, bool TAG>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(96))) void nll_bwd_de3(NllArgs a) {
  static_assert(KS == 4 && DT == 2 && SPML_DE3_NB == 4, "register map and slot plan: tools/gen_nll_de3.py");
  constexpr int NB = 4, MTB = 2, NSLOT = 3;
  constexpr int TSTD = 2 * KS * 1024;            // std hi | lo blocks of one prototype tile
  constexpr int TILE = TSTD + 4 * DT * 1024;     // + T-layout [DT][2][hi|lo]
  constexpr int SLOT = MTB * TILE;
  constexpr int CODES = kFwd2TilesPerChunk * 32 * 4;
  constexpr int NBLK = MTB * (2 * KS + 4 * DT);
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  int* const codes_lds = reinterpret_cast<int*>(sm);
  unsigned char* const ring = sm + CODES;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, jl = lane & 31;
  const int64_t pt0 = ((int64_t)blockIdx.x * 4 + wv) * NB;
  const int nchunk = (int)((a.n.MT + kFwd2TilesPerChunk - 1) / kFwd2TilesPerChunk);
  float mone = -1.0f, minus_inf = -INFINITY;
  asm("" : "+s"(mone));                          // opaque: the residual stays one v_fma_mix_f32

  const unsigned ring_lds = (unsigned)(size_t)((lptr_t)sm) + (unsigned)CODES;      // LDS byte address of the ring
  const unsigned dma_off_std = ((unsigned)wv * 64u + (unsigned)lane) * 16u;         // this lane's bytes of block wv of a tile
  const unsigned dma_off_t = ((unsigned)(wv >> 1) * 64u + (unsigned)lane) * 16u;    // ... of T-layout block pair wv >> 1
  const _Float16* const dma_t = (wv & 1) ? a.ptl : a.pth;                           // (hi / lo alternate with the wave)
  de3_claim_registers();                         // accumulators = 0
  int pcode[NB], own[NB];
  float wa[NB], wb[NB];
  de3_static_for<NB>([&](auto nbc) {
    constexpr int nb = decltype(nbc)::value;
    const int64_t pt = min(pt0 + nb, a.n.PT - 1);
    de3_static_for<KS>([&](auto ksc) {
      constexpr int ks = decltype(ksc)::value;
      de3_ld_frag((nb * KS + ks) * 2, a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      de3_ld_frag((nb * KS + ks) * 2 + 1, a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    });
    const int64_t p = min(32 * pt + jl, a.n.P - 1);
    pcode[nb] = (int)a.px_code[p];
    const PixelCoef cf = a.coef[32 * pt + jl];
    wa[nb] = cf.wa * cf.tscale; wb[nb] = cf.wb * cf.tscale; own[nb] = cf.own;    // (tscale: nll_common.hpp; 0 past P)
    if (pt0 + nb >= a.n.PT) { wa[nb] = 0.f; wb[nb] = 0.f; own[nb] = -1; }
  });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the resident fragments (loads the compiler does not track)

  for (int c = blockIdx.y; c < nchunk; c += gridDim.y) {
    const int64_t mt_lo = (int64_t)c * kFwd2TilesPerChunk;
    const int ntile = (int)(min(a.n.MT, mt_lo + kFwd2TilesPerChunk) - mt_lo);
    const int nstage = (ntile + MTB - 1) / MTB;
    __syncthreads();                                   // the previous chunk's codes / ring are no longer read
    // LDS-DMA of stage st (two prototype tiles, 16 blocks of 1 KB each) into ring slot st % 3: wave wv brings blocks
    // {wv, 4 + wv, 8 + wv, 12 + wv} of either tile = k-step wv of the std fragments (hi, lo) and blocks wv, 4 + wv
    // of the T-layout fragments; every tile is 4 KB in each of the four arrays, so one 32-bit offset per tile serves
    // all of them (scalar base + vector offset form: ~12 instructions per tile)
    auto stage = [&](int st) {
      const unsigned dst = ring_lds + (unsigned)((st % NSLOT) * SLOT) + (unsigned)wv * 1024u;
#pragma unroll
      for (int t = 0; t < MTB; ++t) {
        const unsigned mt = (unsigned)(mt_lo + min(st * MTB + t, ntile - 1));   // past the end: a harmless duplicate
        const unsigned off_std = dma_off_std + mt * 4096u, off_t = dma_off_t + mt * 4096u;
        const unsigned d = dst + (unsigned)(t * TILE);
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d), "v"(off_std), "s"(a.ph) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d + 4096u), "v"(off_std), "s"(a.pl) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(d + 8192u), "v"(off_t), "s"(dma_t) : "memory", "m0");
        asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048" : : "s"(d + 12288u - 2048u), "v"(off_t), "s"(dma_t) : "memory", "m0");   // (the offset also moves the LDS address)
      }
    };
    stage(0);
    if (nstage > 1) stage(1);
    for (int i = threadIdx.x; i < 32 * ntile; i += 256) codes_lds[i] = (int)a.pr_code_pad[32 * mt_lo + i];
    const int my_blocks = NBLK / 4;
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
    int own_t[NB];                                     // chunk-relative tile of the own prototype (or out of range)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) own_t[nb] = (own[nb] - (int)(32 * mt_lo)) >> 5;

    // LDS byte address (32-bit) of this lane's 16 bytes in block 0 of tile t
    auto tile_addr = [&](int t) -> unsigned {
      return ring_lds + (unsigned)(((t / MTB) % NSLOT) * SLOT + (t % MTB) * TILE) + (unsigned)lane * 16u;
    };
    // similarity operands of k-steps {0, 1} / {2, 3}, transposed operands (issued here, waited for by hand)
    // (LDS reads of the streamed operands: issued by the steps, each right after the last MFMA that reads the
    // registers it overwrites, and waited for by hand with counted lgkmcnt -- LDS returns in order)
    // MFMA i (0..5) of half hf (k-steps 2 hf, 2 hf + 1) of the similarity tile of pixel tile nb
    auto r_mfma = [&](auto ic, auto hfc, auto nbc) {
      constexpr int i = decltype(ic)::value, hf = decltype(hfc)::value, nb = decltype(nbc)::value;
      constexpr int ks = 2 * hf + i / 3, w = i % 3;
      de3_r((((nb * KS + ks) * 3 + w) * 2) + ((hf == 0 && i == 0) ? 1 : 0));
    };
    // MFMA i (0..5) of the second contraction into d-tile dt of pixel tile nb (T of parity nb & 1)
    auto c_mfma = [&](auto ic, auto dtc, auto nbc) {
      constexpr int i = decltype(ic)::value, dt = decltype(dtc)::value, nb = decltype(nbc)::value;
      de3_c(((nb * DT + dt) * 2 + i / 3) * 3 + i % 3);
    };
    using I0 = std::integral_constant<int, 0>;
    using I1 = std::integral_constant<int, 1>;

    // pipeline fill: R(0) and the first half of R(1), back to back (dependent chains: the only place)
    de3_zero_pipeline();
    de3_static_for<8>([&](auto ic) { de3_ld_a(decltype(ic)::value, tile_addr(0)); });
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    de3_static_for<6>([&](auto ic) { r_mfma(ic, I0{}, I0{}); });
    de3_static_for<6>([&](auto ic) { r_mfma(ic, I1{}, I0{}); });
    de3_static_for<6>([&](auto ic) { r_mfma(ic, I0{}, I1{}); });
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");   // R(0) -> V(0) without MFMA slots in between
    int code_next = codes_lds[0];                      // first code of the next tile, read a tile ahead

    // one step: 6 groups of [R(k+2) first half | C(k-1) d-tile 0 | R(k+1) second half | C(k-1) d-tile 1],
    // the 64 vector operations of V(k) in slots 2..23
    auto step = [&](auto jc, auto uni_tag, int g, unsigned at, unsigned at_next, int code0) {
      constexpr int J = decltype(jc)::value;
      constexpr bool UNI = decltype(uni_tag)::value;
      constexpr int NBC = (J + NB - 1) % NB, NBV = J, NBR1 = (J + 1) % NB, NBR2 = (J + 2) % NB;
      constexpr int PAR = NBV & 1;                     // T buffer written by V(k); C(k-1) reads the other one
      // the own prototype of one of this tile's pixels is in the prototype tile (rare): its element of z
      // becomes -inf, i.e. T = 0 there; nll_de_finalize adds the own prototype's term (own_term_kernel)
      if (__builtin_expect(__any(own_t[NBV] == g), 0)) {
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");    // z: 16 passes after R's last MFMA
        const int own_rel = own[NBV] - (int)(32 * mt_lo) - 32 * g - 4 * half;   // == tile_row(r, 0) for the own element
        de3_static_for<16>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          de3_own_patch(NBV * 16 + r, minus_inf, __ballot(tile_row(r, 0) == own_rel));
        });
      }
      // row weights: one predicate for a uniform tile; 16 (compiler-scheduled, ahead of the step) otherwise
      float w[16];
      const float wu = code_match<TAG, int>(pcode[NBV], code0) ? wa[NBV] : wb[NBV];
      if constexpr (!UNI) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const int4 v = *reinterpret_cast<const int4*>(codes_lds + 32 * g + 8 * q4 + 4 * half);
          w[4 * q4] = code_match<TAG, int>(pcode[NBV], v.x) ? wa[NBV] : wb[NBV];
          w[4 * q4 + 1] = code_match<TAG, int>(pcode[NBV], v.y) ? wa[NBV] : wb[NBV];
          w[4 * q4 + 2] = code_match<TAG, int>(pcode[NBV], v.z) ? wa[NBV] : wb[NBV];
          w[4 * q4 + 3] = code_match<TAG, int>(pcode[NBV], v.w) ? wa[NBV] : wb[NBV];
        }
      }
      float e[2][2];                                   // the pair in flight and the next one
      // vector operation n (0..63): pair p = values (2p, 2p+1) of the tile: A = 2 x exp2, then, one pair behind,
      // B = 2 x weight, C = hi = rtz pack, D = 2 x (t - hi) by v_fma_mix_f32, E = lo = rtz pack
      auto vop = [&](auto nc) {
        constexpr int n = decltype(nc)::value;
        // order: A(0); then for p = 0..7: A(p+1) [p < 7], B(p), C(p), D(p), E(p)
        // k: 0,1 = A  2,3 = B  4 = C  5,6 = D  7 = E;  8 operations per pair while A(p+1) exists (p < 7), 6 for the last
        constexpr int m = n - 2;
        constexpr int p = n < 2 ? 0 : (m < 56 ? m / 8 + (m % 8 < 2 ? 1 : 0) : 7);
        constexpr int k = n < 2 ? n : (m < 56 ? m % 8 : m - 56 + 2);
        float& e0 = e[p & 1][0];
        float& e1 = e[p & 1][1];
        if constexpr (k == 0) de3_exp(NBV * 16 + 2 * p, e0);
        else if constexpr (k == 1) de3_exp(NBV * 16 + 2 * p + 1, e1);
        else if constexpr (k == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e0) : "v"(UNI ? wu : w[(2 * p) & 15]));
        else if constexpr (k == 3) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e1) : "v"(UNI ? wu : w[(2 * p + 1) & 15]));
        else if constexpr (k == 4) de3_pk_hi(PAR * 8 + p, e0, e1);
        else if constexpr (k == 5) de3_mix0(PAR * 8 + p, e0, mone);
        else if constexpr (k == 6) de3_mix1(PAR * 8 + p, e1, mone);
        else de3_pk_lo(PAR * 8 + p, e0, e1);
      };
      // Operands that change with the prototype tile, and when their registers fall free (slot = MFMA index):
      //   k-step 0 / 1 fragments serve R(k+2): last read in step 1 at slot 8 / 20, first read of the next tile's
      //                in step 2 at slot 0 / 12;
      //   k-step 2 / 3 fragments serve R(k+1): last read in step 2 at slot 10 / 22, next tile's in step 3 at 2 / 14;
      //   transposed fragments of contraction step s2 = 0 / 1 serve C(k-1): last read in step 0 at slot 11 / 23,
      //                this tile's in step 1 at slot 1 / 13.
      // Every load is issued right behind the last reader and has ~15 MFMA slots to land; the waits count the
      // loads issued after the ones they need (the compiler's own LDS reads can only make them stricter).
      de3_static_for<24>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        using G = std::integral_constant<int, (i >> 2)>;
        // outstanding LDS reads, oldest first, at the slot that needs the oldest ones:
        if constexpr (J == 1 && i == 1) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");    // q s2=0 (4) | q s2=1 (4)
        if constexpr (J == 1 && i == 13) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // q s2=1 (4) | k-step 0 (2)
        if constexpr (J == 2 && i == 0) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");    // k-step 0 (2) | k-step 1 (2)
        if constexpr (J == 2 && i == 12) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");   // k-step 1 (2) | k-step 2 (2)
        if constexpr (J == 3 && i == 2) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");    // k-step 2 (2) | k-step 3 (2)
        if constexpr (J == 3 && i == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // k-step 3 (2)
        if constexpr ((i & 3) == 0) r_mfma(G{}, I0{}, std::integral_constant<int, NBR2>{});
        else if constexpr ((i & 3) == 1) c_mfma(G{}, I0{}, std::integral_constant<int, NBC>{});
        else if constexpr ((i & 3) == 2) r_mfma(G{}, I1{}, std::integral_constant<int, NBR1>{});
        else c_mfma(G{}, I1{}, std::integral_constant<int, NBC>{});
        // 64 operations over slots 2..23: 3 per slot, the last two slots take 2
        if constexpr (i >= 2 && i < 22) {
          vop(std::integral_constant<int, 3 * (i - 2)>{});
          vop(std::integral_constant<int, 3 * (i - 2) + 1>{});
          vop(std::integral_constant<int, 3 * (i - 2) + 2>{});
        }
        if constexpr (i == 22) { vop(std::integral_constant<int, 60>{}); vop(std::integral_constant<int, 61>{}); }
        if constexpr (i == 23) { vop(std::integral_constant<int, 62>{}); vop(std::integral_constant<int, 63>{}); }
        if constexpr (J == 0 && i == 11) { de3_ld_q(0, at); de3_ld_q(1, at); de3_ld_q(4, at); de3_ld_q(5, at); }   // s2 = 0
        if constexpr (J == 0 && i == 23) { de3_ld_q(2, at); de3_ld_q(3, at); de3_ld_q(6, at); de3_ld_q(7, at); }   // s2 = 1
        if constexpr (J == 1 && i == 8) { de3_ld_a(0, at_next); de3_ld_a(1, at_next); }      // k-step 0
        if constexpr (J == 1 && i == 20) { de3_ld_a(2, at_next); de3_ld_a(3, at_next); }     // k-step 1
        if constexpr (J == 2 && i == 10) { de3_ld_a(4, at_next); de3_ld_a(5, at_next); }     // k-step 2
        if constexpr (J == 2 && i == 22) { de3_ld_a(6, at_next); de3_ld_a(7, at_next); }     // k-step 3
      });
    };

    unsigned at = tile_addr(0);                        // this lane's LDS address of the current tile, kept incrementally
    int slot = 0;
    auto tile = [&](int g, auto uni_tag) {
      // (past the last tile the recompute runs on this one and is dropped)
      unsigned at_next = at;
      if (g + 1 < ntile) {
        if ((g + 1) % MTB == 0) {
          slot = slot + 1 == NSLOT ? 0 : slot + 1;
          at_next = ring_lds + (unsigned)(slot * SLOT) + (unsigned)lane * 16u;
        } else {
          at_next = at + (unsigned)TILE;
        }
      }
      const int code0 = code_next;
      code_next = codes_lds[32 * min(g + 1, ntile - 1)];
      step(std::integral_constant<int, 0>{}, uni_tag, g, at, at_next, code0);
      // the next tile's first read (end of step 1): if it opens a stage, that stage has to have landed for every
      // wave; the same barrier frees the slot of the stage before the current one
      if ((g + 1) % MTB == 0 && g + 1 < ntile) {
        const int st = (g + 1) / MTB;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        wg_barrier();
        if (st + 1 < nstage) stage(st + 1);
      }
      step(std::integral_constant<int, 1>{}, uni_tag, g, at, at_next, code0);
      step(std::integral_constant<int, 2>{}, uni_tag, g, at, at_next, code0);
      step(std::integral_constant<int, 3>{}, uni_tag, g, at, at_next, code0);
      at = at_next;
    };
    for (int g = 0; g < ntile; ++g) {
      const bool uni = ((g < 64 ? uni_lo >> g : uni_hi >> (g - 64)) & 1ull) != 0;
      if (uni) tile(g, std::true_type{});
      else tile(g, std::false_type{});
    }
    de3_static_for<6>([&](auto ic) {                   // the pipeline's tail: C(last tile, NB - 1)
      c_mfma(ic, I0{}, std::integral_constant<int, NB - 1>{});
      c_mfma(ic, I1{}, std::integral_constant<int, NB - 1>{});
    });
  }

  // the accumulators are read 16 passes after the last MFMA at the earliest
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  // partial gradients in accumulator layout: [y][pt][dt][r][lane]
  de3_static_for<NB * DT * 16>([&](auto ic) {
    constexpr int i = decltype(ic)::value, nb = i / (DT * 16), dt = (i / 16) % DT, r = i % 16;
    const float v = de3_read_acc(i);
    if (pt0 + nb < a.n.PT)
      a.partial_de[((((size_t)blockIdx.y * a.n.PT + pt0 + nb) * DT + dt) * 16 + r) * 64 + lane] = v;
  });
}

}  // namespace

int nll_launch_bwd_de3(const NllArgs& a, int rows, hipStream_t s) {
  if (a.n.KS != 4 || a.n.DT != 2) return SPML_ERR_UNSUPPORTED;
  const unsigned groups = (unsigned)((a.n.PT + 15) / 16);        // 4 waves x 4 pixel tiles
  constexpr int LDS3 = kFwd2TilesPerChunk * 128 + 3 * 2 * (2 * 4 + 4 * 2) * 1024;
  if (a.mode & SPML_NLL_TAGSET) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_de3<4, 2, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
    hipLaunchKernelGGL((nll_bwd_de3<4, 2, true>), dim3(groups, (unsigned)rows), dim3(256), LDS3, s, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&nll_bwd_de3<4, 2, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, LDS3);
    hipLaunchKernelGGL((nll_bwd_de3<4, 2, false>), dim3(groups, (unsigned)rows), dim3(256), LDS3, s, a);
  }
  return launch_status();
}

}  // namespace spml
