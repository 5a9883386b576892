// Pixel-split fused k-means pass on 64-pixel tiles ("v4"): spherical k-means E-step + M-step of
// kmeans_with_initial_labels (segsort/common.py:67-97; find_nearest_prototypes :44-64,
// calculate_prototypes_from_labels :11-41) with X streamed from HBM once per pass.
//
// Same operands and arithmetic as kmeans_pass16 (kmeans.hip): pre-converted split-f16 fragment blocks
// DMA'd into LDS, h*h' + (h*l' + l*h') / 2048 on v_mfma_f32_16x16x32_f16 with the prototypes as
// accumulator ROWS, the M-step as a one-hot matrix product fed by the LDS transpose read, per-workgroup
// slabs, no atomics.  What changed is who does what:
//   * a tile is 64 pixels = two pre-tiles; wave w owns PIXELS 16 w .. 16 w + 15 in the E-step and
//     multiplies them with ALL prototype tiles (K <= 48: 3 x 16 rows), whose A fragments stay in
//     registers for the lifetime of the workgroup (Q = 8: 216 registers per lane -- the kernel is built
//     for ONE wave per SIMD and the whole 512-entry register file).  All four waves carry matrix work
//     (kmeans_pass16: 3 of 4), every B fragment is read from LDS once instead of three times, and a
//     pixel's 48 scores sit in the four lane groups of ONE wave: the arg-max is 12 compares in
//     registers + two cross-lane steps -- no candidate table, no label rebuild, no barrier for it;
//   * the M-step is split over the waves by 16-channel tile as before (the location tile's prototype
//     tiles go to different waves), 64 pixels per barrier instead of 32;
//   * two barriers per 64 pixels (tile landed / labels published), one in the E-only kernel.
// One workgroup per CU moves 2 x 34 KB per tile through a two-tile ring (136 KB of LDS at D = 258).
#include <stdlib.h>

#include <type_traits>

#include "kmeans_tile.hpp"

#ifndef SPML_P64_EXP
#define SPML_P64_EXP 0     // profiling builds: 1 no tile copies after the first, 2 no B reads, 4 no E MFMAs, 8 no M reads, 16 no M-step
#endif

namespace spml {

namespace {

constexpr int k64MaxMT = 3;


// registers a lane needs, roughly: prototype fragments + E accumulators + M accumulators + operands
__host__ __device__ constexpr int p64_regs(int mt, int q, int tail) {
  return (q + tail) * mt * 8 + 12 * mt + ((2 * q + 3) / 4 * mt * 4 + 4) + 96;
}
__host__ __device__ constexpr int p64_wgpc(int mt, int q, int tail) {
  return (p64_regs(mt, q, tail) <= 256 && 2 * p64_lds(q, tail) <= 160 * 1024) ? 2 : 1;
}

// One MFMA of the E-step with the prototype fragment in an ACCUMULATION register (gfx950: srcA may be one; the
// 216 fragment registers of a wave never pass through the architectural half) and an architectural accumulator
// (the arg-max reads it without v_accvgpr_read).  First product of a chain: srcC = 0.  Every hand-written MFMA
// carries two idle issue slots in front of it: the hardware does not interlock a vector-ALU write (a copy the
// register allocator may place) with an MFMA read straight behind it (tools/hw_probes/mfma_valu_raw.hip);
// tools/check_asm_hazards.py re-checks the compiled kernels.  The FIRST MFMA of a run (N_ = 1) carries the slots; the
// ones straight behind another hand-written MFMA (N_ = 0) do not -- 2 of 18 cycles per MFMA, and the checker still
// flags a copy the allocator might place in between.
#define P64_NOP_1 "s_nop 1\n\t"
#define P64_NOP_0 ""
#define P64_MFMA(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(afrag), "v"(bfrag))
#define P64_MFMAV(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(afrag), "v"(bfrag))
#define P64_MFMA0(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(afrag), "v"(bfrag))

template <int MT16, int Q, int TAIL, bool FUSED>
__global__ __launch_bounds__(256, p64_wgpc(MT16, Q, TAIL)) void kmeans_pass64(PassArgs a) {
  constexpr int QE = Q + TAIL;                   // k-steps incl. the location step
  constexpr int NDTW = (2 * Q + 3) / 4;          // full 16-channel tiles per wave (M-step)
  constexpr int PTB = p64_slot_bytes(Q, TAIL);
  constexpr int NDMA = 2 * (Q + TAIL);           // 1-KB copies a wave issues per tile
  constexpr int DPS = (NDMA + QE - 1) / QE;      // ... per k-step of the E loop
  typedef unsigned short us2 __attribute__((ext_vector_type(2)));
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  typedef short short4v __attribute__((vector_size(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4;             // lane group 0..3
  const int lc = lane & 15;             // column inside a 16-wide tile
  const int D = a.D, K = a.K;
  const int img = blockIdx.y, g = blockIdx.x;

  unsigned char* ring = lds;                                   // [2 tiles][2 pre-tiles][PTB]
  const unsigned ring_a = (unsigned)(size_t)(lptr_t)ring;
  const unsigned lab_a = ring_a + 4u * PTB;                    // [64] u16 labels of the tile

  KM_CLOCK_BEGIN
  const unsigned lane16 = 16u * (unsigned)lane;
  // the zero halves of the location blocks: the first ring slot now, the second one once the prototype
  // fragments that travel through it have been read
  auto zero_halves = [&](int sl0) {
    if (TAIL) {
      for (int i = tid; i < 2 * 4 * 64; i += 256) {
        const int sl = sl0 + (i >> 8), blk = (i >> 6) & 3, w = i & 63;
        reinterpret_cast<float*>(ring + (size_t)sl * PTB + Q * 4096 + blk * 512 + 256)[w] = 0.f;
      }
    }
  };
  zero_halves(0);
  // the image's range by a scalar load (its wait is on lgkmcnt: the vector-memory counter stays with the copies)
  uint4v segv;
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(segv) : "s"(a.seg_off + img));
  // (right behind the fetch of the image's range: the copy of the prototype fragments -- it depends on nothing but
  // the kernel arguments and is in flight while the range is still on its way)
  constexpr int NB = MT16 * QE;                                // 1-KB blocks per split half
  static_assert(2 * NB * 1024 <= 2 * PTB, "the prototype fragments must fit into one ring slot");
  {
    const unsigned char* ph = reinterpret_cast<const unsigned char*>(a.cent_h) + (size_t)img * NB * 1024;
    const unsigned char* pl = reinterpret_cast<const unsigned char*>(a.cent_l) + (size_t)img * NB * 1024;
    unsigned char* dst = ring + 2 * PTB;
#pragma unroll
    for (int i = 0; i < (2 * NB + 3) / 4; ++i) {
      const int blk = wave + 4 * i;                            // wave-uniform
      if (blk < 2 * NB) {
        const unsigned char* src = blk < NB ? ph + (size_t)blk * 1024 : pl + (size_t)(blk - NB) * 1024;
        __builtin_amdgcn_global_load_lds((gptr_t)(src + lane16), (lptr_t)(dst + blk * 1024), 16, 0, 0);
      }
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(segv));
  const int64_t seg0 = (int64_t)(((uint64_t)segv[1] << 32) | segv[0]);
  const int64_t len = (int64_t)(((uint64_t)segv[3] << 32) | segv[2]) - seg0;
  const int64_t T32 = (len + 31) >> 5;                         // pre-tiles of the image
  const int64_t T = (T32 + 1) >> 1;                            // 64-pixel tiles
  const int64_t t_step = a.G;
  if (g >= T) {
    if (FUSED) {
      float* z = a.slabs + ((size_t)img * a.G + g) * K * D;
      for (int i = tid; i < K * D; i += 256) z[i] = 0.f;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // (no copy may land in LDS this workgroup has given back)
    KM_CLOCK_END
    return;
  }


  // ---- tile copy: op i of a wave = 1-KB block (wave + 4 b) of pre-tile pt (i = pt * (Q + TAIL) + b), or the
  // wave's 256-B location block (b == Q); uniform base + 16 * lane
  const int64_t tile0 = pre_tile0(seg0, img);
  auto dma_op = [&](int64_t t, int slot, int i) {
    const int pt = i / (Q + TAIL), b = i % (Q + TAIL);
    // (the last tile of an image may be half a tile: its second half is a second copy of the first -- finite
    // numbers whose pixels get the label -1 -- so that the tile loop is ONE straight path)
    const int64_t p32 = 2 * t + pt < T32 ? 2 * t + pt : T32 - 1;
    const unsigned char* sb = a.xc + (size_t)(tile0 + p32) * pre_tile_bytes(Q, TAIL);   // uniform
    unsigned char* dst0 = ring + (size_t)(slot * 2 + pt) * PTB;
    if (b < Q) {
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)(wave + 4 * b) * 1024 + lane16),
                                       (lptr_t)(dst0 + (wave + 4 * b) * 1024), 16, 0, 0);
    } else if (lane < 16) {
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)Q * 4096 + wave * 256 + lane16),
                                       (lptr_t)(dst0 + Q * 4096 + wave * 512), 16, 0, 0);
    }
  };

  KM_TRACE_DECL
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(g, 0, i);

  // ---- all prototype rows -> accumulation registers (A operands), once per workgroup: the fragment-major
  // arrays (kmeans_normalize frag = 1) travel through the still unused second ring slot -- NB 1-KB copies
  // shared by the four waves instead of 2 NB global reads per wave (a quarter of the L2 -> CU traffic of the
  // start-up burst of 256 workgroups) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  wg_barrier();
  half8 ah[MT16][QE], al[MT16][QE];
  {
    const unsigned char* src = ring + 2 * PTB + lane16;
#pragma unroll
    for (int q = 0; q < MT16; ++q)
#pragma unroll
      for (int s = 0; s < QE; ++s) {
        ah[q][s] = *reinterpret_cast<const half8*>(src + (q * QE + s) * 1024);
        al[q][s] = *reinterpret_cast<const half8*>(src + (NB + q * QE + s) * 1024);
      }
  }
  if (TAIL) {
    wg_barrier();                          // every wave has its fragments: the second slot is free
    zero_halves(2);
  }
  // score bias of this lane's rows of the LAST prototype tile: its padding rows (c >= K, all-zero fragments,
  // score 0) must never win against negative scores
  float4a pen;
#pragma unroll
  for (int r = 0; r < 4; ++r) pen[r] = 16 * (MT16 - 1) + 4 * lg + r < K ? 0.f : -INFINITY;

  // ---- M-step accumulators: sums^T[d][k]; this wave owns the channel tiles w, w + 4, ... and
  // (tq < MT16) the prototype tile tq of the location tile ----
  // (hand-written MFMAs: the hardware does not interlock a dependent MFMA issued straight behind its producer
  // -- tools/hw_probes/mfma_chain.hip: wrong sums at distance 1, exact from distance 2 -- so no accumulator is
  // used by two consecutive MFMAs: the low-half products of the location tile, and of every tile when there is
  // only one prototype tile, go to accumulators of their own that are added at the end)
  constexpr int NLO = MT16 == 1 ? NDTW : 1;
  float4a macc[NDTW][MT16], mlo[NLO];
  float4a macc_ta[2] = {float4a{0.f, 0.f, 0.f, 0.f}, float4a{0.f, 0.f, 0.f, 0.f}};     // location tile: hi / lo products
#pragma unroll
  for (int j = 0; j < NDTW; ++j)
#pragma unroll
    for (int q = 0; q < MT16; ++q) macc[j][q] = float4a{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NLO; ++j) mlo[j] = float4a{0.f, 0.f, 0.f, 0.f};
  const int tq = (wave - 2 * Q) & 3;             // location tile: prototype tile q goes to wave (q + 2Q) & 3
  const int tqc = tq < MT16 ? tq : 0;

  // per-lane LDS offsets (the tile base is added per tile; everything else is an instruction offset)
  const int pt_e = wave >> 1, n_e = wave & 1;    // E-step: pre-tile and pixel half of this wave
  // E-step operand (pixel lc, channel group lg) of k-step s: + s * 4096 (hi), + 1024 (lo)
  const unsigned e_off = (unsigned)(pt_e * PTB + n_e * 2048 + frag_slot(lc, lg) * 16);
  // ... of the location k-step: lane group 0 reads the pixel's slot, the others the zero half; + 512 (lo)
  const unsigned e_off_t = (unsigned)(pt_e * PTB + Q * 4096 + n_e * 1024 + (lg == 0 ? lc * 16 : 256));
  // M-step transpose read (row lc>>2 of a [4 pixel][16 channel] sub-block) of channel tile wave + 4 j of
  // pre-tile pt: + pt * PTB + 2 j * 4096, + 512 (pixels 4..7), + 1024 (lo)
  const unsigned m_off = (unsigned)((wave >> 1) * 4096 + (wave & 1) * 256 + (lg >> 1) * 2048 +
                                    frag_slot(8 * (lg & 1) + (lc >> 2), (lc >> 1) & 1) * 16 + 8 * (lc & 1));
  // ... of the location tile: + pt * PTB, + 64 (pixels 4..7), + 512 (lo)
  const unsigned m_off_t = (unsigned)(Q * 4096 + (lg >> 1) * 1024 +
                                      (((lc >> 1) & 1) * 16 + 8 * (lg & 1) + (lc >> 2)) * 16 + 8 * (lc & 1));
  // one-hot operands: (label == 16 q + lc) as packed u16 arithmetic
  unsigned cq[MT16];
#pragma unroll
  for (int q = 0; q < MT16; ++q) cq[q] = 0x10001u * (unsigned)(16 * q + lc);
  unsigned k_one = 0x00010001u, k_h = 0x3C003C00u, k_nh = 0xC400C400u, k_l = 0x10001000u, k_nl = 0xF000F000u;
  asm volatile("" : "+v"(k_one), "+v"(k_h), "+v"(k_nh), "+v"(k_l), "+v"(k_nl));     // (keep them in registers)

  int it = 0;
  KM_MARK(7)
  for (int64_t t = g; t < T; t += t_step, ++it) {
    const int slot = it & 1;
    const unsigned tile_a = ring_a + (unsigned)(slot * 2 * PTB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                               // tile t landed; the other ring slot is free
    KM_MARK(0)
    const bool more = !(SPML_P64_EXP & 1) && t + t_step < T;

    // ================= E-step: 16 pixels x all prototype tiles =================
    // (the copy of the next tile is issued from inside the loop, DPS blocks per k-step; the loop exists
    // with and without it so that no branch sits between the MFMAs)
    int mylab = -1;
    auto estep = [&](auto more_tag) {
      constexpr bool MORE = decltype(more_tag)::value;
      float4a eh[MT16], ex[MT16], ey[MT16];
      // B fragments: three buffers by k-step, issued by hand with counted waits (LDS returns in order: "at most 2
      // outstanding" == "the older pair has landed").  The reads of k-step s + 2 and the two tile copies of a k-step
      // go out BETWEEN its three MFMA groups: a vector-memory / LDS instruction issued while the matrix pipe still
      // runs the previous MFMA costs nothing, the same instructions in a row behind the last MFMA cost their full
      // issue time (~50 cycles per copy) with the pipe idle
      half8 bh[3], bl[3];
      const unsigned eb = tile_a + e_off, ebt = tile_a + e_off_t;
#define P64_LOADB(s_, u_)                                                                                        \
      if ((s_) < Q)                                                                                              \
        asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                            \
                     : "=&v"(bh[u_]), "=&v"(bl[u_]) : "v"(eb), "i"((s_) * 4096), "i"((s_) * 4096 + 1024));       \
      else                                                                                                       \
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512"                                     \
                     : "=&v"(bh[u_]), "=&v"(bl[u_]) : "v"(ebt));
      P64_LOADB(0, 0)
      if (QE > 1) { P64_LOADB(1, 1) }
#pragma unroll
      for (int s = 0; s < QE; ++s) {
        const int u = s % 3;
        if (s + 1 < QE)
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[u]), "+v"(bl[u]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[u]), "+v"(bl[u]));
        if (s == 0) {
#pragma unroll
          for (int q = 0; q < MT16; ++q) { if (q == 0) { P64_MFMA0(1, eh[q], ah[q][s], bh[u]); } else { P64_MFMA0(0, eh[q], ah[q][s], bh[u]); } }
        } else {
#pragma unroll
          for (int q = 0; q < MT16; ++q) { if (q == 0) { P64_MFMA(1, eh[q], ah[q][s], bh[u]); } else { P64_MFMA(0, eh[q], ah[q][s], bh[u]); } }
        }
        if (s + 2 < QE) {
          if ((s + 2) % 3 == 0) { P64_LOADB(s + 2, 0) } else if ((s + 2) % 3 == 1) { P64_LOADB(s + 2, 1) } else { P64_LOADB(s + 2, 2) }
        }
        if (s == 0) {
#pragma unroll
          for (int q = 0; q < MT16; ++q) P64_MFMA0(0, ex[q], ah[q][s], bl[u]);
        } else {
#pragma unroll
          for (int q = 0; q < MT16; ++q) P64_MFMA(0, ex[q], ah[q][s], bl[u]);
        }
        if (MORE && s * DPS < NDMA) dma_op(t + t_step, slot ^ 1, s * DPS);
        if (s == 0) {
#pragma unroll
          for (int q = 0; q < MT16; ++q) P64_MFMA0(0, ey[q], al[q][s], bh[u]);
        } else {
#pragma unroll
          for (int q = 0; q < MT16; ++q) P64_MFMA(0, ey[q], al[q][s], bh[u]);
        }
        if (MORE) {
#pragma unroll
          for (int i = s * DPS + 1; i < (s + 1) * DPS && i < NDMA; ++i) dma_op(t + t_step, slot ^ 1, i);
        }
      }
#undef P64_LOADB
      // (an MFMA result is read by the vector ALU: the hardware wants idle issue slots in between, and the
      // compiler does not see inside the asm statements)
      asm volatile("s_nop 15\n\ts_nop 7");
#pragma unroll
      for (int q = 0; q < MT16; ++q) asm volatile("" : "+v"(eh[q]), "+v"(ex[q]), "+v"(ey[q]));   // (reads stay behind the nops)
      KM_MARK(3)
      float best = -INFINITY;
      int best_i = 0x7fff;
#pragma unroll
      for (int q = 0; q < MT16; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) {                      // ascending prototype rows: ties -> lowest
          float sdot = eh[q][r] + (ex[q][r] + ey[q][r]) * kSplitInv;
          if (q == MT16 - 1) sdot += pen[r];
          if (sdot > best) { best = sdot; best_i = 16 * q + 4 * lg + r; }
        }
      // the four lane groups hold different prototype rows of the same pixel: after a swap of the odd / even
      // rows of 16 lanes (then of the wave halves) every lane sees both candidates of its pair
#pragma unroll
      for (int step = 0; step < 2; ++step) {
        const unsigned bv = __builtin_bit_cast(unsigned, best);
        const auto sv = step == 0 ? __builtin_amdgcn_permlane16_swap(bv, bv, false, false)
                                  : __builtin_amdgcn_permlane32_swap(bv, bv, false, false);
        const auto si = step == 0 ? __builtin_amdgcn_permlane16_swap((unsigned)best_i, (unsigned)best_i, false, false)
                                  : __builtin_amdgcn_permlane32_swap((unsigned)best_i, (unsigned)best_i, false, false);
        const float v0 = __builtin_bit_cast(float, (unsigned)sv[0]), v1 = __builtin_bit_cast(float, (unsigned)sv[1]);
        const int i0 = (int)si[0], i1 = (int)si[1];
        const bool take1 = v1 > v0 || (v1 == v0 && i1 < i0);
        best = take1 ? v1 : v0;
        best_i = take1 ? i1 : i0;
      }
      mylab = (int64_t)t * 64 + 16 * wave + lc < len ? best_i : -1;
    };
    if (more) estep(std::true_type{});
    else estep(std::false_type{});

    if (FUSED) {
      // (every LDS access of the M-step is hand-issued: the compiler orders its own LDS reads behind the
      // LDS-DMA of the NEXT tile with an s_waitcnt vmcnt(0), which would serialise copy and M-step)
      if (lg == 0) asm volatile("ds_write_b16 %0, %1" :: "v"(lab_a + 2u * (unsigned)(16 * wave + lc)), "v"(mylab) : "memory");
      wg_barrier();                              // labels of the 64 pixels published
      KM_MARK(4)
      // ================= M-step: X^T * one-hot, 32 pixels per product =================
      // units of work: (pre-tile, 16-channel tile of this wave) and, last, (pre-tile, location tile);
      // operands of unit u + 1 are in flight while the MFMAs of unit u run (LDS returns in order);
      // buffer = u % 3.  The pipeline is STRAIGHT-LINE code: a branch between a hand-issued read and its
      // counted wait lets the compiler copy the destination registers before the data has landed, and a
      // join of two versions of the accumulators makes it copy them around the hand-written MFMAs.  So
      // every wave runs every unit (a wave without a location / channel tile of its own accumulates
      // into registers nobody stores) and every tile has two pre-tiles (see dma_op).
      const unsigned mb = tile_a + m_off, mbt = tile_a + m_off_t;
      union XA { short4v p[2]; half8 h; };
      union OH { half8 h; uint4v u; };
      {
        constexpr int NPT = 2;                                   // pre-tiles of a tile
        constexpr int NF = NPT * NDTW;                           // full-tile units
        constexpr int NUT = NF + (TAIL ? NPT : 0);
        XA xa[3][2];                                             // [buffer][hi|lo]
        // A operand = X^T (rows = channels, k = pixels) straight out of the channel-major fragment blocks
        // with the LDS transpose read: lane lc receives channel 16*dt + lc of the pixels 8*lg .. 8*lg+7
        // (two reads of 4 pixels); location channels: the same on the lane-linear location block
#define P64_LOADX(u_, b_)                                                                                             \
        if ((u_) < NF)                                                                                                \
          asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"               \
                       "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"                   \
                       : "=&v"(xa[b_][0].p[0]), "=&v"(xa[b_][0].p[1]), "=&v"(xa[b_][1].p[0]), "=&v"(xa[b_][1].p[1])   \
                       : "v"(mb), "i"(((u_) / NDTW) * PTB + ((u_) % NDTW) * 8192),                                    \
                         "i"(((u_) / NDTW) * PTB + ((u_) % NDTW) * 8192 + 512),                                       \
                         "i"(((u_) / NDTW) * PTB + ((u_) % NDTW) * 8192 + 1024),                                      \
                         "i"(((u_) / NDTW) * PTB + ((u_) % NDTW) * 8192 + 1536));                                     \
        else                                                                                                          \
          asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"               \
                       "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"                   \
                       : "=&v"(xa[b_][0].p[0]), "=&v"(xa[b_][0].p[1]), "=&v"(xa[b_][1].p[0]), "=&v"(xa[b_][1].p[1])   \
                       : "v"(mbt), "i"(((u_) - NF) * PTB), "i"(((u_) - NF) * PTB + 64),                               \
                         "i"(((u_) - NF) * PTB + 512), "i"(((u_) - NF) * PTB + 576));
#define P64_LOADXB(u_)                                                          \
        if ((u_) % 3 == 0) { P64_LOADX(u_, 0) } else if ((u_) % 3 == 1) { P64_LOADX(u_, 1) } else { P64_LOADX(u_, 2) }
        uint4v lb[2];                                            // 8 u16 labels of this lane's pixel group
        {
          const unsigned la = lab_a + 2u * (unsigned)(8 * lg);
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:64" : "=&v"(lb[0]), "=&v"(lb[1]) : "v"(la));
        }
        // (two units ahead: a transpose read takes longer to come back than the six MFMAs of a unit run)
        P64_LOADXB(0)
        if (NUT > 1) {
          P64_LOADXB(1)
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(lb[0]), "+v"(lb[1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lb[0]), "+v"(lb[1]));
        }
        // one-hot B operands of the NPT pixel groups: t = |label - (16 q + lc)| clamped to 1 -> 1.0 - t (hi),
        // 2^-11 * (1 - t) (lo: the exact scale of the low split half), two labels per instruction
        OH oh[NPT][MT16], ol[NPT][MT16];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
          for (int q = 0; q < MT16; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              // (asm: the compiler turns min(x, 1) * k + c into 16-bit compares and selects, 3x the instructions)
              unsigned d, m;
              asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"((unsigned)lb[pt][i]), "v"(cq[q]));
              asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(k_one));
              asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(oh[pt][q].u[i]) : "v"(m), "v"(k_nh), "v"(k_h));   // 0x3C00 (1.0) or 0
              asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(ol[pt][q].u[i]) : "v"(m), "v"(k_nl), "v"(k_l));   // 0x1000 (2^-11) or 0
            }
        // the location tile's one-hot operands (prototype tile tqc of this wave)
        half8 oht[NPT], olt[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
          oht[pt] = oh[pt][0].h;
          olt[pt] = ol[pt][0].h;
#pragma unroll
          for (int q = 1; q < MT16; ++q)
            if (tqc == q) { oht[pt] = oh[pt][q].h; olt[pt] = ol[pt][q].h; }
        }
        // (vector ALU results feed hand-written MFMAs: keep the idle slots the hardware wants in between)
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
          for (int q = 0; q < MT16; ++q) asm volatile("" : "+v"(oh[pt][q].u), "+v"(ol[pt][q].u));
          asm volatile("" : "+v"(oht[pt]), "+v"(olt[pt]));
        }
        asm volatile("s_nop 7");
        KM_MARK(5)
#pragma unroll
        for (int u = 0; u < NUT; ++u) {
          const int b = u % 3;
          if (u + 2 < NUT) {
            P64_LOADXB(u + 2)
            asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(xa[b][0].p[0]), "+v"(xa[b][0].p[1]),
                                                  "+v"(xa[b][1].p[0]), "+v"(xa[b][1].p[1]));
          } else if (u + 1 < NUT) {
            asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[b][0].p[0]), "+v"(xa[b][0].p[1]),
                                                  "+v"(xa[b][1].p[0]), "+v"(xa[b][1].p[1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[b][0].p[0]), "+v"(xa[b][0].p[1]),
                                                  "+v"(xa[b][1].p[0]), "+v"(xa[b][1].p[1]));
          }
          // (in-place accumulation, written out: the compiler rotates the accumulators through fresh registers,
          // and a dependent MFMA whose srcC is not its own destination waits for the write-back instead of
          // taking the forwarded result -- 160 instead of 100 cycles per unit)
          if (u < NF) {
            const int pt = u / NDTW, j = u % NDTW;
#pragma unroll
            for (int q = 0; q < MT16; ++q) { if (q == 0) { P64_MFMAV(1, macc[j][q], xa[b][0].h, oh[pt][q].h); } else { P64_MFMAV(0, macc[j][q], xa[b][0].h, oh[pt][q].h); } }
            if (MT16 == 1) {
              P64_MFMAV(0, mlo[j], xa[b][1].h, ol[pt][0].h);
            } else {
#pragma unroll
              for (int q = 0; q < MT16; ++q) P64_MFMAV(0, macc[j][q], xa[b][1].h, ol[pt][q].h);
            }
          } else {
            const int pt = u - NF;
            P64_MFMAV(1, macc_ta[0], xa[b][0].h, oht[pt]);
            P64_MFMAV(0, macc_ta[1], xa[b][1].h, olt[pt]);
          }
        }
#undef P64_LOADXB
#undef P64_LOADX
      }
      KM_MARK(6)
    }
    // labels leave after the M-step: by then the tile copy issued above has drained from the CU's
    // vector-memory queue and the store does not stall behind it
    if (lg == 0 && mylab >= 0) label_store(a, seg0 + t * 64 + 16 * wave + lc, mylab);
  }

  if (FUSED) {
    // (the sums were last written by hand-written MFMAs: idle slots before the stores read them)
    asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
    for (int j = 0; j < NDTW; ++j)
#pragma unroll
      for (int q = 0; q < MT16; ++q) asm volatile("" : "+v"(macc[j][q]));
#pragma unroll
    for (int j = 0; j < NLO; ++j) asm volatile("" : "+v"(mlo[j]));
    asm volatile("" : "+v"(macc_ta[0]), "+v"(macc_ta[1]));
    if (MT16 == 1) {
#pragma unroll
      for (int j = 0; j < NDTW; ++j) macc[j][0] += mlo[j];
    }
    macc_ta[0] += macc_ta[1];
    float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
#pragma unroll
    for (int j = 0; j < NDTW; ++j) {
      const int dt = wave + 4 * j;
      if (dt < 2 * Q) {
#pragma unroll
        for (int q = 0; q < MT16; ++q) {
          const int c = 16 * q + lc;
          // rows 4*lg .. 4*lg+3 of channel tile dt: two 8-byte stores (D is even or the row is 4-B aligned)
          const int d = 16 * dt + 4 * lg;
          if (c < K) {
            float* dst = slab + (size_t)c * D + d;
            if (D & 1) {
#pragma unroll
              for (int r = 0; r < 4; ++r) __builtin_nontemporal_store(macc[j][q][r], dst + r);
            } else {
              // (non-temporal: the slab is read by another kernel; lines that do not wait dirty in L2 shorten the
              // write-back at the end of the launch -- back-to-back launches take 108-117 k cycles with them and
              // 113-122 k with plain stores, although a whole k-means iteration is 1.5 us FASTER with plain ones:
              // kmeans_reduce_slabs then finds the 9.5 MB in L2 / MALL.  kmeans_accum64k, 38 MB of slabs, stores plain)
              typedef float float2v __attribute__((ext_vector_type(2)));
              __builtin_nontemporal_store(float2v{macc[j][q][0], macc[j][q][1]}, reinterpret_cast<float2v*>(dst));
              __builtin_nontemporal_store(float2v{macc[j][q][2], macc[j][q][3]}, reinterpret_cast<float2v*>(dst) + 1);
            }
          }
        }
      }
    }
    if (TAIL && tq < MT16) {
      const int c = 16 * tq + lc;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int d = 32 * Q + 4 * lg + r;
        if (c < K && d < D) __builtin_nontemporal_store(macc_ta[0][r], slab + (size_t)c * D + d);
      }
    }
  }
  KM_TRACE_DRAIN
  KM_MARK(1)
  KM_TRACE_STORE
  KM_CLOCK_END
}
#undef P64_MFMA
#undef P64_MFMA0
#undef P64_MFMAV
#undef P64_NOP_0
#undef P64_NOP_1

template <int MT16, int Q, int TAIL>
int launch64_t(const PassArgs& a, hipStream_t s) {
  const int lds = p64_lds(Q, TAIL);
  if (a.do_accum) {
    auto kern = kmeans_pass64<MT16, Q, TAIL, true>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(a.G, a.n_img), dim3(256), lds, s, a);
  } else {
    auto kern = kmeans_pass64<MT16, Q, TAIL, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(kern, dim3(a.G, a.n_img), dim3(256), lds, s, a);
  }
  return launch_status();
}

}  // namespace

bool pass64_shape(int D, int K) {
  const int q = D / 32, tl = D - 32 * q;
  return K >= 1 && K <= 16 * k64MaxMT && tl <= 8 && (q == 1 || q == 2 || q == 4 || q == 8);
}

int pass64_wg_per_cu(int D, int K) {
  const int q = D / 32, tl = D - 32 * q;
  return p64_wgpc((K + 15) / 16, q, tl ? 1 : 0);
}

size_t pass64_lds_bytes(int D) {
  const int q = D / 32, tl = D - 32 * q;
  return (size_t)p64_lds(q, tl ? 1 : 0);
}

int launch_pass64(const PassArgs& a, hipStream_t s) {
  if (!a.do_assign || !a.xc || !pass64_shape(a.D, a.K)) return SPML_ERR_UNSUPPORTED;
  const int q = a.D / 32, tail = (a.D - 32 * q) ? 1 : 0, mt = (a.K + 15) / 16;
#define SPML_P64(M_, Q_)                                                         \
  if (mt == M_ && q == Q_) return tail ? launch64_t<M_, Q_, 1>(a, s) : launch64_t<M_, Q_, 0>(a, s);
#define SPML_P64Q(M_) SPML_P64(M_, 1) SPML_P64(M_, 2) SPML_P64(M_, 4) SPML_P64(M_, 8)
  SPML_P64Q(1) SPML_P64Q(2) SPML_P64Q(3)
#undef SPML_P64Q
#undef SPML_P64
  return SPML_ERR_UNSUPPORTED;
}

}  // namespace spml
