// E-step of spherical k-means for 64 < K <= 144 prototypes on pre-converted 64-pixel tiles: the assign half
// of kmeans_with_initial_labels (segsort/common.py:44-64 find_nearest_prototypes) for the 12 x 12 grid of the
// reference's 513 x 513 configuration (K = 144, D = 258), with X streamed from HBM once per pass.
//
// kmeans_pass64 (kmeans64.hip) keeps the fragments of ALL prototype tiles in the registers of every wave, which ends
// at three tiles (K <= 48).  Here the prototype tiles are split over EIGHT waves (two per SIMD, <= 256 registers):
//   * wave w owns prototype tile w and multiplies it with ALL 64 pixels of a tile (four groups of 16 pixels, its own
//     group first); the ninth tile (K > 128) is multiplied by waves 0..3 with their own pixel group only -- K = 144:
//     nine tile products per SIMD and 64 pixels, no padded tile, 144 fragment registers per wave at D = 258;
//   * a wave leaves (score, index) of the best row of its tile per pixel in LDS; after one barrier wave w < 4 merges the
//     eight candidates of its 16 pixels with the best of the shared tile (ascending prototype order, ties to the
//     lowest index -- torch.argmax) and stores the labels;
//   * while one wave of a SIMD runs the arg-max of a pixel group on the vector ALU, waits for LDS or blocks in the
//     issue of a tile copy, the other one feeds the matrix core.  (A four-wave variant with three tiles per wave and
//     one wave per SIMD spent 40 % of its time in those phases with the matrix cores idle: 80 instead of 74 us per
//     pass, profiles/r05_kmeans_k144.md.)
// Same operands, arithmetic (h*h' + (h*l' + l*h') / 2048 on v_mfma_f32_16x16x32_f16) and hand-issued schedule as
// kmeans_pass64.  The M-step runs as a pass of its own (kmeans_accum64k below): its accumulators and one-hot operands
// for nine prototype tiles do not fit next to the fragments.
#include <stdlib.h>

#include <type_traits>

#include "kmeans_tile.hpp"

namespace spml {

namespace {

// (N_ = 1: two idle issue slots in front, N_ = 0: none, for an MFMA straight behind another hand-written one --
// kmeans64.hip.  The assign kernel keeps them everywhere: at 240 registers the allocator reloads a fragment register
// between two of its MFMAs in one instantiation, which tools/check_asm_hazards.py flagged)
#define P64_NOP_1 "s_nop 1\n\t"
#define P64_NOP_0 ""
#define P64_MFMA(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "a"(afrag), "v"(bfrag))
#define P64_MFMA0(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc) : "a"(afrag), "v"(bfrag))

template <int MT, int Q, int TAIL>
__global__ __launch_bounds__(512, 2) void kmeans_assign64k(PassArgs a) {
  constexpr int QE = Q + TAIL;
  constexpr int NW = MT > 8 ? 2 : 1;             // tiles whose fragments a wave keeps
  constexpr int PTB = p64_slot_bytes(Q, TAIL);
  constexpr int NFULL = Q / 2;                   // 1-KB copies of a wave per pre-tile (4 Q blocks over 8 waves)
  constexpr int NDMA = 2 * NFULL + (TAIL ? 1 : 0);
  static_assert(MT >= 5 && MT <= 9 && (Q & 1) == 0 && NDMA <= QE, "tiles 0..7 on the eight waves, tile 8 shared");
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  typedef unsigned uint2v __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
  const int K = a.K;
  const int img = blockIdx.y, g = blockIdx.x;

  unsigned char* ring = lds;                                   // [2 tiles][2 pre-tiles][PTB]
  const unsigned ring_a = (unsigned)(size_t)(lptr_t)ring;
  const unsigned cand_a = ring_a + 4u * PTB;                   // [8 waves][4 lane groups][64 pixels] (score, index)

  KM_CLOCK_BEGIN
  const unsigned lane16 = 16u * (unsigned)lane;
  if (TAIL) {                                                  // the zero halves of the location blocks
    for (int i = tid; i < 4 * 4 * 64; i += 512) {
      const int sl = i >> 8, blk = (i >> 6) & 3, w = i & 63;
      reinterpret_cast<float*>(ring + (size_t)sl * PTB + Q * 4096 + blk * 512 + 256)[w] = 0.f;
    }
  }
  uint4v segv;
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(segv) : "s"(a.seg_off + img));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(segv));
  const int64_t seg0 = (int64_t)(((uint64_t)segv[1] << 32) | segv[0]);
  const int64_t len = (int64_t)(((uint64_t)segv[3] << 32) | segv[2]) - seg0;
  const int64_t T32 = (len + 31) >> 5;
  const int64_t T = (T32 + 1) >> 1;
  const int64_t t_step = a.G;
  if (g >= T) {
    KM_CLOCK_END
    return;
  }

  // tile copy: op i < 2 NFULL: 1-KB block (wave + 8 b) of pre-tile pt (i = pt * NFULL + b); op 2 NFULL: the 256-B
  // location block (wave & 3) of pre-tile (wave >> 2)
  const int64_t tile0 = pre_tile0(seg0, img);
  auto dma_op = [&](int64_t t, int slot, int i) {
    const bool loc = i >= 2 * NFULL;
    const int pt = loc ? (wave >> 2) : i / NFULL, b = i % NFULL;
    const int64_t p32 = 2 * t + pt < T32 ? 2 * t + pt : T32 - 1;
    const unsigned char* sb = a.xc + (size_t)(tile0 + p32) * pre_tile_bytes(Q, TAIL);   // uniform
    unsigned char* dst0 = ring + (size_t)(slot * 2 + pt) * PTB;
    if (!loc) {
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)(wave + 8 * b) * 1024 + lane16),
                                       (lptr_t)(dst0 + (wave + 8 * b) * 1024), 16, 0, 0);
    } else if (lane < 16) {
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)Q * 4096 + (wave & 3) * 256 + lane16),
                                       (lptr_t)(dst0 + Q * 4096 + (wave & 3) * 512), 16, 0, 0);
    }
  };
#pragma unroll
  for (int i = 0; i < NDMA; ++i) dma_op(g, 0, i);

  // fragments: tile min(wave, MT - 1) (waves >= MT of a K <= 112 call repeat the last tile; their candidates lose
  // every tie against the owner's: same score, same index) and, K > 128, tile 8
  int tile_of[NW];
  tile_of[0] = wave < MT ? wave : MT - 1;
  if (NW > 1) tile_of[NW - 1] = 8;
  half8 ah[NW][QE], al[NW][QE];
  {
    const unsigned char* ph = reinterpret_cast<const unsigned char*>(a.cent_h) + (size_t)img * MT * QE * 1024 + lane16;
    const unsigned char* pl = reinterpret_cast<const unsigned char*>(a.cent_l) + (size_t)img * MT * QE * 1024 + lane16;
#pragma unroll
    for (int i = 0; i < NW; ++i)
#pragma unroll
      for (int s = 0; s < QE; ++s) {
        ah[i][s] = *reinterpret_cast<const half8*>(ph + (size_t)(tile_of[i] * QE + s) * 1024);
        al[i][s] = *reinterpret_cast<const half8*>(pl + (size_t)(tile_of[i] * QE + s) * 1024);
      }
  }
  float4a pen[NW];
  int row0[NW];
#pragma unroll
  for (int i = 0; i < NW; ++i) {
    row0[i] = 16 * tile_of[i] + 4 * lg;
#pragma unroll
    for (int r = 0; r < 4; ++r) pen[i][r] = row0[i] + r < K ? 0.f : -INFINITY;
  }
  const unsigned e_lane = (unsigned)(frag_slot(lc, lg) * 16);
  const unsigned e_lane_t = (unsigned)(Q * 4096 + (lg == 0 ? lc * 16 : 256));

  auto across_groups = [&](float& best, int& best_i) {
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
  };

  // one pixel group (16 pixels) x NT tiles of this wave; FIRST: the group's k-steps also carry the tile copies
  float xbest = -INFINITY;
  int xbest_i = 0x7fff;
  auto group = [&](auto nt_tag, int pg, int i4, unsigned tile_a, int64_t t_next, int slot) {
    constexpr int NT = decltype(nt_tag)::value;
    const unsigned grp = (unsigned)((pg >> 1) * PTB);
    const unsigned eb = tile_a + grp + (unsigned)((pg & 1) * 2048) + e_lane;
    const unsigned ebt = tile_a + grp + (unsigned)((pg & 1) * 1024) + e_lane_t;
    float4a eh[NT], ex[NT], ey[NT];
    // B fragments: four buffers, the reads of k-step s + 3 go out behind the first MFMA group of s -- a k-step of this
    // kernel is 3 MFMAs (48 cycles of matrix pipe), an LDS read takes longer than two of them
    half8 bh[4], bl[4];
#define P64_LOADB(s_, u_)                                                                                        \
    if ((s_) < Q)                                                                                                \
      asm volatile("ds_read_b128 %0, %2 offset:%3\n\tds_read_b128 %1, %2 offset:%4"                              \
                   : "=&v"(bh[u_]), "=&v"(bl[u_]) : "v"(eb), "i"((s_) * 4096), "i"((s_) * 4096 + 1024));         \
    else                                                                                                         \
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:512"                                       \
                   : "=&v"(bh[u_]), "=&v"(bl[u_]) : "v"(ebt));
#define P64_LOADBQ(s_)                                                                                           \
    if ((s_) % 4 == 0) { P64_LOADB(s_, 0) } else if ((s_) % 4 == 1) { P64_LOADB(s_, 1) }                         \
    else if ((s_) % 4 == 2) { P64_LOADB(s_, 2) } else { P64_LOADB(s_, 3) }
    P64_LOADBQ(0)
    if (QE > 1) { P64_LOADBQ(1) }
    if (QE > 2) { P64_LOADBQ(2) }
    // (two waves per SIMD: the one that feeds the matrix core goes first, the other one's arg-max fills the gaps)
    asm volatile("s_setprio 3");
#pragma unroll
    for (int s = 0; s < QE; ++s) {
      const int u = s % 4;
      if (s + 2 < QE)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(bh[u]), "+v"(bl[u]));
      else if (s + 1 < QE)
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[u]), "+v"(bl[u]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[u]), "+v"(bl[u]));
      if (s == 0) {
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA0(1, eh[q], ah[q][s], bh[u]);
        if (s + 3 < QE) { P64_LOADBQ(s + 3) }
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA0(1, ex[q], ah[q][s], bl[u]);
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA0(1, ey[q], al[q][s], bh[u]);
      } else {
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA(1, eh[q], ah[q][s], bh[u]);
        if (s + 3 < QE) { P64_LOADBQ(s + 3) }
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA(1, ex[q], ah[q][s], bl[u]);
#pragma unroll
        for (int q = 0; q < NT; ++q) P64_MFMA(1, ey[q], al[q][s], bh[u]);
      }
      // the copy of the next tile goes out with the k-steps of the FIRST pixel group (NDMA <= QE): it then has three
      // quarters of the tile's time to land -- spread over the whole tile the last instructions had none, and every
      // tile began with a wait of one memory latency (74 -> .. us per pass)
      if (i4 == 0 && s < NDMA) dma_op(t_next, slot ^ 1, s);
    }
#undef P64_LOADBQ
#undef P64_LOADB
    asm volatile("s_setprio 0");
    asm volatile("s_nop 15\n\ts_nop 7");
#pragma unroll
    for (int q = 0; q < NT; ++q) asm volatile("" : "+v"(eh[q]), "+v"(ex[q]), "+v"(ey[q]));
    float best = -INFINITY;
    int best_i = 0x7fff;
#pragma unroll
    for (int r = 0; r < 4; ++r) {                        // ascending prototype rows: ties -> lowest
      const float sdot = eh[0][r] + (ex[0][r] + ey[0][r]) * kSplitInv + pen[0][r];
      if (sdot > best) { best = sdot; best_i = row0[0] + r; }
    }
    // every lane publishes the best of ITS four rows (table [wave][lane group][pixel]): the cross-lane reduction happens
    // once per tile in the merge instead of once per pixel group here
    {
      const uint2v cv = {__builtin_bit_cast(unsigned, best), (unsigned)best_i};
      asm volatile("ds_write_b64 %0, %1" :: "v"(cand_a + 8u * (unsigned)(256 * wave + 64 * lg + 16 * pg + lc)), "v"(cv) : "memory");
    }
    if (NT > 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sdot = eh[NT - 1][r] + (ex[NT - 1][r] + ey[NT - 1][r]) * kSplitInv + pen[NW - 1][r];
        if (sdot > xbest) { xbest = sdot; xbest_i = row0[NW - 1] + r; }
      }
    }
  };

  int it = 0;
  for (int64_t t = g; t < T; t += t_step, ++it) {
    const int slot = it & 1;
    const unsigned tile_a = ring_a + (unsigned)(slot * 2 * PTB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                               // tile t landed; the other ring slot and the candidate table are free
    const bool more = t + t_step < T;           // uniform
    const int64_t t_next = more ? t + t_step : t;   // (a workgroup's last tile copies itself once more: one path)
    xbest = -INFINITY;
    xbest_i = 0x7fff;
    // own pixel group first: waves 0..3 also multiply the shared ninth tile with it
    if (NW > 1 && wave < 4) group(std::integral_constant<int, NW>{}, wave & 3, 0, tile_a, t_next, slot);
    else group(std::integral_constant<int, 1>{}, wave & 3, 0, tile_a, t_next, slot);
#pragma unroll
    for (int i4 = 1; i4 < 4; ++i4) group(std::integral_constant<int, 1>{}, (wave + i4) & 3, i4, tile_a, t_next, slot);
    wg_barrier();                               // candidates of the 64 pixels published

    // merge (waves 0..3): pixel 16 wave + lc; lane group lg' takes the eight waves' candidates of ITS rows (ascending
    // prototype order: strict > keeps the lowest index), then the shared tile's, then one reduction over the lane groups
    if (wave < 4) {
      uint2v cv[8];
      const unsigned ca = cand_a + 8u * (unsigned)(64 * lg + 16 * wave + lc);
      asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:2048\n\t"
                   "ds_read_b64 %2, %8 offset:4096\n\tds_read_b64 %3, %8 offset:6144\n\t"
                   "ds_read_b64 %4, %8 offset:8192\n\tds_read_b64 %5, %8 offset:10240\n\t"
                   "ds_read_b64 %6, %8 offset:12288\n\tds_read_b64 %7, %8 offset:14336\n\ts_waitcnt lgkmcnt(0)"
                   : "=&v"(cv[0]), "=&v"(cv[1]), "=&v"(cv[2]), "=&v"(cv[3]), "=&v"(cv[4]), "=&v"(cv[5]), "=&v"(cv[6]),
                     "=&v"(cv[7]) : "v"(ca));
      float fb = -INFINITY;
      int fi = 0x7fff;
#pragma unroll
      for (int v = 0; v < 8; ++v) {
        const float sc = __builtin_bit_cast(float, (unsigned)cv[v][0]);
        if (sc > fb) { fb = sc; fi = (int)cv[v][1]; }
      }
      if (NW > 1 && xbest > fb) { fb = xbest; fi = xbest_i; }
      across_groups(fb, fi);
      const int64_t pix = t * 64 + 16 * wave + lc;
      if (lg == 0 && pix < len) label_store(a, seg0 + pix, fi);
    }
  }
  KM_CLOCK_END
}

template <int MT, int Q, int TAIL>
int launch64k_t(const PassArgs& a, hipStream_t s) {
  const int lds = 4 * p64_slot_bytes(Q, TAIL) + 16384;
  auto kern = kmeans_assign64k<MT, Q, TAIL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, dim3(a.G, a.n_img), dim3(512), lds, s, a);
  return launch_status();
}

// ---------------------------------------------------------------------------------------------------------------
// M-step of the same shapes (calculate_prototypes_from_labels, segsort/common.py:11-41: sums of the rows of X by
// label) on the same tiles and eight waves: sums^T[d][k] += X^T (LDS transpose read of the channel-major fragment
// blocks) x one-hot(label) as in kmeans_pass64, with wave w owning the 16-channel tiles w, w + 8 (all MT prototype
// tiles: 2 x 9 accumulators at K = 144, D = 258) and the product of the location tile with prototype tile w (+ the
// ninth, which every wave multiplies and wave 0 stores).  The one-hot operands of a 32-pixel pre-tile (72 registers
// for nine tiles) are rebuilt per pre-tile; while one wave of a SIMD builds them on the vector ALU the other one
// multiplies.  Labels come from memory (the assign kernel's int32 array, or the caller's int64 initial labels).
__device__ const int g_no_label[1] = {-1};      // source of the labels of pixels past the end of an image

#define K64_MFMA(N_, acc, afrag, bfrag) \
  asm volatile(P64_NOP_##N_ "v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(afrag), "v"(bfrag))

template <int MT, int Q, int TAIL>
__global__ __launch_bounds__(512, 2) void kmeans_accum64k(PassArgs a) {
  constexpr int PTB = p64_slot_bytes(Q, TAIL);
  constexpr int NDTW = Q / 4;                    // full 16-channel tiles per wave (2 Q tiles over 8 waves)
  constexpr int NFULL = Q / 2;                   // 1-KB copies of a wave per pre-tile
  constexpr int NU = NDTW + (TAIL ? 1 : 0);      // units per pre-tile
  static_assert(MT >= 5 && MT <= 9 && (Q & 3) == 0, "");
  typedef unsigned uint4v __attribute__((ext_vector_type(4)));
  typedef short short4v __attribute__((vector_size(8)));
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
  const int D = a.D, K = a.K;
  const int img = blockIdx.y, g = blockIdx.x;

  // The ring holds FOUR pre-tiles and is refilled one pre-tile at a time, three ahead (104 KB in flight per CU at
  // D = 258): with two-tile slots and one tile ahead every tile began with ~1 us of waiting for its copy -- 2.7 us
  // from issue to landing against 1.8 us of work (profiles/r05_kmeans_k144.md)
  unsigned char* ring = lds;                                   // [4 pre-tiles][PTB]
  const unsigned ring_a = (unsigned)(size_t)(lptr_t)ring;
  const unsigned lab_a = ring_a + 4u * PTB;                    // [4 pre-tiles][32] int32 labels

  KM_CLOCK_BEGIN
  const unsigned lane16 = 16u * (unsigned)lane;
  if (TAIL) {
    for (int i = tid; i < 4 * 4 * 64; i += 512) {
      const int sl = i >> 8, blk = (i >> 6) & 3, w = i & 63;
      reinterpret_cast<float*>(ring + (size_t)sl * PTB + Q * 4096 + blk * 512 + 256)[w] = 0.f;
    }
  }
  uint4v segv;
  asm volatile("s_load_dwordx4 %0, %1, 0x0" : "=s"(segv) : "s"(a.seg_off + img));
  asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(segv));
  const int64_t seg0 = (int64_t)(((uint64_t)segv[1] << 32) | segv[0]);
  const int64_t len = (int64_t)(((uint64_t)segv[3] << 32) | segv[2]) - seg0;
  const int64_t T32 = (len + 31) >> 5;                         // pre-tiles of the image
  const int64_t T = (T32 + 1) >> 1;                            // 64-pixel tiles: workgroup g takes tiles g, g + G, ...
  if (g >= T) {
    float* z = a.slabs + ((size_t)img * a.G + g) * K * D;
    for (int i = tid; i < K * D; i += 512) z[i] = 0.f;
    KM_CLOCK_END
    return;
  }
  // the workgroup's pre-tiles in order: n -> 2 (g + G (n >> 1)) + (n & 1); N of them (the image may end on half a tile)
  const int64_t n_tiles = (T - 1 - g) / a.G + 1;
  const int64_t last_tile = g + (n_tiles - 1) * (int64_t)a.G;
  const int N = (int)(2 * n_tiles - (2 * last_tile + 1 >= T32 ? 1 : 0));
  auto pre_tile_of = [&](int n) -> int64_t { return 2 * (g + (int64_t)a.G * (n >> 1)) + (n & 1); };
  const int64_t tile0 = pre_tile0(seg0, img);
  __syncthreads();                                             // (the zero halves are in place before any copy lands)

  // copy of pre-tile n into ring slot n & 3: NFULL 1-KB blocks per wave + (waves 0..3) one 256-B location block +
  // (wave 4) its 32 labels, 4 bytes per lane straight into the label table (the same LDS-DMA: a register load would
  // make the compiler wait for every copy in flight before the label's first use)
  auto issue = [&](int n) {
    const int64_t p32 = pre_tile_of(n);
    const unsigned char* sb = a.xc + (size_t)(tile0 + p32) * pre_tile_bytes(Q, TAIL);
    unsigned char* dst0 = ring + (size_t)(n & 3) * PTB;
    if (wave == 4 && lane < 32) {
      const int64_t pix = p32 * 32 + lane;
      const int* src = pix < len ? (a.labels_in64 ? reinterpret_cast<const int*>(a.labels_in64 + seg0 + pix)
                                                  : a.labels + seg0 + pix) : g_no_label;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(ring + 4 * PTB + (n & 3) * 128), 4, 0, 0);
    }
#pragma unroll
    for (int b = 0; b < NFULL; ++b)
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)(wave + 8 * b) * 1024 + lane16),
                                       (lptr_t)(dst0 + (wave + 8 * b) * 1024), 16, 0, 0);
    if (TAIL && wave < 4 && lane < 16)
      __builtin_amdgcn_global_load_lds((gptr_t)(sb + (size_t)Q * 4096 + wave * 256 + lane16),
                                       (lptr_t)(dst0 + Q * 4096 + wave * 512), 16, 0, 0);
  };
  // "pre-tile n has landed": at most `younger` (0..2) younger groups of this wave are still in flight
  auto wait_landed = [&](int younger) {
    if (TAIL ? wave <= 4 : wave == 4) {                        // NFULL + 1 operations per pre-tile
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * (NFULL + 1)) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NFULL + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * NFULL) : "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NFULL) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  };
  if (0 < N) issue(0);
  if (1 < N) issue(1);
  if (2 < N) issue(2);

  float4a macc[NDTW][MT], mta[2], mt8[2];
#pragma unroll
  for (int j = 0; j < NDTW; ++j)
#pragma unroll
    for (int q = 0; q < MT; ++q) macc[j][q] = float4a{0.f, 0.f, 0.f, 0.f};
  mta[0] = mta[1] = mt8[0] = mt8[1] = float4a{0.f, 0.f, 0.f, 0.f};

  // M-step transpose read of channel tile wave + 8 j of a pre-tile: + j * 16384, + 512 (pixels 4..7), + 1024 (lo)
  const unsigned m_off = (unsigned)((wave >> 1) * 4096 + (wave & 1) * 256 + (lg >> 1) * 2048 +
                                    frag_slot(8 * (lg & 1) + (lc >> 2), (lc >> 1) & 1) * 16 + 8 * (lc & 1));
  // ... of the location tile: + 64 (pixels 4..7), + 512 (lo)
  const unsigned m_off_t = (unsigned)(Q * 4096 + (lg >> 1) * 1024 +
                                      (((lc >> 1) & 1) * 16 + 8 * (lg & 1) + (lc >> 2)) * 16 + 8 * (lc & 1));
  unsigned cq[MT];
#pragma unroll
  for (int q = 0; q < MT; ++q) cq[q] = 0x10001u * (unsigned)(16 * q + lc);
  unsigned cqw = 0x10001u * (unsigned)(16 * (wave < MT ? wave : MT - 1) + lc);   // the location tile's prototype tile
  unsigned k_one = 0x00010001u, k_h = 0x3C003C00u, k_nh = 0xC400C400u, k_l = 0x10001000u, k_nl = 0xF000F000u;
  asm volatile("" : "+v"(k_one), "+v"(k_h), "+v"(k_nh), "+v"(k_l), "+v"(k_nl), "+v"(cqw));

  union XA { short4v p[2]; half8 h; };
  union OH { half8 h; uint4v u; };
  KM_TRACE_DECL
  KM_MARK(7)
  // one pre-tile in ring slot SL (a literal at every call: the loop is unrolled by four)
  auto step = [&](const int SL, int n) {
    wait_landed(N - 1 - n);
    KM_MARK(0)
    wg_barrier();                               // pre-tile n landed, its labels published; slot (n + 3) & 3 is free
    KM_MARK(1)
    if (n + 3 < N) issue(n + 3);
    const unsigned mb = ring_a + (unsigned)(SL * PTB) + m_off, mbt = ring_a + (unsigned)(SL * PTB) + m_off_t;
    XA xa[2][2];                                               // [buffer][hi|lo]
#define K64_LOADX(u_, b_)                                                                                            \
    if ((u_) < NDTW)                                                                                                 \
      asm volatile("ds_read_b64_tr_b16 %0, %4 offset:%5\n\tds_read_b64_tr_b16 %1, %4 offset:%6\n\t"                  \
                   "ds_read_b64_tr_b16 %2, %4 offset:%7\n\tds_read_b64_tr_b16 %3, %4 offset:%8"                      \
                   : "=&v"(xa[b_][0].p[0]), "=&v"(xa[b_][0].p[1]), "=&v"(xa[b_][1].p[0]), "=&v"(xa[b_][1].p[1])      \
                   : "v"(mb), "i"((u_) * 16384), "i"((u_) * 16384 + 512), "i"((u_) * 16384 + 1024),                  \
                     "i"((u_) * 16384 + 1536));                                                                      \
    else                                                                                                             \
      asm volatile("ds_read_b64_tr_b16 %0, %4\n\tds_read_b64_tr_b16 %1, %4 offset:64\n\t"                            \
                   "ds_read_b64_tr_b16 %2, %4 offset:512\n\tds_read_b64_tr_b16 %3, %4 offset:576"                    \
                   : "=&v"(xa[b_][0].p[0]), "=&v"(xa[b_][0].p[1]), "=&v"(xa[b_][1].p[0]), "=&v"(xa[b_][1].p[1])      \
                   : "v"(mbt));
    uint4v lb32[2], lb;                                        // the 8 labels of this lane's pixel group -> 8 x u16
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:16" : "=&v"(lb32[0]), "=&v"(lb32[1])
                 : "v"(lab_a + (unsigned)(SL * 128 + 32 * lg)));
    K64_LOADX(0, 0)
    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(lb32[0]), "+v"(lb32[1]));
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned lo = i < 2 ? lb32[0][2 * i] : lb32[1][2 * i - 4], hi = i < 2 ? lb32[0][2 * i + 1] : lb32[1][2 * i - 3];
      lb[i] = (lo & 0xffffu) | (hi << 16);
    }
    KM_MARK(2)
    // one-hot B operands: t = |label - (16 q + lc)| clamped to 1 -> 1.0 - t (hi), 2^-11 * (1 - t) (lo)
    OH oh[MT], ol[MT], ohw, olw;
#pragma unroll
    for (int q = 0; q <= MT; ++q)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned d, m;
        asm("v_pk_sub_u16 %0, %1, %2" : "=v"(d) : "v"((unsigned)lb[i]), "v"(q < MT ? cq[q < MT ? q : 0] : cqw));
        asm("v_pk_min_u16 %0, %1, %2" : "=v"(m) : "v"(d), "v"(k_one));
        if (q < MT) {
          asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(oh[q < MT ? q : 0].u[i]) : "v"(m), "v"(k_nh), "v"(k_h));
          asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(ol[q < MT ? q : 0].u[i]) : "v"(m), "v"(k_nl), "v"(k_l));
        } else {
          asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(ohw.u[i]) : "v"(m), "v"(k_nh), "v"(k_h));
          asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(olw.u[i]) : "v"(m), "v"(k_nl), "v"(k_l));
        }
      }
#pragma unroll
    for (int q = 0; q < MT; ++q) asm volatile("" : "+v"(oh[q].u), "+v"(ol[q].u));
    asm volatile("" : "+v"(ohw.u), "+v"(olw.u));
    asm volatile("s_nop 7");
    KM_MARK(5)
    asm volatile("s_setprio 3");                   // (the wave that multiplies goes first, the other one builds one-hots)
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const int b = u & 1;
      if (u + 1 < NU) {
        K64_LOADX(u + 1, b ^ 1)
        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(xa[b][0].p[0]), "+v"(xa[b][0].p[1]), "+v"(xa[b][1].p[0]), "+v"(xa[b][1].p[1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(xa[b][0].p[0]), "+v"(xa[b][0].p[1]), "+v"(xa[b][1].p[0]), "+v"(xa[b][1].p[1]));
      }
      if (u < NDTW) {
#pragma unroll
        for (int q = 0; q < MT; ++q) { if (q == 0) { K64_MFMA(1, macc[u][q], xa[b][0].h, oh[q].h); } else { K64_MFMA(0, macc[u][q], xa[b][0].h, oh[q].h); } }
#pragma unroll
        for (int q = 0; q < MT; ++q) K64_MFMA(0, macc[u][q], xa[b][1].h, ol[q].h);
      } else {
        K64_MFMA(1, mta[0], xa[b][0].h, ohw.h);
        K64_MFMA(0, mt8[0], xa[b][0].h, oh[MT - 1].h);
        K64_MFMA(0, mta[1], xa[b][1].h, olw.h);
        K64_MFMA(0, mt8[1], xa[b][1].h, ol[MT - 1].h);
      }
    }
#undef K64_LOADX
    asm volatile("s_setprio 0");
    KM_MARK(6)
  };
  for (int n = 0; n < N; n += 4) {
    step(0, n);
    if (n + 1 < N) step(1, n + 1);
    if (n + 2 < N) step(2, n + 2);
    if (n + 3 < N) step(3, n + 3);
  }

  // (the sums were last written by hand-written MFMAs: idle slots before they are read)
  asm volatile("s_nop 15\n\ts_nop 15");
#pragma unroll
  for (int j = 0; j < NDTW; ++j)
#pragma unroll
    for (int q = 0; q < MT; ++q) asm volatile("" : "+a"(macc[j][q]));
  asm volatile("" : "+a"(mta[0]), "+a"(mta[1]), "+a"(mt8[0]), "+a"(mt8[1]));
  mta[0] += mta[1];
  mt8[0] += mt8[1];
  float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
#pragma unroll
  for (int j = 0; j < NDTW; ++j) {
    const int dt = wave + 8 * j;
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      const int c = 16 * q + lc, d = 16 * dt + 4 * lg;
      if (c < K) {
        float* dst = slab + (size_t)c * D + d;
        // (rows are 4-byte aligned only, which is all a 16-byte global store needs: four lanes write 64 contiguous
        // bytes of a row.  Plain stores: the 38 MB of slabs of a launch stay in L2 / MALL for kmeans_finalize --
        // non-temporal ones cost 24 us more per accumulate pass, 82 -> 58.5 us)
        *reinterpret_cast<float4a*>(dst) = macc[j][q];
      }
    }
  }
  if (TAIL) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int d = 32 * Q + 4 * lg + r;
      if (wave < MT) {
        const int c = 16 * wave + lc;
        if (c < K && d < D) slab[(size_t)c * D + d] = mta[0][r];
      }
      if (MT == 9 && wave == 0) {
        const int c = 16 * 8 + lc;
        if (c < K && d < D) slab[(size_t)c * D + d] = mt8[0][r];
      }
    }
  }
  KM_MARK(3)
#ifdef SPML_TRACE
  if (a.trace && blockIdx.x == 7 && lane == 0 && wave < 4) {
    for (int i_ = 0; i_ < 8; ++i_) a.trace[wave * 8 + i_] = tc[i_];
    a.trace[32 + wave] = wall_clock64() - treal0;
  }
#endif
  KM_CLOCK_END
}
#undef K64_MFMA

template <int MT, int Q, int TAIL>
int launch64k_accum_t(const PassArgs& a, hipStream_t s) {
  const int lds = 4 * p64_slot_bytes(Q, TAIL) + 4096;
  auto kern = kmeans_accum64k<MT, Q, TAIL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, dim3(a.G, a.n_img), dim3(512), lds, s, a);
  return launch_status();
}

#undef P64_MFMA
#undef P64_MFMA0
#undef P64_NOP_0
#undef P64_NOP_1

}  // namespace

// prototype tiles for K: 5, 6, 8 or 9 (7 runs as 8 with a padding tile)
int assign64k_kpad(int K) {
  const int mt = (K + 15) / 16;
  return mt <= 5 ? 80 : mt <= 6 ? 96 : mt <= 8 ? 128 : 144;
}

bool assign64k_shape(int D, int K) {
  const int q = D / 32, tl = D - 32 * q;
  return K > 64 && K <= 144 && tl <= 8 && (q == 4 || q == 8);
}

int launch_assign64k(const PassArgs& a, hipStream_t s) {
  if (!a.do_assign || a.do_accum || !a.xc || !assign64k_shape(a.D, a.K) || a.kpad != assign64k_kpad(a.K))
    return SPML_ERR_UNSUPPORTED;
  const int q = a.D / 32, tail = (a.D - 32 * q) ? 1 : 0, mt = a.kpad / 16;
#define SPML_K648(M_, Q_) \
  if (mt == M_ && q == Q_) return tail ? launch64k_t<M_, Q_, 1>(a, s) : launch64k_t<M_, Q_, 0>(a, s);
  SPML_K648(5, 4) SPML_K648(5, 8) SPML_K648(6, 4) SPML_K648(6, 8) SPML_K648(8, 4) SPML_K648(8, 8) SPML_K648(9, 4) SPML_K648(9, 8)
#undef SPML_K648
  return SPML_ERR_UNSUPPORTED;
}

// M-step only on the same tiles, labels from a.labels / a.labels_in64
int launch_accum64k(const PassArgs& a, hipStream_t s) {
  if (a.do_assign || !a.do_accum || !a.xc || !assign64k_shape(a.D, a.K) || a.kpad != assign64k_kpad(a.K))
    return SPML_ERR_UNSUPPORTED;
  const int q = a.D / 32, tail = (a.D - 32 * q) ? 1 : 0, mt = a.kpad / 16;
#define SPML_K64A(M_, Q_) \
  if (mt == M_ && q == Q_) return tail ? launch64k_accum_t<M_, Q_, 1>(a, s) : launch64k_accum_t<M_, Q_, 0>(a, s);
  SPML_K64A(5, 4) SPML_K64A(5, 8) SPML_K64A(6, 4) SPML_K64A(6, 8) SPML_K64A(8, 4) SPML_K64A(8, 8) SPML_K64A(9, 4) SPML_K64A(9, 8)
#undef SPML_K64A
  return SPML_ERR_UNSUPPORTED;
}

}  // namespace spml
