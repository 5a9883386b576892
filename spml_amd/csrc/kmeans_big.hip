// Spherical k-means with MANY clusters per image (K > 64 outside the register budget of
// kmeans_pass16k; the stress configuration is 32x32 = 1024 centroids on 258x258x514).
//
// Replaces, for those shapes, the same reference code as kmeans.hip:
//   protos = normalize(scatter_add(X, labels))      # M-step  segsort/common.py:11-41
//   labels = argmax(X @ protos.T, 1)                # E-step  segsort/common.py:44-64
// The reference materialises the [P,K] fp32 similarity (273 MB per 258^2 image at
// K = 1024); here it only ever exists as 32x32 accumulator tiles.
//
// E-step (bigk_assign, MFMA-bound: 2*P*D*K*3 f16 flops):
//   pixel-stationary GEMM.  A wave keeps 32*NPT pixels as split-f16 B fragments of
//   v_mfma_f32_32x32x16_f16 in REGISTERS for the whole kernel (X is read from HBM once,
//   fp32, converted on the fly); the prototypes stream through a 2-slot LDS ring as
//   fragment-major split-f16 blocks (1 KB = one wave-wide ds_read_b128 / one LDS-DMA
//   instruction), shared by the 4 waves of the workgroup.  Per prototype tile of 32 rows
//   and 16 channels: 3 MFMAs (h*h', h*l', l*h').  Prototypes are the accumulator ROWS, so
//   the arg-max over a lane's 16 rows is in-register; a running (best, index) per lane
//   survives the whole prototype loop; ties -> lowest index.
//   Load balance: P / (128*NPT) pixel tiles rarely fill a whole number of rounds of the
//   256 CUs (258^2 / 128 = 520.03).  The tiles of the last partial round are split S ways
//   over the PROTOTYPE range instead, and partial results meet through a 64-bit atomicMax
//   on (orderable score << 32 | ~index) -- max is associative, so the result is
//   deterministic.
//
// M-step (HBM-bound: X read once, as coalesced 2-KB row gathers):
//   counting sort of the pixels by (image, label) + a gather-sum over the sorted order.
//   Sums are accumulated in 2^-36 fixed point (two int32 partial accumulators per
//   channel, flushed to int64 with atomics): integer addition is associative, so neither
//   the order inside a cluster nor the atomics make the result run-to-run different, and
//   the sum is exact to 2^-36 per element (better than the reference's fp32 scatter_add).
//   Needs |x_d| <= 1 (unit rows, which is what the reference clusters) -- see DESIGN.md.
#include <stdio.h>

#include "common.hpp"

namespace spml {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr float kFix1 = 4096.0f;            // 2^12: integer part of the fixed-point split
constexpr float kFix2 = 16777216.0f;        // 2^24: fractional part
constexpr double kFixInv = 1.0 / 68719476736.0;   // 2^-36

struct BigArgs {
  const float* x;
  int64_t P;
  int D, K, n_img;
  const int64_t* seg_off;        // device [n_img+1]
  const unsigned char* afrag;    // [n_img][MT][NK16][hi|lo] x 1 KB
  unsigned long long* keys;      // [P] (orderable score << 32 | ~index), 0 = empty
  int MT;                        // prototype tiles of 32 rows
  int tiles_per_img;             // pixel tiles (of 128*NPT rows) per image, from max_seg_len
  int n_full;                    // work items [0, n_full) walk all prototype tiles
  int S;                         // the remaining ones are split S ways over the tiles
  // screening (bigk_screen / bigk_assign_list)
  int32_t* amb_list;             // [P]: image b's ambiguous pixels at [seg_off[b], seg_off[b] + amb_count[b])
  int* amb_count;                // [n_img]
  const float* cmax;             // [n_img] largest prototype norm of the image
  float screen_eps;              // 2 * 2^-10 (+ rounding slack)
};


__device__ __forceinline__ unsigned orderable(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ------------------------------------------------------------------------------------
// E-step
// ------------------------------------------------------------------------------------
__device__ __forceinline__ int image_of(const int64_t* seg_off, int n_img, int64_t p) {
  int lo = 0, hi = n_img;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= p) lo = mid; else hi = mid; }
  return lo;
}

// Appends the flagged pixels of this workgroup to the image's ambiguous list with ONE global
// atomic per workgroup (per-wave ballots, counts combined in LDS): thousands of same-address
// returning atomics at the end of a kernel would serialise.
template <int NPT>
__device__ __forceinline__ void append_ambiguous(const BigArgs& a, unsigned char* lds, int img,
                                                 const int64_t (&prow)[NPT], const bool (&amb)[NPT]) {
  int* wcount = reinterpret_cast<int*>(lds);                  // [4 * NPT] + base (ring no longer in use)
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  wg_barrier();                                               // every wave is done with the ring
  unsigned long long mask[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    mask[p] = __ballot(amb[p]);
    if (lane == 0) wcount[wave * NPT + p] = __popcll(mask[p]);
  }
  wg_barrier();
  if (threadIdx.x == 0) {
    int tot = 0;
    for (int i = 0; i < 4 * NPT; ++i) { const int c = wcount[i]; wcount[i] = tot; tot += c; }
    wcount[4 * NPT] = tot ? atomicAdd(a.amb_count + img, tot) : 0;
  }
  wg_barrier();
  const int base = wcount[4 * NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p)
    if (amb[p]) {
      const int pos = base + wcount[wave * NPT + p] + __popcll(mask[p] & ((1ull << lane) - 1ull));
      a.amb_list[a.seg_off[img] + pos] = (int32_t)prow[p];
    }
}

// One work item of the E-step: the 128*NPT pixels `prow` (rows of x; all of image `img`)
// against the prototype tiles [m0, m1).
//   HI = false: exact split-f16 scores (3 MFMA per k-step), result -> keys (atomicMax when the
//               prototype range is split over several workgroups);
//   HI = true : screening on the hi halves only (1 MFMA per k-step, half the pixel registers,
//               half the prototype bytes): tracks the best AND the second-best score; a pixel
//               whose margin exceeds the error bound of the hi-only score is final, the others
//               are appended to the image's ambiguous list for the exact kernel.
template <int NK16, int NPT, bool HI>
__device__ __forceinline__ void assign_item(const BigArgs& a, unsigned char* lds, int img,
                                            const int64_t (&prow)[NPT], const bool (&valid)[NPT],
                                            int m0, int m1, bool split) {
  constexpr int BLK = HI ? 1 : 2;                // 1-KB blocks per k-step in the ring
  constexpr int SLOT = NK16 * BLK * 1024;
  constexpr int SRC_TILE = NK16 * 2048;          // a tile in HBM: [k-step][hi|lo] x 1 KB
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int D = a.D, K = a.K;
  const unsigned char* afrag = a.afrag + (size_t)img * a.MT * SRC_TILE;

  // one 1-KB block of prototype tile `mt` into ring slot `slot` (blocks wave, wave+4, ...)
  auto issue_block = [&](int mt, int slot, int i) {
    const int blk = wave + 4 * i;
    if (blk < BLK * NK16) {                      // wave-uniform
      const size_t so = HI ? (size_t)blk * 2048 : (size_t)blk * 1024;   // HI: skip the lo blocks
      const unsigned char* src = afrag + (size_t)mt * SRC_TILE + so + 16 * lane;
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + slot * SLOT + blk * 1024), 16, 0, 0);
    }
  };
  constexpr int NISSUE = (BLK * NK16 + 3) / 4;   // blocks per wave and tile (at most)
#pragma unroll
  for (int i = 0; i < NISSUE; ++i) issue_block(m0, 0, i);

  // ---- this wave's pixels -> split-f16 B fragments, register resident ----
  //   B[k = 8*half + e][col = j] of k-step s  =  x[pixel j][16*s + 8*half + e]
  half8 bh[NPT][NK16], bl[HI ? 1 : NPT][HI ? 1 : NK16];
  float xnorm2[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const float* row = a.x + prow[p] * D;
    float ssq = 0.f;
    // branch-free: every load is clamped into the row and masked afterwards, so that the
    // loads of all k-steps can be in flight together (one HBM round trip, not NK16)
    if (!(D & 1)) {                                // (wave-uniform) rows are 8-byte aligned
#pragma unroll
      for (int s = 0; s < NK16; ++s) {
        const int c0 = 16 * s + 8 * half;
        float v[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = c0 + 2 * e;
          const float2 f = *reinterpret_cast<const float2*>(row + min(c, D - 2));
          v[2 * e] = c < D ? f.x : 0.f;
          v[2 * e + 1] = c < D ? f.y : 0.f;
        }
        if constexpr (HI) {
          half8 lo_unused;
          split8(v, bh[p][s], lo_unused);
#pragma unroll
          for (int e = 0; e < 8; ++e) ssq += v[e] * v[e];
        } else {
          split8(v, bh[p][s], bl[p][s]);
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < NK16; ++s) {
        const int c0 = 16 * s + 8 * half;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const float f = row[min(c0 + e, D - 1)];
          v[e] = c0 + e < D ? f : 0.f;
        }
        if constexpr (HI) {
          half8 lo_unused;
          split8(v, bh[p][s], lo_unused);
#pragma unroll
          for (int e = 0; e < 8; ++e) ssq += v[e] * v[e];
        } else {
          split8(v, bh[p][s], bl[p][s]);
        }
      }
    }
    xnorm2[p] = HI ? ssq + __shfl_xor(ssq, 32, 64) : 0.f;     // the two halves hold disjoint channels
  }

  float best[NPT], second[NPT];
  int best_i[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) { best[p] = -INFINITY; second[p] = -INFINITY; best_i[p] = 0x7fffffff; }

  // Per tile: the MFMAs fill one accumulator set; at the end the 16 scores per pixel are
  // combined into `sc` (which frees the accumulators), and the running arg-max over them is
  // folded in BETWEEN the MFMAs of the next tile, a few rows per k-step; the DMA of the tile
  // after that is issued one block per k-step as well -- neither sits in front of the
  // matrix pipe.
  float16v zero;
#pragma unroll
  for (int r = 0; r < 16; ++r) zero[r] = 0.f;
  float sc[NPT][16];

  // running arg-max over rows [r0, r1) of the finished tile ft (ascending rows: ties -> lowest)
  auto fold = [&](int ft, int r0, int r1) {
    const bool inside = 32 * (ft + 1) <= K;        // (uniform) no padding rows in this tile
#pragma unroll
    for (int p = 0; p < NPT; ++p)
#pragma unroll
      for (int r = r0; r < r1; ++r) {
        const int c = 32 * ft + (r & 3) + 8 * (r >> 2) + 4 * half;
        const float v = (inside || c < K) ? sc[p][r] : -INFINITY;
        // second <= best always: the median of (best, v, second) is the new runner-up
        if constexpr (HI) second[p] = __builtin_amdgcn_fmed3f(best[p], v, second[p]);
        if (v > best[p]) { best[p] = v; best_i[p] = c; }
      }
  };

  for (int mt = m0; mt < m1; ++mt) {
    const int slot = (mt - m0) & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                     // tile mt has landed for every wave; the other slot is free
    const bool more = mt + 1 < m1;
    const bool has_prev = mt > m0;
    float16v acc_h[NPT], acc_x[HI ? 1 : NPT], acc_y[HI ? 1 : NPT];
    // A operands (prototype rows): hand-issued reads two k-steps ahead, counted waits
    // (LDS returns in order: "at most N outstanding" == "the older ones have landed")
    half8 ah[2], al[2];
    const unsigned cbase = (unsigned)(size_t)(lptr_t)(lds + slot * SLOT) + 16u * lane;
    auto load_a = [&](int s, int bsel) {
      const unsigned addr = cbase + (unsigned)s * (BLK * 1024u);
      if constexpr (HI)
        asm volatile("ds_read_b128 %0, %1" : "=&v"(ah[bsel]) : "v"(addr));
      else
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                     : "=&v"(ah[bsel]), "=&v"(al[bsel]) : "v"(addr));
    };
    load_a(0, 0);
    if (NK16 > 1) load_a(1, 1);
#pragma unroll
    for (int s = 0; s < NK16; ++s) {
      const int bsel = s & 1;
      if constexpr (HI) {
        if (s + 1 < NK16) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[bsel]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[bsel]));
      } else {
        if (s + 1 < NK16) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[bsel]), "+v"(al[bsel]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[bsel]), "+v"(al[bsel]));
      }
#pragma unroll
      for (int p = 0; p < NPT; ++p) {
        acc_h[p] = mfma32(ah[bsel], bh[p][s], s == 0 ? zero : acc_h[p]);
        if constexpr (!HI) {
          acc_x[p] = mfma32(ah[bsel], bl[p][s], s == 0 ? zero : acc_x[p]);
          acc_y[p] = mfma32(al[bsel], bh[p][s], s == 0 ? zero : acc_y[p]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 2 < NK16) load_a(s + 2, bsel);
#ifndef SPML_EXP_NODMA
      if (more && s < NISSUE) issue_block(mt + 1, slot ^ 1, s);
#endif
#ifndef SPML_EXP_NOFOLD
      if (has_prev) fold(mt - 1, (16 * s) / NK16, (16 * (s + 1)) / NK16);
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (NISSUE > NK16 && more) {
#pragma unroll
      for (int i = NK16; i < NISSUE; ++i) issue_block(mt + 1, slot ^ 1, i);
    }
#pragma unroll
    for (int p = 0; p < NPT; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (HI) sc[p][r] = acc_h[p][r];
        else sc[p][r] = acc_h[p][r] + (acc_x[p][r] + acc_y[p][r]) * kSplitInv;
      }
  }
  fold(m1 - 1, 0, 16);

  // the two lane halves hold different prototype rows of the same pixel
  bool amb[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const float ob = __shfl_xor(best[p], 32, 64);
    const int oi = __shfl_xor(best_i[p], 32, 64);
    if constexpr (HI) {
      const float os = __shfl_xor(second[p], 32, 64);
      second[p] = fmaxf(fmaxf(second[p], os), fminf(best[p], ob));
    }
    if (ob > best[p] || (ob == best[p] && oi < best_i[p])) { best[p] = ob; best_i[p] = oi; }
    amb[p] = false;
    if (half == 0 && valid[p]) {
      const unsigned idx = best_i[p] == 0x7fffffff ? 0u : (unsigned)best_i[p];
      const unsigned long long key = ((unsigned long long)orderable(best[p]) << 32) | (0xffffffffu - idx);
      if constexpr (HI) {
        // |exact score - hi-only score| <= 2^-10 |x| |c| per prototype (f16 rounding of both
        // operands, Cauchy-Schwarz): the arg-max is decided when the margin exceeds twice that
        const float bound = a.screen_eps * sqrtf(xnorm2[p]) * a.cmax[img];
        if (best[p] - second[p] > bound) a.keys[prow[p]] = key;
        else amb[p] = true;
      } else {
        if (split) atomicMax(a.keys + prow[p], key);
        else a.keys[prow[p]] = key;
      }
    }
  }
  if constexpr (HI) append_ambiguous<NPT>(a, lds, img, prow, amb);
}

// Exact E-step over all pixels: work item b -> (image, pixel tile, prototype tile range).
template <int NK16, int NPT>
__global__ __launch_bounds__(256, 1) void bigk_assign(BigArgs a) {
  constexpr int TPX = 128 * NPT;                // pixels per workgroup
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;
  int work, m0, m1;
  bool split;
  {
    const int b = blockIdx.x;
    if (b < a.n_full) { work = b; m0 = 0; m1 = a.MT; split = false; }
    else {
      const int r = b - a.n_full;
      work = a.n_full + r / a.S;
      const int part = r % a.S;
      m0 = (a.MT * part) / a.S;
      m1 = (a.MT * (part + 1)) / a.S;
      split = a.S > 1;
    }
  }
  const int img = work / a.tiles_per_img;
  const int t = work - img * a.tiles_per_img;
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  if ((int64_t)t * TPX >= len || m0 >= m1) return;
  bool valid[NPT];
  int64_t prow[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int64_t r = (int64_t)t * TPX + (wave * NPT + p) * 32 + j;
    valid[p] = r < len;
    prow[p] = seg0 + (valid[p] ? r : len - 1);
  }
  assign_item<NK16, NPT, false>(a, lds, img, prow, valid, m0, m1, split);
}

// Screening E-step (hi halves only) over the pixel tiles of whole rounds of the CUs; the
// pixels of the last partial round are handed to the exact kernel as they are (their
// workgroups only append them to the ambiguous lists).
template <int NK16, int NPT>
__global__ __launch_bounds__(256, 1) void bigk_screen(BigArgs a) {
  constexpr int TPX = 128 * NPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31, half = lane >> 5;
  const int work = blockIdx.x;
  const int img = work / a.tiles_per_img;
  const int t = work - img * a.tiles_per_img;
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  if ((int64_t)t * TPX >= len) return;
  bool valid[NPT];
  int64_t prow[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int64_t r = (int64_t)t * TPX + (wave * NPT + p) * 32 + j;
    valid[p] = r < len;
    prow[p] = seg0 + (valid[p] ? r : len - 1);
  }
  if (work >= a.n_full) {                        // last partial round: straight to the exact kernel
    bool amb[NPT];
#pragma unroll
    for (int p = 0; p < NPT; ++p) amb[p] = half == 0 && valid[p];
    append_ambiguous<NPT>(a, lds, img, prow, amb);
    return;
  }
  assign_item<NK16, NPT, true>(a, lds, img, prow, valid, 0, a.MT, false);
}

// Exact E-step over the ambiguous lists: the work (tiles of 128*NPT listed pixels of one image
// x S prototype ranges) is laid out on the device from the list lengths; the grid is sized
// for the worst case and surplus workgroups leave at once.
template <int NK16, int NPT>
__global__ __launch_bounds__(256, 1) void bigk_assign_list(BigArgs a) {
  constexpr int TPX = 128 * NPT;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 31;
  int ntiles = 0;
  for (int i = 0; i < a.n_img; ++i) ntiles += (a.amb_count[i] + TPX - 1) / TPX;
  if (ntiles == 0) return;
  int S = 1;
  while (2 * S * ntiles <= 256 && 2 * S <= a.MT) S *= 2;
  const int w = blockIdx.x;                       // one work item per workgroup; the grid covers
  if (w >= ntiles * S) return;                    // the worst case (every pixel ambiguous)
  int tile = w / S;
  const int part = w - tile * S;
  int img = 0, n_amb = a.amb_count[0];
  while (tile >= (n_amb + TPX - 1) / TPX) {       // (workgroup-uniform; n_img is small)
    tile -= (n_amb + TPX - 1) / TPX;
    ++img;
    n_amb = a.amb_count[img];
  }
  const int64_t seg0 = a.seg_off[img];
  bool valid[NPT];
  int64_t prow[NPT];
#pragma unroll
  for (int p = 0; p < NPT; ++p) {
    const int r = tile * TPX + (wave * NPT + p) * 32 + j;
    valid[p] = r < n_amb;
    prow[p] = a.amb_list[seg0 + (valid[p] ? r : n_amb - 1)];
  }
  const int m0 = (a.MT * part) / S, m1 = (a.MT * (part + 1)) / S;
  if (m0 < m1) assign_item<NK16, NPT, false>(a, lds, img, prow, valid, m0, m1, S > 1);
}

// ------------------------------------------------------------------------------------
// M-step: counting sort by (image, label) + gather-sum in fixed point
// ------------------------------------------------------------------------------------
// keys -> lab32 (keys reset to 0 for the next E-step); counts[img*K + label] += 1.
// A block of 1024 consecutive pixels that lies inside ONE image first builds its histogram
// in LDS and then issues one global atomic per label it saw: with few clusters (P/K
// pixels per counter) the per-pixel global atomics would serialise on K addresses.
constexpr int kSortBlock = 1024;
constexpr int kSortLdsK = 4096;          // LDS path up to this many clusters per image

__global__ __launch_bounds__(kSortBlock) void bigk_hist(unsigned long long* __restrict__ keys,
                                                        int32_t* __restrict__ lab32, int64_t P,
                                                        const int64_t* __restrict__ seg_off, int n_img,
                                                        int K, int* __restrict__ counts) {
  extern __shared__ int sh[];
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * kSortBlock;
  const int64_t p = p0 + tid;
  int l = -1;
  if (p < P) {
    if (keys) {
      l = (int)(0xffffffffu - (unsigned)(keys[p] & 0xffffffffull));
      keys[p] = 0ull;
      lab32[p] = l;
    } else {
      l = lab32[p];
    }
  }
  if (!counts) return;
  const bool ok = p < P && l >= 0 && l < K;
  const int img0 = image_of(seg_off, n_img, p0);
  const int img1 = image_of(seg_off, n_img, min(p0 + kSortBlock, P) - 1);
  if (img0 == img1 && K <= kSortLdsK) {             // block-uniform
    for (int k = tid; k < K; k += kSortBlock) sh[k] = 0;
    __syncthreads();
    if (ok) atomicAdd(sh + l, 1);
    __syncthreads();
    for (int k = tid; k < K; k += kSortBlock)
      if (sh[k]) atomicAdd(counts + (size_t)img0 * K + k, sh[k]);
  } else if (ok) {
    atomicAdd(counts + (size_t)image_of(seg_off, n_img, p) * K + l, 1);
  }
}

// per image: start[k] = seg0 + exclusive prefix of counts; cursor = start; counts -> 0
__global__ __launch_bounds__(1024) void bigk_scan(int* __restrict__ counts, int K,
                                                  const int64_t* __restrict__ seg_off,
                                                  int* __restrict__ start, int* __restrict__ cursor) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (tid == 0) carry = (int)seg_off[img];
  __syncthreads();
  for (int k0 = 0; k0 < K; k0 += 1024) {
    const int k = k0 + tid;
    const int c = k < K ? counts[(size_t)img * K + k] : 0;
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o, 64); if (lane >= o) incl += v; }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = carry;
    for (int i = 0; i < w; ++i) base += wsum[i];
    if (k < K) {
      start[(size_t)img * K + k] = base + incl - c;
      cursor[(size_t)img * K + k] = base + incl - c;
      counts[(size_t)img * K + k] = 0;
    }
    __syncthreads();
    if (tid == 1023) carry = base + incl;
    __syncthreads();
  }
}

__global__ __launch_bounds__(kSortBlock) void bigk_scatter(const int32_t* __restrict__ lab32, int64_t P,
                                                           const int64_t* __restrict__ seg_off,
                                                           int n_img, int K, int* __restrict__ cursor,
                                                           int32_t* __restrict__ order,
                                                           int32_t* __restrict__ order_gid) {
  // same blocking as bigk_hist: ranks inside the block from LDS atomics, one global
  // (returning) atomic per label of the block reserves its range of sorted positions
  extern __shared__ int sh[];
  const int tid = threadIdx.x;
  const int64_t p0 = (int64_t)blockIdx.x * kSortBlock;
  const int64_t p = p0 + tid;
  const int l = p < P ? lab32[p] : -1;
  const bool ok = p < P && l >= 0 && l < K;
  const int img0 = image_of(seg_off, n_img, p0);
  const int img1 = image_of(seg_off, n_img, min(p0 + kSortBlock, P) - 1);
  if (img0 == img1 && K <= kSortLdsK) {             // block-uniform
    int* cnt = sh;
    int* base = sh + K;
    for (int k = tid; k < K; k += kSortBlock) cnt[k] = 0;
    __syncthreads();
    int rank = 0;
    if (ok) rank = atomicAdd(cnt + l, 1);
    __syncthreads();
    for (int k = tid; k < K; k += kSortBlock)
      if (cnt[k]) base[k] = atomicAdd(cursor + (size_t)img0 * K + k, cnt[k]);
    __syncthreads();
    if (ok) {
      const int pos = base[l] + rank;
      order[pos] = (int32_t)p;
      order_gid[pos] = img0 * K + l;
    }
  } else if (ok) {
    const int gid = image_of(seg_off, n_img, p) * K + l;
    const int pos = atomicAdd(cursor + gid, 1);
    order[pos] = (int32_t)p;
    order_gid[pos] = gid;
  }
}

// Gather-sum over the sorted order.  A workgroup takes kGatherRows consecutive sorted
// positions; thread t owns channels (2t, 2t+1) (blockDim = 64 * ceil(D / 128)), so every row
// is one coalesced D*4-byte read.  Rows are fetched kGatherBatch at a time, the next batch in
// flight while the current one is accumulated per channel as (int32 hi, int32 lo) 2^-36
// fixed point; a change of cluster (positions are sorted by cluster) flushes the pair with
// one 64-bit atomic per channel.
constexpr int kGatherRows = 64;
constexpr int kGatherBatch = 8;

__global__ __launch_bounds__(1024) void bigk_gather_sum(const float* __restrict__ x, int64_t P, int D,
                                                        const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ order_gid,
                                                        long long* __restrict__ sums64) {
  __shared__ int s_row[kGatherRows + kGatherBatch], s_gid[kGatherRows + kGatherBatch];
  const int tid = threadIdx.x;
  const int64_t i0 = (int64_t)blockIdx.x * kGatherRows;
  const int n = (int)min((int64_t)kGatherRows, P - i0);
  for (int q = tid; q < kGatherRows + kGatherBatch; q += blockDim.x) {
    // positions never filled (labels outside [0,K)) keep the -1 written by the host memset
    const int r = q < n ? order[i0 + q] : -1;
    s_row[q] = r;
    s_gid[q] = r >= 0 ? order_gid[i0 + q] : -1;
  }
  __syncthreads();
  const int d = 2 * tid;
  const bool act = d < D, pair = d + 1 < D, even = !(D & 1);
  auto fetch = [&](int i, float2 (&v)[kGatherBatch]) {
#pragma unroll
    for (int u = 0; u < kGatherBatch; ++u) {
      const int r = s_row[i + u];                   // (i + u < kGatherRows + kGatherBatch)
      v[u] = float2{0.f, 0.f};
      if (r >= 0 && act) {
        const float* src = x + (size_t)r * D + d;
        if (even) v[u] = *reinterpret_cast<const float2*>(src);
        else { v[u].x = src[0]; if (pair) v[u].y = src[1]; }
      }
    }
  };
  int ahi[2] = {0, 0}, alo[2] = {0, 0};
  auto flush = [&](int gid) {
    if (gid >= 0 && act) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const long long v = (long long)ahi[e] * 16777216ll + (long long)alo[e];
        if ((e == 0 || pair) && v != 0)
          atomicAdd(reinterpret_cast<unsigned long long*>(sums64) + (size_t)gid * D + d + e,
                    (unsigned long long)v);
      }
    }
    ahi[0] = ahi[1] = alo[0] = alo[1] = 0;
  };
  float2 cur[kGatherBatch], nxt[kGatherBatch];
  fetch(0, cur);
  int gcur = s_gid[0];
  for (int i = 0; i < n; i += kGatherBatch) {
    if (i + kGatherBatch < n) fetch(i + kGatherBatch, nxt);
#pragma unroll
    for (int u = 0; u < kGatherBatch; ++u) {
      if (i + u < n) {                       // workgroup-uniform
        const int g = s_gid[i + u];
        if (g != gcur) { flush(gcur); gcur = g; }
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const float f = (e ? cur[u].y : cur[u].x) * kFix1;
          const int hi = (int)f;                           // trunc toward zero
          const int lo = (int)((f - (float)hi) * kFix2);   // exact remainder, |.| < 2^24
          ahi[e] += hi;
          alo[e] += lo;
        }
      }
    }
#pragma unroll
    for (int u = 0; u < kGatherBatch; ++u) cur[u] = nxt[u];
  }
  flush(gcur);
}

// int64 sums (or given fp32 prototypes) -> L2-normalised fp32 prototypes (empty cluster ->
// zero row, as the reference) + the split-f16 fragment blocks the E-step streams, in two
// small kernels: (1) one wave per prototype row: 1 / max(norm, eps); (2) one thread per
// (row, 8-channel group): scale, write fp32 + fragments, zero the sums for the next M-step.
__global__ __launch_bounds__(256) void bigk_norms(const long long* __restrict__ sums64, int64_t rows,
                                                  int D, float* __restrict__ inv) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  const long long* s = sums64 + (size_t)row * D;
  float ssq = 0.f;
  for (int d = lane; d < D; d += 64) { const float v = (float)((double)s[d] * kFixInv); ssq += v * v; }
  ssq = wave_sum(ssq);
  if (lane == 0) { const float nrm = sqrtf(ssq); inv[row] = 1.f / (nrm >= kEps ? nrm : kEps); }
}

__global__ __launch_bounds__(256) void bigk_emit(long long* __restrict__ sums64,
                                                 const float* __restrict__ given,
                                                 const float* __restrict__ inv, int K, int D, int NK16,
                                                 int MT, int n_img, float* __restrict__ cent,
                                                 unsigned char* __restrict__ afrag) {
  // item = (image, prototype tile, 8-channel group, row of the tile): consecutive threads
  // take consecutive rows, i.e. consecutive 16-B slots of a fragment block
  const int64_t it = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_img = (int64_t)MT * 2 * NK16 * 32;
  if (it >= per_img * n_img) return;
  const int img = (int)(it / per_img);
  int rem = (int)(it - img * per_img);
  const int r = rem & 31; rem >>= 5;
  const int grp = rem % (2 * NK16), mt = rem / (2 * NK16);
  const int s16 = grp >> 1, g = grp & 1;
  const int kk = 32 * mt + r, d0 = 16 * s16 + 8 * g;
  const float scale = (given || kk >= K) ? 1.f : inv[(size_t)img * K + kk];
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float f = 0.f;
    if (kk < K && d0 + e < D) {
      const size_t o = ((size_t)img * K + kk) * D + d0 + e;
      if (given) f = given[o];
      else { f = (float)((double)sums64[o] * kFixInv) * scale; sums64[o] = 0; }
      if (cent) cent[o] = f;
    }
    v[e] = f;
  }
  half8 h, l;
  split8(v, h, l);
  unsigned char* dst = afrag + ((size_t)img * MT + mt) * NK16 * 2048 + (size_t)s16 * 2048 +
                       (size_t)(g * 32 + r) * 16;
  *reinterpret_cast<half8*>(dst) = h;
  *reinterpret_cast<half8*>(dst + 1024) = l;
}

// fixed-point sums -> fp32 (un-normalised: the fused-pass entry point returns raw sums)
__global__ __launch_bounds__(256) void bigk_sums_to_f32(long long* __restrict__ sums64, int64_t n,
                                                        float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (float)((double)sums64[i] * kFixInv);
  sums64[i] = 0;
}

struct BigWs {
  size_t keys, order, order_gid, counts, start, cursor, sums64, afrag, inv, amb_list, amb_count, cmax, total;
};

inline int bigk_nk16(int D) {
  const int n = (D + 15) / 16;
  return n <= 5 ? 5 : n <= 9 ? 9 : n <= 17 ? 17 : n <= 33 ? 33 : 0;
}
inline int bigk_npt(int nk16) { return nk16 <= 17 ? 2 : 1; }

BigWs bigk_ws(int64_t P, int D, int K, int n_img) {
  BigWs w{};
  size_t o = 0;
  const int MT = (K + 31) / 32, nk = bigk_nk16(D);
  w.keys = o; o = align_up(o + (size_t)P * 8, 256);
  w.order = o; o = align_up(o + (size_t)P * 4, 256);
  w.order_gid = o; o = align_up(o + (size_t)P * 4, 256);
  w.counts = o; o = align_up(o + (size_t)n_img * K * 4, 256);
  w.start = o; o = align_up(o + (size_t)n_img * K * 4, 256);
  w.cursor = o; o = align_up(o + (size_t)n_img * K * 4, 256);
  w.sums64 = o; o = align_up(o + (size_t)n_img * K * D * 8, 256);
  w.afrag = o; o = align_up(o + (size_t)n_img * MT * nk * 2048, 256);
  w.inv = o; o = align_up(o + (size_t)n_img * K * 4, 256);
  w.amb_list = o; o = align_up(o + (size_t)P * 4, 256);
  w.amb_count = o; o = align_up(o + (size_t)n_img * 4, 256);
  w.cmax = o; o = align_up(o + (size_t)n_img * 4, 256);
  w.total = o;
  return w;
}

// largest row norm of the given centroids of every image (screening bound); the prototypes of
// a k-means run are unit (or zero) rows: the bound is a constant there
__global__ __launch_bounds__(256) void bigk_cmax(const float* __restrict__ cent, int K, int D,
                                                 float* __restrict__ cmax, float constant) {
  __shared__ float red[4];
  const int img = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  float m = 0.f;
  if (cent) {
    for (int k = wv; k < K; k += 4) {
      const float* r = cent + ((size_t)img * K + k) * D;
      float ssq = 0.f;
      for (int d = lane; d < D; d += 64) ssq += r[d] * r[d];
      m = fmaxf(m, sqrtf(wave_sum(ssq)));
    }
  } else {
    m = constant;
  }
  if (lane == 0) red[wv] = m;
  __syncthreads();
  if (threadIdx.x == 0) cmax[img] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])) * 1.000001f;
}

template <int NK16, int NPT, int NPTS>
int launch_screened_t(const BigArgs& a, const BigArgs& as, int grid_screen, int grid_list, hipStream_t s) {
  auto ks = bigk_screen<NK16, NPTS>;
  auto kl = bigk_assign_list<NK16, NPT>;
  const int lds_s = 2 * NK16 * 1024, lds_l = 2 * NK16 * 2048;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(ks), hipFuncAttributeMaxDynamicSharedMemorySize, lds_s);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kl), hipFuncAttributeMaxDynamicSharedMemorySize, lds_l);
  hipLaunchKernelGGL(ks, dim3(grid_screen), dim3(256), lds_s, s, as);
  hipLaunchKernelGGL(kl, dim3(grid_list), dim3(256), lds_l, s, a);
  return launch_status();
}

template <int NK16, int NPT>
int launch_assign_t(const BigArgs& a, int grid, hipStream_t s) {
  auto kern = bigk_assign<NK16, NPT>;
  const int lds = 2 * NK16 * 2048;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a);
  return launch_status();
}

}  // namespace

// shapes the many-cluster kernels cover
bool bigk_shape(int64_t P, int D, int K, int n_img) {
  return K > 64 && K <= 65536 && bigk_nk16(D) != 0 && P < (1ll << 31) &&
         (int64_t)n_img * K < (1ll << 31);
}

size_t bigk_workspace_bytes(int64_t P, int D, int K, int n_img) {
  return bigk_ws(P, D, K, n_img).total;
}

// One k-means run (labels_init given) or one E-step (given_centroids) on the many-cluster
// kernels.  lab32 [P] is the caller's int32 label buffer (already holds labels_init for a
// run); cent_f [n_img,K,D] receives the prototypes of the last M-step.  sums_out (only with
// given_centroids): also the raw sums [n_img,K,D] of X by the new labels (fused pass).
int bigk_run(const float* x, int64_t P, int D, const int64_t* seg_off, int n_img,
             int64_t max_seg_len, int K, const float* given_centroids, int iterations,
             int32_t* lab32, float* cent_f, float* sums_out, int flags, void* ws, hipStream_t s) {
  const BigWs wl = bigk_ws(P, D, K, n_img);
  unsigned char* base = static_cast<unsigned char*>(ws);
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(base + wl.keys);
  int32_t* order = reinterpret_cast<int32_t*>(base + wl.order);
  int32_t* order_gid = reinterpret_cast<int32_t*>(base + wl.order_gid);
  int* counts = reinterpret_cast<int*>(base + wl.counts);
  int* start = reinterpret_cast<int*>(base + wl.start);
  int* cursor = reinterpret_cast<int*>(base + wl.cursor);
  long long* sums64 = reinterpret_cast<long long*>(base + wl.sums64);
  unsigned char* afrag = base + wl.afrag;
  float* inv = reinterpret_cast<float*>(base + wl.inv);
  int32_t* amb_list = reinterpret_cast<int32_t*>(base + wl.amb_list);
  int* amb_count = reinterpret_cast<int*>(base + wl.amb_count);
  float* cmax = reinterpret_cast<float*>(base + wl.cmax);

  const int nk16 = bigk_nk16(D), npt = bigk_npt(nk16);
  const int MT = (K + 31) / 32;
  const int tpx = 128 * npt;
  const unsigned pblocks = (unsigned)((P + kSortBlock - 1) / kSortBlock);
  const size_t sort_lds = K <= kSortLdsK ? (size_t)2 * K * 4 : 0;

  BigArgs a{};
  a.x = x; a.P = P; a.D = D; a.K = K; a.n_img = n_img; a.seg_off = seg_off;
  a.afrag = afrag; a.keys = keys; a.MT = MT;
  a.tiles_per_img = (int)((max_seg_len + tpx - 1) / tpx);
  const int64_t nt = (int64_t)n_img * a.tiles_per_img;
  constexpr int kCUs = 256;
  a.n_full = (int)(nt / kCUs) * kCUs;
  const int rem = (int)(nt - a.n_full);
  int S = 1;
  if (rem > 0) { while (2 * S * rem <= kCUs && 2 * S <= MT) S *= 2; }
  a.S = S;
  const int grid = a.n_full + rem * S;

  if (hipMemsetAsync(keys, 0, (size_t)P * 8, s) != hipSuccess) return SPML_ERR_LAUNCH;
  // Screening (hi halves only, 1/3 of the MFMAs, 1/4 of the prototype bytes per pixel) decides
  // every pixel whose top-2 margin exceeds the error bound of the hi-only score; the exact
  // kernel then runs over the ambiguous pixels only.  Worth it from a few hundred clusters on.
  const bool screen = !(flags & SPML_KMEANS_NO_SCREEN) && K >= 256;
  const int npts = nk16 == 33 ? 2 : 4;            // 32-pixel tiles per wave of the screening kernel
  BigArgs as = a;
  int grid_screen = 0;
  // exact kernel over the lists: at most ceil(P / tile) + n_img tiles, or 256 items when split
  const int64_t list_tiles = (P + tpx - 1) / tpx + n_img;
  const int grid_list = (int)(list_tiles > kCUs ? list_tiles : kCUs);
  if (screen) {
    const int tpx_s = 128 * npts;
    as.tiles_per_img = (int)((max_seg_len + tpx_s - 1) / tpx_s);
    const int64_t nts = (int64_t)n_img * as.tiles_per_img;
    const int rem_s = (int)(nts % kCUs);
    as.n_full = (nts < kCUs || rem_s > 64) ? (int)nts : (int)(nts - rem_s);
    grid_screen = (int)nts;
    as.amb_list = amb_list; as.amb_count = amb_count; as.cmax = cmax;
    as.screen_eps = 2.05e-3f;                     // 2 * 2^-10 (+ slack for the fp32 accumulation)
    a.amb_list = amb_list; a.amb_count = amb_count;
    hipLaunchKernelGGL(bigk_cmax, dim3(n_img), dim3(256), 0, s, given_centroids, K, D, cmax, 1.0f);
  }
  auto estep = [&]() -> int {
    if (screen) {
      if (hipMemsetAsync(amb_count, 0, (size_t)n_img * 4, s) != hipSuccess) return SPML_ERR_LAUNCH;
#define SPML_BIGS(NK_, NP_, NS_) if (nk16 == NK_ && npt == NP_) return launch_screened_t<NK_, NP_, NS_>(a, as, grid_screen, grid_list, s);
      SPML_BIGS(5, 2, 4) SPML_BIGS(9, 2, 4) SPML_BIGS(17, 2, 4) SPML_BIGS(33, 1, 2)
#undef SPML_BIGS
      return SPML_ERR_UNSUPPORTED;
    }
#define SPML_BIG(NK_, NP_) if (nk16 == NK_ && npt == NP_) return launch_assign_t<NK_, NP_>(a, grid, s);
    SPML_BIG(5, 2) SPML_BIG(9, 2) SPML_BIG(17, 2) SPML_BIG(33, 1)
#undef SPML_BIG
    return SPML_ERR_UNSUPPORTED;
  };
  auto finalize = [&](const float* given) {
    if (!given)
      hipLaunchKernelGGL(bigk_norms, dim3((unsigned)(((int64_t)n_img * K + 3) / 4)), dim3(256), 0, s, sums64,
                         (int64_t)n_img * K, D, inv);
    const int64_t items = (int64_t)n_img * MT * 2 * nk16 * 32;
    hipLaunchKernelGGL(bigk_emit, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, sums64, given, inv,
                       K, D, nk16, MT, n_img, given ? (float*)nullptr : cent_f, afrag);
  };
  // labels (from the E-step's keys, or lab32 as is) -> fixed-point sums by (image, label)
  auto msums = [&](bool from_keys) -> int {
    hipLaunchKernelGGL(bigk_hist, dim3(pblocks), dim3(kSortBlock), sort_lds, s,
                       from_keys ? keys : (unsigned long long*)nullptr, lab32, P, seg_off, n_img, K, counts);
    hipLaunchKernelGGL(bigk_scan, dim3(n_img), dim3(1024), 0, s, counts, K, seg_off, start, cursor);
    if (hipMemsetAsync(order, 0xff, (size_t)P * 4, s) != hipSuccess) return SPML_ERR_LAUNCH;
    hipLaunchKernelGGL(bigk_scatter, dim3(pblocks), dim3(kSortBlock), sort_lds, s, lab32, P, seg_off, n_img, K, cursor,
                       order, order_gid);
    const unsigned chunks = (unsigned)((P + kGatherRows - 1) / kGatherRows);
    hipLaunchKernelGGL(bigk_gather_sum, dim3(chunks), dim3(64 * ((D + 127) / 128)), 0, s, x, P, D, order,
                       order_gid, sums64);
    return launch_status();
  };
  auto mstep = [&](bool from_keys) -> int {
    const int r = msums(from_keys);
    if (r != SPML_OK) return r;
    finalize(nullptr);
    return launch_status();
  };

  int rc = SPML_OK;
  if (given_centroids && !sums_out) {
    finalize(given_centroids);
    rc = estep();
    if (rc != SPML_OK) return rc;
    hipLaunchKernelGGL(bigk_hist, dim3(pblocks), dim3(kSortBlock), sort_lds, s, keys, lab32, P, seg_off, n_img, K,
                       (int*)nullptr);
    return launch_status();
  }
  if (!given_centroids && iterations <= 0) return SPML_OK;
  if (hipMemsetAsync(counts, 0, (size_t)n_img * K * 4, s) != hipSuccess ||
      hipMemsetAsync(sums64, 0, (size_t)n_img * K * D * 8, s) != hipSuccess)
    return SPML_ERR_LAUNCH;
  if (given_centroids) {                       // fused pass: E-step + raw sums by the new labels
    finalize(given_centroids);
    rc = estep();
    if (rc != SPML_OK) return rc;
    rc = msums(true);
    if (rc != SPML_OK) return rc;
    const int64_t nsum = (int64_t)n_img * K * D;
    hipLaunchKernelGGL(bigk_sums_to_f32, dim3((unsigned)((nsum + 255) / 256)), dim3(256), 0, s, sums64,
                       nsum, sums_out);
    return launch_status();
  }
  rc = mstep(false);
  if (rc != SPML_OK) return rc;
  for (int it = 0; it < iterations; ++it) {
    rc = estep();
    if (rc != SPML_OK) return rc;
    if (it + 1 < iterations) rc = mstep(true);
    else {
      hipLaunchKernelGGL(bigk_hist, dim3(pblocks), dim3(kSortBlock), sort_lds, s, keys, lab32, P, seg_off, n_img, K,
                         (int*)nullptr);
      rc = launch_status();
    }
    if (rc != SPML_OK) return rc;
  }
  return SPML_OK;
}

}  // namespace spml
