// Spherical (vMF) k-means over a ragged batch of images: fused assign + update.
//
// Replaces kmeans_with_initial_labels (segsort/common.py:67-97) and the
// per-image loop of segment_by_kmeans (common.py:337-373) of the reference:
//   for it in range(iterations):
//       protos = normalize(scatter_add(X, labels))      # M-step  common.py:11-41
//       labels = argmax(X @ protos.T, 1)                # E-step  common.py:44-64
// which materialises a [P,K] fp32 matrix and re-reads X twice per iteration.
//
// Here one *pass* streams X from HBM exactly once and does both steps:
//   E: labels[p] = argmax_k <x_p, c_k> against the current prototypes
//   M: partial sums of x_p by the NEW label, for the next prototypes
// (iterations + 1 passes: the first is M-only on the initial labels, the last
// E-only).  Algorithmic HBM bytes per pass: P*D*4 (+ P*4 labels).
//
// Code paths, chosen per shape by make_plan (spml_kmeans_last_path() names the one taken):
//   kmeans_pass16<MT16,Q,TAIL,PRE>   D = 32q + tail (q in {1,2,4,8}), K <= 64: the main
//       kernel (below, "v3"): 16x16x32 MFMA tiles, prototypes of a wave register-resident,
//       32-pixel tiles as split-f16 fragment blocks in a 2-slot LDS ring fed by direct-to-LDS
//       DMA; E-step arg-max in registers ("swapped operand"), M-step as a one-hot MFMA with
//       operands from the LDS transpose read; no atomics -> bit-reproducible.  PRE = tiles
//       pre-converted once per call (by the seed pass, or by kmeans_preconvert);
//   kmeans_pass16k<MTW,Q,TAIL>       64 < K <= 256 on the same tile pipeline, prototype
//       tiles distributed over the waves in both steps;
//   kmeans_pass<NT,KS,KSPLIT>        ("v2") other even D <= 320 with K <= 64: 32x32x16
//       tiles, channel range split over the waves, fp32 rows DMA'd raw and split on the fly;
//   bigk_assign + sorted gather-sum  K > 64 beyond those (e.g. 32x32 = 1024 clusters on
//       258x258x514): kmeans_big.hip -- pixel-stationary MFMA E-step with a running arg-max,
//       counting-sort + fixed-point gather M-step;
//   generic_assign + segment sums    anything else: fp32 FMA dot products, run-length
//       atomics for the M-step.
// Per-workgroup partial sums of an M-step go to a slab; kmeans_reduce_slabs adds the slabs
// in fixed order and kmeans_normalize normalises (empty cluster -> zero prototype, as the
// reference) and emits the split-f16 prototypes of the next pass.
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "common.hpp"
#include "kmeans_tile.hpp"

namespace spml {

int segment_sum_launch(const float* x, const int64_t* ids, int64_t P, int D, int64_t M,
                       float* sums, hipStream_t s, long long* sums64 = nullptr);
// kmeans_big.hip
bool bigk_shape(int64_t P, int D, int K, int n_img);
size_t bigk_workspace_bytes(int64_t P, int D, int K, int n_img);
int bigk_run(const float* x, int64_t P, int D, const int64_t* seg_off, int n_img,
             int64_t max_seg_len, int K, const float* given_centroids, int iterations,
             int32_t* lab32, float* cent_f, float* sums_out, int flags, void* ws, hipStream_t s);


namespace {


constexpr int kKsMax = 5;

// store of a converted tile by the seed pass (273 MB per 513x513x258 call, read back by the
// next pass): non-temporal stores keep the lines out of the L2 / Infinity-Cache write-back
// queue the first fused pass would otherwise have to wait behind
#ifdef SPML_SEED_PLAIN_STORE
#define SPML_SEED_STORE(ptr, val) (*(ptr) = (val))
#else
#define SPML_SEED_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#endif

constexpr int kNBuf = 3;      // LDS tile ring: one being computed, two in flight

template <int NT, int KS, int KSPLIT>
struct PassCfg {
  static constexpr int PT = 4 / KSPLIT;         // 32-pixel tiles per workgroup step
  static constexpr int TPW = 32 * PT;           // pixels per workgroup step
  static constexpr int RQ = 16 / KSPLIT;        // accumulator regs reduced per wave
};

__host__ __device__ inline int pass_nvt(int D, int KSPLIT) {
  const int TPW = 32 * (4 / KSPLIT);
  return (TPW * D * 4 + 16 + 4095) / 4096;
}

__host__ __device__ inline size_t pass_lds_bytes(int D, int NT, int KSPLIT) {
  size_t b = (size_t)kNBuf * pass_nvt(D, KSPLIT) * 4096;   // X tile ring
  if (KSPLIT > 1) b += (size_t)4 * NT * 16 * 64 * 4;       // partial-dot exchange
  b += 4 * 32 * 8;                                         // candidates (val, idx)
  b += 128 * 4;                                            // labels of the tile
  b += (size_t)kNBuf * 256 * 4;                            // incoming labels (M-only pass)
  return b;
}

// Tiles [t_begin, t_end) of workgroup g of G.  The workgroups of the first dispatch round (g < G / 2: one per CU; the
// second round shares the CUs with them) run ~5 % faster (tools/probe_pass_wgs.py: 52.6 against 55.1 us for 16 tiles at
// 513 x 513 x 258), so a left-over tile belongs there: the tiles are split proportionally over the G / 2 PAIRS
// (g, g + G / 2) and the first workgroup of a pair takes the larger half.  (Proportional, not "T / G each and the rest
// to the first workgroups": equal strides between the workgroups' streams put all of them on the same few HBM channels
// at the same time -- the seed pass, which writes as much as it reads, took 135 instead of 118 us that way.)
__device__ __forceinline__ void tile_range(int64_t T, int g, int G, int64_t& t_begin, int64_t& t_end) {
  if (G & 1) {
    t_begin = (T * g) / G;
    t_end = (T * (g + 1)) / G;
    return;
  }
  const int H = G >> 1, q = g < H ? g : g - H;
  const int64_t s0 = (T * q) / H, s1 = (T * (q + 1)) / H, first = (s1 - s0 + 1) >> 1;
  t_begin = g < H ? s0 : s0 + first;
  t_end = g < H ? s0 + first : s1;
}

template <int NT, int KS, int KSPLIT>
__global__ __launch_bounds__(256) void kmeans_pass(PassArgs a) {
  using Cfg = PassCfg<NT, KS, KSPLIT>;
  constexpr int PT = Cfg::PT, TPW = Cfg::TPW, RQ = Cfg::RQ;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5;
  const int j = lane & 31;              // pixel column inside a 32-pixel tile
  const int pt = wave / KSPLIT;         // which 32-pixel tile of the step
  const int ksub = wave % KSPLIT;       // which slice of the channel range
  const int D = a.D, K = a.K;
  const int img = blockIdx.y, g = blockIdx.x;
  const int nvt = a.nvt;
  const size_t buf_bytes = (size_t)nvt * 4096;

  // ---- LDS carve-up ----
  size_t off = kNBuf * buf_bytes;
  float* xchg = reinterpret_cast<float*>(lds + off);          // [PT][KSPLIT][KSPLIT][NT][RQ/4][64] x float4
  if (KSPLIT > 1) off += (size_t)4 * NT * 16 * 64 * 4;
  float* cand_v = reinterpret_cast<float*>(lds + off);        // [4][32]
  int* cand_i = reinterpret_cast<int*>(lds + off + 4 * 32 * 4);
  off += 4 * 32 * 8;
  int* lab = reinterpret_cast<int*>(lds + off);               // [128]
  off += 128 * 4;
  int* labin = reinterpret_cast<int*>(lds + off);             // [kNBuf][256]

  KM_CLOCK_BEGIN
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  int64_t t_begin, t_end;
  tile_range(T, g, a.G, t_begin, t_end);
  if (t_begin >= t_end) {                // nothing to do: contribute a zero slab
    if (a.do_accum) {
      float* z = a.slabs + ((size_t)img * a.G + g) * K * D;
      for (int i = tid; i < K * D; i += 256) z[i] = 0.f;
    }
    KM_CLOCK_END
    return;
  }

  // ---- prototypes -> registers (A operand), once per workgroup ----
  half8 ah[KS][NT], al[KS][NT];
  if (a.do_assign) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int k0 = 16 * (ksub + KSPLIT * i) + 8 * half;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const size_t o = ((size_t)img * a.kpad + 32 * t + j) * a.dpad + k0;
        ah[i][t] = *reinterpret_cast<const half8*>(a.cent_h + o);
        al[i][t] = *reinterpret_cast<const half8*>(a.cent_l + o);
      }
    }
  }

  // ---- HBM -> LDS tile copy: direct-to-LDS DMA, nvt x 1 KB per wave --------
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x);
  const int ops_per_tile = nvt + (a.do_assign ? 0 : 1);
  auto tile_issue = [&](int64_t t) {
    const int buf = (int)((t - t_begin) % kNBuf);
    const int64_t r0 = seg0 + t * TPW;
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = r0 * D * 4, b1 = b0 + (int64_t)nrows * D * 4;
    const int64_t a0 = b0 & ~(int64_t)15;
    const int nvec = (int)((b1 - a0 + 15) >> 4);
    unsigned char* dst0 = lds + buf * buf_bytes;
    for (int i = 0; i < nvt; ++i) {
      const int v = min(i * 256 + tid, nvec - 1);      // surplus lanes re-read the last vector
      int64_t o = a0 + 16 * (int64_t)v;
      o = min(o, a.x_bytes - 16);                      // see tail fix-up below
      unsigned char* dst = dst0 + (size_t)(i * 256 + wave * 64) * 16;   // wave-uniform
      __builtin_amdgcn_global_load_lds((gptr_t)(xbase + o), (lptr_t)dst, 16, 0, 0);
    }
    if (!a.do_assign) {                                // incoming labels of the tile
      const int64_t p = min(r0 + min(wave * 64 + lane, TPW - 1), a.P - 1);
      int* dst = labin + buf * 256 + wave * 64;        // wave-uniform
      __builtin_amdgcn_global_load_lds(label_src(a, p), (lptr_t)dst, 4, 0, 0);
    }
  };

  // ---- M-step state: this wave owns channel tiles dt = wave + 4*m ----------
  // sums^T[d][k] += X^T[d][p] * onehot[p][k] on the matrix cores; the accumulators
  // stay in registers for the whole workgroup lifetime (no atomics, fixed order).
  constexpr int MT = 3;                  // channel tiles per wave (D <= 384)
  const int n_dt = (D + 31) >> 5;
  float16v macc[MT][NT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int q = 0; q < NT; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) macc[m][q][r] = 0.f;

  tile_issue(t_begin);
  if (t_begin + 1 < t_end) tile_issue(t_begin + 1);

  for (int64_t t = t_begin; t < t_end; ++t) {
    const int buf = (int)((t - t_begin) % kNBuf);
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = (seg0 + t * TPW) * D * 4;
    const int shift = (int)(b0 & 15);                   // 0 or 8
    // tile t has landed once at most the next tile's copies are outstanding
    if (t + 1 < t_end) wait_vmcnt(ops_per_tile);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                                       // ... for every wave; ring slot t-1 is free
    if (t + 2 < t_end) tile_issue(t + 2);

    unsigned char* xs = lds + buf * buf_bytes;
    {
      // x_bytes % 16 == 8 and this tile touches the buffer end: the last vector
      // was fetched 8 bytes early; move its upper half down.
      const int64_t a0 = b0 & ~(int64_t)15;
      const int nvec = (int)((b0 + (int64_t)nrows * D * 4 - a0 + 15) >> 4);
      if (a0 + 16 * (int64_t)nvec > a.x_bytes) {        // workgroup-uniform, rare
        if (tid == 0) {
          float2* slot = reinterpret_cast<float2*>(xs + 16 * (size_t)(nvec - 1));
          slot[0] = slot[1];
        }
        wg_barrier();
      }
    }
    const unsigned char* xrow = xs + shift;
    if (!a.do_assign) {
      if (tid < TPW) lab[tid] = tid < nrows ? labin[buf * 256 + tid] : -1;
      wg_barrier();
    }

    if (a.do_assign) {
      // ================= E-step: MFMA similarity + arg-max =================
      float16v acc_h[NT], acc_x[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[q][r] = 0.f; acc_x[q][r] = 0.f; }

      // all B-fragment reads of this tile are issued up front (no branches in
      // between) so that their LDS latency overlaps; channels >= D read as 0.
      float xv[KS][8];
      const float2* prow = reinterpret_cast<const float2*>(
          xrow + ((size_t)(pt * 32 + j) * D + 16 * ksub + 8 * half) * 4);
#pragma unroll
      for (int i = 0; i < KS; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float2 f = prow[8 * KSPLIT * i + e];   // immediate offsets
          xv[i][2 * e] = f.x;
          xv[i][2 * e + 1] = f.y;
        }
#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int kb = 16 * (ksub + KSPLIT * i);
        if (kb < D) {                                  // wave-uniform
          if (kb + 16 > D) {                           // ragged last step: channels >= D are 0
#pragma unroll
            for (int e = 0; e < 8; ++e) xv[i][e] = (kb + 8 * half + e < D) ? xv[i][e] : 0.f;
          }
          half8 bh, bl;
          split8(xv[i], bh, bl);
#pragma unroll
          for (int q = 0; q < NT; ++q) {
            acc_h[q] = mfma32(ah[i][q], bh, acc_h[q]);
            acc_x[q] = mfma32(ah[i][q], bl, acc_x[q]);
            acc_x[q] = mfma32(al[i][q], bh, acc_x[q]);
          }
        }
      }

      float best = -INFINITY;
      int best_i = 0x7fffffff;
      if (KSPLIT == 1) {
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * q + (r & 3) + 8 * (r >> 2) + 4 * half;
            const float s = acc_h[q][r] + acc_x[q][r] * kSplitInv;
            if (c < K && s > best) { best = s; best_i = c; }
          }
      } else {
        // partial dot products of the KSPLIT channel slices meet in LDS:
        // wave `d` reduces accumulator registers [d*RQ, (d+1)*RQ); 16-B accesses
        float4v* x4 = reinterpret_cast<float4v*>(xchg);
#pragma unroll
        for (int d = 0; d < KSPLIT; ++d)
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r4 = 0; r4 < RQ / 4; ++r4) {
              float4v v;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                const int r = d * RQ + r4 * 4 + e;
                v[e] = acc_h[q][r] + acc_x[q][r] * kSplitInv;
              }
              x4[((((size_t)(pt * KSPLIT + d) * KSPLIT + ksub) * NT + q) * (RQ / 4) + r4) * 64 + lane] = v;
            }
        wg_barrier();
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int r4 = 0; r4 < RQ / 4; ++r4) {
            float4v s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int src = 0; src < KSPLIT; ++src)
              s += x4[((((size_t)(pt * KSPLIT + ksub) * KSPLIT + src) * NT + q) * (RQ / 4) + r4) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int rr = ksub * RQ + r4 * 4 + e;
              const int c = 32 * q + (rr & 3) + 8 * (rr >> 2) + 4 * half;
              if (c < K && s[e] > best) { best = s[e]; best_i = c; }
            }
          }
      }
      // the two lane halves hold different prototype rows of the same pixel
      {
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
      }
      if (KSPLIT > 1) {
        if (lane < 32) {
          cand_v[wave * 32 + lane] = best;
          cand_i[wave * 32 + lane] = best_i;
        }
        wg_barrier();
        if (ksub == 0 && lane < 32) {
#pragma unroll
          for (int s2 = 1; s2 < KSPLIT; ++s2) {
            const float ob = cand_v[(wave + s2) * 32 + lane];
            const int oi = cand_i[(wave + s2) * 32 + lane];
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
          }
        }
      }
      if (ksub == 0 && lane < 32) {
        const int p = pt * 32 + lane;
        lab[p] = p < nrows ? best_i : -1;
        if (p < nrows) label_store(a, seg0 + t * TPW + p, best_i);
      }
      if (a.do_accum) wg_barrier();
    }

    if (a.do_accum && wave < n_dt) {          // wave-uniform: owns at least one channel tile
      // ================= M-step on the matrix cores =========================
      // A = X^T (rows = channels of the owned tile, k = 16 pixels, split-f16),
      // B = one-hot labels (k = pixels, cols = clusters); the low split half is
      // multiplied by an exact 2^-11 folded into B, so one accumulator suffices.
      const bool partial = nrows < TPW;
#pragma unroll 1
      for (int ks = 0; ks < TPW / 16; ++ks) {   // (rolled: keeps register pressure flat)
        const int p0 = 16 * ks + 8 * half;
        if (16 * ks < nrows) {                  // wave-uniform
          int labs[8];
          {
            typedef int int4v __attribute__((ext_vector_type(4)));
            const int4v l0 = *reinterpret_cast<const int4v*>(lab + p0);
            const int4v l1 = *reinterpret_cast<const int4v*>(lab + p0 + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { labs[i] = l0[i]; labs[4 + i] = l1[i]; }
          }
          half8 oh[NT], ol[NT];
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const bool hit = labs[i] == 32 * q + j;
              oh[q][i] = hit ? (_Float16)1.0f : (_Float16)0.0f;
              ol[q][i] = hit ? (_Float16)kSplitInv : (_Float16)0.0f;
            }
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const int dt = wave + 4 * m;
            if (dt < n_dt) {                    // wave-uniform
              const int d = 32 * dt + j;
              const float* col = reinterpret_cast<const float*>(xrow) + min(d, D - 1);
              float v[8];
#pragma unroll
              for (int i = 0; i < 8; ++i) v[i] = col[(size_t)(p0 + i) * D];
              if (partial || 32 * dt + 32 > D) {
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = (d < D && p0 + i < nrows) ? v[i] : 0.f;
              }
              half8 xh, xl;
              split8(v, xh, xl);
#pragma unroll
              for (int q = 0; q < NT; ++q) {
                macc[m][q] = mfma32(xh, oh[q], macc[m][q]);
                macc[m][q] = mfma32(xl, ol[q], macc[m][q]);
              }
            }
          }
        }
      }
    }
  }

  if (a.do_accum) {
    // every (cluster, channel) word of the slab has exactly one owner lane
    float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int dt = wave + 4 * m;
      if (dt < n_dt) {
#pragma unroll
        for (int q = 0; q < NT; ++q) {
          const int c = 32 * q + j;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int d = 32 * dt + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (c < K && d < D) slab[(size_t)c * D + d] = macc[m][q][r];
          }
        }
      }
    }
  }
  KM_CLOCK_END
}

// ===========================================================================
// v3 kernel: 16x16x32 MFMA tiles on split-f16 fragment blocks.
//
// For D = 32*Q + tail (tail = 0, 2 = (y, x) location, or up to 8 on pre-converted tiles)
// and K <= 64.  Differences to kmeans_pass above:
//   * a 32-pixel tile lives in LDS as fragment-major blocks ([k-step][pixel half][hi|lo]
//     x 1 KB, 16 B per (pixel, 8-channel group), slots permuted by frag_slot): every
//     E-step operand is one conflict-free ds_read_b128, every M-step operand a
//     conflict-free ds_read_b64_tr_b16, and every element is split to f16 once -- per
//     call when PRE (the blocks are DMA'd from the pre-converted copy of X into a 2-slot
//     ring), per pass otherwise (raw fp32 rows are DMA'd and converted in LDS);
//   * wave w owns prototype rows [16w, 16w+16) for ALL channels (A fragments in
//     registers): no k-split, no partial-dot exchange; only a 4-KB candidate hand-off
//     for the arg-max across the K/16 waves, after which every wave rebuilds the tile's
//     labels in registers (no further barrier);
//   * the tail channels are a zero-padded extra k-step (a compact 256-B block per
//     (pixel half, hi|lo));
//   * LDS <= 80 KB -> TWO workgroups per CU.
// ===========================================================================
__host__ __device__ inline size_t pass16_lds_bytes(int D, bool pre) {
  const int q = D / 32;
  const size_t conv = (size_t)(q + (D > 32 * q ? 1 : 0)) * 4096;
  const size_t tiles = pre ? 2 * conv : (size_t)pass_nvt(D, 4) * 4096 + conv;
  return tiles + 16 * 32 * 8 + 4 * 32 * 4 + (pre ? 2 : 1) * 256 * 4 + 64;
}

// One-off layout change in front of the iterations: X fp32 [P,D] -> per 32-pixel tile the
// split-f16 fragment blocks the pass kernel wants in LDS, so that every pass DMAs its
// operands straight into place (same 4 B per element as fp32: hi + lo halves) and spends
// no time converting.  Rows past the end of an image are written as zeros.
template <int Q, int TAIL>
__global__ __launch_bounds__(256) void kmeans_preconvert(const float* __restrict__ x, int D,
                                                         const int64_t* __restrict__ seg_off,
                                                         unsigned char* __restrict__ xc) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int lg = lane >> 4, lc = lane & 15;
  const int img = blockIdx.y;
  const int64_t seg0 = seg_off[img];
  const int64_t len = seg_off[img + 1] - seg0;
  const int64_t T = (len + 31) / 32;
  const int64_t tile0 = pre_tile0(seg0, img);
  for (int64_t t = blockIdx.x; t < T; t += gridDim.x) {
    unsigned char* out = xc + (size_t)(tile0 + t) * pre_tile_bytes(Q, TAIL);
    const int nrows = (int)min((int64_t)32, len - 32 * t);
    const float* rows = x + (size_t)(seg0 + 32 * t) * D;
#pragma unroll
    for (int u = wave; u < 2 * Q; u += 4) {
      const int st = u >> 1, n = u & 1;
      const int pix = 16 * n + lc;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (pix < nrows) {
        const float* srow = rows + (size_t)pix * D + 32 * st + 8 * lg;
        if (D & 1) {                               // odd row length: rows are only 4-B aligned
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = srow[e];
        } else {
          const float2* src = reinterpret_cast<const float2*>(srow);
#pragma unroll
          for (int e = 0; e < 4; ++e) { const float2 f = src[e]; v[2 * e] = f.x; v[2 * e + 1] = f.y; }
        }
      }
      half8 h, l;
      split8(v, h, l);
      unsigned char* blk = out + (size_t)(u * 2) * 1024 + (size_t)frag_slot(lc, lg) * 16;
      *reinterpret_cast<half8*>(blk) = h;
      *reinterpret_cast<half8*>(blk + 1024) = l;
    }
    if (TAIL && wave < 2 && lane < 16) {
      const int pix = 16 * wave + lane;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = 0.f;
      if (pix < nrows) {                           // 1..8 channels beyond the 32*Q block
        const float* srow = rows + (size_t)pix * D + 32 * Q;
        const int tl = D - 32 * Q;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = e < tl ? srow[e] : 0.f;
      }
      half8 h, l;
      split8(v, h, l);
      unsigned char* blk = out + (size_t)Q * 4096 + (size_t)(wave * 2) * 256 + (size_t)lane * 16;
      *reinterpret_cast<half8*>(blk) = h;
      *reinterpret_cast<half8*>(blk + 256) = l;
    }
  }
}

// PRE: the tiles arrive pre-converted (kmeans_preconvert) and are DMA'd straight into a
// 2-slot ring of converted buffers: no raw slot, no conversion phase, 2 barriers per tile.
// workgroups per CU the register budget allows (2 = 256 registers per lane, 1 = 512): the
// widest instantiations (K > 48 with D >= 256; K > 128 with D >= 64 on the many-cluster
// kernel) would spill at 256
__host__ __device__ constexpr int pass16_wg_per_cu(int mt16, int q) { return (mt16 == 4 && q == 8) ? 1 : 2; }
__host__ __device__ constexpr int pass16k_wg_per_cu(int mtw, int q, int tail) {
  return mtw * (8 * (q + tail) + 4 * (2 * q + tail)) > 140 ? 1 : 2;
}

template <int MT16, int Q, int TAIL, bool PRE>
__global__ __launch_bounds__(256, pass16_wg_per_cu(MT16, Q)) void kmeans_pass16(PassArgs a) {
  constexpr int TPW = 32;
  constexpr int QE = Q + TAIL;                   // k-steps incl. the (zero padded) location step
  constexpr int NDT = 2 * Q + TAIL;              // 16-channel tiles of the M-step
  constexpr int NDTW = (NDT + 3) / 4;            // ... per wave
  constexpr int NSTW = (Q + 3) / 4;              // 32-channel super tiles per wave (M-step)
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4;             // lane group 0..3
  const int lc = lane & 15;             // column inside a 16-wide tile
  const int D = a.D, K = a.K;
  constexpr int tail = 2 * TAIL;         // D = 32*Q + tail
  const int img = blockIdx.y, g = blockIdx.x;
  const int nvt = a.nvt;

  constexpr int CONV = QE * 4096;                             // [QE][2][hi|lo] x 1 KB
  size_t off = PRE ? 0 : (size_t)nvt * 4096;
  unsigned char* xs = lds;                                    // raw tile (DMA target; !PRE)
  unsigned char* conv0 = lds + off;                           // converted tile(s)
  off += (size_t)(PRE ? 2 : 1) * CONV;
  float* cand_v = reinterpret_cast<float*>(lds + off);        // [4 waves][4 lane groups][32]
  int* cand_i = reinterpret_cast<int*>(lds + off + 16 * 32 * 4);
  off += 16 * 32 * 8;
  int* labw = reinterpret_cast<int*>(lds + off);              // [4 waves][32] labels, per wave
  off += 4 * 32 * 4;
  int* labin = reinterpret_cast<int*>(lds + off);             // [1|2][256] incoming labels (M-only)

  KM_CLOCK_BEGIN
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  int64_t t_begin, t_end, t_step = 1;
  tile_range(T, g, a.G, t_begin, t_end);
  // tiles g, g + G, g + 2 G, ...: at any time the workgroups of an image read ONE contiguous window of X (G tiles = 17 MB
  // at G = 512) instead of G streams spread over the whole tensor: seed pass 112 -> 101 us, fused pass 55.3 -> 53.8 us
  // (tools/bench_kmeans.py, three A/B pairs; SPML_KMEANS_STRIDED=0: contiguous ranges, tile_range).  The left-over
  // tiles fall to workgroups 0 .. T % G - 1, i.e. to the first dispatch round (see tile_range).
  if (a.strided) { t_begin = g; t_end = T; t_step = a.G; }
  if (t_begin >= t_end) {
    if (a.do_accum) {
      float* z = a.slabs + ((size_t)img * a.G + g) * K * D;
      for (int i = tid; i < K * D; i += 256) z[i] = 0.f;
    }
    KM_CLOCK_END
    return;
  }
  // (s_setprio 1 for the workgroups of the second dispatch round, which lose the issue arbitration to their older
  // CU partners, only swaps which half is the slower one: measured, tools/probe_pass_wgs.py)
  // byte offsets of this lane inside a fragment block: E-step operand (pixel lc, channel
  // group lg) and M-step transpose read (row lc>>2 of a [4 pixel][16 channel] sub-block)
  const int eoff = frag_slot(lc, lg) * 16;
  const int troff = (lg >> 1) * 2048 + frag_slot(8 * (lg & 1) + (lc >> 2), (lc >> 1) & 1) * 16 + 8 * (lc & 1);

  // ---- prototypes of this wave's 16 rows -> registers (A operand) ----
  half8 ah[QE], al[QE];
  if (TAIL) {                            // location k-step: only 2 of its 32 channels exist
    for (int i = tid; i < 1024; i += 256) {
      reinterpret_cast<float*>(conv0 + (size_t)Q * 4096)[i] = 0.f;
      if (PRE) reinterpret_cast<float*>(conv0 + CONV + (size_t)Q * 4096)[i] = 0.f;
    }
    if (PRE) wg_barrier();               // before the first DMA lands in those blocks
  }

  // ---- M-step accumulators: sums^T[d][k], this wave owns channel tiles w + 4i ----
  // macc[2*i + ct][q]: super tile s = wave + 4*i, channels 32*s + 16*ct + row;
  // macc[2*NSTW][q]: the location tile (wave 3 only)
  float4a macc[2 * NSTW + 1][MT16];
#pragma unroll
  for (int i = 0; i <= 2 * NSTW; ++i)
#pragma unroll
    for (int q = 0; q < MT16; ++q) macc[i][q] = float4a{0.f, 0.f, 0.f, 0.f};

  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x);
  const int64_t tile0 = pre_tile0(seg0, img);
  auto tile_issue = [&](int64_t t, int slot) {
    const int64_t r0 = seg0 + t * TPW;
    if constexpr (PRE) {
      const unsigned char* tb = a.xc + (size_t)(tile0 + t) * pre_tile_bytes(Q, TAIL) + 16 * lane;
      unsigned char* dst0 = conv0 + (size_t)slot * CONV;
      if (a.do_assign && MT16 < 4) {
        // the E-step keeps waves 0..MT16-1 on the matrix cores: wave 3 issues the whole copy
        if (wave == 3) {
#pragma unroll 4
          for (int b = 0; b < 4 * Q; ++b)
            __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)b * 1024), (lptr_t)(dst0 + b * 1024), 16, 0, 0);
          if (TAIL && lane < 16) {
#pragma unroll
            for (int b = 0; b < 4; ++b)
              __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)Q * 4096 + b * 256),
                                               (lptr_t)(dst0 + (4 * Q + b) * 1024), 16, 0, 0);
          }
        }
      } else {
        for (int b = wave; b < 4 * Q; b += 4)
          __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)b * 1024), (lptr_t)(dst0 + b * 1024), 16, 0, 0);
        if (TAIL && lane < 16)
          __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)Q * 4096 + wave * 256),
                                           (lptr_t)(dst0 + (4 * Q + wave) * 1024), 16, 0, 0);
      }
      if (!a.do_assign) {
        const int64_t p = min(r0 + min(wave * 64 + lane, TPW - 1), a.P - 1);
        int* dst = labin + slot * 256 + wave * 64;
        __builtin_amdgcn_global_load_lds(label_src(a, p), (lptr_t)dst, 4, 0, 0);
      }
      return;
    }
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = r0 * D * 4, b1 = b0 + (int64_t)nrows * D * 4;
    const int64_t a0 = b0 & ~(int64_t)15;
    const int nvec = (int)((b1 - a0 + 15) >> 4);
    const unsigned char* tbase = xbase + a0;                 // wave-uniform 64-bit base
    const int lim = (int)min((int64_t)0x7ffffff0, a.x_bytes - 16 - a0);   // last legal 16-B load
    const int last = min(16 * (nvec - 1), lim);
    if (a.do_assign && MT16 < 4) {
      // the E-step keeps waves 0..MT16-1 on the matrix cores: wave 3 issues the whole copy
      if (wave == 3) {
        int offs = 16 * lane;
        for (int i = 0; i < 4 * nvt; ++i) {
          unsigned char* dst = xs + (size_t)i * 1024;
          __builtin_amdgcn_global_load_lds((gptr_t)(tbase + min(offs, last)), (lptr_t)dst, 16, 0, 0);
          offs += 1024;
        }
      }
    } else {
      int offs = 16 * tid;
      for (int i = 0; i < nvt; ++i) {
        unsigned char* dst = xs + (size_t)(i * 256 + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)(tbase + min(offs, last)), (lptr_t)dst, 16, 0, 0);
        offs += 4096;
      }
    }
    if (!a.do_assign) {
      const int64_t p = min(r0 + min(wave * 64 + lane, TPW - 1), a.P - 1);
      int* dst = labin + wave * 64;
      __builtin_amdgcn_global_load_lds(label_src(a, p), (lptr_t)dst, 4, 0, 0);
    }
  };

  KM_TRACE_DECL
  tile_issue(t_begin, 0);
  // prototype rows -> registers while the first tile is in flight
  if (a.do_assign && wave < MT16) {
#pragma unroll
    for (int s = 0; s < QE; ++s) {
      const size_t o = ((size_t)img * a.kpad + 16 * wave + lc) * a.dpad + 32 * s + 8 * lg;
      ah[s] = *reinterpret_cast<const half8*>(a.cent_h + o);
      al[s] = *reinterpret_cast<const half8*>(a.cent_l + o);
    }
  }
  KM_MARK(7)
  for (int64_t t = t_begin; t < t_end; t += t_step) {
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = (seg0 + t * TPW) * D * 4;
    const int shift = (int)(b0 & 15);
    const int slot = PRE ? (int)(((t - t_begin) / t_step) & 1) : 0;
    unsigned char* conv = conv0 + (size_t)slot * CONV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                               // tile t landed; the other buffer is free
    KM_MARK(0)
    int mylab = -1;                              // label of pixel `lane` (lanes < 32), M-only pass
    if constexpr (PRE) {
      if (t + t_step < t_end) tile_issue(t + t_step, slot ^ 1);   // in flight during E- and M-step
      if (!a.do_assign && lane < 32) mylab = lane < nrows ? labin[slot * 256 + lane] : -1;
      KM_MARK(2)
    } else {
    {
      const int64_t a0 = b0 & ~(int64_t)15;
      const int nvec = (int)((b0 + (int64_t)nrows * D * 4 - a0 + 15) >> 4);
      if (a0 + 16 * (int64_t)nvec > a.x_bytes) {        // see kmeans_pass: tail of the buffer
        if (tid == 0) {
          float2* slot = reinterpret_cast<float2*>(xs + 16 * (size_t)(nvec - 1));
          slot[0] = slot[1];
        }
        wg_barrier();
      }
    }
    const unsigned char* xrow = xs + shift;

    // ---- raw fp32 -> fragment-major split-f16 (each element once) ----
    // item = (pixel p, 4 channels): consecutive lanes take consecutive pixels, so raw
    // reads (row stride = 2 mod 64 banks) are conflict-free
    {
      constexpr int NIT = (32 * 8 * Q) / 256;    // = Q
      float2 raw[NIT][2];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = it * 256 + tid;
        const int pix = id & 31, qd = id >> 5;
        const float2* src = reinterpret_cast<const float2*>(xrow + ((size_t)pix * D + 4 * qd) * 4);
        raw[it][0] = src[0];
        raw[it][1] = src[1];
      }
      float2 tl = {0.f, 0.f};
      if (tail && tid < 32) tl = *reinterpret_cast<const float2*>(xrow + ((size_t)tid * D + D - 2) * 4);
      if (!a.do_assign && lane < 32) mylab = labin[lane];
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int id = it * 256 + tid;
        const int pix = id & 31, qd = id >> 5;
        const bool ok = pix < nrows;             // stale LDS may hold NaN patterns
        const float v0 = ok ? raw[it][0].x : 0.f, v1 = ok ? raw[it][0].y : 0.f;
        const float v2 = ok ? raw[it][1].x : 0.f, v3 = ok ? raw[it][1].y : 0.f;
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        half4v h, l;
        _Float16 hh, ll;
        split_f16(v0, hh, ll); h[0] = hh; l[0] = ll;
        split_f16(v1, hh, ll); h[1] = hh; l[1] = ll;
        split_f16(v2, hh, ll); h[2] = hh; l[2] = ll;
        split_f16(v3, hh, ll); h[3] = hh; l[3] = ll;
        const int c = 4 * qd;
        const int s = c >> 5, lge = (c & 31) >> 3, e0 = c & 7;
        unsigned char* dst = conv + (size_t)((s * 2 + (pix >> 4)) * 2) * 1024 +
                             (size_t)frag_slot(pix & 15, lge) * 16 + 2 * e0;
        *reinterpret_cast<half4v*>(dst) = h;
        *reinterpret_cast<half4v*>(dst + 1024) = l;
      }
      if (TAIL && tid < 32) {
        _Float16 h0, l0, h1, l1;
        split_f16(tid < nrows ? tl.x : 0.f, h0, l0);
        split_f16(tid < nrows ? tl.y : 0.f, h1, l1);
        unsigned char* dst = conv + (size_t)((Q * 2 + (tid >> 4)) * 2) * 1024 + (size_t)(tid & 15) * 16;
        *reinterpret_cast<half2v*>(dst) = half2v{h0, h1};
        *reinterpret_cast<half2v*>(dst + 1024) = half2v{l0, l1};
      }
      if (lane < 32 && lane >= nrows) mylab = -1;
    }
    KM_MARK(1)
    wg_barrier();                               // conv tile ready; raw slot is free again
    if (t + t_step < t_end) tile_issue(t + t_step, 0);    // in flight during E- and M-step
    if (a.xc_out) {
      // the seed pass doubles as kmeans_preconvert: the converted tile leaves for HBM in
      // the layout the later passes DMA back (saves one read of X per k-means call)
      unsigned char* out = a.xc_out + (size_t)(tile0 + t) * pre_tile_bytes(Q, TAIL) + 16 * lane;
      for (int b = wave; b < 4 * Q; b += 4)
        SPML_SEED_STORE(reinterpret_cast<half8*>(out + (size_t)b * 1024),
                        *reinterpret_cast<const half8*>(conv + (size_t)b * 1024 + 16 * lane));
      if (TAIL && lane < 16)
        SPML_SEED_STORE(reinterpret_cast<half8*>(out + (size_t)Q * 4096 + wave * 256),
                        *reinterpret_cast<const half8*>(conv + (size_t)(4 * Q + wave) * 1024 + 16 * lane));
    }
    KM_MARK(2)
    }

    if (a.do_assign) {
      // ================= E-step =================
      if (wave < MT16) {
        float4a eh[2], ex[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) { eh[n] = float4a{0.f, 0.f, 0.f, 0.f}; ex[n] = eh[n]; }
        float4a ey[2];                                   // third chain: al * bh
#pragma unroll
        for (int n = 0; n < 2; ++n) ey[n] = float4a{0.f, 0.f, 0.f, 0.f};
        // one register set, software-pipelined by pixel half: while the three MFMAs of
        // half n run, the operands of the other half (and of the next k-step) are in
        // flight.  The reads are issued by hand with counted waits (LDS returns in order,
        // so "at most 2 outstanding" == "the older pair has landed"): the compiler's own
        // placement waits for lgkmcnt(0) and exposes one LDS round trip per half step.
        half8 bh[2], bl[2];
        const unsigned cbase = (unsigned)(size_t)(lptr_t)(conv);
        auto load_b = [&](int s, int n) {
          // the location k-step (s == Q) keeps the plain lane-linear block
          const unsigned addr = cbase + (unsigned)((s * 2 + n) * 2) * 1024u +
                                (unsigned)(s < Q ? eoff : lane * 16);
          asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                       : "=&v"(bh[n]), "=&v"(bl[n]) : "v"(addr));
        };
        load_b(0, 0);
        load_b(0, 1);
#pragma unroll
        for (int s = 0; s < QE; ++s) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            if (s + 1 < QE || n == 0)
              asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[n]), "+v"(bl[n]));
            else
              asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[n]), "+v"(bl[n]));
            eh[n] = mfma16(ah[s], bh[n], eh[n]);
            ex[n] = mfma16(ah[s], bl[n], ex[n]);
            ey[n] = mfma16(al[s], bh[n], ey[n]);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < QE) load_b(s + 1, n);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) ex[n] += ey[n];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          float best = -INFINITY;
          int best_i = 0x7fffffff;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int c = 16 * wave + 4 * lg + r;
            const float sdot = eh[n][r] + ex[n][r] * kSplitInv;
            if (c < K && sdot > best) { best = sdot; best_i = c; }
          }
          // the 4 lane groups hold different prototype rows of the same pixel: each
          // publishes its own candidate (rows ascend with wave, then lane group)
          cand_v[(wave * 4 + lg) * 32 + 16 * n + lc] = best;
          cand_i[(wave * 4 + lg) * 32 + 16 * n + lc] = best_i;
        }
      }
      KM_MARK(3)
      wg_barrier();
      KM_MARK(4)
      // every wave rebuilds the labels of the 32 pixels (lanes 0..31) in registers
      {
        // all candidate reads are issued before the first compare (one LDS round trip)
        float cv[4 * MT16];
        int ci[4 * MT16];
        const int px = lane & 31;
#pragma unroll
        for (int c2 = 0; c2 < 4 * MT16; ++c2) {
          cv[c2] = cand_v[c2 * 32 + px];
          ci[c2] = cand_i[c2 * 32 + px];
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int c2 = 0; c2 < 4 * MT16; ++c2)             // ascending prototype rows: ties -> lowest
          if (cv[c2] > bv) { bv = cv[c2]; bi = ci[c2]; }
        mylab = lane < nrows ? bi : -1;
      }
    }

    KM_MARK(5)
    if (a.do_accum) {
      // ================= M-step =================
      half8 oh[MT16], ol[MT16];
      {
        typedef int int4v __attribute__((ext_vector_type(4)));
        if (lane < 32) labw[wave * 32 + lane] = mylab;   // same wave reads it back: LDS is in order
        const int4v l0 = *reinterpret_cast<const int4v*>(labw + wave * 32 + 8 * lg);
        const int4v l1 = *reinterpret_cast<const int4v*>(labw + wave * 32 + 8 * lg + 4);
        const half8 scale = {(_Float16)kSplitInv, (_Float16)kSplitInv, (_Float16)kSplitInv,
                             (_Float16)kSplitInv, (_Float16)kSplitInv, (_Float16)kSplitInv,
                             (_Float16)kSplitInv, (_Float16)kSplitInv};
#pragma unroll
        for (int q = 0; q < MT16; ++q) {
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int lb = i < 4 ? l0[i] : l1[i - 4];
            oh[q][i] = lb == 16 * q + lc ? (_Float16)1.0f : (_Float16)0.0f;
          }
          ol[q] = oh[q] * scale;
        }
      }
#pragma unroll
      for (int i = 0; i < NSTW; ++i) {
        const int st = wave + 4 * i;                    // 32-channel super tile = k-step st of conv
        if (Q % 4 == 0 || st < Q) {                     // wave-uniform
          // A operand = X^T (rows = channels, k = pixels) straight out of the channel-major
          // fragment blocks with the LDS transpose read: a 16-lane group reads a
          // [4 pixel][16 channel] sub-block, lane lc receives channel 16*ct + lc of 4 pixels;
          // two reads give the 8 pixels (8*lg .. 8*lg+7) of this lane's k-group
          typedef short short4v __attribute__((vector_size(8)));
          typedef __attribute__((address_space(3))) short4v* trptr_t;
          const unsigned char* cp = conv + (size_t)st * 4096 + troff;
          union { short4v p[2]; half8 h; } xa[2][2];      // [channel tile][hi|lo]
#pragma unroll
          for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int part = 0; part < 2; ++part)
#pragma unroll
              for (int hh = 0; hh < 2; ++hh)
                xa[ct][part].p[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                    (trptr_t)(cp + ct * 256 + part * 1024 + hh * 512));
#pragma unroll
          for (int q = 0; q < MT16; ++q) macc[2 * i][q] = mfma16(xa[0][0].h, oh[q], macc[2 * i][q]);
#pragma unroll
          for (int q = 0; q < MT16; ++q) macc[2 * i + 1][q] = mfma16(xa[1][0].h, oh[q], macc[2 * i + 1][q]);
#pragma unroll
          for (int q = 0; q < MT16; ++q) macc[2 * i][q] = mfma16(xa[0][1].h, ol[q], macc[2 * i][q]);
#pragma unroll
          for (int q = 0; q < MT16; ++q) macc[2 * i + 1][q] = mfma16(xa[1][1].h, ol[q], macc[2 * i + 1][q]);
        }
      }
      if (TAIL && wave == 3) {                          // location channels: rows 0,1 of k-step Q
        // same transpose read on the plain lane-linear location block (16-B slot = pixel,
        // channel group 0 only; groups 1..3 of the block stay zero)
        typedef short short4v __attribute__((vector_size(8)));
        typedef __attribute__((address_space(3))) short4v* trptr_t;
        const unsigned char* cp = conv + (size_t)Q * 4096 + (lg >> 1) * 2048 +
                                  (size_t)(((lc >> 1) & 1) * 16 + 8 * (lg & 1) + (lc >> 2)) * 16 + 8 * (lc & 1);
        union { short4v p[2]; half8 h; } xt[2];           // [hi|lo]
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            xt[part].p[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(cp + part * 1024 + hh * 64));
#pragma unroll
        for (int q = 0; q < MT16; ++q) macc[2 * NSTW][q] = mfma16(xt[0].h, oh[q], macc[2 * NSTW][q]);
#pragma unroll
        for (int q = 0; q < MT16; ++q) macc[2 * NSTW][q] = mfma16(xt[1].h, ol[q], macc[2 * NSTW][q]);
      }
    }
    // labels leave after the M-step: by then the tile copy issued above has drained from
    // the CU's vector-memory queue and the store does not stall behind it
    if (a.do_assign && wave == 1 && lane < nrows) label_store(a, seg0 + t * TPW + lane, mylab);
    KM_MARK(6)
  }

  if (a.do_accum) {
    float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
#pragma unroll
    for (int i = 0; i < NSTW; ++i) {
      const int st = wave + 4 * i;
      if (st < Q) {
#pragma unroll
        for (int par = 0; par < 2; ++par)
#pragma unroll
          for (int q = 0; q < MT16; ++q) {
            const int c = 16 * q + lc;
            // rows 4*lg .. 4*lg+3 of channel tile par: two 8-byte stores (D is even)
            const int d = 32 * st + 16 * par + 4 * lg;
            if (c < K) {
              float2* dst = reinterpret_cast<float2*>(slab + (size_t)c * D + d);
              dst[0] = float2{macc[2 * i + par][q][0], macc[2 * i + par][q][1]};
              dst[1] = float2{macc[2 * i + par][q][2], macc[2 * i + par][q][3]};
            }
          }
      }
    }
    if (TAIL && wave == 3) {
#pragma unroll
      for (int q = 0; q < MT16; ++q) {
        const int c = 16 * q + lc;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int d = 32 * Q + 4 * lg + r;
          if (c < K && d < D) slab[(size_t)c * D + d] = macc[2 * NSTW][q][r];
        }
      }
    }
  }
  KM_TRACE_DRAIN
  KM_MARK(1)                              // (PRE builds: slot 1 = slab write-out, incl. drain)
  KM_TRACE_STORE
  KM_CLOCK_END
}

// ===========================================================================
// Many-cluster variant of kmeans_pass16: 64 < K <= 256 (e.g. the 12x12 grid of the
// DensePose recipe and of full-resolution inference), small D.  Same tile pipeline on
// pre-converted tiles; differences:
//   * wave w owns the prototype tiles {w, w+4, ...} (MTW of them) in BOTH steps: the
//     E-step walks them one after the other against the same LDS operands and keeps a
//     running per-lane best, so the candidate table stays 4 KB whatever K is;
//   * the M-step is split by cluster tile, not by channel: every wave reads all channel
//     tiles of X^T (transpose reads) and multiplies them with the one-hot of ITS
//     clusters only -> NDT * MTW accumulator tiles per wave and balanced MFMA work.
// ===========================================================================
template <int MTW, int Q, int TAIL>
__global__ __launch_bounds__(256, pass16k_wg_per_cu(MTW, Q, TAIL)) void kmeans_pass16k(PassArgs a) {
  constexpr int TPW = 32;
  constexpr int QE = Q + TAIL;
  constexpr int NDT = 2 * Q + TAIL;              // 16-channel tiles
  constexpr int CONV = QE * 4096;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
  const int D = a.D, K = a.K;
  const int MT16 = a.kpad >> 4;
  const int img = blockIdx.y, g = blockIdx.x;

  unsigned char* conv0 = lds;
  size_t off = 2 * (size_t)CONV;
  float* cand_v = reinterpret_cast<float*>(lds + off);        // [4 waves][4 lane groups][32]
  int* cand_i = reinterpret_cast<int*>(lds + off + 16 * 32 * 4);
  off += 16 * 32 * 8;
  int* labw = reinterpret_cast<int*>(lds + off);              // [4 waves][32]
  off += 4 * 32 * 4;
  int* labin = reinterpret_cast<int*>(lds + off);             // [2][256]

  KM_CLOCK_BEGIN
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  int64_t t_begin, t_end, t_step = 1;
  tile_range(T, g, a.G, t_begin, t_end);
  if (a.strided) { t_begin = g; t_end = T; t_step = a.G; }      // (as kmeans_pass16)
  if (t_begin >= t_end) {
    if (a.do_accum) {
      float* z = a.slabs + ((size_t)img * a.G + g) * K * D;
      for (int i = tid; i < K * D; i += 256) z[i] = 0.f;
    }
    KM_CLOCK_END
    return;
  }
  const int eoff = frag_slot(lc, lg) * 16;
  const int troff = (lg >> 1) * 2048 + frag_slot(8 * (lg & 1) + (lc >> 2), (lc >> 1) & 1) * 16 + 8 * (lc & 1);
  const int troff_tail = (lg >> 1) * 2048 + (((lc >> 1) & 1) * 16 + 8 * (lg & 1) + (lc >> 2)) * 16 + 8 * (lc & 1);

  half8 ah[MTW][QE], al[MTW][QE];
  if (TAIL) {
    for (int i = tid; i < 1024; i += 256) {
      reinterpret_cast<float*>(conv0 + (size_t)Q * 4096)[i] = 0.f;
      reinterpret_cast<float*>(conv0 + CONV + (size_t)Q * 4096)[i] = 0.f;
    }
    wg_barrier();
  }
  float4a macc[NDT][MTW];
#pragma unroll
  for (int i = 0; i < NDT; ++i)
#pragma unroll
    for (int j = 0; j < MTW; ++j) macc[i][j] = float4a{0.f, 0.f, 0.f, 0.f};

  const int64_t tile0 = pre_tile0(seg0, img);
  auto tile_issue = [&](int64_t t, int slot) {
    const unsigned char* tb = a.xc + (size_t)(tile0 + t) * pre_tile_bytes(Q, TAIL) + 16 * lane;
    unsigned char* dst0 = conv0 + (size_t)slot * CONV;
    for (int b = wave; b < 4 * Q; b += 4)
      __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)b * 1024), (lptr_t)(dst0 + b * 1024), 16, 0, 0);
    if (TAIL && lane < 16)
      __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)Q * 4096 + wave * 256),
                                       (lptr_t)(dst0 + (4 * Q + wave) * 1024), 16, 0, 0);
    if (!a.do_assign) {
      const int64_t p = min(seg0 + t * TPW + min(wave * 64 + lane, TPW - 1), a.P - 1);
      __builtin_amdgcn_global_load_lds(label_src(a, p), (lptr_t)(labin + slot * 256 + wave * 64), 4, 0, 0);
    }
  };

  tile_issue(t_begin, 0);
  // prototype rows -> registers while the first tile is in flight
  if (a.do_assign) {
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int mt = min(wave + 4 * j, MT16 - 1);
#pragma unroll
      for (int s = 0; s < QE; ++s) {
        const size_t o = ((size_t)img * a.kpad + 16 * mt + lc) * a.dpad + 32 * s + 8 * lg;
        ah[j][s] = *reinterpret_cast<const half8*>(a.cent_h + o);
        al[j][s] = *reinterpret_cast<const half8*>(a.cent_l + o);
      }
    }
  }
  for (int64_t t = t_begin; t < t_end; t += t_step) {
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int slot = (int)(((t - t_begin) / t_step) & 1);
    unsigned char* conv = conv0 + (size_t)slot * CONV;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();
    if (t + t_step < t_end) tile_issue(t + t_step, slot ^ 1);
    int mylab = -1;
    if (!a.do_assign && lane < 32) mylab = lane < nrows ? labin[slot * 256 + lane] : -1;

    if (a.do_assign) {
      // ================= E-step: running best over this wave's prototype tiles =========
      float best[2] = {-INFINITY, -INFINITY};
      int best_i[2] = {0x7fffffff, 0x7fffffff};
      const unsigned cbase = (unsigned)(size_t)(lptr_t)(conv);
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        const int mt = wave + 4 * j;
        if (mt < MT16) {                                   // wave-uniform
          float4a eh[2], ex[2], ey[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) { eh[n] = float4a{0.f, 0.f, 0.f, 0.f}; ex[n] = eh[n]; ey[n] = eh[n]; }
          half8 bh[2], bl[2];
          auto load_b = [&](int s, int n) {
            const unsigned addr = cbase + (unsigned)((s * 2 + n) * 2) * 1024u +
                                  (unsigned)(s < Q ? eoff : lane * 16);
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024"
                         : "=&v"(bh[n]), "=&v"(bl[n]) : "v"(addr));
          };
          load_b(0, 0);
          load_b(0, 1);
#pragma unroll
          for (int s = 0; s < QE; ++s) {
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              if (s + 1 < QE || n == 0)
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bh[n]), "+v"(bl[n]));
              else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bh[n]), "+v"(bl[n]));
              eh[n] = mfma16(ah[j][s], bh[n], eh[n]);
              ex[n] = mfma16(ah[j][s], bl[n], ex[n]);
              ey[n] = mfma16(al[j][s], bh[n], ey[n]);
              __builtin_amdgcn_sched_barrier(0);
              if (s + 1 < QE) load_b(s + 1, n);
              __builtin_amdgcn_sched_barrier(0);
            }
          }
#pragma unroll
          for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int c = 16 * mt + 4 * lg + r;          // ascending in j, then r
              const float sdot = eh[n][r] + (ex[n][r] + ey[n][r]) * kSplitInv;
              if (c < K && sdot > best[n]) { best[n] = sdot; best_i[n] = c; }
            }
          }
        }
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        cand_v[(wave * 4 + lg) * 32 + 16 * n + lc] = best[n];
        cand_i[(wave * 4 + lg) * 32 + 16 * n + lc] = best_i[n];
      }
      wg_barrier();
      {
        float cv[16];
        int ci[16];
        const int px = lane & 31;
#pragma unroll
        for (int c2 = 0; c2 < 16; ++c2) {
          cv[c2] = cand_v[c2 * 32 + px];
          ci[c2] = cand_i[c2 * 32 + px];
        }
        float bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int c2 = 0; c2 < 16; ++c2)      // candidates are not ordered by index: ties -> lowest
          if (cv[c2] > bv || (cv[c2] == bv && ci[c2] < bi)) { bv = cv[c2]; bi = ci[c2]; }
        mylab = lane < nrows ? bi : -1;
      }
    }

    if (a.do_accum) {
      // ================= M-step: all channel tiles x this wave's cluster tiles ==========
      typedef int int4v __attribute__((ext_vector_type(4)));
      typedef short short4v __attribute__((vector_size(8)));
      typedef __attribute__((address_space(3))) short4v* trptr_t;
      if (lane < 32) labw[wave * 32 + lane] = mylab;
      const int4v l0 = *reinterpret_cast<const int4v*>(labw + wave * 32 + 8 * lg);
      const int4v l1 = *reinterpret_cast<const int4v*>(labw + wave * 32 + 8 * lg + 4);
      union { short4v p[2]; half8 h; } xa[NDT][2];         // [channel tile][hi|lo]
#pragma unroll
      for (int dt = 0; dt < 2 * Q; ++dt) {
        const unsigned char* cp = conv + (size_t)(dt >> 1) * 4096 + troff + (dt & 1) * 256;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            xa[dt][part].p[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(cp + part * 1024 + hh * 512));
      }
      if (TAIL) {
        const unsigned char* cp = conv + (size_t)Q * 4096 + troff_tail;
#pragma unroll
        for (int part = 0; part < 2; ++part)
#pragma unroll
          for (int hh = 0; hh < 2; ++hh)
            xa[2 * Q][part].p[hh] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(cp + part * 1024 + hh * 64));
      }
      const half8 scale = {(_Float16)kSplitInv, (_Float16)kSplitInv, (_Float16)kSplitInv,
                           (_Float16)kSplitInv, (_Float16)kSplitInv, (_Float16)kSplitInv,
                           (_Float16)kSplitInv, (_Float16)kSplitInv};
#pragma unroll
      for (int j = 0; j < MTW; ++j) {
        const int mt = wave + 4 * j;
        if (mt < MT16) {
          half8 oh, ol;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int lb = i < 4 ? l0[i] : l1[i - 4];
            oh[i] = lb == 16 * mt + lc ? (_Float16)1.0f : (_Float16)0.0f;
          }
          ol = oh * scale;
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) macc[dt][j] = mfma16(xa[dt][0].h, oh, macc[dt][j]);
#pragma unroll
          for (int dt = 0; dt < NDT; ++dt) macc[dt][j] = mfma16(xa[dt][1].h, ol, macc[dt][j]);
        }
      }
    }
    if (a.do_assign && wave == 1 && lane < nrows) label_store(a, seg0 + t * TPW + lane, mylab);
  }

  if (a.do_accum) {
    float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
#pragma unroll
    for (int j = 0; j < MTW; ++j) {
      const int mt = wave + 4 * j;
      const int c = 16 * mt + lc;
      if (mt < MT16 && c < K) {
#pragma unroll
        for (int dt = 0; dt < 2 * Q; ++dt) {
          float2* dst = reinterpret_cast<float2*>(slab + (size_t)c * D + 16 * dt + 4 * lg);
          dst[0] = float2{macc[dt][j][0], macc[dt][j][1]};
          dst[1] = float2{macc[dt][j][2], macc[dt][j][3]};
        }
        if (TAIL) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int d = 32 * Q + 4 * lg + r;
            if (d < D) slab[(size_t)c * D + d] = macc[2 * Q][j][r];
          }
        }
      }
    }
  }
  KM_CLOCK_END
}

// slabs -> prototypes, two small kernels:
//  (1) kmeans_reduce_slabs: grid (ceil(D/64), K, n_img) x 1024 threads = 64 channels x 16
//      slab groups; every thread adds G/16 slabs, the 16 partials are combined in
//      group order (fixed summation order -> deterministic); also emits the partial
//      sum of squares of its 64 channels;
//  (2) kmeans_normalize: grid (K, n_img): L2 norm from the partials (fixed order),
//      zero sum -> zero prototype (0 / 1e-12), writes fp32 + split-f16 forms.
// sum of p[g * stride], g = g0 .. g1 - 1, added in ascending g (fixed order); the loads of a batch are all in
// flight before the first add
template <int N>
__device__ __forceinline__ void sum_slabs_batch(const float* __restrict__ p, size_t stride, int& g, int g1, float& v) {
  for (; g + N <= g1; g += N) {
    float t[N];
#pragma unroll
    for (int u = 0; u < N; ++u) t[u] = p[(size_t)(g + u) * stride];
#pragma unroll
    for (int u = 0; u < N; ++u) v += t[u];
  }
}
__device__ __forceinline__ float sum_slabs(const float* __restrict__ p, size_t stride, int g0, int g1) {
  float v = 0.f;
  int g = g0;
  sum_slabs_batch<32>(p, stride, g, g1, v);
  sum_slabs_batch<16>(p, stride, g, g1, v);
  sum_slabs_batch<8>(p, stride, g, g1, v);
  sum_slabs_batch<4>(p, stride, g, g1, v);
  sum_slabs_batch<2>(p, stride, g, g1, v);
  sum_slabs_batch<1>(p, stride, g, g1, v);
  return v;
}

__global__ __launch_bounds__(1024) void kmeans_reduce_slabs(const float* __restrict__ slabs,
                                                            int G, int K, int D,
                                                            float* __restrict__ sums,
                                                            float* __restrict__ ssq) {
  __shared__ float part[16][64];
  const int chunk = blockIdx.x, k = blockIdx.y, img = blockIdx.z;
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const int d = chunk * 64 + col;
  const int g0 = (G * grp) / 16, g1 = (G * (grp + 1)) / 16;
  float v = 0.f;
  if (d < D) {
    const float* p = slabs + ((size_t)img * G * K + k) * D + d;
    // all of a thread's slab reads in flight together (G / 16 = 32 at the roofline shape): the
    // kernel is a latency chain, not a bandwidth problem (19 MB)
    // (every batch size is written out: a loop with a run-time trip count waits for each load before it issues
    // the next one -- 16 serial round trips at G = 256, where the 32-wide batch never ran: 9.4 us for 9.5 MB)
    v = sum_slabs(p, (size_t)K * D, g0, g1);
  }
  part[grp][col] = v;
  __syncthreads();
  if (grp == 0) {
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) t += part[i][col];
    if (d < D) sums[((size_t)img * K + k) * D + d] = t;
    const float sq = wave_sum(d < D ? t * t : 0.f);
    if (col == 0) ssq[((size_t)img * K + k) * gridDim.x + chunk] = sq;
  }
}

// q[0] + q[1] + ... in that order, the loads in flight together (batches of 8; a padding 0.f leaves the sum as it is)
__device__ __forceinline__ float sum_chunks(const float* __restrict__ q, int n) {
  float t = 0.f;
  for (int i0 = 0; i0 < n; i0 += 8) {
    float r[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) r[u] = i0 + u < n ? q[i0 + u] : 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) t += r[u];
  }
  return t;
}

__global__ __launch_bounds__(256) void kmeans_normalize(const float* __restrict__ sums,
                                                        const float* __restrict__ ssq,
                                                        int nchunk, int K, int D, int kpad,
                                                        int dpad, int normalize, int frag,
                                                        float* __restrict__ cent,
                                                        _Float16* __restrict__ cent_h,
                                                        _Float16* __restrict__ cent_l) {
  // grid (kpad, n_img): rows k >= K and channels d >= D of the split-f16 arrays are padding
  // and are written as zeros here (the workspace is the caller's: nothing is assumed about it).
  // frag: the split-f16 arrays in the A-fragment order of kmeans_pass64 -- 1-KB blocks
  // [prototype tile k/16][k-step d/32], lane (d%32/8)*16 + k%16 holds 8 consecutive channels -- so that
  // a wave loads a fragment with one coalesced 1-KB read
  const int k = blockIdx.x, img = blockIdx.y;
  auto at = [&](int d) -> size_t {
    if (frag)
      return (((size_t)img * (kpad >> 4) + (k >> 4)) * (dpad >> 5) + (d >> 5)) * 512 +
             (size_t)((((d & 31) >> 3) * 16 + (k & 15)) * 8 + (d & 7));
    return ((size_t)img * kpad + k) * dpad + d;
  };
  if (k >= K) {
    if (cent_h)
      for (int d = threadIdx.x; d < dpad; d += 256) {
        const size_t o = at(d);
        cent_h[o] = (_Float16)0.f;
        cent_l[o] = (_Float16)0.f;
      }
    return;
  }
  // (the row's sums are requested BEFORE the partial norms are added up: one memory round trip for both instead of
  // two in a row -- the kernel is nothing but that latency; the first 9 x 256 channels of a row, i.e. all of it for every
  // shape the tile kernels take)
  const int dmax = cent_h ? max(D, dpad) : D;
  constexpr int kMaxPerThread = 9;
  float raw[kMaxPerThread];
#pragma unroll
  for (int i = 0; i < kMaxPerThread; ++i) {
    const int d = threadIdx.x + 256 * i;
    raw[i] = d < D ? sums[((size_t)img * K + k) * D + d] : 0.f;
  }
  float dn = 1.f;
  if (normalize) {
    const float t = sum_chunks(ssq + ((size_t)img * K + k) * nchunk, nchunk);
    const float n = sqrtf(t);
    dn = n >= kEps ? n : kEps;
  }
#pragma unroll
  for (int i = 0; i < kMaxPerThread; ++i) {
    const int d = threadIdx.x + 256 * i;
    if (d >= dmax) break;
    const float v = d < D ? raw[i] / dn : 0.f;
    if (cent && d < D) cent[((size_t)img * K + k) * D + d] = v;
    if (cent_h && d < dpad) {          // (v3 keeps the 2 tail channels in fp32 only)
      _Float16 h, l;
      split_f16(v, h, l);
      const size_t o = at(d);
      cent_h[o] = h;
      cent_l[o] = l;
    }
  }
  for (int d = threadIdx.x + 256 * kMaxPerThread; d < dmax; d += 256) {       // (rows wider than 2304: not a tile-kernel shape today)
    const float v = d < D ? sums[((size_t)img * K + k) * D + d] / dn : 0.f;
    if (cent && d < D) cent[((size_t)img * K + k) * D + d] = v;
    if (cent_h && d < dpad) {
      _Float16 h, l;
      split_f16(v, h, l);
      const size_t o = at(d);
      cent_h[o] = h;
      cent_l[o] = l;
    }
  }
}

// slabs -> prototypes in ONE launch (the finalize of every M-step of a k-means call): grid (kpad, n_img) x 1024
// threads, D even and <= 2048.  Block k sums its cluster's row over the G slabs -- wave w owns the 128-channel
// chunk w % nchunk (one 8-byte load per lane and slab) and the slab group w / nchunk; ALL loads of a thread are in
// flight together (up to 56: one memory round trip, the kernel is a latency chain, not a bandwidth problem) --
// combines the groups in a fixed order, takes the L2 norm (fixed order) and writes the normalised row as fp32 and
// as split f16 (row-major or in the A-fragment order of kmeans_pass64, see kmeans_normalize).  Same arithmetic as
// kmeans_reduce_slabs + kmeans_normalize up to the order in which the slab groups are combined.
template <int NL>
__global__ __launch_bounds__(1024) void kmeans_finalize(const float* __restrict__ slabs, int G, int K, int D,
                                                        int kpad, int dpad, int frag,
                                                        float* __restrict__ cent,
                                                        _Float16* __restrict__ cent_h,
                                                        _Float16* __restrict__ cent_l) {
  __shared__ float2 part[16][64];
  __shared__ float sq[16];
  const int k = blockIdx.x, img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto at = [&](int d) -> size_t {
    if (frag)
      return (((size_t)img * (kpad >> 4) + (k >> 4)) * (dpad >> 5) + (d >> 5)) * 512 +
             (size_t)((((d & 31) >> 3) * 16 + (k & 15)) * 8 + (d & 7));
    return ((size_t)img * kpad + k) * dpad + d;
  };
  if (k >= K) {
    if (cent_h)
      for (int d = tid; d < dpad; d += 1024) {
        const size_t o = at(d);
        cent_h[o] = (_Float16)0.f;
        cent_l[o] = (_Float16)0.f;
      }
    return;
  }
  const int nchunk = (D + 127) >> 7;                // <= 16 (D <= 2048)
  const int ngrp = 16 / nchunk;
  const int chunk = wave % nchunk, grp = wave / nchunk;
  const int d = chunk * 128 + 2 * lane;
  float2 v = {0.f, 0.f};
  if (grp < ngrp && d < D) {
    const int g0 = (G * grp) / ngrp, g1 = (G * (grp + 1)) / ngrp;     // <= NL slabs
    const float* p = slabs + ((size_t)img * G * K + k) * D + d;
    const size_t stride = (size_t)K * D;
    float2 t[NL];
#pragma unroll
    for (int u = 0; u < NL; ++u)
      t[u] = g0 + u < g1 ? *reinterpret_cast<const float2*>(p + (size_t)(g0 + u) * stride) : float2{0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NL; ++u) { v.x += t[u].x; v.y += t[u].y; }
  }
  part[wave][lane] = v;
  __syncthreads();
  // the first nchunk waves finish their chunk: groups in order, then the chunk's sum of squares
  float2 t = {0.f, 0.f};
  if (wave < nchunk) {
    for (int i = 0; i < ngrp; ++i) { const float2 q = part[wave + i * nchunk][lane]; t.x += q.x; t.y += q.y; }
    const float s2 = wave_sum(wave * 128 + 2 * lane < D ? t.x * t.x + t.y * t.y : 0.f);
    if (lane == 0) sq[wave] = s2;
  }
  __syncthreads();
  if (wave < nchunk) {
    float n2 = 0.f;
    for (int i = 0; i < nchunk; ++i) n2 += sq[i];
    const float n = sqrtf(n2);
    const float dn = n >= kEps ? n : kEps;
    const int dd = wave * 128 + 2 * lane;
    if (dd < D) {
      const float r[2] = {t.x / dn, t.y / dn};
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        if (cent) cent[((size_t)img * K + k) * D + dd + e] = r[e];
        if (cent_h && dd + e < dpad) {
          _Float16 h, l;
          split_f16(r[e], h, l);
          const size_t o = at(dd + e);
          cent_h[o] = h;
          cent_l[o] = l;
        }
      }
    }
  }
  // padding channels D .. dpad-1 of the split arrays
  if (cent_h)
    for (int dd = D + tid; dd < dpad; dd += 1024) {
      const size_t o = at(dd);
      cent_h[o] = (_Float16)0.f;
      cent_l[o] = (_Float16)0.f;
    }
}

// slabs per thread of kmeans_finalize (0: shape not covered)
inline int finalize_loads(int G, int D) {
  if ((D & 1) || D > 2048) return 0;
  const int nchunk = (D + 127) >> 7, ngrp = 16 / nchunk;
  const int per = (G + ngrp - 1) / ngrp + 1;
  return per <= 8 ? 8 : per <= 16 ? 16 : per <= 32 ? 32 : per <= 56 ? 56 : 0;
}

__global__ void fix_to_f32(const long long* __restrict__ acc, int64_t n, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (float)((double)acc[i] * kDetFixInv);
}

__global__ void labels_i64_to_i32(const int64_t* in, int32_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int32_t)in[i];
}
__global__ void labels_i32_to_i64(const int32_t* in, int64_t* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (int64_t)in[i];
}

// ------------------------- generic path ----------------------------------
// one wave handles 4 pixels at a time; lanes split the channels.
template <int NC>
__global__ __launch_bounds__(256) void generic_assign(const float* __restrict__ x, int64_t P,
                                                      int D, const int64_t* __restrict__ seg_off,
                                                      int n_img, int K,
                                                      const float* __restrict__ cent,
                                                      int32_t* __restrict__ labels) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t p0 = w * 4;
  if (p0 >= P) return;
  float xv[4][NC];
  int im[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t p = min(p0 + q, P - 1);
    int lo = 0, hi = n_img;                     // image of pixel p (binary search)
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= p) lo = mid; else hi = mid; }
    im[q] = lo;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int d = lane + 64 * c;
      xv[q][c] = d < D ? x[(size_t)p * D + d] : 0.f;
    }
  }
  float best[4];
  int bi[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { best[q] = -INFINITY; bi[q] = 0; }
  const bool same = (im[0] == im[3]);
  for (int k = 0; k < K; ++k) {
    float dot[4] = {0.f, 0.f, 0.f, 0.f};
    if (same) {
      const float* cr = cent + ((size_t)im[0] * K + k) * D;
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        const int d = lane + 64 * c;
        const float cv = d < D ? cr[d] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) dot[q] += xv[q][c] * cv;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float* cr = cent + ((size_t)im[q] * K + k) * D;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
          const int d = lane + 64 * c;
          dot[q] += xv[q][c] * (d < D ? cr[d] : 0.f);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float s = wave_sum(dot[q]);
      if (s > best[q]) { best[q] = s; bi[q] = k; }
    }
  }
  if (lane < 4 && p0 + lane < P) {
    int v = bi[0];
    if (lane == 1) v = bi[1];
    if (lane == 2) v = bi[2];
    if (lane == 3) v = bi[3];
    labels[p0 + lane] = v;
  }
}

// ids[p] = img(p) * K + label[p]
__global__ void generic_ids(const int32_t* labels, const int64_t* seg_off, int n_img, int K,
                            int64_t P, int64_t* ids) {
  const int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  int lo = 0, hi = n_img;
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (seg_off[mid] <= p) lo = mid; else hi = mid; }
  const int l = labels[p];
  ids[p] = (l >= 0 && l < K) ? (int64_t)lo * K + l : -1;
}

// ------------------------- host side --------------------------------------
struct Plan {
  bool fast;
  bool v3;                     // kmeans_pass16 (16x16x32 tiles, in-LDS split)
  int NT, KS, KSPLIT, G, kpad, dpad, nvt;
  int MT16, Q, TAIL;
  bool pre;                    // v3 on pre-converted tiles
  bool v3k;                    // kmeans_pass16k (64 < K <= 256, small D, pre-converted tiles)
  bool v4k;                    // kmeans_assign64k + kmeans_accum64k (64 < K <= 144, D >= 128, pre-converted tiles)
  int MTW;
  size_t lds;
};

// shapes kmeans_pass16 covers: D = 32*Q + tail, Q in {1, 2, 4, 8}, K <= 64.  tail in {0, 2}
// (embedding, + (y, x) location) runs everywhere; other tails up to 8 (location + colours
// of the DensePose recipe: 5) only on pre-converted tiles, i.e. for >= 3 passes
inline bool v3_shape(int D, int K, bool pre = false) {
  const int q = D / 32, tl = D - 32 * q;
  const bool tail_ok = tl == 0 || tl == 2 || (pre && tl <= 8);
  return K >= 1 && K <= 64 && tail_ok && (q == 1 || q == 2 || q == 4 || q == 8);
}
// shapes kmeans_pass16k covers: 64 < K <= 256 with the wave's prototype fragments and
// M-step accumulators inside the register budget: MTW * (8*QE + 4*NDT) <= 176
inline bool v3k_shape(int D, int K) {
  const int q = D / 32, tl = D - 32 * q;
  if (K <= 64 || K > 256 || tl > 8 || !(q == 1 || q == 2 || q == 4)) return false;
  const int mtw = ((K + 15) / 16 + 3) / 4, qe = q + (tl ? 1 : 0), ndt = 2 * q + (tl ? 1 : 0);
  return mtw * (8 * qe + 4 * ndt) <= 176;
}

Plan make_plan(const float* x, int64_t P, int D, int K, int n_img, int64_t max_seg_len,
               int flags, bool want_pre) {
  Plan pl{};
  pl.fast = false;
  if (flags & SPML_KMEANS_FORCE_GENERIC) return pl;
  if (reinterpret_cast<uintptr_t>(x) & 15) return pl;
  if (P * (int64_t)D * 4 < 16) return pl;
  if (want_pre && !(flags & (SPML_KMEANS_NO_PRECONVERT | SPML_KMEANS_FORCE_V2)) && v3k_shape(D, K)) {
    const int q = D / 32, tl = D - 32 * q;
    pl.fast = true; pl.v3k = true; pl.pre = true;
    pl.Q = q; pl.TAIL = tl ? 1 : 0;
    pl.MT16 = (K + 15) / 16; pl.MTW = (pl.MT16 + 3) / 4;
    pl.kpad = 16 * pl.MT16; pl.dpad = 32 * (q + pl.TAIL);
    pl.lds = pass16_lds_bytes(D, true);
    const int64_t tiles = (max_seg_len + 31) / 32;
    int64_t gI = (256 * pass16k_wg_per_cu(pl.MTW, pl.Q, pl.TAIL) + n_img - 1) / n_img;
    if (gI > tiles) gI = tiles;
    if (gI < 1) gI = 1;
    pl.G = (int)gI;
    return pl;
  }
  if (want_pre && !(flags & (SPML_KMEANS_NO_PRECONVERT | SPML_KMEANS_FORCE_V2 | SPML_KMEANS_NO_V4K)) &&
      assign64k_shape(D, K)) {
    // wide rows, 64 < K <= 144 (the 12 x 12 grid at 513 x 513 x 258): assign and accumulate as two passes over
    // the pre-converted 64-pixel tiles, one workgroup per CU
    const int q = D / 32, tl = D - 32 * q;
    pl.fast = true; pl.v4k = true; pl.pre = true;
    pl.Q = q; pl.TAIL = tl ? 1 : 0;
    pl.kpad = assign64k_kpad(K); pl.MT16 = pl.kpad / 16;
    pl.dpad = 32 * (q + pl.TAIL);
    pl.lds = p64_lds(q, pl.TAIL);
    const int64_t tiles = (max_seg_len + 63) / 64;
    int64_t gI = (256 + n_img - 1) / n_img;
    if (gI > tiles) gI = tiles;
    if (gI < 1) gI = 1;
    pl.G = (int)gI;
    return pl;
  }
  {
    // v3: D = 32*Q + tail, Q in {1, 2, 4, 8}
    const int q = D / 32, tl = D - 32 * q;
    const bool pre = want_pre && !(flags & SPML_KMEANS_NO_PRECONVERT);
    if (!(flags & SPML_KMEANS_FORCE_V2) && v3_shape(D, K, pre)) {
      pl.fast = true; pl.v3 = true;
      pl.pre = pre;
      pl.Q = q; pl.MT16 = (K + 15) / 16;
      pl.kpad = 16 * pl.MT16; pl.dpad = 32 * (q + (tl ? 1 : 0));
      pl.TAIL = tl ? 1 : 0;
      pl.nvt = pass_nvt(D, 4);
      pl.lds = pass16_lds_bytes(D, pl.pre);
      const int64_t tiles = (max_seg_len + 31) / 32;
      int per_cu = (int)(160 * 1024 / pl.lds);
      if (per_cu > pass16_wg_per_cu(pl.MT16, pl.Q)) per_cu = pass16_wg_per_cu(pl.MT16, pl.Q);
      if (per_cu < 1) per_cu = 1;
      int64_t gI = (256 * per_cu + n_img - 1) / n_img;
      if (gI > tiles) gI = tiles;
      if (gI < 1) gI = 1;
      pl.G = (int)gI;
      return pl;
    }
  }
  if (K < 1 || K > 64 || (D & 1) || D < 2) return pl;
  const int steps = (D + 15) / 16;
  int ksplit = 0, ks = 0;
  for (int s : {1, 2, 4}) {
    const int need = (steps + s - 1) / s;
    if (need <= kKsMax) { ksplit = s; ks = need; break; }
  }
  if (!ksplit) return pl;
  ks = ks <= 2 ? 2 : (ks <= 3 ? 3 : 5);
  const int nt = K <= 32 ? 1 : 2;
  const size_t lds = pass_lds_bytes(D, nt, ksplit);
  if (lds > 160 * 1024) return pl;
  pl.fast = true;
  pl.NT = nt; pl.KS = ks; pl.KSPLIT = ksplit; pl.lds = lds;
  pl.kpad = 32 * nt;
  pl.dpad = 16 * ks * ksplit;
  pl.nvt = pass_nvt(D, ksplit);
  const int tpw = 32 * (4 / ksplit);
  const int64_t tiles = (max_seg_len + tpw - 1) / tpw;
  // ~2 workgroups per CU across all images (1 when LDS allows only one)
  const int per_cu = lds > 80 * 1024 ? 1 : 2;
  int64_t gI = (256 * per_cu + n_img - 1) / n_img;
  if (gI > tiles) gI = tiles;
  if (gI < 1) gI = 1;
  pl.G = (int)gI;
  return pl;
}

struct WsLayout {
  size_t lab32, cent_h, cent_l, cent_f, slabs, ids, sums, ssq, det64, xc, big, total;
};

WsLayout ws_layout(int64_t P, int D, int K, int n_img, int64_t max_seg_len) {
  // sized for the worst of the fast / generic plans
  WsLayout w{};
  size_t o = 0;
  w.lab32 = o; o = align_up(o + (size_t)P * 4, 256);
  const size_t kpad = (v3k_shape(D, K) || assign64k_shape(D, K)) ? 256 : 64, dpad = 320;
  w.cent_h = o; o = align_up(o + (size_t)n_img * kpad * dpad * 2, 256);
  w.cent_l = o; o = align_up(o + (size_t)n_img * kpad * dpad * 2, 256);
  w.cent_f = o; o = align_up(o + (size_t)n_img * K * D * 4, 256);
  // slabs: G <= ceil(512 / n_img) per image (fast), or 1 (generic sums)
  const size_t gmax = (size_t)((512 + n_img - 1) / n_img);
  // (per-workgroup slabs exist only on the K <= 256 tile kernels; the generic path needs one)
  w.slabs = o; o = align_up(o + (size_t)n_img * (K <= 256 ? gmax : 1) * K * D * 4, 256);
  w.ids = o; o = align_up(o + (size_t)P * 8, 256);
  w.sums = o; o = align_up(o + (size_t)n_img * K * D * 4, 256);
  w.ssq = o; o = align_up(o + (size_t)n_img * K * ((D + 63) / 64) * 4, 256);
  // deterministic mode, generic route: fixed-point image of the M-step sums (segsum.hip)
  w.det64 = o;
  if (deterministic_mode()) o = align_up(o + (size_t)n_img * K * D * 8, 256);
  // pre-converted tiles (same 4 B per element as X), only for the shapes that use them
  w.xc = o;
  if (v3_shape(D, K, true) || v3k_shape(D, K) || assign64k_shape(D, K))
    o = align_up(o + (size_t)((P >> 5) + n_img + 1) * pre_tile_bytes(D / 32, D & 31), 256);
  // many-cluster kernels (kmeans_big.hip): keys, sort buffers, fixed-point sums, fragments
  w.big = o;
  if (bigk_shape(P, D, K, n_img)) o = align_up(o + bigk_workspace_bytes(P, D, K, n_img), 256);
  w.total = o;
  return w;
}

template <int NT, int KS, int KSPLIT>
int launch_pass_t(const PassArgs& a, const Plan& pl, hipStream_t s) {
  auto kern = kmeans_pass<NT, KS, KSPLIT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  hipLaunchKernelGGL(kern, dim3(pl.G, a.n_img), dim3(256), pl.lds, s, a);
  return launch_status();
}

template <int MT16, int Q, int TAIL, bool PRE>
int launch_pass16_t(const PassArgs& a, const Plan& pl, hipStream_t s) {
  auto kern = kmeans_pass16<MT16, Q, TAIL, PRE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  PassArgs b = a;
  { const char* e = getenv("SPML_KMEANS_STRIDED"); b.strided = !(e && e[0] == '0'); }
  hipLaunchKernelGGL(kern, dim3(pl.G, a.n_img), dim3(256), pl.lds, s, b);
  return launch_status();
}

int launch_preconvert(const float* x, int D, const int64_t* seg_off, int n_img, int64_t max_seg_len,
                      const Plan& pl, unsigned char* xc, hipStream_t s) {
  const int64_t tiles = (max_seg_len + 31) / 32;
  int64_t gx = (4096 + n_img - 1) / n_img;
  if (gx > tiles) gx = tiles;
  if (gx < 1) gx = 1;
  const dim3 grid((unsigned)gx, (unsigned)n_img);
#define SPML_PC(Q_)                                                                              \
  if (pl.Q == Q_) {                                                                              \
    if (pl.TAIL) hipLaunchKernelGGL((kmeans_preconvert<Q_, 1>), grid, dim3(256), 0, s, x, D, seg_off, xc); \
    else hipLaunchKernelGGL((kmeans_preconvert<Q_, 0>), grid, dim3(256), 0, s, x, D, seg_off, xc); \
    return launch_status();                                                                      \
  }
  SPML_PC(1) SPML_PC(2) SPML_PC(4) SPML_PC(8)
#undef SPML_PC
  return SPML_ERR_UNSUPPORTED;
}

template <int MTW, int Q, int TAIL>
int launch_pass16k_t(const PassArgs& a, const Plan& pl, hipStream_t s) {
  auto kern = kmeans_pass16k<MTW, Q, TAIL>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
  PassArgs b = a;
  { const char* e = getenv("SPML_KMEANS_STRIDED"); b.strided = !(e && e[0] == '0'); }
  hipLaunchKernelGGL(kern, dim3(pl.G, a.n_img), dim3(256), pl.lds, s, b);
  return launch_status();
}

int launch_pass(const PassArgs& a, const Plan& pl, hipStream_t s) {
  if (pl.v4k) {                                  // assign, then (fused pass) accumulate on the labels just written
    if (a.do_assign) {
      PassArgs e = a;
      e.do_accum = 0;
      const int rc = launch_assign64k(e, s);
      if (rc != SPML_OK || !a.do_accum) return rc;
    }
    PassArgs m = a;
    m.do_assign = 0;
    if (a.do_assign) {
      m.labels_in64 = nullptr;
      if (m.clocks) m.clocks += (size_t)2 * pl.G * a.n_img;     // (profiling: the second kernel of the pass)
    }
    return launch_accum64k(m, s);
  }
  if (pl.v3k) {
#define SPML_V3K(W_, Q_)                                                            \
  if (pl.MTW == W_ && pl.Q == Q_)                                                   \
    return pl.TAIL ? launch_pass16k_t<W_, Q_, 1>(a, pl, s) : launch_pass16k_t<W_, Q_, 0>(a, pl, s);
    SPML_V3K(2, 1) SPML_V3K(3, 1) SPML_V3K(4, 1)
    SPML_V3K(2, 2) SPML_V3K(3, 2) SPML_V3K(4, 2)
    SPML_V3K(2, 4)
#undef SPML_V3K
    return SPML_ERR_UNSUPPORTED;
  }
  if (pl.v3) {
#define SPML_V3(M_, Q_)                                                             \
  if (pl.MT16 == M_ && pl.Q == Q_) {                                                \
    if (pl.pre)                                                                     \
      return pl.TAIL ? launch_pass16_t<M_, Q_, 1, true>(a, pl, s)                   \
                     : launch_pass16_t<M_, Q_, 0, true>(a, pl, s);                  \
    return pl.TAIL ? launch_pass16_t<M_, Q_, 1, false>(a, pl, s)                    \
                   : launch_pass16_t<M_, Q_, 0, false>(a, pl, s);                   \
  }
#define SPML_V3Q(M_) SPML_V3(M_, 1) SPML_V3(M_, 2) SPML_V3(M_, 4) SPML_V3(M_, 8)
    SPML_V3Q(1) SPML_V3Q(2) SPML_V3Q(3) SPML_V3Q(4)
#undef SPML_V3Q
#undef SPML_V3
    return SPML_ERR_UNSUPPORTED;
  }
#define SPML_CASE(NT_, KS_, SP_) \
  if (pl.NT == NT_ && pl.KS == KS_ && pl.KSPLIT == SP_) return launch_pass_t<NT_, KS_, SP_>(a, pl, s);
#define SPML_CASES(NT_) \
  SPML_CASE(NT_, 2, 1) SPML_CASE(NT_, 3, 1) SPML_CASE(NT_, 5, 1) \
  SPML_CASE(NT_, 2, 2) SPML_CASE(NT_, 3, 2) SPML_CASE(NT_, 5, 2) \
  SPML_CASE(NT_, 2, 4) SPML_CASE(NT_, 3, 4) SPML_CASE(NT_, 5, 4)
  SPML_CASES(1)
  SPML_CASES(2)
#undef SPML_CASES
#undef SPML_CASE
  return SPML_ERR_UNSUPPORTED;
}

int generic_assign_launch(const float* x, int64_t P, int D, const int64_t* seg_off, int n_img,
                          int K, const float* cent, int32_t* labels, hipStream_t s) {
  const dim3 grid((unsigned)((P + 15) / 16));
  const int nc = (D + 63) / 64;
#define SPML_GA(NC) \
  hipLaunchKernelGGL(generic_assign<NC>, grid, dim3(256), 0, s, x, P, D, seg_off, n_img, K, cent, labels)
  if (nc <= 1) SPML_GA(1);
  else if (nc <= 2) SPML_GA(2);
  else if (nc <= 3) SPML_GA(3);
  else if (nc <= 5) SPML_GA(5);
  else if (nc <= 9) SPML_GA(9);
  else if (nc <= 17) SPML_GA(17);
  else return SPML_ERR_UNSUPPORTED;
#undef SPML_GA
  return launch_status();
}

}  // namespace
}  // namespace spml

using namespace spml;

namespace {

// What a k-means call does, decided from host-visible arguments only.
struct Route {
  Plan pl;
  bool big;                 // kmeans_big.hip
  const char* name;
};

// run_iterations: iterations of a spml_kmeans_run_f32 call, 0 for the given-prototype entry points
Route route_for(const float* x, int64_t P, int D, int K, int n_img, int64_t max_seg_len,
                int flags, bool want_pre, int run_iterations = 0) {
  Route r{};
  r.pl = make_plan(x, P, D, K, n_img, max_seg_len, flags, want_pre);
  r.big = !r.pl.fast && !(flags & SPML_KMEANS_FORCE_GENERIC) && bigk_shape(P, D, K, n_img);
  if (r.pl.fast)
    r.name = r.pl.v4k ? "mfma_f16x2_v4k" : r.pl.v3k ? "mfma_f16x2_v3k"
                      : r.pl.v3 ? (r.pl.pre ? ((pass64_shape(D, K) && !(flags & SPML_KMEANS_NO_PASS64)) ? "mfma_f16x2_v4p"
                                                                                                       : "mfma_f16x2_v3p")
                                            : "mfma_f16x2_v3")
                                : "mfma_f16x2";
  else
    r.name = r.big ? "mfma_f16x2_bigk" : "generic";
  return r;
}

}  // namespace

extern "C" const char* spml_kmeans_path_name(int64_t P, int D, int K, int n_img,
                                             int64_t max_seg_len, int iterations,
                                             int given_centroids, int flags) {
  if (P < 0 || D <= 0 || K <= 0 || n_img <= 0 || max_seg_len <= 0) return "invalid";
  // (alignment of x is checked at call time; a 16-byte aligned pointer is assumed here)
  const float* aligned = reinterpret_cast<const float*>(uintptr_t(256));
  const bool want_pre = given_centroids ? (flags & SPML_KMEANS_WS_PRECONVERTED) != 0
                                        : iterations >= 2;
  return route_for(aligned, P, D, K, n_img, max_seg_len, flags, want_pre,
                   given_centroids ? 0 : iterations).name;
}

extern "C" size_t spml_kmeans_workspace_bytes(int64_t P, int D, int K, int n_img,
                                              int64_t max_seg_len) {
  if (P < 0 || D <= 0 || K <= 0 || n_img <= 0) return 0;
  return ws_layout(P, D, K, n_img, max_seg_len).total;
}

extern "C" int spml_kmeans_profile_layout(int64_t P, int D, int K, int n_img, int64_t max_seg_len,
                                          int iterations, int* n_passes, int* workgroups_per_pass) {
  if (!n_passes || !workgroups_per_pass || P < 0 || D <= 0 || K <= 0 || n_img <= 0 ||
      max_seg_len <= 0 || iterations < 1)
    return SPML_ERR_INVALID_ARG;
  const float* aligned = reinterpret_cast<const float*>(uintptr_t(256));
  const Route r = route_for(aligned, P, D, K, n_img, max_seg_len, 0, iterations >= 2, iterations);
  if (!r.pl.fast) return SPML_ERR_UNSUPPORTED;
  // (v4k: an iteration that also accumulates is two kernels = two entries)
  *n_passes = r.pl.v4k ? 2 * iterations : iterations + 1;
  *workgroups_per_pass = r.pl.G * n_img;
  return SPML_OK;
}

// mode: 0 = run (labels_init, iterations), 1 = assign (given centroids -> labels),
//       2 = fused pass (given centroids -> labels + raw sums of X by the new labels)
static int kmeans_common(int mode, const float* x, int64_t P, int D, const int64_t* seg_off,
                         int n_img, int64_t max_seg_len, int K, const int64_t* labels_init,
                         const float* given_centroids, int iterations, int64_t* labels_out,
                         float* centroids_out, float* sums_out, int flags, void* ws,
                         size_t ws_bytes, unsigned long long* clocks, hipStream_t s) {
  if (!x || !seg_off || !labels_out || P < 0 || D <= 0 || K <= 0 || n_img <= 0 ||
      max_seg_len <= 0 || iterations < 0)
    return SPML_ERR_INVALID_ARG;
  if (!given_centroids && !labels_init) return SPML_ERR_INVALID_ARG;
  if (D > 1088 && K > 64) return SPML_ERR_UNSUPPORTED;
  const WsLayout wl = ws_layout(P, D, K, n_img, max_seg_len);
  if (!ws || ws_bytes < wl.total) return SPML_ERR_WORKSPACE;
  if (P == 0) return SPML_OK;
  unsigned char* base = static_cast<unsigned char*>(ws);
  int32_t* lab32 = reinterpret_cast<int32_t*>(base + wl.lab32);
  _Float16* cent_h = reinterpret_cast<_Float16*>(base + wl.cent_h);
  _Float16* cent_l = reinterpret_cast<_Float16*>(base + wl.cent_l);
  float* cent_f = reinterpret_cast<float*>(base + wl.cent_f);
  float* slabs = reinterpret_cast<float*>(base + wl.slabs);
  int64_t* ids = reinterpret_cast<int64_t*>(base + wl.ids);
  float* sums_buf = reinterpret_cast<float*>(base + wl.sums);
  float* ssq_buf = reinterpret_cast<float*>(base + wl.ssq);

  // pre-converting X pays for itself from the third pass on; given centroids: only when
  // the caller says the workspace already holds the converted tiles
  const bool want_pre = given_centroids ? (flags & SPML_KMEANS_WS_PRECONVERTED) != 0
                                        : iterations >= 2;
  const Route route = route_for(x, P, D, K, n_img, max_seg_len, flags, want_pre,
                                (mode == 0 && !given_centroids) ? iterations : 0);
  const Plan& pl = route.pl;
  if (given_centroids && (flags & SPML_KMEANS_WS_PRECONVERTED) && !(pl.fast && pl.pre))
    return SPML_ERR_INVALID_ARG;
  const unsigned pblocks = (unsigned)((P + 255) / 256);
  const int nchunk = (D + 63) / 64;
  int rc = SPML_OK;

  // tile-kernel paths read the int64 labels in their seed pass and write the int64 result in
  // their last pass; the other paths go through the int32 work array
  if (labels_init && !pl.fast)
    hipLaunchKernelGGL(labels_i64_to_i32, dim3(pblocks), dim3(256), 0, s, labels_init, lab32, P);

  if (pl.fast) {
    PassArgs a{};
    a.x = x; a.x_bytes = P * (int64_t)D * 4; a.P = P; a.D = D; a.K = K; a.n_img = n_img; a.G = pl.G;
    a.nvt = pl.nvt;
    a.seg_off = seg_off; a.cent_h = cent_h; a.cent_l = cent_l; a.kpad = pl.kpad;
    a.dpad = pl.dpad; a.labels = lab32; a.slabs = slabs;
    a.cent_f32 = given_centroids ? given_centroids : cent_f;
    a.trace = nullptr;
    a.xc = nullptr;
    a.xc_out = nullptr;
    a.clocks = nullptr;
    a.labels_in64 = nullptr;
    a.labels_out64 = nullptr;
    unsigned char* xc_buf = base + wl.xc;
    // v3: the seed pass converts (in LDS) and writes the tiles out itself; the
    // many-cluster kernel only exists on pre-converted tiles -> separate conversion
    const int tail_ch = D - 32 * (D / 32);
    const bool seed_converts = !given_centroids && pl.pre && pl.v3 && !pl.v3k &&
                               (tail_ch == 0 || tail_ch == 2) &&
                               !(flags & SPML_KMEANS_SEPARATE_PRECONVERT);
    if (given_centroids && pl.pre) {
      a.xc = xc_buf;                            // converted by spml_kmeans_preconvert_f32
    } else if (pl.pre && !seed_converts) {
      rc = launch_preconvert(x, D, seg_off, n_img, max_seg_len, pl, xc_buf, s);
      if (rc != SPML_OK) return rc;
      a.xc = xc_buf;
    }
#ifdef SPML_TRACE
    static unsigned long long* trace_buf = nullptr;
    if (getenv("SPML_KM_TRACE")) {
      if (!trace_buf) (void)hipMalloc(&trace_buf, (40 + 2048) * 8);
      a.trace = trace_buf;
    }
#endif
    // passes with an E-step on pre-converted tiles (K <= 48): the pixel-split kernel on 64-pixel tiles
    // (kmeans64.hip) with its own grid; the M-only seed pass stays on kmeans_pass16
    const bool use64 = pl.v3 && !pl.v3k && pl.pre && pass64_shape(D, K) && !(flags & SPML_KMEANS_NO_PASS64);
    int G64 = pl.G;
    if (use64) {
      const int64_t tiles64 = (max_seg_len + 63) / 64;
      int64_t gI = (256 * pass64_wg_per_cu(D, K) + n_img - 1) / n_img;
      if (gI > pl.G) gI = pl.G;                  // (the slab area of the workspace is sized for pl.G)
      if (gI > tiles64) gI = tiles64;
      G64 = (int)(gI < 1 ? 1 : gI);
    }
    const int frag_major = (use64 || pl.v4k) ? 1 : 0;     // prototype fragments in the order the 64-pixel kernels load them
    auto finalize = [&](int normalize, const float* src, int G) {
      // one launch when there are enough (cluster, image) rows to fill the chip: a block pulls its whole row
      // (G x D x 4 bytes) at ~50 GB/s, so 36 blocks of a single 513 x 513 x 258 image take longer (~10 us) than
      // the two launches of kmeans_reduce_slabs (5 x 36 blocks) + kmeans_normalize (7 + 5 us); 16 training
      // images: 34.8 instead of 38.5 us per iteration
      const int nl = ((flags & SPML_KMEANS_TWO_KERNEL_FINALIZE) || pl.kpad * n_img < 128) ? 0 : finalize_loads(G, D);
      if (normalize && nl) {
#define SPML_FIN(N_)                                                                                              \
        if (nl == N_)                                                                                              \
          hipLaunchKernelGGL(kmeans_finalize<N_>, dim3(pl.kpad, n_img), dim3(1024), 0, s, src, G, K, D, pl.kpad,    \
                             pl.dpad, frag_major, cent_f, cent_h, cent_l);
        SPML_FIN(8) SPML_FIN(16) SPML_FIN(32) SPML_FIN(56)
#undef SPML_FIN
      } else if (normalize) {
        hipLaunchKernelGGL(kmeans_reduce_slabs, dim3(nchunk, K, n_img), dim3(1024), 0, s, src, G, K,
                           D, sums_buf, ssq_buf);
        hipLaunchKernelGGL(kmeans_normalize, dim3(pl.kpad, n_img), dim3(256), 0, s, sums_buf, ssq_buf,
                           nchunk, K, D, pl.kpad, pl.dpad, 1, frag_major, cent_f, cent_h, cent_l);
      } else {                                  // given prototypes: split only
        hipLaunchKernelGGL(kmeans_normalize, dim3(pl.kpad, n_img), dim3(256), 0, s, src,
                           (const float*)nullptr, nchunk, K, D, pl.kpad, pl.dpad, 0, frag_major,
                           (float*)nullptr, cent_h, cent_l);
      }
    };
    int pass_index = 0;
    int last_G = pl.G;                          // slabs the last M-step pass wrote per image
    auto run_pass = [&](const Plan& plan) -> int {
      a.clocks = clocks ? clocks + (size_t)pass_index * 2 * pl.G * n_img : nullptr;
      pass_index += (pl.v4k && a.do_assign && a.do_accum) ? 2 : 1;
      if (use64 && a.do_assign && a.xc) {
        PassArgs b = a;
        b.G = G64;
        last_G = G64;
        return launch_pass64(b, s);
      }
      last_G = plan.G;
      return launch_pass(a, plan, s);
    };
    const bool pass_only = mode == 2 && (flags & SPML_KMEANS_PASS_ONLY) && (flags & SPML_KMEANS_WS_PRECONVERTED);
    if (given_centroids) {
      if (!pass_only) finalize(0, given_centroids, 1);          // split only
      a.do_assign = 1; a.do_accum = mode == 2 ? 1 : 0;
      a.labels_out64 = labels_out;
      rc = run_pass(pl);
      if (rc != SPML_OK) return rc;
      for (int rep = 1; pass_only && rep < ((flags >> 16) & 0xff); ++rep) {    // (measurement: back-to-back launches)
        --pass_index;
        rc = run_pass(pl);
        if (rc != SPML_OK) return rc;
      }
      if (mode == 2 && !pass_only)              // raw sums of X by the new labels
        hipLaunchKernelGGL(kmeans_reduce_slabs, dim3(nchunk, K, n_img), dim3(1024), 0, s, slabs, last_G,
                           K, D, sums_out, ssq_buf);
    } else {
      if (iterations > 0) {
        a.do_assign = 0; a.do_accum = 1;        // M-step on the initial labels
        a.labels_in64 = labels_init;
        if (seed_converts) {
          Plan seed = pl;                       // same grid, in-LDS conversion variant
          seed.pre = false;
          seed.lds = pass16_lds_bytes(D, false);
          a.xc_out = xc_buf;
          rc = run_pass(seed);
          a.xc_out = nullptr;
          a.xc = xc_buf;
        } else {
          rc = run_pass(pl);
        }
        if (rc != SPML_OK) return rc;
        a.labels_in64 = nullptr;
        finalize(1, slabs, last_G);
      } else if (labels_out != labels_init &&
                 hipMemcpyAsync(labels_out, labels_init, (size_t)P * 8, hipMemcpyDeviceToDevice, s) !=
                     hipSuccess) {              // zero iterations: the labels pass through
        return SPML_ERR_LAUNCH;
      }
      for (int it = 0; it < iterations; ++it) {
        const bool last = (it == iterations - 1);
        a.do_assign = 1; a.do_accum = last ? 0 : 1;
        a.labels_out64 = last ? labels_out : nullptr;
        rc = run_pass(pl);
        if (rc != SPML_OK) return rc;
        if (!last) finalize(1, slabs, last_G);
      }
    }
#ifdef SPML_TRACE
    if (a.trace) {
      (void)hipStreamSynchronize(s);
      static unsigned long long h[40 + 2048];
      (void)hipMemcpy(h, a.trace, sizeof(h), hipMemcpyDeviceToHost);
      const char* nm16[8] = {"wait", "convert|epilogue", "barrier+dma", "E", "barrier", "labels", "M", "prologue"};
      const char* nm64[8] = {"wait+barrier", "epilogue", "dma-issue", "E-mfma", "argmax+publish+barrier", "onehot", "M-mfma", "prologue"};
      const char* nm64k[8] = {"dma-wait", "barrier", "label-read", "epilogue", "-", "onehot", "M-mfma", "prologue"};   // (the accumulate kernel)
      const char** nm = pl.v4k ? nm64k : use64 ? nm64 : nm16;
      for (int w = 0; w < 4; ++w) {
        fprintf(stderr, "wave%d:", w);
        for (int i = 0; i < 8; ++i) fprintf(stderr, " %s=%llu", nm[i], h[w * 8 + i]);
        fprintf(stderr, " wall_us=%.2f\n", h[32 + w] * 0.01);
      }
    }
#endif
  } else if (route.big) {
    rc = bigk_run(x, P, D, seg_off, n_img, max_seg_len, K, given_centroids, iterations, lab32, cent_f,
                  mode == 2 ? sums_out : nullptr, flags, base + wl.big, s);
    if (rc != SPML_OK) return rc;
  } else {
    auto assign = [&](const float* cent) -> int {
      return generic_assign_launch(x, P, D, seg_off, n_img, K, cent, lab32, s);
    };
    const int64_t M = (int64_t)n_img * K;
    auto msums = [&](float* dst) -> int {
      hipLaunchKernelGGL(generic_ids, dim3(pblocks), dim3(256), 0, s, lab32, seg_off, n_img, K, P,
                         ids);
      if (deterministic_mode()) {                // fixed-point sums (order-independent), converted once
        long long* acc = reinterpret_cast<long long*>(base + wl.det64);
        if (hipMemsetAsync(acc, 0, (size_t)M * D * 8, s) != hipSuccess) return SPML_ERR_LAUNCH;
        const int rc_ = segment_sum_launch(x, ids, P, D, M, dst, s, acc);
        if (rc_ != SPML_OK) return rc_;
        hipLaunchKernelGGL(fix_to_f32, dim3((unsigned)((M * D + 255) / 256)), dim3(256), 0, s, acc, M * D, dst);
        return launch_status();
      }
      if (hipMemsetAsync(dst, 0, (size_t)M * D * 4, s) != hipSuccess) return SPML_ERR_LAUNCH;
      return segment_sum_launch(x, ids, P, D, M, dst, s);
    };
    if (given_centroids) {
      rc = assign(given_centroids);
      if (rc != SPML_OK) return rc;
      if (mode == 2) {
        rc = msums(sums_out);
        if (rc != SPML_OK) return rc;
      }
    } else {
      for (int it = 0; it < iterations; ++it) {
        rc = msums(slabs);
        if (rc != SPML_OK) return rc;
        rc = spml_normalize_rows_f32(slabs, M, D, cent_f, s);
        if (rc != SPML_OK) return rc;
        rc = assign(cent_f);
        if (rc != SPML_OK) return rc;
      }
    }
  }
  if (!pl.fast)
    hipLaunchKernelGGL(labels_i32_to_i64, dim3(pblocks), dim3(256), 0, s, lab32, labels_out, P);
  if (centroids_out && !given_centroids && iterations > 0) {
    if (hipMemcpyAsync(centroids_out, cent_f, (size_t)n_img * K * D * 4,
                       hipMemcpyDeviceToDevice, s) != hipSuccess)
      return SPML_ERR_LAUNCH;
  }
  return launch_status();
}

extern "C" int spml_kmeans_run_f32(const float* x, int64_t P, int D, const int64_t* seg_offsets,
                                   int n_img, int64_t max_seg_len, int K,
                                   const int64_t* labels_init, int iterations,
                                   int64_t* labels_out, float* centroids_out, int flags,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!labels_init) return SPML_ERR_INVALID_ARG;
  return kmeans_common(0, x, P, D, seg_offsets, n_img, max_seg_len, K, labels_init, nullptr,
                       iterations, labels_out, centroids_out, nullptr, flags, ws, ws_bytes, nullptr,
                       (hipStream_t)stream);
}

extern "C" int spml_kmeans_run_profiled_f32(const float* x, int64_t P, int D,
                                            const int64_t* seg_offsets, int n_img,
                                            int64_t max_seg_len, int K, const int64_t* labels_init,
                                            int iterations, int64_t* labels_out, int flags, void* ws,
                                            size_t ws_bytes, uint64_t* pass_clocks,
                                            size_t pass_clocks_len, void* stream) {
  if (!labels_init || !pass_clocks || iterations < 1) return SPML_ERR_INVALID_ARG;
  int n_pass = 0, wgs = 0;
  const int rc = spml_kmeans_profile_layout(P, D, K, n_img, max_seg_len, iterations, &n_pass, &wgs);
  if (rc != SPML_OK) return rc;
  if (pass_clocks_len < (size_t)n_pass * wgs * 2) return SPML_ERR_WORKSPACE;
  return kmeans_common(0, x, P, D, seg_offsets, n_img, max_seg_len, K, labels_init, nullptr,
                       iterations, labels_out, nullptr, nullptr,
                       flags & ~SPML_KMEANS_FORCE_GENERIC, ws,
                       ws_bytes, reinterpret_cast<unsigned long long*>(pass_clocks),
                       (hipStream_t)stream);
}

extern "C" int spml_kmeans_assign_f32(const float* x, int64_t P, int D,
                                      const int64_t* seg_offsets, int n_img,
                                      int64_t max_seg_len, int K, const float* centroids,
                                      int64_t* labels_out, int flags, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!centroids) return SPML_ERR_INVALID_ARG;
  return kmeans_common(1, x, P, D, seg_offsets, n_img, max_seg_len, K, nullptr, centroids, 1,
                       labels_out, nullptr, nullptr, flags, ws, ws_bytes, nullptr,
                       (hipStream_t)stream);
}

extern "C" int spml_kmeans_fused_pass_f32(const float* x, int64_t P, int D,
                                          const int64_t* seg_offsets, int n_img,
                                          int64_t max_seg_len, int K, const float* centroids_in,
                                          int64_t* labels_out, float* centroid_sums_out, int flags,
                                          void* ws, size_t ws_bytes, void* stream) {
  if (!centroids_in || !centroid_sums_out) return SPML_ERR_INVALID_ARG;
  return kmeans_common(2, x, P, D, seg_offsets, n_img, max_seg_len, K, nullptr, centroids_in, 1,
                       labels_out, nullptr, centroid_sums_out, flags, ws, ws_bytes, nullptr,
                       (hipStream_t)stream);
}

extern "C" int spml_kmeans_preconvert_f32(const float* x, int64_t P, int D,
                                          const int64_t* seg_offsets, int n_img,
                                          int64_t max_seg_len, int K, void* ws, size_t ws_bytes,
                                          void* stream) {
  if (!x || !seg_offsets || P < 0 || D <= 0 || K <= 0 || n_img <= 0 || max_seg_len <= 0)
    return SPML_ERR_INVALID_ARG;
  const WsLayout wl = ws_layout(P, D, K, n_img, max_seg_len);
  if (!ws || ws_bytes < wl.total) return SPML_ERR_WORKSPACE;
  const Route r = route_for(x, P, D, K, n_img, max_seg_len, 0, true);
  if (!(r.pl.fast && r.pl.pre)) return SPML_ERR_UNSUPPORTED;
  if (P == 0) return SPML_OK;
  return launch_preconvert(x, D, seg_offsets, n_img, max_seg_len, r.pl,
                           static_cast<unsigned char*>(ws) + wl.xc, (hipStream_t)stream);
}
