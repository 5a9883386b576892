// Spherical k-means that moves fewer bytes per iteration: hi-half screened E-step + exact
// incremental M-step ("mfma_f16_screened_inc"; K <= 64, D = 32q + {0,2}, >= 2 iterations).
//
// Same algorithm as kmeans.hip (reference: kmeans_with_initial_labels, segsort/common.py:67-97;
// M-step common.py:11-41, E-step common.py:44-64), different decomposition.  The fused pass of
// kmeans.hip streams all of X (P*D*4 bytes) every iteration.  Here, per k-means call:
//
//   kmi_seed    reads X once: M-step on the initial labels with every element rounded to a
//               2^-32 grid and summed in fp64.  Such sums are EXACT (|sum| < 2^20 needs 52 bits),
//               hence associative: any later "+x to the new cluster, -x from the old one" gives
//               bit for bit what a full re-summation would.  Also writes the f16 hi half of X as
//               MFMA operand tiles (P*D*2 bytes) and, per pixel, the margin below which the
//               hi half alone cannot decide the arg-max:  tau_p = 2 * (|x_p - h_p| + slack)
//               (Cauchy-Schwarz: |<c, x - h>| <= |c| |x - h|, prototypes have unit norm);
//   kmi_norm    sums -> unit-norm fp32 prototypes + their split-f16 fragments;
//   per iteration:
//   kmi_screen  streams the hi tiles only: scores (h' + l') . h on the matrix cores, top-2 per
//               pixel; margin > tau_p  =>  the arg-max over the exact scores is the same: label
//               decided (pixels whose label changed go to a list); otherwise the pixel goes to
//               the list of ambiguous pixels (1-4 % on unit vectors);
//   kmi_fix     gathers the fp32 rows of the listed pixels only: exact split-f16 scores on the
//               matrix cores for the ambiguous ones (16 pixels per wave), and
//               sums[new] += x, sums[old] -= x  for every pixel whose label changed (LDS table
//               of fp64 partial sums per workgroup, one 64-bit atomic per touched entry at the end);
//   kmi_norm.
// Lists are per screen workgroup (fixed slots, no atomics, no host round trip); results are
// run-to-run bit-identical.  Labels equal the arg-max of the exact scores wherever the top-2
// margin exceeds fp32 round-off (the contract of spml_kmeans_assign_f32), prototypes are the
// normalised exact sums (2^-32 per element, better than an fp32 scatter_add).
// Input domain: |x| <= 1 element-wise and images of at most 2^20 pixels (exactness of the sums);
// unit-norm rows are the intended use.  Prototypes are normalised by this file.
#include <stdio.h>
#include <stdlib.h>

#include "common.cuh"

namespace spml {

namespace {

typedef float float4a __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// x rounded to the 2^-32 grid, as fp64 (round to nearest even through the magic constant
// 1.5 * 2^20: valid for |x| < 2^19)
constexpr double kGridMagic = 1572864.0;
__device__ __forceinline__ double to_grid(float x) {
  const double d = (double)x + kGridMagic;
  return d - kGridMagic;
}
typedef __attribute__((address_space(1))) double* gdouble_t;
typedef __attribute__((address_space(3))) double* ldouble_t;
__device__ __forceinline__ void global_add_f64(double* p, double v) {
  __builtin_amdgcn_global_atomic_fadd_f64((gdouble_t)p, v);
}

__device__ __forceinline__ float4a mfma16(half8 a, half8 b, float4a c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

struct IncArgs {
  const float* x;
  int64_t x_bytes;
  int64_t P;
  int D, K, n_img, G;
  const int64_t* seg_off;        // device [n_img+1]
  const int64_t* labels_in64;    // seed: the caller's initial labels
  int32_t* labels;               // [P] work labels
  int64_t* labels_out64;         // last iteration: the caller's output, else null
  unsigned char* xh;             // hi-half tiles
  float* tau;                    // [P] decision margin per pixel
  double* sums64;                // [n_copy][n_img][K][D] exact sums of the grid-rounded X by label,
                                 // spread over n_copy partial tables (workgroup g adds to g % n_copy:
                                 // 64-bit float atomics to ONE address serialise at ~85 ns each)
  int n_copy;
  unsigned long long* trace;     // SPML_KMEANS_INC_TRACE: phase stamps of one fix workgroup, or null
  double* slabs;                 // [n_img][G_fix][K][D] per-workgroup deltas of one fix pass
  int32_t* slab_flag;            // [n_img][G_fix] 1 = the slab holds a delta
  const _Float16* cent_h;        // [n_img][kpad][dpad]
  const _Float16* cent_l;
  const float* cent_f;           // [n_img][K][D]
  int kpad, dpad;
  int32_t* amb;                  // [P] ambiguous pixels, per screen-workgroup segment
  unsigned long long* chg;       // [P] changed pixels: pixel | old << 32 | new << 48
  int32_t* counts;               // [n_img][G][2] entries of the two lists of a segment
  int make_lists;                // screen: also the changed list (0 on the last iteration)
  int do_update;                 // fix: update the sums (0 on the last iteration)
  int G_screen;                  // fix: segments per image
  unsigned long long* clocks;    // profiling: [n_img][clock_stride][2] start / end stamps, or null
  int clock_stride;
};

#define KMI_CLOCK_BEGIN                                                                       \
  if (a.clocks && threadIdx.x == 0)                                                           \
    for (int g_ = blockIdx.x; g_ < a.clock_stride; g_ += gridDim.x)                          \
      a.clocks[2 * ((size_t)blockIdx.y * a.clock_stride + g_)] = wall_clock64();
#define KMI_CLOCK_END                                                                         \
  if (a.clocks) {                                                                             \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
    if (threadIdx.x == 0)                                                                     \
      for (int g_ = blockIdx.x; g_ < a.clock_stride; g_ += gridDim.x)                        \
        a.clocks[2 * ((size_t)blockIdx.y * a.clock_stride + g_) + 1] = wall_clock64();       \
  }

// bytes of one hi-half tile of 32 pixels: per 32-channel step and 16-pixel half a 1-KB block
// (16-B slot of (pixel pix, 8-channel group g) at g*256 + pix*16: a lane group of the E-step
// reads 256 contiguous bytes), + 2 x 256 B for the location step (8 channels: 2 real, 6 zero)
__host__ __device__ constexpr int hi_tile_bytes(int q, int tail) { return q * 2048 + (tail ? 512 : 0); }
__host__ __device__ inline int64_t hi_tile0(int64_t seg0, int img) { return (seg0 >> 5) + img; }

// raw fp32 tile of 32 rows in LDS, copied in 1-KB units (one per wave instruction)
__host__ __device__ inline int raw_units(int D) { return (32 * D * 4 + 16 + 1023) / 1024; }
__host__ __device__ inline size_t seed_lds_bytes(int D) {
  const int q = D / 32, tl = D - 32 * q;
  return (size_t)raw_units(D) * 1024 + (size_t)hi_tile_bytes(q, tl) + 256 + 2 * 4 * 32 * 4;
}

// ---------------------------------------------------------------------------------------------
// Seed pass: fp32 X -> hi tiles + tau + fixed-point sums by the initial labels.
// ---------------------------------------------------------------------------------------------
template <int Q, int TAIL>
__global__ __launch_bounds__(256, 3) void kmi_seed(IncArgs a) {
  constexpr int TPW = 32;
  constexpr int tail = 2 * TAIL;
  constexpr int D = 32 * Q + tail;
  constexpr int TILE = hi_tile_bytes(Q, TAIL);
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int img = blockIdx.y, g = blockIdx.x;
  const int K = a.K;
  const int nun = raw_units(D);
  unsigned char* xs = lds;                                         // raw fp32 tile (DMA target)
  unsigned char* conv = lds + (size_t)nun * 1024;                  // hi tile being built
  int* labin = reinterpret_cast<int*>(conv + TILE);
  float* part = reinterpret_cast<float*>(labin + 64);              // [2][4 waves][32]

  KMI_CLOCK_BEGIN
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  const int64_t t_begin = (T * g) / a.G, t_end = (T * (g + 1)) / a.G;
  if (t_begin >= t_end) {
    KMI_CLOCK_END
    return;
  }
  const unsigned char* xbase = reinterpret_cast<const unsigned char*>(a.x);
  const int64_t tile0 = hi_tile0(seg0, img);

  auto tile_issue = [&](int64_t t) {
    const int64_t r0 = seg0 + t * TPW;
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = r0 * D * 4, b1 = b0 + (int64_t)nrows * D * 4;
    const int64_t a0 = b0 & ~(int64_t)15;
    const int nvec = (int)((b1 - a0 + 15) >> 4);
    const unsigned char* tbase = xbase + a0;
    const int lim = (int)min((int64_t)0x7ffffff0, a.x_bytes - 16 - a0);   // last legal 16-B load
    const int last = min(16 * (nvec - 1), lim);
    for (int u = wave; u < nun; u += 4)
      __builtin_amdgcn_global_load_lds((gptr_t)(tbase + min(u * 1024 + 16 * lane, last)),
                                       (lptr_t)(xs + (size_t)u * 1024), 16, 0, 0);
    if (wave == 0) {                     // low words of the int64 labels (little endian, < 2^31)
      const int64_t p = min(r0 + min(lane, TPW - 1), a.P - 1);
      __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const int32_t*>(a.labels_in64 + p)),
                                       (lptr_t)labin, 4, 0, 0);
    }
  };

  // run accumulators (exact fp64 sums of grid-rounded values): lane = channel 64 * wave + lane
  // (waves beyond 32 * Q idle), wave 0 also carries the tail channels on its first lanes
  const int cmain = 64 * wave + lane;
  const bool main_on = cmain < 32 * Q;
  const bool tail_on = TAIL && wave == 0 && lane < tail;
  double macc = 0.0, tacc = 0.0;
  int cur = -1;
  auto flush = [&]() {
    if (cur >= 0 && cur < K) {
      double* dst = a.sums64 + (((size_t)(g % a.n_copy) * a.n_img + img) * K + cur) * D;
      if (main_on && macc != 0.0) global_add_f64(dst + cmain, macc);
      if (TAIL && tail_on && tacc != 0.0) global_add_f64(dst + 32 * Q + lane, tacc);
    }
    macc = 0.0;
    tacc = 0.0;
  };

  tile_issue(t_begin);
  for (int64_t t = t_begin; t < t_end; ++t) {
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t r0 = seg0 + t * TPW;
    const int64_t b0 = r0 * D * 4;
    const int shift = (int)(b0 & 15);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();                                   // raw tile t and its labels landed
    {
      const int64_t a0 = b0 & ~(int64_t)15;
      const int nvec = (int)((b0 + (int64_t)nrows * D * 4 - a0 + 15) >> 4);
      if (a0 + 16 * (int64_t)nvec > a.x_bytes) {    // the clamped last vector sits 8 B low
        if (tid == 0) {
          float2* slot = reinterpret_cast<float2*>(xs + 16 * (size_t)(nvec - 1));
          slot[0] = slot[1];
        }
        wg_barrier();
      }
    }
    const unsigned char* xrow = xs + shift;
    // ---- hi halves into the tile layout; |x - h|^2 and |h|^2 per pixel ----
    float rr = 0.f, hh = 0.f;
    {
      float2 raw[Q][2];
#pragma unroll
      for (int it = 0; it < Q; ++it) {
        const int id = it * 256 + tid;
        const int pix = id & 31, qd = id >> 5;
        const float2* src = reinterpret_cast<const float2*>(xrow + ((size_t)pix * D + 4 * qd) * 4);
        raw[it][0] = src[0];
        raw[it][1] = src[1];
      }
      float2 tl = {0.f, 0.f};
      if (TAIL && tid < 32) tl = *reinterpret_cast<const float2*>(xrow + ((size_t)tid * D + D - 2) * 4);
#pragma unroll
      for (int it = 0; it < Q; ++it) {
        const int id = it * 256 + tid;
        const int pix = id & 31, qd = id >> 5;
        const bool ok = pix < nrows;                 // stale LDS may hold NaN patterns
        float v[4] = {ok ? raw[it][0].x : 0.f, ok ? raw[it][0].y : 0.f,
                      ok ? raw[it][1].x : 0.f, ok ? raw[it][1].y : 0.f};
        typedef _Float16 half4v __attribute__((ext_vector_type(4)));
        half4v h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          h[e] = (_Float16)v[e];
          const float hf = (float)h[e], r = v[e] - hf;
          rr += r * r;
          hh += hf * hf;
        }
        const int c = 4 * qd;
        const int s = c >> 5, grp = (c & 31) >> 3, e0 = c & 7;
        unsigned char* dst = conv + (size_t)(s * 2 + (pix >> 4)) * 1024 + grp * 256 + (pix & 15) * 16 + 2 * e0;
        *reinterpret_cast<half4v*>(dst) = h;
      }
      if (TAIL && tid < 32) {
        const bool ok = tid < nrows;
        const float v0 = ok ? tl.x : 0.f, v1 = ok ? tl.y : 0.f;
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
        const float r0f = v0 - (float)h0, r1f = v1 - (float)h1;
        rr += r0f * r0f + r1f * r1f;
        hh += (float)h0 * (float)h0 + (float)h1 * (float)h1;
        const _Float16 z = (_Float16)0.f;
        *reinterpret_cast<half8*>(conv + (size_t)Q * 2048 + (tid >> 4) * 256 + (tid & 15) * 16) =
            half8{h0, h1, z, z, z, z, z, z};
      }
      rr += __shfl_xor(rr, 32, kWave);
      hh += __shfl_xor(hh, 32, kWave);
      if (lane < 32) {
        part[wave * 32 + lane] = rr;
        part[128 + wave * 32 + lane] = hh;
      }
    }
    // ---- fixed-point sums by label, straight from the raw rows ----
    const int lab_lane = labin[lane & 31];
    for (int p0 = 0; p0 < nrows; p0 += 8) {
      float vm[8], vt[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                  // all LDS reads of the group in flight
        const int pix = min(p0 + u, nrows - 1);
        vm[u] = main_on ? *reinterpret_cast<const float*>(xrow + ((size_t)pix * D + cmain) * 4) : 0.f;
        vt[u] = (TAIL && tail_on)
                    ? *reinterpret_cast<const float*>(xrow + ((size_t)pix * D + 32 * Q + lane) * 4) : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (p0 + u < nrows) {                        // workgroup-uniform
          const int lb = __builtin_amdgcn_readlane(lab_lane, p0 + u);
          if (lb != cur) { flush(); cur = lb; }
          macc += to_grid(vm[u]);
          if (TAIL) tacc += to_grid(vt[u]);
        }
      }
    }
    const int mylab = lab_lane;
    wg_barrier();                                   // raw tile free; hi tile + partials complete
    if (t + 1 < t_end) tile_issue(t + 1);
    {
      unsigned char* out = a.xh + (size_t)(tile0 + t) * TILE + 16 * lane;
      for (int b = wave; b < 2 * Q; b += 4)
        __builtin_nontemporal_store(*reinterpret_cast<const half8*>(conv + (size_t)b * 1024 + 16 * lane),
                                    reinterpret_cast<half8*>(out + (size_t)b * 1024));
      if (TAIL && wave < 2 && lane < 16)
        __builtin_nontemporal_store(
            *reinterpret_cast<const half8*>(conv + (size_t)Q * 2048 + wave * 256 + 16 * lane),
            reinterpret_cast<half8*>(out + (size_t)Q * 2048 + wave * 256));
    }
    if (wave == 1 && lane < nrows) a.labels[r0 + lane] = mylab;
    if (wave == 2 && lane < nrows) {
      const float r2 = part[lane] + part[32 + lane] + part[64 + lane] + part[96 + lane];
      const float h2 = part[128 + lane] + part[160 + lane] + part[192 + lane] + part[224 + lane];
      // decided iff (top-1 - top-2 of the screen scores) > tau: both scores are off by at most
      // |c| |x - h| (|c| <= 1 + 2^-22) plus the fp32 accumulation noise of the two computations
      a.tau[r0 + lane] = 2.0f * (1.001f * sqrtf(r2) + 8e-6f * (sqrtf(h2) + 1e-2f));
    }
  }
  flush();
  KMI_CLOCK_END
}

// ---------------------------------------------------------------------------------------------
// Screen pass: hi tiles -> labels of the decided pixels + the two lists.
// ---------------------------------------------------------------------------------------------
__host__ __device__ constexpr size_t screen_lds_bytes(int q, int tail) {
  return (size_t)2 * (q + (tail ? 1 : 0)) * 2048 + 2 * 16 * 32 * 12;
}

template <int MT16, int Q, int TAIL>
__global__ __launch_bounds__(256, 3) void kmi_screen(IncArgs a) {
  constexpr int TPW = 32;
  constexpr int QE = Q + TAIL;
  constexpr int NHS = 2 * QE;                      // (k-step, pixel half) steps of a tile
  constexpr int TILE = hi_tile_bytes(Q, TAIL);
  constexpr int SLOT = QE * 2048;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
  const int img = blockIdx.y, g = blockIdx.x;
  const int K = a.K;
  unsigned char* slot0 = lds;
  float* cand_b1 = reinterpret_cast<float*>(lds + 2 * SLOT);      // [2][16 rows][32]
  int* cand_i1 = reinterpret_cast<int*>(cand_b1 + 2 * 16 * 32);
  float* cand_b2 = reinterpret_cast<float*>(cand_i1 + 2 * 16 * 32);

  KMI_CLOCK_BEGIN
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  const int64_t t_begin = (T * g) / a.G, t_end = (T * (g + 1)) / a.G;
  int32_t* cnt = a.counts + ((size_t)img * a.G + g) * 2;
  if (t_begin >= t_end) {
    if (tid == 0) { cnt[0] = 0; cnt[1] = 0; }
    KMI_CLOCK_END
    return;
  }
  if (TAIL) {                         // channel groups 1..3 of the location step do not exist
    for (int i = tid; i < 2 * 2 * 192; i += 256) {
      const int blk = i / 192, w = i % 192;
      reinterpret_cast<float*>(slot0 + (size_t)(blk >> 1) * SLOT + (size_t)(2 * Q + (blk & 1)) * 1024 + 256)[w] = 0.f;
    }
  }
  const int64_t tile0 = hi_tile0(seg0, img);
  auto tile_issue = [&](int64_t t, int slot) {
    const unsigned char* tb = a.xh + (size_t)(tile0 + t) * TILE + 16 * lane;
    unsigned char* dst0 = slot0 + (size_t)slot * SLOT;
    if (MT16 < 4) {
      if (wave == 3) {                // waves 0..MT16-1 stay on the matrix cores
#pragma unroll 4
        for (int b = 0; b < 2 * Q; ++b)
          __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)b * 1024), (lptr_t)(dst0 + b * 1024), 16, 0, 0);
        if (TAIL && lane < 16) {
#pragma unroll
          for (int n = 0; n < 2; ++n)
            __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)Q * 2048 + n * 256),
                                             (lptr_t)(dst0 + (2 * Q + n) * 1024), 16, 0, 0);
        }
      }
    } else {
      for (int b = wave; b < 2 * Q; b += 4)
        __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)b * 1024), (lptr_t)(dst0 + b * 1024), 16, 0, 0);
      if (TAIL && wave < 2 && lane < 16)
        __builtin_amdgcn_global_load_lds((gptr_t)(tb + (size_t)Q * 2048 + wave * 256),
                                         (lptr_t)(dst0 + (2 * Q + wave) * 1024), 16, 0, 0);
    }
  };

  // prototypes of this wave's 16 rows -> registers (A operand)
  half8 ah[QE], al[QE];
  // wave 3: labels / margins of the tile in flight and of the tile being finalised
  int lab_nxt = 0, lab_cur = 0;
  float tau_nxt = 0.f, tau_cur = 0.f;
  int n_amb = 0, n_chg = 0;
  const int64_t list0 = seg0 + t_begin * TPW;       // this workgroup's slots in amb / chg

  auto side_loads = [&](int64_t t) {                // wave 3, lanes = pixels
    const int64_t p = min(seg0 + t * TPW + (lane & 31), a.P - 1);
    lab_nxt = a.labels[p];
    tau_nxt = a.tau[p];
  };
  auto finalise = [&](int64_t t, int tb, int lab_old, float tau_v) {   // wave 3
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int px = lane & 31, half = lane >> 5;
    constexpr int NR = 4 * MT16;                    // candidate rows, ascending prototype ranges
    float b1 = -INFINITY, b2 = -INFINITY;
    int i1 = 0x7fffffff;
    float cb1[NR / 2], cb2[NR / 2];
    int ci1[NR / 2];
#pragma unroll
    for (int r = 0; r < NR / 2; ++r) {
      const int row = half * (NR / 2) + r;
      const int w = row >> 2, l4 = row & 3;
      const int o = (tb * 16 + w * 4 + l4) * 32 + px;
      cb1[r] = cand_b1[o];
      ci1[r] = cand_i1[o];
      cb2[r] = cand_b2[o];
    }
#pragma unroll
    for (int r = 0; r < NR / 2; ++r) {
      if (cb1[r] > b1) { b2 = fmaxf(b1, cb2[r]); b1 = cb1[r]; i1 = ci1[r]; }
      else b2 = fmaxf(b2, cb1[r]);
    }
    {   // the upper half of the wave holds the higher prototype ranges
      const float ob1 = __shfl_xor(b1, 32, kWave), ob2 = __shfl_xor(b2, 32, kWave);
      const int oi1 = __shfl_xor(i1, 32, kWave);
      if (half == 0) {
        if (ob1 > b1) { b2 = fmaxf(b1, ob2); b1 = ob1; i1 = oi1; }
        else b2 = fmaxf(b2, ob1);
      }
    }
    const bool valid = lane < nrows;
    const bool decided = (b1 - b2) > tau_v;
    const bool changed = valid && decided && i1 != lab_old;
    const bool ambiguous = valid && !decided;
    const int64_t p = seg0 + t * TPW + lane;
    if (changed) a.labels[p] = i1;
    if (a.labels_out64 && valid) a.labels_out64[p] = (int64_t)(decided ? i1 : lab_old);
    const unsigned long long mc = __ballot(changed), ma = __ballot(ambiguous);
    const unsigned long long below = (1ull << lane) - 1ull;
    if (ambiguous) a.amb[list0 + n_amb + __popcll(ma & below)] = (int32_t)(t * TPW + lane);
    if (changed && a.make_lists)
      a.chg[list0 + n_chg + __popcll(mc & below)] =
          (unsigned long long)(uint32_t)(t * TPW + lane) |
          ((unsigned long long)(uint32_t)(lab_old >= 0 && lab_old < K ? lab_old : 0xffff) << 32) |
          ((unsigned long long)(uint32_t)i1 << 48);
    n_amb += __popcll(ma);
    n_chg += __popcll(mc);
  };

  if (wave == 3) side_loads(t_begin);
  tile_issue(t_begin, 0);
  if (wave < MT16) {
#pragma unroll
    for (int s = 0; s < QE; ++s) {
      const size_t o = ((size_t)img * a.kpad + 16 * wave + lc) * a.dpad + 32 * s + 8 * lg;
      ah[s] = *reinterpret_cast<const half8*>(a.cent_h + o);
      al[s] = *reinterpret_cast<const half8*>(a.cent_l + o);
    }
  }
  for (int64_t t = t_begin; t < t_end; ++t) {
    const int tb = (int)((t - t_begin) & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    wg_barrier();             // tile t landed; the other slot and candidate table t-1 are complete
    if (wave == 3) {
      const int lab_t = lab_nxt;
      const float tau_t = tau_nxt;
      if (t + 1 < t_end) side_loads(t + 1);
      if (MT16 < 4 && t + 1 < t_end) tile_issue(t + 1, tb ^ 1);
      if (MT16 < 4 && t > t_begin) finalise(t - 1, tb ^ 1, lab_cur, tau_cur);
      lab_cur = lab_t;
      tau_cur = tau_t;
    }
    if (MT16 == 4 && t + 1 < t_end) tile_issue(t + 1, tb ^ 1);
    if (wave < MT16) {
      float4a eh[2], ex[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) { eh[n] = float4a{0.f, 0.f, 0.f, 0.f}; ex[n] = eh[n]; }
      half8 bq[4];
      const unsigned cbase = (unsigned)(size_t)(lptr_t)(slot0 + (size_t)tb * SLOT) + (unsigned)(lg * 256 + lc * 16);
      auto load_b = [&](int hs) {
        const unsigned addr = cbase + (unsigned)hs * 1024u;
        asm volatile("ds_read_b128 %0, %1" : "=&v"(bq[hs & 3]) : "v"(addr));
      };
#pragma unroll
      for (int hs = 0; hs < 4 && hs < NHS; ++hs) load_b(hs);
#pragma unroll
      for (int hs = 0; hs < NHS; ++hs) {
        const int n = hs & 1, s = hs >> 1;
        const int rem = NHS - 1 - hs;            // loads issued after this one: min(3, rem)
        if (rem >= 3) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(bq[hs & 3]));
        else if (rem == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(bq[hs & 3]));
        else if (rem == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(bq[hs & 3]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[hs & 3]));
        eh[n] = mfma16(ah[s], bq[hs & 3], eh[n]);
        ex[n] = mfma16(al[s], bq[hs & 3], ex[n]);
        __builtin_amdgcn_sched_barrier(0);
        if (hs + 4 < NHS) load_b(hs + 4);
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        float b1 = -INFINITY, b2 = -INFINITY;
        int i1 = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * wave + 4 * lg + r;
          const float sdot = eh[n][r] + ex[n][r] * kSplitInv;
          if (c < K) {
            if (sdot > b1) { b2 = b1; b1 = sdot; i1 = c; }
            else b2 = fmaxf(b2, sdot);
          }
        }
        const int o = (tb * 16 + wave * 4 + lg) * 32 + 16 * n + lc;
        cand_b1[o] = b1;
        cand_i1[o] = i1;
        cand_b2[o] = b2;
      }
    }
    if (MT16 == 4 && wave == 3 && t > t_begin) finalise(t - 1, tb ^ 1, lab_cur, tau_cur);
  }
  wg_barrier();
  if (wave == 3) {
    finalise(t_end - 1, (int)((t_end - 1 - t_begin) & 1), lab_cur, tau_cur);
    if (lane == 0) { cnt[0] = n_amb; cnt[1] = n_chg; }
  }
  KMI_CLOCK_END
}

// ---------------------------------------------------------------------------------------------
// Fix-up pass over the listed pixels.  A workgroup owns a range of screen segments and keeps a
// [K][D] table of fp64 deltas in LDS.  The pass is a chain of dependent memory round trips
// (counts -> list entries -> rows), so the two kinds of work run side by side on different waves:
//   ambiguous pixels, 16 per wave step: rows gathered straight into MFMA operand layout, split-f16
//            scores (h h' + (h l' + l h') / 2048, the arithmetic of the exact E-step kernels),
//            arg-max, label written; where the label changed the +x / -x update is applied from
//            the same registers;
//   changed pixels, 16 rows in flight per wave step, lane = channel: grid-rounded values added to
//            the new cluster's row and subtracted from the old one's (conflict-free 512-B LDS
//            atomics: 8 cycles per wave instruction, tools/hw_probes/lds_atomics.hip);
//   slab     the table leaves as one coalesced store; kmi_norm adds the slabs to the sums (all
//            exact, any order).  (64-bit float atomics to global memory instead: 2.4 M of them per
//            pass cost 20 us -- measured, dropped.)
// Work units are enumerated statically per segment (even slots: tiles of 16 ambiguous pixels, odd
// slots: batches of 16 changed pixels), so a wave knows where the list entries of its first unit
// are before the counts have arrived and loads them together with the counts.
// ---------------------------------------------------------------------------------------------
constexpr int kFixThreads = 512;
constexpr int kFixWaves = kFixThreads / 64;
constexpr int kFixMaxSeg = 16;        // screen segments per fix workgroup
constexpr int kFixCh = 5;             // channels per lane of a changed row: D <= 320
constexpr int kFixRows = 16;          // changed rows per wave step

__host__ __device__ inline int frag_stride(int dpad) { return dpad + 8; }   // halves; conflict-free rows
__host__ __device__ inline size_t fix_lds_bytes(int K, int D, int kpad, int dpad) {
  return (size_t)K * D * 8 + (size_t)2 * kpad * frag_stride(dpad) * 2 + (size_t)(3 * kFixMaxSeg + 4) * 4;
}

template <int MT16, int Q, int TAIL>
__global__ __launch_bounds__(kFixThreads) void kmi_fix(IncArgs a) {
  constexpr int QE = Q + TAIL;
  constexpr int D = 32 * Q + 2 * TAIL;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lg = lane >> 4, lc = lane & 15;
  const int img = blockIdx.y, g = blockIdx.x, G2 = gridDim.x;
  const int K = a.K, KD = K * D;
  const int fstride = frag_stride(a.dpad);
  double* table = reinterpret_cast<double*>(lds);                              // [K][D]
  _Float16* fh = reinterpret_cast<_Float16*>(lds + (size_t)KD * 8);            // [kpad][fstride]
  _Float16* fl = fh + (size_t)a.kpad * fstride;
  int* seg_base = reinterpret_cast<int*>(fl + (size_t)a.kpad * fstride);       // first pixel slot
  int* seg_namb = seg_base + kFixMaxSeg;
  int* seg_nchg = seg_namb + kFixMaxSeg;
  int* any_update = seg_nchg + kFixMaxSeg;

#define KMI_STAMP(i) if (a.trace && g == 5 && tid == 0) a.trace[i] = wall_clock64();
  KMI_STAMP(0)
  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + 31) / 32;
  const int s0 = (int)(((int64_t)a.G_screen * g) / G2), s1 = (int)(((int64_t)a.G_screen * (g + 1)) / G2);
  const int nseg = s1 - s0;
  if (tid < nseg) {
    const int32_t* c = a.counts + ((size_t)img * a.G_screen + s0 + tid) * 2;
    seg_base[tid] = (int)(32 * ((T * (s0 + tid)) / a.G_screen));
    seg_namb[tid] = c[0];
    seg_nchg[tid] = a.do_update ? c[1] : 0;
  }
  if (tid == 0) any_update[0] = 0;
  // A wave's units, round r: r * 8 + (wave in even rounds, 7 - wave in odd ones: the waves that had
  // the short units of one round get the first units of the next).  The list entries of a unit are
  // requested one round ahead -- those of round 0 together with the counts (entries past the count
  // are stale: never dereferenced).
  auto unit_of = [&](int r) { return r * kFixWaves + ((r & 1) ? kFixWaves - 1 - wave : wave); };
  unsigned long long nxt_ent = 0;
  int nxt_pix = -1;
  auto prefetch = [&](int u) {
    const int j = u % nseg, slot = u / nseg, b = slot >> 1;
    const int64_t lo = 32 * ((T * (s0 + j)) / a.G_screen);
    const int64_t cap = min(len, (int64_t)(32 * ((T * (s0 + j + 1)) / a.G_screen))) - lo;   // slots of the segment
    if (slot & 1) {
      if (a.do_update && lane < kFixRows && kFixRows * b + lane < cap) nxt_ent = a.chg[seg0 + lo + kFixRows * b + lane];
    } else if (16 * b + lc < cap) {
      nxt_pix = a.amb[seg0 + lo + 16 * b + lc];
    }
  };
  prefetch(unit_of(0));
  if (a.do_update)
    for (int i = tid; i < KD; i += kFixThreads) table[i] = 0.0;
  {                      // prototype fragments, rows padded by 16 B (bank-conflict-free A reads)
    const int per_row = a.dpad / 8;
    for (int i = tid; i < a.kpad * per_row; i += kFixThreads) {
      const int r = i / per_row, c8 = i % per_row;
      const size_t src = ((size_t)img * a.kpad + r) * a.dpad + 8 * c8;
      *reinterpret_cast<half8*>(fh + (size_t)r * fstride + 8 * c8) = *reinterpret_cast<const half8*>(a.cent_h + src);
      *reinterpret_cast<half8*>(fl + (size_t)r * fstride + 8 * c8) = *reinterpret_cast<const half8*>(a.cent_l + src);
    }
  }
  __syncthreads();
  KMI_STAMP(1)
  int smax = 0;                                          // slots of the longest segment
  for (int j = 0; j < nseg; ++j)
    smax = max(smax, max(2 * ((seg_namb[j] + 15) / 16) - 1, 2 * ((seg_nchg[j] + kFixRows - 1) / kFixRows)));
  bool touched = false;

  for (int r = 0; r * kFixWaves < nseg * smax; ++r) {
    const int u = unit_of(r);
    const unsigned long long cur_ent = nxt_ent;
    const int cur_pix = nxt_pix;
    if ((r + 1) * kFixWaves < nseg * smax) prefetch(unit_of(r + 1));
    if (u >= nseg * smax) continue;
    const int j = u % nseg, slot = u / nseg;
    const int b = slot >> 1;
    const int64_t lbase = seg0 + seg_base[j];
    if (!(slot & 1)) {
      // ---------------- a tile of ambiguous pixels ----------------
      const int n = seg_namb[j] - 16 * b;
      if (n <= 0) continue;
      int pix = -1, lold = -1;
      if (lc < n) pix = cur_pix;
      float raw[QE][8];
      if (pix >= 0) {
        const float* row = a.x + (size_t)(seg0 + pix) * D + 8 * lg;
        lold = a.labels[seg0 + pix];
#pragma unroll
        for (int s = 0; s < Q; ++s)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float2 f = *reinterpret_cast<const float2*>(row + 32 * s + 2 * e);
            raw[s][2 * e] = f.x;
            raw[s][2 * e + 1] = f.y;
          }
        if (TAIL) {
#pragma unroll
          for (int e = 0; e < 8; ++e) raw[Q][e] = 0.f;
          if (lg == 0) {
            const float2 f = *reinterpret_cast<const float2*>(row + 32 * Q);
            raw[Q][0] = f.x;
            raw[Q][1] = f.y;
          }
        }
      } else {
#pragma unroll
        for (int s = 0; s < QE; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) raw[s][e] = 0.f;
      }
      float best = -INFINITY;
      int bi = 0x7fffffff;
#pragma unroll 1
      for (int mt = 0; mt < MT16; ++mt) {
        float4a eh = {0.f, 0.f, 0.f, 0.f}, ex = eh, ey = eh;
        const _Float16* rh = fh + (size_t)(16 * mt + lc) * fstride + 8 * lg;
        const _Float16* rl = fl + (size_t)(16 * mt + lc) * fstride + 8 * lg;
#pragma unroll
        for (int s = 0; s < QE; ++s) {
          half8 bh, bl;                            // (split again per prototype tile: registers)
          split8(raw[s], bh, bl);
          const half8 ah = *reinterpret_cast<const half8*>(rh + 32 * s);
          const half8 al = *reinterpret_cast<const half8*>(rl + 32 * s);
          eh = mfma16(ah, bh, eh);
          ex = mfma16(ah, bl, ex);
          ey = mfma16(al, bh, ey);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int c = 16 * mt + 4 * lg + r;
          const float sdot = eh[r] + (ex[r] + ey[r]) * kSplitInv;
          if (c < K && sdot > best) { best = sdot; bi = c; }
        }
      }
#pragma unroll
      for (int o = 16; o <= 32; o <<= 1) {       // the 4 lane groups hold different prototype rows
        const float ob = __shfl_xor(best, o, kWave);
        const int oi = __shfl_xor(bi, o, kWave);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      const bool moved = pix >= 0 && bi != lold;
      if (lg == 0 && moved) {
        a.labels[seg0 + pix] = bi;
        if (a.labels_out64) a.labels_out64[seg0 + pix] = (int64_t)bi;
      }
      if (a.do_update && __ballot(moved) != 0ull) {
        touched = true;
        const bool sub_ok = lold >= 0 && lold < K;
        double* tn = table + (size_t)bi * D + 8 * lg;
        double* to = table + (size_t)(sub_ok ? lold : 0) * D + 8 * lg;
#pragma unroll
        for (int s = 0; s < Q; ++s) {
          __builtin_amdgcn_sched_barrier(0);            // (keeps 8 conversions live, not 72)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const double v = to_grid(raw[s][e]);
            if (moved) {
              __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(tn + 32 * s + e), v, 0, 0, false);
              if (sub_ok) __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(to + 32 * s + e), -v, 0, 0, false);
            }
          }
        }
        if (TAIL) {
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const double v = to_grid(raw[Q][e]);
            if (moved && lg == 0) {
              __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(tn + 32 * Q + e), v, 0, 0, false);
              if (sub_ok) __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(to + 32 * Q + e), -v, 0, 0, false);
            }
          }
        }
      }
    } else {
      // ---------------- a batch of changed pixels ----------------
      const int n = min(kFixRows, seg_nchg[j] - kFixRows * b);
      if (n <= 0) continue;
      touched = true;
      unsigned long long ent = 0;
      if (lane < n) ent = cur_ent;
      const uint32_t e_lo = (uint32_t)ent, e_hi = (uint32_t)(ent >> 32);
      float xv[kFixRows][kFixCh];
#pragma unroll
      for (int r = 0; r < kFixRows; ++r) {
        if (r < n) {                                      // wave-uniform
          const uint32_t pix = __builtin_amdgcn_readlane(e_lo, r);
          const float* row = a.x + (size_t)(seg0 + pix) * D + lane;
#pragma unroll
          for (int e = 0; e < kFixCh; ++e) xv[r][e] = lane + 64 * e < D ? row[64 * e] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < kFixRows; ++r) {
        __builtin_amdgcn_sched_barrier(0);
        if (r < n) {
          const uint32_t lab = __builtin_amdgcn_readlane(e_hi, r);
          const int lo = (int)(lab & 0xffff), ln = (int)(lab >> 16);     // 0xffff: no valid old label
          double* tn = table + (size_t)ln * D + lane;
          double* to = table + (size_t)(lo < K ? lo : 0) * D + lane;
#pragma unroll
          for (int e = 0; e < kFixCh; ++e) {
            if (lane + 64 * e < D) {
              const double v = to_grid(xv[r][e]);
              __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(tn + 64 * e), v, 0, 0, false);
              if (lo < K) __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(to + 64 * e), -v, 0, 0, false);
            }
          }
        }
      }
    }
  }
  if (!a.do_update) return;
  if (touched && lane == 0) any_update[0] = 1;
  KMI_STAMP(2)
  __syncthreads();
  KMI_STAMP(3)
  if (!any_update[0]) {
    if (tid == 0) a.slab_flag[(size_t)img * G2 + g] = 0;
    return;
  }
  {
    // slabs [n_img][K][G2][D]: kmi_norm reads the G2 rows of one cluster as one contiguous block
    double* dst = a.slabs + (size_t)img * K * G2 * D + (size_t)g * D;
    for (int i = tid; i < KD; i += kFixThreads) {
      const int k = i / D, d = i - k * D;
      dst[(size_t)k * G2 * D + d] = table[i];
    }
    if (tid == 0) a.slab_flag[(size_t)img * G2 + g] = 1;
    if (a.trace) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      KMI_STAMP(4)
      if (g == 5 && tid == 0) {
        int na = 0, nc = 0;
        for (int j = 0; j < nseg; ++j) { na += seg_namb[j]; nc += seg_nchg[j]; }
        a.trace[7] = na;
        a.trace[8] = nc;
      }
    }
  }
#undef KMI_STAMP
}

// ---------------------------------------------------------------------------------------------
// First stage of the slab reduction: grid (K, kRedParts, n_img) x 256; block (k, s) adds the rows k
// of its share of the flagged slabs (a contiguous block of memory) into part[img][k][s][D].  One
// block per cluster alone (36 CUs pulling 19 MB) took 11 us.
// ---------------------------------------------------------------------------------------------
constexpr int kRedParts = 8;
__global__ __launch_bounds__(256) void kmi_reduce(const double* __restrict__ slabs,
                                                  const int32_t* __restrict__ slab_flag, int n_slab,
                                                  int K, int D, double* __restrict__ part) {
  const int k = blockIdx.x, sp = blockIdx.y, img = blockIdx.z, t = threadIdx.x;
  const int g0 = (n_slab * sp) / kRedParts, g1 = (n_slab * (sp + 1)) / kRedParts;
  const double* base = slabs + ((size_t)img * K + k) * n_slab * D;
  const int32_t* fl = slab_flag + (size_t)img * n_slab;
  double acc[2] = {0.0, 0.0};
  for (int gq = g0; gq < g1; gq += 8) {
    double v[8][2];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int g = gq + u;
      const bool on = g < g1 && fl[g];
      v[u][0] = on && t < D ? base[(size_t)g * D + t] : 0.0;
      v[u][1] = on && t + 256 < D ? base[(size_t)g * D + t + 256] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; }
  }
  double* dst = part + (((size_t)img * K + k) * kRedParts + sp) * D;
  if (t < D) dst[t] = acc[0];
  if (t + 256 < D) dst[t + 256] = acc[1];
}

// ---------------------------------------------------------------------------------------------
// sums (+ the delta slabs of the fix pass) -> unit-norm prototypes: fp32 + split-f16 fragments
// (rows k >= K and channels d >= D of the fragment arrays are padding, written as zeros).  Empty
// cluster -> zero prototype, as the reference (0 / eps).  grid (kpad, n_img) x 1024: 4 groups of
// 256 threads share the slabs of a row, thread t owns channels t and t + 256.  The total goes
// back to partial table 0 (every addend is exact, so is the total, whatever the order).
// ---------------------------------------------------------------------------------------------
constexpr int kNormThreads = 1024;
__global__ __launch_bounds__(kNormThreads) void kmi_norm(double* __restrict__ sums64, int n_copy,
                                                        const double* __restrict__ slabs,
                                                        const int32_t* __restrict__ slab_flag, int n_slab,
                                                        int K, int D, int kpad, int dpad,
                                                        float* __restrict__ cent,
                                                        _Float16* __restrict__ cent_h,
                                                        _Float16* __restrict__ cent_l) {
  __shared__ double part[4][2][256];
  __shared__ int flag[256];
  __shared__ float red[4];
  const int k = blockIdx.x, img = blockIdx.y, tid = threadIdx.x;
  if (k >= K) {
    for (int d = tid; d < dpad; d += kNormThreads) {
      const size_t o = ((size_t)img * kpad + k) * dpad + d;
      cent_h[o] = (_Float16)0.f;
      cent_l[o] = (_Float16)0.f;
    }
    return;
  }
  const int t = tid & 255, grp = tid >> 8;
  if (tid < 256) flag[tid] = tid < n_slab ? (slab_flag ? slab_flag[(size_t)img * n_slab + tid] : 1) : 0;
  __syncthreads();
  double acc[2] = {0.0, 0.0};
  {
    const double* base = slabs + ((size_t)img * K + k) * n_slab * D;
    const size_t stride = (size_t)D;
    for (int g0 = grp; g0 < n_slab; g0 += 4 * 8) {
      double v[8][2];
#pragma unroll
      for (int u = 0; u < 8; ++u) {                    // 16 loads in flight per thread
        const int g = g0 + 4 * u;
        const bool on = g < n_slab && flag[g];
        v[u][0] = on && t < D ? base[(size_t)g * stride + t] : 0.0;
        v[u][1] = on && t + 256 < D ? base[(size_t)g * stride + t + 256] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc[0] += v[u][0]; acc[1] += v[u][1]; }
    }
  }
  part[grp][0][t] = acc[0];
  part[grp][1][t] = acc[1];
  __syncthreads();
  double* s = sums64 + ((size_t)img * K + k) * D;
  const size_t copy_stride = (size_t)gridDim.y * K * D;
  float v[2] = {0.f, 0.f};
  float ssq = 0.f;
  if (tid < 256) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int d = t + 256 * e;
      if (d < D) {
        double tot = part[0][e][t] + part[1][e][t] + part[2][e][t] + part[3][e][t];
        for (int c = 0; c < n_copy; ++c) tot += s[(size_t)c * copy_stride + d];
        s[d] = tot;
        for (int c = 1; c < n_copy; ++c) s[(size_t)c * copy_stride + d] = 0.0;
        v[e] = (float)tot;
        ssq += v[e] * v[e];
      }
    }
  }
  ssq = wave_sum(ssq);
  if (tid < 256 && (tid & 63) == 0) red[tid >> 6] = ssq;
  __syncthreads();
  if (tid >= 256) return;
  ssq = (red[0] + red[1]) + (red[2] + red[3]);
  const float n = sqrtf(ssq);
  const float dn = n >= kEps ? n : kEps;
  const int dmax = max(D, dpad);
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int d = t + 256 * e;
    if (d < dmax) {
      const float c = d < D ? v[e] / dn : 0.f;
      if (d < D) cent[((size_t)img * K + k) * D + d] = c;
      if (d < dpad) {
        _Float16 h, l;
        split_f16(c, h, l);
        const size_t o = ((size_t)img * kpad + k) * dpad + d;
        cent_h[o] = h;
        cent_l[o] = l;
      }
    }
  }
}

struct IncWs {
  size_t tau, amb, chg, counts, sums64, slabs, slab_flag, part, total;
};

constexpr int kIncGMax = 1024;
constexpr int kIncCopies = 4;
constexpr int kIncFixWgMax = 256;

IncWs inc_ws_layout(int64_t P, int D, int K, int n_img) {
  IncWs w{};
  size_t o = 0;
  w.tau = o; o = align_up(o + (size_t)P * 4, 256);
  w.amb = o; o = align_up(o + (size_t)P * 4, 256);
  w.chg = o; o = align_up(o + (size_t)P * 8, 256);
  w.counts = o; o = align_up(o + (size_t)n_img * kIncGMax * 2 * 4, 256);
  w.sums64 = o; o = align_up(o + (size_t)kIncCopies * n_img * K * D * 8, 256);
  const size_t n_slab_max = (size_t)(n_img > kIncFixWgMax ? n_img : kIncFixWgMax);   // G_fix * n_img
  w.slabs = o; o = align_up(o + n_slab_max * K * D * 8, 256);
  w.slab_flag = o; o = align_up(o + n_slab_max * 4, 256);
  w.part = o; o = align_up(o + (size_t)n_img * K * kRedParts * D * 8, 256);
  w.total = o;
  return w;
}

template <int Q, int TAIL>
int launch_seed(const IncArgs& a, int G, hipStream_t s) {
  auto kern = kmi_seed<Q, TAIL>;
  const size_t ldsb = seed_lds_bytes(32 * Q + 2 * TAIL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(kern, dim3(G, a.n_img), dim3(256), ldsb, s, a);
  return launch_status();
}

template <int MT16, int Q, int TAIL>
int launch_screen(const IncArgs& a, hipStream_t s) {
  auto kern = kmi_screen<MT16, Q, TAIL>;
  const size_t ldsb = screen_lds_bytes(Q, TAIL);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(kern, dim3(a.G, a.n_img), dim3(256), ldsb, s, a);
  return launch_status();
}

template <int MT16, int Q, int TAIL>
int launch_fix(const IncArgs& a, int G2, hipStream_t s) {
  auto kern = kmi_fix<MT16, Q, TAIL>;
  const size_t ldsb = fix_lds_bytes(a.K, a.D, a.kpad, a.dpad);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
  hipLaunchKernelGGL(kern, dim3(G2, a.n_img), dim3(kFixThreads), ldsb, s, a);
  return launch_status();
}

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

}  // namespace

// shapes the screened / incremental path covers (decided from host-visible arguments)
bool inc_shape(int64_t P, int D, int K, int n_img, int64_t max_seg_len) {
  const int q = D / 32, tl = D - 32 * q;
  if (!(q == 1 || q == 2 || q == 4 || q == 8) || !(tl == 0 || tl == 2)) return false;
  if (K < 1 || K > 64 || P >= ((int64_t)1 << 31) || max_seg_len > ((int64_t)1 << 20)) return false;
  const int kpad = 16 * ((K + 15) / 16), dpad = 32 * (q + (tl ? 1 : 0));
  if (fix_lds_bytes(K, D, kpad, dpad) > 158 * 1024) return false;
  // the extra kernels per iteration only pay off on large images
  const int64_t min_bytes = (int64_t)env_int("SPML_KMEANS_INC_MIN_MB", 32) << 20;
  (void)n_img;
  return max_seg_len * (int64_t)D * 4 >= min_bytes;
}

size_t inc_workspace_bytes(int64_t P, int D, int K, int n_img) { return inc_ws_layout(P, D, K, n_img).total; }

// workgroups per image of the seed / screen passes (also the stride of the profiling stamps)
int inc_grid(int n_img, int64_t max_seg_len) {
  const int64_t tiles = (max_seg_len + 31) / 32;
  int64_t g = (env_int("SPML_KMEANS_INC_WG", 768) + n_img - 1) / n_img;
  if (g > tiles) g = tiles;
  if (g > kIncGMax) g = kIncGMax;
  if (g < 1) g = 1;
  return (int)g;
}

// xh: room for (P / 32 + n_img + 1) hi tiles; lab32 [P]; cent_* as laid out by kmeans.hip for
// (kpad, dpad); ws: inc_workspace_bytes.  clocks: [iterations + 1][n_img][G][2] or null.
int inc_run(const float* x, int64_t P, int D, const int64_t* seg_off, int n_img, int64_t max_seg_len,
            int K, const int64_t* labels_init, int iterations, int64_t* labels_out, int32_t* lab32,
            float* cent_f, _Float16* cent_h, _Float16* cent_l, int kpad, int dpad, unsigned char* xh,
            void* ws, unsigned long long* clocks, hipStream_t s) {
  const IncWs wl = inc_ws_layout(P, D, K, n_img);
  unsigned char* base = static_cast<unsigned char*>(ws);
  const int Q = D / 32, TAIL = (D & 31) ? 1 : 0, MT16 = (K + 15) / 16;
  const int G = inc_grid(n_img, max_seg_len);
  IncArgs a{};
  a.x = x; a.x_bytes = P * (int64_t)D * 4; a.P = P; a.D = D; a.K = K; a.n_img = n_img;
  a.seg_off = seg_off; a.labels_in64 = labels_init; a.labels = lab32; a.labels_out64 = nullptr;
  a.xh = xh;
  a.tau = reinterpret_cast<float*>(base + wl.tau);
  a.sums64 = reinterpret_cast<double*>(base + wl.sums64);
  a.cent_h = cent_h; a.cent_l = cent_l; a.cent_f = cent_f; a.kpad = kpad; a.dpad = dpad;
  a.amb = reinterpret_cast<int32_t*>(base + wl.amb);
  a.chg = reinterpret_cast<unsigned long long*>(base + wl.chg);
  a.counts = reinterpret_cast<int32_t*>(base + wl.counts);
  a.G_screen = G;
  a.clock_stride = G;
  unsigned long long* trace_buf = nullptr;             // SPML_KMEANS_INC_TRACE=1: phase stamps (debug aid)
  const int want_trace = env_int("SPML_KMEANS_INC_TRACE", 0);
  if (want_trace && hipMalloc(&trace_buf, 16 * 16 * 8) != hipSuccess) return SPML_ERR_LAUNCH;
  a.trace = nullptr;
  a.n_copy = env_int("SPML_KMEANS_INC_COPIES", kIncCopies);
  if (a.n_copy < 1 || a.n_copy > kIncCopies) a.n_copy = kIncCopies;
  if (hipMemsetAsync(a.sums64, 0, (size_t)a.n_copy * n_img * K * D * 8, s) != hipSuccess) return SPML_ERR_LAUNCH;
  int rc;
  int pass_index = 0;
  auto stamp = [&]() {
    a.clocks = clocks ? clocks + (size_t)pass_index * 2 * G * n_img : nullptr;
    ++pass_index;
  };
  a.slabs = reinterpret_cast<double*>(base + wl.slabs);
  a.slab_flag = reinterpret_cast<int32_t*>(base + wl.slab_flag);
  double* part = reinterpret_cast<double*>(base + wl.part);
  auto norm = [&](int n_slab) {
    const double* src = a.slabs;
    const int32_t* flags = a.slab_flag;
    if (n_slab > 2 * kRedParts) {            // two stages: K x 8 blocks stream the slabs, then K blocks finish
      hipLaunchKernelGGL(kmi_reduce, dim3(K, kRedParts, n_img), dim3(256), 0, s, (const double*)a.slabs,
                         (const int32_t*)a.slab_flag, n_slab, K, D, part);
      src = part;
      flags = nullptr;
      n_slab = kRedParts;
    }
    hipLaunchKernelGGL(kmi_norm, dim3(kpad, n_img), dim3(kNormThreads), 0, s, a.sums64, a.n_copy, src, flags,
                       n_slab, K, D, kpad, dpad, cent_f, cent_h, cent_l);
  };
  // ---- seed ----
  {
    int gs = (env_int("SPML_KMEANS_INC_SEED_WG", 768) + n_img - 1) / n_img;
    if (gs > G) gs = G;
    a.G = gs;
    stamp();
    rc = SPML_ERR_UNSUPPORTED;
#define SPML_SEED(Q_)                                                              \
  if (Q == Q_) rc = TAIL ? launch_seed<Q_, 1>(a, gs, s) : launch_seed<Q_, 0>(a, gs, s);
    SPML_SEED(1) SPML_SEED(2) SPML_SEED(4) SPML_SEED(8)
#undef SPML_SEED
    if (rc != SPML_OK) return rc;
    norm(0);
  }
  a.G = G;
  const int fix_wg = (env_int("SPML_KMEANS_INC_FIX_WG", 256) + n_img - 1) / n_img;
  int G2 = fix_wg < G ? fix_wg : G;
  if (G2 > kIncFixWgMax / n_img) G2 = kIncFixWgMax / n_img > 0 ? kIncFixWgMax / n_img : 1;
  if ((G + G2 - 1) / G2 > kFixMaxSeg) G2 = (G + kFixMaxSeg - 1) / kFixMaxSeg;
  for (int it = 0; it < iterations; ++it) {
    const bool last = it == iterations - 1;
    a.make_lists = last ? 0 : 1;
    a.do_update = last ? 0 : 1;
    a.labels_out64 = last ? labels_out : nullptr;
    stamp();
    rc = SPML_ERR_UNSUPPORTED;
#define SPML_SCR(M_, Q_)                                                                     \
  if (MT16 == M_ && Q == Q_) rc = TAIL ? launch_screen<M_, Q_, 1>(a, s) : launch_screen<M_, Q_, 0>(a, s);
#define SPML_SCRQ(M_) SPML_SCR(M_, 1) SPML_SCR(M_, 2) SPML_SCR(M_, 4) SPML_SCR(M_, 8)
    SPML_SCRQ(1) SPML_SCRQ(2) SPML_SCRQ(3) SPML_SCRQ(4)
#undef SPML_SCR
    if (rc != SPML_OK) return rc;
    a.clocks = nullptr;
    a.trace = want_trace && it < 16 ? trace_buf + 16 * it : nullptr;
    rc = SPML_ERR_UNSUPPORTED;
#define SPML_SCR(M_, Q_)                                                                     \
  if (MT16 == M_ && Q == Q_) rc = TAIL ? launch_fix<M_, Q_, 1>(a, G2, s) : launch_fix<M_, Q_, 0>(a, G2, s);
    SPML_SCRQ(1) SPML_SCRQ(2) SPML_SCRQ(3) SPML_SCRQ(4)
#undef SPML_SCRQ
#undef SPML_SCR
    if (rc != SPML_OK) return rc;
    if (!last) norm(G2);
  }
  if (want_trace) {
    (void)hipStreamSynchronize(s);
    unsigned long long h[16 * 16];
    (void)hipMemcpy(h, trace_buf, sizeof(h), hipMemcpyDeviceToHost);
    for (int it = 0; it < iterations - 1 && it < 16; ++it) {
      const unsigned long long* t = h + 16 * it;
      fprintf(stderr, "kmi_fix it%d (wg 5: %llu amb, %llu chg): prologue %.1f  lists + rows + update %.1f  barrier %.1f  slab %.1f us\n",
              it + 1, t[7], t[8], (t[1] - t[0]) * 0.01, (t[2] - t[1]) * 0.01, (t[3] - t[2]) * 0.01,
              (t[4] - t[3]) * 0.01);
    }
    (void)hipFree(trace_buf);
  }
  return launch_status();
}

}  // namespace spml
