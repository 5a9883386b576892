// K1: L2-normalise an NCHW embedding map, transpose it to pixel-major rows,
// append the location features and normalise again -- one pass over HBM.
//
// Replaces the op chain of segsort/common.py:306-310, :346-352, :355-365 and
// general/common.py:101-120 of the reference (6 full-tensor passes there).
//
// Data movement: the NCHW map is read plane by plane with every wave reading 64
// consecutive pixels of one channel (256 B, coalesced); the [C][TPX] tile is
// transposed in LDS (row padding +1 -> conflict-free both ways) and written out
// as pixel-major rows, which for consecutive pixels are one contiguous block.
// Algorithmic HBM bytes per pixel: 4*C read + 4*C + 4*(C+L) written (L local-feature
// channels: 2 for the (y, x) location, 5 with the DensePose recipe's colours).
#include "common.hpp"

#include <stdlib.h>

namespace spml {
namespace {

// torch.linspace(0, 1, n)[i] (fp32, CPU kernel): symmetric evaluation around
// the midpoint; n == 1 -> 0.
__device__ __forceinline__ float linspace01(int i, int n) {
  if (n <= 1) return 0.f;
  const float step = 1.0f / (float)(n - 1);
  return (i < n / 2) ? step * (float)i : 1.0f - step * (float)(n - 1 - i);
}

struct K1Args {
  const float* emb;
  const float* loc;
  const int64_t* row_map;
  float* out_emb;
  float* out_loc;
  const float* d_out_emb;
  const float* d_out_loc;
  float* d_emb;
  int N, C, H, W, tpx, tiles_per_img;
  int L;                      // local-feature channels appended to the embedding (2 = (y, x))
};

constexpr int kMaxLocal = 8;

// LDS: tile[C][tpx+1] (+ for backward: g1[C][tpx+1], g2[C+L][tpx+1]) + scalars
// per-pixel scalars sc[.][tpx]: 0 = |x|, 1..L = local features, L+1 = |[e, local]|,
// L+2 = <o2, g2>, L+3 = <e, de>
__global__ __launch_bounds__(256) void k1_kernel(K1Args a) {
  constexpr bool BWD = false;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int C = a.C, L = a.L, tpx = a.tpx, ld = tpx + 1;
  const int HW = a.H * a.W;
  const int n = blockIdx.x / a.tiles_per_img;
  const int px0 = (blockIdx.x % a.tiles_per_img) * tpx;
  const int npx = min(tpx, HW - px0);
  const int tid = threadIdx.x;

  int64_t* rows = reinterpret_cast<int64_t*>(smem);  // [tpx] destination rows
  float* tile = smem + 2 * tpx;                     // [C][ld]
  float* g1 = tile + (size_t)C * ld;                // BWD only [C][ld]
  float* g2 = BWD ? g1 + (size_t)C * ld : g1;       // BWD only [C+L][ld]
  float* part = BWD ? g2 + (size_t)(C + L) * ld : g1;  // [4][256] partial sums
  float* sc = part + 4 * 256;                       // per-pixel scalars [L+4][tpx]

  const int px = tid % tpx;
  const int prt = tid / tpx;
  const int nprt = 256 / tpx;

  // destination rows
  if (tid < tpx) {
    int64_t r = -1;
    if (tid < npx) {
      const int64_t g = (int64_t)n * HW + px0 + tid;
      r = a.row_map ? a.row_map[g] : g;
    }
    rows[tid] = r;
  }
  // ---- load the NCHW tile (coalesced along pixels) ----
  const float* src = a.emb + (size_t)n * C * HW + px0;
  for (int c = prt; c < C; c += nprt)
    tile[c * ld + px] = (px < npx) ? src[(size_t)c * HW + px] : 0.f;
  __syncthreads();

  // ---- |x|^2 per pixel ----
  float ss = 0.f;
  for (int c = prt; c < C; c += nprt) {
    const float v = tile[c * ld + px];
    ss += v * v;
  }
  part[prt * tpx + px] = ss;
  __syncthreads();
  if (tid < tpx) {
    float t = 0.f;
    for (int i = 0; i < nprt; ++i) t += part[i * tpx + tid];
    const float n1 = sqrtf(t);
    sc[0 * tpx + tid] = n1;
    // local features of this pixel (given, or the (y, x) location generated in place)
    const int p = px0 + tid;
    for (int l = 0; l < L; ++l) {
      float v = 0.f;
      if (tid < npx) {
        if (a.loc) v = a.loc[((size_t)n * HW + p) * L + l];
        else v = l == 0 ? linspace01(p / a.W, a.H) - 0.5f : linspace01(p % a.W, a.W) - 0.5f;
      }
      sc[(1 + l) * tpx + tid] = v;
    }
  }
  __syncthreads();
  // ---- e = x / max(n1, eps), in place; |[e, loc]|^2 ----
  {
    const float n1 = sc[px];
    const float d1 = n1 >= kEps ? n1 : kEps;
    float s2 = 0.f;
    for (int c = prt; c < C; c += nprt) {
      const float e = tile[c * ld + px] / d1;
      tile[c * ld + px] = e;
      s2 += e * e;
    }
    part[prt * tpx + px] = s2;
  }
  __syncthreads();
  if (tid < tpx) {
    float t = 0.f;
    for (int i = 0; i < nprt; ++i) t += part[i * tpx + tid];
    for (int l = 0; l < L; ++l) {
      const float v = sc[(1 + l) * tpx + tid];
      t += v * v;
    }
    sc[(L + 1) * tpx + tid] = sqrtf(t);   // n2
  }
  __syncthreads();

  {
    // ---- write pixel-major rows: one wave per pixel row, lanes along the channels (no
    // integer division per element; a row is one contiguous, coalesced store) ----
    const int lane = tid & 63, wv = tid >> 6;
    if (a.out_emb) {
#pragma unroll 4
      for (int p = wv; p < npx; p += 4) {
        const int64_t r = rows[p];
        if (r < 0) continue;
        for (int c = lane; c < C; c += 64) a.out_emb[(size_t)r * C + c] = tile[c * ld + p];
      }
    }
    if (a.out_loc) {
      const int D = C + L;
#pragma unroll 4
      for (int p = wv; p < npx; p += 4) {
        const int64_t r = rows[p];
        if (r < 0) continue;
        const float n2 = sc[(L + 1) * tpx + p];
        const float d2 = n2 >= kEps ? n2 : kEps;
        for (int c = lane; c < D; c += 64) {
          const float v = c < C ? tile[c * ld + p] : sc[(1 + c - C) * tpx + p];
          a.out_loc[(size_t)r * D + c] = v / d2;
        }
      }
    }
    return;
  }
}

// Backward of K1.  With e = x / d1 (d1 = max(|x|, eps)), o2 = [e, loc] / d2 (d2 = max(|[e, loc]|, eps)):
//   de = g1 + (g2[:C] - o2[:C] <o2, g2>) / d2,   dx = (de - e <e, de>) / d1   (g / eps where a norm is < eps).
// Round 3: every global read of a tile is issued up front (the NCHW tile and the g2 rows into LDS, the
// g1 rows into registers in the row layout they arrive in), all per-pixel norms and dots come out of ONE
// reduction round, and only 2 C + L values per pixel sit in LDS -- 32-pixel tiles (whole 128-byte NCHW
// segments) at the LDS budget the 16-pixel tiles of round 2 needed; 4 barriers per tile instead of 9.
// LDS: rows[tpx] | tile[C][tpx+1] | g2[C+L][tpx+1] | part[3][256] | sc[L+5][tpx]
template <int RPW, int kMaxCG>   // pixel rows per wave = tpx / 4; 64-channel groups of a g1 row held in registers
__global__ __launch_bounds__(256, kMaxCG == 1 ? 8 : 4) void k1_bwd_kernel(K1Args a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int tpx = 4 * RPW;
  constexpr int ld = tpx + 1;
  const int C = a.C, L = a.L, D = C + L;
  const int HW = a.H * a.W;
  const int n = blockIdx.x / a.tiles_per_img;
  const int px0 = (blockIdx.x % a.tiles_per_img) * tpx;
  const int npx = min(tpx, HW - px0);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  int64_t* rows = reinterpret_cast<int64_t*>(smem);
  float* tile = smem + 2 * tpx;                     // [C][ld]   x, NCHW tile
  float* g2 = tile + (size_t)C * ld;                // [D][ld]   upstream gradient of the rows with location; later de
  float* part = g2 + (size_t)D * ld;                // [3][256]
  float* sc = part + 3 * 256;                       // [L + 5][tpx]
  const int px = tid % tpx, prt = tid / tpx;
  constexpr int nprt = 256 / tpx;

  // destination row of pixel p of the tile (every wave computes the ones it needs itself: no barrier)
  auto row_of = [&](int p) -> int64_t {
    if (p >= npx) return -1;
    const int64_t g = (int64_t)n * HW + px0 + p;
    return a.row_map ? a.row_map[g] : g;
  };
  if (tid < tpx) rows[tid] = row_of(tid);
  // ---- all loads of the tile, issued together ----
  const float* src = a.emb + (size_t)n * C * HW + px0;
  for (int c = prt; c < C; c += nprt)
    tile[c * ld + px] = (px < npx) ? src[(size_t)c * HW + px] : 0.f;
  float g1r[RPW][kMaxCG];
  const int ncg = (C + 63) >> 6;
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int p = wv + 4 * i;
    const int64_t r = row_of(p);
#pragma unroll
    for (int q = 0; q < kMaxCG; ++q) {
      const int c = lane + 64 * q;
      g1r[i][q] = (q < ncg && c < C && r >= 0 && a.d_out_emb) ? a.d_out_emb[(size_t)r * C + c] : 0.f;
    }
    for (int c = lane; c < D; c += 64)
      g2[c * ld + p] = (r >= 0 && a.d_out_loc) ? a.d_out_loc[(size_t)r * D + c] : 0.f;
  }
  if (tid < tpx) {                                   // local features of the pixel
    const int p = px0 + tid;
    for (int l = 0; l < L; ++l) {
      float v = 0.f;
      if (tid < npx) {
        if (a.loc) v = a.loc[((size_t)n * HW + p) * L + l];
        else v = l == 0 ? linspace01(p / a.W, a.H) - 0.5f : linspace01(p % a.W, a.W) - 0.5f;
      }
      sc[(1 + l) * tpx + tid] = v;
    }
  }
  __syncthreads();
  // ---- one reduction round: |x|^2 and <x, g2[:C]> per pixel ----
  {
    float ss = 0.f, sg = 0.f;
    for (int c = prt; c < C; c += nprt) {
      const float v = tile[c * ld + px];
      ss += v * v;
      sg += v * g2[c * ld + px];
    }
    part[prt * tpx + px] = ss;
    part[256 + prt * tpx + px] = sg;
  }
  __syncthreads();
  if (tid < tpx) {
    float ss = 0.f, sg = 0.f;
    for (int i = 0; i < nprt; ++i) { ss += part[i * tpx + tid]; sg += part[256 + i * tpx + tid]; }
    const float n1 = sqrtf(ss);
    const float d1 = n1 >= kEps ? n1 : kEps;
    float e2 = ss / (d1 * d1), lg = 0.f;             // |e|^2, <loc, g2[C:]>
    for (int l = 0; l < L; ++l) {
      const float v = sc[(1 + l) * tpx + tid];
      e2 += v * v;
      lg += v * g2[(C + l) * ld + tid];
    }
    const float n2 = sqrtf(e2);
    const float d2 = n2 >= kEps ? n2 : kEps;
    sc[0 * tpx + tid] = n1;
    sc[(L + 1) * tpx + tid] = n2;
    sc[(L + 2) * tpx + tid] = (sg / d1 + lg) / d2;   // <o2, g2>
  }
  __syncthreads();
  // ---- de in the row layout (lane = channel), written over g2; <e, de> per pixel by a wave reduction ----
#pragma unroll
  for (int i = 0; i < RPW; ++i) {
    const int p = wv + 4 * i;
    const float n1 = sc[p], n2 = sc[(L + 1) * tpx + p], t2 = sc[(L + 2) * tpx + p];
    const float d1 = n1 >= kEps ? n1 : kEps;
    const bool ok2 = n2 >= kEps;
    const float d2 = ok2 ? n2 : kEps;
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < kMaxCG; ++q) {
      const int c = lane + 64 * q;
      if (q < ncg && c < C) {
        const float e = tile[c * ld + p] / d1;
        const float gv = g2[c * ld + p];
        const float dv = ok2 ? (gv - (e / d2) * t2) / d2 : gv / kEps;
        const float de = g1r[i][q] + dv;
        g2[c * ld + p] = de;
        s += e * de;
      }
    }
    s = wave_sum(s);
    if (lane == 0) sc[(L + 3) * tpx + p] = s;        // <e, de>
  }
  __syncthreads();
  // ---- dx = (de - e <e, de>) / d1 (or de / eps), written back NCHW (coalesced) ----
  {
    const float n1 = sc[px];
    const bool ok1 = n1 >= kEps;
    const float d1 = ok1 ? n1 : kEps;
    const float t1 = sc[(L + 3) * tpx + px];
    float* dst = a.d_emb + (size_t)n * C * HW + px0;
    if (px < npx) {
      const bool keep = rows[px] >= 0;
      for (int c = prt; c < C; c += nprt) {
        const float de = g2[c * ld + px];
        const float e = tile[c * ld + px] / d1;
        const float dx = ok1 ? (de - e * t1) / d1 : de / kEps;
        dst[(size_t)c * HW + px] = keep ? dx : 0.f;
      }
    }
  }
}

size_t k1_bwd_lds_bytes(int C, int L, int tpx) {
  const size_t ld = tpx + 1;
  return ((size_t)C * ld + (size_t)(C + L) * ld + 3 * 256 + (size_t)(L + 5) * tpx) * sizeof(float) +
         (size_t)tpx * sizeof(int64_t) + 16;
}

size_t k1_lds_bytes(int C, int L, int tpx, bool bwd) {
  const size_t ld = tpx + 1;
  size_t f = (size_t)C * ld;
  if (bwd) f += (size_t)C * ld + (size_t)(C + L) * ld;
  f += 4 * 256 + (size_t)(L + 4) * tpx;
  return f * sizeof(float) + (size_t)tpx * sizeof(int64_t) + 16;
}

// backward: the widest tile (32 / 16 / 8 pixels) whose LDS image stays under ~20 KB -- many resident blocks
// hide the barrier-separated phases (C = 64: 16 pixels, 115 us at 16 x 64 x 130 x 130 against 128 us for the
// round-2 kernel, 32 pixels 118 us; C = 512: 8 pixels, 37 KB, 562 us at 2 x 512 x 258 x 258 against 1 210 us);
// C <= 512 (register-resident g1 rows)
int k1_bwd_launch(K1Args a, hipStream_t s) {
  if (!a.emb || a.N <= 0 || a.C <= 0 || a.H <= 0 || a.W <= 0) return SPML_ERR_INVALID_ARG;
  if (a.L < 1 || a.L > kMaxLocal || (!a.loc && a.L != 2)) return SPML_ERR_INVALID_ARG;
  if (a.C > 512) return SPML_ERR_UNSUPPORTED;
  int tpx = 32;
  while (tpx > 8 && k1_bwd_lds_bytes(a.C, a.L, tpx) > 20 * 1024) tpx >>= 1;
  if (const char* e = getenv("SPML_K1_BWD_TPX")) { const int v = atoi(e); if (v == 8 || v == 16 || v == 32) tpx = v; }
  const size_t lds = k1_bwd_lds_bytes(a.C, a.L, tpx);
  if (lds > 160 * 1024) return SPML_ERR_UNSUPPORTED;
  const int HW = a.H * a.W;
  a.tpx = tpx;
  a.tiles_per_img = (HW + tpx - 1) / tpx;
  const dim3 grid((unsigned)(a.N * a.tiles_per_img));
#define SPML_K1B(RPW_, CG_)                                                                        \
  {                                                                                                \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_bwd_kernel<RPW_, CG_>),            \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);               \
    hipLaunchKernelGGL((k1_bwd_kernel<RPW_, CG_>), grid, dim3(256), lds, s, a);                    \
  }
#define SPML_K1B_CG(RPW_)                                                                          \
  {                                                                                                \
    if (a.C <= 64) SPML_K1B(RPW_, 1) else if (a.C <= 128) SPML_K1B(RPW_, 2) else SPML_K1B(RPW_, 8) \
  }
  if (tpx == 32) SPML_K1B_CG(8) else if (tpx == 16) SPML_K1B_CG(4) else SPML_K1B_CG(2)
#undef SPML_K1B_CG
#undef SPML_K1B
  return launch_status();
}

int k1_launch(K1Args a, bool bwd, hipStream_t s) {
  if (!a.emb || a.N <= 0 || a.C <= 0 || a.H <= 0 || a.W <= 0) return SPML_ERR_INVALID_ARG;
  if (a.L < 1 || a.L > kMaxLocal || (!a.loc && a.L != 2)) return SPML_ERR_INVALID_ARG;
  // pixels per tile: the phases of a block are serialised by barriers, so what hides the latency is
  // the number of resident blocks -- keep a tile under ~20 KB of LDS (>= 7 blocks per CU) down to 8
  // pixels (32-byte NCHW segments); measured 2.1x (backward, C = 64) to 2.8x (forward, C = 512)
  // against the fixed 64-pixel tile (tools/bench_k1.py)
  int tpx = 64;
  while (tpx > 8 && k1_lds_bytes(a.C, a.L, tpx, bwd) > 20 * 1024) tpx >>= 1;
  if (const char* e = getenv("SPML_K1_TPX")) tpx = atoi(e) >= 8 ? atoi(e) : tpx;     // tuning aid
  while (tpx > 4 && k1_lds_bytes(a.C, a.L, tpx, bwd) > 150 * 1024) tpx >>= 1;
  if (k1_lds_bytes(a.C, a.L, tpx, bwd) > 160 * 1024) return SPML_ERR_UNSUPPORTED;
  const int HW = a.H * a.W;
  a.tpx = tpx;
  a.tiles_per_img = (HW + tpx - 1) / tpx;
  const size_t lds = k1_lds_bytes(a.C, a.L, tpx, bwd);
  const dim3 grid((unsigned)(a.N * a.tiles_per_img));
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k1_kernel),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(k1_kernel, grid, dim3(256), lds, s, a);
  return launch_status();
}

// ---- K1 on channels-last input: the network's embedding map arrives as [N, H, W, C] storage when the
// backbone runs channels-last, i.e. the pixel rows are already contiguous and K1 is a row-wise stream:
// LPR lanes share a row (VPL float4 each, C = 4 * LPR * VPL), 64 / LPR rows per wave, norms and dots by
// xor shuffles inside the LPR lanes.  No transposition, no LDS.  Same arithmetic as k1_kernel /
// k1_bwd_kernel (which keep serving NCHW input).
template <int LPR, int VPL>
__device__ __forceinline__ float row_sum(float v) {
#pragma unroll
  for (int o = LPR / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

template <int LPR, int VPL, bool BWD>
__global__ __launch_bounds__(256) void k1_nhwc_kernel(K1Args a) {
  constexpr int C = 4 * LPR * VPL, RW = 64 / LPR;
  const int L = a.L, D = C + L, HW = a.H * a.W;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int lr = lane % LPR, sub = lane / LPR;
  const int64_t total = (int64_t)a.N * HW;
  for (int64_t g0 = ((int64_t)blockIdx.x * 4 + wv) * RW; g0 < total; g0 += (int64_t)gridDim.x * 4 * RW) {
    const int64_t g = g0 + sub;
    const bool live = g < total;
    const int64_t r = live ? (a.row_map ? a.row_map[g] : g) : -1;
    float4v x[VPL];
#pragma unroll
    for (int v = 0; v < VPL; ++v)
      x[v] = live ? *reinterpret_cast<const float4v*>(a.emb + (size_t)g * C + 4 * (lr + LPR * v)) : float4v{0.f, 0.f, 0.f, 0.f};
    // local features of the pixel on the first L lanes of the row group (L <= 8 <= LPR * 4 always holds
    // for the supported C; lanes lr < L carry one each)
    float lf = 0.f;
    if (live && lr < L) {
      if (a.loc) lf = a.loc[(size_t)g * L + lr];
      else {
        const int p = (int)(g % HW);
        lf = lr == 0 ? linspace01(p / a.W, a.H) - 0.5f : linspace01(p % a.W, a.W) - 0.5f;
      }
    }
    float ss = 0.f;
#pragma unroll
    for (int v = 0; v < VPL; ++v)
#pragma unroll
      for (int e = 0; e < 4; ++e) ss += x[v][e] * x[v][e];
    ss = row_sum<LPR, VPL>(ss);
    const float n1 = sqrtf(ss);
    const bool ok1 = n1 >= kEps;
    const float d1 = ok1 ? n1 : kEps;
    if constexpr (!BWD) {
      float s2 = lf * lf;
#pragma unroll
      for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) { x[v][e] = x[v][e] / d1; s2 += x[v][e] * x[v][e]; }
      s2 = row_sum<LPR, VPL>(s2);
      const float n2 = sqrtf(s2);
      const float d2 = n2 >= kEps ? n2 : kEps;
      if (r < 0) continue;
      if (a.out_emb) {
#pragma unroll
        for (int v = 0; v < VPL; ++v)
          *reinterpret_cast<float4v*>(a.out_emb + (size_t)r * C + 4 * (lr + LPR * v)) = x[v];
      }
      if (a.out_loc) {
        float* dst = a.out_loc + (size_t)r * D;          // rows of C + L floats: 4-byte aligned only
#pragma unroll
        for (int v = 0; v < VPL; ++v)
#pragma unroll
          for (int e = 0; e < 4; ++e) dst[4 * (lr + LPR * v) + e] = x[v][e] / d2;
        if (lr < L) dst[C + lr] = lf / d2;
      }
    } else {
      // e2 = |e|^2 + |loc|^2 as the NCHW kernel takes it: |x|^2 / d1^2 + |loc|^2
      const float e2 = ss / (d1 * d1) + row_sum<LPR, VPL>(lf * lf);
      const float n2 = sqrtf(e2);
      const bool ok2 = n2 >= kEps;
      const float d2 = ok2 ? n2 : kEps;
      const bool has = r >= 0;
      float4v g2[VPL], g1[VPL];
      float gl = 0.f;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        g1[v] = (has && a.d_out_emb) ? *reinterpret_cast<const float4v*>(a.d_out_emb + (size_t)r * C + 4 * (lr + LPR * v))
                                      : float4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int e = 0; e < 4; ++e)
          g2[v][e] = (has && a.d_out_loc) ? a.d_out_loc[(size_t)r * D + 4 * (lr + LPR * v) + e] : 0.f;
      }
      if (has && a.d_out_loc && lr < L) gl = a.d_out_loc[(size_t)r * D + C + lr];
      float sg = lf * gl * d1;                               // (<x, g2[:C]> / d1 + <loc, g2[C:]>) * d1, see below
#pragma unroll
      for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) sg += x[v][e] * g2[v][e];
      const float t2 = row_sum<LPR, VPL>(sg) / d1 / d2;     // <o2, g2>
      float s = 0.f;
      float4v de[VPL];
#pragma unroll
      for (int v = 0; v < VPL; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = x[v][e] / d1;
          const float dv = ok2 ? (g2[v][e] - (ev / d2) * t2) / d2 : g2[v][e] / kEps;
          de[v][e] = g1[v][e] + dv;
          s += ev * de[v][e];
        }
      const float t1 = row_sum<LPR, VPL>(s);                // <e, de>
      if (!live) continue;
#pragma unroll
      for (int v = 0; v < VPL; ++v) {
        float4v dx;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float ev = x[v][e] / d1;
          const float d = ok1 ? (de[v][e] - ev * t1) / d1 : de[v][e] / kEps;
          dx[e] = has ? d : 0.f;
        }
        *reinterpret_cast<float4v*>(a.d_emb + (size_t)g * C + 4 * (lr + LPR * v)) = dx;
      }
    }
  }
}

// channel counts the channels-last kernels take (C = 4 * LPR * VPL with LPR a power of two <= 64)
inline bool k1_nhwc_shape(int C, int L) {
  return L >= 1 && L <= kMaxLocal && (C == 16 || C == 32 || C == 64 || C == 128 || C == 256 || C == 512) &&
         L <= (C >= 256 ? 64 : C / 4);                    // one local feature per lane of a row group
}

template <bool BWD>
int k1_nhwc_launch(K1Args a, hipStream_t s) {
  if (!a.emb || a.N <= 0 || a.H <= 0 || a.W <= 0 || !k1_nhwc_shape(a.C, a.L) || (!a.loc && a.L != 2))
    return SPML_ERR_INVALID_ARG;
  const int lpr = a.C >= 256 ? 64 : a.C / 4;
  const int64_t rows = (int64_t)a.N * a.H * a.W;
  const int64_t per_block = 4 * (64 / lpr);
  int64_t blocks = (rows + per_block - 1) / per_block;
  if (blocks > 16384) blocks = 16384;                     // grid-stride beyond that
  const dim3 grid((unsigned)blocks);
#define SPML_K1N(LPR_, VPL_) hipLaunchKernelGGL((k1_nhwc_kernel<LPR_, VPL_, BWD>), grid, dim3(256), 0, s, a)
  switch (a.C) {
    case 16: SPML_K1N(4, 1); break;
    case 32: SPML_K1N(8, 1); break;
    case 64: SPML_K1N(16, 1); break;
    case 128: SPML_K1N(32, 1); break;
    case 256: SPML_K1N(64, 1); break;
    default: SPML_K1N(64, 2); break;
  }
#undef SPML_K1N
  return launch_status();
}

// ---- plain row normalise (one wave per row) ------------------------------
__global__ __launch_bounds__(256) void rownorm_fwd(const float* x, int64_t rows, int D,
                                                   float* y) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)r * D;
  float ss = 0.f;
  for (int d = lane; d < D; d += 64) ss += xr[d] * xr[d];
  ss = wave_sum(ss);
  const float n = sqrtf(ss);
  const float dn = n >= kEps ? n : kEps;
  for (int d = lane; d < D; d += 64) y[(size_t)r * D + d] = xr[d] / dn;
}

__global__ __launch_bounds__(256) void rownorm_bwd(const float* x, const float* dy,
                                                   int64_t rows, int D, float* dx) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + (size_t)r * D;
  const float* gr = dy + (size_t)r * D;
  float ss = 0.f, sg = 0.f;
  for (int d = lane; d < D; d += 64) {
    ss += xr[d] * xr[d];
    sg += xr[d] * gr[d];
  }
  ss = wave_sum(ss);
  sg = wave_sum(sg);
  const float n = sqrtf(ss);
  if (n >= kEps) {
    // y = x/n ; dx = (dy - y <y,dy>)/n
    const float t = sg / n;   // <y, dy>
    for (int d = lane; d < D; d += 64)
      dx[(size_t)r * D + d] = (gr[d] - (xr[d] / n) * t) / n;
  } else {
    for (int d = lane; d < D; d += 64) dx[(size_t)r * D + d] = gr[d] / kEps;
  }
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" int spml_normalize_concat_local_f32(const float* emb, int N, int C, int H, int W,
                                               const float* local, int L,
                                               const int64_t* row_map, float* out_emb,
                                               float* out_loc, void* stream) {
  if (!out_emb && !out_loc) return SPML_ERR_INVALID_ARG;
  K1Args a{};
  a.emb = emb; a.loc = local; a.row_map = row_map; a.out_emb = out_emb; a.out_loc = out_loc;
  a.N = N; a.C = C; a.H = H; a.W = W; a.L = L;
  return k1_launch(a, false, (hipStream_t)stream);
}

extern "C" int spml_normalize_concat_loc_f32(const float* emb, int N, int C, int H, int W,
                                             const float* loc, const int64_t* row_map,
                                             float* out_emb, float* out_loc, void* stream) {
  return spml_normalize_concat_local_f32(emb, N, C, H, W, loc, 2, row_map, out_emb, out_loc,
                                         stream);
}

extern "C" int spml_normalize_concat_local_bwd_f32(const float* emb, int N, int C, int H, int W,
                                                   const float* local, int L,
                                                   const int64_t* row_map,
                                                   const float* d_out_emb,
                                                   const float* d_out_loc, float* d_emb,
                                                   void* stream) {
  if (!d_emb) return SPML_ERR_INVALID_ARG;
  K1Args a{};
  a.emb = emb; a.loc = local; a.row_map = row_map;
  a.d_out_emb = d_out_emb; a.d_out_loc = d_out_loc; a.d_emb = d_emb;
  a.N = N; a.C = C; a.H = H; a.W = W; a.L = L;
  return k1_bwd_launch(a, (hipStream_t)stream);
}

extern "C" int spml_normalize_concat_loc_bwd_f32(const float* emb, int N, int C, int H, int W,
                                                 const float* loc, const int64_t* row_map,
                                                 const float* d_out_emb,
                                                 const float* d_out_loc, float* d_emb,
                                                 void* stream) {
  return spml_normalize_concat_local_bwd_f32(emb, N, C, H, W, loc, 2, row_map, d_out_emb,
                                             d_out_loc, d_emb, stream);
}

extern "C" int spml_normalize_concat_local_nhwc_supported(int C, int L) { return k1_nhwc_shape(C, L) ? 1 : 0; }

extern "C" int spml_normalize_concat_local_nhwc_f32(const float* emb, int N, int C, int H, int W,
                                                    const float* local, int L, const int64_t* row_map,
                                                    float* out_emb, float* out_loc, void* stream) {
  if (!out_emb && !out_loc) return SPML_ERR_INVALID_ARG;
  if (!k1_nhwc_shape(C, L)) return SPML_ERR_UNSUPPORTED;
  K1Args a{};
  a.emb = emb; a.loc = local; a.row_map = row_map; a.out_emb = out_emb; a.out_loc = out_loc;
  a.N = N; a.C = C; a.H = H; a.W = W; a.L = L;
  return k1_nhwc_launch<false>(a, (hipStream_t)stream);
}

extern "C" int spml_normalize_concat_local_nhwc_bwd_f32(const float* emb, int N, int C, int H, int W,
                                                        const float* local, int L, const int64_t* row_map,
                                                        const float* d_out_emb, const float* d_out_loc,
                                                        float* d_emb, void* stream) {
  if (!d_emb) return SPML_ERR_INVALID_ARG;
  if (!k1_nhwc_shape(C, L)) return SPML_ERR_UNSUPPORTED;
  K1Args a{};
  a.emb = emb; a.loc = local; a.row_map = row_map;
  a.d_out_emb = d_out_emb; a.d_out_loc = d_out_loc; a.d_emb = d_emb;
  a.N = N; a.C = C; a.H = H; a.W = W; a.L = L;
  return k1_nhwc_launch<true>(a, (hipStream_t)stream);
}

extern "C" int spml_normalize_rows_f32(const float* x, int64_t rows, int D, float* y,
                                       void* stream) {
  if (!x || !y || rows < 0 || D <= 0) return SPML_ERR_INVALID_ARG;
  if (rows == 0) return SPML_OK;
  hipLaunchKernelGGL(rownorm_fwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, rows, D, y);
  return launch_status();
}

extern "C" int spml_normalize_rows_bwd_f32(const float* x, const float* dy, int64_t rows,
                                           int D, float* dx, void* stream) {
  if (!x || !dy || !dx || rows < 0 || D <= 0) return SPML_ERR_INVALID_ARG;
  if (rows == 0) return SPML_OK;
  hipLaunchKernelGGL(rownorm_bwd, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, x, dy, rows, D, dx);
  return launch_status();
}
