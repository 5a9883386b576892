// Cross-entropy of bilinearly up-sampled logits, forward and backward, without the full-resolution tensors.
//
// Replaces, in the softmax head of the training step (spml/models/predictions/segsort_softmax.py:112-131),
//     logits = F.interpolate(logits, size=labels.shape[-2:], mode='bilinear')      [N, C, H, W]
//     loss   = CrossEntropyLoss(ignore_index)(logits, labels)
// which for 16 x 21 x 513 x 513 moves ~4 GB per step through seven framework kernels (up-sample, layout copy,
// log-softmax, NLL, and their backward passes: 4 ms).  Here the [N, C, H, W] logits never exist:
//   forward   one thread per output pixel interpolates its C logits from the four low-resolution neighbours
//             (the [N, h, w, C] map is L2-resident), reduces them to logsumexp - logit[label], keeps the
//             logsumexp (4 B per pixel) and adds to per-block partial sums (fixed order: deterministic);
//   backward  one thread per LOW-resolution pixel gathers from the ~(H/h + 1)^2 output pixels it feeds:
//             softmax(c) - [c == label], weighted by the pixel's bilinear coefficient -- no atomics, the
//             probabilities are recomputed from the kept logsumexp.
// Interpolation arithmetic as ATen's upsample_bilinear2d (align_corners = False, size given):
//   src = max(scale * (dst + 0.5) - 0.5, 0), scale = in / out, i0 = (int)src, i1 = i0 + (i0 < in - 1).
#include <algorithm>
#include <cmath>

#include "common.hpp"

namespace spml {
namespace {

struct UceArgs {
  const float* logits;       // [N][h][w][C]
  const int64_t* labels;     // [N][H][W]
  float* lse;                // [N][H][W] logsumexp of the interpolated logits (0 for ignored pixels)
  const float* lse_in;       // (backward)
  float* partial;            // [blocks][2] sum of losses, number of counted pixels
  const float* scale;        // backward: d_loss / count
  float* d_logits;           // [N][h][w][C]
  int N, C, h, w, H, W;
  int64_t ignore_index;
  float rh, rw;              // h / H, w / W
};

struct Src {
  int i0, i1;
  float l0, l1;
};
__device__ __forceinline__ Src source(float scale, int dst, int in) {
  const float r = fmaxf(scale * ((float)dst + 0.5f) - 0.5f, 0.f);
  Src s;
  s.i0 = (int)r;
  s.i1 = s.i0 + (s.i0 < in - 1 ? 1 : 0);
  s.l1 = r - (float)s.i0;
  s.l0 = 1.f - s.l1;
  return s;
}

template <int CP>
__device__ __forceinline__ void interpolate(const UceArgs& a, int n, const Src& sy, const Src& sx, float (&v)[CP]) {
  const float* p00 = a.logits + (((size_t)n * a.h + sy.i0) * a.w + sx.i0) * a.C;
  const float* p01 = a.logits + (((size_t)n * a.h + sy.i0) * a.w + sx.i1) * a.C;
  const float* p10 = a.logits + (((size_t)n * a.h + sy.i1) * a.w + sx.i0) * a.C;
  const float* p11 = a.logits + (((size_t)n * a.h + sy.i1) * a.w + sx.i1) * a.C;
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    if (c < a.C)
      v[c] = sy.l0 * (sx.l0 * p00[c] + sx.l1 * p01[c]) + sy.l1 * (sx.l0 * p10[c] + sx.l1 * p11[c]);
    else
      v[c] = -INFINITY;
  }
}

template <int CP>
__global__ __launch_bounds__(256) void uce_fwd(UceArgs a) {
  const int64_t total = (int64_t)a.N * a.H * a.W;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float loss = 0.f, cnt = 0.f;
  if (i < total) {
    const int ox = (int)(i % a.W);
    const int64_t t = i / a.W;
    const int oy = (int)(t % a.H), n = (int)(t / a.H);
    const int64_t lab = a.labels[i];
    float lse = 0.f;
    if (lab != a.ignore_index) {
      float v[CP];
      interpolate<CP>(a, n, source(a.rh, oy, a.h), source(a.rw, ox, a.w), v);
      float m = v[0];
#pragma unroll
      for (int c = 1; c < CP; ++c) m = fmaxf(m, v[c]);
      float s = 0.f, vl = 0.f;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        s += __expf(v[c] - m);
        vl = (c == (int)lab) ? v[c] : vl;
      }
      lse = m + __logf(s);
      loss = lse - vl;
      cnt = 1.f;
    }
    a.lse[i] = lse;
  }
  __shared__ float sh[2][4];
  loss = wave_sum(loss);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = loss; sh[1][threadIdx.x >> 6] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    a.partial[2 * (size_t)blockIdx.x] = (sh[0][0] + sh[0][1]) + (sh[0][2] + sh[0][3]);
    a.partial[2 * (size_t)blockIdx.x + 1] = (sh[1][0] + sh[1][1]) + (sh[1][2] + sh[1][3]);
  }
}

// out[0] = sum of the losses, out[1] = counted pixels, out[2] = mean (NaN without counted pixels, as ATen)
__global__ __launch_bounds__(1024) void uce_reduce(const float* __restrict__ partial, int64_t blocks, float* out) {
  __shared__ double sh[2][16];
  double s = 0.0, c = 0.0;
  for (int64_t i = threadIdx.x; i < blocks; i += 1024) { s += partial[2 * i]; c += partial[2 * i + 1]; }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); c += __shfl_xor(c, o, 64); }
  if ((threadIdx.x & 63) == 0) { sh[0][threadIdx.x >> 6] = s; sh[1][threadIdx.x >> 6] = c; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double ts = 0.0, tc = 0.0;
    for (int i = 0; i < 16; ++i) { ts += sh[0][i]; tc += sh[1][i]; }
    out[0] = (float)ts;
    out[1] = (float)tc;
    out[2] = (float)(ts / tc);
  }
}

// first / last output index whose interpolation can touch low-resolution index i (a superset; the loop
// re-derives the exact source of every candidate)
__device__ __forceinline__ void out_range(float scale, int i, int out, int& lo, int& hi) {
  const float inv = 1.0f / scale;
  lo = max(0, (int)floorf(((float)i - 1.f + 0.5f) * inv - 0.5f) - 1);
  hi = min(out - 1, (int)ceilf(((float)i + 1.f + 0.5f) * inv - 0.5f) + 1);
}

template <int CP>
__global__ __launch_bounds__(256) void uce_bwd(UceArgs a) {
  const int64_t total = (int64_t)a.N * a.h * a.w;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int ix = (int)(i % a.w);
  const int64_t t = i / a.w;
  const int iy = (int)(t % a.h), n = (int)(t / a.h);
  float g[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) g[c] = 0.f;
  int oy0, oy1, ox0, ox1;
  out_range(a.rh, iy, a.H, oy0, oy1);
  out_range(a.rw, ix, a.W, ox0, ox1);
  for (int oy = oy0; oy <= oy1; ++oy) {
    const Src sy = source(a.rh, oy, a.h);
    const float wy = (sy.i0 == iy ? sy.l0 : 0.f) + (sy.i1 == iy ? sy.l1 : 0.f);
    if (wy == 0.f) continue;
    for (int ox = ox0; ox <= ox1; ++ox) {
      const Src sx = source(a.rw, ox, a.w);
      const float wx = (sx.i0 == ix ? sx.l0 : 0.f) + (sx.i1 == ix ? sx.l1 : 0.f);
      if (wx == 0.f) continue;
      const int64_t o = ((int64_t)n * a.H + oy) * a.W + ox;
      const int64_t lab = a.labels[o];
      if (lab == a.ignore_index) continue;
      float v[CP];
      interpolate<CP>(a, n, sy, sx, v);
      const float lse = a.lse_in[o], wgt = wy * wx;
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const float p = __expf(v[c] - lse) - (c == (int)lab ? 1.f : 0.f);
        g[c] += wgt * p;
      }
    }
  }
  const float sc = a.scale[0];
  float* out = a.d_logits + (size_t)i * a.C;
#pragma unroll
  for (int c = 0; c < CP; ++c)
    if (c < a.C) out[c] = g[c] * sc;
}

// Tiled backward: a workgroup owns T x T low-resolution pixels.  Per chunk of CH channels,
//   phase A  every output pixel of the rectangle that feeds the tile gets its softmax - one-hot row computed
//            ONCE (from the tile's logits in LDS, 1-pixel halo) and parked in LDS,
//   phase B  every (low-resolution pixel, channel) gathers its <= (H/h + 2)^2 rows with the bilinear weights
//            (row / column weight tables in LDS), in ascending (oy, ox) order: deterministic.
// The un-tiled kernel above recomputes each output pixel's row for each of its four low-resolution
// neighbours from global memory (1.9 ms at 16 x 21 x 130^2 -> 513^2; this one: see tools/bench_upsample_ce.py).
constexpr int kUceT = 8, kUceCH = 8;
struct UceTile {
  int roy, rox;              // capacity of the output rectangle (rows, columns)
};
__global__ __launch_bounds__(256) void uce_bwd_tiled(UceArgs a, UceTile cap) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  constexpr int T = kUceT, CH = kUceCH, TH = T + 2;
  float* s_log = reinterpret_cast<float*>(sm);                    // [TH][TH][C]
  float* s_wy = s_log + TH * TH * a.C;                            // [T][roy]
  float* s_wx = s_wy + T * cap.roy;                               // [T][rox]
  int* s_rng = reinterpret_cast<int*>(s_wx + T * cap.rox);        // [2][T][2] first / last non-zero index
  float* s_g = reinterpret_cast<float*>(s_rng + 4 * T);           // [roy * rox][CH]
  const int n = blockIdx.z, ty0 = blockIdx.y * T, tx0 = blockIdx.x * T;
  const int tyl = min(ty0 + T - 1, a.h - 1), txl = min(tx0 + T - 1, a.w - 1);
  int oy_lo, oy_hi, ox_lo, ox_hi, dummy;
  out_range(a.rh, ty0, a.H, oy_lo, dummy);
  out_range(a.rh, tyl, a.H, dummy, oy_hi);
  out_range(a.rw, tx0, a.W, ox_lo, dummy);
  out_range(a.rw, txl, a.W, dummy, ox_hi);
  const int roy = oy_hi - oy_lo + 1, rox = ox_hi - ox_lo + 1;      // <= cap (host-checked bound)
  const int tid = threadIdx.x;

  // the tile's logits with a 1-pixel halo (rows / columns outside the image are never referenced)
  for (int e = tid; e < TH * TH * a.C; e += 256) {
    const int c = e % a.C, px = (e / a.C) % TH, py = e / (a.C * TH);
    const int iy = ty0 - 1 + py, ix = tx0 - 1 + px;
    float v = 0.f;
    if (iy >= 0 && iy < a.h && ix >= 0 && ix < a.w) v = a.logits[(((size_t)n * a.h + iy) * a.w + ix) * a.C + c];
    s_log[e] = v;
  }
  // weight tables: s_wy[r][j] = coefficient of low-resolution row ty0 + r in output row oy_lo + j
  for (int e = tid; e < T * roy; e += 256) {
    const int r = e / roy, j = e - r * roy;
    const Src sy = source(a.rh, oy_lo + j, a.h);
    s_wy[r * cap.roy + j] = (sy.i0 == ty0 + r ? sy.l0 : 0.f) + (sy.i1 == ty0 + r ? sy.l1 : 0.f);
  }
  for (int e = tid; e < T * rox; e += 256) {
    const int r = e / rox, j = e - r * rox;
    const Src sx = source(a.rw, ox_lo + j, a.w);
    s_wx[r * cap.rox + j] = (sx.i0 == tx0 + r ? sx.l0 : 0.f) + (sx.i1 == tx0 + r ? sx.l1 : 0.f);
  }
  __syncthreads();
  if (tid < 2 * T) {                                  // non-zero ranges of the 2 * T table rows
    const int which = tid / T, r = tid % T;
    const float* tab = which ? s_wx + r * cap.rox : s_wy + r * cap.roy;
    const int len = which ? rox : roy;
    int lo = len, hi = -1;
    for (int j = 0; j < len; ++j)
      if (tab[j] != 0.f) { lo = min(lo, j); hi = j; }
    s_rng[(which * T + r) * 2] = lo;
    s_rng[(which * T + r) * 2 + 1] = hi;
  }
  const float sc = a.scale[0];
  for (int c0 = 0; c0 < a.C; c0 += CH) {
    __syncthreads();                                  // tables ready / previous chunk's rows consumed
    // phase A
    for (int o = tid; o < roy * rox; o += 256) {
      const int j = o / rox, i = o - j * rox;
      const int oy = oy_lo + j, ox = ox_lo + i;
      const Src sy = source(a.rh, oy, a.h), sx = source(a.rw, ox, a.w);
      const int64_t oo = ((int64_t)n * a.H + oy) * a.W + ox;
      const int64_t lab = a.labels[oo];
      const bool live = lab != a.ignore_index && sy.i0 >= ty0 - 1 && sy.i1 <= tyl + 1 && sx.i0 >= tx0 - 1 &&
                        sx.i1 <= txl + 1;
      float g[CH];
#pragma unroll
      for (int k = 0; k < CH; ++k) g[k] = 0.f;
      if (live) {
        const float lse = a.lse_in[oo];
        const float* p00 = s_log + ((sy.i0 - ty0 + 1) * TH + (sx.i0 - tx0 + 1)) * a.C + c0;
        const float* p01 = s_log + ((sy.i0 - ty0 + 1) * TH + (sx.i1 - tx0 + 1)) * a.C + c0;
        const float* p10 = s_log + ((sy.i1 - ty0 + 1) * TH + (sx.i0 - tx0 + 1)) * a.C + c0;
        const float* p11 = s_log + ((sy.i1 - ty0 + 1) * TH + (sx.i1 - tx0 + 1)) * a.C + c0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          if (c0 + k < a.C) {
            const float v = sy.l0 * (sx.l0 * p00[k] + sx.l1 * p01[k]) + sy.l1 * (sx.l0 * p10[k] + sx.l1 * p11[k]);
            g[k] = __expf(v - lse) - (c0 + k == (int)lab ? 1.f : 0.f);
          }
        }
      }
      float4v* dst = reinterpret_cast<float4v*>(s_g + (size_t)o * CH);
      dst[0] = float4v{g[0], g[1], g[2], g[3]};
      dst[1] = float4v{g[4], g[5], g[6], g[7]};
    }
    __syncthreads();
    // phase B
    for (int e = tid; e < T * T * CH; e += 256) {
      const int k = e % CH, cx = (e / CH) % T, r = e / (CH * T);
      if (ty0 + r >= a.h || tx0 + cx >= a.w || c0 + k >= a.C) continue;
      const int jlo = s_rng[r * 2], jhi = s_rng[r * 2 + 1], ilo = s_rng[(T + cx) * 2], ihi = s_rng[(T + cx) * 2 + 1];
      float acc = 0.f;
      for (int j = jlo; j <= jhi; ++j) {
        const float wy = s_wy[r * cap.roy + j];
        const float* row = s_g + ((size_t)j * rox) * CH + k;
        for (int i = ilo; i <= ihi; ++i) acc += (wy * s_wx[cx * cap.rox + i]) * row[(size_t)i * CH];
      }
      a.d_logits[(((size_t)n * a.h + ty0 + r) * a.w + tx0 + cx) * a.C + c0 + k] = acc * sc;
    }
  }
}

// output rows / columns that can touch T consecutive low-resolution ones (bound used for the LDS layout)
inline int uce_rect(int in, int out) {
  const float scale = (float)in / (float)out, inv = 1.0f / scale;
  int worst = 1;
  for (int t0 = 0; t0 < in; t0 += kUceT) {            // the device's out_range of every tile (+1: rounding slack)
    const int tl = std::min(t0 + kUceT - 1, in - 1);
    const int lo = std::max(0, (int)floorf(((float)t0 - 1.f + 0.5f) * inv - 0.5f) - 1);
    const int hi = std::min(out - 1, (int)ceilf(((float)tl + 1.f + 0.5f) * inv - 0.5f) + 1);
    worst = std::max(worst, hi - lo + 2);
  }
  return worst;
}

inline int64_t uce_blocks(int N, int H, int W) { return ((int64_t)N * H * W + 255) / 256; }

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" int spml_upsample_ce_supported(int C) { return C >= 1 && C <= 64; }

extern "C" size_t spml_upsample_ce_workspace_bytes(int N, int H, int W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  return (size_t)uce_blocks(N, H, W) * 2 * sizeof(float);
}

extern "C" int spml_upsample_ce_fwd_f32(const float* logits, const int64_t* labels, int N, int C, int h, int w,
                                        int H, int W, int64_t ignore_index, float* lse, float* result,
                                        void* ws, size_t ws_bytes, void* stream) {
  if (!logits || !labels || !lse || !result || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return SPML_ERR_INVALID_ARG;
  if (!spml_upsample_ce_supported(C)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_upsample_ce_workspace_bytes(N, H, W)) return SPML_ERR_WORKSPACE;
  UceArgs a{};
  a.logits = logits; a.labels = labels; a.lse = lse; a.partial = static_cast<float*>(ws);
  a.N = N; a.C = C; a.h = h; a.w = w; a.H = H; a.W = W; a.ignore_index = ignore_index;
  a.rh = (float)h / (float)H; a.rw = (float)w / (float)W;
  hipStream_t s = (hipStream_t)stream;
  const int64_t blocks = uce_blocks(N, H, W);
  if (C <= 24) hipLaunchKernelGGL(uce_fwd<24>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else if (C <= 32) hipLaunchKernelGGL(uce_fwd<32>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(uce_fwd<64>, dim3((unsigned)blocks), dim3(256), 0, s, a);
  hipLaunchKernelGGL(uce_reduce, dim3(1), dim3(1024), 0, s, a.partial, blocks, result);
  return launch_status();
}

extern "C" int spml_upsample_ce_bwd_f32(const float* logits, const int64_t* labels, const float* lse, int N,
                                        int C, int h, int w, int H, int W, int64_t ignore_index,
                                        const float* scale, float* d_logits, void* stream) {
  if (!logits || !labels || !lse || !scale || !d_logits || N <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return SPML_ERR_INVALID_ARG;
  if (!spml_upsample_ce_supported(C)) return SPML_ERR_UNSUPPORTED;
  UceArgs a{};
  a.logits = logits; a.labels = labels; a.lse_in = lse; a.scale = scale; a.d_logits = d_logits;
  a.N = N; a.C = C; a.h = h; a.w = w; a.H = H; a.W = W; a.ignore_index = ignore_index;
  a.rh = (float)h / (float)H; a.rw = (float)w / (float)W;
  hipStream_t s = (hipStream_t)stream;
  {
    UceTile cap{uce_rect(h, H), uce_rect(w, W)};
    const size_t lds = ((size_t)(kUceT + 2) * (kUceT + 2) * C + (size_t)kUceT * (cap.roy + cap.rox) + 4 * kUceT +
                        (size_t)cap.roy * cap.rox * kUceCH) * 4 + 16;
    if (lds <= 72 * 1024 && N <= 65535) {             // two workgroups per CU; else: the un-tiled kernel
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(uce_bwd_tiled), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
      hipLaunchKernelGGL(uce_bwd_tiled, dim3((w + kUceT - 1) / kUceT, (h + kUceT - 1) / kUceT, N), dim3(256), lds, s,
                         a, cap);
      return launch_status();
    }
  }
  const unsigned blocks = (unsigned)(((int64_t)N * h * w + 255) / 256);
  if (C <= 24) hipLaunchKernelGGL(uce_bwd<24>, dim3(blocks), dim3(256), 0, s, a);
  else if (C <= 32) hipLaunchKernelGGL(uce_bwd<32>, dim3(blocks), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(uce_bwd<64>, dim3(blocks), dim3(256), 0, s, a);
  return launch_status();
}
