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
// Fast path (K <= 64, D even and <= 320) -- kernel kmeans_pass<NT,KS,KSPLIT>:
//   * persistent 256-thread workgroups; each owns a contiguous range of 32*PT
//     pixel tiles of ONE image, so prototypes are loaded and partial sums are
//     flushed once per workgroup, not once per tile;
//   * X tile: flat, 16-B-vector coalesced HBM -> VGPR -> LDS copy of the
//     contiguous [rows*D] block; the next tile's loads are issued before the
//     current tile is computed (register double buffering);
//   * similarity on the f16 matrix cores at fp32 accuracy (common.cuh,
//     split-f16 x2: 3 MFMA 32x32x16 per 16 channels).  Prototypes are the A
//     operand and live in registers for the whole kernel; pixels are the B
//     operand, read from LDS (ds_read_b64, conflict-free for D = 2*odd) and
//     split on the fly.  With prototypes as rows of the accumulator tile the
//     arg-max over prototypes is in-register per lane plus ONE cross-half
//     exchange (the "swapped operand" trick);
//   * KSPLIT waves share one 32-pixel tile and split the channel range (so that
//     a 258-channel row tile fits LDS); their partial dot products meet in LDS;
//   * M-step: every thread owns two adjacent channels for the whole workgroup
//     lifetime and walks the tile's pixels in order, keeping the running sum of
//     the current label run in registers (labels are spatially coherent) and
//     folding it into the [K][D] LDS accumulator only when the label changes.
//     No atomics: the summation order is fixed -> bit-reproducible results.
//   * per-workgroup partial sums go to a slab; a tiny finalize kernel adds the
//     slabs in fixed order, normalises (empty cluster -> zero prototype, as the
//     reference) and emits the split-f16 prototypes for the next pass.
// Generic path (any K, D): VALU dot products + run-length atomics; correct, not
// tuned (used for K > 64, e.g. the 1024-centroid stress configuration).
#include "common.cuh"

namespace spml {

int segment_sum_launch(const float* x, const int64_t* ids, int64_t P, int D, int64_t M,
                       float* sums, hipStream_t s);

namespace {

thread_local const char* g_last_path = "none";

constexpr int kKsMax = 5;

struct PassArgs {
  const float* x;
  int64_t x_bytes;            // total bytes of x (bounds for the vector copy)
  int D, K, n_img, G;
  const int64_t* seg_off;     // device [n_img+1]
  const _Float16* cent_h;     // [n_img][kpad][dpad]
  const _Float16* cent_l;
  int kpad, dpad;
  int32_t* labels;            // [P] in (accumulate-only) / out (assign)
  float* slabs;               // [n_img][G][K][D]
  int do_assign, do_accum;
};

template <int NT, int KS, int KSPLIT>
struct PassCfg {
  static constexpr int PT = 4 / KSPLIT;         // 32-pixel tiles per workgroup step
  static constexpr int TPW = 32 * PT;           // pixels per workgroup step
  static constexpr int MAXV = 2 * KS + 1;       // 16-B vectors per thread per tile
  static constexpr int RQ = 16 / KSPLIT;        // accumulator regs reduced per wave
};

__host__ __device__ inline size_t pass_lds_bytes(int D, int K, int NT, int KSPLIT) {
  const int PT = 4 / KSPLIT, TPW = 32 * PT;
  size_t b = (size_t)TPW * D * 4 + 32;                    // X tile (+ alignment shift)
  b = (b + 15) / 16 * 16;
  b += (size_t)K * D * 4;                                 // accumulators
  b = (b + 15) / 16 * 16;
  if (KSPLIT > 1) b += (size_t)4 * NT * 16 * 64 * 4;      // partial-dot exchange
  b += 4 * 64 * 8;                                        // candidates (val, idx)
  b += (size_t)TPW * 4;                                   // labels of the tile
  return b + 64;
}

template <int NT, int KS, int KSPLIT>
__global__ __launch_bounds__(256) void kmeans_pass(PassArgs a) {
  using Cfg = PassCfg<NT, KS, KSPLIT>;
  constexpr int PT = Cfg::PT, TPW = Cfg::TPW, MAXV = Cfg::MAXV, RQ = Cfg::RQ;
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

  // ---- LDS carve-up ----
  size_t off = 0;
  unsigned char* xs = lds;                                    // raw tile bytes
  off = ((size_t)TPW * D * 4 + 32 + 15) / 16 * 16;
  float* acc = reinterpret_cast<float*>(lds + off);           // [K][D]
  off += ((size_t)K * D * 4 + 15) / 16 * 16;
  float* xchg = reinterpret_cast<float*>(lds + off);          // [PT][KSPLIT][KSPLIT][NT][RQ][64]
  if (KSPLIT > 1) off += (size_t)4 * NT * 16 * 64 * 4;
  float* cand_v = reinterpret_cast<float*>(lds + off);        // [4][64]
  int* cand_i = reinterpret_cast<int*>(lds + off + 4 * 64 * 4);
  off += 4 * 64 * 8;
  int* lab = reinterpret_cast<int*>(lds + off);               // [TPW]

  const int64_t seg0 = a.seg_off[img];
  const int64_t len = a.seg_off[img + 1] - seg0;
  const int64_t T = (len + TPW - 1) / TPW;
  const int64_t t_begin = (T * g) / a.G, t_end = (T * (g + 1)) / a.G;

  // ---- prototypes -> registers (A operand), once per workgroup ----
  half8 ah[KS][NT], al[KS][NT];
  if (a.do_assign) {
#pragma unroll
    for (int i = 0; i < KS; ++i) {
      const int k0 = 16 * (ksub + KSPLIT * i) + 8 * half;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const size_t o = ((size_t)img * a.kpad + 32 * t + j) * a.dpad + k0;
        if (k0 < a.dpad) {
          ah[i][t] = *reinterpret_cast<const half8*>(a.cent_h + o);
          al[i][t] = *reinterpret_cast<const half8*>(a.cent_l + o);
        } else {
          ah[i][t] = half8{0, 0, 0, 0, 0, 0, 0, 0};
          al[i][t] = half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
      }
    }
  }
  if (a.do_accum)
    for (int i = tid; i < K * D; i += 256) acc[i] = 0.f;

  // column pair owned by this thread in the M-step
  const int cp = tid;
  const bool own_cols = a.do_accum && (2 * cp < D);
  float run0 = 0.f, run1 = 0.f;
  int cur = -1;

  // ---- tile copy helpers ----
  float4v pre[MAXV];
  int64_t a0 = 0;       // 16-B aligned global byte offset of the prefetched tile
  int nvec = 0;
  auto tile_issue = [&](int64_t t) {
    const int64_t r0 = seg0 + t * TPW;
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int64_t b0 = r0 * D * 4, b1 = b0 + (int64_t)nrows * D * 4;
    a0 = b0 & ~(int64_t)15;
    nvec = (int)((b1 - a0 + 15) >> 4);
    const unsigned char* base = reinterpret_cast<const unsigned char*>(a.x);
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      float4v val = {0.f, 0.f, 0.f, 0.f};
      if (v < nvec) {
        const int64_t o = a0 + 16 * (int64_t)v;
        if (o + 16 <= a.x_bytes) {
          val = *reinterpret_cast<const float4v*>(base + o);
        } else if (o + 8 <= a.x_bytes) {
          const float2 h2 = *reinterpret_cast<const float2*>(base + o);
          val[0] = h2.x; val[1] = h2.y;
        }
      }
      pre[i] = val;
    }
  };

  if (t_begin < t_end) tile_issue(t_begin);

  for (int64_t t = t_begin; t < t_end; ++t) {
    const int nrows = (int)min((int64_t)TPW, len - t * TPW);
    const int shift = (int)((seg0 + t * TPW) * D * 4 - a0);   // 0 or 8
    const int cur_nvec = nvec;
    __syncthreads();                       // previous tile fully consumed
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int v = i * 256 + tid;
      if (v < cur_nvec) *reinterpret_cast<float4v*>(xs + 16 * (size_t)v) = pre[i];
    }
    if (!a.do_assign && tid < TPW)
      lab[tid] = (tid < nrows) ? a.labels[seg0 + t * TPW + tid] : -1;
    __syncthreads();
    if (t + 1 < t_end) tile_issue(t + 1);  // in flight while this tile is computed

    const unsigned char* xrow = xs + shift;

    if (a.do_assign) {
      // ================= E-step: MFMA similarity + arg-max =================
      float16v acc_h[NT], acc_x[NT];
#pragma unroll
      for (int q = 0; q < NT; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc_h[q][r] = 0.f; acc_x[q][r] = 0.f; }

#pragma unroll
      for (int i = 0; i < KS; ++i) {
        const int kb = 16 * (ksub + KSPLIT * i);
        if (kb < D) {                       // wave-uniform
          const int k0 = kb + 8 * half;
          float v[8];
          const float2* src =
              reinterpret_cast<const float2*>(xrow + ((size_t)(pt * 32 + j) * D + k0) * 4);
          if (k0 + 8 <= D) {
#pragma unroll
            for (int e = 0; e < 4; ++e) { const float2 f = src[e]; v[2 * e] = f.x; v[2 * e + 1] = f.y; }
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float2 f = {0.f, 0.f};
              if (k0 + 2 * e < D) f = src[e];   // D even -> pairs are all-in or all-out
              v[2 * e] = f.x; v[2 * e + 1] = f.y;
            }
          }
          half8 bh, bl;
          split8(v, bh, bl);
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
        // wave `d` reduces accumulator registers [d*RQ, (d+1)*RQ)
#pragma unroll
        for (int d = 0; d < KSPLIT; ++d)
#pragma unroll
          for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < RQ; ++r) {
              const float s = acc_h[q][d * RQ + r] + acc_x[q][d * RQ + r] * kSplitInv;
              xchg[(((((size_t)pt * KSPLIT + d) * KSPLIT + ksub) * NT + q) * RQ + r) * 64 + lane] = s;
            }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
          for (int r = 0; r < RQ; ++r) {
            float s = 0.f;
#pragma unroll
            for (int src = 0; src < KSPLIT; ++src)
              s += xchg[(((((size_t)pt * KSPLIT + ksub) * KSPLIT + src) * NT + q) * RQ + r) * 64 + lane];
            const int rr = ksub * RQ + r;
            const int c = 32 * q + (rr & 3) + 8 * (rr >> 2) + 4 * half;
            if (c < K && s > best) { best = s; best_i = c; }
          }
      }
      // the two lane halves hold different prototype rows of the same pixel
      {
        const float ob = __shfl_xor(best, 32, 64);
        const int oi = __shfl_xor(best_i, 32, 64);
        if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
      }
      if (KSPLIT > 1) {
        cand_v[wave * 64 + lane] = best;
        cand_i[wave * 64 + lane] = best_i;
        __syncthreads();
        if (ksub == 0 && lane < 32) {
#pragma unroll
          for (int s2 = 1; s2 < KSPLIT; ++s2) {
            const float ob = cand_v[(wave + s2) * 64 + lane];
            const int oi = cand_i[(wave + s2) * 64 + lane];
            if (ob > best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
          }
        }
      }
      if (ksub == 0 && lane < 32) {
        const int p = pt * 32 + lane;
        lab[p] = best_i;
        if (p < nrows) a.labels[seg0 + t * TPW + p] = best_i;
      }
      __syncthreads();
    }

    if (a.do_accum) {
      // ================= M-step: ordered run-length accumulation ============
      if (wave * 64 * 2 < D) {               // wave-uniform: any owned column here?
        const float2* xc = reinterpret_cast<const float2*>(xrow) + cp;
        for (int b = 0; b < nrows; b += 64) {
          const int nb = min(64, nrows - b);
          const int mylab = lab[min(b + lane, TPW - 1)];

          for (int i = 0; i < nb; ++i) {
            const int l = __builtin_amdgcn_readlane(mylab, i);
            if (l != cur) {                  // wave-uniform
              if (own_cols && (unsigned)cur < (unsigned)K) {
                float2* dst = reinterpret_cast<float2*>(acc + (size_t)cur * D) + cp;
                float2 o = *dst;
                o.x += run0; o.y += run1;
                *dst = o;
              }
              run0 = 0.f; run1 = 0.f; cur = l;
            }
            if (own_cols) {
              const float2 f = xc[(size_t)(b + i) * (D / 2)];
              run0 += f.x; run1 += f.y;
            }
          }
        }
      }
    }
  }

  if (a.do_accum) {
    if (own_cols && (unsigned)cur < (unsigned)K) {
      float2* dst = reinterpret_cast<float2*>(acc + (size_t)cur * D) + cp;
      float2 o = *dst;
      o.x += run0; o.y += run1;
      *dst = o;
    }
    __syncthreads();
    float* slab = a.slabs + ((size_t)img * a.G + g) * K * D;
    for (int i = tid; i < K * D; i += 256) slab[i] = acc[i];
  }
}

// slabs -> prototypes: sum the G partial slabs in fixed order, L2-normalise
// (zero sum -> zero prototype: 0 / 1e-12), write fp32 + split-f16 forms.
__global__ __launch_bounds__(256) void kmeans_finalize(const float* __restrict__ slabs, int G,
                                                       int K, int D, int kpad, int dpad,
                                                       int normalize,
                                                       float* __restrict__ cent,
                                                       _Float16* __restrict__ cent_h,
                                                       _Float16* __restrict__ cent_l) {
  __shared__ float red[4];
  const int k = blockIdx.x, img = blockIdx.y;
  const int tid = threadIdx.x;
  constexpr int MAXC = 8;                      // D <= 2048
  float s[MAXC];
  float ss = 0.f;
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int d = tid + 256 * c;
    float v = 0.f;
    if (d < D) {
      const float* p = slabs + ((size_t)img * G * K + k) * D + d;
      for (int gI = 0; gI < G; ++gI) v += p[(size_t)gI * K * D];
    }
    s[c] = v;
    ss += v * v;
  }
  float dn = 1.f;
  if (normalize) {
    const float n = sqrtf(block_sum_256(ss, red));
    dn = n >= kEps ? n : kEps;
  }
#pragma unroll
  for (int c = 0; c < MAXC; ++c) {
    const int d = tid + 256 * c;
    if (d < D) {
      const float v = s[c] / dn;
      if (cent) cent[((size_t)img * K + k) * D + d] = v;
      if (cent_h) {
        _Float16 h, l;
        split_f16(v, h, l);
        const size_t o = ((size_t)img * kpad + k) * dpad + d;
        cent_h[o] = h;
        cent_l[o] = l;
      }
    }
  }
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
  int NT, KS, KSPLIT, G, kpad, dpad;
  size_t lds;
};

Plan make_plan(const float* x, int64_t P, int D, int K, int n_img, int64_t max_seg_len,
               int flags) {
  Plan pl{};
  pl.fast = false;
  if (flags & SPML_KMEANS_FORCE_GENERIC) return pl;
  if (K < 1 || K > 64 || (D & 1) || D < 2) return pl;
  if (reinterpret_cast<uintptr_t>(x) & 15) return pl;
  const int steps = (D + 15) / 16;
  int ksplit = 0, ks = 0;
  for (int s : {1, 2, 4}) {
    const int need = (steps + s - 1) / s;
    if (need <= kKsMax) { ksplit = s; ks = need; break; }
  }
  if (!ksplit) return pl;
  ks = ks <= 2 ? 2 : (ks <= 3 ? 3 : 5);
  const int nt = K <= 32 ? 1 : 2;
  const size_t lds = pass_lds_bytes(D, K, nt, ksplit);
  if (lds > 160 * 1024) return pl;
  pl.fast = true;
  pl.NT = nt; pl.KS = ks; pl.KSPLIT = ksplit; pl.lds = lds;
  pl.kpad = 32 * nt;
  pl.dpad = 16 * ks * ksplit;
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
  size_t lab32, cent_h, cent_l, cent_f, slabs, ids, total;
};

WsLayout ws_layout(int64_t P, int D, int K, int n_img, int64_t max_seg_len) {
  // sized for the worst of the fast / generic plans
  WsLayout w{};
  size_t o = 0;
  w.lab32 = o; o = align_up(o + (size_t)P * 4, 256);
  const size_t kpad = 64, dpad = 320;
  w.cent_h = o; o = align_up(o + (size_t)n_img * kpad * dpad * 2, 256);
  w.cent_l = o; o = align_up(o + (size_t)n_img * kpad * dpad * 2, 256);
  w.cent_f = o; o = align_up(o + (size_t)n_img * K * D * 4, 256);
  // slabs: G <= ceil(512 / n_img) per image (fast), or 1 (generic sums)
  const size_t gmax = (size_t)((512 + n_img - 1) / n_img);
  w.slabs = o; o = align_up(o + (size_t)n_img * gmax * K * D * 4, 256);
  w.ids = o; o = align_up(o + (size_t)P * 8, 256);
  w.total = o;
  (void)max_seg_len;
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

int launch_pass(const PassArgs& a, const Plan& pl, hipStream_t s) {
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

extern "C" const char* spml_kmeans_last_path(void) { return g_last_path; }

extern "C" size_t spml_kmeans_workspace_bytes(int64_t P, int D, int K, int n_img,
                                              int64_t max_seg_len) {
  if (P < 0 || D <= 0 || K <= 0 || n_img <= 0) return 0;
  return ws_layout(P, D, K, n_img, max_seg_len).total;
}

static int kmeans_common(const float* x, int64_t P, int D, const int64_t* seg_off, int n_img,
                         int64_t max_seg_len, int K, const int64_t* labels_init,
                         const float* given_centroids, int iterations, int64_t* labels_out,
                         float* centroids_out, int flags, void* ws, size_t ws_bytes,
                         hipStream_t s) {
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

  const Plan pl = make_plan(x, P, D, K, n_img, max_seg_len, flags);
  const unsigned pblocks = (unsigned)((P + 255) / 256);
  int rc = SPML_OK;

  if (labels_init)
    hipLaunchKernelGGL(labels_i64_to_i32, dim3(pblocks), dim3(256), 0, s, labels_init, lab32, P);

  if (pl.fast) {
    g_last_path = "mfma_f16x2";
    if (hipMemsetAsync(cent_h, 0, (size_t)n_img * pl.kpad * pl.dpad * 2, s) != hipSuccess ||
        hipMemsetAsync(cent_l, 0, (size_t)n_img * pl.kpad * pl.dpad * 2, s) != hipSuccess)
      return SPML_ERR_LAUNCH;
    PassArgs a{};
    a.x = x; a.x_bytes = P * (int64_t)D * 4; a.D = D; a.K = K; a.n_img = n_img; a.G = pl.G;
    a.seg_off = seg_off; a.cent_h = cent_h; a.cent_l = cent_l; a.kpad = pl.kpad;
    a.dpad = pl.dpad; a.labels = lab32; a.slabs = slabs;
    auto finalize = [&](int normalize, const float* src, int G) {
      hipLaunchKernelGGL(kmeans_finalize, dim3(K, n_img), dim3(256), 0, s, src, G, K, D, pl.kpad,
                         pl.dpad, normalize, normalize ? cent_f : (float*)nullptr, cent_h,
                         cent_l);
    };
    if (given_centroids) {
      finalize(0, given_centroids, 1);          // split only
      a.do_assign = 1; a.do_accum = 0;
      rc = launch_pass(a, pl, s);
      if (rc != SPML_OK) return rc;
    } else {
      if (iterations > 0) {
        a.do_assign = 0; a.do_accum = 1;        // M-step on the initial labels
        rc = launch_pass(a, pl, s);
        if (rc != SPML_OK) return rc;
        finalize(1, slabs, pl.G);
      }
      for (int it = 0; it < iterations; ++it) {
        const bool last = (it == iterations - 1);
        a.do_assign = 1; a.do_accum = last ? 0 : 1;
        rc = launch_pass(a, pl, s);
        if (rc != SPML_OK) return rc;
        if (!last) finalize(1, slabs, pl.G);
      }
    }
  } else {
    g_last_path = "generic";
    const int64_t M = (int64_t)n_img * K;
    auto mstep = [&]() -> int {
      hipLaunchKernelGGL(generic_ids, dim3(pblocks), dim3(256), 0, s, lab32, seg_off, n_img, K, P,
                         ids);
      if (hipMemsetAsync(slabs, 0, (size_t)M * D * 4, s) != hipSuccess) return SPML_ERR_LAUNCH;
      int r = segment_sum_launch(x, ids, P, D, M, slabs, s);
      if (r != SPML_OK) return r;
      return spml_normalize_rows_f32(slabs, M, D, cent_f, s);
    };
    if (given_centroids) {
      rc = generic_assign_launch(x, P, D, seg_off, n_img, K, given_centroids, lab32, s);
      if (rc != SPML_OK) return rc;
    } else {
      for (int it = 0; it < iterations; ++it) {
        rc = mstep();
        if (rc != SPML_OK) return rc;
        rc = generic_assign_launch(x, P, D, seg_off, n_img, K, cent_f, lab32, s);
        if (rc != SPML_OK) return rc;
      }
    }
  }
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
  return kmeans_common(x, P, D, seg_offsets, n_img, max_seg_len, K, labels_init, nullptr,
                       iterations, labels_out, centroids_out, flags, ws, ws_bytes,
                       (hipStream_t)stream);
}

extern "C" int spml_kmeans_assign_f32(const float* x, int64_t P, int D,
                                      const int64_t* seg_offsets, int n_img,
                                      int64_t max_seg_len, int K, const float* centroids,
                                      int64_t* labels_out, int flags, void* ws, size_t ws_bytes,
                                      void* stream) {
  if (!centroids) return SPML_ERR_INVALID_ARG;
  return kmeans_common(x, P, D, seg_offsets, n_img, max_seg_len, K, nullptr, centroids, 1,
                       labels_out, nullptr, flags, ws, ws_bytes, (hipStream_t)stream);
}
