// Training-mode batch normalisation fused with the ReLU and the residual add around it, for
// channels-last (NHWC) fp32 activations [R = N*H*W, C].
//
// The reference's ResNet bottleneck (spml/models/backbones/resnet.py:42-63) runs
//   bn(conv(x)) -> relu        (twice)      and      bn(conv(x)) + identity -> relu
// as separate framework ops: per unit 3 batch norms (two passes over the tensor each), 3 ReLUs
// and 1 add -- 20 % of the GPU time of the training step once the convolutions are tuned
// (profiles/r02_train_step_steady_state.md).  Here one statistics pass + one apply pass do
// normalise + scale/shift (+ residual) (+ ReLU), and the backward is one reduction pass + one
// apply pass that also yields the gradient of the residual branch.  HBM-bound, float4 accesses.
//
//   spml_bn_stats_f32          per-channel mean and sum of squared deviations (chunked, merged
//                              with Chan's formula: no E[x^2] - E[x]^2 cancellation)
//   spml_bn_act_apply_f32      y = act((x - mean) * invstd * gamma + beta [+ residual])
//   spml_bn_act_bwd_reduce_f32 sum(dz), sum(dz * xhat)   with dz = dy * (y > 0)
//   spml_bn_act_bwd_apply_f32  dx = gamma * invstd * (dz - sum_dz/n - xhat * sum_dz_xhat/n); dres = dz
// Cross-rank statistics (SyncBatchNorm) are combined by the caller between the two halves.
#include "common.hpp"

namespace spml {
namespace {


// hl8 = split-f16 copy of a tensor (csrc/conv.hip): 16-byte units ((row*C/8 + c/8)*2 + part) of
// 8 channels; a channel quad q is half a unit.
// ReLU mask: one byte per (row, channel quad), bit e = (y[4q+e] > 0) -- 0.25 B per element for the
// backward passes instead of re-reading y.
__device__ __forceinline__ float4v relu_mask_bits(float4v dz, const unsigned char* __restrict__ mask, size_t row,
                                                  int C, int q) {
  const unsigned m = mask[row * (size_t)(C >> 2) + q];
#pragma unroll
  for (int e = 0; e < 4; ++e) dz[e] = (m >> e) & 1u ? dz[e] : 0.f;
  return dz;
}

// S = 2^(14 - e) for the smallest e with bound < 2^e (same rule as csrc/conv.hip)
__device__ __forceinline__ float bn_pow2_scale(float bound) {
  if (!(bound > 0.f) || bound > 1e38f) return 1.f;
  int e;
  (void)frexpf(bound, &e);
  e = 14 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);
}

// One channel quad (4 values) of row `row` into the hl8 tensor.  A 16-byte unit holds 8 channels of one part: the two
// lanes of a quad pair (q, q ^ 1: adjacent lanes, same row) exchange halves so that the even lane stores the whole h
// unit and the odd lane the whole l unit -- one contiguous kilobyte per wave store instead of two instructions that
// each fill every other 8 bytes (bn_bwd_apply ran at 3 TB/s with those).  Every lane of the pair must call it.
__device__ __forceinline__ void store_hl8_quad(uint2* __restrict__ out, size_t row, int C, int q, float4v v,
                                               float s) {
  union { uint2 u; _Float16 h[4]; } hh, ll;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float x = v[e] * s;
    const _Float16 h = (_Float16)x;
    hh.h[e] = h;
    ll.h[e] = (_Float16)(x - (float)h);
  }
  const bool odd = (q & 1) != 0;
  const uint2 send = odd ? hh.u : ll.u;            // the even lane needs the partner's h half, the odd lane its l half
  uint2 recv;
  recv.x = (unsigned)__shfl_xor((int)send.x, 1, kWave);
  recv.y = (unsigned)__shfl_xor((int)send.y, 1, kWave);
  uint4 unit;
  if (odd) { unit.x = recv.x; unit.y = recv.y; unit.z = ll.u.x; unit.w = ll.u.y; }
  else { unit.x = hh.u.x; unit.y = hh.u.y; unit.z = recv.x; unit.w = recv.y; }
  reinterpret_cast<uint4*>(out)[(row * (size_t)(C >> 3) + (q >> 1)) * 2 + (odd ? 1 : 0)] = unit;
}

// block = 256 threads = CQ channel quads x (256 / CQ) row lanes; grid (C / (4*CQ), chunks)
// MODE 0: (sum, sumsq) of x -> (mean, M2) of the chunk;  MODE 1: (sum dz, sum dz * (x - mean))
template <int MODE, int MASK>
__global__ __launch_bounds__(256) void bn_partial(const float* __restrict__ x, const float* __restrict__ dy,
                                                  const float* __restrict__ y, const float* __restrict__ mean,
                                                  int64_t R, int C, int cq, int chunk_rows,
                                                  float* __restrict__ pa, float* __restrict__ pb,
                                                  const unsigned char* __restrict__ yh, float* __restrict__ pc,
                                                  float* __restrict__ pd, float* __restrict__ zero_me) {
  // zero_me: the tensor-bound slot the following bn_merge max-reduces into (single-rank calls)
  if (zero_me && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_me = 0.f;
  // pc / pd (optional): MODE 0 per-channel max / min of x, MODE 1 max |dz| (the bounds that fix
  // the power-of-two scale of the split-f16 copies, csrc/conv.hip); yh (MASK 2): the ReLU mask
  // bytes written by bn_apply instead of an fp32 y
  __shared__ float4v sa[256], sb[256];
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, nty = 256 / cq;
  const int q = blockIdx.x * cq + tx;                       // channel quad
  const int64_t r0 = (int64_t)blockIdx.y * chunk_rows;
  const int nrows = (int)min((int64_t)chunk_rows, R - r0);
  const int cquads = C >> 2;
  float4v a = {0.f, 0.f, 0.f, 0.f}, b = {0.f, 0.f, 0.f, 0.f};
  float4v mu = {0.f, 0.f, 0.f, 0.f};
  float4v hi = {-3.4e38f, -3.4e38f, -3.4e38f, -3.4e38f}, lo = {3.4e38f, 3.4e38f, 3.4e38f, 3.4e38f};
  if (MODE == 1) hi = a;
  if (q < cquads) {
    // MODE 0: sums are taken relative to the chunk's first row (a per-channel shift), so that
    // sumsq - n * mean^2 does not cancel when |mean| >> std;  MODE 1: the batch mean
    mu = MODE == 1 ? *reinterpret_cast<const float4v*>(mean + 4 * q)
                   : *reinterpret_cast<const float4v*>(x + (size_t)r0 * C + 4 * q);
#pragma unroll 4
    for (int r = ty; r < nrows; r += nty) {
      const size_t o = (size_t)(r0 + r) * C + 4 * q;
      const float4v xv = *reinterpret_cast<const float4v*>(x + o);
      if (MODE == 0) {
        const float4v d = xv - mu;
        a += d;
        b += d * d;
#pragma unroll
        for (int e = 0; e < 4; ++e) { hi[e] = fmaxf(hi[e], xv[e]); lo[e] = fminf(lo[e], xv[e]); }
      } else {
        float4v dz = *reinterpret_cast<const float4v*>(dy + o);
        if (MASK == 1) {
          const float4v yv = *reinterpret_cast<const float4v*>(y + o);
#pragma unroll
          for (int e = 0; e < 4; ++e) dz[e] = yv[e] > 0.f ? dz[e] : 0.f;
        } else if (MASK == 2) {
          dz = relu_mask_bits(dz, yh, (size_t)(r0 + r), C, q);
        }
        a += dz;
        b += dz * (xv - mu);
#pragma unroll
        for (int e = 0; e < 4; ++e) hi[e] = fmaxf(hi[e], fabsf(dz[e]));
      }
    }
  }
  sa[threadIdx.x] = a;
  sb[threadIdx.x] = b;
  __syncthreads();
  if (ty == 0 && q < cquads) {
    for (int i = 1; i < nty; ++i) { a += sa[i * cq + tx]; b += sb[i * cq + tx]; }
    if (MODE == 0) {
      const float inv = 1.0f / (float)nrows;
      const float4v m = a * inv;          // mean of the shifted values
      b = b - m * m * (float)nrows;       // chunk M2 (shift invariant)
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = fmaxf(b[e], 0.f);
      a = m + mu;                         // chunk mean
    }
    *reinterpret_cast<float4v*>(pa + (size_t)blockIdx.y * C + 4 * q) = a;
    *reinterpret_cast<float4v*>(pb + (size_t)blockIdx.y * C + 4 * q) = b;
  }
  if (pc) {                                  // block-uniform
    __syncthreads();
    sa[threadIdx.x] = hi;
    sb[threadIdx.x] = lo;
    __syncthreads();
    if (ty == 0 && q < cquads) {
      for (int i = 1; i < nty; ++i) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          hi[e] = fmaxf(hi[e], sa[i * cq + tx][e]);
          lo[e] = fminf(lo[e], sb[i * cq + tx][e]);
        }
      }
      *reinterpret_cast<float4v*>(pc + (size_t)blockIdx.y * C + 4 * q) = hi;
      if (MODE == 0) *reinterpret_cast<float4v*>(pd + (size_t)blockIdx.y * C + 4 * q) = lo;
    }
  }
}

// Merge of the chunk partials: block = 16 channels x 64 chunk lanes, fixed summation order
// (deterministic).  MODE 0 extras (single-rank batch norm: everything in this kernel, no
// framework ops in between): fin != 0 -> out_b receives invstd = rsqrt(M2 / R + eps) instead of M2
// and the running statistics are updated (momentum, unbiased variance).
struct BnFinal { int fin; float eps, momentum; float* running_mean; float* running_var; };
// single-rank calls: the tensor bound (csrc/conv.hip) is max-reduced here instead of a launch of
// its own.  MODE 0: max_c |bn(x)_c| (+ *res_bound) from this kernel's mean / invstd / extremes;
// MODE 1: the bound of |dx| from the sums of this kernel and the forward's extremes.
struct BnBound {
  float* bound;
  const float* res_bound;
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* cmax;
  const float* cmin;
  float inv_count;
};
// 256-thread workgroups (one wave per SIMD): a 1024-thread one needs four free wave slots on every SIMD of
// a CU at once and queued behind the side stream's weight-gradient workgroups in the backward pass
// (median 57 us against 12 us for the same merge in the forward pass)
constexpr int kMergeLanes = 64, kMergeCh = 4;

__device__ inline float merge_lanes(float v, float* sh, int tx, int ty) {
  __syncthreads();
  sh[ty * kMergeCh + tx] = v;
  __syncthreads();
  float t = 0.f;
#pragma unroll 16
  for (int i = 0; i < kMergeLanes; ++i) t += sh[i * kMergeCh + tx];
  return t;
}

template <int MODE>
__global__ __launch_bounds__(kMergeCh * kMergeLanes) void bn_merge(
    const float* __restrict__ pa, const float* __restrict__ pb, int64_t R, int C, int chunks, int chunk_rows,
    const float* __restrict__ invstd, float* __restrict__ out_a, float* __restrict__ out_b, BnFinal f,
    const float* __restrict__ pc, const float* __restrict__ pd, float* __restrict__ out_c,
    float* __restrict__ out_d, BnBound bb) {
  __shared__ float sh[kMergeCh * kMergeLanes];
  __shared__ float sh_bound[kMergeCh];      // the block's (up to) four bound candidates: ONE atomicMax per block
  const int tx = threadIdx.x % kMergeCh, ty = threadIdx.x / kMergeCh;
  // (the live threads of a block are lanes 0 .. 3 of its first wave: they meet wave-synchronously through sh_bound;
  // every channel used to send its own atomicMax to the ONE bound word -- C same-address atomics in a row)
  const int n_live = min(kMergeCh, C - (int)blockIdx.x * kMergeCh);
  auto publish_bound = [&](float v) {
    sh_bound[tx] = v;
    __builtin_amdgcn_wave_barrier();
    if (tx == 0) {
      unsigned m = __float_as_uint(v);       // (bit patterns, as the atomic itself compares them: a NaN stays the maximum)
      for (int i = 1; i < n_live; ++i) m = max(m, __float_as_uint(sh_bound[i]));
      atomicMax(reinterpret_cast<unsigned*>(bb.bound), m);
    }
  };
  const int c = min(blockIdx.x * kMergeCh + tx, C - 1);
  const bool live = blockIdx.x * kMergeCh + tx < C && ty == 0;
  float hi = -3.4e38f, lo = 3.4e38f;
  // Up to kMergeRegs x kMergeLanes chunks: every value this thread folds is requested up front -- pa, pb and the
  // extremes, all loads in flight together -- and the three reductions below run on registers.  The loops they replace
  // waited for a batch of four loads at a time, three dependent phases in a row: ~12 memory round trips per call, and
  // ~200 calls sit on the critical path of a training step (12-14 us each; the same arithmetic in the same order).
  constexpr int kMergeRegs = 16;
  const bool in_regs = chunks <= kMergeRegs * kMergeLanes;
  const bool ext = pc && out_c;
  float ra[kMergeRegs], rb[kMergeRegs], rc[kMergeRegs], rd[kMergeRegs];
  if (in_regs) {
#pragma unroll
    for (int j = 0; j < kMergeRegs; ++j) {
      const int i = ty + j * kMergeLanes;
      const bool ok = i < chunks;
      const size_t o = (size_t)(ok ? i : 0) * C + c;
      ra[j] = ok ? pa[o] : 0.f;
      rb[j] = ok ? pb[o] : 0.f;
      rc[j] = (ok && ext) ? pc[o] : -3.4e38f;
      rd[j] = (ok && ext && MODE == 0) ? pd[o] : 3.4e38f;
    }
  }
  // the per-channel coefficients the epilogue of a live thread needs (gamma / beta / residual bound / running statistics;
  // MODE 1: invstd, mean, extremes, gamma) are requested here, with the chunk values: behind the reductions they were
  // one more dependent round trip per call
  float p_gamma = 0.f, p_beta = 0.f, p_res = 0.f, p_rm = 0.f, p_rv = 0.f, p_is = 0.f, p_mu = 0.f, p_cmax = 0.f, p_cmin = 0.f;
  if (live) {
    if (MODE == 0) {
      if (f.fin && bb.bound) { p_gamma = bb.gamma[c]; p_beta = bb.beta[c]; p_res = bb.res_bound ? *bb.res_bound : 0.f; }
      if (f.fin && f.running_mean) { p_rm = f.running_mean[c]; p_rv = f.running_var[c]; }
    } else {
      p_is = invstd[c];
      if (bb.bound) { p_mu = bb.mean[c]; p_cmax = bb.cmax[c]; p_cmin = bb.cmin[c]; p_gamma = bb.gamma[c]; }
    }
  }
  if (ext) {                               // extremes of the chunks (max / min; MODE 1: max only)
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < kMergeRegs; ++j) {
        hi = fmaxf(hi, rc[j]);
        if (MODE == 0) lo = fminf(lo, rd[j]);
      }
    } else {
#pragma unroll 4
      for (int i = ty; i < chunks; i += kMergeLanes) {
        hi = fmaxf(hi, pc[(size_t)i * C + c]);
        if (MODE == 0) lo = fminf(lo, pd[(size_t)i * C + c]);
      }
    }
    __syncthreads();
    sh[ty * kMergeCh + tx] = hi;
    __syncthreads();
    for (int i = 0; i < kMergeLanes; ++i) hi = fmaxf(hi, sh[i * kMergeCh + tx]);
    if (MODE == 0) {
      __syncthreads();
      sh[ty * kMergeCh + tx] = lo;
      __syncthreads();
      for (int i = 0; i < kMergeLanes; ++i) lo = fminf(lo, sh[i * kMergeCh + tx]);
    }
    if (live) {
      out_c[c] = hi;
      if (MODE == 0) out_d[c] = lo;
    }
  }
  if (MODE == 0) {
    // pooled mean first, then the M2 terms (Chan): two independent sums, no division in the loops
    float sm = 0.f;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < kMergeRegs; ++j) {
        const int i = ty + j * kMergeLanes;
        if (i < chunks) {
          const float nb = (float)min((int64_t)chunk_rows, R - (int64_t)i * chunk_rows);
          sm += ra[j] * nb;
        }
      }
    } else {
#pragma unroll 4
      for (int i = ty; i < chunks; i += kMergeLanes) {
        const float nb = (float)min((int64_t)chunk_rows, R - (int64_t)i * chunk_rows);
        sm += pa[(size_t)i * C + c] * nb;
      }
    }
    const float m = merge_lanes(sm, sh, tx, ty) / (float)R;
    float m2 = 0.f;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < kMergeRegs; ++j) {
        const int i = ty + j * kMergeLanes;
        if (i < chunks) {
          const float nb = (float)min((int64_t)chunk_rows, R - (int64_t)i * chunk_rows);
          const float d = ra[j] - m;
          m2 += rb[j] + d * d * nb;
        }
      }
    } else {
#pragma unroll 4
      for (int i = ty; i < chunks; i += kMergeLanes) {
        const float nb = (float)min((int64_t)chunk_rows, R - (int64_t)i * chunk_rows);
        const float d = pa[(size_t)i * C + c] - m;
        m2 += pb[(size_t)i * C + c] + d * d * nb;
      }
    }
    m2 = merge_lanes(m2, sh, tx, ty);
    if (!live) return;
    out_a[c] = m;
    if (f.fin) {
      const float is = 1.0f / sqrtf(m2 / (float)R + f.eps);
      out_b[c] = is;
      if (bb.bound) {
        const float sc = p_gamma * is, b = p_beta;
        const float v = fmaxf(fabsf((hi - m) * sc + b), fabsf((lo - m) * sc + b)) * 1.0001f + p_res;
        publish_bound(v);
      }
      if (f.running_mean) {
        f.running_mean[c] = (1.f - f.momentum) * p_rm + f.momentum * m;
        f.running_var[c] = (1.f - f.momentum) * p_rv +
                           f.momentum * (m2 / (float)(R > 1 ? R - 1 : 1));
      }
    } else {
      out_b[c] = m2;
    }
  } else {
    float s0 = 0.f, s1 = 0.f;
    if (in_regs) {
#pragma unroll
      for (int j = 0; j < kMergeRegs; ++j)
        if (ty + j * kMergeLanes < chunks) { s0 += ra[j]; s1 += rb[j]; }
    } else {
      for (int i = ty; i < chunks; i += kMergeLanes) { s0 += pa[(size_t)i * C + c]; s1 += pb[(size_t)i * C + c]; }
    }
    s0 = merge_lanes(s0, sh, tx, ty);
    s1 = merge_lanes(s1, sh, tx, ty);
    if (!live) return;
    out_a[c] = s0;                        // sum dz
    out_b[c] = s1 * p_is;                 // sum dz * xhat
    if (bb.bound) {
      const float is = p_is, mu = p_mu;
      const float xh = fmaxf(fabsf(p_cmax - mu), fabsf(p_cmin - mu)) * is;
      const float v = fabsf(p_gamma * is) *
                      (hi + fabsf(s0) * bb.inv_count + xh * fabsf(s1 * is) * bb.inv_count) * 1.0001f;
      publish_bound(v);
    }
  }
}

// apply kernels: block = cq channel quads x (256 / cq) row lanes, grid (C / (4*cq), row tiles);
// a thread keeps its channel quad's coefficients in registers and walks down the rows of its
// tile (bn_apply_rows: ~4096 blocks per launch, at most 256 rows each)

template <bool HAS_RES>
__global__ __launch_bounds__(256) void bn_apply(const float* __restrict__ x, const float* __restrict__ res,
                                                int64_t R, int C, int cq, int apply_rows,
                                                const float* __restrict__ mean,
                                                const float* __restrict__ invstd,
                                                const float* __restrict__ gamma,
                                                const float* __restrict__ beta, int relu,
                                                float* __restrict__ y, uint2* __restrict__ yh,
                                                const float* __restrict__ ybound,
                                                unsigned char* __restrict__ mask) {
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, nty = 256 / cq;
  const int q = blockIdx.x * cq + tx;
  if (q >= (C >> 2)) return;
  const float ys = yh ? bn_pow2_scale(*ybound) : 1.f;
  const float4v mu = *reinterpret_cast<const float4v*>(mean + 4 * q);
  const float4v sc = *reinterpret_cast<const float4v*>(invstd + 4 * q) *
                     *reinterpret_cast<const float4v*>(gamma + 4 * q);
  const float4v sh = *reinterpret_cast<const float4v*>(beta + 4 * q);
  const int64_t r0 = (int64_t)blockIdx.y * apply_rows;
  const int64_t r1 = min(R, r0 + apply_rows);
#pragma unroll 4
  for (int64_t r = r0 + ty; r < r1; r += nty) {
    const size_t o = (size_t)r * C + 4 * q;
    float4v v = (*reinterpret_cast<const float4v*>(x + o) - mu) * sc + sh;
    if (HAS_RES) v += *reinterpret_cast<const float4v*>(res + o);
    if (relu) {
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    if (y) *reinterpret_cast<float4v*>(y + o) = v;
    if (yh) store_hl8_quad(yh, (size_t)r, C, q, v, ys);
    if (mask)
      mask[(size_t)r * (C >> 2) + q] = (unsigned char)((v[0] > 0.f) | ((v[1] > 0.f) << 1) | ((v[2] > 0.f) << 2) |
                                                       ((v[3] > 0.f) << 3));
  }
}

template <int MASK>
__global__ __launch_bounds__(256) void bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ y,
                                                    const float* __restrict__ x, int64_t R, int C, int cq,
                                                    int apply_rows, const float* __restrict__ mean,
                                                    const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ sum_dz,
                                                    const float* __restrict__ sum_dz_xhat, float inv_count,
                                                    const float* __restrict__ count_dev,
                                                    float* __restrict__ dx, float* __restrict__ dres,
                                                    const unsigned char* __restrict__ yh, uint2* __restrict__ dxh,
                                                    const float* __restrict__ dxbound) {
  const int tx = threadIdx.x % cq, ty = threadIdx.x / cq, nty = 256 / cq;
  const int q = blockIdx.x * cq + tx;
  if (q >= (C >> 2)) return;
  if (count_dev) inv_count = 1.f / count_dev[0];      // pooled row count of all ranks (SyncBatchNorm)
  const float ds = dxh ? bn_pow2_scale(*dxbound) : 1.f;
  float4v mu = {0.f, 0.f, 0.f, 0.f}, is = mu, gi = mu, c0 = mu, c1 = mu;
  if (dx || dxh) {
    mu = *reinterpret_cast<const float4v*>(mean + 4 * q);
    is = *reinterpret_cast<const float4v*>(invstd + 4 * q);
    gi = *reinterpret_cast<const float4v*>(gamma + 4 * q) * is;
    c0 = *reinterpret_cast<const float4v*>(sum_dz + 4 * q) * inv_count;
    c1 = *reinterpret_cast<const float4v*>(sum_dz_xhat + 4 * q) * inv_count;
  }
  const int64_t r0 = (int64_t)blockIdx.y * apply_rows;
  const int64_t r1 = min(R, r0 + apply_rows);
#pragma unroll 4
  for (int64_t r = r0 + ty; r < r1; r += nty) {
    const size_t o = (size_t)r * C + 4 * q;
    float4v dz = *reinterpret_cast<const float4v*>(dy + o);
    if (MASK == 1) {
      const float4v yv = *reinterpret_cast<const float4v*>(y + o);
#pragma unroll
      for (int e = 0; e < 4; ++e) dz[e] = yv[e] > 0.f ? dz[e] : 0.f;
    } else if (MASK == 2) {
      dz = relu_mask_bits(dz, yh, (size_t)r, C, q);
    }
    if (dres) *reinterpret_cast<float4v*>(dres + o) = dz;
    if (dx || dxh) {
      const float4v xh = (*reinterpret_cast<const float4v*>(x + o) - mu) * is;
      const float4v g = gi * (dz - c0 - xh * c1);
      if (dx) *reinterpret_cast<float4v*>(dx + o) = g;
      if (dxh) store_hl8_quad(dxh, (size_t)r, C, q, g, ds);
    }
  }
}

// upper bounds of |y| and |dx| over the whole tensor from the per-channel extremes (one block)
__device__ __forceinline__ float block_max_256(float v, float* sh) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, kWave));
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

__global__ __launch_bounds__(256) void bn_bound_fwd(int C, const float* __restrict__ cmax,
                                                    const float* __restrict__ cmin,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ beta,
                                                    const float* __restrict__ res_bound, float* __restrict__ out) {
  __shared__ float sh[4];
  float m = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float sc = gamma[c] * invstd[c];
    m = fmaxf(m, fmaxf(fabsf((cmax[c] - mean[c]) * sc + beta[c]), fabsf((cmin[c] - mean[c]) * sc + beta[c])));
  }
  m = block_max_256(m, sh);
  if (threadIdx.x == 0) out[0] = m * 1.0001f + (res_bound ? res_bound[0] : 0.f);
}

__global__ __launch_bounds__(256) void bn_bound_bwd(int C, const float* __restrict__ max_dz,
                                                    const float* __restrict__ cmax,
                                                    const float* __restrict__ cmin,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ invstd,
                                                    const float* __restrict__ gamma,
                                                    const float* __restrict__ sum_dz,
                                                    const float* __restrict__ sum_dz_xhat, float inv_count,
                                                    const float* __restrict__ count_dev,
                                                    float* __restrict__ out) {
  __shared__ float sh[4];
  float m = 0.f;
  if (count_dev) inv_count = 1.f / count_dev[0];
  for (int c = threadIdx.x; c < C; c += 256) {
    const float xh = fmaxf(fabsf(cmax[c] - mean[c]), fabsf(cmin[c] - mean[c])) * invstd[c];
    m = fmaxf(m, fabsf(gamma[c] * invstd[c]) *
                     (max_dz[c] + fabsf(sum_dz[c]) * inv_count + xh * fabsf(sum_dz_xhat[c]) * inv_count));
  }
  m = block_max_256(m, sh);
  if (threadIdx.x == 0) out[0] = m * 1.0001f;
}

__global__ __launch_bounds__(256) void bn_finalize(int C, const float* __restrict__ mean,
                                                   const float* __restrict__ m2, float count, float eps,
                                                   float momentum, float* __restrict__ running_mean,
                                                   float* __restrict__ running_var, float* __restrict__ invstd) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  invstd[c] = 1.0f / sqrtf(m2[c] / count + eps);
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * mean[c];
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2[c] / (count > 1.f ? count - 1.f : 1.f));
  }
}

// SyncBatchNorm: pooled statistics of `world` ranks (Chan) + finalisation in one launch.
// stats [world][3][C] = per-rank (count, mean, M2) as gathered by the caller.
__global__ __launch_bounds__(256) void bn_finalize_ranks(int C, int world, const float* __restrict__ stats,
                                                         float eps, float momentum,
                                                         float* __restrict__ running_mean,
                                                         float* __restrict__ running_var,
                                                         float* __restrict__ mean_out,
                                                         float* __restrict__ invstd,
                                                         float* __restrict__ total_out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float total = 0.f, sm = 0.f;
  for (int r = 0; r < world; ++r) {
    const float n = stats[((size_t)r * 3 + 0) * C + c];
    total += n;
    sm += n * stats[((size_t)r * 3 + 1) * C + c];
  }
  const float m = sm / total;
  float m2 = 0.f;
  for (int r = 0; r < world; ++r) {
    const float n = stats[((size_t)r * 3 + 0) * C + c];
    const float d = stats[((size_t)r * 3 + 1) * C + c] - m;
    m2 += stats[((size_t)r * 3 + 2) * C + c] + d * d * n;
  }
  mean_out[c] = m;
  invstd[c] = 1.0f / sqrtf(m2 / total + eps);
  if (total_out && c == 0) total_out[0] = total;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * m;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (m2 / (total > 1.f ? total - 1.f : 1.f));
  }
}

inline int bn_cq(int C);
inline int bn_apply_rows(int64_t R, int C);
inline int bn_cq(int C) { const int q = C >> 2; return q >= 64 ? 64 : (q >= 32 ? 32 : (q >= 16 ? 16 : (q >= 8 ? 8 : 4))); }
// rows per partial-statistics chunk: enough chunks to fill the chip (~1024 blocks), at most 1024
// of them (merge depth), never fewer than 128 rows each
inline int bn_chunk_rows(int64_t R, int C) {
  const int cq = bn_cq(C), col_blocks = ((C >> 2) + cq - 1) / cq;
  const int64_t target = std::min<int64_t>(1024, std::max<int64_t>(1, 1024 / col_blocks));
  return (int)std::max<int64_t>(128, (R + target - 1) / target);
}
inline int bn_chunks(int64_t R, int C) { const int cr = bn_chunk_rows(R, C); return (int)((R + cr - 1) / cr); }
inline int bn_apply_rows(int64_t R, int C) {
  const int cq = bn_cq(C), col_blocks = ((C >> 2) + cq - 1) / cq, nty = 256 / cq;
  const int64_t rows = (R * col_blocks + 4095) / 4096;
  return (int)std::min<int64_t>(256, std::max<int64_t>(nty, (rows + nty - 1) / nty * nty));
}
inline bool bn_ok(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace spml

using namespace spml;

// kernel variant dispatch on the (uniform) optional inputs: no pointer tests inside the row loops
#define SPML_BN_BY_MASK(KERNEL, MASK, ...)                                        \
  do {                                                                            \
    if ((MASK) == 1) hipLaunchKernelGGL((KERNEL<1>), __VA_ARGS__);                \
    else if ((MASK) == 2) hipLaunchKernelGGL((KERNEL<2>), __VA_ARGS__);           \
    else hipLaunchKernelGGL((KERNEL<0>), __VA_ARGS__);                            \
  } while (0)
#define SPML_BN_PARTIAL1(MASK, ...)                                               \
  do {                                                                            \
    if ((MASK) == 1) hipLaunchKernelGGL((bn_partial<1, 1>), __VA_ARGS__);         \
    else if ((MASK) == 2) hipLaunchKernelGGL((bn_partial<1, 2>), __VA_ARGS__);    \
    else hipLaunchKernelGGL((bn_partial<1, 0>), __VA_ARGS__);                     \
  } while (0)
#define SPML_BN_APPLY(HAS_RES, ...)                                               \
  do {                                                                            \
    if (HAS_RES) hipLaunchKernelGGL((bn_apply<true>), __VA_ARGS__);               \
    else hipLaunchKernelGGL((bn_apply<false>), __VA_ARGS__);                      \
  } while (0)

extern "C" size_t spml_bn_workspace_bytes(int64_t R, int C) {
  if (R <= 0 || C <= 0) return 0;
  return (size_t)4 * bn_chunks(R, C) * C * 4 + 256;
}

extern "C" int spml_bn_stats_f32(const float* x, int64_t R, int C, float* mean, float* m2, void* ws,
                                 size_t ws_bytes, void* stream) {
  if (!x || !mean || !m2 || R <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(x) || !bn_ok(mean) || !bn_ok(m2)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial<0, 0>), dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, R, C, cq, crows, pa, pb, (const unsigned char*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R, C, chunks, crows,
                     (const float*)nullptr, mean, m2, BnFinal{0, 0.f, 0.f, nullptr, nullptr}, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, BnBound{});
  return launch_status();
}

extern "C" int spml_bn_act_apply_f32(const float* x, const float* residual, int64_t R, int C,
                                     const float* mean, const float* invstd, const float* gamma,
                                     const float* beta, int relu, float* y, void* stream) {
  if (!x || !mean || !invstd || !gamma || !beta || !y || R <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(x) || !bn_ok(y) || (residual && !bn_ok(residual)) || !bn_ok(mean) ||
      !bn_ok(invstd) || !bn_ok(gamma) || !bn_ok(beta))
    return SPML_ERR_UNSUPPORTED;
  const int cq = bn_cq(C), arows = bn_apply_rows(R, C);
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_APPLY(residual != nullptr, grid, dim3(256), 0, (hipStream_t)stream, x, residual, R, C, cq, arows, mean, invstd,
                     gamma, beta, relu, y, (uint2*)nullptr, (const float*)nullptr, (unsigned char*)nullptr);
  return launch_status();
}

extern "C" int spml_bn_act_bwd_reduce_f32(const float* dy, const float* y, const float* x, int64_t R,
                                          int C, const float* mean, const float* invstd,
                                          float* sum_dz, float* sum_dz_xhat, void* ws, size_t ws_bytes,
                                          void* stream) {
  if (!dy || !x || !mean || !invstd || !sum_dz || !sum_dz_xhat || R <= 0 || C <= 0)
    return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(dy) || !bn_ok(x) || (y && !bn_ok(y)) || !bn_ok(mean)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  SPML_BN_PARTIAL1(y ? 1 : 0, dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x, dy, y, mean, R, C, cq, crows,
                   pa, pb, (const unsigned char*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<1>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R, C, chunks, crows, invstd, sum_dz,
                     sum_dz_xhat, BnFinal{0, 0.f, 0.f, nullptr, nullptr}, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, BnBound{});
  return launch_status();
}

extern "C" int spml_bn_act_bwd_apply_f32(const float* dy, const float* y, const float* x, int64_t R,
                                         int C, const float* mean, const float* invstd,
                                         const float* gamma, const float* sum_dz,
                                         const float* sum_dz_xhat, double count,
                                         const float* count_dev, float* dx,
                                         float* d_residual, void* stream) {
  if (!dy || R <= 0 || C <= 0 || (!dx && !d_residual) || (count <= 0 && !count_dev)) return SPML_ERR_INVALID_ARG;
  if (dx && (!x || !mean || !invstd || !gamma || !sum_dz || !sum_dz_xhat)) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(dy) || (y && !bn_ok(y)) || (dx && !bn_ok(dx)) || (d_residual && !bn_ok(d_residual)))
    return SPML_ERR_UNSUPPORTED;
  const int cq = bn_cq(C), arows = bn_apply_rows(R, C);
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_BY_MASK(bn_bwd_apply, y ? 1 : 0, grid, dim3(256), 0, (hipStream_t)stream, dy, y, x, R, C, cq, arows, mean, invstd,
                     gamma, sum_dz, sum_dz_xhat, count > 0 ? (float)(1.0 / count) : 0.f, count_dev, dx, d_residual,
                     (const unsigned char*)nullptr, (uint2*)nullptr, (const float*)nullptr);
  return launch_status();
}

// Single-rank batch norm: the whole forward / backward in one call each (three launches, no
// framework ops in between).
extern "C" int spml_bn_act_fwd_f32(const float* x, const float* residual, int64_t R, int C,
                                   const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, int relu, float* y,
                                   float* mean, float* invstd, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !invstd || R <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(x) || !bn_ok(y) || (residual && !bn_ok(residual)) || !bn_ok(mean) ||
      !bn_ok(invstd) || !bn_ok(gamma) || !bn_ok(beta))
    return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C), arows = bn_apply_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial<0, 0>), dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, R, C, cq, crows, pa, pb, (const unsigned char*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R, C, chunks, crows,
                     (const float*)nullptr, mean, invstd, BnFinal{1, eps, momentum, running_mean, running_var}, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, BnBound{});
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_APPLY(residual != nullptr, grid, dim3(256), 0, s, x, residual, R, C, cq, arows, mean, invstd, gamma, beta, relu, y, (uint2*)nullptr, (const float*)nullptr, (unsigned char*)nullptr);
  return launch_status();
}

extern "C" int spml_bn_act_bwd_f32(const float* dy, const float* y, const float* x, int64_t R, int C,
                                   const float* mean, const float* invstd, const float* gamma,
                                   float* d_gamma, float* d_beta, float* dx, float* d_residual,
                                   void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !d_gamma || !d_beta || R <= 0 || C <= 0)
    return SPML_ERR_INVALID_ARG;
  int rc = spml_bn_act_bwd_reduce_f32(dy, y, x, R, C, mean, invstd, d_beta, d_gamma, ws, ws_bytes, stream);
  if (rc != SPML_OK) return rc;
  if (!dx && !d_residual) return SPML_OK;
  return spml_bn_act_bwd_apply_f32(dy, y, x, R, C, mean, invstd, gamma, d_beta, d_gamma, (double)R, nullptr, dx,
                                   d_residual, stream);
}

// ---- variants that also produce the split-f16 ("hl8") copies the matrix-core convolutions
// consume (csrc/conv.hip), with the tensor bounds that fix their scales -----------------------
extern "C" int spml_bn_stats_ext_f32(const float* x, int64_t R, int C, float* mean, float* m2, float* cmax,
                                     float* cmin, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !mean || !m2 || !cmax || !cmin || R <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || !bn_ok(x) || !bn_ok(mean) || !bn_ok(m2)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  float* pc = pb + (size_t)chunks * C;
  float* pd = pc + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial<0, 0>), dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, R, C, cq, crows, pa, pb,
                     (const unsigned char*)nullptr, pc, pd, (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R, C, chunks, crows,
                     (const float*)nullptr, mean, m2, BnFinal{0, 0.f, 0.f, nullptr, nullptr}, (const float*)pc,
                     (const float*)pd, cmax, cmin, BnBound{});
  return launch_status();
}

// The same statistics pooled from the chunk statistics a convolution's epilogue left
// (spml_conv_hl8_stats_f32): one launch, x is not read.
extern "C" int spml_bn_stats_ext_chunks_f32(const float* chunk_stats, int chunks, int chunk_rows, int64_t R, int C,
                                            float* mean, float* m2, float* cmax, float* cmin, void* stream) {
  if (!chunk_stats || !mean || !m2 || !cmax || !cmin || R <= 0 || C <= 0 || chunks <= 0 || chunk_rows <= 0 ||
      (int64_t)chunks * chunk_rows < R || (int64_t)(chunks - 1) * chunk_rows >= R)
    return SPML_ERR_INVALID_ARG;
  const float* pa = chunk_stats;
  const float* pb = pa + (size_t)chunks * C;
  const float* pc = pb + (size_t)chunks * C;
  const float* pd = pc + (size_t)chunks * C;
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0,
                     (hipStream_t)stream, pa, pb, R, C, chunks, chunk_rows, (const float*)nullptr, mean, m2,
                     BnFinal{0, 0.f, 0.f, nullptr, nullptr}, pc, pd, cmax, cmin, BnBound{});
  return launch_status();
}

extern "C" int spml_bn_finalize_f32(const float* mean, const float* m2, int C, double count, float eps,
                                    float momentum, float* running_mean, float* running_var, float* invstd,
                                    void* stream) {
  if (!mean || !m2 || !invstd || C <= 0 || count <= 0) return SPML_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_finalize, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, mean, m2,
                     (float)count, eps, momentum, running_mean, running_var, invstd);
  return launch_status();
}

extern "C" int spml_bn_act_apply_hl8_f32(const float* x, const float* residual, const float* residual_bound,
                                         int64_t R, int C, const float* mean, const float* invstd,
                                         const float* gamma, const float* beta, const float* cmax,
                                         const float* cmin, int relu, float* y, void* y_hl8, float* y_bound,
                                         unsigned char* relu_mask, void* stream) {
  if (!x || !mean || !invstd || !gamma || !beta || (!y && !y_hl8) || R <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((y_hl8 || y_bound) && (!cmax || !cmin || !y_bound || (residual && !residual_bound))) return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(x) || (y && !bn_ok(y)) || (y_hl8 && !bn_ok(y_hl8)) || (residual && !bn_ok(residual)) ||
      !bn_ok(mean) || !bn_ok(invstd) || !bn_ok(gamma) || !bn_ok(beta))
    return SPML_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (y_bound)
    hipLaunchKernelGGL(bn_bound_fwd, dim3(1), dim3(256), 0, s, C, cmax, cmin, mean, invstd, gamma, beta,
                       residual ? residual_bound : (const float*)nullptr, y_bound);
  const int cq = bn_cq(C), arows = bn_apply_rows(R, C);
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_APPLY(residual != nullptr, grid, dim3(256), 0, s, x, residual, R, C, cq, arows, mean, invstd, gamma, beta, relu, y,
                static_cast<uint2*>(y_hl8), (const float*)y_bound, relu_mask);
  return launch_status();
}

extern "C" int spml_bn_act_bwd_reduce_ext_f32(const float* dy, const float* y, const unsigned char* relu_mask,
                                              const float* x,
                                              int64_t R, int C, const float* mean, const float* invstd,
                                              float* sum_dz, float* sum_dz_xhat, float* max_dz, void* ws,
                                              size_t ws_bytes, void* stream) {
  if (!dy || !x || !mean || !invstd || !sum_dz || !sum_dz_xhat || !max_dz || R <= 0 || C <= 0)
    return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(dy) || !bn_ok(x) || (y && !bn_ok(y)) || !bn_ok(mean)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  float* pc = pb + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  SPML_BN_PARTIAL1(y ? 1 : (relu_mask ? 2 : 0), dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x, dy, y, mean,
                   R, C, cq, crows, pa, pb, relu_mask, pc, (float*)nullptr, (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<1>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R, C, chunks, crows,
                     invstd, sum_dz, sum_dz_xhat, BnFinal{0, 0.f, 0.f, nullptr, nullptr}, (const float*)pc,
                     (const float*)nullptr, max_dz, (float*)nullptr, BnBound{});
  return launch_status();
}

extern "C" int spml_bn_act_bwd_apply_hl8_f32(const float* dy, const float* y, const unsigned char* relu_mask,
                                             const float* x,
                                             int64_t R, int C, const float* mean, const float* invstd,
                                             const float* gamma, const float* sum_dz, const float* sum_dz_xhat,
                                             const float* max_dz, const float* cmax, const float* cmin,
                                             double count, const float* count_dev, float* dx, void* dx_hl8,
                                             float* dx_bound, float* d_residual, void* stream) {
  if (!dy || R <= 0 || C <= 0 || (!dx && !dx_hl8 && !d_residual) || (count <= 0 && !count_dev))
    return SPML_ERR_INVALID_ARG;
  const float inv_count = count > 0 ? (float)(1.0 / count) : 0.f;
  if ((dx || dx_hl8) && (!x || !mean || !invstd || !gamma || !sum_dz || !sum_dz_xhat)) return SPML_ERR_INVALID_ARG;
  if (dx_hl8 && (!max_dz || !cmax || !cmin || !dx_bound)) return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(dy) || (y && !bn_ok(y)) || (dx && !bn_ok(dx)) ||
      (dx_hl8 && !bn_ok(dx_hl8)) || (d_residual && !bn_ok(d_residual)))
    return SPML_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (dx_hl8)
    hipLaunchKernelGGL(bn_bound_bwd, dim3(1), dim3(256), 0, s, C, max_dz, cmax, cmin, mean, invstd, gamma, sum_dz,
                       sum_dz_xhat, inv_count, count_dev, dx_bound);
  const int cq = bn_cq(C), arows = bn_apply_rows(R, C);
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_BY_MASK(bn_bwd_apply, y ? 1 : (relu_mask ? 2 : 0), grid, dim3(256), 0, s, dy, y, x, R, C, cq, arows, mean, invstd, gamma, sum_dz,
                     sum_dz_xhat, inv_count, count_dev, dx, d_residual, relu_mask,
                  static_cast<uint2*>(dx_hl8), (const float*)dx_bound);
  return launch_status();
}

// Single-rank forms of the two halves: statistics + finalisation + bound + apply (three launches),
// reduction + bound + apply (three launches).
extern "C" int spml_bn_fwd_hl8_f32(const float* x, const float* residual, const float* residual_bound, int64_t R,
                                   int C, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float momentum, float eps, int relu, float* y, void* y_hl8,
                                   float* y_bound, unsigned char* relu_mask, float* mean, float* invstd,
                                   float* cmax, float* cmin, void* ws, size_t ws_bytes, void* stream) {
  if (!x || !gamma || !beta || (!y && !y_hl8) || !y_bound || !mean || !invstd || !cmax || !cmin || R <= 0 || C <= 0 ||
      (residual && !residual_bound))
    return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(x) || (y && !bn_ok(y)) || (y_hl8 && !bn_ok(y_hl8)) || (residual && !bn_ok(residual)) ||
      !bn_ok(mean) || !bn_ok(invstd) || !bn_ok(gamma) || !bn_ok(beta))
    return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C), arows = bn_apply_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  float* pc = pb + (size_t)chunks * C;
  float* pd = pc + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial<0, 0>), dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x,
                     (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, R, C, cq, crows, pa, pb,
                     (const unsigned char*)nullptr, pc, pd, y_bound);
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R,
                     C, chunks, crows, (const float*)nullptr, mean, invstd,
                     BnFinal{1, eps, momentum, running_mean, running_var}, (const float*)pc, (const float*)pd, cmax,
                     cmin, BnBound{y_bound, residual ? residual_bound : nullptr, gamma, beta, nullptr, nullptr, nullptr, 0.f});
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_APPLY(residual != nullptr, grid, dim3(256), 0, s, x, residual, R, C, cq, arows, mean, invstd, gamma, beta,
                relu, y, static_cast<uint2*>(y_hl8), (const float*)y_bound, relu_mask);
  return launch_status();
}

// The same forward when the producer of x (a matrix-core convolution, spml_conv_hl8_stats_f32) has already
// left the chunk statistics [4][chunks][C] (mean, M2, max, min per chunk of chunk_rows rows) and reset
// *y_bound: two launches, x is read once.
extern "C" int spml_bn_fwd_hl8_chunks_f32(const float* x, const float* chunk_stats, int chunks, int chunk_rows,
                                          const float* residual, const float* residual_bound, int64_t R, int C,
                                          const float* gamma, const float* beta, float* running_mean,
                                          float* running_var, float momentum, float eps, int relu, float* y,
                                          void* y_hl8, float* y_bound, unsigned char* relu_mask, float* mean,
                                          float* invstd, float* cmax, float* cmin, void* stream) {
  if (!x || !chunk_stats || !gamma || !beta || (!y && !y_hl8) || !y_bound || !mean || !invstd || !cmax || !cmin ||
      R <= 0 || C <= 0 || chunks <= 0 || chunk_rows <= 0 || (int64_t)chunks * chunk_rows < R ||
      (int64_t)(chunks - 1) * chunk_rows >= R || (residual && !residual_bound))
    return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(x) || (y && !bn_ok(y)) || (y_hl8 && !bn_ok(y_hl8)) || (residual && !bn_ok(residual)) ||
      !bn_ok(mean) || !bn_ok(invstd) || !bn_ok(gamma) || !bn_ok(beta))
    return SPML_ERR_UNSUPPORTED;
  const int cq = bn_cq(C), arows = bn_apply_rows(R, C);
  const float* pa = chunk_stats;
  const float* pb = pa + (size_t)chunks * C;
  const float* pc = pb + (size_t)chunks * C;
  const float* pd = pc + (size_t)chunks * C;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_merge<0>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R,
                     C, chunks, chunk_rows, (const float*)nullptr, mean, invstd,
                     BnFinal{1, eps, momentum, running_mean, running_var}, pc, pd, cmax, cmin,
                     BnBound{y_bound, residual ? residual_bound : nullptr, gamma, beta, nullptr, nullptr, nullptr, 0.f});
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_APPLY(residual != nullptr, grid, dim3(256), 0, s, x, residual, R, C, cq, arows, mean, invstd, gamma, beta,
                relu, y, static_cast<uint2*>(y_hl8), (const float*)y_bound, relu_mask);
  return launch_status();
}

extern "C" int spml_bn_bwd_hl8_f32(const float* dy, const float* y, const unsigned char* relu_mask, const float* x,
                                   int64_t R, int C, const float* mean, const float* invstd, const float* gamma,
                                   const float* cmax, const float* cmin, float* d_gamma, float* d_beta, float* dx,
                                   void* dx_hl8, float* dx_bound, float* d_residual, void* ws, size_t ws_bytes,
                                   void* stream) {
  if (!dy || !x || !mean || !invstd || !gamma || !cmax || !cmin || !d_gamma || !d_beta || R <= 0 || C <= 0 ||
      (dx_hl8 && !dx_bound))
    return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !bn_ok(dy) || !bn_ok(x) || (y && !bn_ok(y)) || (dx && !bn_ok(dx)) || (dx_hl8 && !bn_ok(dx_hl8)) ||
      (d_residual && !bn_ok(d_residual)) || !bn_ok(mean))
    return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_bn_workspace_bytes(R, C)) return SPML_ERR_WORKSPACE;
  const int chunks = bn_chunks(R, C), cq = bn_cq(C), crows = bn_chunk_rows(R, C), arows = bn_apply_rows(R, C);
  float* pa = static_cast<float*>(ws);
  float* pb = pa + (size_t)chunks * C;
  float* pc = pb + (size_t)chunks * C;
  float* max_dz = pc + (size_t)chunks * C;              // the fourth quarter of the workspace is free here
  hipStream_t s = (hipStream_t)stream;
  const int mask = y ? 1 : (relu_mask ? 2 : 0);
  SPML_BN_PARTIAL1(mask, dim3(((C >> 2) + cq - 1) / cq, chunks), dim3(256), 0, s, x, dy, y, mean, R, C, cq, crows, pa, pb,
                   relu_mask, pc, (float*)nullptr, dx_hl8 ? dx_bound : (float*)nullptr);
  hipLaunchKernelGGL(bn_merge<1>, dim3((C + kMergeCh - 1) / kMergeCh), dim3(kMergeCh * kMergeLanes), 0, s, pa, pb, R,
                     C, chunks, crows, invstd, d_beta, d_gamma, BnFinal{0, 0.f, 0.f, nullptr, nullptr},
                     (const float*)pc, (const float*)nullptr, max_dz, (float*)nullptr,
                     BnBound{dx_hl8 ? dx_bound : nullptr, nullptr, gamma, nullptr, mean, cmax, cmin, (float)(1.0 / (double)R)});
  if (!dx && !dx_hl8 && !d_residual) return launch_status();
  const dim3 grid(((C >> 2) + cq - 1) / cq, (unsigned)((R + arows - 1) / arows));
  SPML_BN_BY_MASK(bn_bwd_apply, mask, grid, dim3(256), 0, s, dy, y, x, R, C, cq, arows, mean, invstd, gamma,
                  (const float*)d_beta, (const float*)d_gamma, (float)(1.0 / (double)R), (const float*)nullptr, dx,
                  d_residual, relu_mask, static_cast<uint2*>(dx_hl8), (const float*)dx_bound);
  return launch_status();
}

extern "C" int spml_bn_finalize_ranks_f32(const float* stats, int world, int C, float eps, float momentum,
                                          float* running_mean, float* running_var, float* mean, float* invstd,
                                          float* total_count, void* stream) {
  if (!stats || !mean || !invstd || world < 1 || C <= 0) return SPML_ERR_INVALID_ARG;
  hipLaunchKernelGGL(bn_finalize_ranks, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, C, world, stats, eps,
                     momentum, running_mean, running_var, mean, invstd, total_count);
  return launch_status();
}
