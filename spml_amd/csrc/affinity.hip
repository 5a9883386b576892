// N3: pixel x pixel affinity -> random-walk transition matrix.
// Replaces pyscripts/inference/pseudo_camrw_crf.py:146-158 (= pseudo_softmaxrw_crf.py:
// 137-163): per augmented view  aff_b = exp(5 * E_b^T E_b - 5)  ([n,n], n = (H/8)*(W/8)),
// mean over the views, `** 20`, division by the column sums -- five [n,n] temporaries per
// view in the reference.  Here the similarity only exists as 32x32 MFMA accumulator tiles
// (split-f16 x2, fp32-class accuracy as everywhere else); one wave owns a 32-row stripe
// and walks all column tiles, so the row sums (== column sums: the matrix is symmetric)
// are complete inside the wave -- no atomics, deterministic.  The walk itself
// (T <- T T, six times) is a plain fp32 library GEMM on the host side.
#include "common.hpp"

namespace spml {
namespace {

// emb [B][C][n] (unit columns) -> fragment-major split-f16 [B][tiles][KS][64][8]:
// lane holds E[32*tile + (lane&31)][16*ks + 8*(lane>>5) + e]; reads are coalesced along n
__global__ __launch_bounds__(256) void affinity_prep(const float* __restrict__ emb, int B, int C,
                                                     int64_t n, int KS, _Float16* __restrict__ oh,
                                                     _Float16* __restrict__ ol) {
  const int64_t tiles = (n + 31) / 32;
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= (int64_t)B * tiles * KS) return;
  const int lane = threadIdx.x & 63;
  const int ks = (int)(f % KS);
  const int64_t bt = f / KS;
  const int64_t tile = bt % tiles;
  const int b = (int)(bt / tiles);
  const int64_t i = 32 * tile + (lane & 31);
  const int c0 = 16 * ks + 8 * (lane >> 5);
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (i < n && c0 + e < C) v = emb[((size_t)b * C + c0 + e) * n + i];
    _Float16 a, r;
    split_f16(v, a, r);
    h[e] = a; l[e] = r;
  }
  *reinterpret_cast<half8*>(oh + ((size_t)f * 64 + lane) * 8) = h;
  *reinterpret_cast<half8*>(ol + ((size_t)f * 64 + lane) * 8) = l;
}

__device__ __forceinline__ float ipow(float x, int p) {
  float r = 1.f;
  while (p > 0) {
    if (p & 1) r *= x;
    x *= x;
    p >>= 1;
  }
  return r;
}

// one wave = one 32-row stripe; out[i][j] = (mean_b exp(scale * <e_i, e_j> - scale)) ^ power
__global__ __launch_bounds__(64) void affinity_kernel(const _Float16* __restrict__ eh,
                                                      const _Float16* __restrict__ el, int B,
                                                      int KS, int64_t n, float scale_log2e,
                                                      int power, float* __restrict__ out,
                                                      float* __restrict__ rowsum) {
  const int lane = threadIdx.x;
  const int half = lane >> 5, j = lane & 31;
  const int64_t tiles = (n + 31) / 32;
  const int64_t rt = blockIdx.x;
  const float inv_b = 1.0f / (float)B;
  float rs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) rs[r] = 0.f;
  for (int64_t ct = 0; ct < tiles; ++ct) {
    float mean[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) mean[r] = 0.f;
    for (int b = 0; b < B; ++b) {
      float16v zh, zx;
#pragma unroll
      for (int r = 0; r < 16; ++r) { zh[r] = 0.f; zx[r] = 0.f; }
      const size_t ra = (((size_t)b * tiles + rt) * KS) * 512 + (size_t)lane * 8;
      const size_t ca = (((size_t)b * tiles + ct) * KS) * 512 + (size_t)lane * 8;
      for (int ks = 0; ks < KS; ++ks) {
        const half8 a_h = *reinterpret_cast<const half8*>(eh + ra + (size_t)ks * 512);
        const half8 a_l = *reinterpret_cast<const half8*>(el + ra + (size_t)ks * 512);
        const half8 b_h = *reinterpret_cast<const half8*>(eh + ca + (size_t)ks * 512);
        const half8 b_l = *reinterpret_cast<const half8*>(el + ca + (size_t)ks * 512);
        zh = mfma32(a_h, b_h, zh);
        zx = mfma32(a_h, b_l, zx);
        zx = mfma32(a_l, b_h, zx);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float dot = zh[r] + zx[r] * kSplitInv;
        mean[r] += __builtin_amdgcn_exp2f((dot - 1.0f) * scale_log2e);   // exp(scale*dot - scale)
      }
    }
    const int64_t col = 32 * ct + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
      const float v = ipow(mean[r] * inv_b, power);
      if (row < n && col < n) {
        out[(size_t)row * n + col] = v;
        rs[r] += v;
      }
    }
  }
  // row sums: combine the 32 columns held by the lanes of each half
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float v = rs[r];
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int64_t row = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * half;
    if (j == 0 && row < n) rowsum[row] = v;
  }
}

// trans[i][j] = pow[i][j] / colsum[j]   (colsum == rowsum: symmetric matrix)
__global__ __launch_bounds__(256) void affinity_normalize(float* __restrict__ m,
                                                          const float* __restrict__ colsum,
                                                          int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n * n) return;
  m[i] = m[i] / colsum[i % n];
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" size_t spml_affinity_workspace_bytes(int B, int C, int64_t n) {
  if (B <= 0 || C <= 0 || n <= 0) return 0;
  const size_t ks = (size_t)(C + 15) / 16, tiles = (size_t)(n + 31) / 32;
  return align_up((size_t)B * tiles * ks * 512 * 2, 256) * 2 + align_up((size_t)n * 4, 256);
}

extern "C" int spml_affinity_transition_f32(const float* emb, int B, int C, int64_t n, float scale,
                                            int power, float* trans, void* ws, size_t ws_bytes,
                                            void* stream) {
  if (!emb || !trans || B <= 0 || C <= 0 || n <= 0 || power < 1) return SPML_ERR_INVALID_ARG;
  if (C > 1024) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_affinity_workspace_bytes(B, C, n)) return SPML_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int KS = (C + 15) / 16;
  const int64_t tiles = (n + 31) / 32;
  const size_t frag = align_up((size_t)B * tiles * KS * 512 * 2, 256);
  unsigned char* b = static_cast<unsigned char*>(ws);
  _Float16* eh = reinterpret_cast<_Float16*>(b);
  _Float16* el = reinterpret_cast<_Float16*>(b + frag);
  float* rowsum = reinterpret_cast<float*>(b + 2 * frag);
  const int64_t nfrag = (int64_t)B * tiles * KS;
  hipLaunchKernelGGL(affinity_prep, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, emb, B, C, n,
                     KS, eh, el);
  hipLaunchKernelGGL(affinity_kernel, dim3((unsigned)tiles), dim3(64), 0, s, eh, el, B, KS, n,
                     scale * 1.4426950408889634f, power, trans, rowsum);
  hipLaunchKernelGGL(affinity_normalize, dim3((unsigned)((n * n + 255) / 256)), dim3(256), 0, s,
                     trans, rowsum, n);
  return launch_status();
}
