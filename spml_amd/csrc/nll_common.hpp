// Shared declarations of the NLL kernels (csrc/nll.hip, csrc/nll_de3.hip): problem sizes, kernel
// arguments, per-pixel backward coefficients, the positive-set predicate and the LDS-DMA helper.
#pragma once

#include <type_traits>

#include "common.hpp"

namespace spml {

struct NllDims {
  int64_t P, M;
  int D, KS, DT;            // k-steps of 16 channels, d-tiles of 32 channels
  int64_t PT, MT;           // 32-row tiles of pixels / prototypes
};

__host__ __device__ inline NllDims nll_dims(int64_t P, int64_t M, int D) {
  NllDims n;
  n.P = P; n.M = M; n.D = D;
  n.KS = (D + 15) / 16;
  n.DT = (D + 31) / 32;
  n.PT = (P + 31) / 32;
  n.MT = (M + 31) / 32;
  return n;
}

// row of a 32x32 accumulator tile held by (register r, lane half h)
__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// prototype tiles per chunk of the v2 / v3 kernels -- a constant, so that the summation order of a pixel's
// sums depends on M alone (3072 prototypes per chunk: 12 KB of codes in LDS)
constexpr int kFwd2TilesPerChunk = 96;

// v2 / v3 gradient kernels: T = s * w is multiplied by a power of two before its (hi, lo) f16 split with UNSCALED
// residual: an f16 pair resolves 2^-24 absolute, which at M = 1e5 (T ~ 1e-5) left a truncation error of 5e-5 of the
// gradient (found by the M = 100 003 parity test, round 4).  |T| <= 1 for a pixel whose own prototype is of its class
// (every pixel of the training step): scale 2^14, resolution 2^-38.  Otherwise the reference's `sum_same - own_sim`
// (loss.py:61-70) subtracts a similarity that is not in the sum, num can be far below a single same-class similarity
// and |T| <= B = (num + own_sim) |1/den - 1/num|: the pixel's scale is 2^14 / 2^ceil(log2 B) (a per-pixel power of
// two: exact, folded into the pixel's weights and taken out again per pixel).
constexpr float kTScale = 16384.0f;

__host__ __device__ inline float nll_t_scale(float num, float den, float own_sim, bool fallback, bool own_is_same) {
  if (fallback || own_is_same) return kTScale;
  const float b = (num + own_sim) * fabsf(1.0f / den - 1.0f / num);
  if (!(b > 1.0f) || !(b < 3.0e38f)) return kTScale;     // (b = inf / nan when num or den underflowed: no finite scale fits)
  int ex = 0;
  frexpf(b, &ex);                      // b = f 2^ex, f in [0.5, 1)
  return ldexpf(kTScale, -(ex > 40 ? 40 : ex));
}

struct PixelCoef {   // 16 B, one per pixel (tile-padded)
  float wa, wb;      // weights of the same-class / other prototypes (unscaled)
  int own;
  float tscale;      // nll_t_scale of the pixel; 0 for the padding rows past P
};

struct NllArgs {
  NllDims n;
  const _Float16 *eh, *el;     // pixels, std fragments      [PT][KS][64][8]
  const _Float16 *ph, *pl;     // prototypes, std fragments  [MT][KS][64][8]
  const _Float16 *pth, *ptl;   // prototypes, T fragments    [MT][DT][2][64][8]   (bwd_de)
  const _Float16 *eth, *etl;   // pixels (x g*kappa/S), T    [PT][DT][2][64][8]   (bwd_dp)
  const int64_t* own;          // [P]
  const int64_t* px_code;      // [P]
  const int64_t* pr_code;      // [M]
  const int64_t* pr_code_pad;  // [MT*32] copy of pr_code padded with 0 (workspace)
  const int64_t* px_code_pad;  // [PT*32] (backward)
  const struct PixelCoef* coef; // [PT*32] per-pixel backward coefficients (workspace)
  int64_t mt_grad;             // prototype tiles that receive a gradient
  int depth;                   // LDS ring depth of the backward kernels
  int depth_fwd;               // ... of the forward kernel
  int dt0, dt_all;             // backward: this launch covers d-tiles [dt0, dt0 + DT) of dt_all
  float kappa_log2e, kappa;
  int mode;
  float* nll;                  // [P]
  float* stats;                // [P][4] = num, den, own_sim, fallback
  const float* d_nll;          // [P]
  const float* gscale;         // [1]
  float* d_emb;                // [P][D]
  float* d_protos;             // [M][D]
  long long* d_protos64;       // deterministic mode: [m_grad rows][D] fixed-point image of d_protos / gscale (else null)
  int chunks;                  // bwd_dp: pixel chunks per prototype tile
  // wide embeddings (several d-chunk launches): the weight tiles T = s * w of the first launch are kept
  // (one 4-KB block per (pixel tile, prototype tile), 64 B per lane) and re-read by the later launches
  // instead of recomputing the similarity GEMM + exp + predicate for every chunk
  float* tcache_de;            // [strip tiles][MT][64 lanes][16]
  float* tcache_dp;            // [MT][strip tiles][64 lanes][16]
  int64_t spt0, spt1;          // backward kernels: the pixel tiles [spt0, spt1) this launch covers (a strip of the call)
  float* partial;              // nll_fwd2: per-chunk partial sums [chunks][PT*32][4]
  int skip_de;                 // the v2 dE kernel has already run
  float* partial_de;           // nll_bwd_de2: [gridDim.y][PT][DT][16][64] accumulator-layout partial gradients
};

// d_protos[idx] += v.  Deterministic mode: v / gscale (a power of two: exact) joins a 64-bit fixed-point sum instead
// (nll_dpr_from_fix adds the converted sums to d_protos once at the end of the call).
__device__ __forceinline__ void dpr_add(const NllArgs& a, size_t idx, float v, float inv_gscale) {
  if (a.d_protos64) det_atomic_add(a.d_protos64 + idx, v * inv_gscale);
  else unsafeAtomicAdd(a.d_protos + idx, v);
}

// positive-set predicate; TAG is a template parameter of the kernels so that the
// per-(pixel, prototype) work is one compare, not both predicates and a select
template <bool TAG, typename T>
__device__ __forceinline__ bool code_match(T a, T b) {
  if constexpr (sizeof(T) == 4) return TAG ? ((unsigned)(a & b) != 0u) : ((unsigned)a == (unsigned)b);
  else return TAG ? ((a & b) != 0) : (a == b);
}
// C32 (SPML_NLL_CODE32): the caller promises that every code fits in 32 bits, the predicate
// then costs one 32-bit VALU op instead of two to four on 64-bit pairs -- the kernels are
// bound by this per-(pixel, prototype) epilogue, not by the matrix cores
template <bool C32>
using code_t = typename std::conditional<C32, int, int64_t>::type;

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// One 1-KB fragment block (64 lanes x 16 B) global -> LDS, asynchronously.
__device__ __forceinline__ void dma_block(const void* src_lane, void* dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds((gptr_t)src_lane, (lptr_t)dst_wave_uniform, 16, 0, 0);
}
__device__ __forceinline__ void dma_wait_and_sync() {
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// nll_de3.hip: the software-pipelined embedding-gradient kernel (KS <= 4, 32-bit codes); `rows` = grid
// rows (chunk c is taken by row c mod rows), partial sums into a.partial_de as nll_bwd_de2 writes them
int nll_launch_bwd_de3(const NllArgs& a, int rows, hipStream_t s);

// nll_dp3.hip: the software-pipelined prototype-gradient kernel of the same shapes (operand contract at the
// definition); rows: nll_dp3_rows_bytes(PT) bytes of workspace
size_t nll_dp3_rows_bytes(int64_t PT);
int nll_launch_bwd_dp3(const NllArgs& a, const float* own_term, const float* emb, float* rows, hipStream_t s);

}  // namespace spml
