// K4/K5: pixel-to-segment NCA negative log-likelihood, forward and backward.
//
// Replaces _calculate_log_likelihood / _one_hot_calculate_log_likelihood
// (segsort/loss.py:15-82, :85-130 of the reference), which materialise five
// [P,M] fp32 temporaries: sim = exp(kappa * E @ Pr^T), two label masks and two
// masked products.  Here the [P,M] similarity only ever exists as 32x32 MFMA
// accumulator tiles:
//
//   forward   one wave owns 32*NB pixels (B operand, fragments resident in
//             registers) and streams every 32-prototype tile (A operand, 1-KB
//             coalesced fragment blocks from L2); epilogue per tile: exp, label
//             predicate, three running per-pixel sums.  No LDS, no barriers.
//   backward  flash-style recompute of the tile; T = s * dnll/ds is split to
//             f16 (hi, lo) in registers and used directly as the B operand of
//             a second MFMA chain (the accumulator layout IS a valid B
//             fragment for a permuted contraction order; the matching A
//             fragments are laid out by the prep kernels):
//               dE^T [d][pixel] += PrT[d][m] * T[m][pixel]      (kernel bwd_de)
//               dPr^T[d][proto] += ET [d][p] * T'[p][proto]     (kernel bwd_dp)
//
// All contractions run on the f16 matrix cores at fp32 accuracy (split-f16 x2,
// see common.cuh).  Prep kernels convert fp32 rows to fragment-major (hi, lo)
// f16 arrays once per call; algorithmic flops fwd 2*P*M*D, bwd ~6*P*M*D.
#include "common.cuh"

namespace spml {
namespace {

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

// ---------------------------------------------------------------------------
// prep: fp32 rows [R,D] -> fragment-major split-f16.
//  std layout : frag[((tile*KS + ks)*64 + lane)*8 + e] = X[32*tile + (lane&31)][16*ks + 8*(lane>>5) + e]
//  T   layout : frag[(((tile*DT + dt)*2 + s)*64 + lane)*8 + e]
//                 = scale[rho] * X[rho][32*dt + (lane&31)],  rho = 32*tile + tile_row(8*s + e, lane>>5)
// out-of-range rows / channels are written as zero.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prep_std(const float* __restrict__ x, int64_t R, int D,
                                                int KS, _Float16* __restrict__ oh,
                                                _Float16* __restrict__ ol) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // fragment block id
  const int64_t nfrag = ((R + 31) / 32) * KS;
  if (f >= nfrag) return;
  const int lane = threadIdx.x & 63;
  const int64_t tile = f / KS;
  const int ks = (int)(f - tile * KS);
  const int64_t row = 32 * tile + (lane & 31);
  const int k0 = 16 * ks + 8 * (lane >> 5);
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    float v = 0.f;
    if (row < R && k0 + e < D) v = x[(size_t)row * D + k0 + e];
    _Float16 a, b;
    split_f16(v, a, b);
    h[e] = a; l[e] = b;
  }
  *reinterpret_cast<half8*>(oh + ((size_t)f * 64 + lane) * 8) = h;
  *reinterpret_cast<half8*>(ol + ((size_t)f * 64 + lane) * 8) = l;
}

__global__ __launch_bounds__(256) void prep_T(const float* __restrict__ x, int64_t R, int D,
                                              int DT, const float* __restrict__ rowscale,
                                              const float* __restrict__ gscale,
                                              _Float16* __restrict__ oh,
                                              _Float16* __restrict__ ol) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nfrag = ((R + 31) / 32) * DT * 2;
  if (f >= nfrag) return;
  const int lane = threadIdx.x & 63;
  const int s = (int)(f & 1);
  const int64_t td = f >> 1;
  const int64_t tile = td / DT;
  const int dt = (int)(td - tile * DT);
  const int d = 32 * dt + (lane & 31);
  const float gs = gscale ? 1.0f / gscale[0] : 1.0f;     // power of two: exact
  half8 h, l;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int64_t rho = 32 * tile + tile_row(8 * s + e, lane >> 5);
    float v = 0.f;
    if (rho < R && d < D) {
      v = x[(size_t)rho * D + d];
      if (rowscale) v *= rowscale[rho] * gs;
    }
    _Float16 a, b;
    split_f16(v, a, b);
    h[e] = a; l[e] = b;
  }
  *reinterpret_cast<half8*>(oh + ((size_t)f * 64 + lane) * 8) = h;
  *reinterpret_cast<half8*>(ol + ((size_t)f * 64 + lane) * 8) = l;
}

// gscale[0] = power of two >= max |g|  (so that g/gscale is in [-1,1]); 1 if all zero
__global__ __launch_bounds__(1024) void max_abs_pow2(const float* __restrict__ g, int64_t n,
                                                     float* __restrict__ out) {
  __shared__ float red[16];
  float m = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(g[i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) m = fmaxf(m, red[i]);
    float p = 1.0f;
    if (m > 0.f && m < INFINITY) {
      int ex;
      frexpf(m, &ex);            // m = f * 2^ex, f in [0.5, 1)
      p = ldexpf(1.0f, ex);
    }
    out[0] = p;
  }
}

// ---------------------------------------------------------------------------
struct NllArgs {
  NllDims n;
  const _Float16 *eh, *el;     // pixels, std fragments      [PT][KS][64][8]
  const _Float16 *ph, *pl;     // prototypes, std fragments  [MT][KS][64][8]
  const _Float16 *pth, *ptl;   // prototypes, T fragments    [MT][DT][2][64][8]   (bwd_de)
  const _Float16 *eth, *etl;   // pixels (x g*kappa/S), T    [PT][DT][2][64][8]   (bwd_dp)
  const int64_t* own;          // [P]
  const int64_t* px_code;      // [P]
  const int64_t* pr_code;      // [M]
  float kappa_log2e, kappa;
  int mode;
  float* nll;                  // [P]
  float* stats;                // [P][4] = num, den, own_sim, fallback
  const float* d_nll;          // [P]
  const float* gscale;         // [1]
  float* d_emb;                // [P][D]
  float* d_protos;             // [M][D]
  int chunks;                  // bwd_dp: pixel chunks per prototype tile
};

__device__ __forceinline__ bool code_match(int64_t a, int64_t b, int mode) {
  return (mode & SPML_NLL_TAGSET) ? ((a & b) != 0) : (a == b);
}

// z tile (rows = A rows, cols = B cols) for KS k-steps, A streamed from global
template <int KS>
__device__ __forceinline__ void zgemm(const _Float16* __restrict__ ah_g,
                                      const _Float16* __restrict__ al_g, int lane,
                                      const half8 (&bh)[KS], const half8 (&bl)[KS],
                                      float16v& zh, float16v& zx) {
#pragma unroll
  for (int r = 0; r < 16; ++r) { zh[r] = 0.f; zx[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const half8 a_h = *reinterpret_cast<const half8*>(ah_g + ((size_t)ks * 64 + lane) * 8);
    const half8 a_l = *reinterpret_cast<const half8*>(al_g + ((size_t)ks * 64 + lane) * 8);
    zh = mfma32(a_h, bh[ks], zh);
    zx = mfma32(a_h, bl[ks], zx);
    zx = mfma32(a_l, bh[ks], zx);
  }
}

// ------------------------------- forward -----------------------------------
template <int KS, int NB>
__global__ __launch_bounds__(256) void nll_fwd(NllArgs a) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, j = lane & 31;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t pt0 = wave * NB;
  if (pt0 >= a.n.PT) return;

  half8 bh[NB][KS], bl[NB][KS];
  int64_t pcode[NB];
  int own[NB];
  float s_same[NB], s_diff[NB], s_own[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int64_t pt = min(pt0 + nb, a.n.PT - 1);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bh[nb][ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
      bl[nb][ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    }
    const int64_t p = min(32 * pt + j, a.n.P - 1);
    pcode[nb] = a.px_code[p];
    own[nb] = (int)a.own[p];
    s_same[nb] = 0.f; s_diff[nb] = 0.f; s_own[nb] = 0.f;
  }

  for (int64_t mt = 0; mt < a.n.MT; ++mt) {
    int64_t rc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = min(32 * mt + tile_row(r, half), a.n.M - 1);
      rc[r] = a.pr_code[row];
    }
    const _Float16* ahg = a.ph + (size_t)mt * KS * 512;
    const _Float16* alg = a.pl + (size_t)mt * KS * 512;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      float16v zh, zx;
      zgemm<KS>(ahg, alg, lane, bh[nb], bl[nb], zh, zx);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (int)(32 * mt) + tile_row(r, half);
        const float z = zh[r] + zx[r] * kSplitInv;
        float s = __builtin_amdgcn_exp2f(z * a.kappa_log2e);
        s = row < a.n.M ? s : 0.f;
        const bool same = code_match(pcode[nb], rc[r], a.mode);
        s_same[nb] += same ? s : 0.f;
        s_diff[nb] += same ? 0.f : s;
        s_own[nb] += (row == own[nb]) ? s : 0.f;
      }
    }
  }
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    // the two lane halves saw different prototype rows of the same pixel
    const float same = s_same[nb] + __shfl_xor(s_same[nb], 32, 64);
    const float diff = s_diff[nb] + __shfl_xor(s_diff[nb], 32, 64);
    const float osim = s_own[nb] + __shfl_xor(s_own[nb], 32, 64);
    const int64_t p = 32 * (pt0 + nb) + j;
    if (half == 0 && pt0 + nb < a.n.PT && p < a.n.P) {
      // loss.py:61-80: pos = sum_same - own (that order); fallback to own if pos <= 0
      const float pos = same - osim;
      const bool fb = (a.mode & SPML_NLL_PLAIN) || !(pos > 0.f);
      const float num = fb ? osim : pos;
      const float den = diff + num;
      a.nll[p] = -logf(num / den);
      float4v st = {num, den, osim, fb ? 1.f : 0.f};
      *reinterpret_cast<float4v*>(a.stats + (size_t)p * 4) = st;
    }
  }
}

// T = s * dnll/ds  (without the g * kappa factor):  w = dnum*(1/den - 1/num) + diff/den
__device__ __forceinline__ float t_value(float s, bool same, bool is_own, float inv_num,
                                         float inv_den, bool fb) {
  const float dnum = fb ? (is_own ? 1.f : 0.f) : ((same ? 1.f : 0.f) - (is_own ? 1.f : 0.f));
  const float w = dnum * (inv_den - inv_num) + (same ? 0.f : inv_den);
  return s * w;
}

// ------------------------------- backward: dE ------------------------------
template <int KS, int DT>
__global__ __launch_bounds__(256) void nll_bwd_de(NllArgs a) {
  const int lane = threadIdx.x & 63;
  const int half = lane >> 5, j = lane & 31;
  const int64_t pt = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (pt >= a.n.PT) return;

  half8 bh[KS], bl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bh[ks] = *reinterpret_cast<const half8*>(a.eh + (((size_t)pt * KS + ks) * 64 + lane) * 8);
    bl[ks] = *reinterpret_cast<const half8*>(a.el + (((size_t)pt * KS + ks) * 64 + lane) * 8);
  }
  const int64_t p = min(32 * pt + j, a.n.P - 1);
  const int64_t pcode = a.px_code[p];
  const int own = (int)a.own[p];
  const float4v st = *reinterpret_cast<const float4v*>(a.stats + (size_t)p * 4);
  const float inv_num = 1.0f / st[0], inv_den = 1.0f / st[1];
  const bool fb = st[3] != 0.f;

  float16v dacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[dt][r] = 0.f;

  for (int64_t mt = 0; mt < a.n.MT; ++mt) {
    float16v zh, zx;
    zgemm<KS>(a.ph + (size_t)mt * KS * 512, a.pl + (size_t)mt * KS * 512, lane, bh, bl, zh, zx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (int)(32 * mt) + tile_row(r, half);
      const int64_t rcode = a.pr_code[min((int64_t)row, a.n.M - 1)];
      const float z = zh[r] + zx[r] * kSplitInv;
      float s = __builtin_amdgcn_exp2f(z * a.kappa_log2e);
      s = row < a.n.M ? s : 0.f;
      t[r] = t_value(s, code_match(pcode, rcode, a.mode), row == own, inv_num, inv_den, fb);
    }
    // registers [8s, 8s+8) of this lane are the B fragment of k-step s
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = t[8 * s2 + e];
      half8 th, tl;
      split8(v, th, tl);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const size_t o = ((((size_t)mt * DT + dt) * 2 + s2) * 64 + lane) * 8;
        const half8 a_h = *reinterpret_cast<const half8*>(a.pth + o);
        const half8 a_l = *reinterpret_cast<const half8*>(a.ptl + o);
        // lo terms carry an exact 2^-11: fold it in by accumulating them first
        float16v lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) lo[r] = 0.f;
        lo = mfma32(a_h, tl, lo);
        lo = mfma32(a_l, th, lo);
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[dt][r] += lo[r] * kSplitInv;
        dacc[dt] = mfma32(a_h, th, dacc[dt]);
      }
    }
  }
  // dE[p][d] = g_p * kappa * acc[d][p]
  const int64_t pp = 32 * pt + j;
  if (pp < a.n.P) {
    const float gk = a.d_nll[pp] * a.kappa;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * dt + tile_row(r, half);
        if (d < a.n.D) a.d_emb[(size_t)pp * a.n.D + d] = gk * dacc[dt][r];
      }
  }
}

// ------------------------------- backward: dPr -----------------------------
// grid (MT, chunks): one workgroup per (prototype tile, pixel chunk); its 4
// waves stride over the chunk's pixel tiles; partial dPr^T tiles meet in LDS
// and leave with one fp32 atomic per element.
template <int KS, int DT>
__global__ __launch_bounds__(256) void nll_bwd_dp(NllArgs a) {
  __shared__ float red[DT][16][64];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int half = lane >> 5, j = lane & 31;
  const int64_t mt = blockIdx.x;
  const int64_t per = (a.n.PT + a.chunks - 1) / a.chunks;
  const int64_t pt_lo = (int64_t)blockIdx.y * per;
  const int64_t pt_hi = min(a.n.PT, pt_lo + per);

  // prototypes of this tile: B operand (cols), resident
  half8 bh[KS], bl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bh[ks] = *reinterpret_cast<const half8*>(a.ph + (((size_t)mt * KS + ks) * 64 + lane) * 8);
    bl[ks] = *reinterpret_cast<const half8*>(a.pl + (((size_t)mt * KS + ks) * 64 + lane) * 8);
  }
  const int col = (int)(32 * mt) + j;                 // prototype of this lane's column
  const bool col_ok = col < a.n.M;
  const int64_t ccode = a.pr_code[min((int64_t)col, a.n.M - 1)];

  float16v dacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) dacc[dt][r] = 0.f;

  for (int64_t pt = pt_lo + wave; pt < pt_hi; pt += 4) {
    // z'[row = pixel][col = prototype]
    float16v zh, zx;
    zgemm<KS>(a.eh + (size_t)pt * KS * 512, a.el + (size_t)pt * KS * 512, lane, bh, bl, zh, zx);
    float t[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t p = 32 * pt + tile_row(r, half);
      const int64_t pc = min(p, a.n.P - 1);
      const float4v st = *reinterpret_cast<const float4v*>(a.stats + (size_t)pc * 4);
      const int64_t pcode = a.px_code[pc];
      const int own = (int)a.own[pc];
      const float z = zh[r] + zx[r] * kSplitInv;
      float s = __builtin_amdgcn_exp2f(z * a.kappa_log2e);
      s = (col_ok && p < a.n.P) ? s : 0.f;
      t[r] = t_value(s, code_match(pcode, ccode, a.mode), col == own, 1.0f / st[0],
                     1.0f / st[1], st[3] != 0.f);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = t[8 * s2 + e];
      half8 th, tl;
      split8(v, th, tl);
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const size_t o = ((((size_t)pt * DT + dt) * 2 + s2) * 64 + lane) * 8;
        const half8 a_h = *reinterpret_cast<const half8*>(a.eth + o);
        const half8 a_l = *reinterpret_cast<const half8*>(a.etl + o);
        float16v lo;
#pragma unroll
        for (int r = 0; r < 16; ++r) lo[r] = 0.f;
        lo = mfma32(a_h, tl, lo);
        lo = mfma32(a_l, th, lo);
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[dt][r] += lo[r] * kSplitInv;
        dacc[dt] = mfma32(a_h, th, dacc[dt]);
      }
    }
  }
  // waves 1..3 hand their partial tiles to wave 0, one at a time (fixed order)
  for (int w = 1; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) red[dt][r][lane] = dacc[dt][r];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) dacc[dt][r] += red[dt][r][lane];
    }
    __syncthreads();
  }
  if (wave == 0 && col_ok) {
    const float gs = a.gscale[0];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int d = 32 * dt + tile_row(r, half);
        if (d < a.n.D) unsafeAtomicAdd(a.d_protos + (size_t)col * a.n.D + d, dacc[dt][r] * gs);
      }
  }
}

__global__ void rowscale_kernel(const float* g, float kappa, int64_t n, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = g[i] * kappa;
}

// ------------------------------- host --------------------------------------
struct NllWs {
  size_t eh, el, ph, pl, pth, ptl, eth, etl, gscale, rowscale, total;
};

NllWs nll_ws(const NllDims& n) {
  NllWs w{};
  size_t o = 0;
  const size_t e_std = (size_t)n.PT * n.KS * 512 * 2, p_std = (size_t)n.MT * n.KS * 512 * 2;
  const size_t e_t = (size_t)n.PT * n.DT * 2 * 512 * 2, p_t = (size_t)n.MT * n.DT * 2 * 512 * 2;
  w.eh = o; o = align_up(o + e_std, 256);
  w.el = o; o = align_up(o + e_std, 256);
  w.ph = o; o = align_up(o + p_std, 256);
  w.pl = o; o = align_up(o + p_std, 256);
  w.pth = o; o = align_up(o + p_t, 256);
  w.ptl = o; o = align_up(o + p_t, 256);
  w.eth = o; o = align_up(o + e_t, 256);
  w.etl = o; o = align_up(o + e_t, 256);
  w.gscale = o; o = align_up(o + 16, 256);
  w.rowscale = o; o = align_up(o + (size_t)n.P * 4, 256);
  w.total = o;
  return w;
}

int ks_bucket(int ks) {
  if (ks <= 2) return 2;
  if (ks <= 3) return 3;
  if (ks <= 5) return 5;
  if (ks <= 9) return 9;
  if (ks <= 17) return 17;
  return 0;
}

void launch_prep_std(const float* x, int64_t R, int D, int KS, _Float16* h, _Float16* l,
                     hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * KS;
  hipLaunchKernelGGL(prep_std, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, KS, h, l);
}
void launch_prep_T(const float* x, int64_t R, int D, int DT, const float* rowscale,
                   const float* gscale, _Float16* h, _Float16* l, hipStream_t s) {
  const int64_t nfrag = ((R + 31) / 32) * DT * 2;
  hipLaunchKernelGGL(prep_T, dim3((unsigned)((nfrag + 3) / 4)), dim3(256), 0, s, x, R, D, DT,
                     rowscale, gscale, h, l);
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" size_t spml_segsort_nll_workspace_bytes(int64_t P, int64_t M, int D) {
  if (P < 0 || M <= 0 || D <= 0) return 0;
  NllDims n = nll_dims(P, M, D);
  if (ks_bucket(n.KS)) { n.KS = ks_bucket(n.KS); n.DT = (n.KS + 1) / 2; }
  return nll_ws(n).total;
}

static int nll_common(bool backward, const float* emb, const int64_t* own,
                      const int64_t* px_code, int64_t P, const float* protos,
                      const int64_t* pr_code, int64_t M, int D, float kappa, int mode,
                      float* nll, float* stats, const float* d_nll, float* d_emb,
                      float* d_protos, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!emb || !own || !px_code || !protos || !pr_code || P < 0 || M <= 0 || D <= 0 || !stats)
    return SPML_ERR_INVALID_ARG;
  if (mode < 0 || mode > (SPML_NLL_TAGSET | SPML_NLL_PLAIN)) return SPML_ERR_INVALID_ARG;
  if (backward ? (!d_nll || !d_emb || !d_protos) : !nll) return SPML_ERR_INVALID_ARG;
  NllDims n = nll_dims(P, M, D);
  const int ksb = ks_bucket(n.KS);
  if (!ksb) return SPML_ERR_UNSUPPORTED;                    // D <= 272
  n.KS = ksb;
  n.DT = (ksb + 1) / 2;                                     // template tile counts
  const NllWs w = nll_ws(n);
  if (!ws || ws_bytes < w.total) return SPML_ERR_WORKSPACE;
  if (P == 0) return SPML_OK;
  unsigned char* b = static_cast<unsigned char*>(ws);
  NllArgs a{};
  a.n = n;
  _Float16* eh = reinterpret_cast<_Float16*>(b + w.eh);
  _Float16* el = reinterpret_cast<_Float16*>(b + w.el);
  _Float16* ph = reinterpret_cast<_Float16*>(b + w.ph);
  _Float16* pl = reinterpret_cast<_Float16*>(b + w.pl);
  a.eh = eh; a.el = el; a.ph = ph; a.pl = pl;
  a.own = own; a.px_code = px_code; a.pr_code = pr_code;
  a.kappa = kappa;
  a.kappa_log2e = kappa * 1.4426950408889634f;
  a.mode = mode; a.nll = nll; a.stats = stats; a.d_nll = d_nll; a.d_emb = d_emb;
  a.d_protos = d_protos;
  launch_prep_std(emb, P, D, n.KS, eh, el, s);
  launch_prep_std(protos, M, D, n.KS, ph, pl, s);

#define SPML_KS_SWITCH(MACRO)             \
  switch (n.KS) {                         \
    case 2: MACRO(2); break;              \
    case 3: MACRO(3); break;              \
    case 5: MACRO(5); break;              \
    case 9: MACRO(9); break;              \
    case 17: MACRO(17); break;            \
    default: return SPML_ERR_UNSUPPORTED; \
  }
  if (!backward) {
#define SPML_FWD(KS_)                                                                      \
  {                                                                                        \
    constexpr int NB = KS_ <= 9 ? 2 : 1;                                                   \
    const int64_t waves = (n.PT + NB - 1) / NB;                                            \
    hipLaunchKernelGGL((nll_fwd<KS_, NB>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, a); \
  }
    SPML_KS_SWITCH(SPML_FWD)
#undef SPML_FWD
    return launch_status();
  }

  // ---- backward ----
  _Float16* pth = reinterpret_cast<_Float16*>(b + w.pth);
  _Float16* ptl = reinterpret_cast<_Float16*>(b + w.ptl);
  _Float16* eth = reinterpret_cast<_Float16*>(b + w.eth);
  _Float16* etl = reinterpret_cast<_Float16*>(b + w.etl);
  float* gscale = reinterpret_cast<float*>(b + w.gscale);
  float* rowscale = reinterpret_cast<float*>(b + w.rowscale);
  a.pth = pth; a.ptl = ptl; a.eth = eth; a.etl = etl; a.gscale = gscale;
  hipLaunchKernelGGL(max_abs_pow2, dim3(1), dim3(1024), 0, s, d_nll, P, gscale);
  hipLaunchKernelGGL(rowscale_kernel, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, s, d_nll,
                     kappa, P, rowscale);
  // gscale bounds |g|; kappa is folded into rowscale (kappa * g / gscale stays O(kappa))
  launch_prep_T(protos, M, D, n.DT, nullptr, nullptr, pth, ptl, s);
  launch_prep_T(emb, P, D, n.DT, rowscale, gscale, eth, etl, s);
  // pixel chunks so that the grid has a few thousand workgroups
  int64_t chunks = (4096 + n.MT - 1) / n.MT;
  if (chunks > (n.PT + 3) / 4) chunks = (n.PT + 3) / 4;
  if (chunks < 1) chunks = 1;
  if (chunks > 65535) chunks = 65535;
  a.chunks = (int)chunks;

#define SPML_BWD_DT(KS_, DT_)                                                                    \
  {                                                                                              \
    hipLaunchKernelGGL((nll_bwd_de<KS_, DT_>), dim3((unsigned)((n.PT + 3) / 4)), dim3(256), 0, s, a); \
    hipLaunchKernelGGL((nll_bwd_dp<KS_, DT_>), dim3((unsigned)n.MT, (unsigned)chunks), dim3(256), \
                       0, s, a);                                                                 \
  }
#define SPML_BWD(KS_)                                          \
  {                                                            \
    constexpr int DTM = (KS_ + 1) / 2;                         \
    SPML_BWD_DT(KS_, DTM)                                      \
  }
  SPML_KS_SWITCH(SPML_BWD)
#undef SPML_BWD
#undef SPML_BWD_DT
#undef SPML_KS_SWITCH
  return launch_status();
}

extern "C" int spml_segsort_nll_fwd_f32(const float* emb, const int64_t* own,
                                        const int64_t* px_code, int64_t P, const float* protos,
                                        const int64_t* pr_code, int64_t M, int D, float kappa,
                                        int mode, float* nll, float* stats, void* ws,
                                        size_t ws_bytes, void* stream) {
  return nll_common(false, emb, own, px_code, P, protos, pr_code, M, D, kappa, mode, nll, stats,
                    nullptr, nullptr, nullptr, ws, ws_bytes, (hipStream_t)stream);
}

extern "C" int spml_segsort_nll_bwd_f32(const float* emb, const int64_t* own,
                                        const int64_t* px_code, int64_t P, const float* protos,
                                        const int64_t* pr_code, int64_t M, int D, float kappa,
                                        int mode, const float* stats, const float* d_nll,
                                        float* d_emb, float* d_protos, void* ws, size_t ws_bytes,
                                        void* stream) {
  return nll_common(true, emb, own, px_code, P, protos, pr_code, M, D, kappa, mode, nullptr,
                    const_cast<float*>(stats), d_nll, d_emb, d_protos, ws, ws_bytes,
                    (hipStream_t)stream);
}
