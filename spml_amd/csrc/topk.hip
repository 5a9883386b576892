// K6/K7: top-k retrieval by cosine affinity (k <= 32).
//
// Replaces eval.py:32-35 of the reference (mm -> full argsort of the [Q,M]
// affinity -> first k columns) and models/utils.py:198-214 (mm -> where(mask)
// -> topk).  A workgroup owns 32 queries (B operand, fragments in registers); its
// WAVES waves each stream a contiguous share of the 32-prototype tiles through the f16
// matrix cores (split-f16 x2, fp32 accuracy); each lane keeps a sorted top-k list of the
// 16 prototype rows it sees per tile, and the 2*WAVES lists of a query (two lane halves
// per wave) are merged once at the end.  Ties resolve to the lowest prototype index.
#include "common.hpp"

namespace spml {
namespace {

__device__ __forceinline__ int tile_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(256) void topk_prep(const float* __restrict__ x, int64_t R, int D,
                                                 int KS, _Float16* __restrict__ oh,
                                                 _Float16* __restrict__ ol) {
  const int64_t f = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
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

struct TopkArgs {
  const _Float16 *qh, *ql, *ph, *pl;
  int64_t Q, M, QT, MT;
  int k;
  const int64_t* q_group;
  const int64_t* pr_group;
  const uint8_t* pr_valid;
  float masked_value;
  int64_t* idx;
  float* val;
};

template <int KS, int KMAX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void topk_kernel(TopkArgs a) {
  __shared__ float sv[WAVES * 64][KMAX + 1];
  __shared__ int si[WAVES * 64][KMAX + 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, j = lane & 31;
  const int64_t qt = blockIdx.x;
  half8 bh[KS], bl[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    bh[ks] = *reinterpret_cast<const half8*>(a.qh + (((size_t)qt * KS + ks) * 64 + lane) * 8);
    bl[ks] = *reinterpret_cast<const half8*>(a.ql + (((size_t)qt * KS + ks) * 64 + lane) * 8);
  }
  const int64_t q = min(32 * qt + j, a.Q - 1);
  const int64_t qg = a.q_group ? a.q_group[q] : 0;

  float tv[KMAX];
  int ti[KMAX];
#pragma unroll
  for (int i = 0; i < KMAX; ++i) { tv[i] = -INFINITY; ti[i] = 0x7fffffff; }

  const int64_t mt_lo = (a.MT * wave) / WAVES, mt_hi = (a.MT * (wave + 1)) / WAVES;
  for (int64_t mt = mt_lo; mt < mt_hi; ++mt) {
    float16v zh, zx;
#pragma unroll
    for (int r = 0; r < 16; ++r) { zh[r] = 0.f; zx[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const size_t o = (((size_t)mt * KS + ks) * 64 + lane) * 8;
      const half8 a_h = *reinterpret_cast<const half8*>(a.ph + o);
      const half8 a_l = *reinterpret_cast<const half8*>(a.pl + o);
      zh = mfma32(a_h, bh[ks], zh);
      zx = mfma32(a_h, bl[ks], zx);
      zx = mfma32(a_l, bh[ks], zx);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (int)(32 * mt) + tile_row(r, half);
      float z = zh[r] + zx[r] * kSplitInv;
      bool ok = row < a.M;
      if (a.q_group && ok) {
        const bool allowed = a.pr_group[row] == qg && (!a.pr_valid || a.pr_valid[row] != 0);
        z = allowed ? z : a.masked_value;
      }
      const bool cand = ok && (z > tv[KMAX - 1]);
      if (__any(cand)) {                       // wave-uniform early out
        // insert (z,row) keeping the list sorted (desc value; earlier rows first on ties)
        float cv = cand ? z : -INFINITY;
        int ci = cand ? row : 0x7fffffff;
#pragma unroll
        for (int i = 0; i < KMAX; ++i) {
          const bool sw = cv > tv[i];
          const float ov = tv[i];
          const int oi = ti[i];
          tv[i] = sw ? cv : ov;
          ti[i] = sw ? ci : oi;
          cv = sw ? ov : cv;
          ci = sw ? oi : ci;
        }
      }
    }
  }
#pragma unroll
  for (int i = 0; i < KMAX; ++i) { sv[threadIdx.x][i] = tv[i]; si[threadIdx.x][i] = ti[i]; }
  __syncthreads();
  if (threadIdx.x < 32 && 32 * qt + j < a.Q) {
    // merge the 2*WAVES sorted lists of this query: (value desc, index asc)
    int pos[2 * WAVES];
#pragma unroll
    for (int l = 0; l < 2 * WAVES; ++l) pos[l] = 0;
    for (int o = 0; o < a.k; ++o) {
      float bv = -INFINITY;
      int bx = 0x7fffffff, bl = 0;
#pragma unroll
      for (int l = 0; l < 2 * WAVES; ++l) {
        const int row = (l >> 1) * 64 + (l & 1) * 32 + j;
        const float v = sv[row][pos[l]];
        const int x = si[row][pos[l]];
        if (v > bv || (v == bv && x < bx)) { bv = v; bx = x; bl = l; }
      }
#pragma unroll
      for (int l = 0; l < 2 * WAVES; ++l) pos[l] += (l == bl) ? 1 : 0;
      if (bx == 0x7fffffff) bx = 0;            // fewer than k prototypes exist
      a.idx[(size_t)(32 * qt + j) * a.k + o] = bx;
      a.val[(size_t)(32 * qt + j) * a.k + o] = bv;
    }
  }
}

int ks_bucket(int ks) {
  if (ks <= 2) return 2;
  if (ks <= 3) return 3;
  if (ks <= 4) return 4;
  if (ks <= 5) return 5;
  if (ks <= 9) return 9;
  if (ks <= 17) return 17;
  if (ks <= 33) return 33;
  return 0;
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" size_t spml_topk_workspace_bytes(int64_t Q, int64_t M, int D, int k) {
  if (Q < 0 || M <= 0 || D <= 0 || k <= 0) return 0;
  const int ks = ks_bucket((D + 15) / 16);
  if (!ks) return 0;
  const size_t qf = (size_t)((Q + 31) / 32) * ks * 1024, pf = (size_t)((M + 31) / 32) * ks * 1024;
  return 2 * align_up(qf, 256) + 2 * align_up(pf, 256) + 256;
}

extern "C" int spml_topk_affinity_f32(const float* q, int64_t Q, const float* protos, int64_t M,
                                      int D, int k, const int64_t* q_group,
                                      const int64_t* pr_group, const uint8_t* pr_valid,
                                      float masked_value, int64_t* idx, float* val, void* ws,
                                      size_t ws_bytes, void* stream) {
  if (!q || !protos || !idx || !val || Q < 0 || M <= 0 || D <= 0 || k <= 0)
    return SPML_ERR_INVALID_ARG;
  if ((q_group == nullptr) != (pr_group == nullptr)) return SPML_ERR_INVALID_ARG;
  if (k > 32) return SPML_ERR_UNSUPPORTED;
  const int ks = ks_bucket((D + 15) / 16);
  if (!ks) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_topk_workspace_bytes(Q, M, D, k)) return SPML_ERR_WORKSPACE;
  if (Q == 0) return SPML_OK;
  hipStream_t s = (hipStream_t)stream;
  const int64_t QT = (Q + 31) / 32, MT = (M + 31) / 32;
  const size_t qf = align_up((size_t)QT * ks * 1024, 256), pf = align_up((size_t)MT * ks * 1024, 256);
  unsigned char* b = static_cast<unsigned char*>(ws);
  _Float16* qh = reinterpret_cast<_Float16*>(b);
  _Float16* ql = reinterpret_cast<_Float16*>(b + qf);
  _Float16* ph = reinterpret_cast<_Float16*>(b + 2 * qf);
  _Float16* pl = reinterpret_cast<_Float16*>(b + 2 * qf + pf);
  hipLaunchKernelGGL(topk_prep, dim3((unsigned)((QT * ks + 3) / 4)), dim3(256), 0, s, q, Q, D, ks,
                     qh, ql);
  hipLaunchKernelGGL(topk_prep, dim3((unsigned)((MT * ks + 3) / 4)), dim3(256), 0, s, protos, M, D,
                     ks, ph, pl);
  TopkArgs a{};
  a.qh = qh; a.ql = ql; a.ph = ph; a.pl = pl; a.Q = Q; a.M = M; a.QT = QT; a.MT = MT; a.k = k;
  a.q_group = q_group; a.pr_group = pr_group; a.pr_valid = pr_valid;
  a.masked_value = masked_value; a.idx = idx; a.val = val;
#define SPML_TK(KS_, KM_, W_) \
  hipLaunchKernelGGL((topk_kernel<KS_, KM_, W_>), dim3((unsigned)QT), dim3(64 * W_), 0, s, a)
#define SPML_TK_KS(KM_, W_)                              \
  switch (ks) {                                          \
    case 2: SPML_TK(2, KM_, W_); break;                  \
    case 3: SPML_TK(3, KM_, W_); break;                  \
    case 4: SPML_TK(4, KM_, W_); break;                  \
    case 5: SPML_TK(5, KM_, W_); break;                  \
    case 9: SPML_TK(9, KM_, W_); break;                  \
    case 17: SPML_TK(17, KM_, W_); break;                \
    case 33: SPML_TK(33, KM_, W_); break;                \
    default: return SPML_ERR_UNSUPPORTED;                \
  }
  // 4 waves share a query tile (more waves in flight, 1/4 of the prototype stream each);
  // the k <= 32 lists are 4x larger: 2 waves keep the merge table inside 64 KB of LDS
  if (k <= 8) { SPML_TK_KS(8, 4) } else { SPML_TK_KS(32, 2) }
#undef SPML_TK_KS
#undef SPML_TK
  return launch_status();
}
