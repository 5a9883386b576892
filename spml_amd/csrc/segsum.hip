// A4: segment prototypes = scatter-sum of pixel rows by segment id, then L2
// normalise (segsort/common.py:11-41 of the reference: zeros + scatter_add_ with
// a [P,D]-expanded index + normalize_embedding).
//
// Pixels arrive image-major / row-major, so the segment id is constant over
// long runs.  Every wave walks a contiguous chunk of pixels with its lanes over
// the D columns (coalesced 256-B row reads), keeps the running sum of the
// current run in registers and only touches memory when the id changes: one
// fp32 atomic per (run, column) instead of one per (pixel, column).
// Algorithmic HBM bytes: P*D*4 (X once) + P*8 (ids) + 2*M*D*4.
#include "common.hpp"

#include <type_traits>

namespace spml {

namespace {

constexpr int kChunk = 32;    // pixels per wave (short chunks: the per-image calls have few pixels)

// FIX: deterministic mode -- every element is converted to 64-bit fixed point (2^36 steps per unit: exact for
// |x| >= 2^-12, truncated below 2^-36 otherwise) and the runs leave with integer atomics into `sums64`; integer adds
// commute, so the sums do not depend on the order in which the waves arrive (segsum_from_fix converts them back).
template <int NC, bool FIX>   // NC = ceil(D / 64) columns per lane
__global__ __launch_bounds__(256) void segsum_kernel(const float* __restrict__ x,
                                                     const int64_t* __restrict__ ids,
                                                     int64_t P, int D, int64_t M,
                                                     int64_t id_offset_stride,
                                                     float* __restrict__ sums,
                                                     long long* __restrict__ sums64) {
  typedef typename std::conditional<FIX, long long, float>::type acc_t;
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t p0 = wave * kChunk;
  if (p0 >= P) return;
  const int np = (int)min((int64_t)kChunk, P - p0);

  acc_t run[NC];
#pragma unroll
  for (int j = 0; j < NC; ++j) run[j] = 0;
  int64_t cur = -1;
  auto flush = [&](int64_t id) {
    if (id >= 0 && id < M) {
#pragma unroll
      for (int j = 0; j < NC; ++j) {
        const int d = lane + 64 * j;
        if (d < D) {
          if constexpr (FIX) {
            if (run[j] != 0) atomicAdd(reinterpret_cast<unsigned long long*>(sums64) + (size_t)id * D + d, (unsigned long long)run[j]);
          } else {
            atomicAdd(&sums[(size_t)id * D + d], run[j]);
          }
        }
      }
    }
  };

  for (int b = 0; b < np; b += 64) {
    const int nb = min(64, np - b);
    // 64 ids at a time, one per lane (two 32-bit halves for readlane)
    int64_t myid = (lane < nb) ? ids[p0 + b + lane] : -1;
    int id_lo = (int)(myid & 0xffffffff), id_hi = (int)(myid >> 32);
    // rows are fetched kAhead at a time, independent of the run logic: a wave walks its
    // pixels in order, so without this every row load would sit on the critical path
    constexpr int kAhead = 8;
    for (int i0 = 0; i0 < nb; i0 += kAhead) {
      float rows[kAhead][NC];
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const float* xr = x + (size_t)(p0 + b + min(i0 + u, nb - 1)) * D;
#pragma unroll
        for (int j = 0; j < NC; ++j) {
          const int d = lane + 64 * j;
          rows[u][j] = d < D ? xr[d] : 0.f;
        }
      }
#pragma unroll
      for (int u = 0; u < kAhead; ++u) {
        const int i = i0 + u;
        if (i < nb) {                                     // wave-uniform
          const int64_t id = ((int64_t)__builtin_amdgcn_readlane(id_hi, i) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane(id_lo, i);
          if (id != cur) {
            flush(cur);
#pragma unroll
            for (int j = 0; j < NC; ++j) run[j] = 0;
            cur = id;
          }
#pragma unroll
          for (int j = 0; j < NC; ++j) {
            if constexpr (FIX) run[j] += det_to_fix(rows[u][j]);
            else run[j] += rows[u][j];
          }
        }
      }
    }
  }
  flush(cur);
}

// deterministic mode: fixed-point sums -> fp32 (one rounding per element)
__global__ void segsum_from_fix(const long long* __restrict__ acc, int64_t n, float* __restrict__ sums) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) sums[i] = (float)((double)acc[i] * kDetFixInv);
}

// d_sums[m] = (dP_m - p_m <p_m, dP_m>) / |s_m|   with p_m = s_m/|s_m|
//           = dP_m / eps                          where |s_m| < eps
__global__ __launch_bounds__(256) void proto_bwd_rows(const float* __restrict__ dprotos,
                                                      const float* __restrict__ sums,
                                                      int64_t M, int D,
                                                      float* __restrict__ dsums) {
  const int64_t m = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= M) return;
  const int lane = threadIdx.x & 63;
  const float* s = sums + (size_t)m * D;
  const float* g = dprotos + (size_t)m * D;
  float ss = 0.f, sg = 0.f;
  for (int d = lane; d < D; d += 64) {
    ss += s[d] * s[d];
    sg += s[d] * g[d];
  }
  ss = wave_sum(ss);
  sg = wave_sum(sg);
  const float n = sqrtf(ss);
  if (n >= kEps) {
    const float t = sg / n;
    for (int d = lane; d < D; d += 64) dsums[(size_t)m * D + d] = (g[d] - (s[d] / n) * t) / n;
  } else {
    for (int d = lane; d < D; d += 64) dsums[(size_t)m * D + d] = g[d] / kEps;
  }
}

template <bool ACC>
__global__ __launch_bounds__(256) void gather_rows(const float* __restrict__ rows,
                                                   const int64_t* __restrict__ ids,
                                                   int64_t P, int D, int64_t M,
                                                   float* __restrict__ out) {
  const int64_t total = P * (int64_t)D;
  for (int64_t f = (int64_t)blockIdx.x * 256 + threadIdx.x; f < total;
       f += (int64_t)gridDim.x * 256) {
    const int64_t p = f / D;
    const int d = (int)(f - p * D);
    const int64_t id = ids[p];
    const float v = (id >= 0 && id < M) ? rows[(size_t)id * D + d] : 0.f;
    if (ACC) out[f] += v; else out[f] = v;
  }
}

}  // namespace

// internal: sums[ids[p]] += x[p]   (sums must be zeroed by the caller); sums64 != null: the deterministic form --
// fixed-point sums into the (zeroed) [M][D] 64-bit scratch, `sums` is not touched
int segment_sum_launch(const float* x, const int64_t* ids, int64_t P, int D, int64_t M,
                       float* sums, hipStream_t s, long long* sums64) {
  if (P == 0) return SPML_OK;
  const int64_t waves = (P + kChunk - 1) / kChunk;
  const dim3 grid((unsigned)((waves + 3) / 4));
  const int nc = (D + 63) / 64;
#define SPML_SEGSUM(NC)                                                                                              \
  do {                                                                                                               \
    if (sums64) hipLaunchKernelGGL((segsum_kernel<NC, true>), grid, dim3(256), 0, s, x, ids, P, D, M, (int64_t)0, sums, sums64); \
    else hipLaunchKernelGGL((segsum_kernel<NC, false>), grid, dim3(256), 0, s, x, ids, P, D, M, (int64_t)0, sums, sums64);       \
  } while (0)
  if (nc <= 1) SPML_SEGSUM(1);
  else if (nc <= 2) SPML_SEGSUM(2);
  else if (nc <= 3) SPML_SEGSUM(3);
  else if (nc <= 5) SPML_SEGSUM(5);
  else if (nc <= 9) SPML_SEGSUM(9);
  else if (nc <= 17) SPML_SEGSUM(17);
  else return SPML_ERR_UNSUPPORTED;   // D > 1088
#undef SPML_SEGSUM
  return launch_status();
}

}  // namespace spml

using namespace spml;

extern "C" size_t spml_segment_sum_det_workspace_bytes(int64_t M, int D) {
  return M > 0 && D > 0 ? (size_t)M * D * 8 : 0;
}

extern "C" int spml_segment_sum_normalize_det_f32(const float* x, const int64_t* ids, int64_t P, int D, int64_t M,
                                                  float* sums, float* protos, void* ws, size_t ws_bytes,
                                                  void* stream) {
  if (!x || !ids || !sums || !protos || P < 0 || D <= 0 || M <= 0) return SPML_ERR_INVALID_ARG;
  if (!ws || ws_bytes < (size_t)M * D * 8) return SPML_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  long long* acc = static_cast<long long*>(ws);
  if (hipMemsetAsync(acc, 0, (size_t)M * D * 8, s) != hipSuccess) return SPML_ERR_LAUNCH;
  int rc = segment_sum_launch(x, ids, P, D, M, sums, s, acc);
  if (rc != SPML_OK) return rc;
  hipLaunchKernelGGL(segsum_from_fix, dim3((unsigned)(((int64_t)M * D + 255) / 256)), dim3(256), 0, s, acc,
                     (int64_t)M * D, sums);
  return spml_normalize_rows_f32(sums, M, D, protos, stream);
}

extern "C" int spml_segment_sum_normalize_f32(const float* x, const int64_t* ids, int64_t P,
                                              int D, int64_t M, float* sums, float* protos,
                                              void* stream) {
  if (!x || !ids || !sums || !protos || P < 0 || D <= 0 || M <= 0) return SPML_ERR_INVALID_ARG;
  // deterministic mode: this entry point accumulates with fp32 atomics -- refused, loudly, instead of silently giving
  // run-to-run different bits (callers use spml_segment_sum_normalize_det_f32, which needs a workspace)
  if (deterministic_mode()) return SPML_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(sums, 0, (size_t)M * D * sizeof(float), s) != hipSuccess)
    return SPML_ERR_LAUNCH;
  int rc = segment_sum_launch(x, ids, P, D, M, sums, s, nullptr);
  if (rc != SPML_OK) return rc;
  return spml_normalize_rows_f32(sums, M, D, protos, stream);
}

extern "C" int spml_segment_sum_normalize_bwd_f32(const float* d_protos, const float* sums,
                                                  const int64_t* ids, int64_t P, int D,
                                                  int64_t M, float* d_sums_scratch, float* dx,
                                                  int accumulate, void* stream) {
  if (!d_protos || !sums || !ids || !d_sums_scratch || !dx || P < 0 || D <= 0 || M <= 0)
    return SPML_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(proto_bwd_rows, dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, d_protos,
                     sums, M, D, d_sums_scratch);
  if (P == 0) return launch_status();
  const int64_t total = P * (int64_t)D;
  const unsigned blocks = (unsigned)min((int64_t)4096, (total + 255) / 256);
  if (accumulate)
    hipLaunchKernelGGL(gather_rows<true>, dim3(blocks), dim3(256), 0, s, d_sums_scratch, ids, P,
                       D, M, dx);
  else
    hipLaunchKernelGGL(gather_rows<false>, dim3(blocks), dim3(256), 0, s, d_sums_scratch, ids,
                       P, D, M, dx);
  return launch_status();
}
