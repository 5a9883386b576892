// Dense re-indexing of integer keys without a sort of the whole input:
//   uniq = sorted distinct keys, inv[i] = position of keys[i] in uniq
// -- what `torch.unique(keys, return_inverse=True)` returns in the reference's label algebra
// (segsort/common.py:192-218,398-405; models/utils.py:94-111), where the number of distinct keys
// (segments, a few thousand .. tens of thousands) is small against the number of pixels (hundreds
// of thousands).  Open-addressing hash set of the keys (insert, 64-bit CAS), compaction of the
// occupied slots, rank of every distinct key among the distinct keys, lookup.  Rank = number of smaller
// distinct keys: the list of distinct keys is sorted in 2048-key tiles (bitonic, in LDS), and a key's rank is the
// sum over the tiles of its lower bound in the tile -- U * (U / 2048) * 11 compares (round 3 counted all U^2
// pairs: 1.27 ms at U = 139 k, where every rank of an 8-GPU job re-indexes the global prototype set).  Ranks
// are a function of the key SET only: the result is deterministic although the insertion order is not.
#include "common.hpp"

namespace spml {
namespace {

constexpr long long kEmpty = (long long)0x8000000000000000ull;   // INT64_MIN: not a valid key

__device__ __forceinline__ unsigned long long mix64(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

__global__ void relabel_init(long long* table, int64_t T, unsigned long long* count) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < T) table[i] = kEmpty;
  if (i == 0) *count = 0ull;
}

__global__ void relabel_insert(const int64_t* __restrict__ keys, int64_t P, long long* table, int64_t mask) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const long long k = keys[i];
  int64_t h = (int64_t)(mix64((unsigned long long)k) & (unsigned long long)mask);
  while (true) {
    const long long seen = table[h];                       // most keys are already there: no atomic
    if (seen == k) return;
    if (seen == kEmpty) {
      const long long old = (long long)atomicCAS(reinterpret_cast<unsigned long long*>(table + h),
                                                 (unsigned long long)kEmpty, (unsigned long long)k);
      if (old == kEmpty || old == k) return;
    }
    h = (h + 1) & mask;
  }
}

// occupied slots -> list (any order: the ranks are a function of the key SET).  A block scans 4096 slots and claims
// its range of the list with ONE atomic (a per-slot -- even a per-wave -- atomic on the single counter serialises:
// 16 k waves took 200 us at T = 1 M)
constexpr int kCompactPerThread = 16;
__global__ __launch_bounds__(256) void relabel_compact(const long long* __restrict__ table, int64_t T, long long* list,
                                                       unsigned long long* count) {
  __shared__ unsigned wave_total[4];
  __shared__ unsigned long long block_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kCompactPerThread) + threadIdx.x;
  long long k[kCompactPerThread];
  unsigned before[kCompactPerThread];                    // occupied slots of this wave in front of slot (q, lane)
  unsigned mine = 0;
#pragma unroll
  for (int q = 0; q < kCompactPerThread; ++q) {
    const int64_t i = i0 + 256 * q;
    k[q] = i < T ? table[i] : kEmpty;
    const unsigned long long m = __ballot(k[q] != kEmpty);
    before[q] = mine + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
    mine += (unsigned)__popcll(m);                       // (wave-uniform running total)
  }
  if (lane == 0) wave_total[wave] = mine;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned total = wave_total[0] + wave_total[1] + wave_total[2] + wave_total[3];
    block_base = total ? atomicAdd(count, (unsigned long long)total) : 0ull;
  }
  __syncthreads();
  unsigned long long base = block_base;
  for (int w = 0; w < wave; ++w) base += wave_total[w];
#pragma unroll
  for (int q = 0; q < kCompactPerThread; ++q)
    if (k[q] != kEmpty) list[base + before[q]] = k[q];
}

// list[2048 t .. 2048 t + 2047] sorted in place (ascending; the last tile holds U - 2048 t keys)
constexpr int kTileKeys = 2048;
__global__ __launch_bounds__(256) void relabel_sort_tiles(long long* __restrict__ list,
                                                          const unsigned long long* __restrict__ count) {
  __shared__ long long tile[kTileKeys];
  const int64_t U = (int64_t)*count;
  const int64_t base = (int64_t)blockIdx.x * kTileKeys;
  if (base >= U) return;
  const int n = (int)min((int64_t)kTileKeys, U - base);
  for (int t = threadIdx.x; t < kTileKeys; t += 256) tile[t] = t < n ? list[base + t] : (long long)0x7fffffffffffffffll;
  __syncthreads();
  // pair index t of a stage with stride j <= 32 lies in the 64-pair group t / 64, whose 128 elements no other wave
  // touches before the stride grows again: those stages need no workgroup barrier (51 of the 66 stages)
  for (int k = 2; k <= kTileKeys; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = threadIdx.x; t < kTileKeys / 2; t += 256) {
        const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi = lo | j;     // the pair (lo, lo + j)
        const long long a = tile[lo], b = tile[hi];
        const bool up = (lo & k) == 0;
        if ((a > b) == up) { tile[lo] = b; tile[hi] = a; }
      }
      if (j > 32 || j == 1) __syncthreads();
      else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"), __builtin_amdgcn_wave_barrier(), __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  for (int t = threadIdx.x; t < n; t += 256) list[base + t] = tile[t];     // (padding sorts to the end)
}

// rank of list[i] among the distinct keys = number of smaller distinct keys = sum over the sorted tiles of the
// key's lower bound: blockIdx.y splits the tiles (kRankSplit at a time), integer atomics combine the partial counts
constexpr int kRankSplit = 16;
constexpr int kRankPerThread = 4;      // distinct keys ranked per thread
__global__ __launch_bounds__(256) void relabel_rank(const long long* __restrict__ list,
                                                    const unsigned long long* __restrict__ count,
                                                    int* __restrict__ rank) {
  __shared__ long long tile[kTileKeys];
  const int64_t U = (int64_t)*count;
  const int64_t i0 = (int64_t)blockIdx.x * (256 * kRankPerThread) + threadIdx.x;
  if ((int64_t)blockIdx.x * (256 * kRankPerThread) >= U) return;
  long long mine[kRankPerThread];
  int part[kRankPerThread];
#pragma unroll
  for (int q = 0; q < kRankPerThread; ++q) {
    const int64_t i = i0 + 256 * q;
    mine[q] = i < U ? list[i] : kEmpty;
    part[q] = 0;
  }
  for (int64_t base = (int64_t)blockIdx.y * kTileKeys; base < U; base += (int64_t)gridDim.y * kTileKeys) {
    const int n = (int)min((int64_t)kTileKeys, U - base);
    __syncthreads();
    for (int t = threadIdx.x; t < n; t += 256) tile[t] = list[base + t];
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kRankPerThread; ++q) {
      int lo = 0, hi = n;                                  // first position with tile[pos] >= mine
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (tile[mid] < mine[q]) lo = mid + 1; else hi = mid;
      }
      part[q] += lo;
    }
  }
#pragma unroll
  for (int q = 0; q < kRankPerThread; ++q) {
    const int64_t i = i0 + 256 * q;
    if (i < U && part[q]) atomicAdd(rank + i, part[q]);
  }
}

// uniq[rank] = key; slot_rank[slot of key] = rank
__global__ void relabel_place(const long long* __restrict__ list, const unsigned long long* __restrict__ count,
                              const int* __restrict__ rank, const long long* __restrict__ table, int64_t mask,
                              int64_t* __restrict__ uniq, int64_t cap, int* __restrict__ slot_rank) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)*count) return;
  const long long mine = list[i];
  const int r = rank[i];
  if (r < cap) uniq[r] = mine;
  int64_t h = (int64_t)(mix64((unsigned long long)mine) & (unsigned long long)mask);
  while (table[h] != mine) h = (h + 1) & mask;
  slot_rank[h] = r;
}

__global__ void relabel_lookup(const int64_t* __restrict__ keys, int64_t P, const long long* __restrict__ table,
                               int64_t mask, const int* __restrict__ slot_rank, int64_t* __restrict__ inv) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const long long k = keys[i];
  int64_t h = (int64_t)(mix64((unsigned long long)k) & (unsigned long long)mask);
  while (table[h] != k) h = (h + 1) & mask;
  inv[i] = slot_rank[h];
}

inline int64_t table_size(int64_t P) {
  int64_t t = 1024;
  while (t < 2 * P) t <<= 1;
  return t;
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" size_t spml_relabel_unique_workspace_bytes(int64_t P) {
  if (P < 0) return 0;
  const int64_t T = table_size(P);
  return (size_t)T * 8 + (size_t)T * 4 + align_up((size_t)P * 8 + 8, 256) + align_up((size_t)P * 4 + 4, 256) + 256;
}

// keys [P] (any int64 except INT64_MIN) -> inv [P]; uniq [uniq_capacity] receives the first uniq_capacity
// sorted distinct keys; count [1] (device) = number of distinct keys.  Stream-ordered, no host sync.
extern "C" int spml_relabel_unique_i64(const int64_t* keys, int64_t P, int64_t* inv, int64_t* uniq,
                                       int64_t uniq_capacity, int64_t* count, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (P < 0 || !count || (P > 0 && (!keys || !inv)) || uniq_capacity < 0 || (uniq_capacity > 0 && !uniq))
    return SPML_ERR_INVALID_ARG;
  if (!ws || ws_bytes < spml_relabel_unique_workspace_bytes(P)) return SPML_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int64_t T = table_size(P);
  unsigned char* b = static_cast<unsigned char*>(ws);
  long long* table = reinterpret_cast<long long*>(b);
  int* slot_rank = reinterpret_cast<int*>(b + (size_t)T * 8);
  long long* list = reinterpret_cast<long long*>(b + (size_t)T * 12);
  int* rank = reinterpret_cast<int*>(b + (size_t)T * 12 + align_up((size_t)P * 8 + 8, 256));
  unsigned long long* cnt = reinterpret_cast<unsigned long long*>(count);
  hipLaunchKernelGGL(relabel_init, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, s, table, T, cnt);
  if (P > 0) {
    const unsigned pb = (unsigned)((P + 255) / 256);
    hipLaunchKernelGGL(relabel_insert, dim3(pb), dim3(256), 0, s, keys, P, table, T - 1);
    hipLaunchKernelGGL(relabel_compact, dim3((unsigned)((T + 256 * kCompactPerThread - 1) / (256 * kCompactPerThread))), dim3(256), 0, s,
                       table, T, list, cnt);
    if (hipMemsetAsync(rank, 0, (size_t)P * 4, s) != hipSuccess) return SPML_ERR_LAUNCH;
    hipLaunchKernelGGL(relabel_sort_tiles, dim3((unsigned)((P + kTileKeys - 1) / kTileKeys)), dim3(256), 0, s, list, cnt);
    hipLaunchKernelGGL(relabel_rank, dim3((pb + kRankPerThread - 1) / kRankPerThread, kRankSplit), dim3(256), 0, s, list,
                       cnt, rank);
    hipLaunchKernelGGL(relabel_place, dim3(pb), dim3(256), 0, s, list, cnt, rank, table, T - 1, uniq, uniq_capacity,
                       slot_rank);
    hipLaunchKernelGGL(relabel_lookup, dim3(pb), dim3(256), 0, s, keys, P, table, T - 1, slot_rank, inv);
  }
  return launch_status();
}
