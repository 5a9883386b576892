// Micro-benchmark: what it costs 256 workgroups (one per CU, 256 threads) to add their [K][D] partial sums into a
// shared table with 64-bit integer atomics (no return value) instead of storing a slab each -- the end of a fused
// k-means pass (K x D = 36 x 258 = 9288 values per workgroup).  Integer adds are associative: the table is
// bit-reproducible whatever the arrival order.
//   mode 0: plain 8-byte stores to a slab per workgroup (what kmeans_pass64 does today, 4-byte values there)
//   mode 1: atomics, agent scope, ONE table
//   mode 2: atomics, agent scope, one table per XCD (hardware XCC_ID)
//   mode 3: atomics, workgroup scope (performed in the XCD's own L2), one table per XCD
// Times: HIP events around 20 launches, and the sum over the 8 tables is checked against the expected value.
// Build: hipcc --offload-arch=gfx950 -O3 global_atomics.hip -o global_atomics.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
constexpr int N = 36 * 258;
__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int MODE>
__global__ __launch_bounds__(256) void probe(long long* tab, long long* slabs, unsigned* xcd_of) {
  const int g = blockIdx.x;
  const unsigned x = xcc_id();
  if (threadIdx.x == 0) xcd_of[g] = x;
  long long* t = MODE >= 2 ? tab + (size_t)x * N : tab;
  for (int i = threadIdx.x; i < N; i += 256) {
    const long long v = (long long)(g + 1) * (i + 1);
    if (MODE == 0) __builtin_nontemporal_store(v, slabs + (size_t)g * N + i);
    if (MODE == 1 || MODE == 2) __hip_atomic_fetch_add(t + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 3) __hip_atomic_fetch_add(t + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
template <int MODE>
void run(const char* name) {
  long long *tab, *slabs; unsigned* xo;
  hipMalloc(&tab, 8 * N * 8); hipMalloc(&slabs, (size_t)256 * N * 8); hipMalloc(&xo, 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 256>>>(tab, slabs, xo);
  hipDeviceSynchronize();
  hipMemset(tab, 0, 8 * N * 8);
  const int reps = 20;
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) probe<MODE><<<256, 256>>>(tab, slabs, xo);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(8 * N); std::vector<unsigned> hx(256);
  hipMemcpy(h.data(), tab, 8 * N * 8, hipMemcpyDeviceToHost);
  hipMemcpy(hx.data(), xo, 256 * 4, hipMemcpyDeviceToHost);
  long long bad = 0;
  if (MODE) for (int i = 0; i < N; ++i) {
    long long s = 0; for (int x = 0; x < (MODE >= 2 ? 8 : 1); ++x) s += h[(size_t)x * N + i];
    if (s != (long long)reps * (i + 1) * (256LL * 257 / 2)) ++bad;
  }
  int cnt[16] = {0}; int rr = 1; for (int g = 0; g < 256; ++g) { cnt[hx[g]]++; if (hx[g] != (unsigned)(g % 8)) rr = 0; }
  printf("%-44s %7.2f us per launch, wrong sums %lld, workgroups per XCD %d %d %d %d %d %d %d %d, g %% 8 == xcd: %d\n", name,
         ms * 1e3 / reps, bad, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7], rr);
  hipFree(tab); hipFree(slabs); hipFree(xo);
}
int main() {
  run<0>("stores, slab per workgroup");
  run<1>("atomics agent scope, one table");
  run<2>("atomics agent scope, table per XCD");
  run<3>("atomics workgroup scope, table per XCD");
  run<0>("stores, slab per workgroup");
  return 0;
}
