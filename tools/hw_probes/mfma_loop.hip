// Micro-benchmark: how many cycles does one "k-step" of the split-f16 GEMM inner loop take on
// one wave per SIMD (4 waves per CU, 1 workgroup per CU)?
//   mode 0: 3 MFMA 32x32x16 per step, operands in registers only
//   mode 1: + 2 ds_read_b128 (A operand) per step, two steps ahead, counted lgkmcnt waits
//   mode 2: mode 1 + 6 VALU ops per step
//   mode 3: mode 1 + one global_load_lds (1 KB) per step
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_loop.hip -o mfma_loop
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr int NK = 33;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const half8* __restrict__ bsrc, const unsigned char* gsrc,
                                                float* out, unsigned long long* cyc, int tiles) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  half8 bh[NK], bl[NK];
#pragma unroll
  for (int s = 0; s < NK; ++s) { bh[s] = bsrc[(s * 64 + lane)]; bl[s] = bsrc[((NK + s) * 64 + lane)]; }
  for (int i = threadIdx.x; i < 2 * NK * 1024 / 4; i += 256) reinterpret_cast<float*>(lds)[i] = 0.001f * (i & 15);
  __syncthreads();
  float16v acc_h, acc_x, acc_y;
  for (int r = 0; r < 16; ++r) { acc_h[r] = 0; acc_x[r] = 0; acc_y[r] = 0; }
  float best = 0.f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int t = 0; t < tiles; ++t) {
    half8 ah[2], al[2];
    const unsigned cbase = (unsigned)(size_t)(lptr_t)(lds) + 16u * lane;
    auto load_a = [&](int s, int b) {
      const unsigned addr = cbase + (unsigned)s * 2048u;
      asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024" : "=&v"(ah[b]), "=&v"(al[b]) : "v"(addr));
    };
    if (MODE >= 1) { load_a(0, 0); load_a(1, 1); }
    else { ah[0] = bh[0]; al[0] = bl[0]; ah[1] = bh[1]; al[1] = bl[1]; }
#pragma unroll
    for (int s = 0; s < NK; ++s) {
      const int b = s & 1;
      if (MODE >= 1) {
        if (s + 1 < NK) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[b]), "+v"(al[b]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[b]), "+v"(al[b]));
      }
      acc_h = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bh[s], acc_h, 0, 0, 0);
      acc_x = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[b], bl[s], acc_x, 0, 0, 0);
      acc_y = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[b], bh[s], acc_y, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE >= 1 && s + 2 < NK) load_a(s + 2, b);
      if (MODE == 2) {
#pragma unroll
        for (int q = 0; q < 6; ++q) best = best * 1.0001f + (float)(s + q);
      }
      if (MODE == 3 && s < 17)
        __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + (size_t)(s * 4 + (threadIdx.x >> 6)) * 1024 + 16 * lane),
                                         (lptr_t)(lds + NK * 2048 + (s * 4 + (threadIdx.x >> 6)) * 1024), 16, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (MODE == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = best;
  for (int r = 0; r < 16; ++r) s += acc_h[r] + acc_x[r] + acc_y[r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const half8* b, const unsigned char* g, float* out, unsigned long long* cyc, int tiles) {
  const int lds = 2 * NK * 2048 + 4096;
  hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  probe<MODE><<<256, 256, lds>>>(b, g, out, cyc, tiles);
  hipEventRecord(e0);
  probe<MODE><<<256, 256, lds>>>(b, g, out, cyc, tiles);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i];
  m /= 256;
  printf("mode %d: %.1f us, %.0f shader cycles per workgroup -> %.1f cycles per k-step (3 MFMA), clock %.2f GHz, %.0f TFLOP/s\n",
         MODE, ms * 1e3, m, m / (tiles * NK), m / (ms * 1e3) / 1e3,
         256.0 * 4 * tiles * NK * 3 * 32768.0 / (ms * 1e-3) / 1e12);
}

int main() {
  half8* b; unsigned char* g; float* out; unsigned long long* cyc;
  hipMalloc(&b, 2 * NK * 64 * 16); hipMemset(b, 0x11, 2 * NK * 64 * 16);
  hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int tiles = 64;
  run<0>(b, g, out, cyc, tiles);
  run<1>(b, g, out, cyc, tiles);
  run<2>(b, g, out, cyc, tiles);
  run<3>(b, g, out, cyc, tiles);
  return 0;
}
