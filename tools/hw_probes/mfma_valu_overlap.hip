// Micro-benchmark: do the MFMA work of one wave and the VALU work of ANOTHER wave on the same SIMD
// overlap?  One workgroup of 512 threads per CU: waves 0-3 (one per SIMD) and waves 4-7 (their SIMD
// partners).  Per iteration a "matrix" wave issues NM v_mfma_f32_32x32x16_f16 on two independent
// accumulators, a "vector" wave NV dependent-chain-free VALU ops (4 chains of v_fma / v_exp mix).
//   mode 0: waves 0-3 matrix, waves 4-7 idle          mode 1: waves 0-3 idle, waves 4-7 vector
//   mode 2: waves 0-3 matrix, waves 4-7 vector         mode 3: all 8 waves: matrix burst then vector burst
//   mode 4: all 8 waves: 1 MFMA then NV/NM VALU, interleaved in program order
//   mode 5: 4 waves only (one per SIMD): matrix burst then vector burst (no partner)
//   mode 6: 4 waves only: interleaved
//   mode 7: all 8 waves vector only        mode 8: all 8 waves matrix only
// Build: hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form mfma_valu_overlap.hip -o mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
constexpr int NM = 24;     // MFMAs per iteration (as one prototype-tile step of nll_fwd2)
constexpr int NV = 96;     // VALU ops per iteration

__device__ __forceinline__ void matrix_burst(float16v& z0, float16v& z1, half8 a, half8 b) {
#pragma unroll
  for (int i = 0; i < NM / 2; ++i) {
    z0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z0, 0, 0, 0);
    z1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z1, 0, 0, 0);
  }
}
__device__ __forceinline__ void vector_burst(float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < NV / 8; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(1.0001f), "v"(0.5f));
  }
}
template <int MODE>
__global__ __launch_bounds__(512, 1) void probe(float* out, unsigned long long* cyc, int iters) {
  const int wv = threadIdx.x >> 6;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.01f * (threadIdx.x & 7)); b[e] = (_Float16)(0.02f * e); }
  float16v z0, z1;
  for (int r = 0; r < 16; ++r) { z0[r] = 0.f; z1[r] = 0.f; }
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = 0.001f * (threadIdx.x + k);
  const bool matrix = (MODE == 0 || MODE == 2) ? wv < 4 : (MODE >= 3 && MODE != 7);
  const bool vector = (MODE == 1 || MODE == 2) ? wv >= 4 : (MODE >= 3 && MODE != 8);
  const bool active = (MODE == 5 || MODE == 6) ? wv < 4 : true;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  if (active) {
    for (int it = 0; it < iters; ++it) {
      if (MODE == 4 || MODE == 6) {
#pragma unroll
        for (int i = 0; i < NM; ++i) {
          if (i & 1) z1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z1, 0, 0, 0);
          else z0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z0, 0, 0, 0);
#pragma unroll
          for (int k = 0; k < NV / NM; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k]) : "v"(1.0001f), "v"(0.5f));
        }
      } else {
        if (matrix) matrix_burst(z0, z1, a, b);
        if (vector) vector_burst(v);
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int r = 0; r < 16; ++r) s += z0[r] + z1[r];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == (MODE == 1 ? 256 : 0)) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* what, int iters) {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(512), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double avg = 0; for (int i = 0; i < 256; ++i) avg += h[i]; avg /= 256;
  printf("mode %d  %-58s %8.1f us  %7.0f s_memtime ticks / iteration  (%.0f ns)\n", MODE, what, ms * 1e3, avg / iters, ms * 1e6 / iters);
  hipFree(out); hipFree(cyc);
}
int main() {
  const int iters = 2000;
  printf("per iteration: %d MFMA 32x32x16 f16 (2 accumulators) = %d pipe cycles; %d v_fma_f32\n", NM, NM * 32, NV);
  run<0>("waves 0-3 matrix, partners idle", iters);
  run<1>("waves 4-7 vector, partners idle", iters);
  run<2>("waves 0-3 matrix beside waves 4-7 vector", iters);
  run<3>("8 waves, each: matrix burst then vector burst", iters);
  run<4>("8 waves, each: 1 MFMA + 4 VALU interleaved", iters);
  run<5>("4 waves, each: matrix burst then vector burst", iters);
  run<6>("4 waves, each: 1 MFMA + 4 VALU interleaved", iters);
  run<7>("8 waves, vector only", iters);
  run<8>("8 waves, matrix only", iters);
  return 0;
}
