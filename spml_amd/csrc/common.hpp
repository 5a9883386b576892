// Shared device/host helpers for libspml_hip.so (gfx950 / CDNA4 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "spml_hip.h"

namespace spml {

constexpr int kWave = 64;          // CDNA wavefront
constexpr float kEps = 1e-12f;     // normalize_embedding eps (general/common.py:101)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

// The similarity contraction <x, c> runs on the f16 matrix cores at fp32
// accuracy: every fp32 operand v is written as  v = h + l / 2048  with
// h = f16(v) and l = f16((v - h) * 2048)  (two 11-bit mantissas = 22 bits, the
// residual is pre-scaled so that it stays a normal f16).  The product keeps
// the three terms  h*h' + (h*l' + l*h') / 2048 ; the dropped l*l' term is
// 2^-22 relative.  f16 x f16 products are exact in the fp32 accumulator, so the
// result is fp32-class (measured max |err| 2.3e-7 on unit vectors, the same as
// an fp32 GEMM) at 3/16 of the cost of the fp32-input MFMA.
constexpr float kSplitScale = 2048.0f;
constexpr float kSplitInv = 1.0f / 2048.0f;

__device__ __forceinline__ void split_f16(float v, _Float16& h, _Float16& l) {
  h = (_Float16)v;
  l = (_Float16)((v - (float)h) * kSplitScale);
}

// 8 consecutive fp32 -> (hi, lo) f16x8 fragments.
__device__ __forceinline__ void split8(const float (&v)[8], half8& h, half8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    _Float16 hj, lj;
    split_f16(v[j], hj, lj);
    h[j] = hj;
    l[j] = lj;
  }
}

// D(32x32) += A(32x16) * B(16x32) on the f16 matrix cores.
//   A fragment: lane holds A[row = lane & 31][k = 8*(lane>>5) + 0..7]
//   B fragment: lane holds B[k = 8*(lane>>5) + 0..7][col = lane & 31]
//   C/D:        lane holds D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane & 31]
__device__ __forceinline__ float16v mfma32(half8 a, half8 b, float16v c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// row index inside a 32x32 accumulator tile for register r of this lane
__device__ __forceinline__ int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
  return v;
}

__device__ __forceinline__ float block_sum_256(float v, float* smem4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) smem4[w] = v;
  __syncthreads();
  float t = 0.f;
  const int nw = (blockDim.x + 63) >> 6;
  for (int i = 0; i < nw; ++i) t += smem4[i];
  __syncthreads();
  return t;
}

// s_waitcnt vmcnt(n) for a run-time (wave-uniform) n; loads retire in order, so
// "at most n vector-memory ops outstanding" == "everything older has landed".
__device__ __forceinline__ void wait_vmcnt(int n) {
#define SPML_VM(N) case N: asm volatile("s_waitcnt vmcnt(" #N ")" ::: "memory"); break;
  switch (n) {
    SPML_VM(0) SPML_VM(1) SPML_VM(2) SPML_VM(3) SPML_VM(4) SPML_VM(5) SPML_VM(6) SPML_VM(7)
    SPML_VM(8) SPML_VM(9) SPML_VM(10) SPML_VM(11) SPML_VM(12) SPML_VM(13) SPML_VM(14)
    SPML_VM(15) SPML_VM(16) SPML_VM(17) SPML_VM(18) SPML_VM(19) SPML_VM(20) SPML_VM(21)
    SPML_VM(22) SPML_VM(23) SPML_VM(24) SPML_VM(25) SPML_VM(26) SPML_VM(27) SPML_VM(28)
    SPML_VM(29) SPML_VM(30) SPML_VM(31) SPML_VM(32) SPML_VM(33) SPML_VM(34) SPML_VM(35)
    SPML_VM(36)
    default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
  }
#undef SPML_VM
}

// raw workgroup barrier that does NOT drain in-flight LDS-DMA (vmcnt untouched)
__device__ __forceinline__ void wg_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Process-wide deterministic mode (spml_set_deterministic, misc.hip): every fp32 atomic accumulation of the hot path
// (segment sums, the prototype gradient of the NLL backward) is replaced by 64-bit integer atomics on a fixed-point
// image of the same values -- integer adds commute, so the result does not depend on the arrival order.
bool deterministic_mode();

// Fixed point of the deterministic accumulations: 2^36 steps per unit (kFixShiftF), values up to +-2^26.
constexpr float kDetFix = 68719476736.0f;            // 2^36
constexpr double kDetFixInv = 1.0 / 68719476736.0;
__device__ __forceinline__ long long det_to_fix(float v) {
  const float f = v * kDetFix;
  // (|f| >= 2^62 -- more than 2^26 units -- would overflow the conversion: saturate; the sums such values
  // would need do not fit the format either, and the host side documents the domain)
  const float lim = 4611686018427387904.0f;          // 2^62
  return (long long)(f > lim ? lim : (f < -lim ? -lim : f));
}
__device__ __forceinline__ void det_atomic_add(long long* p, float v) {
  const long long q = det_to_fix(v);
  if (q != 0) atomicAdd(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q);
}

inline int launch_status() {
  return hipGetLastError() == hipSuccess ? SPML_OK : SPML_ERR_LAUNCH;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

}  // namespace spml
