// Probe: v_mfma_f32_16x16x32_f16 with -inf in srcC (a score bias for padding rows): what comes out when the
// products are 0 * x, for x finite / negative / f16-denormal, and over a chain of accumulating MFMAs.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4a __attribute__((ext_vector_type(4)));
__global__ void k(float* out, int mode) {
  const int lane = threadIdx.x;
  half8 a, b;
  for (int i = 0; i < 8; ++i) {
    a[i] = (_Float16)0.f;
    float v = (mode == 0 || mode == 3) ? 0.37f * ((lane * 7 + i) % 11 - 5) : (mode == 1 ? -1.5f : 3.0e-7f * ((lane + i) % 5 - 2));
    b[i] = (_Float16)v;
  }
  float4a c = {-INFINITY, -INFINITY, 0.f, -INFINITY};
  float4a d;
  if (mode >= 3) {      // the form kmeans_pass64 uses: A in an accumulation register, srcC another register than vdst
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %3" : "=&v"(d) : "a"(a), "v"(b), "v"(c));
    asm volatile("s_nop 15\n\ts_nop 15" : "+v"(d));
  } else
  d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int s = 0; s < 8; ++s) d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main() {
  float* o; hipMalloc(&o, 256 * 4);
  float h[256];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, mode);
    hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) { float v = h[l * 4 + r]; bool ok = (r == 2) ? v == 0.f : (isinf(v) && v < 0); if (!ok) { if (bad < 8) printf("mode %d lane %d r %d -> %g\n", mode, l, r, v); ++bad; } }
    printf("mode %d: %d unexpected values\n", mode, bad);
  }
  return 0;
}
