// Probe: is a vector-ALU write interlocked with a hand-written MFMA that reads the register straight after it
// (as srcC / srcA / srcB)?  The compiler inserts wait states for MFMAs it knows; inline asm gets none.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4a __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  half8 a, b, a2;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((lane + i) % 3); b[i] = (_Float16)(float)((lane * 3 + i) % 4); a2[i] = (_Float16)0.f; }
  float4a c = {1.f, 2.f, 3.f, 4.f}, d = {0, 0, 0, 0}, junk = {100.f, 200.f, 300.f, 400.f};
  asm volatile("s_nop 7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(junk), "+v"(a2));
  if (MODE == 0)       // srcC written by v_mov right before
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tv_mfma_f32_16x16x32_f16 %4, %5, %6, %7"
                 : "+v"(junk[0]), "+v"(junk[3]), "+v"(c[0]), "+v"(c[3]), "=&v"(d) : "v"(a), "v"(b), "v"(junk));
  if (MODE == 1)       // srcA written by v_mov right before (a2 = 0 -> a)
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %3\n\tv_mfma_f32_16x16x32_f16 %4, %5, %6, %7"
                 : "+v"(a2[0]), "+v"(a2[3]), "+v"(a[0]), "+v"(a[3]), "=&v"(d) : "v"(a2), "v"(b), "v"(c));
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(d));
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
template <int MODE>
__global__ void ref(float* out) {
  const int lane = threadIdx.x;
  half8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(float)((lane + i) % 3); b[i] = (_Float16)(float)((lane * 3 + i) % 4); }
  float4a c = {1.f, 2.f, 3.f, 4.f};
  if (MODE == 0) c = float4a{1.f, 200.f, 300.f, 4.f};
  if (MODE == 1) { half8 z; for (int i = 0; i < 8; ++i) z[i] = (_Float16)0.f; z[0] = a[0]; z[1] = a[1]; z[6] = a[6]; z[7] = a[7]; a = z; }
  float4a d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = d[r];
}
int main() {
  float *o, *p; (void)hipMalloc(&o, 1024); (void)hipMalloc(&p, 1024);
  float h[256], g[256];
#define RUN(M_) { hipLaunchKernelGGL(ref<M_>, dim3(1), dim3(64), 0, 0, p); (void)hipMemcpy(g, p, 1024, hipMemcpyDeviceToHost); \
    hipLaunchKernelGGL(k<M_>, dim3(1), dim3(64), 0, 0, o); (void)hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost); \
    int bad = 0; for (int i = 0; i < 256; ++i) bad += h[i] != g[i]; printf("mode %d (%s written by the vector ALU right before the MFMA): %d of 256 values differ (e.g. %g vs %g)\n", M_, M_ ? "srcA" : "srcC", bad, h[4], g[4]); }
  RUN(0) RUN(1)
  return 0;
}
