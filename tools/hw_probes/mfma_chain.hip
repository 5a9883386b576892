// Probe: are dependent in-place v_mfma_f32_16x16x32_f16 (srcC = vdst) correct when issued back to back from inline asm
// (no compiler-inserted wait states), at dependency distance 1, 2, 3?  Sums 64 products with small integers (exact).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float4a __attribute__((ext_vector_type(4)));
#define MV(acc, a_, b_) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a_), "v"(b_))
template <int DIST>
__global__ void k(float* out) {
  const int lane = threadIdx.x;
  half8 a[4], b[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) { a[j][i] = (_Float16)(float)((lane + i + j) % 3); b[j][i] = (_Float16)(float)((lane * 3 + i + 2 * j) % 4); }
  float4a acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  asm volatile("s_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
  for (int it = 0; it < 16; ++it) {
    if (DIST == 1) { MV(acc[0], a[0], b[0]); MV(acc[0], a[1], b[1]); MV(acc[0], a[2], b[2]); MV(acc[0], a[3], b[3]); }
    if (DIST == 2) { MV(acc[0], a[0], b[0]); MV(acc[1], a[0], b[0]); MV(acc[0], a[1], b[1]); MV(acc[1], a[1], b[1]);
                     MV(acc[0], a[2], b[2]); MV(acc[1], a[2], b[2]); MV(acc[0], a[3], b[3]); MV(acc[1], a[3], b[3]); }
    if (DIST == 3) { for (int j = 0; j < 4; ++j) { MV(acc[0], a[j], b[j]); MV(acc[1], a[j], b[j]); MV(acc[2], a[j], b[j]); } }
  }
  asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]));
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[0][r];
}
template <int DIST>
__global__ void ref(float* out) {
  const int lane = threadIdx.x;
  half8 a[4], b[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) { a[j][i] = (_Float16)(float)((lane + i + j) % 3); b[j][i] = (_Float16)(float)((lane * 3 + i + 2 * j) % 4); }
  float4a acc = {0, 0, 0, 0};
  for (int it = 0; it < 16; ++it)
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[j], b[j], acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[lane * 4 + r] = acc[r];
}
int main() {
  float *o, *p; (void)hipMalloc(&o, 1024); (void)hipMalloc(&p, 1024);
  float h[256], g[256];
  hipLaunchKernelGGL(ref<1>, dim3(1), dim3(64), 0, 0, p); (void)hipMemcpy(g, p, 1024, hipMemcpyDeviceToHost);
#define RUN(D_) { hipLaunchKernelGGL(k<D_>, dim3(1), dim3(64), 0, 0, o); (void)hipMemcpy(h, o, 1024, hipMemcpyDeviceToHost); \
    int bad = 0; for (int i = 0; i < 256; ++i) bad += h[i] != g[i]; printf("distance %d: %d of 256 values differ (e.g. %g vs %g)\n", D_, bad, h[5], g[5]); }
  RUN(1) RUN(2) RUN(3)
  return 0;
}
