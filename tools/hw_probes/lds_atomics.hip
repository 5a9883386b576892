// Micro-benchmark: cost of a 64-lane LDS atomic instruction (no return value) per operand type,
// conflict-free (lane i -> consecutive element i) and with all waves of a workgroup hammering the
// same LDS: ds_add_u32, ds_add_f32, ds_add_u64, ds_add_f64, plus plain ds_write_b64 for scale.
// One workgroup of W waves per CU, N instructions per wave.  Output: cycles per instruction of the
// workgroup (wall cycles * 1 / (N * W)) -- i.e. the reciprocal throughput of the CU's LDS pipe.
// Build: hipcc --offload-arch=gfx950 -O3 lds_atomics.hip -o lds_atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) double* ldouble_t;
template <int MODE>
__global__ __launch_bounds__(1024) void probe(unsigned long long* cyc, int n, float* sink) {
  __shared__ double tab[8 * 1024];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < 8 * 1024; i += blockDim.x) tab[i] = 0.0;
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < n; ++i) {
    const int row = (i * 7 + wv * 3) & 63;            // 64 rows of 64 x 8 B = 512 B
    if (MODE == 0) atomicAdd(reinterpret_cast<unsigned int*>(tab) + row * 128 + lane, 1u);
    if (MODE == 1) __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float*)(reinterpret_cast<float*>(tab) + row * 128 + lane), 1.0f, 0, 0, false);
    if (MODE == 2) atomicAdd(reinterpret_cast<unsigned long long*>(tab) + row * 64 + lane, 1ull);
    if (MODE == 3) __builtin_amdgcn_ds_atomic_fadd_f64((ldouble_t)(tab + row * 64 + lane), 1.0, 0, 0, false);
    if (MODE == 4) tab[row * 64 + lane] = (double)i;
  }
  __syncthreads();
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
  if (tab[threadIdx.x] == 12345.678) sink[0] = 1.f;
}
template <int MODE>
void run(const char* name, int waves) {
  unsigned long long* d; float* s;
  hipMalloc(&d, 256 * 8); hipMalloc(&s, 4);
  const int n = 4096;
  probe<MODE><<<256, waves * 64>>>(d, n, s);
  probe<MODE><<<256, waves * 64>>>(d, n, s);
  hipDeviceSynchronize();
  unsigned long long h[256];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  double m = 0; for (int i = 0; i < 256; ++i) m += h[i]; m /= 256;
  printf("%-14s %2d waves: %7.1f cycles per wave instruction of the CU (%.1f per wave)\n", name, waves,
         m / ((double)n * waves), m / n);
  hipFree(d); hipFree(s);
}
int main() {
  for (int w : {1, 8, 16}) {
    run<0>("ds_add_u32", w); run<1>("ds_add_f32", w); run<2>("ds_add_u64", w); run<3>("ds_add_f64", w);
    run<4>("ds_write_b64", w);
  }
  return 0;
}
