// Hardware probe: lane/element mapping of gfx950's ds_read_b64_tr_b16.
//   hipcc --offload-arch=gfx950 -O2 ds_read_tr16.hip -o ds_read_tr16.bin && ./ds_read_tr16.bin
// Every lane l passes the address of 4 consecutive 16-bit elements (4l .. 4l+3).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short short4v __attribute__((vector_size(8)));
__global__ void k(short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  for (int i = threadIdx.x; i < 1024; i += 64) reinterpret_cast<short*>(lds)[i] = (short)i;
  __syncthreads();
  short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
      (__attribute__((address_space(3))) short4v*)(lds + threadIdx.x * 8));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d;
  (void)hipMalloc(&d, 64 * 4 * 2);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d);
  short h[256];
  (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  // expected: within a 16-lane group the lanes' 4-element rows form a [4][16] matrix
  // (row r = lanes 4r..4r+3), and lane i receives column i
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4, i = l & 15;
    for (int j = 0; j < 4; ++j) {
      const int e = 4 * (16 * g + 4 * j + (i >> 2)) + (i & 3);
      if (h[l * 4 + j] != e) ++bad;
    }
  }
  for (int l = 0; l < 20; ++l)
    printf("lane %d: %d %d %d %d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  printf("bad=%d\n", bad);
  return 0;
}
