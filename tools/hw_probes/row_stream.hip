// Micro-benchmark: how fast can FEW workgroups each pull one contiguous row block?  The finalize of a k-means M-step
// would be ONE launch (no second kernel, no fence) if block k could sum its cluster's G x D partial sums (264 KB at
// G = 256, D = 258) in about the time the two-kernel form takes (5.5 + 4 us + a kernel boundary): K = 36 blocks x 1024
// threads, 16-byte loads, all of a thread's loads in flight, contiguous [K][G][D] layout against the strided [G][K][D]
// layout the passes write today (8-byte loads, stride K x D x 4).
// Build: hipcc --offload-arch=gfx950 -O3 row_stream.hip -o row_stream.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
constexpr int K = 36, G = 256, D = 264;          // D padded to a multiple of 4 for the float4 form
template <int MODE>
__global__ __launch_bounds__(1024) void probe(const float* __restrict__ slabs, float* __restrict__ out) {
  const int k = blockIdx.x, tid = threadIdx.x;
  __shared__ float part[16][D];
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    // contiguous: row block of cluster k = G x D floats = 16 896 float4; thread t takes float4 t, t + 1024, ...
    const float4* p = reinterpret_cast<const float4*>(slabs + (size_t)k * G * D);
    float4 t[17];
#pragma unroll
    for (int u = 0; u < 17; ++u) { const int i = tid + 1024 * u; t[u] = i < G * D / 4 ? p[i] : float4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
    for (int u = 0; u < 17; ++u) { acc.x += t[u].x; acc.y += t[u].y; acc.z += t[u].z; acc.w += t[u].w; }
  } else {
    // strided, as today: wave w -> (channel chunk w % 3 of 128 channels, slab group w / 3), 8-byte loads
    const int lane = tid & 63, wave = tid >> 6, chunk = wave % 3, grp = wave / 3;
    const int d = chunk * 128 + 2 * lane;
    if (grp < 5 && d < 258) {
      const float* p = slabs + (size_t)k * D + d;
      float2 t[52];
#pragma unroll
      for (int u = 0; u < 52; ++u) { const int g = grp * 52 + u; t[u] = g < G ? *reinterpret_cast<const float2*>(p + (size_t)g * K * D) : float2{0.f, 0.f}; }
#pragma unroll
      for (int u = 0; u < 52; ++u) { acc.x += t[u].x; acc.y += t[u].y; }
    }
  }
  part[tid >> 6][tid & 63] = acc.x + acc.y + acc.z + acc.w;
  __syncthreads();
  if (tid < 64) { float s = 0.f; for (int i = 0; i < 16; ++i) s += part[i][tid]; out[k * 64 + tid] = s; }
}
template <int MODE>
void run(const char* name) {
  float *slabs, *out;
  hipMalloc(&slabs, (size_t)K * G * D * 4 + 4096); hipMalloc(&out, K * 64 * 4);
  hipMemset(slabs, 0, (size_t)K * G * D * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float* big; hipMalloc(&big, 512u << 20);
  const int reps = 20;
  float tot = 0.f;
  for (int r = 0; r < reps; ++r) {
    hipMemsetAsync(big, r, 512u << 20, 0);          // evict L2 / MALL between launches (the passes stream 272 MB in between)
    hipEventRecord(e0);
    probe<MODE><<<K, 1024>>>(slabs, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); if (r >= 2) tot += ms;
  }
  printf("%-52s %7.2f us per launch (9.7 MB, cold)\n", name, tot * 1e3 / (reps - 2));
  hipFree(slabs); hipFree(out); hipFree(big);
}
int main() {
  run<0>("36 blocks, contiguous [K][G][D] rows, 16-byte loads");
  run<1>("36 blocks, strided [G][K][D] slabs, 8-byte loads");
  run<0>("36 blocks, contiguous [K][G][D] rows, 16-byte loads");
  return 0;
}
