// Status strings, ABI version and the k-means grid initialisation (A3).
#include "common.hpp"

#include <algorithm>
#include <atomic>

namespace spml {
namespace {

// torch.linspace(0, end, n)[i] in fp32 as ATen's CPU kernel evaluates it:
// step = end/(n-1); first half counts up from 0, second half down from `end`.
__device__ __forceinline__ float linspace_at(float end, int i, int n) {
  if (n <= 1) return 0.f;
  const float step = end / (float)(n - 1);
  return (i < n / 2) ? step * (float)i : end - step * (float)(n - 1 - i);
}

// segsort/common.py:129-153: y_labels + (y_labels.max() + 1) * x_labels with
// labels = round_half_even(linspace(0, K-1, size)).
__global__ void init_grid_kernel(int H, int W, int Ky, int Kx, int64_t* out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= H * W) return;
  const int y = i / W, x = i - y * W;
  const int ly = (int)rintf(linspace_at((float)(Ky - 1), y, H));
  const int lx = (int)rintf(linspace_at((float)(Kx - 1), x, W));
  const int ymax = (int)rintf(linspace_at((float)(Ky - 1), H - 1, H));
  out[i] = (int64_t)ly + (int64_t)(ymax + 1) * lx;
}

// shader cycles against the 100-MHz wall clock over `spin_us` microseconds of sleeping
__global__ void clock_probe_kernel(unsigned long long* out, int spin_us) {
  const unsigned long long r0 = wall_clock64();
  const unsigned long long c0 = __builtin_readcyclecounter();
  unsigned long long r1 = r0;
  while (r1 - r0 < (unsigned long long)spin_us * 100ull) {
    __builtin_amdgcn_s_sleep(32);
    r1 = wall_clock64();
  }
  const unsigned long long c1 = __builtin_readcyclecounter();
  r1 = wall_clock64();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
}

}  // namespace
}  // namespace spml

namespace spml {
namespace {

// 3 x 3 max-pool, stride 2, padding 1 (the stem's `nn.MaxPool2d(3, 2, 1)`, spml/models/backbones/resnet.py:66-110) on a
// channels-last map: one thread per (output pixel, 4 channels), 16-byte loads of the up to nine taps -- 128 channels =
// 512 contiguous bytes per tap and pixel.  Forward only: conv1 / res2 are outside the autograd graph (frozen).
__global__ __launch_bounds__(256) void maxpool3x3s2_nhwc(const float* __restrict__ x, int N, int H, int W, int C4,
                                                        int OH, int OW, float* __restrict__ y) {
  const int64_t total = (int64_t)N * OH * OW * C4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int c = (int)(i % C4);
    int64_t r = i / C4;
    const int ow = (int)(r % OW); r /= OW;
    const int oh = (int)(r % OH);
    const int n = (int)(r / OH);
    float4v m = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
    for (int dh = 0; dh < 3; ++dh) {
      const int ih = 2 * oh - 1 + dh;
      if ((unsigned)ih >= (unsigned)H) continue;
#pragma unroll
      for (int dw = 0; dw < 3; ++dw) {
        const int iw = 2 * ow - 1 + dw;
        if ((unsigned)iw >= (unsigned)W) continue;
        const float4v v = reinterpret_cast<const float4v*>(x)[(((int64_t)n * H + ih) * W + iw) * C4 + c];
#pragma unroll
        for (int e = 0; e < 4; ++e) m[e] = v[e] > m[e] || v[e] != v[e] ? v[e] : m[e];      // (NaN propagates, as the framework's)
      }
    }
    reinterpret_cast<float4v*>(y)[i] = m;
  }
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" int spml_clock_probe(uint64_t* out, int spin_us, void* stream) {
  if (!out || spin_us <= 0) return SPML_ERR_INVALID_ARG;
  hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream,
                     reinterpret_cast<unsigned long long*>(out), spin_us);
  return launch_status();
}

extern "C" int spml_maxpool3x3s2_nhwc_f32(const float* x, int n, int H, int W, int C, float* y, void* stream) {
  if (!x || !y || n <= 0 || H <= 0 || W <= 0 || C <= 0) return SPML_ERR_INVALID_ARG;
  if ((C & 3) || (reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(y) & 15)) return SPML_ERR_UNSUPPORTED;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;          // floor((H + 2 - 3) / 2) + 1
  const int64_t total = (int64_t)n * OH * OW * (C >> 2);
  const int grid = (int)std::min<int64_t>((total + 255) / 256, 65536);
  hipLaunchKernelGGL(maxpool3x3s2_nhwc, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, n, H, W, C >> 2, OH, OW, y);
  return launch_status();
}

extern "C" const char* spml_status_string(int status) {
  switch (status) {
    case SPML_OK: return "ok";
    case SPML_ERR_INVALID_ARG: return "invalid argument (null pointer, bad size or alignment)";
    case SPML_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
    case SPML_ERR_WORKSPACE: return "workspace missing or too small";
    case SPML_ERR_LAUNCH: return "HIP launch failed";
    default: return "unknown spml status";
  }
}

extern "C" int spml_abi_version(void) { return SPML_ABI_VERSION; }

namespace spml {
namespace {
std::atomic<int> g_deterministic{0};
}
bool deterministic_mode() { return g_deterministic.load(std::memory_order_relaxed) != 0; }
}  // namespace spml

extern "C" int spml_set_deterministic(int on) {
  return spml::g_deterministic.exchange(on ? 1 : 0, std::memory_order_relaxed);
}

extern "C" int spml_get_deterministic(void) { return spml::deterministic_mode() ? 1 : 0; }

#ifndef SPML_BUILD_EXPERIMENT
#define SPML_BUILD_EXPERIMENT 0
#endif
extern "C" int spml_build_experiment(void) { return SPML_BUILD_EXPERIMENT; }

extern "C" int spml_kmeans_init_grid_i64(int H, int W, int Ky, int Kx, int64_t* out,
                                         void* stream) {
  if (!out || H <= 0 || W <= 0 || Ky <= 0 || Kx <= 0) return SPML_ERR_INVALID_ARG;
  hipLaunchKernelGGL(init_grid_kernel, dim3((unsigned)((H * W + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, H, W, Ky, Kx, out);
  return launch_status();
}
