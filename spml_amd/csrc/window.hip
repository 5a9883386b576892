// N2: overlap accumulation of sliding-window crop embeddings into the
// full-resolution map.  Replaces, per crop, the chain of
// pyscripts/inference/prototype.py:163-178: permute -> normalize_embedding -> permute ->
// `embeddings[:, :, sh:eh, sw:ew] += crop_emb` -> `counts[:, :, sh:eh, sw:ew] += 1`
// (five passes over the crop) by one pass: the channel vector of a pixel is normalised
// in registers and added in place.  HBM-bound: reads the crop once (the second channel
// loop hits L2), one read-modify-write of the window.
#include "common.hpp"

namespace spml {
namespace {

// thread = one crop pixel (x fastest -> coalesced channel-plane accesses)
__global__ __launch_bounds__(256) void window_accumulate_kernel(
    const float* __restrict__ patch, int C, int h, int w, float* __restrict__ acc,
    float* __restrict__ counts, int H, int W, int sh, int sw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= h * w) return;
  const int y = i / w, x = i - y * w;
  const size_t plane = (size_t)h * w;
  float ss = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = patch[c * plane + i];
    ss += v * v;
  }
  const float nrm = sqrtf(ss);
  const float inv = 1.0f / (nrm >= kEps ? nrm : kEps);     // general/common.py:101-120
  const size_t o = (size_t)(sh + y) * W + (sw + x);
  const size_t big = (size_t)H * W;
  for (int c = 0; c < C; ++c) acc[c * big + o] += patch[c * plane + i] * inv;
  counts[o] += 1.0f;
}

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" int spml_window_accumulate_f32(const float* patch, int C, int h, int w, float* acc,
                                          float* counts, int H, int W, int sh, int sw,
                                          void* stream) {
  if (!patch || !acc || !counts || C <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0)
    return SPML_ERR_INVALID_ARG;
  if (sh < 0 || sw < 0 || sh + h > H || sw + w > W) return SPML_ERR_INVALID_ARG;
  hipLaunchKernelGGL(window_accumulate_kernel, dim3((unsigned)((h * w + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, patch, C, h, w, acc, counts, H, W, sh, sw);
  return launch_status();
}
