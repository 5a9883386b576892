// Shared between the k-means pass kernels (kmeans.hip, kmeans64.hip): the argument block of a
// pass, the pre-converted tile format (fragment-major split-f16 blocks, one 32-pixel "pre-tile"
// at a time) and the small device helpers every pass uses.
#pragma once

#include "common.hpp"

namespace spml {

struct PassArgs {
  const float* x;
  int64_t x_bytes;            // total bytes of x (bounds for the tile copy)
  int64_t P;
  int D, K, n_img, G;
  const int64_t* seg_off;     // device [n_img+1]
  const _Float16* cent_h;     // [n_img][kpad][dpad]
  const _Float16* cent_l;
  int kpad, dpad;
  int nvt;                    // 4-KB copy rounds per tile (tile buffer = nvt * 4096 B)
  int32_t* labels;            // [P] in (accumulate-only) / out (assign)
  const int64_t* labels_in64; // accumulate-only pass: read the caller's int64 labels directly (their
                              // low words, stride 8) instead of `labels`; or null
  int64_t* labels_out64;      // assign pass: also the caller's int64 output (the last pass of a
                              // call writes it itself: no widening kernel); or null
  float* slabs;               // [n_img][G][K][D], fully overwritten by an M-step pass
  int do_assign, do_accum;
  int strided;                // kmeans_pass16: workgroup g takes tiles g, g + G, ... instead of a contiguous range
  const float* cent_f32;      // [n_img][K][D] fp32 prototypes
  unsigned long long* trace;  // per-phase cycle counters (only in -DSPML_TRACE builds)
  const unsigned char* xc;    // pre-converted tiles (kmeans_preconvert), or null
  unsigned char* xc_out;      // !PRE passes: also write every converted tile here (or null)
  unsigned long long* clocks; // profiling: [n_img][G][2] start / end of every workgroup in
                              // 100-MHz s_memrealtime ticks, or null (spml_kmeans_run_profiled_f32)
};

// first / last instruction of a pass kernel when a.clocks is set (one lane per workgroup; the
// end stamp is taken after this workgroup's stores have drained)
#define KM_CLOCK_BEGIN                                                                     \
  if (a.clocks && threadIdx.x == 0)                                                        \
    a.clocks[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x)] = wall_clock64();
#define KM_CLOCK_END                                                                       \
  if (a.clocks) {                                                                          \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                       \
    if (threadIdx.x == 0)                                                                  \
      a.clocks[2 * ((size_t)blockIdx.y * gridDim.x + blockIdx.x) + 1] = wall_clock64();   \
  }

// Phase instrumentation of kmeans_pass16 (build with SPML_TRACE=1 python -m spml_amd._build
// --force, run with SPML_KM_TRACE=1): cycles per phase of workgroup 7, fused passes.
#ifdef SPML_TRACE
#define KM_TRACE_DECL unsigned long long tc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; \
  const unsigned long long treal0 = wall_clock64();                      \
  unsigned long long tprev = __builtin_readcyclecounter();
#define KM_MARK(i) { const unsigned long long n_ = __builtin_readcyclecounter(); tc[i] += n_ - tprev; tprev = n_; }
#define KM_TRACE_DRAIN asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#define KM_TRACE_STORE                                                                            \
  if (a.trace && a.do_assign && a.do_accum && blockIdx.x == 7 && lane == 0) {                     \
    for (int i_ = 0; i_ < 8; ++i_) a.trace[wave * 8 + i_] = tc[i_];                               \
    a.trace[32 + wave] = wall_clock64() - treal0;   /* 100 MHz */                                 \
  }                                                                                               \
  if (a.trace && a.do_assign && a.do_accum && tid == 0 && blockIdx.x < 1024 && blockIdx.y == 0) { \
    a.trace[40 + 2 * blockIdx.x] = treal0;                                                        \
    a.trace[41 + 2 * blockIdx.x] = wall_clock64();                                                \
  }
#else
#define KM_TRACE_DECL
#define KM_MARK(i)
#define KM_TRACE_DRAIN
#define KM_TRACE_STORE
#endif

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float float4a __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4a mfma16(half8 a, half8 b, float4a c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

// label of row p for the LDS-DMA of an accumulate-only pass: the int32 work array, or the low
// word of the caller's int64 label (little endian; labels are < 2^31)
__device__ __forceinline__ gptr_t label_src(const PassArgs& a, int64_t p) {
  return a.labels_in64 ? (gptr_t)(reinterpret_cast<const int32_t*>(a.labels_in64 + p))
                       : (gptr_t)(a.labels + p);
}
__device__ __forceinline__ void label_store(const PassArgs& a, int64_t p, int v) {
  a.labels[p] = v;
  if (a.labels_out64) a.labels_out64[p] = (int64_t)v;
}

// Position (in 16-B units) of (pixel pix of a 16-pixel half, 8-channel group g) inside
// a 1-KB fragment block.  The permutation keeps both consumers conflict-free: the
// E-step's ds_read_b128 (every 16-lane service group sees all 16 residues mod 16) and
// the M-step's ds_read_b64_tr_b16 (a 32-lane half reads 8 pixels x 2 channel groups,
// again all 16 residues).
__host__ __device__ inline int frag_slot(int pix, int g) {
  return 16 * ((((pix >> 2) & 1) << 1) | (g >> 1)) + ((((pix & 3) | ((pix >> 3) << 2)) << 1) | (g & 1));
}

// bytes of one pre-converted 32-pixel tile: 4*Q fragment blocks of 1 KB ([k-step][pixel
// half][hi|lo], the LDS layout of the E-step operands) + 4 compact 256-B blocks for the
// location k-step (only its first 8 channels are stored: 2 real + 6 zero)
__host__ __device__ inline size_t pre_tile_bytes(int q, int tail) {
  return (size_t)q * 4096 + (tail ? 1024 : 0);
}
// first tile of image `img` in the pre-converted buffer (closed form, no prefix sum:
// sum_{i<img} ceil(len_i/32) <= floor(seg0/32) + img)
__host__ __device__ inline int64_t pre_tile0(int64_t seg0, int img) { return (seg0 >> 5) + img; }

// LDS bytes of one pre-tile slot of the 64-pixel-tile kernels: Q k-steps of 4 KB + the location k-step as four
// 512-B blocks ([pixel half][hi|lo]: 256 B of data + 256 B that stay zero: channel groups 1..3 of that k-step);
// a workgroup holds a ring of two tiles = four slots + 2 KB (labels of a tile)
__host__ __device__ constexpr int p64_slot_bytes(int q, int tail) { return q * 4096 + (tail ? 2048 : 0); }
__host__ __device__ constexpr int p64_lds(int q, int tail) { return 4 * p64_slot_bytes(q, tail) + 2048; }

// ---- kmeans64.hip: the pixel-split pass (64-pixel tiles, one workgroup of 4 waves per CU for wide rows) ----
// shapes it covers (pre-converted tiles, an E-step in the pass): K <= 48, D = 32 q + tail
bool pass64_shape(int D, int K);
int pass64_wg_per_cu(int D, int K);
size_t pass64_lds_bytes(int D);
// a.G workgroups per image; a.do_assign must be set; a.do_accum selects the fused / E-only kernel
int launch_pass64(const PassArgs& a, hipStream_t s);

// ---- kmeans64k.hip: E-step for 64 < K <= 144 on the same tiles, the prototype tiles split over eight waves ----
bool assign64k_shape(int D, int K);
int assign64k_kpad(int K);           // prototype rows incl. padding tiles (80, 96, 128 or 144)
int launch_assign64k(const PassArgs& a, hipStream_t s);
int launch_accum64k(const PassArgs& a, hipStream_t s);   // M-step only (a.do_accum), labels from a.labels / a.labels_in64

}  // namespace spml
