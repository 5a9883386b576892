// Stride-1 "same" convolutions (1x1 and dilated 3x3) of the ResNet bottleneck stack
// (spml/models/backbones/resnet.py:11-63: conv1/conv2/conv3 + downsample of every unit in
// res4/res5 = 83 % of the training step's flops) as implicit GEMMs on the f16 matrix cores at
// fp32-class accuracy.
//
// The fp32 matrix path of gfx950 peaks at 157 TFLOP/s; the library kernels already run at 80 %
// of it (profiles/r02_train_step_steady_state.md).  The f16 path is 16x faster, and the split
//     v * S = h + l,   h = f16(v * S),  l = f16(v * S - h)            (S = per-tensor power of two)
// carries 22 mantissa bits for every element within 2^15 of the tensor's largest magnitude
// (smaller ones keep an absolute error of 2^-39 of it), so that
//     a * b = (ha*hb + ha*lb + la*hb) / (Sa * Sb)       (la*lb, 2^-24 relative, is dropped)
// costs three f16 MFMAs -- all exact products accumulated in fp32 -- instead of one fp32 MFMA
// at 1/16 of the rate.  S comes from an upper bound of max|v| that the producing kernel knows
// (batch-norm statistics) or a reduction over the tensor (weights): stored next to the tensor as
// one float, never read by the host.
//
// "hl8" layout of a split tensor [rows][C]: 16-byte units, unit ((row * C/8 + c/8) * 2 + part)
// holds 8 consecutive channels of part h (0) or l (1) -- 4 B per element like fp32, and the
// operand fragment of v_mfma_f32_32x32x16_f16 (8 consecutive k of one row) is one unit, so the
// LDS-DMA gathers fragments straight from HBM/L2 into fragment-major 1-KB blocks.
//
//   forward        out[r][co] = sum_{tap,ci} x[r + shift(tap)][ci] * w[co][tap][ci]
//   data gradient  the same kernel on (dy, w transposed and tap-flipped [ci][tap'][co]),
//                  optionally accumulating the gradient of the residual branch (addend)
//   weight gradient  dw[co][tap][ci] = sum_r dy[r][co] * x[r + shift(tap)][ci]   (conv_wgrad)
#include "common.hpp"

#include <math.h>
#include <stdlib.h>

#ifndef SPML_CONV_EXP
#define SPML_CONV_EXP 0     // profiling builds of conv_gemm: 1 A operand from the zero page, 2 B operand from it, 4 no MFMAs,
                            // 16 shader clocks / 100-MHz ticks of tile 7 -> out[0..1], 32 per-wave phase cycles -> out[8..23]
                            // (timing only: both overwrite output values; tools/probe_conv_power.py reads them)
#endif

namespace spml {
namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ uint4 g_zero_page[16];       // source of the padded taps (never written)

// S = 2^(14 - e) with bound < 2^e: the scaled tensor stays below 2^14 (f16 max 65504)
__device__ __forceinline__ float pow2_scale(float bound) {
  if (!(bound > 0.f) || bound > 1e38f) return 1.f;
  int e;
  (void)frexpf(bound, &e);
  e = 14 - e;
  e = e < -100 ? -100 : (e > 100 ? 100 : e);
  return ldexpf(1.f, e);
}

__device__ __forceinline__ void split_unscaled(const float (&v)[8], float s, half8& h, half8& l) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * s;
    const _Float16 hj = (_Float16)x;
    h[j] = hj;
    l[j] = (_Float16)(x - (float)hj);
  }
}

// ---------------------------------------------------------------------------------------
// converters
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void absmax_bound(const float* __restrict__ x, int64_t n,
                                                    unsigned* __restrict__ bound) {
  __shared__ float sm[4];
  float m = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4v v = reinterpret_cast<const float4v*>(x)[i];
    m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);
  }
  if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) m = fmaxf(m, fabsf(x[(n4 << 2) + threadIdx.x]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
    atomicMax(bound, __float_as_uint(m));          // non-negative floats order like their bits
  }
}

// fp32 [rows][C] -> hl8; one thread per 8-channel unit pair
__global__ __launch_bounds__(256) void hl8_convert(const float* __restrict__ x, int64_t units,
                                                   const float* __restrict__ bound,
                                                   uint4* __restrict__ out) {
  const float s = bound ? pow2_scale(*bound) : 1.f;
  for (int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (int64_t)gridDim.x * 256) {
    const float4v v0 = reinterpret_cast<const float4v*>(x)[2 * u];
    const float4v v1 = reinterpret_cast<const float4v*>(x)[2 * u + 1];
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    union { half8 h; uint4 u; } hh, ll;
    split_unscaled(v, s, hh.h, ll.h);
    out[2 * u] = hh.u;
    out[2 * u + 1] = ll.u;
  }
}

// weights [Cout][taps][Cin] fp32 -> hl8 [Cin][taps'][Cout], taps' = taps - 1 - tap (the
// data-gradient convolution runs over the mirrored taps); thread = (ci fastest, co-block, tap)
__global__ __launch_bounds__(256) void hl8_convert_wt(const float* __restrict__ w, int Cout, int taps,
                                                      int Cin, const float* __restrict__ bound,
                                                      uint4* __restrict__ out, int taps_total, int tap_offset) {
  const float s = bound ? pow2_scale(*bound) : 1.f;
  const int64_t total = (int64_t)Cin * taps * (Cout >> 3);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int ci = (int)(i % Cin);
    const int64_t rest = i / Cin;
    const int cb = (int)(rest % (Cout >> 3)), tapo = (int)(rest / (Cout >> 3));
    const int tap = taps - 1 - tapo;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = w[((size_t)(cb * 8 + e) * taps + tap) * Cin + ci];
    union { half8 h; uint4 u; } hh, ll;
    split_unscaled(v, s, hh.h, ll.h);
    const size_t u = ((size_t)ci * taps_total + tap_offset + tapo) * (Cout >> 3) + cb;
    out[2 * u] = hh.u;
    out[2 * u + 1] = ll.u;
  }
}

// All convolution weights of one bottleneck unit in two launches (instead of four per weight):
// bounds first, then both operand layouts of every weight.
struct WeightSet {
  const float* w[4];          // [Cout][taps][Cin] fp32 (channels-last storage)
  uint4* fwd[4];              // hl8 [Cout][taps*Cin]
  uint4* tr[4];               // hl8 [Cin][taps*Cout], mirrored taps
  float* bound;               // [4]
  int cout[4], cin[4], taps[4];
  int n;
  int64_t start[5];           // prefix of 8-element work items per weight
};

__global__ __launch_bounds__(256) void weightset_absmax(const WeightSet ws) {
  __shared__ float sm[4];
  const int wi = blockIdx.y;
  const float* x = ws.w[wi];
  const int64_t n4 = ((int64_t)ws.cout[wi] * ws.cin[wi] * ws.taps[wi]) >> 2;
  float m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4v v = reinterpret_cast<const float4v*>(x)[i];
    m = fmaxf(fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))), m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, kWave));
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    atomicMax(reinterpret_cast<unsigned*>(ws.bound + wi), __float_as_uint(fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]))));
}

__global__ __launch_bounds__(256) void weightset_convert(const WeightSet ws) {
  const int64_t total = ws.start[ws.n];
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    int wi = 0;
    while (wi + 1 < ws.n && i >= ws.start[wi + 1]) ++wi;
    const int64_t u = i - ws.start[wi];                   // unit index in the forward layout
    const int cout = ws.cout[wi], cin = ws.cin[wi], taps = ws.taps[wi];
    const float s = pow2_scale(ws.bound[wi]);
    const float4v v0 = reinterpret_cast<const float4v*>(ws.w[wi])[2 * u];
    const float4v v1 = reinterpret_cast<const float4v*>(ws.w[wi])[2 * u + 1];
    const float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    union { half8 h; uint4 q; _Float16 e[8]; } hh, ll;
    split_unscaled(v, s, hh.h, ll.h);
    ws.fwd[wi][2 * u] = hh.q;
    ws.fwd[wi][2 * u + 1] = ll.q;
    // the same 8 values (co, tap, ci0..ci0+7) scattered into the transposed operand:
    // element (ci, taps-1-tap, co) = channel co & 7 of unit ((ci*taps + tapo)*(cout/8) + co/8)
    const int c8 = cin >> 3;
    const int ci0 = (int)(u % c8) * 8;
    const int64_t rest = u / c8;
    const int tap = (int)(rest % taps), co = (int)(rest / taps);
    const int tapo = taps - 1 - tap;
    _Float16* th = reinterpret_cast<_Float16*>(ws.tr[wi]);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const size_t tu = ((size_t)(ci0 + e) * taps + tapo) * (cout >> 3) + (co >> 3);
      th[(2 * tu) * 8 + (co & 7)] = hh.e[e];
      th[(2 * tu + 1) * 8 + (co & 7)] = ll.e[e];
    }
  }
}

// ---------------------------------------------------------------------------------------
// implicit GEMM: out[R][N] = A[R][taps*K] * B[N][taps*K]^T, both operands hl8
// ---------------------------------------------------------------------------------------
struct ConvArgs {
  const uint4* a;            // activations hl8 [R][K/8][2]
  const uint4* b;            // weights hl8 [N][taps*K/8][2]
  const float* a_bound;      // scale bounds (nullptr = unscaled)
  const float* b_bound;
  const float* addend;       // optional [R][N], added to the result
  const unsigned char* addend_mask;   // optional [R][N/4]: bit (n & 3) of byte n / 4 gates the addend (ReLU mask)
  const float* bias;         // optional [N] (inference: batch norm folded into the weights)
  int relu;                  // inference: out = max(out, 0)
  float* out_bound;          // optional: max |out| over the tensor (atomic max; zeroed by the host wrapper)
  float* out;                // [R][N] fp32
  int64_t R;
  int H, W, K, N, taps, dil;
  int n_col_tiles, n_tiles;
  int dils[4];               // taps == 9 * groups (groups > 1): group g = tap / 9 has dilation dils[g]
  int tap_groups;            // > 1: gridDim.y workgroups per tile, each takes taps / tap_groups consecutive taps and
                             // ADDS its partial tile to `out` (zeroed by the host) with fp32 atomics
  float* stats;              // optional (RG == 1 tiles): [4][row tiles][N] = per row tile and column the mean, the
                             // sum of squared deviations, the max and the min of `out` -- the chunk statistics the
                             // batch norm that follows would otherwise take from a pass of its own (bn_partial)
  float* stats_zero;         // optional: a float this launch resets to 0 (the bound slot bn_merge max-reduces into)
};

// Workgroup = 4 waves side by side along N: wave w owns output columns [64w, 64w+64) of the
// 256-column tile for all RB 32-row blocks (RB x 2 accumulator tiles of 32x32).  A pipeline stage
// holds 16 k of the A tile (RB*32 rows, shared by the four waves) and of the B tile (256 columns,
// each wave reads only its own two blocks), filled by LDS-DMA three stages deep.  One DMA
// instruction moves 16 rows x 64 contiguous bytes (k16, h and l): the four lanes of a quad read
// one 64-byte segment, which is what the texture addresser coalesces (a 16-byte gather per lane
// from 64 different lines runs at a quarter of the rate).  The 16-byte slots of a row are rotated
// by (row >> 2) so that the fragment reads (ds_read_b128, 16 lanes per pass) stay conflict-free.  LDS <= 78 KB and <= 256 registers: two workgroups per CU, the
// second one's MFMAs cover the first one's barriers.  Fragment reads are hand-issued one row
// block ahead of the MFMAs that consume them (counted lgkmcnt waits).
// NC < 4 (narrow outputs: N a multiple of 128 / 64 only): NC waves side by side along N, the tile is
// 64 * NC columns wide and RG row groups tall -- (NC, RG) = (2, 2) or (1, 4) keep four waves per workgroup.
template <int RB, int kStages, int WGS, int RG, bool CHUNK = false, int NC = 4>
__global__ __launch_bounds__(64 * NC * RG, WGS) void conv_gemm(const ConvArgs a) {
  // CHUNK (very long reductions, e.g. the 36-tap forward of a wide ASPP head, K = 73 728): the MFMA chain
  // is cut every 1024 k -- the running accumulator is added to a second register set and restarted -- so
  // that the fp32 accumulation error stays at the level of a K = 1024 convolution
  // RG row groups of 4 waves: RG = 2 doubles the tile height to 2 * RB row blocks sharing one B tile
  constexpr int kABlocks = 2 * RB * RG;          // A half blocks (16 rows) per stage
  constexpr int kBlocks = kABlocks + 4 * NC;     // + B: 4 half blocks (16 columns each) per column group
  constexpr int kStage = kBlocks * 1024;
  constexpr int kWaves = NC * RG;
  constexpr int NQ = (kABlocks + kWaves - 1) / kWaves;   // A blocks a wave may load per stage
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wc = wave % NC, wg = wave / NC;      // column group (64 columns), row group
  const int lr = lane & 31;
  // DMA role of this lane: row (lane >> 2) of a 16-row half block, 16-byte piece (k8 group, part)
  const int drow = lane >> 2, dpiece = ((lane & 3) - (drow >> 2)) & 3;
  // fragment (row lr, k8 group lane >> 5): byte offsets of its h and l slots inside a 2-KB block
  const unsigned foff_h = (unsigned)((lr >> 4) * 1024 + 16 * (4 * (lr & 15) + (((lane >> 5) * 2 + ((lr & 15) >> 2)) & 3)));
  const unsigned foff_l = (unsigned)((lr >> 4) * 1024 + 16 * (4 * (lr & 15) + (((lane >> 5) * 2 + 1 + ((lr & 15) >> 2)) & 3)));

  unsigned long long exp_c0 = 0, exp_r0 = 0;
  if (SPML_CONV_EXP & 16) { exp_c0 = __builtin_amdgcn_s_memtime(); exp_r0 = __builtin_amdgcn_s_memrealtime(); }
  // consecutive tiles (sharing A rows / the weight slab) stay on one XCD's L2
  int t = blockIdx.x;
  if ((a.n_tiles & 7) == 0) t = (t & 7) * (a.n_tiles >> 3) + (t >> 3);
  const int row_tile = t / a.n_col_tiles, col_tile = t - row_tile * a.n_col_tiles;
  const int64_t m0 = (int64_t)row_tile * (RG * RB * 32);
  const int n0 = col_tile * (64 * NC) + wc * 64;
  const int k8 = a.K >> 3, nk = a.K >> 4;
  const int hw = a.H * a.W;
  // taps of this workgroup (gridDim.y tap groups), minus those that read nothing but padding for every row
  // of the tile (a tile of consecutive pixels covers a few image rows: with the large dilations of the
  // pyramid head whole taps fall outside the map for the tiles near its top / bottom edge)
  const int tpg = a.taps / (int)gridDim.y, tap_begin = (int)blockIdx.y * tpg, tap_end = tap_begin + tpg;
  int oh_lo = 0, oh_hi = a.H - 1;                  // image rows the tile touches (whole map if it spans images)
  {
    const int64_t rows_t = RG * RB * 32;
    const int64_t last = (m0 + rows_t < a.R ? m0 + rows_t : a.R) - 1;
    if (m0 / hw == last / hw) { oh_lo = (int)((m0 % hw) / a.W); oh_hi = (int)((last % hw) / a.W); }
  }
  auto tap_active = [&](int tap) -> bool {
    if (a.taps < 9) return true;
    const int g = tap / 9, t3 = (tap - 9 * g) / 3;
    const int d = a.taps == 9 ? a.dil : (g == 0 ? a.dils[0] : (g == 1 ? a.dils[1] : (g == 2 ? a.dils[2] : a.dils[3])));
    const int dh = (t3 - 1) * d;
    return oh_hi + dh >= 0 && oh_lo + dh < a.H;
  };
  int n_active = 0, first_tap = tap_end;
  for (int tp = tap_end - 1; tp >= tap_begin; --tp)
    if (tap_active(tp)) { ++n_active; first_tap = tp; }
  const int total = n_active * nk;

  // A half blocks q = wave, wave + kWaves, ...: rows m0 + 16 q .. + 15
  int64_t arow[NQ];
  int aoh[NQ], aow[NQ];
  int my_dma = 4 / RG;                           // + this wave's share of its column group's four B blocks
#pragma unroll
  for (int i = 0; i < NQ; ++i) {
    const int q = wave + kWaves * i;
    if (q < kABlocks) ++my_dma;
    const int64_t row = m0 + q * 16 + drow;
    const bool ok = row < a.R && q < kABlocks;
    const int64_t rr = row < a.R ? row : a.R - 1;
    const int pix = (int)(rr % hw);
    aoh[i] = ok ? pix / a.W : -(1 << 20);        // rows past the end read the zero page
    aow[i] = pix % a.W;
    arow[i] = rr;
  }
  const uint4* bsrc[4 / RG];                     // the 16-column half blocks wg, wg + RG, ... of the group
#pragma unroll
  for (int j = 0; j < 4 / RG; ++j)
    bsrc[j] = a.b + (size_t)(n0 + (wg + RG * j) * 16 + drow) * ((size_t)a.taps * k8) * 2 + dpiece;

  int i_tap = first_tap, i_kc = 0, i_slot = 0;     // stages are issued in order
  auto issue = [&](int) {
    const int tap = i_tap, kc = i_kc;
    int dh = 0, dw = 0;
    if (a.taps >= 9) {
      const int g = tap / 9, t9 = tap - 9 * g, t3 = t9 / 3;
      const int d = a.taps == 9 ? a.dil : (g == 0 ? a.dils[0] : (g == 1 ? a.dils[1] : (g == 2 ? a.dils[2] : a.dils[3])));
      dh = (t3 - 1) * d;
      dw = (t9 - 3 * t3 - 1) * d;
    }
    unsigned char* base = lds + i_slot * kStage;
    if (++i_kc == nk) {
      i_kc = 0;
      do { ++i_tap; } while (i_tap < tap_end && !tap_active(i_tap));
    }
    if (++i_slot == kStages) i_slot = 0;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
      const int q = wave + kWaves * i;
      if (q < kABlocks) {                         // wave-uniform
        const bool ok = (unsigned)(aoh[i] + dh) < (unsigned)a.H && (unsigned)(aow[i] + dw) < (unsigned)a.W;
        const uint4* src = (ok && !(SPML_CONV_EXP & 1)) ? a.a + ((arow[i] + dh * a.W + dw) * k8 + kc * 2) * 2 + dpiece : g_zero_page;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + q * 1024), 16, 0, 0);
      }
    }
    const size_t o = ((size_t)tap * k8 + kc * 2) * 2;
    unsigned char* bb = base + (kABlocks + 4 * wc) * 1024;
#pragma unroll
    for (int j = 0; j < 4 / RG; ++j)
      __builtin_amdgcn_global_load_lds((gptr_t)((SPML_CONV_EXP & 2) ? g_zero_page : bsrc[j] + o), (lptr_t)(bb + (wg + RG * j) * 1024), 16, 0, 0);
  };

  float16v acc[RB][2];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  float16v tot[CHUNK ? RB : 1][2];
  if (CHUNK) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[i][j][r] = 0.f;
  }
  const unsigned lbase = (unsigned)(size_t)(lptr_t)lds;
  unsigned exp_ph[4] = {0, 0, 0, 0};
  unsigned long long exp_last = 0;
  issue(0);
  if (kStages > 2 && total > 1) issue(1);
  int c_slot = 0;
  for (int s = 0; s < total; ++s) {
    // this wave's share of stage s has landed (kStages - 2 younger stages may still be in flight)
    unsigned long long ph0 = 0, ph1 = 0, ph2 = 0;
    if (SPML_CONV_EXP & 32) ph0 = __builtin_amdgcn_s_memtime();
    wait_vmcnt(kStages > 2 && s + 1 < total ? my_dma : 0);
    if (SPML_CONV_EXP & 32) ph1 = __builtin_amdgcn_s_memtime();
    wg_barrier();                                 // ... everyone's; stage s-1 is fully consumed
    if (SPML_CONV_EXP & 32) ph2 = __builtin_amdgcn_s_memtime();
    if (s + kStages - 1 < total) issue(s + kStages - 1);
    if (SPML_CONV_EXP & 32) {                     // cycles in: DMA wait, barrier, DMA issue, (previous stage's) fragment reads + MFMAs
      const unsigned long long ph3 = __builtin_amdgcn_s_memtime();
      exp_ph[0] += (unsigned)(ph1 - ph0); exp_ph[1] += (unsigned)(ph2 - ph1); exp_ph[2] += (unsigned)(ph3 - ph2);
      if (s) exp_ph[3] += (unsigned)(ph0 - exp_last);
      exp_last = ph3;
    }
    const unsigned sb = lbase + (unsigned)(c_slot * kStage);
    if (++c_slot == kStages) c_slot = 0;
    half8 bh0, bl0, bh1, bl1, ah[2], al[2];
    {
      const unsigned bb = sb + (unsigned)((kABlocks + 4 * wc) * 1024);
      asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\t"
                   "ds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %5 offset:2048"
                   : "=&v"(bh0), "=&v"(bl0), "=&v"(bh1), "=&v"(bl1) : "v"(bb + foff_h), "v"(bb + foff_l));
    }
    const unsigned sa = sb + (unsigned)(wg * RB * 2048);          // this row group's A blocks
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3" : "=&v"(ah[0]), "=&v"(al[0])
                 : "v"(sa + foff_h), "v"(sa + foff_l));
#pragma unroll
    for (int i = 0; i < RB; ++i) {
      const int c = i & 1;
      if (i + 1 < RB) {
        const unsigned ad = sa + (unsigned)((i + 1) * 2048);
        asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %3"
                     : "=&v"(ah[c ^ 1]), "=&v"(al[c ^ 1]) : "v"(ad + foff_h), "v"(ad + foff_l));
        if (i == 0)
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[c]), "+v"(al[c]), "+v"(bh0), "+v"(bl0), "+v"(bh1), "+v"(bl1));
        else
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[c]), "+v"(al[c]));
      } else {
        if (i == 0)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[c]), "+v"(al[c]), "+v"(bh0), "+v"(bl0), "+v"(bh1), "+v"(bl1));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[c]), "+v"(al[c]));
      }
      if (!(SPML_CONV_EXP & 4)) {
      acc[i][0] = mfma32(al[c], bh0, acc[i][0]);
      acc[i][1] = mfma32(al[c], bh1, acc[i][1]);
      acc[i][0] = mfma32(ah[c], bl0, acc[i][0]);
      acc[i][1] = mfma32(ah[c], bl1, acc[i][1]);
      acc[i][0] = mfma32(ah[c], bh0, acc[i][0]);
      acc[i][1] = mfma32(ah[c], bh1, acc[i][1]);
      } else {
        asm volatile("" :: "v"(al[c]), "v"(ah[c]), "v"(bh0), "v"(bl0), "v"(bh1), "v"(bl1));
      }
    }
    if (CHUNK && (s & 63) == 63) {                // every 64 stages = 1024 k
#pragma unroll
      for (int i = 0; i < RB; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) { tot[i][j][r] += acc[i][j][r]; acc[i][j][r] = 0.f; }
    }
  }
  if (CHUNK) {
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] += tot[i][j][r];
  }

  const float mult = 1.0f / ((a.a_bound ? pow2_scale(*a.a_bound) : 1.f) * (a.b_bound ? pow2_scale(*a.b_bound) : 1.f));
  float vmax = 0.f;
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int64_t row = m0 + (wg * RB + i) * 32 + acc_row(r, lane);
      if (row < a.R) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const size_t o = (size_t)row * a.N + n0 + j * 32 + lr;
          float v = acc[i][j][r] * mult;
          if (gridDim.y > 1) {                      // tap groups: partial tiles meet in `out`
            if (a.bias && blockIdx.y == 0) v += a.bias[n0 + j * 32 + lr];
            unsafeAtomicAdd(a.out + o, v);
            continue;
          }
          if (a.bias) v += a.bias[n0 + j * 32 + lr];
          if (a.addend) {
            float ad = a.addend[o];
            if (a.addend_mask) ad = (a.addend_mask[o >> 2] >> (o & 3)) & 1 ? ad : 0.f;     // N % 4 == 0
            v += ad;
          }
          if (a.relu) v = fmaxf(v, 0.f);
          vmax = fmaxf(vmax, fabsf(v));
          a.out[o] = v;
        }
      }
    }
  if constexpr (RG == 1) {
    if (a.stats && gridDim.y == 1) {               // uniform
      // a wave owns its 64 columns over all RB * 32 rows of the tile: column statistics stay inside the wave
      // (lane and lane ^ 32 hold the two halves of a column).  Two passes over the registers: mean, then the
      // squared deviations from it (no E[x^2] - E[x]^2 cancellation); bn_merge pools the tiles (Chan).
      if (a.stats_zero && blockIdx.x == 0 && threadIdx.x == 0) *a.stats_zero = 0.f;
      const int64_t rows_here = (m0 + RB * 32 < a.R ? (int64_t)RB * 32 : a.R - m0);
      const float inv_rows = 1.0f / (float)rows_here;
      const size_t plane = (size_t)gridDim.x / a.n_col_tiles * a.N;       // row tiles x N
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float sm = 0.f, hi = -3.4e38f, lo = 3.4e38f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = m0 + i * 32 + acc_row(r, lane) < a.R;
            const float v = acc[i][j][r] * mult;
            sm += ok ? v : 0.f;
            hi = fmaxf(hi, ok ? v : -3.4e38f);
            lo = fminf(lo, ok ? v : 3.4e38f);
          }
        sm += __shfl_xor(sm, 32, kWave);
        hi = fmaxf(hi, __shfl_xor(hi, 32, kWave));
        lo = fminf(lo, __shfl_xor(lo, 32, kWave));
        const float mean = sm * inv_rows;
        float m2 = 0.f;
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool ok = m0 + i * 32 + acc_row(r, lane) < a.R;
            const float d = acc[i][j][r] * mult - mean;
            m2 += ok ? d * d : 0.f;
          }
        m2 += __shfl_xor(m2, 32, kWave);
        if (lane < 32) {
          float* dst = a.stats + (size_t)row_tile * a.N + n0 + j * 32 + lr;
          dst[0] = mean;
          dst[plane] = m2;
          dst[2 * plane] = hi;
          dst[3 * plane] = lo;
        }
      }
    }
  }
  if ((SPML_CONV_EXP & 32) && blockIdx.x == 7 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.out[8 + wave * 4 + i] = (float)exp_ph[i];
  }
  if ((SPML_CONV_EXP & 16) && blockIdx.x == 7 && threadIdx.x == 0) {
    a.out[0] = (float)(__builtin_amdgcn_s_memtime() - exp_c0);
    a.out[1] = (float)(__builtin_amdgcn_s_memrealtime() - exp_r0);
  }
  if (a.out_bound) {                             // wave-uniform
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o, kWave));
    if (lane == 0) atomicMax(reinterpret_cast<unsigned*>(a.out_bound), __float_as_uint(vmax));
  }
}

template <int RB, int kStages, int WGS, int RG = 1, bool CHUNK = false, int NC = 4>
int launch_conv(const ConvArgs& a0, hipStream_t s) {
  ConvArgs a = a0;
  a.n_col_tiles = a.N / (64 * NC);
  const int64_t row_tiles = (a.R + RG * RB * 32 - 1) / (RG * RB * 32);
  a.n_tiles = (int)(row_tiles * a.n_col_tiles);
  auto kern = conv_gemm<RB, kStages, WGS, RG, CHUNK, NC>;
  const int lds = kStages * (2 * RB * RG + 4 * NC) * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int groups = a.tap_groups > 1 ? a.tap_groups : 1;
  if (groups > 1 && hipMemsetAsync(a.out, 0, (size_t)a.R * a.N * sizeof(float), s) != hipSuccess) return SPML_ERR_LAUNCH;
  hipLaunchKernelGGL(kern, dim3(a.n_tiles, groups), dim3(64 * NC * RG), lds, s, a);
  return launch_status();
}

// narrow outputs: N % 256 != 0.  128-column tiles: 2 x 2 waves, 256 rows (RB = 4: 72 KB of LDS, two workgroups
// per CU); 64-column tiles: 1 x 4 waves, 256 rows (RB = 2: 60 KB) -- SPML_CONV_NARROW_RB=3 takes 384 rows
// (84 KB, one workgroup per CU)
template <bool CHUNK>
int launch_conv_narrow(const ConvArgs& c, hipStream_t s) {
  if ((c.N & 127) == 0) return launch_conv<CHUNK ? 3 : 4, 3, 2, 2, CHUNK, 2>(c, s);   // (chunked: two accumulator sets)
  const char* e = getenv("SPML_CONV_NARROW_RB");         // (read per call: the library keeps no state)
  if (e && e[0] == '3') return launch_conv<3, 3, 1, 4, CHUNK, 1>(c, s);
  return launch_conv<2, 3, 2, 4, CHUNK, 1>(c, s);
}

// rows per tile: the tallest tile (RB = 5: the weight blocks and the barrier are amortised over the most
// MFMAs -- 5-10 % faster than RB = 3 / 4 on every res4 / res5 shape, tools/bench_conv.py with
// SPML_CONV_RB) unless the problem is too small to give every CU a workgroup
// SPML_CONV_CHUNK_MIN_K (tuning / A-B switch): smallest K * taps that takes the chunked accumulation
inline bool chunk_long_reduction(int64_t k_total) {
  int64_t min_k = 4096;
  if (const char* e = getenv("SPML_CONV_CHUNK_MIN_K")) min_k = atoll(e);
  return k_total >= min_k;
}

inline int pick_rb(int64_t R, int N) {
  if (const char* e = getenv("SPML_CONV_RB")) {          // tuning aid
    const int v = atoi(e);
    if (v >= 3 && v <= 5) return v;
  }
  const int64_t tiles5 = (R + 159) / 160 * (N / 256);
  return tiles5 >= 384 ? 5 : (tiles5 >= 192 ? 4 : 3);
}

// ---------------------------------------------------------------------------------------
// weight gradient: dw[n][tap][k] = sum_r dy[r][n] * x[r + shift(tap)][k]
// ---------------------------------------------------------------------------------------
// The reduction runs over pixel rows, i.e. both operands are needed transposed ([channel][8
// pixels] fragments out of [pixel][channel] tensors): the LDS image is built for the hardware
// transpose read (ds_read_b64_tr_b16: a 16-lane group reads a [4 pixels][16 channels] block and
// every lane receives one channel of the 4 pixels).  One 1-KB DMA block = 8 pixels x 32 channels,
// h and l: quad Q = (pixel & 3) + 4 * (16-channel half) + 8 * (pixel >> 2) holds one pixel's 16
// channels (64 contiguous bytes of the hl8 tensor); inside the quad the h units sit at slots
// {0,1} for the first channel half and {2,3} for the second, so that the 16 units one 32-lane pass
// reads cover all 16 slot residues (conflict-free).
//
// Workgroup = 8 waves (2 along n x 4 along k), output tile TN (n) x TK (k) of one tap (256 x 256: wave
// tile 128 x 64 = 4 x 2 accumulators of 32x32; 128-wide tiles halve that dimension); a stage = 16 pixel
// rows (32 KB at 256 x 256), 3-stage ring.
// The pixel range is split over `splits` workgroups per tile; partial tiles go to the workspace
// and conv_wgrad_reduce sums them in a fixed order (deterministic).
struct WgradArgs {
  const uint4* dy;           // hl8 [R][N/8][2]
  const uint4* x;            // hl8 [R][K/8][2]
  float* partial;            // [splits][tiles][256][256]
  int64_t R;
  int H, W, K, N, taps, dil;
  int splits, rows_per_split; // rows_per_split % 16 == 0
  int k_tiles;               // K / 256
  int pyr_taps;              // PYR: 9 * branches (taps past it are padding of the last group of four)
  unsigned pyr_dils;         // PYR: dilation of branch b in byte b
};

typedef short short4v __attribute__((vector_size(8)));
typedef __attribute__((address_space(3))) short4v* trptr_t;

// TN x TK = output tile of one tap (256 or 128 each): channel counts that are multiples of 128 only (res3:
// 512 -> 128 -> 128 -> 512) run 128-wide tiles in that dimension -- half the accumulators per wave, the same
// LDS image and transpose reads.
//
// PYR (the 64-column branches of the pyramid head, `spml/models/heads/spp.py:8-43`: up to four dilated 3x3
// convolutions of ONE input whose outputs are summed, so all share one dy): with 64 output channels a tile of one tap
// would stream x once per tap for a quarter of a tile's products.  Here the roles of the shift are swapped,
//   dw_b[n][tap][k] = sum_r x[r][k] * dy[r - shift_b(tap)][n],
// and the 256 "n" columns of a tile are FOUR taps x 64 channels: x is read unshifted, once per group of four taps (9
// groups for 36 taps), and each 32-channel group of the dy image comes from its tap's shifted rows (the zero page
// where the shifted pixel leaves the image).  Everything after the DMA is the 256 x 256 kernel.
template <int TN, int TK, int kStages = 3, bool PAIR = false, bool PYR = false>
__global__ __launch_bounds__(512, 2) void conv_wgrad(const WgradArgs a) {
  static_assert(!PYR || (TN == 256 && TK == 256), "pyramid mode: four taps x 64 channels");
  constexpr int GN = TN / 32, GK = TK / 32;              // 32-channel groups of dy / x per stage
  constexpr int NBLK = 2 * (GN + GK);                    // 1-KB blocks per stage (two 8-pixel blocks per group)
  constexpr int NI = TN / 64, NJ = TK / 128;             // accumulator tiles per wave (2 x 4 waves)
  constexpr int NDMA = NBLK / 8;                         // DMA instructions per wave and stage
  constexpr int kStage = NBLK * 1024;
  extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wn = wave >> 2, wk = wave & 3;
  const int tile = blockIdx.x, split = blockIdx.y;
  const int tap = tile % a.taps, rest = tile / a.taps;
  const int kt = rest % a.k_tiles, nt = rest / a.k_tiles;
  const int n8 = a.N >> 3, k8 = a.K >> 3, hw = a.H * a.W;
  int dh = 0, dw = 0;
  if (!PYR && a.taps == 9) { dh = (tap / 3 - 1) * a.dil; dw = (tap % 3 - 1) * a.dil; }
  // PYR: this wave's two dy blocks (b = wave, wave + 8) are 32-channel groups cg = b >> 1 of the tile's 256 columns:
  // tap 4 * (tile's group) + (cg >> 1), channel half cg & 1
  int ph[2] = {0, 0}, pw[2] = {0, 0};
  bool pok[2] = {false, false};
  if constexpr (PYR) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = 4 * tap + (((wave + 8 * i) >> 1) >> 1);
      if (t < a.pyr_taps) {
        const int br = t / 9, t9 = t - 9 * br, d = (int)((a.pyr_dils >> (8 * br)) & 255u);
        ph[i] = (t9 / 3 - 1) * d;
        pw[i] = (t9 % 3 - 1) * d;
        pok[i] = true;
      }
    }
  }

  // DMA role: quad Q = lane >> 2 -> pixel (Q & 3) + 4 * (Q >> 3) of the 8-pixel block, channel
  // half (Q >> 2) & 1; slot lane & 3 -> unit (lane & 1) of the half, part ((lane >> 1) & 1) ^ half
  const int dq = lane >> 2, dpix = (dq & 3) + 4 * (dq >> 3), dhalf = (dq >> 2) & 1;
  const int dunit = dhalf * 2 + (lane & 1), dpart = ((lane >> 1) & 1) ^ dhalf;
  // this wave loads blocks b = wave, wave + 8, ... of the stage's NBLK:
  // b < 2 GN: dy, 32-channel group b >> 1, pixel block b & 1;  the others: x likewise
  const int64_t r_begin = (int64_t)split * a.rows_per_split;
  const int64_t r_end = r_begin + a.rows_per_split < a.R ? r_begin + a.rows_per_split : a.R;
  const int stages = (int)((r_end - r_begin + 15) >> 4);
  // (oh, ow) of this lane's pixel in the two x blocks it loads (pixel block b & 1 = wave & 1),
  // advanced by 16 rows per issued stage: no division in the loop
  int xoh, xow;
  {
    const int64_t row = r_begin + (wave & 1) * 8 + dpix;
    const int pix = (int)((row < a.R ? row : 0) % hw);
    xoh = pix / a.W;
    xow = pix - xoh * a.W;
  }

  auto issue = [&](int s) {
    unsigned char* base = lds + (s % kStages) * kStage;
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
      const int b = wave + 8 * i;                 // wave-uniform
      const bool is_dy = b < 2 * GN;
      const int cg = (is_dy ? b : b - 2 * GN) >> 1, pb = b & 1;
      const int64_t row = r_begin + (int64_t)s * 16 + pb * 8 + dpix;
      const uint4* src = g_zero_page;
      if (row < r_end) {
        if constexpr (PYR) {
          if (is_dy) {
            const int oh = xoh - ph[i & 1], ow = xow - pw[i & 1];
            if (pok[i & 1] && (unsigned)oh < (unsigned)a.H && (unsigned)ow < (unsigned)a.W)
              src = a.dy + ((size_t)(row - ph[i & 1] * a.W - pw[i & 1]) * n8 + (cg & 1) * 4 + dunit) * 2 + dpart;
          } else {
            src = a.x + ((size_t)row * k8 + kt * (TK / 8) + cg * 4 + dunit) * 2 + dpart;
          }
        } else if (is_dy) {
          src = a.dy + ((size_t)row * n8 + nt * (TN / 8) + cg * 4 + dunit) * 2 + dpart;
        } else {
          const int ih = xoh + dh, iw = xow + dw;
          if ((unsigned)ih < (unsigned)a.H && (unsigned)iw < (unsigned)a.W)
            src = a.x + ((size_t)(row + dh * a.W + dw) * k8 + kt * (TK / 8) + cg * 4 + dunit) * 2 + dpart;
        }
      }
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(base + b * 1024), 16, 0, 0);
    }
    xow += 16;                                    // stages are issued in order
    while (xow >= a.W) { xow -= a.W; ++xoh; }
    while (xoh >= a.H) xoh -= a.H;
  };

  // transpose-read role: group g = lane >> 4: channel half g & 1, pixel block g >> 1; lane lc:
  // pixel lc >> 2 (+4 by instruction offset), chunk lc & 3 = unit (lc >> 1) & 1, 8-byte half lc & 1
  const int g = lane >> 4, lc = lane & 15, th = g & 1;
  const unsigned tq = (unsigned)((lc >> 2) + 4 * th);                       // quad, pixels 0..3
  const unsigned toff_h = (unsigned)((g >> 1) * 1024) + (tq * 4 + (unsigned)((0 ^ th) * 2 + ((lc >> 1) & 1))) * 16 + 8 * (lc & 1);
  const unsigned toff_l = (unsigned)((g >> 1) * 1024) + (tq * 4 + (unsigned)((1 ^ th) * 2 + ((lc >> 1) & 1))) * 16 + 8 * (lc & 1);
  const unsigned lbase = (unsigned)(size_t)(lptr_t)lds;

  float16v acc[NI][NJ];
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  union Frag { short4v p[2]; half8 h; };
  auto tr_read = [&](unsigned addr_h, unsigned addr_l, Frag& fh, Frag& fl) {
    // pixels 0..3 / 4..7 of the lane's 8-pixel block (quads +8 = 512 bytes further)
    fh.p[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(size_t)addr_h);
    fh.p[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(size_t)(addr_h + 512));
    fl.p[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(size_t)addr_l);
    fl.p[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((trptr_t)(size_t)(addr_l + 512));
  };

  auto compute = [&](int s) {
    const unsigned sb = lbase + (unsigned)((s % kStages) * kStage);
    Frag bh[NJ], bl[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const unsigned blk = sb + (unsigned)((2 * GN + (wk * NJ + j) * 2) * 1024);   // x, 32-channel group wk*NJ+j
      tr_read(blk + toff_h, blk + toff_l, bh[j], bl[j]);
    }
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      Frag ah, al;
      const unsigned blk = sb + (unsigned)(((wn * NI + i) * 2) * 1024);          // dy, 32-channel group wn*NI+i
      tr_read(blk + toff_h, blk + toff_l, ah, al);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        acc[i][j] = mfma32(al.h, bh[j].h, acc[i][j]);
        acc[i][j] = mfma32(ah.h, bl[j].h, acc[i][j]);
        acc[i][j] = mfma32(ah.h, bh[j].h, acc[i][j]);
      }
    }
  };
  if constexpr (PAIR) {
    // one workgroup barrier per TWO stages (32 pixel rows): four ring slots, two stages issued at a time
    static_assert(!PAIR || kStages == 4, "pair mode: four slots");
    if (stages > 0) issue(0);
    if (stages > 1) issue(1);
    for (int s = 0; s < stages; s += 2) {
      wait_vmcnt(0);
      wg_barrier();
      if (s + 2 < stages) issue(s + 2);
      if (s + 3 < stages) issue(s + 3);
      compute(s);
      if (s + 1 < stages) compute(s + 1);
    }
  } else {
#pragma unroll
    for (int s = 0; s < kStages - 1; ++s)
      if (s < stages) issue(s);
    for (int s = 0; s < stages; ++s) {
      // stage s has landed: at most kStages - 2 younger stages of this wave's DMA are still in flight
      const int younger = stages - 1 - s < kStages - 2 ? stages - 1 - s : kStages - 2;
      wait_vmcnt(younger * NDMA);
      wg_barrier();
      if (s + kStages - 1 < stages) issue(s + kStages - 1);
      compute(s);
    }
  }

  float* out = a.partial + ((size_t)split * gridDim.x + tile) * (TN * TK);
#pragma unroll
  for (int i = 0; i < NI; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = (wn * NI + i) * 32 + acc_row(r, lane);
#pragma unroll
      for (int j = 0; j < NJ; ++j) out[n * TK + (wk * NJ + j) * 32 + (lane & 31)] = acc[i][j][r];
    }
}

// dw[n][tap][k] = (sum over splits of the partial tiles) / (S_dy * S_x)
// One thread per 4 consecutive k of one (tile, n) row: 16-byte loads, 8 splits in flight, the splits added left to right
// (the order of a plain loop: deterministic).  Round 3 read scalars, one row per 256-thread block: 1.1 TB/s, 61 us per
// convolution = 5.7 ms per training step.
__global__ __launch_bounds__(256) void conv_wgrad_reduce(const float* __restrict__ partial, int splits,
                                                         int tiles, int taps, int k_tiles, int K,
                                                         const float* __restrict__ dy_bound,
                                                         const float* __restrict__ x_bound,
                                                         float* __restrict__ dw, int TN, int TK,
                                                         int pyr_taps) {
  const int tk4 = TK >> 2;                               // float4 columns of a tile row
  const int64_t item = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t per_tile = (int64_t)TN * tk4;
  if (item >= per_tile * tiles) return;
  const int tile = (int)(item / per_tile);
  const int rem = (int)(item - (int64_t)tile * per_tile);
  const int n = rem / tk4, c4 = rem - n * tk4;
  const int tap = tile % taps, rest = tile / taps;
  const int kt = rest % k_tiles, nt = rest / k_tiles;
  const float mult = 1.0f / ((dy_bound ? pow2_scale(*dy_bound) : 1.f) * (x_bound ? pow2_scale(*x_bound) : 1.f));
  const float4v* p = reinterpret_cast<const float4v*>(partial + (size_t)tile * (TN * TK) + (size_t)n * TK) + c4;
  const size_t stride = (size_t)tiles * (TN * TK) / 4;   // float4 units between two splits
  float4v v = {0.f, 0.f, 0.f, 0.f};
  int s = 0;
  for (; s + 8 <= splits; s += 8) {
    float4v t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(p + (size_t)(s + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) v += t[u];
  }
  // (the remainder in written-out batches of 4, 2, 1 -- a loop with a run-time trip count waits for every load before
  // it issues the next: 7 splits -- 3x3 512 -> 512 -- were 7 round trips in a row; same left-to-right order)
  if (s + 4 <= splits) {
    float4v t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = __builtin_nontemporal_load(p + (size_t)(s + u) * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) v += t[u];
    s += 4;
  }
  if (s + 2 <= splits) {
    const float4v t0 = __builtin_nontemporal_load(p + (size_t)s * stride);
    const float4v t1 = __builtin_nontemporal_load(p + (size_t)(s + 1) * stride);
    v += t0;
    v += t1;
    s += 2;
  }
  if (s < splits) v += __builtin_nontemporal_load(p + (size_t)s * stride);
  v *= mult;
  if (pyr_taps) {
    // pyramid tiles: column n = (tap 4 * group + (n >> 6), channel n & 63); dw = [branch][64][9][K]
    const int t = 4 * tap + (n >> 6);
    if (t >= pyr_taps) return;
    const int br = t / 9, t9 = t - 9 * br;
    *reinterpret_cast<float4v*>(dw + ((size_t)(br * 64 + (n & 63)) * 9 + t9) * K + kt * TK + 4 * c4) = v;
    return;
  }
  *reinterpret_cast<float4v*>(dw + ((size_t)(nt * TN + n) * taps + tap) * K + kt * TK + 4 * c4) = v;
}

inline int wgrad_splits(int64_t R, int tiles) {
  const int64_t max_s = (R + 255) / 256;                  // at least 16 stages each
  int s = 256 / tiles;                                    // one workgroup per CU, one round
  if (tiles > 256) {
    // more tiles than CUs (e.g. 4096 -> 512 x 9 taps: 288): one split would leave a second round of 32
    // workgroups alone on the chip; take the smallest split count whose last round is (almost) full
    s = 1;
    double best = 1e9;
    for (int c = 1; c <= 16 && c <= max_s; ++c) {
      const int64_t wgs = (int64_t)tiles * c;
      const double waste = (double)((wgs + 255) / 256 * 256) / (double)wgs - 1.0;
      if (waste < best - 0.02) { best = waste; s = c; }
    }
  }
  if (s > max_s) s = (int)max_s;
  return s < 1 ? 1 : s;
}

// 8-wave workgroups (two row groups of 4 waves sharing one B tile, 320 x 256 outputs) are 3-6 % faster
// than two independent 160-row workgroups per CU in isolation (tools/bench_conv.py, SPML_CONV_RG=2), but
// their 108 KB of LDS keep the side-stream weight gradient off the CU: the training step is not faster
// (141.6 vs 141.0 ms), so the variant stays behind the tuning switch.
inline bool use_two_row_groups(int64_t, int, int) {
  const char* e = getenv("SPML_CONV_RG");
  return e && e[0] == '2';
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace spml

using namespace spml;

extern "C" int spml_hl8_from_f32(const float* x, int64_t rows, int C, float* bound, int compute_bound,
                                 void* out, void* stream) {
  if (!x || !out || rows <= 0 || C <= 0 || (compute_bound && !bound)) return SPML_ERR_INVALID_ARG;
  if ((C & 7) || !al16(x) || !al16(out)) return SPML_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = rows * C;
  if (compute_bound) {
    if (hipMemsetAsync(bound, 0, sizeof(float), s) != hipSuccess) return SPML_ERR_LAUNCH;
    const int grid = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256 + 1);
    hipLaunchKernelGGL(absmax_bound, dim3(grid), dim3(256), 0, s, x, n, reinterpret_cast<unsigned*>(bound));
  }
  const int64_t units = n >> 3;
  const int grid = (int)std::min<int64_t>(8192, (units + 255) / 256);
  hipLaunchKernelGGL(hl8_convert, dim3(grid), dim3(256), 0, s, x, units, (const float*)bound,
                     static_cast<uint4*>(out));
  return launch_status();
}

extern "C" int spml_hl8_weight_transposed_f32(const float* w, int Cout, int taps, int Cin,
                                              const float* bound, void* out, void* stream) {
  if (!w || !out || Cout <= 0 || Cin <= 0 || (taps != 1 && taps != 9)) return SPML_ERR_INVALID_ARG;
  if ((Cout & 7) || !al16(out)) return SPML_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)Cin * taps * (Cout >> 3);
  const int grid = (int)std::min<int64_t>(8192, (total + 255) / 256);
  hipLaunchKernelGGL(hl8_convert_wt, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, taps, Cin, bound,
                     static_cast<uint4*>(out), taps, 0);
  return launch_status();
}

extern "C" int spml_hl8_weight_transposed_into_f32(const float* w, int Cout, int taps, int Cin, const float* bound,
                                                   void* out, int taps_total, int tap_offset, void* stream) {
  if (!w || !out || Cout <= 0 || Cin <= 0 || (taps != 1 && taps != 9) || tap_offset < 0 ||
      tap_offset + taps > taps_total)
    return SPML_ERR_INVALID_ARG;
  if ((Cout & 7) || !al16(out)) return SPML_ERR_UNSUPPORTED;
  const int64_t total = (int64_t)Cin * taps * (Cout >> 3);
  const int grid = (int)std::min<int64_t>(8192, (total + 255) / 256);
  hipLaunchKernelGGL(hl8_convert_wt, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, Cout, taps, Cin, bound,
                     static_cast<uint4*>(out), taps_total, tap_offset);
  return launch_status();
}

extern "C" int spml_hl8_weight_set_f32(const float* const* w, const int* cout, const int* cin, const int* taps,
                                       int n, float* bounds, void* const* fwd, void* const* transposed,
                                       void* stream) {
  if (!w || !cout || !cin || !taps || !bounds || !fwd || !transposed || n < 1 || n > 4) return SPML_ERR_INVALID_ARG;
  WeightSet ws{};
  ws.n = n;
  ws.bound = bounds;
  int64_t acc = 0, largest = 0;
  for (int i = 0; i < n; ++i) {
    if (!w[i] || !fwd[i] || !transposed[i] || cout[i] <= 0 || cin[i] <= 0 || (taps[i] != 1 && taps[i] != 9))
      return SPML_ERR_INVALID_ARG;
    if ((cout[i] & 7) || (cin[i] & 7) || !al16(w[i]) || !al16(fwd[i]) || !al16(transposed[i])) return SPML_ERR_UNSUPPORTED;
    ws.w[i] = w[i];
    ws.fwd[i] = static_cast<uint4*>(fwd[i]);
    ws.tr[i] = static_cast<uint4*>(transposed[i]);
    ws.cout[i] = cout[i]; ws.cin[i] = cin[i]; ws.taps[i] = taps[i];
    ws.start[i] = acc;
    const int64_t units = (int64_t)cout[i] * cin[i] * taps[i] / 8;
    acc += units;
    largest = std::max(largest, units);
  }
  for (int i = n; i <= 4; ++i) ws.start[i] = acc;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(bounds, 0, 4 * sizeof(float), s) != hipSuccess) return SPML_ERR_LAUNCH;
  const int gx = (int)std::min<int64_t>(256, (largest * 2 + 255) / 256);
  hipLaunchKernelGGL(weightset_absmax, dim3(gx, n), dim3(256), 0, s, ws);
  const int grid = (int)std::min<int64_t>(4096, (acc + 255) / 256);
  hipLaunchKernelGGL(weightset_convert, dim3(grid), dim3(256), 0, s, ws);
  return launch_status();
}

extern "C" int spml_absmax_bound_f32(const float* x, int64_t n, float* bound, int zero_first, void* stream) {
  if (!x || !bound || n <= 0) return SPML_ERR_INVALID_ARG;
  if (!al16(x)) return SPML_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (zero_first && hipMemsetAsync(bound, 0, sizeof(float), s) != hipSuccess) return SPML_ERR_LAUNCH;
  const int grid = (int)std::min<int64_t>(2048, (n / 4 + 255) / 256 + 1);
  hipLaunchKernelGGL(absmax_bound, dim3(grid), dim3(256), 0, s, x, n, reinterpret_cast<unsigned*>(bound));
  return launch_status();
}

extern "C" int spml_conv_hl8_supported(int K, int N, int taps) {
  return (taps == 1 || taps == 9) && K > 0 && (K & 15) == 0 && N > 0 && (N & 63) == 0;
}

extern "C" int spml_conv_hl8_f32(const void* a, const float* a_bound, const void* b, const float* b_bound,
                                 const float* addend, const unsigned char* addend_mask, float* out, int n_img,
                                 int H, int W, int K, int N, int taps, int dilation, void* stream) {
  if (!a || !b || !out || n_img <= 0 || H <= 0 || W <= 0 || dilation < 1) return SPML_ERR_INVALID_ARG;
  if (!spml_conv_hl8_supported(K, N, taps) || !al16(a) || !al16(b) || !al16(out) || (addend && !al16(addend)))
    return SPML_ERR_UNSUPPORTED;
  ConvArgs c{};
  c.a = static_cast<const uint4*>(a);
  c.b = static_cast<const uint4*>(b);
  c.a_bound = a_bound;
  c.b_bound = b_bound;
  c.addend = addend;
  c.addend_mask = addend ? addend_mask : nullptr;
  c.out = out;
  c.R = (int64_t)n_img * H * W;
  c.H = H; c.W = W; c.K = K; c.N = N; c.taps = taps; c.dil = dilation;
  hipStream_t s = (hipStream_t)stream;
  // long reductions (K * taps >= 4096: the 3x3 convolutions of res5): the accumulation chain is cut every
  // 1024 k, so that its rounding error stays at the level of the fp32 library's split-K kernels
  // (profiles/r03_conv_accuracy.md: 1.1e-6 -> 2.6e-7 of max|out| at K = 4608).  It costs the 3-row-block
  // tile, 9-21 % of the convolution's time: at K = 2048 / 2304 (un-chunked 6.6e-7) it stays off by default
  if (chunk_long_reduction(c.K * c.taps)) return (N & 255) ? launch_conv_narrow<true>(c, s) : launch_conv<3, 3, 2, 1, true>(c, s);
  if (N & 255) return launch_conv_narrow<false>(c, s);
  if (use_two_row_groups(c.R, N, c.K * c.taps)) return launch_conv<5, 3, 2, 2>(c, s);
  switch (pick_rb(c.R, N)) {
    case 3: return launch_conv<3, 3, 2>(c, s);
    case 5: return launch_conv<5, 3, 2>(c, s);
    default: return launch_conv<4, 3, 2>(c, s);
  }
}

// rows per row tile of the launch spml_conv_hl8_f32 picks for this shape (0: a tile spans several row
// groups or the columns are narrow -- no fused statistics)
static int conv_stats_rows(int64_t R, int K, int N, int taps) {
  if ((N & 255) || !spml_conv_hl8_supported(K, N, taps) || use_two_row_groups(R, N, K * taps)) return 0;
  return 32 * (chunk_long_reduction((int64_t)K * taps) ? 3 : pick_rb(R, N));
}

extern "C" int spml_conv_hl8_stats_layout(int n_img, int H, int W, int K, int N, int taps, int* chunks,
                                          int* chunk_rows) {
  if (!chunks || !chunk_rows || n_img <= 0 || H <= 0 || W <= 0) return SPML_ERR_INVALID_ARG;
  const int64_t R = (int64_t)n_img * H * W;
  const int rows = conv_stats_rows(R, K, N, taps);
  if (!rows || (R + rows - 1) / rows > 4096) return SPML_ERR_UNSUPPORTED;
  *chunk_rows = rows;
  *chunks = (int)((R + rows - 1) / rows);
  return SPML_OK;
}

extern "C" int spml_conv_hl8_stats_f32(const void* a, const float* a_bound, const void* b, const float* b_bound,
                                       float* out, float* chunk_stats, float* zero_me, int n_img, int H, int W,
                                       int K, int N, int taps, int dilation, void* stream) {
  if (!a || !b || !out || !chunk_stats || n_img <= 0 || H <= 0 || W <= 0 || dilation < 1) return SPML_ERR_INVALID_ARG;
  int chunks = 0, rows = 0;
  if (spml_conv_hl8_stats_layout(n_img, H, W, K, N, taps, &chunks, &rows) != SPML_OK || !al16(a) || !al16(b) ||
      !al16(out))
    return SPML_ERR_UNSUPPORTED;
  ConvArgs c{};
  c.a = static_cast<const uint4*>(a);
  c.b = static_cast<const uint4*>(b);
  c.a_bound = a_bound;
  c.b_bound = b_bound;
  c.out = out;
  c.stats = chunk_stats;
  c.stats_zero = zero_me;
  c.R = (int64_t)n_img * H * W;
  c.H = H; c.W = W; c.K = K; c.N = N; c.taps = taps; c.dil = dilation;
  hipStream_t s = (hipStream_t)stream;
  if (chunk_long_reduction((int64_t)c.K * c.taps)) return launch_conv<3, 3, 2, 1, true>(c, s);
  switch (rows / 32) {
    case 3: return launch_conv<3, 3, 2>(c, s);
    case 5: return launch_conv<5, 3, 2>(c, s);
    default: return launch_conv<4, 3, 2>(c, s);
  }
}

extern "C" int spml_conv_hl8_affine_f32(const void* a, const float* a_bound, const void* b, const float* b_bound,
                                        const float* bias, const float* addend, int relu, float* out,
                                        float* out_bound, int n_img, int H, int W, int K, int N, int taps,
                                        int dilation, void* stream) {
  if (!a || !b || !out || n_img <= 0 || H <= 0 || W <= 0 || dilation < 1) return SPML_ERR_INVALID_ARG;
  if (!spml_conv_hl8_supported(K, N, taps) || !al16(a) || !al16(b) || !al16(out) || (addend && !al16(addend)))
    return SPML_ERR_UNSUPPORTED;
  ConvArgs c{};
  c.a = static_cast<const uint4*>(a);
  c.b = static_cast<const uint4*>(b);
  c.a_bound = a_bound;
  c.b_bound = b_bound;
  c.bias = bias;
  c.addend = addend;
  c.relu = relu;
  c.out = out;
  c.out_bound = out_bound;
  c.R = (int64_t)n_img * H * W;
  c.H = H; c.W = W; c.K = K; c.N = N; c.taps = taps; c.dil = dilation;
  hipStream_t s = (hipStream_t)stream;
  if (out_bound && hipMemsetAsync(out_bound, 0, sizeof(float), s) != hipSuccess) return SPML_ERR_LAUNCH;
  if (chunk_long_reduction(c.K * c.taps)) return (N & 255) ? launch_conv_narrow<true>(c, s) : launch_conv<3, 3, 2, 1, true>(c, s);
  if (N & 255) return launch_conv_narrow<false>(c, s);
  switch (pick_rb(c.R, N)) {
    case 3: return launch_conv<3, 3, 2>(c, s);
    case 5: return launch_conv<5, 3, 2>(c, s);
    default: return launch_conv<4, 3, 2>(c, s);
  }
}

extern "C" int spml_conv_hl8_pyramid_f32(const void* a, const float* a_bound, const void* b, const float* b_bound,
                                         const float* bias, const float* addend, float* out, int n_img, int H,
                                         int W, int K, int N, int groups, const int* dilations, void* stream) {
  if (!a || !b || !out || !dilations || n_img <= 0 || H <= 0 || W <= 0 || groups < 1 || groups > 4)
    return SPML_ERR_INVALID_ARG;
  if (!spml_conv_hl8_supported(K, N, 9) || !al16(a) || !al16(b) || !al16(out) || (addend && !al16(addend)))
    return SPML_ERR_UNSUPPORTED;
  ConvArgs c{};
  c.a = static_cast<const uint4*>(a);
  c.b = static_cast<const uint4*>(b);
  c.a_bound = a_bound;
  c.b_bound = b_bound;
  c.bias = bias;
  c.addend = addend;
  c.out = out;
  c.R = (int64_t)n_img * H * W;
  c.H = H; c.W = W; c.K = K; c.N = N; c.taps = 9 * groups; c.dil = dilations[0];
  for (int g = 0; g < 4; ++g) {
    c.dils[g] = dilations[g < groups ? g : 0];
    if (c.dils[g] < 1) return SPML_ERR_INVALID_ARG;
  }
  hipStream_t s = (hipStream_t)stream;
  // narrow heads (the 64-d embedding: 264 row tiles of 256 pixels -- one workgroup per CU, nothing to cover its
  // barriers): one workgroup per (tile, dilation group), the four partial tiles are added with fp32 atomics
  if ((N & 255) && groups > 1 && !addend && !deterministic_mode()) c.tap_groups = groups;     // (partial tiles meet through fp32 atomics)
  if (N & 255) return chunk_long_reduction((int64_t)K * c.taps) ? launch_conv_narrow<true>(c, s) : launch_conv_narrow<false>(c, s);
  if (chunk_long_reduction((int64_t)K * c.taps)) return launch_conv<3, 3, 2, 1, true>(c, s);     // chunked accumulation
  if (use_two_row_groups(c.R, N, c.K * c.taps)) return launch_conv<5, 3, 2, 2>(c, s);
  switch (pick_rb(c.R, N)) {
    case 3: return launch_conv<3, 3, 2>(c, s);
    case 5: return launch_conv<5, 3, 2>(c, s);
    default: return launch_conv<4, 3, 2>(c, s);
  }
}

// ---------------------------------------------------------------------------------------
// narrow pyramid head, forward, as ONE plain GEMM + a gather
// ---------------------------------------------------------------------------------------
// out[p][n] = bias[n] + sum_tap x[p + shift(tap)] . w_tap[n]  with N = 64: a tile of one output pixel block is 64
// columns wide and x is streamed once per tap (36 x 554 MB).  With z[r][tap * N + n] = x[r] . w_tap[n] -- a 1x1
// convolution with 36 * N "channels": x streamed once per 256 columns = 9 times -- the head's output is
//   out[p][n] = bias[n] + sum_tap z[p + shift(tap)][tap * N + n]      (taps whose shifted pixel leaves the image: 0),
// which this kernel gathers: N / 4 lanes per pixel (16-byte loads of 4 N bytes contiguous per tap), the taps added in
// order 0 .. taps-1 (deterministic, unlike the atomics of the tap-group launch).
__global__ __launch_bounds__(256) void conv_tap_gather(const float* __restrict__ z, const float* __restrict__ bias,
                                                       float* __restrict__ out, int64_t R, int H, int W, int N,
                                                       int taps, unsigned dils) {
  const int lanes = N >> 2;                               // threads per pixel (a power of two <= 256)
  const int per_block = 256 / lanes;
  const int64_t p = (int64_t)blockIdx.x * per_block + threadIdx.x / lanes;
  const int c4 = threadIdx.x % lanes;
  if (p >= R) return;
  const int hw = H * W;
  const int pix = (int)(p % hw), oh = pix / W, ow = pix - oh * W;
  const int64_t zrow = (int64_t)taps * N;
  float4v acc = bias ? *reinterpret_cast<const float4v*>(bias + 4 * c4) : float4v{0.f, 0.f, 0.f, 0.f};
  for (int t0 = 0; t0 < taps; t0 += 9) {                   // one branch: its nine loads in flight together
    const int d = (int)((dils >> (8 * (t0 / 9))) & 255u);
    float4v v[9];
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) {
      const int dh = (t9 / 3 - 1) * d, dw = (t9 % 3 - 1) * d;
      const bool ok = (unsigned)(oh + dh) < (unsigned)H && (unsigned)(ow + dw) < (unsigned)W;
      v[t9] = ok ? __builtin_nontemporal_load(reinterpret_cast<const float4v*>(
                       z + (p + (int64_t)dh * W + dw) * zrow + (int64_t)(t0 + t9) * N) + c4)
                 : float4v{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int t9 = 0; t9 < 9; ++t9) acc += v[t9];
  }
  *reinterpret_cast<float4v*>(out + p * N + 4 * c4) = acc;
}

extern "C" int spml_conv_tap_gather_f32(const float* z, const float* bias, float* out, int n_img, int H, int W,
                                        int N, int branches, const int* dilations, void* stream) {
  if (!z || !out || !dilations || n_img <= 0 || H <= 0 || W <= 0) return SPML_ERR_INVALID_ARG;
  if (N < 16 || N > 1024 || (N & (N - 1)) || branches < 1 || branches > 4 || !al16(z) || !al16(out) ||
      (bias && !al16(bias)))
    return SPML_ERR_UNSUPPORTED;
  unsigned dils = 0;
  for (int b = 0; b < branches; ++b) {
    if (dilations[b] < 1 || dilations[b] > 255) return SPML_ERR_INVALID_ARG;
    dils |= (unsigned)dilations[b] << (8 * b);
  }
  const int64_t R = (int64_t)n_img * H * W;
  const int per_block = 256 / (N >> 2);
  hipLaunchKernelGGL(conv_tap_gather, dim3((unsigned)((R + per_block - 1) / per_block)), dim3(256), 0,
                     (hipStream_t)stream, z, bias, out, R, H, W, N, 9 * branches, dils);
  return launch_status();
}

extern "C" int spml_conv_wgrad_hl8_supported(int K, int N, int taps) {
  return (taps == 1 || taps == 9) && K > 0 && (K & 127) == 0 && N > 0 && (N & 127) == 0;
}

// tile widths: 256 where the channel count allows it, 128 otherwise
static inline int wgrad_tile(int channels) { return (channels & 255) == 0 ? 256 : 128; }

extern "C" size_t spml_conv_wgrad_workspace_bytes(int n_img, int H, int W, int K, int N, int taps) {
  if (!spml_conv_wgrad_hl8_supported(K, N, taps) || n_img <= 0 || H <= 0 || W <= 0) return 0;
  const int tn = wgrad_tile(N), tk = wgrad_tile(K);
  const int tiles = (N / tn) * (K / tk) * taps;
  return (size_t)wgrad_splits((int64_t)n_img * H * W, tiles) * tiles * tn * tk * sizeof(float);
}

extern "C" int spml_conv_wgrad_hl8_f32(const void* dy, const float* dy_bound, const void* x,
                                       const float* x_bound, float* dw, int n_img, int H, int W, int K,
                                       int N, int taps, int dilation, void* ws, size_t ws_bytes,
                                       void* stream) {
  if (!dy || !x || !dw || n_img <= 0 || H <= 0 || W <= 0 || dilation < 1) return SPML_ERR_INVALID_ARG;
  if (!spml_conv_wgrad_hl8_supported(K, N, taps) || !al16(dy) || !al16(x) || !al16(dw)) return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_conv_wgrad_workspace_bytes(n_img, H, W, K, N, taps)) return SPML_ERR_WORKSPACE;
  WgradArgs a{};
  a.dy = static_cast<const uint4*>(dy);
  a.x = static_cast<const uint4*>(x);
  a.partial = static_cast<float*>(ws);
  a.R = (int64_t)n_img * H * W;
  a.H = H; a.W = W; a.K = K; a.N = N; a.taps = taps; a.dil = dilation;
  const int tn = wgrad_tile(N), tk = wgrad_tile(K);
  a.k_tiles = K / tk;
  const int tiles = (N / tn) * a.k_tiles * taps;
  a.splits = wgrad_splits(a.R, tiles);
  a.rows_per_split = (int)(((a.R + a.splits - 1) / a.splits + 15) / 16 * 16);
  hipStream_t s = (hipStream_t)stream;
  // 256 x 256 tiles: four ring slots and one workgroup barrier per TWO stages (32 pixel rows) -- 5-12 % faster than a
  // barrier per stage with three slots on every res4 / res5 shape (tools/bench_conv.py; SPML_WGRAD_STAGES=3 selects the
  // latter, 2 / 4 / 5 its other ring depths, which all measure the same: the stage time is matrix-pipe + barrier time,
  // not load latency).  The 128-wide tiles keep three slots.
  // (the experiment switch only exists for the 256 x 256 tiles, and only with the depths instantiated below: anything
  // else takes the default -- a launch that matches no instantiation would leave conv_wgrad_reduce summing an
  // unwritten workspace)
  int nst = 42;
  if (tn == 256 && tk == 256) {
    if (const char* e = getenv("SPML_WGRAD_STAGES")) {
      const int v = atoi(e);
      if (v == 2 || v == 3 || v == 4 || v == 5) nst = v;
    }
  } else {
    nst = 3;
  }
  bool launched = false;
#define SPML_WGRAD(TN_, TK_, ST_)                                                                              \
  if (tn == TN_ && tk == TK_ && nst == ST_) {                                                                  \
    const int lds = ST_ * 2 * (TN_ / 32 + TK_ / 32) * 1024;                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad<TN_, TK_, ST_>),                        \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                \
    hipLaunchKernelGGL((conv_wgrad<TN_, TK_, ST_>), dim3(tiles, a.splits), dim3(512), lds, s, a);              \
    launched = true;                                                                                           \
  }
  SPML_WGRAD(256, 256, 3) SPML_WGRAD(256, 128, 3) SPML_WGRAD(128, 256, 3) SPML_WGRAD(128, 128, 3)
  SPML_WGRAD(256, 256, 4) SPML_WGRAD(256, 256, 5) SPML_WGRAD(256, 256, 2)
#undef SPML_WGRAD
  if (tn == 256 && tk == 256 && nst == 42) {
    const int lds = 4 * 2 * (256 / 32 + 256 / 32) * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad<256, 256, 4, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL((conv_wgrad<256, 256, 4, true>), dim3(tiles, a.splits), dim3(512), lds, s, a);
    launched = true;
  }
  if (!launched) return SPML_ERR_UNSUPPORTED;
  const int64_t items = (int64_t)tiles * tn * (tk / 4);
  hipLaunchKernelGGL(conv_wgrad_reduce, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (const float*)a.partial,
                     a.splits, tiles, taps, a.k_tiles, K, dy_bound, x_bound, dw, tn, tk, 0);
  return launch_status();
}

// The weight gradients of the pyramid head's 64-channel branches (conv_wgrad<..., PYR>): dw fp32
// [branches][64][9][K] = the channels-last storage of each branch's [64, K, 3, 3] gradient, one after the other.
static inline int pyr_groups(int branches) { return (9 * branches + 3) / 4; }

static inline int pyr_splits(int64_t R, int tiles) {
  // one 128-KB workgroup per CU: the smallest split count whose last round of 256 is (almost) full
  const int64_t max_s = (R + 255) / 256;
  int s = 1;
  double best = 1e9;
  for (int c = 1; c <= 16 && c <= max_s; ++c) {
    const int64_t wgs = (int64_t)tiles * c;
    const double waste = (double)((wgs + 255) / 256 * 256) / (double)wgs - 1.0;
    if (waste < best - 0.02) { best = waste; s = c; }
  }
  return s;
}

extern "C" int spml_conv_wgrad_pyramid_hl8_supported(int K, int N, int branches) {
  return N == 64 && K > 0 && (K & 255) == 0 && branches >= 1 && branches <= 4;
}

extern "C" size_t spml_conv_wgrad_pyramid_workspace_bytes(int n_img, int H, int W, int K, int N, int branches) {
  if (!spml_conv_wgrad_pyramid_hl8_supported(K, N, branches) || n_img <= 0 || H <= 0 || W <= 0) return 0;
  const int tiles = pyr_groups(branches) * (K / 256);
  return (size_t)pyr_splits((int64_t)n_img * H * W, tiles) * tiles * 256 * 256 * sizeof(float);
}

extern "C" int spml_conv_wgrad_pyramid_hl8_f32(const void* dy, const float* dy_bound, const void* x,
                                               const float* x_bound, float* dw, int n_img, int H, int W, int K,
                                               int N, int branches, const int* dilations, void* ws,
                                               size_t ws_bytes, void* stream) {
  if (!dy || !x || !dw || !dilations || n_img <= 0 || H <= 0 || W <= 0) return SPML_ERR_INVALID_ARG;
  if (!spml_conv_wgrad_pyramid_hl8_supported(K, N, branches) || !al16(dy) || !al16(x) || !al16(dw))
    return SPML_ERR_UNSUPPORTED;
  if (!ws || ws_bytes < spml_conv_wgrad_pyramid_workspace_bytes(n_img, H, W, K, N, branches)) return SPML_ERR_WORKSPACE;
  WgradArgs a{};
  for (int b = 0; b < branches; ++b) {
    if (dilations[b] < 1 || dilations[b] > 255) return SPML_ERR_INVALID_ARG;
    a.pyr_dils |= (unsigned)dilations[b] << (8 * b);
  }
  a.dy = static_cast<const uint4*>(dy);
  a.x = static_cast<const uint4*>(x);
  a.partial = static_cast<float*>(ws);
  a.R = (int64_t)n_img * H * W;
  a.H = H; a.W = W; a.K = K; a.N = N; a.dil = 1;
  a.taps = pyr_groups(branches);                 // (the tile index decodes as (group of four taps, k tile))
  a.pyr_taps = 9 * branches;
  a.k_tiles = K / 256;
  const int tiles = a.taps * a.k_tiles;
  a.splits = pyr_splits(a.R, tiles);
  a.rows_per_split = (int)(((a.R + a.splits - 1) / a.splits + 15) / 16 * 16);
  hipStream_t s = (hipStream_t)stream;
  const int lds = 4 * 2 * (256 / 32 + 256 / 32) * 1024;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad<256, 256, 4, true, true>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL((conv_wgrad<256, 256, 4, true, true>), dim3(tiles, a.splits), dim3(512), lds, s, a);
  const int64_t items = (int64_t)tiles * 256 * (256 / 4);
  hipLaunchKernelGGL(conv_wgrad_reduce, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, s, (const float*)a.partial,
                     a.splits, tiles, a.taps, a.k_tiles, K, dy_bound, x_bound, dw, 256, 256, a.pyr_taps);
  return launch_status();
}
