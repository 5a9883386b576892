// Micro-benchmark: issue rate of v_mfma_f32_16x16x32_f16 for a lone wave per SIMD (the kmeans_pass64 setting),
// nine independent accumulation chains in rotation, and what the instructions of the E-/M-step loops cost between them.
//   mode 0: MFMAs alone, accumulators in accumulation registers (AGPR form), A/B architectural
//   mode 1: accumulators architectural (VGPR form)
//   mode 2: mode 0 + A operand from accumulation registers
//   mode 3: mode 0 + 2 x ds_read_b128 per 9 MFMAs (issued one group ahead, counted wait)
//   mode 4: mode 0 + 8 x v_accvgpr_read_b32 per 9 MFMAs
//   mode 5: mode 0 + 1 x global_load_lds_dwordx4 (1 KB LDS-DMA) per 9 MFMAs
//   mode 6: mode 0 + 4 x ds_read_b64_tr_b16 per 6 MFMAs (M-step shape)
//   mode 7: 3 chains only (each accumulator every 3rd MFMA)
//   mode 8: 1 chain (dependent MFMAs back to back)
// Build: hipcc --offload-arch=gfx950 -O3 mfma16_rate.hip -o mfma16_rate.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#define MA(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 a[" #d ":" #d "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #d ":" #d "+3]")
#define MV(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 v[" #d ":" #d "+3], v[" #a ":" #a "+3], v[" #b ":" #b "+3], v[" #d ":" #d "+3]")
#define MAA(d, a, b) asm volatile("v_mfma_f32_16x16x32_f16 a[" #d ":" #d "+3], a[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #d ":" #d "+3]")
template <int MODE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) void probe(float* out, int iters, unsigned long long* stamps,
                                                                                       const float* src) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
               "a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35",
               "a64","a65","a66","a67","a68","a69","a70","a71",
               "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",
               "v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95",
               "v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115",
               "v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143",
               "v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159",
               "v160","v161","v162","v163","v255");
  if (iters < 0) {      // (negative count: operands = pseudo-random f16 in (-2, 2) instead of whatever the registers hold)
    iters = -iters;
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u;
#define RNDV(r) h = h * 1664525u + 1013904223u; asm volatile("v_mov_b32 v" #r ", %0" :: "v"((h & 0xbfffbfffu) | 0x20002000u));
    RNDV(64) RNDV(65) RNDV(66) RNDV(67) RNDV(68) RNDV(69) RNDV(70) RNDV(71) RNDV(72) RNDV(73) RNDV(74) RNDV(75)
    RNDV(76) RNDV(77) RNDV(78) RNDV(79) RNDV(80) RNDV(81) RNDV(82) RNDV(83) RNDV(84) RNDV(85) RNDV(86) RNDV(87)
    RNDV(88) RNDV(89) RNDV(90) RNDV(91) RNDV(92) RNDV(93) RNDV(94) RNDV(95)
  } else {
#define ZERV(r) asm volatile("v_mov_b32 v" #r ", 0");
    ZERV(64) ZERV(65) ZERV(66) ZERV(67) ZERV(68) ZERV(69) ZERV(70) ZERV(71) ZERV(72) ZERV(73) ZERV(74) ZERV(75)
    ZERV(76) ZERV(77) ZERV(78) ZERV(79) ZERV(80) ZERV(81) ZERV(82) ZERV(83) ZERV(84) ZERV(85) ZERV(86) ZERV(87)
    ZERV(88) ZERV(89) ZERV(90) ZERV(91) ZERV(92) ZERV(93) ZERV(94) ZERV(95)
  }
  const unsigned laddr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds + 16u * (threadIdx.x & 63);
  const float* gsrc = src + (threadIdx.x & 63) * 4;
  const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (MODE == 3) {
        asm volatile("ds_read_b128 v[100:103], %0\n\tds_read_b128 v[104:107], %0 offset:1024" :: "v"(laddr));
        asm volatile("s_waitcnt lgkmcnt(2)");
      }
      if (MODE == 6) {
        asm volatile("ds_read_b64_tr_b16 v[100:101], %0\n\tds_read_b64_tr_b16 v[102:103], %0 offset:512\n\t"
                     "ds_read_b64_tr_b16 v[104:105], %0 offset:1024\n\tds_read_b64_tr_b16 v[106:107], %0 offset:1536" :: "v"(laddr));
        asm volatile("s_waitcnt lgkmcnt(4)");
      }
      if (MODE == 5)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gsrc + g * 256),
                                         (__attribute__((address_space(3))) void*)(lds + 4096 + g * 1024), 16, 0, 0);
      if (MODE == 1) {
        MV(128, 64, 80); MV(132, 68, 80); MV(136, 72, 80); MV(140, 64, 84); MV(144, 68, 84); MV(148, 72, 84);
        MV(152, 76, 80); MV(156, 88, 80); MV(160, 92, 80);
      } else if (MODE == 2) {
        MA(0, 64, 80); MA(4, 68, 80); MA(8, 72, 80); MA(12, 64, 84); MA(16, 68, 84); MA(20, 72, 84);
        MAA(24, 64, 80); MAA(28, 68, 80); MAA(32, 64, 80);
      } else if (MODE == 4) {
        MA(0, 64, 80); MA(4, 68, 80); MA(8, 72, 80);
        asm volatile("v_accvgpr_read_b32 v108, a64\n\tv_accvgpr_read_b32 v109, a65\n\tv_accvgpr_read_b32 v110, a66\n\tv_accvgpr_read_b32 v111, a67");
        MA(12, 64, 84); MA(16, 108, 84); MA(20, 72, 84);
        asm volatile("v_accvgpr_read_b32 v112, a68\n\tv_accvgpr_read_b32 v113, a69\n\tv_accvgpr_read_b32 v114, a70\n\tv_accvgpr_read_b32 v115, a71");
        MA(24, 76, 80); MA(28, 112, 80); MA(32, 92, 80);
      } else if (MODE == 6) {
        MA(0, 64, 80); MA(4, 68, 80); MA(8, 72, 80); MA(0, 64, 84); MA(4, 68, 84); MA(8, 72, 84);
      } else if (MODE == 9 || MODE == 10 || MODE == 11) {
        // the E-step loop as hipcc emits it: a spilled A operand is re-read into the registers of the B fragment the
        // three MFMAs before it used (write after read), and used as A one MFMA later (read after write)
        MA(0, 64, 84); MA(4, 68, 84); MA(8, 72, 84);
        if (MODE == 9) asm volatile("v_accvgpr_read_b32 v84, a64\n\tv_accvgpr_read_b32 v85, a65\n\tv_accvgpr_read_b32 v86, a66\n\tv_accvgpr_read_b32 v87, a67");
        if (MODE == 10) asm volatile("v_accvgpr_read_b32 v108, a64\n\tv_accvgpr_read_b32 v109, a65\n\tv_accvgpr_read_b32 v110, a66\n\tv_accvgpr_read_b32 v111, a67");
        MA(12, 64, 80);
        if (MODE == 9) MA(16, 84, 80); else MA(16, 108, 80);
        if (MODE == 9) asm volatile("v_accvgpr_read_b32 v84, a68\n\tv_accvgpr_read_b32 v85, a69\n\tv_accvgpr_read_b32 v86, a70\n\tv_accvgpr_read_b32 v87, a71");
        if (MODE == 10) asm volatile("v_accvgpr_read_b32 v112, a68\n\tv_accvgpr_read_b32 v113, a69\n\tv_accvgpr_read_b32 v114, a70\n\tv_accvgpr_read_b32 v115, a71");
        MA(20, 68, 80); MA(24, 72, 80);
        if (MODE == 9) MA(28, 84, 80); else MA(28, 112, 80);
        MA(32, 92, 80);
        if (MODE == 11) {    // the B fragments of the step after next land in the registers the MFMAs above read
          asm volatile("ds_read_b128 v[80:83], %0\n\tds_read_b128 v[84:87], %0 offset:1024" :: "v"(laddr));
          asm volatile("s_waitcnt lgkmcnt(2)");
        }
      } else if (MODE == 7) {
        MA(0, 64, 80); MA(4, 68, 80); MA(8, 72, 80); MA(0, 64, 84); MA(4, 68, 84); MA(8, 72, 84);
        MA(0, 76, 80); MA(4, 88, 80); MA(8, 92, 80);
      } else if (MODE == 8) {
        MA(0, 64, 80); MA(0, 68, 80); MA(0, 72, 80); MA(0, 64, 84); MA(0, 68, 84); MA(0, 72, 84);
        MA(0, 76, 80); MA(0, 88, 80); MA(0, 92, 80);
      } else {
        MA(0, 64, 80); MA(4, 68, 80); MA(8, 72, 80); MA(12, 64, 84); MA(16, 68, 84); MA(20, 72, 84);
        MA(24, 76, 80); MA(28, 88, 80); MA(32, 92, 80);
      }
    }
    if (MODE == 5) asm volatile("s_waitcnt vmcnt(0)");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = t1 - t0; }
  float v, w;
  asm volatile("v_accvgpr_read_b32 %0, a0\n\tv_mov_b32 %1, v128" : "=v"(v), "=v"(w));
  out[blockIdx.x * 256 + threadIdx.x] = v + w;
}
template <int MODE>
void run(const char* what, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  float* src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
  unsigned long long* stamps; hipMalloc(&stamps, 512 * 8);
  unsigned long long host[512];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 65536, 0, out, iters, stamps, src);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 65536, 0, out, iters, stamps, src);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(host, stamps, 512 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)host[2 * i]; rt += (double)host[2 * i + 1]; }
  const int per = MODE == 6 ? 24 : 36;
  const int n = iters < 0 ? -iters : iters;
  printf("mode %d %s %-62s %8.1f us  %6.1f shader cycles per MFMA at %.0f MHz\n", MODE, iters < 0 ? "random" : "zeros ", what, ms * 1e3,
         cyc / 256 / n / per, cyc / rt * 100.0);
  hipFree(out); hipFree(stamps); hipFree(src);
}
int main() {
  const int iters = 2000;
  run<0>("9 chains, AGPR accumulators", iters);
  run<1>("9 chains, VGPR accumulators", iters);
  run<2>("9 chains, 3 of 9 with A from AGPRs", iters);
  run<3>("+ 2 ds_read_b128 per 9", iters);
  run<4>("+ 8 v_accvgpr_read per 9", iters);
  run<5>("+ 1 LDS-DMA (1 KB) per 9", iters);
  run<6>("4 ds_read_b64_tr_b16 per 6, 3 chains twice", iters);
  run<7>("3 chains (accumulator reused every 3rd MFMA)", iters);
  run<8>("1 chain (dependent back to back)", iters);
  run<9>("E-loop: accvgpr_read INTO the B registers just read, used as A next", iters);
  run<10>("E-loop: the same reads into other registers", iters);
  run<11>("E-loop: ds_read_b128 into the B registers just read", iters);
  run<0>("9 chains, AGPR accumulators", -iters);
  run<1>("9 chains, VGPR accumulators", -iters);
  run<3>("+ 2 ds_read_b128 per 9", -iters);
  run<4>("+ 8 v_accvgpr_read per 9", -iters);
  run<7>("3 chains (accumulator reused every 3rd MFMA)", -iters);
  return 0;
}
