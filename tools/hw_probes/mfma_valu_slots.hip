// Micro-benchmark: what a lone wave per SIMD pays for vector instructions placed between its own MFMAs.
// 48 v_mfma_f32_32x32x16_f16 per iteration in four independent chains issued in rotation (accumulators
// architectural, B in accumulation registers: the forward step of nll_fwd3), NV vector instructions after each:
//   mode 0: no vector work                    mode 1: 3 x v_add_f32 per MFMA
//   mode 2: 1 x v_exp_f32 + 2 x v_add_f32     mode 3: 2 x v_exp_f32 + 1 x v_add_f32
//   mode 4: 3 x v_exp_f32                     mode 5: 6 x v_add_f32
//   mode 6: vector work of mode 1 alone       mode 7: vector work of mode 4 alone
//   mode 8: as 2, the v_exp_f32 reading a register of the OTHER accumulator set (as the kernel does)
//   mode 9: 1 x v_add_f32                     mode 10: 2 x v_add_f32
//   mode 11: v_exp_f32, v_add_f32, then an add that CONSUMES the exp (one instruction between: the form of a naive
//            exp -> accumulate epilogue)      mode 12: the consumer one MFMA slot later (4 instructions)
//   mode 13: the consumer two MFMA slots later (8 instructions)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_valu_slots.hip -o mfma_valu_slots
#include <hip/hip_runtime.h>
#include <stdio.h>
#define M_VVA(d, a, b) asm volatile("v_mfma_f32_32x32x16_f16 v[" #d ":" #d "+15], v[" #a ":" #a "+3], a[" #b ":" #b "+3], v[" #d ":" #d "+15]")
#define ADD(x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(one))
#define EXP(x) asm volatile("v_exp_f32 %0, %1" : "=v"(x) : "v"(src))
#define DEP(x, y) asm volatile("v_add_f32 %0, %0, %1" : "+v"(x) : "v"(y))
#define EXPZ(x, r) asm volatile("v_exp_f32 %0, v" #r : "=v"(x))
template <int MODE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) void probe(float* out, int iters, unsigned long long* stamps) {
  asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15",
               "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79",
               "v128","v143","v144","v159","v160","v175","v176","v191","v192","v207","v255");
  float x0 = threadIdx.x, x1 = 1.f, x2 = 2.f, x3 = 3.f, x4 = 4.f, x5 = 5.f, one = 1.f, src = 0.5f;
  asm volatile("" : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(one), "+v"(src));
  float e[3] = {0.f, 0.f, 0.f};
  asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]));
  constexpr bool MAT = MODE != 6 && MODE != 7;
  const unsigned long long c0 = __builtin_readcyclecounter(), t0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 48; ++g) {
      if (MAT) {
        if ((g & 3) == 0) M_VVA(128, 64, 0);
        if ((g & 3) == 1) M_VVA(144, 68, 4);
        if ((g & 3) == 2) M_VVA(160, 72, 8);
        if ((g & 3) == 3) M_VVA(176, 76, 12);
      }
      if (MODE == 1 || MODE == 6) { ADD(x0); ADD(x1); ADD(x2); }
      if (MODE == 2) { EXP(x3); ADD(x0); ADD(x1); }
      if (MODE == 3) { EXP(x3); ADD(x0); EXP(x4); }
      if (MODE == 4 || MODE == 7) { EXP(x3); EXP(x4); EXP(x5); }
      if (MODE == 5) { ADD(x0); ADD(x1); ADD(x2); ADD(x3); ADD(x4); ADD(x5); }
      if (MODE == 8) { EXPZ(x3, 200); ADD(x0); ADD(x1); }
      if (MODE == 11) { EXP(e[0]); ADD(x0); DEP(x1, e[0]); }
      if (MODE == 12) { EXP(e[g & 1]); ADD(x0); DEP(x1, e[(g + 1) & 1]); }
      if (MODE == 13) { EXP(e[g % 3]); ADD(x0); DEP(x1, e[(g + 1) % 3]); }
      if (MODE == 9) { ADD(x0); }
      if (MODE == 10) { ADD(x0); ADD(x1); }
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  const unsigned long long c1 = __builtin_readcyclecounter(), t1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = t1 - t0; }
  float v;
  asm volatile("v_mov_b32 %0, v128" : "=v"(v));
  out[blockIdx.x * 256 + threadIdx.x] = v + x0 + x1 + x2 + x3 + x4 + x5 + e[0] + e[1] + e[2];
}
template <int MODE>
void run(const char* what, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  unsigned long long* stamps; hipMalloc(&stamps, 512 * 8);
  unsigned long long host[512];
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters, stamps);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters, stamps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(host, stamps, 512 * 8, hipMemcpyDeviceToHost);
  double cyc = 0, rt = 0;
  for (int i = 0; i < 256; ++i) { cyc += (double)host[2 * i]; rt += (double)host[2 * i + 1]; }
  // s_memtime counts shader cycles, s_memrealtime a constant 100 MHz
  printf("mode %2d  %-52s %8.1f us  %6.1f ns per MFMA slot  %6.1f shader cycles per slot at %.0f MHz\n", MODE, what, ms * 1e3,
         ms * 1e6 / iters / 48, cyc / 256 / iters / 48, cyc / rt * 100.0);
  hipFree(out); hipFree(stamps);
}
int main() {
  const int iters = 2000;
  run<0>("MFMA only", iters);
  run<9>("MFMA + 1 add", iters);
  run<10>("MFMA + 2 add", iters);
  run<1>("MFMA + 3 add", iters);
  run<5>("MFMA + 6 add", iters);
  run<2>("MFMA + 1 exp + 2 add", iters);
  run<8>("MFMA + 1 exp (other accumulator set) + 2 add", iters);
  run<3>("MFMA + 2 exp + 1 add", iters);
  run<4>("MFMA + 3 exp", iters);
  run<11>("MFMA + exp, add, consumer of the exp", iters);
  run<12>("MFMA + exp, add, consumer of the previous slot's exp", iters);
  run<13>("MFMA + exp, add, consumer of the exp two slots back", iters);
  run<6>("3 add alone", iters);
  run<7>("3 exp alone", iters);
  return 0;
}
