// Micro-benchmark: issue cadence of v_mfma_f32_32x32x16_f16 by operand form, one wave per SIMD, four
// independent accumulation chains issued in rotation (what nll_bwd_de3 does), 24 MFMAs per iteration:
//   mode 0: accumulators in accumulation registers, A and B architectural            (the compiler's usual form)
//   mode 1: accumulators architectural, A architectural, B in accumulation registers (the recompute of de3)
//   mode 2: rotation of the two forms (mode 1, mode 0, mode 1, mode 0)               (a step of de3)
//   mode 3: as 2, every MFMA with its own A / B registers                            (operand variety of a step)
//   mode 4: as 0 with TWO chains only (dependent MFMAs 64 cycles apart)
//   mode 5: as 0 with ONE chain (back-to-back dependent)
// Build: hipcc --offload-arch=gfx950 -O3 mfma_forms.hip -o mfma_forms
#include <hip/hip_runtime.h>
#include <stdio.h>
#define M_AVV(d, a, b) asm volatile("v_mfma_f32_32x32x16_f16 a[" #d ":" #d "+15], v[" #a ":" #a "+3], v[" #b ":" #b "+3], a[" #d ":" #d "+15]")
#define M_VVA(d, a, b) asm volatile("v_mfma_f32_32x32x16_f16 v[" #d ":" #d "+15], v[" #a ":" #a "+3], a[" #b ":" #b "+3], v[" #d ":" #d "+15]")
template <int MODE>
__global__ __launch_bounds__(256, 1) __attribute__((amdgpu_num_vgpr(64))) void probe(float* out, int iters) {
  asm volatile("v_accvgpr_write_b32 a0, 0" ::: "a0","a1","a2","a3","a4","a5","a6","a7","a8","a9","a10","a11","a12","a13","a14","a15","a16","a17","a18","a19","a20","a21","a22","a23","a24","a25","a26","a27","a28","a29","a30","a31","a32","a33","a34","a35","a36","a37","a38","a39","a40","a41","a42","a43","a44","a45","a46","a47","a48","a49","a50","a51","a52","a53","a54","a55","a56","a57","a58","a59","a60","a61","a62","a63","a64","a65","a66","a67","a68","a69","a70","a71","a72","a73","a74","a75","a76","a77","a78","a79","a80","a81","a82","a83","a84","a85","a86","a87","a88","a89","a90","a91","a92","a93","a94","a95","a96","a97","a98","a99","a100","a101","a102","a103","a104","a105","a106","a107","a108","a109","a110","a111","a112","a113","a114","a115","a116","a117","a118","a119","a120","a121","a122","a123","a124","a125","a126","a127","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127","v128","v129","v130","v131","v132","v133","v134","v135","v136","v137","v138","v139","v140","v141","v142","v143","v144","v145","v146","v147","v148","v149","v150","v151","v152","v153","v154","v155","v156","v157","v158","v159","v160","v161","v162","v163","v164","v165","v166","v167","v168","v169","v170","v171","v172","v173","v174","v175","v176","v177","v178","v179","v180","v181","v182","v183","v184","v185","v186","v187","v188","v189","v190","v191");
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 6; ++g) {
      if (MODE == 0) { M_AVV(0, 64, 68); M_AVV(16, 72, 76); M_AVV(32, 64, 68); M_AVV(48, 72, 76); }
      if (MODE == 1) { M_VVA(128, 64, 64); M_VVA(144, 72, 68); M_VVA(160, 64, 64); M_VVA(176, 72, 68); }
      if (MODE == 2) { M_VVA(128, 64, 64); M_AVV(0, 72, 76); M_VVA(144, 64, 68); M_AVV(16, 72, 76); }
      if (MODE == 3) {
        if (g == 0) { M_VVA(128, 64, 64); M_AVV(0, 68, 72); M_VVA(144, 76, 68); M_AVV(16, 80, 84); }
        if (g == 1) { M_VVA(128, 88, 72); M_AVV(0, 92, 96); M_VVA(144, 100, 76); M_AVV(16, 104, 108); }
        if (g == 2) { M_VVA(128, 112, 80); M_AVV(0, 116, 120); M_VVA(144, 124, 84); M_AVV(16, 64, 68); }
        if (g == 3) { M_VVA(128, 72, 88); M_AVV(0, 76, 80); M_VVA(144, 84, 92); M_AVV(16, 88, 92); }
        if (g == 4) { M_VVA(128, 96, 96); M_AVV(0, 100, 104); M_VVA(144, 108, 100); M_AVV(16, 112, 116); }
        if (g == 5) { M_VVA(128, 120, 104); M_AVV(0, 124, 64); M_VVA(144, 68, 108); M_AVV(16, 72, 76); }
      }
      if (MODE == 4) { M_AVV(0, 64, 68); M_AVV(16, 72, 76); M_AVV(0, 64, 68); M_AVV(16, 72, 76); }
      if (MODE == 5) { M_AVV(0, 64, 68); M_AVV(0, 72, 76); M_AVV(0, 64, 68); M_AVV(0, 72, 76); }
    }
  }
  asm volatile("s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15");
  float v;
  asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v));
  out[blockIdx.x * 256 + threadIdx.x] = v;
}
template <int MODE>
void run(const char* what, int iters) {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL(probe<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("mode %d  %-70s %8.1f us  %.1f ns per MFMA\n", MODE, what, ms * 1e3, ms * 1e6 / iters / 24);
  hipFree(out);
}
int main() {
  const int iters = 4000;
  run<0>("4 chains, acc in AGPR, A/B in VGPR", iters);
  run<1>("4 chains, acc in VGPR, A VGPR, B AGPR", iters);
  run<2>("4 chains, the two forms in rotation", iters);
  run<3>("as 2, a different A/B register set per MFMA", iters);
  run<4>("2 chains (dependent MFMAs 64 cycles apart)", iters);
  run<5>("1 chain (back-to-back dependent)", iters);
  return 0;
}
