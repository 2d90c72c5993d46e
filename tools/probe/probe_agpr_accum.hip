// Feasibility probe (compile-only: hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only): MFMAs written as inline asm with the
// "+a" constraint keep their accumulators in AGPRs across a loop with NO v_accvgpr_* moves inside it, while "=v" MFMAs deliver
// scores to VGPRs.  This is what the 64-rows-per-wave / one-wave-per-SIMD attention form needs (DESIGN.md sections 6 and 9):
// hipcc's own allocation of builtin MFMAs shuffled ~570 registers per tile through v_accvgpr there.
#include <hip/hip_runtime.h>
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
// O accumulators pinned to AGPRs through the "a" constraint; S produced into VGPRs
__global__ __launch_bounds__(256, 1) void k(const bf16x8_t* a, const bf16x8_t* b, float* out, int iters) {
  f32x16_t o[8];
  for (int i = 0; i < 8; ++i) o[i] = (f32x16_t){0};
  bf16x8_t av = a[threadIdx.x], bv = b[threadIdx.x];
  float acc = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x16_t s;
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(s) : "v"(av), "v"(bv));
    asm volatile("s_nop 15\ns_nop 7" ::: );
    for (int r = 0; r < 16; ++r) acc += s[r];
#pragma unroll
    for (int i = 0; i < 8; ++i)
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(o[i]) : "v"(av), "v"(bv));
    bv[0] = (__bf16)acc;
  }
  float t = acc;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) t += o[i][r];
  out[threadIdx.x] = t;
}
