// MFMA / VALU co-issue probe for gfx950 (not part of the product).  One 512-thread workgroup per CU = 2 waves per SIMD
// (waves w and w+4 share SIMD w%4, checked through HW_ID).  Role A waves run only v_mfma_f32_32x32x16_bf16, role B waves
// only VALU (fma or exp).  Modes: 0 = A alone, 1 = B alone, 2 = A and B together on the same SIMDs,
// 3 = ONE wave per SIMD alternating 1 MFMA + K VALU in its own stream (K = argv[2]).
//   hipcc --offload-arch=gfx950 -O2 probe_overlap.hip -o probe_overlap && ./probe_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int ITERS = 4000;

template <int MODE, int VOP, int K>
__global__ __launch_bounds__(512, 1) void probe(float* out, long long* cyc) {
  __shared__ char pad[100 * 1024];            // one workgroup per CU
  const int wave = threadIdx.x >> 6;
  const bool roleA = wave < 4;
  if (threadIdx.x == 9999) pad[0] = 1;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
  bf16x8_t a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(threadIdx.x * 0.001f + e); b[e] = (__bf16)(e * 0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.01f + i;
  const float c1 = 0.999f, c2 = 0.001f;
  long long t0 = __builtin_readcyclecounter();
  if (MODE == 3 || MODE == 4) {
    // hand-written stream: 4 x { MFMA ; K VALU } per iteration; MODE 3: one wave per SIMD, MODE 4: two waves per SIMD
    if (roleA || MODE == 4) {
      for (int it = 0; it < ITERS; ++it) {
#define VOPS1(R) VOP == 0 ? "v_fma_f32 " R ", " R ", %12, %13\n" : "v_exp_f32 " R ", " R "\n"
#define MF(ACC) "v_mfma_f32_32x32x16_bf16 " ACC ", %14, %15, " ACC "\n"
        if (K == 0) asm volatile(MF("%0") MF("%1") MF("%2") MF("%3")
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
            : "v"(c1), "v"(c2), "v"(a), "v"(b));
#define STREAM(OPS) asm volatile(MF("%0") OPS MF("%1") OPS MF("%2") OPS MF("%3") OPS \
            : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]) \
            : "v"(c1), "v"(c2), "v"(a), "v"(b))
#define F(R) "v_fma_f32 " R ", " R ", %12, %13\n"
#define E(R) "v_exp_f32 " R ", " R "\n"
        if (VOP == 0) {
          if (K == 2) STREAM(F("%4") F("%5"));
          if (K == 4) STREAM(F("%4") F("%5") F("%6") F("%7"));
          if (K == 5) STREAM(F("%4") F("%5") F("%6") F("%7") F("%8"));
          if (K == 6) STREAM(F("%4") F("%5") F("%6") F("%7") F("%8") F("%9"));
          if (K == 8) STREAM(F("%4") F("%5") F("%6") F("%7") F("%8") F("%9") F("%10") F("%11"));
          if (K == 12) STREAM(F("%4") F("%5") F("%6") F("%7") F("%8") F("%9") F("%10") F("%11") F("%4") F("%5") F("%6") F("%7"));
        } else {
          if (K == 1) STREAM(E("%4"));
          if (K == 2) STREAM(E("%4") E("%5"));
          if (K == 3) STREAM(E("%4") E("%5") E("%6"));
          if (K == 4) STREAM(E("%4") E("%5") E("%6") E("%7"));
          if (K == 5) STREAM(F("%4") F("%5") F("%6") F("%7") E("%8"));         // 4 fma + 1 exp
          if (K == 7) STREAM(F("%4") F("%5") F("%6") F("%7") F("%9") F("%10") E("%8"));   // 6 fma + 1 exp
        }
      }
    }
  } else {
    const bool runA = (MODE == 0 || MODE == 2) && roleA;
    const bool runB = (MODE == 1 || MODE == 2) && !roleA;
    if (runA) {
      for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[j], 0, 0, 0);
      }
    }
    if (runB) {
      for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
          if (VOP == 0) v[k & 7] = __builtin_fmaf(v[k & 7], c1, c2);
          else v[k & 7] = __builtin_amdgcn_exp2f(v[k & 7]);
        }
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int VOP, int K>
void run(const char* name) {
  float* out; long long* cyc;
  CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  probe<MODE, VOP, K><<<256, 512>>>(out, cyc);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  probe<MODE, VOP, K><<<256, 512>>>(out, cyc);
  CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[8]; CK(hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost));
  // per-iteration figures: A does 4 MFMA / iter, B does 32 VALU / iter
  printf("%-44s %.3f ms | wave0 (A) %.1f clk/iter, wave4 (B) %.1f clk/iter\n", name, ms, (double)h[0] / ITERS, (double)h[4] / ITERS);
  CK(hipFree(out)); CK(hipFree(cyc));
}

int main() {
  run<0, 0, 0>("A alone: 4 MFMA/iter");
  run<1, 0, 0>("B alone: 32 v_fma/iter");
  run<2, 0, 0>("A + B(fma) on the same SIMD");
  run<1, 1, 0>("B alone: 32 v_exp/iter");
  run<2, 1, 0>("A + B(exp) on the same SIMD");
  run<3, 0, 0>("1 wave/SIMD asm: MFMA only");
  run<3, 0, 2>("1 wave/SIMD asm: MFMA + 2 fma");
  run<3, 0, 4>("1 wave/SIMD asm: MFMA + 4 fma");
  run<3, 0, 5>("1 wave/SIMD asm: MFMA + 5 fma");
  run<3, 0, 6>("1 wave/SIMD asm: MFMA + 6 fma");
  run<3, 0, 8>("1 wave/SIMD asm: MFMA + 8 fma");
  run<3, 0, 12>("1 wave/SIMD asm: MFMA + 12 fma");
  run<3, 1, 1>("1 wave/SIMD asm: MFMA + 1 exp");
  run<3, 1, 2>("1 wave/SIMD asm: MFMA + 2 exp");
  run<3, 1, 3>("1 wave/SIMD asm: MFMA + 3 exp");
  run<3, 1, 4>("1 wave/SIMD asm: MFMA + 4 exp");
  run<3, 1, 5>("1 wave/SIMD asm: MFMA + 4 fma + 1 exp");
  run<3, 1, 7>("1 wave/SIMD asm: MFMA + 6 fma + 1 exp");
  run<4, 0, 0>("2 waves/SIMD asm: MFMA only");
  run<4, 0, 4>("2 waves/SIMD asm: MFMA + 4 fma");
  run<4, 0, 6>("2 waves/SIMD asm: MFMA + 6 fma");
  run<4, 0, 8>("2 waves/SIMD asm: MFMA + 8 fma");
  run<4, 0, 12>("2 waves/SIMD asm: MFMA + 12 fma");
  run<4, 1, 5>("2 waves/SIMD asm: MFMA + 4 fma + 1 exp");
  run<4, 1, 7>("2 waves/SIMD asm: MFMA + 6 fma + 1 exp");
  return 0;
}
