// Does gfx950's hardware fp32 -> bf16 conversion (v_cvt_pk_bf16_f32, what `(__bf16)f` compiles to) give the same bits as the library's
// software round-to-nearest-even (qp_common.h: f32_to_bf16_bits) for EVERY fp32 pattern?  Exhaustive: 2^32 inputs.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/probe_cvt_bf16 tools/probe/probe_cvt_bf16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ unsigned short sw(float f) {
  unsigned u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__device__ __forceinline__ unsigned short hw(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }

__global__ void cmp(unsigned long long* counts, unsigned* examples) {
  const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
  unsigned long long bad = 0, bad_nan = 0;
  for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const float f = __uint_as_float((unsigned)i);
    const unsigned short a = sw(f), b = hw(f);
    if (a != b) {
      const bool isnan = ((unsigned)i & 0x7fffffffu) > 0x7f800000u;
      if (isnan) ++bad_nan; else { ++bad; }
      unsigned slot = atomicAdd(&examples[0], 1u);
      if (slot < 8) { examples[1 + 3 * slot] = (unsigned)i; examples[2 + 3 * slot] = a; examples[3 + 3 * slot] = b; }
    }
  }
  atomicAdd(&counts[0], bad); atomicAdd(&counts[1], bad_nan);
}

int main() {
  unsigned long long* c; unsigned* ex;
  hipMalloc(&c, 16); hipMalloc(&ex, 4 * 32); hipMemset(c, 0, 16); hipMemset(ex, 0, 4 * 32);
  cmp<<<4096, 256>>>(c, ex);
  unsigned long long hc[2]; unsigned he[32];
  hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost); hipMemcpy(he, ex, 4 * 32, hipMemcpyDeviceToHost);
  printf("fp32 -> bf16, hardware (v_cvt_pk_bf16_f32) vs software RNE over all 2^32 patterns: %llu finite/inf mismatches, %llu NaN-pattern mismatches\n", hc[0], hc[1]);
  for (unsigned i = 0; i < (he[0] < 8 ? he[0] : 8); ++i) printf("  in %08x  sw %04x  hw %04x\n", he[1 + 3 * i], he[2 + 3 * i], he[3 + 3 * i]);
  return 0;
}
