// Hardware-layout probe for gfx950 (MI355X).  Not part of the product: it pins the
// register layouts the attention kernel relies on (MFMA operand/result maps,
// ds_read_b64_tr_b16 gather, permlane32_swap) by checking hypotheses on the device.
//   hipcc --offload-arch=gfx950 -O2 probe_layouts.hip -o probe_layouts && ./probe_layouts
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <cstdint>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

static inline uint16_t f2bf(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)(u >> 16); }

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

// C[32][32] = A[32][16] * B[16][32]; hypothesis: A lane l -> row l&31, k=(l>>5)*8+e ; B lane l -> col l&31, k=(l>>5)*8+e
// C lane l reg r -> col l&31, row (r&3)+8*(r>>2)+4*(l>>5)
__global__ void k_mfma32(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)A[(l & 31) * 16 + (l >> 5) * 8 + e];
    b[e] = (short)B[((l >> 5) * 8 + e) * 32 + (l & 31)];
  }
  f32x16 c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    C[row * 32 + (l & 31)] = c[r];
  }
}

// C[16][16] = A[16][32]*B[32][16]; A lane l -> row l&15, k=(l>>4)*8+e; C lane l reg r -> col l&15,row (l>>4)*4+r
__global__ void k_mfma16(const uint16_t* A, const uint16_t* B, float* C) {
  int l = threadIdx.x;
  bf16x8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (short)A[(l & 15) * 32 + (l >> 4) * 8 + e];
    b[e] = (short)B[((l >> 4) * 8 + e) * 16 + (l & 15)];
  }
  f32x4 c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// tr read: LDS[i] = i (u16). lane address = lane*stride_bytes. Dump 4 values per lane.
__global__ void k_tr(uint16_t* out, int mode) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[8192];
  int l = threadIdx.x;
  for (int i = l; i < 8192; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  int idx;  // element index (u16 units), must be multiple of 4
  if (mode == 0) idx = l * 32;                        // each lane its own 64-B row
  else idx = ((l & 15) >> 2) * 64 + (l & 3) * 4 + (l >> 4) * 1024;  // [4 rows of 64 elems][16 cols] per 16-lane group
  s16x4 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + idx));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}

__global__ void k_swap(uint32_t* out) {
  int l = threadIdx.x;
  uint32_t a = 1000 + l, b = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[l * 2] = r[0];
  out[l * 2 + 1] = r[1];
}

__global__ void k_cvt(const float* in, uint32_t* out) {
  int l = threadIdx.x;
  uint32_t r;
  float lo = in[2 * l], hi = in[2 * l + 1];
  asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  out[l] = r;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s arch=%s CUs=%d lds/block=%zu clock=%d MHz mem=%.1f GB\n", p.name, p.gcnArchName, p.multiProcessorCount,
         p.sharedMemPerBlock, p.clockRate / 1000, p.totalGlobalMem / 1e9);
  srand(1);
  {  // mfma 32x32x16
    std::vector<uint16_t> A(32 * 16), B(16 * 32); std::vector<float> Af(32 * 16), Bf(16 * 32), C(32 * 32), R(32 * 32, 0.f);
    for (int i = 0; i < 32 * 16; ++i) { Af[i] = (float)(rand() % 17 - 8); A[i] = f2bf(Af[i]); }
    for (int i = 0; i < 16 * 32; ++i) { Bf[i] = (float)(rand() % 13 - 6); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) for (int k = 0; k < 16; ++k) R[i * 32 + j] += Af[i * 16 + k] * Bf[k * 32 + j];
    uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, A.size() * 2)); CK(hipMalloc(&dB, B.size() * 2)); CK(hipMalloc(&dC, C.size() * 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice));
    k_mfma32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; ++i) bad += (C[i] != R[i]);
    printf("mfma_32x32x16_bf16 layout hypothesis: %s (mismatches=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  {  // mfma 16x16x32
    std::vector<uint16_t> A(16 * 32), B(32 * 16); std::vector<float> Af(16 * 32), Bf(32 * 16), C(256), R(256, 0.f);
    for (int i = 0; i < 512; ++i) { Af[i] = (float)(rand() % 17 - 8); A[i] = f2bf(Af[i]); Bf[i] = (float)(rand() % 13 - 6); B[i] = f2bf(Bf[i]); }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) for (int k = 0; k < 32; ++k) R[i * 16 + j] += Af[i * 32 + k] * Bf[k * 16 + j];
    uint16_t *dA, *dB; float* dC; CK(hipMalloc(&dA, 1024)); CK(hipMalloc(&dB, 1024)); CK(hipMalloc(&dC, 1024));
    CK(hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice));
    k_mfma16<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(C.data(), dC, 1024, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 256; ++i) bad += (C[i] != R[i]);
    printf("mfma_16x16x32_bf16 layout hypothesis: %s (mismatches=%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  for (int mode = 0; mode < 2; ++mode) {
    uint16_t* d; CK(hipMalloc(&d, 64 * 4 * 2)); std::vector<uint16_t> h(256);
    k_tr<<<1, 64>>>(d, mode); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    printf("ds_read_b64_tr_b16 mode %d (lane: 4 values as u16 LDS element index):\n", mode);
    for (int l = 0; l < 64; ++l) {
      printf("  lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        int v = h[l * 4 + j];
        if (mode == 0) printf(" [src_lane %2d elem %d]", v / 32, v % 32); else printf(" [grp %d row %d col %2d]", v / 1024, (v % 1024) / 64, v % 64);
      }
      printf("\n");
    }
    // hypothesis check mode 0: lane l elem j <- lane ((l&~15) + 4*j + ((l&15)>>2)), element (l&3)
    if (mode == 0) { int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int exp = ((l & ~15) + 4 * j + ((l & 15) >> 2)) * 32 + (l & 3); bad += (h[l * 4 + j] != exp); }
      printf("tr16_b64 hypothesis (result[l][j] = src lane (l&~15)+4j+((l&15)>>2), elem l&3): %s (%d)\n", bad ? "FAIL" : "PASS", bad); }
    else { int bad = 0; for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int exp = (l >> 4) * 1024 + j * 64 + (l & 15); bad += (h[l * 4 + j] != exp); }
      printf("tr16_b64 4x16 block hypothesis (lane gets column l&15 of rows 0..3): %s (%d)\n", bad ? "FAIL" : "PASS", bad); }
  }
  {
    uint32_t* d; CK(hipMalloc(&d, 512)); std::vector<uint32_t> h(128);
    k_swap<<<1, 64>>>(d); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost));
    // hypothesis: r0 (from a): lanes<32 keep a, lanes>=32 get b of lane-32 ; r1 (from b): lanes<32 get a of lane+32; lanes>=32 keep b
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      uint32_t e0 = l < 32 ? 1000 + l : 2000 + (l - 32), e1 = l < 32 ? 1000 + (l + 32) : 2000 + l;
      bad += (h[2 * l] != e0) + (h[2 * l + 1] != e1);
    }
    printf("permlane32_swap hypothesis: %s (%d)  lane0=(%u,%u) lane32=(%u,%u)\n", bad ? "FAIL" : "PASS", bad, h[0], h[1], h[64], h[65]);
  }
  {
    std::vector<float> in(128); for (int i = 0; i < 128; ++i) in[i] = 1.0f + i * 0.00390625f * 0.37f;
    float* di; uint32_t* d; CK(hipMalloc(&di, 512)); CK(hipMalloc(&d, 256)); CK(hipMemcpy(di, in.data(), 512, hipMemcpyHostToDevice));
    std::vector<uint32_t> h(64);
    k_cvt<<<1, 64>>>(di, d); CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, 256, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
      auto rne = [](float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (uint16_t)(u >> 16); };
      uint32_t e = (uint32_t)rne(in[2 * l]) | ((uint32_t)rne(in[2 * l + 1]) << 16);
      bad += (h[l] != e);
    }
    printf("v_cvt_pk_bf16_f32 (lo in [15:0], RNE): %s (%d)\n", bad ? "FAIL" : "PASS", bad);
  }
  return 0;
}
