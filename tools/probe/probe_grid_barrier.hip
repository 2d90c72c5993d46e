// How long does a grid-wide barrier (global atomic counter + spin) take on MI355X, vs the ~4 us of a kernel boundary inside a graph?
// Decides whether a persistent whole-layer decode kernel could beat six dependent launches per layer.
// build: hipcc --offload-arch=gfx950 -O3 -o probe_grid_barrier probe_grid_barrier.hip ; run: ./probe_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ bool grid_barrier(unsigned* ctr, unsigned nblocks, unsigned& epoch) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __threadfence();
    const unsigned target = (++epoch) * nblocks;
    atomicAdd(ctr, 1u);
    long spins = 0;
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
      if (++spins > 200000000L) { ok = false; break; }           // never hang the box
    __threadfence();
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k(unsigned* ctr, int iters, float* sink, int* err) {
  unsigned epoch = 0;
  float acc = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    acc = acc * 1.0001f + 1.f;
    if (!grid_barrier(ctr, gridDim.x, epoch)) { if (threadIdx.x == 0) *err = 1; break; }
  }
  if (acc == 12345.f) *sink = acc;
}

int main() {
  unsigned* ctr; float* sink; int* err;
  hipMalloc(&ctr, 4); hipMalloc(&sink, 4); hipMalloc(&err, 4);
  for (int blocks : {64, 256, 512, 1024}) {
    int maxb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k, 256, 0);
    if (blocks > maxb * 256) continue;                            // all workgroups must be co-resident
    for (int iters : {200, 2000}) {
      hipMemset(ctr, 0, 4); hipMemset(err, 0, 4);
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      k<<<blocks, 256>>>(ctr, iters, sink, err);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      int e = 0; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost);
      printf("blocks=%d iters=%d: %.3f ms total, %.3f us per barrier%s\n", blocks, iters, ms, ms * 1e3f / iters, e ? "  (TIMEOUT)" : "");
    }
  }
  return 0;
}
