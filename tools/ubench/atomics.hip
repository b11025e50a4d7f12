// microbenchmark: throughput of device-scope global atomicAdd (u32) scattered over a table,
// and latency of dependent kernel boundaries.  Build: hipcc --offload-arch=gfx950 -O3 atomics.hip -o atomics
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
__global__ void k_scatter_add(uint32_t* table, uint32_t table_size, uint32_t n, uint32_t seed) {
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t h = (i + seed) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
  atomicAdd(&table[h % table_size], 1u);
}
__global__ void k_scatter_add_runs(uint32_t* table, uint32_t table_size, uint32_t n, uint32_t seed) {
  // like the radix case: 8 consecutive lanes hit the same 2 KB row (dest tile), random column
  uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint32_t run = i >> 3;
  uint32_t h = (run + seed) * 2654435761u; h ^= h >> 15;
  uint32_t row = h % (table_size / 512);
  uint32_t g = (i + seed) * 2246822519u; g ^= g >> 13;
  atomicAdd(&table[row * 512 + (g & 511)], 1u);
}
__global__ void k_empty(uint32_t* p) { if (p == nullptr) p[0] = 1; }
int main() {
  const uint32_t n = 1u << 20;
  uint32_t* table;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (uint32_t tsz : {1024u, 8192u, 125440u, 1u << 20}) {
    hipMalloc(&table, tsz * 4); hipMemset(table, 0, tsz * 4);
    for (int variant = 0; variant < 2; ++variant) {
      float best = 1e9;
      for (int it = 0; it < 10; ++it) {
        hipEventRecord(e0);
        if (variant == 0) hipLaunchKernelGGL(k_scatter_add, dim3(n / 1024), dim3(1024), 0, 0, table, tsz, n, it);
        else hipLaunchKernelGGL(k_scatter_add_runs, dim3(n / 1024), dim3(1024), 0, 0, table, tsz, n, it);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("table %8u words, %s: 1M atomicAdd in %.2f us\n", tsz, variant ? "runs-of-8" : "random   ", best * 1e3);
    }
    hipFree(table);
  }
  // kernel boundary
  { float best = 1e9; for (int it = 0; it < 10; ++it) { hipEventRecord(e0); for (int k = 0; k < 20; ++k) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, (uint32_t*)0x10);
    hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
    printf("20 empty dependent kernels: %.2f us each\n", best * 1e3 / 20); }
  return 0;
}
