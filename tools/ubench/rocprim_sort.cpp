// Calibration only (not linked into the product): what rocPRIM's device radix sort -- AMD's own tuned onesweep -- takes
// for the key/payload shapes of the hot path, next to k_sort_pass.  u64 keys [code | index], u32 payload, sort bits
// [begin, end).   hipcc --offload-arch=gfx950 -O3 -o rocprim_sort rocprim_sort.cpp && ./rocprim_sort
#include <cstring>
#include <string.h>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <random>
#include <vector>

static void run(size_t n, int ibits, int cbits) {
  std::vector<uint64_t> h(n);
  std::vector<uint32_t> v(n);
  std::mt19937_64 rng(42);
  for (size_t i = 0; i < n; ++i) { h[i] = ((rng() & ((1ull << cbits) - 1)) << ibits) | i; v[i] = (uint32_t)i; }
  uint64_t *ki, *ko; uint32_t *vi, *vo;
  hipMalloc(&ki, n * 8); hipMalloc(&ko, n * 8); hipMalloc(&vi, n * 4); hipMalloc(&vo, n * 4);
  hipMemcpy(ki, h.data(), n * 8, hipMemcpyHostToDevice);
  hipMemcpy(vi, v.data(), n * 4, hipMemcpyHostToDevice);
  size_t tmp_bytes = 0;
  rocprim::radix_sort_pairs(nullptr, tmp_bytes, ki, ko, vi, vo, n, ibits, ibits + cbits);
  void* tmp; hipMalloc(&tmp, tmp_bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int w = 0; w < 3; ++w) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, ibits, ibits + cbits);
  hipDeviceSynchronize();
  const int reps = 20;
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) rocprim::radix_sort_pairs(tmp, tmp_bytes, ki, ko, vi, vo, n, ibits, ibits + cbits);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms = 0; hipEventElapsedTime(&ms, a, b);
  printf("rocprim radix_sort_pairs  n=%zu  key u64 bits [%d,%d) + u32 payload: %.1f us per sort  (%.2f Gkeys/s, %d code bits)\n", n, ibits, ibits + cbits,
         1e3 * ms / reps, n / (ms / reps) / 1e6, cbits);
  // keys only (the packed key carries the index): what the 8-byte-per-key sort costs
  rocprim::radix_sort_keys(nullptr, tmp_bytes, ki, ko, n, ibits, ibits + cbits);
  void* tmp2; hipMalloc(&tmp2, tmp_bytes);
  for (int w = 0; w < 3; ++w) rocprim::radix_sort_keys(tmp2, tmp_bytes, ki, ko, n, ibits, ibits + cbits);
  hipEventRecord(a);
  for (int r = 0; r < reps; ++r) rocprim::radix_sort_keys(tmp2, tmp_bytes, ki, ko, n, ibits, ibits + cbits);
  hipEventRecord(b); hipEventSynchronize(b);
  hipEventElapsedTime(&ms, a, b);
  printf("rocprim radix_sort_keys   n=%zu  key u64 bits [%d,%d):               %.1f us per sort\n", n, ibits, ibits + cbits, 1e3 * ms / reps);
  hipFree(ki); hipFree(ko); hipFree(vi); hipFree(vo); hipFree(tmp); hipFree(tmp2);
}

int main() {
  run(1000000, 20, 36);    // cfg2: 36 varying code bits, 20 index bits
  run(1000000, 20, 33);
  run(10000000, 24, 39);   // cfg4
  run(10000000, 24, 42);
  return 0;
}
