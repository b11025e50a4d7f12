// What shader clock does a short kernel get?  A dependent chain of integer adds (4 cycles per wave64 VALU operation on
// CDNA) timed on the constant 100 MHz clock: (a) after the GPU sat idle for a while, (b) ~200 us into a burst of
// kernels, (c) after a long busy kernel.   hipcc --offload-arch=gfx950 -O3 clock_probe.cpp -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <vector>
__global__ void chain(unsigned* out, unsigned long long* t, int n) {
  unsigned v = threadIdx.x;
  const unsigned long long t0 = wall_clock64();
  for (int i = 0; i < n; ++i) { v = v * 3u + 1u; __asm__ volatile("" : "+v"(v)); }
  const unsigned long long t1 = wall_clock64();
  if (threadIdx.x == 0) { t[blockIdx.x] = t1 - t0; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
__global__ void busy(unsigned* out, int n) {
  unsigned v = threadIdx.x + blockIdx.x;
  for (int i = 0; i < n; ++i) { v = v * 3u + 1u; __asm__ volatile("" : "+v"(v)); }
  out[(blockIdx.x * blockDim.x + threadIdx.x) & 1023] = v;
}
int main() {
  unsigned* out; unsigned long long* t;
  hipMalloc(&out, 1 << 22); hipMalloc(&t, 4096);
  const int n = 20000;  // v_mul_lo + v_add per iteration, dependent
  auto probe = [&](const char* what) {
    hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, out, t, n);
    hipDeviceSynchronize();
    unsigned long long ticks; hipMemcpy(&ticks, t, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.1f us for %d dependent mul+add pairs: %.2f ns per pair\n", what, ticks / 100.0, n, ticks * 10.0 / n);
  };
  probe("first kernel of the process");
  usleep(200000);
  probe("after 200 ms idle");
  usleep(2000);
  probe("after 2 ms idle");
  usleep(300);
  probe("after 0.3 ms idle");
  probe("back to back");
  hipLaunchKernelGGL(busy, dim3(2048), dim3(256), 0, 0, out, 2000000 / 8);
  probe("behind a busy kernel on the whole GPU");
  for (int k = 0; k < 5; ++k) probe("back to back again");
  usleep(300);
  probe("after 0.3 ms idle");
  return 0;
}
