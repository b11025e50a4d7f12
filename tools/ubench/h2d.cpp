// h2d.cpp -- how a 32 MB frame gets from host memory into HBM fastest (numbers behind DESIGN.md, host-input pipeline):
// pageable hipMemcpy, pinned hipMemcpyAsync, hipHostRegister + copy + unregister, CPU memcpy into a pinned stage
// (1..T threads side by side), several pinned copies in flight on several streams.
//   hipcc -O2 -pthread tools/ubench/h2d.cpp -o tools/ubench/h2d && tools/ubench/h2d
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <chrono>
#include <thread>
#include <vector>

typedef std::chrono::steady_clock Clock;
static double ms_since(Clock::time_point t0) { return std::chrono::duration<double, std::milli>(Clock::now() - t0).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  const size_t bytes = 32u << 20;
  const int reps = 10;
  void* dev[8];
  for (auto& d : dev) CK(hipMalloc(&d, bytes));
  char* pageable = (char*)aligned_alloc(4096, bytes);
  memset(pageable, 1, bytes);
  char* pinned[8];
  for (auto& p : pinned) { CK(hipHostMalloc((void**)&p, bytes, hipHostMallocDefault)); memset(p, 2, bytes); }
  hipStream_t st[8];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));

  CK(hipMemcpy(dev[0], pageable, bytes, hipMemcpyHostToDevice));  // warm up
  auto t0 = Clock::now();
  for (int r = 0; r < reps; ++r) CK(hipMemcpy(dev[0], pageable, bytes, hipMemcpyHostToDevice));
  double ms = ms_since(t0) / reps;
  printf("pageable hipMemcpy              %7.3f ms  %6.1f GB/s\n", ms, bytes / ms / 1e6);

  CK(hipMemcpyAsync(dev[0], pinned[0], bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0]));
  t0 = Clock::now();
  for (int r = 0; r < reps; ++r) { CK(hipMemcpyAsync(dev[0], pinned[0], bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0])); }
  ms = ms_since(t0) / reps;
  printf("pinned hipMemcpyAsync + sync    %7.3f ms  %6.1f GB/s\n", ms, bytes / ms / 1e6);

  for (int k : {2, 4, 8}) {
    t0 = Clock::now();
    for (int r = 0; r < reps; ++r) {
      for (int i = 0; i < k; ++i) CK(hipMemcpyAsync(dev[i], pinned[i], bytes, hipMemcpyHostToDevice, st[i]));
      for (int i = 0; i < k; ++i) CK(hipStreamSynchronize(st[i]));
    }
    ms = ms_since(t0) / reps / k;
    printf("%d pinned copies in flight        %7.3f ms per frame  %6.1f GB/s\n", k, ms, bytes / ms / 1e6);
  }

  double reg = 0, cp = 0, unreg = 0;
  for (int r = 0; r < reps; ++r) {
    t0 = Clock::now();
    CK(hipHostRegister(pageable, bytes, hipHostRegisterDefault));
    reg += ms_since(t0);
    t0 = Clock::now();
    CK(hipMemcpyAsync(dev[0], pageable, bytes, hipMemcpyHostToDevice, st[0])); CK(hipStreamSynchronize(st[0]));
    cp += ms_since(t0);
    t0 = Clock::now();
    CK(hipHostUnregister(pageable));
    unreg += ms_since(t0);
  }
  printf("hipHostRegister %7.3f ms + copy %7.3f ms + hipHostUnregister %7.3f ms\n", reg / reps, cp / reps, unreg / reps);

  for (int T : {1, 2, 4, 8}) {
    std::vector<char*> src(T);
    for (int i = 0; i < T; ++i) { src[i] = (char*)aligned_alloc(4096, bytes); memset(src[i], 3 + i, bytes); }
    t0 = Clock::now();
    for (int r = 0; r < reps; ++r) {
      std::vector<std::thread> th;
      for (int i = 0; i < T; ++i) th.emplace_back([&, i] { memcpy(pinned[i], src[i], bytes); });
      for (auto& t : th) t.join();
    }
    ms = ms_since(t0) / reps;
    printf("%d threads memcpy pageable->pinned %7.3f ms per round  %6.1f GB/s aggregate\n", T, ms, T * (double)bytes / ms / 1e6);
    for (int i = 0; i < T; ++i) free(src[i]);
  }
  return 0;
}
