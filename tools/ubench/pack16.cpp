// Host-side cost of packing pcl::PointXYZRGB (32 B) to the 16 bytes the kernels read (pack_points_16, csrc/pcc_host_codec.cpp):
//   g++ -O2 -march=x86-64-v3 -pthread -I cwi-pcl-codec_amd/csrc tools/ubench/pack16.cpp cwi-pcl-codec_amd/csrc/pcc_host_codec.o -o tools/ubench/pack16
//   tools/ubench/pack16 [points per frame] [threads]
// prints ms per frame and Mpoints/s for 1..threads threads packing different frames side by side.
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace pcc { void pack_points_16(uint8_t* dst, const uint8_t* src, size_t n, size_t stride, size_t rgb_offset); }

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1000000;
  const int tmax = argc > 2 ? atoi(argv[2]) : 8;
  for (int t = 1; t <= tmax; t *= 2) {
    std::vector<uint8_t*> src(t), dst(t);
    for (int k = 0; k < t; ++k) {
      src[k] = (uint8_t*)aligned_alloc(64, 32 * n); dst[k] = (uint8_t*)aligned_alloc(64, 16 * n);
      memset(src[k], k + 1, 32 * n); memset(dst[k], 0, 16 * n);
    }
    const int reps = 20;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int k = 0; k < t; ++k) th.emplace_back([&, k] { for (int r = 0; r < reps; ++r) pcc::pack_points_16(dst[k], src[k], n, 32, 16); });
    for (auto& x : th) x.join();
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%d thread(s): %.3f ms per frame of %zu points each, %.0f Mpoints/s together\n", t, s / reps * 1e3, n, (double)t * reps * n / s / 1e6);
    for (int k = 0; k < t; ++k) { free(src[k]); free(dst[k]); }
  }
  return 0;
}
