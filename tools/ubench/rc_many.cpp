// Host range coder alone, no GPU: 1 ... 16 streams in one call on one thread (ns per symbol and stream; up to four share a
// scalar loop, ten and more go through AVX-512 lanes where the CPU has them -- PCC_RC_WIDE=0 switches that off), on a
// byte stream with the statistics of an octree occupancy stream (one or two bits set per byte, mostly).
//   g++ -O3 -march=x86-64-v3 -I../../cwi-pcl-codec_amd/csrc -I../../include rc_many.cpp ../../cwi-pcl-codec_amd/csrc/pcc_host_codec.o -o rc_many
#include "pcc_host_codec.h"
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
using namespace pcc;
static uint64_t sm(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ull); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
int main(int argc, char** argv) {
  const size_t n = 942000;
  std::vector<std::vector<uint8_t>> in(16, std::vector<uint8_t>(n));
  uint64_t seed = 42;
  for (auto& v : in)
    for (auto& b : v) {
      const uint64_t r = sm(seed);
      uint8_t x = (uint8_t)(1u << (r & 7));
      if ((r >> 8) % 100 < 45) x |= (uint8_t)(1u << ((r >> 16) & 7));
      if ((r >> 24) % 100 < 15) x |= (uint8_t)(1u << ((r >> 32) & 7));
      b = x;
    }
  // the symbol counts come with the streams, as they do from the GPU stage (k_occ_histogram / k_jpeg_rows)
  std::vector<std::vector<uint32_t>> hist(in.size(), std::vector<uint32_t>(256, 0));
  for (size_t v = 0; v < in.size(); ++v) for (uint8_t b : in[v]) ++hist[v][b];
  uint64_t check = 0;
  for (int k = 1; k <= 16; ++k) {
    double best = 1e9;
    for (int rep = 0; rep < 7; ++rep) {
      std::vector<Bytes> out(k);
      const uint8_t* src[16]; size_t len[16]; Bytes* dst[16]; size_t got[16]; const uint32_t* cnt[16];
      for (int i = 0; i < k; ++i) { src[i] = in[i].data(); len[i] = n; dst[i] = &out[i]; cnt[i] = hist[i].data(); }
      const auto t0 = std::chrono::steady_clock::now();
      StaticRangeCoder::encode_many(k, src, len, dst, got, cnt);
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt < best) best = dt;
      if (rep == 0) for (int i = 0; i < k; ++i) { for (size_t j = 0; j < out[i].size(); j += 97) check = check * 131 + out[i][j]; check += got[i]; }
    }
    printf("%2d stream(s) in the call: %.3f ms per stream, %.2f ns per symbol and stream\n", k, best * 1e3 / k, best * 1e9 / k / n);
  }
  printf("checksum %016llx\n", (unsigned long long)check);
  return 0;
}
