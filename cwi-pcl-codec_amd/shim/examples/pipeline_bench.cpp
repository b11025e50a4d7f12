// pipeline_bench -- what a C++ caller gets from the sequence interface of the plain C ABI (include/pcc_codec.h), without
// Python or torch around it: N copies of a synthetic frame (points on a sphere shell in random order, like bench.py's
// headline frame) through pcc_pipeline_encode (frames resident in HBM) and pcc_pipeline_encode_host (frames in host
// memory), frames per second = Mpoints/s for the 1 M-point default.
//   pipeline_bench [points per frame = 1000000] [frames = 1024] [octree bits = 10] [host threads = CPUs of the job]
// Built with a plain host compiler (make -C cwi-pcl-codec_amd/shim/examples); nothing HIP-specific crosses the boundary.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "pcc_codec.h"

static int job_cpus() {  // the CPUs this process may really use (cgroup quota), like bench.py's default_workers()
  int cpus = (int)std::thread::hardware_concurrency();
  std::ifstream f("/sys/fs/cgroup/cpu.max");
  std::string quota;
  long period = 0;
  if (f >> quota >> period && quota != "max" && period > 0) cpus = std::min(cpus, (int)std::max(1L, atol(quota.c_str()) / period));
  return std::max(1, std::min(32, cpus));
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 1000000;
  const size_t frames = argc > 2 ? (size_t)atol(argv[2]) : 1024;
  const int bits = argc > 3 ? atoi(argv[3]) : 10;
  const int workers = argc > 4 ? atoi(argv[4]) : job_cpus();
  std::vector<std::vector<pcc_point_xyzrgb>> cloud(4, std::vector<pcc_point_xyzrgb>(n));
  for (size_t c = 0; c < cloud.size(); ++c) {
    std::mt19937 rng(1234 + (unsigned)c);
    std::uniform_real_distribution<float> u(0.f, 1.f);
    for (size_t i = 0; i < n; ++i) {
      const float ct = 2.f * u(rng) - 1.f, st = std::sqrt(std::max(0.f, 1.f - ct * ct)), ph = 6.2831853f * u(rng);
      const float r = 0.45f + 0.004f * (u(rng) - 0.5f);
      pcc_point_xyzrgb& p = cloud[c][i];
      memset(&p, 0, sizeof(p));
      p.x = 0.5f + r * st * std::cos(ph); p.y = 0.5f + r * st * std::sin(ph); p.z = 0.5f + r * ct; p.w = 1.f;
      const uint32_t cr = (uint32_t)(255.f * p.x), cg = (uint32_t)(255.f * p.y), cb = (uint32_t)(255.f * p.z);
      p.rgba = cb | (cg << 8) | (cr << 16) | 0xff000000u;
    }
  }
  pcc_params prm;
  memset(&prm, 0, sizeof(prm));
  prm.octree_resolution = std::ldexp(1.0, -bits); prm.point_resolution = prm.octree_resolution;
  prm.do_color_encoding = 1; prm.color_bit_resolution = 8; prm.color_coding_type = 1; prm.jpeg_quality = 85;
  prm.macroblock_size = 16; prm.frame_id = 1;

  pcc_pipeline* pipe = pcc_pipeline_create(0, workers);
  if (!pipe) { fprintf(stderr, "no usable HIP device (there is no CPU fallback)\n"); return 2; }
  for (int k = 0; k < pcc_pipeline_contexts(pipe); ++k) pcc_set_option(pcc_pipeline_context(pipe, k), "copy_image", 0);
  pcc_ctx* ctx0 = pcc_pipeline_context(pipe, 0);
  std::vector<void*> dev(cloud.size(), nullptr);
  for (size_t c = 0; c < cloud.size(); ++c) {
    if (pcc_device_alloc(ctx0, n * sizeof(pcc_point_xyzrgb), &dev[c]) != PCC_OK ||
        pcc_device_upload(ctx0, dev[c], cloud[c].data(), n * sizeof(pcc_point_xyzrgb)) != PCC_OK) {
      fprintf(stderr, "upload: %s\n", pcc_last_error(ctx0));
      return 1;
    }
  }
  std::vector<const void*> seq_dev(frames), seq_host(frames);
  std::vector<size_t> counts(frames, n);
  for (size_t f = 0; f < frames; ++f) { seq_dev[f] = dev[f % dev.size()]; seq_host[f] = cloud[f % cloud.size()].data(); }
  std::vector<pcc_bitstream> out(frames);
  typedef std::chrono::steady_clock Clock;
  auto run = [&](bool host, size_t count) {
    const Clock::time_point t0 = Clock::now();
    const int rc = host ? pcc_pipeline_encode_host(pipe, seq_host.data(), counts.data(), count, sizeof(pcc_point_xyzrgb), 16, &prm, out.data())
                        : pcc_pipeline_encode(pipe, seq_dev.data(), counts.data(), count, sizeof(pcc_point_xyzrgb), 16, &prm, out.data());
    if (rc != PCC_OK) { fprintf(stderr, "encode: %s\n", pcc_pipeline_last_error(pipe)); exit(1); }
    return std::chrono::duration<double>(Clock::now() - t0).count();
  };
  const size_t warm = std::min(frames, (size_t)pcc_pipeline_contexts(pipe));
  run(false, warm);                                                  // every context allocates its arena
  pcc_pipeline_reserve(pipe, frames, out[0].len, n);                 // landing buffers and output memory up front
  printf("%zu points per frame, %d host threads, bitstream %zu bytes\n", n, workers, out[0].len);
  for (int rep = 0; rep < 3; ++rep) {
    const double s = run(false, frames);
    printf("frames in HBM        : %5zu frames in %8.2f ms  %8.1f frames/s  %8.1f Mpoints/s\n", frames, 1e3 * s, frames / s, frames * (double)n / s / 1e6);
  }
  const size_t hf = std::min(frames, (size_t)256);
  run(true, std::min(hf, 2 * warm));
  for (int rep = 0; rep < 2; ++rep) {
    const double s = run(true, hf);
    printf("frames in host memory: %5zu frames in %8.2f ms  %8.1f frames/s  %8.1f Mpoints/s\n", hf, 1e3 * s, hf / s, hf * (double)n / s / 1e6);
  }
  for (void* d : dev) pcc_device_free(ctx0, d);
  pcc_pipeline_destroy(pipe);
  return 0;
}
