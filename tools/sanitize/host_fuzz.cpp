// Host stages under AddressSanitizer + UBSan (CPU build only; GPU sanitizers are not available on the pool):
//   range coder round trips (1..4 streams at once, empty / tiny / skewed / long inputs),
//   JPEG encode -> decode on random images of awkward sizes,
//   rigid-transform coding on random matrices,
//   a frame built by the entropy stage from synthetic leaf products, decoded again,
//   the decoder on truncated and bit-flipped streams: must fail cleanly, never read or write out of bounds.
// Build + run: tools/sanitize/run.sh
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../cwi-pcl-codec_amd/csrc/pcc_delta.h"
#include "../../cwi-pcl-codec_amd/csrc/pcc_host_codec.h"

using namespace pcc;

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() {
  g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;
  return (uint32_t)(g_state >> 32);
}
#define CHECK(c) do { if (!(c)) { fprintf(stderr, "CHECK failed line %d: %s\n", __LINE__, #c); exit(1); } } while (0)

static void range_coder() {
  const size_t sizes[] = {0, 1, 2, 255, 256, 257, 4099, 70000, 300000};
  for (size_t n : sizes)
    for (int skew = 0; skew < 3; ++skew) {
      std::vector<std::vector<uint8_t>> in(4);
      for (int s = 0; s < 4; ++s) {
        in[s].resize(n + (size_t)s * (n / 3));
        for (auto& b : in[s]) b = skew == 0 ? (uint8_t)rnd() : (skew == 1 ? (uint8_t)((rnd() % 100) < 97 ? 7 : rnd()) : (uint8_t)(rnd() & 3));
      }
      for (int count = 1; count <= 4; ++count) {
        Bytes out[4];
        const uint8_t* src[4];
        size_t len[4], got[4];
        Bytes* dst[4];
        for (int s = 0; s < count; ++s) { src[s] = in[s].data(); len[s] = in[s].size(); dst[s] = &out[s]; out[s].push_back(0xAB); }
        StaticRangeCoder::encode_many(count, src, len, dst, got);
        for (int s = 0; s < count; ++s) {
          CHECK(out[s].size() == 1 + got[s] && out[s][0] == 0xAB);
          std::vector<uint8_t> back(len[s] + 1, 0xCD);
          const size_t used = StaticRangeCoder::decode(out[s].data() + 1, got[s], back.data(), len[s]);
          CHECK(used == got[s]);
          CHECK((len[s] == 0 || memcmp(back.data(), in[s].data(), len[s]) == 0) && back[len[s]] == 0xCD);
          // truncated input: must stop, not overrun
          if (got[s] > 8) (void)StaticRangeCoder::decode(out[s].data() + 1, got[s] / 2, back.data(), len[s]);
        }
      }
    }
}

static void jpeg() {
  const int dims[][2] = {{1, 1}, {3, 2}, {16, 16}, {17, 15}, {256, 1}, {256, 37}, {2, 300}, {33, 65}};
  for (auto& d : dims)
    for (int q : {1, 50, 85, 100}) {
      const int w = d[0], h = d[1];
      std::vector<uint8_t> img((size_t)w * h * 3);
      for (size_t i = 0; i < img.size(); ++i) img[i] = (uint8_t)((i * 7) ^ rnd() >> 28);
      Bytes jpg, back;
      BaselineJpeg::encode_rgb(img.data(), w, h, q, jpg);
      int bw = 0, bh = 0;
      CHECK(BaselineJpeg::decode_rgb(jpg.data(), jpg.size(), back, bw, bh));
      CHECK(bw == w && bh == h && back.size() == img.size());
      for (size_t cut : {(size_t)0, (size_t)2, jpg.size() / 2, jpg.size() - 1}) {  // truncated files
        Bytes tmp;
        (void)BaselineJpeg::decode_rgb(jpg.data(), cut, tmp, bw, bh);
      }
      for (int k = 0; k < 40; ++k) {  // corrupted files
        Bytes bad(jpg.begin(), jpg.end());
        bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() & 7));
        Bytes tmp;
        (void)BaselineJpeg::decode_rgb(bad.data(), bad.size(), tmp, bw, bh);
      }
    }
}

static void rigid() {
  for (int k = 0; k < 2000; ++k) {
    float m[16];
    for (float& v : m) v = ((int)(rnd() % 4001) - 2000) / (k % 2 ? 1000.0f : 300.0f);
    m[12] = m[13] = m[14] = 0; m[15] = 1;
    std::vector<int16_t> comp;
    rigid_compress(m, comp);
    CHECK(comp.size() == 6 || comp.size() == 10);
    float back[16];
    rigid_decompress(comp.data(), comp.size(), back);
  }
}

static void frames() {
  for (int round = 0; round < 12; ++round) {
    // leaf products as the GPU stage would hand them over (contents arbitrary: the host stage only packs them)
    const size_t L = 1 + rnd() % 3000, B = 1 + rnd() % 4000;
    std::vector<uint8_t> occ(B), bgr(3 * L), cen(3 * L);
    for (auto& v : occ) v = (uint8_t)(1u << (rnd() & 7));
    for (auto& v : bgr) v = (uint8_t)rnd();
    for (auto& v : cen) v = (uint8_t)rnd();
    const uint32_t W = 256, H = (uint32_t)(L / 256 + 1);
    std::vector<uint8_t> image((size_t)3 * W * H);
    for (auto& v : image) v = (uint8_t)rnd();
    pcc_hot_result hot;
    memset(&hot, 0, sizeof(hot));
    for (int a = 0; a < 3; ++a) { hot.bbox[a] = 0.0; hot.bbox[3 + a] = 1.0; }
    hot.depth = 8; hot.n_epochs = 1; hot.n_points_in = L; hot.n_leaves = L; hot.n_branches = B;
    hot.occupancy = occ.data(); hot.bgr = bgr.data(); hot.centroid = cen.data(); hot.image = image.data();
    hot.image_w = W; hot.image_h = H;
    pcc_params prm;
    memset(&prm, 0, sizeof(prm));
    prm.octree_resolution = 1.0 / 256; prm.point_resolution = 1.0 / 256; prm.do_color_encoding = 1; prm.color_bit_resolution = 8;
    prm.color_coding_type = round % 4; prm.do_voxel_centroid = round & 1; prm.jpeg_quality = 75; prm.macroblock_size = 16; prm.frame_id = 1;
    Bytes out;
    uint64_t perf[3];
    entropy_encode_frame(hot, prm, out, perf, nullptr);
    CHECK(out.size() > 140);
    // the occupancy bytes are random, not a tree: the decoder has to cope with whatever it finds
    PointVec pts;
    pcc_cloud info;
    (void)decode_frame(out.data(), out.size(), pts, info);
    for (int k = 0; k < 60; ++k) {
      Bytes bad(out.begin(), out.end());
      if (k % 3 == 0) bad.resize(rnd() % bad.size());
      else bad[rnd() % bad.size()] ^= (uint8_t)(1u << (rnd() & 7));
      (void)decode_frame(bad.data(), bad.size(), pts, info);
    }
  }
}

int main() {
  range_coder(); printf("range coder ok\n");
  jpeg(); printf("jpeg ok\n");
  rigid(); printf("rigid transform coding ok\n");
  frames(); printf("frames / decoder robustness ok\n");
  return 0;
}
