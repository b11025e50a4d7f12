// pcc_kernels.h -- launch interface between the C-ABI layer (pcc_api.cpp) and the HIP kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "pcc_device.h"

namespace pcc {

// How to read the caller's point array (pcl::PointXYZRGB: stride 32, colour word at 16).
struct PointView {
  const uint8_t* base;
  uint32_t stride;
  uint32_t rgb_off;
  uint32_t aligned16;  // base and stride are multiples of 16: x,y,z,w come in one 16-byte load
};

// defineBoundingBox() before the points are added (the delta path builds its trees in [0,1]^3, impl.hpp:340,426):
// the box is given instead of grown around the first point; it still grows if a point lies outside.
struct FixedBox {
  int enabled;
  double mn[3], mx[3];
};

struct LeafParams {
  uint32_t do_color;         // cloud_with_color_
  uint32_t color_reduction;  // colorBitReduction_ (only the PCL colour coder, type 0, ever has one)
  uint32_t do_centroid;      // do_voxel_centroid_enDecoding_
  uint32_t write_image;      // colour coding type 1: emit the snake-mapped 256 x H image
  uint32_t simplify_only;    // simplifyPCloud (impl.hpp:318-403): only the simplified cloud, centre = (key + 0.5) * res + min
  uint32_t linear_rows;      // developer knob (PCC_LEAF_ROWS=linear): block row = blockIdx instead of the per-XCD ranges of k_leaf_tile
  uint32_t uniform_probes;   // developer knob (PCC_LEAF_PROBES=uniform): evenly spaced first probes of the parent search (round 2's layout)
};

// Quantiser of the JPEG front end: per component (0 luma, 1 chroma), natural order:
// half = (8q) >> 1, magic = floor(2^32 / (8q)) + 1 (exact reciprocal for |coef| + half < 2^17).
struct JpegQuant {
  uint16_t half[2][64];
  uint32_t magic[2][64];
};

// Per-MCU-row record of the GPU Huffman stage (u32 words; bits are MSB-first inside each word):
//   [0] total bits   [1] bit offset of the first MCU's Cb block   [2] ... of its Cr block   [3] 1 = did not fit
//   [4..6] DC of the first Y / Cb / Cr block of the row (their DC codes are left out: they depend on the row
//   before and are inserted by the host)   [7..9] DC of the last Y / Cb / Cr block   [16..] the bits
constexpr int kJpegTileWords = 512;
constexpr int kJpegTileHeader = 16;
constexpr int kJpegTileBits = (kJpegTileWords - kJpegTileHeader) * 32;
// "lines" mode: a strip has at most 256 MCUs = 1536 blocks of at most 219 bits
constexpr int kJpegLineWords = (1536 * 219 + 31) / 32 + 15;
// Huffman tables as (length << 16 | code): DC [component][size 0..11], AC [component][run << 4 | size]
struct JpegHuffTables {
  uint32_t dc[2][12];
  uint32_t ac[2][256];
};

struct HotPathArgs {
  PointView pv;
  uint32_t n;
  double res;
  double inv_res_pow2;  // 1/res when res is a power of two (then x * inv == x / res exactly), else 0
  LeafParams lp;
  int max_passes;  // sort passes to enqueue (the device decides how many do work; more needed => kErrPasses)
  int force_pairs; // testing: use the (key, index) pair sort even when the packed key would fit
  int no_cell_ranks;  // testing: sort the full varying Morton code even where cell ranks would save a pass
  int need_index;  // somebody reads the point index of the sorted elements (centroids, macroblock trees): keep it in the key
  FixedBox box;    // defineBoundingBox before addPointsFromInputCloud
  int stop_after_leaf_scan;  // macroblock trees: only the sorted points and the leaf (= block) arrays are wanted
  uint64_t* boxes;     // eight {value, frame_seq} words per 2048-point chunk (zeroed when allocated)
  uint32_t frame_seq;  // never 0, different from the frame before on this arena: tells this frame's chunk boxes from older ones
  FrameState* state;
  uint64_t* keys_a;
  uint64_t* keys_b;
  uint32_t* idx_a;
  uint32_t* idx_b;
  int deep_launch;     // 1: enqueue the DEEP instantiations (two-word Morton codes: trees of 22 to 31 levels); the device sends
                       // a deep frame that meets the single-word kernels back with kErrDeep
  uint32_t* idx2_a;    // deep only: second payload of the sort (point index or colour word), ping-pong
  uint32_t* idx2_b;
  uint32_t* leaf_hi;   // deep only: high word of each leaf's code
  uint16_t* hist_rows;  // [sort tiles][kMaxPasses][kMaxBins] digit counts (at most 4096 each) from k_make_keys
  uint32_t* digit_tot;  // [kMaxPasses][kMaxBins]
  uint32_t* tile_prefix0;  // [sort tiles][kMaxBins] exclusive tile prefix of the pass-0 digit counts
  uint8_t* sync_area;   // tickets | leaf scan status | sort status (sync_area_bytes), zeroed by k_boxes_events
  uint32_t* leaf_start;
  uint64_t* leaf_code;
  uint32_t* leaf_base;
  uint8_t* leaf_t;
  uint8_t* occ;
  uint8_t* bgr;
  uint8_t* centroid;
  uint8_t* image;
  void* simplified;  // float4 per leaf (x, y, z, rgba bits)
  int16_t* coefs;    // quantised JPEG coefficients, 6 x 64 per MCU in zigzag order (null: JPEG on the host)
  JpegQuant jq;
  uint32_t* jpeg_tiles;          // per-MCU-row Huffman records (null: Huffman coding on the host)
  const JpegHuffTables* huff;    // device copy of the standard tables
  // colour coding type 2 on the GPU: directory {word offset, bits, width, overflow} per strip and the strips' bit strings
  uint32_t* jpeg_lines_dir;
  uint32_t* jpeg_lines_data;
  uint32_t jpeg_lines_capacity;  // words
  std::vector<const char*>* span_names;  // profiling: kernel name of every span slot, in launch order (host side)
  unsigned long long* spans;     // profiling: kMaxSpans x 2 x kSpanShards words {workgroup starts | ~(wave ends)} in launch order, preset to ~0 (null: off)
};
constexpr int kMaxSpans = 24;
constexpr int kSpanShards = 256;
constexpr size_t kSpanWords = (size_t)kMaxSpans * 2 * kSpanShards;

// Optional per-kernel timing: one HIP event after every launch, on the launch stream.
class KernelTimer {
 public:
  void reset() { used_ = 0; }
  void stamp(const char* name, hipStream_t s) {
    if (used_ == events_.size()) {
      hipEvent_t e;
      (void)hipEventCreate(&e);
      events_.push_back(e);
      names_.push_back(name);
    }
    names_[used_] = name;
    (void)hipEventRecord(events_[used_], s);
    ++used_;
  }
  // after the stream is synchronised: (name, ms) per launch, stamp 0 is the origin
  void collect(std::vector<std::pair<const char*, float>>& out) const {
    out.clear();
    for (size_t i = 1; i < used_; ++i) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, events_[i - 1], events_[i]);
      out.emplace_back(names_[i], ms);
    }
  }
  ~KernelTimer() {
    for (auto e : events_) (void)hipEventDestroy(e);
  }

 private:
  std::vector<hipEvent_t> events_;
  std::vector<const char*> names_;
  size_t used_ = 0;
};

size_t sync_area_bytes(uint32_t n, int passes);
void launch_hot_path(const HotPathArgs& a, hipStream_t stream, KernelTimer* tm);

}  // namespace pcc
