// pcc_host_codec.h -- host-side (serial) stages of the codec that stay on the CPU by design:
// frame header, static range coder, baseline JPEG Huffman coding, stream assembly, and the
// decoder.  North star: "the serial range/entropy coder runs on the host over the
// GPU-produced occupancy byte stream".
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <functional>
#include <vector>

#include "../../include/pcc_codec_tools.h"

namespace pcc {

// Byte buffers are resized to a worst case and then written: value-initialising the new bytes (what std::allocator
// does) would zero megabytes per frame for nothing.  resize() leaves new bytes indeterminate; every user writes
// before it reads.
template <typename T>
struct NoInitAllocator : std::allocator<T> {
  template <typename U> struct rebind { typedef NoInitAllocator<U> other; };
  NoInitAllocator() = default;
  template <typename U> NoInitAllocator(const NoInitAllocator<U>&) {}
  template <typename U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }  // default-init: no store
  template <typename U, typename... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(static_cast<A&&>(a)...); }
};
typedef std::vector<uint8_t, NoInitAllocator<uint8_t>> Bytes;

// pcl::StaticRangeCoder (call sites impl.hpp:1694,1706,1719 / 1778,1789,1798)
class StaticRangeCoder {
 public:
  // appends 1028-byte table + payload + 4 flush bytes to `out`; returns bytes appended
  static size_t encode(const uint8_t* in, size_t n, Bytes& out);
  // up to kMaxStreams independent streams coded in one loop (same bytes as separate encode() calls; see the
  // .cpp); the output vectors must be distinct; got[i] = bytes appended to *out[i].  Up to kInterleave streams share a
  // scalar loop; five to sixteen go through the lanes of AVX-512 registers where the CPU has them (wide_available()),
  // else through scalar loops of four, one group after the other.
  static constexpr int kMaxStreams = 16;
  static constexpr int kInterleave = 4;
  static bool wide_available();   // AVX-512 F/CD/BW/DQ/VL on this CPU and not switched off (PCC_RC_WIDE=0)
  // counts[i]: the 256-bin histogram of in[i] if it is known already, else null (counts itself may be null)
  static void encode_many(int count, const uint8_t* const in[], const size_t n[], Bytes* const out[], size_t got[],
                          const uint32_t* const counts[] = nullptr);
  // two streams
  static void encode2(const uint8_t* in_a, size_t n_a, Bytes& out_a, size_t& len_a,
                      const uint8_t* in_b, size_t n_b, Bytes& out_b, size_t& len_b);
  // returns bytes consumed, 0 on a truncated stream
  static size_t decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t n);
};

// libjpeg-turbo-compatible baseline JPEG, YCbCr 4:2:0, as driven by jpeg_io.hpp:211-330 / 90-192
class BaselineJpeg {
 public:
  static void encode_rgb(const uint8_t* rgb, int w, int h, int quality, Bytes& out);
  // Same file from quantised coefficients (6 x 64 per MCU, zigzag order) produced by the GPU front end:
  // headers + Huffman coding only.  Blocks of dummy rows below the image carry zeros; their DC is
  // taken from the preceding block here (jccoefct.c).
  static void encode_coefs(const int16_t* coefs, int w, int h, int quality, Bytes& out);
  // Same file from the per-MCU-row bit strings of the GPU Huffman stage (layout: pcc_hot_result.jpeg_tiles).
  // Returns false if a row did not fit its record (then the caller Huffman-codes the coefficients instead).
  static bool encode_tiles(const uint32_t* tiles, uint32_t tile_words, uint32_t n_tiles, int w, int h, int quality, Bytes& out);
  // A complete JPEG file around an entropy-coded bit string that the GPU produced for a whole image (colour coding type
  // 2: one strip): headers, the bits with 0xFF stuffing, padding, end marker.
  static void wrap_bits(const uint32_t* words, uint32_t n_bits, int w, int h, int quality, Bytes& out);
  // The Huffman tables of jpeg_set_defaults as (length << 16 | code): dc[component][size], ac[component][run << 4 | size]
  static void huffman_tables(uint32_t dc[2][12], uint32_t ac[2][256]);
  // The quantiser the GPU front end needs for `quality` (natural order; see pcc_kernels.h JpegQuant)
  static void quantiser(int quality, uint16_t half[2][64], uint32_t magic[2][64]);
  static bool decode_rgb(const uint8_t* jpg, size_t len, Bytes& rgb, int& w, int& h, uint64_t max_pixels = 0);  // max_pixels: refuse larger images (0: no bound)
  // Entropy decoding only (the sequential part of a JPEG decoder): quantised coefficients, six blocks of 64 per MCU
  // (Y00 Y01 Y10 Y11 Cb Cr) in natural order, and the quantisation tables of the three components.
  struct JpegCoefs {
    std::vector<int16_t> blocks;
    uint16_t q[3][64];
    int mcus_x = 0, mcus_y = 0;
  };
  static bool decode_coefs(const uint8_t* jpg, size_t len, int& w, int& h, JpegCoefs& out, uint64_t max_pixels = 0);

 private:
  static bool decode_impl(const uint8_t* jpg, size_t len, Bytes* rgb_out, int& w, int& h, JpegCoefs* coefs_out, uint64_t max_pixels);
};

// SnakeGridMapping (snake.h): position of linear element i in a w x h image (w multiple of 8)
uint32_t snake_position(uint32_t i, uint32_t w, uint32_t h);

// writeFrameHeader + entropyEncoding (impl.hpp:1472-1486, 1682-1760)
// `times_us` (optional, 4 doubles): occupancy range coder, JPEG (Huffman or full), colour range coder, whole stage
void entropy_encode_frame(const pcc_hot_result& hot, const pcc_params& prm, Bytes& out, uint64_t perf[3],
                          double* times_us = nullptr);

// the same for n = 1..4 frames at once: the range-coder stages of all frames share one loop
void entropy_encode_frames(int n, const pcc_hot_result* const hot[], const pcc_params* const prm[], Bytes* const out[],
                           uint64_t* const perf[], double* const times_us[]);

// decodePointCloud (impl.hpp:224-310); returns PCC_OK or PCC_ERR_STREAM
// (the cloud is written voxel by voxel: no zero-fill of 32 bytes per voxel first)
typedef std::vector<pcc_point_xyzrgb, NoInitAllocator<pcc_point_xyzrgb>> PointVec;
int decode_frame(const uint8_t* stream, size_t len, PointVec& points, pcc_cloud& info);
// pieces of the entropy stage for callers that run the range coders elsewhere (pcc_entropy_batch: on the GPU):
// the 140-byte frame header (impl.hpp:1472-1486), and what the colour range coder gets (jpegcc.h:115-139)
void frame_header_bytes(const pcc_hot_result& hot, const pcc_params& prm, Bytes& out);
void colour_stream_source(const pcc_hot_result& hot, const pcc_params& prm, Bytes& payload, const uint8_t*& src, size_t& src_len);

// The two sequential halves of decode_frame, for the decoder that does the rest on the GPU (pcc_decode_intra_gpu):
// header + the three range-coded vectors (+ the per-voxel colour bytes unless `colours_too` is false and the colours are
// one snake-mapped JPEG, which then stays in `payload`), and the walk over the occupancy stream.
struct FrameStreams {
  uint64_t count = 0;  // voxels announced by the header
  bool with_color = false;
  uint32_t cct = 0;
  Bytes occ, cen, col, payload;
};
// `after_occupancy` (may be empty) is called once the header is parsed and fs.occ holds the occupancy bytes, before the
// other two vectors are decoded: the host decoder starts the walk over the tree on a second thread there.
int decode_frame_streams(const uint8_t* stream, size_t len, pcc_cloud& info, FrameStreams& fs, bool colours_too,
                         const std::function<void()>& after_occupancy = std::function<void()>());
struct LeafParents {             // per node of level D-1, in stream order:
  std::vector<uint64_t> prefix;  //   its key, 3 bits per level, x-major triples (the 21 low triples)
  std::vector<uint32_t> prefix_hi;  // the triples above them: only filled for trees of more than 22 levels
  std::vector<uint8_t> bits;     //   its occupancy byte = which of its eight voxels exist
  std::vector<uint32_t> first;   //   how many voxels the nodes before it hold
};
int walk_leaf_parents(const Bytes& occ, unsigned depth, uint64_t count, LeafParents& lp);

// n points of `stride` bytes (x, y, z floats at 0, colour word at rgb_offset) -> 16 bytes each: x, y, z, colour word.
// What the kernels read of a pcl::PointXYZRGB is 16 of its 32 bytes; packing on the host halves what crosses PCIe.
// AVX2 with streaming stores for the PCL layout (32 / 16), scalar otherwise.  dst: 32-byte aligned.
void pack_points_16(uint8_t* dst, const uint8_t* src, size_t n, size_t stride, size_t rgb_offset);

}  // namespace pcc
