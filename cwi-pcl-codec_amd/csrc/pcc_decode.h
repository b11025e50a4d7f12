// pcc_decode.h -- launch interface of the GPU half of decodePointCloud (impl.hpp:224-310): everything behind the
// sequential stages (range decoders, JPEG entropy decoding, the walk over the depth-first occupancy stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcc {

struct DecodeArgs {
  // geometry: one entry per node of level D-1 ("leaf parent"), in stream order
  const uint64_t* prefix;   // key of the node, 3 bits per level, x-major triples (the 21 low triples)
  const uint32_t* prefix_hi;  // the triples above them, or null (trees of up to 22 levels)
  const uint8_t* bits;      // its occupancy byte
  const uint32_t* first;    // voxels before it
  uint32_t n_parents;
  uint32_t n_leaves;
  double res, mn[3];        // octree resolution, bounding-box minimum (header)
  const uint8_t* centroid;  // 3 bytes per voxel or null (PointCodingV2::decodePoint, ptv2.h:103-118)
  // colours: either 3 bytes per voxel in voxel order (types 0, 2, 3; already un-snaked on the host) ...
  const uint8_t* colours;
  uint32_t colour_shift;    // colorBitReduction_ of the PCL colour coder (type 0)
  int with_colour;
  // ... or the planes of the snake-mapped JPEG (type 1): Y (y_stride wide), Cb, Cr (c_stride wide), image w x h
  const uint8_t* plane_y;
  const uint8_t* plane_cb;
  const uint8_t* plane_cr;
  uint32_t img_w, img_h, y_stride, c_stride;
  void* points;             // out: 32-byte pcl::PointXYZRGB per voxel, depth-first (= Morton) order
};

struct IdctArgs {
  const int16_t* blocks;    // quantised coefficients, six blocks of 64 per MCU (Y00 Y01 Y10 Y11 Cb Cr), natural order
  uint16_t q[3][64];        // quantisation tables of Y, Cb, Cr
  uint32_t mcus_x, mcus_y;
  uint8_t* plane_y;         // (16 mcus_x) x (16 mcus_y)
  uint8_t* plane_cb;        // (8 mcus_x) x (8 mcus_y)
  uint8_t* plane_cr;
};

void launch_decode_idct(const IdctArgs& a, hipStream_t stream);
void launch_decode_points(const DecodeArgs& a, hipStream_t stream);

}  // namespace pcc
