// pcc_delta.h -- the inter-frame ("delta") path: encodePointCloudDeltaFrame / decodePointCloudDeltaFrame
// (impl.hpp:787-1235), do_icp_prediction (impl.hpp:443-568), RigidTransformCoding / QuaternionCoding
// (impl/rigid_transform_coding_impl.hpp:63-203, impl/quaternion_coding_impl.hpp:55-222).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace pcc {

// ---- host: transform coding (float arithmetic as the reference's, Eigen restated) ----
// comp_dat: 6 int16 (quaternion + translation) or 10 (two rotation rows + sign word + translation)
void rigid_compress(const float tr[16] /* row-major 4x4 */, std::vector<int16_t>& comp_dat);
void rigid_decompress(const int16_t* comp_dat, size_t count, float tr_out[16]);

// ---- device ----
struct BlockTree {            // a macroblock tree = the leaf arrays of a hot-path run with a defined box at res * macroblock
  const uint64_t* sorted_keys;  // packed [code | point index], sorted
  const uint32_t* leaf_start;   // [n_blocks + 1]
  const uint64_t* leaf_code;    // [n_blocks] morton code inside the varying-bit window
  uint64_t prefix_code;         // morton code of the constant high key bits
  uint64_t index_mask;          // low bits of sorted_keys = point index
  uint32_t n_blocks, n_points;
  const uint8_t* points;        // the cloud the tree was built on
  uint32_t stride, rgb_off;
};

struct BlockResult {          // one per macroblock of the predictive frame, in depth-first (Morton) order
  int32_t i_block;            // macroblock of the I frame with the same key, or -1 (exclusive block)
  uint32_t n_p, n_i;
  int32_t do_icp;             // passed the size and colour-variance gates (impl.hpp:453-521)
  int32_t converged;          // ICP converged and fitness < 2 * point_resolution (impl.hpp:560)
  int32_t iterations;
  int8_t rgb_offsets[4];
  uint16_t key[4];            // kx, ky, kz of the block
  float fitness;
  float rt[16];               // final ICP transformation, row-major
};

struct DeltaArgs {
  BlockTree i_tree, p_tree;
  uint64_t* i_full;            // [i blocks] full morton keys (scratch)
  uint64_t* p_full;            // [p blocks]
  float4* i_xyzc;              // [n_i] block-ordered x, y, z, rgba bits of the I cloud
  float4* p_xyzc;              // [n_p] same for the predictive cloud
  float4* cur;                 // [n_i] scratch: the source points of a block while ICP moves them
  uint32_t* nn;                // [n_i] scratch: nearest target of every source point
  BlockResult* results;        // [p blocks]
  uint32_t* work;              // [p blocks] n_i * n_p of the blocks that go through ICP, else 0
  uint32_t* order;             // [p blocks] those blocks by falling work
  uint32_t* counts;            // [2] how many there are, how many of them are heavy
  int shape;                   // 0: heavy blocks get a workgroup each, the others a wave; 1: a workgroup each; 2: a wave each
  double point_resolution;
  double octree_resolution;    // voxel size of the coder; a macroblock is macroblock_size voxels a side
  int macroblock_size;
  int max_iterations;
  float transformation_epsilon;
  float var_threshold;
  int do_icp_color_offset;
};

// aux, ev_fork, ev_join: an idle second stream and two events (or null): the two ICP kernel shapes then run side by side
void launch_delta_blocks(const DeltaArgs& a, hipStream_t stream, hipStream_t aux = nullptr, hipEvent_t ev_fork = nullptr,
                         hipEvent_t ev_join = nullptr);

// second phase, after the host has decided which blocks are predicted:
//   out_intra[dst_intra[b] ..] = points of block b (exclusive / failed blocks), 32-byte PointXYZRGB
//   out_cloud[dst_out[b] ..]   = predicted (transformed I block) or copied points
struct GatherArgs {
  const float4* i_xyzc;
  const float4* p_xyzc;
  const uint32_t* i_leaf_start;
  const uint32_t* p_leaf_start;
  const BlockResult* results;
  const float* mdec;            // [p blocks][16] decoded transforms (valid where predicted)
  const uint32_t* dst_intra;    // [p blocks] or 0xffffffff
  const uint32_t* dst_out;      // [p blocks] or 0xffffffff
  uint32_t n_blocks;
  int do_icp_color_offset;
  int colour_doubled;           // decoder: p.r += p.r + offset (impl.hpp:1187-1189); encoder's out cloud: pt.r += offset (impl.hpp:901-905)
  uint8_t* out_intra;           // PointXYZRGB[]
  uint8_t* out_cloud;           // PointXYZRGB[]
};
void launch_delta_gather(const GatherArgs& a, hipStream_t stream);

}  // namespace pcc
