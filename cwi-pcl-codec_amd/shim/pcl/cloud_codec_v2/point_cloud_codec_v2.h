// point_cloud_codec_v2.h -- drop-in for the reference header of the same path
// (cloud_codec_v2/include/pcl/cloud_codec_v2/point_cloud_codec_v2.h, "codec.h" below).
//
// Same namespace, class name, constructor signature and public methods as
// pcl::io::OctreePointCloudCodecV2<PointT> (codec.h:70-227), implemented over the C ABI of
// libpcc_hip.so (include/pcc_codec.h) instead of on top of PCL's pointer octree.  An application
// written against the reference (apps/evaluate_compression) keeps its source and links
// -lpcc_hip instead of -lcloud_codec_v2; see INTEGRATION.md.
//
// What is behind each method:
//   ctor                         -> pcc_create            (eval.hpp:377-395 builds the codec)
//   encodePointCloud             -> pcc_encode_intra      (codec.h:174-175, impl.hpp:80-213)
//   decodePointCloud             -> pcc_decode_intra      (codec.h:177-178, impl.hpp:224-310)
//   getPerformanceMetrics        -> pcc_bitstream.perf    (codec.h:193-197)
//   getOutputCloud               -> pcc_get_output_cloud  (used at eval.hpp:862)
//   normalize_pointclouds / restore_scaling -> pcc_normalize_group / pcc_restore_scaling (codec.h:216-227)
//   encodePointCloudDeltaFrame / decodePointCloudDeltaFrame: the inter-frame path is outside this
//   round's scope; they throw std::logic_error rather than silently differ.
#pragma once
#ifndef PCL_POINT_TYPES_H_
#include "../pcl_lite.h"
#endif

#include <stdint.h>
#include <string.h>

#include <iostream>
#include <istream>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcc_codec.h"

namespace pcl {
namespace io {

struct BoundingBox {  // codec.h:64-68 (Eigen::Vector4f there; four floats here)
  float min_xyz[4];
  float max_xyz[4];
};

template <typename PointT>
class OctreePointCloudCodecV2 {
 public:
  typedef pcl::PointCloud<PointT> PointCloud;
  typedef typename PointCloud::Ptr PointCloudPtr;
  typedef typename PointCloud::ConstPtr PointCloudConstPtr;
  static_assert(sizeof(PointT) == 32, "the codec is instantiated for PointXYZRGB only (point_cloud_codec_v2.cpp:45)");

  // codec.h:108-121: identical argument list and defaults
  OctreePointCloudCodecV2(compression_Profiles_e compressionProfile_arg = MED_RES_ONLINE_COMPRESSION_WITH_COLOR,
                          bool showStatistics_arg = false, const double pointResolution_arg = 0.001,
                          const double octreeResolution_arg = 0.01, bool doVoxelGridDownDownSampling_arg = false,
                          const unsigned int iFrameRate_arg = 0, bool doColorEncoding_arg = true,
                          const unsigned char colorBitResolution_arg = 6, const unsigned char colorCodingType_arg = 0,
                          bool doVoxelGridCentroid_arg = true, bool createScalableStream_arg = true,
                          bool codeConnectivity_arg = false, int jpeg_quality_arg = 75, int num_threads = 0)
      : show_statistics_(showStatistics_arg), frame_id_(0) {
    if (compressionProfile_arg != MANUAL_CONFIGURATION)
      throw std::invalid_argument("OctreePointCloudCodecV2: only MANUAL_CONFIGURATION is supported (eval.hpp:379)");
    if (!doVoxelGridDownDownSampling_arg || iFrameRate_arg != 0)
      throw std::invalid_argument("OctreePointCloudCodecV2: voxel-grid intra coding only (eval.hpp:385-386)");
    (void)num_threads;
    memset(&prm_, 0, sizeof(prm_));
    prm_.octree_resolution = octreeResolution_arg;
    prm_.point_resolution = pointResolution_arg;
    prm_.do_color_encoding = doColorEncoding_arg;
    prm_.color_bit_resolution = colorBitResolution_arg;
    prm_.color_coding_type = colorCodingType_arg;
    prm_.do_voxel_centroid = doVoxelGridCentroid_arg;
    prm_.create_scalable = createScalableStream_arg;
    prm_.do_connectivity = codeConnectivity_arg;
    prm_.jpeg_quality = jpeg_quality_arg;
    prm_.macroblock_size = 16;     // codec.h:138
    prm_.do_icp_color_offset = 0;  // codec.h:141
    perf_[0] = perf_[1] = perf_[2] = 0;
    ctx_ = pcc_create(0);
    if (!ctx_) throw std::runtime_error("OctreePointCloudCodecV2: no usable MI355X/HIP device (there is no CPU fallback)");
  }
  ~OctreePointCloudCodecV2() { pcc_destroy(ctx_); }
  OctreePointCloudCodecV2(const OctreePointCloudCodecV2&) = delete;
  OctreePointCloudCodecV2& operator=(const OctreePointCloudCodecV2&) = delete;

  void initialization() {}                                     // codec.h:145
  void setMacroblockSize(int size) { prm_.macroblock_size = size; }  // codec.h:149
  void setDoICPColorOffset(bool doit) { prm_.do_icp_color_offset = doit; }  // codec.h:164

  // codec.h:174-175
  void encodePointCloud(const PointCloudConstPtr& cloud_arg, std::ostream& compressed_tree_data_out_arg) {
    prm_.frame_id = frame_id_ + 1;
    pcc_bitstream bs;
    const int rc = pcc_encode_intra(ctx_, cloud_arg->points.data(), cloud_arg->points.size(), sizeof(PointT), 16, &prm_, &bs);
    if (rc == PCC_ERR_EMPTY) {  // impl.hpp:206-212: frame silently dropped
      if (show_statistics_) std::cerr << "Info: Dropping empty point cloud\n";
      return;
    }
    if (rc != PCC_OK) throw std::runtime_error(std::string("encodePointCloud: ") + pcc_last_error(ctx_));
    ++frame_id_;
    for (int i = 0; i < 3; ++i) perf_[i] = bs.perf[i];
    compressed_tree_data_out_arg.write(reinterpret_cast<const char*>(bs.data), (std::streamsize)bs.len);
    compressed_tree_data_out_arg.flush();
  }

  // codec.h:177-178.  Reads the rest of the stream, decodes the first frame found and leaves the
  // read position right behind it (the reference consumes exactly one frame as well).
  void decodePointCloud(std::istream& compressed_tree_data_in_arg, PointCloudPtr& cloud_arg) {
    const std::streampos start = compressed_tree_data_in_arg.tellg();
    std::vector<char> buf((std::istreambuf_iterator<char>(compressed_tree_data_in_arg)), std::istreambuf_iterator<char>());
    pcc_cloud out;
    const int rc = pcc_decode_intra(ctx_, reinterpret_cast<const uint8_t*>(buf.data()), buf.size(), &out);
    compressed_tree_data_in_arg.clear();
    if (rc != PCC_OK) {  // impl.hpp:231: header not found -> output untouched
      compressed_tree_data_in_arg.seekg(start);
      return;
    }
    compressed_tree_data_in_arg.seekg(start + (std::streamoff)out.consumed);
    cloud_arg->points.resize(out.n);
    if (out.n) memcpy(static_cast<void*>(cloud_arg->points.data()), out.points, out.n * sizeof(PointT));
    cloud_arg->height = 1;  // impl.hpp:284-286
    cloud_arg->width = (uint32_t)out.n;
    cloud_arg->is_dense = false;
  }

  uint64_t* getPerformanceMetrics() { return perf_; }  // codec.h:193-197

  // OctreePointCloudCompression::getOutputCloud(): the simplified cloud of the last encode (impl.hpp:1576)
  PointCloudPtr getOutputCloud() {
    PointCloudPtr c(new PointCloud());
    const pcc_point_xyzrgb* p = nullptr;
    size_t n = 0;
    if (pcc_get_output_cloud(ctx_, &p, &n) == PCC_OK && n) {
      c->points.resize(n);
      memcpy(static_cast<void*>(c->points.data()), p, n * sizeof(PointT));
    }
    return c;
  }

  // codec.h:181-186 / impl.hpp:787-1118
  virtual void encodePointCloudDeltaFrame(const PointCloudConstPtr& icloud_arg, const PointCloudConstPtr& pcloud_arg, PointCloudPtr& out_cloud_arg,
                                          std::ostream& i_coded_data, std::ostream& p_coded_data, bool icp_on_original = false,
                                          bool write_out_cloud = true) {
    pcc_delta_params dp;
    memset(&dp, 0, sizeof(dp));
    dp.codec = prm_;
    dp.icp_on_original = icp_on_original;
    dp.write_out_cloud = write_out_cloud;
    pcc_delta_result r;
    const int rc = pcc_encode_delta(ctx_, reinterpret_cast<const pcc_point_xyzrgb*>(icloud_arg->points.data()), icloud_arg->points.size(),
                                    reinterpret_cast<const pcc_point_xyzrgb*>(pcloud_arg->points.data()), pcloud_arg->points.size(), &dp, &r);
    if (rc != PCC_OK) throw std::runtime_error(std::string("encodePointCloudDeltaFrame: ") + pcc_last_error(ctx_));
    i_coded_data.write(reinterpret_cast<const char*>(r.i_data), (std::streamsize)r.i_len);
    p_coded_data.write(reinterpret_cast<const char*>(r.p_data), (std::streamsize)r.p_len);
    if (write_out_cloud && out_cloud_arg) {  // push_back onto whatever the caller had in it (impl.hpp:899-935)
      const size_t before = out_cloud_arg->points.size();
      out_cloud_arg->points.resize(before + r.out_n);
      if (r.out_n) memcpy(static_cast<void*>(out_cloud_arg->points.data() + before), r.out_cloud, r.out_n * sizeof(PointT));
      out_cloud_arg->width = (uint32_t)out_cloud_arg->points.size();
      out_cloud_arg->height = 1;
    }
    shared_macroblock_percentage_ = r.shared_macroblock_percentage;
    shared_macroblock_convergence_percentage_ = r.shared_macroblock_convergence_percentage;
  }
  // codec.h:188-191 / impl.hpp:1120-1235
  virtual void decodePointCloudDeltaFrame(const PointCloudConstPtr& icloud_arg, PointCloudPtr& cloud_out_arg, std::istream& i_coded_data,
                                          std::istream& p_coded_data) {
    std::vector<char> ib((std::istreambuf_iterator<char>(i_coded_data)), std::istreambuf_iterator<char>());
    std::vector<char> pb((std::istreambuf_iterator<char>(p_coded_data)), std::istreambuf_iterator<char>());
    pcc_delta_params dp;
    memset(&dp, 0, sizeof(dp));
    dp.codec = prm_;
    pcc_cloud out;
    const int rc = pcc_decode_delta(ctx_, reinterpret_cast<const pcc_point_xyzrgb*>(icloud_arg->points.data()), icloud_arg->points.size(),
                                    reinterpret_cast<const uint8_t*>(ib.data()), ib.size(), reinterpret_cast<const uint8_t*>(pb.data()),
                                    pb.size(), &dp, &out);
    if (rc != PCC_OK) throw std::runtime_error(std::string("decodePointCloudDeltaFrame: ") + pcc_last_error(ctx_));
    const size_t before = cloud_out_arg->points.size();
    cloud_out_arg->points.resize(before + out.n);
    if (out.n) memcpy(static_cast<void*>(cloud_out_arg->points.data() + before), out.points, out.n * sizeof(PointT));
    cloud_out_arg->width = (uint32_t)cloud_out_arg->points.size();
    cloud_out_arg->height = 1;
  }
  float getMacroBlockPercentage() { return shared_macroblock_percentage_; }                        // codec.h:200-204
  float getMacroBlockConvergencePercentage() { return shared_macroblock_convergence_percentage_; }  // codec.h:207-210

  // codec.h:216-217 / impl.hpp:1840-1866: radius outlier filter on every cloud of the group (GPU; a context of its own,
  // the function is static in the reference as well)
  static void remove_outliers(std::vector<PointCloudPtr>& point_clouds, int min_points, double radius, unsigned int debug_level = 0) {
    if (min_points <= 0) return;
    pcc_ctx* c = pcc_create(0);
    if (!c) throw std::runtime_error("remove_outliers: no usable MI355X/HIP device (there is no CPU fallback)");
    for (auto& pc : point_clouds) {
      std::vector<uint8_t> keep(pc->points.size() + 1);
      size_t kept = 0;
      const int rc = pcc_remove_outliers(c, reinterpret_cast<const pcc_point_xyzrgb*>(pc->points.data()), pc->points.size(), min_points, radius,
                                         keep.data(), &kept);
      if (rc != PCC_OK) {
        const std::string msg = std::string("remove_outliers: ") + pcc_last_error(c);
        pcc_destroy(c);
        throw std::runtime_error(msg);
      }
      PointCloudPtr out(new PointCloud());
      out->points.reserve(kept);
      for (size_t i = 0; i < pc->points.size(); ++i)
        if (keep[i]) out->points.push_back(pc->points[i]);
      out->width = (uint32_t)out->points.size();
      out->height = 1;
      if (debug_level > 2) std::cout << "filtered out a total of: " << pc->points.size() - kept << " outliers" << std::endl;
      pc = out;  // "swap the pointer"
    }
    pcc_destroy(c);
  }

  // codec.h:223-227 (vectors of dyn_range/offset are unused by the reference as well)
  static BoundingBox normalize_pointclouds(std::vector<PointCloudPtr>& point_clouds, std::vector<BoundingBox>& bounding_boxes,
                                           double bb_expand_factor, std::vector<float> = std::vector<float>(),
                                           std::vector<float> = std::vector<float>(), unsigned int = 0) {
    std::vector<pcc_point_xyzrgb*> ptrs;
    std::vector<size_t> sizes;
    for (auto& c : point_clouds) {
      ptrs.push_back(reinterpret_cast<pcc_point_xyzrgb*>(c->points.data()));
      sizes.push_back(c->points.size());
    }
    BoundingBox bb;
    memset(&bb, 0, sizeof(bb));
    pcc_normalize_group(ptrs.data(), sizes.data(), ptrs.size(), bb_expand_factor, bb.min_xyz, bb.max_xyz);
    bounding_boxes.assign(point_clouds.size(), bb);
    return bb;
  }
  static void restore_scaling(PointCloudPtr& point_cloud, const BoundingBox& bb) {
    pcc_restore_scaling(reinterpret_cast<pcc_point_xyzrgb*>(point_cloud->points.data()), point_cloud->points.size(), bb.min_xyz,
                        bb.max_xyz);
  }

 private:
  pcc_ctx* ctx_;
  pcc_params prm_;
  bool show_statistics_;
  uint32_t frame_id_;
  uint64_t perf_[3];
  float shared_macroblock_percentage_ = 0.f, shared_macroblock_convergence_percentage_ = 0.f;
};

}  // namespace io
}  // namespace pcl
