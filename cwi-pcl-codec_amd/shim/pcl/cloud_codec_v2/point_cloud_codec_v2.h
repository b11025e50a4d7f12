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
//   normalize_pointclouds / restore_scaling -> pcc_normalize_group_boxes / pcc_restore_scaling (codec.h:216-227)
//   remove_outliers              -> pcc_remove_outliers   (codec.h:216-217)
//   encodePointCloudDeltaFrame / generatePointCloudDeltaFrame / decodePointCloudDeltaFrame
//                                -> pcc_encode_delta / pcc_decode_delta (codec.h:181-191)
// The GPU is chosen with the environment variable PCC_DEVICE (default 0): the reference's constructor has no
// argument for it, and its signature is kept.
#pragma once
#ifndef PCL_POINT_TYPES_H_
#include "../pcl_lite.h"
#endif

#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <iostream>
#include <istream>
#include <iterator>
#include <ostream>
#include <stdexcept>
#include <string>
#include <vector>

#include "pcc_codec.h"

namespace pcl {
namespace io {

struct BoundingBox {  // codec.h:64-68
  Eigen::Vector4f min_xyz;
  Eigen::Vector4f max_xyz;
};

inline int pcc_shim_device() {  // which GPU the codec objects of this process use
  const char* e = getenv("PCC_DEVICE");
  return e ? atoi(e) : 0;
}

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
    // What constructs is what the evaluation app passes (eval.hpp:377-395): MANUAL_CONFIGURATION, voxel-grid downsampling on,
    // iFrameRate 0.  The reference's OWN default arguments (codec.h:108-112: a PCL profile, downsampling off) select PCL's
    // profile table and the point-detail tail of the bitstream (impl.hpp:1728-1757), neither of which is built here: refused,
    // before any device is touched, rather than coded differently from the reference.
    if (compressionProfile_arg != MANUAL_CONFIGURATION)
      throw std::invalid_argument("OctreePointCloudCodecV2 (libpcc_hip): only MANUAL_CONFIGURATION is supported -- PCL's compression "
                                  "profiles, the reference's default first argument included, are not built (eval.hpp:379 passes MANUAL_CONFIGURATION)");
    if (!doVoxelGridDownDownSampling_arg || iFrameRate_arg != 0)
      throw std::invalid_argument("OctreePointCloudCodecV2 (libpcc_hip): only doVoxelGridDownDownSampling = true with iFrameRate = 0 is "
                                  "supported -- the point-detail stream (impl.hpp:1728-1757) is not built (eval.hpp:385-386 passes true, 0)");
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
    icp_max_iterations_ = 50;      // codec.h:139-142
    transformationepsilon_ = 1e-8f;
    perf_[0] = perf_[1] = perf_[2] = 0;
    ctx_ = pcc_create(pcc_shim_device());
    if (!ctx_) throw std::runtime_error("OctreePointCloudCodecV2: no usable MI355X/HIP device (there is no CPU fallback)");
  }
  ~OctreePointCloudCodecV2() { pcc_destroy(ctx_); }
  OctreePointCloudCodecV2(const OctreePointCloudCodecV2&) = delete;
  OctreePointCloudCodecV2& operator=(const OctreePointCloudCodecV2&) = delete;

  void initialization() {}                                     // codec.h:145
  void setMacroblockSize(int size) { prm_.macroblock_size = size; }  // codec.h:149
  void setColorVarThreshold(int size) { prm_.macroblock_size = size; }  // codec.h:154-157: the reference assigns macroblock_size_ here
  void setMaxIterations(int max_in) { icp_max_iterations_ = max_in; }  // codec.h:159-162
  void setDoICPColorOffset(bool doit) { prm_.do_icp_color_offset = doit; }  // codec.h:164
  void setDoICPColorOffset(float tfeps) { transformationepsilon_ = tfeps; }  // codec.h:169-172: the float overload sets transformationepsilon_

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

  // codec.h:177-178.  Consumes exactly one frame, like the reference (syncToHeader, impl.hpp:1660-1676, then the
  // pieces of impl.hpp:1766-1835).  The range-coded pieces do not state their coded length, so the frame is read in
  // a bounded piece sized from its header (doubled if the decoder runs out of bytes) and what was read beyond the
  // frame is given back to the stream: a concatenation of frames costs every byte a constant number of times.
  // A stream that cannot seek keeps the surplus inside this object; it is used first at the next call on that stream.
  void decodePointCloud(std::istream& compressed_tree_data_in_arg, PointCloudPtr& cloud_arg) {
    std::vector<uint8_t> buf;
    pcc_cloud out;
    if (!read_and_decode(compressed_tree_data_in_arg, buf, out)) return;  // impl.hpp:231: no header -> output untouched
    cloud_arg->points.resize(out.n);
    if (out.n) memcpy(static_cast<void*>(cloud_arg->points.data()), out.points, out.n * sizeof(PointT));
    cloud_arg->height = 1;  // impl.hpp:284-286
    cloud_arg->width = (uint32_t)out.n;
    cloud_arg->is_dense = false;
  }

  uint64_t* getPerformanceMetrics() { return perf_; }  // codec.h:193-197

  // Not in the reference: what a caller needs to hand whole groups of frames to pcc_pipeline / pcc_multi_pipeline
  // (several GPUs) and keep this object's frame counter in step with the frames coded there.
  const pcc_params& native_params() const { return prm_; }
  uint32_t next_frame_id() const { return frame_id_ + 1; }
  void advance_frame_id(uint32_t frames_coded) { frame_id_ += frames_coded; }

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
                                          bool write_out_cloud = false) {
    delta_frame(icloud_arg, pcloud_arg, out_cloud_arg, i_coded_data, p_coded_data, icp_on_original, write_out_cloud, true);
  }
  // codec.h:177-179 / impl.hpp:579-777: the serial predecessor of encodePointCloudDeltaFrame.  Same blocks, gates, ICP and
  // intra part; its p_coded_data chunks have no leading size byte (impl.hpp:657-662 against :1021-1027), and it resets
  // out_cloud_arg's width/height before the blocks are visited (impl.hpp:603-604).
  virtual void generatePointCloudDeltaFrame(const PointCloudConstPtr& icloud_arg, const PointCloudConstPtr& pcloud_arg, PointCloudPtr& out_cloud_arg,
                                            std::ostream& i_coded_data, std::ostream& p_coded_data, bool icp_on_original = false,
                                            bool write_out_cloud = true) {
    out_cloud_arg->height = 1;
    out_cloud_arg->width = 0;
    delta_frame(icloud_arg, pcloud_arg, out_cloud_arg, i_coded_data, p_coded_data, icp_on_original, write_out_cloud, false);
  }

 private:
  void delta_frame(const PointCloudConstPtr& icloud_arg, const PointCloudConstPtr& pcloud_arg, PointCloudPtr& out_cloud_arg,
                   std::ostream& i_coded_data, std::ostream& p_coded_data, bool icp_on_original, bool write_out_cloud, bool chunk_sizes) {
    pcc_delta_params dp;
    memset(&dp, 0, sizeof(dp));
    dp.codec = prm_;
    dp.icp_on_original = icp_on_original;
    dp.write_out_cloud = write_out_cloud;
    dp.icp_max_iterations = icp_max_iterations_;
    dp.transformation_epsilon = transformationepsilon_;
    pcc_delta_result r;
    const int rc = pcc_encode_delta(ctx_, reinterpret_cast<const pcc_point_xyzrgb*>(icloud_arg->points.data()), icloud_arg->points.size(),
                                    reinterpret_cast<const pcc_point_xyzrgb*>(pcloud_arg->points.data()), pcloud_arg->points.size(), &dp, &r);
    if (rc != PCC_OK) throw std::runtime_error(std::string("encodePointCloudDeltaFrame: ") + pcc_last_error(ctx_));
    i_coded_data.write(reinterpret_cast<const char*>(r.i_data), (std::streamsize)r.i_len);
    if (chunk_sizes) {
      p_coded_data.write(reinterpret_cast<const char*>(r.p_data), (std::streamsize)r.p_len);
    } else {
      for (size_t at = 0; at < r.p_len;) {  // u8 size | chunk
        const size_t len = r.p_data[at];
        p_coded_data.write(reinterpret_cast<const char*>(r.p_data + at + 1), (std::streamsize)len);
        at += 1 + len;
      }
    }
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

 public:
  // codec.h:188-191 / impl.hpp:1120-1235.  Like the reference: the chunk list is read to the end of p_coded_data
  // (impl.hpp:1144-1150 loops until the stream is exhausted) and exactly one intra frame is taken from i_coded_data.
  virtual void decodePointCloudDeltaFrame(const PointCloudConstPtr& icloud_arg, PointCloudPtr& cloud_out_arg, std::istream& i_coded_data,
                                          std::istream& p_coded_data) {
    std::vector<uint8_t> ib;
    {
      pcc_cloud unused;
      read_and_decode(i_coded_data, ib, unused);  // exactly one intra frame leaves i_coded_data; it is decoded again below with the chunks
    }
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
    pcc_ctx* c = pcc_create(pcc_shim_device());  // one context for the whole group (the function is static: there is no object to keep it in)
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

  // codec.h:223-227 / impl.hpp:1871-1967.  The box vector's allocator is a template parameter: the reference declares
  // vector<BoundingBox, Eigen::aligned_allocator<BoundingBox>> and the app passes exactly that (eval.hpp:442-443).
  // dyn_range / offset are unused by the reference as well.  bounding_boxes[k] = the box in force for cloud k.
  template <class BoxAlloc>
  static BoundingBox normalize_pointclouds(std::vector<PointCloudPtr>& point_clouds, std::vector<BoundingBox, BoxAlloc>& bounding_boxes,
                                           double bb_expand_factor, std::vector<float> = std::vector<float>(),
                                           std::vector<float> = std::vector<float>(), unsigned int = 0) {
    std::vector<pcc_point_xyzrgb*> ptrs;
    std::vector<size_t> sizes;
    for (auto& c : point_clouds) {
      ptrs.push_back(reinterpret_cast<pcc_point_xyzrgb*>(c->points.data()));
      sizes.push_back(c->points.size());
    }
    float mn[3] = {0.f, 0.f, 0.f}, mx[3] = {0.f, 0.f, 0.f};
    std::vector<float> per_cloud(6 * point_clouds.size() + 6);
    pcc_normalize_group_boxes(ptrs.data(), sizes.data(), ptrs.size(), bb_expand_factor, mn, mx, per_cloud.data());
    BoundingBox bb;
    for (int a = 0; a < 3; ++a) { bb.min_xyz[a] = mn[a]; bb.max_xyz[a] = mx[a]; }
    bb.min_xyz[3] = 0.f;  // impl.hpp:1874: min_pt_bb(0,0,0,0); max_pt_bb[3] is never set there
    bb.max_xyz[3] = 0.f;
    if (bounding_boxes.size() < point_clouds.size()) bounding_boxes.resize(point_clouds.size());
    for (size_t k = 0; k < point_clouds.size(); ++k)
      for (int a = 0; a < 3; ++a) {
        bounding_boxes[k].min_xyz[a] = per_cloud[6 * k + a];
        bounding_boxes[k].max_xyz[a] = per_cloud[6 * k + 3 + a];
      }
    return bb;
  }
  static void restore_scaling(PointCloudPtr& point_cloud, const BoundingBox& bb) {
    const float mn[3] = {bb.min_xyz[0], bb.min_xyz[1], bb.min_xyz[2]}, mx[3] = {bb.max_xyz[0], bb.max_xyz[1], bb.max_xyz[2]};
    pcc_restore_scaling(reinterpret_cast<pcc_point_xyzrgb*>(point_cloud->points.data()), point_cloud->points.size(), mn, mx);
  }

 private:
  // bytes from the carry-over of `in` first, then from `in` itself
  size_t pull(std::istream& in, std::vector<uint8_t>& buf, size_t n) {
    size_t got = 0;
    if (carry_stream_ == &in && !carry_.empty()) {
      got = n < carry_.size() ? n : carry_.size();
      buf.insert(buf.end(), carry_.begin(), carry_.begin() + (std::ptrdiff_t)got);
      carry_.erase(carry_.begin(), carry_.begin() + (std::ptrdiff_t)got);
    }
    // the rest comes off the stream in bounded pieces: `n` may derive from an untrusted header field, and a buffer is only
    // ever grown by what the stream really delivered
    while (got < n && in) {
      const size_t piece = (n - got) < ((size_t)1 << 22) ? (n - got) : ((size_t)1 << 22);
      const size_t at = buf.size();
      buf.resize(at + piece);
      in.read(reinterpret_cast<char*>(buf.data() + at), (std::streamsize)piece);
      const size_t r = (size_t)in.gcount();
      buf.resize(at + r);
      got += r;
      if (r < piece) break;
    }
    return got;
  }
  // one frame off the stream, decoded into the context's cloud buffer; `buf` holds exactly the frame afterwards
  bool read_and_decode(std::istream& in, std::vector<uint8_t>& buf, pcc_cloud& out) {
    static const char kId[] = "<PCL-OCT-CODECV2-COMPRESSED>";  // codec.h:371; 28 bytes on the wire
    const size_t id_len = sizeof(kId) - 1;
    if (carry_stream_ != &in) carry_.clear();
    size_t matched = 0;
    std::vector<uint8_t> one;
    while (matched < id_len) {  // syncToHeader: byte by byte until the identifier has gone by
      one.clear();
      if (pull(in, one, 1) != 1) { carry_.clear(); carry_stream_ = nullptr; return false; }  // end of the stream
      const char ch = (char)one[0];
      matched = (ch == kId[matched]) ? matched + 1 : ((ch == kId[0]) ? 1 : 0);
    }
    buf.assign(kId, kId + id_len);
    // 20-byte base identifier, frame id, flags, then the voxel count at byte 55: the size of the frame follows it loosely
    if (pull(in, buf, 140 - id_len) != 140 - id_len) { carry_.clear(); carry_stream_ = nullptr; return false; }
    uint64_t voxels = 0;
    memcpy(&voxels, buf.data() + 55, sizeof(voxels));
    // (the count only sizes the FIRST read, and only up to 4 MB: a corrupt header must not be able to ask for gigabytes)
    size_t want = 8192 + (size_t)(voxels < (1ull << 21) ? voxels : (1ull << 21)) * 2;
    for (;;) {
      const size_t got = pull(in, buf, want);
      const int rc = pcc_decode_intra(ctx_, buf.data(), buf.size(), &out);
      if (rc == PCC_OK) {
        const size_t surplus = buf.size() - out.consumed;
        if (surplus) {
          in.clear();  // reading to the end of the stream sets eofbit / failbit
          if (carry_stream_ == &in && !carry_.empty()) {  // part of the surplus may have come from the carry-over: it goes back there
            carry_.insert(carry_.begin(), buf.end() - (std::ptrdiff_t)surplus, buf.end());
          } else if (in.tellg() != std::streampos(-1) && in.seekg(-(std::streamoff)surplus, std::ios_base::cur)) {
            carry_.clear();
          } else {
            in.clear();
            carry_stream_ = &in;
            carry_.assign(buf.end() - (std::ptrdiff_t)surplus, buf.end());
          }
          buf.resize(out.consumed);
        }
        return true;
      }
      if (got < want) {  // the stream has no more to give: truncated or corrupt frame (impl.hpp has no error path either)
        in.clear();
        carry_.clear();  // nothing of this stream is kept: a later stream object at the same address starts clean
        carry_stream_ = nullptr;
        return false;
      }
      want = buf.size();  // double what has been read
    }
  }
  std::vector<uint8_t> carry_;
  const std::istream* carry_stream_ = nullptr;

  pcc_ctx* ctx_;
  pcc_params prm_;
  int icp_max_iterations_;
  float transformationepsilon_;
  bool show_statistics_;
  uint32_t frame_id_;
  uint64_t perf_[3];
  float shared_macroblock_percentage_ = 0.f, shared_macroblock_convergence_percentage_ = 0.f;
};

}  // namespace io
}  // namespace pcl
