// ref_codec_driver.cpp -- C entry points around the REAL reference codec (test infrastructure, not product).
//
// Compiled by oracle/pin_with_pcl.sh -- only where a PCL 1.8.1-1.10 installation exists (it does not in the build
// container nor on the GPU boxes) -- from the reference's own sources where they lie under
// $REF (/root/reference): this file includes the reference's header and implementation and instantiates
// pcl::io::OctreePointCloudCodecV2<pcl::PointXYZRGB> exactly like cloud_codec_v2/src/point_cloud_codec_v2.cpp:45.
// Nothing of the reference is copied into this repository.  oracle/pin_check.py drives the functions below with the
// clouds of tests/golden/make_codec_golden.py and compares every byte with the CPU oracle.
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include <pcl/cloud_codec_v2/point_cloud_codec_v2.h>
#include <pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp>

#include <cstdint>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

template class pcl::io::OctreePointCloudCodecV2<pcl::PointXYZRGB>;

namespace {
typedef pcl::PointXYZRGB PointT;
typedef pcl::io::OctreePointCloudCodecV2<PointT> Codec;

struct Point32 {  // the layout the oracle and the product use: pcl::PointXYZRGB, 32 bytes, colour word at 16
  float x, y, z, w;
  uint32_t rgba;
  uint32_t pad[3];
};
static_assert(sizeof(Point32) == 32 && sizeof(PointT) == 32, "PointXYZRGB is 32 bytes");

std::string g_stream;
std::vector<Point32> g_cloud;

pcl::PointCloud<PointT>::Ptr to_cloud(const Point32* p, size_t n) {
  pcl::PointCloud<PointT>::Ptr c(new pcl::PointCloud<PointT>());
  c->points.resize(n);
  if (n) std::memcpy(static_cast<void*>(c->points.data()), p, n * sizeof(Point32));
  c->width = (uint32_t)n;
  c->height = 1;
  c->is_dense = false;
  return c;
}
void from_cloud(const pcl::PointCloud<PointT>& c) {
  g_cloud.resize(c.points.size());
  if (!c.points.empty()) std::memcpy(g_cloud.data(), static_cast<const void*>(c.points.data()), c.points.size() * sizeof(Point32));
}
Codec* make(double point_res, double octree_res, int color_bits, int color_coding_type, int keep_centroid, int jpeg_quality) {
  // the argument list of eval.hpp:377-395
  return new Codec(pcl::io::MANUAL_CONFIGURATION, false, point_res, octree_res, true, 0, color_bits > 0, (unsigned char)color_bits,
                   (unsigned char)color_coding_type, keep_centroid != 0, false, false, jpeg_quality, 1);
}
}  // namespace

extern "C" {

// encodePointCloud on a fresh codec object, `frame_id - 1` empty frames... the reference has no setter for frame_ID_: the
// id is advanced by encoding the same cloud `frame_id` times and keeping the last stream.
// Returns the stream length; *perf = getPerformanceMetrics(); the simplified cloud (getOutputCloud) is kept for ref_cloud().
size_t ref_encode(const Point32* pts, size_t n, double point_res, double octree_res, int color_bits, int color_coding_type,
                  int keep_centroid, int jpeg_quality, unsigned frame_id, uint64_t perf[3]) {
  Codec* enc = make(point_res, octree_res, color_bits, color_coding_type, keep_centroid, jpeg_quality);
  pcl::PointCloud<PointT>::Ptr cloud = to_cloud(pts, n);
  g_stream.clear();
  for (unsigned k = 0; k < (frame_id ? frame_id : 1u); ++k) {
    std::stringstream ss;
    enc->encodePointCloud(cloud, ss);
    g_stream = ss.str();
  }
  const std::uint64_t* m = enc->getPerformanceMetrics();
  for (int i = 0; i < 3; ++i) perf[i] = m[i];
  if (enc->getOutputCloud()) from_cloud(*enc->getOutputCloud()); else g_cloud.clear();
  delete enc;
  return g_stream.size();
}
const char* ref_stream() { return g_stream.data(); }
size_t ref_cloud_size() { return g_cloud.size(); }
const Point32* ref_cloud() { return g_cloud.data(); }

// decodePointCloud on a fresh codec object; the decoded cloud is kept for ref_cloud()
size_t ref_decode(const char* stream, size_t len, double point_res, double octree_res, int color_bits, int color_coding_type,
                  int keep_centroid) {
  Codec* dec = make(point_res, octree_res, color_bits, color_coding_type, keep_centroid, 75);
  std::stringstream ss(std::string(stream, len));
  pcl::PointCloud<PointT>::Ptr out(new pcl::PointCloud<PointT>());
  dec->decodePointCloud(ss, out);
  from_cloud(*out);
  delete dec;
  return g_cloud.size();
}

}  // extern "C"
