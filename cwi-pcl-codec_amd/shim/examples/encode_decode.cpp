// encode_decode.cpp -- a caller written exactly like the reference app's encode/decode loop
// (evaluate_compression_impl.hpp:377-395 constructs the codecs, :463 encodes, :490 decodes),
// compiled against the drop-in header.  Build:  make -C cwi-pcl-codec_amd/shim/examples
#include <pcl/cloud_codec_v2/point_cloud_codec_v2.h>

#include <cmath>
#include <cstdio>
#include <sstream>
#include <streambuf>

// a stream that can only be read forwards (a socket, a pipe): tellg() is -1 and seekg() fails
struct forward_only_buf : std::streambuf {
  std::string data;
  explicit forward_only_buf(const std::string& s) : data(s) { setg(&data[0], &data[0], &data[0] + data.size()); }
};

typedef pcl::PointXYZRGB PointT;
typedef pcl::io::OctreePointCloudCodecV2<PointT> Codec;

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 100000;
  const int octree_bits = argc > 2 ? atoi(argv[2]) : 8;
  pcl::PointCloud<PointT>::Ptr cloud(new pcl::PointCloud<PointT>());
  uint64_t s = 0x9E3779B97F4A7C15ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / (double)(1ull << 53); };
  for (int i = 0; i < n; ++i) {  // a sphere shell, like SURVEY.md's cfg1
    const double ct = 2 * rnd() - 1, st = std::sqrt(1 - ct * ct), ph = 6.283185307179586 * rnd();
    PointT p;
    p.x = (float)(0.5 + 0.3 * st * std::cos(ph)); p.y = (float)(0.5 + 0.3 * st * std::sin(ph)); p.z = (float)(0.5 + 0.3 * ct);
    p.r = (uint8_t)(255 * p.x); p.g = (uint8_t)(255 * p.y); p.b = (uint8_t)(255 * p.z);
    cloud->points.push_back(p);
  }
  std::vector<pcl::PointCloud<PointT>::Ptr> group(1, cloud);
  std::vector<pcl::io::BoundingBox> boxes;
  pcl::io::BoundingBox bb = Codec::normalize_pointclouds(group, boxes, 0.2);

  const double res = std::pow(2.0, -1.0 * octree_bits);
  Codec encoder(pcl::io::MANUAL_CONFIGURATION, false, res, res, true, 0, true, 8, 1, false, false, false, 85, 1);
  Codec decoder(pcl::io::MANUAL_CONFIGURATION, false, res, res, true, 0, true, 8, 1, false, false, false, 85, 1);
  std::stringstream stream;
  encoder.encodePointCloud(cloud, stream);
  uint64_t* sizes = encoder.getPerformanceMetrics();
  pcl::PointCloud<PointT>::Ptr out(new pcl::PointCloud<PointT>());
  decoder.decodePointCloud(stream, out);
  Codec::restore_scaling(out, bb);
  std::printf("points in %d, compressed %zu bytes (octree %llu, centroid %llu, colour %llu), decoded voxels %zu\n", n,
              stream.str().size(), (unsigned long long)sizes[0], (unsigned long long)sizes[1],
              (unsigned long long)sizes[2], out->points.size());
  if (out->points.size() != encoder.getOutputCloud()->points.size()) return 1;

  // Several frames in ONE stream (the reference decodes them one call at a time, each call consuming exactly one
  // frame, impl.hpp:224-310): once through a seekable stream, once through a forward-only one, with garbage in front.
  std::stringstream many;
  many << "junk before the first header";
  const int kFrames = 3;
  for (int f = 0; f < kFrames; ++f) encoder.encodePointCloud(cloud, many);
  const std::string all = many.str();
  forward_only_buf fb(all);
  std::istream forward(&fb);
  std::istream* inputs[2] = {&many, &forward};
  for (int k = 0; k < 2; ++k) {
    for (int f = 0; f < kFrames; ++f) {
      pcl::PointCloud<PointT>::Ptr o(new pcl::PointCloud<PointT>());
      decoder.decodePointCloud(*inputs[k], o);
      if (o->points.size() != out->points.size()) { std::printf("stream %d frame %d: %zu voxels\n", k, f, o->points.size()); return 2; }
    }
    pcl::PointCloud<PointT>::Ptr none(new pcl::PointCloud<PointT>());
    decoder.decodePointCloud(*inputs[k], none);  // nothing left: output untouched (impl.hpp:231)
    if (!none->points.empty()) return 3;
  }
  std::printf("%d frames decoded one by one from a seekable and from a forward-only stream\n", kFrames);
  return 0;
}
