// pcl_lite.h -- the handful of PCL types the codec's public interface mentions, so that
// reference-style callers compile where PCL itself is not installed.  When the real PCL is
// available, do not include this file: point_cloud_codec_v2.h only needs pcl::PointXYZRGB,
// pcl::PointCloud<T> and a shared_ptr, and picks the real ones if PCL_POINT_TYPES_H_ is defined.
#pragma once
#include <stdint.h>

#include <memory>
#include <vector>

namespace pcl {

// Memory layout of pcl::PointXYZRGB (PCL point_types.hpp): 32 bytes, colour word at offset 16.
struct alignas(16) PointXYZRGB {
  union {
    float data[4];
    struct { float x, y, z; };
  };
  union {
    struct { uint8_t b, g, r, a; };
    float rgb;
    uint32_t rgba;
  };
  uint32_t pad_[3];
  PointXYZRGB() : pad_{0, 0, 0} {
    x = y = z = 0.0f;
    data[3] = 1.0f;
    r = g = b = 0;
    a = 255;
  }
};
static_assert(sizeof(PointXYZRGB) == 32, "PointXYZRGB must be 32 bytes");

template <typename PointT>
struct PointCloud {
  typedef std::shared_ptr<PointCloud<PointT>> Ptr;
  typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
  std::vector<PointT> points;
  uint32_t width = 0, height = 0;
  bool is_dense = true;
  size_t size() const { return points.size(); }
  PointT& at(size_t i) { return points.at(i); }
  const PointT& at(size_t i) const { return points.at(i); }
  PointT& operator[](size_t i) { return points[i]; }
  const PointT& operator[](size_t i) const { return points[i]; }
};

namespace io {
enum compression_Profiles_e {
  LOW_RES_ONLINE_COMPRESSION_WITHOUT_COLOR, LOW_RES_ONLINE_COMPRESSION_WITH_COLOR,
  MED_RES_ONLINE_COMPRESSION_WITHOUT_COLOR, MED_RES_ONLINE_COMPRESSION_WITH_COLOR,
  HIGH_RES_ONLINE_COMPRESSION_WITHOUT_COLOR, HIGH_RES_ONLINE_COMPRESSION_WITH_COLOR,
  LOW_RES_OFFLINE_COMPRESSION_WITHOUT_COLOR, LOW_RES_OFFLINE_COMPRESSION_WITH_COLOR,
  MED_RES_OFFLINE_COMPRESSION_WITHOUT_COLOR, MED_RES_OFFLINE_COMPRESSION_WITH_COLOR,
  HIGH_RES_OFFLINE_COMPRESSION_WITHOUT_COLOR, HIGH_RES_OFFLINE_COMPRESSION_WITH_COLOR,
  COMPRESSION_PROFILE_COUNT, MANUAL_CONFIGURATION
};
}  // namespace io
}  // namespace pcl
