// pcl_lite.h -- the handful of PCL types the codec's public interface mentions, so that
// reference-style callers compile where PCL itself is not installed.  When the real PCL is
// available, do not include this file: point_cloud_codec_v2.h only needs pcl::PointXYZRGB,
// pcl::PointCloud<T> and a shared_ptr, and picks the real ones if PCL_POINT_TYPES_H_ is defined.
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <memory>
#include <vector>

// The reference's interface names two Eigen types (codec.h:64-68,224) and the app holds its clouds in
// boost::shared_ptr (eval.hpp:377,439-442).  Where those libraries are installed their headers come first and
// these stand-ins stay out of the way; where they are not, callers written against the reference still compile.
#if defined(__has_include)
#if __has_include(<Eigen/Core>)
#include <Eigen/Core>
#include <Eigen/StdVector>
#define PCC_SHIM_HAVE_EIGEN 1
#endif
#if __has_include(<boost/shared_ptr.hpp>)
#include <boost/shared_ptr.hpp>
#define PCC_SHIM_HAVE_BOOST 1
#endif
#endif

#ifndef PCC_SHIM_HAVE_EIGEN
namespace Eigen {
struct Vector4f {  // four floats with the accessors the codec's callers use
  float v[4];
  Vector4f() : v{0.f, 0.f, 0.f, 0.f} {}
  Vector4f(float a, float b, float c, float d) : v{a, b, c, d} {}
  float& operator[](int i) { return v[i]; }
  const float& operator[](int i) const { return v[i]; }
  float& operator()(int i) { return v[i]; }
  const float& operator()(int i) const { return v[i]; }
  float& x() { return v[0]; }
  float& y() { return v[1]; }
  float& z() { return v[2]; }
  float& w() { return v[3]; }
  const float& x() const { return v[0]; }
  const float& y() const { return v[1]; }
  const float& z() const { return v[2]; }
  const float& w() const { return v[3]; }
  float* data() { return v; }
  const float* data() const { return v; }
  Vector4f operator-(const Vector4f& o) const { return Vector4f(v[0] - o.v[0], v[1] - o.v[1], v[2] - o.v[2], v[3] - o.v[3]); }
  Vector4f operator+(const Vector4f& o) const { return Vector4f(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2], v[3] + o.v[3]); }
};
template <class T>
struct aligned_allocator : public std::allocator<T> {  // a distinct allocator type, as Eigen's is
  typedef T value_type;
  aligned_allocator() {}
  template <class U> aligned_allocator(const aligned_allocator<U>&) {}
  template <class U> struct rebind { typedef aligned_allocator<U> other; };
};
}  // namespace Eigen
#endif

#ifndef PCC_SHIM_HAVE_BOOST
namespace boost {
using std::shared_ptr;  // PCL 1.8-1.10 clouds are boost::shared_ptr; same interface for what the app does with them
}
#endif

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
  bool empty() const { return points.empty(); }
  void clear() { points.clear(); width = 0; height = 0; }
  void push_back(const PointT& p) { points.push_back(p); width = (uint32_t)points.size(); height = 1; }
  Ptr makeShared() const { return Ptr(new PointCloud<PointT>(*this)); }  // deep copy, as pcl::PointCloud::makeShared
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
