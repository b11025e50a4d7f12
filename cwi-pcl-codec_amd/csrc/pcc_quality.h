// pcc_quality.h -- launch interface of the quality-metric kernels (pcc_quality.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pcc {

// one direction of the metric: every point of `query` finds its nearest point in `target`
struct QualityArgs {
  const void* query;   // pcl::PointXYZRGB (32 bytes) on the device
  const void* target;
  uint32_t n_query, n_target;
  float origin[3];     // grid origin (below every target point)
  float cell;          // grid cell size
  size_t table_slots;  // power of two >= 2 * n_target
  unsigned long long* keys;  // [table_slots]
  uint32_t* heads;           // [table_slots]
  uint32_t* next;            // [n_target]
  float* d2;                 // [n_query] squared distance to the nearest target point
  uint32_t* idx;             // [n_query] its index
  int with_colour;           // also sum the YUV errors
  double* partials;          // [ceil(n_query / 256)][8] per-workgroup partial sums / maxima
};

// pcl::RadiusOutlierRemoval (remove_outliers, impl.hpp:1840-1866): keep[i] = 1 if at least min_points other points lie within radius
struct RadiusArgs {
  const void* cloud;  // pcl::PointXYZRGB on the device
  uint32_t n;
  float origin[3];    // grid origin (below every point)
  float radius;       // = grid cell size
  uint32_t min_points;
  size_t table_slots;
  unsigned long long* keys;
  uint32_t* heads;
  uint32_t* next;
  uint8_t* keep;      // [n]
};
void launch_radius_filter(const RadiusArgs& a, hipStream_t stream);

size_t quality_table_slots(size_t n_target);
void launch_quality_direction(const QualityArgs& a, hipStream_t stream);

}  // namespace pcc
