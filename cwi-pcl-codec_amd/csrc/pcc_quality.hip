// pcc_quality.hip -- the quality metric of the reference's evaluation app on the GPU
// (apps/evaluate_compression quality_metrics_impl.hpp:82-239):
// for every point of cloud A the nearest point of cloud B (squared distance + YUV colour error),
// for every point of B the nearest point of A (squared distance), then symmetric RMS / Hausdorff /
// geometric PSNR and per-channel colour PSNR.
//
// The reference answers the nearest-neighbour queries with pcl::search::KdTree (FLANN, exact,
// L2_Simple on float).  Here the target cloud is binned into a uniform grid held in an open-addressing
// hash table (cell code -> head of a linked list of points); a query walks the 3x3x3 cells around its
// own cell and is exact as soon as the best distance is not larger than the cell size (for a decoded
// cloud against its original that is always the case when the cell is the voxel size); otherwise the
// search cube grows.  Squared distances are accumulated in float in x, y, z order like FLANN's
// L2_Simple; equal distances resolve to the lower point index (FLANN's tie order is not specified).
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "pcc_quality.h"

namespace pcc {

namespace {

constexpr int kQBlock = 256;
constexpr uint32_t kNoPoint = 0xffffffffu;
constexpr int kMaxRadius = 4;  // cells; beyond that the query scans the whole target cloud

struct GridParams {
  float origin[3];
  float inv_cell;
  float cell;
  uint32_t table_mask;
};

struct QPoint {  // pcl::PointXYZRGB, 32 bytes
  float x, y, z, w;
  uint32_t rgba;
  uint32_t pad[3];
};

__device__ __forceinline__ bool finite3(float x, float y, float z) { return isfinite(x) && isfinite(y) && isfinite(z); }

__device__ __forceinline__ void cell_of(const GridParams& g, float x, float y, float z, int c[3]) {
  c[0] = (int)floorf((x - g.origin[0]) * g.inv_cell);
  c[1] = (int)floorf((y - g.origin[1]) * g.inv_cell);
  c[2] = (int)floorf((z - g.origin[2]) * g.inv_cell);
}
// cells are addressed with 21 bits per axis (offset so that small negative indices stay valid); 0 = empty slot
__device__ __forceinline__ uint64_t cell_code(int cx, int cy, int cz) {
  const uint64_t x = (uint64_t)(uint32_t)(cx + 1024) & 0x1fffffu, y = (uint64_t)(uint32_t)(cy + 1024) & 0x1fffffu,
                 z = (uint64_t)(uint32_t)(cz + 1024) & 0x1fffffu;
  return ((x << 42) | (y << 21) | z) + 1ull;
}
__device__ __forceinline__ uint32_t hash_code(uint64_t code) {
  code ^= code >> 33; code *= 0xff51afd7ed558ccdULL; code ^= code >> 33; code *= 0xc4ceb9fe1a85ec53ULL; code ^= code >> 33;
  return (uint32_t)code;
}

__global__ __launch_bounds__(kQBlock) void k_nn_clear(unsigned long long* keys, uint32_t* heads, uint32_t slots) {
  for (uint32_t i = blockIdx.x * kQBlock + threadIdx.x; i < slots; i += gridDim.x * kQBlock) { keys[i] = 0ull; heads[i] = kNoPoint; }
}

__global__ __launch_bounds__(kQBlock) void k_nn_build(const QPoint* __restrict__ t, uint32_t n, GridParams g,
                                                      unsigned long long* keys, uint32_t* heads, uint32_t* __restrict__ next) {
  const uint32_t i = blockIdx.x * kQBlock + threadIdx.x;
  if (i >= n) return;
  const QPoint p = t[i];
  if (!finite3(p.x, p.y, p.z)) { next[i] = kNoPoint; return; }  // the KdTree leaves non-finite points out
  int c[3];
  cell_of(g, p.x, p.y, p.z, c);
  const unsigned long long code = cell_code(c[0], c[1], c[2]);
  uint32_t slot = hash_code(code) & g.table_mask;
  for (;;) {
    const unsigned long long prev = atomicCAS(&keys[slot], 0ull, code);
    if (prev == 0ull || prev == code) break;
    slot = (slot + 1u) & g.table_mask;
  }
  next[i] = atomicExch(&heads[slot], i);
}

__device__ __forceinline__ void visit_cell(const QPoint* __restrict__ t, const GridParams& g, const unsigned long long* __restrict__ keys,
                                           const uint32_t* __restrict__ heads, const uint32_t* __restrict__ next, int cx, int cy, int cz,
                                           float qx, float qy, float qz, float& best, uint32_t& best_i) {
  const unsigned long long code = cell_code(cx, cy, cz);
  uint32_t slot = hash_code(code) & g.table_mask;
  for (;;) {
    const unsigned long long k = keys[slot];
    if (k == code) break;
    if (k == 0ull) return;
    slot = (slot + 1u) & g.table_mask;
  }
  for (uint32_t i = heads[slot]; i != kNoPoint; i = next[i]) {
    const float dx = qx - t[i].x, dy = qy - t[i].y, dz = qz - t[i].z;
    float d = __fmul_rn(dx, dx);                 // FLANN L2_Simple: result += diff * diff, float, in order
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    if (d < best || (d == best && i < best_i)) { best = d; best_i = i; }
  }
}

// one thread per query point: nearest target point (index and squared distance)
__global__ __launch_bounds__(kQBlock) void k_nn_query(const QPoint* __restrict__ q, uint32_t nq, const QPoint* __restrict__ t, uint32_t nt,
                                                      GridParams g, const unsigned long long* __restrict__ keys,
                                                      const uint32_t* __restrict__ heads, const uint32_t* __restrict__ next,
                                                      float* __restrict__ out_d2, uint32_t* __restrict__ out_idx) {
  const uint32_t i = blockIdx.x * kQBlock + threadIdx.x;
  if (i >= nq) return;
  const float qx = q[i].x, qy = q[i].y, qz = q[i].z;
  float best = FLT_MAX;
  uint32_t best_i = kNoPoint;
  if (finite3(qx, qy, qz)) {
    int c[3];
    cell_of(g, qx, qy, qz, c);
    bool exact = false;
    for (int r = 1; r <= kMaxRadius && !exact; ++r) {
      for (int dz = -r; dz <= r; ++dz)
        for (int dy = -r; dy <= r; ++dy)
          for (int dx = -r; dx <= r; ++dx) {
            if (r > 1 && abs(dx) < r && abs(dy) < r && abs(dz) < r) continue;  // the inner cube was searched already
            visit_cell(t, g, keys, heads, next, c[0] + dx, c[1] + dy, c[2] + dz, qx, qy, qz, best, best_i);
          }
      // everything outside the searched cube is at least r cells away (minus rounding of the cell index)
      const float reach = (float)r * g.cell * 0.9999f;
      exact = best_i != kNoPoint && best <= reach * reach;
    }
    if (!exact) {  // far from everything: scan the target cloud
      for (uint32_t k = 0; k < nt; ++k) {
        if (!finite3(t[k].x, t[k].y, t[k].z)) continue;
        const float dx = qx - t[k].x, dy = qy - t[k].y, dz = qz - t[k].z;
        float d = __fmul_rn(dx, dx);
        d = __fadd_rn(d, __fmul_rn(dy, dy));
        d = __fadd_rn(d, __fmul_rn(dz, dz));
        if (d < best || (d == best && k < best_i)) { best = d; best_i = k; }
      }
    }
  }
  out_d2[i] = best;
  out_idx[i] = best_i;
}

__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_fl(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// convertRGBtoYUV (quality_metrics_impl.hpp:63-70): double arithmetic, stored to float
__device__ __forceinline__ void rgb_to_yuv(uint32_t rgba, float yuv[3]) {
  const double r = (double)((rgba >> 16) & 0xffu), g = (double)((rgba >> 8) & 0xffu), b = (double)(rgba & 0xffu);
  yuv[0] = (float)__ddiv_rn(__dadd_rn(__dadd_rn(__dmul_rn(0.299, r), __dmul_rn(0.587, g)), __dmul_rn(0.114, b)), 255.0);
  yuv[1] = (float)__ddiv_rn(__dadd_rn(__dsub_rn(__dmul_rn(-0.147, r), __dmul_rn(0.289, g)), __dmul_rn(0.436, b)), 255.0);
  yuv[2] = (float)__ddiv_rn(__dsub_rn(__dsub_rn(__dmul_rn(0.615, r), __dmul_rn(0.515, g)), __dmul_rn(0.100, b)), 255.0);
}

// per-workgroup partial sums: [0] sum d2, [1..3] sum of squared Y, U, V errors, [4] max d2, [5..7] max x, y, z of the queries
__global__ __launch_bounds__(kQBlock) void k_quality_partials(const QPoint* __restrict__ q, uint32_t nq, const QPoint* __restrict__ t,
                                                              const float* __restrict__ d2, const uint32_t* __restrict__ idx,
                                                              int with_colour, double* __restrict__ partials) {
  __shared__ double s_sum[4][kQBlock / 64];
  __shared__ float s_max[4][kQBlock / 64];
  const uint32_t i = blockIdx.x * kQBlock + threadIdx.x;
  double sd = 0.0, se[3] = {0.0, 0.0, 0.0};
  float md = -FLT_MAX, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < nq && idx[i] != kNoPoint) {
    sd = (double)d2[i];
    md = d2[i];
    mx[0] = q[i].x; mx[1] = q[i].y; mx[2] = q[i].z;
    if (with_colour) {
      float a[3], b[3];
      rgb_to_yuv(q[i].rgba, a);
      rgb_to_yuv(t[idx[i]].rgba, b);
      for (int c = 0; c < 3; ++c) {
        const float e = __fsub_rn(a[c], b[c]);
        se[c] = (double)__fmul_rn(e, e);
      }
    }
  }
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  sd = wave_sum_d(sd); md = wave_max_fl(md);
  for (int c = 0; c < 3; ++c) { se[c] = wave_sum_d(se[c]); mx[c] = wave_max_fl(mx[c]); }
  if (lane == 0) {
    s_sum[0][w] = sd; s_max[0][w] = md;
    for (int c = 0; c < 3; ++c) { s_sum[1 + c][w] = se[c]; s_max[1 + c][w] = mx[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double* out = partials + (size_t)blockIdx.x * 8;
    if (threadIdx.x < 4) {
      double v = 0.0;
      for (int k = 0; k < kQBlock / 64; ++k) v += s_sum[threadIdx.x][k];
      out[threadIdx.x] = v;
    } else {
      float v = -FLT_MAX;
      for (int k = 0; k < kQBlock / 64; ++k) v = fmaxf(v, s_max[threadIdx.x - 4][k]);
      out[threadIdx.x] = (double)v;
    }
  }
}


// pcl::RadiusOutlierRemoval as the reference uses it (remove_outliers, impl.hpp:1840-1866): a point stays if at least
// `min_points` OTHER points lie within `radius` (squared float distance <= radius^2).  Grid cell = radius: every such
// neighbour is in the 3x3x3 cells around the point's own.  The walk stops as soon as enough neighbours were seen.
__global__ __launch_bounds__(kQBlock) void k_radius_keep(const QPoint* __restrict__ t, uint32_t n, GridParams g, const unsigned long long* __restrict__ keys,
                                                         const uint32_t* __restrict__ heads, const uint32_t* __restrict__ next, float r2,
                                                         uint32_t min_points, uint8_t* __restrict__ keep) {
  const uint32_t i = blockIdx.x * kQBlock + threadIdx.x;
  if (i >= n) return;
  const float qx = t[i].x, qy = t[i].y, qz = t[i].z;
  if (!finite3(qx, qy, qz)) { keep[i] = 0; return; }
  int c[3];
  cell_of(g, qx, qy, qz, c);
  uint32_t found = 0;
  for (int dz = -1; dz <= 1 && found < min_points; ++dz)
    for (int dy = -1; dy <= 1 && found < min_points; ++dy)
      for (int dx = -1; dx <= 1 && found < min_points; ++dx) {
        const unsigned long long code = cell_code(c[0] + dx, c[1] + dy, c[2] + dz);
        uint32_t slot = hash_code(code) & g.table_mask;
        bool present = true;
        for (;;) {
          const unsigned long long k = keys[slot];
          if (k == code) break;
          if (k == 0ull) { present = false; break; }
          slot = (slot + 1u) & g.table_mask;
        }
        if (!present) continue;
        for (uint32_t j = heads[slot]; j != kNoPoint && found < min_points; j = next[j]) {
          if (j == i) continue;
          const float ex = qx - t[j].x, ey = qy - t[j].y, ez = qz - t[j].z;
          float d = __fmul_rn(ex, ex);
          d = __fadd_rn(d, __fmul_rn(ey, ey));
          d = __fadd_rn(d, __fmul_rn(ez, ez));
          if (d <= r2) ++found;
        }
      }
  keep[i] = found >= min_points ? 1 : 0;
}

}  // namespace

size_t quality_table_slots(size_t n_target) {
  size_t s = 1024;
  while (s < 2 * n_target) s <<= 1;
  return s;
}

void launch_quality_direction(const QualityArgs& a, hipStream_t stream) {
  GridParams g;
  for (int k = 0; k < 3; ++k) g.origin[k] = a.origin[k];
  g.cell = a.cell;
  g.inv_cell = 1.0f / a.cell;
  g.table_mask = (uint32_t)(a.table_slots - 1);
  const QPoint* q = reinterpret_cast<const QPoint*>(a.query);
  const QPoint* t = reinterpret_cast<const QPoint*>(a.target);
  hipLaunchKernelGGL(k_nn_clear, dim3(1024), dim3(kQBlock), 0, stream, a.keys, a.heads, (uint32_t)a.table_slots);
  hipLaunchKernelGGL(k_nn_build, dim3((a.n_target + kQBlock - 1) / kQBlock), dim3(kQBlock), 0, stream, t, a.n_target, g, a.keys, a.heads, a.next);
  const uint32_t qb = (a.n_query + kQBlock - 1) / kQBlock;
  hipLaunchKernelGGL(k_nn_query, dim3(qb), dim3(kQBlock), 0, stream, q, a.n_query, t, a.n_target, g, a.keys, a.heads, a.next, a.d2, a.idx);
  hipLaunchKernelGGL(k_quality_partials, dim3(qb), dim3(kQBlock), 0, stream, q, a.n_query, t, a.d2, a.idx, a.with_colour, a.partials);
}

void launch_radius_filter(const RadiusArgs& a, hipStream_t stream) {
  GridParams g;
  for (int k = 0; k < 3; ++k) g.origin[k] = a.origin[k];
  g.cell = a.radius;
  g.inv_cell = 1.0f / a.radius;
  g.table_mask = (uint32_t)(a.table_slots - 1);
  const QPoint* t = reinterpret_cast<const QPoint*>(a.cloud);
  const uint32_t nb = (a.n + kQBlock - 1) / kQBlock;
  hipLaunchKernelGGL(k_nn_clear, dim3(1024), dim3(kQBlock), 0, stream, a.keys, a.heads, (uint32_t)a.table_slots);
  hipLaunchKernelGGL(k_nn_build, dim3(nb), dim3(kQBlock), 0, stream, t, a.n, g, a.keys, a.heads, a.next);
  hipLaunchKernelGGL(k_radius_keep, dim3(nb), dim3(kQBlock), 0, stream, t, a.n, g, a.keys, a.heads, a.next, a.radius * a.radius, a.min_points, a.keep);
}

}  // namespace pcc
