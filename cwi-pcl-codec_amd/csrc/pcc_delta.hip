// pcc_delta.hip -- kernels of the inter-frame ("delta") path:
//   shared-macroblock detection = intersection of the two frames' sorted macroblock keys ("Morton diff"),
//   the size / colour-variance gates of do_icp_prediction (impl.hpp:453-521),
//   one point-to-point ICP per shared block (the reference runs pcl::IterativeClosestPoint per block, in an
//   OpenMP loop over blocks, impl.hpp:544-567, 974-989): here one workgroup per block,
//   and the assembly of the residual (intra coded) points and of the predicted cloud.
// The macroblock trees themselves are hot-path runs (defined box [0,1]^3, resolution * macroblock size, stopped
// after k_leaf_scan): leaf = macroblock, sorted keys = points in block order.
//
// ICP: PCL is not in the reference tree, so its IterativeClosestPoint cannot be matched bit for bit ("parity
// unpinned"); what is restated is PCL 1.10's default pipeline: nearest-neighbour correspondences, Umeyama / SVD
// transformation estimation in float, DefaultConvergenceCriteria with the thresholds of impl.hpp:549-553.
#include <hip/hip_runtime.h>
#include <float.h>
#include <math.h>
#include <stdint.h>

#include "pcc_delta.h"

namespace pcc {
namespace {

constexpr int kDBlock = 256;

__device__ __forceinline__ uint32_t compact3_21(uint64_t x) {
  x &= 0x1249249249249249ULL;
  x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ULL;
  x = (x ^ (x >> 4)) & 0x100f00f00f00f00fULL;
  x = (x ^ (x >> 8)) & 0x1f0000ff0000ffULL;
  x = (x ^ (x >> 16)) & 0x1f00000000ffffULL;
  x = (x ^ (x >> 32)) & 0x1fffffULL;
  return (uint32_t)x;
}

__global__ __launch_bounds__(kDBlock) void k_block_keys(const uint64_t* __restrict__ leaf_code, uint64_t prefix, uint32_t n,
                                                        uint64_t* __restrict__ full) {
  const uint32_t j = blockIdx.x * kDBlock + threadIdx.x;
  if (j < n) full[j] = leaf_code[j] | prefix;
}

// points in block order: x, y, z and the colour word
__global__ __launch_bounds__(kDBlock) void k_block_points(BlockTree t, float4* __restrict__ out) {
  const uint32_t i = blockIdx.x * kDBlock + threadIdx.x;
  if (i >= t.n_points) return;
  const uint32_t idx = (uint32_t)(t.sorted_keys[i] & t.index_mask);
  const uint8_t* p = t.points + (size_t)idx * t.stride;
  const float* f = reinterpret_cast<const float*>(p);
  const uint32_t rgba = *reinterpret_cast<const uint32_t*>(p + t.rgb_off);
  out[i] = make_float4(f[0], f[1], f[2], __uint_as_float(rgba));
}

// one thread per macroblock of the predictive frame: partner in the I frame, gates, colour offsets
__global__ __launch_bounds__(kDBlock) void k_block_match(DeltaArgs a) {
  const uint32_t b = blockIdx.x * kDBlock + threadIdx.x;
  if (b >= a.p_tree.n_blocks) return;
  BlockResult r;
  const uint64_t key = a.p_full[b];
  r.key[0] = (uint16_t)compact3_21(key >> 2); r.key[1] = (uint16_t)compact3_21(key >> 1); r.key[2] = (uint16_t)compact3_21(key); r.key[3] = 0;
  uint32_t lo = 0, hi = a.i_tree.n_blocks;  // findLeaf (impl.hpp:843): binary search over the sorted keys of the I frame
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.i_full[mid] < key) lo = mid + 1; else hi = mid;
  }
  r.i_block = (lo < a.i_tree.n_blocks && a.i_full[lo] == key) ? (int32_t)lo : -1;
  const uint32_t p0 = a.p_tree.leaf_start[b], p1 = a.p_tree.leaf_start[b + 1];
  r.n_p = p1 - p0;
  r.n_i = 0;
  r.do_icp = 0; r.converged = 0; r.iterations = 0; r.fitness = 0.f;
  r.rgb_offsets[0] = r.rgb_offsets[1] = r.rgb_offsets[2] = r.rgb_offsets[3] = 0;
  for (int k = 0; k < 16; ++k) r.rt[k] = (k % 5 == 0) ? 1.f : 0.f;
  if (r.i_block >= 0) {
    const uint32_t i0 = a.i_tree.leaf_start[r.i_block], i1 = a.i_tree.leaf_start[r.i_block + 1];
    r.n_i = i1 - i0;
    // impl.hpp:453-456
    bool do_icp = r.n_p > 6 ? ((double)r.n_p < (double)r.n_i * 2) && ((double)r.n_p >= (double)r.n_i * 0.5) : false;
    if (do_icp) {  // colour means and variances, sequential double sums in point order (impl.hpp:470-517)
      double av[2][3], var[2];
      for (int s = 0; s < 2; ++s) {
        const float4* pts = s == 0 ? a.i_xyzc + i0 : a.p_xyzc + p0;
        const uint32_t n = s == 0 ? r.n_i : r.n_p;
        double m[3] = {0, 0, 0};
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t w = __float_as_uint(pts[k].w);
          m[0] += (double)((w >> 16) & 0xffu); m[1] += (double)((w >> 8) & 0xffu); m[2] += (double)(w & 0xffu);
        }
        for (int c = 0; c < 3; ++c) m[c] = __ddiv_rn(m[c], (double)n);
        double v = 0;
        for (uint32_t k = 0; k < n; ++k) {
          const uint32_t w = __float_as_uint(pts[k].w);
          const double dr = __dsub_rn((double)((w >> 16) & 0xffu), m[0]), dg = __dsub_rn((double)((w >> 8) & 0xffu), m[1]),
                       db = __dsub_rn((double)(w & 0xffu), m[2]);
          const double val = __dadd_rn(__dadd_rn(__dmul_rn(dr, dr), __dmul_rn(dg, dg)), __dmul_rn(db, db));
          v = __dadd_rn(v, val);
        }
        var[s] = __ddiv_rn(v, (double)(3 * n));
        for (int c = 0; c < 3; ++c) av[s][c] = m[c];
      }
      if (var[0] > (double)a.var_threshold || var[1] > (double)a.var_threshold) do_icp = false;
      if (a.do_icp_color_offset) {  // impl.hpp:528-535 (computed before the variance gate returns)
        for (int c = 0; c < 3; ++c) {
          const double d = __dsub_rn(av[1][c], av[0][c]);
          if (fabs(d) < 32) r.rgb_offsets[c] = (int8_t)d;
        }
      }
    }
    r.do_icp = do_icp ? 1 : 0;
  }
  a.results[b] = r;
}

// ---- ICP helpers ----
__device__ __forceinline__ float block_sum_f(float v, float* s_red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}
__device__ __forceinline__ double block_sum_d(double v, double* s_red) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__device__ __forceinline__ void nearest_in(const float4* __restrict__ tgt, uint32_t nt, float x, float y, float z, uint32_t& best_j, float& best) {
  best = FLT_MAX;
  best_j = 0;
  for (uint32_t j = 0; j < nt; ++j) {
    const float dx = x - tgt[j].x, dy = y - tgt[j].y, dz = z - tgt[j].z;
    float d = __fmul_rn(dx, dx);
    d = __fadd_rn(d, __fmul_rn(dy, dy));
    d = __fadd_rn(d, __fmul_rn(dz, dz));
    if (d < best) { best = d; best_j = j; }
  }
}

// pcl::transformPointCloud with a Matrix4f (PCL 1.10 Transformer::se3): x*c0 + (y*c1 + (z*c2 + c3)) per row, float
__device__ __forceinline__ void se3(const float* m, float x, float y, float z, float& ox, float& oy, float& oz) {
  ox = __fadd_rn(__fmul_rn(x, m[0]), __fadd_rn(__fmul_rn(y, m[1]), __fadd_rn(__fmul_rn(z, m[2]), m[3])));
  oy = __fadd_rn(__fmul_rn(x, m[4]), __fadd_rn(__fmul_rn(y, m[5]), __fadd_rn(__fmul_rn(z, m[6]), m[7])));
  oz = __fadd_rn(__fmul_rn(x, m[8]), __fadd_rn(__fmul_rn(y, m[9]), __fadd_rn(__fmul_rn(z, m[10]), m[11])));
}

// Rotation of Eigen::umeyama (without scaling) from the 3x3 covariance sigma = (1/n) sum (dst - dm)(src - sm)^T:
// R = U diag(1, 1, +-1) V^T.  SVD by Jacobi rotations on sigma^T sigma, in double.
__device__ void umeyama_rotation(const float sg[9], float Rm[9]) {
  double A[3][3], B[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) A[r][c] = (double)sg[3 * r + c];
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) B[r][c] = A[0][r] * A[0][c] + A[1][r] * A[1][c] + A[2][r] * A[2][c];
  for (int sweep = 0; sweep < 12; ++sweep) {
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        if (fabs(B[p][q]) < 1e-300) continue;
        const double theta = (B[q][q] - B[p][p]) / (2.0 * B[p][q]);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) { const double bkp = B[k][p], bkq = B[k][q]; B[k][p] = c * bkp - s * bkq; B[k][q] = s * bkp + c * bkq; }
        for (int k = 0; k < 3; ++k) { const double bpk = B[p][k], bqk = B[q][k]; B[p][k] = c * bpk - s * bqk; B[q][k] = s * bpk + c * bqk; }
        for (int k = 0; k < 3; ++k) { const double vkp = V[k][p], vkq = V[k][q]; V[k][p] = c * vkp - s * vkq; V[k][q] = s * vkp + c * vkq; }
      }
  }
  int ord[3] = {0, 1, 2};  // singular values descending
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (B[ord[j]][ord[j]] > B[ord[i]][ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
  double Vs[3][3], U[3][3], sv[3];
  for (int i = 0; i < 3; ++i) {
    sv[i] = sqrt(fmax(B[ord[i]][ord[i]], 0.0));
    for (int k = 0; k < 3; ++k) Vs[k][i] = V[k][ord[i]];
  }
  const double tiny = 1e-12 * fmax(sv[0], 1e-300);
  for (int i = 0; i < 3; ++i) {
    if (sv[i] > tiny) {
      for (int k = 0; k < 3; ++k) U[k][i] = (A[k][0] * Vs[0][i] + A[k][1] * Vs[1][i] + A[k][2] * Vs[2][i]) / sv[i];
    } else if (i == 2) {  // complete the basis
      U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
      U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
      U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    } else if (i == 1) {  // rank 1: any unit vector orthogonal to u0
      const double ax = fabs(U[0][0]), ay = fabs(U[1][0]), az = fabs(U[2][0]);
      double e[3] = {0, 0, 0};
      e[(ax <= ay && ax <= az) ? 0 : (ay <= az ? 1 : 2)] = 1.0;
      const double dot = e[0] * U[0][0] + e[1] * U[1][0] + e[2] * U[2][0];
      double w[3] = {e[0] - dot * U[0][0], e[1] - dot * U[1][0], e[2] - dot * U[2][0]};
      const double nw = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
      for (int k = 0; k < 3; ++k) U[k][1] = w[k] / nw;
    } else {  // zero matrix
      U[0][0] = 1; U[1][0] = 0; U[2][0] = 0;
    }
  }
  auto det3 = [](double M[3][3]) {
    return M[0][0] * (M[1][1] * M[2][2] - M[1][2] * M[2][1]) - M[0][1] * (M[1][0] * M[2][2] - M[1][2] * M[2][0]) +
           M[0][2] * (M[1][0] * M[2][1] - M[1][1] * M[2][0]);
  };
  const double s2 = det3(U) * det3(Vs) < 0 ? -1.0 : 1.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) Rm[3 * r + c] = (float)(U[r][0] * Vs[c][0] + U[r][1] * Vs[c][1] + s2 * U[r][2] * Vs[c][2]);
}

// one workgroup per macroblock of the predictive frame
__global__ __launch_bounds__(kDBlock) void k_block_icp(DeltaArgs a) {
  __shared__ float s_redf[kDBlock / 64];
  __shared__ double s_redd[kDBlock / 64];
  __shared__ float s_tr[16], s_final[16];
  __shared__ int s_flag;  // 0 go on, 1 converged, 2 failed
  const uint32_t b = blockIdx.x;
  const BlockResult r0 = a.results[b];
  if (!r0.do_icp) return;
  const uint32_t s0 = a.i_tree.leaf_start[r0.i_block], ns = r0.n_i;
  const uint32_t t0 = a.p_tree.leaf_start[b], nt = r0.n_p;
  const float4* src = a.i_xyzc + s0;  // source = the I frame's block, target = the predictive frame's block (impl.hpp:547-548)
  const float4* tgt = a.p_xyzc + t0;
  float4* cur = a.cur + s0;
  uint32_t* nn = a.nn + s0;
  for (uint32_t i = threadIdx.x; i < ns; i += kDBlock) cur[i] = src[i];
  if (threadIdx.x < 16) s_final[threadIdx.x] = (threadIdx.x % 5 == 0) ? 1.f : 0.f;
  if (threadIdx.x == 0) s_flag = 0;
  __syncthreads();
  const float fn = (float)ns;
  double prev_mse = DBL_MAX;
  int it = 0;
  const double rot_thr = 1.0 - (double)a.transformation_epsilon, trans_thr = (double)a.transformation_epsilon;
  const double mse_rel = 3.0 * (double)a.transformation_epsilon, mse_abs = 1e-12;
  while (true) {
    // correspondences: nearest target of every (moved) source point; sums for the means
    float sx = 0, sy = 0, sz = 0, tx = 0, ty = 0, tz = 0;
    double sd = 0;
    for (uint32_t i = threadIdx.x; i < ns; i += kDBlock) {
      const float4 p = cur[i];
      uint32_t j; float d;
      nearest_in(tgt, nt, p.x, p.y, p.z, j, d);
      nn[i] = j;
      sx += p.x; sy += p.y; sz += p.z;
      tx += tgt[j].x; ty += tgt[j].y; tz += tgt[j].z;
      sd += (double)d;
    }
    const float smx = block_sum_f(sx, s_redf) / fn, smy = block_sum_f(sy, s_redf) / fn, smz = block_sum_f(sz, s_redf) / fn;
    const float tmx = block_sum_f(tx, s_redf) / fn, tmy = block_sum_f(ty, s_redf) / fn, tmz = block_sum_f(tz, s_redf) / fn;
    const double mse = block_sum_d(sd, s_redd) / (double)ns;
    float sg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = threadIdx.x; i < ns; i += kDBlock) {
      const float4 p = cur[i];
      const float4 q = tgt[nn[i]];
      const float ds[3] = {p.x - smx, p.y - smy, p.z - smz}, dt[3] = {q.x - tmx, q.y - tmy, q.z - tmz};
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) sg[3 * r + c] += dt[r] * ds[c];
    }
    for (int k = 0; k < 9; ++k) sg[k] = block_sum_f(sg[k], s_redf) / fn;
    if (threadIdx.x == 0) {
      float Rm[9];
      umeyama_rotation(sg, Rm);
      float tr[16];
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) tr[4 * r + c] = Rm[3 * r + c];
        const float sm[3] = {smx, smy, smz}, tm[3] = {tmx, tmy, tmz};
        tr[4 * r + 3] = tm[r] - (Rm[3 * r] * sm[0] + Rm[3 * r + 1] * sm[1] + Rm[3 * r + 2] * sm[2]);
      }
      tr[12] = tr[13] = tr[14] = 0.f; tr[15] = 1.f;
      float nf[16];  // final = tr * final
      for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
          nf[4 * r + c] = ((tr[4 * r] * s_final[c] + tr[4 * r + 1] * s_final[4 + c]) + tr[4 * r + 2] * s_final[8 + c]) + tr[4 * r + 3] * s_final[12 + c];
      for (int k = 0; k < 16; ++k) { s_tr[k] = tr[k]; s_final[k] = nf[k]; }
      // DefaultConvergenceCriteria::hasConverged (PCL 1.10) with the settings of impl.hpp:549-553
      int flag = 0;
      if (it + 1 >= a.max_iterations) flag = 1;
      else {
        const double cos_angle = 0.5 * ((double)tr[0] + (double)tr[5] + (double)tr[10] - 1.0);
        const double tsq = (double)tr[3] * tr[3] + (double)tr[7] * tr[7] + (double)tr[11] * tr[11];
        if (cos_angle >= rot_thr && tsq <= trans_thr) flag = 1;
        else if (fabs(mse - prev_mse) < mse_abs || fabs(mse - prev_mse) / prev_mse < mse_rel) flag = 1;
      }
      s_flag = flag;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < ns; i += kDBlock) {  // move the source points
      const float4 p = cur[i];
      float4 o = p;
      se3(s_tr, p.x, p.y, p.z, o.x, o.y, o.z);
      cur[i] = o;
    }
    prev_mse = mse;
    ++it;
    const int flag = s_flag;
    __syncthreads();
    if (flag) break;
  }
  // getFitnessScore: mean squared distance of the source moved by the final transformation to its nearest target
  double fs = 0;
  for (uint32_t i = threadIdx.x; i < ns; i += kDBlock) {
    float x, y, z;
    se3(s_final, src[i].x, src[i].y, src[i].z, x, y, z);
    uint32_t j; float d;
    nearest_in(tgt, nt, x, y, z, j, d);
    fs += (double)d;
  }
  const double fitness = block_sum_d(fs, s_redd) / (double)ns;
  if (threadIdx.x == 0) {
    BlockResult* out = a.results + b;
    out->iterations = it;
    out->fitness = (float)fitness;
    out->converged = fitness < a.point_resolution * 2.0 ? 1 : 0;  // hasConverged() is true also at the iteration limit
    for (int k = 0; k < 16; ++k) out->rt[k] = s_final[k];
  }
}

__device__ __forceinline__ void store_point(uint8_t* dst, float x, float y, float z, uint32_t rgba) {
  float4* o = reinterpret_cast<float4*>(dst);
  o[0] = make_float4(x, y, z, 1.0f);
  reinterpret_cast<uint4*>(dst)[1] = make_uint4(rgba, 0u, 0u, 0u);
}

// one workgroup per macroblock: copy / predict its points to their places
__global__ __launch_bounds__(kDBlock) void k_delta_gather(GatherArgs a) {
  const uint32_t b = blockIdx.x;
  const BlockResult r = a.results[b];
  const uint32_t p0 = a.p_leaf_start[b], np = a.p_leaf_start[b + 1] - p0;
  const uint32_t di = a.dst_intra[b], dout = a.dst_out[b];
  if (di != 0xffffffffu)
    for (uint32_t k = threadIdx.x; k < np; k += kDBlock) {
      const float4 p = a.p_xyzc[p0 + k];
      store_point(a.out_intra + (size_t)(di + k) * 32, p.x, p.y, p.z, __float_as_uint(p.w));
    }
  if (dout == 0xffffffffu) return;
  if (di != 0xffffffffu) {  // not predicted: the block's own points (impl.hpp:918-935)
    for (uint32_t k = threadIdx.x; k < np; k += kDBlock) {
      const float4 p = a.p_xyzc[p0 + k];
      store_point(a.out_cloud + (size_t)(dout + k) * 32, p.x, p.y, p.z, __float_as_uint(p.w));
    }
  } else {  // predicted: the I frame's block moved by the DEcoded transform, colours offset (impl.hpp:885-910)
    const uint32_t i0 = a.i_leaf_start[r.i_block];
    const float* m = a.mdec + (size_t)b * 16;
    for (uint32_t k = threadIdx.x; k < r.n_i; k += kDBlock) {
      const float4 p = a.i_xyzc[i0 + k];
      float x, y, z;
      se3(m, p.x, p.y, p.z, x, y, z);
      uint32_t w = __float_as_uint(p.w);
      if (a.do_icp_color_offset) {
        const uint32_t mul = a.colour_doubled ? 2u : 1u;
        const uint32_t rr = (mul * ((w >> 16) & 0xffu) + (uint32_t)(int)r.rgb_offsets[0]) & 0xffu, gg = (mul * ((w >> 8) & 0xffu) + (uint32_t)(int)r.rgb_offsets[1]) & 0xffu,
                       bb = (mul * (w & 0xffu) + (uint32_t)(int)r.rgb_offsets[2]) & 0xffu;
        w = (w & 0xff000000u) | (rr << 16) | (gg << 8) | bb;
      }
      store_point(a.out_cloud + (size_t)(dout + k) * 32, x, y, z, w);
    }
  }
}

}  // namespace

void launch_delta_blocks(const DeltaArgs& a, hipStream_t stream) {
  const uint32_t nbi = a.i_tree.n_blocks, nbp = a.p_tree.n_blocks;
  if (nbi) hipLaunchKernelGGL(k_block_keys, dim3((nbi + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.i_tree.leaf_code, a.i_tree.prefix_code, nbi, a.i_full);
  if (nbp) hipLaunchKernelGGL(k_block_keys, dim3((nbp + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.p_tree.leaf_code, a.p_tree.prefix_code, nbp, a.p_full);
  if (a.i_tree.n_points) hipLaunchKernelGGL(k_block_points, dim3((a.i_tree.n_points + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.i_tree, a.i_xyzc);
  if (a.p_tree.n_points) hipLaunchKernelGGL(k_block_points, dim3((a.p_tree.n_points + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.p_tree, a.p_xyzc);
  if (!nbp) return;
  hipLaunchKernelGGL(k_block_match, dim3((nbp + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a);
  hipLaunchKernelGGL(k_block_icp, dim3(nbp), dim3(kDBlock), 0, stream, a);
}

void launch_delta_gather(const GatherArgs& a, hipStream_t stream) {
  if (a.n_blocks) hipLaunchKernelGGL(k_delta_gather, dim3(a.n_blocks), dim3(kDBlock), 0, stream, a);
}

}  // namespace pcc
