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
#include <stdlib.h>

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
  // two lanes per macroblock: lane 0 takes the I frame's points, lane 1 the P frame's (the colour statistics are
  // sequential sums in point order, impl.hpp:470-517: the only parallelism inside a block is between its two clouds)
  const uint32_t t = blockIdx.x * kDBlock + threadIdx.x;
  const uint32_t b = min(t >> 1, a.p_tree.n_blocks - 1);
  const bool owner = (t >> 1) < a.p_tree.n_blocks && (t & 1u) == 0;
  const int side = (int)(t & 1u);
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
  uint32_t i0 = 0;
  bool do_icp = false;
  if (r.i_block >= 0) {
    i0 = a.i_tree.leaf_start[r.i_block];
    r.n_i = a.i_tree.leaf_start[r.i_block + 1] - i0;
    // impl.hpp:453-456
    do_icp = r.n_p > 6 ? ((double)r.n_p < (double)r.n_i * 2) && ((double)r.n_p >= (double)r.n_i * 0.5) : false;
  }
  double m[3] = {0, 0, 0}, var = 0;
  if (do_icp) {  // colour mean and variance of this lane's cloud
    const float4* pts = side == 0 ? a.i_xyzc + i0 : a.p_xyzc + p0;
    const uint32_t n = side == 0 ? r.n_i : r.n_p;
    uint32_t sr = 0, sg = 0, sb = 0;  // sums of 8-bit values: exact in any arithmetic (the reference adds them in double)
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t w = __float_as_uint(pts[k].w);
      sr += (w >> 16) & 0xffu; sg += (w >> 8) & 0xffu; sb += w & 0xffu;
    }
    m[0] = __ddiv_rn((double)sr, (double)n); m[1] = __ddiv_rn((double)sg, (double)n); m[2] = __ddiv_rn((double)sb, (double)n);
    double v = 0;
    for (uint32_t k = 0; k < n; ++k) {
      const uint32_t w = __float_as_uint(pts[k].w);
      const double dr = __dsub_rn((double)((w >> 16) & 0xffu), m[0]), dg = __dsub_rn((double)((w >> 8) & 0xffu), m[1]),
                   db = __dsub_rn((double)(w & 0xffu), m[2]);
      const double val = __dadd_rn(__dadd_rn(__dmul_rn(dr, dr), __dmul_rn(dg, dg)), __dmul_rn(db, db));
      v = __dadd_rn(v, val);
    }
    var = __ddiv_rn(v, (double)(3 * n));
  }
  // the partner lane's numbers (lanes 2k and 2k+1 hold the same block and took the same branches)
  const double om0 = __shfl_xor(m[0], 1), om1 = __shfl_xor(m[1], 1), om2 = __shfl_xor(m[2], 1), ovar = __shfl_xor(var, 1);
  if (!owner) return;
  if (do_icp) {
    if (var > (double)a.var_threshold || ovar > (double)a.var_threshold) do_icp = false;
    if (a.do_icp_color_offset) {  // impl.hpp:528-535 (computed before the variance gate returns)
      const double d[3] = {__dsub_rn(om0, m[0]), __dsub_rn(om1, m[1]), __dsub_rn(om2, m[2])};  // P mean - I mean
      for (int c = 0; c < 3; ++c)
        if (fabs(d[c]) < 32) r.rgb_offsets[c] = (int8_t)d[c];
    }
  }
  r.do_icp = do_icp ? 1 : 0;
  a.results[b] = r;
  a.work[b] = do_icp ? (uint32_t)min((uint64_t)r.n_i * r.n_p, (uint64_t)0xffffffffu) : 0u;  // distance evaluations per ICP iteration
}

// Order of the ICP work: macroblocks by falling n_source x n_target (power-of-two classes; the order inside a class is
// whatever the atomics give -- it only decides which wave takes which block, not any result).  Long blocks first keeps
// the tail of the ICP kernel short; blocks above kIcpHeavy go to the workgroup-per-block shape.
constexpr uint32_t kIcpHeavy = 128u * 128u;
__global__ __launch_bounds__(1024) void k_block_order(const uint32_t* __restrict__ work, uint32_t n, uint32_t* __restrict__ order, uint32_t* __restrict__ counts) {
  __shared__ uint32_t s_count[33], s_start[33], s_heavy;
  if (threadIdx.x < 33) s_count[threadIdx.x] = 0u;
  if (threadIdx.x == 0) s_heavy = 0u;
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n; b += 1024u) {
    const uint32_t w = work[b];
    if (!w) continue;
    atomicAdd(&s_count[32 - __clz(w)], 1u);  // class 1..32
    if (w >= kIcpHeavy) atomicAdd(&s_heavy, 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t at = 0;
    for (int c = 32; c >= 1; --c) { s_start[c] = at; at += s_count[c]; }
    counts[0] = at;       // blocks that go through ICP
    counts[1] = s_heavy;  // the first so many of them are heavy
  }
  __syncthreads();
  for (uint32_t b = threadIdx.x; b < n; b += 1024u) {
    const uint32_t w = work[b];
    if (!w) continue;
    order[atomicAdd(&s_start[32 - __clz(w)], 1u)] = b;
  }
}

// ---- ICP helpers ----
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
// R = U diag(1, 1, det(U) det(V)) V^T for sigma = U S V^T.  With u2' = u0 x u1 and v2' = v0 x v1 (both right-handed) this
// is R = u0 v0^T + u1 v1^T + u2' v2'^T for either sign of the determinants, so only the two leading singular pairs are
// needed -- and those are well conditioned even for the flat (surface patch) blocks that are the rule here.
// V from Jacobi rotations on sigma^T sigma; float, like Eigen's JacobiSVD<Matrix3f> inside PCL.
__device__ __forceinline__ void jacobi_rotate(float& bpp, float& bqq, float& bpq, float& bpr, float& bqr, float& v0p, float& v0q, float& v1p,
                                              float& v1q, float& v2p, float& v2q) {
  if (fabsf(bpq) < 1e-37f) return;
  const float theta = (bqq - bpp) / (2.0f * bpq);
  const float t = copysignf(1.0f, theta) / (fabsf(theta) + sqrtf(theta * theta + 1.0f));
  const float c = 1.0f / sqrtf(t * t + 1.0f), sn = t * c;
  bpp -= t * bpq;
  bqq += t * bpq;
  bpq = 0.0f;
  const float r0 = bpr, r1 = bqr;
  bpr = c * r0 - sn * r1;
  bqr = sn * r0 + c * r1;
  float a0 = v0p, a1 = v0q;
  v0p = c * a0 - sn * a1; v0q = sn * a0 + c * a1;
  a0 = v1p; a1 = v1q;
  v1p = c * a0 - sn * a1; v1q = sn * a0 + c * a1;
  a0 = v2p; a1 = v2q;
  v2p = c * a0 - sn * a1; v2q = sn * a0 + c * a1;
}

__device__ void umeyama_rotation(const float A[9], float Rm[9]) {
  // B = A^T A (symmetric): b00 b11 b22 b01 b02 b12
  float b00 = A[0] * A[0] + A[3] * A[3] + A[6] * A[6], b11 = A[1] * A[1] + A[4] * A[4] + A[7] * A[7], b22 = A[2] * A[2] + A[5] * A[5] + A[8] * A[8];
  float b01 = A[0] * A[1] + A[3] * A[4] + A[6] * A[7], b02 = A[0] * A[2] + A[3] * A[5] + A[6] * A[8], b12 = A[1] * A[2] + A[4] * A[5] + A[7] * A[8];
  float v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;  // v[row][col]: columns = eigenvectors
  for (int sweep = 0; sweep < 8; ++sweep) {
    const float off = b01 * b01 + b02 * b02 + b12 * b12;
    if (off <= 1e-15f * (b00 * b00 + b11 * b11 + b22 * b22)) break;
    jacobi_rotate(b00, b11, b01, b02, b12, v00, v01, v10, v11, v20, v21);  // (p, q, r) = (0, 1, 2)
    jacobi_rotate(b00, b22, b02, b01, b12, v00, v02, v10, v12, v20, v22);  // (0, 2, 1)
    jacobi_rotate(b11, b22, b12, b01, b02, v01, v02, v11, v12, v21, v22);  // (1, 2, 0)
  }
  // the two largest eigenvalues' vectors
  float e[3] = {b00, b11, b22};
  float vc[3][3] = {{v00, v10, v20}, {v01, v11, v21}, {v02, v12, v22}};  // vc[k] = eigenvector k
  int i0 = 0, i1 = 1, i2 = 2;
  if (e[i1] > e[i0]) { const int t = i0; i0 = i1; i1 = t; }
  if (e[i2] > e[i0]) { const int t = i0; i0 = i2; i2 = t; }
  if (e[i2] > e[i1]) { const int t = i1; i1 = i2; i2 = t; }
  float va[3], vb[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {  // selects instead of dynamic indexing (registers, no scratch)
    va[k] = i0 == 0 ? vc[0][k] : (i0 == 1 ? vc[1][k] : vc[2][k]);
    vb[k] = i1 == 0 ? vc[0][k] : (i1 == 1 ? vc[1][k] : vc[2][k]);
  }
  auto normalise = [](float w[3]) {
    const float n2 = w[0] * w[0] + w[1] * w[1] + w[2] * w[2];
    if (!(n2 > 1e-37f)) return false;
    const float inv = 1.0f / sqrtf(n2);
    w[0] *= inv; w[1] *= inv; w[2] *= inv;
    return true;
  };
  auto any_orthogonal = [](const float u[3], float w[3]) {
    const float ax = fabsf(u[0]), ay = fabsf(u[1]), az = fabsf(u[2]);
    float ex = 0.f, ey = 0.f, ez = 0.f;
    if (ax <= ay && ax <= az) ex = 1.f; else if (ay <= az) ey = 1.f; else ez = 1.f;
    const float dot = ex * u[0] + ey * u[1] + ez * u[2];
    w[0] = ex - dot * u[0]; w[1] = ey - dot * u[1]; w[2] = ez - dot * u[2];
  };
  normalise(va);
  {  // vb orthogonal to va
    const float dot = va[0] * vb[0] + va[1] * vb[1] + va[2] * vb[2];
    vb[0] -= dot * va[0]; vb[1] -= dot * va[1]; vb[2] -= dot * va[2];
    if (!normalise(vb)) { any_orthogonal(va, vb); normalise(vb); }
  }
  float ua[3], ub[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    ua[r] = A[3 * r] * va[0] + A[3 * r + 1] * va[1] + A[3 * r + 2] * va[2];
    ub[r] = A[3 * r] * vb[0] + A[3 * r + 1] * vb[1] + A[3 * r + 2] * vb[2];
  }
  const float s0sq = ua[0] * ua[0] + ua[1] * ua[1] + ua[2] * ua[2];
  if (!normalise(ua)) { ua[0] = va[0]; ua[1] = va[1]; ua[2] = va[2]; }  // zero covariance: identity
  {
    const float dot = ua[0] * ub[0] + ua[1] * ub[1] + ua[2] * ub[2];
    ub[0] -= dot * ua[0]; ub[1] -= dot * ua[1]; ub[2] -= dot * ua[2];
    const float n2 = ub[0] * ub[0] + ub[1] * ub[1] + ub[2] * ub[2];
    if (n2 > 1e-12f * s0sq && normalise(ub)) {
    } else if (s0sq > 1e-37f) {  // rank 1: the rotation about u0 is free; take the one that keeps vb's image closest
      float w[3] = {vb[0] - (vb[0] * ua[0] + vb[1] * ua[1] + vb[2] * ua[2]) * ua[0], vb[1] - (vb[0] * ua[0] + vb[1] * ua[1] + vb[2] * ua[2]) * ua[1],
                    vb[2] - (vb[0] * ua[0] + vb[1] * ua[1] + vb[2] * ua[2]) * ua[2]};
      if (!normalise(w)) { any_orthogonal(ua, w); normalise(w); }
      ub[0] = w[0]; ub[1] = w[1]; ub[2] = w[2];
    } else {
      ub[0] = vb[0]; ub[1] = vb[1]; ub[2] = vb[2];
    }
  }
  const float uc[3] = {ua[1] * ub[2] - ua[2] * ub[1], ua[2] * ub[0] - ua[0] * ub[2], ua[0] * ub[1] - ua[1] * ub[0]};
  const float vcx[3] = {va[1] * vb[2] - va[2] * vb[1], va[2] * vb[0] - va[0] * vb[2], va[0] * vb[1] - va[1] * vb[0]};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int c = 0; c < 3; ++c) Rm[3 * r + c] = ua[r] * va[c] + ub[r] * vb[c] + uc[r] * vcx[c];
}

__device__ __forceinline__ float wave_sum_f(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

__device__ __forceinline__ float dist2(float x, float y, float z, const float4& q) {
  const float dx = x - q.x, dy = y - q.y, dz = z - q.z;
  float d = __fmul_rn(dx, dx);
  d = __fadd_rn(d, __fmul_rn(dy, dy));
  return __fadd_rn(d, __fmul_rn(dz, dz));
}

// nearest target with the targets staged in LDS (every lane reads the same address: a broadcast); slot = target index
__device__ __forceinline__ void nearest_lds(const float4* s_tgt, uint32_t nt, float x, float y, float z, uint32_t& best_slot, float& best) {
  best = FLT_MAX;
  best_slot = 0;
  uint32_t j = 0;
  for (; j + 4 <= nt; j += 4) {  // four targets in flight: the LDS reads and the arithmetic of one overlap the compares of another
    const float4 q0 = s_tgt[j], q1 = s_tgt[j + 1], q2 = s_tgt[j + 2], q3 = s_tgt[j + 3];
    const float d0 = dist2(x, y, z, q0), d1 = dist2(x, y, z, q1), d2 = dist2(x, y, z, q2), d3 = dist2(x, y, z, q3);
    // the lowest of the four, the earlier one on ties; then against the best so far (strictly smaller wins: lowest index)
    const bool a = d1 < d0, b = d3 < d2;
    const float m01 = a ? d1 : d0, m23 = b ? d3 : d2;
    const uint32_t i01 = a ? j + 1 : j, i23 = b ? j + 3 : j + 2;
    const bool c = m23 < m01;
    const float m = c ? m23 : m01;
    const uint32_t im = c ? i23 : i01;
    if (m < best) { best = m; best_slot = im; }
  }
  for (; j < nt; ++j) {
    const float d = dist2(x, y, z, s_tgt[j]);
    if (d < best) { best = d; best_slot = j; }
  }
}

__device__ __forceinline__ void wave_sync() {  // LDS written by some lanes of this wave, read by others
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}

// ICP of one macroblock by WAVES waves.  WAVES = 1: one wave per macroblock, four macroblocks per workgroup, no
// barrier inside the iteration (sums by cross-lane shuffles) -- the shape for frames with thousands of blocks, where the
// chip is full anyway.  WAVES = 4: the whole workgroup works on one macroblock (two barriers per iteration to add the
// waves' partial sums) -- for frames with few blocks, where the latency of the longest block is what counts.
// The 3x3 SVD is computed redundantly by every lane; all lanes take the same decisions from the same sums.
constexpr int kIcpWaves = kDBlock / 64;
constexpr uint32_t kIcpTargetCap = 512;  // per wave: targets staged in LDS (8 KB) and as many moving source points (8 KB);
                                         // larger blocks keep them in HBM/L2

template <int WAVES>
__global__ __launch_bounds__(kDBlock) void k_block_icp(DeltaArgs a) {
  __shared__ float4 s_tgt_all[kIcpWaves * kIcpTargetCap];
  __shared__ float4 s_cur_all[kIcpWaves * kIcpTargetCap];
  __shared__ float s_redf[2][kIcpWaves][9];
  __shared__ double s_redd[2][kIcpWaves];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // rank in the work order: the workgroup shape takes ranks [0, first_light), the wave shape [first_light, n_live)
  const uint32_t n_live = a.counts[0], first_light = a.shape == 1 ? n_live : (a.shape == 2 ? 0u : a.counts[1]);
  const uint32_t rank = WAVES == 1 ? first_light + blockIdx.x * kIcpWaves + (uint32_t)wave : blockIdx.x;
  if (rank >= (WAVES == 1 ? n_live : first_light)) return;  // WAVES == 1: waves are independent; WAVES == 4: the whole workgroup leaves
  const uint32_t b = a.order[rank];
  const BlockResult r0 = a.results[b];
  const uint32_t s0 = a.i_tree.leaf_start[r0.i_block], ns = r0.n_i;
  const uint32_t t0 = a.p_tree.leaf_start[b], nt = r0.n_p;
  const float4* src = a.i_xyzc + s0;  // source = the I frame's block, target = the predictive frame's block (impl.hpp:547-548)
  const float4* tgt = a.p_xyzc + t0;
  float4* s_tgt = WAVES == 1 ? s_tgt_all + (size_t)wave * kIcpTargetCap : s_tgt_all;
  const uint32_t first = WAVES == 1 ? (uint32_t)lane : threadIdx.x, step = 64u * WAVES;
  const bool staged = nt <= kIcpTargetCap * (WAVES == 1 ? 1u : (uint32_t)kIcpWaves);
  if (staged) {
    for (uint32_t j = first; j < nt; j += step) s_tgt[j] = tgt[j];
    if (WAVES == 1) wave_sync(); else __syncthreads();
  }
  auto nearest = [&](float x, float y, float z, uint32_t& slot, float& d) {
    if (staged) nearest_lds(s_tgt, nt, x, y, z, slot, d);
    else nearest_in(tgt, nt, x, y, z, slot, d);
  };
  int parity = 0;
  // sums over the block's points: v[0..count) floats and one double
  auto block_sums = [&](float* v, int count, double& dv) {
    for (int k = 0; k < count; ++k) v[k] = wave_sum_f(v[k]);
    dv = wave_sum_d(dv);
    if (WAVES > 1) {
      if (lane == 0) {
        for (int k = 0; k < count; ++k) s_redf[parity][wave][k] = v[k];
        s_redd[parity][wave] = dv;
      }
      __syncthreads();
      for (int k = 0; k < count; ++k) v[k] = (s_redf[parity][0][k] + s_redf[parity][1][k]) + (s_redf[parity][2][k] + s_redf[parity][3][k]);
      dv = (s_redd[parity][0] + s_redd[parity][1]) + (s_redd[parity][2] + s_redd[parity][3]);
      parity ^= 1;  // the other buffer next time: no second barrier needed
    }
  };
  // the moving copy of the source points, w = slot of the nearest target: in LDS when the block fits (an iteration
  // then has no HBM round trip at all), else in HBM.  Thread-private elements (i = first, first + step, ...): no
  // synchronisation needed.
  const uint32_t cap = kIcpTargetCap * (WAVES == 1 ? 1u : (uint32_t)kIcpWaves);
  float4* cur = ns <= cap ? (WAVES == 1 ? s_cur_all + (size_t)wave * kIcpTargetCap : s_cur_all) : a.cur + s0;
  for (uint32_t i = first; i < ns; i += step) cur[i] = src[i];
  float fin[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) fin[k] = (k % 5 == 0) ? 1.f : 0.f;
  const float fn = (float)ns;
  double prev_mse = DBL_MAX;
  int it = 0;
  const double rot_thr = 1.0 - (double)a.transformation_epsilon, trans_thr = (double)a.transformation_epsilon;
  const double mse_rel = 3.0 * (double)a.transformation_epsilon, mse_abs = 1e-12;
  while (true) {
    // correspondences: nearest target of every (moved) source point; sums for the means
    float m6[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double sd = 0;
    for (uint32_t i = first; i < ns; i += step) {
      const float4 p = cur[i];
      uint32_t j; float d;
      nearest(p.x, p.y, p.z, j, d);
      cur[i].w = __uint_as_float(j);
      const float4 q = staged ? s_tgt[j] : tgt[j];
      m6[0] += p.x; m6[1] += p.y; m6[2] += p.z;
      m6[3] += q.x; m6[4] += q.y; m6[5] += q.z;
      sd += (double)d;
    }
    block_sums(m6, 6, sd);
    const float smx = m6[0] / fn, smy = m6[1] / fn, smz = m6[2] / fn, tmx = m6[3] / fn, tmy = m6[4] / fn, tmz = m6[5] / fn;
    const double mse = sd / (double)ns;
    float sg[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = first; i < ns; i += step) {
      const float4 p = cur[i];
      const uint32_t j = __float_as_uint(p.w);
      const float4 q = staged ? s_tgt[j] : tgt[j];
      const float ds[3] = {p.x - smx, p.y - smy, p.z - smz}, dt[3] = {q.x - tmx, q.y - tmy, q.z - tmz};
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) sg[3 * r + c] += dt[r] * ds[c];
    }
    double unused = 0;
    block_sums(sg, 9, unused);
#pragma unroll
    for (int k = 0; k < 9; ++k) sg[k] /= fn;
    // every lane: the same transformation from the same sums
    float Rm[9], tr[16];
    umeyama_rotation(sg, Rm);
    {
      const float sm[3] = {smx, smy, smz}, tm[3] = {tmx, tmy, tmz};
#pragma unroll
      for (int r = 0; r < 3; ++r) {
#pragma unroll
        for (int c = 0; c < 3; ++c) tr[4 * r + c] = Rm[3 * r + c];
        tr[4 * r + 3] = tm[r] - (Rm[3 * r] * sm[0] + Rm[3 * r + 1] * sm[1] + Rm[3 * r + 2] * sm[2]);
      }
      tr[12] = tr[13] = tr[14] = 0.f; tr[15] = 1.f;
    }
    float nf[16];  // final = tr * final
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        nf[4 * r + c] = ((tr[4 * r] * fin[c] + tr[4 * r + 1] * fin[4 + c]) + tr[4 * r + 2] * fin[8 + c]) + tr[4 * r + 3] * fin[12 + c];
#pragma unroll
    for (int k = 0; k < 16; ++k) fin[k] = nf[k];
    for (uint32_t i = first; i < ns; i += step) {  // move the source points
      const float4 p = cur[i];
      float4 o = p;
      se3(tr, p.x, p.y, p.z, o.x, o.y, o.z);
      cur[i] = o;
    }
    ++it;
    // DefaultConvergenceCriteria::hasConverged (PCL 1.10) with the settings of impl.hpp:549-553
    bool stop = false;
    if (it >= a.max_iterations) stop = true;
    else {
      const double cos_angle = 0.5 * ((double)tr[0] + (double)tr[5] + (double)tr[10] - 1.0);
      const double tsq = (double)tr[3] * tr[3] + (double)tr[7] * tr[7] + (double)tr[11] * tr[11];
      if (cos_angle >= rot_thr && tsq <= trans_thr) stop = true;
      else if (fabs(mse - prev_mse) < mse_abs || fabs(mse - prev_mse) / prev_mse < mse_rel) stop = true;
    }
    prev_mse = mse;
    if (stop) break;  // uniform: every lane holds the same values
  }
  // getFitnessScore: mean squared distance of the source moved by the final transformation to its nearest target
  double fs = 0;
  for (uint32_t i = first; i < ns; i += step) {
    float x, y, z;
    se3(fin, src[i].x, src[i].y, src[i].z, x, y, z);
    uint32_t j; float d;
    nearest(x, y, z, j, d);
    fs += (double)d;
  }
  float none[1] = {0.f};
  block_sums(none, 0, fs);
  const double fitness = fs / (double)ns;
  if (first == 0) {
    BlockResult* out = a.results + b;
    out->iterations = it;
    out->fitness = (float)fitness;
    out->converged = fitness < a.point_resolution * 2.0 ? 1 : 0;  // hasConverged() is true also at the iteration limit
    for (int k = 0; k < 16; ++k) out->rt[k] = fin[k];
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

void launch_delta_blocks(const DeltaArgs& a, hipStream_t stream, hipStream_t aux, hipEvent_t ev_fork, hipEvent_t ev_join) {
  const uint32_t nbi = a.i_tree.n_blocks, nbp = a.p_tree.n_blocks;
  if (nbi) hipLaunchKernelGGL(k_block_keys, dim3((nbi + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.i_tree.leaf_code, a.i_tree.prefix_code, nbi, a.i_full);
  if (nbp) hipLaunchKernelGGL(k_block_keys, dim3((nbp + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.p_tree.leaf_code, a.p_tree.prefix_code, nbp, a.p_full);
  if (a.i_tree.n_points) hipLaunchKernelGGL(k_block_points, dim3((a.i_tree.n_points + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.i_tree, a.i_xyzc);
  if (a.p_tree.n_points) hipLaunchKernelGGL(k_block_points, dim3((a.p_tree.n_points + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a.p_tree, a.p_xyzc);
  if (!nbp) return;
  hipLaunchKernelGGL(k_block_match, dim3((2 * nbp + kDBlock - 1) / kDBlock), dim3(kDBlock), 0, stream, a);
  hipLaunchKernelGGL(k_block_order, dim3(1), dim3(1024), 0, stream, a.work, nbp, a.order, a.counts);
  // Few blocks: every one gets a workgroup (the longest block's latency is what counts).  Many blocks: the heavy ones
  // (first in the order) get a workgroup each, the rest a wave each; both launches cover the worst case and the
  // surplus workgroups leave at once.  a.shape = 1 / 2 on entry forces one shape for all blocks (test hook: option "icp_waves").
  DeltaArgs b = a;
  b.shape = a.shape ? a.shape : (nbp <= 2048 ? 1 : 0);
  // the two shapes work on disjoint blocks: side by side on two streams when the caller has a second one
  const bool fork = b.shape == 0 && aux && ev_fork && ev_join;
  if (fork) {
    (void)hipEventRecord(ev_fork, stream);
    (void)hipStreamWaitEvent(aux, ev_fork, 0);
  }
  if (b.shape != 2) hipLaunchKernelGGL(k_block_icp<kIcpWaves>, dim3(nbp), dim3(kDBlock), 0, stream, b);
  if (b.shape != 1) hipLaunchKernelGGL(k_block_icp<1>, dim3((nbp + kIcpWaves - 1) / kIcpWaves), dim3(kDBlock), 0, fork ? aux : stream, b);
  if (fork) {
    (void)hipEventRecord(ev_join, aux);
    (void)hipStreamWaitEvent(stream, ev_join, 0);
  }
}

void launch_delta_gather(const GatherArgs& a, hipStream_t stream) {
  if (a.n_blocks) hipLaunchKernelGGL(k_delta_gather, dim3(a.n_blocks), dim3(kDBlock), 0, stream, a);
}

}  // namespace pcc
