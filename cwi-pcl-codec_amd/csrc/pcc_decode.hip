// pcc_decode.hip -- the data-parallel half of decodePointCloud (impl.hpp:224-310) on the GPU (gfx950, wave64).
//
// What stays on the host is sequential by construction: the three range decoders (impl.hpp:1766-1835), the Huffman
// decoding of the colour JPEG, and the walk over the depth-first occupancy stream that tells a byte's level (the
// level of byte i+1 depends on the remaining-children counters of all ancestors of byte i).  The walk visits branch
// nodes only and hands over one record per node of level D-1; from there on every voxel is independent:
//   k_dec_idct    dequantisation + jidctint.c inverse DCT of every 8x8 block (JPEGReader::readJPEG, jpeg_io.hpp:90-192)
//   k_dec_points  voxel key -> centre or centroid (deserializeTreeCallback, impl.hpp:1584-1653), colour = the voxel's
//                 pixel of the snake-mapped image (decodeJPEGSnake, jpegcc.h:228-242) with libjpeg's fancy h2v2
//                 chroma upsampling and YCbCr -> RGB, or the voxel's bytes of the colour vector
// Integer arithmetic as in the host decoder (pcc_host_codec.cpp), doubles individually rounded (-ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pcc_decode.h"

namespace pcc {

namespace {

typedef long long idct_t;  // like the host decoder: nothing overflows on corrupt coefficients either
__device__ __forceinline__ idct_t descale64(idct_t x, int n) { return (x + ((idct_t)1 << (n - 1))) >> n; }
__device__ __forceinline__ uint8_t idct_clamp(idct_t v) {  // sample_range_limit + CENTERJSAMPLE, index masked to 10 bits
  const int i = (int)(v & 1023);
  if (i < 128) return (uint8_t)(128 + i);
  if (i < 512) return 255;
  if (i < 896) return 0;
  return (uint8_t)(i - 896);
}
__device__ __forceinline__ void idct_1d(const idct_t in[8], idct_t o[8]) {  // jidctint.c butterfly, unscaled outputs
  idct_t z2 = in[2], z3 = in[6];
  idct_t z1 = (z2 + z3) * 4433;
  const idct_t e2 = z1 + z3 * (-15137), e3 = z1 + z2 * 6270;
  const idct_t e0 = (in[0] + in[4]) * 8192, e1 = (in[0] - in[4]) * 8192;
  const idct_t t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
  idct_t t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  idct_t z4 = t1 + t3;
  const idct_t z5 = (z3 + z4) * 9633;
  t0 *= 2446; t1 *= 16819; t2 *= 25172; t3 *= 12299;
  z1 *= -7373; z2 *= -20995;
  z3 = z3 * (-16069) + z5;
  z4 = z4 * (-3196) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  o[0] = t10 + t3; o[7] = t10 - t3; o[1] = t11 + t2; o[6] = t11 - t2;
  o[2] = t12 + t1; o[5] = t12 - t1; o[3] = t13 + t0; o[4] = t13 - t0;
}

constexpr int kIdctBlocksPerWg = 32;  // 8 threads per block

// thread = (block, column c) for the column pass, then (block, row r) for the row pass; the workspace goes through LDS
__global__ __launch_bounds__(kIdctBlocksPerWg * 8) void k_dec_idct(IdctArgs a) {
  __shared__ idct_t s_ws[kIdctBlocksPerWg][64 + 8];
  const uint32_t n_blocks = a.mcus_x * a.mcus_y * 6u;
  const uint32_t lb = threadIdx.x >> 3, k = threadIdx.x & 7u;
  const uint32_t b = blockIdx.x * kIdctBlocksPerWg + lb;
  const bool live = b < n_blocks;
  const uint32_t mcu = b / 6u, slot = b % 6u;
  const int comp = slot < 4u ? 0 : (slot == 4u ? 1 : 2);
  if (live) {
    const int16_t* coef = a.blocks + (size_t)b * 64;
    idct_t in[8], o[8];
#pragma unroll
    for (int r = 0; r < 8; ++r) in[r] = (idct_t)coef[8 * r + k] * (idct_t)a.q[comp][8 * r + k];
    idct_1d(in, o);
#pragma unroll
    for (int r = 0; r < 8; ++r) s_ws[lb][8 * r + k] = descale64(o[r], 13 - 2);
  }
  __syncthreads();
  if (!live) return;
  idct_t in[8], o[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) in[c] = s_ws[lb][8 * k + c];
  idct_1d(in, o);
  const uint32_t mx = mcu % a.mcus_x, my = mcu / a.mcus_x;
  uint8_t* dst;
  if (slot < 4u) {
    const uint32_t stride = a.mcus_x * 16u;
    dst = a.plane_y + (size_t)(16u * my + 8u * (slot >> 1) + k) * stride + 16u * mx + 8u * (slot & 1u);
  } else {
    const uint32_t stride = a.mcus_x * 8u;
    dst = (slot == 4u ? a.plane_cb : a.plane_cr) + (size_t)(8u * my + k) * stride + 8u * mx;
  }
  uint32_t w0 = 0, w1 = 0;  // the eight samples of the row in two aligned dwords
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    w0 |= (uint32_t)idct_clamp(descale64(o[c], 13 + 2 + 3)) << (8 * c);
    w1 |= (uint32_t)idct_clamp(descale64(o[4 + c], 13 + 2 + 3)) << (8 * c);
  }
  reinterpret_cast<uint32_t*>(dst)[0] = w0;
  reinterpret_cast<uint32_t*>(dst)[1] = w1;
}

// SnakeGridIterator (snake.h:46-71) in closed form: linear element i -> pixel index, W multiple of 8
__device__ __forceinline__ uint32_t snake_pos(uint32_t i, uint32_t W, uint32_t H) {
  const uint32_t full = H / 8u, hl = H % 8u, per_row = W * 8u;
  if (i < full * per_row) {
    const uint32_t br = i / per_row, rem = i % per_row;
    const uint32_t bw = rem / 64u, q = rem % 64u, r = q / 8u, c = q % 8u;
    return (br * 8u + r) * W + bw * 8u + ((r & 1u) ? 7u - c : c);
  }
  const uint32_t rem = i - full * per_row, blk = 8u * hl;
  const uint32_t bw = rem / blk, q = rem % blk, r = q / 8u, c = q % 8u;
  const uint32_t flip = (r + ((hl & 1u) ? bw : 0u)) & 1u;
  return (full * 8u + r) * W + bw * 8u + (flip ? 7u - c : c);
}

__device__ __forceinline__ uint32_t compact3(uint64_t x) {
  x &= 0x1249249249249249ULL;
  x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ULL;
  x = (x ^ (x >> 4)) & 0x100f00f00f00f00fULL;
  x = (x ^ (x >> 8)) & 0x1f0000ff0000ffULL;
  x = (x ^ (x >> 16)) & 0x1f00000000ffffULL;
  x = (x ^ (x >> 32)) & 0x1fffffULL;
  return (uint32_t)x;
}

// jdsample.c h2v2_fancy_upsample at output pixel (r, c): 3/4 of the nearer chroma sample, 1/4 of the farther one, in
// both directions; plain replication when the chroma plane is at most two samples wide
__device__ __forceinline__ int upsampled(const uint8_t* plane, uint32_t stride, uint32_t cw, uint32_t ch, uint32_t r, uint32_t c) {
  const uint32_t cr = r >> 1, cc = c >> 1;
  const uint8_t* in0 = plane + (size_t)cr * stride;
  if (cw <= 2u) return in0[cc];
  const uint32_t nr = (r & 1u) ? min(cr + 1u, ch - 1u) : (cr ? cr - 1u : 0u);
  const uint8_t* in1 = plane + (size_t)nr * stride;
  const int cur = 3 * in0[cc] + in1[cc];
  if (c == 0u) return (cur * 4 + 8) >> 4;
  if (c == 2u * cw - 1u) return (cur * 4 + 7) >> 4;
  if (c & 1u) return (cur * 3 + (3 * in0[cc + 1u] + in1[cc + 1u]) + 7) >> 4;
  return (cur * 3 + (3 * in0[cc - 1u] + in1[cc - 1u]) + 8) >> 4;
}
__device__ __forceinline__ uint32_t sat8(int v) { return (uint32_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

// one thread per node of level D-1: its (up to eight) voxels are consecutive in the output
__global__ __launch_bounds__(256) void k_dec_points(DecodeArgs a) {
  const uint32_t t = blockIdx.x * 256u + threadIdx.x;
  if (t >= a.n_parents) return;
  const uint64_t prefix = a.prefix[t];
  const uint32_t prefix_hi = a.prefix_hi ? a.prefix_hi[t] : 0u;
  // the node's key per axis (x-major triples); a voxel's key is that, shifted up, plus its child bit
  const uint32_t pk[3] = {compact3(prefix >> 2) | (compact3((uint64_t)prefix_hi >> 2) << 21), compact3(prefix >> 1) | (compact3((uint64_t)prefix_hi >> 1) << 21),
                          compact3(prefix) | (compact3((uint64_t)prefix_hi) << 21)};
  uint32_t bits = a.bits[t];
  uint32_t leaf = a.first[t];
  const uint32_t cw = (a.img_w + 1u) / 2u, chh = (a.img_h + 1u) / 2u;
  uint4* out = reinterpret_cast<uint4*>(a.points);
  while (bits) {
    const uint32_t c = (uint32_t)__ffs((int)bits) - 1u;
    bits &= bits - 1u;
    if (leaf >= a.n_leaves) return;  // (the host has checked the counts)
    const uint32_t key[3] = {(pk[0] << 1) | ((c >> 2) & 1u), (pk[1] << 1) | ((c >> 1) & 1u), (pk[2] << 1) | (c & 1u)};
    float xyz[3];
#pragma unroll
    for (int ax = 0; ax < 3; ++ax) {
      if (a.centroid) {  // ptv2.h:103-118: an unsigned byte times the float precision 0.001f, added to the lower voxel corner
        const double lc = __dadd_rn(__dmul_rn((double)key[ax], a.res), a.mn[ax]);
        xyz[ax] = (float)__dadd_rn(lc, (double)__fmul_rn((float)a.centroid[3 * (size_t)leaf + ax], 0.001f));
      } else {           // impl.hpp:1630-1632
        xyz[ax] = (float)__dadd_rn(__dmul_rn(__dadd_rn((double)key[ax], 0.5), a.res), a.mn[ax]);
      }
    }
    uint32_t rgba = 0x00FFFFFFu;  // ColorCoding::setDefaultColor
    if (a.with_colour) {
      uint32_t c0, c1, c2;
      if (a.plane_y) {
        const uint32_t px = snake_pos(leaf, a.img_w, a.img_h), r = px / a.img_w, col = px % a.img_w;
        const int y = a.plane_y[(size_t)r * a.y_stride + col];
        const int xb = upsampled(a.plane_cb, a.c_stride, cw, chh, r, col) - 128;
        const int xr = upsampled(a.plane_cr, a.c_stride, cw, chh, r, col) - 128;
        c0 = sat8(y + ((91881 * xr + 32768) >> 16));
        c1 = sat8(y + ((-22554 * xb + 32768 - 46802 * xr) >> 16));
        c2 = sat8(y + ((116130 * xb + 32768) >> 16));
      } else {
        c0 = (uint8_t)(a.colours[3 * (size_t)leaf] << a.colour_shift);
        c1 = (uint8_t)(a.colours[3 * (size_t)leaf + 1] << a.colour_shift);
        c2 = (uint8_t)(a.colours[3 * (size_t)leaf + 2] << a.colour_shift);
      }
      rgba = c0 | (c1 << 8) | (c2 << 16);
    }
    out[2 * (size_t)leaf] = make_uint4(__float_as_uint(xyz[0]), __float_as_uint(xyz[1]), __float_as_uint(xyz[2]), __float_as_uint(1.0f));
    out[2 * (size_t)leaf + 1] = make_uint4(rgba, 0u, 0u, 0u);
    ++leaf;
  }
}

}  // namespace

void launch_decode_idct(const IdctArgs& a, hipStream_t stream) {
  const uint32_t n_blocks = a.mcus_x * a.mcus_y * 6u;
  if (!n_blocks) return;
  hipLaunchKernelGGL(k_dec_idct, dim3((n_blocks + kIdctBlocksPerWg - 1) / kIdctBlocksPerWg), dim3(kIdctBlocksPerWg * 8), 0, stream, a);
}

void launch_decode_points(const DecodeArgs& a, hipStream_t stream) {
  if (!a.n_parents) return;
  hipLaunchKernelGGL(k_dec_points, dim3((a.n_parents + 255u) / 256u), dim3(256), 0, stream, a);
}

}  // namespace pcc
