// pcc_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4, wave64) for the intra-frame
// hot path of the CWI point-cloud codec.  They replace, with breadth-first array passes,
// what the reference does with a pointer octree:
//
//   P1/P2  addPointsFromInputCloud + adoptBoundingBoxToPoint   -> k_boxes_events (chunk boxes + growth replay in one launch)
//   P3     genOctreeKeyforPoint                                -> k_make_keys
//   P4     createLeafRecursive + addPointIndex                 -> LSD radix sort (k_make_keys histograms, k_digit_totals, k_sort_pass)
//   P5     serializeTree (depth-first occupancy bytes)         -> k_leaf_scan + k_leaf_tile
//   C2/P6  serializeTreeCallback / encodeAverageOfPoints       -> k_leaf_tile
//   C3b    SnakeGridMapping::doMapping                         -> k_leaf_tile (closed-form position)
//   C4     PointCodingV2::encodePoint                          -> k_leaf_tile
//   C3/C5  encodeJPEGSnake / encodeJPEGLines, JPEGWriter       -> k_leaf_tile (image rows), k_jpeg_rows, k_jpeg_lines
//
// These are the forms that have run on an MI355X (round 2) plus layout-only changes; forms that have not are not compiled
// into this file (branch experiments/r04-optin-forms; DESIGN.md, status).
//
// All of it is integer / byte / fp64-scalar work bound by HBM bandwidth and launch latency:
// no MFMA anywhere (there is no dense contraction in this path).
//
// Floating point discipline: keys and voxel centres are defined by individually rounded
// double operations (no FMA contraction, no fast-math): explicit __d*_rn intrinsics are
// used and the file is compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <float.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>

#include "pcc_device.h"
#include "pcc_kernels.h"
#include "pcc_dev.h"

namespace pcc {

// ------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ int wave_id() { return threadIdx.x >> 6; }

__device__ __forceinline__ void load_xyz(const PointView& pv, uint32_t i, float& x, float& y, float& z) {
  const uint8_t* p = pv.base + (size_t)i * pv.stride;
  if (pv.aligned16) {  // one 16-byte load per lane (x,y,z,w)
    const float4 v = *reinterpret_cast<const float4*>(p);
    x = v.x; y = v.y; z = v.z;
  } else {
    const float* f = reinterpret_cast<const float*>(p);
    x = f[0]; y = f[1]; z = f[2];
  }
}
__device__ __forceinline__ uint32_t load_rgba(const PointView& pv, uint32_t i) {
  return *reinterpret_cast<const uint32_t*>(pv.base + (size_t)i * pv.stride + pv.rgb_off);
}
__device__ __forceinline__ bool finite3(float x, float y, float z) {
  return isfinite(x) && isfinite(y) && isfinite(z);
}

// 21-bit -> every third bit
__device__ __forceinline__ uint64_t split3(uint32_t v) {
  uint64_t x = v & 0x1fffffu;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
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
// x-major triples: child index = (xbit<<2)|(ybit<<1)|zbit  (OctreeKey::getChildIdxWithDepthMask)
__device__ __forceinline__ uint64_t morton3(uint32_t kx, uint32_t ky, uint32_t kz) {
  return (split3(kx) << 2) | (split3(ky) << 1) | split3(kz);
}

// ---- wave-wide reductions and scans through DPP: lanes read each other's registers inside the VALU (a __shfl is a
//      ds_bpermute, an LDS round trip of a hundred cycles or more per step, six steps per reduction; these run in the
//      latency-bound stretches of every kernel: digit scans of the sort pass, the replay workgroup of k_boxes_events).
//      All 64 lanes must be active.
#ifndef PCC_WAVE_OPS_SHFL
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t identity, uint32_t v) {  // lanes without a source keep `identity`
  return (uint32_t)__builtin_amdgcn_update_dpp((int)identity, (int)v, CTRL, ROW_MASK, 0xf, false);
}
// inclusive scan over the 64 lanes: row_shr 1, 2, 4, 8 inside the rows of 16 lanes, then lane 15 of rows 0 and 2 into
// rows 1 and 3 (row_bcast:15), then lane 31 into the upper half (row_bcast:31)
template <typename Op>
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v, uint32_t id, Op op) {
  v = op(v, dpp_mov<0x111, 0xf>(id, v));
  v = op(v, dpp_mov<0x112, 0xf>(id, v));
  v = op(v, dpp_mov<0x114, 0xf>(id, v));
  v = op(v, dpp_mov<0x118, 0xf>(id, v));
  v = op(v, dpp_mov<0x142, 0xa>(id, v));
  v = op(v, dpp_mov<0x143, 0xc>(id, v));
  return v;
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint64_t dpp_add_u64(uint64_t v) {
  const uint32_t lo = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)v), hi = dpp_mov<CTRL, ROW_MASK>(0u, (uint32_t)(v >> 32));
  return v + (((uint64_t)hi << 32) | lo);
}
// lane `l` of v for everybody, l uniform: v_readlane (a __shfl with a uniform index is still a ds_bpermute)
__device__ __forceinline__ uint32_t lane_of(uint32_t v, int l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ float lane_of(float v, int l) { return __uint_as_float(lane_of(__float_as_uint(v), l)); }
__device__ __forceinline__ uint64_t lane_of(uint64_t v, int l) { return ((uint64_t)lane_of((uint32_t)(v >> 32), l) << 32) | lane_of((uint32_t)v, l); }
// the value of the lane below (wave_shr:1); lane 0 keeps its own `first`
__device__ __forceinline__ uint64_t wave_shr1(uint64_t v, uint64_t first) {
  const uint32_t lo = dpp_mov<0x138, 0xf>((uint32_t)first, (uint32_t)v), hi = dpp_mov<0x138, 0xf>((uint32_t)(first >> 32), (uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}
__device__ __forceinline__ uint32_t lane63(uint32_t v) { return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint64_t lane63(uint64_t v) { return ((uint64_t)lane63((uint32_t)(v >> 32)) << 32) | lane63((uint32_t)v); }

#else
// -DPCC_WAVE_OPS_SHFL: the same helpers through __shfl (ds_bpermute), the form they had before the DPP rewrite -- a build
// switch for bisecting (make shfl -> libpcc_hip_shfl.so; the parity tests run against either library through PCC_LIB)
template <typename Op>
__device__ __forceinline__ uint32_t wave_scan_u32(uint32_t v, uint32_t id, Op op) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl((int)v, lane - o);
    v = op(v, lane >= o ? t : id);
  }
  return v;
}
__device__ __forceinline__ uint32_t lane_of(uint32_t v, int l) { return (uint32_t)__shfl((int)v, l); }
__device__ __forceinline__ float lane_of(float v, int l) { return __shfl(v, l); }
__device__ __forceinline__ uint64_t lane_of(uint64_t v, int l) { return ((uint64_t)lane_of((uint32_t)(v >> 32), l) << 32) | lane_of((uint32_t)v, l); }
__device__ __forceinline__ uint64_t wave_shr1(uint64_t v, uint64_t first) {
  const int lane = lane_id();
  const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, lane - 1), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), lane - 1);
  return lane ? (((uint64_t)hi << 32) | lo) : first;
}
__device__ __forceinline__ uint32_t lane63(uint32_t v) { return lane_of(v, 63); }
__device__ __forceinline__ uint64_t lane63(uint64_t v) { return lane_of(v, 63); }
#endif

__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
  return wave_scan_u32(v, 0u, [](uint32_t a, uint32_t b) { return a + b; });
}
__device__ __forceinline__ uint64_t wave_incl_scan_u64(uint64_t v) {
#ifndef PCC_WAVE_OPS_SHFL
  v = dpp_add_u64<0x111, 0xf>(v);
  v = dpp_add_u64<0x112, 0xf>(v);
  v = dpp_add_u64<0x114, 0xf>(v);
  v = dpp_add_u64<0x118, 0xf>(v);
  v = dpp_add_u64<0x142, 0xa>(v);
  v = dpp_add_u64<0x143, 0xc>(v);
#else
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t lo = (uint32_t)__shfl((int)(uint32_t)v, lane - o), hi = (uint32_t)__shfl((int)(uint32_t)(v >> 32), lane - o);
    if (lane >= o) v += ((uint64_t)hi << 32) | lo;
  }
#endif
  return v;
}
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) { return wave_incl_scan_u32(v); }
__device__ __forceinline__ uint64_t wave_incl_scan(uint64_t v) { return wave_incl_scan_u64(v); }
__device__ __forceinline__ float wave_min_f(float v) {
  const uint32_t r = wave_scan_u32(__float_as_uint(v), __float_as_uint(FLT_MAX),
                                   [](uint32_t a, uint32_t b) { return __float_as_uint(fminf(__uint_as_float(a), __uint_as_float(b))); });
  return __uint_as_float(lane63(r));
}
__device__ __forceinline__ float wave_max_f(float v) {
  const uint32_t r = wave_scan_u32(__float_as_uint(v), __float_as_uint(-FLT_MAX),
                                   [](uint32_t a, uint32_t b) { return __float_as_uint(fmaxf(__uint_as_float(a), __uint_as_float(b))); });
  return __uint_as_float(lane63(r));
}
__device__ __forceinline__ int wave_min_i(int v) {
  return (int)lane63(wave_scan_u32((uint32_t)v, 0x7fffffffu, [](uint32_t a, uint32_t b) { return (uint32_t)min((int)a, (int)b); }));
}
__device__ __forceinline__ uint64_t wave_sum_u64(uint64_t v) { return lane63(wave_incl_scan_u64(v)); }
// block-wide exclusive scan of one value per thread (NW wave64 per workgroup); `total` = block sum
template <int NW, typename T>
__device__ __forceinline__ T block_excl_scan(T v, T* s_wave /*[NW]*/, T& total) {
  const T incl = wave_incl_scan(v);
  const int lane = lane_id();
  if (lane == 63) s_wave[wave_id()] = incl;
  __syncthreads();
  T off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const T x = s_wave[w];
    if (w < wave_id()) off += x;
    tot += x;
  }
  total = tot;
  __syncthreads();
  return off + incl - v;
}
__device__ __forceinline__ uint64_t block_excl_scan_u64(uint64_t v, uint64_t* s_wave /*[4]*/, uint64_t& total) {
  return block_excl_scan<kBlock / 64, uint64_t>(v, s_wave, total);
}
__device__ __forceinline__ uint32_t block_excl_scan_u32(uint32_t v, uint32_t* s_wave /*[4]*/, uint32_t& total) {
  return block_excl_scan<kBlock / 64, uint32_t>(v, s_wave, total);
}

// words other workgroups of the SAME launch poll: always agent-scope relaxed atomics (one self-describing
// word per hand-off, so no fence is needed; cdna_hip_programming.md guideline 16, form R2)
__device__ __forceinline__ void publish_u32(uint32_t* p, uint32_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t poll_u32(const uint32_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void publish_u64(uint64_t* p, uint64_t v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t poll_u64(const uint64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The frame's sticky error word.  A workgroup that gives up stores its code; kernels of LATER launches read the word like any
// other field of the state, but k_leaf_scan's tiles read it in the launch in which other tiles may store into it -- so inside
// kernels it is written, and there read, like every word that crosses workgroups inside a launch.  (Several workgroups may
// raise: any of their codes sends the frame back or fails it.)
__device__ __forceinline__ void raise_error(FrameState* st, int code) { publish_u32(reinterpret_cast<uint32_t*>(&st->error), (uint32_t)code); }
__device__ __forceinline__ int poll_error(const FrameState* st) { return (int)poll_u32(reinterpret_cast<const uint32_t*>(&st->error)); }

// A place where the code relies on all 64 lanes of the wave having executed everything above it before any lane goes
// on (the LDS operations of one wave are issued in program order): nothing on the GPU; a meeting point of the lanes in
// the CPU executor of tests/emu, which runs the lanes of a wave one after the other between such points.
#ifdef PCC_EMU
#define PCC_WAVE_LOCKSTEP() emu::wave_barrier()
#else
#define PCC_WAVE_LOCKSTEP() do { } while (0)
#endif
// The same between an LDS write of one lane and an LDS read of ANOTHER address by another lane of the wave: the hardware
// keeps a wave's LDS operations in order, but to the compiler the two accesses of one thread are unrelated and may be
// swapped -- a wavefront-scope fence and a wave barrier (no instruction beyond a wait for the LDS counter) forbid that.
#ifdef PCC_EMU
#define PCC_WAVE_SYNC_LDS() emu::wave_barrier()
#else
#define PCC_WAVE_SYNC_LDS() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)
#endif

#ifdef PCC_KTIME  // developer build only: phase time stamps of k_sort_pass (shader clock), read by tools/ktime.py
__device__ unsigned long long g_ktime[(7 + 2) * 1024 * 8];
#define PCC_KTR(row, slot)                                                                             \
  do {                                                                                                 \
    const unsigned ks_ = (gridDim.x + 1023u) / 1024u; /* every ks_-th workgroup of a large grid is sampled */ \
    if (threadIdx.x == 0 && blockIdx.x % ks_ == 0) g_ktime[((size_t)(row) * 1024 + blockIdx.x / ks_) * 8 + (slot)] = wall_clock64(); \
  } while (0)
#else
#define PCC_KTR(row, slot) do { } while (0)
#endif
#define PCC_KT(slot) PCC_KTR(pass, slot)

// Optional device-side span of a launch: [earliest workgroup start, latest wave end] on the GPU's real-time clock.
// HIP events between launches add a few microseconds to each short kernel; this is what a kernel trace reports.
// A launch owns 2 x kSpanShards words, preset to ~0 and only ever lowered (ends are stored inverted); workgroups and
// waves spread over the shards, so the recording itself does not queue thousands of atomics on one word (that
// tripled the duration of the short kernels when tried).  The host takes the minimum of each half.  Null: nothing.
struct KSpan {
  unsigned long long* p;
  __device__ explicit KSpan(unsigned long long* q) : p(q) {
    if (p && threadIdx.x == 0) atomicMin(p + (blockIdx.x % kSpanShards), wall_clock64());
  }
  __device__ ~KSpan() {
    if (p && (threadIdx.x & 63u) == 0u)
      atomicMin(p + kSpanShards + ((blockIdx.x * 16u + (threadIdx.x >> 6)) % kSpanShards), ~wall_clock64());
  }
};

// ------------------------------------------------------------------------------------------
// Stage 0 + 1 in ONE launch: per-chunk bounding boxes (first read of the cloud: 16 of every 32 bytes per point)
// by workgroups 1..n_chunks, and the adaptive bounding box (P2) -- sequential and order dependent by
// definition -- by workgroup 0, which
//   * replays the growth events of chunk 0 straight away (with points in random order the box reaches its
//     final size within the first few points), while the other workgroups stream the cloud;
//   * then collects the chunk boxes as they appear, skips every chunk whose box fits the current bounding box
//     (all of them, usually), and writes the epoch table and the sort plan.
// Workgroup 0 waits for workgroups that never wait themselves, so the launch makes progress in whatever
// order the workgroups are dispatched.  Hand-off: a chunk box is eight self-describing 8-byte words
// {value, frame sequence number}, each written by one agent-scope store and read by one agent-scope load
// (cdna_hip_programming.md guideline 16, form R2: no fence, no flag, the writer does not wait for anything).
// ------------------------------------------------------------------------------------------
constexpr int kBoxWords = 8;  // mn xyz, mx xyz, first finite index, finite count

// ---- octree key of a point (P3, genOctreeKeyforPoint) -> the code that is sorted
constexpr uint64_t kInvalidKey = ~0ull;
struct KeyGeom {
  int vb, cm, np, ibits, payload, payload2;
  bool packed_mode, colour_in_key, ranked, deep;
  unsigned cbase[3], cdim[3], lm, m;
  uint32_t prefix[3];
  int pshift[kMaxPasses];
  uint32_t pmask[kMaxPasses];
};
// `mn`, `shift`: origin and key offset of the epoch the point belongs to; `cell_rank`: FrameState::cell_rank or a copy
__device__ __forceinline__ uint64_t point_code(const KeyGeom& g, const double* mn, const uint32_t* shift, const uint8_t* cell_rank, double res,
                                               double inv_res_pow2, float x, float y, float z, bool& ok, uint32_t& hi) {
  hi = 0u;
  const float p[3] = {x, y, z};
  unsigned kk[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // x / res; when res is a power of two the product with its (exact) reciprocal is the same double
    const double diff = __dsub_rn((double)p[a], mn[a]);
    const double d = inv_res_pow2 != 0.0 ? __dmul_rn(diff, inv_res_pow2) : __ddiv_rn(diff, res);
    kk[a] = (unsigned)d + shift[a];
    ok &= g.vb >= 32 || ((kk[a] >> g.vb) == (g.prefix[a] >> g.vb));
  }
  if (g.deep) {  // two-word code: `hi` = the triples above the 21 low ones (pcc_device.h: kMaxDepthDeep)
    hi = (uint32_t)morton3((kk[0] & g.m) >> 21, (kk[1] & g.m) >> 21, (kk[2] & g.m) >> 21);
    return morton3(kk[0] & g.m & 0x1fffffu, kk[1] & g.m & 0x1fffffu, kk[2] & g.m & 0x1fffffu);
  }
  if (!g.ranked) return morton3(kk[0] & g.m, kk[1] & g.m, kk[2] & g.m);
  unsigned d[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) { d[a] = (kk[a] >> g.cm) - g.cbase[a]; ok &= d[a] < g.cdim[a]; }
  const unsigned cell = ok ? d[2] + g.cdim[2] * (d[1] + g.cdim[1] * d[0]) : 0u;
  return ((uint64_t)cell_rank[cell] << (3 * g.cm)) | morton3(kk[0] & g.lm, kk[1] & g.lm, kk[2] & g.lm);
}

// digit of pass q: of the low word, or (deep trees, shift >= 63) of the high word
__device__ __forceinline__ uint32_t code_digit(const KeyGeom& g, int q, uint64_t lo, uint32_t hi) {
  return (g.pshift[q] >= 63 ? (hi >> (g.pshift[q] - 63)) : (uint32_t)(lo >> g.pshift[q])) & g.pmask[q];
}

__device__ __forceinline__ void publish_box(uint64_t* dst, const ChunkBox& b, uint32_t seq) {
  uint32_t v[kBoxWords];
  __builtin_memcpy(v, &b, sizeof(b));
#pragma unroll
  for (int k = 0; k < kBoxWords; ++k) publish_u64(dst + k, ((uint64_t)seq << 32) | v[k]);
}
__device__ __forceinline__ bool decode_box(const uint64_t* w, uint32_t seq, ChunkBox& b) {
  uint32_t v[kBoxWords];
  bool all = true;
#pragma unroll
  for (int k = 0; k < kBoxWords; ++k) {
    all &= (uint32_t)(w[k] >> 32) == seq;
    v[k] = (uint32_t)w[k];
  }
  __builtin_memcpy(&b, v, sizeof(b));
  return all;
}
__device__ __forceinline__ bool fetch_box(const uint64_t* src, uint32_t seq, ChunkBox& b) {
  uint64_t w[kBoxWords];
#pragma unroll
  for (int k = 0; k < kBoxWords; ++k) w[k] = poll_u64(src + k);
  return decode_box(w, seq, b);
}

__device__ __forceinline__ void chunk_box_block(const PointView& pv, uint32_t n, uint32_t c, uint32_t n_chunks, uint64_t* __restrict__ boxes,
                                                uint32_t seq, uint4* __restrict__ sync_area, uint32_t sync_vec16, float* s_f, int* s_i) {
  float(*s_mn)[kBlock / 64] = reinterpret_cast<float(*)[kBlock / 64]>(s_f);
  float(*s_mx)[kBlock / 64] = reinterpret_cast<float(*)[kBlock / 64]>(s_f + 3 * (kBlock / 64));
  int* s_first = s_i;
  int* s_cnt = s_i + kBlock / 64;
  // every word another workgroup polls later in this frame (tile tickets, look-back status) starts at zero
  for (uint32_t k = c * kBlock + threadIdx.x; k < sync_vec16; k += n_chunks * kBlock)
    sync_area[k] = make_uint4(0, 0, 0, 0);
  const uint32_t base = c * kTile;
  float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  int first = 0x7fffffff, cnt = 0;
#pragma unroll
  for (int k = 0; k < kItems; ++k) {
    const uint32_t i = base + k * kBlock + threadIdx.x;
    if (i < n) {
      float x, y, z;
      load_xyz(pv, i, x, y, z);
      if (finite3(x, y, z)) {
        mn[0] = fminf(mn[0], x); mx[0] = fmaxf(mx[0], x);
        mn[1] = fminf(mn[1], y); mx[1] = fmaxf(mx[1], y);
        mn[2] = fminf(mn[2], z); mx[2] = fmaxf(mx[2], z);
        first = min(first, (int)i);
        ++cnt;
      }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; ++a) { mn[a] = wave_min_f(mn[a]); mx[a] = wave_max_f(mx[a]); }
  first = wave_min_i(first);
  cnt = (int)wave_sum_u64((uint64_t)cnt);
  if (lane_id() == 0) {
    const int w = wave_id();
    for (int a = 0; a < 3; ++a) { s_mn[a][w] = mn[a]; s_mx[a][w] = mx[a]; }
    s_first[w] = first; s_cnt[w] = cnt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    ChunkBox b;
    int cc = 0, f = 0x7fffffff;
    for (int a = 0; a < 3; ++a) { b.mn[a] = FLT_MAX; b.mx[a] = -FLT_MAX; }
    for (int w = 0; w < kBlock / 64; ++w) {
      for (int a = 0; a < 3; ++a) { b.mn[a] = fminf(b.mn[a], s_mn[a][w]); b.mx[a] = fmaxf(b.mx[a], s_mx[a][w]); }
      f = min(f, s_first[w]); cc += s_cnt[w];
    }
    b.first_finite = cc ? f : -1;
    b.n_finite = cc;
    publish_box(boxes + (size_t)c * kBoxWords, b, seq);
  }
  PCC_KTR(6, 7);
}

// what the host tells the plan (workgroup 0): which sort it enqueued and what the stages behind the sort will read
struct PlanOptions {
  int force_pairs;      // testing: (key, index) pairs even where the packed key would fit
  int need_index;       // somebody reads the point index of the sorted elements (centroids, macroblock trees)
  int no_cell_ranks;    // testing: sort the full varying Morton code
  int do_color;         // the frame has colours (they ride in the key or in the payload)
  int passes_launched;  // sort passes enqueued: a frame that needs more comes back with kErrPasses
  int deep_launched;    // the DEEP instantiations are enqueued: a deep frame that meets the others comes back with kErrDeep
};

__global__ __launch_bounds__(kBlock) void k_boxes_events(PointView pv, uint32_t n, uint32_t n_chunks, uint64_t* boxes, uint32_t seq,
                                                         uint4* __restrict__ sync_area, uint32_t sync_vec16, double res, PlanOptions opt, FixedBox box,
                                                         FrameState* __restrict__ st, unsigned long long* span) {
  const KSpan kspan(span);
  __shared__ float s_p[3][kTile];  // workgroup 0: the chunk being replayed; the others: a few words for their reduction
  __shared__ int s_redi[2][kBlock / 64];
  if (blockIdx.x != 0) {
    chunk_box_block(pv, n, blockIdx.x - 1u, n_chunks, boxes, seq, sync_area, sync_vec16, &s_p[0][0], &s_redi[0][0]);
    return;
  }
  const int force_pairs = opt.force_pairs, need_index = opt.need_index, no_cell_ranks = opt.no_cell_ranks, do_color = opt.do_color;
  const int passes_launched = opt.passes_launched, deep_launched = opt.deep_launched;
  __shared__ int ev_index[kMaxEpochs], ev_lowered[kMaxEpochs], ev_depth_before[kMaxEpochs];
  __shared__ double ev_mn[kMaxEpochs][3];
  __shared__ float s_g[6][kBlock / 64];
  __shared__ unsigned s_nfin;

  PCC_KTR(6, 0);
  const double eps = (double)FLT_EPSILON;  // PCL: const float minValue = numeric_limits<float>::epsilon()
  int parity = 0;
  // one barrier per reduction; the result buffers alternate so that a fast wave cannot overwrite a pending read
  auto block_min1 = [&](int v) {
    v = wave_min_i(v);
    if (lane_id() == 0) s_redi[parity][wave_id()] = v;
    __syncthreads();
    const int r = min(min(s_redi[parity][0], s_redi[parity][1]), min(s_redi[parity][2], s_redi[parity][3]));
    parity ^= 1;
    return r;
  };

  // the bounding box, replayed identically by every thread in registers: a growth event costs one barrier.
  // Points and chunk boxes are floats: p < mn  <=>  p < RU(mn) and p >= mx  <=>  p >= RU(mx) with RU = the double
  // rounded up to float, so the tests run on floats and give what PCL's double comparisons give.
  double mn[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
  float fmn[3] = {0, 0, 0}, fmx[3] = {0, 0, 0};
  int depth = 0, nev = 0, err = kErrNone;
  bool have_box = false;
  int i0 = 0x7fffffff, cur = 0, loaded = -1;
  auto float_bounds = [&]() {
    for (int a = 0; a < 3; ++a) { fmn[a] = __double2float_ru(mn[a]); fmx[a] = __double2float_ru(mx[a]); }
  };
  auto violates = [&](float x, float y, float z) {
    return (x < fmn[0]) | (y < fmn[1]) | (z < fmn[2]) | (x >= fmx[0]) | (y >= fmx[1]) | (z >= fmx[2]);
  };

  auto load_chunk = [&](int c) {  // into LDS; NaN never violates: non-finite points are skipped
    __syncthreads();              // nobody reads the chunk before any more
#pragma unroll
    for (int k = 0; k < kItems; ++k) {
      const int e = k * kBlock + (int)threadIdx.x;
      const uint32_t i = (uint32_t)c * kTile + (uint32_t)e;
      float x = __builtin_nanf(""), y = x, z = x;
      if (i < n) {
        load_xyz(pv, i, x, y, z);
        if (!finite3(x, y, z)) { x = y = z = __builtin_nanf(""); }
      }
      s_p[0][e] = x; s_p[1][e] = y; s_p[2][e] = z;
    }
    loaded = c;
    __syncthreads();
  };
  // adoptBoundingBoxToPoint, empty-tree branch + getKeyBitSize: the box around the first finite point
  auto first_box = [&](float fx, float fy, float fz, int index) {
    const float p[3] = {fx, fy, fz};
    for (int a = 0; a < 3; ++a) {
      mn[a] = __dsub_rn((double)p[a], __ddiv_rn(res, 2.0));
      mx[a] = __dadd_rn((double)p[a], __ddiv_rn(res, 2.0));
      if (box.enabled) { mn[a] = box.mn[a]; mx[a] = box.mx[a]; }  // defineBoundingBox, then getKeyBitSize as below
    }
    unsigned max_voxels = 2;
    for (int a = 0; a < 3; ++a) {
      const unsigned k = (unsigned)ceil(__ddiv_rn(__dsub_rn(__dsub_rn(mx[a], mn[a]), eps), res));
      max_voxels = max(max_voxels, k);
    }
    depth = (int)ceil(__dsub_rn(log2((double)max_voxels), eps));
    depth = min(depth, 32);
    const double side = __dmul_rn((double)(1u << depth), res);
    for (int a = 0; a < 3; ++a) {
      const double over = __ddiv_rn(__dsub_rn(side, __dsub_rn(mx[a], mn[a])), 2.0);
      if (over > eps) { mn[a] = __dsub_rn(mn[a], over); mx[a] = __dadd_rn(mx[a], over); }
    }
    float_bounds();
    if (threadIdx.x == 0) {
      for (int a = 0; a < 3; ++a) ev_mn[0][a] = mn[a];
      ev_index[0] = index; ev_lowered[0] = 0; ev_depth_before[0] = 0;
    }
    nev = 1;
    have_box = true;
    i0 = index;
    cur = box.enabled ? index : index + 1;  // a given box has to be checked against the first point too
  };
  // adoptBoundingBoxToPoint, bounding_box_defined_ branch: the box doubles towards the point until it holds it
  // (uniform: every calling thread replays it in its own registers; thread 0 records the events)
  auto grow_to = [&](float fx, float fy, float fz, int index) {
    const double p[3] = {(double)fx, (double)fy, (double)fz};
    for (;;) {
      bool up[3], any = false;
      for (int a = 0; a < 3; ++a) { up[a] = p[a] >= mx[a]; any |= (p[a] < mn[a]) | up[a]; }
      if (!any) break;
      if (nev >= kMaxEpochs || depth >= 31) { err = kErrEpochs; break; }
      double side = __dmul_rn((double)(1u << depth), res);
      int lowered = 0;
      for (int a = 0; a < 3; ++a)
        if (!up[a]) { mn[a] = __dsub_rn(mn[a], side); lowered |= 1 << a; }
      if (threadIdx.x == 0) {
        ev_depth_before[nev] = depth;
        ev_index[nev] = index;
        ev_lowered[nev] = lowered;
        for (int a = 0; a < 3; ++a) ev_mn[nev][a] = mn[a];
      }
      ++depth;
      side = __dsub_rn(__dmul_rn((double)(1u << depth), res), eps);
      for (int a = 0; a < 3; ++a) mx[a] = __dadd_rn(mn[a], side);
      ++nev;
    }
    float_bounds();
  };
  // all growth events of the loaded chunk from `cur` on
#ifdef PCC_KTIME
  int dbg_round = 0;
#endif
  auto replay_loaded_chunk = [&]() {
    while (err == kErrNone) {
      const int start_e = max(cur - loaded * kTile, 0);
      int ce = 0x7fffffff;
#pragma unroll
      for (int k = kItems - 1; k >= 0; --k) {
        const int e = k * kBlock + (int)threadIdx.x;
        if (e >= start_e && violates(s_p[0][e], s_p[1][e], s_p[2][e])) ce = e;
      }
      const int emin = block_min1(ce);
#ifdef PCC_KTIME
      if (dbg_round < 8) { PCC_KTR(8, dbg_round); ++dbg_round; }
#endif
      if (emin == 0x7fffffff) {  // nothing (left) in this chunk: go on with the chunks behind it
        cur = (loaded + 1) * kTile;
        return;
      }
      grow_to(s_p[0][emin], s_p[1][emin], s_p[2][emin], loaded * kTile + emin);
      cur = loaded * kTile + emin + 1;
    }
  };
  // The same for 64 consecutive elements of the loaded chunk, by wave 0 alone: a lane holds one point, the first
  // violating lane comes from a ballot and its coordinates from a broadcast -- no LDS round, no barrier per event.
  // With points in random order the box reaches its final size within the first few dozen points, so this takes
  // the events that used to cost one block-wide round each.  The other waves pick the state up from LDS.
  __shared__ double s_state_d[6];
  __shared__ int s_state_i[4];
  auto replay_first_lanes = [&]() {
    const int e0 = max(cur - loaded * kTile, 0);
    if (wave_id() == 0) {
      const int e = e0 + lane_id();
      float x = __builtin_nanf(""), y = x, z = x;
      if (e < kTile) { x = s_p[0][e]; y = s_p[1][e]; z = s_p[2][e]; }
      int from = 0;  // lanes below `from` are behind the replay
      while (err == kErrNone) {
        const uint64_t viol = __ballot(lane_id() >= from && violates(x, y, z));
        if (viol == 0ull) break;
        const int l = __ffsll((long long)viol) - 1;
        grow_to(lane_of(x, l), lane_of(y, l), lane_of(z, l), loaded * kTile + e0 + l);
        from = l + 1;
      }
      if (lane_id() == 0) {
        for (int a = 0; a < 3; ++a) { s_state_d[a] = mn[a]; s_state_d[3 + a] = mx[a]; }
        s_state_i[0] = depth; s_state_i[1] = nev; s_state_i[2] = err;
      }
    }
    __syncthreads();
    for (int a = 0; a < 3; ++a) { mn[a] = s_state_d[a]; mx[a] = s_state_d[3 + a]; }
    depth = s_state_i[0]; nev = s_state_i[1]; err = s_state_i[2];
    float_bounds();
    cur = loaded * kTile + min(e0 + 64, kTile);
  };

  // ---- early: chunk 0, while the other workgroups read the cloud ----
  uint64_t pre_w[kBoxWords] = {0, 0, 0, 0, 0, 0, 0, 0};  // (sequence numbers start at 1: all zero = not there)
  load_chunk(0);
  {
    int ce = 0x7fffffff;
#pragma unroll
    for (int k = kItems - 1; k >= 0; --k) {
      const int e = k * kBlock + (int)threadIdx.x;
      const float x = s_p[0][e];
      if (x == x) ce = e;
    }
    const int f = block_min1(ce);
    PCC_KTR(6, 1);
    if (f != 0x7fffffff) {
      first_box(s_p[0][f], s_p[1][f], s_p[2][f], f);
      PCC_KTR(8, 5);
#ifndef PCC_WAVE_OPS_SHFL  // (the switch also restores the block-wide rounds for the first points: same commit as the DPP helpers)
      replay_first_lanes();
#else
      (void)replay_first_lanes;
#endif
      PCC_KTR(8, 6);
      // the first chunk box of every thread is requested now and looked at after the block-wide rounds below: by then
      // the streaming workgroups have usually published, and the sweep starts with its first round trip behind it
      if (threadIdx.x < n_chunks) {
#pragma unroll
        for (int k = 0; k < kBoxWords; ++k) pre_w[k] = poll_u64(boxes + (size_t)threadIdx.x * kBoxWords + k);
      }
      replay_loaded_chunk();
    }
  }
  PCC_KTR(6, 2);

  // ---- A: the chunk boxes as they appear (workgroups 1..n_chunks never wait, so this terminates): first finite
  //         point, finite count, global AABB; in the same sweep, the first chunk behind the loaded one whose box
  //         does not fit the bounding box as it stands.  A thread takes its chunks in ascending order and comes back
  //         to the first one that was not there yet.
  if (threadIdx.x == 0) s_nfin = 0;
  st->occ_hist[threadIdx.x] = 0u;  // kBlock = 256 threads: one counter each (filled by k_occ_histogram at the end of the frame)
  if (threadIdx.x == 0) st->jpeg_line_words = 0u;
  int first = 0x7fffffff, cand = 0x7fffffff;
  unsigned nfin = 0;
  float g[6] = {FLT_MAX, FLT_MAX, FLT_MAX, -FLT_MAX, -FLT_MAX, -FLT_MAX};
  auto box_violates = [&](const ChunkBox& b) { return violates(b.mn[0], b.mn[1], b.mn[2]) | violates(b.mx[0], b.mx[1], b.mx[2]); };
  {
    uint32_t c = threadIdx.x, spins = 0;
    for (;;) {
      while (c < n_chunks) {
        ChunkBox b;
        if (!(c == threadIdx.x && spins == 0 && decode_box(pre_w, seq, b)) && !fetch_box(boxes + (size_t)c * kBoxWords, seq, b)) break;
        if (b.n_finite > 0) {
          first = min(first, b.first_finite);
          nfin += (unsigned)b.n_finite;
          for (int a = 0; a < 3; ++a) { g[a] = fminf(g[a], b.mn[a]); g[3 + a] = fmaxf(g[3 + a], b.mx[a]); }
          if (have_box && (int)c > loaded && cand == 0x7fffffff && box_violates(b)) cand = (int)c;  // ascending per thread
        }
        c += kBlock;
      }
      if (__syncthreads_and(c >= n_chunks)) break;
      __builtin_amdgcn_s_sleep(16);
      if (++spins > kSpinLimit) { err = kErrSpin; break; }  // uniform: every thread counts the same rounds
    }
  }
  PCC_KTR(6, 3);
  nfin = (unsigned)wave_sum_u64(nfin);
  for (int a = 0; a < 3; ++a) { g[a] = wave_min_f(g[a]); g[3 + a] = wave_max_f(g[3 + a]); }
  if (lane_id() == 0) {
    atomicAdd(&s_nfin, nfin);
    for (int a = 0; a < 6; ++a) s_g[a][wave_id()] = g[a];
  }
  int pending = block_min1(have_box ? cand : first);  // (its barrier also covers s_nfin and s_g)
  bool pending_valid = have_box;
  if (!have_box && err == kErrNone) {
    if (pending == 0x7fffffff) {  // no finite point: the reference drops the frame (impl.hpp:206-212)
      if (threadIdx.x == 0) {
        st->n_epochs = 0; st->depth = 0; st->n_finite = 0; st->first_finite = -1;
        st->n_leaves = 0; st->n_branches = 0; st->npasses = 0; st->error = kErrNone;
        st->vbits = 0; st->vbits_axis = 0; st->ibits = 0; st->n_growth_events = 0; st->packed = 1; st->payload = 0; st->colour_in_key = 0; st->deep = 0; st->payload2 = 0; st->keys_final = 0;
        st->code_low_bits = 0; st->code_bits = 0;
        st->passes_launched = passes_launched;
      }
      return;
    }
    float fx, fy, fz;  // chunk 0 held no finite point: the first one of the cloud comes from the chunk boxes
    load_xyz(pv, (uint32_t)pending, fx, fy, fz);
    first_box(fx, fy, fz, pending);
  }
  PCC_KTR(6, 4);

  // ---- C: walk forward; only chunks whose AABB violates the current box are opened ----
  while (cur < (int)n && err == kErrNone) {
    const int c0 = cur / kTile;
    if (c0 != loaded) {
      int cmin;
      if (pending_valid) {
        cmin = pending;  // found during the sweep above, for exactly this box and this range of chunks
      } else {
        int cd = 0x7fffffff;
        for (int c = c0 + (int)threadIdx.x; c < (int)n_chunks; c += kBlock) {
          ChunkBox b;
          (void)fetch_box(boxes + (size_t)c * kBoxWords, seq, b);  // all there since the sweep
          if (b.n_finite > 0 && box_violates(b)) { cd = c; break; }  // ascending per thread: the first hit is this thread's minimum
        }
        cmin = block_min1(cd);
      }
      if (cmin == 0x7fffffff) break;  // everything that is left fits
      load_chunk(cmin);
      cur = max(cur, cmin * kTile);
    }
    pending_valid = false;
    replay_loaded_chunk();
  }
  __syncthreads();  // ev_* complete
  PCC_KTR(6, 5);

  // ---- D: epoch table (one thread per growth event), sort geometry ----
  // an epoch = a run of point indices with one box origin; the same point may grow the box several times, the last
  // growth of such a run stands for the epoch.  The key offset of an epoch is the sum of the re-rootings that came
  // after it.
  if (wave_id() == 0) {
    const int k = lane_id();  // kMaxEpochs <= 64
    const bool live = k < nev;
    const bool last_of_run = live && !(k + 1 < nev && ev_index[k + 1] == ev_index[k]);
    const uint64_t runs = __ballot(last_of_run);
    const int ne = __popcll(runs);
    const int w = __popcll(runs & (k ? (~0ull >> (64 - k)) : 0ull));
    uint32_t later[3];
    for (int a = 0; a < 3; ++a) {
      const uint32_t add = (live && (ev_lowered[k] & (1 << a))) ? (1u << ev_depth_before[k]) : 0u;
      const uint32_t incl = wave_incl_scan_u32(add);
      later[a] = lane63(incl) - incl;  // growths after event k
    }
    if (last_of_run) {
      st->ep_index[w] = ev_index[k];
      for (int a = 0; a < 3; ++a) { st->ep_mn[w][a] = ev_mn[k][a]; st->ep_shift[w][a] = later[a]; }
    }
    if (k == 0) {
      st->n_epochs = ne;
      st->n_growth_events = nev - 1;
      st->depth = depth;
      st->first_finite = i0;
      st->n_finite = s_nfin;
      for (int a = 0; a < 3; ++a) { st->mn[a] = mn[a]; st->mx[a] = mx[a]; }
      st->passes_launched = passes_launched;
      st->n_leaves = 0;
      st->n_branches = 0;
    }
  } else if (wave_id() == 1) {
    if (depth > kMaxDepthDeep && err == kErrNone) err = kErrDepth;
    // two-word codes (pcc_device.h): no cell ranks, pairs of payloads.  Which instantiations of the kernels
    // behind this one were enqueued is the host's decision (it cannot know the depth): a frame in the deep sequence is
    // treated as deep whatever its depth, a deep frame in the single-word sequence is sent back (kErrDeep)
    const bool deep = deep_launched != 0;
    if (depth > kMaxDepth && !deep && err == kErrNone) err = kErrDeep;
    // varying key bits from the global AABB under the final origin, +-1 voxel of slack
    int vb = 0;
    unsigned kmin[3], kmax[3];
    const unsigned klim = depth >= 32 ? 0xffffffffu : ((1u << depth) - 1u);
    for (int a = 0; a < 3; ++a) {
      float gmin = FLT_MAX, gmax = -FLT_MAX;
      for (int w = 0; w < kBlock / 64; ++w) { gmin = fminf(gmin, s_g[a][w]); gmax = fmaxf(gmax, s_g[3 + a][w]); }
      const double lo = __ddiv_rn(__dsub_rn((double)gmin, mn[a]), res);
      const double hi = __ddiv_rn(__dsub_rn((double)gmax, mn[a]), res);
      unsigned kl = lo > 0.0 ? (unsigned)lo : 0u;
      unsigned kh = hi > 0.0 ? (unsigned)hi : 0u;
      kl = kl > 0 ? kl - 1 : 0;
      kh = kh < klim ? kh + 1 : klim;
      kmin[a] = kl; kmax[a] = kh;
      const unsigned x = kl ^ kh;
      const int nb = x ? 32 - __clz((int)x) : 0;
      vb = max(vb, nb);
    }
    // ---- cell ranks (FrameState::code_low_bits): the shortest sorted code over the low-bit counts m that leave at most
    //      64 cells; fewer passes first, then fewer cells
    auto passes_for = [](int bits) { return bits <= kMaxDigitBits ? 1 : (bits + kMaxDigitBits - 1) / kMaxDigitBits; };
    int cm = vb, cbits = 3 * vb;
    unsigned cdim[3] = {1u, 1u, 1u};
    for (int m = vb - 1; m >= 0 && vb < 32; --m) {
      unsigned d[3];
      unsigned long long nc = 1;
      for (int a = 0; a < 3; ++a) { d[a] = (kmax[a] >> m) - (kmin[a] >> m) + 1u; nc *= d[a]; }
      if (nc > 64ull) break;
      const int rb = nc > 1ull ? 32 - __clz((int)(nc - 1ull)) : 0;
      const int bits = 3 * m + rb;
      if (passes_for(bits) < passes_for(cbits)) { cm = m; cbits = bits; for (int a = 0; a < 3; ++a) cdim[a] = d[a]; }
    }
    if (no_cell_ranks || deep) { cm = vb; cbits = 3 * vb; cdim[0] = cdim[1] = cdim[2] = 1u; }
    {
      const unsigned nc = cdim[0] * cdim[1] * cdim[2];
      const int i = lane_id();
      const unsigned hm = (vb - cm) >= 32 ? 0xffffffffu : ((1u << (vb - cm)) - 1u);  // the varying part of a cell coordinate
      uint64_t mort = ~0ull;
      if ((unsigned)i < nc) {
        const unsigned dz = (unsigned)i % cdim[2], dy = ((unsigned)i / cdim[2]) % cdim[1], dx = (unsigned)i / (cdim[2] * cdim[1]);
        mort = morton3(((kmin[0] >> cm) + dx) & hm, ((kmin[1] >> cm) + dy) & hm, ((kmin[2] >> cm) + dz) & hm);
      }
      unsigned rank = 0;
      for (unsigned j = 0; j < nc && nc > 1u; ++j) {  // (one cell: no ranks, nobody reads the tables)
        const uint64_t other = lane_of(mort, (int)j);
        rank += (other < mort) ? 1u : 0u;
      }
      if ((unsigned)i < nc && nc > 1u) { st->cell_rank[i] = (uint8_t)rank; st->cell_abs[rank] = mort; }
      if (i < 3) { st->cell_base[i] = kmin[i] >> cm; st->cell_dim[i] = cdim[i]; }
      if (i == 0) { st->code_low_bits = cm; st->code_bits = cbits; }
    }
    int ibits = 32 - __clz((int)n);  // bit length of n: index < 2^ibits - 1
    // Nothing downstream needs the point index unless centroids are coded or the caller wants the sorted points
    // themselves (macroblock trees): the sort is stable, so equal codes keep their input order without it.  Then the
    // key is the code alone, or [code | the point's 24 colour bits] -- 8 B per key and pass, no payload array.
    // Otherwise code + index in one u64 when they fit (+ the colour word as a u32 payload: 12 B); otherwise u64 code
    // keys with a u32 index payload (12 B).  All of them give the same order.
    const bool bare = !deep && !need_index && !force_pairs && (!do_color || cbits + 24 <= 63);
    const int packed = (!deep && (bare || (cbits + ibits <= 64 && !force_pairs))) ? 1 : 0;
    if (bare) ibits = do_color ? 24 : 0;
    if (!packed) ibits = 0;
    // digit plan: as few passes as 9-bit digits allow, the code bits spread evenly over them.  Deep trees: the low word's
    // bits first, then the high word's (a digit never straddles the two; a shift of 63 or more means "of the high word")
    const int vbits = cbits;  // what the passes sort
    const int lo_bits = vbits < 63 ? vbits : 63, hi_bits = vbits - lo_bits;
    int np_lo = (lo_bits + kMaxDigitBits - 1) / kMaxDigitBits;
    if (np_lo < 1) np_lo = 1;
    const int np_hi = (hi_bits + kMaxDigitBits - 1) / kMaxDigitBits;
    const int np = np_lo + np_hi;
    if (err == kErrNone && np > passes_launched) err = kErrPasses;  // the host re-launches with more passes
    // digit q of `n` digits over `bits` bits: width, and offset of its first bit
    auto digit = [](int q, int n, int bits, int& width, int& offset) {
      width = 0; offset = 0;
      for (int r = 0; r <= q; ++r) {
        int b = 0;
        if (r < n) { b = bits / n + (r < bits % n ? 1 : 0); if (b < 1) b = 1; }
        if (r < q) offset += b; else width = b;
      }
    };
    const int p = lane_id();
    if (p < kMaxPasses) {
      int bits = 0, sh = 0;
      if (p < np_lo) digit(p, np_lo, lo_bits, bits, sh);
      else if (p < np) { digit(p - np_lo, np_hi, hi_bits, bits, sh); sh += 63; }
      else { bits = 0; sh = vbits; }
      st->pass_bits[p] = bits;
      st->pass_shift[p] = sh;
    }
    if (p == 8) {
      for (int a = 0; a < 3; ++a) st->prefix[a] = vb >= 32 ? 0u : ((kmin[a] >> vb) << vb);
      st->vbits_axis = vb;
      st->vbits = 3 * vb;
      st->ibits = ibits;
      st->packed = packed;
      st->payload = deep ? 3 : (bare ? 0 : (packed ? (do_color ? 2 : 0) : 1));
      st->colour_in_key = (bare && do_color) ? 1 : 0;
      st->deep = deep ? 1 : 0;
      st->payload2 = !deep ? 0 : (need_index ? 1 : (do_color ? 2 : 0));
      st->keys_final = np & 1;
      st->npasses = err != kErrNone ? 0 : np;
      st->error = err;
    }
  }
  PCC_KTR(6, 6);
}

// ------------------------------------------------------------------------------------------
// Stage 2: octree keys (P3) -> packed sort keys  [morton(vbits) | point index(ibits)]; a non-finite
// point gets the marker ~0 and is dropped by the first sort pass.  The same kernel counts, per
// 4096-point tile, the digits of EVERY sort pass in LDS and writes them as one row of hist_rows.
// ------------------------------------------------------------------------------------------

// (1024 threads x 4 points; 512 x 8 was tried for the sake of frames in flight -- a smaller workgroup finds room on a
// busy CU sooner, which took k_digit_totals from 22 to 14 us under load -- but here it lost both ways: 15.4 -> 18.9 us
// alone, 38.6 -> 41.9 us under load)
template <int kKeyThreads, int KEY_ITEMS>
__global__ __launch_bounds__(kKeyThreads) void k_make_keys(PointView pv, uint32_t n, double res, double inv_res_pow2,
                                                            FrameState* __restrict__ st, uint64_t* __restrict__ keys,
                                                            uint32_t* __restrict__ idx, uint32_t* __restrict__ idx2, uint16_t* __restrict__ hist_rows,
                                                            unsigned long long* span) {
  const KSpan kspan(span);
  constexpr int kKeyTile = kKeyThreads * KEY_ITEMS;
  static_assert(kKeyTile == kSortTile, "one row of digit counts per sort tile");
  __shared__ uint32_t s_h[kMaxPasses][kMaxBins];
  const int ne = st->n_epochs;
  // No plan (the frame holds no finite point, or k_boxes_events refused it: npasses is 0 then): nothing to do.  The error
  // word itself is NOT looked at here: workgroups of this very launch store into it (kErrPrefix, below), and lanes that read
  // it before and after such a store would disagree about leaving -- part of a workgroup at the barriers below, counting
  // digits in LDS nobody zeroed (found by the happens-before checker of tests/emu as reads of uninitialised LDS).
  if (ne == 0 || st->npasses == 0) return;
  const uint32_t tile = blockIdx.x;  // one workgroup per sort tile
  KeyGeom g;
  g.np = st->npasses;
  g.vb = st->vbits_axis; g.ibits = st->ibits;
  g.packed_mode = st->packed != 0;
  g.payload = st->payload;
  g.colour_in_key = st->colour_in_key != 0;
  g.deep = st->deep != 0; g.payload2 = st->payload2;
  g.cm = st->code_low_bits;
  g.ranked = g.cm < g.vb;  // the high key bits go into the code as the rank of their cell (FrameState::code_low_bits)
#pragma unroll
  for (int a = 0; a < 3; ++a) { g.cbase[a] = st->cell_base[a]; g.cdim[a] = st->cell_dim[a]; g.prefix[a] = st->prefix[a]; }
  g.lm = g.cm >= 32 ? 0xffffffffu : ((1u << g.cm) - 1u);
  g.m = g.vb >= 32 ? 0xffffffffu : ((1u << g.vb) - 1u);
  const int ep0 = st->ep_index[0], ep_last = st->ep_index[ne - 1];
#pragma unroll
  for (int p = 0; p < kMaxPasses; ++p) { g.pshift[p] = st->pass_shift[p]; g.pmask[p] = (1u << st->pass_bits[p]) - 1u; }
  for (int k = threadIdx.x; k < g.np * kMaxBins; k += kKeyThreads) (&s_h[0][0])[k] = 0u;
  __syncthreads();
  const uint32_t base = tile * kKeyTile;
  const bool late = (int)base >= ep_last;  // the whole tile lies in the last epoch (all but the first tiles)
#pragma unroll
  for (int k = 0; k < KEY_ITEMS; ++k) {
    const uint32_t i = base + k * kKeyThreads + threadIdx.x;
    if (i >= n) break;
    float x, y, z;
    load_xyz(pv, i, x, y, z);
    uint64_t key = kInvalidKey;
    uint32_t hi = 0u;
    if (finite3(x, y, z) && (int)i >= ep0) {
      int e = ne - 1;
      if (!late) {  // rare: the tile overlaps an earlier epoch
        while (e > 0 && st->ep_index[e] > (int)i) --e;
      }
      bool ok = true;
      const uint64_t code = point_code(g, st->ep_mn[e], st->ep_shift[e], st->cell_rank, res, inv_res_pow2, x, y, z, ok, hi);
      if (!ok) raise_error(st, kErrPrefix);  // the +-1 voxel slack was not enough: refuse rather than mis-sort
#pragma unroll
      for (int p = 0; p < kMaxPasses; ++p)
        if (p < g.np) atomicAdd(&s_h[p][code_digit(g, p, code, hi)], 1u);
      // low bits: the point index, or (nobody needs the index) the point's colour, or nothing
      const uint64_t low = g.colour_in_key ? (uint64_t)(load_rgba(pv, i) & 0xffffffu) : (g.ibits ? (uint64_t)i : 0ull);
      key = g.packed_mode ? ((code << g.ibits) | low) : code;
    }
    if (g.payload == 1) idx[i] = i;
    else if (g.payload == 2) idx[i] = load_rgba(pv, i);  // same 32-byte point as x,y,z: no extra traffic
    else if (g.payload == 3) idx[i] = hi;                // deep trees: the high word of the code
    if (g.payload2 == 1) idx2[i] = i;
    else if (g.payload2 == 2) idx2[i] = load_rgba(pv, i);
    keys[i] = key;
  }
  __syncthreads();
  uint16_t* row = hist_rows + (size_t)tile * kMaxPasses * kMaxBins;  // (a tile has at most 4096 keys: a count fits 16 bits)
  for (int k = threadIdx.x; k < g.np * kMaxBins; k += kKeyThreads) row[k] = (uint16_t)(&s_h[0][0])[k];
}

// column sums of hist_rows: digit_tot[pass][digit] = number of keys with that digit in that pass.
// One workgroup per 16 columns, 64 row groups (contiguous row ranges) per column, LDS combine: the grid is a few
// hundred workgroups however many tiles there are, and a thread walks rows / 64 of them.
// For pass 0 the input order of the sort is the tile order of k_make_keys, so the same kernel also
// writes the exclusive prefix over the tiles (tile_prefix0[tile][digit]): pass 0 needs no look-back.
// (256 threads per workgroup for frames of up to 512 tiles: a 1024-thread workgroup needs sixteen free wave slots on ONE
// CU at the same moment, and with the kernels of three other frames on the GPU it waits for them: 22 us under load
// against 5.5 us alone; the narrow shape 14 us against 7 us)
// (for frames of many tiles the wide shape stays: a thread of the narrow one would walk rows / 16 of them)
constexpr uint32_t kDtCols = 16;
template <uint32_t kDtThreads>
__global__ __launch_bounds__(kDtThreads) void k_digit_totals(const FrameState* __restrict__ st, uint32_t n_rows,
                                                       const uint16_t* __restrict__ hist_rows,
                                                       uint32_t* __restrict__ digit_tot, uint32_t* __restrict__ tile_prefix0, unsigned long long* span) {
  const KSpan kspan(span);
  constexpr uint32_t kDtGroups = kDtThreads / kDtCols;
  __shared__ uint32_t s_part[kDtGroups][kDtCols];
  const uint32_t c = threadIdx.x % kDtCols;
  // 16 columns are half a 128-byte line of every row: the two workgroups that share the lines are put on the same XCD
  // (workgroup b runs on XCD b mod 8: every XCD takes a contiguous eighth of the column blocks; the grid is a multiple of 8)
  const uint32_t cb = (gridDim.x & 7u) ? blockIdx.x : (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const uint32_t col = cb * kDtCols + c;
  const uint32_t pass = col / kMaxBins;
  if ((int)pass >= st->npasses) return;  // uniform per workgroup (512 columns per pass)
  const uint32_t g = threadIdx.x / kDtCols;
  const uint32_t per = (n_rows + kDtGroups - 1u) / kDtGroups;
  const uint32_t r0 = min(g * per, n_rows), r1 = min(r0 + per, n_rows);
  // The narrow shape serves frames of up to 512 tiles: a thread's rows (at most 32) are requested together and stay in
  // registers for the prefixes of pass 0 below, instead of a walk with a load per step and a second walk over the same rows.
  constexpr uint32_t kHeld = kDtThreads == 256 ? 32u : 1u;
  const bool held = kDtThreads == 256 && per <= kHeld;
  uint32_t mine[kHeld];
  uint32_t acc = 0;
  if (held) {
#pragma unroll
    for (uint32_t k = 0; k < kHeld; ++k) mine[k] = (r0 + k < r1) ? hist_rows[(size_t)(r0 + k) * kMaxPasses * kMaxBins + col] : 0u;
#pragma unroll
    for (uint32_t k = 0; k < kHeld; ++k) acc += mine[k];
  } else {
    for (uint32_t r = r0; r < r1; ++r) acc += hist_rows[(size_t)r * kMaxPasses * kMaxBins + col];
  }
  s_part[g][c] = acc;
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t k = 0; k < kDtGroups; ++k) {
    const uint32_t v = s_part[k][c];
    if (k < g) before += v;
    all += v;
  }
  if (g == 0) digit_tot[col] = all;
  if (pass == 0 && held) {
    uint32_t run = before;
#pragma unroll
    for (uint32_t k = 0; k < kHeld; ++k) {
      if (r0 + k < r1) tile_prefix0[(size_t)(r0 + k) * kMaxBins + col] = run;
      run += mine[k];
    }
  } else if (pass == 0) {
    uint32_t run = before;
    for (uint32_t r = r0; r < r1; ++r) {
      tile_prefix0[(size_t)r * kMaxBins + col] = run;
      run += hist_rows[(size_t)r * kMaxPasses * kMaxBins + col];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Stage 3: stable LSD radix sort, ONE kernel per pass ("onesweep"): a tile of 4096 keys ranks its
// keys per digit with wave ballots against wave-private LDS counters, publishes its digit counts
// and finds the counts of all earlier tiles by decoupled look-back (one self-describing word per
// (tile, digit): flag | count), then scatters.  Tile ids come from a ticket counter, so a tile only
// ever waits for tiles that have already started.  Passes beyond st->npasses return at once.
// ------------------------------------------------------------------------------------------
// (second launch bound = waves per SIMD the register allocation has to leave room for: four, i.e. one 1024-thread or
// two 512-thread workgroups per CU.  Without it the 512-thread shape takes 151 registers -- it is allowed 256 -- and
// only one workgroup fits a CU, which is the whole point of that shape gone.)
constexpr size_t kTicketBytes = ((kMaxPasses + 1) * sizeof(uint32_t) + 15) / 16 * 16;  // tickets[pass], tickets[kMaxPasses] = the leaf scan's
// (DEEP: two-word codes -- the u32 payload is the code's high word, whose digits the last passes sort by, and a second
// payload array carries the point index or the colour word)
template <int THREADS, int ITEMS, bool DEEP = false>
__global__ __launch_bounds__(THREADS, 4) void k_sort_pass(const uint64_t* buf_a, const uint64_t* buf_b,
                                                            uint64_t* out_a, uint64_t* out_b,
                                                            uint32_t* idx_a, uint32_t* idx_b, uint32_t* idx2_a, uint32_t* idx2_b, uint32_t n, int pass,
                                                            FrameState* st, const uint32_t* __restrict__ digit_tot,
                                                            const uint32_t* __restrict__ tile_prefix0,
                                                            uint32_t* status_all, uint32_t* tickets,
                                                            uint32_t n_tiles_max, unsigned long long* span) {
  const KSpan kspan(span);
  PCC_KT(0);
  if (pass >= st->npasses) return;
  constexpr int NW = THREADS / 64;
  static_assert(THREADS * ITEMS == kSortTile && THREADS >= kMaxBins, "a tile is 4096 keys (the histogram rows of k_make_keys); one thread per digit");
  // s_raw is used twice: while ranking, one 64-bit lane mask per (wave, digit); afterwards the tile's
  // keys (and payload) in digit order, so that the global writes are runs
  constexpr int kTileWords = DEEP ? kSortTile * 2 : kSortTile * 3 / 2;  // keys + one payload (+ a second one)
  constexpr int kRawWords = NW * kMaxBins > kTileWords ? NW * kMaxBins : kTileWords;
  __shared__ __attribute__((aligned(16))) uint64_t s_raw[kRawWords];
  uint64_t* s_match = s_raw;
  uint64_t* s_keys = s_raw;
  uint32_t* s_pay = reinterpret_cast<uint32_t*>(s_raw + kSortTile);
  uint32_t* s_pay2 = s_pay + kSortTile;  // DEEP only
  __shared__ uint16_t s_cnt[NW][kMaxBins];  // per-wave running digit counts, then wave start ranks
  __shared__ uint32_t s_gofs[kMaxBins];     // global position of a digit's first key minus its position in s_keys
  __shared__ uint16_t s_dstart[kMaxBins];   // position of a digit's first key in s_keys
  __shared__ uint32_t s_scan[NW];
  __shared__ uint32_t s_tile;

  const uint32_t count = pass == 0 ? n : st->n_finite;  // pass 0 still holds the non-finite markers
  const uint32_t out_count = st->n_finite;
  const uint32_t n_tiles = (count + kSortTile - 1) / kSortTile;
  const int bits = st->pass_bits[pass];
  const uint32_t nbins = 1u << bits, mask = nbins - 1u;
  const uint32_t d_me = threadIdx.x;  // thread = digit in the per-digit steps
  // Tile id = a ticket: a tile only ever waits for lower tile ids, and those belong to workgroups that have
  // started.  (blockIdx would do only if the whole grid were co-resident; with several frames in flight on
  // other streams it is not, and workgroups are dispatched per XCD: a resident workgroup could then wait for
  // one that cannot start because of workgroups waiting the other way round.)  Pass 0 waits for nobody -- its
  // tile prefixes come from k_digit_totals -- and keeps blockIdx.  The ticket, 245 workgroups queueing on one
  // word, takes a microsecond or two to come back: the digit totals are fetched meanwhile.
  uint32_t ticket = blockIdx.x;
  if (pass != 0 && threadIdx.x == 0) ticket = atomicAdd(&tickets[pass], 1u);
  const uint32_t dtot = d_me < nbins ? digit_tot[(size_t)pass * kMaxBins + d_me] : 0u;
  for (int k = threadIdx.x; k < NW * kMaxBins / 2; k += THREADS) reinterpret_cast<uint32_t*>(&s_cnt[0][0])[k] = 0u;
  for (int k = threadIdx.x; k < NW * kMaxBins; k += THREADS) s_match[k] = 0ull;
  if (threadIdx.x == 0) s_tile = ticket;
  // global start of every digit = exclusive scan of the digit totals (its barriers also cover the LDS set up above)
  uint32_t gsum;
  const uint32_t gbase = block_excl_scan<NW, uint32_t>(dtot, s_scan, gsum);
  const uint32_t tile = s_tile;
  PCC_KT(1);
  if (tile >= n_tiles) return;

  const bool with_payload = st->payload != 0;
  const bool with_payload2 = DEEP && st->payload2 != 0;
  const int shift = st->ibits + st->pass_shift[pass];
  // the digit of a key: of the key word or, in the last passes over a two-word code, of the payload (the high word)
  const bool of_high = DEEP && shift >= 63;
  const int hshift = of_high ? shift - 63 : 0;
  auto digit_of = [&](uint64_t k, uint32_t p) { return (of_high ? (p >> hshift) : (uint32_t)(k >> shift)) & mask; };
  const uint64_t* in = (pass & 1) ? buf_b : buf_a;  // ping-pong: pass 0 reads a writes b
  uint64_t* out = (pass & 1) ? out_a : out_b;
  const uint32_t* pay_in = (pass & 1) ? idx_b : idx_a;
  uint32_t* pay_out = (pass & 1) ? idx_a : idx_b;
  const uint32_t* pay2_in = (pass & 1) ? idx2_b : idx2_a;
  uint32_t* pay2_out = (pass & 1) ? idx2_a : idx2_b;
  const uint32_t n_groups_max = (n_tiles_max + kLookBackGroup - 1) / kLookBackGroup;
  uint32_t* status = status_all + ((size_t)pass * (n_tiles_max + n_groups_max)) * kMaxBins;  // one word per (tile, digit)
  uint32_t* gstatus = status + (size_t)n_tiles_max * kMaxBins;                                // one per (group, digit)
  const int lane = lane_id(), wave = wave_id();
  const uint64_t lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
  const uint32_t g = tile / kLookBackGroup, q = tile % kLookBackGroup;
  const bool closes_group = q == kLookBackGroup - 1;
  uint32_t spins = 0;

  // Tile order = (wave, round, lane): every wave owns consecutive keys, read as rows of 64.
  const uint32_t wbase = tile * kSortTile + (uint32_t)wave * (kSortTile / NW);
  uint64_t key[ITEMS];
  uint32_t pay[ITEMS], pay2[DEEP ? ITEMS : 1];
  uint16_t lrank[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
    key[r] = i < count ? in[i] : kInvalidKey;
    pay[r] = (with_payload && i < count) ? pay_in[i] : 0u;
    if (DEEP) pay2[DEEP ? r : 0] = (with_payload2 && i < count) ? pay2_in[i] : 0u;
  }
  // ---- ranking first: the tile's digit counts fall out of it (sum of the per-wave counts), so the separate counting
  //      pass over the keys that used to come first is gone; the tile's word is published right after, and the keys go
  //      into digit order in LDS BEFORE the look-back, whose round trips then have something to overlap with (and the
  //      registers of keys, payload and ranks are free by then).
  // Peers = lanes of this wave whose key has the same digit.  Every lane ORs its lane bit
  // into the (wave, digit) mask in LDS and reads the mask back: three LDS operations instead of one ballot
  // and a handful of 64-bit VALU operations per digit bit.  The first peer clears the mask again and advances
  // the wave's digit counter (LDS operations of one wave execute in program order, so no barrier is needed).
  uint64_t* wmatch = s_match + (size_t)wave * kMaxBins;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const bool valid = key[r] != kInvalidKey;
    const uint32_t d = digit_of(key[r], pay[r]);
    // the peers of the row's first key come from one ballot (the most significant digit of a clustered cloud has few
    // values, and 64 lanes ORing into one LDS word would serialise), the others through LDS
    const uint64_t vm = __ballot(valid);
    const int first = vm ? __ffsll((long long)vm) - 1 : 0;
    const uint32_t d0 = lane_of(d, first);
    const uint64_t same = __ballot(valid && d == d0);
    const bool via_lds = valid && d != d0;
    if (via_lds) atomicOr(reinterpret_cast<unsigned long long*>(&wmatch[d]), 1ull << lane);
    PCC_WAVE_LOCKSTEP();
    const uint64_t peers = via_lds ? wmatch[d] : (valid ? same : 0ull);
    const uint32_t rank = (uint32_t)__popcll(peers & lt_mask);
    const uint32_t prior = s_cnt[wave][d];
    PCC_WAVE_LOCKSTEP();
    if (valid && rank == 0) {
      if (via_lds) wmatch[d] = 0ull;
      s_cnt[wave][d] = (uint16_t)(prior + (uint32_t)__popcll(peers));
    }
    lrank[r] = (uint16_t)(prior + rank);
  }
  __syncthreads();
  PCC_KT(4);
  uint32_t run = 0;  // keys of this tile with digit d_me
  if (d_me < nbins) {  // per-wave counts -> wave start ranks inside the tile
    uint32_t sum = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const uint32_t c = s_cnt[w][d_me];
      s_cnt[w][d_me] = (uint16_t)sum;
      sum += c;
    }
    run = sum;
  }
  if (pass != 0 && d_me < nbins) publish_u32(status + (size_t)tile * kMaxBins + d_me, kStatusAggregate | run);
  uint32_t tile_valid;
  const uint32_t dstart = block_excl_scan<NW, uint32_t>(run, s_scan, tile_valid);
  if (d_me < nbins) s_dstart[d_me] = (uint16_t)dstart;
  __syncthreads();
  PCC_KT(2);
  // keys into digit order in LDS ...
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (key[r] != kInvalidKey) {
      const uint32_t d = digit_of(key[r], pay[r]);
      const uint32_t lp = (uint32_t)s_dstart[d] + s_cnt[wave][d] + lrank[r];
      s_keys[lp] = key[r];
      if (with_payload) s_pay[lp] = pay[r];
      if (DEEP && with_payload2) s_pay2[lp] = pay2[DEEP ? r : 0];
    }
  }
  // Two-level decoupled look-back over one self-describing word per (tile, digit) / (group of 16 tiles, digit).
  // To keep the polling traffic off the memory system only ONE lane per awaited tile polls (the digit-0 word)
  // until it is there; then every thread reads its own digit's words.
  uint32_t partial = 0;  // keys of this digit in the earlier tiles of the own group
  auto wait_group_mates = [&](bool also_wait_for_previous_group) {
    if (threadIdx.x < q) {
      while ((poll_u32(status + (size_t)(g * kLookBackGroup + threadIdx.x) * kMaxBins) >> 30) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit) { raise_error(st, kErrSpin); break; }
      }
    } else if (also_wait_for_previous_group && g > 0 && threadIdx.x == 64) {
      while ((poll_u32(gstatus + (size_t)(g - 1) * kMaxBins) >> 30) == 0) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > kSpinLimit) { raise_error(st, kErrSpin); break; }
      }
    }
    __syncthreads();
  };
  auto read_group_mates = [&]() {  // every thread reads its digit's words of the earlier tiles of the group
    for (;;) {
      uint32_t sum = 0;
      bool all = true;
#pragma unroll
      for (uint32_t k = 0; k < kLookBackGroup - 1; ++k) {
        if (k < q) {
          const uint32_t v = poll_u32(status + (size_t)(g * kLookBackGroup + k) * kMaxBins + d_me);
          all &= (v >> 30) != 0;
          sum += v & kStatusValue;
        }
      }
      if (all) { partial = sum; break; }
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit) { raise_error(st, kErrSpin); break; }
    }
  };
  // The last tile of a group publishes the group's count before it looks back itself: the two hops of the
  // look-back (tile counts -> group count -> the tiles that need it) then overlap.
  if (pass != 0 && closes_group) {
    wait_group_mates(false);
    if (d_me < nbins) read_group_mates();
    if (d_me < nbins) publish_u32(gstatus + (size_t)g * kMaxBins + d_me, (g == 0 ? kStatusInclusive : kStatusAggregate) | (partial + run));
  }
  PCC_KT(3);

  // ---- keys of each digit in the tiles before this one ----
  uint32_t acc = 0;
  if (pass == 0) {  // known up front (k_digit_totals)
    if (d_me < nbins) acc = tile_prefix0[(size_t)tile * kMaxBins + d_me];
  } else {
    // one barrier interval waits for the own group's earlier tiles AND the group before; then the tile words
    // and the group words are read back to back
    if (!closes_group) {
      wait_group_mates(true);
    } else {
      if (g > 0 && threadIdx.x == 64) {
        while ((poll_u32(gstatus + (size_t)(g - 1) * kMaxBins) >> 30) == 0) {
          __builtin_amdgcn_s_sleep(2);
          if (++spins > kSpinLimit) { raise_error(st, kErrSpin); break; }
        }
      }
      __syncthreads();
    }
    PCC_KT(6);
    if (d_me < nbins) {
      // the first batch of group words is requested together with the tile words: one round trip for both
      uint32_t v[kLookBackGroup];
      int j = (int)g - 1;
#pragma unroll
      for (int k = 0; k < kLookBackGroup; ++k)
        v[k] = (j - k >= 0) ? poll_u32(gstatus + (size_t)(j - k) * kMaxBins + d_me) : kStatusInclusive;
      if (!closes_group) read_group_mates();
      // the groups before: 16 per poll, ending at the first one that knows its inclusive prefix
      uint32_t before = 0;
      while (j >= 0) {
        int used = 0;
        bool done = false;
#pragma unroll
        for (int k = 0; k < kLookBackGroup; ++k) {
          if (!done && used == k) {
            const uint32_t f = v[k] >> 30;
            if (f != 0) {
              before += v[k] & kStatusValue;
              ++used;
              done = f == 2;
            }
          }
        }
        if (done) break;
        j -= used;
        if (used == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit) { raise_error(st, kErrSpin); break; }
        }
#pragma unroll
        for (int k = 0; k < kLookBackGroup; ++k)
          v[k] = (j - k >= 0) ? poll_u32(gstatus + (size_t)(j - k) * kMaxBins + d_me) : kStatusInclusive;
      }
      if (closes_group && g != 0) publish_u32(gstatus + (size_t)g * kMaxBins + d_me, kStatusInclusive | ((before + partial + run) & kStatusValue));
      acc = before + partial;
    }
  }
  if (d_me < nbins) s_gofs[d_me] = gbase + acc - dstart;
  __syncthreads();
  PCC_KT(7);
  // ... and out in runs: consecutive lanes hold consecutive keys of (mostly) the same digit
#pragma unroll
  for (int k = 0; k < ITEMS; ++k) {
    const uint32_t lp = (uint32_t)k * THREADS + threadIdx.x;
    if (lp < tile_valid) {
      const uint64_t kk = s_keys[lp];
      const uint32_t d = digit_of(kk, DEEP ? s_pay[lp] : 0u);
      const uint32_t pos = s_gofs[d] + lp;
      if (pos < out_count) {  // always true unless a look-back gave up (kErrSpin)
        out[pos] = kk;
        if (with_payload) pay_out[pos] = s_pay[lp];
        if (DEEP && with_payload2) pay2_out[pos] = s_pay2[lp];
      }
    }
  }
  PCC_KT(5);
}

// ------------------------------------------------------------------------------------------
// Stage 4: leaves.  head(i) = code(i) != code(i-1);  t(j) = index of the highest 3-bit triple in
// which leaf j differs from leaf j-1 (= number of branch nodes whose first leaf is j; t(0) = D).
// One chained scan carries both sums (leaf id, DFS byte offset: closed form of the pre-order stream,
// SURVEY.md row P5) across the tiles: word = flag(2) | sum t (32) | leaf count (30).  The tile that
// opens a piece of the DFS stream also zeroes it (B is only known on the device).
// ------------------------------------------------------------------------------------------
// ---- Morton codes of one word (trees of up to 21 levels) or two (deeper ones, pcc_device.h): the few operations the
//      leaf kernels need, so that their bodies read the same for both ----
struct Code2 {
  uint64_t lo;  // the 21 low triples
  uint32_t hi;  // the triples above
};
template <bool DEEP> struct CodeOf { using type = uint64_t; };
template <> struct CodeOf<true> { using type = Code2; };
__device__ __forceinline__ bool code_eq(uint64_t a, uint64_t b) { return a == b; }
__device__ __forceinline__ bool code_eq(Code2 a, Code2 b) { return a.lo == b.lo && a.hi == b.hi; }
__device__ __forceinline__ bool code_lt(uint64_t a, uint64_t b) { return a < b; }
__device__ __forceinline__ bool code_lt(Code2 a, Code2 b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); }
// the highest triple in which two different codes differ
__device__ __forceinline__ int code_top_triple(uint64_t a, uint64_t b) { return (63 - __clzll((long long)(a ^ b))) / 3; }
__device__ __forceinline__ int code_top_triple(Code2 a, Code2 b) {
  const uint32_t xh = a.hi ^ b.hi;
  return xh ? 21 + (31 - __clz((int)xh)) / 3 : (63 - __clzll((long long)(a.lo ^ b.lo))) / 3;
}
// the code with its v low triples cleared
__device__ __forceinline__ uint64_t code_clear_low(uint64_t c, int v) { return 3 * v >= 64 ? 0ull : ((c >> (3 * v)) << (3 * v)); }
__device__ __forceinline__ Code2 code_clear_low(Code2 c, int v) {
  if (v >= 21) { const int sh = 3 * (v - 21); return Code2{0ull, sh >= 32 ? 0u : ((c.hi >> sh) << sh)}; }
  return Code2{(c.lo >> (3 * v)) << (3 * v), c.hi};
}
__device__ __forceinline__ uint32_t code_triple(uint64_t c, int t) { return (uint32_t)(c >> (3 * t)) & 7u; }
__device__ __forceinline__ uint32_t code_triple(Code2 c, int t) { return t >= 21 ? ((c.hi >> (3 * (t - 21))) & 7u) : ((uint32_t)(c.lo >> (3 * t)) & 7u); }
__device__ __forceinline__ uint64_t code_or(uint64_t a, uint64_t b) { return a | b; }
__device__ __forceinline__ Code2 code_or(Code2 a, Code2 b) { return Code2{a.lo | b.lo, a.hi | b.hi}; }
// key bits of axis a (0 x, 1 y, 2 z)
__device__ __forceinline__ uint32_t code_axis(uint64_t c, int a) { return compact3(c >> (2 - a)); }
__device__ __forceinline__ uint32_t code_axis(Code2 c, int a) { return compact3(c.lo >> (2 - a)) | (compact3((uint64_t)c.hi >> (2 - a)) << 21); }
__device__ __forceinline__ void code_make(uint64_t& c, uint64_t lo, uint32_t) { c = lo; }
__device__ __forceinline__ void code_make(Code2& c, uint64_t lo, uint32_t hi) { c.lo = lo; c.hi = hi; }
__device__ __forceinline__ uint64_t code_low(uint64_t c) { return c; }
__device__ __forceinline__ uint64_t code_low(Code2 c) { return c.lo; }
__device__ __forceinline__ uint32_t code_high(uint64_t) { return 0u; }
__device__ __forceinline__ uint32_t code_high(Code2 c) { return c.hi; }
// Morton code of three key prefixes (the constant high key bits of a frame)
__device__ __forceinline__ void code_of_keys(uint64_t& c, const uint32_t k[3]) { c = morton3(k[0], k[1], k[2]); }
__device__ __forceinline__ void code_of_keys(Code2& c, const uint32_t k[3]) {
  c.lo = morton3(k[0] & 0x1fffffu, k[1] & 0x1fffffu, k[2] & 0x1fffffu);
  c.hi = (uint32_t)morton3(k[0] >> 21, k[1] >> 21, k[2] >> 21);
}

template <typename CodeT>
__device__ __forceinline__ uint64_t head_t(CodeT code, CodeT prev, bool is_first, int depth) {
  if (is_first) return ((uint64_t)depth << 32) | 1ull;
  if (code_eq(code, prev)) return 0ull;
  return ((uint64_t)code_top_triple(code, prev) << 32) | 1ull;
}
__device__ __forceinline__ uint32_t wave_shr1_u32(uint32_t v, uint32_t first) {  // the value of the lane below; lane 0 keeps `first`
  return (uint32_t)wave_shr1((uint64_t)v, (uint64_t)first);
}
__device__ __forceinline__ uint64_t scan_pack(uint64_t ht) { return ((ht >> 32) << 30) | (ht & 0x3fffffffull); }
__device__ __forceinline__ uint64_t scan_unpack(uint64_t w) { return (((w >> 30) & 0xffffffffull) << 32) | (w & 0x3fffffffull); }

// (two workgroup shapes, like k_sort_pass: 1024 threads x 4 keys for small grids, 512 x 8 -- two workgroups per CU, one
// looks back while the other scans -- for the rest)
template <int THREADS, int ITEMS, bool DEEP = false>
__global__ __launch_bounds__(THREADS, 4) void k_leaf_scan(const uint64_t* buf_a, const uint64_t* buf_b,
                                                            const uint32_t* __restrict__ idx_a, const uint32_t* __restrict__ idx_b,
                                                            FrameState* st, uint64_t* leaf_status, uint32_t* ticket,
                                                            uint32_t* __restrict__ leaf_start, uint64_t* __restrict__ leaf_code, uint32_t* __restrict__ leaf_hi,
                                                            uint32_t* __restrict__ leaf_base, uint8_t* __restrict__ leaf_t,
                                                            uint8_t* __restrict__ occ, unsigned long long* span) {
  using CodeT = typename CodeOf<DEEP>::type;
  const KSpan kspan(span);
  constexpr int NW = THREADS / 64;
  static_assert(THREADS * ITEMS == kSortTile, "a tile is 4096 keys");
  constexpr uint64_t kFlagAgg = 1ull << 62, kFlagIncl = 2ull << 62, kVal = (1ull << 62) - 1ull;
  __shared__ uint64_t s_w[NW];
  __shared__ uint64_t s_prefix;
  __shared__ uint32_t s_tile, s_nfin;
  // The error word is read ONCE per workgroup: tiles of this very launch store kErrSpin into it (look-back below), and lanes
  // that read it before and after such a store would disagree about leaving -- part of a workgroup at the barriers below.
  if (threadIdx.x == 0) {
    s_tile = atomicAdd(ticket, 1u);  // tile id = ticket: see k_sort_pass
    s_nfin = (poll_error(st) == kErrNone) ? st->n_finite : 0u;
  }
  __syncthreads();
  const uint32_t tile = s_tile, nfin = s_nfin;
  if ((uint64_t)tile * kSortTile >= nfin) return;
  const uint64_t* keys = (st->npasses & 1) ? buf_b : buf_a;
  const uint32_t* highs = (st->npasses & 1) ? idx_b : idx_a;  // DEEP: the codes' high words (the sort's payload)
  const int ibits = st->ibits, depth = st->depth;
  const int lane = lane_id(), wave = wave_id();
  // sorted codes whose high part is a cell rank (FrameState::code_low_bits) become Morton codes again here: nothing
  // downstream of this kernel sees a rank
  __shared__ uint64_t s_cell_abs[64];
  const int cm3 = 3 * st->code_low_bits;
  const bool ranked = st->code_low_bits < st->vbits_axis;
  if (ranked) {
    if (threadIdx.x < 64) s_cell_abs[threadIdx.x] = st->cell_abs[threadIdx.x];
    __syncthreads();
  }
  const uint64_t lowmask = ranked ? (1ull << cm3) - 1ull : ~0ull;  // (an unranked frame's cm3 can be 64 or more: no shift by it)
  auto unrank = [&](uint64_t c) { return ranked ? ((s_cell_abs[(c >> cm3) & 63u] << cm3) | (c & lowmask)) : c; };
  // Every wave owns 512 consecutive sorted keys, read as 8 rows of 64 (coalesced); element (r, lane)
  // is key wbase + 64 r + lane, so scan order is row-major inside the wave, then wave-major.
  const uint32_t wbase = tile * kSortTile + (uint32_t)wave * (kSortTile / NW);
  uint64_t ht[ITEMS], inc[ITEMS];
  CodeT code[ITEMS];
  auto code_at = [&](uint32_t i) {  // the code of sorted element i
    CodeT c;
    if (DEEP) code_make(c, keys[i], highs[i]);  // (no index bits, no cell ranks in a deep frame's keys)
    else code_make(c, unrank(keys[i] >> ibits), 0u);
    return c;
  };
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
    if (i < nfin) code[r] = code_at(i); else code_make(code[r], 0ull, 0u);
  }
  CodeT carry;  // key before the segment
  if (lane == 0 && wbase > 0 && wbase < nfin) carry = code_at(wbase - 1); else code_make(carry, 0ull, 0u);
  uint64_t wave_tot = 0;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    const uint32_t i = wbase + (uint32_t)r * 64u + (uint32_t)lane;
    CodeT prev;  // the key of the lane before; lane 0: the end of the row before
    code_make(prev, wave_shr1(code_low(code[r]), code_low(carry)), DEEP ? wave_shr1_u32(code_high(code[r]), code_high(carry)) : 0u);
    ht[r] = i < nfin ? head_t(code[r], prev, i == 0, depth) : 0ull;
    code_make(carry, lane63(code_low(code[r])), DEEP ? lane63(code_high(code[r])) : 0u);  // lane 0 of the next row compares against the end of this one
    inc[r] = wave_incl_scan_u64(ht[r]) + wave_tot;
    wave_tot = lane63(inc[r]);
  }
  if (lane == 63) s_w[wave] = wave_tot;
  __syncthreads();
  uint64_t woff = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    const uint64_t x = s_w[w];
    if (w < wave) woff += x;
    tot += x;
  }
  if (wave == 0) {  // one wave looks back, 64 earlier tiles per step
    const uint64_t mine = scan_pack(tot);
    uint64_t before = 0;
    if (tile == 0) {
      if (lane == 0) publish_u64(leaf_status, kFlagIncl | mine);
    } else {
      if (lane == 0) publish_u64(leaf_status + tile, kFlagAgg | mine);
      int j = (int)tile - 1;
      uint32_t spins = 0;
      while (j >= 0) {
        const int jj = j - lane;
        const uint64_t v = jj >= 0 ? poll_u64(leaf_status + jj) : kFlagIncl;
        const uint32_t f = (uint32_t)(v >> 62);
        const uint64_t not_ready = __ballot(f == 0), incl = __ballot(f == 2);
        const int first_nr = not_ready ? __ffsll((long long)not_ready) - 1 : 64;
        const int first_in = incl ? __ffsll((long long)incl) - 1 : 64;
        const int take = first_in < first_nr ? first_in + 1 : first_nr;  // lanes [0, take) are usable
        before += wave_sum_u64(lane < take ? (v & kVal) : 0ull);
        if (first_in < first_nr) break;
        j -= take;
        if (take == 0) {
          __builtin_amdgcn_s_sleep(1);
          if (++spins > kSpinLimit) { if (lane == 0) raise_error(st, kErrSpin); break; }
        }
      }
      if (lane == 0) publish_u64(leaf_status + tile, kFlagIncl | ((before + mine) & kVal));
    }
    if (lane == 0) s_prefix = scan_unpack(before);
  }
  __syncthreads();
  const uint64_t pre = s_prefix;
#pragma unroll
  for (int r = 0; r < ITEMS; ++r) {
    if (ht[r] & 1ull) {
      const uint64_t ex = pre + woff + inc[r] - ht[r];
      const uint32_t id = (uint32_t)(ex & 0xffffffffu);
      leaf_start[id] = wbase + (uint32_t)r * 64u + (uint32_t)lane;
      leaf_code[id] = code_low(code[r]);
      if (DEEP) leaf_hi[id] = code_high(code[r]);
      leaf_base[id] = (uint32_t)(ex >> 32);
      leaf_t[id] = (uint8_t)(ht[r] >> 32);
    }
  }
  // zero the piece of the DFS stream this tile's leaves open: bytes [b0, b1)
  const uint32_t b0 = (uint32_t)(pre >> 32), b1 = b0 + (uint32_t)(tot >> 32);
  const uint32_t a0 = min((b0 + 15u) & ~15u, b1), a1 = max(b1 & ~15u, a0);
  for (uint32_t k = b0 + threadIdx.x; k < a0; k += THREADS) occ[k] = 0;
  for (uint32_t k = a0 + threadIdx.x * 16u; k < a1; k += THREADS * 16u) *reinterpret_cast<uint4*>(occ + k) = make_uint4(0, 0, 0, 0);
  for (uint32_t k = a1 + threadIdx.x; k < b1; k += THREADS) occ[k] = 0;
  if ((uint64_t)(tile + 1) * kSortTile >= nfin && threadIdx.x == 0) {  // the last tile closes the frame
    const uint64_t all = pre + tot;
    const uint32_t L = (uint32_t)(all & 0xffffffffu);
    st->n_leaves = L;
    st->n_branches = (uint32_t)(all >> 32);
    leaf_start[L] = nfin;
    // keep the dword that holds the last stream byte clean beyond B (k_leaf_finalize ORs whole dwords)
    for (uint32_t k = b1; k < ((b1 + 3u) & ~3u); ++k) occ[k] = 0;
  }
}

// ------------------------------------------------------------------------------------------
// Stage 5: one thread per leaf: colour mean (P6), voxel centre / centroid (C2, C4), snake-mapped
// image pixel (C3b), and the leaf's contributions to the occupancy bytes (P5).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void or_byte(uint8_t* occ, uint32_t off, uint32_t bits) {
  atomicOr(reinterpret_cast<unsigned int*>(occ + (off & ~3u)), bits << (8u * (off & 3u)));
}

// SnakeGridIterator (snake.h:46-71) in closed form: linear element i -> pixel index, W multiple of 8
__device__ __forceinline__ uint32_t snake_pos(uint32_t i, uint32_t W, uint32_t H) {
  const uint32_t full = H / 8u, hl = H % 8u, per_row = W * 8u;
  if (i < full * per_row) {
    const uint32_t br = i / per_row, rem = i % per_row;
    const uint32_t bw = rem / 64u, q = rem % 64u, r = q / 8u, c = q % 8u;
    const uint32_t cc = (r & 1u) ? 7u - c : c;
    return (br * 8u + r) * W + bw * 8u + cc;
  }
  // last, partial block row: hl rows per block; with an odd hl the direction flag is not reset
  // between blocks, so every other block starts right-to-left (reference quirk, kept)
  const uint32_t rem = i - full * per_row, blk = 8u * hl;
  const uint32_t bw = rem / blk, q = rem % blk, r = q / 8u, c = q % 8u;
  const uint32_t flip = (r + ((hl & 1u) ? bw : 0u)) & 1u;
  const uint32_t cc = flip ? 7u - c : c;
  return (full * 8u + r) * W + bw * 8u + cc;
}

// original point index of sorted element i: low bits of the packed key, or the index payload
struct IndexOf {
  const uint64_t* keys;
  const uint32_t* idx;  // null in packed mode
  uint64_t imask;
  __device__ __forceinline__ uint32_t operator()(uint32_t i) const { return idx ? idx[i] : (uint32_t)(keys[i] & imask); }
};

// `colour_pay`: the sorted colour words themselves (payload 2), else they are gathered through the point index
// (`colour_keys`: the colour sits in the low 24 bits of the sorted keys)
__device__ __forceinline__ void leaf_colour(const PointView& pv, const IndexOf& index_of, const uint32_t* __restrict__ colour_pay,
                                            const uint64_t* __restrict__ colour_keys,
                                            uint32_t s, uint32_t e, uint32_t red, uint32_t& b, uint32_t& g, uint32_t& r) {
  uint32_t s0 = 0, s1 = 0, s2 = 0;
  for (uint32_t i = s; i < e; ++i) {
    const uint32_t w = colour_keys ? (uint32_t)colour_keys[i] : (colour_pay ? colour_pay[i] : load_rgba(pv, index_of(i)));
    s0 += w & 0xffu; s1 += (w >> 8) & 0xffu; s2 += (w >> 16) & 0xffu;
  }
  const uint32_t cnt = e - s;
  if (cnt > 1) { s0 /= cnt; s1 /= cnt; s2 /= cnt; }
  b = (s0 >> red) & 0xffu; g = (s1 >> red) & 0xffu; r = (s2 >> red) & 0xffu;
}

// ------------------------------------------------------------------------------------------
// JPEG front-end helpers (jpeg_io.hpp:259-314 drives libjpeg with
// JCS_RGB in, YCbCr 4:2:0, islow FDCT, quality-scaled Annex-K tables).  Everything up to the
// quantised coefficients is exact integer arithmetic and embarrassingly parallel; only the Huffman
// bit packing stays on the host.  One thread per 8x8 block, six blocks per 16x16 MCU
// (Y00 Y01 Y10 Y11 Cb Cr), coefficients written in zigzag order.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int jpeg_descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c butterfly (13-bit constants).  PASS1: even outputs shifted up by 2, odd descaled by 11;
// PASS2: even descaled by 2, odd by 15.
template <bool PASS1>
__device__ __forceinline__ void jpeg_fdct_1d(int& d0, int& d1, int& d2, int& d3, int& d4, int& d5, int& d6, int& d7) {
  const int t0 = d0 + d7, t7 = d0 - d7, t1 = d1 + d6, t6 = d1 - d6;
  const int t2 = d2 + d5, t5 = d2 - d5, t3 = d3 + d4, t4 = d3 - d4;
  const int t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  constexpr int DN = PASS1 ? 11 : 15;
  // (jfdctint's "<< PASS1_BITS", on the unsigned representation: a left shift of a negative int is undefined before C++20)
  d0 = PASS1 ? (int)((unsigned)(t10 + t11) << 2) : jpeg_descale(t10 + t11, 2);
  d4 = PASS1 ? (int)((unsigned)(t10 - t11) << 2) : jpeg_descale(t10 - t11, 2);
  int z1 = (t12 + t13) * 4433;
  d2 = jpeg_descale(z1 + t13 * 6270, DN);
  d6 = jpeg_descale(z1 + t12 * (-15137), DN);
  z1 = t4 + t7;
  int z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int z5 = (z3 + z4) * 9633;
  const int a4 = t4 * 2446, a5 = t5 * 16819, a6 = t6 * 25172, a7 = t7 * 12299;
  z1 *= -7373; z2 *= -20995;
  z3 = z3 * (-16069) + z5;
  z4 = z4 * (-3196) + z5;
  d7 = jpeg_descale(a4 + z1 + z3, DN);
  d5 = jpeg_descale(a5 + z2 + z4, DN);
  d3 = jpeg_descale(a6 + z2 + z3, DN);
  d1 = jpeg_descale(a7 + z1 + z4, DN);
}


__device__ const uint8_t kZigzagDev[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ------------------------------------------------------------------------------------------
// Stage 5: one workgroup per BLOCK ROW (8 image rows) of the snake-mapped image.  In snake order (snake.h:46-71) the
// 8 image rows of a block row are filled by ONE contiguous range of 2048 leaves (the last, partial block row holds
// 256 x (H mod 8)), so a workgroup owns <= 2048 consecutive leaves and everything they produce:
//   per leaf   colour mean (P6), voxel centre / centroid (C2, C4), simplified-cloud point, the leaf's
//              bits of the occupancy stream (P5; collected in an LDS window of the DFS stream)
//   per tile   the 8 x 256 image window, assembled in LDS and written as whole rows (k_jpeg_rows reads them)
// All global writes are contiguous runs (bgr, simplified, image rows).
// Round 1 had one 1024-thread workgroup per MCU row do this AND the JPEG stage: 118 KB of LDS, one workgroup per CU
// for 40 us.  Now two 512-thread workgroups share a CU (56 KB each), and the JPEG stage (45 KB, 768 threads, no
// per-leaf state) is a launch of its own: more boundaries for a lone frame, less CU time per frame when frames overlap.
// ------------------------------------------------------------------------------------------
constexpr int kFinThreads = 512;
constexpr int kFinRounds = 4;                          // leaves per thread
constexpr int kFinTile = kFinThreads * kFinRounds;     // 2048 leaf positions
constexpr int kFinSlots = kFinTile / 64;               // (wave, round) slots of 64 consecutive leaves
constexpr int kOccWindow = 3072;                       // dwords of the DFS stream collected in LDS (a surface needs ~300, a dense cloud ~1 800)
constexpr int kColourStage = 2560;                     // colour words staged in LDS per tile (1.25 points per leaf)

// (second launch bound: six waves per SIMD, i.e. three of these workgroups per CU -- the kernel spends two thirds of its
// time waiting for its leaf records and for the parent search, and only other workgroups can fill that)
// First probes of the parent search: distances back from the tile's first leaf that grow geometrically (ratio 2^(30/63): 64
// probes reach 2^30 leaves back).  The ancestor of height v of the tile's first leaf was opened some 4^v leaves earlier (a
// surface), so for all but the one or two highest levels the answer lies between two probes a few hundred leaves apart,
// inside leaf records the previous block rows of the same XCD have just read; evenly spaced probes (round 2's layout,
// PCC_LEAF_PROBES=uniform) send every level into a second round of 64 cold lines of its own.
__device__ const uint32_t kGeoBack[64] = {
    1u, 2u, 2u, 3u, 4u, 6u, 8u, 11u,
    15u, 20u, 28u, 38u, 53u, 74u, 102u, 142u,
    197u, 274u, 381u, 530u, 737u, 1024u, 1425u, 1982u,
    2757u, 3835u, 5334u, 7420u, 10322u, 14358u, 19973u, 27783u,
    38648u, 53762u, 74786u, 104032u, 144716u, 201309u, 280034u, 389545u,
    541882u, 753794u, 1048576u, 1458639u, 2029062u, 2822558u, 3926363u, 5461828u,
    7597761u, 10568984u, 14702150u, 20451656u, 28449595u, 39575254u, 55051774u, 76580630u,
    106528682u, 148188387u, 206139769u, 286753946u, 398893555u, 554887110u, 771884381u, 1073741824u};
__device__ __forceinline__ uint32_t geo_probe(int l, uint32_t pos0) {  // ascending in l; probe 0 is leaf 0 (pos0 < 2^30)
  const uint32_t off = kGeoBack[63 - l];
  return pos0 > off ? pos0 - off : 0u;
}

// (DEEP: trees of 22 to 31 levels, two-word codes: per-level tables for 31 levels, the codes' high words beside the low
// ones, point index or colour word from the sort's second payload)
template <bool DEEP>
__global__ __launch_bounds__(kFinThreads, 6) void k_leaf_tile(PointView pv, double res, LeafParams lp,
                                                           const uint64_t* __restrict__ buf_a, const uint64_t* __restrict__ buf_b,
                                                           const uint32_t* __restrict__ idx_a, const uint32_t* __restrict__ idx_b,
                                                           const uint32_t* __restrict__ idx2_a, const uint32_t* __restrict__ idx2_b,
                                                           const FrameState* __restrict__ st,
                                                           const uint32_t* __restrict__ leaf_start, const uint64_t* __restrict__ leaf_code,
                                                           const uint32_t* __restrict__ leaf_hi,
                                                           const uint32_t* __restrict__ leaf_base, const uint8_t* __restrict__ leaf_t,
                                                           uint8_t* __restrict__ occ, uint8_t* __restrict__ bgr, uint8_t* __restrict__ centroid,
                                                           uint8_t* __restrict__ image, float4* __restrict__ simplified, unsigned long long* span) {
  const KSpan kspan(span);
  using CodeT = typename CodeOf<DEEP>::type;
  constexpr int kDepthCap = DEEP ? kMaxDepthDeep : kMaxDepth;
  constexpr int kMaskStride = kDepthCap + 1;
  PCC_KTR(5, 0);
  const uint32_t L = st->n_leaves;
  if (L == 0 || st->error != kErrNone) return;  // after an error upstream the leaf arrays are not to be trusted
  if ((st->deep != 0) != DEEP) return;          // (cannot happen: k_boxes_events marks the frame by the sequence that was enqueued)
  const uint32_t W = 256u, H = L / 256u + 1u;  // jpegcc.h:194-198
  // Block row: image rows [8 br, 8 br + 8).  Workgroups go to the XCDs round robin (workgroup b to XCD b % 8), and a
  // block row looks at the leaf records just before its own (parent search): every XCD takes one contiguous range of
  // the block rows in use, so that those records are in its own L2 already instead of being fetched a second time.
  const uint32_t rows_used = (H + 7u) / 8u, rows_per_xcd = (rows_used + 7u) / 8u;
  const uint32_t br = lp.linear_rows ? blockIdx.x : (blockIdx.x & 7u) * rows_per_xcd + (blockIdx.x >> 3);
  if ((!lp.linear_rows && (blockIdx.x >> 3) >= rows_per_xcd) || 8u * br >= H) return;
  const uint32_t full = H / 8u, hl = H % 8u;
  const uint32_t pos0 = 2048u * br, npos = br < full ? 2048u : 256u * hl;  // leaf positions [pos0, pos0 + npos) fill this block row (br <= full here)
  const uint32_t nl = min(npos, L - pos0);                  // real leaves among them (pos0 <= L always)

  __shared__ __attribute__((aligned(16))) uint8_t s_img[8 * 768];
  __shared__ __attribute__((aligned(16))) uint8_t s_bgr[3 * kFinTile];
  // s_base | s_occ | s_mask | s_t
  __shared__ __attribute__((aligned(16))) uint32_t s_scratch[kFinTile + kOccWindow + kFinSlots * kMaskStride * 2 + kFinTile / 4];
  __shared__ uint32_t s_col[kColourStage];  // the tile's sorted colour words, loaded as one contiguous run
  __shared__ uint32_t s_pad;
  __shared__ unsigned long long s_slotbits[kDepthCap + 2];  // per level v: which slots hold a leaf with t >= v
  __shared__ uint32_t s_far[kDepthCap + 2];  // stream offset of the level-(D-v) node that was open when this tile starts
  __shared__ uint64_t s_probe[64];
  __shared__ uint32_t s_probe_hi[DEEP ? 64 : 1];
  uint32_t* s_base = s_scratch;
  uint32_t* s_occ = s_scratch + kFinTile;
  uint64_t* s_mask = reinterpret_cast<uint64_t*>(s_scratch + kFinTile + kOccWindow);
  uint8_t* s_t = reinterpret_cast<uint8_t*>(s_mask + kFinSlots * kMaskStride);

  const uint64_t* keys = st->keys_final ? buf_b : buf_a;  // (= the last pass's output)
  const int ibits = st->ibits, D = st->depth;
  const int lane = lane_id(), wave = wave_id();
  IndexOf index_of;
  index_of.keys = keys;
  const uint32_t* pay_sorted = (st->npasses & 1) ? idx_b : idx_a;
  const uint32_t* pay2_sorted = (st->npasses & 1) ? idx2_b : idx2_a;
  index_of.idx = st->payload == 1 ? pay_sorted : ((DEEP && st->payload2 == 1) ? pay2_sorted : nullptr);
  index_of.imask = (ibits >= 64) ? ~0ull : ((1ull << ibits) - 1ull);
  const uint32_t* colour_pay = st->payload == 2 ? pay_sorted : ((DEEP && st->payload2 == 2) ? pay2_sorted : nullptr);
  const uint64_t* colour_keys = st->colour_in_key ? keys : nullptr;
  auto leaf_code_at = [&](uint32_t j) { CodeT c; code_make(c, leaf_code[j], DEEP ? leaf_hi[j] : 0u); return c; };

  // ---- A1: leaf records, per-level "opens a node at level >= v" masks, LDS init ----
  if (threadIdx.x < kDepthCap + 2) s_slotbits[threadIdx.x] = 0ull;
  __syncthreads();
  PCC_KTR(7, 0);
  // first probes of the parent search below (the same for every level): requested together with the leaf records
  CodeT probe_code;
  code_make(probe_code, ~0ull, ~0u);
  const bool probes_here = wave == kFinThreads / 64 - 1 && nl && pos0 > 255u;
  if (probes_here) {
    const uint32_t step = (pos0 + 63u) / 64u, probe = lp.uniform_probes ? (uint32_t)lane * step : geo_probe(lane, pos0);
    if (probe < pos0) probe_code = leaf_code_at(probe);
  }
  // every load that does not depend on another one is requested first (leaf records of all four rounds, the ends of
  // the tile's run of points, the code the parent search starts from), then the LDS work
  int t[kFinRounds];
  uint32_t base[kFinRounds], ls[kFinRounds], le[kFinRounds];
  CodeT code[kFinRounds];
#pragma unroll
  for (int r = 0; r < kFinRounds; ++r) {
    const uint32_t lj = ((uint32_t)wave * kFinRounds + r) * 64u + (uint32_t)lane, j = pos0 + lj;
    const bool is_leaf = lj < nl;
    t[r] = is_leaf ? (int)leaf_t[j] : 0;
    base[r] = is_leaf ? leaf_base[j] : 0u;
    if (is_leaf) code[r] = leaf_code_at(j); else code_make(code[r], 0ull, 0u);
    ls[r] = is_leaf ? leaf_start[j] : 0u;
    le[r] = is_leaf ? leaf_start[j + 1] : 0u;
  }
  // the points of this tile's leaves are one contiguous run of the sorted arrays: their colour words are staged in LDS
  const uint32_t run0 = nl ? leaf_start[pos0] : 0u;
  const uint32_t run1 = nl ? leaf_start[pos0 + nl] : 0u;
  CodeT code0;
  if (nl && pos0) code0 = leaf_code_at(pos0); else code_make(code0, 0ull, 0u);
  for (int k = threadIdx.x; k < kOccWindow; k += kFinThreads) s_occ[k] = 0u;
  const bool staged = (colour_pay != nullptr || colour_keys != nullptr) && lp.do_color;
  const uint32_t ncol = staged ? min(run1 - run0, (uint32_t)kColourStage) : 0u;
  uint32_t colreg[kColourStage / kFinThreads];
#pragma unroll
  for (int q = 0; q < kColourStage / kFinThreads; ++q) {
    const uint32_t k = threadIdx.x + (uint32_t)q * kFinThreads;
    colreg[q] = k < ncol ? (colour_keys ? (uint32_t)colour_keys[run0 + k] : colour_pay[run0 + k]) : 0u;
  }
#pragma unroll
  for (int r = 0; r < kFinRounds; ++r) {
    const uint32_t slot = (uint32_t)wave * kFinRounds + r, lj = slot * 64u + (uint32_t)lane;
    const bool is_leaf = lj < nl;
    s_t[lj] = (uint8_t)t[r];
    s_base[lj] = base[r];
    // t >= v thins out quickly with v (64 consecutive leaves rarely open a node more than three or four levels up):
    // all levels are cleared with one store, then only the levels that have a leaf are visited
    if (lane <= D) s_mask[slot * kMaskStride + lane] = 0ull;
    for (int v = 1; v <= D; ++v) {
      const uint64_t mk = __ballot(is_leaf && t[r] >= v);
      if (mk == 0ull) break;
      if (lane == 0) {
        s_mask[slot * kMaskStride + v] = mk;
        atomicOr(&s_slotbits[v], 1ull << slot);
      }
    }
  }
  PCC_KTR(7, 1);
  if (probes_here) { s_probe[lane] = code_low(probe_code); if (DEEP) s_probe_hi[DEEP ? lane : 0] = code_high(probe_code); }
  __syncthreads();
  PCC_KTR(7, 2);
  // Parents that were opened before this tile: for every v the nearest earlier leaf f with t(f) >= v is
  // the first leaf of the level-(D-v) ancestor of the tile's first leaf, i.e. lower_bound over the sorted
  // leaf codes.  One wave per level.  The first round of 64 probes does not depend on the level: it was
  // requested at the top of the kernel (s_probe); the last round takes a whole range of <= 256 leaves at once,
  // four per lane, together with their stream offsets, so that a level costs two dependent round trips.
  if (nl && pos0) {
    // a leaf with t nodes of its own asks for the level-(t + 1) parent: levels beyond the tile's largest t + 1 are never
    // looked at (2048 consecutive leaves rarely open a node more than five or six levels up, so one round of waves does it)
    int vtop = 0;
    {
      const uint64_t any = __ballot(lane >= 1 && lane <= D && s_slotbits[lane] != 0ull);
      vtop = any ? 64 - __clzll((long long)any) : 1;  // largest t in the tile, + 1
    }
    for (int v = wave + 1; v <= min(D, vtop); v += kFinThreads / 64) {
      const CodeT pcode = code_clear_low(code0, v);
      // invariant: the answer (first leaf with code >= pcode) lies in [lo, hi]; leaf hi has code >= pcode
      uint32_t lo = 0, hi = pos0;
      if (pos0 > 255u && !lp.uniform_probes) {  // first round: the shared, geometrically spaced probes (s_probe)
        CodeT pc;
        code_make(pc, s_probe[lane], DEEP ? s_probe_hi[DEEP ? lane : 0] : 0u);
        const int cnt = __popcll(__ballot(geo_probe(lane, pos0) < pos0 && code_lt(pc, pcode)));  // ascending probes: a prefix of the lanes
        if (cnt == 0) {
          hi = 0u;  // probe 0 is leaf 0
        } else {
          lo = geo_probe(cnt - 1, pos0) + 1u;
          if (cnt < 64) hi = geo_probe(cnt, pos0);
        }
      }
      bool first_round = lp.uniform_probes != 0u;
      while (hi - lo > 255u) {
        const uint32_t step = (hi - lo + 63u) / 64u;
        const uint32_t probe = lo + (uint32_t)lane * step;
        CodeT pc;
        if (first_round) code_make(pc, s_probe[lane], DEEP ? s_probe_hi[DEEP ? lane : 0] : 0u);
        else if (probe < hi) pc = leaf_code_at(probe);
        else code_make(pc, ~0ull, ~0u);
        first_round = false;
        const bool less = probe < hi && code_lt(pc, pcode);
        const int cnt = __popcll(__ballot(less));  // the probes are ascending, so `less` holds for a prefix of the lanes
        if (cnt == 0) {
          hi = lo;
        } else {
          const uint32_t nhi = min(lo + (uint32_t)cnt * step, hi);
          lo = lo + (uint32_t)(cnt - 1) * step + 1u;
          hi = nhi;
        }
      }
      uint32_t below = 0, fb[4], ft[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const uint32_t i = lo + 4u * (uint32_t)lane + (uint32_t)k;
        const bool in = i <= hi;
        CodeT c;
        if (in) c = leaf_code_at(i); else code_make(c, ~0ull, ~0u);
        fb[k] = in ? leaf_base[i] : 0u;
        ft[k] = in ? (uint32_t)leaf_t[i] : 0u;
        below += (uint32_t)__popcll(__ballot(i < hi && code_lt(c, pcode)));
      }
      const uint32_t at = below;  // answer = lo + below
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (at == 4u * (uint32_t)lane + (uint32_t)k) s_far[v] = fb[k] + ft[k] - (uint32_t)v;
    }
  }
  PCC_KTR(7, 3);
#pragma unroll
  for (int q = 0; q < kColourStage / kFinThreads; ++q) {
    const uint32_t k = threadIdx.x + (uint32_t)q * kFinThreads;
    if (k < ncol) s_col[k] = colreg[q];
  }
  if (threadIdx.x == 0 && lp.write_image && nl < npos) {  // padding pixels repeat the last voxel's colour (jpegcc.h:203-213)
    uint32_t b, g, r;
    leaf_colour(pv, index_of, colour_pay, colour_keys, leaf_start[L - 1], leaf_start[L], lp.color_reduction, b, g, r);
    s_pad = b | (g << 8) | (r << 16);
  }
  __syncthreads();
  PCC_KTR(5, 1);
  const uint32_t seg0 = nl ? (s_base[0] & ~3u) : 0u;  // dword-aligned start of the tile's piece of the DFS stream
  const uint32_t seg1 = nl ? s_base[nl - 1] + s_t[nl - 1] : 0u;
  CodeT pfx;
  {
    const uint32_t pk[3] = {st->prefix[0], st->prefix[1], st->prefix[2]};
    code_of_keys(pfx, pk);
  }

  // ---- A2: per leaf ----
#pragma unroll
  for (int r = 0; r < kFinRounds; ++r) {
    const uint32_t slot = (uint32_t)wave * kFinRounds + r, lj = slot * 64u + (uint32_t)lane, j = pos0 + lj;
    if (lj >= nl) {
      if (lp.write_image && lj < npos) {
        const uint32_t px = snake_pos(j, W, H) - 8u * br * W;
        s_img[3 * px] = (uint8_t)s_pad; s_img[3 * px + 1] = (uint8_t)(s_pad >> 8); s_img[3 * px + 2] = (uint8_t)(s_pad >> 16);
      }
      continue;
    }
    const CodeT fullcode = code_or(code[r], pfx);
    const uint32_t key[3] = {code_axis(fullcode, 0), code_axis(fullcode, 1), code_axis(fullcode, 2)};
    uint32_t cb = 0, cg = 0, cr = 0;
    if (lp.do_color) {
      if (staged && le[r] - run0 <= (uint32_t)kColourStage) {  // the usual case: sum straight out of LDS
        uint32_t s0 = 0, s1 = 0, s2 = 0;
        for (uint32_t i = ls[r] - run0; i < le[r] - run0; ++i) {
          const uint32_t w = s_col[i];
          s0 += w & 0xffu; s1 += (w >> 8) & 0xffu; s2 += (w >> 16) & 0xffu;
        }
        const uint32_t cnt = le[r] - ls[r];
        if (cnt > 1) { s0 /= cnt; s1 /= cnt; s2 /= cnt; }
        cb = (s0 >> lp.color_reduction) & 0xffu; cg = (s1 >> lp.color_reduction) & 0xffu; cr = (s2 >> lp.color_reduction) & 0xffu;
      } else {
        leaf_colour(pv, index_of, colour_pay, colour_keys, ls[r], le[r], lp.color_reduction, cb, cg, cr);
      }
      s_bgr[3 * lj] = (uint8_t)cb; s_bgr[3 * lj + 1] = (uint8_t)cg; s_bgr[3 * lj + 2] = (uint8_t)cr;
      if (lp.write_image) {
        const uint32_t px = snake_pos(j, W, H) - 8u * br * W;
        s_img[3 * px] = (uint8_t)cb; s_img[3 * px + 1] = (uint8_t)cg; s_img[3 * px + 2] = (uint8_t)cr;
      }
    }
    if (r == 0) PCC_KTR(5, 2);
    // lower voxel corner (impl.hpp:1519-1521), then centre (impl.hpp:1560-1562) or centroid (:1566-1573)
    double lc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) lc[a] = __dadd_rn(__dmul_rn((double)key[a], res), st->mn[a]);
    float c[3];
    if (!lp.do_centroid) {
#pragma unroll
      for (int a = 0; a < 3; ++a)
        c[a] = lp.simplify_only ? (float)__dadd_rn(__dmul_rn(__dadd_rn((double)key[a], 0.5), res), st->mn[a])  // genLeafNodeCenterFromOctreeKey
                                : (float)__dadd_rn(lc[a], __dmul_rn(0.5, res));
    } else {
      float sx = 0.f, sy = 0.f, sz = 0.f;  // pcl::compute3DCentroid: float sums in index order
      for (uint32_t i = ls[r]; i < le[r]; ++i) {
        float x, y, z;
        load_xyz(pv, index_of(i), x, y, z);
        sx = __fadd_rn(sx, x); sy = __fadd_rn(sy, y); sz = __fadd_rn(sz, z);
      }
      const float cntf = (float)(le[r] - ls[r]);
      c[0] = __fdiv_rn(sx, cntf); c[1] = __fdiv_rn(sy, cntf); c[2] = __fdiv_rn(sz, cntf);
      const double prec = (double)0.001f;  // PointCoding default precision (ptv2.h:89-91)
      if (!lp.simplify_only) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
          int d = (int)__ddiv_rn(__dsub_rn((double)c[a], lc[a]), prec);
          d = max(-127, min(127, d));
          centroid[3 * (size_t)j + a] = (uint8_t)d;
        }
      }
    }
    if (simplified) {
      const uint32_t rgba = cb | (cg << 8) | (cr << 16) | 0xff000000u;
      simplified[j] = make_float4(c[0], c[1], c[2], __uint_as_float(rgba));
    }

    if (r == 0) PCC_KTR(5, 3);
    // occupancy: the t(j) branch nodes this leaf opens sit at base(j).. in the stream; each gets the
    // child bit on this leaf's path.  The topmost one is itself a new child of an older node: the node
    // opened by the nearest earlier leaf f with t(f) > t(j) (ballot masks inside the tile, binary
    // search over the leaf codes before it).
    if (lp.simplify_only) continue;
    const int tt = t[r];
    for (int q = 0; q < tt; ++q) {
      const int level = D - tt + q;
      const uint32_t child = code_triple(fullcode, D - 1 - level);
      const uint32_t lo = base[r] + (uint32_t)q - seg0;
      if ((lo >> 2) < (uint32_t)kOccWindow) atomicOr(&s_occ[lo >> 2], (1u << child) << (8u * (lo & 3u)));
      else or_byte(occ, base[r] + (uint32_t)q, 1u << child);
    }
    if (j > 0) {
      const int v = tt + 1;  // t < D for every leaf but the first
      const uint32_t child = code_triple(fullcode, tt);
      int fl = -1;
      uint64_t mk = s_mask[slot * kMaskStride + v] & (lane ? (~0ull >> (64 - lane)) : 0ull);
      if (mk) {
        fl = (int)slot * 64 + 63 - __clzll((long long)mk);
      } else {  // the nearest earlier slot that holds such a leaf, if any
        const uint64_t sb = s_slotbits[v] & (slot ? (~0ull >> (64 - slot)) : 0ull);
        if (sb) {
          const int sl = 63 - __clzll((long long)sb);
          mk = s_mask[sl * kMaskStride + v];
          fl = sl * 64 + 63 - __clzll((long long)mk);
        }
      }
      uint32_t off;
      if (fl >= 0) {
        off = s_base[fl] + (uint32_t)((D - tt - 1) - (D - (int)s_t[fl]));
      } else {
        off = s_far[v];  // opened before this tile
      }
      const uint32_t rel = off - seg0;
      if (off >= seg0 && (rel >> 2) < (uint32_t)kOccWindow) atomicOr(&s_occ[rel >> 2], (1u << child) << (8u * (rel & 3u)));
      else or_byte(occ, off, 1u << child);
    }
    if (r == 0) PCC_KTR(5, 4);
  }
  __syncthreads();
  PCC_KTR(5, 5);

  // ---- A3: contiguous writes: DFS stream piece, bgr, image rows ----
  if (seg1 > seg0) {
    const uint32_t ndw = min((seg1 - seg0 + 3u) >> 2, (uint32_t)kOccWindow);
    unsigned int* dst = reinterpret_cast<unsigned int*>(occ + seg0);
    for (uint32_t k = threadIdx.x; k < ndw; k += kFinThreads) {
      const uint32_t v = s_occ[k];
      if (v) atomicOr(dst + k, v);  // neighbouring tiles share the boundary dwords and set parent bits here
    }
  }
  if (lp.do_color && nl) {
    const uint32_t nbytes = 3u * nl, ndw = nbytes >> 2;
    uint8_t* dstb = bgr + 3 * (size_t)pos0;  // 3 * pos0 is a multiple of 4
    for (uint32_t k = threadIdx.x; k < ndw; k += kFinThreads) reinterpret_cast<uint32_t*>(dstb)[k] = reinterpret_cast<const uint32_t*>(s_bgr)[k];
    for (uint32_t k = 4u * ndw + threadIdx.x; k < nbytes; k += kFinThreads) dstb[k] = s_bgr[k];
  }
  if (lp.write_image && image) {
    const uint32_t rows_here = min(8u, H - 8u * br), ndw = rows_here * 768u / 4u;
    uint32_t* dsti = reinterpret_cast<uint32_t*>(image + (size_t)8u * br * 768u);
    for (uint32_t k = threadIdx.x; k < ndw; k += kFinThreads) dsti[k] = reinterpret_cast<const uint32_t*>(s_img)[k];
  }
  PCC_KTR(5, 6);
}

// ---- the range coder's symbol counts of the occupancy stream ----
// The static range coder starts with a histogram of its input (one more serial pass over ~1 MB on the host, an
// eighth of the host stage); the bytes are final once k_leaf_tile is through, so the counts ride back inside the
// FrameState for free.  Workgroup `wg` of `n_wgs`, the first `threads` threads of each (a multiple of 256 is not needed:
// any count >= 256 works).  Runs as extra workgroups of k_jpeg_rows (both only need k_leaf_tile to be over) or, for
// frames without a JPEG stage, as the launch k_occ_histogram.
template <int THREADS>
__device__ __forceinline__ void occ_histogram_block(FrameState* __restrict__ st, const uint8_t* __restrict__ occ, uint32_t wg, uint32_t n_wgs) {
  __shared__ uint32_t s_h[4][256];  // four copies: runs of equal bytes do not pile up on one LDS word
  if (st->error != kErrNone || st->n_epochs == 0) return;
  for (int k = threadIdx.x; k < 4 * 256; k += THREADS) (&s_h[0][0])[k] = 0u;
  __syncthreads();
  const uint32_t B = st->n_branches;
  const uint32_t vec = B / 16u;
  const uint4* v = reinterpret_cast<const uint4*>(occ);
  const int copy = threadIdx.x & 3;
  // Most bytes of a deep tree's stream have ONE bit set (a node with a single child; 8 values take most of the counts of a
  // sparse cloud's stream): LDS atomics on eight hot words serialise, four copies or not.  Those eight values are counted
  // in a register instead -- eight 8-bit counters in one 64-bit word, emptied into LDS before one can overflow -- and
  // only the other values go to LDS one by one.
  uint64_t hot = 0ull;   // counter k (bits 8k .. 8k+7): bytes equal to 1 << k
  uint32_t hot_n = 0u;   // bytes counted in `hot` since it was emptied (<= 255)
  auto spill_hot = [&]() {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t c = (uint32_t)(hot >> (8 * k)) & 0xffu;
      if (c) atomicAdd(&s_h[copy][1u << k], c);
    }
    hot = 0ull; hot_n = 0u;
  };
  auto count = [&](uint32_t b) {
    if (b != 0u && (b & (b - 1u)) == 0u) {
      hot += 1ull << (8 * (__ffs((int)b) - 1));
      ++hot_n;
    } else {
      atomicAdd(&s_h[copy][b], 1u);
    }
  };
  for (uint32_t i = wg * THREADS + threadIdx.x; i < vec; i += n_wgs * THREADS) {
    const uint4 q = v[i];
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    if (hot_n > 255u - 16u) spill_hot();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      count(w[k] & 0xffu); count((w[k] >> 8) & 0xffu); count((w[k] >> 16) & 0xffu); count(w[k] >> 24);
    }
  }
  spill_hot();
  if (wg == 0)  // the tail
    for (uint32_t i = vec * 16u + threadIdx.x; i < B; i += THREADS) atomicAdd(&s_h[0][occ[i]], 1u);
  __syncthreads();
  if (threadIdx.x < 256) {
    const uint32_t c = s_h[0][threadIdx.x] + s_h[1][threadIdx.x] + s_h[2][threadIdx.x] + s_h[3][threadIdx.x];
    if (c) atomicAdd(&st->occ_hist[threadIdx.x], c);
  }
}
__global__ __launch_bounds__(256) void k_occ_histogram(FrameState* __restrict__ st, const uint8_t* __restrict__ occ, unsigned long long* span) {
  const KSpan kspan(span);
  occ_histogram_block<256>(st, occ, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------
// Stage 6: one workgroup per MCU ROW (16 image rows) of the snake-mapped image: libjpeg's front end on it
// (jpeg_io.hpp:259-314: RGB->YCbCr, h2v2 downsample, islow FDCT, quantisation) -> 96 blocks of 64 zigzag-ordered
// coefficients, and their Huffman coding (jchuff.c); the host adds the file headers and stitches the rows.
// The rows come from k_leaf_tile through `image` (2.8 MB per 1 M-voxel frame, written and read once).
// ------------------------------------------------------------------------------------------
constexpr int kJpegThreads = 768;  // thread = (8x8 block, line) in the FDCT; 12 waves x 8 blocks in the Huffman stage

__global__ __launch_bounds__(kJpegThreads) void k_jpeg_rows(FrameState* st, const uint8_t* __restrict__ image,
                                                            JpegQuant jq, int16_t* __restrict__ coefs, uint32_t* __restrict__ jpeg_tiles,
                                                            const JpegHuffTables* __restrict__ huff, uint32_t n_row_wgs, const uint8_t* __restrict__ occ,
                                                            unsigned long long* span) {
  const KSpan kspan(span);
  if (blockIdx.x >= n_row_wgs) {  // the workgroups behind the MCU rows count the occupancy bytes (see occ_histogram_block)
    occ_histogram_block<kJpegThreads>(st, occ, blockIdx.x - n_row_wgs, gridDim.x - n_row_wgs);
    return;
  }
  const uint32_t L = st->n_leaves;
  if (L == 0 || st->error != kErrNone) return;
  const uint32_t H = L / 256u + 1u;  // jpegcc.h:194-198; the image is 256 pixels wide
  const uint32_t m = blockIdx.x;     // MCU row
  if (16u * m >= H) return;
  __shared__ __attribute__((aligned(16))) uint8_t s_img[16 * 768];
  __shared__ int s_ws[96 * 8 * 9];  // FDCT workspace: 96 blocks of 8 lines, 9 ints apart
  __shared__ uint32_t s_hbits[kJpegTileWords];  // header + Huffman bits of this MCU row
  __shared__ uint32_t s_hdc[2 * 12], s_hac[2 * 256];
  __shared__ uint32_t s_blen[96], s_boff[96];
  const int lane = lane_id(), wave = wave_id();
  {
    const uint32_t rows_here = min(16u, H - 16u * m), ndw = rows_here * 768u / 4u;
    const uint32_t* src = reinterpret_cast<const uint32_t*>(image + (size_t)16u * m * 768u);
    for (uint32_t k = threadIdx.x; k < ndw; k += kJpegThreads) reinterpret_cast<uint32_t*>(s_img)[k] = src[k];
  }
  if (jpeg_tiles) {  // what stage C needs from global memory travels together with the image rows
    for (int k = threadIdx.x; k < kJpegTileWords; k += kJpegThreads) s_hbits[k] = 0u;
    for (int k = threadIdx.x; k < 24; k += kJpegThreads) s_hdc[k] = (&huff->dc[0][0])[k];
    for (int k = threadIdx.x; k < 512; k += kJpegThreads) s_hac[k] = (&huff->ac[0][0])[k];
  }
  __syncthreads();
  PCC_KTR(7, 4);


  // ---- B: JPEG front end on the 16-row window: thread = (8x8 block, line) ----
  const int y_hb = (int)(H + 7u) / 8, ch = (int)(H + 1u) / 2, Hi = (int)H, row0 = 16 * (int)m;
  const int blk = threadIdx.x >> 3, line = threadIdx.x & 7;
  const int mx = blk / 6, slot6 = blk % 6;
  const bool active = true;  // 768 threads = 96 blocks x 8 lines
  bool dummy = false;
  if (active) {
    int d0, d1, d2, d3, d4, d5, d6, d7;
    int* dd[8] = {&d0, &d1, &d2, &d3, &d4, &d5, &d6, &d7};
    if (slot6 < 4) {
      const int by = 2 * (int)m + (slot6 >> 1), bx = 2 * mx + (slot6 & 1);
      dummy = by >= y_hb;  // block row below the image: the host copies the DC of the previous block
      const int y = min(8 * by + line, Hi - 1) - row0;
      // 8 pixels = 24 bytes = 6 aligned dwords of the window row
      const uint32_t* rowp = reinterpret_cast<const uint32_t*>(s_img + (dummy ? 0 : y) * 768 + 24 * bx);
      uint32_t w[6];
#pragma unroll
      for (int k = 0; k < 6; ++k) w[k] = rowp[k];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int R = (w[(3 * c) >> 2] >> (8 * ((3 * c) & 3))) & 0xff;
        const int G = (w[(3 * c + 1) >> 2] >> (8 * ((3 * c + 1) & 3))) & 0xff;
        const int B = (w[(3 * c + 2) >> 2] >> (8 * ((3 * c + 2) & 3))) & 0xff;
        *dd[c] = ((19595 * R + 38470 * G + 7471 * B + 32768) >> 16) - 128;
      }
    } else {
      const bool is_cr = slot6 == 5;
      const int crow = min(8 * (int)m + line, ch - 1);  // component rows below the image repeat the last real one
      // 16 pixels = 48 bytes = 12 aligned dwords of each of the two window rows
      const uint32_t* r0 = reinterpret_cast<const uint32_t*>(s_img + (min(2 * crow, Hi - 1) - row0) * 768 + 48 * mx);
      const uint32_t* r1 = reinterpret_cast<const uint32_t*>(s_img + (min(2 * crow + 1, Hi - 1) - row0) * 768 + 48 * mx);
      uint32_t w0[12], w1[12];
#pragma unroll
      for (int k = 0; k < 12; ++k) { w0[k] = r0[k]; w1[k] = r1[k]; }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        int sum = (c & 1) ? 2 : 1;  // jcsample.c h2v2_downsample bias 1,2,1,2
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t* ww = (k & 2) ? w1 : w0;
          const int o = 3 * (2 * c + (k & 1));
          const int R = (ww[o >> 2] >> (8 * (o & 3))) & 0xff;
          const int G = (ww[(o + 1) >> 2] >> (8 * ((o + 1) & 3))) & 0xff;
          const int B = (ww[(o + 2) >> 2] >> (8 * ((o + 2) & 3))) & 0xff;
          sum += is_cr ? ((32768 * R - 27439 * G - 5329 * B + (128 << 16) + 32767) >> 16)
                       : ((-11059 * R - 21709 * G + 32768 * B + (128 << 16) + 32767) >> 16);
        }
        *dd[c] = (sum >> 2) - 128;
      }
    }
    jpeg_fdct_1d<true>(d0, d1, d2, d3, d4, d5, d6, d7);
    int* wr = s_ws + (blk * 8 + line) * 9;
    wr[0] = d0; wr[1] = d1; wr[2] = d2; wr[3] = d3; wr[4] = d4; wr[5] = d5; wr[6] = d6; wr[7] = d7;
  }
  __syncthreads();
  if (active) {  // column `line` of the block
    int* col = s_ws + blk * 72 + line;
    int d0 = col[0], d1 = col[9], d2 = col[18], d3 = col[27], d4 = col[36], d5 = col[45], d6 = col[54], d7 = col[63];
    jpeg_fdct_1d<false>(d0, d1, d2, d3, d4, d5, d6, d7);
    const int comp = slot6 < 4 ? 0 : 1;
    const int dv[8] = {d0, d1, d2, d3, d4, d5, d6, d7};
#pragma unroll
    for (int r = 0; r < 8; ++r) {  // quantise: (|v| + 4q) / 8q with the sign restored; exact reciprocal
      const int v = dv[r], i = 8 * r + line;
      const uint32_t a = (uint32_t)(v < 0 ? -v : v) + jq.half[comp][i];
      const int qv = (int)(((uint64_t)a * jq.magic[comp][i]) >> 32);
      col[9 * r] = dummy ? 0 : (v < 0 ? -qv : qv);
    }
  }
  __syncthreads();
  int16_t* out = coefs + (size_t)m * 16 * 6 * 64;
  for (int k = threadIdx.x; k < 96 * 64; k += kJpegThreads) {
    const int b = k >> 6, nat = kZigzagDev[k & 63];
    out[k] = (int16_t)s_ws[b * 72 + (nat >> 3) * 9 + (nat & 7)];
  }
  if (!jpeg_tiles) return;

  // ---- C: Huffman coding of the 96 blocks (jchuff.c encode_one_block), one wave per block, lane = zigzag
  // index.  The row's bit string is assembled in LDS; the DC codes of the first Y, Cb and Cr block depend on
  // the MCU row before and are left to the host, which stitches the rows together.
  constexpr int kBlocksPerWave = 96 / (kJpegThreads / 64);  // 6
  uint64_t hb[kBlocksPerWave];
  uint32_t hlen[kBlocksPerWave], ho[kBlocksPerWave];
  // effective DC of a block: a dummy block below the image carries the DC of the block before its row (Y01)
  auto eff_dc = [&](int b) {
    const int bmx = b / 6, bs = b % 6;
    const bool dmy = bs >= 2 && bs < 4 && (2 * (int)m + 1) >= y_hb;
    return s_ws[(dmy ? bmx * 6 + 1 : b) * 72];
  };
#pragma unroll
  for (int i = 0; i < kBlocksPerWave; ++i) {
    const int b = wave + i * (kJpegThreads / 64);
    const int bmx = b / 6, bs = b % 6, comp = bs < 4 ? 0 : 1;
    const bool dmy = bs >= 2 && bs < 4 && (2 * (int)m + 1) >= y_hb;
    const int nat = kZigzagDev[lane];
    const int v = (lane == 0 || dmy) ? 0 : s_ws[b * 72 + (nat >> 3) * 9 + (nat & 7)];
    const uint64_t nzmask = __ballot(v != 0);
    uint64_t bits = 0;
    uint32_t len = 0;
    if (lane == 0) {
      const bool first_of_chain = bmx == 0 && (bs == 0 || bs >= 4);
      if (!first_of_chain) {
        const int pb = bs == 0 ? (bmx - 1) * 6 + 3 : (bs < 4 ? b - 1 : (bmx - 1) * 6 + bs);
        const int diff = eff_dc(b) - eff_dc(pb);
        const uint32_t a = (uint32_t)(diff < 0 ? -diff : diff);
        const uint32_t nb = a ? 32u - (uint32_t)__clz((int)a) : 0u;
        const uint32_t val = (uint32_t)(diff < 0 ? diff - 1 : diff) & ((1u << nb) - 1u);
        const uint32_t e = s_hdc[comp * 12 + nb];
        bits = ((uint64_t)(e & 0xffffu) << nb) | val;
        len = (e >> 16) + nb;
      }
    } else if (v != 0) {
      const uint64_t lower = nzmask & ((1ull << lane) - 1ull);
      const int prev = lower ? 63 - __clzll((long long)lower) : 0;
      const int run = lane - prev - 1;
      const uint32_t a = (uint32_t)(v < 0 ? -v : v);
      const uint32_t nb = 32u - (uint32_t)__clz((int)a);
      const uint32_t val = (uint32_t)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u);
      const uint32_t zrl = s_hac[comp * 256 + 0xF0];
      for (int z = 0; z < (run >> 4); ++z) { bits = (bits << (zrl >> 16)) | (zrl & 0xffffu); len += zrl >> 16; }
      const uint32_t e = s_hac[comp * 256 + (((uint32_t)run & 15u) << 4 | nb)];
      bits = (bits << (e >> 16)) | (e & 0xffffu);
      bits = (bits << nb) | val;
      len += (e >> 16) + nb;
    }
    const int last = nzmask ? 63 - __clzll((long long)nzmask) : 0;
    if (last < 63 && lane == last) {  // end of block
      const uint32_t e = s_hac[comp * 256];
      bits = (bits << (e >> 16)) | (e & 0xffffu);
      len += e >> 16;
    }
    const uint32_t incl = wave_incl_scan_u32(len);
    hb[i] = bits; hlen[i] = len; ho[i] = incl - len;
    if (lane == 63) s_blen[b] = incl;
  }
  __syncthreads();
  if (wave == 0) {  // exclusive scan of the 96 block lengths
    const uint32_t l0 = s_blen[lane], l1 = lane < 32 ? s_blen[64 + lane] : 0u;
    const uint32_t i0 = wave_incl_scan_u32(l0);
    const uint32_t t0 = lane63(i0);
    const uint32_t i1 = wave_incl_scan_u32(l1);
    s_boff[lane] = i0 - l0;
    if (lane < 32) s_boff[64 + lane] = t0 + i1 - l1;
    if (lane == 31) {
      const uint32_t total = t0 + i1;
      s_hbits[0] = total;
      s_hbits[3] = total > (uint32_t)kJpegTileBits ? 1u : 0u;
    }
  }
  __syncthreads();
  const bool fits = s_hbits[3] == 0u;
#pragma unroll
  for (int i = 0; i < kBlocksPerWave; ++i) {
    const int b = wave + i * (kJpegThreads / 64);
    uint32_t n = hlen[i];
    if (n && fits) {
      uint32_t p = s_boff[b] + ho[i];
      const uint64_t v = hb[i];
      while (n) {  // at most three words
        const uint32_t off = p & 31u, take = min(n, 32u - off);
        const uint32_t chunk = (uint32_t)(v >> (n - take)) & (take == 32u ? ~0u : ((1u << take) - 1u));
        atomicOr(&s_hbits[kJpegTileHeader + (p >> 5)], chunk << (32u - off - take));
        n -= take; p += take;
      }
    }
  }
  if (threadIdx.x == 0) {
    s_hbits[1] = s_boff[4]; s_hbits[2] = s_boff[5];
    s_hbits[4] = (uint32_t)eff_dc(0); s_hbits[5] = (uint32_t)eff_dc(4); s_hbits[6] = (uint32_t)eff_dc(5);
    s_hbits[7] = (uint32_t)eff_dc(93); s_hbits[8] = (uint32_t)eff_dc(94); s_hbits[9] = (uint32_t)eff_dc(95);
  }
  __syncthreads();
  {
    const uint32_t words = kJpegTileHeader + (fits ? (s_hbits[0] + 31u) / 32u : 0u);
    uint32_t* rec = jpeg_tiles + (size_t)m * kJpegTileWords;
    for (uint32_t k = threadIdx.x; k < words; k += kJpegThreads) rec[k] = s_hbits[k];
  }
  PCC_KTR(7, 5);
}

// ------------------------------------------------------------------------------------------
// Colour coding type 2, "lines" (jpegcc.h:244-317): the per-voxel colours are cut into strips of 2048 voxels (the last
// strip takes the remainder, 2048..4095; fewer than 2048 voxels make one strip) and every strip is a JPEG image of
// its own, w x 1 pixels.  One workgroup per strip does what libjpeg does with such an image (jpeg_io.hpp:259-314):
// the single row is repeated down the 16 rows of its MCU row (jcprepct.c), so every 8x8 block has eight equal rows and
// only the first row of its DCT is non-zero: one 1-D FDCT per block, coefficient (0,c) = 2 * row_pass[c]; the second
// luma block row of the MCU lies below the image and consists of dummy blocks (DC of the block before, no AC,
// jccoefct.c).  Huffman coding (jchuff.c) per block by one thread; the DC chain never leaves the strip.
// Output: the strip's bit string (MSB first in each u32) appended to `data` at a position taken from a cursor, and a
// directory entry {word offset, bits, width, 0}; the host adds the headers and the 0xFF stuffing.
// A block costs at most 22 + 7 * 26 + 11 + 4 = 219 bits, 1536 blocks at most: the LDS string cannot overflow.
// ------------------------------------------------------------------------------------------
constexpr int kLineThreads = 1024;
constexpr int kLineMaxBlocks = 6 * 256;

struct LineBlockCoder {  // the symbols of one block, either counted or written into the LDS bit string
  uint32_t* bits;  // null: count only
  uint32_t pos;
  __device__ __forceinline__ void put(uint32_t code, uint32_t len) {
    if (bits && len) {
      const uint32_t off = pos & 31u;
      if (off + len <= 32u) {
        atomicOr(&bits[pos >> 5], code << (32u - off - len));
      } else {
        const uint32_t second = off + len - 32u;
        atomicOr(&bits[pos >> 5], code >> second);
        atomicOr(&bits[(pos >> 5) + 1u], code << (32u - second));
      }
    }
    pos += len;
  }
};

// jchuff.c encode_one_block for a block whose only non-zero coefficients are q[0] (DC) and q[1..7] = the rest of the
// first row; zigzag positions of (0,1)..(0,7): 1, 5, 6, 14, 15, 27, 28
__device__ __forceinline__ void code_line_block(LineBlockCoder& c, int dc_diff, const int* q, bool dummy, const uint32_t* s_hdc,
                                                const uint32_t* s_hac, int comp) {
  {
    const uint32_t a = (uint32_t)(dc_diff < 0 ? -dc_diff : dc_diff);
    const uint32_t nb = a ? 32u - (uint32_t)__clz((int)a) : 0u;
    const uint32_t e = s_hdc[comp * 12 + nb];
    c.put(e & 0xffffu, e >> 16);
    if (nb) c.put((uint32_t)(dc_diff < 0 ? dc_diff - 1 : dc_diff) & ((1u << nb) - 1u), nb);
  }
  if (!dummy) {
    constexpr int kZz[8] = {0, 1, 5, 6, 14, 15, 27, 28};
    int prev = 0;
#pragma unroll
    for (int k = 1; k < 8; ++k) {
      const int v = q[k];
      if (v == 0) continue;
      int run = kZz[k] - prev - 1;
      prev = kZz[k];
      if (run >= 16) {
        const uint32_t z = s_hac[comp * 256 + 0xF0];
        c.put(z & 0xffffu, z >> 16);
        run -= 16;
      }
      const uint32_t a = (uint32_t)(v < 0 ? -v : v);
      const uint32_t nb = 32u - (uint32_t)__clz((int)a);
      const uint32_t e = s_hac[comp * 256 + ((uint32_t)run << 4 | nb)];
      c.put(e & 0xffffu, e >> 16);
      c.put((uint32_t)(v < 0 ? v - 1 : v) & ((1u << nb) - 1u), nb);
    }
  }
  const uint32_t eob = s_hac[comp * 256];  // the last coefficient of a block is never reached: always an end-of-block
  c.put(eob & 0xffffu, eob >> 16);
}

__global__ __launch_bounds__(kLineThreads) void k_jpeg_lines(FrameState* __restrict__ st, const uint8_t* __restrict__ bgr, JpegQuant jq,
                                                             const JpegHuffTables* __restrict__ huff, uint32_t* __restrict__ dir,
                                                             uint32_t* __restrict__ data, uint32_t capacity_words, unsigned long long* span) {
  const KSpan kspan(span);
  if (st->error != kErrNone || st->n_epochs == 0) return;
  const uint32_t L = st->n_leaves;
  const uint32_t lines = L / 2048u, count = lines ? lines : 1u;
  const uint32_t line = blockIdx.x;
  if (line >= count || L == 0) return;
  const uint32_t start = 2048u * line;
  const uint32_t width = lines == 0 ? L : (line + 1u != count ? 2048u : L - start);
  const uint32_t mcus = (width + 15u) / 16u, nblk = 6u * mcus;

  __shared__ __attribute__((aligned(16))) uint8_t s_px[3 * 4096 + 16];
  __shared__ int s_dc[kLineMaxBlocks];
  __shared__ uint32_t s_bits[kJpegLineWords];
  __shared__ uint32_t s_hdc[2 * 12], s_hac[2 * 256];
  __shared__ uint32_t s_scan[kLineThreads / 64];
  __shared__ uint32_t s_where;

  for (uint32_t k = threadIdx.x; k < (uint32_t)kJpegLineWords; k += kLineThreads) s_bits[k] = 0u;
  for (uint32_t k = threadIdx.x; k < 24u; k += kLineThreads) s_hdc[k] = (&huff->dc[0][0])[k];
  for (uint32_t k = threadIdx.x; k < 512u; k += kLineThreads) s_hac[k] = (&huff->ac[0][0])[k];
  {
    const uint32_t nbytes = 3u * width, ndw = nbytes >> 2;  // 3 * start is a multiple of 4
    const uint8_t* src = bgr + 3 * (size_t)start;
    for (uint32_t k = threadIdx.x; k < ndw; k += kLineThreads) reinterpret_cast<uint32_t*>(s_px)[k] = reinterpret_cast<const uint32_t*>(src)[k];
    for (uint32_t k = 4u * ndw + threadIdx.x; k < nbytes; k += kLineThreads) s_px[k] = src[k];
  }
  __syncthreads();

  // thread t owns blocks 2t and 2t+1 (MCU order: Y00 Y01 Y10 Y11 Cb Cr), so that the scan below runs over threads
  int q[2][8];
  bool live[2], dummy[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t b = 2u * threadIdx.x + (uint32_t)h;
    live[h] = b < nblk;
    const uint32_t mcu = b / 6u, slot = b % 6u;
    dummy[h] = slot == 2u || slot == 3u;
#pragma unroll
    for (int k = 0; k < 8; ++k) q[h][k] = 0;
    if (!live[h] || dummy[h]) continue;
    int d[8];
    if (slot < 2u) {
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const uint32_t x = min(16u * mcu + 8u * slot + (uint32_t)c, width - 1u);  // right edge: the last pixel repeated (jcprepct.c)
        const int R = s_px[3 * x], G = s_px[3 * x + 1], B = s_px[3 * x + 2];
        d[c] = ((19595 * R + 38470 * G + 7471 * B + 32768) >> 16) - 128;
      }
    } else {
      const bool is_cr = slot == 5u;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        int sum = (c & 1) ? 2 : 1;  // jcsample.c h2v2_downsample bias 1,2,1,2; both rows of the 2x2 box are the same row
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const uint32_t x = min(16u * mcu + 2u * (uint32_t)c + (uint32_t)k, width - 1u);
          const int R = s_px[3 * x], G = s_px[3 * x + 1], B = s_px[3 * x + 2];
          const int v = is_cr ? ((32768 * R - 27439 * G - 5329 * B + (128 << 16) + 32767) >> 16)
                              : ((-11059 * R - 21709 * G + 32768 * B + (128 << 16) + 32767) >> 16);
          sum += 2 * v;
        }
        d[c] = (sum >> 2) - 128;
      }
    }
    jpeg_fdct_1d<true>(d[0], d[1], d[2], d[3], d[4], d[5], d[6], d[7]);
    const int comp = slot < 4u ? 0 : 1;
#pragma unroll
    for (int c = 0; c < 8; ++c) {  // column pass over eight equal values: (0,c) = 2 * d[c]; then quantisation as in k_leaf_tile
      const int v = 2 * d[c];
      const uint32_t a = (uint32_t)(v < 0 ? -v : v) + jq.half[comp][c];
      const int qv = (int)(((uint64_t)a * jq.magic[comp][c]) >> 32);
      q[h][c] = v < 0 ? -qv : qv;
    }
    s_dc[b] = q[h][0];
  }
  // A strip whose width leaves its last MCU at most eight pixels has an odd number of luma blocks: the MCU's second
  // luma block lies beyond the component and is a dummy block too (jccoefct.c: DC of the block before, no AC) -- not
  // the transform of the repeated edge pixel.  Blocks 6m and 6m+1 belong to the same thread.
  if (live[1] && (2u * threadIdx.x + 1u) % 6u == 1u && 2u * ((2u * threadIdx.x + 1u) / 6u) + 1u >= (width + 7u) / 8u) {
#pragma unroll
    for (int k = 1; k < 8; ++k) q[1][k] = 0;
    q[1][0] = q[0][0];
    dummy[1] = true;
    s_dc[2u * threadIdx.x + 1u] = q[0][0];
  }
  __syncthreads();
  // DC differences along the chains Y00 Y01 Y10 Y11 | Cb | Cr (a dummy block repeats the DC of the block before it)
  int diff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const uint32_t b = 2u * threadIdx.x + (uint32_t)h;
    diff[h] = 0;
    if (!live[h]) continue;
    const uint32_t mcu = b / 6u, slot = b % 6u;
    if (slot == 0u) diff[h] = q[h][0] - (mcu ? s_dc[6u * (mcu - 1u) + 1u] : 0);
    else if (slot == 1u) diff[h] = q[h][0] - s_dc[b - 1u];
    else if (slot >= 4u) diff[h] = q[h][0] - (mcu ? s_dc[b - 6u] : 0);
  }
  uint32_t len[2] = {0u, 0u};
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!live[h]) continue;
    LineBlockCoder c{nullptr, 0u};
    code_line_block(c, diff[h], q[h], dummy[h], s_hdc, s_hac, (2u * threadIdx.x + (uint32_t)h) % 6u < 4u ? 0 : 1);
    len[h] = c.pos;
  }
  uint32_t total;
  const uint32_t at = block_excl_scan<kLineThreads / 64, uint32_t>(len[0] + len[1], s_scan, total);
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!live[h]) continue;
    LineBlockCoder c{s_bits, at + (h ? len[0] : 0u)};
    code_line_block(c, diff[h], q[h], dummy[h], s_hdc, s_hac, (2u * threadIdx.x + (uint32_t)h) % 6u < 4u ? 0 : 1);
  }
  const uint32_t nwords = (total + 31u) / 32u;
  if (threadIdx.x == 0) s_where = atomicAdd(&st->jpeg_line_words, nwords);
  __syncthreads();
  const uint32_t where = s_where;
  const bool fits = where + nwords <= capacity_words;  // always, the region holds the worst case of every strip
  if (threadIdx.x == 0) {
    dir[4u * line] = where; dir[4u * line + 1u] = fits ? total : 0u; dir[4u * line + 2u] = width; dir[4u * line + 3u] = fits ? 0u : 1u;
  }
  if (fits)
    for (uint32_t k = threadIdx.x; k < nwords; k += kLineThreads) data[where + k] = s_bits[k];
}

// ------------------------------------------------------------------------------------------
// host-side launch sequence
// ------------------------------------------------------------------------------------------
#define PCC_STAMP(name)                                      \
  do {                                                       \
    if (tm) tm->stamp(name, stream);                         \
  } while (0)

#ifdef PCC_KTIME
extern "C" int pcc_debug_read_ktime(unsigned long long* out, size_t count) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ktime), count * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

size_t sync_area_bytes(uint32_t n, int passes) {
  const size_t tiles = ((size_t)n + kSortTile - 1) / kSortTile;
  size_t b = kTicketBytes;                         // tickets: one per pass, the leaf scan's
  b += ((tiles * sizeof(uint64_t) + 15) / 16) * 16;  // leaf scan status
  const size_t groups = (tiles + kLookBackGroup - 1) / kLookBackGroup;
  b += (size_t)passes * (tiles + groups) * kMaxBins * sizeof(uint32_t);  // sort status: per tile, per group of tiles
  return b;
}

// Up to this many tiles the wide workgroup (1024 threads x 4 keys: 16 waves share a tile's latency-bound steps) is used,
// beyond it the narrow one (512 x 8: two workgroups per CU).  One frame at a time the wide shape is a little faster at 1 M
// points (176.5 against 179.8 us per frame), but with frames in flight on other streams the narrow one wins by more
// (cfg2, ten streams: 9 170 against 8 550 frames/s), and throughput is what the pipeline is for.
constexpr uint32_t kSortSmallGridTiles = 96;

void launch_hot_path(const HotPathArgs& a, hipStream_t stream, KernelTimer* tm) {
  int span_slot = 0;
  auto span = [&](const char* name) -> unsigned long long* {  // the next launch's pair of words, while there are any
    if (!a.spans || span_slot >= kMaxSpans) return nullptr;
    if (a.span_names) a.span_names->push_back(name);
    return a.spans + (size_t)2 * kSpanShards * (span_slot++);
  };
  const uint32_t n = a.n;
  const uint32_t n_tiles = (n + kTile - 1) / kTile;              // bounding-box chunks (2048 points)
  const uint32_t s_tiles = (n + kSortTile - 1) / kSortTile;      // sort / scan tiles (4096 keys)
  const int passes = a.max_passes;
  uint8_t* sync = a.sync_area;
  uint32_t* tickets = reinterpret_cast<uint32_t*>(sync);
  uint64_t* leaf_status = reinterpret_cast<uint64_t*>(sync + kTicketBytes);
  uint32_t* sort_status = reinterpret_cast<uint32_t*>(sync + kTicketBytes + (((size_t)s_tiles * sizeof(uint64_t) + 15) / 16) * 16);
  const uint32_t sync_vec16 = (uint32_t)(sync_area_bytes(n, passes) / 16);
  PCC_STAMP("begin");
  const bool deep = a.deep_launch != 0;  // the DEEP instantiations (two-word codes): a frame deeper than 21 levels came by
  static const int forced_shape = [] {  // developer knob: PCC_SORT_SHAPE=narrow|wide
    const char* e = dev_env("PCC_SORT_SHAPE");
    return !e ? 0 : (!strcmp(e, "narrow") ? 1 : (!strcmp(e, "wide") ? 2 : 0));
  }();
  const bool many_tiles = forced_shape ? forced_shape == 1 : s_tiles > kSortSmallGridTiles;
  const PlanOptions opt{a.force_pairs, a.need_index, a.no_cell_ranks, (int)a.lp.do_color, passes, deep ? 1 : 0};
  hipLaunchKernelGGL(k_boxes_events, dim3(n_tiles + 1u), dim3(kBlock), 0, stream, a.pv, n, n_tiles, a.boxes, a.frame_seq, reinterpret_cast<uint4*>(sync), sync_vec16,
                     a.res, opt, a.box, a.state, span("k_boxes_events"));
  PCC_STAMP("k_boxes_events");
  hipLaunchKernelGGL((k_make_keys<kSortThreads, kSortItems>), dim3(s_tiles), dim3(kSortThreads), 0, stream, a.pv, n, a.res, a.inv_res_pow2, a.state, a.keys_a, a.idx_a, a.idx2_a, a.hist_rows,
                     span("k_make_keys"));
  PCC_STAMP("k_make_keys");
  if (s_tiles <= 512)
    hipLaunchKernelGGL(k_digit_totals<256>, dim3((uint32_t)passes * kMaxBins / kDtCols), dim3(256), 0, stream, a.state, s_tiles, a.hist_rows, a.digit_tot, a.tile_prefix0, span("k_digit_totals"));
  else
    hipLaunchKernelGGL(k_digit_totals<1024>, dim3((uint32_t)passes * kMaxBins / kDtCols), dim3(1024), 0, stream, a.state, s_tiles, a.hist_rows, a.digit_tot, a.tile_prefix0, span("k_digit_totals"));
  PCC_STAMP("k_digit_totals");
  // Few tiles (every tile has a CU to itself): 16 waves share a tile's latency-bound steps.  Many tiles: 8 waves with
  // twice the keys per thread need 62 KB of LDS instead of 87 KB, so two tiles share a CU and one loads or waits for
  // its predecessors while the other ranks and writes.
  // One ticket counter for all workgroups of a pass: a tile only ever waits for tiles that have started.
#define PCC_SORT_ARGS a.keys_a, a.keys_b, a.keys_a, a.keys_b, a.idx_a, a.idx_b, a.idx2_a, a.idx2_b, n, pass, a.state, a.digit_tot, a.tile_prefix0, sort_status, tickets, s_tiles, span("k_sort_pass")
  for (int pass = 0; pass < passes; ++pass) {
    if (many_tiles && deep) hipLaunchKernelGGL((k_sort_pass<512, 8, true>), dim3(s_tiles), dim3(512), 0, stream, PCC_SORT_ARGS);
    else if (many_tiles) hipLaunchKernelGGL((k_sort_pass<512, 8, false>), dim3(s_tiles), dim3(512), 0, stream, PCC_SORT_ARGS);
    else if (deep) hipLaunchKernelGGL((k_sort_pass<kSortThreads, kSortItems, true>), dim3(s_tiles), dim3(kSortThreads), 0, stream, PCC_SORT_ARGS);
    else hipLaunchKernelGGL((k_sort_pass<kSortThreads, kSortItems, false>), dim3(s_tiles), dim3(kSortThreads), 0, stream, PCC_SORT_ARGS);
    PCC_STAMP("k_sort_pass");
  }
#undef PCC_SORT_ARGS
#define PCC_SCAN_ARGS a.keys_a, a.keys_b, a.idx_a, a.idx_b, a.state, leaf_status, tickets + kMaxPasses, a.leaf_start, a.leaf_code, a.leaf_hi, a.leaf_base, a.leaf_t, a.occ, span("k_leaf_scan")
  if (many_tiles && deep) hipLaunchKernelGGL((k_leaf_scan<512, 8, true>), dim3(s_tiles), dim3(512), 0, stream, PCC_SCAN_ARGS);
  else if (many_tiles) hipLaunchKernelGGL((k_leaf_scan<512, 8, false>), dim3(s_tiles), dim3(512), 0, stream, PCC_SCAN_ARGS);
  else if (deep) hipLaunchKernelGGL((k_leaf_scan<kSortThreads, kSortItems, true>), dim3(s_tiles), dim3(kSortThreads), 0, stream, PCC_SCAN_ARGS);
  else hipLaunchKernelGGL((k_leaf_scan<kSortThreads, kSortItems, false>), dim3(s_tiles), dim3(kSortThreads), 0, stream, PCC_SCAN_ARGS);
#undef PCC_SCAN_ARGS
  PCC_STAMP("k_leaf_scan");
  if (a.stop_after_leaf_scan) return;
  const uint32_t max_h = n / 256u + 1u;  // tallest possible snake image
  static const bool linear_rows = [] { const char* e = dev_env("PCC_LEAF_ROWS"); return e && !strcmp(e, "linear"); }();
  static const bool uniform_probes = [] { const char* e = dev_env("PCC_LEAF_PROBES"); return e && !strcmp(e, "uniform"); }();
  LeafParams lp = a.lp;
  lp.linear_rows = linear_rows ? 1u : 0u;
  lp.uniform_probes = uniform_probes ? 1u : 0u;
#define PCC_TILE_ARGS a.pv, a.res, lp, a.keys_a, a.keys_b, a.idx_a, a.idx_b, a.idx2_a, a.idx2_b, a.state, a.leaf_start, a.leaf_code, a.leaf_hi, a.leaf_base, a.leaf_t, \
                      a.occ, a.bgr, a.centroid, a.image, reinterpret_cast<float4*>(a.simplified), span("k_leaf_tile")
  if (deep) hipLaunchKernelGGL(k_leaf_tile<true>, dim3(((max_h + 7u) / 8u + 7u) / 8u * 8u), dim3(kFinThreads), 0, stream, PCC_TILE_ARGS);
  else hipLaunchKernelGGL(k_leaf_tile<false>, dim3(((max_h + 7u) / 8u + 7u) / 8u * 8u), dim3(kFinThreads), 0, stream, PCC_TILE_ARGS);
#undef PCC_TILE_ARGS
  PCC_STAMP("k_leaf_tile");
  // B is only known on the device: enough workgroups for the worst usual case (a few bytes per point), at least 64 x 256 threads
  const uint32_t hist_wgs = std::min(1024u, std::max(64u, (n + 16383u) / 16384u));
  bool counted = a.lp.simplify_only != 0;  // (the simplified cloud alone: no stream, nothing to count)
  if (a.lp.write_image && a.coefs && a.image) {
    const uint32_t row_wgs = (max_h + 15u) / 16u, extra = counted ? 0u : (hist_wgs + 2u) / 3u;  // 768 threads each instead of 256
    hipLaunchKernelGGL(k_jpeg_rows, dim3(row_wgs + extra), dim3(kJpegThreads), 0, stream, a.state, a.image, a.jq, a.coefs, a.jpeg_tiles, a.huff, row_wgs,
                       a.occ, span("k_jpeg_rows"));
    PCC_STAMP("k_jpeg_rows");
    counted = true;
  }
  if (a.jpeg_lines_dir) {
    const uint32_t max_lines = std::max(1u, n / 2048u);
    hipLaunchKernelGGL(k_jpeg_lines, dim3(max_lines), dim3(kLineThreads), 0, stream, a.state, a.bgr, a.jq, a.huff, a.jpeg_lines_dir, a.jpeg_lines_data,
                       a.jpeg_lines_capacity, span("k_jpeg_lines"));
    PCC_STAMP("k_jpeg_lines");
  }
  if (!counted) {
    hipLaunchKernelGGL(k_occ_histogram, dim3(hist_wgs), dim3(256), 0, stream, a.state, a.occ, span("k_occ_histogram"));
    PCC_STAMP("k_occ_histogram");
  }
}

}  // namespace pcc
