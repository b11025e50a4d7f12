// pcc_rc_device.hip -- pcl::StaticRangeCoder::encodeCharVectorToStream (the 32-bit char variant, see
// pcc_host_codec.cpp and DESIGN.md (c)) for MANY independent streams at once on the GPU.
//
// A range coder is serial per stream, but frames are independent: one WAVE codes one stream.  The coder state
// (low, range, pending output bytes) is wave-uniform, so the compiler keeps it in scalar registers and the per-symbol
// chain runs on the scalar ALU; the only per-lane work is fetching 64 symbols and their (cumulative frequency, width)
// pairs at a time -- the table lookups of the next 64 symbols are one LDS gather.  A wave is roughly ten times slower
// than a CPU core running four interleaved coders, but the chip has 1024 SIMDs with eight wave slots each: the point
// is aggregate throughput for pipelines whose host has fewer cores than the GPUs can feed.
// Bytes identical to the host coder (tests/test_rc_device.py).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>

#include "pcc_rc_device.h"

namespace pcc {
namespace {

constexpr uint32_t kTop = 1u << 24, kBottom = 1u << 16, kMaxRange = 1u << 16;

__global__ __launch_bounds__(64) void k_range_encode(const RcJob* __restrict__ jobs, uint32_t n_jobs) {
  __shared__ uint32_t s_freq[257];
  __shared__ uint32_t s_cnt[256];
  const uint32_t jb = blockIdx.x;
  if (jb >= n_jobs) return;
  const RcJob job = jobs[jb];
  const int lane = threadIdx.x;
  const uint8_t* __restrict__ in = job.in;
  const uint32_t n = job.n;

  // ---- symbol counts ----
  for (int s = lane; s < 256; s += 64) s_cnt[s] = job.hist ? job.hist[s] : 0u;
  __syncthreads();
  if (!job.hist) {
    for (uint32_t i = lane; i < n; i += 64) atomicAdd(&s_cnt[in[i]], 1u);
    __syncthreads();
  }
  // ---- cumulative table: freq[s + 1] = freq[s] + max(count, 1) ("forced strictly increasing"), halved while
  // freq[256] >= 2^16 with the same repair, exactly as the host (serial by nature, 256 entries) ----
  if (lane == 0) {
    uint32_t run = 0;
    s_freq[0] = 0;
    for (int s = 0; s < 256; ++s) {
      const uint32_t c = s_cnt[s];
      run += c ? c : 1u;
      s_freq[s + 1] = run;
    }
    while (s_freq[256] >= kMaxRange) {
      for (int f = 1; f <= 256; ++f) {
        uint32_t v = s_freq[f] / 2;
        if (v <= s_freq[f - 1]) v = s_freq[f - 1] + 1;
        s_freq[f] = v;
      }
    }
  }
  __syncthreads();
  uint8_t* __restrict__ out = job.out;
  for (int s = lane; s < 257; s += 64) reinterpret_cast<uint32_t*>(out)[s] = s_freq[s];  // the 1028-byte table
  uint8_t* __restrict__ pay = out + 1028;

  const uint32_t total = s_freq[256];
  const uint64_t magic = ~0ull / total + 1ull;  // exact floor(x / total) for 32-bit x: mulhi64(magic, x)
  const uint32_t mh = (uint32_t)(magic >> 32), ml = (uint32_t)magic;

  uint32_t low = 0, range = ~0u;
  uint64_t acc = 0;   // bytes settled but not stored yet (at most 6), most recent in the low bits
  uint32_t nacc = 0, pos = 0;
  auto put_bytes = [&](uint32_t bytes, uint32_t k) {  // k in 1..3 bytes, first one in the high end
    acc = (acc << (8 * k)) | bytes;
    nacc += k;
    if (nacc >= 4) {
      const uint32_t word = (uint32_t)(acc >> (8 * (nacc - 4)));  // oldest four
      if (lane == 0) *reinterpret_cast<uint32_t*>(pay + pos) = __builtin_bswap32(word);
      pos += 4;
      nacc -= 4;
    }
  };

  for (uint32_t i0 = 0; i0 < n; i0 += 64) {
    const uint32_t cnt = min(64u, n - i0);
    uint32_t fw = 0;  // start | width << 16 of the symbol's interval: both are below 2^16 (the table total is), one lane read per symbol
    if ((uint32_t)lane < cnt) {  // the next 64 symbols and their table entries: the data-parallel part
      const uint32_t sym = in[i0 + lane];
      const uint32_t fv = s_freq[sym];
      fw = fv | ((s_freq[sym + 1] - fv) << 16);
    }
    for (uint32_t l = 0; l < cnt; ++l) {  // the serial part: wave-uniform, scalar ALU
      const uint32_t fwl = (uint32_t)__builtin_amdgcn_readlane((int)fw, (int)l);
      const uint32_t f = fwl & 0xffffu, w = fwl >> 16;
      const uint64_t t = (uint64_t)range * mh + __umulhi(range, ml);
      const uint32_t q = (uint32_t)(t >> 32);  // range / total
      low += f * q;
      range = q * w;
      const uint32_t x = low ^ (low + range);
      const uint32_t k = (uint32_t)__builtin_clz(x | 1u) >> 3;  // settled top bytes (0..3)
      if (k) {
        put_bytes(low >> (32 - 8 * k), k);
        low <<= 8 * k;
        range <<= 8 * k;
      }
      if (range < kBottom) {  // rare: range underflow while the top byte is still open (as PCL writes it)
        range = (0u - low) & (kBottom - 1);
        for (;;) {
          put_bytes(low >> 24, 1);
          range <<= 8;
          low <<= 8;
          if ((low ^ (low + range)) >= kTop) {
            if (range >= kBottom) break;
            range = (0u - low) & (kBottom - 1);
          }
        }
      }
    }
  }
  for (int k = 0; k < 4; ++k) {  // "flush remaining data"
    put_bytes(low >> 24, 1);
    low <<= 8;
  }
  if (lane == 0) {
    for (uint32_t k = 0; k < nacc; ++k) pay[pos + k] = (uint8_t)(acc >> (8 * (nacc - 1 - k)));
    *job.out_len = 1028u + pos + nacc;
  }
}

// ------------------------------------------------------------------------------------------------------------------------
// The same coder with one LANE per stream (opt-in per context / batch / pipeline: option "rc_device_lanes").
//
// k_range_encode above keeps one stream's state in scalar registers and leaves 63 of a wave's 64 lanes to fetch symbols: a
// wave is a slow scalar core, and a chip full of them is bounded by the CUs' scalar units (measured, round 2: 109 ns per
// symbol, 2 048 waves for 2 048 streams, 16 200 streams/s).  Frames are independent, so the data-parallel axis is the
// STREAM: here the 64 lanes of a wave code 64 streams in lockstep, the coder state lives in vector registers, and one step
// of the loop settles one symbol of every stream.  What is per lane instead of per wave:
//   * the table: 256 words start | width << 16 in a lane-private LDS column (s_tab[symbol * 64 + lane]: the lanes of a wave
//     hit 64 different banks-modulo-32 whatever their symbols are -- one ds_read per symbol, no conflicts beyond the
//     wave64 pair);  the cumulative counts are built and halved in that column by the lane itself (no barrier anywhere:
//     no lane ever touches another lane's column)
//   * the input: 16 symbols per 16-byte load of the lane's own stream (the callers place streams on 64-byte boundaries, so
//     all lanes reload in the same iteration); a byte-wise tail
//   * the output: settled bytes collect in a 64-bit register per lane and leave as aligned dwords
//   * the division by the table total: the same exact multiply-high by a per-lane 64-bit reciprocal
// 2 048 streams are 32 waves instead of 2 048; the arithmetic per symbol is the same, so are the bytes
// (tests/test_rc_device.py holds both kernels against the host coder).  The symbol counts of streams that come without them
// are made first by k_stream_histograms, one workgroup per stream.
__global__ __launch_bounds__(256) void k_stream_histograms(const RcJob* __restrict__ jobs, uint32_t n_jobs, uint32_t* __restrict__ hists) {
  __shared__ uint32_t s_h[4][256];  // four copies: runs of equal bytes do not pile up on one LDS word
  const uint32_t jb = blockIdx.x;
  if (jb >= n_jobs) return;
  const RcJob job = jobs[jb];
  if (job.hist) return;
  for (int c = 0; c < 4; ++c) s_h[c][threadIdx.x] = 0u;
  __syncthreads();
  const uint8_t* __restrict__ in = job.in;
  const uint32_t n = job.n, copy = threadIdx.x & 3u;
  for (uint32_t i = threadIdx.x; i < n; i += 256u) atomicAdd(&s_h[copy][in[i]], 1u);
  __syncthreads();
  hists[(size_t)jb * 256u + threadIdx.x] = s_h[0][threadIdx.x] + s_h[1][threadIdx.x] + s_h[2][threadIdx.x] + s_h[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void k_range_encode_lanes(const RcJob* __restrict__ jobs, uint32_t n_jobs, const uint32_t* __restrict__ hists) {
  __shared__ uint32_t s_tab[256 * 64];  // [symbol][lane]: first the cumulative counts freq[symbol + 1], then start | width << 16
  const uint32_t lane = threadIdx.x;
  const uint32_t jb = blockIdx.x * 64u + lane;
  if (jb >= n_jobs) return;  // (no barrier and no cross-lane operation below: a lane is on its own)
  const RcJob job = jobs[jb];
  const uint8_t* __restrict__ in = job.in;
  const uint32_t n = job.n;
  uint32_t* col = s_tab + lane;

  // ---- cumulative table freq[s + 1] = freq[s] + max(count, 1), halved while its total is 2^16 or more (see k_range_encode) ----
  {
    const uint32_t* __restrict__ h = job.hist ? job.hist : hists + (size_t)jb * 256u;
    uint32_t run = 0;
    for (int s = 0; s < 256; ++s) {
      const uint32_t c = h[s];
      run += c ? c : 1u;
      col[s * 64] = run;
    }
    while (col[255 * 64] >= kMaxRange) {
      uint32_t prev = 0;
      for (int s = 0; s < 256; ++s) {
        uint32_t v = col[s * 64] / 2;
        if (v <= prev) v = prev + 1;
        col[s * 64] = v;
        prev = v;
      }
    }
  }
  uint8_t* __restrict__ out = job.out;
  const uint32_t total = col[255 * 64];
  {
    uint32_t* __restrict__ out32 = reinterpret_cast<uint32_t*>(out);  // the 1028-byte table, then the column becomes start | width << 16
    out32[0] = 0u;
    uint32_t prev = 0;
    for (int s = 0; s < 256; ++s) {
      const uint32_t cur = col[s * 64];
      out32[s + 1] = cur;
      col[s * 64] = prev | ((cur - prev) << 16);
      prev = cur;
    }
  }
  uint8_t* __restrict__ pay = out + 1028;
  const uint64_t magic = ~0ull / total + 1ull;  // exact floor(x / total) for 32-bit x: mulhi64(magic, x)
  const uint32_t mh = (uint32_t)(magic >> 32), ml = (uint32_t)magic;

  // The step is written without branches on per-lane conditions (64 streams in lockstep: "some lane settles a byte" and
  // "some lane has four bytes to store" are true in almost every step, so a branch would only add the mask bookkeeping):
  //   * settled bytes (0..3 per step) go into `acc` left-aligned -- the oldest pending byte is its top byte -- and the top
  //     dword is stored at pay + pos in EVERY step, complete or not; once it is complete pos moves on.  The last store of
  //     a dword is the one that counts; the trailing bytes of the stream are in memory by the same rule (the buffer has 64
  //     bytes of slack behind the longest possible stream)
  //   * shifting by 8 * k with k = 0 changes nothing, so the "byte settled" case needs no test at all
  // Only the range underflow stays a branch: it is rare per lane (a fraction of a per cent of the steps).
  uint32_t low = 0, range = ~0u;
  uint64_t acc = 0;
  uint32_t nacc = 0, pos = 0;
  auto put_bytes = [&](uint32_t bytes, uint32_t k) {  // k in 0..3 bytes (bytes < 2^(8k)), first one in the high end
    nacc += k;                                         // at most 3 + 3
    acc |= (uint64_t)bytes << ((64u - 8u * nacc) & 63u);
    *reinterpret_cast<uint32_t*>(pay + pos) = __builtin_bswap32((uint32_t)(acc >> 32));
    const bool full = nacc >= 4u;
    acc = full ? acc << 32 : acc;
    pos += full ? 4u : 0u;
    nacc -= full ? 4u : 0u;
  };
  auto step = [&](uint32_t fw) {  // fw = start | width << 16 of the symbol's interval
    const uint32_t f = fw & 0xffffu, w = fw >> 16;
    const uint64_t t = (uint64_t)range * mh + __umulhi(range, ml);
    const uint32_t q = (uint32_t)(t >> 32);  // range / total
    low += f * q;
    range = q * w;
    const uint32_t x = low ^ (low + range);
    const uint32_t s8 = ((uint32_t)__builtin_clz(x | 1u) >> 3) * 8u;  // settled top bytes (0..3), in bits
    const uint64_t sh = (uint64_t)low << s8;
    put_bytes((uint32_t)(sh >> 32), s8 >> 3);
    low = (uint32_t)sh;
    range <<= s8;
    if (range < kBottom) {  // rare: range underflow while the top byte is still open (as PCL writes it)
      range = (0u - low) & (kBottom - 1);
      for (;;) {
        put_bytes(low >> 24, 1);
        range <<= 8;
        low <<= 8;
        if ((low ^ (low + range)) >= kTop) {
          if (range >= kBottom) break;
          range = (0u - low) & (kBottom - 1);
        }
      }
    }
  };

  uint32_t i = 0;
  const bool vec = (reinterpret_cast<uintptr_t>(in) & 15u) == 0u;  // (both callers place their streams on 64-byte boundaries; anything else: byte by byte)
  for (; vec && i + 16u <= n; i += 16u) {
    const uint4 v = *reinterpret_cast<const uint4*>(in + i);
    const uint32_t wd[4] = {v.x, v.y, v.z, v.w};
    uint32_t fw[16];  // the table look-ups do not depend on the coder's state: all sixteen are in flight before the first step needs one
#pragma unroll
    for (int b = 0; b < 16; ++b) fw[b] = col[((wd[b >> 2] >> (8 * (b & 3))) & 0xffu) * 64u];
#pragma unroll
    for (int b = 0; b < 16; ++b) step(fw[b]);
  }
  for (; i < n; ++i) step(col[(uint32_t)in[i] * 64u]);
  for (int k = 0; k < 4; ++k) {  // "flush remaining data"
    put_bytes(low >> 24, 1);
    low <<= 8;
  }
  *job.out_len = 1028u + pos + nacc;
}

// coded streams side by side: stream j's out_len[j] bytes go to packed + offset[j]
__global__ __launch_bounds__(256) void k_pack_streams(const RcJob* __restrict__ jobs, const uint32_t* __restrict__ offsets, uint8_t* __restrict__ packed,
                                                      uint32_t n_jobs) {
  const uint32_t jb = blockIdx.x;
  if (jb >= n_jobs) return;
  const RcJob job = jobs[jb];
  const uint32_t len = *job.out_len;
  uint8_t* dst = packed + offsets[jb];  // 16-byte aligned, like job.out
  const uint32_t vec = len / 16u;
  for (uint32_t k = threadIdx.x; k < vec; k += 256u) reinterpret_cast<uint4*>(dst)[k] = reinterpret_cast<const uint4*>(job.out)[k];
  for (uint32_t k = vec * 16u + threadIdx.x; k < len; k += 256u) dst[k] = job.out[k];
}

}  // namespace

void launch_pack_streams(const RcJob* dev_jobs, const uint32_t* dev_offsets, uint8_t* dev_packed, uint32_t n_jobs, hipStream_t stream) {
  if (n_jobs) hipLaunchKernelGGL(k_pack_streams, dim3(n_jobs), dim3(256), 0, stream, dev_jobs, dev_offsets, dev_packed, n_jobs);
}

// dev_hists null: one wave per stream (the form that has run on the chip); else one lane per stream
void launch_range_encode(const RcJob* dev_jobs, uint32_t n_jobs, uint32_t* dev_hists, hipStream_t stream) {
  if (!n_jobs) return;
  if (dev_hists) {
    hipLaunchKernelGGL(k_stream_histograms, dim3(n_jobs), dim3(256), 0, stream, dev_jobs, n_jobs, dev_hists);
    hipLaunchKernelGGL(k_range_encode_lanes, dim3((n_jobs + 63u) / 64u), dim3(64), 0, stream, dev_jobs, n_jobs, dev_hists);
  } else {
    hipLaunchKernelGGL(k_range_encode, dim3(n_jobs), dim3(64), 0, stream, dev_jobs, n_jobs);
  }
}

}  // namespace pcc
