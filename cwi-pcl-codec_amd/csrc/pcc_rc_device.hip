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

void launch_range_encode(const RcJob* dev_jobs, uint32_t n_jobs, hipStream_t stream) {
  if (n_jobs) hipLaunchKernelGGL(k_range_encode, dim3(n_jobs), dim3(64), 0, stream, dev_jobs, n_jobs);
}

}  // namespace pcc
