// pcc_host_codec.cpp -- host stages: frame header, static range coder, baseline JPEG, decoder.
//
// These are the serial parts of the reference that the north star keeps on the CPU.  They are
// written for throughput (one pass, table driven, no per-symbol allocation) but compute exactly
// what the reference's libraries compute:
//   range coder = pcl::StaticRangeCoder (PCL 1.10 entropy_range_coder.hpp; call sites impl.hpp:1694..1798)
//   JPEG        = libjpeg-turbo as configured by jpeg_io.hpp:259-314 (encode) and :110-188 (decode)
//   header      = impl.hpp:1472-1486 + the PCL base header it calls at :1477
#include "pcc_host_codec.h"
#include "pcc_dev.h"

#include <float.h>
#include <immintrin.h>
#include <math.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <memory>
#include <system_error>
#include <thread>
#include <cstdio>
#include <cstdlib>

namespace pcc {

// =============================================================================================
// static range coder
// =============================================================================================
// The reference only ever calls the CHAR-vector variant (encodeCharVectorToStream /
// decodeStreamToCharVector).  That variant is the 32-bit coder: DWord (uint32_t) freq[257] -- hence
// the 1028-byte table -- DWord low/range, top = 1<<24, bottom = 1<<16, maxRange = 1<<16 (the
// cumulative table is halved until its total is below 2^16), one output byte = low >> 24, 4 flush
// bytes.  (The 1<<56 / 1<<48 constants belong to the INT-vector variant, which this path never uses.)
namespace {
constexpr uint32_t kTop = 1u << 24;
constexpr uint32_t kBottom = 1u << 16;
constexpr uint32_t kMaxRange = 1u << 16;

// `counts`: the histogram of `in` if somebody has it already (the GPU counts the occupancy bytes), else null
inline void cumulative_table(const uint8_t* in, size_t n, uint32_t freq[257], const uint32_t* counts = nullptr) {
  // four interleaved histograms: avoids store-to-load stalls on runs of equal bytes
  uint32_t h[4][256];
  memset(h, 0, sizeof(h));
  if (counts) {
    memcpy(h[0], counts, 256 * sizeof(uint32_t));
  } else {
    size_t i = 0;
    for (; i + 4 <= n; i += 4) {
      ++h[0][in[i]]; ++h[1][in[i + 1]]; ++h[2][in[i + 2]]; ++h[3][in[i + 3]];
    }
    for (; i < n; ++i) ++h[0][in[i]];
  }
  freq[0] = 0;
  for (int s = 0; s < 256; ++s) {
    const uint64_t c = (uint64_t)h[0][s] + h[1][s] + h[2][s] + h[3][s];
    uint32_t v = freq[s] + (uint32_t)c;
    if (v <= freq[s]) v = freq[s] + 1;  // forced strictly increasing: absent symbols get count 1
    freq[s + 1] = v;
  }
  // "rescale if numerical limits are reached"
  while (freq[256] >= kMaxRange) {
    for (int f = 1; f <= 256; ++f) {
      freq[f] /= 2;
      if (freq[f] <= freq[f - 1]) freq[f] = freq[f - 1] + 1;
    }
  }
}

// Exact floor(n / d) for every 32-bit n and 2 <= d < 2^32: q = mulhi64(ceil(2^64 / d), n).
// The coder divides by the same total for every symbol; this replaces the hardware divide.
struct InvariantDiv32 {
  uint64_t magic;
  explicit InvariantDiv32(uint32_t d) : magic(~0ull / d + 1) {}
  inline uint32_t div(uint32_t n) const { return (uint32_t)(((unsigned __int128)magic * n) >> 64); }
};
}  // namespace

namespace {
// One coder: the state of pcl::StaticRangeCoder::encodeCharVectorToStream between two symbols.
struct RcStream {
  uint32_t freq[257];
  uint64_t fw[256];     // freq[s] | (freq[s + 1] - freq[s]) << 32: one load per symbol
  uint64_t magic = 0;   // InvariantDiv32(total)
  uint32_t low = 0, range = ~0u;
  Bytes* out = nullptr;
  size_t start = 0;     // where this stream's table begins in *out
  size_t pos = 0;       // payload bytes written behind the table
  size_t cap = 0;       // payload bytes available
  const uint8_t* in = nullptr;
  size_t n = 0;

  uint8_t* payload() { return out->data() + start + sizeof(freq); }
  void begin(const uint8_t* src, size_t count, Bytes& dst, const uint32_t* counts = nullptr) {
    in = src; n = count; out = &dst;
    cumulative_table(src, count, freq, counts);
    for (int s = 0; s < 256; ++s) fw[s] = (uint64_t)freq[s] | ((uint64_t)(freq[s + 1] - freq[s]) << 32);
    magic = InvariantDiv32(freq[256]).magic;  // 256 <= total < 2^16
    start = dst.size();
    cap = count + count / 2 + 2048;  // grows on demand: a static order-0 coder cannot expand its input much
    dst.resize(start + sizeof(freq) + cap);
    memcpy(dst.data() + start, freq, sizeof(freq));
    pos = 0; low = 0; range = ~0u;
  }
  void ensure(size_t more) {  // room for `more` payload bytes
    if (pos + more > cap) {
      cap = 2 * cap + more;
      out->resize(start + sizeof(freq) + cap);
    }
  }
  size_t finish() {
    uint8_t* p = payload();
    for (int i = 0; i < 4; ++i) {  // "flush remaining data"
      p[pos++] = (uint8_t)(low >> 24);
      low <<= 8;
    }
    out->resize(start + sizeof(freq) + pos);
    return sizeof(freq) + pos;
  }
};

// One symbol of stream S (state in the scalars low##S, range##S, p##S).
// PCL's loop emits one byte per turn while the top byte is settled ((low ^ (low + range)) < 2^24).  Shifting low
// and range left by 8 shifts that XOR by 8 as well, so the number of settled bytes is the number of leading zero
// BYTES of the XOR: they are emitted in one step without a branch (4 bytes are stored, k of them count; the XOR
// cannot be 0 because range > 0).  Rare: the range underflows while the top byte is still open; then it is
// clamped to the distance to the next 2^16 boundary (range = -int(low) & (bottom - 1)) and the loop goes on as
// PCL writes it.
#define PCC_RC_STEP(S)                                                                              \
  {                                                                                                 \
    const uint64_t fw = sb[S].fw[in##S[i]];                                                         \
    const uint64_t r_ = (uint64_t)(((unsigned __int128)sb[S].magic * range##S) >> 64); /* range / total */ \
    /* start * r and size * r in ONE 64-bit multiply: fw = start | size << 32, and start * r <= (start + size) * r <= range  \
       < 2^32, so the low product does not carry into the high one (the loop of three or four streams is bound by the       \
       multiplier, not by a chain of dependent operations) */                                       \
    const uint64_t pr_ = fw * r_;                                                                   \
    low##S += (uint32_t)pr_;                                                                        \
    range##S = (uint32_t)(pr_ >> 32);                                                               \
    const uint32_t x = low##S ^ (low##S + range##S);                                                \
    const unsigned k = (unsigned)_lzcnt_u32(x) >> 3; /* x != 0: range > 0 */                           \
    const uint32_t be = __builtin_bswap32(low##S);                                                  \
    memcpy(p##S, &be, 4);                                                                           \
    p##S += k;                                                                                      \
    low##S = (uint32_t)((uint64_t)low##S << (8 * k));                                               \
    range##S = (uint32_t)((uint64_t)range##S << (8 * k));                                           \
    if (__builtin_expect(range##S < kBottom, 0)) {                                                  \
      range##S = (0u - low##S) & (kBottom - 1);                                                     \
      for (;;) {                                                                                    \
        *p##S++ = (uint8_t)(low##S >> 24);                                                          \
        range##S <<= 8;                                                                             \
        low##S <<= 8;                                                                               \
        if ((low##S ^ (low##S + range##S)) >= kTop) {                                               \
          if (range##S >= kBottom) break;                                                           \
          range##S = (0u - low##S) & (kBottom - 1);                                                 \
        }                                                                                           \
      }                                                                                             \
    }                                                                                               \
  }
#define PCC_RC_LOAD(S, st)                                                                          \
  const uint8_t* in##S = (st)->in;                                                                  \
  uint32_t low##S = (st)->low, range##S = (st)->range;                                              \
  uint8_t* p##S = (st)->payload() + (st)->pos;
#define PCC_RC_STORE(S, st)                                                                         \
  (st)->low = low##S; (st)->range = range##S; (st)->pos = (size_t)(p##S - (st)->payload());

// The same symbol with a shorter chain of dependent operations, for loops of one or two streams (whose speed is the
// latency of that chain, not the number of instructions: a lone frame, the last frames of a call).  What the next
// symbol needs first is r = range / total AFTER the renormalisation, and the renormalisation is a shift by 8 k bits
// whose k comes out of a five-operation chain (add, xor, count leading zeros, mask).  floor(magic * (range << 8k) >> 64)
// is the 128-bit product magic * range shifted left by 8k (no bit is lost: magic < 2^56, range << 8k < 2^32), so the
// multiply starts as soon as the new range exists, beside the chain that finds k, and r is two shifts and an OR
// behind k: 10 cycles per symbol instead of 13 (multiply, then k, then shift, then the 64 x 64 multiply).  The upper end
// of the new interval, low + (start + size) * r, is a third multiply next to start * r and size * r instead of an add
// behind them.
// State between symbols: low, range and r = range / total.
#ifndef PCC_RC_VARIANT
#define PCC_RC_VARIANT 1
#endif
#if PCC_RC_VARIANT == 2  // two shifts and an OR, kept apart (a double shift by CL is microcoded on some cores)
#define PCC_RC_TAIL(phi, plo, sh, r) { uint64_t tail_ = (plo) >> (63u - (sh)); __asm__("" : "+r"(tail_)); r = (uint32_t)(((phi) << (sh)) | tail_); }
#else                    // the compiler makes one double shift (shld) of it
#define PCC_RC_TAIL(phi, plo, sh, r) r = (uint32_t)(((phi) << (sh)) | ((plo) >> (63u - (sh))));
#endif
#define PCC_RC_STEP_FAST(S) PCC_RC_STEP_FAST_(S, low##S + range##S)
// (a lone stream only -- in a loop of two the extra multiply costs more than the shorter chain gains: 1.42 -> 1.51 ms per frame)
#define PCC_RC_STEP_FAST3(S) PCC_RC_STEP_FAST_(S, top3_)
#define PCC_RC_STEP_FAST_(S, TOP)                                                                         \
  {                                                                                                 \
    const uint64_t fw = sb[S].fw[in##S[i]];                                                         \
    const uint32_t top3_ = low##S + ((uint32_t)fw + (uint32_t)(fw >> 32)) * r##S; /* low + range of the new interval by a multiply of its own, beside the other two (dead code unless TOP names it) */ \
    low##S += (uint32_t)fw * r##S;                                                                  \
    range##S = (uint32_t)(fw >> 32) * r##S;                                                         \
    const unsigned __int128 prod = (unsigned __int128)sb[S].magic * range##S;                       \
    const uint64_t phi = (uint64_t)(prod >> 64), plo = (uint64_t)prod >> 1;                         \
    (void)top3_;                                                                                    \
    const uint32_t x = low##S ^ (TOP);                                                              \
    const unsigned sh = (unsigned)_lzcnt_u32(x) & 0x38u; /* 8 * settled bytes; x != 0: range > 0 */  \
    const uint32_t be = __builtin_bswap32(low##S);                                                  \
    memcpy(p##S, &be, 4);                                                                           \
    p##S += sh >> 3;                                                                                \
    low##S = (uint32_t)((uint64_t)low##S << sh);                                                    \
    range##S = (uint32_t)((uint64_t)range##S << sh);                                                \
    PCC_RC_TAIL(phi, plo, sh, r##S)                                                                 \
    if (__builtin_expect(range##S < kBottom, 0)) {                                                  \
      range##S = (0u - low##S) & (kBottom - 1);                                                     \
      for (;;) {                                                                                    \
        *p##S++ = (uint8_t)(low##S >> 24);                                                          \
        range##S <<= 8;                                                                             \
        low##S <<= 8;                                                                               \
        if ((low##S ^ (low##S + range##S)) >= kTop) {                                               \
          if (range##S >= kBottom) break;                                                           \
          range##S = (0u - low##S) & (kBottom - 1);                                                 \
        }                                                                                           \
      }                                                                                             \
      r##S = (uint32_t)(((unsigned __int128)sb[S].magic * range##S) >> 64);                         \
    }                                                                                               \
  }
#define PCC_RC_LOAD_FAST(S, st)                                                                     \
  PCC_RC_LOAD(S, st)                                                                                \
  uint32_t r##S = (uint32_t)(((unsigned __int128)(st)->magic * range##S) >> 64);

constexpr size_t kRcBlock = 256;                      // symbols between two capacity checks
constexpr size_t kRcBlockBytes = 4 * kRcBlock + 16;   // a symbol emits at most 4 bytes (+ store slack)

// symbols [i0, i1) of 1..4 streams in one loop: every symbol is a chain of dependent multiplies, shifts and a
// count-leading-zeros (about a dozen cycles), so one coder leaves most of the core idle; several independent
// coders in the same loop fill it.  The bytes of each stream do not depend on how many run together.
// `sb` = the streams of the loop, consecutive in memory: tables and reciprocals are addressed from the one base
// pointer (constant offsets), which leaves the registers to the coder states.
void rc_run1(RcStream* sb, size_t i0, size_t i1) {
  for (size_t b = i0; b < i1; b += kRcBlock) {
    const size_t e = std::min(b + kRcBlock, i1);
    sb[0].ensure(kRcBlockBytes);
#if PCC_RC_VARIANT == 0
    PCC_RC_LOAD(0, &sb[0])
    for (size_t i = b; i < e; ++i) PCC_RC_STEP(0)
#else
    PCC_RC_LOAD_FAST(0, &sb[0])
    for (size_t i = b; i < e; ++i) PCC_RC_STEP_FAST3(0)
#endif
    PCC_RC_STORE(0, &sb[0])
  }
}
void rc_run2(RcStream* sb, size_t i0, size_t i1) {
  for (size_t b = i0; b < i1; b += kRcBlock) {
    const size_t e = std::min(b + kRcBlock, i1);
    sb[0].ensure(kRcBlockBytes); sb[1].ensure(kRcBlockBytes);
#if PCC_RC_VARIANT == 0
    PCC_RC_LOAD(0, &sb[0]) PCC_RC_LOAD(1, &sb[1])
    for (size_t i = b; i < e; ++i) { PCC_RC_STEP(0) PCC_RC_STEP(1) }
#else
    PCC_RC_LOAD_FAST(0, &sb[0]) PCC_RC_LOAD_FAST(1, &sb[1])
    for (size_t i = b; i < e; ++i) { PCC_RC_STEP_FAST(0) PCC_RC_STEP_FAST(1) }
#endif
    PCC_RC_STORE(0, &sb[0]) PCC_RC_STORE(1, &sb[1])
  }
}
void rc_run3(RcStream* sb, size_t i0, size_t i1) {
  for (size_t b = i0; b < i1; b += kRcBlock) {
    const size_t e = std::min(b + kRcBlock, i1);
    sb[0].ensure(kRcBlockBytes); sb[1].ensure(kRcBlockBytes); sb[2].ensure(kRcBlockBytes);
    PCC_RC_LOAD(0, &sb[0]) PCC_RC_LOAD(1, &sb[1]) PCC_RC_LOAD(2, &sb[2])
    for (size_t i = b; i < e; ++i) { PCC_RC_STEP(0) PCC_RC_STEP(1) PCC_RC_STEP(2) }
    PCC_RC_STORE(0, &sb[0]) PCC_RC_STORE(1, &sb[1]) PCC_RC_STORE(2, &sb[2])
  }
}
void rc_run4(RcStream* sb, size_t i0, size_t i1) {
  for (size_t b = i0; b < i1; b += kRcBlock) {
    const size_t e = std::min(b + kRcBlock, i1);
    sb[0].ensure(kRcBlockBytes); sb[1].ensure(kRcBlockBytes); sb[2].ensure(kRcBlockBytes); sb[3].ensure(kRcBlockBytes);
    PCC_RC_LOAD(0, &sb[0]) PCC_RC_LOAD(1, &sb[1]) PCC_RC_LOAD(2, &sb[2]) PCC_RC_LOAD(3, &sb[3])
    for (size_t i = b; i < e; ++i) { PCC_RC_STEP(0) PCC_RC_STEP(1) PCC_RC_STEP(2) PCC_RC_STEP(3) }
    PCC_RC_STORE(0, &sb[0]) PCC_RC_STORE(1, &sb[1]) PCC_RC_STORE(2, &sb[2]) PCC_RC_STORE(3, &sb[3])
  }
}
// ---- five to sixteen streams: one stream per 32-bit lane of AVX-512 registers --------------------------------------------
// The scalar loops above stop gaining at four or five streams (measured: 2.29 ns per symbol and stream with four, 2.15 with
// six to eight: the multiplier and the store port are full).  The same arithmetic on sixteen streams at once in vector
// registers: the table entry start | width << 16 of each lane's symbol by one gather out of a 16 KB table block, range / total
// as the multiply-high by the 64-bit reciprocal done on the even and the odd lanes (32 x 32 -> 64 multiplies), the settled
// bytes of all lanes found by one vplzcntd and written by two scatters of the byte-swapped `low` (four bytes are stored, k of
// them count, as in the scalar step), the rare range underflow lane by lane in PCL's own loop.  Same bytes by construction:
// every lane computes exactly the scalar step (tests/test_host_stage.py holds all widths against each other).
// tools/ubench/rc_many.cpp: 1.35-1.40 ns per symbol and stream on the build container's Xeon (AVX-512 on two ports) against
// 2.29 for four scalar streams; not yet timed on the GPU box's EPYC.
struct alignas(64) RcWide {
  uint32_t tab[16 * 256];
  uint32_t ml[16], mh[16], low[16], range[16];
  uint64_t inp[16], outp[16];
};
#define PCC_T512 __attribute__((target("avx512f,avx512cd,avx512bw,avx512dq,avx512vl")))
// symbols [i0, i1) of the lanes in `live`; (i1 - i0) % 4 == 0 and every live lane's stream holds i1 symbols at least
PCC_T512 void rc_run_wide(RcWide& L, __mmask16 live, size_t i0, size_t i1) {
  const __mmask8 live_lo = (__mmask8)live, live_hi = (__mmask8)(live >> 8);
  const __m512i lane256 = _mm512_mullo_epi32(_mm512_set_epi32(15, 14, 13, 12, 11, 10, 9, 8, 7, 6, 5, 4, 3, 2, 1, 0), _mm512_set1_epi32(256));
  const __m512i ML = _mm512_load_si512(L.ml), MH = _mm512_load_si512(L.mh);
  const __m512i MLo = _mm512_srli_epi64(ML, 32), MHo = _mm512_srli_epi64(MH, 32);
  const __m512i himask = _mm512_set1_epi64((long long)0xffffffff00000000ull);
  const __m512i bsw = _mm512_set4_epi32(0x0c0d0e0f, 0x08090a0b, 0x04050607, 0x00010203);
  const __m512i kbot = _mm512_set1_epi32((int)kBottom);
  __m512i low = _mm512_load_si512(L.low), range = _mm512_load_si512(L.range);
  const __m512i in0 = _mm512_load_si512(L.inp), in1 = _mm512_load_si512(L.inp + 8);
  __m512i p0 = _mm512_load_si512(L.outp), p1 = _mm512_load_si512(L.outp + 8);
  for (size_t i = i0; i < i1; i += 4) {
    const __m512i off = _mm512_set1_epi64((long long)i);
    const __m256i w0 = _mm512_mask_i64gather_epi32(_mm256_setzero_si256(), live_lo, _mm512_add_epi64(in0, off), nullptr, 1);
    const __m256i w1 = _mm512_mask_i64gather_epi32(_mm256_setzero_si256(), live_hi, _mm512_add_epi64(in1, off), nullptr, 1);
    __m512i word = _mm512_inserti64x4(_mm512_castsi256_si512(w0), w1, 1);  // four symbols of every lane
    for (int j = 0; j < 4; ++j) {
      const __m512i sym = _mm512_and_si512(word, _mm512_set1_epi32(0xff));
      word = _mm512_srli_epi32(word, 8);
      const __m512i fw = _mm512_i32gather_epi32(_mm512_add_epi32(sym, lane256), L.tab, 4);
      const __m512i f = _mm512_and_si512(fw, _mm512_set1_epi32(0xffff)), w = _mm512_srli_epi32(fw, 16);
      // q = range / total = mulhi64(magic, range): even lanes and odd lanes apart, the quotient of an odd lane lands in place
      const __m512i ro = _mm512_srli_epi64(range, 32);
      const __m512i ae = _mm512_srli_epi64(_mm512_mul_epu32(range, ML), 32), ao = _mm512_srli_epi64(_mm512_mul_epu32(ro, MLo), 32);
      const __m512i be = _mm512_add_epi64(_mm512_mul_epu32(range, MH), ae), bo = _mm512_add_epi64(_mm512_mul_epu32(ro, MHo), ao);
      const __m512i q = _mm512_or_si512(_mm512_srli_epi64(be, 32), _mm512_and_si512(bo, himask));
      low = _mm512_add_epi32(low, _mm512_mullo_epi32(f, q));
      range = _mm512_mullo_epi32(q, w);
      const __m512i x = _mm512_xor_si512(low, _mm512_add_epi32(low, range));
      const __m512i lz = _mm512_lzcnt_epi32(x);  // x != 0 because range != 0
      const __m512i k8 = _mm512_and_si512(lz, _mm512_set1_epi32(0x18)), k = _mm512_srli_epi32(lz, 3);
      const __m512i bytes = _mm512_shuffle_epi8(low, bsw);
      _mm512_mask_i64scatter_epi32(nullptr, live_lo, p0, _mm512_castsi512_si256(bytes), 1);
      _mm512_mask_i64scatter_epi32(nullptr, live_hi, p1, _mm512_extracti64x4_epi64(bytes, 1), 1);
      p0 = _mm512_add_epi64(p0, _mm512_cvtepu32_epi64(_mm512_castsi512_si256(k)));
      p1 = _mm512_add_epi64(p1, _mm512_cvtepu32_epi64(_mm512_extracti64x4_epi64(k, 1)));
      low = _mm512_sllv_epi32(low, k8);
      range = _mm512_sllv_epi32(range, k8);
      const __mmask16 under = _mm512_mask_cmplt_epu32_mask(live, range, kbot);
      if (__builtin_expect(under != 0, 0)) {  // rare: the lanes concerned go through PCL's loop one by one
        alignas(64) uint32_t lo_[16], ra_[16];
        alignas(64) uint64_t pp[16];
        _mm512_store_si512(lo_, low); _mm512_store_si512(ra_, range); _mm512_store_si512(pp, p0); _mm512_store_si512(pp + 8, p1);
        for (unsigned m = under; m; m &= m - 1) {
          const int l = __builtin_ctz(m);
          uint32_t lw = lo_[l], rg = (0u - lo_[l]) & (kBottom - 1);
          uint8_t* p = reinterpret_cast<uint8_t*>(pp[l]);
          for (;;) {
            *p++ = (uint8_t)(lw >> 24);
            rg <<= 8;
            lw <<= 8;
            if ((lw ^ (lw + rg)) >= kTop) {
              if (rg >= kBottom) break;
              rg = (0u - lw) & (kBottom - 1);
            }
          }
          lo_[l] = lw; ra_[l] = rg; pp[l] = reinterpret_cast<uint64_t>(p);
        }
        low = _mm512_load_si512(lo_); range = _mm512_load_si512(ra_); p0 = _mm512_load_si512(pp); p1 = _mm512_load_si512(pp + 8);
      }
    }
  }
  _mm512_store_si512(L.low, low); _mm512_store_si512(L.range, range); _mm512_store_si512(L.outp, p0); _mm512_store_si512(L.outp + 8, p1);
}
// symbols [i0, i1) of streams sb[0 .. live): blocks of kRcBlock through the vector loop, what is left (fewer than four symbols
// per stream) through the scalar one
void rc_run_many_wide(RcStream* sb, int live, size_t i0, size_t i1, RcWide& L) {
  const __mmask16 mask = (__mmask16)((1u << live) - 1u);
  for (int l = 0; l < 16; ++l) {
    const RcStream& s = sb[l < live ? l : 0];
    for (int y = 0; y < 256; ++y) L.tab[l * 256 + y] = (uint32_t)s.fw[y] | ((uint32_t)(s.fw[y] >> 32) << 16);  // start, width < 2^16
    L.ml[l] = (uint32_t)s.magic; L.mh[l] = (uint32_t)(s.magic >> 32);
    L.inp[l] = reinterpret_cast<uint64_t>(s.in);
  }
  const size_t vec_end = i0 + ((i1 - i0) & ~(size_t)3);
  for (size_t b = i0; b < vec_end; b += kRcBlock) {
    const size_t e = std::min(b + kRcBlock, vec_end);
    for (int l = 0; l < 16; ++l) {
      RcStream& s = sb[l < live ? l : 0];
      if (l < live) s.ensure(kRcBlockBytes);
      L.low[l] = s.low; L.range[l] = l < live ? s.range : ~0u;   // (a lane that is switched off keeps a harmless state)
      L.outp[l] = reinterpret_cast<uint64_t>(s.payload() + s.pos);
    }
    rc_run_wide(L, mask, b, e);
    for (int l = 0; l < live; ++l) {
      sb[l].low = L.low[l]; sb[l].range = L.range[l];
      sb[l].pos = (size_t)(reinterpret_cast<uint8_t*>(L.outp[l]) - sb[l].payload());
    }
  }
  for (int l = 0; l < live && vec_end < i1; ++l) rc_run1(sb + l, vec_end, i1);
}
}  // namespace

bool StaticRangeCoder::wide_available() {
  static const bool ok = [] {
    const char* e = dev_env("PCC_RC_WIDE");
    if (e && e[0] == '0') return false;
    return __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512cd") && __builtin_cpu_supports("avx512bw") &&
           __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512vl");
  }();
  return ok;
}

void StaticRangeCoder::encode_many(int count, const uint8_t* const in[], const size_t n[], Bytes* const out[], size_t got[],
                                   const uint32_t* const counts[]) {
  RcStream st[kMaxStreams];
  if (count > kMaxStreams) count = kMaxStreams;
  // longest first: all streams run together up to the length of the shortest, then one fewer, ...
  int idx[kMaxStreams];
  for (int i = 0; i < count; ++i) idx[i] = i;
  std::stable_sort(idx, idx + count, [&](int x, int y) { return n[x] > n[y]; });
  for (int k = 0; k < count; ++k) st[k].begin(in[idx[k]], n[idx[k]], *out[idx[k]], counts ? counts[idx[k]] : nullptr);
  size_t done = 0;
  std::unique_ptr<RcWide> wide;
  for (int live = count; live > 0; --live) {
    const size_t upto = st[live - 1].n;  // the shortest stream still running ends here
    if (upto > done) {
      if (live > kInterleave) {
        // (a vector step costs the same however many of its sixteen lanes are in use: below ten streams two or three scalar
        //  loops of four are the faster way)
        if (live >= 10 && wide_available()) {
          if (!wide) wide.reset(new RcWide());
          rc_run_many_wide(st, live, done, upto, *wide);
        } else {  // no AVX-512 here: scalar loops of four streams (and what is left), one group after the other
          for (int g = 0; g < live; g += kInterleave) {
            const int m = std::min(kInterleave, live - g);
            if (m == 4) rc_run4(st + g, done, upto);
            else if (m == 3) rc_run3(st + g, done, upto);
            else if (m == 2) rc_run2(st + g, done, upto);
            else rc_run1(st + g, done, upto);
          }
        }
      } else if (live == 4) rc_run4(st, done, upto);
      else if (live == 3) rc_run3(st, done, upto);
      else if (live == 2) rc_run2(st, done, upto);
      else rc_run1(st, done, upto);
      done = upto;
    }
  }
  for (int k = 0; k < count; ++k) {
    st[k].ensure(8);
    got[idx[k]] = st[k].finish();
  }
}

size_t StaticRangeCoder::encode(const uint8_t* in, size_t n, Bytes& out) {
  const uint8_t* src[1] = {in};
  const size_t len[1] = {n};
  Bytes* dst[1] = {&out};
  size_t got[1];
  encode_many(1, src, len, dst, got);
  return got[0];
}

void StaticRangeCoder::encode2(const uint8_t* in_a, size_t n_a, Bytes& out_a, size_t& len_a,
                               const uint8_t* in_b, size_t n_b, Bytes& out_b, size_t& len_b) {
  const uint8_t* src[2] = {in_a, in_b};
  const size_t len[2] = {n_a, n_b};
  Bytes* dst[2] = {&out_a, &out_b};
  size_t got[2];
  encode_many(2, src, len, dst, got);
  len_a = got[0];
  len_b = got[1];
}

size_t StaticRangeCoder::decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t n) {
  uint32_t freq[257];
  if (in_len < sizeof(freq) + 4) return 0;
  memcpy(freq, in, sizeof(freq));
  size_t pos = sizeof(freq);
  uint32_t code = 0, low = 0, range = ~0u;
  for (int i = 0; i < 4; ++i) code = (code << 8) | in[pos++];
  const uint32_t total = freq[256];
  if (total == 0) return 0;
  // PCL finds the symbol of a cumulative count by an eight-step binary search; each step is an unpredictable branch,
  // most of the decoder's time.  The encoder keeps the total below 2^16 (it halves the table otherwise), so a table of
  // 4096 entries -- the symbol that holds count 16 b, for every b -- followed by a walk up the cumulative table answers
  // instead: symbols wider than 16 counts (all the frequent ones) need no step of the walk unless the count falls into the
  // bucket of their first count, and table and cumulative counts stay in the first-level cache (a 64 K table of one
  // entry per count does not: that load was a second-level hit on the path of every symbol).  Any other table (foreign or
  // corrupt stream) takes the search.  The table must be non-decreasing for the two to agree: checked.
  bool monotone = true;
  for (int k = 0; k < 256; ++k) monotone &= freq[k] <= freq[k + 1];
  constexpr unsigned kBucketShift = 4;
  uint8_t first_of_bucket[65536u >> kBucketShift];
  bool have_table = false;
  if (monotone && freq[0] == 0 && total <= 65536u && n >= 256) {
    unsigned sy = 0;
    for (uint32_t b = 0; (b << kBucketShift) < total; ++b) {
      const uint32_t c = b << kBucketShift;
      while (freq[sy + 1] <= c) ++sy;  // c < total = freq[256]: ends at sy <= 255
      first_of_bucket[b] = (uint8_t)sy;
    }
    have_table = true;
  }
  const uint8_t* const table = have_table ? first_of_bucket : nullptr;
  const InvariantDiv32 by_total(total >= 2 ? total : 2);  // the same divisor for every symbol: a multiply instead of a divide
  for (size_t i = 0; i < n; ++i) {
    range = total >= 2 ? by_total.div(range) : range / total;
    if (range == 0) return 0;  // corrupt table: PCL would divide by zero here
    const uint32_t count = (code - low) / range;
    unsigned sym = 0;
    if (table && count < total) {
      sym = table[count >> kBucketShift];
      while (freq[sym + 1] <= count) ++sym;  // the largest symbol whose cumulative count does not exceed `count`, as the search finds it
    } else {
      for (unsigned step = 128; step; step >>= 1)
        if (freq[sym + step] <= count) sym += step;
    }
    out[i] = (uint8_t)sym;
    low += freq[sym] * range;
    range *= freq[sym + 1] - freq[sym];
    if (range == 0) return 0;  // a symbol of width zero: corrupt table or stream (PCL's loop would never end)
    // PCL's loop takes one byte per turn while the top byte of the interval is settled; as in the encoder, the number of
    // settled bytes is the number of leading zero BYTES of low ^ (low + range): they are taken in one step without a branch
    // (the loop's trip count is as good as random, a mispredicted branch per symbol otherwise).  Near the end of the input,
    // and when the range underflows with the top byte still open, the loop runs as PCL writes it.
    if (__builtin_expect(pos + 4 <= in_len, 1)) {
      const uint32_t x = low ^ (low + range);
      const unsigned sh = (unsigned)_lzcnt_u32(x) & 0x38u;  // 8 * settled bytes (x != 0 because range != 0)
      uint32_t next4;
      memcpy(&next4, in + pos, 4);
      next4 = __builtin_bswap32(next4);
      code = (uint32_t)((((uint64_t)code << 32) | next4) >> (32u - sh));
      low = (uint32_t)((uint64_t)low << sh);
      range = (uint32_t)((uint64_t)range << sh);
      pos += sh >> 3;
      if (__builtin_expect(range >= kBottom, 1)) continue;
      range = (0u - low) & (kBottom - 1);
      {
        const uint8_t b = pos < in_len ? in[pos] : 0;
        ++pos;
        code = (code << 8) | b;
        range <<= 8;
        low <<= 8;
      }
    }
    for (;;) {
      if ((low ^ (low + range)) >= kTop) {
        if (range >= kBottom) break;
        range = (0u - low) & (kBottom - 1);
      }
      const uint8_t b = pos < in_len ? in[pos] : 0;
      ++pos;
      code = (code << 8) | b;
      range <<= 8;
      low <<= 8;
    }
  }
  return pos <= in_len ? pos : 0;
}

// =============================================================================================
// baseline JPEG
// =============================================================================================
namespace {

const uint8_t kLumaQ[64] = {16, 11, 10, 16, 24,  40,  51,  61,  12, 12, 14, 19, 26,  58,  60,  55,
                            14, 13, 16, 24, 40,  57,  69,  56,  14, 17, 22, 29, 51,  87,  80,  62,
                            18, 22, 37, 56, 68,  109, 103, 77,  24, 35, 55, 64, 81,  104, 113, 92,
                            49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
                              24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                              99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kZigzag[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,
                             12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                             35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
                             58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffSpec {
  uint8_t bits[16];
  const uint8_t* vals;
  int nvals;
};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71,
    0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72,
    0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59,
    0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
    0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
    0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22,
    0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1,
    0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58,
    0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a,
    0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba,
    0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
    0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
const HuffSpec kDcLuma = {{0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, kDcVals, 12};
const HuffSpec kDcChroma = {{0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0}, kDcVals, 12};
const HuffSpec kAcLuma = {{0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, kAcLumaVals, 162};
const HuffSpec kAcChroma = {{0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77}, kAcChromaVals, 162};

struct HuffEnc {
  uint32_t code[256];
  uint8_t len[256];
  explicit HuffEnc(const HuffSpec& s) {
    memset(code, 0, sizeof(code));
    memset(len, 0, sizeof(len));
    uint32_t c = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
      for (int i = 0; i < s.bits[l - 1]; ++i, ++k, ++c) {
        code[s.vals[k]] = c;
        len[s.vals[k]] = (uint8_t)l;
      }
      c <<= 1;
    }
  }
};

// quantiser: divide by 8*q with round-half-up on the magnitude, exactly as jcdctmgr.c; the
// division is done with a 32-bit reciprocal that is exact for every 17-bit dividend
struct Quant {
  uint16_t q[64];      // natural order
  uint32_t magic[64];  // floor(2^32 / (8q)) + 1
  uint16_t half[64];   // (8q) >> 1
  Quant(const uint8_t* basic, int quality) {
    quality = std::min(100, std::max(1, quality));
    const int scale = quality < 50 ? 5000 / quality : 200 - 2 * quality;
    for (int i = 0; i < 64; ++i) {
      long t = ((long)basic[i] * scale + 50L) / 100L;
      t = std::min(255L, std::max(1L, t));
      q[i] = (uint16_t)t;
      const uint32_t d = (uint32_t)t << 3;
      magic[i] = (uint32_t)((1ull << 32) / d) + 1u;
      half[i] = (uint16_t)(d >> 1);
    }
  }
  inline int16_t apply(int32_t v, int i) const {
    const uint32_t a = (uint32_t)(v < 0 ? -v : v) + half[i];
    const uint32_t r = (uint32_t)(((uint64_t)a * magic[i]) >> 32);
    return (int16_t)(v < 0 ? -(int32_t)r : (int32_t)r);
  }
};

#define PCC_FIX_0_298631336 2446
#define PCC_FIX_0_390180644 3196
#define PCC_FIX_0_541196100 4433
#define PCC_FIX_0_765366865 6270
#define PCC_FIX_0_899976223 7373
#define PCC_FIX_1_175875602 9633
#define PCC_FIX_1_501321110 12299
#define PCC_FIX_1_847759065 15137
#define PCC_FIX_1_961570560 16069
#define PCC_FIX_2_053119869 16819
#define PCC_FIX_2_562915447 20995
#define PCC_FIX_3_072711026 25172
inline int32_t descale(int32_t x, int n) { return (x + (1 << (n - 1))) >> n; }

// jfdctint.c: 13-bit constants, 2 extra bits carried between the passes, output scaled by 8
inline void fdct_1d(const int32_t in[8], int32_t out[8], int shift_even_up, int down_even, int down_odd) {
  const int32_t t0 = in[0] + in[7], t7 = in[0] - in[7];
  const int32_t t1 = in[1] + in[6], t6 = in[1] - in[6];
  const int32_t t2 = in[2] + in[5], t5 = in[2] - in[5];
  const int32_t t3 = in[3] + in[4], t4 = in[3] - in[4];
  const int32_t t10 = t0 + t3, t13 = t0 - t3, t11 = t1 + t2, t12 = t1 - t2;
  if (shift_even_up) {
    out[0] = (t10 + t11) * (1 << shift_even_up);  // libjpeg shifts the (possibly negative) sum left: same value, defined behaviour
    out[4] = (t10 - t11) * (1 << shift_even_up);
  } else {
    out[0] = descale(t10 + t11, down_even);
    out[4] = descale(t10 - t11, down_even);
  }
  int32_t z1 = (t12 + t13) * PCC_FIX_0_541196100;
  out[2] = descale(z1 + t13 * PCC_FIX_0_765366865, down_odd);
  out[6] = descale(z1 + t12 * (-PCC_FIX_1_847759065), down_odd);
  z1 = t4 + t7;
  int32_t z2 = t5 + t6, z3 = t4 + t6, z4 = t5 + t7;
  const int32_t z5 = (z3 + z4) * PCC_FIX_1_175875602;
  const int32_t a4 = t4 * PCC_FIX_0_298631336, a5 = t5 * PCC_FIX_2_053119869;
  const int32_t a6 = t6 * PCC_FIX_3_072711026, a7 = t7 * PCC_FIX_1_501321110;
  z1 *= -PCC_FIX_0_899976223;
  z2 *= -PCC_FIX_2_562915447;
  z3 = z3 * (-PCC_FIX_1_961570560) + z5;
  z4 = z4 * (-PCC_FIX_0_390180644) + z5;
  out[7] = descale(a4 + z1 + z3, down_odd);
  out[5] = descale(a5 + z2 + z4, down_odd);
  out[3] = descale(a6 + z2 + z3, down_odd);
  out[1] = descale(a7 + z1 + z4, down_odd);
}

// samples (already level shifted) -> quantised coefficients in natural order
inline void fdct_quant(const int32_t samples[64], const Quant& q, int16_t out[64]) {
  int32_t ws[64];
  for (int r = 0; r < 8; ++r) fdct_1d(samples + 8 * r, ws + 8 * r, 2, 0, 13 - 2);
  for (int c = 0; c < 8; ++c) {
    int32_t col[8], res[8];
    for (int r = 0; r < 8; ++r) col[r] = ws[8 * r + c];
    fdct_1d(col, res, 0, 2, 13 + 2);
    for (int r = 0; r < 8; ++r) out[8 * r + c] = q.apply(res[r], 8 * r + c);
  }
}

class BitSink {
 public:
  explicit BitSink(Bytes& out) : out_(out) {}
  // `len` <= 27 bits per call; bytes leave the accumulator four at a time
  inline void put(uint32_t code, int len) {
    acc_ = (acc_ << len) | (uint64_t)(code & ((1u << len) - 1u));
    n_ += len;
    if (n_ >= 32) {
      n_ -= 32;
      const uint32_t w = (uint32_t)(acc_ >> n_);
      if (used_ + 8 > out_.size()) out_.resize(out_.size() * 2 + 4096);
      uint8_t* p = out_.data() + used_;
      // 0xFF anywhere in the word?  (SWAR test on the complement for a zero byte)
      const uint32_t inv = ~w;
      if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0) {
        p[0] = (uint8_t)(w >> 24); p[1] = (uint8_t)(w >> 16); p[2] = (uint8_t)(w >> 8); p[3] = (uint8_t)w;
        used_ += 4;
      } else {
        size_t k = 0;
        for (int sft = 24; sft >= 0; sft -= 8) {
          const uint8_t b = (uint8_t)(w >> sft);
          p[k++] = b;
          if (b == 0xFF) p[k++] = 0;  // byte stuffing
        }
        used_ += k;
      }
    }
  }
  // bits [from, to) of a string kept as u32 words, MSB first inside each word (what the GPU's Huffman stages produce).
  // Behind a short head that brings the source to a word boundary the string moves 64 bits at a time: one shift-merge, one
  // test for 0xFF bytes, one 8-byte store (stitching the 231 MCU rows of a 1 M-voxel frame: 0.23 ms -> 0.06 ms per frame
  // compared with 24-bit pieces through put()).
  void append(const uint32_t* words, uint32_t from, uint32_t to) {
    uint32_t p = from;
    auto piece = [&](uint32_t upto) {  // through put(), at most 24 bits at a time, never across a source word
      while (p < upto) {
        const uint32_t off = p & 31u, take = std::min(std::min(upto - p, 32u - off), 24u);
        put(words[p >> 5] >> (32u - off - take), (int)take);
        p += take;
      }
    };
    piece(std::min(to, (p + 31u) & ~31u));
    while (to - p >= 64u) {  // p is a multiple of 32 here; n_ < 32 bits are pending in acc_
      const uint64_t w = ((uint64_t)words[p >> 5] << 32) | words[(p >> 5) + 1u];
      p += 64u;
      const uint64_t o = n_ ? ((acc_ << (64 - n_)) | (w >> n_)) : w;  // the next 64 bits of the output
      acc_ = n_ ? (w & ((1ull << n_) - 1ull)) : 0ull;                 // n_ stays what it was
      if (used_ + 16 > out_.size()) out_.resize(out_.size() * 2 + 4096);
      uint8_t* q = out_.data() + used_;
      const uint64_t inv = ~o;
      if (((inv - 0x0101010101010101ull) & ~inv & 0x8080808080808080ull) == 0) {  // no 0xFF byte among the eight
        const uint64_t be = __builtin_bswap64(o);
        memcpy(q, &be, 8);
        used_ += 8;
      } else {
        size_t k = 0;
        for (int sft = 56; sft >= 0; sft -= 8) {
          const uint8_t b = (uint8_t)(o >> sft);
          q[k++] = b;
          if (b == 0xFF) q[k++] = 0;  // byte stuffing
        }
        used_ += k;
      }
    }
    piece(to);
  }
  void start() { used_ = out_.size(); out_.resize(out_.size() + 4096); }
  void flush() {
    // remaining whole bytes, then pad the last one with ones (jchuff.c flush_bits)
    while (n_ >= 8) {
      n_ -= 8;
      emit((uint8_t)(acc_ >> n_));
    }
    if (n_ > 0) {
      const uint8_t b = (uint8_t)(((acc_ << (8 - n_)) | ((1u << (8 - n_)) - 1u)) & 0xFF);
      emit(b);
    }
    acc_ = 0;
    n_ = 0;
    out_.resize(used_);
  }

 private:
  inline void emit(uint8_t b) {
    if (used_ + 2 > out_.size()) out_.resize(out_.size() * 2 + 4096);
    out_[used_++] = b;
    if (b == 0xFF) out_[used_++] = 0;
  }
  Bytes& out_;
  size_t used_ = 0;
  uint64_t acc_ = 0;
  int n_ = 0;
};

inline int bit_length(uint32_t v) { return v ? 32 - __builtin_clz(v) : 0; }

inline void huff_block(BitSink& bs, const int16_t blk[64], int& last_dc, const HuffEnc& dc, const HuffEnc& ac) {
  int diff = blk[0] - last_dc;
  last_dc = blk[0];
  int mag = diff < 0 ? -diff : diff, low = diff < 0 ? diff - 1 : diff;
  int nb = bit_length((uint32_t)mag);
  bs.put(dc.code[nb], dc.len[nb]);
  if (nb) bs.put((uint32_t)low, nb);
  int run = 0;
  for (int k = 1; k < 64; ++k) {
    const int v = blk[kZigzag[k]];
    if (v == 0) { ++run; continue; }
    while (run > 15) { bs.put(ac.code[0xF0], ac.len[0xF0]); run -= 16; }
    mag = v < 0 ? -v : v;
    low = v < 0 ? v - 1 : v;
    nb = bit_length((uint32_t)mag);
    const int sym = (run << 4) | nb;
    bs.put(ac.code[sym], ac.len[sym]);
    bs.put((uint32_t)low, nb);
    run = 0;
  }
  if (run) bs.put(ac.code[0], ac.len[0]);
}

inline void be16(Bytes& o, unsigned v) { o.push_back((uint8_t)(v >> 8)); o.push_back((uint8_t)v); }
void put_dqt(Bytes& o, int id, const Quant& q) {
  be16(o, 0xFFDB); be16(o, 67); o.push_back((uint8_t)id);
  for (int i = 0; i < 64; ++i) o.push_back((uint8_t)q.q[kZigzag[i]]);
}
void put_dht(Bytes& o, int tc_th, const HuffSpec& s) {
  be16(o, 0xFFC4); be16(o, (unsigned)(19 + s.nvals)); o.push_back((uint8_t)tc_th);
  o.insert(o.end(), s.bits, s.bits + 16);
  o.insert(o.end(), s.vals, s.vals + s.nvals);
}
// headers: SOI, JFIF APP0, DQT x2, SOF0 (2x2,1x1,1x1), DHT x4, SOS (jcmarker.c order)
void put_headers(Bytes& out, int w, int h, const Quant& ql, const Quant& qc) {
  be16(out, 0xFFD8);
  be16(out, 0xFFE0); be16(out, 16);
  const uint8_t jfif[14] = {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0};
  out.insert(out.end(), jfif, jfif + 14);
  put_dqt(out, 0, ql);
  put_dqt(out, 1, qc);
  be16(out, 0xFFC0); be16(out, 17); out.push_back(8); be16(out, (unsigned)h); be16(out, (unsigned)w); out.push_back(3);
  const uint8_t comps[9] = {1, 0x22, 0, 2, 0x11, 1, 3, 0x11, 1};
  out.insert(out.end(), comps, comps + 9);
  put_dht(out, 0x00, kDcLuma); put_dht(out, 0x10, kAcLuma);
  put_dht(out, 0x01, kDcChroma); put_dht(out, 0x11, kAcChroma);
  be16(out, 0xFFDA); be16(out, 12);
  const uint8_t sos[10] = {3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 63, 0};
  out.insert(out.end(), sos, sos + 10);
}

// one block whose 64 coefficients are already in zigzag order; `dc` overrides zz[0] (dummy blocks)
inline void huff_block_zz(BitSink& bs, const int16_t* zz, int dc, int& last_dc, const HuffEnc& dct, const HuffEnc& act) {
  int diff = dc - last_dc;
  last_dc = dc;
  int mag = diff < 0 ? -diff : diff, low = diff < 0 ? diff - 1 : diff;
  int nb = bit_length((uint32_t)mag);
  bs.put(dct.code[nb], dct.len[nb]);
  if (nb) bs.put((uint32_t)low, nb);
  // skip the zero tail in 8-byte steps: most blocks of smooth colour fields end early
  int last = 63;
  {
    const uint64_t* w = reinterpret_cast<const uint64_t*>(zz);
    int q = 15;
    while (q > 0 && w[q] == 0) --q;
    last = 4 * q + 3;
    while (last > 0 && zz[last] == 0) --last;
  }
  int run = 0;
  for (int k = 1; k <= last; ++k) {
    const int v = zz[k];
    if (v == 0) { ++run; continue; }
    while (run > 15) { bs.put(act.code[0xF0], act.len[0xF0]); run -= 16; }
    mag = v < 0 ? -v : v;
    low = v < 0 ? v - 1 : v;
    nb = bit_length((uint32_t)mag);
    const int sym = (run << 4) | nb;
    bs.put(act.code[sym], act.len[sym]);
    bs.put((uint32_t)low, nb);
    run = 0;
  }
  if (last < 63) bs.put(act.code[0], act.len[0]);
}
}  // namespace

void BaselineJpeg::quantiser(int quality, uint16_t half[2][64], uint32_t magic[2][64]) {
  const Quant ql(kLumaQ, quality), qc(kChromaQ, quality);
  for (int i = 0; i < 64; ++i) {
    half[0][i] = ql.half[i]; magic[0][i] = ql.magic[i];
    half[1][i] = qc.half[i]; magic[1][i] = qc.magic[i];
  }
}

void BaselineJpeg::huffman_tables(uint32_t dc[2][12], uint32_t ac[2][256]) {
  static const HuffEnc dcl(kDcLuma), acl(kAcLuma), dcc(kDcChroma), acc(kAcChroma);
  for (int i = 0; i < 12; ++i) {
    dc[0][i] = ((uint32_t)dcl.len[i] << 16) | dcl.code[i];
    dc[1][i] = ((uint32_t)dcc.len[i] << 16) | dcc.code[i];
  }
  for (int i = 0; i < 256; ++i) {
    ac[0][i] = ((uint32_t)acl.len[i] << 16) | acl.code[i];
    ac[1][i] = ((uint32_t)acc.len[i] << 16) | acc.code[i];
  }
}

bool BaselineJpeg::encode_tiles(const uint32_t* tiles, uint32_t tile_words, uint32_t n_tiles, int w, int h, int quality,
                                Bytes& out) {
  for (uint32_t m = 0; m < n_tiles; ++m)
    if (tiles[(size_t)m * tile_words + 3] != 0) return false;
  const Quant ql(kLumaQ, quality), qc(kChromaQ, quality);
  static const HuffEnc dcl(kDcLuma), dcc(kDcChroma);
  put_headers(out, w, h, ql, qc);
  BitSink bs(out);
  bs.start();
  auto append = [&](const uint32_t* words, uint32_t from, uint32_t to) { bs.append(words, from, to); };
  auto put_dc = [&](int diff, const HuffEnc& t) {  // jchuff.c encode_one_block, DC part
    const int mag = diff < 0 ? -diff : diff, low = diff < 0 ? diff - 1 : diff;
    const int nb = bit_length((uint32_t)mag);
    bs.put(t.code[nb], t.len[nb]);
    if (nb) bs.put((uint32_t)low, nb);
  };
  int last_dc[3] = {0, 0, 0};
  for (uint32_t m = 0; m < n_tiles; ++m) {
    const uint32_t* rec = tiles + (size_t)m * tile_words;
    const uint32_t* bits = rec + 16;
    const uint32_t total = rec[0], b1 = rec[1], b2 = rec[2];
    put_dc((int)rec[4] - last_dc[0], dcl);
    append(bits, 0, b1);
    put_dc((int)rec[5] - last_dc[1], dcc);
    append(bits, b1, b2);
    put_dc((int)rec[6] - last_dc[2], dcc);
    append(bits, b2, total);
    last_dc[0] = (int)rec[7]; last_dc[1] = (int)rec[8]; last_dc[2] = (int)rec[9];
  }
  bs.flush();
  be16(out, 0xFFD9);
  return true;
}

void BaselineJpeg::wrap_bits(const uint32_t* words, uint32_t n_bits, int w, int h, int quality, Bytes& out) {
  const Quant ql(kLumaQ, quality), qc(kChromaQ, quality);
  put_headers(out, w, h, ql, qc);
  BitSink bs(out);
  bs.start();
  bs.append(words, 0, n_bits);
  bs.flush();
  be16(out, 0xFFD9);
}

void BaselineJpeg::encode_coefs(const int16_t* coefs, int w, int h, int quality, Bytes& out) {
  const Quant ql(kLumaQ, quality), qc(kChromaQ, quality);
  static const HuffEnc dcl(kDcLuma), acl(kAcLuma), dcc(kDcChroma), acc(kAcChroma);
  put_headers(out, w, h, ql, qc);
  const int y_hb = (h + 7) / 8;
  const int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16;
  BitSink bs(out);
  bs.start();
  int last_dc[3] = {0, 0, 0};
  for (int my = 0; my < mcus_y; ++my) {
    const bool dummy_row = 2 * my + 1 >= y_hb;  // the MCU's second luma block row lies below the image
    for (int mx = 0; mx < mcus_x; ++mx) {
      const int16_t* m = coefs + ((size_t)my * mcus_x + mx) * 6 * 64;
      huff_block_zz(bs, m, m[0], last_dc[0], dcl, acl);
      huff_block_zz(bs, m + 64, m[64], last_dc[0], dcl, acl);
      // dummy blocks are all zero with the DC of the block just before their row (the top-right one)
      huff_block_zz(bs, m + 128, dummy_row ? m[64] : m[128], last_dc[0], dcl, acl);
      huff_block_zz(bs, m + 192, dummy_row ? m[64] : m[192], last_dc[0], dcl, acl);
      huff_block_zz(bs, m + 256, m[256], last_dc[1], dcc, acc);
      huff_block_zz(bs, m + 320, m[320], last_dc[2], dcc, acc);
    }
  }
  bs.flush();
  be16(out, 0xFFD9);
}

void BaselineJpeg::encode_rgb(const uint8_t* rgb, int w, int h, int quality, Bytes& out) {
  const Quant ql(kLumaQ, quality), qc(kChromaQ, quality);
  static const HuffEnc dcl(kDcLuma), acl(kAcLuma), dcc(kDcChroma), acc(kAcChroma);

  put_headers(out, w, h, ql, qc);

  // colour planes (jccolor.c rgb_ycc_convert, 16-bit fixed point)
  const size_t npx = (size_t)w * h;
  std::vector<uint8_t> planes(3 * npx);
  uint8_t* Y = planes.data();
  uint8_t* Cb = Y + npx;
  uint8_t* Cr = Cb + npx;
  for (size_t i = 0; i < npx; ++i) {
    const int32_t r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    Y[i] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
    Cb[i] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
    Cr[i] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
  }

  const int ch = (h + 1) / 2;                    // real chroma rows
  const int y_wb = (w + 7) / 8, y_hb = (h + 7) / 8;  // real luma blocks
  const int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16;
  BitSink bs(out);
  bs.start();
  int last_dc[3] = {0, 0, 0};
  int16_t blk[6][64];
  int32_t smp[64];

  auto clampi = [](int v, int hi) { return v < hi ? v : hi; };
  for (int my = 0; my < mcus_y; ++my) {
    for (int mx = 0; mx < mcus_x; ++mx) {
      for (int yi = 0; yi < 2; ++yi) {
        const int by = 2 * my + yi;
        for (int xi = 0; xi < 2; ++xi) {
          const int bx = 2 * mx + xi;
          int16_t* dst = blk[2 * yi + xi];
          if (by >= y_hb) {  // dummy block row below the image: DC of the block before this row
            memset(dst, 0, sizeof(blk[0]));
            dst[0] = blk[2 * yi - 1][0];
          } else if (bx >= y_wb) {  // dummy block right of the image: DC of the left neighbour
            memset(dst, 0, sizeof(blk[0]));
            dst[0] = blk[2 * yi + xi - 1][0];
          } else {
            for (int r = 0; r < 8; ++r) {
              const uint8_t* row = Y + (size_t)clampi(8 * by + r, h - 1) * w;
              for (int c = 0; c < 8; ++c) smp[8 * r + c] = (int32_t)row[clampi(8 * bx + c, w - 1)] - 128;
            }
            fdct_quant(smp, ql, dst);
          }
        }
      }
      for (int k = 0; k < 2; ++k) {  // h2v2 box downsample with alternating bias 1,2 (jcsample.c)
        const uint8_t* P = k ? Cr : Cb;
        for (int r = 0; r < 8; ++r) {
          const int cr_row = clampi(8 * my + r, ch - 1);  // component rows replicated below the image
          const uint8_t* r0 = P + (size_t)clampi(2 * cr_row, h - 1) * w;
          const uint8_t* r1 = P + (size_t)clampi(2 * cr_row + 1, h - 1) * w;
          for (int c = 0; c < 8; ++c) {
            const int cc = 8 * mx + c;
            const int c0 = clampi(2 * cc, w - 1), c1 = clampi(2 * cc + 1, w - 1);
            smp[8 * r + c] = ((r0[c0] + r0[c1] + r1[c0] + r1[c1] + ((cc & 1) ? 2 : 1)) >> 2) - 128;
          }
        }
        fdct_quant(smp, qc, blk[4 + k]);
      }
      for (int i = 0; i < 4; ++i) huff_block(bs, blk[i], last_dc[0], dcl, acl);
      huff_block(bs, blk[4], last_dc[1], dcc, acc);
      huff_block(bs, blk[5], last_dc[2], dcc, acc);
    }
  }
  bs.flush();
  be16(out, 0xFFD9);
}

// ---------------------------------------------------------------------------------------------
// decoder (libjpeg defaults: islow IDCT, fancy upsampling when downsampled_width > 2)
// ---------------------------------------------------------------------------------------------
namespace {
// PCC_DECODE_TRACE=1: where the host decoder spends its time (stderr)
struct DecodeTrace {
  bool on;
  std::chrono::steady_clock::time_point t;
  DecodeTrace() : on(dev_env("PCC_DECODE_TRACE") != nullptr), t(std::chrono::steady_clock::now()) {}
  void lap(const char* what) {
    if (!on) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[pcc decode] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t).count());
    t = now;
  }
};
struct HuffDec {
  int32_t maxcode[18] = {};
  int32_t valoff[17] = {};
  uint8_t vals[256] = {};
  // the first kLook bits of the stream -> (code length << 8) | symbol, 0 if the code is longer than kLook bits.  Filled
  // by running the length-by-length search below on every kLook-bit pattern, so a table that is not a proper prefix code
  // (corrupt stream) decodes exactly as the search alone would.
  static constexpr int kLook = 9;
  uint16_t fast[1 << kLook] = {};
  bool ok = false;
  void build(const uint8_t bits[16], const uint8_t* v, int n) {
    memcpy(vals, v, (size_t)n);
    int32_t code = 0, p = 0;
    for (int l = 1; l <= 16; ++l) {
      if (bits[l - 1]) {
        valoff[l] = p - code;
        p += bits[l - 1];
        code += bits[l - 1];
        maxcode[l] = code - 1;
      } else {
        maxcode[l] = -1;
      }
      code <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    for (uint32_t pat = 0; pat < (1u << kLook); ++pat) {
      fast[pat] = 0;
      int32_t c = 0;
      for (int l = 1; l <= kLook; ++l) {
        c = (c << 1) | (int32_t)((pat >> (kLook - l)) & 1u);
        if (maxcode[l] >= 0 && c <= maxcode[l]) {
          fast[pat] = (uint16_t)((l << 8) | vals[(c + valoff[l]) & 0xFF]);
          break;
        }
      }
    }
    ok = true;
  }
};
// The entropy-coded segment as a bit string: 0xFF 0x00 is a data byte 0xFF, any other 0xFF xx is a marker, behind which
// (and behind the end of the data) the decoder reads zeros, as libjpeg does.  `pos` never moves past a marker.
struct BitSource {
  const uint8_t* p;
  size_t len, pos;
  uint64_t acc = 0;  // the low n bits are the next n bits of the string
  int n = 0;
  bool marker = false;
  inline void refill() {  // afterwards n > 56
    while (n <= 56) {
      uint32_t v = 0;
      if (!marker && pos < len) {
        v = p[pos];
        if (v == 0xFF) {
          if (pos + 1 < len && p[pos + 1] == 0) pos += 2;
          else { marker = true; v = 0; }
        } else {
          ++pos;
        }
      }
      acc = (acc << 8) | v;
      n += 8;
    }
  }
  inline int bits(int k) {  // k <= 16 for any valid stream
    if (k <= 0) return 0;
    if (k > 31) k = 31;
    if (n < k) refill();
    n -= k;
    return (int)((uint32_t)(acc >> n) & ((1u << k) - 1u));
  }
  inline int sym(const HuffDec& t) {
    if (n < 16) refill();
    const uint32_t look = (uint32_t)(acc >> (n - 16)) & 0xFFFFu;  // the next 16 bits
    const uint16_t e = t.fast[look >> (16 - HuffDec::kLook)];
    if (e) {
      n -= e >> 8;
      return e & 0xFF;
    }
    int32_t code = (int32_t)(look >> (16 - HuffDec::kLook));
    for (int l = HuffDec::kLook + 1; l <= 16; ++l) {
      code = (code << 1) | (int32_t)((look >> (16 - l)) & 1u);
      if (t.maxcode[l] >= 0 && code <= t.maxcode[l]) {
        n -= l;
        return t.vals[(code + t.valoff[l]) & 0xFF];
      }
    }
    n -= 16;  // no code of any length: sixteen bits are gone, the symbol is 0 (what the bit-by-bit search did)
    return 0;
  }
};
inline int extend_sign(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

// The inverse DCT works in 64-bit integers: for a valid stream every value fits 32 bits and the results are libjpeg's;
// for a corrupt one (coefficient x quantiser up to 2^31) nothing overflows, so garbage in is garbage out and not
// undefined behaviour.
typedef int64_t idct_t;
inline idct_t descale64(idct_t x, int n) { return (x + ((idct_t)1 << (n - 1))) >> n; }
inline uint8_t idct_clamp(idct_t v) {  // sample_range_limit + CENTERJSAMPLE, index masked to 10 bits
  const int i = (int)(v & 1023);
  if (i < 128) return (uint8_t)(128 + i);
  if (i < 512) return 255;
  if (i < 896) return 0;
  return (uint8_t)(i - 896);
}
inline void idct_1d(const idct_t in[8], idct_t o[8]) {  // jidctint.c butterfly, unscaled outputs
  idct_t z2 = in[2], z3 = in[6];
  idct_t z1 = (z2 + z3) * PCC_FIX_0_541196100;
  const idct_t e2 = z1 + z3 * (-PCC_FIX_1_847759065), e3 = z1 + z2 * PCC_FIX_0_765366865;
  const idct_t e0 = (in[0] + in[4]) * 8192, e1 = (in[0] - in[4]) * 8192;
  const idct_t t10 = e0 + e3, t13 = e0 - e3, t11 = e1 + e2, t12 = e1 - e2;
  idct_t t0 = in[7], t1 = in[5], t2 = in[3], t3 = in[1];
  z1 = t0 + t3; z2 = t1 + t2; z3 = t0 + t2;
  idct_t z4 = t1 + t3;
  const idct_t z5 = (z3 + z4) * PCC_FIX_1_175875602;
  t0 *= PCC_FIX_0_298631336; t1 *= PCC_FIX_2_053119869; t2 *= PCC_FIX_3_072711026; t3 *= PCC_FIX_1_501321110;
  z1 *= -PCC_FIX_0_899976223; z2 *= -PCC_FIX_2_562915447;
  z3 = z3 * (-PCC_FIX_1_961570560) + z5;
  z4 = z4 * (-PCC_FIX_0_390180644) + z5;
  t0 += z1 + z3; t1 += z2 + z4; t2 += z2 + z3; t3 += z1 + z4;
  o[0] = t10 + t3; o[7] = t10 - t3; o[1] = t11 + t2; o[6] = t11 - t2;
  o[2] = t12 + t1; o[5] = t12 - t1; o[3] = t13 + t0; o[4] = t13 - t0;
}
void idct_block(const int16_t coef[64], const uint16_t q[64], uint8_t* dst, int stride) {
  idct_t ws[64], in[8], o[8];
  // jidctint.c's short cuts, which give what the butterfly gives: a column (row) whose AC terms are all zero is its DC
  // term times 8192 in every output, i.e. 4 * DC after the first descale and (DC + 16) >> 5 after the second.  Most
  // columns of a quantised block are like that.
  for (int c = 0; c < 8; ++c) {
    if ((coef[8 + c] | coef[16 + c] | coef[24 + c] | coef[32 + c] | coef[40 + c] | coef[48 + c] | coef[56 + c]) == 0) {
      const idct_t dc = (idct_t)coef[c] * q[c] * 4;
      for (int r = 0; r < 8; ++r) ws[8 * r + c] = dc;
      continue;
    }
    for (int r = 0; r < 8; ++r) in[r] = (idct_t)coef[8 * r + c] * q[8 * r + c];
    idct_1d(in, o);
    for (int r = 0; r < 8; ++r) ws[8 * r + c] = descale64(o[r], 13 - 2);
  }
  for (int r = 0; r < 8; ++r) {
    const idct_t* w = ws + 8 * r;
    uint8_t* d = dst + (size_t)r * stride;
    if ((w[1] | w[2] | w[3] | w[4] | w[5] | w[6] | w[7]) == 0) {
      const uint8_t v = idct_clamp(descale64(w[0], 2 + 3));
      for (int c = 0; c < 8; ++c) d[c] = v;
      continue;
    }
    idct_1d(w, o);
    for (int c = 0; c < 8; ++c) d[c] = idct_clamp(descale64(o[c], 13 + 2 + 3));
  }
}
}  // namespace

bool BaselineJpeg::decode_rgb(const uint8_t* jpg, size_t len, Bytes& rgb, int& w, int& h, uint64_t max_pixels) {
  return decode_impl(jpg, len, &rgb, w, h, nullptr, max_pixels);
}
bool BaselineJpeg::decode_coefs(const uint8_t* jpg, size_t len, int& w, int& h, JpegCoefs& out, uint64_t max_pixels) {
  return decode_impl(jpg, len, nullptr, w, h, &out, max_pixels);
}

bool BaselineJpeg::decode_impl(const uint8_t* jpg, size_t len, Bytes* rgb_out, int& w, int& h, JpegCoefs* coefs_out, uint64_t max_pixels) {
  uint16_t qt[4][64] = {};
  HuffDec hd[2][4];
  int restart = 0;
  int samp[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, tdc[3] = {0, 0, 0}, tac[3] = {0, 0, 0};
  w = h = 0;
  if (len < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return false;
  size_t pos = 2;
  bool sos = false;
  while (!sos && pos + 4 <= len) {
    if (jpg[pos] != 0xFF) return false;
    const int m = jpg[pos + 1];
    pos += 2;
    if (m == 0xFF) { --pos; continue; }
    if (m == 0xD8 || m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
    const size_t seg = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
    if (seg < 2 || pos + seg > len) return false;
    const uint8_t* s = jpg + pos + 2;
    const size_t n = seg - 2;
    if (m == 0xDB) {
      for (size_t i = 0; i < n;) {
        const int pq = s[i] >> 4, id = s[i] & 15;
        ++i;
        if (id > 3 || i + (pq ? 128 : 64) > n) return false;
        for (int k = 0; k < 64; ++k) {
          qt[id][kZigzag[k]] = pq ? (uint16_t)((s[i] << 8) | s[i + 1]) : s[i];
          i += pq ? 2 : 1;
        }
      }
    } else if (m == 0xC4) {
      for (size_t i = 0; i < n;) {
        const int tc = s[i] >> 4, th = s[i] & 15;
        ++i;
        if (tc > 1 || th > 3 || i + 16 > n) return false;
        int cnt = 0;
        for (int k = 0; k < 16; ++k) cnt += s[i + k];
        if (cnt > 256 || i + 16 + cnt > n) return false;
        hd[tc][th].build(s + i, s + i + 16, cnt);
        i += 16 + (size_t)cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (n < 15 || s[0] != 8 || s[5] != 3) return false;
      h = (s[1] << 8) | s[2];
      w = (s[3] << 8) | s[4];
      for (int c = 0; c < 3; ++c) { samp[c] = s[7 + 3 * c]; tq[c] = s[8 + 3 * c] & 3; }
    } else if (m == 0xC2 || (m > 0xC4 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return false;
    } else if (m == 0xDD) {
      if (n < 2) return false;
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (n < 10 || s[0] != 3) return false;
      for (int c = 0; c < 3; ++c) { tdc[c] = (s[2 + 2 * c] >> 4) & 3; tac[c] = s[2 + 2 * c] & 3; }
      sos = true;
    }
    pos += seg;
  }
  if (!sos || w <= 0 || h <= 0 || samp[0] != 0x22 || samp[1] != 0x11 || samp[2] != 0x11) return false;
  for (int c = 0; c < 3; ++c)  // a scan that names a table no DHT segment defined
    if (!hd[0][tdc[c]].ok || !hd[1][tac[c]].ok) return false;
  // An MCU of six blocks takes at least six bits of entropy data per block pair ... in any case more than one byte: a
  // header that promises more MCUs than the payload has bytes is corrupt (and would ask for gigabytes below)
  // (with one-bit codes for "size 0" and "end of block" an MCU is still twelve bits: 2 MCUs per 3 bytes at the very most)
  if ((uint64_t)((w + 15) / 16) * (uint64_t)((h + 15) / 16) * 3u > 2u * (uint64_t)len) return false;
  // a caller that knows how many pixels the image can have (a frame's voxel count) says so: 768 bytes of coefficients
  // per MCU against a few bytes of payload is otherwise a 500-fold amplification a hostile frame header could ask for
  if (max_pixels && (uint64_t)w * (uint64_t)h > max_pixels) return false;

  const int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16;
  const int cw = (w + 1) / 2, chh = (h + 1) / 2;
  const int yw = mcus_x * 16, cpw = mcus_x * 8;
  std::vector<uint8_t> Y, C0, C1;
  if (coefs_out) {  // entropy decoding only: the quantised coefficients, six blocks per MCU (Y00 Y01 Y10 Y11 Cb Cr), natural order
    coefs_out->blocks.assign((size_t)mcus_x * mcus_y * 6 * 64, 0);
    for (int k = 0; k < 64; ++k) { coefs_out->q[0][k] = qt[tq[0]][k]; coefs_out->q[1][k] = qt[tq[1]][k]; coefs_out->q[2][k] = qt[tq[2]][k]; }
    coefs_out->mcus_x = mcus_x; coefs_out->mcus_y = mcus_y;
  } else {
    Y.resize((size_t)yw * mcus_y * 16); C0.resize((size_t)cpw * mcus_y * 8); C1.resize((size_t)cpw * mcus_y * 8);
  }
  uint8_t* C[2] = {C0.data(), C1.data()};
  DecodeTrace jt;
  BitSource br{jpg, len, pos};
  int last_dc[3] = {0, 0, 0}, count = 0, next_rst = 0;
  int16_t blk_local[64];
  int16_t* blk = blk_local;
  size_t blocks_done = 0;
  auto one_block = [&](int c, uint8_t* dst, int stride) {
    if (coefs_out) blk = coefs_out->blocks.data() + 64 * blocks_done++;
    memset(blk, 0, 64 * sizeof(int16_t));
    int sz = br.sym(hd[0][tdc[c]]);
    if (sz > 16) sz = 0;  // corrupt table: libjpeg treats a bad code as zero as well
    last_dc[c] = (int16_t)(last_dc[c] + (sz ? extend_sign(br.bits(sz), sz) : 0));  // stays in range on corrupt data too
    blk[0] = (int16_t)last_dc[c];
    for (int k = 1; k < 64; ++k) {
      const int rs = br.sym(hd[1][tac[c]]), r = rs >> 4, s4 = rs & 15;
      if (s4) {
        k += r;
        if (k > 63) break;
        blk[kZigzag[k]] = (int16_t)extend_sign(br.bits(s4), s4);
      } else if (r == 15) {
        k += 15;
      } else {
        break;
      }
    }
    if (!coefs_out) idct_block(blk, qt[tq[c]], dst, stride);
  };
  for (int my = 0; my < mcus_y; ++my)
    for (int mx = 0; mx < mcus_x; ++mx) {
      if (restart && count && count % restart == 0) {
        br.n = 0; br.marker = false;
        while (br.pos + 1 < br.len && !(br.p[br.pos] == 0xFF && br.p[br.pos + 1] == 0xD0 + next_rst)) ++br.pos;
        br.pos += 2;
        next_rst = (next_rst + 1) & 7;
        last_dc[0] = last_dc[1] = last_dc[2] = 0;
      }
      for (int yi = 0; yi < 2; ++yi)
        for (int xi = 0; xi < 2; ++xi) one_block(0, coefs_out ? nullptr : Y.data() + (size_t)(16 * my + 8 * yi) * yw + 16 * mx + 8 * xi, yw);
      one_block(1, coefs_out ? nullptr : C[0] + (size_t)8 * my * cpw + 8 * mx, cpw);
      one_block(2, coefs_out ? nullptr : C[1] + (size_t)8 * my * cpw + 8 * mx, cpw);
      ++count;
    }
  if (coefs_out) return true;
  Bytes& rgb = *rgb_out;
  jt.lap("  jpeg: huffman + idct");

  // chroma upsampling (jdsample.c): triangle filter if downsampled_width > 2, else replication
  const int uw = cpw * 2;
  std::vector<uint8_t> U0((size_t)uw * chh * 2), U1((size_t)uw * chh * 2);
  uint8_t* U[2] = {U0.data(), U1.data()};
  for (int k = 0; k < 2; ++k)
    for (int r = 0; r < chh; ++r)
      for (int v = 0; v < 2; ++v) {
        uint8_t* o = U[k] + (size_t)(2 * r + v) * uw;
        const uint8_t* in0 = C[k] + (size_t)r * cpw;
        if (cw <= 2) {
          for (int c = 0; c < cw; ++c) o[2 * c] = o[2 * c + 1] = in0[c];
          continue;
        }
        const int nr = v == 0 ? std::max(r - 1, 0) : std::min(r + 1, chh - 1);
        const uint8_t* in1 = C[k] + (size_t)nr * cpw;
        int cur = in0[0] * 3 + in1[0], nxt = in0[1] * 3 + in1[1], prv;
        *o++ = (uint8_t)((cur * 4 + 8) >> 4);
        *o++ = (uint8_t)((cur * 3 + nxt + 7) >> 4);
        prv = cur; cur = nxt;
        for (int c = 2; c < cw; ++c) {
          nxt = in0[c] * 3 + in1[c];
          *o++ = (uint8_t)((cur * 3 + prv + 8) >> 4);
          *o++ = (uint8_t)((cur * 3 + nxt + 7) >> 4);
          prv = cur; cur = nxt;
        }
        *o++ = (uint8_t)((cur * 3 + prv + 8) >> 4);
        *o++ = (uint8_t)((cur * 4 + 7) >> 4);
      }
  jt.lap("  jpeg: chroma upsampling");
  rgb.resize((size_t)w * h * 3);
  // jdcolor.c: the chroma terms of every pixel come out of four 256-entry tables (as libjpeg builds them)
  int32_t cr_r[256], cb_b[256], cb_g[256], cr_g[256];
  for (int v = 0; v < 256; ++v) {
    const int32_t x = v - 128;
    cr_r[v] = (91881 * x + 32768) >> 16;
    cb_b[v] = (116130 * x + 32768) >> 16;
    cb_g[v] = -22554 * x + 32768;
    cr_g[v] = -46802 * x;
  }
  auto sat = [](int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); };
  for (int r = 0; r < h; ++r) {
    const uint8_t* yr = Y.data() + (size_t)r * yw;
    const uint8_t* ub = U[0] + (size_t)r * uw;
    const uint8_t* ur = U[1] + (size_t)r * uw;
    uint8_t* o = rgb.data() + (size_t)r * w * 3;
    for (int c = 0; c < w; ++c, o += 3) {
      const int y = yr[c];
      o[0] = sat(y + cr_r[ur[c]]);
      o[1] = sat(y + ((cb_g[ub[c]] + cr_g[ur[c]]) >> 16));
      o[2] = sat(y + cb_b[ub[c]]);
    }
  }
  jt.lap("  jpeg: colour conversion");
  return true;
}

// =============================================================================================
// snake grid mapping, closed form (same as the device code; snake.h:46-71)
// =============================================================================================
uint32_t snake_position(uint32_t i, uint32_t W, uint32_t H) {
  const uint32_t full = H / 8u, hl = H % 8u, per_row = W * 8u;
  if (i < full * per_row) {
    const uint32_t br = i / per_row, rem = i % per_row, bw = rem / 64u, q = rem % 64u, r = q / 8u, c = q % 8u;
    return (br * 8u + r) * W + bw * 8u + ((r & 1u) ? 7u - c : c);
  }
  const uint32_t rem = i - full * per_row, blk = 8u * hl, bw = rem / blk, q = rem % blk, r = q / 8u, c = q % 8u;
  const uint32_t flip = (r + ((hl & 1u) ? bw : 0u)) & 1u;
  return (full * 8u + r) * W + bw * 8u + (flip ? 7u - c : c);
}

// =============================================================================================
// frame header + entropy stage
// =============================================================================================
namespace {
template <typename T>
inline void put_le(Bytes& o, T v) {
  const uint8_t* p = reinterpret_cast<const uint8_t*>(&v);
  o.insert(o.end(), p, p + sizeof(T));
}
const char kV2Id[] = "<PCL-OCT-CODECV2-COMPRESSED>";
const char kV1Id[] = "<PCL-OCT-COMPRESSED>";
}  // namespace

namespace {
void write_frame_header(const pcc_hot_result& hot, const pcc_params& prm, Bytes& out) {
  out.clear();
  // --- 140-byte header (impl.hpp:1472-1486 + the PCL base header it calls) ---
  out.insert(out.end(), kV2Id, kV2Id + 28);
  out.insert(out.end(), kV1Id, kV1Id + 20);
  put_le<uint32_t>(out, prm.frame_id);
  put_le<uint8_t>(out, 1);  // i_frame_
  put_le<uint8_t>(out, 1);  // do_voxel_grid_enDecoding_
  put_le<uint8_t>(out, prm.do_color_encoding ? 1 : 0);
  put_le<uint64_t>(out, (uint64_t)hot.n_leaves);
  put_le<double>(out, prm.octree_resolution);
  put_le<uint8_t>(out, (uint8_t)prm.color_bit_resolution);
  put_le<double>(out, (double)(float)prm.point_resolution);
  for (int i = 0; i < 6; ++i) put_le<double>(out, hot.bbox[i]);
  put_le<uint8_t>(out, prm.do_voxel_centroid ? 1 : 0);
  put_le<uint8_t>(out, prm.do_connectivity ? 1 : 0);
  put_le<uint8_t>(out, prm.create_scalable ? 1 : 0);
  put_le<uint32_t>(out, (uint32_t)prm.color_coding_type);
  put_le<int32_t>(out, prm.macroblock_size);
  put_le<uint8_t>(out, prm.do_icp_color_offset ? 1 : 0);
}

// what the colour range coder gets (jpegcc.h:115-139): raw b,g,r bytes, or the JPEG file(s) made of them
void colour_payload(const pcc_hot_result& hot, const pcc_params& prm, Bytes& payload, const uint8_t*& src, size_t& src_len) {
  const size_t L = (size_t)hot.n_leaves;
  src = hot.bgr;
  src_len = 3 * L;
  if (prm.color_coding_type == 1) {  // one 256-wide snake-mapped image (jpegcc.h:187-226)
    if (hot.jpeg_tiles &&
        BaselineJpeg::encode_tiles(hot.jpeg_tiles, hot.jpeg_tile_words, hot.jpeg_n_tiles, (int)hot.image_w, (int)hot.image_h,
                                   prm.jpeg_quality, payload)) {
      // stitched from the GPU's per-row bit strings
    } else if (hot.jpeg_coefs) {
      BaselineJpeg::encode_coefs(hot.jpeg_coefs, (int)hot.image_w, (int)hot.image_h, prm.jpeg_quality, payload);
    } else {
      BaselineJpeg::encode_rgb(hot.image, (int)hot.image_w, (int)hot.image_h, prm.jpeg_quality, payload);
    }
    src = payload.data();
    src_len = payload.size();
  } else if (prm.color_coding_type == 2) {  // 2048x1 lines (jpegcc.h:244-317)
    const size_t lines = L / 2048;
    const uint32_t count = lines ? (uint32_t)lines : 1u;
    put_le<uint32_t>(payload, count);
    Bytes one;
    const bool from_gpu = hot.jpeg_lines_dir && hot.jpeg_lines_data && hot.jpeg_n_lines == count;
    for (uint32_t i = 0; i < count; ++i) {
      const size_t start = (size_t)2048 * i;
      const size_t width = lines == 0 ? L : (i + 1 != count ? 2048 : L - start);
      one.clear();
      if (from_gpu) {  // entropy-coded on the GPU: headers, 0xFF stuffing and the end marker are added here
        const uint32_t* d = hot.jpeg_lines_dir + 4 * (size_t)i;
        BaselineJpeg::wrap_bits(hot.jpeg_lines_data + d[0], d[1], (int)width, 1, prm.jpeg_quality, one);
      } else
      BaselineJpeg::encode_rgb(hot.bgr + 3 * start, (int)width, 1, prm.jpeg_quality, one);
      put_le<uint32_t>(payload, (uint32_t)one.size());
      payload.insert(payload.end(), one.begin(), one.end());
    }
    src = payload.data();
    src_len = payload.size();
  }
}
}  // namespace

void frame_header_bytes(const pcc_hot_result& hot, const pcc_params& prm, Bytes& out) { write_frame_header(hot, prm, out); }
void colour_stream_source(const pcc_hot_result& hot, const pcc_params& prm, Bytes& payload, const uint8_t*& src, size_t& src_len) {
  colour_payload(hot, prm, payload, src, src_len);
}

// Up to sixteen frames at a time (four by default): every range-coder stage codes the streams of all frames in one call
// (StaticRangeCoder::encode_many): same bytes, a fraction of the time per frame.
void entropy_encode_frames(int n, const pcc_hot_result* const hot[], const pcc_params* const prm[], Bytes* const out[],
                           uint64_t* const perf[], double* const times_us[]) {
  typedef std::chrono::steady_clock Clock;
  auto us_since = [](Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); };
  const Clock::time_point t_begin = Clock::now();
  constexpr int kMax = StaticRangeCoder::kMaxStreams;
  if (n > kMax) n = kMax;
  double t_occ = 0, t_jpeg = 0, t_col = 0;
  bool use[kMax];
  const uint8_t* src[kMax];
  size_t len[kMax];
  uint64_t got[kMax];
  const uint32_t* cnt[kMax];  // symbol counts of the stage's input where known, else null
  for (int i = 0; i < kMax; ++i) cnt[i] = nullptr;
  // a range-coder stage over the frames that take part in it
  auto stage = [&]() {
    const uint8_t* s2[kMax];
    size_t l2[kMax], g2[kMax];
    Bytes* o2[kMax];
    const uint32_t* c2[kMax];
    int idx[kMax], m = 0;
    for (int i = 0; i < n; ++i) {
      got[i] = 0;
      if (use[i]) { s2[m] = src[i]; l2[m] = len[i]; o2[m] = out[i]; c2[m] = cnt[i]; idx[m] = i; ++m; }
    }
    if (m) StaticRangeCoder::encode_many(m, s2, l2, o2, g2, c2);
    for (int k = 0; k < m; ++k) got[idx[k]] = g2[k];
  };

  // One or two frames leave coder slots free (four streams share a loop, and one coder alone uses a fraction of a
  // core): their colour streams ride along with their occupancy streams instead of waiting for them -- a lone frame's
  // stage 3.3 -> 2.9 ms.  The colour stream is coded into a buffer of its own and appended where it belongs.
  const bool ride_along = 2 * n <= StaticRangeCoder::kInterleave;
  Bytes payload[kMax], colour_rc[kMax];
  uint64_t colour_got[kMax] = {};
  size_t colour_len[kMax] = {};
  const uint8_t* colour_src[kMax] = {};
  Clock::time_point t0 = Clock::now();
  if (ride_along) {
    for (int i = 0; i < n; ++i)
      if (prm[i]->do_color_encoding != 0) colour_payload(*hot[i], *prm[i], payload[i], colour_src[i], colour_len[i]);
    t_jpeg = us_since(t0);
  }

  // --- header, occupancy bytes (impl.hpp:1692-1697) ---
  for (int i = 0; i < n; ++i) {
    write_frame_header(*hot[i], *prm[i], *out[i]);
    put_le<uint64_t>(*out[i], hot[i]->n_branches);
    use[i] = true; src[i] = hot[i]->occupancy; len[i] = (size_t)hot[i]->n_branches;
    cnt[i] = hot[i]->occupancy_histogram;
  }
  t0 = Clock::now();
  if (ride_along) {
    const uint8_t* s2[kMax];
    size_t l2[kMax], g2[kMax];
    Bytes* o2[kMax];
    const uint32_t* c2[kMax];
    int m = 0, col_slot[kMax];
    for (int i = 0; i < n; ++i) { s2[m] = hot[i]->occupancy; l2[m] = (size_t)hot[i]->n_branches; o2[m] = out[i]; c2[m] = hot[i]->occupancy_histogram; ++m; }
    for (int i = 0; i < n; ++i) {
      col_slot[i] = -1;
      if (prm[i]->do_color_encoding != 0) {
        col_slot[i] = m; s2[m] = colour_src[i]; l2[m] = colour_len[i]; o2[m] = &colour_rc[i]; c2[m] = nullptr; ++m;
      }
    }
    StaticRangeCoder::encode_many(m, s2, l2, o2, g2, c2);
    for (int i = 0; i < n; ++i) { got[i] = g2[i]; if (col_slot[i] >= 0) colour_got[i] = g2[col_slot[i]]; }
  } else {
    stage();
  }
  t_occ = us_since(t0);
  for (int i = 0; i < n; ++i) cnt[i] = nullptr;
  for (int i = 0; i < n; ++i) perf[i][0] = got[i];

  // --- centroid bytes (impl.hpp:1700-1710) ---
  for (int i = 0; i < n; ++i) {
    use[i] = prm[i]->do_voxel_centroid != 0;
    if (use[i]) {
      put_le<uint32_t>(*out[i], (uint32_t)(3 * hot[i]->n_leaves));
      src[i] = hot[i]->centroid; len[i] = (size_t)(3 * hot[i]->n_leaves);
    }
  }
  stage();
  for (int i = 0; i < n; ++i) perf[i][1] = got[i];

  // --- colour (impl.hpp:1713-1723) ---
  if (ride_along) {  // coded above, next to the occupancy bytes
    for (int i = 0; i < n; ++i) {
      got[i] = 0;
      if (prm[i]->do_color_encoding != 0) {
        put_le<uint64_t>(*out[i], (uint64_t)colour_len[i]);
        out[i]->insert(out[i]->end(), colour_rc[i].begin(), colour_rc[i].end());
        got[i] = colour_got[i];
      }
    }
  } else {
    t0 = Clock::now();
    for (int i = 0; i < n; ++i) {
      use[i] = prm[i]->do_color_encoding != 0;
      if (use[i]) {
        colour_payload(*hot[i], *prm[i], payload[i], src[i], len[i]);
        put_le<uint64_t>(*out[i], (uint64_t)len[i]);
      }
    }
    t_jpeg = us_since(t0);
    t0 = Clock::now();
    stage();
    t_col = us_since(t0);
  }
  for (int i = 0; i < n; ++i) perf[i][2] = got[i];

  const double total = us_since(t_begin);
  for (int i = 0; i < n; ++i)
    if (times_us && times_us[i]) {  // per frame: a stage shared by several frames counts in equal parts
      times_us[i][0] = t_occ / n; times_us[i][1] = t_jpeg / n; times_us[i][2] = t_col / n; times_us[i][3] = total / n;
    }
}

void entropy_encode_frame(const pcc_hot_result& hot, const pcc_params& prm, Bytes& out, uint64_t perf[3],
                          double* times_us) {
  const pcc_hot_result* h[1] = {&hot};
  const pcc_params* p[1] = {&prm};
  Bytes* o[1] = {&out};
  uint64_t* pf[1] = {perf};
  double* t[1] = {times_us};
  entropy_encode_frames(1, h, p, o, pf, t);
}

// =============================================================================================
// decoder
// =============================================================================================
namespace {
struct Reader {
  const uint8_t* p;
  size_t len, pos;
  template <typename T>
  bool get(T& v) {
    if (pos + sizeof(T) > len) return false;
    memcpy(&v, p + pos, sizeof(T));
    pos += sizeof(T);
    return true;
  }
  bool sync(const char* id) {  // impl.hpp:1660-1676
    const size_t n = strlen(id);
    size_t k = 0;
    while (k < n) {
      if (pos >= len) return false;
      const char c = (char)p[pos++];
      if (c != id[k++]) k = (id[0] == c) ? 1 : 0;
    }
    return true;
  }
};
}  // namespace


// How many symbols the range-coded vector at `in` (its 257-entry cumulative table, then the coded bytes) can hold at most --
// asked BEFORE the output is allocated, because the count comes from a header a hostile stream controls.  The coder starts
// with a range of 2^32, every symbol narrows it by at least the symbol's share of the table and every coded byte widens it
// by 2^8: n symbols of at least b bits each need n * b <= 8 * coded bytes + 32.  b comes from the stream's OWN table (the
// widest symbol over the total), which is read before anything else anyway; a table whose total is 0 or beyond 2^16 (the
// encoder halves its table below that), that is not non-decreasing, or in which one symbol takes everything (every symbol
// has width 1 at least: a symbol then costs 0.0056 bits or more, 1424 symbols per byte at most) is not the encoder's.
// (Round 3's guard allowed 64 symbols per byte whatever the table: a frame of coincident points -- 64 k voxels, every
// centroid byte the same: 192 k symbols in 1.2 KB -- is valid and was refused; round 4's allowed 1424 per byte whatever the
// table.  With the table's own bound a stream has to BE that skewed to claim that many symbols.)
static bool plausible_symbol_count(uint64_t n, const uint8_t* in, size_t in_len) {
  if (n == 0) return true;
  uint32_t freq[257];
  if (in_len < sizeof(freq) + 4) return false;
  memcpy(freq, in, sizeof(freq));
  const uint32_t total = freq[256];
  if (total == 0 || total > 65536u || freq[0] != 0) return false;
  uint32_t widest = 0;
  for (int k = 0; k < 256; ++k) {
    if (freq[k + 1] < freq[k]) return false;
    widest = std::max(widest, freq[k + 1] - freq[k]);
  }
  if (widest >= total) return false;
  const double bits = -log2((double)widest / (double)total);
  const double room = 8.0 * (double)(in_len - sizeof(freq)) + 64.0;
  return (double)n <= room / bits + 1.0;
}

int decode_frame_streams(const uint8_t* stream, size_t len, pcc_cloud& info, FrameStreams& fs, bool colours_too,
                         const std::function<void()>& after_occupancy) {
  DecodeTrace tr;
  memset(&info, 0, sizeof(info));
  Reader r{stream, len, 0};
  if (!r.sync(kV2Id) || !r.sync(kV1Id)) return PCC_ERR_STREAM;
  pcc_params& p = info.params;
  uint8_t i_frame = 0, vg = 0, with_color = 0, cbits = 0, u8 = 0;
  uint64_t count = 0;
  if (!r.get(p.frame_id) || !r.get(i_frame) || !i_frame) return PCC_ERR_STREAM;
  if (!r.get(vg) || !r.get(with_color) || !r.get(count) || !r.get(p.octree_resolution) || !r.get(cbits) ||
      !r.get(p.point_resolution))
    return PCC_ERR_STREAM;
  for (int i = 0; i < 6; ++i)
    if (!r.get(info.bbox[i])) return PCC_ERR_STREAM;
  p.color_bit_resolution = cbits;
  p.do_color_encoding = with_color;
  uint32_t cct = 0;
  if (!r.get(u8)) return PCC_ERR_STREAM;
  p.do_voxel_centroid = u8;
  if (!r.get(u8)) return PCC_ERR_STREAM;
  p.do_connectivity = u8;
  if (!r.get(u8)) return PCC_ERR_STREAM;
  p.create_scalable = u8;
  if (!r.get(cct) || !r.get(p.macroblock_size) || !r.get(u8)) return PCC_ERR_STREAM;
  p.color_coding_type = (int32_t)cct;
  p.do_icp_color_offset = u8;
  const double res = p.octree_resolution;
  if (!(res > 0.0)) return PCC_ERR_STREAM;
  // do_voxel_grid_enDecoding_ false: `count` counts points, not voxels, and a point-detail tail follows the streams
  // (impl.hpp:1728-1757) -- a frame this decoder would misread as a voxel-grid one; the encoder never writes it
  if (!vg) return PCC_ERR_UNSUPPORTED;

  // defineBoundingBox + getKeyBitSize
  {
    const float mv = FLT_EPSILON;
    unsigned mk = 2;
    for (int a = 0; a < 3; ++a) {
      const double k = ceil((info.bbox[3 + a] - info.bbox[a] - mv) / res);
      if (!(k >= 0.0 && k < 4294967296.0)) return PCC_ERR_STREAM;
      mk = std::max(mk, (unsigned)k);
    }
    unsigned d = (unsigned)ceil(log2((double)mk) - mv);
    d = std::min(d, 32u);
    info.depth = d;
    const double side = (double)(1ull << d) * res;
    for (int a = 0; a < 3; ++a) {
      const double over = (side - (info.bbox[3 + a] - info.bbox[a])) / 2.0;
      if (over > mv) { info.bbox[a] -= over; info.bbox[3 + a] += over; }
    }
  }

  uint64_t occ_n = 0;
  // (a leaf opens at most one branch node per level: a count beyond that is not a voxel-grid frame's; 2^40 voxels: no overflow below)
  if (count >= (1ull << 40)) return PCC_ERR_STREAM;
  if (!r.get(occ_n) || occ_n > count * (uint64_t)std::max(info.depth, 1u) || !plausible_symbol_count(occ_n, r.p + r.pos, r.len - r.pos)) return PCC_ERR_STREAM;
  // ... and a branch byte opens at most eight voxels: with occ_n tied to the coded bytes above, this ties `count` -- which sizes
  // everything allocated from here on -- to the stream's size too
  if (count > 8u * occ_n) return PCC_ERR_STREAM;
  Bytes& occ = fs.occ;
  occ.resize((size_t)occ_n);
  size_t used = StaticRangeCoder::decode(r.p + r.pos, r.len - r.pos, occ.data(), occ.size());
  if (!used) return PCC_ERR_STREAM;
  r.pos += used;
  tr.lap("occupancy range decoder");
  fs.count = count;
  fs.with_color = with_color != 0;
  if (after_occupancy) after_occupancy();
  Bytes& cen = fs.cen;
  cen.clear();
  if (p.do_voxel_centroid) {
    uint32_t n = 0;
    if (!r.get(n) || (uint64_t)n != 3u * count || !plausible_symbol_count(n, r.p + r.pos, r.len - r.pos)) return PCC_ERR_STREAM;   // (three bytes per voxel, impl.hpp:1706)
    cen.resize(n);
    used = StaticRangeCoder::decode(r.p + r.pos, r.len - r.pos, cen.data(), cen.size());
    if (!used) return PCC_ERR_STREAM;
    r.pos += used;
  }
  Bytes& col = fs.col;
  col.clear();
  fs.payload.clear();
  if (with_color) {
    uint64_t n = 0;
    if (!r.get(n) || !plausible_symbol_count(n, r.p + r.pos, r.len - r.pos)) return PCC_ERR_STREAM;
    Bytes& payload = fs.payload;
    payload.resize((size_t)n);
    used = StaticRangeCoder::decode(r.p + r.pos, r.len - r.pos, payload.data(), payload.size());
    if (!used) return PCC_ERR_STREAM;
    r.pos += used;
    tr.lap("colour range decoder");
    if (!colours_too && cct == 1) {
      // the caller takes the JPEG from fs.payload (the GPU decoder: inverse DCT, upsampling and un-snaking on the device)
    } else if (cct == 1) {  // decodeJPEGSnake (jpegcc.h:228-242)
      Bytes img;
      int w = 0, h = 0;
      const bool jpeg_ok = BaselineJpeg::decode_rgb(payload.data(), payload.size(), img, w, h, count + 4096u);  // 256 x (count / 256 + 1)
      tr.lap("jpeg decoder");
      if (jpeg_ok && w % 8 == 0) {
        col.resize(img.size());
        const uint64_t npix = (uint64_t)w * (uint64_t)h;  // < 2^32 (two 16-bit fields of the frame header)
        // the snake walks the image in runs of eight pixels of one row, forwards or backwards (snake.h:46-71; w is a
        // multiple of 8): one position and one direction per run instead of a closed form with divisions per pixel
        for (uint64_t i = 0; i < npix; i += 8) {
          const uint32_t p0 = snake_position((uint32_t)i, (uint32_t)w, (uint32_t)h);
          const uint32_t p1 = snake_position((uint32_t)i + 1u, (uint32_t)w, (uint32_t)h);
          if (p1 == p0 + 1u) {
            memcpy(&col[3 * i], &img[3 * (size_t)p0], 24);
          } else {
            for (uint32_t c = 0; c < 8; ++c) {
              const size_t px = (size_t)p0 - c;
              col[3 * (i + c)] = img[3 * px]; col[3 * (i + c) + 1] = img[3 * px + 1]; col[3 * (i + c) + 2] = img[3 * px + 2];
            }
          }
        }
      }
    } else if (cct == 2) {  // decodeJPEGLines (jpegcc.h:319-344)
      Reader lr{payload.data(), payload.size(), 0};
      uint32_t lines = 0;
      lr.get(lines);
      for (uint32_t i = 0; i < lines; ++i) {
        uint32_t sz = 0;
        if (!lr.get(sz) || lr.pos + sz > lr.len) break;
        Bytes img;
        int w = 0, h = 0;
        // (a strip holds 2048 voxels, the last one up to 4095; the reference's own encoder writes a lone strip 2048 wide
        // whatever the voxel count, jpegcc.h:256-275)
        if (BaselineJpeg::decode_rgb(lr.p + lr.pos, sz, img, w, h, 4096u)) col.insert(col.end(), img.begin(), img.end());
        lr.pos += sz;
      }
    } else {
      col.swap(payload);
    }
  }
  tr.lap("colours in voxel order");
  info.consumed = r.pos;
  fs.count = count;
  fs.with_color = with_color != 0;
  fs.cct = cct;
  return PCC_OK;
}

// The leaf parents of the depth-first occupancy stream: a node of level D-1 has its voxels as children, so everything
// a voxel needs is its parent's key, its parent's byte and how many voxels came before.  The walk is the sequential
// part of deserializeTree (the level of a byte depends on every byte before it); it visits branch nodes only.
namespace {
template <typename PrefixT>
int walk_leaf_parents_t(const Bytes& occ, unsigned D, uint64_t count, LeafParents& lp) {
  constexpr bool kWide = sizeof(PrefixT) > 8;
  uint8_t rem[40];
  int sp = 0;
  size_t op = 0;
  PrefixT prefix = 0;
  uint64_t leaves = 0;
  rem[0] = occ[op++];
  while (sp >= 0) {
    if ((unsigned)sp == D - 1) {
      lp.prefix.push_back((uint64_t)prefix & 0x7fffffffffffffffull);
      if (kWide) lp.prefix_hi.push_back((uint32_t)(prefix >> 63));
      lp.bits.push_back(rem[sp]);
      lp.first.push_back((uint32_t)leaves);
      leaves += (uint64_t)__builtin_popcount(rem[sp]);
      if (leaves > count) return PCC_ERR_STREAM;
      --sp; prefix >>= 3;
      continue;
    }
    if (!rem[sp]) { --sp; prefix >>= 3; continue; }
    const int c = __builtin_ctz(rem[sp]);
    rem[sp] = (uint8_t)(rem[sp] & (rem[sp] - 1));
    if (op >= occ.size()) return PCC_ERR_STREAM;
    prefix = (prefix << 3) | (PrefixT)c;
    rem[++sp] = occ[op++];
  }
  return leaves == count ? PCC_OK : PCC_ERR_STREAM;
}
}  // namespace

int walk_leaf_parents(const Bytes& occ, unsigned D, uint64_t count, LeafParents& lp) {
  lp.prefix.clear(); lp.prefix_hi.clear(); lp.bits.clear(); lp.first.clear();
  if (D == 0 || D > 31 || occ.empty()) return count == 0 ? PCC_OK : PCC_ERR_STREAM;
  if (count >= (1ull << 32)) return PCC_ERR_STREAM;  // `first` counts voxels in 32 bits (and B < 2^32 bytes cannot hold more)
  {  // a level-(D-1) node holds at least one voxel and is one byte of the stream
    const size_t most = (size_t)std::min<uint64_t>(count, occ.size());
    lp.prefix.reserve(most); lp.bits.reserve(most); lp.first.reserve(most);
    if (D > 22) lp.prefix_hi.reserve(most);
  }
  // the key of a level-(D-1) node has 3 (D - 1) bits: one word up to 22 levels, two beyond
  return D > 22 ? walk_leaf_parents_t<unsigned __int128>(occ, D, count, lp) : walk_leaf_parents_t<uint64_t>(occ, D, count, lp);
}


namespace {
// deserializeTreeCallback (impl.hpp:1584-1653), position part: centre of the voxel (impl.hpp:1630-1632) or, with a
// centroid stream, lower corner + the decoded offset (ptv2.h:115-117)
inline void voxel_position(pcc_point_xyzrgb& np, const uint32_t key[3], double res, const double* bbox, const uint8_t* cen3) {
  float xyz[3];
  for (int a = 0; a < 3; ++a) {
    if (cen3) {
      const double lc = (double)key[a] * res + bbox[a];
      xyz[a] = (float)(lc + (float)cen3[a] * 0.001f);
    } else {
      xyz[a] = (float)(((double)key[a] + 0.5) * res + bbox[a]);
    }
  }
  np.x = xyz[0]; np.y = xyz[1]; np.z = xyz[2]; np.w = 1.0f;
  np.pad[0] = np.pad[1] = np.pad[2] = 0u;
}

// Positions of all voxels, leaves in Morton order.  Depth <= 21: the walk only visits branch nodes (walk_leaf_parents);
// a parent's key comes out of its 3-bits-per-level path with three bit extractions and its voxels are its set bits in
// ascending order.  Deeper trees (keys beyond 63 path bits): the plain pre-order walk with an explicit stack (Appendix B).
int voxel_positions(const Bytes& occ, const Bytes& cen, bool centroids, unsigned D, double res, const double* bbox, PointVec& points) {
  const size_t count = points.size();
  if (D == 0 || occ.empty()) return count == 0 ? PCC_OK : PCC_ERR_STREAM;
  if (D <= 21) {
    DecodeTrace wt;
    LeafParents lp;
    const int rc = walk_leaf_parents(occ, D, count, lp);
    if (rc != PCC_OK) return rc;
    wt.lap("  walk over the branch nodes");
    const size_t np_ = lp.bits.size();
    for (size_t i = 0; i < np_; ++i) {
      const uint64_t path = lp.prefix[i];  // x-major triples, the level-(D-2) triple lowest
      const uint32_t px = (uint32_t)_pext_u64(path, 0x4924924924924924ull), py = (uint32_t)_pext_u64(path, 0x2492492492492492ull),
                     pz = (uint32_t)_pext_u64(path, 0x9249249249249249ull);
      size_t leaf = lp.first[i];
      for (unsigned bits = lp.bits[i]; bits; bits &= bits - 1u, ++leaf) {
        const unsigned c = (unsigned)__builtin_ctz(bits);
        const uint32_t key[3] = {(px << 1) | ((c >> 2) & 1u), (py << 1) | ((c >> 1) & 1u), (pz << 1) | (c & 1u)};
        if (leaf >= count) return PCC_ERR_STREAM;
        voxel_position(points[leaf], key, res, bbox, (centroids && 3 * leaf + 2 < cen.size()) ? &cen[3 * leaf] : nullptr);
      }
    }
    return PCC_OK;
  }
  size_t leaf = 0, op = 0;
  struct Frame { uint8_t bits; int8_t next; };
  Frame stack[40];
  uint32_t key[3] = {0, 0, 0};
  int sp = 0;
  stack[0] = {occ[op++], 0};
  while (sp >= 0) {
    Frame& f = stack[sp];
    int c = f.next;
    while (c < 8 && !(f.bits & (1u << c))) ++c;
    if (c == 8) {  // popBranch
      --sp;
      key[0] >>= 1; key[1] >>= 1; key[2] >>= 1;
      continue;
    }
    f.next = (int8_t)(c + 1);
    key[0] = (key[0] << 1) | ((c >> 2) & 1);
    key[1] = (key[1] << 1) | ((c >> 1) & 1);
    key[2] = (key[2] << 1) | (c & 1);
    if ((unsigned)(sp + 1) < D) {
      if (op >= occ.size() || sp + 1 >= 40) return PCC_ERR_STREAM;
      stack[++sp] = {occ[op++], 0};
      continue;
    }
    if (leaf >= count) return PCC_ERR_STREAM;
    voxel_position(points[leaf], key, res, bbox, (centroids && 3 * leaf + 2 < cen.size()) ? &cen[3 * leaf] : nullptr);
    ++leaf;
    key[0] >>= 1; key[1] >>= 1; key[2] >>= 1;
  }
  return leaf == count ? PCC_OK : PCC_ERR_STREAM;
}

struct JoinOnExit {  // (an exception on the way -- a corrupt header asking for more memory than there is -- must not leave a thread behind)
  std::thread& t;
  ~JoinOnExit() { if (t.joinable()) t.join(); }
};
}  // namespace

int decode_frame(const uint8_t* stream, size_t len, PointVec& points, pcc_cloud& info) {
  points.clear();
  FrameStreams fs;
  // The three vectors sit one behind the other and a range-coded vector does not say how long it is, so they are decoded
  // in order; but once the occupancy bytes are there, the walk over the tree (positions of all voxels) and the colour side
  // (colour range decoder, JPEG, un-snaking) do not need each other: the walk runs on a second thread meanwhile.  With a
  // centroid stream the positions need the second vector: then everything runs in order.
  std::thread walker;
  JoinOnExit joiner{walker};
  int walk_rc = PCC_OK;
  bool walked = false;
  auto start_walk = [&]() {
    static const bool serial = dev_env("PCC_DECODE_SERIAL") != nullptr;  // developer knob: no second thread
    if (serial || info.params.do_voxel_centroid || fs.count > 8 * (uint64_t)fs.occ.size()) return;
    points.resize((size_t)fs.count);
    try {
      walker = std::thread([&]() {
        DecodeTrace wt;
        try {
          walk_rc = voxel_positions(fs.occ, fs.cen, false, info.depth, info.params.octree_resolution, info.bbox, points);
        } catch (...) {
          walk_rc = PCC_ERR_STREAM;
        }
        wt.lap("(second thread) tree walk");
      });
      walked = true;
    } catch (const std::system_error&) {
      // no thread to be had (process limits): the walk runs on this thread after the colour side
    }
  };
  const int rc = decode_frame_streams(stream, len, info, fs, true, start_walk);
  if (walker.joinable()) walker.join();
  DecodeTrace tr;
  if (rc != PCC_OK) { points.clear(); return rc; }
  const pcc_params& p = info.params;
  const uint64_t count = fs.count;
  // a tree of occ_n branch nodes has at most eight leaves per node: anything else is a corrupt header
  if (count > 8 * (uint64_t)fs.occ.size()) { points.clear(); return PCC_ERR_STREAM; }
  if (!walked) {
    points.resize((size_t)count);
    walk_rc = voxel_positions(fs.occ, fs.cen, p.do_voxel_centroid != 0, info.depth, p.octree_resolution, info.bbox, points);
  }
  if (walk_rc != PCC_OK) { points.clear(); return walk_rc; }
  tr.lap("tree walk + positions (rest)");
  // colour part of deserializeTreeCallback
  const Bytes& col = fs.col;
  const unsigned shift = (fs.cct == 0) ? (unsigned)(8 - p.color_bit_resolution) & 7u : 0u;
  const size_t n = points.size();
  if (fs.with_color) {
    for (size_t leaf = 0; leaf < n; ++leaf) {
      uint32_t c0 = 0, c1 = 0, c2 = 0;
      if (3 * leaf + 2 < col.size()) { c0 = col[3 * leaf]; c1 = col[3 * leaf + 1]; c2 = col[3 * leaf + 2]; }
      c0 = (uint8_t)(c0 << shift); c1 = (uint8_t)(c1 << shift); c2 = (uint8_t)(c2 << shift);
      points[leaf].rgba = c0 | (c1 << 8) | (c2 << 16);
    }
  } else {
    for (size_t leaf = 0; leaf < n; ++leaf) points[leaf].rgba = 0x00FFFFFFu;  // ColorCoding::setDefaultColor
  }
  tr.lap("colours into the cloud");
  info.n = n;
  return PCC_OK;
}

void pack_points_16(uint8_t* dst, const uint8_t* src, size_t n, size_t stride, size_t rgb_offset) {
  size_t i = 0;
#ifdef __AVX2__
  if (stride == 32 && rgb_offset == 16 && (reinterpret_cast<uintptr_t>(dst) & 31) == 0) {
    // dwords 0,1,2 (x,y,z) and 4 (colour) of every point; two points per 32-byte streaming store
    const __m256i pick = _mm256_setr_epi32(0, 1, 2, 4, 0, 1, 2, 4);
    for (; i + 4 <= n; i += 4) {
      const __m256i a = _mm256_permutevar8x32_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32 * i)), pick);
      const __m256i b = _mm256_permutevar8x32_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32 * i + 32)), pick);
      const __m256i c = _mm256_permutevar8x32_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32 * i + 64)), pick);
      const __m256i d = _mm256_permutevar8x32_epi32(_mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + 32 * i + 96)), pick);
      _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 16 * i), _mm256_permute2x128_si256(a, b, 0x20));
      _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + 16 * i + 32), _mm256_permute2x128_si256(c, d, 0x20));
    }
    _mm_sfence();
  }
#endif
  for (; i < n; ++i) {
    memcpy(dst + 16 * i, src + stride * i, 12);
    memcpy(dst + 16 * i + 12, src + stride * i + rgb_offset, 4);
  }
}

}  // namespace pcc
