/*
 * rangecoder_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Restatement of pcl::StaticRangeCoder::encodeCharVectorToStream /
 * decodeStreamToCharVector (PCL 1.10.0,
 * io/include/pcl/compression/impl/entropy_range_coder.hpp) as used by the
 * reference at impl.hpp:1694,1706,1719 (encode) and impl.hpp:1778,1789,1798
 * (decode).  PCL is not vendored in /root/reference: "parity unpinned"
 * (SURVEY.md section 8 row P7).
 *
 * The CHAR-vector variant (the only one the reference calls) is the 32-bit
 * Subbotin carry-less coder: `DWord freq[257]` with DWord = uint32_t (hence the
 * 1028-byte table), `DWord low, range`, top = 1<<24, bottom = 1<<16,
 * maxRange = 1<<16, output byte = low >> 24, 4 flush bytes, and the cumulative
 * table is halved until freq[256] < 2^16.  The 64-bit constants (1<<56, 1<<48,
 * 8 flush bytes) that SURVEY.md row P7 quotes belong to the INT-vector variant
 * (encodeIntVectorToStream), which this path never calls; the survey's own
 * probe ("the -int(low) form cannot round-trip on a 64-bit state") is what one
 * expects when the two variants are mixed.  DESIGN.md (c) records the correction.
 */
#include "oracle_util.h"

#define RC_TOP       ((uint32_t)1 << 24)
#define RC_BOTTOM    ((uint32_t)1 << 16)
#define RC_MAX_RANGE ((uint32_t)1 << 16)

static void build_freq(const uint8_t *in, size_t n, uint32_t freq[257]) {
  uint64_t hist[257]; /* "uint64_t FreqHist[257]" */
  memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < n; i++) hist[(size_t)in[i] + 1]++;
  freq[0] = 0;
  for (int f = 1; f <= 256; f++) {
    freq[f] = freq[f - 1] + (uint32_t)hist[f];
    if (freq[f] <= freq[f - 1]) freq[f] = freq[f - 1] + 1;
  }
  /* "rescale if numerical limits are reached" */
  while (freq[256] >= RC_MAX_RANGE) {
    for (int f = 1; f <= 256; f++) {
      freq[f] /= 2;
      if (freq[f] <= freq[f - 1]) freq[f] = freq[f - 1] + 1;
    }
  }
}

size_t pcco_rc_encode(const uint8_t *in, size_t n, pcco_buf *out) {
  uint32_t freq[257];
  size_t start = out->len;
  build_freq(in, n, freq);
  buf_write(out, freq, sizeof(freq)); /* raw little-endian table, 1028 B */

  uint32_t low = 0, range = (uint32_t)-1;
  for (size_t i = 0; i < n; i++) {
    uint8_t ch = in[i];
    /* low += freq[ch] * (range /= freq[256]); range *= freq[ch+1] - freq[ch]; */
    range /= freq[256];
    low += freq[ch] * range;
    range *= freq[ch + 1] - freq[ch];
    /* range = -int(low) & (bottom - 1) */
    while ((low ^ (low + range)) < RC_TOP ||
           ((range < RC_BOTTOM) && ((range = (0u - low) & (RC_BOTTOM - 1)), 1))) {
      buf_put(out, (uint8_t)(low >> 24));
      range <<= 8;
      low <<= 8;
    }
  }
  for (int i = 0; i < 4; i++) {
    buf_put(out, (uint8_t)(low >> 24));
    low <<= 8;
  }
  return out->len - start;
}

size_t pcco_rc_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t n) {
  uint32_t freq[257];
  size_t pos = 0;
  if (in_len < sizeof(freq) + 4) return 0;
  memcpy(freq, in, sizeof(freq));
  pos += sizeof(freq);

  uint32_t code = 0, low = 0, range = (uint32_t)-1;
  for (int i = 0; i < 4; i++) code = (code << 8) | in[pos++];

  for (size_t i = 0; i < n; i++) {
    uint8_t symbol = 0, ssize = 128;
    range /= freq[256];
    uint32_t count = (code - low) / range;
    while (ssize > 0) {
      if (freq[symbol + ssize] <= count) symbol = (uint8_t)(symbol + ssize);
      ssize = (uint8_t)(ssize / 2);
    }
    out[i] = symbol;
    low += freq[symbol] * range;
    range *= freq[symbol + 1] - freq[symbol];
    while ((low ^ (low + range)) < RC_TOP ||
           ((range < RC_BOTTOM) && ((range = (0u - low) & (RC_BOTTOM - 1)), 1))) {
      /* an istream read past EOF leaves the char untouched in PCL; streams
       * produced by the encoder never run out, so 0 is fed here */
      uint8_t ch = pos < in_len ? in[pos] : 0;
      pos++;
      code = (code << 8) | ch;
      range <<= 8;
      low <<= 8;
    }
  }
  return pos;
}

void pcco_buf_free(pcco_buf *b) {
  free(b->data);
  b->data = NULL;
  b->len = b->cap = 0;
}
