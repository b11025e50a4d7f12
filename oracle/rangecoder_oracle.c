/*
 * rangecoder_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Restatement of pcl::StaticRangeCoder (PCL 1.10.0,
 * io/include/pcl/compression/impl/entropy_range_coder.hpp) as used by the
 * reference at impl.hpp:1694,1706,1719 (encode) and impl.hpp:1778,1789,1798
 * (decode).  PCL is not vendored in /root/reference: "parity unpinned"
 * (SURVEY.md section 8 row P7).  Order-0 static model, Subbotin-style
 * carry-less range coder on a 64-bit state.
 */
#include "oracle_util.h"

/* Shared by encoder and decoder: the renormalisation predicate.
 * PCL text:  while ((low ^ (low + range)) < top ||
 *                   ((range < bottom) && ((range = -low & (bottom - 1)), 1)))
 * with low/range uint64_t (full 64-bit negation, SURVEY.md P7 note). */
#define RC_TOP    ((uint64_t)1 << 56)
#define RC_BOTTOM ((uint64_t)1 << 48)

static void build_freq(const uint8_t *in, size_t n, uint32_t freq[257]) {
  uint64_t hist[257];
  memset(hist, 0, sizeof(hist));
  for (size_t i = 0; i < n; i++) hist[(size_t)in[i] + 1]++;
  freq[0] = 0;
  for (int f = 1; f <= 256; f++) {
    freq[f] = freq[f - 1] + (uint32_t)hist[f];
    if (freq[f] <= freq[f - 1]) freq[f] = freq[f - 1] + 1;
  }
  /* "rescale if numerical limits are reached": freq is 32-bit so
   * freq[256] >= 2^48 can never hold; kept for literalness. */
  while ((uint64_t)freq[256] >= RC_BOTTOM) {
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

  uint64_t low = 0, range = (uint64_t)-1;
  for (size_t i = 0; i < n; i++) {
    uint8_t ch = in[i];
    range /= freq[256];
    low += (uint64_t)freq[ch] * range;
    range *= (uint64_t)(freq[ch + 1] - freq[ch]);
    while ((low ^ (low + range)) < RC_TOP ||
           ((range < RC_BOTTOM) && ((range = (0 - low) & (RC_BOTTOM - 1)), 1))) {
      buf_put(out, (uint8_t)(low >> 56));
      range <<= 8;
      low <<= 8;
    }
  }
  for (int i = 0; i < 8; i++) {
    buf_put(out, (uint8_t)(low >> 56));
    low <<= 8;
  }
  return out->len - start;
}

size_t pcco_rc_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t n) {
  uint32_t freq[257];
  size_t pos = 0;
  if (in_len < sizeof(freq) + 8) return 0;
  memcpy(freq, in, sizeof(freq));
  pos += sizeof(freq);

  uint64_t code = 0, low = 0, range = (uint64_t)-1;
  for (int i = 0; i < 8; i++) code = (code << 8) | in[pos++];

  for (size_t i = 0; i < n; i++) {
    uint8_t symbol = 0, ssize = 128;
    range /= freq[256];
    uint64_t count = (code - low) / range;
    while (ssize > 0) {
      if ((uint64_t)freq[symbol + ssize] <= count) symbol = (uint8_t)(symbol + ssize);
      ssize = (uint8_t)(ssize / 2);
    }
    out[i] = symbol;
    low += (uint64_t)freq[symbol] * range;
    range *= (uint64_t)(freq[symbol + 1] - freq[symbol]);
    while ((low ^ (low + range)) < RC_TOP ||
           ((range < RC_BOTTOM) && ((range = (0 - low) & (RC_BOTTOM - 1)), 1))) {
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
