/* oracle_util.h -- CPU ORACLE helper (test infrastructure): growable byte buffer. */
#ifndef PCC_ORACLE_UTIL_H
#define PCC_ORACLE_UTIL_H
#include "pcc_oracle.h"
#include <stdlib.h>
#include <string.h>

static inline void buf_reserve(pcco_buf *b, size_t extra) {
  if (b->len + extra <= b->cap) return;
  size_t nc = b->cap ? b->cap * 2 : 4096;
  while (nc < b->len + extra) nc *= 2;
  b->data = (uint8_t *)realloc(b->data, nc);
  b->cap = nc;
}
static inline void buf_put(pcco_buf *b, uint8_t v) {
  buf_reserve(b, 1);
  b->data[b->len++] = v;
}
static inline void buf_write(pcco_buf *b, const void *p, size_t n) {
  buf_reserve(b, n);
  if (n) memcpy(b->data + b->len, p, n);
  b->len += n;
}
#endif
