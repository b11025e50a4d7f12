/*
 * jpeg_oracle.c -- CPU ORACLE (test infrastructure, not product code).
 *
 * Restatement of what libjpeg-turbo computes for the reference's JPEG stage:
 *   encode: JPEGWriter::writeJPEG  (jpeg_io.hpp:211-330): JCS_RGB input,
 *           jpeg_set_defaults + jpeg_set_quality(q, TRUE)  => baseline,
 *           YCbCr 4:2:0, Annex-K tables, JFIF header, islow FDCT.
 *   decode: JPEGReader::readJPEG   (jpeg_io.hpp:90-192): library defaults
 *           => islow IDCT, fancy (triangle) h2v2 upsampling, JCS_RGB out.
 *
 * libjpeg-turbo is a third-party dependency of the reference
 * (CMakeLists.txt:121,167), not vendored.  PINNED against libjpeg-turbo
 * output through Pillow's bundled copy: tests/golden/jpeg_*.bin, generator
 * tests/golden/make_jpeg_golden.py.
 */
#include "oracle_util.h"
#include <stdio.h>

/* ---------- tables (ITU-T T.81 Annex K, as in libjpeg jcparam.c / jstdhuff.c) ---------- */

static const uint8_t std_luma_q[64] = {
  16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55,
  14, 13, 16, 24, 40, 57, 69, 56, 14, 17, 22, 29, 51, 87, 80, 62,
  18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
  49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
static const uint8_t std_chroma_q[64] = {
  17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
  24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
  99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
static const uint8_t zigzag[64] = {
  0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5,
  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
  35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
  58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

static const uint8_t bits_dc_luma[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
static const uint8_t val_dc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t bits_dc_chroma[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t bits_ac_luma[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
static const uint8_t val_ac_luma[162] = {
  0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07,
  0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0,
  0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28,
  0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49,
  0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
  0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89,
  0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7,
  0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5,
  0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2,
  0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
  0xf9, 0xfa};
static const uint8_t bits_ac_chroma[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t val_ac_chroma[162] = {
  0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71,
  0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0,
  0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26,
  0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48,
  0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68,
  0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87,
  0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5,
  0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3,
  0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda,
  0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8,
  0xf9, 0xfa};

/* jfdctint.c / jidctint.c constants: CONST_BITS 13, PASS1_BITS 2 */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + ((int32_t)1 << ((n)-1))) >> (n))

/* ---------- encoder ---------- */

typedef struct {
  uint16_t code[256];
  uint8_t size[256];
} huff_enc;

static void make_enc_table(const uint8_t bits[17], const uint8_t *vals, huff_enc *t) {
  /* jchuff.c jpeg_make_c_derived_tbl: canonical code assignment */
  memset(t, 0, sizeof(*t));
  unsigned code = 0;
  int k = 0;
  for (int l = 1; l <= 16; l++) {
    for (int i = 0; i < bits[l]; i++) {
      t->code[vals[k]] = (uint16_t)code;
      t->size[vals[k]] = (uint8_t)l;
      code++;
      k++;
    }
    code <<= 1;
  }
}

static void quant_table(const uint8_t *basic, int quality, uint16_t out[64]) {
  /* jcparam.c jpeg_quality_scaling + jpeg_add_quant_table(force_baseline=TRUE) */
  if (quality <= 0) quality = 1;
  if (quality > 100) quality = 100;
  int scale = quality < 50 ? 5000 / quality : 200 - quality * 2;
  for (int i = 0; i < 64; i++) {
    long t = ((long)basic[i] * scale + 50L) / 100L;
    if (t <= 0) t = 1;
    if (t > 32767) t = 32767;
    if (t > 255) t = 255;
    out[i] = (uint16_t)t;
  }
}

static void fdct_islow(int32_t *d) {
  int32_t tmp0, tmp1, tmp2, tmp3, tmp4, tmp5, tmp6, tmp7, tmp10, tmp11, tmp12, tmp13;
  int32_t z1, z2, z3, z4, z5;
  int32_t *p = d;
  for (int ctr = 0; ctr < 8; ctr++, p += 8) { /* pass 1: rows */
    tmp0 = p[0] + p[7]; tmp7 = p[0] - p[7];
    tmp1 = p[1] + p[6]; tmp6 = p[1] - p[6];
    tmp2 = p[2] + p[5]; tmp5 = p[2] - p[5];
    tmp3 = p[3] + p[4]; tmp4 = p[3] - p[4];
    tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
    p[0] = (tmp10 + tmp11) << PASS1_BITS;
    p[4] = (tmp10 - tmp11) << PASS1_BITS;
    z1 = (tmp12 + tmp13) * FIX_0_541196100;
    p[2] = DESCALE(z1 + tmp13 * FIX_0_765366865, CONST_BITS - PASS1_BITS);
    p[6] = DESCALE(z1 + tmp12 * (-FIX_1_847759065), CONST_BITS - PASS1_BITS);
    z1 = tmp4 + tmp7; z2 = tmp5 + tmp6; z3 = tmp4 + tmp6; z4 = tmp5 + tmp7;
    z5 = (z3 + z4) * FIX_1_175875602;
    tmp4 *= FIX_0_298631336; tmp5 *= FIX_2_053119869;
    tmp6 *= FIX_3_072711026; tmp7 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    p[7] = DESCALE(tmp4 + z1 + z3, CONST_BITS - PASS1_BITS);
    p[5] = DESCALE(tmp5 + z2 + z4, CONST_BITS - PASS1_BITS);
    p[3] = DESCALE(tmp6 + z2 + z3, CONST_BITS - PASS1_BITS);
    p[1] = DESCALE(tmp7 + z1 + z4, CONST_BITS - PASS1_BITS);
  }
  p = d;
  for (int ctr = 0; ctr < 8; ctr++, p++) { /* pass 2: columns */
    tmp0 = p[0] + p[56]; tmp7 = p[0] - p[56];
    tmp1 = p[8] + p[48]; tmp6 = p[8] - p[48];
    tmp2 = p[16] + p[40]; tmp5 = p[16] - p[40];
    tmp3 = p[24] + p[32]; tmp4 = p[24] - p[32];
    tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
    p[0] = DESCALE(tmp10 + tmp11, PASS1_BITS);
    p[32] = DESCALE(tmp10 - tmp11, PASS1_BITS);
    z1 = (tmp12 + tmp13) * FIX_0_541196100;
    p[16] = DESCALE(z1 + tmp13 * FIX_0_765366865, CONST_BITS + PASS1_BITS);
    p[48] = DESCALE(z1 + tmp12 * (-FIX_1_847759065), CONST_BITS + PASS1_BITS);
    z1 = tmp4 + tmp7; z2 = tmp5 + tmp6; z3 = tmp4 + tmp6; z4 = tmp5 + tmp7;
    z5 = (z3 + z4) * FIX_1_175875602;
    tmp4 *= FIX_0_298631336; tmp5 *= FIX_2_053119869;
    tmp6 *= FIX_3_072711026; tmp7 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    p[56] = DESCALE(tmp4 + z1 + z3, CONST_BITS + PASS1_BITS);
    p[40] = DESCALE(tmp5 + z2 + z4, CONST_BITS + PASS1_BITS);
    p[24] = DESCALE(tmp6 + z2 + z3, CONST_BITS + PASS1_BITS);
    p[8] = DESCALE(tmp7 + z1 + z4, CONST_BITS + PASS1_BITS);
  }
}

/* sample plane padded to whole blocks (edge replication as libjpeg does it) */
typedef struct {
  int w, h;       /* padded dims (multiples of 8 * sampling) */
  uint8_t *s;
} plane;

/* forward DCT + quantise one 8x8 block at (bx,by) of a plane (jcdctmgr.c forward_DCT) */
static void fdct_quant_block(const plane *pl, int bx, int by, const uint16_t q[64], int16_t out[64]) {
  int32_t ws[64];
  for (int r = 0; r < 8; r++)
    for (int c = 0; c < 8; c++)
      ws[r * 8 + c] = (int32_t)pl->s[(size_t)(by * 8 + r) * pl->w + bx * 8 + c] - 128;
  fdct_islow(ws);
  for (int i = 0; i < 64; i++) {
    int32_t qv = (int32_t)q[i] << 3;
    int32_t t = ws[i];
    if (t < 0) {
      t = -t;
      t += qv >> 1;
      t = (t >= qv) ? t / qv : 0;
      t = -t;
    } else {
      t += qv >> 1;
      t = (t >= qv) ? t / qv : 0;
    }
    out[i] = (int16_t)t;
  }
}

typedef struct {
  pcco_buf *out;
  uint32_t acc;
  int nbits;
} bitw;

static void emit_byte(bitw *b, uint8_t v) {
  buf_put(b->out, v);
  if (v == 0xFF) buf_put(b->out, 0);
}
static void emit_bits(bitw *b, unsigned code, int size) {
  if (size == 0) return;
  b->acc = (b->acc << size) | (code & ((1u << size) - 1));
  b->nbits += size;
  while (b->nbits >= 8) {
    emit_byte(b, (uint8_t)((b->acc >> (b->nbits - 8)) & 0xFF));
    b->nbits -= 8;
  }
}
static void flush_bits(bitw *b) {
  emit_bits(b, 0x7F, 7);
  b->acc = 0;
  b->nbits = 0;
}
static int bitlen(int v) {
  int n = 0;
  while (v) { n++; v >>= 1; }
  return n;
}

/* jchuff.c encode_one_block */
static void encode_block(bitw *b, const int16_t blk[64], int *last_dc, const huff_enc *dc, const huff_enc *ac) {
  int temp = blk[0] - *last_dc, temp2 = temp;
  *last_dc = blk[0];
  if (temp < 0) { temp = -temp; temp2--; }
  int nbits = bitlen(temp);
  emit_bits(b, dc->code[nbits], dc->size[nbits]);
  if (nbits) emit_bits(b, (unsigned)temp2, nbits);
  int r = 0;
  for (int k = 1; k < 64; k++) {
    temp = blk[zigzag[k]];
    if (temp == 0) { r++; continue; }
    while (r > 15) { emit_bits(b, ac->code[0xF0], ac->size[0xF0]); r -= 16; }
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    nbits = bitlen(temp);
    int sym = (r << 4) + nbits;
    emit_bits(b, ac->code[sym], ac->size[sym]);
    emit_bits(b, (unsigned)temp2, nbits);
    r = 0;
  }
  if (r > 0) emit_bits(b, ac->code[0], ac->size[0]);
}

static void put16(pcco_buf *o, unsigned v) { buf_put(o, (uint8_t)(v >> 8)); buf_put(o, (uint8_t)v); }
static void emit_dqt(pcco_buf *o, int id, const uint16_t q[64]) {
  put16(o, 0xFFDB); put16(o, 67); buf_put(o, (uint8_t)id);
  for (int i = 0; i < 64; i++) buf_put(o, (uint8_t)q[zigzag[i]]);
}
static void emit_dht(pcco_buf *o, int tc_th, const uint8_t bits[17], const uint8_t *vals) {
  int n = 0;
  for (int i = 1; i <= 16; i++) n += bits[i];
  put16(o, 0xFFC4); put16(o, (unsigned)(2 + 1 + 16 + n)); buf_put(o, (uint8_t)tc_th);
  for (int i = 1; i <= 16; i++) buf_put(o, bits[i]);
  buf_write(o, vals, (size_t)n);
}

int pcco_jpeg_encode_rgb(const uint8_t *rgb, int w, int h, int quality, pcco_buf *out) {
  if (w <= 0 || h <= 0 || !rgb) return -1;
  uint16_t ql[64], qc[64];
  quant_table(std_luma_q, quality, ql);
  quant_table(std_chroma_q, quality, qc);
  huff_enc dcl, acl, dcc, acc;
  make_enc_table(bits_dc_luma, val_dc, &dcl);
  make_enc_table(bits_ac_luma, val_ac_luma, &acl);
  make_enc_table(bits_dc_chroma, val_dc, &dcc);
  make_enc_table(bits_ac_chroma, val_ac_chroma, &acc);

  /* geometry (jcmaster.c initial_setup / per_scan_setup), 4:2:0 */
  int cw = (w + 1) / 2, ch = (h + 1) / 2;     /* chroma downsampled dims */
  int y_wb = (w + 7) / 8, y_hb = (h + 7) / 8; /* real luma blocks */
  int c_wb = (cw + 7) / 8, c_hb = (ch + 7) / 8;
  int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16;

  /* colour conversion (jccolor.c rgb_ycc_convert), full-res planes of w x h */
  size_t npx = (size_t)w * h;
  uint8_t *Y = (uint8_t *)malloc(npx), *Cb = (uint8_t *)malloc(npx), *Cr = (uint8_t *)malloc(npx);
  for (size_t i = 0; i < npx; i++) {
    int32_t r = rgb[3 * i], g = rgb[3 * i + 1], b = rgb[3 * i + 2];
    Y[i] = (uint8_t)((19595 * r + 38470 * g + 7471 * b + 32768) >> 16);
    Cb[i] = (uint8_t)((-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16);
    Cr[i] = (uint8_t)((32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16);
  }

  /* luma plane: right edge replicated to y_wb*8, bottom replicated to the iMCU height */
  plane py, pcb, pcr;
  py.w = mcus_x * 16; py.h = mcus_y * 16; py.s = (uint8_t *)calloc((size_t)py.w * py.h, 1);
  pcb.w = pcr.w = mcus_x * 8; pcb.h = pcr.h = mcus_y * 8;
  pcb.s = (uint8_t *)calloc((size_t)pcb.w * pcb.h, 1);
  pcr.s = (uint8_t *)calloc((size_t)pcr.w * pcr.h, 1);

  for (int r = 0; r < py.h; r++) {
    int sr = r < h ? r : h - 1; /* jcprepct.c expand_bottom_edge (input rows, then component rows) */
    for (int c = 0; c < py.w; c++) {
      int sc = c < w ? c : w - 1; /* jcsample.c expand_right_edge */
      py.s[(size_t)r * py.w + c] = Y[(size_t)sr * w + sc];
    }
  }
  /* chroma: input padded to an even number of rows (last row replicated) and to
   * c_wb*16 columns (last column replicated); h2v2_downsample with bias 1,2,1,2;
   * then the component's last row is replicated down to the iMCU boundary */
  const uint8_t *src[2] = {Cb, Cr};
  plane *dst[2] = {&pcb, &pcr};
  for (int k = 0; k < 2; k++) {
    for (int r = 0; r < ch; r++) {
      int r0 = 2 * r < h ? 2 * r : h - 1, r1 = 2 * r + 1 < h ? 2 * r + 1 : h - 1;
      for (int c = 0; c < c_wb * 8; c++) {
        int c0 = 2 * c < w ? 2 * c : w - 1, c1 = 2 * c + 1 < w ? 2 * c + 1 : w - 1;
        int bias = (c & 1) ? 2 : 1;
        int v = src[k][(size_t)r0 * w + c0] + src[k][(size_t)r0 * w + c1] +
                src[k][(size_t)r1 * w + c0] + src[k][(size_t)r1 * w + c1] + bias;
        dst[k]->s[(size_t)r * dst[k]->w + c] = (uint8_t)(v >> 2);
      }
    }
    for (int r = ch; r < dst[k]->h; r++)
      memcpy(dst[k]->s + (size_t)r * dst[k]->w, dst[k]->s + (size_t)(ch - 1) * dst[k]->w, (size_t)c_wb * 8);
  }
  free(Y); free(Cb); free(Cr);

  /* markers (jcmarker.c write_file_header / write_frame_header / write_scan_header) */
  put16(out, 0xFFD8);
  put16(out, 0xFFE0); put16(out, 16);
  buf_write(out, "JFIF\0", 5);
  buf_put(out, 1); buf_put(out, 1); buf_put(out, 0); put16(out, 1); put16(out, 1); buf_put(out, 0); buf_put(out, 0);
  emit_dqt(out, 0, ql);
  emit_dqt(out, 1, qc);
  put16(out, 0xFFC0); put16(out, 17); buf_put(out, 8); put16(out, (unsigned)h); put16(out, (unsigned)w); buf_put(out, 3);
  buf_put(out, 1); buf_put(out, 0x22); buf_put(out, 0);
  buf_put(out, 2); buf_put(out, 0x11); buf_put(out, 1);
  buf_put(out, 3); buf_put(out, 0x11); buf_put(out, 1);
  emit_dht(out, 0x00, bits_dc_luma, val_dc);
  emit_dht(out, 0x10, bits_ac_luma, val_ac_luma);
  emit_dht(out, 0x01, bits_dc_chroma, val_dc);
  emit_dht(out, 0x11, bits_ac_chroma, val_ac_chroma);
  put16(out, 0xFFDA); put16(out, 12); buf_put(out, 3);
  buf_put(out, 1); buf_put(out, 0x00); buf_put(out, 2); buf_put(out, 0x11); buf_put(out, 3); buf_put(out, 0x11);
  buf_put(out, 0); buf_put(out, 63); buf_put(out, 0);

  /* entropy-coded segment (jccoefct.c compress_data + jchuff.c encode_mcu_huff) */
  bitw bw = {out, 0, 0};
  int last_dc[3] = {0, 0, 0};
  int16_t mcu[6][64];
  for (int my = 0; my < mcus_y; my++) {
    for (int mx = 0; mx < mcus_x; mx++) {
      /* luma 2x2 */
      for (int yi = 0; yi < 2; yi++) {
        int by = my * 2 + yi;
        if (by < y_hb) {
          for (int xi = 0; xi < 2; xi++) {
            int bx = mx * 2 + xi;
            if (bx < y_wb) {
              fdct_quant_block(&py, bx, by, ql, mcu[yi * 2 + xi]);
            } else { /* dummy block at the right edge: zero AC, DC of the left neighbour */
              memset(mcu[yi * 2 + xi], 0, sizeof(mcu[0]));
              mcu[yi * 2 + xi][0] = mcu[yi * 2 + xi - 1][0];
            }
          }
        } else { /* dummy block row at the bottom: DC of the block just before this row */
          for (int xi = 0; xi < 2; xi++) {
            memset(mcu[yi * 2 + xi], 0, sizeof(mcu[0]));
            mcu[yi * 2 + xi][0] = mcu[yi * 2 - 1][0];
          }
        }
      }
      /* chroma: MCU_width = MCU_height = 1 so every MCU has a real block */
      (void)c_hb;
      fdct_quant_block(&pcb, mx, my, qc, mcu[4]);
      fdct_quant_block(&pcr, mx, my, qc, mcu[5]);
      for (int i = 0; i < 4; i++) encode_block(&bw, mcu[i], &last_dc[0], &dcl, &acl);
      encode_block(&bw, mcu[4], &last_dc[1], &dcc, &acc);
      encode_block(&bw, mcu[5], &last_dc[2], &dcc, &acc);
    }
  }
  flush_bits(&bw);
  put16(out, 0xFFD9);
  free(py.s); free(pcb.s); free(pcr.s);
  return 0;
}

/* ---------- decoder ---------- */

typedef struct {
  /* jdhuff.c-style canonical decoding tables */
  int32_t maxcode[18];
  int32_t valoffset[17];
  uint8_t vals[256];
  int present;
} huff_dec;

static void make_dec_table(const uint8_t bits[17], const uint8_t *vals, int n, huff_dec *t) {
  memset(t, 0, sizeof(*t));
  memcpy(t->vals, vals, (size_t)n);
  int32_t code = 0;
  int p = 0;
  for (int l = 1; l <= 16; l++) {
    if (bits[l]) {
      t->valoffset[l] = p - code;
      p += bits[l];
      code += bits[l];
      t->maxcode[l] = code - 1;
    } else {
      t->maxcode[l] = -1;
    }
    code <<= 1;
  }
  t->maxcode[17] = 0x7FFFFFFF;
  t->present = 1;
}

typedef struct {
  const uint8_t *p;
  size_t len, pos;
  uint32_t acc;
  int nbits;
  int hit_marker;
} bitr;

static int get_bit(bitr *b) {
  if (b->nbits == 0) {
    uint8_t v = 0;
    if (!b->hit_marker && b->pos < b->len) {
      v = b->p[b->pos];
      if (v == 0xFF) {
        if (b->pos + 1 < b->len && b->p[b->pos + 1] == 0x00) {
          b->pos += 2;
        } else { /* marker: feed zeros (libjpeg: insufficient data -> zero bits) */
          b->hit_marker = 1;
          v = 0;
        }
      } else {
        b->pos++;
      }
    }
    b->acc = v;
    b->nbits = 8;
  }
  b->nbits--;
  return (int)((b->acc >> b->nbits) & 1);
}
static int get_bits(bitr *b, int n) {
  int v = 0;
  for (int i = 0; i < n; i++) v = (v << 1) | get_bit(b);
  return v;
}
static int huff_decode(bitr *b, const huff_dec *t) {
  int32_t code = 0;
  for (int l = 1; l <= 16; l++) {
    code = (code << 1) | get_bit(b);
    if (t->maxcode[l] >= 0 && code <= t->maxcode[l]) return t->vals[(code + t->valoffset[l]) & 0xFF];
  }
  return 0;
}
static int extend(int v, int n) { return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v; }

/* jidctint.c jpeg_idct_islow (dequantise + 2-pass IDCT), output samples 0..255 */
static uint8_t range_limit(int32_t v) {
  /* jdmaster.c prepare_range_limit_table, indexed with (v & RANGE_MASK) after CENTERJSAMPLE offset */
  int idx = (int)(v & 1023);
  if (idx < 128) return (uint8_t)(128 + idx);
  if (idx < 512) return 255;
  if (idx < 896) return 0;
  return (uint8_t)(idx - 896);
}
static void idct_islow(const int16_t coef[64], const uint16_t q[64], uint8_t *dst, int stride) {
  int32_t ws[64];
  int32_t tmp0, tmp1, tmp2, tmp3, tmp10, tmp11, tmp12, tmp13, z1, z2, z3, z4, z5;
  for (int c = 0; c < 8; c++) { /* pass 1: columns */
    int32_t in[8];
    for (int r = 0; r < 8; r++) in[r] = (int32_t)coef[r * 8 + c] * (int32_t)q[r * 8 + c];
    z2 = in[2]; z3 = in[6];
    z1 = (z2 + z3) * FIX_0_541196100;
    tmp2 = z1 + z3 * (-FIX_1_847759065);
    tmp3 = z1 + z2 * FIX_0_765366865;
    z2 = in[0]; z3 = in[4];
    tmp0 = (z2 + z3) * (1 << CONST_BITS);
    tmp1 = (z2 - z3) * (1 << CONST_BITS);
    tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    ws[0 * 8 + c] = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
    ws[7 * 8 + c] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
    ws[1 * 8 + c] = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
    ws[6 * 8 + c] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
    ws[2 * 8 + c] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
    ws[5 * 8 + c] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
    ws[3 * 8 + c] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
    ws[4 * 8 + c] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
  }
  for (int r = 0; r < 8; r++) { /* pass 2: rows */
    const int32_t *w = ws + r * 8;
    z2 = w[2]; z3 = w[6];
    z1 = (z2 + z3) * FIX_0_541196100;
    tmp2 = z1 + z3 * (-FIX_1_847759065);
    tmp3 = z1 + z2 * FIX_0_765366865;
    tmp0 = (w[0] + w[4]) * (1 << CONST_BITS);
    tmp1 = (w[0] - w[4]) * (1 << CONST_BITS);
    tmp10 = tmp0 + tmp3; tmp13 = tmp0 - tmp3;
    tmp11 = tmp1 + tmp2; tmp12 = tmp1 - tmp2;
    tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2; z4 = tmp1 + tmp3;
    z5 = (z3 + z4) * FIX_1_175875602;
    tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869;
    tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
    z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447;
    z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    uint8_t *o = dst + (size_t)r * stride;
    o[0] = range_limit(DESCALE(tmp10 + tmp3, CONST_BITS + PASS1_BITS + 3));
    o[7] = range_limit(DESCALE(tmp10 - tmp3, CONST_BITS + PASS1_BITS + 3));
    o[1] = range_limit(DESCALE(tmp11 + tmp2, CONST_BITS + PASS1_BITS + 3));
    o[6] = range_limit(DESCALE(tmp11 - tmp2, CONST_BITS + PASS1_BITS + 3));
    o[2] = range_limit(DESCALE(tmp12 + tmp1, CONST_BITS + PASS1_BITS + 3));
    o[5] = range_limit(DESCALE(tmp12 - tmp1, CONST_BITS + PASS1_BITS + 3));
    o[3] = range_limit(DESCALE(tmp13 + tmp0, CONST_BITS + PASS1_BITS + 3));
    o[4] = range_limit(DESCALE(tmp13 - tmp0, CONST_BITS + PASS1_BITS + 3));
  }
}

static int decode_block(bitr *b, int16_t blk[64], int *last_dc, const huff_dec *dc, const huff_dec *ac) {
  memset(blk, 0, sizeof(int16_t) * 64);
  int s = huff_decode(b, dc);
  int diff = s ? extend(get_bits(b, s), s) : 0;
  *last_dc += diff;
  blk[0] = (int16_t)*last_dc;
  for (int k = 1; k < 64; k++) {
    int rs = huff_decode(b, ac);
    int r = rs >> 4, sz = rs & 15;
    if (sz) {
      k += r;
      if (k > 63) break;
      blk[zigzag[k]] = (int16_t)extend(get_bits(b, sz), sz);
    } else {
      if (r != 15) break;
      k += 15;
    }
  }
  return 0;
}

int pcco_jpeg_decode_rgb(const uint8_t *jpg, size_t len, uint8_t **rgb_out, int *w_out, int *h_out) {
  uint16_t qt[4][64];
  huff_dec hd[2][4];
  memset(qt, 0, sizeof(qt));
  memset(hd, 0, sizeof(hd));
  int w = 0, h = 0, ncomp = 0, restart = 0;
  int comp_h[3] = {0}, comp_v[3] = {0}, comp_q[3] = {0}, comp_dc[3] = {0}, comp_ac[3] = {0};
  size_t pos = 0;
  if (len < 4 || jpg[0] != 0xFF || jpg[1] != 0xD8) return -1;
  pos = 2;
  int found_sos = 0;
  while (pos + 4 <= len && !found_sos) {
    if (jpg[pos] != 0xFF) return -2;
    int m = jpg[pos + 1];
    pos += 2;
    if (m == 0xFF) { pos--; continue; }
    if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
    size_t seglen = ((size_t)jpg[pos] << 8) | jpg[pos + 1];
    if (pos + seglen > len) return -3;
    const uint8_t *s = jpg + pos + 2;
    size_t n = seglen - 2;
    if (m == 0xDB) {
      size_t i = 0;
      while (i < n) {
        int pq = s[i] >> 4, tq = s[i] & 15;
        i++;
        if (tq > 3) return -4;
        for (int k = 0; k < 64; k++) {
          unsigned v = pq ? (((unsigned)s[i] << 8) | s[i + 1]) : s[i];
          i += pq ? 2 : 1;
          qt[tq][zigzag[k]] = (uint16_t)v;
        }
      }
    } else if (m == 0xC4) {
      size_t i = 0;
      while (i < n) {
        int tc = s[i] >> 4, th = s[i] & 15;
        i++;
        if (tc > 1 || th > 3) return -5;
        uint8_t bits[17];
        bits[0] = 0;
        int cnt = 0;
        for (int k = 1; k <= 16; k++) { bits[k] = s[i++]; cnt += bits[k]; }
        if (cnt > 256) return -5;
        make_dec_table(bits, s + i, cnt, &hd[tc][th]);
        i += (size_t)cnt;
      }
    } else if (m == 0xC0 || m == 0xC1) {
      if (s[0] != 8) return -6;
      h = (s[1] << 8) | s[2];
      w = (s[3] << 8) | s[4];
      ncomp = s[5];
      if (ncomp != 3) return -7;
      for (int c = 0; c < 3; c++) {
        comp_h[c] = s[7 + 3 * c] >> 4;
        comp_v[c] = s[7 + 3 * c] & 15;
        comp_q[c] = s[8 + 3 * c];
      }
    } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC8 && m != 0xCC)) {
      return -8; /* not baseline */
    } else if (m == 0xDD) {
      restart = (s[0] << 8) | s[1];
    } else if (m == 0xDA) {
      if (s[0] != 3) return -9;
      for (int c = 0; c < 3; c++) {
        comp_dc[c] = s[2 + 2 * c] >> 4;
        comp_ac[c] = s[2 + 2 * c] & 15;
      }
      found_sos = 1;
    }
    pos += seglen;
  }
  if (!found_sos || w <= 0 || h <= 0) return -10;
  if (!(comp_h[0] == 2 && comp_v[0] == 2 && comp_h[1] == 1 && comp_v[1] == 1 && comp_h[2] == 1 && comp_v[2] == 1))
    return -11; /* oracle handles the 4:2:0 layout the encoder side produces */

  int mcus_x = (w + 15) / 16, mcus_y = (h + 15) / 16;
  int cw = (w + 1) / 2, ch = (h + 1) / 2;
  int yw = mcus_x * 16, yh = mcus_y * 16, cpw = mcus_x * 8, cph = mcus_y * 8;
  uint8_t *Y = (uint8_t *)calloc((size_t)yw * yh, 1);
  uint8_t *C[2];
  C[0] = (uint8_t *)calloc((size_t)cpw * cph, 1);
  C[1] = (uint8_t *)calloc((size_t)cpw * cph, 1);

  bitr br = {jpg, len, pos, 0, 0, 0};
  int last_dc[3] = {0, 0, 0};
  int16_t blk[64];
  int mcu_count = 0, next_rst = 0;
  for (int my = 0; my < mcus_y; my++) {
    for (int mx = 0; mx < mcus_x; mx++) {
      if (restart && mcu_count && (mcu_count % restart) == 0) {
        /* process_restart: discard partial byte, read RSTn, reset predictors */
        br.nbits = 0;
        br.hit_marker = 0;
        while (br.pos + 1 < br.len && !(br.p[br.pos] == 0xFF && br.p[br.pos + 1] == (0xD0 + next_rst))) br.pos++;
        br.pos += 2;
        next_rst = (next_rst + 1) & 7;
        last_dc[0] = last_dc[1] = last_dc[2] = 0;
      }
      for (int yi = 0; yi < 2; yi++)
        for (int xi = 0; xi < 2; xi++) {
          decode_block(&br, blk, &last_dc[0], &hd[0][comp_dc[0]], &hd[1][comp_ac[0]]);
          idct_islow(blk, qt[comp_q[0]], Y + (size_t)(my * 16 + yi * 8) * yw + mx * 16 + xi * 8, yw);
        }
      for (int c = 1; c < 3; c++) {
        decode_block(&br, blk, &last_dc[c], &hd[0][comp_dc[c]], &hd[1][comp_ac[c]]);
        idct_islow(blk, qt[comp_q[c]], C[c - 1] + (size_t)(my * 8) * cpw + mx * 8, cpw);
      }
      mcu_count++;
    }
  }

  /* jdsample.c h2v2_fancy_upsample on the ch x cw real chroma samples (vertical
   * context clamped to the real rows, jdmainct.c) -- or plain 2x2 replication when
   * downsampled_width <= 2 -- then jdcolor.c ycc_rgb_convert */
  int uw = cpw * 2;
  uint8_t *U[2];
  U[0] = (uint8_t *)calloc((size_t)uw * (size_t)(ch * 2), 1);
  U[1] = (uint8_t *)calloc((size_t)uw * (size_t)(ch * 2), 1);
  for (int k = 0; k < 2; k++) {
    for (int r = 0; r < ch; r++) {
      for (int v = 0; v < 2; v++) {
        if (cw <= 2) { /* jdsample.c jinit_upsampler: fancy only if downsampled_width > 2, else h2v2_upsample (box) */
          uint8_t *ob = U[k] + (size_t)(2 * r + v) * uw;
          for (int cc = 0; cc < cw; cc++) ob[2 * cc] = ob[2 * cc + 1] = C[k][(size_t)r * cpw + cc];
          continue;
        }
        int nr = v == 0 ? (r > 0 ? r - 1 : 0) : (r + 1 < ch ? r + 1 : ch - 1);
        const uint8_t *in0 = C[k] + (size_t)r * cpw, *in1 = C[k] + (size_t)nr * cpw;
        uint8_t *o = U[k] + (size_t)(2 * r + v) * uw;
        int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
        const uint8_t *p0 = in0 + 2, *p1 = in1 + 2;
        *o++ = (uint8_t)((thiscol * 4 + 8) >> 4);
        *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
        for (int cc = cw - 2; cc > 0; cc--) {
          nextcol = (*p0++) * 3 + (*p1++);
          *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
          *o++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
          lastcol = thiscol; thiscol = nextcol;
        }
        *o++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        *o++ = (uint8_t)((thiscol * 4 + 7) >> 4);
      }
    }
  }
  uint8_t *rgb = (uint8_t *)malloc((size_t)w * h * 3);
  for (int r = 0; r < h; r++) {
    for (int c = 0; c < w; c++) {
      int y = Y[(size_t)r * yw + c];
      int cb = U[0][(size_t)r * uw + c], cr = U[1][(size_t)r * uw + c];
      int32_t xr = cr - 128, xb = cb - 128;
      int32_t r_add = (91881 * xr + 32768) >> 16;
      int32_t b_add = (116130 * xb + 32768) >> 16;
      int32_t g_add = ((-22554 * xb + 32768) + (-46802 * xr)) >> 16;
      int R = y + r_add, G = y + g_add, B = y + b_add;
      rgb[((size_t)r * w + c) * 3 + 0] = (uint8_t)(R < 0 ? 0 : R > 255 ? 255 : R);
      rgb[((size_t)r * w + c) * 3 + 1] = (uint8_t)(G < 0 ? 0 : G > 255 ? 255 : G);
      rgb[((size_t)r * w + c) * 3 + 2] = (uint8_t)(B < 0 ? 0 : B > 255 ? 255 : B);
    }
  }
  free(Y); free(C[0]); free(C[1]); free(U[0]); free(U[1]);
  *rgb_out = rgb; *w_out = w; *h_out = h;
  return 0;
}
