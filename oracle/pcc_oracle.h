/*
 * pcc_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the intra-frame hot path of cwi-dis/cwi-pcl-codec
 * (pcl::io::OctreePointCloudCodecV2<PointXYZRGB>::encodePointCloud /
 * decodePointCloud) including the pieces it inherits from PCL 1.8.1-1.10.0
 * (adaptive bounding box, pointer octree, depth-first serialisation, static
 * range coder) and from libjpeg-turbo (baseline JPEG, via jpeg_io).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product library (libpcc_hip.so) never links it.
 *
 * PARITY STATUS
 *   - snake grid mapping: PINNED against the reference's own header compiled
 *     from /root/reference (oracle/_ref, see oracle/Makefile).
 *   - JPEG stage: PINNED against libjpeg-turbo output (Pillow's bundled
 *     libjpeg-turbo, fixtures in tests/golden/, generator
 *     tests/golden/make_jpeg_golden.py).
 *   - PCL-inherited arithmetic (bbox growth, keys, DFS stream, range coder,
 *     colour average): "parity unpinned" -- PCL is not vendored in the
 *     reference and is not installed; the reference has no tests or golden
 *     vectors.  Restated from PCL 1.10.0 behaviour and
 *     anchored on the reference's call sites cited at each function.
 *
 * Reference file abbreviations (relative to /root/reference):
 *   impl.hpp = cloud_codec_v2/include/pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp
 *   codec.h  = cloud_codec_v2/include/pcl/cloud_codec_v2/point_cloud_codec_v2.h
 *   jpegcc.h = cloud_codec_v2/include/pcl/cloud_codec_v2/color_coding_jpeg.h
 *   snake.h  = cloud_codec_v2/include/pcl/cloud_codec_v2/snake_grid_mapping.h
 *   ptv2.h   = cloud_codec_v2/include/pcl/cloud_codec_v2/point_coding_v2.h
 *   jpeg_io.hpp = jpeg_io/include/pcl/io/impl/jpeg_io.hpp
 */
#ifndef PCC_ORACLE_H
#define PCC_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* pcl::PointXYZRGB memory layout: 32 bytes, rgb word at byte offset 16. */
typedef struct {
  float x, y, z, w;   /* w = 1.0f in PCL */
  uint32_t rgba;      /* b | g<<8 | r<<16 | a<<24 */
  uint32_t pad[3];
} pcco_point;

typedef struct {
  uint8_t *data;
  size_t len, cap;
} pcco_buf;

/* Codec configuration = the 14-argument ctor (codec.h:108-143) plus setters. */
typedef struct {
  double octree_resolution;     /* octreeResolution_arg */
  double point_resolution;      /* pointResolution_arg (header only) */
  int do_color_encoding;        /* doColorEncoding_arg */
  int color_bit_resolution;     /* colorBitResolution_arg (header byte) */
  int color_coding_type;        /* 0 PCL raw, 1 SNAKE jpeg, 2 LINES jpeg, 3 GRID(raw) */
  int do_voxel_centroid;        /* doVoxelGridCentroid_arg */
  int create_scalable;          /* header flag only */
  int do_connectivity;          /* header flag only */
  int jpeg_quality;             /* jpeg_quality_arg */
  int macroblock_size;          /* header field, default 16 (codec.h:138) */
  int do_icp_color_offset;      /* header flag, default 0 (codec.h:141) */
  uint32_t frame_id;            /* value of frame_ID_ written to the header */
} pcco_params;

/* Everything one intra encode produces (all intermediate products exposed
 * so the HIP path can be compared stage by stage). */
typedef struct {
  double bbox[6];               /* min_x,min_y,min_z,max_x,max_y,max_z */
  uint32_t depth;               /* final octree depth D */
  uint64_t n_points_in;         /* finite input points (object_count_) */
  uint64_t n_leaves;            /* L */
  uint64_t n_branches;          /* B */
  uint32_t *leaf_keys;          /* 3*L: key.x,key.y,key.z in DFS leaf order */
  uint32_t *leaf_counts;        /* L: points per leaf */
  pcco_buf occupancy;           /* B bytes, DFS pre-order */
  pcco_buf bgr;                 /* 3*L bytes, (b,g,r) per leaf (P6) */
  pcco_buf centroid_bytes;      /* 3*L bytes if do_voxel_centroid */
  pcco_buf color_payload;       /* what gets range-coded: JPEG bytes / raw */
  pcco_buf snake_image;         /* 3*W*H mapped image (mode 1 only) */
  uint32_t image_w, image_h;
  pcco_point *simplified;       /* L points = output_ cloud (impl.hpp:1576) */
  pcco_buf bitstream;           /* header + entropy-coded payload */
  uint64_t perf[3];             /* compression_performance_metrics */
} pcco_frame;

typedef struct {
  pcco_point *points;
  uint64_t n;
  pcco_params params;           /* recovered from the header */
  double bbox[6];
  uint32_t depth;
  size_t consumed;              /* bytes of the input stream consumed */
} pcco_cloud;

void pcco_buf_free(pcco_buf *b);
void pcco_frame_free(pcco_frame *f);
void pcco_cloud_free(pcco_cloud *c);

/* ---- stage functions ---- */

/* P7: pcl::StaticRangeCoder::encodeCharVectorToStream (call sites impl.hpp:1694,1706,1719). */
size_t pcco_rc_encode(const uint8_t *in, size_t n, pcco_buf *out);
/* P7 mirror: decodeStreamToCharVector (impl.hpp:1778,1789,1798).  Returns bytes consumed. */
size_t pcco_rc_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t n);

/* C3b: snake.h:23-118.  perm[i] = pixel index of linear element i. */
void pcco_snake_perm(int w, int h, int32_t *perm);

/* C5: jpeg_io.hpp:211-330 (libjpeg: jpeg_set_defaults + jpeg_set_quality(q,TRUE)). */
int pcco_jpeg_encode_rgb(const uint8_t *rgb, int w, int h, int quality, pcco_buf *out);
/* C8: jpeg_io.hpp:90-192 (libjpeg defaults: islow IDCT, fancy upsampling). */
int pcco_jpeg_decode_rgb(const uint8_t *jpg, size_t len, uint8_t **rgb, int *w, int *h);

/* C1: impl.hpp:80-213.  Returns 0 on success, 1 if the cloud was dropped (empty). */
int pcco_encode_intra(const pcco_point *pts, size_t n, const pcco_params *p, pcco_frame *out);
/* C8: impl.hpp:224-310. Returns 0 on success. */
int pcco_decode_intra(const uint8_t *bs, size_t len, pcco_cloud *out);

/* normalize_pointclouds (impl.hpp:1871-1967) for a single cloud with is_bb_init=false. */
void pcco_normalize_single(pcco_point *pts, size_t n, double bb_expand_factor,
                           float bb_min[3], float bb_max[3]);

#ifdef __cplusplus
}
#endif
#endif
