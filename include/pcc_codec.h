/*
 * pcc_codec.h -- C ABI of libpcc_hip.so: the MI355X-native intra-frame hot path of the
 * CWI point-cloud codec (cwi-dis/cwi-pcl-codec).
 *
 * The reference has no plugin / FFI layer: its boundary is the C++ class template
 * pcl::io::OctreePointCloudCodecV2<PointT> (codec.h:70-368, instantiated once at
 * cloud_codec_v2/src/point_cloud_codec_v2.cpp:45).  The drop-in is therefore a
 * same-named header-only C++ shim (cwi-pcl-codec_amd/shim/) whose methods forward to
 * the entry points below; INTEGRATION.md shows the binding.  Everything that crosses
 * this boundary is plain C: pointers, sizes, PODs.  No C++ or torch types.
 *
 * Reference file abbreviations (relative to /root/reference):
 *   codec.h  = cloud_codec_v2/include/pcl/cloud_codec_v2/point_cloud_codec_v2.h
 *   impl.hpp = cloud_codec_v2/include/pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp
 *   eval.hpp = apps/evaluate_compression/include/pcl/apps/evaluate_compression/impl/evaluate_compression_impl.hpp
 *
 * Threading: a pcc_ctx is bound to one GPU and one HIP stream and is not re-entrant
 * (like the reference's codec object, which holds per-frame state).  Contexts share
 * nothing, so "one frame per GPU" is just N contexts on N host threads.
 * Ownership: input buffers are caller-owned; every pointer returned in an out-struct
 * is library-owned and stays valid until the next call on the same context.
 * Errors: every call returns PCC_OK (0) or a negative code; pcc_last_error() gives text.
 *
 * Layout of the interface: THIS header is the drop-in boundary -- everything the class shim (cwi-pcl-codec_amd/shim/), the
 * evaluation app and a frame loop call.  pcc_codec_tools.h declares what measurements, tests and tools use on top of it
 * (kernel and host-stage timings, the pipeline's pieces one by one, the host stages' building blocks, the device range
 * coder's harness, pcc_debug_*); a caller of the codec never needs it.
 *
 * Environment: the shipped library reads NINE variables, all of them deployment configuration of pcc_pipeline (none changes
 * an output byte): PCC_PIPELINE_ENTROPY = host | gpu (where the entropy stage runs; default host),
 * PCC_PIPELINE_GPU_THREADS, PCC_PIPELINE_UPLOAD_THREADS (frames in flight on the GPU / uploads), PCC_PIPELINE_BATCH
 * (frames per coder loop, 1..16, default 4), PCC_PIPELINE_PIN = groups | cores | none with PCC_PIPELINE_PIN_OFFSET and
 * PCC_PIPELINE_PIN_SPAN (core pinning of the entropy threads), and torchrun's LOCAL_RANK / LOCAL_WORLD_SIZE (each rank pins
 * inside its own share of the cores: cores of its GPU's own NUMA node where /sys names one for every GPU, rank i on GPU i).  Developer switches (traces, bisecting forms) exist only in the `make dev` build:
 * csrc/pcc_dev.h.
 */
#ifndef PCC_CODEC_H
#define PCC_CODEC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCC_OK 0
#define PCC_ERR_ARG (-1)        /* bad argument */
#define PCC_ERR_HIP (-2)        /* a HIP runtime call failed (no GPU, OOM, ...) */
#define PCC_ERR_EMPTY (-3)      /* no finite input point: the reference drops the frame (impl.hpp:206-212) */
#define PCC_ERR_UNSUPPORTED (-4) /* a tree of more than 31 levels, 2^30 points or more, a stream of 2 GB or more */
#define PCC_ERR_STREAM (-5)     /* decode: header not found / truncated / corrupt */
#define PCC_ERR_STATE (-6)      /* call order (finish without launch, ...) */
#define PCC_NO_NUMA_NODE (-100) /* pcc_pipeline_get(p, "numa_node"): the cores were not chosen by NUMA node */

typedef struct pcc_ctx pcc_ctx;

/* pcl::PointXYZRGB as the reference lays it out: 32 bytes, colour word at offset 16. */
typedef struct pcc_point_xyzrgb {
  float x, y, z, w;
  uint32_t rgba; /* b | g<<8 | r<<16 | a<<24 */
  uint32_t pad[3];
} pcc_point_xyzrgb;

/* Codec configuration: the arguments of the reference constructor (codec.h:108-121, as
 * called at eval.hpp:377-395) plus the two setters the app uses (codec.h:149,164). */
typedef struct pcc_params {
  double octree_resolution;      /* octreeResolution_arg */
  double point_resolution;       /* pointResolution_arg (written to the header) */
  int32_t do_color_encoding;     /* doColorEncoding_arg */
  int32_t color_bit_resolution;  /* colorBitResolution_arg */
  int32_t color_coding_type;     /* colorCodingType_arg: 0 PCL, 1 JPEG snake, 2 JPEG lines, 3 raw */
  int32_t do_voxel_centroid;     /* doVoxelGridCentroid_arg */
  int32_t create_scalable;       /* createScalableStream_arg (header flag only) */
  int32_t do_connectivity;       /* codeConnectivity_arg (header flag only) */
  int32_t jpeg_quality;          /* jpeg_quality_arg */
  int32_t macroblock_size;       /* setMacroblockSize (header field) */
  int32_t do_icp_color_offset;   /* setDoICPColorOffset(bool) (header flag) */
  uint32_t frame_id;             /* value of frame_ID_ after the ++ at impl.hpp:133 */
} pcc_params;

/* What the GPU stage hands to the host entropy stage (all host pointers). */
typedef struct pcc_hot_result {
  double bbox[6];            /* adaptive bounding box: min xyz, max xyz (header bytes) */
  uint32_t depth;            /* final octree depth D */
  uint32_t n_epochs;         /* bounding-box growth epochs seen */
  uint64_t n_points_in;      /* finite input points (object_count_) */
  uint64_t n_leaves;         /* L = occupied voxels = point_count_ in the header */
  uint64_t n_branches;       /* B = occupancy bytes */
  const uint8_t *occupancy;  /* B bytes, depth-first pre-order (serializeTree) */
  const uint8_t *bgr;        /* 3L bytes (b,g,r) per voxel in leaf order; NULL if no colour */
  const uint8_t *centroid;   /* 3L bytes; NULL unless do_voxel_centroid */
  const uint8_t *image;      /* snake-mapped 3*W*H image (color_coding_type 1), else NULL */
  uint32_t image_w, image_h;
  float gpu_ms;              /* HIP-event time of the kernel sequence */
  /* JPEG front end done on the GPU for the snake image: quantised coefficients, 6 blocks of 64 per
   * 16x16 MCU (Y00 Y01 Y10 Y11 Cb Cr) in zigzag order; NULL if the host has to start from `image`. */
  const int16_t *jpeg_coefs;
  /* JPEG Huffman coding done on the GPU as well: one record of `jpeg_tile_words` u32 per MCU row (16 image
   * rows): [0] bits, [1],[2] bit offsets of the first MCU's Cb / Cr block, [3] overflow flag, [4..6] DC of the
   * row's first Y/Cb/Cr block (their DC codes are inserted by the host), [7..9] DC of the last ones, [16..]
   * the bit string, MSB first inside each u32.  NULL if the host Huffman-codes `jpeg_coefs` (or `image`). */
  const uint32_t *jpeg_tiles;
  uint32_t jpeg_tile_words, jpeg_n_tiles;
  /* 256 counts: how often each byte value occurs in `occupancy` (counted on the GPU; saves the range coder its
   * histogram pass).  NULL: the host counts. */
  const uint32_t *occupancy_histogram;
  /* Colour coding type 2 ("lines", jpegcc.h:244-317) coded on the GPU: per strip {word offset into jpeg_lines_data, bits,
   * width in pixels, 1 = did not fit}, and the strips' entropy-coded bit strings (MSB first inside each u32, no 0xFF
   * stuffing, no headers: the host adds those).  NULL: the host codes the strips from `bgr`. */
  const uint32_t *jpeg_lines_dir;
  const uint32_t *jpeg_lines_data;
  uint32_t jpeg_n_lines;
} pcc_hot_result;

typedef struct pcc_bitstream {
  const uint8_t *data;
  size_t len;
  uint64_t perf[3];          /* getPerformanceMetrics(): octree, centroid, colour bytes (codec.h:193-197) */
} pcc_bitstream;

typedef struct pcc_cloud {
  const pcc_point_xyzrgb *points;
  size_t n;
  pcc_params params;         /* as recovered from the frame header */
  double bbox[6];
  uint32_t depth;
  size_t consumed;           /* bytes of the input consumed */
} pcc_cloud;

/* ---- lifetime ---- */
pcc_ctx *pcc_create(int device);                 /* replaces `new OctreePointCloudCodecV2` (eval.hpp:377) */
void pcc_destroy(pcc_ctx *ctx);
const char *pcc_last_error(pcc_ctx *ctx);
/* What this binary is: "pcc_hip 0.1 (gfx950)" for the product library and nothing else -- the same sources compiled for the
 * CPU executor of tests/emu answer "pcc_emu ...", developer builds name themselves.  bench.py and smoke() refuse anything but
 * the product string (a line timed on another build must not be mistaken for a measurement). */
const char *pcc_version(void);

/* ---- encodePointCloud (codec.h:174-175, impl.hpp:80-213) ----
 * One call = one I-frame: H2D of the caller's cloud, GPU hot path, host entropy stage. */
int pcc_encode_intra(pcc_ctx *ctx, const void *host_points, size_t n, size_t stride, size_t rgb_offset,
                     const pcc_params *params, pcc_bitstream *out);

/* Same with the cloud already resident in HBM (bench / pipelines). */
int pcc_encode_intra_device(pcc_ctx *ctx, const void *dev_points, size_t n, size_t stride, size_t rgb_offset,
                            const pcc_params *params, pcc_bitstream *out);

/* Optional: allocate the context's HBM arena, pinned landing buffers and bitstream buffer for frames of up to
 * `max_points` points / `bitstream_bytes` bytes now instead of on first use (the first frame on a fresh context pays
 * some twenty hipMalloc / hipHostMalloc calls otherwise). */
int pcc_reserve(pcc_ctx *ctx, size_t max_points, size_t bitstream_bytes);

/* ---- the three stages of encodePointCloud, separately (for overlap across frames) ---- */
/* addPointsFromInputCloud + serializeTree + leaf callbacks (impl.hpp:99,166,1509-1578) on the GPU; asynchronous. */
int pcc_hotpath_launch(pcc_ctx *ctx, const void *dev_points, size_t n, size_t stride, size_t rgb_offset,
                       const pcc_params *params);
/* wait for the kernels, bring occupancy bytes / colour image / centroid bytes to the host */
int pcc_hotpath_finish(pcc_ctx *ctx, pcc_hot_result *out);
/* writeFrameHeader + entropyEncoding (impl.hpp:175-178, 1472-1486, 1682-1760): host only, no GPU calls;
 * may run on another thread while the context's GPU stage works on the next frame IF `ctx_for_output`
 * is a different context (the bitstream is stored there). */
int pcc_entropy_encode(pcc_ctx *ctx_for_output, const pcc_hot_result *hot, const pcc_params *params,
                       pcc_bitstream *out);

/* getOutputCloud() (eval.hpp:862): the simplified cloud of the last encode, L points (impl.hpp:1576). */
int pcc_get_output_cloud(pcc_ctx *ctx, const pcc_point_xyzrgb **points, size_t *n);

/* ---- decodePointCloud (codec.h:177-178, impl.hpp:224-310) ---- */
int pcc_decode_intra(pcc_ctx *ctx, const uint8_t *stream, size_t len, pcc_cloud *out);

/* The same with the data-parallel half on the GPU: the host runs what is sequential by construction (the three range
 * decoders, the JPEG's Huffman decoding, the walk over the depth-first occupancy stream that finds every byte's
 * level), the GPU the rest (voxel keys -> centres or centroids, inverse DCT, chroma upsampling, colour conversion and
 * un-snaking of the colour image).  Same cloud, bit for bit; out->points is page-locked library memory. */
int pcc_decode_intra_gpu(pcc_ctx *ctx, const uint8_t *stream, size_t len, pcc_cloud *out);

/* ---- helpers around the path ---- */
/* device memory for callers that keep clouds resident (bench, multi-frame pipelines) */
int pcc_device_alloc(pcc_ctx *ctx, size_t bytes, void **dev_ptr);
int pcc_device_free(pcc_ctx *ctx, void *dev_ptr);
int pcc_device_upload(pcc_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes);
/* context knobs (do not change any output byte):
 *   "jpeg_on_gpu" (default 2): 2 = the whole JPEG stage but the file headers runs on the GPU (colour conversion,
 *                 4:2:0 downsample, FDCT, quantisation, Huffman coding per MCU row; the host stitches the rows
 *                 together);  1 = up to the quantised coefficients, the host Huffman-codes;  0 = the host starts
 *                 from the image.
 *   "copy_image"  (default 1): bring the snake-mapped image itself back in pcc_hot_result.image (needed only
 *                 for inspection when jpeg_on_gpu is 1).
 *   "pack_upload" (default 0): a frame from host memory is packed to 16 bytes per point before it crosses PCIe.
 *   (measurement and test hooks -- "profile_events", "force_pairs", "no_cell_ranks", "icp_waves", "rc_device_lanes" -- are
 *   described in pcc_codec_tools.h) */
int pcc_set_option(pcc_ctx *ctx, const char *name, int value);

/* ---- a sequence of frames on one GPU (the app's frame loop, eval.hpp:818-835) ----
 * A pipeline owns a ring of pcc_ctx, a few GPU-stage threads that keep frames in flight on the GPU, and
 * `n_workers` entropy threads that run the serial host stage, a few frames per coder loop.  Frames are independent I-frames
 * (impl.hpp:89-90,126-130);
 * the frames get consecutive frame ids starting at params->frame_id, dropped (empty / all non-finite) frames
 * do not consume one (frame_ID_ is the only state the reference carries from frame to frame, impl.hpp:133,
 * 206-212), so the bitstreams equal those of the reference's serial loop; a dropped frame yields len 0.  The calls block
 * until every frame is done; `out[f]` stays valid until the next call on the pipeline. */
typedef struct pcc_pipeline pcc_pipeline;
pcc_pipeline *pcc_pipeline_create(int device, int n_workers);
void pcc_pipeline_destroy(pcc_pipeline *p);
/* pipeline knobs (no output byte changes), to be set between calls:
 *   "entropy_on_gpu" 1: the range coders of the entropy stage run on the GPU (pcc_entropy_batch), the entropy threads
 *                 only copy, stitch JPEG rows and assemble -- for hosts with fewer cores than the GPU stage can feed; a flush
 *                 takes ~0.1 s whatever its size.  0 (default; PCC_PIPELINE_ENTROPY=host|gpu sets it): on the host.  Any other
 *                 value is PCC_ERR_ARG; any other word in the variable means host, and says so on stderr once.
 *   "entropy_gpu_batch" (default 256): frames per flush and entropy thread.
 *   "rc_device_lanes" (default 0): with the entropy stage on the GPU, the device range coder codes one stream per LANE
 *                 instead of one per wave (same bytes; never timed on an MI355X).
 *   "pack_upload" (default 0): frames from host memory are packed to 16 bytes per point before they cross PCIe. */
int pcc_pipeline_set_option(pcc_pipeline *p, const char *name, int value);
/* what the pipeline runs with: "workers" (entropy threads), "gpu_threads", "contexts", "frames_per_coder_call" (the batch
 * size in force: PCC_PIPELINE_BATCH after its range check), "last_entropy_mode" (where the entropy stage of the last call
 * ran: 0 host, 1 GPU), "rc_device_lanes", "entropy_gpu_batch", "numa_node" (the host's NUMA node whose cores the pipeline's
 * threads were given -- the node its GPU hangs off -- or PCC_NO_NUMA_NODE); PCC_ERR_ARG for an unknown name */
int pcc_pipeline_get(pcc_pipeline *p, const char *name);
int pcc_pipeline_contexts(pcc_pipeline *p);
pcc_ctx *pcc_pipeline_context(pcc_pipeline *p, int index); /* for pcc_set_option on the contexts of the ring */
int pcc_pipeline_encode(pcc_pipeline *p, const void *const *dev_frames, const size_t *n_points, size_t n_frames,
                        size_t stride, size_t rgb_offset, const pcc_params *params, pcc_bitstream *out);
/* The same sequence with the frames in HOST memory -- the call the reference app's frame loop maps to (eval.hpp:818-835
 * encodes clouds that sit in std::vectors): upload of frame k+1 (PCIe, one frame at a time through the pipeline's
 * upload lane), kernels of frame k and the host entropy stage of frame k-1 overlap.  Same bitstreams. */
int pcc_pipeline_encode_host(pcc_pipeline *p, const void *const *host_frames, const size_t *n_points, size_t n_frames,
                             size_t stride, size_t rgb_offset, const pcc_params *params, pcc_bitstream *out);
/* Optional: set aside (and touch) the memory for the bitstreams of `n_frames` frames of up to `bytes_per_frame` bytes
 * before the frames arrive; otherwise the first call allocates per frame and later calls reuse what the largest call
 * needed.  With max_points_per_frame > 0 every context of the ring is prepared as well (pcc_reserve): a context used
 * for the first time in the middle of a sequence otherwise stalls its thread for several milliseconds. */
int pcc_pipeline_reserve(pcc_pipeline *p, size_t n_frames, size_t bytes_per_frame, size_t max_points_per_frame);
const char *pcc_pipeline_last_error(pcc_pipeline *p);

/* ---- the same frame loop over several GPUs of one node (frames shard one per GPU, no collective) ----
 * One pipeline per entry of `devices` (a GPU may be named more than once); frame f goes to devices[f mod n_devices];
 * the bitstreams come back in sequence order with the frame ids of the reference's serial loop (dropped frames do not
 * consume one).  pcc_pipeline_create_multi returns NULL if any of the devices is missing.  Every pipeline's host threads get
 * cores of their own: cores of the NUMA node the pipeline's GPU hangs off (/sys/bus/pci/devices/<address>/numa_node), split
 * among the pipelines whose GPUs share that node, if the host names a node for every GPU; an n-th of the allowed cores otherwise.  With device-resident
 * frames, frame f has to live on devices[f mod n_devices]. */
typedef struct pcc_multi_pipeline pcc_multi_pipeline;
pcc_multi_pipeline *pcc_pipeline_create_multi(const int *devices, int n_devices, int n_workers_per_device);
void pcc_multi_pipeline_destroy(pcc_multi_pipeline *m);
int pcc_multi_pipeline_size(pcc_multi_pipeline *m);
pcc_pipeline *pcc_multi_pipeline_member(pcc_multi_pipeline *m, int index); /* e.g. for pcc_pipeline_reserve */
int pcc_multi_pipeline_encode_host(pcc_multi_pipeline *m, const void *const *host_frames, const size_t *n_points,
                                   size_t n_frames, size_t stride, size_t rgb_offset, const pcc_params *params,
                                   pcc_bitstream *out);
int pcc_multi_pipeline_encode(pcc_multi_pipeline *m, const void *const *dev_frames, const size_t *n_points, size_t n_frames,
                              size_t stride, size_t rgb_offset, const pcc_params *params, pcc_bitstream *out);
const char *pcc_multi_pipeline_last_error(pcc_multi_pipeline *m);

/* ---- computeQualityMetric (apps/evaluate_compression quality_metrics_impl.hpp:82-239; quality_metrics.h:53-80) ----
 * cloud_a = original, cloud_b = decoded (host pointers).  Nearest neighbours on the GPU (uniform grid of cell size
 * `cell_hint`, e.g. the octree resolution; <= 0: chosen from the clouds), exact like the reference's KdTree;
 * among equally distant neighbours the lower index counts.  Non-finite points are left out of the searches. */
typedef struct pcc_quality {
  uint64_t in_point_count, out_point_count;
  float symm_rms, symm_hausdorff, left_hausdorff, right_hausdorff, left_rms, right_rms;
  double psnr_db;
  double psnr_yuv[3];
  float gpu_ms;
} pcc_quality;
int pcc_quality_metrics(pcc_ctx *ctx, const pcc_point_xyzrgb *cloud_a, size_t n_a, const pcc_point_xyzrgb *cloud_b,
                        size_t n_b, double cell_hint, pcc_quality *out);

/* ---- remove_outliers (codec.h:216-217, impl.hpp:1840-1866): pcl::RadiusOutlierRemoval on the GPU ----
 * keep[i] = 1 if at least `min_points` other points of the cloud lie within `radius` of point i (squared float distance
 * <= radius^2; non-finite points are dropped); min_points <= 0 keeps everything, as the reference does.  The caller
 * compacts (the reference returns the kept points in their original order). */
int pcc_remove_outliers(pcc_ctx *ctx, const pcc_point_xyzrgb *cloud, size_t n, int min_points, double radius, uint8_t *keep,
                        size_t *n_kept);

/* ---- encodePointCloudDeltaFrame / decodePointCloudDeltaFrame (codec.h:181-191, impl.hpp:787-1235): predictive frames ----
 * The P frame is simplified to voxel centres (simplifyPCloud, impl.hpp:318-403), both frames are cut into macroblocks
 * (octree leaves at octree_resolution * macroblock_size inside [0,1]^3, impl.hpp:411-434); a macroblock that exists in
 * both frames, passes the size and colour-variance gates (impl.hpp:453-521) and whose ICP converges (impl.hpp:544-567) is
 * coded as a rigid transform of the I frame's block (p_data); all other points are intra coded (i_data).  Clouds are
 * host pointers; macroblock trees, gates, one ICP per block and the cloud assembly run on the GPU.
 * PCL's IterativeClosestPoint is not part of the reference tree: transforms agree with PCL's to ICP convergence
 * accuracy, not bit for bit ("parity unpinned"). */
typedef struct pcc_delta_params {
  pcc_params codec;              /* the coder's configuration: octree_resolution, point_resolution, color_bit_resolution,
                                    color_coding_type, do_voxel_centroid, macroblock_size, do_icp_color_offset are used */
  int32_t icp_on_original;       /* argument of encodePointCloudDeltaFrame: skip the simplification */
  int32_t write_out_cloud;       /* argument: also build the predicted cloud (what the decoder will produce, before the
                                    intra part's own quantisation) */
  int32_t icp_max_iterations;    /* icp_max_iterations_ (codec.h: 50); <= 0: default */
  float icp_var_threshold;       /* icp_var_threshold_ (100); <= 0: default */
  float transformation_epsilon;  /* transformationepsilon_ (1e-8); <= 0: default */
} pcc_delta_params;

typedef struct pcc_delta_result {
  const uint8_t *i_data; size_t i_len;   /* i_coded_data: the intra coded residual points */
  const uint8_t *p_data; size_t p_len;   /* p_coded_data: chunks  u8 size | 3 x int16 key | 6 or 10 x int16 transform | [3 x int8 rgb] */
  const pcc_point_xyzrgb *out_cloud; size_t out_n;
  uint32_t macro_block_count, shared_macroblock_count, convergence_count;
  float shared_macroblock_percentage;              /* getMacroBlockPercentage() */
  float shared_macroblock_convergence_percentage;  /* getMacroBlockConvergencePercentage() */
  uint64_t n_intra_points, n_simplified;
  float gpu_ms;
} pcc_delta_result;

int pcc_encode_delta(pcc_ctx *ctx, const pcc_point_xyzrgb *i_cloud, size_t n_i, const pcc_point_xyzrgb *p_cloud, size_t n_p,
                     const pcc_delta_params *params, pcc_delta_result *out);
int pcc_decode_delta(pcc_ctx *ctx, const pcc_point_xyzrgb *i_cloud, size_t n_i, const uint8_t *i_stream, size_t i_len,
                     const uint8_t *p_stream, size_t p_len, const pcc_delta_params *params, pcc_cloud *out);

/* ---- normalize_pointclouds / restore_scaling for one group (codec.h:216-227, impl.hpp:1871-1986), host side ----
 * also reports the box in force for every cloud (bounding_boxes[k], impl.hpp:1928-1929): per_cloud_boxes = n_clouds x
 * {min x,y,z, max x,y,z}, or NULL */
int pcc_normalize_group_boxes(pcc_point_xyzrgb **clouds, const size_t *sizes, size_t n_clouds, double bb_expand_factor,
                              float bb_min[3], float bb_max[3], float *per_cloud_boxes);
int pcc_restore_scaling(pcc_point_xyzrgb *cloud, size_t n, const float bb_min[3], const float bb_max[3]);

#ifdef __cplusplus
}
#endif
#endif /* PCC_CODEC_H */
