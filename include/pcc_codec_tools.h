/*
 * pcc_codec_tools.h -- the part of libpcc_hip.so's C ABI that measurements, tests and tools use ON TOP OF the drop-in
 * boundary of pcc_codec.h: timings, the frame pipeline's pieces one by one, the building blocks of the host stages, the
 * device range coder's harness, and a pcc_debug_* block.  A caller of the codec (the class shim, the evaluation app, a frame
 * loop) needs none of it.  Same conventions as pcc_codec.h (plain C, library-owned out pointers, PCC_OK or a negative code).
 */
#ifndef PCC_CODEC_TOOLS_H
#define PCC_CODEC_TOOLS_H

#include "pcc_codec.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- contexts without a GPU ---- */
/* A context without a GPU: only the host stages work on it (pcc_entropy_encode, pcc_decode_intra).
 * Every GPU entry point returns PCC_ERR_STATE -- there is no CPU fallback for the hot path. */
pcc_ctx *pcc_create_host(void);

/* ---- timings ---- */
/* Per-kernel timing of the last hot-path run (HIP events on the context's stream). */
#define PCC_MAX_KERNEL_TIMES 64
typedef struct pcc_kernel_times {
  int32_t count;
  const char *name[PCC_MAX_KERNEL_TIMES];
  float ms[PCC_MAX_KERNEL_TIMES];
} pcc_kernel_times;

int pcc_get_kernel_times(pcc_ctx *ctx, pcc_kernel_times *out);
/* The same launches measured on the GPU itself: from the start of a launch's first workgroup to the end of its last
 * wave, on the device's real-time clock (what a kernel trace reports; the events above sit BETWEEN the launches and
 * add a few microseconds of their own to every short kernel).  Sort passes a frame did not need are left out. */
int pcc_get_kernel_spans(pcc_ctx *ctx, pcc_kernel_times *out);
/* ... and when each of those launches started, milliseconds after the first one: the distance between the starts of two
 * consecutive launches is what a launch costs its stream (its span plus the dispatch and the end-of-kernel write-back
 * that the span leaves out) -- the figure a kernel trace calls the kernel's duration. */
int pcc_get_kernel_span_starts(pcc_ctx *ctx, pcc_kernel_times *out);
/* wall time of the last pcc_entropy_encode on this context, microseconds: occupancy range coder, JPEG
 * stage, colour range coder, whole stage */
int pcc_get_host_times(pcc_ctx *ctx, double out_us[4]);
/* enable per-kernel HIP-event timing (off by default: events between launches cost a little) */
int pcc_set_profiling(pcc_ctx *ctx, int enabled);
/* test and measurement hooks of pcc_set_option (pcc_codec.h; none changes an output byte):
 *   "profile_events" (default 1): with pcc_set_profiling, also record HIP events between the launches
 *                 (pcc_get_kernel_times); 0 leaves only the launch spans on the GPU clock, so that the launches run
 *                 back to back as they do unprofiled.
 *   "force_pairs", "no_cell_ranks" (default 0): which key layout the sort is given (1: (code, index) pairs on small
 *                 frames; 2: the point index kept in the key although nothing reads it) / the full varying Morton code
 *                 sorted instead of cell ranks.
 *   "icp_waves" (default 0): delta path, one ICP kernel shape for every macroblock: 4 = a workgroup per block, 1 = a wave
 *                 per block; 0 = by block count.
 *   "rc_device_lanes" (default 0): the device range coder launched through THIS context (pcc_device_range_encode; a batch
 *                 or pipeline has the same option) codes one stream per LANE, 64 per wave, instead of one per wave. */
/* milliseconds of the last pcc_decode_intra_gpu: sequential host stages | upload + kernels + download | whole call */
int pcc_get_decode_times(pcc_ctx *ctx, double out_ms[3]);


/* ---- the pipeline's pieces, one by one ---- */
/* The same for a cloud in HOST memory (the reference's timed span starts there, eval.hpp:462-464): the points are
 * copied to the context's HBM arena asynchronously, the kernels start when they have arrived.  Ordinary (pageable)
 * memory is page-locked for the time of the copy; memory from pcc_host_alloc is used as it is.  `lane` (may be NULL)
 * is a stream shared by the contexts of one GPU that carries the uploads one after the other, at full PCIe rate,
 * while the kernels of earlier frames run; with NULL the copy is queued in front of the kernels on the context's own
 * stream.  The caller's buffer must stay untouched until pcc_hotpath_finish returns. */
typedef struct pcc_upload_lane pcc_upload_lane;
pcc_upload_lane *pcc_upload_lane_create(int device);
void pcc_upload_lane_destroy(pcc_upload_lane *lane);
int pcc_hotpath_launch_host(pcc_ctx *ctx, pcc_upload_lane *lane, const void *host_points, size_t n, size_t stride,
                            size_t rgb_offset, const pcc_params *params);
/* A context's GPU work runs on a stream of its own unless it borrows one.  Why a caller would care: the HIP runtime
 * spreads the streams of a process over four hardware queues, round robin in the order the streams are created, and
 * the frames in flight overlap best when every queue carries the same number of them (cfg2, saturated GPU stage: 12
 * streams as 3 3 3 3: 11 200 frames/s, 10 streams as 3 3 2 2: 10 400, 12 as 4 4 4 0: 10 000; tools/queue_balance.py).
 * With more contexts than frames in flight -- a context is held until its host stage is over -- the streams that
 * happen to be busy are not balanced; pcc_pipeline therefore creates one stream per GPU-stage thread, one after the
 * other, and lends it to whichever context the thread is driving.  Only between frames (no launch in flight on the
 * context); NULL gives the context its own stream back. */
typedef struct pcc_stream pcc_stream;
pcc_stream *pcc_stream_create(int device);
void pcc_stream_destroy(pcc_stream *stream);
int pcc_use_stream(pcc_ctx *ctx, pcc_stream *stream);
/* page-locked host memory for callers that fill their frames themselves (capture, file readers) */
void *pcc_host_alloc(size_t bytes);
void pcc_host_free(void *p);
/* The same for several frames at once (different contexts): the serial range-coder loops of the frames are
 * interleaved in one loop, which costs a fraction of the time per frame (each symbol is a chain of dependent
 * operations that leaves most of a core idle).  Bytes identical to separate pcc_entropy_encode calls.  Up to four frames share
 * a scalar loop; ten to sixteen go through AVX-512 lanes where the host CPU has them (else through scalar loops of four). */
#define PCC_MAX_FRAMES_AT_ONCE 16
int pcc_entropy_encode_many(int n, pcc_ctx *const ctx[], const pcc_hot_result *const hot[], const pcc_params *const prm[],
                            pcc_bitstream *const out[]);

/* the GPU stage alone (kernels + device->host hand-over), for capacity measurements */
int pcc_pipeline_gpu_stage_only(pcc_pipeline *p, const void *const *dev_frames, const size_t *n_points, size_t n_frames,
                                size_t stride, size_t rgb_offset, const pcc_params *params);
/* per-frame means of the last call, microseconds: launch, finish, entropy call wall time; then the four values of
 * pcc_get_host_times; out_us[7] = frames processed */
int pcc_pipeline_stats(pcc_pipeline *p, double out_us[8]);
/* CPU time (not wall time) the pipeline's threads spent in the same three calls, per-frame means, microseconds;
 * out_us[3] = frames processed.  Wall minus CPU = time asleep waiting for the GPU. */
int pcc_pipeline_cpu_times(pcc_pipeline *p, double out_us[4]);
/* HIP-event kernel times of the last call, summed over the frames that ran on a context with profiling
 * enabled: sums->ms[i] = total milliseconds of kernel sums->name[i], launches[i] = number of launches
 * (arrays of PCC_MAX_KERNEL_TIMES), *frames = profiled frames */
int pcc_pipeline_kernel_times(pcc_pipeline *p, pcc_kernel_times *sums, int32_t *launches, int32_t *frames);

/* ---- the delta path's per-block decisions ---- */
/* one macroblock of the P frame as the GPU judged it (inspection / tests) */
typedef struct pcc_delta_block {
  int32_t i_block;            /* index of the I frame's macroblock with the same key, -1: none */
  uint32_t n_p, n_i;          /* points in the P / I block */
  int32_t do_icp;             /* passed the gates */
  int32_t converged;          /* ICP converged and fitness < 2 * point_resolution */
  int32_t iterations;
  int8_t rgb_offsets[4];
  uint16_t key[4];            /* x, y, z */
  float fitness;
  float rt[16];               /* final transformation, row-major */
} pcc_delta_block;

int pcc_delta_blocks(pcc_ctx *ctx, const pcc_delta_block **blocks, size_t *n); /* of the last pcc_encode_delta */

/* ---- the static range coder for MANY independent streams on the GPU (csrc/pcc_rc_device.hip) ----
 * One wave per stream, coder state in scalar registers: roughly ten times slower per stream than a CPU core, but a
 * thousand streams run side by side -- for pipelines whose host has fewer cores than the GPUs can feed.  Same bytes as
 * pcc_host_range_encode.  Host pointers in and out (out[i]: room for 1028 + n[i] + n[i]/2 + 64 bytes); this entry point
 * is the measurement / test harness of the kernel; the frame pipeline reaches the coder through pcc_entropy_batch. */
int pcc_device_range_encode(pcc_ctx *ctx, int n_streams, const uint8_t *const *in, const size_t *n, uint8_t *const *out,
                            size_t *out_len, float *gpu_ms);

/* ---- the entropy stage of MANY frames with the range coders on the GPU ----
 * For hosts with fewer CPU cores than their GPUs can feed (the north star keeps the serial coder on the host, and with
 * 16 cores per GPU that is the faster place).  pcc_entropy_batch_add copies what a frame's entropy stage needs out of
 * the frame's hot-path products -- so the context that produced it is free again -- and puts the colour JPEG together on the host;
 * pcc_entropy_batch_flush range-codes every stream of the batch on the GPU (one wave per stream, ~0.1 s per flush
 * whatever the batch size: use batches of hundreds of frames) and assembles the bitstreams -- byte-identical to
 * pcc_entropy_encode.  out[i] (i-th frame added) stays valid until the next flush. */
typedef struct pcc_entropy_batch pcc_entropy_batch;
pcc_entropy_batch *pcc_entropy_batch_create(int device, size_t max_frames);
void pcc_entropy_batch_destroy(pcc_entropy_batch *b);
size_t pcc_entropy_batch_size(pcc_entropy_batch *b);
size_t pcc_entropy_batch_capacity(pcc_entropy_batch *b);
int pcc_entropy_batch_add(pcc_entropy_batch *b, const pcc_hot_result *hot, const pcc_params *params); /* index, or < 0 */
int pcc_entropy_batch_flush(pcc_entropy_batch *b, pcc_bitstream *out, size_t out_capacity, size_t *n_out);
const char *pcc_entropy_batch_last_error(pcc_entropy_batch *b);
/* options of the batch's own context: "rc_device_lanes" (which form of the device range coder its flushes launch) */
int pcc_entropy_batch_set_option(pcc_entropy_batch *b, const char *name, int value);

/* ---- building blocks of the host stages (serial by nature; exposed for tests and tools) ---- */
/* pcl::StaticRangeCoder::encodeCharVectorToStream / decodeStreamToCharVector (impl.hpp:1694 / :1778).
 * encode: writes at most out_cap bytes, returns the encoded size (or 0 if out_cap is too small). */
size_t pcc_host_range_encode(const uint8_t *in, size_t n, uint8_t *out, size_t out_cap);
size_t pcc_host_range_decode(const uint8_t *in, size_t in_len, uint8_t *out, size_t n);
/* The same coder for up to sixteen independent vectors in ONE call -- how the entropy stage codes the streams of the frames
 * it holds (a lone coder is a chain of dependent operations and leaves most of a core idle): up to four share a scalar loop,
 * ten and more go through the lanes of AVX-512 registers where the CPU has them.  Every out[i] gets exactly the bytes
 * pcc_host_range_encode gives for in[i]; out_len[i] = encoded size, 0 if out_cap[i] is too small.  Returns PCC_OK, or
 * PCC_ERR_ARG for count outside 1..16. */
int pcc_host_range_encode_many(int count, const uint8_t *const *in, const size_t *n, uint8_t *const *out,
                               const size_t *out_cap, size_t *out_len);
/* JPEGWriter::writeJPEG / JPEGReader::readJPEG (jpeg_io.hpp:211-330 / 90-192), RGB, 4:2:0 */
size_t pcc_host_jpeg_encode(const uint8_t *rgb, int w, int h, int quality, uint8_t *out, size_t out_cap);
int pcc_host_jpeg_decode(const uint8_t *jpg, size_t len, uint8_t *rgb, size_t rgb_cap, int *w, int *h);
/* SnakeGridMapping iterator position (snake_grid_mapping.h:46-71) in closed form */
uint32_t pcc_host_snake_position(uint32_t i, uint32_t w, uint32_t h);

/* RigidTransformCoding::compressRigidTransform / deCompressRigidTransform (rigid_transform_coding_impl.hpp:63-203):
 * row-major 4x4 -> 6 int16 (quaternion + translation) or 10 (two rotation rows, sign word, translation) */
size_t pcc_host_rigid_compress(const float tr[16], int16_t *comp_out, size_t cap);
int pcc_host_rigid_decompress(const int16_t *comp, size_t count, float tr_out[16]);


/* ---- pcc_debug_*: developer aids, no stability promise ---- */
/* the sort geometry of the last frame whose state came back: {sort passes, code bits that were sorted, varying Morton bits,
 * key bits per axis below the cell ranks, bytes per key and pass} */
int pcc_debug_sort_plan(pcc_ctx *ctx, int32_t out[5]);
/* the CPUs entropy thread `worker` of a pipeline may run on, lowest first; returns how many there are (at most `cap` are
 * written), -1 for a bad argument */
int pcc_debug_pipeline_cpus(pcc_pipeline *p, int worker, int *out, int cap);
/* 1 if the host range coder's AVX-512 path is in use on this machine (tests skip its cases where it is not) */
int pcc_debug_host_rc_wide(void);
/* where a GPU hangs off the host (csrc/pcc_numa.h; what the pipelines choose their cores by): the device's PCI address as the
 * runtime prints it ("0000:c1:00.0"; cap >= 13), and the host's NUMA node nearest to it, -1 if nobody says --
 * <sysfs_root>/bus/pci/devices/<address>/numa_node first (sysfs_root NULL = "/sys"), the runtime's HostNumaId attribute otherwise */
int pcc_debug_device_pci_bus_id(int device, char *out, int cap);
int pcc_debug_device_numa_node(int device, const char *sysfs_root);
/* the planning alone, no GPU needed: `n_devices` pipelines whose GPUs sit at the PCI addresses `pci[d]`, a process that may
 * run on `cpus[0..n_cpus)` (ascending), the topology read from `sysfs_root`.  Writes the node of pipeline d's share to
 * node_out[d] (-1: shares are plain n-ths of the cores) and its cores to cores_out[d * cap .. ), their number to n_cores_out[d]
 * (at most `cap` are written).  PCC_OK or PCC_ERR_ARG. */
int pcc_debug_numa_plan(const char *sysfs_root, const char *const *pci, int n_devices, const int *cpus, int n_cpus,
                        int *node_out, int *cores_out, int *n_cores_out, int cap);
/* the NUMA node the page behind `p` lives on right now, -1 if the kernel does not say (not touched yet, no NUMA): where a
 * page-locked landing buffer went */
int pcc_debug_address_node(const void *p);

#ifdef __cplusplus
}
#endif
#endif /* PCC_CODEC_TOOLS_H */
