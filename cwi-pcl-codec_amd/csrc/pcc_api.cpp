// pcc_api.cpp -- the C ABI declared in include/pcc_codec.h (the boundary) and include/pcc_codec_tools.h (measurement, tests): context, HBM arena, launch of the
// HIP hot path, device->host hand-over and the host entropy stage.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/prctl.h>
#include <time.h>

#include <algorithm>
#include <map>
#include <mutex>
#include <new>
#include <string>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/pcc_codec_tools.h"
#include "pcc_device.h"
#include "pcc_host_codec.h"
#include "pcc_kernels.h"
#include "pcc_quality.h"
#include "pcc_decode.h"
#include "pcc_delta.h"
#include "pcc_rc_device.h"
#include "pcc_dev.h"
#include "pcc_numa.h"

using namespace pcc;

namespace {

template <typename T>
struct DevBuf {  // grow-only device allocation
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t count) {
    if (count <= cap) return hipSuccess;
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e == hipSuccess) cap = count;
    return e;
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

template <typename T>
struct PinnedBuf {  // grow-only page-locked host allocation (D2H landing zone)
  T* p = nullptr;
  size_t cap = 0;
  hipError_t ensure(size_t count) {
    if (count <= cap) return hipSuccess;
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
    const size_t want = std::max(count, (4096 + sizeof(T) - 1) / sizeof(T));  // at least a page (not 4096 elements: a FrameState is 3 KB)
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), want * sizeof(T), hipHostMallocDefault);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (p) (void)hipHostFree(p);
    p = nullptr;
    cap = 0;
  }
};

}  // namespace

struct pcc_ctx {
  int device = 0;
  hipStream_t stream = nullptr;      // where this context's GPU work goes: its own stream, or a borrowed one (pcc_use_stream)
  hipStream_t own_stream = nullptr;
  hipEvent_t ev_begin = nullptr, ev_end = nullptr;
  hipEvent_t ev_wait = nullptr;  // blocking-sync event: a host thread that waits for the GPU sleeps instead of
                                 // spinning, so its core is free for the entropy stage of another frame
  hipEvent_t ev_h2d = nullptr;   // the frame's points have arrived (recorded on an upload lane's stream)
  const void* locked_host = nullptr;  // host range page-locked for the frame in flight (released by pcc_hotpath_finish)
  std::string err;
  bool profiling = false;
  bool profile_events = true;  // with profiling: HIP events between the launches as well (they lengthen what they measure)
  KernelTimer timer;
  std::vector<std::pair<const char*, float>> times;
  // device-side launch spans (first workgroup start .. last wave end on the GPU's real-time clock), profiling only
  DevBuf<unsigned long long> d_spans;
  PinnedBuf<unsigned long long> h_spans;
  std::vector<const char*> span_names;
  std::vector<std::pair<const char*, float>> span_times;
  std::vector<float> span_starts;  // ms from the first launch's first workgroup to this launch's first workgroup
  double wall_clock_khz = 100000.0;

  // HBM arena (see pcc_device.h for the layout)
  DevBuf<uint8_t> d_points;  // only for the host-input entry point
  DevBuf<uint64_t> d_boxes;   // eight self-describing words per chunk (k_boxes_events)
  uint32_t frame_seq = 0;     // sequence number of the last enqueued frame; stamps its chunk boxes
  DevBuf<FrameState> d_state;
  DevBuf<uint64_t> d_keys_a, d_keys_b, d_leaf_code;
  DevBuf<uint16_t> d_hist_rows;
  DevBuf<uint32_t> d_digit_tot, d_tile_prefix0, d_leaf_start, d_leaf_base, d_idx_a, d_idx_b;
  DevBuf<uint32_t> d_idx2_a, d_idx2_b, d_leaf_hi;  // trees deeper than 21 levels only (two-word codes): allocated when one comes by
  bool deep_hint = false;                          // the frame before was one: enqueue the deep kernels straight away
  int force_pairs = 0;                             // test hooks (pcc_set_option): which key layout the sort is given
  bool no_cell_ranks = false;
  bool rc_lanes = false;                           // device range coder of this context / batch: one lane per stream instead of one wave (option "rc_device_lanes")
  int icp_waves = 0;                               // delta path: 4 = a workgroup per macroblock, 1 = a wave per macroblock, 0 = by block count
  DevBuf<uint8_t> d_leaf_t, d_bgr, d_centroid, d_image, d_sync;
  // the per-MCU-row Huffman records and, right behind them, the occupancy stream: what the host stage needs of a
  // colour frame is one contiguous piece of HBM and comes back in ONE copy (a copy costs some 40 us to set up)
  DevBuf<uint8_t> d_occ;
  size_t out_off = 0;  // bytes of the records region of the frame in flight = offset of its occupancy bytes
  DevBuf<float> d_simplified;  // 4 floats per leaf
  DevBuf<uint32_t> d_lines;    // colour coding type 2: directory (4 words per strip) | the strips' bit strings
  PinnedBuf<uint32_t> h_lines;
  size_t lines_dir_words = 0;  // of the frame in flight
  DevBuf<int16_t> d_coefs;     // JPEG coefficients of the snake image
  DevBuf<JpegHuffTables> d_huff;
  bool huff_uploaded = false;

  // quality metric
  DevBuf<uint8_t> d_qa, d_qb;
  DevBuf<unsigned long long> d_qkeys;
  DevBuf<uint32_t> d_qheads, d_qnext, d_qidx;
  DevBuf<float> d_qd2;
  DevBuf<double> d_qpart;
  std::vector<double> h_qpart;

  // inter-frame path: sub-contexts for the simplified cloud and the two macroblock trees, block scratch
  pcc_ctx* sub[4] = {nullptr, nullptr, nullptr, nullptr};  // simplification, I blocks, P blocks, residual intra coder
  DevBuf<uint8_t> d_delta_i, d_delta_p, d_delta_intra, d_delta_out;
  DevBuf<uint8_t> d_rc;  // pcc_device_range_encode: inputs, outputs, lengths, jobs
  DevBuf<uint64_t> d_ifull, d_pfull;
  DevBuf<float4> d_ixyzc, d_pxyzc, d_cur;
  DevBuf<uint32_t> d_nn, d_dst_intra, d_dst_out, d_fake_start, d_work, d_order, d_counts;
  DevBuf<BlockResult> d_blocks;
  DevBuf<float> d_mdec;
  std::vector<BlockResult> h_blocks;
  std::vector<uint32_t> h_pstart, h_istart, h_dst_intra, h_dst_out;
  std::vector<uint64_t> h_ifull;
  std::vector<float> h_mdec;
  Bytes p_stream, i_stream;
  PinnedBuf<pcc_point_xyzrgb> delta_cloud;  // predicted (+ decoded intra) points of the last delta call: page-locked, the
  size_t delta_cloud_n = 0;                 // device-to-host copy of a million points is a DMA, not a staged copy

  // host landing buffers
  PinnedBuf<FrameState> h_state;
  PinnedBuf<uint8_t> h_occ, h_bgr, h_centroid, h_image;
  PinnedBuf<float> h_simplified;
  PinnedBuf<int16_t> h_coefs;
  int jpeg_on_gpu = 2;  // 0 host JPEG, 1 coefficients from the GPU, 2 Huffman-coded MCU rows from the GPU
  bool copy_image = true;
  // frames that start in host memory: 16 bytes per point (x, y, z, colour) are packed into a page-locked buffer of the
  // context and uploaded instead of the caller's 32-byte points (option "pack_upload"; PCC_PACK_UPLOAD=1 makes it the default)
  bool pack_upload = false;
  PinnedBuf<uint8_t> h_pack;
  std::vector<pcc_point_xyzrgb> out_cloud;   // getOutputCloud()
  PointVec dec_points;  // decodePointCloud()
  // decodePointCloud with the data-parallel half on the GPU (pcc_decode_intra_gpu)
  FrameStreams dec_streams;
  LeafParents dec_parents;
  BaselineJpeg::JpegCoefs dec_coefs;
  DevBuf<uint8_t> d_dec;             // one arena: prefixes | first | bits | centroid | colours | coefficients | planes
  DevBuf<uint8_t> d_dec_points;
  PinnedBuf<uint8_t> h_dec_stage;    // the same inputs on the host side, page-locked: one upload
  PinnedBuf<pcc_point_xyzrgb> h_dec_points;
  double dec_ms[4] = {0, 0, 0, 0};   // last pcc_decode_intra_gpu: sequential host stages, upload + kernels + download, total
  Bytes bitstream;
  double host_us[4] = {0, 0, 0, 0};  // last entropy stage: occupancy coder, JPEG, colour coder, total

  // frame in flight
  HotPathArgs args{};          // what was enqueued (kept so that a frame that needs more sort passes can be re-run)
  int pass_hint = kMaxPasses;  // sort passes to enqueue: the previous frame's need (sequences are coherent)
  bool launched = false;
  size_t n = 0;
  pcc_params params{};
  bool simplified_valid = false;
  size_t last_L = 0;
  double usual_wait_ns[3] = {0, 0, 0};  // how long the waits took lately: kernels of a frame, copies, anything else
  pcc_hot_result last_hot{};
  // the occupancy byte counts of the last finished frame.  They arrive inside the pinned FrameState, which the NEXT
  // pcc_hotpath_launch overwrites asynchronously; this copy is only rewritten by pcc_hotpath_finish, like the other
  // products a pcc_hot_result points to
  uint32_t occ_hist[256] = {0};
};

namespace {

int fail(pcc_ctx* c, int code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}
int hip_fail(pcc_ctx* c, hipError_t e, const char* what) {
  return fail(c, PCC_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define PCC_HIP(call)                                  \
  do {                                                 \
    hipError_t e_ = (call);                            \
    if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
  } while (0)

#define PCC_NEED_GPU()                                                                              \
  do {                                                                                             \
    if (ctx->device < 0) return fail(ctx, PCC_ERR_STATE, "host-only context: the hot path needs a GPU"); \
  } while (0)

// bytes of the Huffman records of a cloud of n points (one record per MCU row of the tallest possible snake image)
size_t tiles_region(size_t n) { return ((n / 256 + 1 + 15) / 16) * (size_t)kJpegTileWords * sizeof(uint32_t); }

int reserve(pcc_ctx* ctx, size_t n) {
  const size_t tiles = (n + kTile - 1) / kTile;
  const size_t stiles = (n + kSortTile - 1) / kSortTile;
  if (ctx->d_boxes.cap < 8 * tiles) {  // a fresh array must not hold anything that looks like a chunk box of a coming frame
    PCC_HIP(ctx->d_boxes.ensure(8 * tiles));
    PCC_HIP(hipMemsetAsync(ctx->d_boxes.p, 0, ctx->d_boxes.cap * sizeof(uint64_t), ctx->stream));
  }
  PCC_HIP(ctx->d_state.ensure(1));
  PCC_HIP(ctx->d_keys_a.ensure(n));
  PCC_HIP(ctx->d_keys_b.ensure(n));
  PCC_HIP(ctx->d_idx_a.ensure(n));
  PCC_HIP(ctx->d_idx_b.ensure(n));
  PCC_HIP(ctx->d_hist_rows.ensure(stiles * kMaxPasses * kMaxBins));  // a row per sort tile
  PCC_HIP(ctx->d_digit_tot.ensure((size_t)kMaxPasses * kMaxBins));
  PCC_HIP(ctx->d_tile_prefix0.ensure(stiles * kMaxBins));
  PCC_HIP(ctx->d_sync.ensure(sync_area_bytes((uint32_t)n, kMaxPasses)));
  PCC_HIP(ctx->d_leaf_start.ensure(n + 1));
  PCC_HIP(ctx->d_leaf_code.ensure(n));
  PCC_HIP(ctx->d_leaf_base.ensure(n));
  PCC_HIP(ctx->d_leaf_t.ensure(n));
  PCC_HIP(ctx->d_occ.ensure(tiles_region(n) + n * (size_t)kMaxDepthDeep + 64));  // records + worst case B = L * D
  PCC_HIP(ctx->d_bgr.ensure(3 * n + 16));
  PCC_HIP(ctx->d_centroid.ensure(3 * n + 16));
  PCC_HIP(ctx->d_image.ensure(3 * 256 * (n / 256 + 1) + 16));
  PCC_HIP(ctx->d_simplified.ensure(4 * n));
  PCC_HIP(ctx->d_coefs.ensure((size_t)16 * ((n / 256 + 1 + 15) / 16) * 6 * 64 + 64));
  PCC_HIP(ctx->d_huff.ensure(1));
  if (!ctx->huff_uploaded) {
    JpegHuffTables t;
    BaselineJpeg::huffman_tables(t.dc, t.ac);
    PCC_HIP(hipMemcpy(ctx->d_huff.p, &t, sizeof(t), hipMemcpyHostToDevice));
    ctx->huff_uploaded = true;
  }
  PCC_HIP(ctx->h_state.ensure(1));
  return PCC_OK;
}

// Wait for everything enqueued on the context's stream so far WITHOUT burning the core: the host entropy stage of other
// frames needs it.  hipEventSynchronize spins in user space even on a hipEventBlockingSync event (measured: 0.71 ms of
// CPU time for 0.77 ms of waiting per frame), so the default is to poll the event between short sleeps; the few
// microseconds of extra latency are hidden by the other frames in flight.  PCC_WAIT=event restores the runtime's wait.
int wait_stream(pcc_ctx* ctx, int site = 2) {
  static const int mode = [] {
    const char* e = dev_env("PCC_WAIT");
    return (e && !strcmp(e, "event")) ? 1 : 0;
  }();
  PCC_HIP(hipEventRecord(ctx->ev_wait, ctx->stream));
  if (mode == 1) {
    PCC_HIP(hipEventSynchronize(ctx->ev_wait));
    return PCC_OK;
  }
  static thread_local bool slack_set = false;
  if (!slack_set) {  // default timer slack is 50 us: ask for precise short sleeps on this thread
    (void)prctl(PR_SET_TIMERSLACK, 1000UL, 0UL, 0UL, 0UL);
    slack_set = true;
  }
  // A sleep costs a few microseconds of CPU (two kernel crossings and a context switch), so few long sleeps beat many
  // short ones: sleep through most of what this wait took the last times on this context (kernels of a frame: some
  // hundred microseconds; copies: some ten), then look every 30 us.
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  double& usual = ctx->usual_wait_ns[site];
  long ns = usual > 60000.0 ? (long)(0.7 * usual) : 5000;
  for (;;) {
    const hipError_t q = hipEventQuery(ctx->ev_wait);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return hip_fail(ctx, q, "hipEventQuery");
    timespec ts{0, ns};
    nanosleep(&ts, nullptr);
    ns = ns < 30000 ? std::min(2 * ns, 30000L) : 30000;
  }
  timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double took = (double)(t1.tv_sec - t0.tv_sec) * 1e9 + (double)(t1.tv_nsec - t0.tv_nsec);
  usual = usual == 0.0 ? took : 0.75 * usual + 0.25 * took;
  return PCC_OK;
}

// the kernel sequence of one frame + the FrameState read-back, all asynchronous on the context's stream
int enqueue(pcc_ctx* ctx) {
  if (++ctx->frame_seq == 0) ctx->frame_seq = 1;
  ctx->args.frame_seq = ctx->frame_seq;
  ctx->args.spans = nullptr;
  ctx->args.span_names = nullptr;
  if (ctx->profiling) {
    PCC_HIP(ctx->d_spans.ensure(kSpanWords));
    PCC_HIP(ctx->h_spans.ensure(kSpanWords));
    PCC_HIP(hipMemsetAsync(ctx->d_spans.p, 0xFF, kSpanWords * sizeof(unsigned long long), ctx->stream));
    ctx->span_names.clear();
    ctx->args.spans = ctx->d_spans.p;
    ctx->args.span_names = &ctx->span_names;
  }
  PCC_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
  if (ctx->profiling) ctx->timer.reset(); else ctx->times.clear();
  launch_hot_path(ctx->args, ctx->stream, (ctx->profiling && ctx->profile_events) ? &ctx->timer : nullptr);
  {
    const hipError_t le = hipGetLastError();
    if (le != hipSuccess) return hip_fail(ctx, le, "kernel launch");
  }
  PCC_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->h_state.p, ctx->d_state.p, sizeof(FrameState), hipMemcpyDeviceToHost, ctx->stream));
  if (ctx->profiling)
    PCC_HIP(hipMemcpyAsync(ctx->h_spans.p, ctx->d_spans.p, kSpanWords * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
  return PCC_OK;
}

// ---- host ranges page-locked on behalf of frames in flight ----
// A frame that arrives in ordinary (pageable) host memory is page-locked for the time of its upload, so that the
// copy is an asynchronous DMA at PCIe speed (hipHostRegister of 32 MB: 0.075 ms, tools/ubench/h2d.cpp; an upload from
// pageable memory blocks the calling thread for the 0.6 ms it takes).  The same buffer may be in flight on several
// contexts at once (a sequence that repeats its frames), hence the reference counts.
struct LockedRange { size_t bytes; int refs; };
std::mutex g_lock_mu;
std::map<const void*, LockedRange> g_locked;

// 0: the range is (now) page-locked and must be released with unlock_host_range; 1: it was page-locked by the caller;
// 2: could not be locked (the copy goes the pageable way)
int lock_host_range(const void* p, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_lock_mu);
  auto it = g_locked.find(p);
  if (it != g_locked.end()) {
    if (bytes > it->second.bytes) return 2;
    ++it->second.refs;
    return 0;
  }
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess && attr.type == hipMemoryTypeHost) return 1;
  (void)hipGetLastError();  // an unknown pointer is reported as an error
  if (hipHostRegister(const_cast<void*>(p), bytes, hipHostRegisterDefault) != hipSuccess) {
    (void)hipGetLastError();
    return 2;
  }
  g_locked[p] = LockedRange{bytes, 1};
  return 0;
}
void unlock_host_range(const void* p) {
  std::lock_guard<std::mutex> lk(g_lock_mu);
  auto it = g_locked.find(p);
  if (it == g_locked.end()) return;
  if (--it->second.refs == 0) {
    (void)hipHostUnregister(const_cast<void*>(p));
    g_locked.erase(it);
  }
}

}  // namespace

// One stream that carries the host-to-device copies of a GPU, one after the other: copies issued side by side on
// several streams share the link and finish later in total (51 GB/s with eight in flight against 56 GB/s one at a
// time, tools/ubench/h2d.cpp), and a frame's kernels should start when ITS points are there, not when everybody's are.
struct pcc_stream {
  int device = 0;
  hipStream_t stream = nullptr;
};

struct pcc_upload_lane {
  int device = 0;
  hipStream_t stream = nullptr;
  std::mutex mu;
};

extern "C" {

// What this binary is: the product says gfx950; the same sources compiled for the CPU wave64 executor of tests/emu say so
// too (bench.py and smoke() refuse a library that is not the gfx950 one: a line timed on the executor must not be
// mistaken for a measurement), and so do the developer builds.
const char* pcc_version(void) {
#if defined(PCC_EMU)
  return "pcc_emu 0.1 (CPU wave64 executor: test infrastructure, not the product)";
#elif defined(PCC_DEV)
  return "pcc_hip 0.1 (gfx950, dev build)";
#elif defined(PCC_WAVE_OPS_SHFL)
  return "pcc_hip 0.1 (gfx950, shfl bisect build)";
#elif defined(PCC_KTIME)
  return "pcc_hip 0.1 (gfx950, ktime developer build)";
#else
  return "pcc_hip 0.1 (gfx950)";
#endif
}

pcc_ctx* pcc_create(int device) {
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  pcc_ctx* c = new pcc_ctx();
  c->device = device;
  { const char* e = dev_env("PCC_PACK_UPLOAD"); c->pack_upload = e && e[0] == '1'; }
  { const char* e = dev_env("PCC_RC_DEVICE"); c->rc_lanes = e && !strcmp(e, "lanes"); }
  {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device) == hipSuccess && khz > 0) c->wall_clock_khz = (double)khz;
  }
  if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess || ((c->stream = c->own_stream), false) ||
      hipEventCreate(&c->ev_begin) != hipSuccess || hipEventCreate(&c->ev_end) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_wait, hipEventBlockingSync | hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_h2d, hipEventDisableTiming) != hipSuccess) {
    delete c;
    return nullptr;
  }
  return c;
}

// ---- where a GPU hangs off the host (pcc_numa.h): asked by the pipelines when they choose cores ----
// "0000:c1:00.0"-style PCI address of a device, as the runtime prints it
int pcc_debug_device_pci_bus_id(int device, char* out, int cap) {
  int count = 0;
  if (!out || cap < 13 || hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) {
    (void)hipGetLastError();
    return PCC_ERR_ARG;
  }
  if (hipDeviceGetPCIBusId(out, cap, device) != hipSuccess) {
    (void)hipGetLastError();
    out[0] = 0;
    return PCC_ERR_HIP;
  }
  return PCC_OK;
}
// The host's NUMA node nearest to a device, -1 if nobody says: sysfs through the PCI address
// (/sys/bus/pci/devices/<address>/numa_node) first, the runtime's own answer (hipDeviceAttributeHostNumaId) where sysfs has
// none.  `sysfs_root` NULL = "/sys".
int pcc_debug_device_numa_node(int device, const char* sysfs_root) {
  char bdf[64];
  if (pcc_debug_device_pci_bus_id(device, bdf, (int)sizeof(bdf)) == PCC_OK) {
    const int node = pcc::numa::pci_numa_node(sysfs_root && *sysfs_root ? sysfs_root : "/sys", bdf);
    if (node >= 0) return node;
    if (sysfs_root && *sysfs_root) return -1;  // a made-up tree (tests): what it does not say stays unknown
  }
  int node = -1;
  if (hipDeviceGetAttribute(&node, hipDeviceAttributeHostNumaId, device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  return node >= 0 ? node : -1;
}
// the node the page behind `p` is on right now (-1: not touched yet, or the kernel does not say): where a landing buffer went
int pcc_debug_address_node(const void* p) { return p ? pcc::numa::node_of_address(p) : -1; }

pcc_ctx* pcc_create_host(void) {
  pcc_ctx* c = new pcc_ctx();
  c->device = -1;
  return c;
}

void pcc_destroy(pcc_ctx* c) {
  if (!c) return;
  if (c->device < 0) {
    delete c;
    return;
  }
  (void)hipSetDevice(c->device);
  if (dev_env("PCC_WAIT_STATS") && c->usual_wait_ns[0] > 0)
    fprintf(stderr, "[pcc_ctx %p] usual waits: kernels %.0f us, copies %.0f us, other %.0f us\n", (void*)c, c->usual_wait_ns[0] / 1e3,
            c->usual_wait_ns[1] / 1e3, c->usual_wait_ns[2] / 1e3);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->locked_host) { unlock_host_range(c->locked_host); c->locked_host = nullptr; }
  c->d_dec.release(); c->d_dec_points.release(); c->h_dec_stage.release(); c->h_dec_points.release();
  c->delta_cloud.release(); c->d_points.release(); c->d_spans.release(); c->h_spans.release(); c->d_boxes.release(); c->d_state.release(); c->d_keys_a.release(); c->d_keys_b.release(); c->d_idx_a.release(); c->d_idx_b.release(); c->d_idx2_a.release(); c->d_idx2_b.release(); c->d_leaf_hi.release();
  c->d_leaf_code.release(); c->d_hist_rows.release(); c->d_digit_tot.release(); c->d_tile_prefix0.release(); c->d_sync.release();
  c->d_leaf_start.release(); c->d_leaf_base.release(); c->d_leaf_t.release(); c->d_occ.release(); c->d_bgr.release();
  c->d_centroid.release(); c->d_image.release(); c->d_simplified.release(); c->d_coefs.release(); c->h_coefs.release(); c->d_lines.release(); c->h_lines.release();
  c->d_huff.release();
  c->d_qa.release(); c->d_qb.release(); c->d_qkeys.release(); c->d_qheads.release(); c->d_qnext.release();
  c->d_qidx.release(); c->d_qd2.release(); c->d_qpart.release();
  c->h_state.release(); c->h_occ.release(); c->h_bgr.release(); c->h_centroid.release(); c->h_image.release();
  c->h_simplified.release(); c->h_pack.release();
  for (pcc_ctx*& sc : c->sub) { if (sc) pcc_destroy(sc); sc = nullptr; }
  c->d_delta_i.release(); c->d_delta_p.release(); c->d_delta_intra.release(); c->d_delta_out.release(); c->d_ifull.release();
  c->d_pfull.release(); c->d_ixyzc.release(); c->d_pxyzc.release(); c->d_cur.release(); c->d_nn.release(); c->d_dst_intra.release();
  c->d_rc.release(); c->d_dst_out.release(); c->d_fake_start.release(); c->d_work.release(); c->d_order.release(); c->d_counts.release(); c->d_blocks.release(); c->d_mdec.release();
  if (c->ev_begin) (void)hipEventDestroy(c->ev_begin);
  if (c->ev_end) (void)hipEventDestroy(c->ev_end);
  if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
  if (c->ev_h2d) (void)hipEventDestroy(c->ev_h2d);
  if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
  delete c;
}

const char* pcc_last_error(pcc_ctx* c) { return c ? c->err.c_str() : "no context (no usable HIP device?)"; }

int pcc_set_profiling(pcc_ctx* ctx, int enabled) {
  if (!ctx) return PCC_ERR_ARG;
  ctx->profiling = enabled != 0;
  if (enabled && ctx->device >= 0) {  // the buffers of the launch spans now, not inside somebody's timed region
    PCC_HIP(hipSetDevice(ctx->device));
    PCC_HIP(ctx->d_spans.ensure(kSpanWords));
    PCC_HIP(ctx->h_spans.ensure(kSpanWords));
  }
  return PCC_OK;
}

int pcc_set_option(pcc_ctx* ctx, const char* name, int value) {
  if (!ctx || !name) return PCC_ERR_ARG;
  if (!strcmp(name, "jpeg_on_gpu")) ctx->jpeg_on_gpu = value < 0 ? 0 : (value > 2 ? 2 : value);
  else if (!strcmp(name, "copy_image")) ctx->copy_image = value != 0;
  else if (!strcmp(name, "pack_upload")) ctx->pack_upload = value != 0;
  else if (!strcmp(name, "profile_events")) ctx->profile_events = value != 0;
  else if (!strcmp(name, "force_pairs")) ctx->force_pairs = value;          // test hooks: see launch_frame
  else if (!strcmp(name, "no_cell_ranks")) ctx->no_cell_ranks = value != 0;
  else if (!strcmp(name, "icp_waves")) ctx->icp_waves = value;               // test hook: one ICP kernel shape for every macroblock
  else if (!strcmp(name, "rc_device_lanes")) ctx->rc_lanes = value != 0;     // the device range coder of THIS context: one lane per stream instead of one wave
  else return fail(ctx, PCC_ERR_ARG, std::string("unknown option ") + name);
  return PCC_OK;
}

int pcc_get_kernel_times(pcc_ctx* ctx, pcc_kernel_times* out) {
  if (!ctx || !out) return PCC_ERR_ARG;
  out->count = (int32_t)std::min(ctx->times.size(), (size_t)PCC_MAX_KERNEL_TIMES);
  for (int i = 0; i < out->count; ++i) {
    out->name[i] = ctx->times[i].first;
    out->ms[i] = ctx->times[i].second;
  }
  return PCC_OK;
}

int pcc_get_kernel_spans(pcc_ctx* ctx, pcc_kernel_times* out) {
  if (!ctx || !out) return PCC_ERR_ARG;
  out->count = (int32_t)std::min(ctx->span_times.size(), (size_t)PCC_MAX_KERNEL_TIMES);
  for (int i = 0; i < out->count; ++i) {
    out->name[i] = ctx->span_times[i].first;
    out->ms[i] = ctx->span_times[i].second;
  }
  return PCC_OK;
}

int pcc_get_kernel_span_starts(pcc_ctx* ctx, pcc_kernel_times* out) {
  if (!ctx || !out) return PCC_ERR_ARG;
  out->count = (int32_t)std::min(ctx->span_starts.size(), (size_t)PCC_MAX_KERNEL_TIMES);
  for (int i = 0; i < out->count; ++i) {
    out->name[i] = ctx->span_times[(size_t)i].first;
    out->ms[i] = ctx->span_starts[(size_t)i];
  }
  return PCC_OK;
}

int pcc_get_host_times(pcc_ctx* ctx, double out_us[4]) {
  if (!ctx || !out_us) return PCC_ERR_ARG;
  for (int i = 0; i < 4; ++i) out_us[i] = ctx->host_us[i];
  return PCC_OK;
}

int pcc_device_alloc(pcc_ctx* ctx, size_t bytes, void** dev_ptr) {
  if (!ctx || !dev_ptr) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  PCC_HIP(hipSetDevice(ctx->device));
  PCC_HIP(hipMalloc(dev_ptr, bytes ? bytes : 16));
  return PCC_OK;
}
int pcc_device_free(pcc_ctx* ctx, void* dev_ptr) {
  if (!ctx) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  PCC_HIP(hipSetDevice(ctx->device));
  PCC_HIP(hipFree(dev_ptr));
  return PCC_OK;
}
int pcc_device_upload(pcc_ctx* ctx, void* dev_dst, const void* host_src, size_t bytes) {
  if (!ctx || !dev_dst || (!host_src && bytes)) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  PCC_HIP(hipSetDevice(ctx->device));
  PCC_HIP(hipMemcpy(dev_dst, host_src, bytes, hipMemcpyHostToDevice));
  return PCC_OK;
}

// launch of one frame; `box`, `simplify_only`, `stop_after_leaf_scan` are the variations the inter-frame path needs
static int launch_frame(pcc_ctx* ctx, const void* dev_points, size_t n, size_t stride, size_t rgb_offset, const pcc_params* prm,
                        const FixedBox* box, int simplify_only, int stop_after_leaf_scan) {
  if (!ctx || !prm) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (n && !dev_points) return fail(ctx, PCC_ERR_ARG, "null point array");
  if (stride < 12 || (stride & 3) || rgb_offset + 4 > stride || (rgb_offset & 3))
    return fail(ctx, PCC_ERR_ARG, "stride/rgb_offset: need stride >= 12, 4-byte aligned fields");
  if (n >= (1ull << 30)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 2^30 points");
  if (!(prm->octree_resolution > 0.0) || !std::isfinite(prm->octree_resolution))
    return fail(ctx, PCC_ERR_ARG, "octree_resolution must be positive");
  PCC_HIP(hipSetDevice(ctx->device));
  ctx->launched = false;
  ctx->simplified_valid = false;
  ctx->n = n;
  ctx->params = *prm;
  if (n == 0) {  // empty cloud: nothing to launch; finish() reports PCC_ERR_EMPTY
    ctx->launched = true;
    return PCC_OK;
  }
  int rc = reserve(ctx, n);
  if (rc != PCC_OK) return rc;

  HotPathArgs a{};
  a.pv.base = static_cast<const uint8_t*>(dev_points);
  a.pv.stride = (uint32_t)stride;
  a.pv.rgb_off = (uint32_t)rgb_offset;
  a.pv.aligned16 = ((reinterpret_cast<uintptr_t>(dev_points) & 15) == 0 && (stride & 15) == 0 && stride >= 16) ? 1u : 0u;
  a.n = (uint32_t)n;
  a.res = prm->octree_resolution;
  {
    int ex = 0;
    const double mant = frexp(a.res, &ex);  // power of two <=> mantissa 0.5; keep 1/res finite and normal
    a.inv_res_pow2 = (mant == 0.5 && ex > -1000 && ex < 1000) ? 1.0 / a.res : 0.0;
  }
  a.lp.do_color = prm->do_color_encoding ? 1u : 0u;
  a.lp.color_reduction = (prm->color_coding_type == 0) ? (uint32_t)((8 - prm->color_bit_resolution) & 7) : 0u;
  if (prm->color_coding_type == 0 && prm->color_bit_resolution <= 0) a.lp.color_reduction = 8;
  a.lp.do_centroid = prm->do_voxel_centroid ? 1u : 0u;
  a.lp.write_image = (a.lp.do_color && prm->color_coding_type == 1) ? 1u : 0u;
  a.max_passes = std::min(std::max(ctx->pass_hint, 1), (int)kMaxPasses);
  a.deep_launch = ctx->deep_hint ? 1 : 0;
  // test hooks (pcc_set_option "force_pairs" / "no_cell_ranks"; nothing reads the environment per frame):
  // force_pairs 1: the pair sort on small frames; 2: keep the point index in the key although nothing needs it (the packed
  // [code | index] + colour payload sort); no_cell_ranks 1: the full varying Morton code is sorted
  a.force_pairs = ctx->force_pairs == 1 ? 1 : 0;
  a.need_index = (a.lp.do_centroid || stop_after_leaf_scan || ctx->force_pairs == 2) ? 1 : 0;
  a.no_cell_ranks = ctx->no_cell_ranks ? 1 : 0;
  a.boxes = ctx->d_boxes.p; a.state = ctx->d_state.p;
  a.keys_a = ctx->d_keys_a.p; a.keys_b = ctx->d_keys_b.p;
  a.idx_a = ctx->d_idx_a.p; a.idx_b = ctx->d_idx_b.p;
  if (a.deep_launch) {
    PCC_HIP(ctx->d_idx2_a.ensure(n)); PCC_HIP(ctx->d_idx2_b.ensure(n)); PCC_HIP(ctx->d_leaf_hi.ensure(n));
    a.idx2_a = ctx->d_idx2_a.p; a.idx2_b = ctx->d_idx2_b.p; a.leaf_hi = ctx->d_leaf_hi.p;
  }
  a.hist_rows = ctx->d_hist_rows.p; a.digit_tot = ctx->d_digit_tot.p; a.tile_prefix0 = ctx->d_tile_prefix0.p; a.sync_area = ctx->d_sync.p;
  a.leaf_start = ctx->d_leaf_start.p; a.leaf_code = ctx->d_leaf_code.p; a.leaf_base = ctx->d_leaf_base.p;
  a.leaf_t = ctx->d_leaf_t.p; a.occ = ctx->d_occ.p + tiles_region(n); a.bgr = ctx->d_bgr.p; a.centroid = ctx->d_centroid.p;
  a.simplified = ctx->d_simplified.p;
  a.coefs = nullptr;
  if (a.lp.write_image && ctx->jpeg_on_gpu) {
    a.coefs = ctx->d_coefs.p;
    BaselineJpeg::quantiser(prm->jpeg_quality, a.jq.half, a.jq.magic);
  }
  // the snake image itself only leaves the chip if somebody wants to look at it, or the host does the JPEG
  a.image = a.lp.write_image ? ctx->d_image.p : nullptr;  // k_leaf_tile -> k_jpeg_rows; it only leaves the chip if somebody wants to look at it, or the host does the JPEG
  a.jpeg_tiles = (a.coefs && ctx->jpeg_on_gpu >= 2) ? reinterpret_cast<uint32_t*>(ctx->d_occ.p) : nullptr;
  ctx->lines_dir_words = 0;
  if (a.lp.do_color && prm->color_coding_type == 2 && ctx->jpeg_on_gpu >= 2 && !simplify_only && !stop_after_leaf_scan) {
    const size_t max_lines = std::max<size_t>(1, n / 2048);
    ctx->lines_dir_words = 4 * max_lines;
    const size_t cap = max_lines * (size_t)kJpegLineWords + (size_t)kJpegLineWords;  // the last strip may be two strips long
    PCC_HIP(ctx->d_lines.ensure(ctx->lines_dir_words + cap));
    a.jpeg_lines_dir = ctx->d_lines.p;
    a.jpeg_lines_data = ctx->d_lines.p + ctx->lines_dir_words;
    a.jpeg_lines_capacity = (uint32_t)std::min<size_t>(cap, 0xffffffffu);
    BaselineJpeg::quantiser(prm->jpeg_quality, a.jq.half, a.jq.magic);
  }
  ctx->out_off = tiles_region(n);
  a.huff = ctx->d_huff.p;
  if (box) a.box = *box;
  a.lp.simplify_only = simplify_only ? 1u : 0u;
  a.stop_after_leaf_scan = stop_after_leaf_scan;
  ctx->args = a;
  rc = enqueue(ctx);
  if (rc != PCC_OK) return rc;
  ctx->launched = true;
  return PCC_OK;
}

int pcc_hotpath_launch(pcc_ctx* ctx, const void* dev_points, size_t n, size_t stride, size_t rgb_offset,
                       const pcc_params* prm) {
  return launch_frame(ctx, dev_points, n, stride, rgb_offset, prm, nullptr, 0, 0);
}

pcc_upload_lane* pcc_upload_lane_create(int device) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  pcc_upload_lane* l = new pcc_upload_lane();
  l->device = device;
  if (hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking) != hipSuccess) {
    delete l;
    return nullptr;
  }
  return l;
}
void pcc_upload_lane_destroy(pcc_upload_lane* l) {
  if (!l) return;
  (void)hipSetDevice(l->device);
  (void)hipStreamSynchronize(l->stream);
  (void)hipStreamDestroy(l->stream);
  delete l;
}

// developer aid (include/pcc_codec_tools.h, the pcc_debug_* block): the sort geometry of the last frame whose state came back --
// {sort passes, code bits that were sorted, varying Morton bits, key bits per axis below the cell ranks, bytes per key and pass}
int pcc_debug_sort_plan(pcc_ctx* ctx, int32_t out[5]) {
  if (!ctx || !out) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  const FrameState& st = *ctx->h_state.p;
  out[0] = st.npasses; out[1] = st.code_bits; out[2] = st.vbits; out[3] = st.code_low_bits; out[4] = st.payload ? 12 : 8;
  return PCC_OK;
}

pcc_stream* pcc_stream_create(int device) {
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  pcc_stream* st = new pcc_stream();
  st->device = device;
  if (hipStreamCreateWithFlags(&st->stream, hipStreamNonBlocking) != hipSuccess) {
    delete st;
    return nullptr;
  }
  return st;
}
void pcc_stream_destroy(pcc_stream* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  (void)hipStreamSynchronize(st->stream);
  (void)hipStreamDestroy(st->stream);
  delete st;
}
int pcc_use_stream(pcc_ctx* ctx, pcc_stream* st) {
  if (!ctx) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (ctx->launched) return fail(ctx, PCC_ERR_STATE, "pcc_use_stream: a frame is in flight on this context");
  if (st && st->device != ctx->device) return fail(ctx, PCC_ERR_ARG, "pcc_use_stream: the stream belongs to another device");
  ctx->stream = st ? st->stream : ctx->own_stream;
  return PCC_OK;
}

void* pcc_host_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 16, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return p;
}
void pcc_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int pcc_hotpath_launch_host(pcc_ctx* ctx, pcc_upload_lane* lane, const void* host_points, size_t n, size_t stride, size_t rgb_offset,
                            const pcc_params* prm) {
  if (!ctx || !prm) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (n && !host_points) return fail(ctx, PCC_ERR_ARG, "null point array");
  if (lane && lane->device != ctx->device) return fail(ctx, PCC_ERR_ARG, "upload lane and context belong to different GPUs");
  if (n == 0) return launch_frame(ctx, nullptr, 0, stride, rgb_offset, prm, nullptr, 0, 0);
  PCC_HIP(hipSetDevice(ctx->device));
  if (ctx->pack_upload && stride >= 16 && rgb_offset + 4 <= stride) {
    // 16 of a point's bytes are read by the kernels: they are packed here (the calling thread: a GPU-stage thread of the
    // pipeline, which would otherwise wait) and half the bytes cross the link
    const size_t packed = 16 * n;
    PCC_HIP(ctx->d_points.ensure(packed + 16));
    PCC_HIP(ctx->h_pack.ensure(packed + 32));
    if (ctx->locked_host) { unlock_host_range(ctx->locked_host); ctx->locked_host = nullptr; }
    pack_points_16(ctx->h_pack.p, static_cast<const uint8_t*>(host_points), n, stride, rgb_offset);
    hipError_t e;
    if (lane) {
      std::lock_guard<std::mutex> lk(lane->mu);
      e = hipMemcpyAsync(ctx->d_points.p, ctx->h_pack.p, packed, hipMemcpyHostToDevice, lane->stream);
      if (e == hipSuccess) e = hipEventRecord(ctx->ev_h2d, lane->stream);
      if (e == hipSuccess) e = hipStreamWaitEvent(ctx->stream, ctx->ev_h2d, 0);
    } else {
      e = hipMemcpyAsync(ctx->d_points.p, ctx->h_pack.p, packed, hipMemcpyHostToDevice, ctx->stream);
    }
    if (e != hipSuccess) return hip_fail(ctx, e, "upload of the packed frame");
    return launch_frame(ctx, ctx->d_points.p, n, 16, 12, prm, nullptr, 0, 0);
  }
  const size_t bytes = n * stride;
  PCC_HIP(ctx->d_points.ensure(bytes + 16));
  if (ctx->locked_host) { unlock_host_range(ctx->locked_host); ctx->locked_host = nullptr; }  // a frame that was never finished
  if (lock_host_range(host_points, bytes) == 0) ctx->locked_host = host_points;
  hipError_t e;
  if (lane) {
    std::lock_guard<std::mutex> lk(lane->mu);
    e = hipMemcpyAsync(ctx->d_points.p, host_points, bytes, hipMemcpyHostToDevice, lane->stream);
    if (e == hipSuccess) e = hipEventRecord(ctx->ev_h2d, lane->stream);
  } else {
    e = hipMemcpyAsync(ctx->d_points.p, host_points, bytes, hipMemcpyHostToDevice, ctx->stream);
  }
  if (e == hipSuccess && lane) e = hipStreamWaitEvent(ctx->stream, ctx->ev_h2d, 0);
  int rc = e == hipSuccess ? launch_frame(ctx, ctx->d_points.p, n, stride, rgb_offset, prm, nullptr, 0, 0) : hip_fail(ctx, e, "upload of the frame");
  if (rc != PCC_OK && ctx->locked_host) {
    (void)hipStreamSynchronize(lane ? lane->stream : ctx->stream);
    unlock_host_range(ctx->locked_host);
    ctx->locked_host = nullptr;
  }
  return rc;
}

// wait for the frame's kernels and its FrameState; a frame that needs more sort passes than were enqueued runs again
static int wait_frame_state(pcc_ctx* ctx) {
  { const int wrc = wait_stream(ctx, 0); if (wrc != PCC_OK) return wrc; }
  const FrameState& st = *ctx->h_state.p;
  // Two device errors mean "run the frame again": more sort passes needed than were enqueued (a deeper tree than the
  // frames before: every pass is enqueued the second time), and a bounded poll that ran out (a workgroup this frame
  // waited for was held up for tens of milliseconds -- another process on the GPU, a debugger; nothing is wrong with
  // the frame).  Either can show up on the re-run the other one caused, hence a loop; a poll may time out once.
  int spin_retries = 0;
  for (int attempt = 0; attempt < 6; ++attempt) {
    if (st.error == kErrPasses && ctx->args.max_passes < (int)kMaxPasses) {
      ctx->args.max_passes = kMaxPasses;
    } else if (st.error == kErrDeep && !ctx->args.deep_launch) {
      // a tree deeper than 21 levels: its Morton codes need two words; the kernels' DEEP instantiations and their arrays
      ctx->args.deep_launch = 1;
      ctx->args.max_passes = kMaxPasses;
      PCC_HIP(ctx->d_idx2_a.ensure(ctx->n)); PCC_HIP(ctx->d_idx2_b.ensure(ctx->n)); PCC_HIP(ctx->d_leaf_hi.ensure(ctx->n));
      ctx->args.idx2_a = ctx->d_idx2_a.p; ctx->args.idx2_b = ctx->d_idx2_b.p; ctx->args.leaf_hi = ctx->d_leaf_hi.p;
    } else if (st.error == kErrSpin && spin_retries == 0) {
      ++spin_retries;
    } else {
      break;
    }
    const int rc = enqueue(ctx);
    if (rc != PCC_OK) return rc;
    { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  }
  if (st.error == kErrSpin) return fail(ctx, PCC_ERR_HIP, "the GPU did not make progress on this frame (look-back poll timed out twice)");
  if (st.error != kErrNone) {
    char buf[160];
    snprintf(buf, sizeof(buf), "unsupported frame geometry (device error %d: depth %d > %d, or key window %d bits missed)",
             st.error, st.depth, kMaxDepthDeep, st.vbits);
    return fail(ctx, PCC_ERR_UNSUPPORTED, buf);
  }
  if (st.n_epochs == 0) return fail(ctx, PCC_ERR_EMPTY, "no finite point: frame dropped");
  ctx->pass_hint = st.npasses;
  ctx->deep_hint = st.depth > kMaxDepth;  // (a shallow frame behind a deep one goes back to the single-word kernels)
  return PCC_OK;
}

int pcc_reserve(pcc_ctx* ctx, size_t max_points, size_t bitstream_bytes) {
  if (!ctx) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (max_points >= (1ull << 30)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 2^30 points");
  PCC_HIP(hipSetDevice(ctx->device));
  if (max_points) {
    const int rc = reserve(ctx, max_points);
    if (rc != PCC_OK) return rc;
    // landing buffers of the usual products (occupancy bytes: about one per point for surfaces; more is fetched on demand)
    PCC_HIP(ctx->h_occ.ensure(tiles_region(max_points) + std::max(2 * max_points, 2 * bitstream_bytes) + 16));
    {
      // A device-to-host copy into the landing buffer, here and now, behind a device-wide synchronisation.
      // hipMemcpyAsync into a page-locked buffer sometimes costs the calling thread 2-15 ms of CPU (PCC_FINISH_TRACE shows
      // it as "copies enqueued"): always the first copy into a fresh buffer -- but whole short sequences also ran at half
      // speed with buffers that had been copied into before, unless the runtime had been through a hipDeviceSynchronize
      // (or a hipHostFree, which implies one) since the sequences before (tools/bench20.sh, 20-frame calls after a
      // 92-frame warm-up: 15 of 17 calls at 900-1 600 Mpoints/s without it, 0 of 24 with it; event queries, stream
      // queries and stream synchronisations do not have that effect).  A reservation is where a caller prepares a
      // sequence, the GPU is idle, and the synchronisation costs nothing.
      PCC_HIP(hipDeviceSynchronize());
      const size_t bytes = std::min(ctx->h_occ.cap, ctx->d_occ.cap);
      PCC_HIP(hipMemcpyAsync(ctx->h_occ.p, ctx->d_occ.p, bytes, hipMemcpyDeviceToHost, ctx->stream));
      PCC_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  if (bitstream_bytes) {
    const size_t want = 2 * bitstream_bytes + 4096;
    if (ctx->bitstream.capacity() < want) {
      ctx->bitstream.reserve(want);
      const size_t had = ctx->bitstream.size();
      ctx->bitstream.resize(want);
      memset(ctx->bitstream.data() + had, 0, want - had);  // first touch now
      ctx->bitstream.resize(had);
    }
  }
  return PCC_OK;
}

int pcc_hotpath_finish(pcc_ctx* ctx, pcc_hot_result* out) {
  if (!ctx || !out) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (!ctx->launched) return fail(ctx, PCC_ERR_STATE, "pcc_hotpath_finish without pcc_hotpath_launch");
  ctx->launched = false;
  memset(out, 0, sizeof(*out));
  if (ctx->n == 0) return fail(ctx, PCC_ERR_EMPTY, "empty cloud: frame dropped");
  PCC_HIP(hipSetDevice(ctx->device));
  // developer aid (PCC_FINISH_TRACE=1): calls that take longer than 2 ms say where the time went (wall / CPU of this thread)
  static const bool trace = [] { const char* e = dev_env("PCC_FINISH_TRACE"); return e && e[0] == '1'; }();
  auto now_pair = [](double t[2]) {
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    clock_gettime(CLOCK_THREAD_CPUTIME_ID, &b);
    t[0] = a.tv_sec * 1e3 + a.tv_nsec * 1e-6;
    t[1] = b.tv_sec * 1e3 + b.tv_nsec * 1e-6;
  };
  double tr0[2] = {0, 0}, tr1[2] = {0, 0}, tr2[2] = {0, 0}, tr3[2] = {0, 0};
  if (trace) now_pair(tr0);
  const int src = wait_frame_state(ctx);
  if (trace) now_pair(tr1);
  if (ctx->locked_host) {  // the upload was over before the first kernel started
    unlock_host_range(ctx->locked_host);
    ctx->locked_host = nullptr;
  }
  const FrameState& st = *ctx->h_state.p;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end);
  out->gpu_ms = ms;
  if (ctx->profiling) {
    if (ctx->profile_events) ctx->timer.collect(ctx->times); else ctx->times.clear();
    ctx->span_times.clear();
    ctx->span_starts.clear();
    unsigned long long origin = ~0ull;
    int sort_seen = 0;
    for (size_t i = 0; i < ctx->span_names.size() && i < (size_t)kMaxSpans; ++i) {
      const unsigned long long* w = ctx->h_spans.p + i * 2 * kSpanShards;
      unsigned long long t0 = ~0ull, e1 = ~0ull;
      for (int k = 0; k < kSpanShards; ++k) { t0 = std::min(t0, w[k]); e1 = std::min(e1, w[kSpanShards + k]); }
      const unsigned long long t1 = ~e1;
      const bool is_sort = !strcmp(ctx->span_names[i], "k_sort_pass");
      if (is_sort && ++sort_seen > st.npasses) continue;  // enqueued, but the frame did not need the pass
      if (t0 == ~0ull || t1 < t0) continue;
      if (origin == ~0ull) origin = t0;
      ctx->span_times.emplace_back(ctx->span_names[i], (float)((double)(t1 - t0) / ctx->wall_clock_khz));
      ctx->span_starts.push_back((float)((double)(t0 - origin) / ctx->wall_clock_khz));
    }
  }
  if (src != PCC_OK) return src;
  const size_t L = st.n_leaves, B = st.n_branches;
  const pcc_params& prm = ctx->params;
  const bool color = prm.do_color_encoding != 0;
  const bool image = color && prm.color_coding_type == 1;
  const uint32_t W = 256, H = (uint32_t)(L / 256 + 1);

  const size_t off = ctx->out_off;
  const bool tiles_too = image && ctx->jpeg_on_gpu >= 2;  // records + occupancy bytes in one copy
  // (two bytes per point even if this frame needs fewer: the buffer then survives pcc_reserve and the other frames of a
  // sequence -- freeing and re-allocating page-locked memory in the middle of a sequence is what makes later copies slow)
  PCC_HIP(ctx->h_occ.ensure(off + std::max<size_t>(B, 2 * ctx->n) + 16));
  if (tiles_too) PCC_HIP(hipMemcpyAsync(ctx->h_occ.p, ctx->d_occ.p, off + B, hipMemcpyDeviceToHost, ctx->stream));
  else PCC_HIP(hipMemcpyAsync(ctx->h_occ.p + off, ctx->d_occ.p + off, B, hipMemcpyDeviceToHost, ctx->stream));
  const uint32_t* h_tiles = reinterpret_cast<const uint32_t*>(ctx->h_occ.p);
  const bool lines_on_gpu = ctx->lines_dir_words != 0 && color && prm.color_coding_type == 2;
  // the per-voxel colours themselves only leave the chip if the host codes them (types 0, 3; type 2 without the GPU
  // stage) or somebody wants to look at them
  const bool want_bgr = color && (((prm.color_coding_type == 1 || lines_on_gpu) && ctx->copy_image) ||
                                  (prm.color_coding_type != 1 && !lines_on_gpu));
  if (want_bgr) {
    PCC_HIP(ctx->h_bgr.ensure(3 * L + 16));
    PCC_HIP(hipMemcpyAsync(ctx->h_bgr.p, ctx->d_bgr.p, 3 * L, hipMemcpyDeviceToHost, ctx->stream));
  }
  const bool coefs = image && ctx->jpeg_on_gpu >= 1;
  const bool tiles = image && ctx->jpeg_on_gpu >= 2;
  const bool want_image = image && (ctx->copy_image || !coefs);
  const size_t n_coefs = (size_t)16 * ((H + 15) / 16) * 6 * 64;
  const size_t n_tiles = (H + 15) / 16;
  if (want_image) {
    PCC_HIP(ctx->h_image.ensure((size_t)3 * W * H + 16));
    PCC_HIP(hipMemcpyAsync(ctx->h_image.p, ctx->d_image.p, (size_t)3 * W * H, hipMemcpyDeviceToHost, ctx->stream));
  }
  if (tiles) {  // the Huffman-coded MCU rows came with the occupancy bytes; the coefficients only follow if a row did not fit its record
  } else if (coefs) {
    PCC_HIP(ctx->h_coefs.ensure(n_coefs + 64));
    PCC_HIP(hipMemcpyAsync(ctx->h_coefs.p, ctx->d_coefs.p, n_coefs * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (prm.do_voxel_centroid) {
    PCC_HIP(ctx->h_centroid.ensure(3 * L + 16));
    PCC_HIP(hipMemcpyAsync(ctx->h_centroid.p, ctx->d_centroid.p, 3 * L, hipMemcpyDeviceToHost, ctx->stream));
  }
  const size_t n_lines = L / 2048 ? L / 2048 : 1;
  if (lines_on_gpu) {  // directory and bit strings are one piece of HBM: one copy
    const size_t words = ctx->lines_dir_words + st.jpeg_line_words;
    PCC_HIP(ctx->h_lines.ensure(words + 16));
    PCC_HIP(hipMemcpyAsync(ctx->h_lines.p, ctx->d_lines.p, words * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  }
  if (trace) now_pair(tr2);
  { const int wrc = wait_stream(ctx, 1); if (wrc != PCC_OK) return wrc; }
  if (trace) {
    now_pair(tr3);
    if (tr3[0] - tr0[0] > 2.0)
      fprintf(stderr, "[pcc_hotpath_finish %p] wait for the kernels %.3f ms (cpu %.3f), copies enqueued %.3f (cpu %.3f), wait for the copies %.3f (cpu %.3f); gpu %.3f ms\n",
              (void*)ctx, tr1[0] - tr0[0], tr1[1] - tr0[1], tr2[0] - tr1[0], tr2[1] - tr1[1], tr3[0] - tr2[0], tr3[1] - tr2[1], (double)ms);
  }
  bool lines_ok = lines_on_gpu;
  if (lines_on_gpu) {
    for (size_t i = 0; i < n_lines; ++i)
      if (ctx->h_lines.p[4 * i + 3] != 0) lines_ok = false;
    if (!lines_ok) {  // cannot happen (the region holds the worst case); the host codes the strips from the colours then
      PCC_HIP(ctx->h_bgr.ensure(3 * L + 16));
      PCC_HIP(hipMemcpyAsync(ctx->h_bgr.p, ctx->d_bgr.p, 3 * L, hipMemcpyDeviceToHost, ctx->stream));
      { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
    }
  }

  for (int a = 0; a < 3; ++a) { out->bbox[a] = st.mn[a]; out->bbox[3 + a] = st.mx[a]; }
  out->depth = (uint32_t)st.depth;
  out->n_epochs = (uint32_t)st.n_epochs;
  out->n_points_in = st.n_finite;
  out->n_leaves = L;
  out->n_branches = B;
  out->occupancy = ctx->h_occ.p + off;
  out->bgr = (want_bgr || (lines_on_gpu && !lines_ok)) ? ctx->h_bgr.p : nullptr;
  out->jpeg_lines_dir = lines_ok ? ctx->h_lines.p : nullptr;
  out->jpeg_lines_data = lines_ok ? ctx->h_lines.p + ctx->lines_dir_words : nullptr;
  out->jpeg_n_lines = lines_ok ? (uint32_t)n_lines : 0u;
  out->centroid = prm.do_voxel_centroid ? ctx->h_centroid.p : nullptr;
  out->image = want_image ? ctx->h_image.p : nullptr;
  bool tiles_ok = tiles;
  if (tiles) {
    for (size_t m = 0; m < n_tiles; ++m)
      if (h_tiles[m * kJpegTileWords + 3] != 0) tiles_ok = false;
    if (!tiles_ok) {  // rare: a very busy image; fall back to the coefficients for this frame
      PCC_HIP(ctx->h_coefs.ensure(n_coefs + 64));
      PCC_HIP(hipMemcpyAsync(ctx->h_coefs.p, ctx->d_coefs.p, n_coefs * sizeof(int16_t), hipMemcpyDeviceToHost, ctx->stream));
      { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
    }
  }
  out->jpeg_coefs = (coefs && !tiles_ok) ? ctx->h_coefs.p : nullptr;
  out->jpeg_tiles = tiles_ok ? h_tiles : nullptr;
  out->jpeg_tile_words = kJpegTileWords;
  out->jpeg_n_tiles = (uint32_t)n_tiles;
  memcpy(ctx->occ_hist, ctx->h_state.p->occ_hist, sizeof(ctx->occ_hist));  // counted by k_occ_histogram, came back with the FrameState
  out->occupancy_histogram = ctx->occ_hist;
  out->image_w = image ? W : 0;
  out->image_h = image ? H : 0;
  ctx->last_L = L;
  ctx->simplified_valid = true;
  ctx->last_hot = *out;
  return PCC_OK;
}

int pcc_entropy_encode(pcc_ctx* ctx, const pcc_hot_result* hot, const pcc_params* prm, pcc_bitstream* out) {
  if (!ctx || !hot || !prm || !out) return PCC_ERR_ARG;
  entropy_encode_frame(*hot, *prm, ctx->bitstream, out->perf, ctx->host_us);
  out->data = ctx->bitstream.data();
  out->len = ctx->bitstream.size();
  return PCC_OK;
}

int pcc_entropy_encode_many(int n, pcc_ctx* const ctx[], const pcc_hot_result* const hot[], const pcc_params* const prm[],
                            pcc_bitstream* const out[]) {
  if (n < 1 || n > PCC_MAX_FRAMES_AT_ONCE || !ctx || !hot || !prm || !out) return PCC_ERR_ARG;
  Bytes* o[PCC_MAX_FRAMES_AT_ONCE] = {};
  uint64_t* pf[PCC_MAX_FRAMES_AT_ONCE] = {};
  double* t[PCC_MAX_FRAMES_AT_ONCE] = {};
  for (int i = 0; i < n; ++i) {
    if (!ctx[i] || !hot[i] || !prm[i] || !out[i]) return PCC_ERR_ARG;
    for (int k = 0; k < i; ++k)
      if (ctx[k] == ctx[i]) return PCC_ERR_ARG;  // every frame's bitstream lives in its own context
    o[i] = &ctx[i]->bitstream; pf[i] = out[i]->perf; t[i] = ctx[i]->host_us;
  }
  entropy_encode_frames(n, hot, prm, o, pf, t);
  for (int i = 0; i < n; ++i) { out[i]->data = ctx[i]->bitstream.data(); out[i]->len = ctx[i]->bitstream.size(); }
  return PCC_OK;
}

int pcc_encode_intra_device(pcc_ctx* ctx, const void* dev_points, size_t n, size_t stride, size_t rgb_offset,
                            const pcc_params* prm, pcc_bitstream* out) {
  if (!ctx || !out) return PCC_ERR_ARG;
  memset(out, 0, sizeof(*out));
  int rc = pcc_hotpath_launch(ctx, dev_points, n, stride, rgb_offset, prm);
  if (rc != PCC_OK) return rc;
  pcc_hot_result hot;
  rc = pcc_hotpath_finish(ctx, &hot);
  if (rc != PCC_OK) return rc;
  return pcc_entropy_encode(ctx, &hot, prm, out);
}

int pcc_encode_intra(pcc_ctx* ctx, const void* host_points, size_t n, size_t stride, size_t rgb_offset,
                     const pcc_params* prm, pcc_bitstream* out) {
  if (!ctx || !out || !prm) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  memset(out, 0, sizeof(*out));
  if (n && !host_points) return fail(ctx, PCC_ERR_ARG, "null point array");
  if (n == 0) return fail(ctx, PCC_ERR_EMPTY, "empty cloud: frame dropped");
  PCC_HIP(hipSetDevice(ctx->device));
  PCC_HIP(ctx->d_points.ensure(n * stride + 16));
  PCC_HIP(hipMemcpyAsync(ctx->d_points.p, host_points, n * stride, hipMemcpyHostToDevice, ctx->stream));
  return pcc_encode_intra_device(ctx, ctx->d_points.p, n, stride, rgb_offset, prm, out);
}

int pcc_get_output_cloud(pcc_ctx* ctx, const pcc_point_xyzrgb** points, size_t* n) {
  if (!ctx || !points || !n) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  if (!ctx->simplified_valid) return fail(ctx, PCC_ERR_STATE, "no encoded frame to take the simplified cloud from");
  PCC_HIP(hipSetDevice(ctx->device));
  const size_t L = ctx->last_L;
  PCC_HIP(ctx->h_simplified.ensure(4 * L + 4));
  PCC_HIP(hipMemcpyAsync(ctx->h_simplified.p, ctx->d_simplified.p, 16 * L, hipMemcpyDeviceToHost, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  ctx->out_cloud.resize(L);
  const bool color = ctx->params.do_color_encoding != 0;
  for (size_t i = 0; i < L; ++i) {  // a default-constructed PointXYZRGB with x,y,z,r,g,b overwritten (impl.hpp:1515,1554-1576)
    pcc_point_xyzrgb& p = ctx->out_cloud[i];
    const float* s = ctx->h_simplified.p + 4 * i;
    p.x = s[0]; p.y = s[1]; p.z = s[2]; p.w = 1.0f;
    uint32_t rgba;
    memcpy(&rgba, s + 3, 4);
    p.rgba = color ? rgba : 0xFF000000u;
    p.pad[0] = p.pad[1] = p.pad[2] = 0;
  }
  *points = ctx->out_cloud.data();
  *n = L;
  return PCC_OK;
}

int pcc_quality_metrics(pcc_ctx* ctx, const pcc_point_xyzrgb* cloud_a, size_t n_a, const pcc_point_xyzrgb* cloud_b, size_t n_b,
                        double cell_hint, pcc_quality* out) {
  if (!ctx || !out || (!cloud_a && n_a) || (!cloud_b && n_b)) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  memset(out, 0, sizeof(*out));
  if (n_a == 0 || n_b == 0) return fail(ctx, PCC_ERR_EMPTY, "quality metric of an empty cloud");
  if (n_a >= (1ull << 31) || n_b >= (1ull << 31)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 2^31 points");
  PCC_HIP(hipSetDevice(ctx->device));
  const size_t n_max = std::max(n_a, n_b);
  const size_t slots = quality_table_slots(n_max);
  const size_t blocks = (n_max + 255) / 256;
  PCC_HIP(ctx->d_qa.ensure(32 * n_a));
  PCC_HIP(ctx->d_qb.ensure(32 * n_b));
  PCC_HIP(ctx->d_qkeys.ensure(slots));
  PCC_HIP(ctx->d_qheads.ensure(slots));
  PCC_HIP(ctx->d_qnext.ensure(n_max));
  PCC_HIP(ctx->d_qidx.ensure(n_max));
  PCC_HIP(ctx->d_qd2.ensure(n_max));
  PCC_HIP(ctx->d_qpart.ensure(blocks * 8));
  ctx->h_qpart.resize(blocks * 8);
  PCC_HIP(hipMemcpyAsync(ctx->d_qa.p, cloud_a, 32 * n_a, hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->d_qb.p, cloud_b, 32 * n_b, hipMemcpyHostToDevice, ctx->stream));

  // grid: origin below both clouds, cell = hint or a size that puts about two points of a volume-filling cloud in a cell
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  auto span = [&](const pcc_point_xyzrgb* c, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      const float q[3] = {c[i].x, c[i].y, c[i].z};
      if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
      for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
    }
  };
  span(cloud_a, n_a);
  span(cloud_b, n_b);
  if (!(lo[0] <= hi[0])) return fail(ctx, PCC_ERR_EMPTY, "quality metric: no finite point");
  float ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  if (!(ext > 0.f)) ext = 1.f;
  float cell = (float)cell_hint;
  if (!(cell > 0.f)) cell = ext / std::cbrt((float)n_max / 2.0f);
  cell = std::max(cell, ext / 1000000.0f);  // 21 bits of cell index per axis, with room to spare

  QualityArgs qa{};
  qa.keys = ctx->d_qkeys.p; qa.heads = ctx->d_qheads.p; qa.next = ctx->d_qnext.p;
  qa.d2 = ctx->d_qd2.p; qa.idx = ctx->d_qidx.p; qa.partials = ctx->d_qpart.p;
  qa.cell = cell;
  for (int a = 0; a < 3; ++a) qa.origin[a] = lo[a] - cell;
  double sum_d[2] = {0, 0}, sum_e[3] = {0, 0, 0};
  float max_d[2] = {0, 0}, max_xyz[3] = {0, 0, 0};
  PCC_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
  for (int dir = 0; dir < 2; ++dir) {  // 0: A -> B with colour, 1: B -> A geometry only
    qa.query = dir == 0 ? ctx->d_qa.p : ctx->d_qb.p;
    qa.target = dir == 0 ? ctx->d_qb.p : ctx->d_qa.p;
    qa.n_query = (uint32_t)(dir == 0 ? n_a : n_b);
    qa.n_target = (uint32_t)(dir == 0 ? n_b : n_a);
    qa.table_slots = quality_table_slots(qa.n_target);
    qa.with_colour = dir == 0;
    launch_quality_direction(qa, ctx->stream);
    PCC_HIP(hipGetLastError());
    const size_t nb = ((size_t)qa.n_query + 255) / 256;
    PCC_HIP(hipMemcpyAsync(ctx->h_qpart.data(), ctx->d_qpart.p, nb * 8 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (dir == 1) PCC_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
    float md = -3.4e38f;
    for (size_t b = 0; b < nb; ++b) {  // fixed order: the result does not depend on scheduling
      const double* p = ctx->h_qpart.data() + 8 * b;
      sum_d[dir] += p[0];
      if (dir == 0) { sum_e[0] += p[1]; sum_e[1] += p[2]; sum_e[2] += p[3]; }
      md = std::max(md, (float)p[4]);
      if (dir == 0) for (int a = 0; a < 3; ++a) max_xyz[a] = b == 0 ? (float)p[5 + a] : std::max(max_xyz[a], (float)p[5 + a]);
    }
    max_d[dir] = md;
  }
  (void)hipEventElapsedTime(&out->gpu_ms, ctx->ev_begin, ctx->ev_end);
  // quality_metrics_impl.hpp:163-198
  const float max_dist_a = std::sqrt(max_d[0]), max_dist_b = std::sqrt(max_d[1]);
  const double rms_a = std::sqrt(sum_d[0] / (double)n_a), rms_b = std::sqrt(sum_d[1] / (double)n_b);
  const float dist_h = std::max(max_dist_a, max_dist_b);
  const float dist_rms = (float)std::max(rms_a, rms_b);
  const float energy = max_xyz[0] * max_xyz[0] + max_xyz[1] * max_xyz[1] + max_xyz[2] * max_xyz[2];
  const float psnr = 10 * std::log10(energy / (dist_rms * dist_rms));
  out->in_point_count = n_a;
  out->out_point_count = n_b;
  out->left_hausdorff = max_dist_a; out->right_hausdorff = max_dist_b; out->symm_hausdorff = dist_h;
  out->left_rms = (float)rms_a; out->right_rms = (float)rms_b; out->symm_rms = dist_rms;
  out->psnr_db = psnr;
  for (int c = 0; c < 3; ++c) out->psnr_yuv[c] = 10 * std::log10(1.0 / (sum_e[c] / (double)n_a));
  return PCC_OK;
}

int pcc_remove_outliers(pcc_ctx* ctx, const pcc_point_xyzrgb* cloud, size_t n, int min_points, double radius, uint8_t* keep, size_t* n_kept) {
  if (!ctx || (!cloud && n) || !keep || !n_kept) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  *n_kept = 0;
  if (n == 0) return PCC_OK;
  if (min_points <= 0) {  // the reference does nothing then (impl.hpp:1844)
    memset(keep, 1, n);
    *n_kept = n;
    return PCC_OK;
  }
  if (!(radius > 0.0)) return fail(ctx, PCC_ERR_ARG, "remove_outliers: radius must be positive");
  if (n >= (1ull << 31)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 2^31 points");
  PCC_HIP(hipSetDevice(ctx->device));
  float lo[3] = {3.4e38f, 3.4e38f, 3.4e38f}, hi[3] = {-3.4e38f, -3.4e38f, -3.4e38f};
  for (size_t i = 0; i < n; ++i) {
    const float q[3] = {cloud[i].x, cloud[i].y, cloud[i].z};
    if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
    for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], q[a]); hi[a] = std::max(hi[a], q[a]); }
  }
  if (!(lo[0] <= hi[0])) { memset(keep, 0, n); return PCC_OK; }
  const float ext = std::max(std::max(hi[0] - lo[0], hi[1] - lo[1]), hi[2] - lo[2]);
  const float cell = std::max((float)radius, ext / 1000000.0f);  // 21 bits of cell index per axis
  const size_t slots = quality_table_slots(n);
  PCC_HIP(ctx->d_qa.ensure(32 * n));
  PCC_HIP(ctx->d_qkeys.ensure(slots));
  PCC_HIP(ctx->d_qheads.ensure(slots));
  PCC_HIP(ctx->d_qnext.ensure(n));
  PCC_HIP(ctx->d_qb.ensure(n));
  PCC_HIP(hipMemcpyAsync(ctx->d_qa.p, cloud, 32 * n, hipMemcpyHostToDevice, ctx->stream));
  RadiusArgs ra{};
  ra.cloud = ctx->d_qa.p; ra.n = (uint32_t)n;
  for (int a = 0; a < 3; ++a) ra.origin[a] = lo[a] - cell;
  ra.radius = (float)radius;
  if (cell > ra.radius) return fail(ctx, PCC_ERR_UNSUPPORTED, "remove_outliers: radius too small for the cloud's extent");
  ra.min_points = (uint32_t)min_points;
  ra.table_slots = slots;
  ra.keys = ctx->d_qkeys.p; ra.heads = ctx->d_qheads.p; ra.next = ctx->d_qnext.p; ra.keep = ctx->d_qb.p;
  launch_radius_filter(ra, ctx->stream);
  PCC_HIP(hipGetLastError());
  PCC_HIP(hipMemcpyAsync(keep, ctx->d_qb.p, n, hipMemcpyDeviceToHost, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  size_t k = 0;
  for (size_t i = 0; i < n; ++i) k += keep[i] ? 1 : 0;
  *n_kept = k;
  return PCC_OK;
}

int pcc_decode_intra(pcc_ctx* ctx, const uint8_t* stream, size_t len, pcc_cloud* out) {
  if (!ctx || !out || (!stream && len)) return PCC_ERR_ARG;
  int rc;
  try {  // a corrupt header can ask for more memory than there is: that must not leave the C ABI as an exception
    rc = decode_frame(stream, len, ctx->dec_points, *out);
  } catch (const std::exception&) {  // bad_alloc, length_error, system_error: all of them mean "not a stream we can decode"
    ctx->dec_points.clear();
    rc = PCC_ERR_STREAM;
  }
  out->points = ctx->dec_points.data();
  out->n = ctx->dec_points.size();
  if (rc != PCC_OK) return fail(ctx, rc, "decode: frame header not found, or stream truncated/corrupt");
  return PCC_OK;
}

// decodePointCloud (impl.hpp:224-310) with everything behind the sequential stages on the GPU.  Falls back to the
// host decoder for what the kernels do not cover (trees deeper than 31 levels, inconsistent vector lengths of a
// corrupt stream): same results either way.
int pcc_decode_intra_gpu(pcc_ctx* ctx, const uint8_t* stream, size_t len, pcc_cloud* out) {
  if (!ctx || !out || (!stream && len)) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  auto ms_since = [&](const timespec& a) {
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    return (double)(t.tv_sec - a.tv_sec) * 1e3 + (double)(t.tv_nsec - a.tv_nsec) * 1e-6;
  };
  FrameStreams& fs = ctx->dec_streams;
  LeafParents& lp = ctx->dec_parents;
  int rc;
  bool host_only = false;
  try {
    rc = decode_frame_streams(stream, len, *out, fs, false);
    if (rc != PCC_OK) return fail(ctx, rc, "decode: frame header not found, or stream truncated/corrupt");
    const size_t L = (size_t)fs.count;
    int img_w = 0, img_h = 0;
    const bool jpeg = fs.with_color && fs.cct == 1;
    if (out->depth > (uint32_t)kMaxDepthDeep || L >= (1ull << 31) || L == 0) host_only = true;
    if (!host_only && out->params.do_voxel_centroid && fs.cen.size() < 3 * L) host_only = true;
    if (!host_only && fs.with_color && !jpeg && fs.col.size() < 3 * L) host_only = true;
    // the JPEG's Huffman decoding and the walk over the occupancy stream are both sequential, but not on each other:
    // they run side by side (a second thread for the time of the call)
    bool jpeg_ok = true, jpeg_oom = false;
    auto decode_jpeg = [&] {
      try {
        jpeg_ok = BaselineJpeg::decode_coefs(fs.payload.data(), fs.payload.size(), img_w, img_h, ctx->dec_coefs, (uint64_t)fs.count + 4096u);
      } catch (const std::exception&) {  // (nothing may leave a thread's function)
        jpeg_ok = false;
        jpeg_oom = true;
      }
    };
    {
      // an exception on the way (the walk's vectors are sized from the untrusted stream) must not unwind past a
      // joinable thread: that is std::terminate
      struct JoinOnExit {
        std::thread t;
        ~JoinOnExit() { if (t.joinable()) t.join(); }
      } helper;
      bool inline_jpeg = false;
      if (!host_only && jpeg) {
        try {
          helper.t = std::thread(decode_jpeg);
        } catch (const std::system_error&) {  // no thread to be had: the JPEG is decoded here, after the walk
          inline_jpeg = true;
        }
      }
      if (!host_only) rc = walk_leaf_parents(fs.occ, out->depth, fs.count, lp);
      if (inline_jpeg) decode_jpeg();
    }
    if (jpeg_oom) throw std::bad_alloc();
    if (!host_only && rc != PCC_OK) return fail(ctx, rc, "decode: occupancy stream does not describe the announced number of voxels");
    if (!host_only && jpeg && (!jpeg_ok || img_w % 8 != 0 || (size_t)img_w * (size_t)img_h < L)) host_only = true;
    if (host_only) {
      ctx->dec_ms[1] = 0.0;  // (pcc_get_decode_times: no GPU half this time)
      rc = decode_frame(stream, len, ctx->dec_points, *out);
      out->points = ctx->dec_points.data();
      out->n = ctx->dec_points.size();
      if (rc != PCC_OK) return fail(ctx, rc, "decode: frame header not found, or stream truncated/corrupt");
      return PCC_OK;
    }
    ctx->dec_ms[0] = ms_since(t0);
    timespec t1;
    clock_gettime(CLOCK_MONOTONIC, &t1);
    PCC_HIP(hipSetDevice(ctx->device));
    // one staging buffer, one upload
    const size_t np = lp.prefix.size();
    auto up16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t n_hi = lp.prefix_hi.size() == np ? 4 * np : 0;  // two-word keys: trees of more than 22 levels
    const size_t o_prefix = 0, o_first = up16(o_prefix + 8 * np), o_bits = up16(o_first + 4 * np), o_hi = up16(o_bits + np);
    const size_t o_cen = up16(o_hi + n_hi), n_cen = out->params.do_voxel_centroid ? 3 * L : 0;
    const size_t o_col = up16(o_cen + n_cen), n_col = (fs.with_color && !jpeg) ? 3 * L : 0;
    const size_t o_coef = up16(o_col + n_col), n_coef = jpeg ? ctx->dec_coefs.blocks.size() * sizeof(int16_t) : 0;
    const size_t staged = up16(o_coef + n_coef);
    const size_t mx = jpeg ? (size_t)ctx->dec_coefs.mcus_x : 0, my = jpeg ? (size_t)ctx->dec_coefs.mcus_y : 0;
    const size_t o_py = staged, o_pcb = up16(o_py + 256 * mx * my), o_pcr = up16(o_pcb + 64 * mx * my), total = up16(o_pcr + 64 * mx * my);
    PCC_HIP(ctx->h_dec_stage.ensure(staged + 16));
    PCC_HIP(ctx->d_dec.ensure(total + 16));
    PCC_HIP(ctx->d_dec_points.ensure(32 * L + 32));
    PCC_HIP(ctx->h_dec_points.ensure(L + 1));
    uint8_t* h = ctx->h_dec_stage.p;
    memcpy(h + o_prefix, lp.prefix.data(), 8 * np);
    memcpy(h + o_first, lp.first.data(), 4 * np);
    memcpy(h + o_bits, lp.bits.data(), np);
    if (n_hi) memcpy(h + o_hi, lp.prefix_hi.data(), n_hi);
    if (n_cen) memcpy(h + o_cen, fs.cen.data(), n_cen);
    if (n_col) memcpy(h + o_col, fs.col.data(), n_col);
    if (n_coef) memcpy(h + o_coef, ctx->dec_coefs.blocks.data(), n_coef);
    PCC_HIP(hipMemcpyAsync(ctx->d_dec.p, h, staged, hipMemcpyHostToDevice, ctx->stream));
    uint8_t* d = ctx->d_dec.p;
    DecodeArgs da{};
    if (jpeg) {
      IdctArgs ia{};
      ia.blocks = reinterpret_cast<const int16_t*>(d + o_coef);
      memcpy(ia.q, ctx->dec_coefs.q, sizeof(ia.q));
      ia.mcus_x = (uint32_t)mx; ia.mcus_y = (uint32_t)my;
      ia.plane_y = d + o_py; ia.plane_cb = d + o_pcb; ia.plane_cr = d + o_pcr;
      launch_decode_idct(ia, ctx->stream);
      da.plane_y = ia.plane_y; da.plane_cb = ia.plane_cb; da.plane_cr = ia.plane_cr;
      da.img_w = (uint32_t)img_w; da.img_h = (uint32_t)img_h;
      da.y_stride = (uint32_t)(16 * mx); da.c_stride = (uint32_t)(8 * mx);
    }
    da.prefix = reinterpret_cast<const uint64_t*>(d + o_prefix);
    da.prefix_hi = n_hi ? reinterpret_cast<const uint32_t*>(d + o_hi) : nullptr;
    da.first = reinterpret_cast<const uint32_t*>(d + o_first);
    da.bits = d + o_bits;
    da.n_parents = (uint32_t)np;
    da.n_leaves = (uint32_t)L;
    da.res = out->params.octree_resolution;
    for (int a = 0; a < 3; ++a) da.mn[a] = out->bbox[a];
    da.centroid = n_cen ? d + o_cen : nullptr;
    da.colours = n_col ? d + o_col : nullptr;
    da.colour_shift = (fs.cct == 0) ? (uint32_t)(8 - out->params.color_bit_resolution) & 7u : 0u;
    da.with_colour = fs.with_color ? 1 : 0;
    da.points = ctx->d_dec_points.p;
    launch_decode_points(da, ctx->stream);
    PCC_HIP(hipGetLastError());
    PCC_HIP(hipMemcpyAsync(ctx->h_dec_points.p, ctx->d_dec_points.p, 32 * L, hipMemcpyDeviceToHost, ctx->stream));
    { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
    ctx->dec_ms[1] = ms_since(t1);
    ctx->dec_ms[2] = ms_since(t0);
    out->points = ctx->h_dec_points.p;
    out->n = L;
    return PCC_OK;
  } catch (const std::bad_alloc&) {
    return fail(ctx, PCC_ERR_STREAM, "decode: the stream asks for more memory than there is");
  } catch (const std::exception& e) {  // nothing crosses the C ABI as an exception
    return fail(ctx, PCC_ERR_STREAM, e.what());
  }
}

int pcc_get_decode_times(pcc_ctx* ctx, double out_ms[3]) {
  if (!ctx || !out_ms) return PCC_ERR_ARG;
  for (int i = 0; i < 3; ++i) out_ms[i] = ctx->dec_ms[i];
  return PCC_OK;
}

// =====================================================================================================
// The entropy stage of MANY frames with the range coders on the GPU (csrc/pcc_rc_device.hip: one wave per stream).
// For hosts with fewer cores than their GPUs can feed: a CPU core codes ~700 frames/s of the headline size, a GPU
// ~16 000 streams/s -- at 0.1 s per batch, whatever its size, hence batches of hundreds of frames.  What a frame
// needs of its context is copied out when it is added, so the context goes straight back to the GPU stage.
// Same bytes as pcc_entropy_encode.
// =====================================================================================================
struct pcc_entropy_batch {
  pcc_ctx* ctx = nullptr;  // stream, error text
  size_t max_frames = 0;
  struct Frame {
    Bytes header;                // 140-byte frame header
    uint64_t n_branches = 0, n_leaves = 0;
    size_t occ_off = 0, occ_len = 0, hist_off = 0, cen_off = 0, cen_len = 0, col_off = 0, col_len = 0;
    bool has_hist = false, has_cen = false, has_col = false;
    int job[3] = {-1, -1, -1};
  };
  std::vector<Frame> frames;
  PinnedBuf<uint8_t> h_in;       // every stream of the batch, 64-byte aligned, and the occupancy counts
  size_t in_used = 0;
  DevBuf<uint8_t> d_in, d_out, d_packed;
  DevBuf<RcJob> d_jobs;
  DevBuf<uint32_t> d_lens, d_offs, d_hists;
  PinnedBuf<uint32_t> h_lens;
  PinnedBuf<uint8_t> h_packed;
  std::vector<Bytes> streams;    // the finished bitstreams of the last flush
  float gpu_ms = 0.f;

  uint8_t* room(size_t bytes, size_t& off) {  // grows the pinned arena, keeping what is in it
    off = (in_used + 63) & ~(size_t)63;
    const size_t need = off + bytes + 64;
    if (need > h_in.cap) {
      PinnedBuf<uint8_t> bigger;
      if (bigger.ensure(std::max(need, 2 * h_in.cap)) != hipSuccess) return nullptr;
      if (in_used) memcpy(bigger.p, h_in.p, in_used);
      h_in.release();
      h_in = bigger;
    }
    in_used = off + bytes;
    return h_in.p + off;
  }
};

pcc_entropy_batch* pcc_entropy_batch_create(int device, size_t max_frames) {
  pcc_ctx* c = pcc_create(device);
  if (!c) return nullptr;
  pcc_entropy_batch* b = new pcc_entropy_batch();
  b->ctx = c;
  b->max_frames = max_frames ? max_frames : 256;
  return b;
}

// options of the batch's own context ("rc_device_lanes": which form of the device range coder its flushes launch)
int pcc_entropy_batch_set_option(pcc_entropy_batch* b, const char* name, int value) {
  return b ? pcc_set_option(b->ctx, name, value) : PCC_ERR_ARG;
}

void pcc_entropy_batch_destroy(pcc_entropy_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->ctx->device);
  (void)hipStreamSynchronize(b->ctx->stream);
  b->h_in.release(); b->d_in.release(); b->d_out.release(); b->d_packed.release(); b->d_jobs.release(); b->d_lens.release();
  b->d_offs.release(); b->d_hists.release(); b->h_lens.release(); b->h_packed.release();
  pcc_destroy(b->ctx);
  delete b;
}

size_t pcc_entropy_batch_size(pcc_entropy_batch* b) { return b ? b->frames.size() : 0; }
size_t pcc_entropy_batch_capacity(pcc_entropy_batch* b) { return b ? b->max_frames : 0; }
const char* pcc_entropy_batch_last_error(pcc_entropy_batch* b) { return b ? b->ctx->err.c_str() : "no batch (no usable HIP device?)"; }

int pcc_entropy_batch_add(pcc_entropy_batch* b, const pcc_hot_result* hot, const pcc_params* prm) {
  if (!b || !hot || !prm) return PCC_ERR_ARG;
  pcc_ctx* ctx = b->ctx;
  if (b->frames.size() >= b->max_frames) return fail(ctx, PCC_ERR_STATE, "the batch is full: flush it first");
  pcc_entropy_batch::Frame f;
  // a frame that cannot be added leaves nothing behind in the arena
  struct Rollback {
    pcc_entropy_batch* b;
    size_t used;
    bool keep = false;
    ~Rollback() { if (!keep) b->in_used = used; }
  } rollback{b, b->in_used};
  frame_header_bytes(*hot, *prm, f.header);
  f.n_branches = hot->n_branches;
  f.n_leaves = hot->n_leaves;
  uint8_t* p = b->room((size_t)hot->n_branches, f.occ_off);
  if (!p) return fail(ctx, PCC_ERR_HIP, "out of page-locked memory");
  memcpy(p, hot->occupancy, (size_t)hot->n_branches);
  f.occ_len = (size_t)hot->n_branches;
  if (hot->occupancy_histogram) {
    p = b->room(1024, f.hist_off);
    if (!p) return fail(ctx, PCC_ERR_HIP, "out of page-locked memory");
    memcpy(p, hot->occupancy_histogram, 1024);
    f.has_hist = true;
  }
  if (prm->do_voxel_centroid) {
    f.cen_len = (size_t)(3 * hot->n_leaves);
    p = b->room(f.cen_len, f.cen_off);
    if (!p) return fail(ctx, PCC_ERR_HIP, "out of page-locked memory");
    memcpy(p, hot->centroid, f.cen_len);
    f.has_cen = true;
  }
  if (prm->do_color_encoding) {  // the JPEG is put together here, on the host (headers, row stitching): a fraction of a millisecond
    Bytes payload;
    const uint8_t* src = nullptr;
    size_t n = 0;
    colour_stream_source(*hot, *prm, payload, src, n);
    f.col_len = n;
    p = b->room(n, f.col_off);
    if (!p) return fail(ctx, PCC_ERR_HIP, "out of page-locked memory");
    if (n) memcpy(p, src, n);
    f.has_col = true;
  }
  b->frames.push_back(std::move(f));
  rollback.keep = true;
  return (int)b->frames.size() - 1;
}

static int entropy_batch_flush_frames(pcc_entropy_batch* b, pcc_bitstream* out, size_t* n_out);

int pcc_entropy_batch_flush(pcc_entropy_batch* b, pcc_bitstream* out, size_t out_capacity, size_t* n_out) {
  if (!b || (!out && out_capacity) || !n_out) return PCC_ERR_ARG;
  pcc_ctx* ctx = b->ctx;
  *n_out = 0;
  const size_t nf = b->frames.size();
  if (out_capacity < nf) return fail(ctx, PCC_ERR_ARG, "room for fewer bitstreams than frames in the batch");  // (the batch stays as it is)
  b->streams.assign(nf, Bytes());
  if (nf == 0) return PCC_OK;
  const int rc = entropy_batch_flush_frames(b, out, n_out);
  // whatever happened, the batch is empty afterwards: a failed flush loses its frames (the caller is told), it does not
  // poison the flushes that follow
  b->frames.clear();
  b->in_used = 0;
  return rc;
}

static int entropy_batch_flush_frames(pcc_entropy_batch* b, pcc_bitstream* out, size_t* n_out) {
  pcc_ctx* ctx = b->ctx;
  const size_t nf = b->frames.size();
  PCC_HIP(hipSetDevice(ctx->device));
  // jobs: the occupancy bytes (with their counts where the GPU stage delivered them), centroid bytes, colour payload
  std::vector<RcJob> jobs;
  std::vector<size_t> out_off;
  size_t out_bytes = 0;
  PCC_HIP(b->d_in.ensure(b->in_used + 64));
  auto add_job = [&](size_t off, size_t len, bool hist, size_t hist_off) {
    RcJob j;
    j.in = b->d_in.p + off;
    j.n = (uint32_t)len;
    j.hist = hist ? reinterpret_cast<const uint32_t*>(b->d_in.p + hist_off) : nullptr;
    j.out = nullptr;
    j.out_len = nullptr;
    out_off.push_back(out_bytes);
    out_bytes += (1028 + len + len / 2 + 64 + 63) & ~(size_t)63;
    jobs.push_back(j);
    return (int)jobs.size() - 1;
  };
  for (auto& f : b->frames) {
    if (f.occ_len >= (1ull << 31) || f.cen_len >= (1ull << 31) || f.col_len >= (1ull << 31)) return fail(ctx, PCC_ERR_UNSUPPORTED, "stream of 2 GB or more");
    f.job[0] = add_job(f.occ_off, f.occ_len, f.has_hist, f.hist_off);
    f.job[1] = f.has_cen ? add_job(f.cen_off, f.cen_len, false, 0) : -1;
    f.job[2] = f.has_col ? add_job(f.col_off, f.col_len, false, 0) : -1;
  }
  const uint32_t nj = (uint32_t)jobs.size();
  PCC_HIP(b->d_out.ensure(out_bytes + 64));
  PCC_HIP(b->d_lens.ensure(nj));
  PCC_HIP(b->d_offs.ensure(nj));
  PCC_HIP(b->d_jobs.ensure(nj));
  const bool lanes = ctx->rc_lanes;
  if (lanes) PCC_HIP(b->d_hists.ensure((size_t)nj * 256));
  PCC_HIP(b->h_lens.ensure(nj));
  for (uint32_t k = 0; k < nj; ++k) { jobs[k].out = b->d_out.p + out_off[k]; jobs[k].out_len = b->d_lens.p + k; }
  PCC_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
  PCC_HIP(hipMemcpyAsync(b->d_in.p, b->h_in.p, b->in_used, hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipMemcpyAsync(b->d_jobs.p, jobs.data(), (size_t)nj * sizeof(RcJob), hipMemcpyHostToDevice, ctx->stream));
  launch_range_encode(b->d_jobs.p, nj, lanes ? b->d_hists.p : nullptr, ctx->stream);
  PCC_HIP(hipGetLastError());
  PCC_HIP(hipMemcpyAsync(b->h_lens.p, b->d_lens.p, (size_t)nj * 4, hipMemcpyDeviceToHost, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  // the coded streams side by side, one copy back
  std::vector<uint32_t> offs(nj);
  size_t packed = 0;
  for (uint32_t k = 0; k < nj; ++k) { offs[k] = (uint32_t)packed; packed += ((size_t)b->h_lens.p[k] + 15) & ~(size_t)15; }
  if (packed >= (1ull << 32)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 4 GB of coded streams in one batch");
  PCC_HIP(b->d_packed.ensure(packed + 64));
  PCC_HIP(b->h_packed.ensure(packed + 64));
  PCC_HIP(hipMemcpyAsync(b->d_offs.p, offs.data(), (size_t)nj * 4, hipMemcpyHostToDevice, ctx->stream));
  launch_pack_streams(b->d_jobs.p, b->d_offs.p, b->d_packed.p, nj, ctx->stream);
  PCC_HIP(hipGetLastError());
  PCC_HIP(hipMemcpyAsync(b->h_packed.p, b->d_packed.p, packed, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  (void)hipEventElapsedTime(&b->gpu_ms, ctx->ev_begin, ctx->ev_end);
  // writeFrameHeader + entropyEncoding layout (impl.hpp:1682-1760)
  for (size_t i = 0; i < nf; ++i) {
    const pcc_entropy_batch::Frame& f = b->frames[i];
    Bytes& s = b->streams[i];
    s = f.header;
    auto put = [&](const void* q, size_t n) { const uint8_t* u = static_cast<const uint8_t*>(q); s.insert(s.end(), u, u + n); };
    auto coded = [&](int job) { put(b->h_packed.p + offs[(size_t)job], b->h_lens.p[job]); return (uint64_t)b->h_lens.p[job]; };
    const uint64_t nb = f.n_branches;
    put(&nb, 8);
    out[i].perf[0] = coded(f.job[0]);
    out[i].perf[1] = out[i].perf[2] = 0;
    if (f.has_cen) {
      const uint32_t n3 = (uint32_t)f.cen_len;
      put(&n3, 4);
      out[i].perf[1] = coded(f.job[1]);
    }
    if (f.has_col) {
      const uint64_t nc = f.col_len;
      put(&nc, 8);
      out[i].perf[2] = coded(f.job[2]);
    }
    out[i].data = s.data();
    out[i].len = s.size();
  }
  *n_out = nf;
  return PCC_OK;
}

int pcc_device_range_encode(pcc_ctx* ctx, int n_streams, const uint8_t* const* in, const size_t* n, uint8_t* const* out, size_t* out_len,
                            float* gpu_ms) {
  if (!ctx || n_streams < 0 || (n_streams && (!in || !n || !out || !out_len))) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  PCC_HIP(hipSetDevice(ctx->device));
  if (gpu_ms) *gpu_ms = 0.f;
  if (n_streams == 0) return PCC_OK;
  // one device buffer: [inputs | outputs | lengths | jobs]
  std::vector<size_t> in_off((size_t)n_streams), out_off((size_t)n_streams);
  size_t at = 0;
  for (int i = 0; i < n_streams; ++i) {
    if (n[i] >= (1ull << 31) || (n[i] && !in[i]) || !out[i]) return fail(ctx, PCC_ERR_ARG, "pcc_device_range_encode: stream arguments");
    in_off[(size_t)i] = at;
    at += (n[i] + 63) & ~(size_t)63;
  }
  const size_t in_bytes = at;
  for (int i = 0; i < n_streams; ++i) {
    out_off[(size_t)i] = at;
    at += (1028 + n[i] + n[i] / 2 + 64 + 63) & ~(size_t)63;
  }
  const size_t len_off = at;
  at += ((size_t)n_streams * sizeof(uint32_t) + 63) & ~(size_t)63;
  const size_t job_off = at;
  at += ((size_t)n_streams * sizeof(RcJob) + 63) & ~(size_t)63;
  const size_t hist_off = at;   // (the lane-per-stream form counts the symbols of every stream here first)
  const bool lanes = ctx->rc_lanes;
  if (lanes) at += (size_t)n_streams * 256 * sizeof(uint32_t);
  PCC_HIP(ctx->d_rc.ensure(at));
  std::vector<RcJob> jobs((size_t)n_streams);
  for (int i = 0; i < n_streams; ++i) {
    if (n[i]) PCC_HIP(hipMemcpyAsync(ctx->d_rc.p + in_off[(size_t)i], in[i], n[i], hipMemcpyHostToDevice, ctx->stream));
    RcJob& j = jobs[(size_t)i];
    j.in = ctx->d_rc.p + in_off[(size_t)i];
    j.n = (uint32_t)n[i];
    j.hist = nullptr;
    j.out = ctx->d_rc.p + out_off[(size_t)i];
    j.out_len = reinterpret_cast<uint32_t*>(ctx->d_rc.p + len_off) + i;
  }
  (void)in_bytes;
  PCC_HIP(hipMemcpyAsync(ctx->d_rc.p + job_off, jobs.data(), jobs.size() * sizeof(RcJob), hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
  launch_range_encode(reinterpret_cast<const RcJob*>(ctx->d_rc.p + job_off), (uint32_t)n_streams,
                      lanes ? reinterpret_cast<uint32_t*>(ctx->d_rc.p + hist_off) : nullptr, ctx->stream);
  PCC_HIP(hipGetLastError());
  PCC_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
  std::vector<uint32_t> lens((size_t)n_streams);
  PCC_HIP(hipMemcpyAsync(lens.data(), ctx->d_rc.p + len_off, lens.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  for (int i = 0; i < n_streams; ++i) {
    out_len[i] = lens[(size_t)i];
    PCC_HIP(hipMemcpyAsync(out[i], ctx->d_rc.p + out_off[(size_t)i], lens[(size_t)i], hipMemcpyDeviceToHost, ctx->stream));
  }
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  if (gpu_ms) (void)hipEventElapsedTime(gpu_ms, ctx->ev_begin, ctx->ev_end);
  return PCC_OK;
}

size_t pcc_host_range_encode(const uint8_t* in, size_t n, uint8_t* out, size_t out_cap) {
  Bytes b;
  StaticRangeCoder::encode(in, n, b);
  if (b.size() > out_cap) return 0;
  memcpy(out, b.data(), b.size());
  return b.size();
}
int pcc_host_range_encode_many(int count, const uint8_t* const* in, const size_t* n, uint8_t* const* out, const size_t* out_cap,
                               size_t* out_len) {
  if (count < 1 || count > StaticRangeCoder::kMaxStreams || !in || !n || !out || !out_cap || !out_len) return PCC_ERR_ARG;
  try {
    Bytes b[StaticRangeCoder::kMaxStreams];
    Bytes* dst[StaticRangeCoder::kMaxStreams];
    size_t got[StaticRangeCoder::kMaxStreams];
    for (int i = 0; i < count; ++i) dst[i] = &b[i];
    StaticRangeCoder::encode_many(count, in, n, dst, got);
    for (int i = 0; i < count; ++i) {
      out_len[i] = b[i].size() <= out_cap[i] ? b[i].size() : 0;
      if (out_len[i]) memcpy(out[i], b[i].data(), b[i].size());
    }
  } catch (const std::bad_alloc&) {
    return PCC_ERR_HIP;  // out of memory
  }
  return PCC_OK;
}
size_t pcc_host_range_decode(const uint8_t* in, size_t in_len, uint8_t* out, size_t n) {
  return StaticRangeCoder::decode(in, in_len, out, n);
}
size_t pcc_host_jpeg_encode(const uint8_t* rgb, int w, int h, int quality, uint8_t* out, size_t out_cap) {
  if (!rgb || w <= 0 || h <= 0) return 0;
  Bytes b;
  BaselineJpeg::encode_rgb(rgb, w, h, quality, b);
  if (b.size() > out_cap) return 0;
  memcpy(out, b.data(), b.size());
  return b.size();
}
int pcc_host_jpeg_decode(const uint8_t* jpg, size_t len, uint8_t* rgb, size_t rgb_cap, int* w, int* h) {
  Bytes b;
  try {
    if (!w || !h || !BaselineJpeg::decode_rgb(jpg, len, b, *w, *h)) return PCC_ERR_STREAM;
  } catch (const std::bad_alloc&) {
    return PCC_ERR_STREAM;
  }
  if (b.size() > rgb_cap) return PCC_ERR_ARG;
  memcpy(rgb, b.data(), b.size());
  return PCC_OK;
}
uint32_t pcc_host_snake_position(uint32_t i, uint32_t w, uint32_t h) { return snake_position(i, w, h); }
int pcc_debug_host_rc_wide(void) { return StaticRangeCoder::wide_available() ? 1 : 0; }

int pcc_normalize_group_boxes(pcc_point_xyzrgb** clouds, const size_t* sizes, size_t n_clouds, double f, float bb_min[3],
                              float bb_max[3], float* per_cloud) {
  // normalize_pointclouds (impl.hpp:1871-1967): a running box that is re-initialised from a
  // frame's own extent whenever that frame does not fit strictly inside it
  if (!clouds || !sizes || !bb_min || !bb_max) return PCC_ERR_ARG;
  float mnb[3] = {1000.f, 1000.f, 1000.f}, mxb[3] = {-1000.f, -1000.f, -1000.f};
  bool init = false;
  for (size_t k = 0; k < n_clouds; ++k) {
    pcc_point_xyzrgb* c = clouds[k];
    float mn[3] = {3.402823466e+38f, 3.402823466e+38f, 3.402823466e+38f};
    float mx[3] = {-3.402823466e+38f, -3.402823466e+38f, -3.402823466e+38f};
    for (size_t i = 0; i < sizes[k]; ++i) {  // pcl::getMinMax3D skips non-finite points
      const float q[3] = {c[i].x, c[i].y, c[i].z};
      if (!std::isfinite(q[0]) || !std::isfinite(q[1]) || !std::isfinite(q[2])) continue;
      for (int a = 0; a < 3; ++a) { mn[a] = std::min(mn[a], q[a]); mx[a] = std::max(mx[a], q[a]); }
    }
    if (!(mn[0] > mnb[0] && mn[1] > mnb[1] && mn[2] > mnb[2])) init = false;
    if (!(mx[0] < mxb[0] && mx[1] < mxb[1] && mx[2] < mxb[2])) init = false;
    if (!init) {
      for (int a = 0; a < 3; ++a) {
        mnb[a] = (float)((double)mn[a] - f * (double)fabsf(mx[a] - mn[a]));
        mxb[a] = (float)((double)mx[a] + f * (double)fabsf(mx[a] - mn[a]));
      }
      init = true;
    }
    const float dyn[3] = {mxb[0] - mnb[0], mxb[1] - mnb[1], mxb[2] - mnb[2]};
    if (per_cloud)  // bounding_boxes[k] = the box in force for cloud k (impl.hpp:1928-1929)
      for (int a = 0; a < 3; ++a) { per_cloud[6 * k + a] = mnb[a]; per_cloud[6 * k + 3 + a] = mxb[a]; }
    for (size_t i = 0; i < sizes[k]; ++i) {
      c[i].x -= mnb[0]; c[i].y -= mnb[1]; c[i].z -= mnb[2];
      c[i].x /= dyn[0]; c[i].y /= dyn[1]; c[i].z /= dyn[2];
    }
  }
  for (int a = 0; a < 3; ++a) { bb_min[a] = mnb[a]; bb_max[a] = mxb[a]; }
  return PCC_OK;
}

int pcc_restore_scaling(pcc_point_xyzrgb* cloud, size_t n, const float bb_min[3], const float bb_max[3]) {
  if ((!cloud && n) || !bb_min || !bb_max) return PCC_ERR_ARG;
  const float dyn[3] = {bb_max[0] - bb_min[0], bb_max[1] - bb_min[1], bb_max[2] - bb_min[2]};
  for (size_t i = 0; i < n; ++i) {  // impl.hpp:1969-1986: multiply, then add
    cloud[i].x *= dyn[0]; cloud[i].y *= dyn[1]; cloud[i].z *= dyn[2];
    cloud[i].x += bb_min[0]; cloud[i].y += bb_min[1]; cloud[i].z += bb_min[2];
  }
  return PCC_OK;
}

// =====================================================================================================
// inter-frame ("delta") path: encodePointCloudDeltaFrame / decodePointCloudDeltaFrame (impl.hpp:787-1235)
// =====================================================================================================
static_assert(sizeof(pcc_delta_block) == sizeof(BlockResult), "pcc_delta_block mirrors BlockResult");

static uint64_t host_split3(uint64_t x) {
  x &= 0x1fffffULL;
  x = (x | x << 32) & 0x1f00000000ffffULL;
  x = (x | x << 16) & 0x1f0000ff0000ffULL;
  x = (x | x << 8) & 0x100f00f00f00f00fULL;
  x = (x | x << 4) & 0x10c30c30c30c30c3ULL;
  x = (x | x << 2) & 0x1249249249249249ULL;
  return x;
}
static uint64_t host_morton3(uint32_t kx, uint32_t ky, uint32_t kz) {
  return (host_split3(kx) << 2) | (host_split3(ky) << 1) | host_split3(kz);
}

static int delta_subcontexts(pcc_ctx* ctx) {
  for (pcc_ctx*& sc : ctx->sub)
    if (!sc) {
      sc = pcc_create(ctx->device);
      if (!sc) return fail(ctx, PCC_ERR_HIP, "inter-frame path: cannot create a sub-context");
    }
  return PCC_OK;
}

// generate_macroblock_tree (impl.hpp:411-434): an octree over [0,1]^3 at resolution * macroblock size; the leaves are
// the macroblocks.  Only the sorted points and the leaf arrays are needed.
static int launch_block_tree(pcc_ctx* sc, const void* dev_points, size_t n, size_t stride, size_t rgb_off, double res_mb) {
  pcc_params tp{};
  tp.octree_resolution = res_mb;
  tp.point_resolution = res_mb;
  tp.do_color_encoding = 0;
  tp.color_bit_resolution = 8;
  tp.color_coding_type = 3;
  FixedBox box{};
  box.enabled = 1;
  for (int a = 0; a < 3; ++a) { box.mn[a] = 0.0; box.mx[a] = 1.0; }
  return launch_frame(sc, dev_points, n, stride, rgb_off, &tp, &box, 0, 1);
}

static int block_tree_of(pcc_ctx* owner, pcc_ctx* sc, const void* dev_points, size_t stride, size_t rgb_off, BlockTree& t) {
  const int rc = wait_frame_state(sc);
  if (rc != PCC_OK) return fail(owner, rc, std::string("macroblock tree: ") + sc->err);
  const FrameState& st = *sc->h_state.p;
  if (!st.packed) return fail(owner, PCC_ERR_UNSUPPORTED, "macroblock tree: key and point index do not fit 64 bits");
  t.sorted_keys = st.keys_final ? sc->d_keys_b.p : sc->d_keys_a.p;
  t.leaf_start = sc->d_leaf_start.p;
  t.leaf_code = sc->d_leaf_code.p;
  t.prefix_code = host_morton3(st.prefix[0], st.prefix[1], st.prefix[2]);
  t.index_mask = st.ibits >= 64 ? ~0ull : ((1ull << st.ibits) - 1ull);
  t.n_blocks = st.n_leaves;
  t.n_points = st.n_finite;
  t.points = static_cast<const uint8_t*>(dev_points);
  t.stride = (uint32_t)stride;
  t.rgb_off = (uint32_t)rgb_off;
  return PCC_OK;
}

int pcc_encode_delta(pcc_ctx* ctx, const pcc_point_xyzrgb* i_cloud, size_t n_i, const pcc_point_xyzrgb* p_cloud, size_t n_p,
                     const pcc_delta_params* dp, pcc_delta_result* out) {
  if (!ctx || !dp || !out || (!i_cloud && n_i) || (!p_cloud && n_p)) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  memset(out, 0, sizeof(*out));
  const pcc_params& cp = dp->codec;
  if (n_i == 0 || n_p == 0) return fail(ctx, PCC_ERR_EMPTY, "inter-frame coding of an empty cloud");
  if (n_i >= (1ull << 30) || n_p >= (1ull << 30)) return fail(ctx, PCC_ERR_UNSUPPORTED, "more than 2^30 points");
  if (!(cp.octree_resolution > 0.0) || cp.macroblock_size < 1) return fail(ctx, PCC_ERR_ARG, "octree_resolution / macroblock_size");
  PCC_HIP(hipSetDevice(ctx->device));
  { const int rc = delta_subcontexts(ctx); if (rc != PCC_OK) return rc; }
  pcc_ctx *s_simp = ctx->sub[0], *s_it = ctx->sub[1], *s_pt = ctx->sub[2];
  const double res = cp.octree_resolution, res_mb = res * (double)cp.macroblock_size;

  PCC_HIP(ctx->d_delta_i.ensure(32 * n_i));
  PCC_HIP(ctx->d_delta_p.ensure(32 * n_p));
  PCC_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->d_delta_i.p, i_cloud, 32 * n_i, hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->d_delta_p.p, p_cloud, 32 * n_p, hipMemcpyHostToDevice, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }

  // the I frame's macroblocks and the simplification of the P frame are independent: two streams
  int rc = launch_block_tree(s_it, ctx->d_delta_i.p, n_i, 32, 16, res_mb);
  if (rc != PCC_OK) return fail(ctx, rc, "I macroblock tree: " + s_it->err);
  const uint8_t* p_points = ctx->d_delta_p.p;
  size_t p_count = n_p, p_stride = 32, p_rgb = 16;
  if (!dp->icp_on_original) {  // simplifyPCloud (impl.hpp:318-403): voxel centres (or float centroids) and mean colours
    pcc_params sp{};
    sp.octree_resolution = res;
    sp.point_resolution = res;
    sp.do_color_encoding = 1;
    sp.color_bit_resolution = 8;
    sp.color_coding_type = 3;
    sp.do_voxel_centroid = cp.do_voxel_centroid;
    FixedBox box{};
    box.enabled = 1;
    for (int a = 0; a < 3; ++a) { box.mn[a] = 0.0; box.mx[a] = 1.0; }
    rc = launch_frame(s_simp, ctx->d_delta_p.p, n_p, 32, 16, &sp, &box, 1, 0);
    if (rc != PCC_OK) return fail(ctx, rc, "simplification: " + s_simp->err);
    rc = wait_frame_state(s_simp);
    if (rc != PCC_OK) return fail(ctx, rc, "simplification: " + s_simp->err);
    p_points = reinterpret_cast<const uint8_t*>(s_simp->d_simplified.p);
    p_count = s_simp->h_state.p->n_leaves;
    p_stride = 16;
    p_rgb = 12;
  }
  rc = launch_block_tree(s_pt, p_points, p_count, p_stride, p_rgb, res_mb);
  if (rc != PCC_OK) return fail(ctx, rc, "P macroblock tree: " + s_pt->err);

  DeltaArgs da{};
  rc = block_tree_of(ctx, s_it, ctx->d_delta_i.p, 32, 16, da.i_tree);
  if (rc != PCC_OK) return rc;
  rc = block_tree_of(ctx, s_pt, p_points, p_stride, p_rgb, da.p_tree);
  if (rc != PCC_OK) return rc;
  const uint32_t nbi = da.i_tree.n_blocks, nbp = da.p_tree.n_blocks;
  PCC_HIP(ctx->d_ifull.ensure(nbi));
  PCC_HIP(ctx->d_pfull.ensure(nbp));
  PCC_HIP(ctx->d_ixyzc.ensure(da.i_tree.n_points));
  PCC_HIP(ctx->d_pxyzc.ensure(da.p_tree.n_points));
  PCC_HIP(ctx->d_cur.ensure(da.i_tree.n_points));
  PCC_HIP(ctx->d_nn.ensure(da.i_tree.n_points));
  PCC_HIP(ctx->d_blocks.ensure(nbp));
  PCC_HIP(ctx->d_work.ensure(nbp));
  PCC_HIP(ctx->d_order.ensure(nbp));
  PCC_HIP(ctx->d_counts.ensure(4));
  da.work = ctx->d_work.p; da.order = ctx->d_order.p; da.counts = ctx->d_counts.p;
  da.i_full = ctx->d_ifull.p; da.p_full = ctx->d_pfull.p;
  da.i_xyzc = ctx->d_ixyzc.p; da.p_xyzc = ctx->d_pxyzc.p; da.cur = ctx->d_cur.p; da.nn = ctx->d_nn.p;
  da.results = ctx->d_blocks.p;
  da.point_resolution = cp.point_resolution;
  da.octree_resolution = res;
  da.macroblock_size = cp.macroblock_size;
  da.max_iterations = dp->icp_max_iterations > 0 ? dp->icp_max_iterations : 50;
  da.transformation_epsilon = dp->transformation_epsilon > 0.f ? dp->transformation_epsilon : 1e-8f;
  da.var_threshold = dp->icp_var_threshold > 0.f ? dp->icp_var_threshold : 100.f;
  da.do_icp_color_offset = cp.do_icp_color_offset ? 1 : 0;
  // second stream: the residual intra coder's (idle until the blocks are decided); its two timing events do as fork / join
  da.shape = ctx->icp_waves == 4 ? 1 : (ctx->icp_waves == 1 ? 2 : 0);
  launch_delta_blocks(da, ctx->stream, ctx->sub[3]->stream, ctx->sub[3]->ev_begin, ctx->sub[3]->ev_end);
  PCC_HIP(hipGetLastError());
  ctx->h_blocks.resize(nbp);
  ctx->h_pstart.resize((size_t)nbp + 1);
  PCC_HIP(hipMemcpyAsync(ctx->h_blocks.data(), ctx->d_blocks.p, (size_t)nbp * sizeof(BlockResult), hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->h_pstart.data(), da.p_tree.leaf_start, ((size_t)nbp + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }

  // chunks of the predicted blocks, in the depth-first order of the P frame's macroblocks (impl.hpp:1003-1070)
  const uint32_t kNone = 0xffffffffu;
  ctx->p_stream.clear();
  ctx->h_dst_intra.assign(nbp, kNone);
  ctx->h_dst_out.assign(nbp, kNone);
  ctx->h_mdec.assign((size_t)nbp * 16, 0.f);
  uint32_t shared = 0, converged = 0;
  size_t intra_n = 0, out_n = 0;
  std::vector<int16_t> comp;
  for (uint32_t b = 0; b < nbp; ++b) {
    const BlockResult& r = ctx->h_blocks[b];
    if (r.i_block >= 0) ++shared;
    const bool predicted = r.i_block >= 0 && r.do_icp && r.converged;
    if (predicted) {
      ++converged;
      rigid_compress(r.rt, comp);
      const size_t chunk = 3 * 2 + comp.size() * 2 + (cp.do_icp_color_offset ? 3 : 0);
      ctx->p_stream.push_back((uint8_t)chunk);
      for (int a = 0; a < 3; ++a) { const int16_t k = (int16_t)r.key[a]; ctx->p_stream.push_back((uint8_t)(k & 0xff)); ctx->p_stream.push_back((uint8_t)((k >> 8) & 0xff)); }
      for (int16_t v : comp) { ctx->p_stream.push_back((uint8_t)(v & 0xff)); ctx->p_stream.push_back((uint8_t)((v >> 8) & 0xff)); }
      if (cp.do_icp_color_offset) for (int a = 0; a < 3; ++a) ctx->p_stream.push_back((uint8_t)r.rgb_offsets[a]);
      rigid_decompress(comp.data(), comp.size(), &ctx->h_mdec[(size_t)b * 16]);  // the out cloud shows what the decoder will see
      if (dp->write_out_cloud) { ctx->h_dst_out[b] = (uint32_t)out_n; out_n += r.n_i; }
    } else {
      ctx->h_dst_intra[b] = (uint32_t)intra_n; intra_n += r.n_p;
      if (dp->write_out_cloud) { ctx->h_dst_out[b] = (uint32_t)out_n; out_n += r.n_p; }
    }
  }

  PCC_HIP(ctx->d_dst_intra.ensure(nbp));
  PCC_HIP(ctx->d_dst_out.ensure(nbp));
  PCC_HIP(ctx->d_mdec.ensure((size_t)nbp * 16));
  PCC_HIP(ctx->d_delta_intra.ensure(32 * intra_n + 32));
  PCC_HIP(ctx->d_delta_out.ensure(32 * out_n + 32));
  PCC_HIP(hipMemcpyAsync(ctx->d_dst_intra.p, ctx->h_dst_intra.data(), (size_t)nbp * 4, hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->d_dst_out.p, ctx->h_dst_out.data(), (size_t)nbp * 4, hipMemcpyHostToDevice, ctx->stream));
  PCC_HIP(hipMemcpyAsync(ctx->d_mdec.p, ctx->h_mdec.data(), (size_t)nbp * 64, hipMemcpyHostToDevice, ctx->stream));
  GatherArgs ga{};
  ga.i_xyzc = da.i_xyzc; ga.p_xyzc = da.p_xyzc;
  ga.i_leaf_start = da.i_tree.leaf_start; ga.p_leaf_start = da.p_tree.leaf_start;
  ga.results = da.results; ga.mdec = ctx->d_mdec.p;
  ga.dst_intra = ctx->d_dst_intra.p; ga.dst_out = ctx->d_dst_out.p;
  ga.n_blocks = nbp;
  ga.do_icp_color_offset = cp.do_icp_color_offset ? 1 : 0;
  ga.colour_doubled = 0;
  ga.out_intra = ctx->d_delta_intra.p; ga.out_cloud = ctx->d_delta_out.p;
  launch_delta_gather(ga, ctx->stream);
  PCC_HIP(hipGetLastError());
  PCC_HIP(ctx->delta_cloud.ensure(out_n + 1));
  ctx->delta_cloud_n = out_n;
  if (out_n) PCC_HIP(hipMemcpyAsync(ctx->delta_cloud.p, ctx->d_delta_out.p, 32 * out_n, hipMemcpyDeviceToHost, ctx->stream));
  PCC_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
  { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end);

  // the points that were not predicted go through the intra coder; it is built with the constructor's defaults for the
  // arguments the reference does not pass (impl.hpp:1089-1101 vs codec.h:108-121), and a fresh coder's first frame has id 1
  ctx->i_stream.clear();
  if (intra_n) {
    pcc_params ip = cp;
    ip.do_color_encoding = 1;
    ip.create_scalable = 1;
    ip.do_connectivity = 0;
    ip.jpeg_quality = 75;
    ip.macroblock_size = 16;
    ip.do_icp_color_offset = 0;
    ip.frame_id = 1;
    pcc_bitstream bs;
    pcc_ctx* s_intra = ctx->sub[3];  // a coder of its own, as in the reference: this one's getOutputCloud() stays the I frame's
    rc = pcc_encode_intra_device(s_intra, ctx->d_delta_intra.p, intra_n, 32, 16, &ip, &bs);
    if (rc != PCC_OK && rc != PCC_ERR_EMPTY) return fail(ctx, rc, "residual intra coder: " + s_intra->err);
    if (rc == PCC_OK) {
      ctx->i_stream.assign(bs.data, bs.data + bs.len);
      ms += s_intra->last_hot.gpu_ms;
    }
  }
  out->i_data = ctx->i_stream.data(); out->i_len = ctx->i_stream.size();
  out->p_data = ctx->p_stream.data(); out->p_len = ctx->p_stream.size();
  out->out_cloud = ctx->delta_cloud.p; out->out_n = out_n;
  out->macro_block_count = nbp; out->shared_macroblock_count = shared; out->convergence_count = converged;
  out->shared_macroblock_percentage = (float)shared / (float)nbp;                 // impl.hpp:1105-1106
  out->shared_macroblock_convergence_percentage = (float)converged / (float)shared;
  out->n_intra_points = intra_n;
  out->n_simplified = dp->icp_on_original ? 0 : p_count;
  out->gpu_ms = ms;
  return PCC_OK;
}

int pcc_delta_blocks(pcc_ctx* ctx, const pcc_delta_block** blocks, size_t* n) {
  if (!ctx || !blocks || !n) return PCC_ERR_ARG;
  *blocks = reinterpret_cast<const pcc_delta_block*>(ctx->h_blocks.data());
  *n = ctx->h_blocks.size();
  return PCC_OK;
}

int pcc_decode_delta(pcc_ctx* ctx, const pcc_point_xyzrgb* i_cloud, size_t n_i, const uint8_t* i_stream, size_t i_len,
                     const uint8_t* p_stream, size_t p_len, const pcc_delta_params* dp, pcc_cloud* out) {
  if (!ctx || !dp || !out || (!i_cloud && n_i) || (!i_stream && i_len) || (!p_stream && p_len)) return PCC_ERR_ARG;
  PCC_NEED_GPU();
  memset(out, 0, sizeof(*out));
  const pcc_params& cp = dp->codec;
  if (!(cp.octree_resolution > 0.0) || cp.macroblock_size < 1) return fail(ctx, PCC_ERR_ARG, "octree_resolution / macroblock_size");
  PCC_HIP(hipSetDevice(ctx->device));
  ctx->delta_cloud_n = 0;
  bool delta_copy_pending = false;
  static const bool trace = dev_env("PCC_TRACE_DELTA") != nullptr;  // developer knob: where the call's time goes
  timespec tr0;
  clock_gettime(CLOCK_MONOTONIC, &tr0);
  auto mark = [&](const char* what) {
    if (!trace) return;
    timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);
    fprintf(stderr, "[pcc_decode_delta] %-28s %8.3f ms\n", what, (double)(t.tv_sec - tr0.tv_sec) * 1e3 + (double)(t.tv_nsec - tr0.tv_nsec) * 1e-6);
  };
  size_t out_n = 0;
  if (n_i && p_len) {
    { const int rc = delta_subcontexts(ctx); if (rc != PCC_OK) return rc; }
    pcc_ctx* s_it = ctx->sub[1];
    PCC_HIP(ctx->d_delta_i.ensure(32 * n_i));
    PCC_HIP(hipMemcpyAsync(ctx->d_delta_i.p, i_cloud, 32 * n_i, hipMemcpyHostToDevice, ctx->stream));
    { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
    mark("I cloud uploaded");
    int rc = launch_block_tree(s_it, ctx->d_delta_i.p, n_i, 32, 16, cp.octree_resolution * (double)cp.macroblock_size);
    if (rc != PCC_OK) return fail(ctx, rc, "I macroblock tree: " + s_it->err);
    DeltaArgs da{};
    rc = block_tree_of(ctx, s_it, ctx->d_delta_i.p, 32, 16, da.i_tree);
    mark("I macroblock tree");
    if (rc == PCC_OK) {
      const uint32_t nbi = da.i_tree.n_blocks;
      PCC_HIP(ctx->d_ifull.ensure(nbi));
      PCC_HIP(ctx->d_ixyzc.ensure(da.i_tree.n_points));
      da.i_full = ctx->d_ifull.p; da.i_xyzc = ctx->d_ixyzc.p;
      launch_delta_blocks(da, ctx->stream);  // no P tree: only the I frame's keys and block-ordered points
      PCC_HIP(hipGetLastError());
      ctx->h_ifull.resize(nbi);
      ctx->h_istart.resize((size_t)nbi + 1);
      PCC_HIP(hipMemcpyAsync(ctx->h_ifull.data(), ctx->d_ifull.p, (size_t)nbi * 8, hipMemcpyDeviceToHost, ctx->stream));
      PCC_HIP(hipMemcpyAsync(ctx->h_istart.data(), da.i_tree.leaf_start, ((size_t)nbi + 1) * 4, hipMemcpyDeviceToHost, ctx->stream));
      { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; }
      mark("block keys on the host");
      // the chunks (impl.hpp:1135-1200)
      ctx->h_blocks.clear();
      ctx->h_mdec.clear();
      ctx->h_dst_out.clear();
      size_t pos = 0;
      const size_t off_bytes = cp.do_icp_color_offset ? 3 : 0;
      std::vector<int16_t> comp;
      while (pos < p_len) {
        const size_t chunk = p_stream[pos++];
        if (chunk == 0 || pos + chunk > p_len || chunk < 6 + off_bytes) break;
        int16_t key[3];
        for (int a = 0; a < 3; ++a) key[a] = (int16_t)(p_stream[pos + 2 * a] | (p_stream[pos + 2 * a + 1] << 8));
        const size_t n_comp = (chunk - 6 - off_bytes) / 2;
        comp.resize(n_comp);
        for (size_t k = 0; k < n_comp; ++k) comp[k] = (int16_t)(p_stream[pos + 6 + 2 * k] | (p_stream[pos + 6 + 2 * k + 1] << 8));
        BlockResult r{};
        if (cp.do_icp_color_offset) for (int a = 0; a < 3; ++a) r.rgb_offsets[a] = (int8_t)p_stream[pos + 6 + 2 * n_comp + a];
        pos += chunk;
        if (key[0] < 0 || key[1] < 0 || key[2] < 0 || (n_comp != 6 && n_comp < 10)) continue;
        const uint64_t full = host_morton3((uint32_t)key[0], (uint32_t)key[1], (uint32_t)key[2]);
        const auto it = std::lower_bound(ctx->h_ifull.begin(), ctx->h_ifull.end(), full);
        if (it == ctx->h_ifull.end() || *it != full) continue;  // "no corresponding i block"
        r.i_block = (int32_t)(it - ctx->h_ifull.begin());
        r.n_i = ctx->h_istart[r.i_block + 1] - ctx->h_istart[r.i_block];
        float m[16];
        rigid_decompress(comp.data(), comp.size(), m);
        ctx->h_blocks.push_back(r);
        ctx->h_mdec.insert(ctx->h_mdec.end(), m, m + 16);
        ctx->h_dst_out.push_back((uint32_t)out_n);
        out_n += r.n_i;
      }
      const uint32_t nch = (uint32_t)ctx->h_blocks.size();
      mark("chunks parsed");
      if (nch) {
        ctx->h_dst_intra.assign(nch, 0xffffffffu);
        ctx->h_pstart.assign((size_t)nch + 1, 0u);
        PCC_HIP(ctx->d_blocks.ensure(nch));
        PCC_HIP(ctx->d_dst_intra.ensure(nch));
        PCC_HIP(ctx->d_dst_out.ensure(nch));
        PCC_HIP(ctx->d_fake_start.ensure((size_t)nch + 1));
        PCC_HIP(ctx->d_mdec.ensure((size_t)nch * 16));
        PCC_HIP(ctx->d_delta_out.ensure(32 * out_n + 32));
        PCC_HIP(hipMemcpyAsync(ctx->d_blocks.p, ctx->h_blocks.data(), (size_t)nch * sizeof(BlockResult), hipMemcpyHostToDevice, ctx->stream));
        PCC_HIP(hipMemcpyAsync(ctx->d_dst_intra.p, ctx->h_dst_intra.data(), (size_t)nch * 4, hipMemcpyHostToDevice, ctx->stream));
        PCC_HIP(hipMemcpyAsync(ctx->d_dst_out.p, ctx->h_dst_out.data(), (size_t)nch * 4, hipMemcpyHostToDevice, ctx->stream));
        PCC_HIP(hipMemcpyAsync(ctx->d_fake_start.p, ctx->h_pstart.data(), ((size_t)nch + 1) * 4, hipMemcpyHostToDevice, ctx->stream));
        PCC_HIP(hipMemcpyAsync(ctx->d_mdec.p, ctx->h_mdec.data(), (size_t)nch * 64, hipMemcpyHostToDevice, ctx->stream));
        GatherArgs ga{};
        ga.i_xyzc = da.i_xyzc; ga.p_xyzc = nullptr;
        ga.i_leaf_start = da.i_tree.leaf_start; ga.p_leaf_start = ctx->d_fake_start.p;
        ga.results = ctx->d_blocks.p; ga.mdec = ctx->d_mdec.p;
        ga.dst_intra = ctx->d_dst_intra.p; ga.dst_out = ctx->d_dst_out.p;
        ga.n_blocks = nch;
        ga.do_icp_color_offset = cp.do_icp_color_offset ? 1 : 0;
        ga.colour_doubled = 1;  // p.r += p.r + offset (impl.hpp:1187-1189)
        ga.out_intra = nullptr; ga.out_cloud = ctx->d_delta_out.p;
        launch_delta_gather(ga, ctx->stream);
        PCC_HIP(hipGetLastError());
        // room for the intra coded points behind the predicted ones (at most 16384 per byte of a corrupt stream: bounded by decode_frame)
        PCC_HIP(ctx->delta_cloud.ensure(out_n + 1));
        ctx->delta_cloud_n = out_n;
        PCC_HIP(hipMemcpyAsync(ctx->delta_cloud.p, ctx->d_delta_out.p, 32 * out_n, hipMemcpyDeviceToHost, ctx->stream));
        // the copy lands while the host decodes the intra part below
        delta_copy_pending = true;
        mark("gather enqueued");
      }
    } else if (rc != PCC_ERR_EMPTY) {
      return rc;
    }
  }
  if (i_len) {  // the intra coded points follow the predicted ones (impl.hpp:1207-1229)
    pcc_cloud ic;
    // the decoder with its data-parallel half on the GPU; it waits for the context's stream, i.e. also for the copy above
    const int rc = pcc_decode_intra_gpu(ctx, i_stream, i_len, &ic);
    if (rc != PCC_OK) return fail(ctx, rc, "decode: intra part of the delta frame: header not found, or stream truncated/corrupt");
    out->params = ic.params; out->depth = ic.depth; out->consumed = ic.consumed;
    for (int a = 0; a < 6; ++a) out->bbox[a] = ic.bbox[a];
    const pcc_point_xyzrgb* intra_points = ic.points;
    mark("intra part decoded");
    const size_t n_intra = ic.n;
    if (ctx->delta_cloud.cap < ctx->delta_cloud_n + n_intra + 1) {  // grow, keeping the predicted points
      if (delta_copy_pending) { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; delta_copy_pending = false; }
      PinnedBuf<pcc_point_xyzrgb> bigger;
      PCC_HIP(bigger.ensure(ctx->delta_cloud_n + n_intra + 1));
      if (ctx->delta_cloud_n) memcpy(bigger.p, ctx->delta_cloud.p, ctx->delta_cloud_n * sizeof(pcc_point_xyzrgb));
      ctx->delta_cloud.release();
      ctx->delta_cloud = bigger;
    }
    if (n_intra) memcpy(ctx->delta_cloud.p + ctx->delta_cloud_n, intra_points, n_intra * sizeof(pcc_point_xyzrgb));
    ctx->delta_cloud_n += n_intra;
  }
  if (delta_copy_pending) { const int wrc = wait_stream(ctx); if (wrc != PCC_OK) return wrc; delta_copy_pending = false; }
  mark("done");
  out->points = ctx->delta_cloud.p;
  out->n = ctx->delta_cloud_n;
  return PCC_OK;
}

size_t pcc_host_rigid_compress(const float tr[16], int16_t* comp_out, size_t cap) {
  if (!tr || !comp_out) return 0;
  std::vector<int16_t> comp;
  rigid_compress(tr, comp);
  if (comp.size() > cap) return 0;
  memcpy(comp_out, comp.data(), comp.size() * sizeof(int16_t));
  return comp.size();
}
int pcc_host_rigid_decompress(const int16_t* comp, size_t count, float tr_out[16]) {
  if (!comp || !tr_out || (count != 6 && count != 10)) return PCC_ERR_ARG;
  rigid_decompress(comp, count, tr_out);
  return PCC_OK;
}

}  // extern "C"
