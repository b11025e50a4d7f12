// pcc_device.h -- data layout shared by the HIP kernels and the host-side launcher.
//
// HBM layout of one frame in flight (all arrays allocated once per context for the
// largest N seen; 288 GB of HBM3E makes worst-case sizing cheap and removes every
// size-dependent host<->device round trip from the middle of the pipeline):
//
//   points      N x stride B   caller's pcl::PointXYZRGB array (x,y,z at 0, colour word at rgb_off)
//   chunk_box   ceil(N/2048) x 32 B   per-chunk AABB + first finite index + finite count
//   state       1 x FrameState  epochs of the adaptive bounding box, sort geometry, L, B
//   keys[2]     N x u64         sort keys, ping-pong; ~0 marks a non-finite point.  [morton | colour] or the morton code
//                               alone when nobody needs the point index (no centroids), else [morton | index]
//   idx[2]      N x u32         sort payload, only with an index in the key: the point's colour word (so that no random
//                               gather is needed after the sort), or the point index in pairs mode (code + index bits > 64)
//   hist_rows   ceil(N/4096) x kMaxPasses x 512 x u32   per-tile digit counts of every pass (from k_make_keys)
//   digit_tot   kMaxPasses x 512 x u32                  column sums of hist_rows
//   status      kMaxPasses x (tiles + tiles/16) x 512 x u32   look-back words of the sort passes: per tile, per group of 16 tiles
//   leaf_status ceil(N/4096) x u64                      decoupled look-back words of the leaf scan
//   tickets     (kMaxPasses + 1) x u32                  dynamic tile ids (forward progress of the look-back)
//   leaf_start  (N+1) x u32     first sorted position of each leaf
//   leaf_code   N x u64         morton code of each leaf (sorted, unique)
//   leaf_base   N x u32         DFS byte offset of the first branch node a leaf opens
//   leaf_t      N x u8          number of branch nodes a leaf opens
//   occupancy   N x 21 B (+pad) DFS occupancy stream, worst case L*D
//   bgr / centroid 3N B each, image 3*256*(N/256+1) B, simplified N x 16 B
#pragma once
#include <stdint.h>

namespace pcc {

constexpr int kBlock = 256;        // threads per workgroup (4 wave64)
constexpr int kItems = 8;          // items per thread in tiled kernels
constexpr int kTile = kBlock * kItems;  // 2048 items per tile / bbox chunk
constexpr int kMaxEpochs = 40;     // depth <= 32 => at most 33 growth events (+ first point)
constexpr int kMaxDepth = 21;      // deepest tree whose Morton code (3 bits per level) fits one 64-bit word
// Deeper trees (PCL allows them; a frame gets one when extent / resolution passes 2^19 or so) use TWO-WORD codes:
// `lo` = the 21 low triples (63 bits, the sort key), `hi` = the triples above (up to 10: the sort's u32 payload); what
// would ride in the payload otherwise (point index or colour word) rides in a second payload array.  Every kernel that
// looks at a code has a DEEP instantiation; frames of up to 21 levels run the single-word instantiations unchanged.
constexpr int kMaxDepthDeep = 31;  // 10 triples in the 32-bit high word (and PCL's own box growth stops at 31)
// sort: onesweep-style LSD radix sort, one kernel per pass, digit width chosen per frame (<= 9 bits)
constexpr int kSortThreads = 1024;         // 16 wave64 per workgroup: four per SIMD hide the ranking latencies
constexpr int kSortItems = 4;              // keys per thread
constexpr int kSortTile = kSortThreads * kSortItems;  // 4096 keys per tile
constexpr int kMaxDigitBits = 9;
constexpr int kMaxBins = 1 << kMaxDigitBits;  // 512 = one digit per thread in the look-back
constexpr int kMaxPasses = 11;             // 63 low code bits / 9 + 30 high code bits / 9 (deep trees); 7 for single-word codes
constexpr uint32_t kStatusAggregate = 1u << 30, kStatusInclusive = 2u << 30, kStatusValue = (1u << 30) - 1u;
constexpr int kLookBackGroup = 16;         // tiles per look-back group (two-level look-back)
constexpr uint32_t kSpinLimit = 1u << 20;  // bounded polling: a lost predecessor becomes an error, not a hang

struct ChunkBox {          // 32 bytes
  float mn[3];
  float mx[3];
  int32_t first_finite;    // global index of the first finite point in the chunk, or -1
  int32_t n_finite;
};

enum FrameError : int32_t {
  kErrNone = 0,
  kErrDepth = 1,           // depth > kMaxDepthDeep
  kErrPrefix = 2,          // a key fell outside the predicted varying-bit window (should not happen)
  kErrEpochs = 3,          // more than kMaxEpochs growth epochs
  kErrPasses = 4,          // the frame needs more sort passes than the host enqueued (host re-launches)
  kErrSpin = 5,            // a look-back poll ran into kSpinLimit (should not happen)
  kErrDeep = 6,            // the tree is deeper than kMaxDepth and the host enqueued the single-word kernels (host re-launches)
};

struct FrameState {
  // ---- adaptive bounding box (P2) ----
  double mn[3], mx[3];               // final box (header bytes)
  int32_t depth;                     // final depth D
  int32_t n_epochs;                  // 0 => no finite point
  int32_t first_finite;
  uint32_t n_finite;
  int32_t ep_index[kMaxEpochs];      // epoch e holds for point indices in [ep_index[e], ep_index[e+1])
  double ep_mn[kMaxEpochs][3];       // box origin in force during epoch e
  uint32_t ep_shift[kMaxEpochs][3];  // key offset from the re-rootings that came after epoch e
  int32_t n_growth_events;           // raw growth count (reporting)
  // ---- sort geometry ----
  int32_t vbits_axis;                // varying key bits per axis
  int32_t vbits;                     // 3 * vbits_axis
  int32_t ibits;                     // index bits in the packed key (0 in pairs mode)
  int32_t packed;                    // 1: [code|index] in one u64;  0: u64 code keys + u32 index payload
  int32_t payload;                   // what the u32 payload of the sort carries: 0 nothing, 1 point index (pairs
                                     // mode), 2 the point's colour word (packed mode with colour: no gather later)
  int32_t colour_in_key;             // 1: the low 24 key bits (ibits = 24) are the point's colour, no payload, no index
  int32_t keys_final;                // which key buffer the last sort pass wrote: 0 = a, 1 = b
  int32_t deep;                      // 1: two-word codes (depth > kMaxDepth): key = low 63 code bits, payload = 3: the high code bits
  int32_t payload2;                  // deep only, the second payload array: 0 nothing, 1 point index, 2 the point's colour word
  // cell ranks: the code that is SORTED may be shorter than the 3 * vbits_axis varying Morton bits.  A cloud that
  // straddles a high power-of-two boundary of its box varies in key bits far above its extent (a 1024-voxel capture in a
  // box of 8192: 13 bits per axis, 39 code bits, five sort passes), but touches only a few cells of side 2^code_low_bits
  // up there: the high part of the code is replaced by the cell's rank in Morton order (order preserving), which
  // k_leaf_scan turns back into the Morton bits (cell_abs) before anything downstream sees a code.
  int32_t code_low_bits;             // m: key bits per axis that go into the sorted code verbatim (= vbits_axis: no ranks)
  int32_t code_bits;                 // 3 * m + bits of a rank: what the sort passes cover
  uint32_t cell_base[3], cell_dim[3];  // the cells the cloud's box touches: first cell per axis (key >> m), cells per axis
  uint8_t cell_rank[64];             // cell (dz + dim_z * (dy + dim_y * dx)) -> rank
  uint64_t cell_abs[64];             // rank -> Morton code of the cell's varying high key bits
  int32_t npasses;                   // radix passes actually needed (>= 1)
  int32_t pass_bits[kMaxPasses];     // digit width of each pass
  int32_t pass_shift[kMaxPasses];    // bit position of each digit inside the code (add ibits for the packed key)
  int32_t passes_launched;           // what the host enqueued (k_boxes_events checks npasses against it)
  uint32_t prefix[3];                // constant high key bits per axis (in place)
  // ---- leaves ----
  uint32_t n_leaves;                 // L
  uint32_t n_branches;               // B
  int32_t error;                     // FrameError
  // ---- for the host entropy stage ----
  uint32_t occ_hist[256];            // how often each occupancy byte value occurs (k_occ_histogram): the range coder's table
  uint32_t jpeg_line_words;          // colour coding type 2: words of the strips' bit strings written so far (cursor of k_jpeg_lines)
};

}  // namespace pcc
