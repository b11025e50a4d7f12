// Runtime of the wave64 executor (see include/hip/hip_runtime.h): TEST INFRASTRUCTURE ONLY.
//
//   launch      -> the workgroups of the grid are handed out in blockIdx order to a pool of OS threads; the call returns
//                  when the grid is through (streams are synchronous)
//   workgroup   -> one OS thread; its LDS is that thread's thread-local storage
//   lane        -> a fibre (own stack, hand-written x86-64 context switch); the lanes of a wave run one after the other
//                  up to their next meeting point: a cross-lane operation, a barrier, a sleep, the end of the kernel
//   cross-lane  -> executed by the wave's scheduler once every lane of the wave that can still get there has arrived
//                  (lanes waiting at a barrier or finished are inactive, exactly the EXEC mask of the hardware)
//   barrier     -> released when every lane of the workgroup has arrived or finished
//   s_sleep     -> the OS thread yields: polls on other workgroups' words make progress because those workgroups run
//                  on other threads of the pool (at least two)
#include <hip/hip_runtime.h>

#include "race.h"

#include <pthread.h>
#include <sched.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

thread_local emu::Idx3 threadIdx, blockIdx, blockDim, gridDim;

// AddressSanitizer builds of the executor (make OUT=_build_asan OPT="-O1 -fsanitize=address"): the lanes' stacks are mapped and
// unmapped here, and a region that still carries the redzone poison of frames that once lived on it must not be handed back
// to the process like that (the next owner of the addresses would be reported for touching "stack redzones")
extern "C" void __asan_unpoison_memory_region(void const volatile*, size_t) __attribute__((weak));

namespace emu {
namespace {

// ------------------------------------------------------------------ context switch
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

enum State : int { kRunnable = 0, kCollective, kBarrier, kSleep, kDone };
enum Kind : int { kBallot = 1, kDpp, kReadlane, kReadfirst, kShfl, kWaveBarrier, kSync, kSyncAnd, kSyncOr, kSyncCount };

struct Lane {
  void* sp;
  int state;
  int kind;
  int site;    // the source line through which the lane came to the meeting point it waits at
  uint64_t a0, a1;  // operands
  int c0, c1, c2, c3;  // constants of the operation (must agree over the wave)
  uint64_t result;
};

constexpr size_t kStackBytes = 256 * 1024;
constexpr int kMaxThreads = 1024;

struct Worker {
  char* stacks = nullptr;
  Lane lanes[kMaxThreads];
  void* sched_sp = nullptr;
  int cur = -1;          // lane that is running
  int nthreads = 0;
  Launcher* job = nullptr;
  const char* kname = "";
  uint32_t block_linear = 0;
};
thread_local Worker* tl_worker = nullptr;

[[noreturn]] void die(const char* what) {
  Worker* w = tl_worker;
  fprintf(stderr, "wave64 executor: %s (kernel %s, workgroup %u, lane %d)\n", what, w ? w->kname : "?", w ? w->block_linear : 0u, w ? w->cur : -1);
  abort();
}

void lane_entry() {
  Worker* w = tl_worker;
  w->job->run_lane();
  w = tl_worker;
  Lane& l = w->lanes[w->cur];
  l.state = kDone;
  void* dummy;
  emu_switch(&dummy, w->sched_sp);
  die("a finished lane was resumed");
}

void prepare_lane(Worker* w, int i) {
  char* top = w->stacks + (size_t)(i + 1) * kStackBytes;
  if (__asan_unpoison_memory_region) __asan_unpoison_memory_region(top - kStackBytes, kStackBytes);  // (a lane that ended inside frames left their redzones)
  uintptr_t sp = (uintptr_t)top & ~(uintptr_t)15;
  // layout popped by emu_switch: r15 r14 r13 r12 rbx rbp, then the return address; after `ret` rsp % 16 must be 8
  sp -= 8;  // alignment slot: entry sees rsp % 16 == 8
  uint64_t* s = reinterpret_cast<uint64_t*>(sp);
  *--s = (uint64_t)(uintptr_t)&lane_entry;
  for (int k = 0; k < 6; ++k) *--s = 0;
  w->lanes[i].sp = s;
  w->lanes[i].state = kRunnable;
  w->lanes[i].kind = 0;
}

// called on a lane's stack: hand control back to the scheduler of the workgroup
inline uint64_t yield_lane(int site, int state, int kind, uint64_t a0, uint64_t a1, int c0 = 0, int c1 = 0, int c2 = 0, int c3 = 0) {
  Worker* w = tl_worker;
  if (!w || w->cur < 0) die("device operation outside a kernel");
  Lane& l = w->lanes[w->cur];
  l.site = site;
  l.state = state; l.kind = kind; l.a0 = a0; l.a1 = a1; l.c0 = c0; l.c1 = c1; l.c2 = c2; l.c3 = c3;
  emu_switch(&l.sp, w->sched_sp);
  return tl_worker->lanes[tl_worker->cur].result;
}

// ------------------------------------------------------------------ cross-lane operations
// source lane of a DPP control for destination lane i, or -1 (out of range)
int dpp_source(int ctrl, int i) {
  const int row = i & ~15, r = i & 15;
  if (ctrl >= 0x000 && ctrl <= 0x0ff) return (i & ~3) | ((ctrl >> (2 * (i & 3))) & 3);  // quad_perm
  if (ctrl >= 0x101 && ctrl <= 0x10f) { const int s = r + (ctrl & 15); return s < 16 ? row + s : -1; }   // row_shl
  if (ctrl >= 0x111 && ctrl <= 0x11f) { const int s = r - (ctrl & 15); return s >= 0 ? row + s : -1; }   // row_shr
  if (ctrl >= 0x121 && ctrl <= 0x12f) return row + ((r - (ctrl & 15)) & 15);                           // row_ror
  switch (ctrl) {
    case 0x130: return i + 1 < 64 ? i + 1 : -1;   // wave_shl:1
    case 0x134: return (i + 1) & 63;              // wave_rol:1
    case 0x138: return i - 1;                     // wave_shr:1 (lane 0: -1)
    case 0x13c: return (i - 1) & 63;              // wave_ror:1
    case 0x140: return row + (15 - r);            // row_mirror
    case 0x141: return row + (r & 8) + (7 - (r & 7));  // row_half_mirror
    case 0x142: return row >= 16 ? row - 1 : -1;  // row_bcast:15: lane 15 of the row before
    case 0x143: return i >= 32 ? 31 : -1;         // row_bcast:31: lane 31 into the upper half
  }
  die("unknown DPP control");
}

void run_collective(Worker* w, int wave_first, uint64_t active) {
  Lane* L = w->lanes + wave_first;
  int first = __builtin_ctzll(active);
  const int kind = L[first].kind;
  for (int i = 0; i < 64; ++i)
    if ((active >> i) & 1) {
      // On the chip lanes of a wave that stand at two different instructions never execute them together: each call site runs
      // with the lanes that are there.  Here all waiting lanes would be served as one operation -- so lanes of one wave meeting
      // at different call sites is something the executor cannot stand in for, and says so instead of computing something.
      if (L[i].site != L[first].site) {
        fprintf(stderr, "wave64 executor: lanes %d and %d of a wave wait at cross-lane operations of two different source lines (%d, %d)\n", first, i,
                L[first].site, L[i].site);
        die("cross-lane operation reached by lanes of one wave through different source lines");
      }
      if (L[i].kind != kind || L[i].c0 != L[first].c0 || L[i].c1 != L[first].c1 || L[i].c2 != L[first].c2 || L[i].c3 != L[first].c3) {
        fprintf(stderr, "wave64 executor: lanes %d and %d of a wave are at different cross-lane operations (%d/%x vs %d/%x)\n", first, i, kind,
                L[first].c0, L[i].kind, L[i].c0);
        die("divergent cross-lane operation");
      }
    }
  auto on = [&](int i) { return i >= 0 && i < 64 && ((active >> i) & 1); };
  switch (kind) {
    case kBallot: {
      uint64_t m = 0;
      for (int i = 0; i < 64; ++i) if (on(i) && L[i].a0) m |= 1ull << i;
      for (int i = 0; i < 64; ++i) if (on(i)) L[i].result = m;
      break;
    }
    case kDpp: {
      const int ctrl = L[first].c0, row_mask = L[first].c1, bank_mask = L[first].c2, bound = L[first].c3;
      for (int i = 0; i < 64; ++i) {
        if (!on(i)) continue;
        uint64_t r = L[i].a0;  // old
        if (((row_mask >> (i >> 4)) & 1) && ((bank_mask >> ((i >> 2) & 3)) & 1)) {
          const int s = dpp_source(ctrl, i);
          if (on(s)) r = L[s].a1;
          else if (bound) r = 0;
        }
        L[i].result = r;
      }
      break;
    }
    case kReadlane: {
      const int s = L[first].c0;
      if (s < 0 || s > 63) die("readlane: lane out of range");
      // an inactive source lane's register is read as it is on the hardware; here the value is not known
      if (!on(s)) die("readlane from a lane that is not active");
      for (int i = 0; i < 64; ++i) if (on(i)) L[i].result = L[s].a0;
      break;
    }
    case kReadfirst:
      for (int i = 0; i < 64; ++i) if (on(i)) L[i].result = L[first].a0;
      break;
    case kShfl: {
      const int width = L[first].c0, mode = L[first].c1;
      if (width <= 0 || width > 64 || (width & (width - 1))) die("shuffle width");
      for (int i = 0; i < 64; ++i) {
        if (!on(i)) continue;
        const int a = (int)(int64_t)L[i].a1, seg = i & ~(width - 1), r = i & (width - 1);
        int s;
        switch (mode) {
          case 0: s = seg + (a & (width - 1)); break;
          case 1: s = (r ^ a) < width ? seg + (r ^ a) : i; break;
          case 2: s = r - a >= 0 ? seg + r - a : i; break;
          default: s = r + a < width ? seg + r + a : i; break;
        }
        L[i].result = on(s) ? L[s].a0 : 0ull;  // ds_bpermute: a lane that is switched off contributes zero
      }
      break;
    }
    case kWaveBarrier: break;
    default: die("unknown cross-lane operation");
  }
  for (int i = 0; i < 64; ++i) if (on(i)) L[i].state = kRunnable;
}

// PCC_EMU_SHUFFLE=<seed>: the waves of a workgroup take their turns in a different pseudo-random order in every sweep, and
// so do the lanes of a wave between two meeting points -- nothing a correct kernel can notice (the hardware promises no
// order between waves, and between meeting points the lanes' effects are simultaneous), but a missing barrier or a
// forgotten lockstep marker shows up as a parity failure instead of passing by the luck of "wave 0 first, lane 0 first".
uint64_t shuffle_seed() {
  static const uint64_t s = [] { const char* e = getenv("PCC_EMU_SHUFFLE"); return e ? (uint64_t)strtoull(e, nullptr, 0) * 2u + 1u : 0ull; }();
  return s;
}
inline uint32_t next_rand(uint64_t& st) {
  st = st * 6364136223846793005ull + 1442695040888963407ull;
  return (uint32_t)(st >> 33);
}

void ask_for_a_thread();

void run_workgroup(Worker* w) {
  const int T = w->nthreads, nw = (T + 63) / 64;
  int sleepy_sweeps = 0;
  const bool racing = race::on();   // the happens-before checker (race.cpp) wants to know which lane runs and how many barriers it has passed
  race::Where& where = race::tl_where;
  for (int i = 0; i < T; ++i) prepare_lane(w, i);
  uint64_t rnd = shuffle_seed() ? shuffle_seed() ^ ((uint64_t)w->block_linear * 0x9E3779B97F4A7C15ull) : 0ull;
  int wave_order[kMaxThreads / 64], lane_order[64];
  for (int i = 0; i < nw; ++i) wave_order[i] = i;
  for (int i = 0; i < 64; ++i) lane_order[i] = i;
  for (;;) {
    bool slept = false;
    int n_done = 0, n_barrier = 0;
    if (rnd) for (int i = nw - 1; i > 0; --i) std::swap(wave_order[i], wave_order[next_rand(rnd) % (uint32_t)(i + 1)]);
    for (int wi = 0; wi < nw; ++wi) {
      const int wv = wave_order[wi];
      const int f = wv * 64, cnt = std::min(64, T - f);
      for (;;) {
        if (rnd) for (int i = 63; i > 0; --i) std::swap(lane_order[i], lane_order[next_rand(rnd) % (uint32_t)(i + 1)]);
        for (int li = 0; li < 64; ++li) {
          const int i = lane_order[li];
          if (i >= cnt) continue;
          Lane& l = w->lanes[f + i];
          if (l.state != kRunnable) continue;
          w->cur = f + i;
          if (racing) where.lane = f + i;
          threadIdx.x = (uint32_t)(f + i) % blockDim.x;
          threadIdx.y = ((uint32_t)(f + i) / blockDim.x) % blockDim.y;
          threadIdx.z = (uint32_t)(f + i) / (blockDim.x * blockDim.y);
          emu_switch(&w->sched_sp, l.sp);
        }
        w->cur = -1;
        if (racing) where.lane = -1;
        uint64_t coll = 0;
        bool sleepers = false;
        for (int i = 0; i < cnt; ++i) {
          const int s = w->lanes[f + i].state;
          if (s == kCollective) coll |= 1ull << i;
          else if (s == kSleep) sleepers = true;
        }
        if (sleepers) {  // come back to this wave after the others (and other workgroups) had their turn
          for (int i = 0; i < cnt; ++i) if (w->lanes[f + i].state == kSleep) w->lanes[f + i].state = kRunnable;
          slept = true;
          break;
        }
        if (!coll) break;
        run_collective(w, f, coll);
      }
    }
    for (int i = 0; i < T; ++i) {
      n_done += w->lanes[i].state == kDone;
      n_barrier += w->lanes[i].state == kBarrier;
    }
    if (n_done == T) return;
    if (n_done + n_barrier == T) {
      int kind = 0, all = 1, any = 0, count = 0;
      for (int wv = 0; wv < nw; ++wv) {  // (a wave executes s_barrier once, whatever its lanes do: its lanes must wait at ONE __syncthreads())
        int site = 0;
        for (int i = wv * 64; i < std::min(T, wv * 64 + 64); ++i) {
          if (w->lanes[i].state != kBarrier) continue;
          if (!site) site = w->lanes[i].site;
          else if (site != w->lanes[i].site) die("the lanes of one wave wait at two different __syncthreads()");
        }
      }
      for (int i = 0; i < T; ++i) {
        Lane& l = w->lanes[i];
        if (l.state != kBarrier) continue;
        if (kind == 0) kind = l.kind;
        else if (kind != l.kind) die("the lanes of a workgroup wait at different kinds of barriers");
        all &= l.a0 != 0; any |= l.a0 != 0; count += l.a0 != 0;
      }
      for (int i = 0; i < T; ++i) {
        Lane& l = w->lanes[i];
        if (l.state != kBarrier) continue;
        l.result = kind == kSyncAnd ? all : kind == kSyncOr ? any : kind == kSyncCount ? count : 0;
        l.state = kRunnable;
      }
      if (racing) ++where.bepoch;
      continue;
    }
    if (slept) {
      if (++sleepy_sweeps % 16 == 0) ask_for_a_thread();
      sched_yield();
      continue;
    }
    die("deadlock inside a workgroup: lanes wait for a cross-lane operation that the others cannot reach");
  }
}

// ------------------------------------------------------------------ pool
// Workgroups are handed out in blockIdx order.  A workgroup that keeps sleeping (it polls for something another
// workgroup has to produce) while workgroups of the launch are still waiting for a thread asks for one more thread:
// what the hardware gives a kernel whose grid fits the chip -- every workgroup resident -- the pool gives up to
// PCC_EMU_MAX_THREADS (default 1100) workgroups; beyond that a waiting workgroup runs into its own bound, as it
// would on a GPU that is busy with other kernels.
struct Pool {
  std::mutex launch_mu;   // one launch at a time (host threads of the pipeline share the pool)
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  // the launch in flight
  Launcher* job = nullptr;
  const char* kname = "";
  dim3 grid, block;
  uint32_t total = 0;
  std::atomic<uint32_t> next{0};
  uint32_t blocks_done = 0;
  int busy = 0;           // workers inside the launch
  uint64_t generation = 0;
  size_t max_threads = 1100;

  int parked = 0, tokens = 0;  // helper threads: asleep until a sleeping workgroup asks for one
  std::condition_variable cv_helper;

  void worker_main(bool helper) {
    Worker* w = new Worker();
    size_t stack_lanes = 0;  // the lanes' stacks: address space for as many as the largest workgroup this thread has run
    auto stacks_for = [&](size_t lanes) {
      if (lanes <= stack_lanes) return;
      if (w->stacks) {
        if (__asan_unpoison_memory_region) __asan_unpoison_memory_region(w->stacks, kStackBytes * stack_lanes);
        munmap(w->stacks, kStackBytes * stack_lanes);
      }
      w->stacks = (char*)mmap(nullptr, kStackBytes * lanes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (w->stacks == MAP_FAILED) { perror("wave64 executor: mmap of the lanes' stacks"); abort(); }
      stack_lanes = lanes;
    };
    tl_worker = w;
    uint64_t seen = 0;
    bool first = true;
    for (;;) {
      dim3 g, b;
      {
        std::unique_lock<std::mutex> lk(mu);
        if (helper) {
          // a helper joins the launch that is in flight when it is asked for (the first time: the one it was made for)
          if (!first) {
            ++parked;
            cv_helper.wait(lk, [&] { return tokens > 0; });
            --tokens; --parked;
          }
          first = false;
          if (!job) continue;  // that launch is over already
        } else {
          cv_work.wait(lk, [&] { return generation != seen; });
          seen = generation;
        }
        ++busy;
        w->job = job; w->kname = kname;
        g = grid; b = block;
      }
      uint32_t mine = 0;
      for (;;) {
        const uint32_t bi = next.fetch_add(1u, std::memory_order_relaxed);
        if (bi >= total) break;
        w->block_linear = bi;
        w->nthreads = (int)(b.x * b.y * b.z);
        stacks_for((size_t)w->nthreads);
        blockDim = {b.x, b.y, b.z};
        gridDim = {g.x, g.y, g.z};
        blockIdx.x = bi % g.x; blockIdx.y = (bi / g.x) % g.y; blockIdx.z = bi / (g.x * g.y);
        if (race::on()) {
          race::Where& where = race::tl_where;
          where.wg = bi; where.lane = -1; where.bepoch = 0; where.kname = w->kname;
          if (!where.own_lo) {   // the four index variables lie side by side in this file's thread-local block
            char* v[4] = {(char*)&threadIdx, (char*)&blockIdx, (char*)&blockDim, (char*)&gridDim};
            where.own_lo = *std::min_element(v, v + 4);
            where.own_hi = *std::max_element(v, v + 4) + sizeof(emu::Idx3);
          }
          race::workgroup_begin();
          run_workgroup(w);
          race::workgroup_end();
        } else {
          run_workgroup(w);
        }
        ++mine;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        blocks_done += mine;
        --busy;
      }
      cv_done.notify_all();
    }
  }
  void start() {
    const char* e = getenv("PCC_EMU_WORKERS");
    int n = e ? atoi(e) : (int)std::min<long>(8, sysconf(_SC_NPROCESSORS_ONLN));
    n = std::max(2, n);  // workgroup 0 of k_boxes_events waits for the others: they need a thread to run on
    if (const char* m = getenv("PCC_EMU_MAX_THREADS")) max_threads = (size_t)std::max(atoi(m), n);
    std::lock_guard<std::mutex> lk(mu);
    for (int i = 0; i < n; ++i) threads.emplace_back([this] { worker_main(false); });
  }
  // called by a workgroup that has been sleeping for a while: one more thread for the launch in flight (a parked helper,
  // or a new one) while workgroups are still waiting to be started
  void more_threads_please() {
    std::lock_guard<std::mutex> lk(mu);
    if (!job || next.load(std::memory_order_relaxed) >= total) return;
    if (parked > tokens) { ++tokens; cv_helper.notify_one(); }
    else if (threads.size() < max_threads) threads.emplace_back([this] { worker_main(true); });
  }
  void run(const char* name, dim3 g, dim3 b, Launcher& l) {
    std::lock_guard<std::mutex> one(launch_mu);
    if (threads.empty()) start();
    const uint64_t nt = (uint64_t)b.x * b.y * b.z;
    if (nt == 0 || nt > (uint64_t)kMaxThreads) { fprintf(stderr, "wave64 executor: %s: %llu threads per workgroup\n", name, (unsigned long long)nt); abort(); }
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return busy == 0; });  // (a worker that woke up late for the launch before)
    if (race::on()) race::launch_begin(name, g.x * g.y * g.z);
    job = &l; kname = name; grid = g; block = b;
    total = g.x * g.y * g.z;
    next.store(0);
    blocks_done = 0;
    ++generation;
    cv_work.notify_all();
    cv_done.wait(lk, [&] { return blocks_done == total && busy == 0; });
    job = nullptr;
    tokens = 0;
    if (race::on()) race::launch_end();
  }
};
Pool& pool() {
  // never destroyed: the workers live as long as the process.  (call_once rather than a function-local static: the `tsan`
  // build runs this uninstrumented file under a ThreadSanitizer that sees pthread_once but not an inline guard check)
  static Pool* p = nullptr;
  static std::once_flag once;
  std::call_once(once, [] { p = new Pool(); });
  return *p;
}
void ask_for_a_thread() { pool().more_threads_please(); }

}  // namespace

uint64_t ballot(int pred, int site) { return yield_lane(site, kCollective, kBallot, (uint64_t)(pred != 0), 0); }
int update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl, int site) {
  return (int)(uint32_t)yield_lane(site, kCollective, kDpp, (uint32_t)old, (uint32_t)src, ctrl, row_mask, bank_mask, bound_ctrl ? 1 : 0);
}
uint32_t readlane(uint32_t v, int lane, int site) { return (uint32_t)yield_lane(site, kCollective, kReadlane, v, 0, lane); }
uint32_t readfirstlane(uint32_t v, int site) { return (uint32_t)yield_lane(site, kCollective, kReadfirst, v, 0); }
uint64_t shfl64(uint64_t v, int a, int width, int mode, int site) { return yield_lane(site, kCollective, kShfl, v, (uint64_t)(int64_t)a, width, mode); }
void wave_barrier(int site) { (void)yield_lane(site, kCollective, kWaveBarrier, 0, 0); }
void syncthreads(int site) { (void)yield_lane(site, kBarrier, kSync, 1, 0); }
int syncthreads_and(int pred, int site) { return (int)yield_lane(site, kBarrier, kSyncAnd, pred != 0, 0); }
int syncthreads_or(int pred, int site) { return (int)yield_lane(site, kBarrier, kSyncOr, pred != 0, 0); }
int syncthreads_count(int pred, int site) { return (int)yield_lane(site, kBarrier, kSyncCount, pred != 0, 0); }
void sleep(int) { (void)yield_lane(0, kSleep, 0, 0, 0); }
// HW_REG_XCC_ID: workgroup b on XCD b mod 8, as observed on the chip.  PCC_EMU_XCC=<n>: every workgroup claims XCD n;
// PCC_EMU_XCC=rand: a pseudo-random one -- the kernels may use the register as a hint only, and must give the same bytes
int getreg(int imm) {
  if ((imm & 63) != 20 || !tl_worker) return 0;
  static const int mode = [] { const char* e = getenv("PCC_EMU_XCC"); return !e ? -1 : (!strcmp(e, "rand") ? -2 : atoi(e) & 7); }();
  if (mode >= 0) return mode;
  if (mode == -2) return (int)((tl_worker->block_linear * 2654435761u) >> 29);
  return (int)(tl_worker->block_linear & 7u);
}
unsigned long long wall_clock() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (unsigned long long)ts.tv_sec * 1000000000ull + (unsigned long long)ts.tv_nsec;  // "100 MHz" clock: see hipDeviceGetAttribute
}
void launch(void*, const char* name, dim3 grid, dim3 block, Launcher& l) {
  if ((uint64_t)grid.x * grid.y * grid.z == 0) return;
  pool().run(name, grid, block, l);
}

}  // namespace emu

// -------------------------------------------------------------------- host API
struct emuStream { int id; };
struct emuEvent { unsigned long long t; };

const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : e == hipErrorNotReady ? "hipErrorNotReady" : "hipError (wave64 executor)"; }
hipError_t hipGetLastError() { return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) {
  const char* e = getenv("PCC_EMU_DEVICES");
  *n = e ? atoi(e) : 1;
  return hipSuccess;
}
static thread_local int tl_device = 0;
hipError_t hipGetDevice(int* d) { *d = tl_device; return hipSuccess; }
hipError_t hipSetDevice(int d) {
  int n = 0;
  hipGetDeviceCount(&n);
  if (d < 0 || d >= n) return hipErrorInvalidDevice;
  tl_device = d;
  return hipSuccess;
}
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "wave64 executor (CPU)");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "emu");
  p->totalGlobalMem = (size_t)16 << 30;
  p->multiProcessorCount = 256;
  p->maxSharedMemoryPerMultiProcessor = 160 * 1024;
  p->sharedMemPerBlock = 64 * 1024;
  p->regsPerMultiprocessor = 128 * 1024;
  p->maxThreadsPerMultiProcessor = 2048;
  p->warpSize = 64;
  p->clockRate = 2400000;
  return hipSuccess;
}
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  if (a == hipDeviceAttributeWallClockRate) { *v = 1000000; return hipSuccess; }  // kHz: wall_clock() counts nanoseconds
  if (a == hipDeviceAttributeMultiprocessorCount) { *v = 256; return hipSuccess; }
  return hipErrorInvalidValue;
}
hipError_t hipDeviceGetPCIBusId(char* out, int len, int d) {
  int n = 0;
  hipGetDeviceCount(&n);
  if (!out || len < 13 || d < 0 || d >= n) return hipErrorInvalidValue;
  snprintf(out, (size_t)len, "0000:%02X:00.0", d + 1);   // (upper-case, as the runtime prints it)
  return hipSuccess;
}
// The `race` build's access callbacks feed the happens-before checker of race.cpp when PCC_EMU_RACE=1 (the return address is the access)
static inline void emu_touch(uintptr_t a, size_t n, bool store, void* pc) {
  if (emu::race::on()) emu::race::access(a, n, store, pc);
}
// The `race` build is instrumented by gcc's -fsanitize=thread pass (the address sanitizer's outline pass leaves direct
// accesses to thread-local variables alone -- every scalar __shared__ variable -- and this one does not): these are its
// callbacks.  libtsan is NOT linked: the checker is race.cpp.  The atomics below are the ones inside the
// hooks of hip_runtime.h (already booked by the hook; a bare one from a running lane is booked here as agent scope).
#ifndef PCC_EMU_REAL_TSAN   // (the `tsan` build of the Makefile links the real ThreadSanitizer for the HOST sources: its symbols, not these)
#define PCC_TSAN_CB(N) \
  extern "C" void __tsan_read##N(void* a) { emu_touch((uintptr_t)a, N, false, __builtin_return_address(0)); } \
  extern "C" void __tsan_write##N(void* a) { emu_touch((uintptr_t)a, N, true, __builtin_return_address(0)); } \
  extern "C" void __tsan_unaligned_read##N(void* a) { emu_touch((uintptr_t)a, N, false, __builtin_return_address(0)); } \
  extern "C" void __tsan_unaligned_write##N(void* a) { emu_touch((uintptr_t)a, N, true, __builtin_return_address(0)); }
PCC_TSAN_CB(1) PCC_TSAN_CB(2) PCC_TSAN_CB(4) PCC_TSAN_CB(8) PCC_TSAN_CB(16)
extern "C" void __tsan_read_range(void* a, size_t n) { emu_touch((uintptr_t)a, n, false, __builtin_return_address(0)); }
extern "C" void __tsan_write_range(void* a, size_t n) { emu_touch((uintptr_t)a, n, true, __builtin_return_address(0)); }
extern "C" void __tsan_init() {}
extern "C" void __tsan_func_entry(void*) {}
extern "C" void __tsan_func_exit() {}
extern "C" void __tsan_vptr_update(void**, void*) {}
extern "C" void __tsan_vptr_read(void**) {}
namespace {
struct BareAtomic {   // an atomic builtin that did not come through a hook of hip_runtime.h
  BareAtomic(const volatile void* p, size_t n, int order, int kind, void* pc) { emu::race::atomic_begin_bare((const void*)p, n, order, kind, pc); }
  ~BareAtomic() { emu::race::atomic_end_bare(); }
};
}
#define PCC_TSAN_ATOMIC(BITS, T) \
  extern "C" T __tsan_atomic##BITS##_load(const volatile T* p, int mo) { BareAtomic g(p, sizeof(T), mo, 0, __builtin_return_address(0)); return __atomic_load_n(p, __ATOMIC_SEQ_CST); } \
  extern "C" void __tsan_atomic##BITS##_store(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 1, __builtin_return_address(0)); __atomic_store_n(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_exchange(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_add(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_sub(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_and(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_or(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_xor(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_xor(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_fetch_nand(volatile T* p, T v, int mo) { BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_fetch_nand(p, v, __ATOMIC_SEQ_CST); } \
  extern "C" int __tsan_atomic##BITS##_compare_exchange_strong(volatile T* p, T* expect, T v, int mo, int) { \
    BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_compare_exchange_n(p, expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); } \
  extern "C" int __tsan_atomic##BITS##_compare_exchange_weak(volatile T* p, T* expect, T v, int mo, int) { \
    BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); return __atomic_compare_exchange_n(p, expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); } \
  extern "C" T __tsan_atomic##BITS##_compare_exchange_val(volatile T* p, T expect, T v, int mo, int) { \
    BareAtomic g(p, sizeof(T), mo, 2, __builtin_return_address(0)); __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }
PCC_TSAN_ATOMIC(8, uint8_t) PCC_TSAN_ATOMIC(16, uint16_t) PCC_TSAN_ATOMIC(32, uint32_t) PCC_TSAN_ATOMIC(64, uint64_t)
extern "C" void __tsan_atomic_thread_fence(int mo) { if (emu::race::on()) emu::race::fence_bare(mo); __atomic_thread_fence(__ATOMIC_SEQ_CST); }
extern "C" void __tsan_atomic_signal_fence(int) {}
#endif  // PCC_EMU_REAL_TSAN

hipError_t hipMalloc(void** p, size_t bytes) {
  if (void* t = emu::race::arena_alloc(bytes ? bytes : 256)) { memset(t, 0xA5, bytes); *p = t; return hipSuccess; }
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
  // fresh device memory holds whatever was there before: poison it so that a kernel that relies on zeroes is caught
  memset(q, 0xA5, bytes);
  *p = q;
  return hipSuccess;
}
hipError_t hipFree(void* p) { if (!emu::race::arena_owns(p)) free(p); return hipSuccess; }  // (the checker's arena is never reused)
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) {
  if (void* t = emu::race::arena_alloc(bytes ? bytes : 256)) { *p = t; return hipSuccess; }  // pinned memory the kernels write has a shadow too
  void* q = nullptr;
  if (posix_memalign(&q, 256, bytes ? bytes : 256) != 0) return hipErrorOutOfMemory;
  *p = q;
  return hipSuccess;
}
hipError_t hipHostFree(void* p) { if (!emu::race::arena_owns(p)) free(p); return hipSuccess; }
hipError_t hipHostRegister(void*, size_t, unsigned) { return hipSuccess; }
hipError_t hipHostUnregister(void*) { return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  a->type = hipMemoryTypeUnregistered; a->device = 0; a->devicePointer = const_cast<void*>(p); a->hostPointer = const_cast<void*>(p);
  return hipErrorInvalidValue;  // "not known to the runtime", what real HIP says of plain host memory
}
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) { memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t) { memmove(dst, src, bytes); return hipSuccess; }
hipError_t hipMemsetAsync(void* dst, int v, size_t bytes, hipStream_t) { memset(dst, v, bytes); return hipSuccess; }
hipError_t hipMemset(void* dst, int v, size_t bytes) { memset(dst, v, bytes); return hipSuccess; }
hipError_t hipMemcpyFromSymbol(void* dst, const void* sym, size_t bytes, size_t off, hipMemcpyKind) {
  memcpy(dst, (const char*)sym + off, bytes);
  return hipSuccess;
}
hipError_t hipStreamCreate(hipStream_t* s) { *s = new emuStream{0}; return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new emuStream{0}; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new emuEvent{0}; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new emuEvent{0}; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = emu::wall_clock(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)((double)(b->t - a->t) * 1e-6); return hipSuccess; }
