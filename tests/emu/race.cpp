// Happens-before checker ("race detector") of the wave64 executor: TEST INFRASTRUCTURE ONLY.
//
// The executor runs workgroups as OS threads on one coherent CPU: every store is visible to every other workgroup at
// once.  An MI355X is eight XCDs with an L2 each: a plain store of a workgroup sits in its XCD's L2 (or on its way there)
// until something writes it back, and a plain load of a workgroup on another XCD may be served from a stale line of ITS
// L2 -- inside one launch only agent-scope atomics (which go to memory) and what an agent-scope release -> acquire chain
// orders cross from one workgroup to another.  Code that hands data from workgroup to workgroup through plain loads and
// stores is therefore "executor-green" and wrong on the chip.  This checker finds exactly that class, without modelling
// the caches: in the `race` build of the library (tests/emu/Makefile) every load and store of the kernel sources calls
// back (gcc's outline address-sanitizer instrumentation, the callbacks are ours), every atomic and fence goes through
// the hooks of hip_runtime.h, and the checker reports
//
//   GLOBAL memory (the arena hipMalloc / hipHostMalloc hand out in this mode)
//     two accesses of the same bytes in ONE launch by two different workgroups, at least one of them a write, not both
//     agent-scope atomics, and not ordered by a release -> acquire chain of agent-scope atomics / fences
//     (vector clocks: a release publishes the workgroup's clock on the atomic's address, an acquire that reads it joins;
//     relaxed atomics carry the clock of the last release fence / feed the next acquire fence; workgroup-scope atomics
//     and fences order nothing between workgroups).  Launches of a stream are ordered by the hardware (the L2s are
//     written back and invalidated between kernels): records of earlier launches never conflict.
//   GLOBAL and LDS memory, inside one workgroup
//     two accesses of the same bytes by two different WAVES of the workgroup, at least one a write, not both atomics, with
//     no workgroup barrier between them -- the missing __syncthreads(), which the executor's fixed wave order hides.
//     (Lanes of one wave are in program order: never reported.)
//   LDS, uninitialised
//     a read of LDS bytes that no lane of the workgroup has written since the workgroup started.  On the chip LDS holds
//     whatever the workgroup before left there; here it holds what the OS thread's previous workgroup left (or zeroes the
//     first time) -- a kernel that relies on either is wrong on both, and only by luck visibly so.
//
// Any pattern a per-XCD store buffer would break (a plain store read by another XCD's workgroup before the launch ends)
// is by definition a report of the first kind, so "zero reports" covers what a visibility model would show as wrong
// bytes, and names the two source lines instead.
//
// Per 4-byte granule the shadow keeps the last writer, the last writer before it that was somebody else, and two readers
// (with the bytes each touched); granules whose bytes have different owners are kept per byte.  Two of each is the one
// approximation: an access that races only with a third, older writer or reader of a granule is missed.  Never a false
// report: a record is only ever replaced, not merged.
#include "race.h"

#include <dlfcn.h>
#include <link.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <unistd.h>

#include <array>
#include <atomic>
#include <map>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

namespace emu {
namespace race {

thread_local Where tl_where;

namespace {

constexpr size_t kArena = (size_t)32 << 30;
constexpr int kPageShift = 14;               // granules per shadow page
constexpr size_t kStripes = 4096;
constexpr int kScopeWorkgroup = 3, kScopeAgent = 4;

// who: (workgroup + 1) << 5 | atomic << 4 | bytes of the granule;  bw: barrier epoch << 5 | wave
struct Rec { uint32_t who, epoch, bw, pc; };
struct Cell { uint32_t stamp, split; Rec w[2], r[2]; };  // w[0]: the last writer; w[1]: the last one before it that was somebody else
inline uint32_t wg1(const Rec& r) { return r.who >> 5; }
inline uint32_t bytes(const Rec& r) { return r.who & 15u; }
inline bool atomic_(const Rec& r) { return (r.who >> 4) & 1u; }

using VC = std::map<uint32_t, uint32_t>;  // workgroup -> epoch (tiny: the kernels' hand-offs are self-describing words)
void join(VC& into, const VC& from) {
  for (auto& kv : from) { uint32_t& e = into[kv.first]; if (kv.second > e) e = kv.second; }
}

struct WgState {
  uint32_t wg = 0, clock = 1;
  bool released = false;     // a release has published `clock`: the next plain access starts a new epoch
  VC vc;                     // what this workgroup has acquired
  bool has_rel_fence = false;
  VC rel_fence, acq_pending;
  bool suppress = false;     // inside an atomic operation that atomic_begin has booked
  bool nested = false, in_fence = false;
  uintptr_t at_addr = 0; int at_order = 0, at_kind = 0; bool at_agent = false, at_global = false;
  std::unordered_map<uint64_t, std::array<Cell, 4>> lds_split;
};

struct Acc { uint32_t wg, epoch, bw, pc, mask; bool write, atomic, check_init; };

struct Report {
  std::string kernel;
  bool lds;
  bool uninit;
  uint32_t pc_now, pc_then;
  bool now_write, now_atomic, then_write, then_atomic;
  uint32_t wg_now, wave_now, wg_then, wave_then;
  uint64_t offset, alloc_index, alloc_off, alloc_size, count;
};

struct Stripe {
  std::atomic_flag lock = ATOMIC_FLAG_INIT;
  std::unordered_map<uint64_t, std::array<Cell, 4>> split;
  std::unordered_map<uintptr_t, VC> sync;   // the clock a release left on an atomic's address
};

struct Global {
  bool on = false;
  char* lo = nullptr;
  char* hi = nullptr;
  std::atomic<size_t> bump{0};
  std::atomic<Cell*>* pages = nullptr;
  size_t npages = 0;
  uint32_t launch = 0;
  std::string kernel;
  uintptr_t lib_base = 0;
  size_t tls_size = 0;
  Stripe* stripes = nullptr;
  std::mutex alloc_mu;
  std::vector<std::pair<size_t, size_t>> allocs;
  std::mutex rep_mu;
  std::map<std::array<uint64_t, 3>, Report> reports;
  bool abort_on_report = false;
  std::atomic<uint64_t> seen[4];   // accesses checked: global plain, global atomic, LDS plain, LDS atomic

  void init() {
    const char* e = getenv("PCC_EMU_RACE");
    on = e && e[0] == '1';
    if (!on) return;
    abort_on_report = getenv("PCC_EMU_RACE_ABORT") != nullptr;
    for (auto& c : seen) c.store(0);
    lo = (char*)mmap(nullptr, kArena, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (lo == MAP_FAILED) { perror("race checker: arena"); abort(); }
    hi = lo + kArena;
    npages = (kArena >> 2) >> kPageShift;
    pages = new std::atomic<Cell*>[npages];
    for (size_t i = 0; i < npages; ++i) pages[i].store(nullptr, std::memory_order_relaxed);
    stripes = new Stripe[kStripes];
    Dl_info di;
    if (dladdr((void*)&emu::race::on, &di)) lib_base = (uintptr_t)di.dli_fbase;
  }
  Cell* page(size_t pi) {
    Cell* p = pages[pi].load(std::memory_order_acquire);
    if (p) return p;
    const size_t bytes = sizeof(Cell) << kPageShift;
    Cell* fresh = (Cell*)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (fresh == MAP_FAILED) { perror("race checker: shadow page"); abort(); }
    Cell* expect = nullptr;
    if (pages[pi].compare_exchange_strong(expect, fresh, std::memory_order_acq_rel)) return fresh;
    munmap(fresh, bytes);
    return expect;
  }
};
Global& g() {
  static Global* x = [] { Global* y = new Global(); y->init(); return y; }();
  return *x;
}

struct Guard {
  std::atomic_flag& f;
  explicit Guard(std::atomic_flag& x) : f(x) { while (f.test_and_set(std::memory_order_acquire)) {} }
  ~Guard() { f.clear(std::memory_order_release); }
};

inline Rec rec_of(const Acc& a, uint32_t mask) { return Rec{((a.wg + 1u) << 5) | (a.atomic ? 16u : 0u) | (mask & 15u), a.epoch, a.bw, a.pc}; }
inline bool same_actor_and_time(const Rec& r, const Acc& a) {
  return wg1(r) == a.wg + 1u && r.bw == a.bw && r.epoch == a.epoch && atomic_(r) == a.atomic;
}

// does the recorded access happen before the access `a` of the running lane?
bool ordered(const Rec& r, const Acc& a, const WgState* st) {
  if (wg1(r) == a.wg + 1u) {
    if ((r.bw & 31u) == (a.bw & 31u)) return true;   // the same wave: program order
    return (r.bw >> 5) != (a.bw >> 5);               // another wave of the workgroup: a barrier in between?
  }
  auto it = st->vc.find(wg1(r) - 1u);
  return it != st->vc.end() && it->second >= r.epoch;
}

void report(const Rec& then, bool then_write, const Acc& now, bool lds, uint64_t offset, bool uninit = false) {
  Global& G = g();
  std::lock_guard<std::mutex> lk(G.rep_mu);
  const std::array<uint64_t, 3> key{{(uint64_t)now.pc << 32 | then.pc, (uint64_t)lds | (uninit ? 2u : 0u), std::hash<std::string>()(G.kernel)}};
  auto it = G.reports.find(key);
  if (it != G.reports.end()) { ++it->second.count; return; }
  if (G.reports.size() >= 4096) return;
  Report r{};
  r.kernel = G.kernel; r.lds = lds; r.uninit = uninit; r.pc_now = now.pc; r.pc_then = then.pc;
  r.now_write = now.write; r.now_atomic = now.atomic; r.then_write = then_write; r.then_atomic = atomic_(then);
  r.wg_now = now.wg; r.wave_now = now.bw & 31u; r.wg_then = wg1(then) - 1u; r.wave_then = then.bw & 31u;
  r.offset = offset; r.count = 1;
  if (!lds) {
    std::lock_guard<std::mutex> al(G.alloc_mu);
    for (size_t i = 0; i < G.allocs.size(); ++i)
      if (offset >= G.allocs[i].first && offset < G.allocs[i].first + G.allocs[i].second) { r.alloc_index = i; r.alloc_off = offset - G.allocs[i].first; r.alloc_size = G.allocs[i].second; }
  }
  G.reports.emplace(key, r);
  if (G.abort_on_report) {
    fprintf(stderr, "race checker: %s %s: pc %x vs %x (workgroups %u, %u)\n", G.kernel.c_str(), lds ? "LDS" : "global", now.pc, then.pc, r.wg_now, r.wg_then);
    abort();
  }
}

// one cell (a granule, or one byte of a split granule): conflicts of `a` with what is recorded, then record `a`
void one_cell(Cell& c, const Acc& a, uint32_t record_mask, const WgState* st, bool lds, uint64_t offset) {
  for (const Rec& w : c.w)
    if (wg1(w) && (bytes(w) & a.mask) && !(atomic_(w) && a.atomic) && !ordered(w, a, st)) report(w, true, a, lds, offset);
  // LDS: bytes nobody in this workgroup has written yet (w[1]'s bytes are among w[0]'s; an atomic read-modify-write of
  // such bytes reads them too)
  if (lds && a.check_init && (a.mask & ~bytes(c.w[0])) != 0u) report(Rec{}, false, a, true, offset, true);
  if (a.write) {
    for (Rec& r : c.r)
      if (wg1(r) && (bytes(r) & a.mask) && !(atomic_(r) && a.atomic) && !ordered(r, a, st)) report(r, false, a, lds, offset);
    // (the bytes of w[1] are always among those of w[0], which this write covers -- granule() sees to that: nothing is lost)
    if (wg1(c.w[0]) && !(wg1(c.w[0]) == a.wg + 1u && (c.w[0].bw & 31u) == (a.bw & 31u))) c.w[1] = c.w[0];
    c.w[0] = rec_of(a, record_mask);
    for (Rec& r : c.r)   // a reader this write comes after, all of whose bytes are overwritten: the write stands for it from now on
      if (wg1(r) && (bytes(r) & ~a.mask) == 0u && ordered(r, a, st)) r = Rec{};
    return;
  }
  for (Rec& r : c.r)
    if (wg1(r) && same_actor_and_time(r, a)) { r.who |= a.mask & 15u; return; }
  for (Rec& r : c.r)
    if (!wg1(r)) { r = rec_of(a, a.mask); return; }
  for (Rec& r : c.r)   // an older read that happens before this one and covers no other bytes: whatever races with it races with this one
    if ((bytes(r) & ~a.mask) == 0u && ordered(r, a, st)) { r = rec_of(a, a.mask); return; }
  c.r[1] = rec_of(a, a.mask);   // (the approximation: two readers per granule)
}

template <typename SplitMap>
void granule(Cell& c, uint32_t stamp, const Acc& a, SplitMap& sm, uint64_t key, const WgState* st, bool lds, uint64_t offset) {
  if (c.stamp != stamp) { c = Cell{}; c.stamp = stamp; }
  uint32_t record_mask = a.mask;
  if (!c.split && a.write && wg1(c.w[0]) && (bytes(c.w[0]) & ~a.mask)) {
    // the recorded writer owns bytes this write does not touch: the same actor at the same time -> one record for both;
    // anybody else -> the granule is kept per byte from now on
    if (same_actor_and_time(c.w[0], a)) record_mask |= bytes(c.w[0]);
    else {
      std::array<Cell, 4>& sub = sm[key];
      for (int b = 0; b < 4; ++b) {
        sub[b] = Cell{};
        sub[b].stamp = stamp;
        for (int k = 0; k < 2; ++k) {
          if ((bytes(c.w[k]) >> b) & 1u) { sub[b].w[k] = c.w[k]; sub[b].w[k].who = (c.w[k].who & ~15u) | (1u << b); }
          if ((bytes(c.r[k]) >> b) & 1u) { sub[b].r[k] = c.r[k]; sub[b].r[k].who = (c.r[k].who & ~15u) | (1u << b); }
        }
        if (!wg1(sub[b].w[0])) { sub[b].w[0] = sub[b].w[1]; sub[b].w[1] = Rec{}; }
      }
      c.split = 1;
    }
  }
  if (c.split) {
    std::array<Cell, 4>& sub = sm[key];
    for (int b = 0; b < 4; ++b)
      if ((a.mask >> b) & 1u) {
        Acc ab = a;
        ab.mask = 1u << b;
        if (sub[b].stamp != stamp) { sub[b] = Cell{}; sub[b].stamp = stamp; }
        one_cell(sub[b], ab, ab.mask, st, lds, offset + (uint64_t)b);
      }
  } else {
    one_cell(c, a, record_mask, st, lds, offset);
  }
}

void touch(uintptr_t addr, size_t n, bool write, bool atomic, void* pc, bool reads = true) {
  Global& G = g();
  Where& w = tl_where;
  WgState* st = (WgState*)w.wgstate;
  const bool global = addr >= (uintptr_t)G.lo && addr < (uintptr_t)G.hi;
  const bool lds = !global && addr >= (uintptr_t)w.lds_lo && addr < (uintptr_t)w.lds_hi;
  if (!global && !lds) return;
  G.seen[(global ? 0 : 2) + (atomic ? 1 : 0)].fetch_add(1, std::memory_order_relaxed);

  if (global && !atomic && st->released) { ++st->clock; st->released = false; }
  Acc a{};
  a.wg = w.wg; a.epoch = st->clock; a.bw = (w.bepoch << 5) | ((uint32_t)w.lane >> 6); a.write = write; a.atomic = atomic;
  a.pc = (uint32_t)((uintptr_t)pc - G.lib_base);
  // reads (and the read half of a read-modify-write) of LDS want the bytes written before -- but not the executor's own
  // thread-local variables in the same block (threadIdx & co: written by the scheduler, which is not instrumented)
  a.check_init = lds && reads && !(addr >= (uintptr_t)w.own_lo && addr < (uintptr_t)w.own_hi);
  const uintptr_t base = global ? (uintptr_t)G.lo : (uintptr_t)w.lds_lo;
  const uint64_t first = (addr - base) >> 2, last = (addr + n - 1 - base) >> 2;
  for (uint64_t gi = first; gi <= last; ++gi) {
    const uintptr_t g_lo = base + (gi << 2);
    uint32_t mask = 0;
    for (int b = 0; b < 4; ++b)
      if (g_lo + b >= addr && g_lo + b < addr + n) mask |= 1u << b;
    a.mask = mask;
    if (global) {
      Cell* page = G.page(gi >> kPageShift);
      Stripe& s = G.stripes[(gi * 0x9E3779B97F4A7C15ull) >> 52];
      Guard lk(s.lock);
      granule(page[gi & ((1u << kPageShift) - 1u)], G.launch, a, s.split, gi, st, false, gi << 2);
    } else {
      granule(((Cell*)w.lds_shadow)[gi], w.lds_stamp, a, st->lds_split, gi, st, true, gi << 2);
    }
  }
}

VC snapshot(const WgState* st) {
  VC v = st->vc;
  v[st->wg] = st->clock;
  return v;
}
inline bool is_release(int order) { return order == __ATOMIC_RELEASE || order == __ATOMIC_ACQ_REL || order == __ATOMIC_SEQ_CST; }
inline bool is_acquire(int order) { return order == __ATOMIC_ACQUIRE || order == __ATOMIC_ACQ_REL || order == __ATOMIC_SEQ_CST || order == __ATOMIC_CONSUME; }

int tls_callback(dl_phdr_info* info, size_t, void* out) {
  Global& G = g();
  if ((uintptr_t)info->dlpi_addr != G.lib_base) return 0;
  for (int i = 0; i < info->dlpi_phnum; ++i)
    if (info->dlpi_phdr[i].p_type == PT_TLS) {
      char** o = (char**)out;
      o[0] = (char*)info->dlpi_tls_data;
      o[1] = o[0] ? o[0] + info->dlpi_phdr[i].p_memsz : nullptr;
    }
  return 1;
}

}  // namespace

bool on() { return g().on; }

void* arena_alloc(size_t bytes) {
  Global& G = g();
  if (!G.on) return nullptr;
  const size_t need = (bytes + 4095) & ~(size_t)4095;
  const size_t at = G.bump.fetch_add(need);
  if (at + need > kArena) { fprintf(stderr, "race checker: arena exhausted\n"); abort(); }
  std::lock_guard<std::mutex> lk(G.alloc_mu);
  G.allocs.emplace_back(at, bytes);
  return G.lo + at;
}
bool arena_owns(const void* p) { Global& G = g(); return G.on && (const char*)p >= G.lo && (const char*)p < G.hi; }

void launch_begin(const char* kernel, uint32_t) {
  Global& G = g();
  if (!G.on) return;
  if (++G.launch == 0) G.launch = 1;
  std::string n(kernel);
  while (!n.empty() && (n[0] == '(' || n[0] == ' ')) n.erase(0, 1);
  while (!n.empty() && (n.back() == ')' || n.back() == ' ')) n.pop_back();
  G.kernel = n;
  for (size_t i = 0; i < kStripes; ++i) {
    if (!G.stripes[i].split.empty()) G.stripes[i].split.clear();
    if (!G.stripes[i].sync.empty()) G.stripes[i].sync.clear();
  }
}
void launch_end() {}

void workgroup_begin() {
  Global& G = g();
  if (!G.on) return;
  Where& w = tl_where;
  WgState* st = new WgState();
  st->wg = w.wg;
  w.wgstate = st;
  if (!w.lds_lo) {   // (tl_where lives in the library's thread-local block: touching it above has made this thread's copy)
    char* range[2] = {nullptr, nullptr};
    dl_iterate_phdr(tls_callback, range);
    if (range[0]) {
      w.lds_lo = range[0]; w.lds_hi = range[1];
      const size_t cells = ((size_t)(range[1] - range[0]) + 3) / 4;
      w.lds_shadow = mmap(nullptr, cells * sizeof(Cell), PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (w.lds_shadow == MAP_FAILED) { perror("race checker: LDS shadow"); abort(); }
    }
  }
  if (++w.lds_stamp == 0) w.lds_stamp = 1;
}
void workgroup_end() {
  Where& w = tl_where;
  delete (WgState*)w.wgstate;
  w.wgstate = nullptr;
}

void access(uintptr_t addr, size_t n, bool store, void* pc) {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;   // host code (or the checker is off)
  if (((WgState*)w.wgstate)->suppress) return;
  touch(addr, n, store, false, pc, !store);
}

void atomic_begin(const void* p, size_t n, int order, int scope, int kind, void* pc) {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  Global& G = g();
  WgState* st = (WgState*)w.wgstate;
  const uintptr_t addr = (uintptr_t)p;
  const bool global = addr >= (uintptr_t)G.lo && addr < (uintptr_t)G.hi;
  // in global memory only agent scope (or wider) is atomic between workgroups; in LDS workgroup scope is all there is
  const bool agent = scope >= kScopeAgent;
  const bool counts_as_atomic = global ? agent : scope >= kScopeWorkgroup;
  touch(addr, n, kind != 0, counts_as_atomic, pc, kind != 1);   // (a plain atomic store reads nothing)
  st->suppress = true;
  st->at_addr = addr; st->at_order = order; st->at_kind = kind; st->at_agent = agent; st->at_global = global;
  if (!global || !agent || kind == 0) return;
  // the store half: what a reader of this value synchronises with
  Stripe& s = G.stripes[(addr * 0x9E3779B97F4A7C15ull) >> 52];
  Guard lk(s.lock);
  if (is_release(order)) {
    VC snap = snapshot(st);
    if (kind == 1) s.sync[addr] = snap; else join(s.sync[addr], snap);
    st->released = true;
  } else if (st->has_rel_fence) {
    if (kind == 1) s.sync[addr] = st->rel_fence; else join(s.sync[addr], st->rel_fence);
  } else if (kind == 1) {
    s.sync.erase(addr);   // a relaxed store by somebody else ends the release sequence
  }
}
void atomic_end() {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  WgState* st = (WgState*)w.wgstate;
  st->suppress = false;
  if (!st->at_global || !st->at_agent || st->at_kind == 1) return;
  Global& G = g();
  Stripe& s = G.stripes[(st->at_addr * 0x9E3779B97F4A7C15ull) >> 52];
  Guard lk(s.lock);
  auto it = s.sync.find(st->at_addr);
  if (it == s.sync.end()) return;
  if (is_acquire(st->at_order)) join(st->vc, it->second); else join(st->acq_pending, it->second);
}
void atomic_begin_bare(const void* p, size_t n, int order, int kind, void* pc) {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  WgState* st = (WgState*)w.wgstate;
  if (st->suppress) { st->nested = true; return; }   // the atomic inside a hook: the hook has booked it
  atomic_begin(p, n, order, kScopeAgent, kind, pc);
}
void atomic_end_bare() {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  WgState* st = (WgState*)w.wgstate;
  if (st->nested) { st->nested = false; return; }
  atomic_end();
}
void fence_bare(int order) {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  if (((WgState*)w.wgstate)->in_fence) return;   // the fence inside the hook of __builtin_amdgcn_fence
  fence(order, "agent");
}
void fence(int order, const char* scope) {
  Where& w = tl_where;
  if (w.lane < 0 || !w.wgstate) return;
  if (scope && (!strcmp(scope, "workgroup") || !strcmp(scope, "wavefront") || !strcmp(scope, "singlethread"))) return;  // orders nothing between workgroups
  WgState* st = (WgState*)w.wgstate;
  if (is_release(order)) { st->rel_fence = snapshot(st); st->has_rel_fence = true; st->released = true; }
  if (is_acquire(order)) join(st->vc, st->acq_pending);
}

}  // namespace race
}  // namespace emu

// what the checker has found since the last call: the number of distinct reports; one line each into `out`:
//   kernel space offset | now: W/R a/p pc workgroup wave | then: W/R a/p pc workgroup wave | count | allocation index offset size
extern "C" size_t pcc_emu_race_report(char* out, size_t cap) {
  using namespace emu::race;
  Global& G = g();
  std::lock_guard<std::mutex> lk(G.rep_mu);
  std::string s;
  for (auto& kv : G.reports) {
    const Report& r = kv.second;
    char line[512];
    snprintf(line, sizeof(line), "%s %s %llu | %c %c 0x%x %u %u | %c %c 0x%x %u %u | %llu | %llu %llu %llu\n", r.kernel.c_str(), r.uninit ? "lds-uninitialised" : (r.lds ? "lds" : "global"),
             (unsigned long long)r.offset, r.now_write ? 'W' : 'R', r.now_atomic ? 'a' : 'p', r.pc_now, r.wg_now, r.wave_now, r.then_write ? 'W' : 'R',
             r.then_atomic ? 'a' : 'p', r.pc_then, r.wg_then, r.wave_then, (unsigned long long)r.count, (unsigned long long)r.alloc_index,
             (unsigned long long)r.alloc_off, (unsigned long long)r.alloc_size);
    s += line;
  }
  const size_t n = G.reports.size();
  G.reports.clear();
  if (out && cap) { strncpy(out, s.c_str(), cap - 1); out[cap - 1] = 0; }
  return n;
}
// PCC_EMU_RACE_LOG=<file>: when the process ends, one "seen" line (proof that the checker was watching) and the reports
// that nobody has collected are appended to it -- how the unchanged `-m gpu` tests run under the checker (tests/test_emu_race.py)
__attribute__((destructor)) static void pcc_emu_race_at_exit() {
  const char* path = getenv("PCC_EMU_RACE_LOG");
  if (!path || !emu::race::on()) return;
  static char text[1 << 20];
  unsigned long long seen[4];
  for (int i = 0; i < 4; ++i) seen[i] = emu::race::g().seen[i].load();
  const size_t n = pcc_emu_race_report(text, sizeof(text));
  FILE* f = fopen(path, "a");
  if (!f) return;
  fprintf(f, "seen %d %llu %llu %llu %llu reports %zu\n", (int)getpid(), seen[0], seen[1], seen[2], seen[3], n);
  fputs(text, f);
  fclose(f);
}
extern "C" int pcc_emu_race_enabled(void) { return emu::race::on() ? 1 : 0; }
// accesses the checker has looked at so far: global plain, global atomic, LDS plain, LDS atomic (proof that it was watching)
extern "C" void pcc_emu_race_seen(unsigned long long out[4]) {
  for (int i = 0; i < 4; ++i) out[i] = emu::race::g().seen[i].load();
}
