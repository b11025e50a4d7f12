// pcc_pipeline.cpp -- multi-frame encoder on one GPU.  Two kinds of host threads around a ring of pcc_ctx
// (each a HIP stream + HBM arena + pinned landing buffers):
//   GPU-stage threads   take a free context and the next frame, enqueue the kernels, sleep until the
//                       occupancy stream / JPEG rows have landed in the context's pinned buffers, and hand
//                       the context to the ready queue;
//   entropy threads     take a batch of ready contexts -- up to four; how many follows what is left of the call, see
//                       batch_wanted() -- and run the serial host stage (static range coder, JPEG stitching) for all of
//                       them in one loop (pcc_entropy_encode_many), which costs little more than one frame alone.  Then
//                       the contexts go back to the free list.
// So the GPU always has a few frames in flight, and every core that the host stage can get codes symbols.
//
// The reference encodes the frames of a sequence one after the other on one thread (eval.hpp:818-835).
// Frames are independent I-frames (impl.hpp:89-90,126-130); the only thing that ties them together is the
// header field frame_ID_ (impl.hpp:133), which is assigned here by sequence index, so the bitstreams are
// the ones the serial loop would have produced.
//
// Built on the public C ABI only (pcc_hotpath_launch / pcc_hotpath_finish / pcc_entropy_encode[2]).
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pcc_codec_tools.h"
#include "pcc_dev.h"
#include "pcc_numa.h"

namespace {
typedef std::chrono::steady_clock Clock;
inline double thread_cpu_us() {  // CPU time this thread has consumed
  timespec ts;
  clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts);
  return (double)ts.tv_sec * 1e6 + (double)ts.tv_nsec * 1e-3;
}
inline double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct Job {  // one pcc_pipeline_encode call
  const void* const* frames = nullptr;
  const size_t* counts = nullptr;
  size_t n_frames = 0, stride = 0, rgb_offset = 0;
  pcc_params params{};
  int mode = 0;  // 0 full encode, 1 GPU stage only (launch + finish)
  bool host_input = false;  // the frames are host pointers: upload through the pipeline's lane, then as above
};

struct Ready {  // a context whose GPU stage is done
  pcc_ctx* ctx;
  size_t frame;
  pcc_params prm;
  pcc_hot_result hot;
  Clock::time_point since;  // when it joined the ready queue
};
}  // namespace

struct pcc_pipeline {
  int device = 0;
  int n_entropy = 0, n_gpu = 0;
  int n_gpu_device = 0;  // of the n_gpu GPU-stage threads, how many take part when the frames already sit in HBM
  pcc_upload_lane* lane = nullptr;  // host-to-device copies of host-input jobs, one after the other
  std::vector<pcc_stream*> gpu_streams;  // one per GPU-stage thread, created one after the other: see pcc_use_stream in pcc_codec.h
  // Opt-in (pcc_pipeline_set_option "entropy_on_gpu"): the range coders run on the GPU, one wave per stream, in batches of
  // `gpu_batch` frames per entropy thread -- for hosts with fewer cores than the GPU stage can feed.
  bool entropy_on_gpu = false;   // what the job at hand uses (decided when the job starts)
  // 0 (default): the host coders, whose cost on this path has been measured.  1: the device coder -- a deployment choice for
  // hosts with few cores per GPU (round 2: a flush costs ~110 ns per symbol of its longest stream however many streams it
  // holds).  Option "entropy_on_gpu", PCC_PIPELINE_ENTROPY=host|gpu.  (Rounds 3-4 had a third setting that decided per call from
  // a cost estimate with two constants nobody had calibrated; it went with the other unmeasured forms.)
  int entropy_mode = 0;
  int numa_node = -1;                         // the node whose cores this pipeline's threads were given (-1: no placement by node)
  size_t pin_start = 0, pin_cores_taken = 0;  // this pipeline's range in g_pin_ranges, given back when the pipeline is destroyed
  int gpu_batch = 256;
  bool rc_lanes = false;  // option "rc_device_lanes": the form of the device range coder the entropy threads' batches launch
  std::vector<pcc_entropy_batch*> batches;  // one per entropy thread, made on first use
  // most frames an entropy thread codes in one call: four share a scalar loop (the default, measured on the GPU box);
  // PCC_PIPELINE_BATCH=16: sixteen through AVX-512 lanes (1.47 against 2.00 ns per symbol and stream on the build container's
  // Xeon, tools/ubench/rc_many.cpp; not yet timed on the GPU box's EPYC) -- every entropy thread then holds up to sixteen contexts
  int batch = 4;
  // developer aid (PCC_PIPELINE_TRACE=1): when each frame of a call left the GPU stage, when its coder loop started and ended,
  // how many frames shared the loop (microseconds since the call started; printed to stderr for calls of up to 64 frames)
  struct FrameTrace { double launched = 0, gpu_done = 0, ent_start = 0, ent_end = 0, stored = 0; int batch = 0, thread = -1; };
  std::vector<FrameTrace> trace;
  Clock::time_point job_t0;
  bool tracing = false;
  size_t taken = 0;  // frames of the job at hand that entropy threads have taken so far
  // How many frames the next entropy thread should code in one loop.  Four coders in one loop use the least CPU per frame
  // (1.3 ms against 3.3 ms for one alone) but the frames come out later: right while the frames outnumber the threads,
  // wrong for the last frames of a call and for short calls.  So: what is left, spread over HALF the entropy threads (the
  // CPUs of a job are usually hardware-thread pairs, and two coders on one core run at half speed each; measured on the
  // GPU box with tools/short_calls.py: 20 frames on 16 threads 7.5 ms with a divisor of 16, 6.0 ms with 8, 6.6 ms with
  // 5; PCC_PIPELINE_SPREAD = 16 / divisor) -- a call of 20 frames on 16 threads runs as 3 3 2 2 2 2 2 1 1 1 1, the early
  // frames in the larger batches; a long call runs in fours and tapers off at its end.  (Callers hold `mu`.)
  size_t batch_wanted() const {
    const size_t left = job.n_frames > taken ? job.n_frames - taken : 0;
    // (with the entropy threads on cores of their own -- see pcc_pipeline_create -- the last frames of a call are best
    // spread wider: 20 frames on 16 threads as five pairs and ten singles, 4.6 ms against 5.0 ms)
    static const double forced = [] { const char* e = pcc::dev_env("PCC_PIPELINE_SPREAD"); const double v = e ? atof(e) : 0.0; return v > 0.0 ? v : 0.0; }();
    const double spread = forced > 0.0 ? forced : (left <= 2 * (size_t)std::max(n_entropy, 1) ? 1.4 : 2.0);
    const size_t per = (size_t)(((double)left * spread + (double)n_entropy - 1.0) / (double)std::max(n_entropy, 1));
    return std::min<size_t>(std::max<size_t>(per, 1), (size_t)batch);
  }
  std::vector<pcc_ctx*> ctxs;
  std::vector<std::thread> threads;
  std::mutex mu;  // guards everything below up to `stat_mu`
  std::condition_variable cv_work, cv_done, cv_free, cv_ready;
  uint64_t generation = 0;   // bumped for every job
  bool stopping = false;
  int busy = 0;              // threads still inside the current job
  Job job;
  size_t next_frame = 0;     // next frame to hand to a GPU-stage thread
  size_t gpu_done = 0;       // frames whose GPU stage is over (ready, failed or dropped)
  std::vector<pcc_ctx*> free_ctx;
  std::deque<Ready> ready;
  // results of the current / last job
  std::vector<std::vector<uint8_t>> streams;  // bitstreams that did not fit the arena (first call, much larger frames)
  // One block of memory for the bitstreams of a call, kept from call to call: a megabyte per frame from the allocator
  // means mmap, page faults and munmap for every frame, on 20 threads at once.
  uint8_t* arena = nullptr;
  size_t arena_cap = 0;
  std::atomic<size_t> arena_used{0};
  size_t seen_max_len = 0;  // longest bitstream so far: sizes the arena of the next call

  bool arena_ensure(size_t bytes) {  // grows (and touches) the arena; false: out of memory, the vectors take over
    if (bytes <= arena_cap) return true;
    uint8_t* q = static_cast<uint8_t*>(malloc(bytes));
    if (!q) return false;
    // first touch here, not inside the frame loop.  (Not memset(q, 0, ...): the compiler folds malloc + memset-to-zero into
    // calloc, whose pages are only mapped when they are written -- 200 page faults per 820 KB bitstream, 0.35 ms of every
    // frame's entropy thread, seen with PCC_PIPELINE_TRACE as the time between "coder loop" and "stored".  Streaming
    // stores for the copy itself were tried as well: slower, the coder loops on the other hardware threads suffer.)
    for (size_t k = 0; k < bytes; k += 4096) static_cast<volatile uint8_t*>(q)[k] = 1;
    free(arena);
    arena = q;
    arena_cap = bytes;
    return true;
  }
  std::vector<pcc_bitstream> results;
  std::vector<int> status;
  std::string err;
  // timing of the last job, summed over frames (microseconds)
  std::mutex stat_mu;
  double t_launch = 0, t_finish = 0, t_entropy = 0, host_us[4] = {0, 0, 0, 0};
  double cpu_launch = 0, cpu_finish = 0, cpu_entropy = 0;  // CPU time (not wall time) of the same calls
  size_t frames_done = 0;
  // HIP-event kernel times of the frames that ran on a context with profiling enabled (sums per kernel name)
  std::vector<const char*> k_name;
  std::vector<double> k_ms;
  std::vector<int> k_launches;
  size_t k_frames = 0;

  void note_error(pcc_ctx* c, int rc) {
    if (rc == PCC_OK || rc == PCC_ERR_EMPTY) return;
    std::lock_guard<std::mutex> lk(stat_mu);
    if (err.empty()) err = pcc_last_error(c);
  }

  // wait for the next job; false when the pipeline shuts down
  bool next_job(uint64_t& seen) {
    std::unique_lock<std::mutex> lk(mu);
    cv_work.wait(lk, [&] { return stopping || generation != seen; });
    if (stopping) return false;
    seen = generation;
    return true;
  }
  double t_last_thread = 0;  // (trace) when the last thread left the job, and which one it was (GPU-stage threads first)
  int last_thread = -1;
  void job_done(int who = -1) {
    std::lock_guard<std::mutex> lk(mu);
    if (tracing) {
      const double t = us_since(job_t0);
      if (t > t_last_thread) { t_last_thread = t; last_thread = who; }
    }
    if (--busy == 0) cv_done.notify_all();
  }

  void gpu_thread(int index) {
    uint64_t seen = 0;
    while (next_job(seen)) {
      // A frame that comes from host memory spends most of its time waiting for its turn on the PCIe link (and, if
      // its memory is pageable, in the page-locking call): more threads keep the link busy.  With the frames in HBM
      // the extra frames in flight only get in each other's way on the GPU.
      if (!job.host_input && index >= n_gpu_device) { job_done(index); continue; }
      double tl = 0, tf = 0, cl = 0, cf = 0;
      for (;;) {
        Ready r;
        {
          std::unique_lock<std::mutex> lk(mu);
          if (next_frame >= job.n_frames) break;
          cv_free.wait(lk, [&] { return !free_ctx.empty(); });
          if (next_frame >= job.n_frames) break;
          r.frame = next_frame++;
          r.ctx = free_ctx.back();
          free_ctx.pop_back();
        }
        r.prm = job.params;
        r.prm.frame_id = job.params.frame_id + (uint32_t)r.frame;  // frame_ID_ by sequence index
        if ((size_t)index < gpu_streams.size() && gpu_streams[(size_t)index]) (void)pcc_use_stream(r.ctx, gpu_streams[(size_t)index]);
        Clock::time_point t0 = Clock::now();
        if (tracing) trace[r.frame].launched = us_since(job_t0);
        double c0 = thread_cpu_us();
        int rc = job.host_input
                     ? pcc_hotpath_launch_host(r.ctx, lane, job.frames[r.frame], job.counts[r.frame], job.stride, job.rgb_offset, &r.prm)
                     : pcc_hotpath_launch(r.ctx, job.frames[r.frame], job.counts[r.frame], job.stride, job.rgb_offset, &r.prm);
        tl += us_since(t0);
        cl += thread_cpu_us() - c0;
        if (rc == PCC_OK) {
          t0 = Clock::now();
          c0 = thread_cpu_us();
          rc = pcc_hotpath_finish(r.ctx, &r.hot);
          tf += us_since(t0);
          cf += thread_cpu_us() - c0;
        }
        pcc_kernel_times kt;
        if (rc == PCC_OK && pcc_get_kernel_times(r.ctx, &kt) == PCC_OK && kt.count > 0) {
          std::lock_guard<std::mutex> lk(stat_mu);
          for (int q = 0; q < kt.count; ++q) {
            size_t k = 0;
            while (k < k_name.size() && strcmp(k_name[k], kt.name[q]) != 0) ++k;
            if (k == k_name.size()) { k_name.push_back(kt.name[q]); k_ms.push_back(0.0); k_launches.push_back(0); }
            k_ms[k] += kt.ms[q];
            ++k_launches[k];
          }
          ++k_frames;
        }
        note_error(r.ctx, rc);
        if (tracing) trace[r.frame].gpu_done = us_since(job_t0);
        bool wake_one = false, wake_all = false;
        {
          std::lock_guard<std::mutex> lk(mu);
          status[r.frame] = rc;
          ++gpu_done;
          if (rc == PCC_OK && job.mode == 0) {
            r.since = Clock::now();
            ready.push_back(r);
          } else {  // nothing for the host stage to do: the context is free again
            free_ctx.push_back(r.ctx);
            cv_free.notify_one();
          }
          // a full batch is waiting: one entropy thread is enough (short calls: somebody has to start the frame's clock)
          wake_one = ready.size() >= batch_wanted() || (job.n_frames <= 4 * (size_t)n_entropy && job.mode == 0);
          wake_all = gpu_done >= job.n_frames;
        }
        if (wake_all) cv_ready.notify_all();
        else if (wake_one) cv_ready.notify_one();
      }
      {
        std::lock_guard<std::mutex> lk(stat_mu);
        t_launch += tl; t_finish += tf;
        cpu_launch += cl; cpu_finish += cf;
      }
      cv_ready.notify_all();
      job_done(index);
    }
  }

  // the results of a flushed batch go where the host stage would have put them
  void store_result(size_t frame, const pcc_bitstream& bs) {
    const size_t room = (bs.len + 63) & ~(size_t)63;
    const size_t off = arena_used.fetch_add(room);
    results[frame] = bs;
    if (off + room <= arena_cap) {
      memcpy(arena + off, bs.data, bs.len);
      results[frame].data = arena + off;
    } else {
      streams[frame].assign(bs.data, bs.data + bs.len);
      results[frame].data = streams[frame].data();
    }
  }

  // entropy stage on the GPU: take ready frames one by one, copy what their entropy stage needs into the thread's batch
  // (the context goes back to the GPU stage at once), flush when the batch is full or the job runs out of frames
  void entropy_thread_gpu(int index, double& te, double& ce, size_t& done) {
    // (option "entropy_gpu_batch" may have changed since the thread's batch was made: a batch of another capacity is replaced --
    //  a smaller one used to refuse the frames beyond its capacity, "the batch is full", and the call failed with PCC_ERR_STATE;
    //  found by random pipeline runs)
    if (batches[(size_t)index] && pcc_entropy_batch_capacity(batches[(size_t)index]) != (size_t)gpu_batch) {
      pcc_entropy_batch_destroy(batches[(size_t)index]);
      batches[(size_t)index] = nullptr;
    }
    if (!batches[(size_t)index]) batches[(size_t)index] = pcc_entropy_batch_create(device, (size_t)gpu_batch);
    pcc_entropy_batch* batch = batches[(size_t)index];
    if (batch) (void)pcc_entropy_batch_set_option(batch, "rc_device_lanes", rc_lanes ? 1 : 0);
    std::vector<size_t> frames_in_batch;
    std::vector<pcc_bitstream> outs;
    auto flush = [&]() {
      if (frames_in_batch.empty()) return;
      outs.assign(frames_in_batch.size(), pcc_bitstream());
      size_t n = 0;
      const int rc = batch ? pcc_entropy_batch_flush(batch, outs.data(), outs.size(), &n) : PCC_ERR_HIP;
      for (size_t k = 0; k < frames_in_batch.size(); ++k) {
        if (rc == PCC_OK && k < n) store_result(frames_in_batch[k], outs[k]);
        else { std::lock_guard<std::mutex> lk(mu); status[frames_in_batch[k]] = rc == PCC_OK ? PCC_ERR_STATE : rc; }
      }
      if (rc != PCC_OK) { std::lock_guard<std::mutex> lk(stat_mu); if (err.empty()) err = pcc_entropy_batch_last_error(batch); }
      done += frames_in_batch.size();
      frames_in_batch.clear();
    };
    for (;;) {
      Ready r;
      bool have = false, finished = false;
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_ready.wait(lk, [&] { return !ready.empty() || gpu_done >= job.n_frames; });
        if (!ready.empty()) { r = ready.front(); ready.pop_front(); have = true; }
        else finished = true;
      }
      if (have) {
        Clock::time_point t0 = Clock::now();
        const double c0 = thread_cpu_us();
        const int rc = batch ? pcc_entropy_batch_add(batch, &r.hot, &r.prm) : PCC_ERR_HIP;
        {
          std::lock_guard<std::mutex> lk(mu);
          if (rc < 0) status[r.frame] = rc;
          free_ctx.push_back(r.ctx);
        }
        if (rc < 0) { std::lock_guard<std::mutex> lk(stat_mu); if (err.empty()) err = pcc_entropy_batch_last_error(batch); }
        cv_free.notify_all();
        if (rc >= 0) frames_in_batch.push_back(r.frame);
        if ((int)frames_in_batch.size() >= gpu_batch) flush();
        te += us_since(t0);
        ce += thread_cpu_us() - c0;
      }
      if (finished) {
        Clock::time_point t0 = Clock::now();
        const double c0 = thread_cpu_us();
        flush();
        te += us_since(t0);
        ce += thread_cpu_us() - c0;
        break;
      }
    }
  }

  void entropy_thread(int index) {
    uint64_t seen = 0;
    while (next_job(seen)) {
      double te = 0, ce = 0, hu[4] = {0, 0, 0, 0};
      size_t done = 0;
      const bool on_gpu = entropy_on_gpu && job.mode == 0;
      if (on_gpu) entropy_thread_gpu(index, te, ce, done);
      while (!on_gpu) {
        constexpr int kAtOnce = PCC_MAX_FRAMES_AT_ONCE;
        Ready r[kAtOnce];
        int nr = 0;
        {
          std::unique_lock<std::mutex> lk(mu);
          // Four frames in one coder loop cost 1.5 ms of CPU per frame, one frame alone 3.3 ms (tools/rc_speed.py), and
          // the CPU is what limits the pipeline: wait for a full batch unless the GPU stage has nothing more to give.
          // Short calls: a frame does not sit in the queue for long just because the thread would like a larger batch -- the
          // frames of a short call trickle out of the GPU stage 100-400 us apart, and a loop that waits for its third
          // frame ends last (PCC_PIPELINE_TRACE: 20 frames, the triple that waited 430 us finished 400 us behind everybody else).
          const bool impatient = job.n_frames <= 4 * (size_t)n_entropy;
          auto enough = [&] { return ready.size() >= batch_wanted() || gpu_done >= job.n_frames; };
          while (!enough()) {
            if (impatient && !ready.empty()) {
              const Clock::time_point deadline = ready.front().since + std::chrono::microseconds(150);
              if (Clock::now() >= deadline) break;
              cv_ready.wait_until(lk, deadline);
            } else {
              cv_ready.wait(lk);
            }
          }
          const size_t want = batch_wanted();
          while (nr < kAtOnce && (size_t)nr < want && !ready.empty()) { r[nr++] = ready.front(); ready.pop_front(); }
          if (nr == 0) break;  // every frame went through the GPU stage and the queue is empty
          taken += (size_t)nr;
          if (!ready.empty() && ready.size() >= batch_wanted()) cv_ready.notify_one();  // enough left for another thread
        }
        pcc_bitstream bs[kAtOnce];
        memset(bs, 0, sizeof(bs));
        int rc[kAtOnce];
        pcc_ctx* c[kAtOnce];
        const pcc_hot_result* h[kAtOnce];
        const pcc_params* pp[kAtOnce];
        pcc_bitstream* o[kAtOnce];
        for (int i = 0; i < nr; ++i) { c[i] = r[i].ctx; h[i] = &r[i].hot; pp[i] = &r[i].prm; o[i] = &bs[i]; }
        Clock::time_point t0 = Clock::now();
        const double c0 = thread_cpu_us();
        if (tracing) for (int i = 0; i < nr; ++i) { trace[r[i].frame].ent_start = us_since(job_t0); trace[r[i].frame].batch = nr; trace[r[i].frame].thread = index; }
        const int rc_all = pcc_entropy_encode_many(nr, c, h, pp, o);
        if (tracing) for (int i = 0; i < nr; ++i) trace[r[i].frame].ent_end = us_since(job_t0);
        for (int i = 0; i < nr; ++i) rc[i] = rc_all;
        te += us_since(t0);
        ce += thread_cpu_us() - c0;
        for (int i = 0; i < nr; ++i) {
          if (rc[i] == PCC_OK) {
            double hh[4];
            if (pcc_get_host_times(r[i].ctx, hh) == PCC_OK)
              for (int q = 0; q < 4; ++q) hu[q] += hh[q];
            // the context's buffer is reused by its next frame: the bitstream moves to the call's arena
            const size_t room = (bs[i].len + 63) & ~(size_t)63;
            const size_t off = arena_used.fetch_add(room);
            results[r[i].frame] = bs[i];
            if (off + room <= arena_cap) {
              memcpy(arena + off, bs[i].data, bs[i].len);
              results[r[i].frame].data = arena + off;
            } else {
              streams[r[i].frame].assign(bs[i].data, bs[i].data + bs[i].len);
              results[r[i].frame].data = streams[r[i].frame].data();
            }
          }
          note_error(r[i].ctx, rc[i]);
          if (tracing) trace[r[i].frame].stored = us_since(job_t0);
          ++done;
        }
        {
          std::lock_guard<std::mutex> lk(mu);
          for (int i = 0; i < nr; ++i) {
            if (rc[i] != PCC_OK) status[r[i].frame] = rc[i];
            free_ctx.push_back(r[i].ctx);
          }
        }
        cv_free.notify_all();
      }
      {
        std::lock_guard<std::mutex> lk(stat_mu);
        t_entropy += te;
        cpu_entropy += ce;
        for (int i = 0; i < 4; ++i) host_us[i] += hu[i];
        frames_done += done;
      }
      job_done(1000 + index);
    }
  }
};

// set by pcc_pipeline_create_multi around the creation of each of its pipelines: which of the allowed cores are whose (the
// cores of the GPU's own NUMA node where the host says which that is: pcc_numa.h)
static thread_local const pcc::numa::Share* pin_share_hint = nullptr;
// cores handed to the entropy threads of earlier pipelines of this process: a pipeline created without a range of its own
// (no hint, no PCC_PIPELINE_PIN_OFFSET) starts behind them, so two plain pcc_pipeline_create calls do not pin to the same cores
// Which positions of a process's core range the entropy threads of its live pipelines sit on: [start, start + len) each, in any
// order of creation and destruction (a bump counter that pipelines decrement is only right when they die in LIFO order: create A
// and B, destroy A, create C -- C landed on B's cores, the half-speed case the pinning exists to avoid, while A's sat idle).
static std::mutex g_pin_mu;
static std::vector<std::pair<size_t, size_t>> g_pin_ranges;
// the lowest free stretch of `len` positions inside [0, span); if the span is full, behind everything taken so far, wrapping
static size_t pin_take(size_t len, size_t span) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  std::vector<std::pair<size_t, size_t>> taken = g_pin_ranges;
  std::sort(taken.begin(), taken.end());
  size_t at = 0, total = 0;
  bool found = false;
  for (const auto& r : taken) {
    if (!found && r.first >= at + len) found = true;
    if (!found) at = std::max(at, r.first + r.second);
    total += r.second;
  }
  if (!found && at + len > span) at = total % std::max<size_t>(span, 1);
  g_pin_ranges.emplace_back(at, len);
  return at;
}
static void pin_give_back(size_t start, size_t len) {
  std::lock_guard<std::mutex> lk(g_pin_mu);
  for (size_t i = 0; i < g_pin_ranges.size(); ++i)
    if (g_pin_ranges[i].first == start && g_pin_ranges[i].second == len) { g_pin_ranges.erase(g_pin_ranges.begin() + (long)i); return; }
}

// one logical CPU per physical core out of `allowed` (ascending): the lowest allowed hardware thread of each core
static std::vector<int> one_cpu_per_core(const std::string& root, const std::vector<int>& allowed) {
  std::vector<int> out;
  for (int c : allowed) {
    std::string text;
    int first = c;
    if (pcc::numa::read_text(root + "/devices/system/cpu/cpu" + std::to_string(c) + "/topology/thread_siblings_list", &text)) {
      const std::vector<int> siblings = pcc::numa::parse_cpulist(text);
      for (int sib : siblings)
        if (std::binary_search(allowed.begin(), allowed.end(), sib)) { first = sib; break; }
    }
    if (first == c) out.push_back(c);
  }
  return out;
}
static std::vector<int> allowed_cpus() {
  std::vector<int> out;
  cpu_set_t allowed;
  CPU_ZERO(&allowed);
  if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &allowed)) out.push_back(c);
  return out;
}
static std::string sysfs_root();
static std::vector<int> one_cpu_per_core() { return one_cpu_per_core(sysfs_root(), allowed_cpus()); }

// developer builds and the executor can point the planning at a made-up sysfs tree (tests); the shipped library reads /sys
static std::string sysfs_root() {
  const char* e = pcc::dev_env("PCC_SYSFS_ROOT");
  return e && *e ? e : "/sys";
}
// every allowed logical CPU of `node` (both hardware threads of a core): where a placed pipeline's GPU-stage threads may run
static std::vector<int> allowed_cpus_of_node(int node) {
  const std::vector<int> all = allowed_cpus();
  const std::vector<int> nodes = pcc::numa::nodes_of_cpus(sysfs_root(), all);
  std::vector<int> out;
  for (size_t i = 0; i < all.size(); ++i)
    if (nodes[i] == node) out.push_back(all[i]);
  return out;
}
// the shares of `n` pipelines on GPUs devices[0..n): cores of each GPU's own node where the host names it for all of them
static std::vector<pcc::numa::Share> plan_shares(const int* devices, int n) {
  const std::vector<int> cores = one_cpu_per_core();
  const std::string root = sysfs_root();
  std::vector<int> device_node((size_t)n, -1);
  for (int d = 0; d < n; ++d) device_node[(size_t)d] = pcc_debug_device_numa_node(devices[d], root == "/sys" ? nullptr : root.c_str());
  return pcc::numa::plan(cores, pcc::numa::nodes_of_cpus(root, cores), device_node);
}

extern "C" {

pcc_pipeline* pcc_pipeline_create(int device, int n_workers) {
  if (n_workers < 1) n_workers = 1;
  pcc_pipeline* p = new pcc_pipeline();
  p->device = device;
  p->n_entropy = n_workers;
  // frames in flight on the GPU: six saturate it to within a few per cent (tools/gpu_throughput.py: 4 streams 6 600, 8 streams
  // 7 750, 12 streams 8 070 frames/s); with the entropy stage no longer the bottleneck the last per cent count
  // (tools/sweep_gpu_threads.sh: 7 000 frames/s end to end with 6 threads, 7 400-7 500 with 10 or 12)
  // Twelve since the streams are balanced over the runtime's four hardware queues (three each; ten were 3 3 2 2, and the
  // ten contexts that happened to be in flight out of 92 anything): tools/queue_balance.py, 10 400 -> 11 200 frames/s.
  p->n_gpu_device = n_workers < 12 ? n_workers : 12;
  if (const char* e = getenv("PCC_PIPELINE_GPU_THREADS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) p->n_gpu_device = v;
  }
  p->n_gpu = std::max(p->n_gpu_device, std::min(n_workers, 12));
  if (const char* e = getenv("PCC_PIPELINE_UPLOAD_THREADS")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 64) p->n_gpu = std::max(p->n_gpu_device, v);
  }
  if (const char* e = getenv("PCC_PIPELINE_BATCH")) {
    const int v = atoi(e);
    if (v >= 1 && v <= PCC_MAX_FRAMES_AT_ONCE) p->batch = v;
  }
  // an entropy thread holds `batch` contexts, a GPU-stage thread one, plus a batch or two waiting in the queue
  const int n_ctx = p->batch * p->n_entropy + 2 * p->n_gpu + 2 * p->batch;
  for (int w = 0; w < n_ctx; ++w) {
    pcc_ctx* c = pcc_create(device);
    if (!c) {  // no usable GPU: there is no CPU fallback
      for (pcc_ctx* k : p->ctxs) pcc_destroy(k);
      delete p;
      return nullptr;
    }
    p->ctxs.push_back(c);
  }
  p->lane = pcc_upload_lane_create(device);
  // one stream per GPU-stage thread, created back to back (the runtime hands out its hardware queues round robin)
  // (PCC_PIPELINE_OWN_STREAMS=1: every context keeps its own stream, as before)
  if (!(pcc::dev_env("PCC_PIPELINE_OWN_STREAMS") && pcc::dev_env("PCC_PIPELINE_OWN_STREAMS")[0] == '1'))
    for (int w = 0; w < p->n_gpu; ++w) p->gpu_streams.push_back(pcc_stream_create(device));
  for (int w = 0; w < p->n_gpu; ++w) p->threads.emplace_back([p, w] { p->gpu_thread(w); });
  p->batches.assign((size_t)p->n_entropy, nullptr);
  if (const char* e = pcc::dev_env("PCC_RC_DEVICE")) p->rc_lanes = !strcmp(e, "lanes");
  if (const char* e = getenv("PCC_PIPELINE_ENTROPY")) {
    p->entropy_mode = !strcmp(e, "gpu") ? 1 : 0;
    if (strcmp(e, "gpu") && strcmp(e, "host")) {  // (rounds 3-4 knew "auto"; it is gone) -- said once per process, the host stage it is
      static std::atomic<bool> said{false};
      if (!said.exchange(true)) fprintf(stderr, "pcc_pipeline: PCC_PIPELINE_ENTROPY=%s is neither host nor gpu: the entropy stage runs on the host\n", e);
    }
  }
  for (int w = 0; w < p->n_entropy; ++w) p->threads.emplace_back([p, w] { p->entropy_thread(w); });
  // The entropy stage is a chain of dependent integer operations per symbol: two such threads on the two hardware
  // threads of one core run at about half speed each, and the scheduler does put them there (20 frames on 16 threads:
  // 4.8-7.4 ms from call to call; 256 frames: 34.6 ms).  So every entropy thread gets a GROUP of physical cores of its
  // own (up to eight; first hardware thread of each): the threads of this pipeline never share a core, and the scheduler
  // can still step aside when somebody else's work sits on one of them -- a thread pinned to ONE core that another
  // tenant of the host uses takes twice as long (13-25 ms outliers of a 5 ms call).  Measured (tools/pin_probe.sh):
  // 20 frames 6.1 -> 4.6 ms, 64 frames 12.1 -> 10.2 ms, 256 frames 34.6 -> 28.9 ms, 1 024 frames unchanged.
  //   PCC_PIPELINE_PIN = groups (default when the process may use at least two cores per entropy thread) | cores (one
  //   core per thread) | none;  PCC_PIPELINE_PIN_OFFSET = first core (in the list of allowed cores) of this pipeline,
  //   PCC_PIPELINE_PIN_SPAN = how many cores it may use: several pipelines / ranks on one host take different ranges.
  {
    std::vector<int> cores = one_cpu_per_core();
    const char* e = getenv("PCC_PIPELINE_PIN");
    // Whose cores: (1) the share pcc_pipeline_create_multi planned for this pipeline; (2) the caller's range
    // (PCC_PIPELINE_PIN_OFFSET / _SPAN, positions in the list of allowed cores); (3) one process per GPU on one host (torchrun:
    // LOCAL_RANK r of LOCAL_WORLD_SIZE n, rank i on GPU i): the share of pipeline r in the plan for GPUs 0..n-1 -- the cores of
    // this GPU's own NUMA node, split among the ranks whose GPUs hang off the same node, where the host says which node that is,
    // else an n-th of all cores; (4) all allowed cores.
    const bool env_range = getenv("PCC_PIPELINE_PIN_SPAN") || getenv("PCC_PIPELINE_PIN_OFFSET");
    size_t rank_base = 0, span = cores.size();
    if (pin_share_hint) {
      cores = pin_share_hint->cores;
      p->numa_node = pin_share_hint->node;
      span = cores.size();
    } else if (env_range) {
      if (const char* o = getenv("PCC_PIPELINE_PIN_SPAN")) span = std::min<size_t>(cores.size(), (size_t)std::max(atoi(o), 1));
    } else {
      const char* lr = getenv("LOCAL_RANK");
      const char* lw = getenv("LOCAL_WORLD_SIZE");
      const int nr = lw ? atoi(lw) : 1, r = lr ? atoi(lr) : 0;
      if (nr > 1 && r >= 0 && r < nr) {
        std::vector<pcc::numa::Share> shares;
        char last[64];
        if (device == r && pcc_debug_device_pci_bus_id(nr - 1, last, (int)sizeof(last)) == PCC_OK) {  // (GPU n-1 is there: rank i <-> GPU i holds)
          std::vector<int> devs((size_t)nr);
          for (int i = 0; i < nr; ++i) devs[(size_t)i] = i;
          shares = plan_shares(devs.data(), nr);
        }
        if (!shares.empty() && shares[(size_t)r].node >= 0) {
          cores = shares[(size_t)r].cores;
          p->numa_node = shares[(size_t)r].node;
          span = cores.size();
        } else {
          span = std::max<size_t>(cores.size() / (size_t)nr, 1);
          rank_base = (size_t)r * span;
        }
      }
    }
    int mode = (span >= 2 * (size_t)p->n_entropy) ? 2 : 0;  // 0 none, 1 cores, 2 groups
    if (e) mode = !strcmp(e, "cores") ? 1 : (!strcmp(e, "groups") ? 2 : 0);
    if (!mode) p->numa_node = -1;  // nobody is pinned (too few cores for this many threads, or PCC_PIPELINE_PIN=none): "numa_node" must not claim a placement
    const size_t per = mode == 2 ? std::min<size_t>(8, std::max<size_t>(1, span / (size_t)std::max(p->n_entropy, 1))) : 1;
    // the range [base, base + span) of `cores` is this pipeline's (the caller's, or this rank's share); `start` is where
    // inside it the first entropy thread goes -- behind the cores of earlier pipelines of this process, wrapping INSIDE
    // the range (a second pipeline of one rank must not land on the next rank's cores)
    size_t base = rank_base, start = 0;
    if (const char* o = getenv("PCC_PIPELINE_PIN_OFFSET")) base = (size_t)std::max(atoi(o), 0);
    else if (!pin_share_hint && mode) {
      p->pin_cores_taken = (size_t)p->n_entropy * per;
      start = p->pin_start = pin_take(p->pin_cores_taken, span);
    }
    if (mode && !cores.empty()) {
      for (int w = 0; w < p->n_entropy; ++w) {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (size_t k = 0; k < per; ++k) CPU_SET(cores[(base + (start + (size_t)w * per + k) % span) % cores.size()], &set);
        (void)pthread_setaffinity_np(p->threads[(size_t)p->n_gpu + w].native_handle(), sizeof(set), &set);
      }
    }
    // a placed pipeline's GPU-stage threads (launch calls, waits for the landings, the packing of host frames) stay on the
    // GPU's node too -- any of its allowed CPUs, they are not pinned to cores
    if (mode && p->numa_node >= 0) {
      const std::vector<int> of_node = allowed_cpus_of_node(p->numa_node);
      if (!of_node.empty()) {
        cpu_set_t set;
        CPU_ZERO(&set);
        for (int c : of_node) CPU_SET(c, &set);
        for (int w = 0; w < p->n_gpu; ++w) (void)pthread_setaffinity_np(p->threads[(size_t)w].native_handle(), sizeof(set), &set);
      }
    }
  }
  return p;
}

void pcc_pipeline_destroy(pcc_pipeline* p) {
  if (!p) return;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->stopping = true;
  }
  p->cv_work.notify_all();
  for (std::thread& t : p->threads) t.join();
  for (pcc_ctx* c : p->ctxs) pcc_destroy(c);
  for (pcc_entropy_batch* b : p->batches) pcc_entropy_batch_destroy(b);
  pcc_upload_lane_destroy(p->lane);
  for (pcc_stream* st : p->gpu_streams) pcc_stream_destroy(st);
  free(p->arena);
  if (p->pin_cores_taken) pin_give_back(p->pin_start, p->pin_cores_taken);  // the next pipeline of this process may have them
  delete p;
}

int pcc_pipeline_reserve(pcc_pipeline* p, size_t n_frames, size_t bytes_per_frame, size_t max_points_per_frame) {
  if (!p) return PCC_ERR_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  if (max_points_per_frame)
    for (pcc_ctx* c : p->ctxs) {
      const int rc = pcc_reserve(c, max_points_per_frame, bytes_per_frame);
      if (rc != PCC_OK) { p->err = pcc_last_error(c); return rc; }
    }
  p->seen_max_len = std::max(p->seen_max_len, bytes_per_frame);
  return p->arena_ensure(n_frames * ((p->seen_max_len + p->seen_max_len / 16 + 127) & ~(size_t)63)) ? PCC_OK : PCC_ERR_HIP;
}

int pcc_pipeline_set_option(pcc_pipeline* p, const char* name, int value) {
  if (!p || !name) return PCC_ERR_ARG;
  std::lock_guard<std::mutex> lk(p->mu);  // between jobs: the threads read these when a job starts
  if (!strcmp(name, "entropy_on_gpu")) {
    if (value != 0 && value != 1) return PCC_ERR_ARG;  // (-1 meant "decide per call" in rounds 3-4: refused, not silently taken for the GPU)
    p->entropy_mode = value;
  }
  else if (!strcmp(name, "entropy_gpu_batch")) p->gpu_batch = value < 1 ? 1 : (value > 4096 ? 4096 : value);
  else if (!strcmp(name, "rc_device_lanes")) p->rc_lanes = value != 0;
  else if (!strcmp(name, "pack_upload")) { for (pcc_ctx* c : p->ctxs) (void)pcc_set_option(c, "pack_upload", value); }  // host frames: 16 B per point over PCIe
  else return PCC_ERR_ARG;
  return PCC_OK;
}

int pcc_pipeline_get(pcc_pipeline* p, const char* name) {
  if (!p || !name) return PCC_ERR_ARG;
  std::lock_guard<std::mutex> lk(p->mu);
  if (!strcmp(name, "workers")) return p->n_entropy;
  if (!strcmp(name, "gpu_threads")) return p->n_gpu_device;
  if (!strcmp(name, "contexts")) return (int)p->ctxs.size();
  if (!strcmp(name, "frames_per_coder_call")) return p->batch;
  if (!strcmp(name, "last_entropy_mode")) return p->entropy_on_gpu ? 1 : 0;
  if (!strcmp(name, "rc_device_lanes")) return p->rc_lanes ? 1 : 0;
  if (!strcmp(name, "entropy_gpu_batch")) return p->gpu_batch;
  if (!strcmp(name, "numa_node")) return p->numa_node >= 0 ? p->numa_node : PCC_NO_NUMA_NODE;
  return PCC_ERR_ARG;
}
// developer aid (include/pcc_codec_tools.h, the pcc_debug_* block): the CPUs entropy thread `worker` may run on, lowest first; returns how many
// there are (at most `cap` are written), -1 for a bad argument
int pcc_debug_pipeline_cpus(pcc_pipeline* p, int worker, int* out, int cap) {
  if (!p || worker < 0 || worker >= p->n_entropy || (cap > 0 && !out)) return -1;
  cpu_set_t set;
  CPU_ZERO(&set);
  if (pthread_getaffinity_np(p->threads[(size_t)p->n_gpu + (size_t)worker].native_handle(), sizeof(set), &set) != 0) return -1;
  int n = 0;
  for (int c = 0; c < CPU_SETSIZE; ++c)
    if (CPU_ISSET(c, &set)) { if (n < cap) out[n] = c; ++n; }
  return n;
}
// developer aid (include/pcc_codec_tools.h): the planning of pcc_pipeline_create_multi / of the ranks of one host without a GPU,
// on a sysfs tree and a CPU list the caller names
int pcc_debug_numa_plan(const char* root, const char* const* pci, int n_devices, const int* cpus, int n_cpus, int* node_out, int* cores_out,
                        int* n_cores_out, int cap) {
  if (!root || !pci || n_devices < 1 || !cpus || n_cpus < 1 || !node_out || !cores_out || !n_cores_out || cap < 1) return PCC_ERR_ARG;
  std::vector<int> allowed(cpus, cpus + n_cpus);
  std::sort(allowed.begin(), allowed.end());
  const std::vector<int> cores = one_cpu_per_core(root, allowed);
  std::vector<int> device_node((size_t)n_devices);
  for (int d = 0; d < n_devices; ++d) device_node[(size_t)d] = pcc::numa::pci_numa_node(root, pci[d]);
  const std::vector<pcc::numa::Share> shares = pcc::numa::plan(cores, pcc::numa::nodes_of_cpus(root, cores), device_node);
  for (int d = 0; d < n_devices; ++d) {
    const pcc::numa::Share& sh = shares[(size_t)d];
    node_out[d] = sh.node;
    n_cores_out[d] = (int)sh.cores.size();
    for (size_t k = 0; k < sh.cores.size() && (int)k < cap; ++k) cores_out[(size_t)d * (size_t)cap + k] = sh.cores[k];
  }
  return PCC_OK;
}
int pcc_pipeline_contexts(pcc_pipeline* p) { return p ? (int)p->ctxs.size() : 0; }

pcc_ctx* pcc_pipeline_context(pcc_pipeline* p, int index) {
  if (!p || index < 0 || index >= (int)p->ctxs.size()) return nullptr;
  return p->ctxs[(size_t)index];
}

static int run_job(pcc_pipeline* p, const void* const* dev_frames, const size_t* n_points, size_t n_frames, size_t stride,
                   size_t rgb_offset, const pcc_params* params, int mode, bool host_input = false) {
  if (!p || !params || (n_frames && (!dev_frames || !n_points))) return PCC_ERR_ARG;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->job.frames = dev_frames; p->job.counts = n_points; p->job.n_frames = n_frames;
    p->job.stride = stride; p->job.rgb_offset = rgb_offset; p->job.params = *params; p->job.mode = mode;
    p->job.host_input = host_input;
    {
      // where the entropy stage of this call runs
      const bool on_gpu = p->entropy_mode == 1;
      p->entropy_on_gpu = on_gpu && mode == 0;
    }
    p->streams.assign(n_frames, std::vector<uint8_t>());
    if (mode == 0 && p->seen_max_len) p->arena_ensure(n_frames * ((p->seen_max_len + p->seen_max_len / 16 + 127) & ~(size_t)63));
    p->arena_used = 0;
    p->taken = 0;
    {
      static const bool want = [] { const char* e = pcc::dev_env("PCC_PIPELINE_TRACE"); return e && e[0] == '1'; }();
      p->tracing = want && mode == 0 && n_frames <= 64;
      if (p->tracing) p->trace.assign(n_frames, pcc_pipeline::FrameTrace());
      p->t_last_thread = 0;
      p->job_t0 = Clock::now();
    }
    p->results.assign(n_frames, pcc_bitstream());
    p->status.assign(n_frames, PCC_OK);
    p->err.clear();
    p->t_launch = p->t_finish = p->t_entropy = 0;
    p->cpu_launch = p->cpu_finish = p->cpu_entropy = 0;
    p->host_us[0] = p->host_us[1] = p->host_us[2] = p->host_us[3] = 0;
    p->frames_done = 0;
    p->k_name.clear(); p->k_ms.clear(); p->k_launches.clear(); p->k_frames = 0;
    p->next_frame = 0;
    p->gpu_done = 0;
    p->ready.clear();
    p->free_ctx = p->ctxs;
    p->busy = (int)p->threads.size();
    ++p->generation;
  }
  p->cv_work.notify_all();
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->busy == 0; });
  }
  if (p->tracing) {
    fprintf(stderr, "[pcc_pipeline] call of %zu frames done after %.0f us (the last thread, %d, left the job at %.0f us)\n", n_frames, us_since(p->job_t0), p->last_thread, p->t_last_thread);
    for (size_t f = 0; f < n_frames; ++f) {
      const pcc_pipeline::FrameTrace& t = p->trace[f];
      fprintf(stderr, "  frame %2zu: launched %6.0f  left the GPU stage %6.0f  coder loop %6.0f .. %6.0f, stored %6.0f  (%d in the loop, entropy thread %d)\n", f, t.launched,
              t.gpu_done, t.ent_start, t.ent_end, t.stored, t.batch, t.thread);
    }
  }
  // frame_ID_ is only incremented for frames that are not dropped (impl.hpp:133 vs :206-212): renumber the
  // headers in sequence order (u32 behind the two 28 + 20 byte identifiers)
  if (mode == 0) {
    uint32_t id = params->frame_id;
    for (size_t f = 0; f < n_frames; ++f) {
      if (p->status[f] != PCC_OK || p->results[f].len < 52) continue;
      memcpy(const_cast<uint8_t*>(p->results[f].data) + 48, &id, sizeof(id));
      ++id;
      p->seen_max_len = std::max(p->seen_max_len, p->results[f].len);
    }
  }
  for (size_t f = 0; f < n_frames; ++f)
    if (p->status[f] != PCC_OK && p->status[f] != PCC_ERR_EMPTY) return p->status[f];
  return PCC_OK;
}

int pcc_pipeline_encode(pcc_pipeline* p, const void* const* dev_frames, const size_t* n_points, size_t n_frames,
                        size_t stride, size_t rgb_offset, const pcc_params* params, pcc_bitstream* out) {
  const int rc = run_job(p, dev_frames, n_points, n_frames, stride, rgb_offset, params, 0);
  if (out && p)
    for (size_t f = 0; f < n_frames && f < p->results.size(); ++f) out[f] = p->results[f];
  return rc;
}

int pcc_pipeline_encode_host(pcc_pipeline* p, const void* const* host_frames, const size_t* n_points, size_t n_frames,
                             size_t stride, size_t rgb_offset, const pcc_params* params, pcc_bitstream* out) {
  const int rc = run_job(p, host_frames, n_points, n_frames, stride, rgb_offset, params, 0, true);
  if (out && p)
    for (size_t f = 0; f < n_frames && f < p->results.size(); ++f) out[f] = p->results[f];
  return rc;
}

int pcc_pipeline_gpu_stage_only(pcc_pipeline* p, const void* const* dev_frames, const size_t* n_points, size_t n_frames,
                                size_t stride, size_t rgb_offset, const pcc_params* params) {
  return run_job(p, dev_frames, n_points, n_frames, stride, rgb_offset, params, 1);
}

int pcc_pipeline_stats(pcc_pipeline* p, double out_us[8]) {
  if (!p || !out_us) return PCC_ERR_ARG;
  const double k = p->frames_done ? 1.0 / (double)p->frames_done : 0.0;
  const double kg = p->gpu_done ? 1.0 / (double)p->gpu_done : 0.0;  // frames that went through the GPU stage
  out_us[0] = p->t_launch * kg; out_us[1] = p->t_finish * kg; out_us[2] = p->t_entropy * k;
  for (int i = 0; i < 4; ++i) out_us[3 + i] = p->host_us[i] * k;
  out_us[7] = (double)p->frames_done;
  return PCC_OK;
}

int pcc_pipeline_cpu_times(pcc_pipeline* p, double out_us[4]) {
  if (!p || !out_us) return PCC_ERR_ARG;
  const double k = p->frames_done ? 1.0 / (double)p->frames_done : 0.0;
  const double kg = p->gpu_done ? 1.0 / (double)p->gpu_done : 0.0;
  out_us[0] = p->cpu_launch * kg; out_us[1] = p->cpu_finish * kg; out_us[2] = p->cpu_entropy * k;
  out_us[3] = (double)p->frames_done;
  return PCC_OK;
}

int pcc_pipeline_kernel_times(pcc_pipeline* p, pcc_kernel_times* sums, int32_t* launches, int32_t* frames) {
  if (!p || !sums || !launches || !frames) return PCC_ERR_ARG;
  sums->count = (int32_t)(p->k_name.size() < (size_t)PCC_MAX_KERNEL_TIMES ? p->k_name.size() : (size_t)PCC_MAX_KERNEL_TIMES);
  for (int i = 0; i < sums->count; ++i) {
    sums->name[i] = p->k_name[(size_t)i];
    sums->ms[i] = (float)p->k_ms[(size_t)i];
    launches[i] = p->k_launches[(size_t)i];
  }
  *frames = (int32_t)p->k_frames;
  return PCC_OK;
}

const char* pcc_pipeline_last_error(pcc_pipeline* p) { return p ? p->err.c_str() : "no pipeline (no usable HIP device?)"; }

// ---- several GPUs behind one call: frame f of a sequence goes to devices[f mod n] ----
// One pcc_pipeline per entry of `devices` (the same GPU may be named twice: two pipelines share it).  Frames are
// independent I-frames, so there is no exchange between the GPUs; the only thing that ties the frames together,
// frame_ID_, is written into the finished bitstreams in sequence order, dropped frames not counting (impl.hpp:133 vs
// 206-212), which makes the result byte-identical to the reference's serial loop on one codec object.
struct pcc_multi_pipeline {
  std::vector<pcc_pipeline*> pipes;
  std::vector<int> devices;
  std::string err;
};

pcc_multi_pipeline* pcc_pipeline_create_multi(const int* devices, int n_devices, int n_workers_per_device) {
  if (!devices || n_devices < 1) return nullptr;
  pcc_multi_pipeline* m = new pcc_multi_pipeline();
  // every pipeline's entropy threads on cores of its own: of its GPU's NUMA node where the host names one for every GPU
  // (pcc_numa.h), else an n-th of the allowed cores each
  const std::vector<pcc::numa::Share> shares = plan_shares(devices, n_devices);
  for (int d = 0; d < n_devices; ++d) {
    pin_share_hint = shares[(size_t)d].cores.empty() ? nullptr : &shares[(size_t)d];
    pcc_pipeline* p = pcc_pipeline_create(devices[d], n_workers_per_device);
    pin_share_hint = nullptr;
    if (!p) {  // a device that does not exist: nothing is silently left out
      for (pcc_pipeline* q : m->pipes) pcc_pipeline_destroy(q);
      delete m;
      return nullptr;
    }
    m->pipes.push_back(p);
    m->devices.push_back(devices[d]);
  }
  return m;
}

void pcc_multi_pipeline_destroy(pcc_multi_pipeline* m) {
  if (!m) return;
  for (pcc_pipeline* p : m->pipes) pcc_pipeline_destroy(p);
  delete m;
}

int pcc_multi_pipeline_size(pcc_multi_pipeline* m) { return m ? (int)m->pipes.size() : 0; }
pcc_pipeline* pcc_multi_pipeline_member(pcc_multi_pipeline* m, int index) {
  return (m && index >= 0 && index < (int)m->pipes.size()) ? m->pipes[(size_t)index] : nullptr;
}
const char* pcc_multi_pipeline_last_error(pcc_multi_pipeline* m) { return m ? m->err.c_str() : "no multi-GPU pipeline (a device is missing)"; }

static int multi_run(pcc_multi_pipeline* m, const void* const* frames, const size_t* n_points, size_t n_frames, size_t stride,
                     size_t rgb_offset, const pcc_params* params, pcc_bitstream* out, bool host_input) {
  if (!m || !params || (n_frames && (!frames || !n_points || !out))) return PCC_ERR_ARG;
  const size_t nd = m->pipes.size();
  std::vector<std::vector<const void*>> fr(nd);
  std::vector<std::vector<size_t>> cn(nd);
  std::vector<std::vector<pcc_bitstream>> res(nd);
  for (size_t f = 0; f < n_frames; ++f) {
    fr[f % nd].push_back(frames[f]);
    cn[f % nd].push_back(n_points[f]);
  }
  std::vector<int> rc(nd, PCC_OK);
  std::vector<std::thread> th;
  for (size_t d = 0; d < nd; ++d) {
    res[d].assign(fr[d].size(), pcc_bitstream());
    th.emplace_back([&, d] {
      rc[d] = host_input ? pcc_pipeline_encode_host(m->pipes[d], fr[d].data(), cn[d].data(), fr[d].size(), stride, rgb_offset, params, res[d].data())
                         : pcc_pipeline_encode(m->pipes[d], fr[d].data(), cn[d].data(), fr[d].size(), stride, rgb_offset, params, res[d].data());
    });
  }
  for (std::thread& t : th) t.join();
  m->err.clear();
  int first_bad = PCC_OK;
  for (size_t d = 0; d < nd; ++d)
    if (rc[d] != PCC_OK && first_bad == PCC_OK) {
      first_bad = rc[d];
      m->err = pcc_pipeline_last_error(m->pipes[d]);
    }
  uint32_t id = params->frame_id;
  for (size_t f = 0; f < n_frames; ++f) {
    out[f] = res[f % nd][f / nd];
    if (out[f].len >= 52) {  // frame_ID_: u32 behind the two identifiers (28 + 20 bytes)
      memcpy(const_cast<uint8_t*>(out[f].data) + 48, &id, sizeof(id));
      ++id;
    }
  }
  return first_bad;
}

int pcc_multi_pipeline_encode_host(pcc_multi_pipeline* m, const void* const* host_frames, const size_t* n_points, size_t n_frames,
                                   size_t stride, size_t rgb_offset, const pcc_params* params, pcc_bitstream* out) {
  return multi_run(m, host_frames, n_points, n_frames, stride, rgb_offset, params, out, true);
}
int pcc_multi_pipeline_encode(pcc_multi_pipeline* m, const void* const* dev_frames, const size_t* n_points, size_t n_frames,
                              size_t stride, size_t rgb_offset, const pcc_params* params, pcc_bitstream* out) {
  return multi_run(m, dev_frames, n_points, n_frames, stride, rgb_offset, params, out, false);
}

}  // extern "C"
