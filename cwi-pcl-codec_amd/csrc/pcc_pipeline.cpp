// pcc_pipeline.cpp -- multi-frame encoder on one GPU: a pool of host threads, each with its own
// pcc_ctx (HIP stream + HBM arena + pinned landing buffers), so that the serial host stage of one frame
// (JPEG Huffman + static range coder) overlaps the GPU stage of the others.
//
// The reference encodes the frames of a sequence one after the other on one thread (eval.hpp:818-835).
// Frames are independent I-frames (impl.hpp:89-90,126-130); the only thing that ties them together is the
// header field frame_ID_ (impl.hpp:133), which is assigned here by sequence index, so the bitstreams are
// the ones the serial loop would have produced.
//
// Built on the public C ABI only (pcc_hotpath_launch / pcc_hotpath_finish / pcc_entropy_encode).
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/pcc_codec.h"

namespace {
typedef std::chrono::steady_clock Clock;
inline double us_since(Clock::time_point t0) { return std::chrono::duration<double, std::micro>(Clock::now() - t0).count(); }

struct Job {  // one pcc_pipeline_encode call
  const void* const* frames = nullptr;
  const size_t* counts = nullptr;
  size_t n_frames = 0, stride = 0, rgb_offset = 0;
  pcc_params params{};
  int mode = 0;  // 0 full encode, 1 GPU stage only (launch + finish)
};
}  // namespace

struct pcc_pipeline {
  int device = 0;
  std::vector<pcc_ctx*> ctxs;
  std::vector<std::thread> threads;
  std::mutex mu;
  std::condition_variable cv_work, cv_done;
  uint64_t generation = 0;   // bumped for every job
  bool stopping = false;
  int busy = 0;              // workers still inside the current job
  Job job;
  std::atomic<size_t> next{0};
  // results of the current / last job
  std::vector<std::vector<uint8_t>> streams;
  std::vector<pcc_bitstream> results;
  std::vector<int> status;
  std::string err;
  // timing of the last job, summed over frames (microseconds)
  std::mutex stat_mu;
  double t_launch = 0, t_finish = 0, t_entropy = 0, host_us[4] = {0, 0, 0, 0};
  size_t frames_done = 0;
  // HIP-event kernel times of the frames that ran on a context with profiling enabled (sums per kernel name)
  std::vector<const char*> k_name;
  std::vector<double> k_ms;
  std::vector<int> k_launches;
  size_t k_frames = 0;

  void worker(int w) {
    pcc_ctx* ctx = ctxs[(size_t)w];
    uint64_t seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return stopping || generation != seen; });
        if (stopping) return;
        seen = generation;
      }
      double tl = 0, tf = 0, te = 0, hu[4] = {0, 0, 0, 0};
      size_t done = 0;
      for (;;) {
        const size_t f = next.fetch_add(1);
        if (f >= job.n_frames) break;
        pcc_params prm = job.params;
        prm.frame_id = job.params.frame_id + (uint32_t)f;  // frame_ID_ by sequence index
        Clock::time_point t0 = Clock::now();
        int rc = pcc_hotpath_launch(ctx, job.frames[f], job.counts[f], job.stride, job.rgb_offset, &prm);
        tl += us_since(t0);
        pcc_hot_result hot;
        if (rc == PCC_OK) {
          t0 = Clock::now();
          rc = pcc_hotpath_finish(ctx, &hot);
          tf += us_since(t0);
          pcc_kernel_times kt;
          if (rc == PCC_OK && pcc_get_kernel_times(ctx, &kt) == PCC_OK && kt.count > 0) {
            std::lock_guard<std::mutex> lk(stat_mu);
            for (int i = 0; i < kt.count; ++i) {
              size_t k = 0;
              while (k < k_name.size() && strcmp(k_name[k], kt.name[i]) != 0) ++k;
              if (k == k_name.size()) { k_name.push_back(kt.name[i]); k_ms.push_back(0.0); k_launches.push_back(0); }
              k_ms[k] += kt.ms[i];
              ++k_launches[k];
            }
            ++k_frames;
          }
        }
        pcc_bitstream bs;
        memset(&bs, 0, sizeof(bs));
        if (rc == PCC_OK && job.mode == 0) {
          t0 = Clock::now();
          rc = pcc_entropy_encode(ctx, &hot, &prm, &bs);
          te += us_since(t0);
          double h[4];
          if (pcc_get_host_times(ctx, h) == PCC_OK)
            for (int i = 0; i < 4; ++i) hu[i] += h[i];
          if (rc == PCC_OK) {
            streams[f].assign(bs.data, bs.data + bs.len);  // the context's buffer is reused by its next frame
            results[f] = bs;
            results[f].data = streams[f].data();
          }
        }
        status[f] = rc;
        if (rc != PCC_OK && rc != PCC_ERR_EMPTY) {
          std::lock_guard<std::mutex> lk(stat_mu);
          if (err.empty()) err = pcc_last_error(ctx);
        }
        ++done;
      }
      {
        std::lock_guard<std::mutex> lk(stat_mu);
        t_launch += tl; t_finish += tf; t_entropy += te;
        for (int i = 0; i < 4; ++i) host_us[i] += hu[i];
        frames_done += done;
      }
      {
        std::lock_guard<std::mutex> lk(mu);
        if (--busy == 0) cv_done.notify_all();
      }
    }
  }
};

extern "C" {

pcc_pipeline* pcc_pipeline_create(int device, int n_workers) {
  if (n_workers < 1) n_workers = 1;
  pcc_pipeline* p = new pcc_pipeline();
  p->device = device;
  for (int w = 0; w < n_workers; ++w) {
    pcc_ctx* c = pcc_create(device);
    if (!c) {  // no usable GPU: there is no CPU fallback
      for (pcc_ctx* k : p->ctxs) pcc_destroy(k);
      delete p;
      return nullptr;
    }
    p->ctxs.push_back(c);
  }
  for (int w = 0; w < n_workers; ++w) p->threads.emplace_back([p, w] { p->worker(w); });
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
  delete p;
}

int pcc_pipeline_workers(pcc_pipeline* p) { return p ? (int)p->ctxs.size() : 0; }

pcc_ctx* pcc_pipeline_context(pcc_pipeline* p, int worker) {
  if (!p || worker < 0 || worker >= (int)p->ctxs.size()) return nullptr;
  return p->ctxs[(size_t)worker];
}

static int run_job(pcc_pipeline* p, const void* const* dev_frames, const size_t* n_points, size_t n_frames, size_t stride,
                   size_t rgb_offset, const pcc_params* params, int mode) {
  if (!p || !params || (n_frames && (!dev_frames || !n_points))) return PCC_ERR_ARG;
  {
    std::lock_guard<std::mutex> lk(p->mu);
    p->job.frames = dev_frames; p->job.counts = n_points; p->job.n_frames = n_frames;
    p->job.stride = stride; p->job.rgb_offset = rgb_offset; p->job.params = *params; p->job.mode = mode;
    p->streams.assign(n_frames, std::vector<uint8_t>());
    p->results.assign(n_frames, pcc_bitstream());
    p->status.assign(n_frames, PCC_OK);
    p->err.clear();
    p->t_launch = p->t_finish = p->t_entropy = 0;
    p->host_us[0] = p->host_us[1] = p->host_us[2] = p->host_us[3] = 0;
    p->frames_done = 0;
    p->k_name.clear(); p->k_ms.clear(); p->k_launches.clear(); p->k_frames = 0;
    p->next.store(0);
    p->busy = (int)p->threads.size();
    ++p->generation;
  }
  p->cv_work.notify_all();
  {
    std::unique_lock<std::mutex> lk(p->mu);
    p->cv_done.wait(lk, [&] { return p->busy == 0; });
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

int pcc_pipeline_gpu_stage_only(pcc_pipeline* p, const void* const* dev_frames, const size_t* n_points, size_t n_frames,
                                size_t stride, size_t rgb_offset, const pcc_params* params) {
  return run_job(p, dev_frames, n_points, n_frames, stride, rgb_offset, params, 1);
}

int pcc_pipeline_stats(pcc_pipeline* p, double out_us[8]) {
  if (!p || !out_us) return PCC_ERR_ARG;
  const double k = p->frames_done ? 1.0 / (double)p->frames_done : 0.0;
  out_us[0] = p->t_launch * k; out_us[1] = p->t_finish * k; out_us[2] = p->t_entropy * k;
  for (int i = 0; i < 4; ++i) out_us[3 + i] = p->host_us[i] * k;
  out_us[7] = (double)p->frames_done;
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

}  // extern "C"
