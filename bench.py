#!/usr/bin/env python3
"""Headline benchmark: encoded Mpoints/s for 1M-point XYZRGB frames, octree depth 10, intra-only.

A "step" is one complete encodePointCloud-equivalent of one frame whose points already sit in
HBM: GPU hot path (bounding box, keys, sort, leaves, occupancy stream, colour image + JPEG front
end), device->host hand-over, then the host entropy stage (JPEG Huffman + static range coder) down
to the final bitstream.  Frames of the GOP are sharded one per GPU (rank r takes frames r, r+N, ...)
with no collective on the data path; inside a rank the native pipeline (pcc_pipeline, C++ threads,
one pcc_ctx each) overlaps the serial host stage of one frame with the GPU stage of the others.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract in the task statement) carrying `roofline` and
`cpu_baseline` (the CPU oracle on a bounded sample; rank 0, N=1 only).

roofline: the kernel durations come from a SERIALISED leg -- one frame at a time on one stream, nothing
else on the GPU -- because inside the timed region six frames share the GPU and a kernel's duration then
contains the other frames' work.  Two clocks are reported: `kernel_avg_ms`, the launch's span on the GPU's
own real-time clock (first workgroup start to last wave end: what `rocprofv3 --kernel-trace` reports, see
profiles/), and `kernel_avg_ms_hip_events`, HIP events recorded on the launch stream between the kernels
(they add a few microseconds of their own to every short kernel).  The dominant kernel is the one with
the largest sum of spans per frame.  `path_gpu_ms` is the HIP-event time from before the first to behind
the last kernel of a frame without any events in between.

`--gpus N` without a torchrun environment starts the N ranks itself (and fails loudly if it cannot).
"""
import argparse
import json
import os
import resource   # (at the top on purpose: a module imported -- a shared object loaded -- while the pipeline's threads exist
                  #  costs the run a quarter of its throughput, see main())
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--workload", default="cfg2", help="cfg2 (headline), cfg2u, cfg1, cfg3, cfg4")
    ap.add_argument("--workers", type=int, default=0, help="host threads (= contexts) per GPU; 0 = auto")
    ap.add_argument("--distinct-frames", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-frames", type=int, default=0, help="frames for the CPU baseline sample (0 = auto)")
    ap.add_argument("--no-host-input", action="store_true", help="skip the legs that start from host memory")
    ap.add_argument("--host-frames", type=int, default=0, help="frames of the host-input legs (0 = min(steps, 256))")
    ap.add_argument("--traffic-json", default="",
                    help="per-kernel HBM bytes per launch from a separate rocprofv3 --pmc run (tools/pmc_traffic.py)")
    return ap.parse_args()


# must-move bytes of each kernel per launch, as a function of the frame ("Kernels")
def kernel_algorithmic_bytes(name, n, L, B, with_color, image_bytes):
    # no centroids in the bench workloads: the sort key is [code | colour] or the code alone, 8 bytes, no payload array
    col = 4 if with_color else 0
    if name == "k_boxes_events":
        return 16 * n                      # x,y,z(,w) of every point
    if name == "k_make_keys":
        return (16 + col) * n + 8 * n      # x,y,z(,w) (+ colour word) of every point, key out
    if name == "k_sort_pass":
        return 2 * 8 * n                   # read keys, write keys
    if name == "k_leaf_scan":
        return 8 * n + 17 * L + B          # read keys; write start, code, base, t per leaf; zero the DFS stream
    if name == "k_leaf_tile":              # leaf records, keys (their colour bits); bgr, image rows, simplified cloud, DFS stream
        return 17 * L + (8 * n + 3 * L + image_bytes if with_color else 0) + 16 * L + B
    if name == "k_jpeg_rows":              # image rows in; 1.5 int16 coefficients per pixel = 3 bytes per pixel out
        return 2 * image_bytes if with_color else 0
    return 0


def cgroup_throttled():
    """(nr_throttled, throttled_usec) of this job's cgroup: the CPU quota (cpu.max) stalls threads until the next 100 ms
    period when it runs out -- on a CPU whose local slice is used up, others go on -- which shows up as a 5 ms call
    taking 15 ms.  None where the file is not there.  (Plain os.open / os.read: this runs right next to the timed region,
    where nothing may be imported -- see the note there -- and Python's text layer imports on first use.)"""
    try:
        fd = os.open("/sys/fs/cgroup/cpu.stat", os.O_RDONLY)
        try:
            d = dict(l.split() for l in os.read(fd, 4096).split(b"\n") if l)
        finally:
            os.close(fd)
        return int(d[b"nr_throttled"]), int(d[b"throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def default_workers(world):
    """Entropy threads per GPU: the CPUs this process may really use (cgroup quota, affinity), shared by the ranks."""
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cpus = min(cpus, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(32, cpus // max(world, 1)))


def main():
    args = parse()
    # before anything else: what is about to be measured?  Exits unless the gfx950 library is what the binding loads
    # (PCC_LIB can point it at the CPU executor's build of the same sources, or at a developer build)
    import __graft_entry__ as G
    library = G.load_package().binding.require_product_library("bench.py")
    cgroup_throttled()  # first use here, long before the pipeline's threads exist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torchrun: start the ranks ourselves (one process per GPU) and pass rank 0's line through
        import subprocess
        port = 29500 + (os.getpid() % 2000)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        r = subprocess.run(cmd)
        raise SystemExit(r.returncode)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    # testing hook: several ranks on ONE GPU over gloo (a 1-GPU box cannot run RCCL between two ranks)
    share_gpu = os.environ.get("PCC_BENCH_SHARE_GPU0") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1 or os.environ.get("PCC_BENCH_FORCE_DIST") == "1":  # the hook runs the RCCL calls on a single rank (1-GPU box)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            # RCCL prints a banner ("Hostname : ...", "Librccl path : ...") on STDOUT when the communicator is made;
            # stdout is for the one JSON line: send it to stderr while the communicator comes up
            import ctypes
            saved = os.dup(1)
            sys.stdout.flush()
            os.dup2(2, 1)
            try:
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
                dist.barrier()
                torch.cuda.synchronize()
            finally:
                ctypes.CDLL(None).fflush(None)
                os.dup2(saved, 1)
                os.close(saved)

    import __graft_entry__ as G
    pkg = G.load_package()
    b, syn = pkg.binding, pkg.synthetic
    trace = (lambda m: print("[rank %d] %s" % (rank, m), file=sys.stderr, flush=True)) if os.environ.get("PCC_BENCH_TRACE") else (lambda m: None)

    cfg = syn.CONFIGS[args.workload]
    kw = dict(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
              jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
    params = b.make_params(frame_id=1, **kw)
    n_points = cfg["n"]
    workers = args.workers or default_workers(world)

    # ---- frames of this rank, resident in HBM before the clock starts ----
    # The ranks of one host share its cores: the library gives every rank's pipeline cores of its own from torchrun's LOCAL_RANK /
    # LOCAL_WORLD_SIZE -- cores of the NUMA node its GPU hangs off where the host names one for every GPU (csrc/pcc_numa.h), an
    # n-th of the allowed cores otherwise.  PCC_PIPELINE_PIN_SPAN / _OFFSET override it.  Which it was: "numa_node" per rank.
    pipe = b.Pipeline(local_rank, workers)
    ctx0 = pipe.context(0)
    for w in range(pipe.n_contexts):
        pipe.context(w).set_option("copy_image", 0)  # the host stage only needs the quantised JPEG coefficients
    n_distinct = max(1, args.distinct_frames)
    host_frames = [syn.make_frame(args.workload, frame=rank * n_distinct + f) for f in range(n_distinct)]
    dev_frames = [ctx0.upload(f) for f in host_frames]
    trace("frames uploaded, %d workers" % pipe.workers)

    def sync_all():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # one frame through a plain context call: L, B, depth for the report (and a first warm-up)
    ctx0.hotpath_launch(dev_frames[0], n_points, params)
    hot0 = ctx0.hotpath_finish(copy=False)
    L, B, depth = hot0.n_leaves, hot0.n_branches, hot0.depth
    lib0 = b.load_library()
    placement = {"numa_node": pipe.get("numa_node"),                                            # whose cores this rank's host threads sit on (None: no placement by node)
                 "gpu_numa_node": (lambda v: None if v < 0 else v)(lib0.pcc_debug_device_numa_node(local_rank, None)),
                 "landing_buffer_numa_node": (lambda v: None if v < 0 else v)(lib0.pcc_debug_address_node(hot0.raw.occupancy))}  # where the runtime put the page-locked landing buffer
    image_bytes = hot0.image_w * hot0.image_h * 3

    trace("first frame done: L=%d B=%d D=%d" % (L, B, depth))
    warm = max(args.warmup, pipe.n_contexts)
    wres = pipe.encode([dev_frames[s % n_distinct] for s in range(warm)], [n_points] * warm, params, copy=False)
    # output memory for the timed frames is set aside beforehand, like the input frames are resident beforehand
    pipe.reserve(args.steps, max(r[0] for r in wres), n_points)

    # HIP events between the kernels of ONE context: live kernel durations from inside the timed region
    # without taxing every stream (the free list is a stack: its top context takes part in every round)
    # (only in long runs: a profiled frame creates its events on first use, which a 20-frame region would mostly consist of)
    prof_ctxs = [pipe.context(pipe.n_contexts - 1 - k) for k in range(min(4, pipe.n_contexts))] if (args.steps >= 256 and not os.environ.get("PCC_BENCH_NO_KERNEL_EVENTS")) else []
    for c in prof_ctxs:
        c.set_profiling(True)
    seq = [dev_frames[s % n_distinct] for s in range(args.steps)]
    trace("warm-up done")
    sync_all()
    # Nothing may be imported from here to the end of the timed region.  Measured: `import resource` at this point -- the
    # dlopen of a small extension module -- took the 1024-frame run from 8 500 to 6 300 Mpoints/s.  glibc 2.35 sends every
    # access to a thread-local variable of a dlopen-ed library (the HIP runtime and libpcc_hip.so are loaded that way by
    # ctypes / torch) through the slow path of __tls_get_addr once a later dlopen has bumped the TLS generation counter.
    # The warm-up ends with one untimed call of the timed call's own shape.  The first long call of a process is slower than
    # every later one (1 024 frames: 148 ms, then 101-106 ms; 3 000-11 000 page faults against 56): contexts that the short
    # warm-up call never reached -- the free list is a stack, a short call uses its top -- make their host-side buffers
    # inside it.  (PCC_BENCH_SHAPE_WARMUP=0 leaves it out, =N repeats it.)
    for rep in range(int(os.environ.get("PCC_BENCH_SHAPE_WARMUP", "1"))):
        tx = time.perf_counter()
        pipe.encode(seq, [n_points] * args.steps, params, copy=False)
        trace("warm-up call %d of the timed call's shape: %.3f ms" % (rep, 1e3 * (time.perf_counter() - tx)))
        pipe.reserve(args.steps, max(r[0] for r in wres), n_points)  # what a caller does before a sequence (see pcc_reserve)
    sync_all()
    thr0 = cgroup_throttled()
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    res = pipe.encode(seq, [n_points] * args.steps, params, copy=False)
    t_call = time.perf_counter()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    thr1 = cgroup_throttled()
    throttled_ms = None if thr0 is None or thr1 is None else round((thr1[1] - thr0[1]) / 1e3, 3)
    trace("cgroup CPU quota: throttled %s times for %s ms inside the timed region" % (
        "?" if thr0 is None or thr1 is None else thr1[0] - thr0[0], throttled_ms))
    trace("timed region: the call %.3f ms, the synchronisation behind it %.3f ms; CPU user %.1f ms, system %.1f ms, %d minor page faults, "
          "%d + %d context switches" % (1e3 * (t_call - t0), 1e3 * (t1 - t_call), 1e3 * (ru1.ru_utime - ru0.ru_utime), 1e3 * (ru1.ru_stime - ru0.ru_stime),
                                       ru1.ru_minflt - ru0.ru_minflt, ru1.ru_nvcsw - ru0.ru_nvcsw, ru1.ru_nivcsw - ru0.ru_nivcsw))
    trace("timed region done: %.3f s" % elapsed)
    stats = pipe.stats()
    entropy_mode_timed = pipe.last_entropy_mode()   # (option "entropy_on_gpu": 0 host, the default; 1 GPU)
    ktimes, profiled = pipe.kernel_times()
    for c in prof_ctxs:
        c.set_profiling(False)
    own_elapsed = elapsed
    if dist is not None:
        dist.barrier()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    nbytes = res[0][0]

    # GPU side alone (kernels + device->host hand-over, no host entropy stage), same streams: the capacity
    # the host stage has to keep up with
    g_steps = max(4 * pipe.n_contexts, 64)
    gseq = [dev_frames[s % n_distinct] for s in range(g_steps)]
    pipe.gpu_stage_only(gseq[:pipe.n_contexts], [n_points] * pipe.n_contexts, params)
    torch.cuda.synchronize()
    tg = time.perf_counter()
    pipe.gpu_stage_only(gseq, [n_points] * g_steps, params)
    gpu_only_fps = g_steps / (time.perf_counter() - tg)

    total_points = n_points * args.steps * world
    value = total_points / elapsed / 1e6

    # ---- what every rank saw: the host entropy stage is what bounds a node with few CPUs per GPU, and a flat scaling curve
    #      has to be readable from the line itself (value per rank, what its GPU stage alone sustains, its share of the CPUs,
    #      where its entropy stage ran and what that stage can sustain there) ----
    cpus_here = default_workers(world)
    host_bound_fps = (cpus_here / (stats["entropy_cpu_us"] * 1e-6)) if stats["entropy_cpu_us"] > 0 and not entropy_mode_timed else None
    mine = {
        "rank": rank, "value": round(n_points * args.steps / own_elapsed / 1e6, 3), "ms_per_step": round(own_elapsed / args.steps * 1e3, 4),
        "gpu_only_mpoints_per_s": round(gpu_only_fps * n_points / 1e6, 1), "host_cpus_for_this_rank": cpus_here, "host_threads": pipe.workers,
        "entropy_stage": {"ran_on": "gpu" if entropy_mode_timed else "host",
                          "host_frames_per_s_bound": None if host_bound_fps is None else round(host_bound_fps, 0),
                          "gpu_stage_frames_per_s": round(gpu_only_fps, 0),
                          "host_cpu_ms_per_frame": round(stats["entropy_cpu_us"] / 1e3, 3)},
        # the host stage of this rank cannot keep up with what its GPU stage delivers: more GPUs on these CPUs add nothing
        "host_bound": bool(host_bound_fps is not None and host_bound_fps < gpu_only_fps),
    }
    mine.update(placement)
    ranks = [mine]
    if dist is not None:
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)

    # ---- kernel durations: serialised leg (one frame at a time on one stream, nothing else on the GPU) ----
    with_color = cfg["color_bits"] > 0
    path_bytes = 32 * n_points + L * (3 if with_color else 0) + B + 16 * L  # SURVEY.md 8(d), with output_ cloud
    roofline = None
    span_frame_ms, event_frame_ms, launches = {}, {}, {}
    pitch_ms, pitch_n = {}, {}
    path_ms = None
    if rank == 0:
        reps = 24 if n_points <= 2_000_000 else 8
        plain = []
        for k in range(reps + 4):           # no events between the kernels: the frame's kernel sequence as it runs
            ctx0.hotpath_launch(dev_frames[k % n_distinct], n_points, params)
            h = ctx0.hotpath_finish(copy=False)
            if k >= 4:
                plain.append(h.gpu_ms)
        path_ms = float(np.mean(plain))
        ctx0.set_profiling(True)
        for events in (0, 1):   # the launch spans alone (launches back to back, as unprofiled), then with HIP events in between
            ctx0.set_option("profile_events", events)
            for k in range(reps + 2):
                ctx0.hotpath_launch(dev_frames[k % n_distinct], n_points, params)
                ctx0.hotpath_finish(copy=False)
                if k < 2:
                    continue
                if events:
                    for name, ms in ctx0.kernel_times():
                        event_frame_ms[name] = event_frame_ms.get(name, 0.0) + ms / reps
                    continue
                for name, ms in ctx0.kernel_spans():
                    span_frame_ms[name] = span_frame_ms.get(name, 0.0) + ms / reps
                    launches[name] = launches.get(name, 0.0) + 1.0 / reps
                for name, ms in ctx0.kernel_pitches():
                    pitch_ms[name] = pitch_ms.get(name, 0.0) + ms
                    pitch_n[name] = pitch_n.get(name, 0) + 1
        ctx0.set_profiling(False)
    if span_frame_ms:
        dominant = max(span_frame_ms, key=span_frame_ms.get)
        # duration of a launch = from its start to the start of the next launch of the frame: the span of its workgroups plus
        # the dispatch and the end-of-kernel write-back, i.e. what `rocprofv3 --kernel-trace` calls the duration
        span_ms = span_frame_ms[dominant] / launches[dominant]
        dom_ms = pitch_ms[dominant] / pitch_n[dominant] if pitch_n.get(dominant) else span_ms
        dom_bytes = kernel_algorithmic_bytes(dominant, n_points, L, B, with_color, image_bytes)
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        sum_spans = float(sum(span_frame_ms.values()))
        # PMC counters need their own rocprofv3 passes (tools/pmc_run.sh): the figure comes from a file.  It counts as THIS line's
        # roofline.traffic only if it was taken on the kernel sources this library was built from (the file records their hash);
        # a figure of other sources -- round 2's, while no GPU session has refreshed it -- is reported beside it, labelled, and
        # traffic stays null: not a measurement of the run that prints it.
        traffic = traffic_taken = traffic_of_other_sources = None
        try:
            import hashlib
            with open(args.traffic_json or os.path.join(ROOT, "profiles", "hbm_traffic_%s.json" % args.workload)) as fh:
                tj = json.load(fh)
            with open(os.path.join(ROOT, "cwi-pcl-codec_amd", "csrc", "pcc_kernels.hip"), "rb") as fh:
                sha_now = hashlib.sha256(fh.read()).hexdigest()[:16]
            if tj.get("workload") == args.workload:
                figure = tj.get("kernels", {}).get(dominant, {}).get("hbm_bytes_per_launch")
                taken = tj.get("taken", "a separate rocprofv3 --pmc run (tools/pmc_run.sh)")
                if figure is not None and tj.get("kernels_sha16") == sha_now:
                    traffic, traffic_taken = figure, taken
                elif figure is not None:
                    traffic_of_other_sources = {"hbm_bytes_per_launch": figure, "taken": taken}
        except (OSError, ValueError):
            pass
        # inside the timed region several frames share the GPU: what the same kernel takes there (HIP events on ONE
        # context's stream, the other streams unprofiled) is reported for comparison, not as the kernel's duration
        profiled = max(1, profiled)
        shared = {k: v[0] / v[1] for k, v in ktimes.items() if v[1] and not k.startswith("begin")}
        roofline = {
            "bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBPS, 5), "traffic": traffic, "traffic_taken": traffic_taken,
            "traffic_of_other_sources": traffic_of_other_sources,
            "kernel_avg_ms": round(dom_ms, 5), "kernel_bytes_per_launch": int(dom_bytes),
            "kernel_launches_per_frame": round(launches[dominant], 2),
            "kernel_span_ms": round(span_ms, 5),
            "kernel_ms_per_frame": round(dom_ms * launches[dominant], 5),
            "kernel_avg_ms_hip_events": round(event_frame_ms.get(dominant, 0.0) / max(launches[dominant], 1e-9), 5),
            "how": "one frame at a time on one stream, GPU real-time clock: kernel_avg_ms = start of the launch to start of the next "
                   "launch of the frame (what a kernel trace reports as duration); kernel_span_ms = first workgroup start to last wave end",
            "path_bytes_per_frame": int(path_bytes), "path_gpu_ms": round(path_ms, 5),
            "path_sum_of_kernel_spans_ms": round(sum_spans, 5),
            "path_achieved": round(path_bytes / (path_ms * 1e-3) / 1e9, 2),
            "path_frac": round(path_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 5),
            "frames_in_flight_in_timed_region": pipe.get("gpu_threads"),
            "kernel_avg_ms_sharing_the_gpu": round(shared.get(dominant, 0.0), 5) if shared else None,
        }

    # ---- the same work starting from HOST memory (the reference's timed span starts there, eval.hpp:462-464) ----
    host_input = None
    if rank == 0 and not args.no_host_input:
        lib = b.load_library()
        hf = args.host_frames or min(args.steps, 256)
        pinned = [b.pinned_array(lib, f) for f in host_frames]
        legs = {}
        for label, src in (("pinned", pinned), ("pageable", host_frames)):
            seq_h = [src[s % n_distinct] for s in range(hf)]
            pipe.encode_host(seq_h[:min(hf, 2 * pipe.n_contexts)], params, copy=False)   # warm-up: every context gets its upload buffer
            torch.cuda.synchronize()
            th = time.perf_counter()
            pipe.encode_host(seq_h, params, copy=False)
            legs[label] = hf * n_points / (time.perf_counter() - th) / 1e6
        # the same with "pack_upload": the GPU-stage threads pack x, y, z, colour (16 of a point's 32 bytes) into page-locked
        # buffers and half the bytes cross the link -- pays where the host has memory bandwidth and cores to spare
        try:
            pipe.set_option("pack_upload", 1)
            seq_h = [host_frames[s % n_distinct] for s in range(hf)]
            pipe.encode_host(seq_h[:min(hf, 2 * pipe.n_contexts)], params, copy=False)
            torch.cuda.synchronize()
            th = time.perf_counter()
            pipe.encode_host(seq_h, params, copy=False)
            legs["packed"] = hf * n_points / (time.perf_counter() - th) / 1e6
        finally:
            pipe.set_option("pack_upload", 0)
        # one frame, one call, ordinary memory: what a caller of the reference's class gets per encodePointCloud
        lat = []
        for k in range(6):
            tl = time.perf_counter()
            ctx0.encode_intra_host(host_frames[k % n_distinct], params)
            lat.append(time.perf_counter() - tl)
        host_input = {
            "e2e_from_host_mpoints_per_s": round(legs["pinned"], 1),
            "e2e_from_pageable_host_mpoints_per_s": round(legs["pageable"], 1),
            "e2e_from_host_packed_16B_mpoints_per_s": round(legs["packed"], 1),   # option pack_upload (off by default)
            "frames": hf,
            "pcie_bound_mpoints_per_s": round(56.0e9 / (32 * 1e6), 0),   # 56 GB/s measured host->device (tools/ubench/h2d.cpp), 32 B per point
            "single_call_latency_ms": round(1e3 * float(np.median(lat[1:])), 3),
            "note": "pipelined: upload of frame k+1, kernels of frame k, host entropy stage of frame k-1 overlap; single call: "
                    "upload + kernels + the serial range coder of one frame (the coder alone is ~3 ms for ~1M occupancy symbols)",
        }
        for a in pinned:
            lib.pcc_host_free(a.ctypes.data)

    # ---- CPU baseline: the pointer-octree oracle, single thread, bounded sample ----
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as O
        po = O.make_params(**kw)
        tw = time.perf_counter()
        O.encode_intra(host_frames[0], po, keep=False)          # warm-up + cost probe
        one = time.perf_counter() - tw
        frames = args.cpu_frames or int(max(2, min(10, round(15.0 / max(one, 1e-3)))))
        tc = time.perf_counter()
        for f in range(frames):
            r = O.encode_intra(host_frames[f % n_distinct], po, keep=False)
        cpu_s = time.perf_counter() - tc
        assert r.n_leaves == L or n_distinct > 1
        cpu_baseline = {
            "value": round(n_points * frames / cpu_s / 1e6, 4), "unit": "Mpoints/s", "cores": 1, "kind": "port",
            "sample": "%d full encodes of the %s frame (%d points) by oracle/liboracle.so (gcc -O2, pointer octree + "
                      "JPEG + range coder, 1 thread), %.1f s" % (frames, args.workload, n_points, cpu_s),
            "host_cpus": os.cpu_count(),
        }
        try:  # the reference's own CMake hard-wires -g -O0 (CMakeLists.txt:85-87): the same port at -O0, for orientation
            t0 = time.perf_counter()
            O.encode_intra(host_frames[0], po, opt0=True, keep=False)
            O.encode_intra(host_frames[0], po, opt0=True, keep=False)
            cpu_baseline["value_at_O0"] = round(n_points * 2 / (time.perf_counter() - t0) / 1e6, 4)
        except Exception:
            pass

    if rank == 0:
        out = {
            "metric": ("encoded Mpoints/s, 1M-pt XYZRGB depth-10 intra" if args.workload in ("cfg2", "cfg2u") else
                       "encoded Mpoints/s, %s" % args.workload) + " (full encode to bitstream, input resident in HBM)",
            "value": round(value, 3), "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u64 keys / u8 streams (fp64 key quantisation)", "data": "synthetic",
            # the same encode with the cloud in HOST memory when the clock starts -- the span the reference's own clock brackets
            # (evaluate_compression_impl.hpp:462-464: cloud in host memory in, bitstream out): pipelined over the frames of a GOP
            # (page-locked source), and one frame through one blocking encodePointCloud-shaped call (ordinary memory).  `value`
            # above is the contract's figure (input resident in HBM); these two are the ones to hold against the reference.
            "value_from_host_memory": None if host_input is None else host_input["e2e_from_host_mpoints_per_s"],
            "single_call_ms": None if host_input is None else host_input["single_call_latency_ms"],
            "single_call_mpoints_per_s": None if host_input is None else round(n_points / host_input["single_call_latency_ms"] / 1e3, 1),
            "reference_timed_span": "host memory in, bitstream out, one call (eval.hpp:462-464) = single_call_ms; value_from_host_memory is "
                                    "the same span pipelined over a GOP; value starts with the frames resident in HBM",
            "config": {"workload": "%s: %d-point %s XYZRGB frame, octree_bits=%d, %s, intra-only" %
                                   (args.workload, n_points, "sphere-shell" if cfg["gen"] == "sphere" else "uniform-volume",
                                    cfg["octree_bits"],
                                    "colour JPEG snake q%d" % cfg["jpeg_quality"] if with_color else "geometry only"),
                       "frames_per_gpu": args.steps, "host_threads_per_gpu": pipe.workers,
                       "frames_per_coder_call": pipe.get("frames_per_coder_call"), "L": int(L), "B": int(B),
                       "depth": int(depth), "bitstream_bytes": int(nbytes),
                       "sharding": "frame f -> gpu f mod N, no collectives"},
            "gpu_only_mpoints_per_s": round(sum(r["gpu_only_mpoints_per_s"] for r in ranks), 1),   # all ranks, each its own GPU
            "host_ms_per_frame": {"launch_call": round(stats["launch_us"] / 1e3, 3),
                                  "finish_call": round(stats["finish_us"] / 1e3, 3),
                                  "entropy_call": round(stats["entropy_us"] / 1e3, 3),
                                  "occupancy_range_coder": round(stats["occupancy_coder_us"] / 1e3, 3),
                                  "jpeg_huffman": round(stats["jpeg_us"] / 1e3, 3),
                                  "colour_range_coder": round(stats["colour_coder_us"] / 1e3, 3)},
            "host_cpu_ms_per_frame": {"launch_call": round(stats["launch_cpu_us"] / 1e3, 3),
                                      "finish_call": round(stats["finish_cpu_us"] / 1e3, 3),
                                      "entropy_call": round(stats["entropy_cpu_us"] / 1e3, 3)},
            "roofline": roofline,
            "kernels_ms_per_frame": {k: round(v, 5) for k, v in sorted(span_frame_ms.items(), key=lambda kv: -kv[1])},
            "host_input": host_input,
            "host_cpus_for_this_rank": cpus_here,
            # what the host entropy stage can sustain on a rank's CPUs (CPU time per frame measured in the timed region) against
            # what its GPU stage delivers: rank 0's here, every rank's under "ranks"
            "entropy_stage": mine["entropy_stage"],
            "host_bound": any(r["host_bound"] for r in ranks),
            "ranks": ranks,
            # a call of few frames cannot be shorter than the frames one entropy thread codes one after the other
            "short_call_floor_ms": round(-(-args.steps // max(pipe.workers, 1)) * stats["entropy_us"] / 1e3, 3),
            "library": library,
            "entropy_coder": {"frames_per_coder_call": pipe.get("frames_per_coder_call"), "device_form": "lanes" if pipe.get("rc_device_lanes") else "waves"},
            "cgroup_throttled_ms_in_timed_region": throttled_ms,
            "warmup_frames_run": warm + args.steps * int(os.environ.get("PCC_BENCH_SHAPE_WARMUP", "1")),
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(out))
    pipe.close()
    if dist is not None:
        dist.barrier()   # rank 0 has a little more to do after the timed region: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
