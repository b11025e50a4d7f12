#!/usr/bin/env python3
"""Single-context, one-frame-at-a-time GPU latency of the hot path (no host entropy stage, no
concurrency): HIP-event time from the first to the last kernel, and per-kernel event times.
    python tools/gpu_latency.py [workload] [frames] [distinct frames]
With `distinct frames` > 1 that many different frames (frame 0, 1, ...) sit in HBM and take turns: twelve 1 M-point frames
are 384 MB, more than the 256 MB Infinity Cache, so that a PMC traffic figure taken over this tool is HBM traffic and not
cache hits (one frame alone is re-read from the Infinity Cache from the second iteration on)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as G
sys.path.insert(0, os.path.join(ROOT, "tools"))
from frame_digests import check_frame

pkg = G.load_package()
b, syn = pkg.binding, pkg.synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
DISTINCT = int(sys.argv[3]) if len(sys.argv) > 3 else 1
cfg = syn.CONFIGS[wl]
p = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                  jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
ctx = b.Context(0)
frames = [syn.make_frame(wl, frame=f) for f in range(DISTINCT)]
devs = [ctx.upload(f) for f in frames]
pts, dev = frames[0], devs[0]
n = len(pts)
# parity line first: the frame that is about to be timed, with host copies and its bitstream, against the oracle's digests
ctx.hotpath_launch(dev, n, p)
_hot = ctx.hotpath_finish()
_stream, _ = ctx.entropy_encode(_hot.raw, p)
print("parity: %s frame 0 %s" % (wl, "matches the oracle's golden digests (bbox, occupancy, colours, bitstream)"
                                  if check_frame(wl, 0, _hot, _stream) else "NOT CHECKED (no digest)"), flush=True)
ctx.set_option("copy_image", 0)   # (the timed frames leave image and colours on the device)
for prof in (0, 1, 2):   # 0 unprofiled, 1 HIP events between the launches + spans, 2 spans only (launches back to back)
    ctx.set_profiling(prof != 0)
    ctx.set_option("profile_events", 1 if prof == 1 else 0)
    ms, wall = [], []
    agg, spans = {}, {}
    for k in range(K + 3):
        t = time.perf_counter()
        ctx.hotpath_launch(devs[k % DISTINCT], len(frames[k % DISTINCT]), p)
        hot = ctx.hotpath_finish(copy=False)
        w = time.perf_counter() - t
        check_frame(wl, k % DISTINCT, hot)   # L, B, D and the bounding box of EVERY timed frame (no host copies in the timed loop)
        if k >= 3:
            ms.append(hot.gpu_ms); wall.append(w * 1e3)
            if prof:
                for name, v in (ctx.kernel_times() if prof == 1 else ctx.kernel_pitches()):
                    e = agg.setdefault(name, [0.0, 0]); e[0] += v; e[1] += 1
                for name, v in ctx.kernel_spans():
                    e = spans.setdefault(name, [0.0, 0]); e[0] += v; e[1] += 1
    print("%s N=%d L=%d B=%d D=%d  profiling=%d: gpu %.1f us (min %.1f)  launch+finish wall %.1f us" %
          (wl, n, hot.n_leaves, hot.n_branches, hot.depth, prof, 1e3 * np.mean(ms), 1e3 * np.min(ms), 1e3 * np.mean(wall)))
    if prof:
        print("   (%s)" % ("first column: HIP events between the launches" if prof == 1 else "first column: start of a launch to start of the next one, GPU clock"))
        for name, (tot, cnt) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
            sp = spans.get(name, [0.0, 1])
            print("   %-22s %5.1f launches/frame  %8.2f us each  %8.2f us/frame   on the GPU clock: %8.2f us each  %8.2f us/frame" %
                  (name, cnt / K, 1e3 * tot / cnt, 1e3 * tot / K, 1e3 * sp[0] / max(1, sp[1]), 1e3 * sp[0] / K))
        print("   sum of the launch spans on the GPU clock: %.1f us/frame" % (1e3 * sum(v[0] for v in spans.values()) / K))
ctx.close()
