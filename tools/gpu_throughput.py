#!/usr/bin/env python3
"""GPU-side capacity: frames/s of the hot path alone (launch + finish, no host entropy stage) with
several contexts (streams) in flight on one GPU.   python tools/gpu_throughput.py [workload] [threads...]"""
import os, sys, time, threading
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
sys.path.insert(0, os.path.join(ROOT, "tools"))
from frame_digests import check_frame
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
threads = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8, 16]
cfg = syn.CONFIGS[wl]
p = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                  jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
pts = syn.make_frame(wl); n = len(pts)
for T in threads:
    ctxs = [b.Context(0) for _ in range(T)]
    dev = ctxs[0].upload(pts)
    K = 40 * T
    for c in ctxs:   # parity line: every context's frame with host copies and bitstream against the oracle's digests
        c.hotpath_launch(dev, n, p); h = c.hotpath_finish(); s, _ = c.entropy_encode(h.raw, p)
        checked = check_frame(wl, 0, h, s)
    print("parity: %s on %d contexts %s" % (wl, T, "matches the oracle's golden digests" if checked else "NOT CHECKED (no digest)"), flush=True)
    for c in ctxs: c.set_option("copy_image", 0)
    def work(i):
        c = ctxs[i]
        for _ in range(K // T):
            c.hotpath_launch(dev, n, p)
            check_frame(wl, 0, c.hotpath_finish(copy=False))   # L, B, D, bounding box of every timed frame
    with ThreadPoolExecutor(T) as ex:
        list(ex.map(work, range(T)))            # warm-up
        t = time.perf_counter(); list(ex.map(work, range(T))); dt = time.perf_counter() - t
    print("%s: %2d streams: %.1f frames/s  (%.1f us per frame, %.0f Mpoints/s)" % (wl, T, K / dt, dt / K * 1e6, K * n / dt / 1e6))
    for c in ctxs: c.close()
