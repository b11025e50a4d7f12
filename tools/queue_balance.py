#!/usr/bin/env python3
"""Does the saturated GPU stage depend on how the streams in flight are spread over the runtime's hardware queues
(GPU_MAX_HW_QUEUES = 4 by default; a stream gets the least-used queue when it is created)?  16 contexts are created once
(stream k -> queue k mod 4 if the assignment is round robin), then subsets of them run frames side by side.
    python tools/queue_balance.py"""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
cfg = syn.CONFIGS["cfg2"]
p = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                  jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
pts = syn.make_frame("cfg2"); n = len(pts)
ctxs = [b.Context(0) for _ in range(16)]
for c in ctxs: c.set_option("copy_image", 0)
dev = ctxs[0].upload(pts)
sets = {
    "8 balanced (2 2 2 2)": [0, 1, 2, 3, 4, 5, 6, 7],
    "8 on two queues (4 4 0 0)": [0, 4, 8, 12, 1, 5, 9, 13],
    "10 (3 3 2 2)": list(range(10)),
    "10 (4 4 2 0)": [0, 4, 8, 12, 1, 5, 9, 13, 2, 6],
    "12 balanced (3 3 3 3)": list(range(12)),
    "12 (4 4 4 0)": [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14],
    "16 balanced": list(range(16)),
}
for rep in range(2):
    for name, ids in sets.items():
        T = len(ids); per = 60
        def work(i):
            c = ctxs[ids[i]]
            for _ in range(per):
                c.hotpath_launch(dev, n, p); c.hotpath_finish(copy=False)
        with ThreadPoolExecutor(T) as ex:
            list(ex.map(work, range(T)))
            t = time.perf_counter(); list(ex.map(work, range(T))); dt = time.perf_counter() - t
        print("%-28s %8.1f frames/s" % (name, T * per / dt), flush=True)
