#!/usr/bin/env python3
"""Short calls of the frame pipeline (the driver's bench uses --steps 20): frames/s and per-call times for a few call
lengths, repeated.  PCC_PIPELINE_BATCH caps the frames per coder loop.   python tools/short_calls.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("SHORT_CALLS_WITH_TORCH"):  # does the process behave differently with torch (its threads, its HIP context) loaded?
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import __graft_entry__ as G
pkg = G.load_package(); B = pkg.binding
ns = [int(x) for x in sys.argv[1:]] or [20, 64, 256]
pts = [pkg.synthetic.make_frame("cfg2", frame=f) for f in range(4)]
prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
pipe = B.Pipeline(0, 16)
for w in range(pipe.n_contexts):
    pipe.context(w).set_option("copy_image", 0)
dev = [pipe.context(0).upload(p) for p in pts]
w = pipe.encode([dev[i % 4] for i in range(pipe.n_contexts)], [len(pts[0])] * pipe.n_contexts, prm, copy=False)
pipe.reserve(max(ns), max(r[0] for r in w), len(pts[0]))
for n in ns:
    for rep in range(4):
        t0 = time.perf_counter()
        pipe.encode([dev[i % 4] for i in range(n)], [len(pts[0])] * n, prm, copy=False)
        dt = time.perf_counter() - t0
        s = pipe.stats()
        print("batch<=%s  %4d frames: %7.2f ms  %6.0f frames/s   launch %.3f  finish %.3f  entropy %.3f ms per frame" % (
            os.environ.get("PCC_PIPELINE_BATCH", "4"), n, dt * 1e3, n / dt, s["launch_us"] / 1e3, s["finish_us"] / 1e3, s["entropy_us"] / 1e3))
pipe.close()
