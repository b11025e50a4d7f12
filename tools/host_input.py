#!/usr/bin/env python3
"""Host-input pipeline (pcc_pipeline_encode_host): frames/s from pinned and from pageable host memory against the
device-resident pipeline, for a few GPU-stage thread counts (PCC_PIPELINE_GPU_THREADS).  GPU box.
    python tools/host_input.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G
pkg = G.load_package(); B = pkg.binding
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
pts = [pkg.synthetic.make_frame("cfg2", frame=f) for f in range(4)]
prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
lib = B.load_library()
pin = [B.pinned_array(lib, p) for p in pts]
for gthreads in (os.environ.get("PCC_PIPELINE_GPU_THREADS", "6"),):
    pipe = B.Pipeline(0, 16)
    for w in range(pipe.n_contexts):
        pipe.context(w).set_option("copy_image", 0)
    dev = [pipe.context(0).upload(p) for p in pts]
    w = pipe.encode([dev[i % 4] for i in range(64)], [len(pts[0])] * 64, prm, copy=False)
    pipe.reserve(n, max(r[0] for r in w), len(pts[0]))
    for label in ("device", "pinned", "pageable", "pinned", "pageable"):
        t0 = time.perf_counter()
        if label == "device":
            pipe.encode([dev[i % 4] for i in range(n)], [len(pts[0])] * n, prm, copy=False)
        else:
            src = pin if label == "pinned" else pts
            pipe.encode_host([src[i % 4] for i in range(n)], prm, copy=False)
        dt = time.perf_counter() - t0
        s = pipe.stats()
        print("gpu threads %s  %-8s %6.0f frames/s   launch %.3f ms (cpu %.3f)  finish %.3f ms (cpu %.3f)  entropy %.3f ms" %
              (gthreads, label, n / dt, s["launch_us"] / 1e3, s["launch_cpu_us"] / 1e3, s["finish_us"] / 1e3, s["finish_cpu_us"] / 1e3, s["entropy_us"] / 1e3))
    pipe.close()
