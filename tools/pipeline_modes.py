#!/usr/bin/env python3
"""Per-frame wall and CPU times of the pipeline's calls in GPU-stage-only mode and in full mode (cfg2).  GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    pkg = G.load_package()
    B = pkg.binding
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    pts = [pkg.synthetic.make_frame("cfg2", frame=f) for f in range(4)]
    prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
    pipe = B.Pipeline(0, workers)
    for w in range(pipe.n_contexts):
        pipe.context(w).set_option("copy_image", 0)
    dev = [pipe.context(0).upload(p) for p in pts]
    n = 1024
    seq, cnt = [dev[i % 4] for i in range(n)], [len(pts[0])] * n
    w = pipe.encode(seq[:128], cnt[:128], prm, copy=False)
    if len(sys.argv) > 2:
        pipe.reserve(n, max(r[0] for r in w), len(pts[0]))
    for mode in ("gpu", "full", "gpu", "full"):
        t0 = time.perf_counter()
        if mode == "gpu":
            pipe.gpu_stage_only(seq, cnt, prm)
        else:
            pipe.encode(seq, cnt, prm, copy=False)
        dt = time.perf_counter() - t0
        s = pipe.stats()
        print("%-4s %6.0f frames/s  launch %.3f (cpu %.3f)  finish %.3f (cpu %.3f)  entropy %.3f (cpu %.3f) ms" % (
            mode, n / dt, s["launch_us"] / 1e3, s["launch_cpu_us"] / 1e3, s["finish_us"] / 1e3, s["finish_cpu_us"] / 1e3,
            s["entropy_us"] / 1e3, s["entropy_cpu_us"] / 1e3))


if __name__ == "__main__":
    main()
