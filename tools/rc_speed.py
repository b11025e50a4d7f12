#!/usr/bin/env python3
"""Speed of the host entropy stage alone on this machine: 1, 2, 4 frames per call on one thread, then T threads of
4-frame calls at once (ns per coded symbol).  Frames come from the GPU hot path (cfg2).  Run on the GPU box."""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    pkg = G.load_package()
    B = pkg.binding
    pts = pkg.synthetic.make_frame("cfg2")
    prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
    g = B.Context(0)
    dev = g.upload(pts)
    g.hotpath_launch(dev, len(pts), prm)
    hot = g.hotpath_finish()
    symbols = hot.n_branches
    hosts = [B.Context(None) for _ in range(64)]
    for k in (1, 2, 4):
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            B.Context.entropy_encode_many(hosts[:k], [hot.raw] * k, [prm] * k)
            best = min(best, time.perf_counter() - t0)
        print("%d frame(s) per call, 1 thread: %.3f ms per frame, %.2f ns per occupancy symbol (whole stage)" % (k, best * 1e3 / k, best * 1e9 / k / symbols))
    for T in (4, 8, 16):
        reps = 6
        def work(i):
            for _ in range(reps):
                B.Context.entropy_encode_many(hosts[4 * i:4 * i + 4], [hot.raw] * 4, [prm] * 4)
        th = [threading.Thread(target=work, args=(i,)) for i in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print("%2d threads x 4 frames: %.3f ms of thread time per frame, %.0f frames/s" % (T, dt * 1e3 / (reps * 4), T * reps * 4 / dt))


if __name__ == "__main__":
    main()
