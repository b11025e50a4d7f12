#!/usr/bin/env python3
"""Throughput of the inter-frame path when several contexts work at once (contexts share nothing, so the P frames of a
sequence -- each depends only on the intra coding of the frame before, never on another P frame -- can be coded side by
side): cfg5 pairs, T host threads with a context each.  Run on the GPU box:  python tools/delta_throughput.py [n] [T...]"""
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    pkg = G.load_package()
    B = pkg.binding
    cfg = pkg.synthetic.CONFIGS["cfg5"]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n"]
    threads = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
    frames = pkg.synthetic.moving_sphere_group(n, cfg["seed"], 9)
    prm = B.make_params(octree_bits=cfg["octree_bits"], color_bits=8, color_coding_type=1, jpeg_quality=85)
    c0 = B.Context(0)
    pairs = []
    for f in range(8):
        c0.encode_intra_host(frames[f], prm)
        pairs.append((c0.output_cloud(), frames[f + 1]))
    for T in threads:
        ctxs = [B.Context(0) for _ in range(T)]
        for c in ctxs:
            c.encode_delta(pairs[0][0], pairs[0][1], prm, write_out_cloud=False)   # warm-up: allocations
        reps = 24

        def work(k):
            for r in range(reps):
                i_cloud, p_cloud = pairs[(k + r) % len(pairs)]
                ctxs[k].encode_delta(i_cloud, p_cloud, prm, write_out_cloud=False)
        th = [threading.Thread(target=work, args=(k,)) for k in range(T)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
        print("%d context(s): %.2f ms per P frame per context, %.0f P frames/s (%.1f Mpoints/s of input)" % (
            T, dt / reps * 1e3, T * reps / dt, T * reps * n / dt / 1e6))
        for c in ctxs:
            c.close()


if __name__ == "__main__":
    main()
