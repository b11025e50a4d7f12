#!/usr/bin/env python3
"""Speed of the GPU range coder (one wave per stream) on the occupancy stream of the cfg2 frame: latency of one stream,
aggregate streams/s with many streams side by side.  Run on the GPU box."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    pkg = G.load_package()
    B = pkg.binding
    pts = pkg.synthetic.make_frame("cfg2")
    prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
    ctx = B.Context(0)
    dev = ctx.upload(pts)
    ctx.hotpath_launch(dev, len(pts), prm)
    hot = ctx.hotpath_finish()
    occ = hot.occupancy.tobytes()
    want = B.host_range_encode(occ)
    for k in (1, 16, 64, 256, 1024, 2048):
        best = 1e9
        for _ in range(2):
            got, ms = ctx.device_range_encode([occ] * k)
            best = min(best, ms)
        assert got[0] == want and got[-1] == want
        print("%5d streams of %d symbols: kernel %.2f ms -> %.1f ns per symbol per stream, %.0f streams/s" % (
            k, len(occ), best, best * 1e6 / len(occ), k / best * 1e3))


if __name__ == "__main__":
    main()
