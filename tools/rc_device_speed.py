#!/usr/bin/env python3
"""Speed of the GPU range coder on the occupancy stream of the cfg2 frame: latency of one stream, aggregate streams/s with many
streams side by side -- for both forms: one wave per stream (default) and one lane per stream (option "rc_device_lanes").
Run on the GPU box.  RC_SPEED_BIG=1 adds 4096 streams (4 GB of coded streams on the host)."""
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
    sizes = (1, 16, 64, 256, 1024, 2048) + ((4096,) if os.environ.get("RC_SPEED_BIG") == "1" else ())
    for lanes in (0, 1):
        ctx.set_option("rc_device_lanes", lanes)
        print("== one %s per stream" % ("LANE" if lanes else "wave"))
        for k in sizes:
            best = 1e9
            for _ in range(2):
                got, ms = ctx.device_range_encode([occ] * k)
                best = min(best, ms)
            assert got[0] == want and got[-1] == want and all(len(g) == len(want) for g in got)
            print("%5d streams of %d symbols: kernel %.2f ms -> %.1f ns per symbol per stream, %.0f streams/s, %d waves" % (
                k, len(occ), best, best * 1e6 / len(occ), k / best * 1e3, (k + 63) // 64 if lanes else k))
            del got
    ctx.set_option("rc_device_lanes", 0)


if __name__ == "__main__":
    main()
