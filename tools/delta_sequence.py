#!/usr/bin/env python3
"""SURVEY.md 8(d) cfg5 as the reference app runs it with do_delta_coding=1 (eval.hpp:818-890): every frame is intra
coded, and every frame but the first is also coded as a P frame against the simplified cloud of the frame before.
Reports per-frame times and sizes through the codec class interface.  Run on the GPU box."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    pkg = G.load_package()
    B = pkg.binding
    cfg = pkg.synthetic.CONFIGS["cfg5"]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else cfg["n"]
    frames = pkg.synthetic.moving_sphere_group(n, cfg["seed"], cfg["frames"])
    res = 2.0 ** -cfg["octree_bits"]
    codec = B.OctreePointCloudCodecV2(B.MANUAL_CONFIGURATION, False, res, res, True, 0, True, 8, 1, False, False, False, 85)
    codec.setMacroblockSize(cfg["macroblock_size"])
    t_i = t_p = 0.0
    bytes_i = bytes_pi = bytes_pp = 0
    shared = conv = 0.0
    prev = None
    for f, cloud in enumerate(frames):
        if prev is not None:
            t0 = time.perf_counter()
            _, i_data, p_data = codec.encodePointCloudDeltaFrame(prev, cloud, False, False)
            t_p += time.perf_counter() - t0
            bytes_pi += len(i_data); bytes_pp += len(p_data)
            shared += codec.getMacroBlockPercentage(); conv += codec.getMacroBlockConvergencePercentage()
        t0 = time.perf_counter()
        stream = codec.encodePointCloud(cloud)
        t_i += time.perf_counter() - t0
        bytes_i += len(stream)
        prev = codec.getOutputCloud()
    k = len(frames)
    print("cfg5: %d frames of %d points, octree_bits=%d, macroblock %d" % (k, n, cfg["octree_bits"], cfg["macroblock_size"]))
    print("intra: %.2f ms per frame (host buffers in, bitstream out), %.1f KB per frame" % (t_i / k * 1e3, bytes_i / k / 1e3))
    print("delta: %.2f ms per P frame, %.1f KB per frame (%.1f intra part + %.1f predicted part); shared blocks %.3f, converged %.3f" % (
        t_p / (k - 1) * 1e3, (bytes_pi + bytes_pp) / (k - 1) / 1e3, bytes_pi / (k - 1) / 1e3, bytes_pp / (k - 1) / 1e3,
        shared / (k - 1), conv / (k - 1)))


if __name__ == "__main__":
    main()
