#!/usr/bin/env python3
"""The rate when the caller hands over HOST buffers (pcc_encode_intra: H2D of the 32-byte points + GPU stage + host
stage, one frame at a time on one context) -- the PCIe-inclusive figure DESIGN.md quotes next to the bench's `value`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
B = pkg.binding
pts = pkg.synthetic.make_frame("cfg2")
prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
ctx = B.Context(0)
ctx.set_option("copy_image", 0)
for _ in range(3):
    ctx.encode_intra_host(pts, prm)
t0 = time.perf_counter()
K = 20
for _ in range(K):
    ctx.encode_intra_host(pts, prm)
dt = (time.perf_counter() - t0) / K
print("pcc_encode_intra from pageable host memory, one context, serial: %.2f ms per 1 M-point frame = %.0f Mpoints/s" % (dt * 1e3, len(pts) / dt / 1e6))
