#!/usr/bin/env python3
"""Time the product's host entropy stage (header + range coder + JPEG) on one oracle-produced frame. CPU only."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G
from oracle import oracle as O

pkg = G.load_package()
b = pkg.binding
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = pkg.synthetic.CONFIGS[wl]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n"]
pts = pkg.synthetic.make_frame(wl, n=n)
kw = dict(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
          jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
t = time.perf_counter(); r = O.encode_intra(pts, O.make_params(**kw)); t_or = time.perf_counter() - t
hr = b.HotResult()
for i in range(6): hr.bbox[i] = r.bbox[i]
hr.depth, hr.n_points_in, hr.n_leaves, hr.n_branches = r.depth, r.n_points_in, r.n_leaves, r.n_branches
keep = [np.ascontiguousarray(a) for a in (r.occupancy, r.bgr, r.centroid_bytes, r.snake_image)]
hr.occupancy = keep[0].ctypes.data
hr.bgr = keep[1].ctypes.data if keep[1].size else None
hr.centroid = keep[2].ctypes.data if keep[2].size else None
hr.image = keep[3].ctypes.data if keep[3].size else None
hr.image_w, hr.image_h = r.image_w, r.image_h
host = b.Context(None)
p = b.make_params(**kw)
for _ in range(2): host.entropy_encode(hr, p, copy=False)
t = time.perf_counter()
K = 5
for _ in range(K): nb, perf = host.entropy_encode(hr, p, copy=False)
dt = (time.perf_counter() - t) / K
stream, perf = host.entropy_encode(hr, p)
print("oracle full encode %.1f ms; product host stage %.2f ms; L=%d B=%d bytes=%d perf=%s match=%s" %
      (t_or * 1e3, dt * 1e3, r.n_leaves, r.n_branches, nb, perf, stream == r.bitstream))
# pieces
occ = keep[0].tobytes()
t = time.perf_counter(); e = b.host_range_encode(occ); t1 = time.perf_counter() - t
print("range encode of %d occupancy bytes: %.2f ms -> %d" % (len(occ), t1 * 1e3, len(e)))
if r.image_w:
    img = keep[3].reshape(r.image_h, r.image_w, 3)
    t = time.perf_counter(); j = b.host_jpeg_encode(img, cfg["jpeg_quality"]); t2 = time.perf_counter() - t
    print("jpeg encode from rgb %dx%d: %.2f ms -> %d" % (r.image_w, r.image_h, t2 * 1e3, len(j)))
    t = time.perf_counter(); e = b.host_range_encode(j); t3 = time.perf_counter() - t
    print("range encode of jpeg bytes: %.2f ms -> %d" % (t3 * 1e3, len(e)))
