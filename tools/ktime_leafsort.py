#!/usr/bin/env python3
"""Developer tool: phase timing inside k_leaf_sort (library built with -DPCC_KTIME: make -C cwi-pcl-codec_amd/csrc ktime;
PCC_LIB=cwi-pcl-codec_amd/libpcc_hip_ktime.so python tools/ktime_leafsort.py [workload])."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = syn.CONFIGS[wl]
p = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                  jpeg_quality=cfg["jpeg_quality"])
ctx = b.Context(0); pts = syn.make_frame(wl); dev = ctx.upload(pts)
for _ in range(5):
    ctx.hotpath_launch(dev, len(pts), p); hot = ctx.hotpath_finish(copy=False)
print("sort plan:", ctx.sort_plan())
buf = np.zeros(9 * 1024 * 8, dtype=np.uint64)
lib = b.load_library()
print("rc", lib.pcc_debug_read_ktime(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.size)))
t = buf.reshape(9, 1024, 8).astype(np.int64)
grid = (len(pts) + 2047) // 2048
ks = (grid + 1023) // 1024
nt = (grid + ks - 1) // ks
d = t[4, :nt, :].astype(np.float64) / 100.0
ok = d[:, 5] > 0
d = d[ok]
names = ["ticket + ends of the tile", "keys into LDS", "bucket ranks", "local radix passes", "run written", "leaf scan + look-back", "leaf arrays + zeroing"]
order = [0, 1, 2, 6, 7, 3, 4, 5]
print("k_leaf_sort, %d workgroups sampled; per workgroup, median us (max):" % len(d))
for k in range(len(order) - 1):
    x = d[:, order[k + 1]] - d[:, order[k]]
    print("   %-28s %6.2f  (%.2f)" % (names[k], np.median(x), x.max()))
print("   whole workgroup              %6.2f  (%.2f);  first start to last end %.1f us; starts spread over %.1f us" % (
    np.median(d[:, 5] - d[:, 0]), (d[:, 5] - d[:, 0]).max(), d[:, 5].max() - d[:, 0].min(), d[:, 0].max() - d[:, 0].min()))
q = t[3, :nt, :].astype(np.float64) / 100.0
q = q[ok]
print("   first local pass: ranking %.2f, barrier %.2f, digit starts %.2f, scatter + barrier %.2f, reload %.2f" % tuple(
    np.median(q[:, k + 1] - q[:, k]) for k in range(5)))
ctx.close()
