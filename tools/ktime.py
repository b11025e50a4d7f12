#!/usr/bin/env python3
"""Developer tool: phase timing inside k_sort_pass (library built with -DPCC_KTIME)."""
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
grid = (len(pts) + 4095) // 4096
ntiles = (grid + (grid + 1023) // 1024 - 1) // ((grid + 1023) // 1024)   # every k-th workgroup of a large grid is sampled
for _ in range(5):
    ctx.hotpath_launch(dev, len(pts), p); hot = ctx.hotpath_finish(copy=False)
buf = np.zeros(9 * 1024 * 8, dtype=np.uint64)
lib = b.load_library()
print("rc", lib.pcc_debug_read_ktime(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.size)))
t = buf.reshape(9, 1024, 8).astype(np.int64)
for ps in range(hot.depth * 3 // 9 + 1 if wl != "cfg2" else 4):
    x = t[ps, :ntiles, :8]
    t0 = x[:, 0].min()
    rel = (x - t0) / 100.0
    m = np.median(rel, axis=0)  # slots: 0 start, 1 tile id + digit totals, 4 ranked, 2 tile word published + digit starts, 3 group closers done, 6 group mates / group before there, 7 look-back done, 5 end
    print("pass %d: stamps (median us): start %.2f, tile id + totals %.2f, ranked %.2f, published %.2f, closers done %.2f, mates / group before there %.2f, look-back done %.2f, end %.2f" %
          (ps, m[0], m[1], m[4], m[2], m[3], m[6], m[7], m[5]))
    print("        stamps (max us, slot order 0..7):", np.round(rel.max(axis=0), 2))
    d = x.astype(np.float64) / 100.0
    print("        per workgroup (median us): ticket+totals %.2f  keys+ranking %.2f  publish+digit starts %.2f  look-back (closers, waiting, reading) %.2f  write-out %.2f  whole %.2f" % (
        np.median(d[:, 1] - d[:, 0]), np.median(d[:, 4] - d[:, 1]), np.median(d[:, 2] - d[:, 4]), np.median(d[:, 7] - d[:, 2]),
        np.median(d[:, 5] - d[:, 7]), np.median(d[:, 5] - d[:, 0])))
    order = np.argsort(d[:, 0])
    q = max(1, len(order) // 4)
    for name, part in (("first quarter to start", order[:q]), ("last quarter to start", order[-q:])):
        e = d[part]
        print("          %-22s: start at %.1f us, look-back %.2f us, whole %.2f us" % (name, np.median(e[:, 0] - d[:, 0].min()), np.median(e[:, 7] - e[:, 4]), np.median(e[:, 5] - e[:, 0])))
    print("          launch: first start to last end %.1f us" % (d[:, 5].max() - d[:, 0].min()))
gl = len(pts) // 2048 + 1   # k_leaf_tile: one workgroup per block row of 2048 leaves
nlt = (gl + (gl + 1023) // 1024 - 1) // ((gl + 1023) // 1024)
x = t[5, :nlt, :7]
rel = (x - x[:, 0].min()) / 100.0
print("k_leaf_tile stamps (median us): start, A1 done, r0 colour, r0 centre+simplified, r0 occupancy, A2 done (4 rounds), end (stores issued):", np.round(np.median(rel, axis=0), 2))
print("            stamps (max us):", np.round(rel.max(axis=0), 2))
x = t[7, :nlt, :4]
rel = (x - t[5, :nlt, 0:1]) / 100.0
print("k_leaf_tile A1 (median us since kernel start): first barrier, records landed + masks, barrier, parent search done:", np.round(np.median(rel, axis=0), 2))
x8 = t[8, 0, :8]
x = t[6, 0, :7]
print("k_boxes_events replay rounds (us since start, after each block-wide minimum):", np.round((x8 - x[0]) / 100.0, 2))
nb = (len(pts) + 2047) // 2048 + 1
pub = (t[6, 1:nb, 7] - x[0]) / 100.0
print("k_boxes_events, streaming workgroups: box published (us since workgroup 0 started): min %.2f median %.2f max %.2f" % (pub.min(), np.median(pub), pub.max()))
print("k_boxes_events, workgroup 0, stamps (us): start, first point found, chunk 0 replayed, chunk boxes all there, sweep done, events done, end:", np.round((x - x[0]) / 100.0, 2))
ctx.close()
