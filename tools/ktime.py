#!/usr/bin/env python3
"""Developer tool: phase timing inside k_sort_pass (library built with -DPCC_KTIME)."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
cfg = syn.CONFIGS["cfg2"]
p = b.make_params(octree_bits=cfg["octree_bits"])
ctx = b.Context(0); pts = syn.make_frame("cfg2"); dev = ctx.upload(pts)
for _ in range(5):
    ctx.hotpath_launch(dev, len(pts), p); hot = ctx.hotpath_finish(copy=False)
buf = np.zeros(9 * 1024 * 8, dtype=np.uint64)
lib = b.load_library()
print("rc", lib.pcc_debug_read_ktime(C.c_void_p(buf.ctypes.data), C.c_size_t(buf.size)))
t = buf.reshape(9, 1024, 8).astype(np.int64)
for ps in range(4):
    x = t[ps, :245, :8]
    t0 = x[:, 0].min()
    rel = (x - t0) / 100.0
    print("pass %d: stamps (median us): start %.2f tile %.2f published %.2f closers-done %.2f ranked %.2f mates-read %.2f prev-group-ready %.2f lookback-done %.2f" % ((ps,) + tuple(np.median(rel, axis=0))))
    print("        stamps (max us):", np.round(rel.max(axis=0), 2))
x = t[5, :231, :7]
rel = (x - x[:, 0].min()) / 100.0
print("k_leaf_tile stamps (median us): start, A1 done, r0 colour, r0 centre+simplified, r0 occupancy, A2 done (4 rounds), end:", np.round(np.median(rel, axis=0), 2))
print("            stamps (max us):", np.round(rel.max(axis=0), 2))
x = t[7, :231, :4]
rel = (x - t[5, :231, 0:1]) / 100.0
print("k_leaf_tile A1 (median us since kernel start): first barrier, records landed + masks, barrier, parent search done:", np.round(np.median(rel, axis=0), 2))
x8 = t[8, 0, :8]
x = t[6, 0, :7]
print("k_boxes_events replay rounds (us since start, after each block-wide minimum):", np.round((x8 - x[0]) / 100.0, 2))
print("k_boxes_events, workgroup 0, stamps (us): start, first point found, chunk 0 replayed, chunk boxes all there, sweep done, events done, end:", np.round((x - x[0]) / 100.0, 2))
ctx.close()
