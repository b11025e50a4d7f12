#!/usr/bin/env python3
"""The host decoder alone (no GPU needed): pcc_decode_intra on a host-only context, for a bitstream made by the oracle.
    python tools/decode_time_host.py [workload] [repetitions]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b = pkg.binding; lib = b.load_library()
from oracle import oracle as O
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
cfg = pkg.synthetic.CONFIGS[wl]
pts = pkg.synthetic.make_frame(wl)
prm = O.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                    jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
t = time.perf_counter(); enc = O.encode_intra(pts, prm, keep=False); t_enc = time.perf_counter() - t
stream = enc.bitstream if hasattr(enc, "bitstream") else enc.stream
buf = np.frombuffer(stream, np.uint8)
ctx = b.Context(None)
c = b.Cloud()
best = 1e9
for k in range(reps):
    t = time.perf_counter(); rc = lib.pcc_decode_intra(ctx.h, buf.ctypes.data, len(buf), C.byref(c)); best = min(best, time.perf_counter() - t)
    assert rc == 0, rc
print("%s: oracle encode %.0f ms; host decoder: %d voxels from %d bytes in %.2f ms (best of %d)" % (wl, t_enc * 1e3, c.n, len(buf), best * 1e3, reps))
