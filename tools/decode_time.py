#!/usr/bin/env python3
"""decodePointCloud of the headline frame: host decoder against the decoder with its data-parallel half on the GPU.
    python tools/decode_time.py [workload]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b = pkg.binding; lib = b.load_library()
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = pkg.synthetic.CONFIGS[wl]
prm = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                    jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"], frame_id=1)
ctx = b.Context(0)
stream, _ = ctx.encode_intra_host(pkg.synthetic.make_frame(wl), prm)
buf = np.frombuffer(stream, np.uint8)
c = b.Cloud()
for name, fn in (("host", lib.pcc_decode_intra), ("gpu ", lib.pcc_decode_intra_gpu)):
    best = 1e9
    for k in range(5):
        t = time.perf_counter(); rc = fn(ctx.h, buf.ctypes.data, len(buf), C.byref(c)); best = min(best, time.perf_counter() - t)
    extra = ""
    if name == "gpu ":
        extra = "  (sequential host stages %.2f ms, upload + kernels + download %.2f ms)" % (ctx.decode_times()["host_sequential_ms"], ctx.decode_times()["gpu_ms"])
    print("%s %s: %d voxels from %d bytes in %.2f ms%s" % (wl, name, c.n, len(buf), best * 1e3, extra))
