import sys, time; sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b = pkg.binding
ctx = b.Context(0)
pts = pkg.synthetic.make_frame("cfg2")
s, _ = ctx.encode_intra_host(pts, b.make_params(octree_bits=10, jpeg_quality=85))
dec, _ = ctx.decode_intra(s)
for hint in (2.0 ** -10, 0.0):
    for _ in range(3):
        t = time.perf_counter(); m = ctx.quality_metrics(pts, dec, cell_hint=hint); dt = time.perf_counter() - t
    print("cell hint %g: gpu %.3f ms, call %.1f ms; symm_rms %.6g psnr %.3f psnr_yuv %s" % (hint, m["gpu_ms"], dt * 1e3, m["symm_rms"], m["psnr_db"], [round(x, 3) for x in m["psnr_yuv"]]))
