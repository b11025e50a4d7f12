#!/usr/bin/env python3
"""Developer tool: colour coding type 2 on the GPU (k_jpeg_lines) against the host JPEG coder, strip by strip."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as G
pkg = G.load_package(); b = pkg.binding; lib = b.load_library()
L = int(sys.argv[1]) if len(sys.argv) > 1 else 33000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rng = np.random.default_rng(L)
side = int(np.ceil(np.sqrt(L)))
ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:L]
xyz = np.concatenate([(2 * ij + 0.5) / 256.0 + 0.1, np.full((L, 1), 0.5)], 1).astype(np.float32)
pts = np.zeros(L, dtype=b.POINT_DTYPE)
pts["x"], pts["y"], pts["z"], pts["w"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], 1.0
rgb = rng.integers(0, 256, (L, 3)).astype(np.uint32)
pts["rgba"] = rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(0xFF000000)
prm = b.make_params(octree_resolution=1 / 256.0, point_resolution=1 / 256.0, color_coding_type=2, jpeg_quality=q)
c = b.Context(0)
dev = c.upload(pts)
c.hotpath_launch(dev, L, prm); hot = c.hotpath_finish()
n = hot.raw.jpeg_n_lines
d = np.frombuffer(C.string_at(hot.raw.jpeg_lines_dir, 16 * n), dtype=np.uint32).reshape(n, 4)
print("lines", n, "L", hot.n_leaves)
bgr = hot.bgr.reshape(-1, 3)
lib.pcc_host_jpeg_encode.restype = C.c_size_t
lib.pcc_host_jpeg_encode.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
for i in range(n):
    off, bits, width, ovf = [int(x) for x in d[i]]
    words = np.frombuffer(C.string_at(hot.raw.jpeg_lines_data + 4 * off, 4 * ((bits + 31) // 32)), dtype=np.uint32)
    got = np.unpackbits(words.astype(">u4").view(np.uint8))[:bits]
    px = np.ascontiguousarray(bgr[2048 * i: 2048 * i + width])
    out = np.zeros(3 * width + 8192, np.uint8)
    m = lib.pcc_host_jpeg_encode(px.ctypes.data, width, 1, q, out.ctypes.data, len(out))
    j = bytes(out[:m])
    sos = j.index(b"\xff\xda")
    ent = j[sos + 2 + int.from_bytes(j[sos + 2:sos + 4], "big"):-2].replace(b"\xff\x00", b"\xff")
    want = np.unpackbits(np.frombuffer(ent, np.uint8))
    k = min(len(got), len(want))
    diff = np.flatnonzero(got[:k] != want[:k])
    ok = len(diff) == 0 and len(want) - len(got) < 8
    print("line %2d width %4d bits %6d (host %6d) offset %7d  %s" % (i, width, bits, len(want), off, "same" if ok else "DIFFERS at bit %d" % (diff[0] if len(diff) else k)))
