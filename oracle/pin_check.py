#!/usr/bin/env python3
"""pin_check.py -- compare the CPU oracle with the REAL reference codec (oracle/_ref/libcodec_ref.so, built by
oracle/pin_with_pcl.sh where PCL is installed) on the clouds of tests/golden/make_codec_golden.py plus seeded
larger clouds: complete bitstream, performance counters, simplified cloud, decoded cloud -- byte for byte.

Writes oracle/_ref/pin_report.json ({"pinned": true/false, "cases": {...}}).  With --regenerate the golden file
tests/golden/codec_golden.npz is rewritten from the REFERENCE's outputs (so the fixtures become vectors of PCL
itself, no longer of the restatement) -- only if every case agreed.

Test infrastructure: nothing in the product imports this."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import oracle as O          # noqa: E402
import make_codec_golden as G           # noqa: E402


def load(path):
    lib = C.CDLL(path)
    lib.ref_encode.restype = C.c_size_t
    lib.ref_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint,
                               C.POINTER(C.c_uint64)]
    lib.ref_stream.restype = C.c_void_p
    lib.ref_cloud.restype = C.c_void_p
    lib.ref_cloud_size.restype = C.c_size_t
    lib.ref_decode.restype = C.c_size_t
    lib.ref_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
    return lib


def ref_cloud(lib):
    n = lib.ref_cloud_size()
    return np.frombuffer(C.string_at(lib.ref_cloud(), 32 * n), dtype=O.POINT_DTYPE).copy() if n else np.zeros(0, dtype=O.POINT_DTYPE)


def extra_cases():
    """Larger seeded clouds: more growth events, the range coder's table rescale (> 65535 symbols), every colour mode."""
    rng = np.random.default_rng(424242)
    out = []
    for k, (n, bits, mode, cen) in enumerate([(60_000, 8, 1, 0), (90_000, 9, 0, 1), (120_000, 10, 2, 0), (70_000, 7, 3, 1)]):
        u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        xyz = np.stack([0.15 + 0.7 * u, 0.15 + 0.7 * v, 0.5 + 0.25 * np.sin(6 * u) * np.cos(5 * v)], 1)
        col = np.clip(np.stack([255 * u, 255 * v, 128 + 100 * np.sin(9 * u)], 1) + rng.integers(-8, 9, (n, 3)), 0, 255)
        out.append(("large_%d" % k, G.cloud(xyz, col), dict(octree_bits=bits, color_coding_type=mode, keep_centroid=cen, jpeg_quality=85, frame_id=1 + k)))
    return out


def main():
    lib = load(sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else os.path.join(HERE, "_ref", "libcodec_ref.so"))
    report, all_ok = {}, True
    for name, pts, kw in G.cases() + extra_cases():
        p = O.make_params(**kw)
        want = O.encode_intra(pts, p)                       # the restatement
        perf = (C.c_uint64 * 3)()
        n = lib.ref_encode(pts.ctypes.data, len(pts), p.point_resolution, p.octree_resolution,
                           p.color_bit_resolution if p.do_color_encoding else 0, p.color_coding_type, p.do_voxel_centroid,
                           p.jpeg_quality, p.frame_id, perf)
        got_stream = C.string_at(lib.ref_stream(), n)
        got_simplified = ref_cloud(lib)
        lib.ref_decode(got_stream, len(got_stream), p.point_resolution, p.octree_resolution,
                       p.color_bit_resolution if p.do_color_encoding else 0, p.color_coding_type, p.do_voxel_centroid)
        got_decoded = ref_cloud(lib)
        checks = {
            "bitstream": got_stream == want.bitstream,
            "perf": [int(x) for x in perf] == list(want.perf),
            "simplified_cloud": got_simplified.tobytes() == want.simplified.tobytes(),
            "decoded_cloud": got_decoded.tobytes() == O.decode_intra(want.bitstream).points.tobytes(),
        }
        if not checks["bitstream"]:   # where do they part?  (header: 0..139, then u64 B + 1028-byte table + payload, ...)
            a, b = np.frombuffer(got_stream, np.uint8), np.frombuffer(want.bitstream, np.uint8)
            m = min(len(a), len(b))
            diff = np.flatnonzero(a[:m] != b[:m])
            checks["first_difference_at_byte"] = int(diff[0]) if len(diff) else m
            checks["lengths"] = [len(a), len(b)]
        report[name] = checks
        ok = all(v is True for k, v in checks.items() if k in ("bitstream", "perf", "simplified_cloud", "decoded_cloud"))
        all_ok &= ok
        print("%-28s %s" % (name, "identical" if ok else "DIFFERS: %r" % checks))
    os.makedirs(os.path.join(HERE, "_ref"), exist_ok=True)
    json.dump({"pinned": bool(all_ok), "cases": report}, open(os.path.join(HERE, "_ref", "pin_report.json"), "w"), indent=1)
    print("oracle %s the reference on %d clouds" % ("EQUALS" if all_ok else "differs from", len(report)))
    if all_ok and "--regenerate" in sys.argv:
        import subprocess
        subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_codec_golden.py")], check=True)
        print("tests/golden/codec_golden.npz rewritten (oracle == reference on every case, so these are now the reference's vectors)")
    return 0 if all_ok else 1


if __name__ == "__main__":
    sys.exit(main())
