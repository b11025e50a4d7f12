#!/usr/bin/env python3
"""Frozen vectors of the codec path: small clouds -> what the CPU oracle (oracle/, the restatement of the reference) makes of
them: header fields, occupancy bytes, per-voxel colours, the complete bitstream; and rigid-transform coding vectors of the
inter-frame path.  Written once with

    python tests/golden/make_codec_golden.py

and committed (tests/golden/codec_golden.npz).  The tests hold the oracle, the product's host stage and the GPU path
against them, so none of the three can drift without a test noticing.  (The reference itself cannot run here -- PCL is
not available -- so these are vectors of the restatement, not of PCL: see DESIGN.md (c), "parity unpinned".)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O            # noqa: E402
from oracle import delta_oracle as D      # noqa: E402


def cloud(xyz, rgb):
    xyz = np.asarray(xyz, dtype=np.float32)
    pts = np.zeros(len(xyz), dtype=O.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    rgb = np.asarray(rgb, dtype=np.uint32)
    pts["rgba"] = rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


def cases():
    rng = np.random.default_rng(20260930)
    out = []
    # the worked example of SURVEY.md appendix F
    out.append(("appendix_f", cloud([(0.50, 0.50, 0.50), (0.60, 0.40, 0.52), (0.95, 0.10, 0.50), (0.50, 0.50, 0.51)],
                                    [(3, 1, 0), (33, 21, 10), (63, 41, 20), (8, 2, 1)]),
                dict(octree_resolution=0.25, point_resolution=0.25, color_coding_type=0)))
    out.append(("single_point", cloud([(0.3, 0.4, 0.5)], [(10, 20, 30)]), dict(octree_bits=4)))
    out.append(("growth_every_direction", cloud([(0.5, 0.5, 0.5), (0.9, 0.5, 0.5), (0.1, 0.5, 0.5), (0.5, 0.9, 0.5), (0.5, 0.1, 0.5),
                                                 (0.5, 0.5, 0.9), (0.5, 0.5, 0.1), (0.95, 0.95, 0.95), (0.02, 0.02, 0.02)],
                                                rng.integers(0, 256, (9, 3))), dict(octree_bits=5, color_coding_type=0)))
    xyz = rng.uniform(0.2, 0.8, (1500, 3)).astype(np.float32)
    xyz[rng.integers(0, 1500, 60), rng.integers(0, 3, 60)] = np.nan
    out.append(("nan_points", cloud(xyz, rng.integers(0, 256, (1500, 3))), dict(octree_bits=6)))
    base = rng.uniform(0.3, 0.7, (150, 3)).astype(np.float32)
    out.append(("duplicates_centroid", cloud(np.repeat(base, 5, axis=0), rng.integers(0, 256, (750, 3))),
                dict(octree_bits=7, color_coding_type=0, keep_centroid=1, color_bits=6)))
    for mode in (1, 2, 3):
        n = 1200 + 300 * mode
        u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        xyz = np.stack([0.2 + 0.6 * u, 0.2 + 0.6 * v, 0.5 + 0.2 * np.sin(5 * u) * np.cos(4 * v)], 1)
        col = np.stack([255 * u, 255 * v, 128 + 100 * np.sin(9 * u)], 1).astype(np.int64) + rng.integers(-6, 7, (n, 3))
        out.append(("surface_mode%d" % mode, cloud(xyz, np.clip(col, 0, 255)),
                    dict(octree_bits=7, color_coding_type=mode, jpeg_quality=80, frame_id=7 + mode)))
    xyz = (rng.normal(size=(2000, 3)) * 0.15 + 3.0).astype(np.float32)
    out.append(("odd_resolution", cloud(xyz, rng.integers(0, 256, (2000, 3))),
                dict(octree_resolution=0.0037, point_resolution=0.0037, color_coding_type=0, keep_centroid=1)))
    out.append(("geometry_only", cloud(rng.uniform(0, 1, (1800, 3)), rng.integers(0, 256, (1800, 3))), dict(octree_bits=6, color_bits=0)))
    return out


def rigid_cases():
    rng = np.random.default_rng(77)
    ms = []
    for k in range(24):
        axis = rng.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = rng.uniform(-1, 1) * [0.02, 0.4, 3.1][k % 3]
        K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)).astype(np.float32)
        m[:3, 3] = rng.uniform(-0.3, 0.3, 3)
        ms.append(m)
    for k in range(4):   # no rotation at all: the two-rows-and-signs mode
        m = np.eye(4, dtype=np.float32)
        m[:3, :3] = (rng.uniform(-1, 1, (3, 3)) * 0.7).astype(np.float32)
        ms.append(m)
    return ms


def main():
    z = {}
    names = []
    for name, pts, kw in cases():
        r = O.encode_intra(pts, O.make_params(**kw))
        names.append(name)
        z[name + "/points"] = pts.view(np.uint8).reshape(len(pts), 32)
        z[name + "/params"] = np.array(sorted(kw.items()), dtype=object).astype(str)
        z[name + "/bitstream"] = np.frombuffer(r.bitstream, dtype=np.uint8)
        z[name + "/occupancy"] = r.occupancy
        z[name + "/bgr"] = r.bgr
        z[name + "/centroid"] = r.centroid_bytes
        z[name + "/header"] = np.array([r.depth, r.n_points_in, r.n_leaves, r.n_branches], dtype=np.int64)
        z[name + "/bbox"] = r.bbox
        z[name + "/decoded"] = O.decode_intra(r.bitstream).points.view(np.uint8).reshape(-1, 32)
    z["names"] = np.array(names)
    ms = rigid_cases()
    z["rigid/matrices"] = np.stack(ms)
    comps = [D.rigid_compress(m) for m in ms]
    z["rigid/lengths"] = np.array([len(c) for c in comps], dtype=np.int64)
    z["rigid/comp"] = np.array([c + [0] * (10 - len(c)) for c in comps], dtype=np.int16)
    z["rigid/decoded"] = np.stack([D.rigid_decompress(c) for c in comps])
    np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "codec_golden.npz"), **z)
    print("wrote", len(names), "codec cases and", len(ms), "rigid transforms")


if __name__ == "__main__":
    main()
