#!/usr/bin/env python3
"""Golden digests of the bench / profiling workloads, from the CPU oracle (run here, committed as data):

    python tests/golden/make_frame_digests.py      ->  tests/golden/frame_digests.json

For frames 0..3 of cfg1, cfg2u, cfg3v, frames 0..11 of cfg2 and for "cfg4-1M" (cfg4's settings on its first 1 000 000 points) the
SHA-256 of the bounding box, the occupancy stream, the per-voxel colours and the final bitstream (frame_id 1), plus
L, B and D.  tools/frame_digests.py holds a timed frame against them, so that a timing tool cannot time wrong bytes
without noticing."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import __graft_entry__ as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

syn = G.load_package().synthetic


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def entry(pts, cfg):
    kw = dict(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
              jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"], frame_id=1)
    w = O.encode_intra(pts, O.make_params(**kw))
    return dict(n=int(len(pts)), L=int(w.n_leaves), B=int(w.n_branches), D=int(w.depth),
                bbox=sha(np.asarray(w.bbox, dtype=np.float64).tobytes()), occupancy=sha(np.asarray(w.occupancy).tobytes()),
                bgr=sha(np.asarray(w.bgr).tobytes()), bitstream=sha(w.bitstream), bitstream_bytes=len(w.bitstream))


def main():
    out = {}
    for wl in ("cfg1", "cfg2", "cfg2u", "cfg3v"):
        for f in range(12 if wl == "cfg2" else 4):   # cfg2: twelve distinct frames = 384 MB, more than the 256 MB Infinity Cache
            out["%s/%d" % (wl, f)] = entry(syn.make_frame(wl, frame=f), syn.CONFIGS[wl])
            print(wl, f, out["%s/%d" % (wl, f)]["L"], flush=True)
    for f in range(2):
        out["cfg4-1M/%d" % f] = entry(syn.make_frame("cfg4", frame=f, n=1_000_000), syn.CONFIGS["cfg4"])
    with open(os.path.join(ROOT, "tests", "golden", "frame_digests.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)
        fh.write("\n")


if __name__ == "__main__":
    main()
