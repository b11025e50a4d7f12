"""Every frame a timing tool runs is held against the oracle's golden digests (tests/golden/frame_digests.json, made
by tests/golden/make_frame_digests.py): a timing run that cannot notice wrong bytes is how five kernel commits of round
2 ended up timed but never compared.  Usage inside a tool:

    from frame_digests import check_frame
    check_frame("cfg2", frame_index, hot, stream)      # raises AssertionError with the field that differs

`hot` is a binding.HotProducts (with or without host copies), `stream` the final bitstream or None.  Workloads without
digests (cfg3, cfg4 at full size, cfg5) are checked for L/B/D only when an entry exists, otherwise reported once."""
import hashlib
import json
import os

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_DIGESTS = None
_warned = set()


def digests():
    global _DIGESTS
    if _DIGESTS is None:
        with open(os.path.join(_ROOT, "tests", "golden", "frame_digests.json")) as fh:
            _DIGESTS = json.load(fh)
    return _DIGESTS


def _sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


def check_frame(workload, frame, hot, stream=None, frame_id_is_one=True):
    key = "%s/%d" % (workload, frame)
    want = digests().get(key)
    if want is None:
        if key not in _warned:
            _warned.add(key)
            print("frame_digests: no golden digest for %s (not checked)" % key)
        return False
    got = dict(L=hot.n_leaves, B=hot.n_branches, D=hot.depth)
    for k in ("L", "B", "D"):
        assert got[k] == want[k], "%s: %s = %d, the oracle says %d" % (key, k, got[k], want[k])
    import numpy as np
    assert _sha(np.asarray(hot.bbox, dtype=np.float64).tobytes()) == want["bbox"], "%s: bounding box differs from the oracle's" % key
    if hasattr(hot, "occupancy"):
        assert _sha(hot.occupancy.tobytes()) == want["occupancy"], "%s: occupancy stream differs from the oracle's" % key
        assert _sha(hot.bgr.tobytes()) == want["bgr"], "%s: voxel colours differ from the oracle's" % key
    if stream is not None and frame_id_is_one:
        assert len(stream) == want["bitstream_bytes"] and _sha(stream) == want["bitstream"], "%s: bitstream differs from the oracle's" % key
    return True
