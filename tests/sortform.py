"""Sort-based (breadth-first) restatement of the octree stage in numpy.

Test helper only.  It states, in array form, the algorithm the HIP kernels
implement (epoch-aware keys -> Morton sort -> leaf runs -> closed-form DFS
offsets) so that the *formulation* can be
checked against the pointer-octree oracle on the CPU, independent of any GPU.
"""
import numpy as np

FLT_EPS = float(np.finfo(np.float32).eps)


def bbox_epochs(xyz, res):
    """Simulate adoptBoundingBoxToPoint over the cloud in order.

    Returns (events, mn, mx, depth): events = list of dicts with the index of the
    point that triggered the change, the box origin in force from that point on,
    and the depth; mn/mx/depth = final box.
    """
    x = xyz.astype(np.float64)
    finite = np.isfinite(xyz).all(axis=1)
    idx = np.nonzero(finite)[0]
    if len(idx) == 0:
        return [], None, None, 0
    i0 = int(idx[0])
    p = x[i0]
    mn = p - res / 2
    mx = p + res / 2
    # getKeyBitSize on the empty tree
    mk = max(int(np.ceil((mx[a] - mn[a] - FLT_EPS) / res)) for a in range(3))
    mk = max(mk, 2)
    depth = int(np.ceil(np.log2(mk) - FLT_EPS))
    side = float(1 << depth) * res
    for a in range(3):
        over = (side - (mx[a] - mn[a])) / 2.0
        if over > FLT_EPS:
            mn[a] -= over
            mx[a] += over
    events = [dict(index=i0, mn=mn.copy(), depth=depth, lowered=(0, 0, 0), depth_before=0)]
    cur = i0 + 1
    n = len(x)
    while cur < n:
        seg = x[cur:]
        fin = finite[cur:]
        viol = (((seg < mn) | (seg >= mx)).any(axis=1)) & fin
        nz = np.nonzero(viol)[0]
        if len(nz) == 0:
            break
        i = cur + int(nz[0])
        p = x[i]
        while True:
            lo = p < mn
            up = p >= mx
            if not (lo.any() or up.any()):
                break
            side = float(1 << depth) * res
            lowered = tuple(int(not u) for u in up)
            for a in range(3):
                if not up[a]:
                    mn[a] -= side
            depth_before = depth
            depth += 1
            side = float(1 << depth) * res - FLT_EPS
            mx = mn + side
            events.append(dict(index=i, mn=mn.copy(), depth=depth, lowered=lowered, depth_before=depth_before))
        cur = i + 1
    return events, mn, mx, depth


def point_keys(xyz, res, events, depth):
    """Final D-bit keys per finite point (epoch origin + later re-rootings)."""
    n = len(xyz)
    keys = np.zeros((n, 3), dtype=np.uint64)
    finite = np.isfinite(xyz).all(axis=1)
    x = xyz.astype(np.float64)
    # epochs: after the LAST event at a given index
    bounds = []
    for k, e in enumerate(events):
        if k + 1 < len(events) and events[k + 1]["index"] == e["index"]:
            continue
        bounds.append(k)
    for bi, k in enumerate(bounds):
        start = events[k]["index"]
        end = events[bounds[bi + 1]]["index"] if bi + 1 < len(bounds) else n
        shift = np.zeros(3, dtype=np.uint64)
        for later in events[k + 1:]:
            for a in range(3):
                if later["lowered"][a]:
                    shift[a] += np.uint64(1 << later["depth_before"])
        seg = x[start:end]
        kk = ((seg - events[k]["mn"]) / res)
        kk = np.where(np.isfinite(kk), kk, 0.0)
        keys[start:end] = kk.astype(np.uint64) + shift
    return keys, finite


def morton(keys, depth):
    code = np.zeros(len(keys), dtype=np.uint64)
    for b in range(depth):
        tri = (((keys[:, 0] >> np.uint64(b)) & np.uint64(1)) << np.uint64(2)) | \
              (((keys[:, 1] >> np.uint64(b)) & np.uint64(1)) << np.uint64(1)) | \
              ((keys[:, 2] >> np.uint64(b)) & np.uint64(1))
        code |= tri << np.uint64(3 * b)
    return code


def demorton(code, depth):
    keys = np.zeros((len(code), 3), dtype=np.uint64)
    for b in range(depth):
        tri = (code >> np.uint64(3 * b)) & np.uint64(7)
        keys[:, 0] |= ((tri >> np.uint64(2)) & np.uint64(1)) << np.uint64(b)
        keys[:, 1] |= ((tri >> np.uint64(1)) & np.uint64(1)) << np.uint64(b)
        keys[:, 2] |= (tri & np.uint64(1)) << np.uint64(b)
    return keys


def occupancy_stream(leaf_codes, depth):
    """Closed-form DFS occupancy bytes from sorted unique leaf Morton codes."""
    L = len(leaf_codes)
    t = np.zeros(L, dtype=np.int64)
    t[0] = depth
    if L > 1:
        x = leaf_codes[1:] ^ leaf_codes[:-1]
        msb = np.floor(np.log2(x.astype(np.float64))).astype(np.int64)
        # guard against float rounding of log2 near powers of two
        msb = np.where((np.uint64(1) << msb.astype(np.uint64)) > x, msb - 1, msb)
        msb = np.where((x >> (msb + 1).astype(np.uint64)) > 0, msb + 1, msb)
        t[1:] = msb // 3
    base = np.concatenate([[0], np.cumsum(t)[:-1]])
    B = int(t.sum())
    occ = np.zeros(B, dtype=np.uint8)
    # node (level l, first leaf f) lives at base[f] + l - (depth - t[f])
    for l in range(depth):
        shift = np.uint64(3 * (depth - l))
        node = leaf_codes >> shift if 3 * (depth - l) < 64 else np.zeros(L, dtype=np.uint64)
        first = np.ones(L, dtype=bool)
        first[1:] = node[1:] != node[:-1]
        f_idx = np.maximum.accumulate(np.where(first, np.arange(L), 0))
        off = base[f_idx] + l - (depth - t[f_idx])
        child = ((leaf_codes >> np.uint64(3 * (depth - 1 - l))) & np.uint64(7)).astype(np.uint8)
        np.bitwise_or.at(occ, off, (np.uint8(1) << child))
    return occ, t, base


def encode_geometry(points, res):
    """Full sort-formulation: returns dict with the same products as the oracle."""
    xyz = np.stack([points["x"], points["y"], points["z"]], 1)
    events, mn, mx, depth = bbox_epochs(xyz, res)
    if not events:
        return None
    keys, finite = point_keys(xyz, res, events, depth)
    idx = np.nonzero(finite)[0]
    code = morton(keys[idx], depth)
    order = np.argsort(code, kind="stable")
    scode = code[order]
    sidx = idx[order]
    head = np.ones(len(scode), dtype=bool)
    head[1:] = scode[1:] != scode[:-1]
    leaf_codes = scode[head]
    starts = np.nonzero(head)[0]
    counts = np.diff(np.concatenate([starts, [len(scode)]]))
    occ, t, base = occupancy_stream(leaf_codes, depth)
    # per-leaf colour mean (P6): integer truncation, (b, g, r)
    rgba = points["rgba"][sidx].astype(np.uint64)
    ch = np.stack([(rgba >> np.uint64(s)) & np.uint64(0xFF) for s in (0, 8, 16)], 1)
    sums = np.add.reduceat(ch, starts, axis=0)
    bgr = np.where(counts[:, None] > 1, sums // counts[:, None].astype(np.uint64), sums).astype(np.uint8)
    return dict(bbox=np.concatenate([mn, mx]), depth=depth, leaf_keys=demorton(leaf_codes, depth),
                leaf_counts=counts, occupancy=occ, bgr=bgr.reshape(-1), n_events=len(events),
                sorted_idx=sidx, starts=starts)


def reference_style_lines_stream(oracle, pts, L_expected=None, tail_value=0x5A, **kw):
    """A colour-coding-type-2 frame as the REFERENCE's own encoder writes it when the frame has fewer than 2048 voxels:
    jpegcc.h:256-275 leaves `im_in.width = 2048` in the `num_lines == 0` branch, so writeJPEG (jpeg_io.hpp:259,302-309)
    codes a 2048-pixel strip whose first L pixels are the voxels' colours and whose tail is whatever lies behind the
    3 L-byte buffer (undefined; a fixed byte here).  Returns (stream, strip as decoded by the oracle's JPEG decoder)."""
    import struct
    want = oracle.encode_intra(pts, oracle.make_params(color_coding_type=2, **kw))
    L = want.n_leaves
    assert L < 2048 and (L_expected is None or L == L_expected)
    own_payload = want.color_payload
    tail_len = 8 + len(oracle.rc_encode(own_payload))          # u64 length + range-coded payload close the frame
    prefix = want.bitstream[:-tail_len]
    strip = np.full((1, 2048, 3), tail_value, dtype=np.uint8)
    strip[0, :L, :] = want.bgr.reshape(L, 3)
    jpg = oracle.jpeg_encode(strip, kw.get("jpeg_quality", 85))
    payload = struct.pack("<II", 1, len(jpg)) + jpg                 # JPEGLineData::serialize: line count, size, bytes
    stream = prefix + struct.pack("<Q", len(payload)) + oracle.rc_encode(payload)
    return stream, oracle.jpeg_decode(jpg)[0], want
