"""CPU ORACLE (test infrastructure, not product code): the inter-frame ("delta") path of the reference,
OctreePointCloudCodecV2::encodePointCloudDeltaFrame / decodePointCloudDeltaFrame
(impl.hpp = cloud_codec_v2/include/pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp:318-568, 787-1235),
RigidTransformCoding (impl/rigid_transform_coding_impl.hpp:63-203) and QuaternionCoding
(impl/quaternion_coding_impl.hpp:55-222).

PARITY STATUS
  * rigid-transform / quaternion coding: restated line by line from the reference's own sources (they need Eigen, which
    is not in the image, so they cannot be compiled); Eigen's Quaternion(Matrix3) and toRotationMatrix() are restated
    from Eigen 3.3 (Geometry/Quaternion.h).
  * macroblock trees, shared-block detection, size and colour-variance gates, chunk format, residual points: restated
    from impl.hpp; the trees themselves are PCL octrees with a defined bounding box (oracle/octree_oracle.c).
  * ICP: the reference calls pcl::IterativeClosestPoint (PCL, un-vendored): "parity unpinned".  `icp()` below restates
    PCL 1.10's default pipeline (nearest-neighbour correspondences, Umeyama/SVD estimation in float, the
    DefaultConvergenceCriteria); bit-exactness with PCL's float arithmetic (Eigen JacobiSVD, FLANN) is not claimed.
"""
import ctypes as C

import numpy as np

from . import oracle as O

F = np.float32


# ------------------------------------------------------------------------------------------------
# trees with a defined bounding box
# ------------------------------------------------------------------------------------------------
def tree(points, res, box=(0.0, 0.0, 0.0, 1.0, 1.0, 1.0)):
    """defineBoundingBox(box) + addPointsFromInputCloud: leaves in depth-first order.
    Returns keys (L,3) uint32, list of index arrays, bbox (6,), depth."""
    lib = O.lib()
    pts = np.ascontiguousarray(points)
    keys_p, counts_p, idx_p = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)(), C.POINTER(C.c_int)()
    n_leaves, depth = C.c_uint64(), C.c_uint()
    bbox = (C.c_double * 6)()
    lib.pcco_tree_with_defined_box.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.POINTER(C.c_double), C.POINTER(C.POINTER(C.c_uint32)),
                                               C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.c_uint64),
                                               C.POINTER(C.c_double), C.POINTER(C.c_uint)]
    lib.pcco_tree_with_defined_box(pts.ctypes.data, len(pts), float(res), (C.c_double * 6)(*box), C.byref(keys_p), C.byref(counts_p),
                                   C.byref(idx_p), C.byref(n_leaves), bbox, C.byref(depth))
    L = n_leaves.value
    keys = np.ctypeslib.as_array(keys_p, shape=(max(L, 1) * 3,))[:3 * L].reshape(L, 3).copy()
    counts = np.ctypeslib.as_array(counts_p, shape=(max(L, 1),))[:L].copy()
    total = int(counts.sum())
    flat = np.ctypeslib.as_array(idx_p, shape=(max(total, 1),))[:total].copy()
    libc = C.CDLL(None)
    for p in (keys_p, counts_p, idx_p):
        libc.free(p)
    offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    return keys, [flat[offs[i]:offs[i + 1]] for i in range(L)], np.array(list(bbox)), depth.value


def simplify(points, res, keep_centroid=False):
    """simplifyPCloud (impl.hpp:318-403): voxel centres (genLeafNodeCenterFromOctreeKey) or float centroids, mean colours."""
    keys, lists, bbox, _ = tree(points, res)
    out = np.zeros(len(keys), dtype=O.POINT_DTYPE)
    out["w"] = 1.0
    if not keep_centroid:
        for a, name in enumerate("xyz"):
            out[name] = ((keys[:, a].astype(np.float64) + 0.5) * res + bbox[a]).astype(np.float32)
    r = (points["rgba"] >> 16) & 0xFF
    g = (points["rgba"] >> 8) & 0xFF
    b = points["rgba"] & 0xFF
    for i, idx in enumerate(lists):
        if keep_centroid:   # pcl::compute3DCentroid: float sums in index order
            s = np.zeros(3, dtype=np.float32)
            for k in idx:
                s[0] += points["x"][k]; s[1] += points["y"][k]; s[2] += points["z"][k]
            c = s / F(len(idx))
            out["x"][i], out["y"][i], out["z"][i] = c
        n = len(idx)
        rr, gg, bb = int(r[idx].sum()) // n, int(g[idx].sum()) // n, int(b[idx].sum()) // n
        out["rgba"][i] = (bb & 0xFF) | ((gg & 0xFF) << 8) | ((rr & 0xFF) << 16) | 0xFF000000
    return out


# ------------------------------------------------------------------------------------------------
# quaternion / rigid transform coding
# ------------------------------------------------------------------------------------------------
def quat_from_matrix(m):
    """Eigen::Quaternion<float>(Matrix3f): returns (x, y, z, w) float32."""
    m = m.astype(np.float32)
    q = np.zeros(4, dtype=np.float32)  # x y z w
    t = F(F(m[0, 0] + m[1, 1]) + m[2, 2])
    if t > 0:
        t = np.sqrt(F(t + F(1.0)))
        q[3] = F(0.5) * t
        t = F(0.5) / t
        q[0] = F(m[2, 1] - m[1, 2]) * t
        q[1] = F(m[0, 2] - m[2, 0]) * t
        q[2] = F(m[1, 0] - m[0, 1]) * t
    else:
        i = 0
        if m[1, 1] > m[0, 0]:
            i = 1
        if m[2, 2] > m[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(F(F(F(m[i, i] - m[j, j]) - m[k, k]) + F(1.0)))
        q[i] = F(0.5) * t
        t = F(0.5) / t
        q[3] = F(m[k, j] - m[j, k]) * t
        q[j] = F(m[j, i] + m[i, j]) * t
        q[k] = F(m[k, i] + m[i, k]) * t
    return q


def quat_to_matrix(q):
    """Eigen::Quaternion<float>::toRotationMatrix()."""
    x, y, z, w = [F(v) for v in q]
    tx, ty, tz = F(2) * x, F(2) * y, F(2) * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    m = np.zeros((3, 3), dtype=np.float32)
    m[0, 0] = F(1) - F(tyy + tzz); m[0, 1] = txy - twz; m[0, 2] = txz + twy
    m[1, 0] = txy + twz; m[1, 1] = F(1) - F(txx + tzz); m[1, 2] = tyz - twx
    m[2, 0] = txz - twy; m[2, 1] = tyz + twx; m[2, 2] = F(1) - F(txx + tyy)
    return m


def _i16(v):
    return int(np.int16(np.int64(v) & 0xFFFF))


def _clamp1(v):
    return F(-1) if v < -1 else (F(1) if v > 1 else v)


def quat_compress(q):
    """QuaternionCoding::compressQuaternion (quaternion_coding_impl.hpp:55-166): q = (x,y,z,w) -> 3 int16."""
    scale = F(1.41421)
    x, y, z, w = [F(v) for v in q]

    def pack(a, b, c, neg, bit1, bit2):
        ra, rb, rc = a * scale, b * scale, c * scale
        if neg:
            ra, rb, rc = -ra, -rb, -rc
        ra, rb, rc = _clamp1(ra), _clamp1(rb), _clamp1(rc)
        s0 = _i16(int(ra * F(32767)))
        s1 = _i16((int(rb * F(32767)) & 0xFFFE) | bit1)
        s2 = _i16((int(rc * F(32767)) & 0xFFFE) | bit2)
        return [s0, s1, s2]

    if w > x and w > y and w > z:
        return pack(x, y, z, w < 0, 1, 1)
    if z > x and z > y:
        return pack(x, y, w, z < 0, 1, 0)
    if y > x:
        return pack(x, z, w, y < 0, 0, 1)
    return pack(y, z, w, x < 0, 0, 0)


def quat_decompress(s):
    """deCompressQuaternion (:168-222); like the reference it clears the low bits of s[1], s[2] in place."""
    which = ((s[1] & 1) << 1) | (s[2] & 1)
    s[1] = _i16(s[1] & 0xFFFE)
    s[2] = _i16(s[2] & 0xFFFE)
    scale = F(F(F(1.0) / F(32767.0)) / F(1.41421))
    a, b, c = F(s[0]) * scale, F(s[1]) * scale, F(s[2]) * scale
    d = F(F(F(F(1) - F(a * a)) - F(b * b)) - F(c * c))
    if d > np.finfo(np.float32).eps:
        d = np.sqrt(d)
    if which == 3:
        return np.array([a, b, c, d], dtype=np.float32)       # x y z w
    if which == 2:
        return np.array([a, b, d, c], dtype=np.float32)       # x y w given, z derived
    if which == 1:
        return np.array([a, d, b, c], dtype=np.float32)       # x z w given, y derived
    return np.array([d, a, b, c], dtype=np.float32)           # y z w given, x derived


def rigid_compress(tr):
    """RigidTransformCoding::compressRigidTransform (rigid_transform_coding_impl.hpp:63-149): 4x4 float -> list of int16."""
    tr = tr.astype(np.float32)
    scaling = F(32767.0 / 2.5)
    rot = tr[:3, :3]
    q = quat_from_matrix(rot)
    comp = quat_compress(q)
    test = quat_to_matrix(quat_decompress(comp))     # note: this clears the low bits of comp[1], comp[2]
    stable = True
    for i in range(9):
        if abs(F(test[i // 3, i % 3] - tr[i // 3, i % 3])) > 0.001:
            stable = False
            break
    if not stable:
        comp = comp + [0, 0, 0, 0]
        for l in range(3):
            comp[l] = _i16(int(rot[0, l] * F(32766)))
            comp[l + 3] = _i16(int(rot[1, l] * F(32766)))
            comp[6] = _i16(comp[6] + ((1 << l) if rot[2, l] < 0 else 0))
    else:
        comp = quat_compress(q)
    for a in range(3):
        t = F(tr[a, 3])
        t = F(2.5) if t > 2.5 else (F(-2.5) if t < -2.5 else t)
        comp.append(_i16(int(t * F(scaling - F(1)))))
    return comp


def rigid_decompress(comp):
    """deCompressRigidTransform (:158-203)."""
    comp = list(comp)
    scaling = F(32767.0 / 2.5)
    tr = np.zeros((4, 4), dtype=np.float32)
    if len(comp) == 6:
        tr[:3, :3] = quat_to_matrix(quat_decompress(comp))
    else:
        for l in range(3):
            tr[0, l] = F(comp[l]) / F(32766)
            tr[1, l] = F(comp[l + 3]) / F(32766)
            tr[2, l] = np.sqrt(F(F(F(1) - F(tr[0, l] * tr[0, l])) - F(tr[1, l] * tr[1, l])))
            if ((1 << l) & int(comp[6])) == (1 << l):
                tr[2, l] = -tr[2, l]
    n = len(comp)
    for a in range(3):
        tr[a, 3] = F(comp[n - 3 + a]) / F(scaling - F(1))
    tr[3, 3] = 1
    return tr


def transform_points(xyz, m):
    """pcl::transformPointCloud with a Matrix4f (PCL 1.10 Transformer::se3): x*c0 + (y*c1 + (z*c2 + c3)), float."""
    xyz = xyz.astype(np.float32)
    m = m.astype(np.float32)
    out = np.zeros_like(xyz)
    for r in range(3):
        out[:, r] = (xyz[:, 0] * m[r, 0] + (xyz[:, 1] * m[r, 1] + (xyz[:, 2] * m[r, 2] + m[r, 3]).astype(np.float32)).astype(np.float32)).astype(np.float32)
    return out


# ------------------------------------------------------------------------------------------------
# ICP (restatement of PCL 1.10 defaults; see the module docstring)
# ------------------------------------------------------------------------------------------------
def _nearest(src, tgt):
    d = src[:, None, :] - tgt[None, :, :]
    d2 = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2]).astype(np.float32)
    idx = d2.argmin(axis=1)
    return idx, d2[np.arange(len(src)), idx]


def umeyama(src, dst):
    """Eigen::umeyama(src, dst, with_scaling=false), float."""
    n = F(len(src))
    sm, dm = (src.sum(axis=0) / n).astype(np.float32), (dst.sum(axis=0) / n).astype(np.float32)
    sigma = ((dst - dm).T.astype(np.float32) @ (src - sm).astype(np.float32) / n).astype(np.float32)
    u, _, vt = np.linalg.svd(sigma.astype(np.float32))
    s = np.ones(3, dtype=np.float32)
    if np.linalg.det(u) * np.linalg.det(vt) < 0:
        s[2] = -1
    r = (u * s) @ vt
    t = dm - r @ sm
    m = np.eye(4, dtype=np.float32)
    m[:3, :3], m[:3, 3] = r, t
    return m


def icp(src, tgt, max_iterations=50, transformation_epsilon=1e-8):
    """IterativeClosestPoint::computeTransformation with the settings of impl.hpp:544-556.
    Returns (converged, final 4x4 float, fitness score)."""
    src, tgt = src.astype(np.float32), tgt.astype(np.float32)
    final = np.eye(4, dtype=np.float32)
    cur = src.copy()
    prev_mse = np.finfo(np.float64).max
    rot_thr, trans_thr = 1.0 - transformation_epsilon, transformation_epsilon
    mse_rel, mse_abs = 3 * transformation_epsilon, 1e-12
    it = 0
    converged = False
    while not converged:
        idx, d2 = _nearest(cur, tgt)
        if len(idx) < 3:
            break
        tr = umeyama(cur, tgt[idx])
        cur = transform_points(cur, tr)
        final = (tr @ final).astype(np.float32)
        it += 1
        # DefaultConvergenceCriteria::hasConverged
        if it >= max_iterations:
            converged = True
            break
        cos_angle = 0.5 * (float(tr[0, 0]) + float(tr[1, 1]) + float(tr[2, 2]) - 1)
        tsq = float(tr[0, 3]) ** 2 + float(tr[1, 3]) ** 2 + float(tr[2, 3]) ** 2
        if cos_angle >= rot_thr and tsq <= trans_thr:
            converged = True
            break
        mse = float(d2.astype(np.float64).sum() / len(d2))
        if abs(mse - prev_mse) < mse_abs or abs(mse - prev_mse) / prev_mse < mse_rel:
            converged = True
            break
        prev_mse = mse
    _, d2 = _nearest(transform_points(src, final), tgt)
    return converged, final, float(d2.astype(np.float64).sum() / len(d2))


# ------------------------------------------------------------------------------------------------
# do_icp_prediction (impl.hpp:443-568) and the frame level
# ------------------------------------------------------------------------------------------------
def _rgb(points):
    return np.stack([(points["rgba"] >> 16) & 0xFF, (points["rgba"] >> 8) & 0xFF, points["rgba"] & 0xFF], 1).astype(np.float64)


def gates(i_block, p_block, do_icp_color_offset=False, var_threshold=100.0):
    """Everything of do_icp_prediction before the ICP itself: (do_icp, rgb_offsets, in_var, out_var)."""
    ni, np_ = len(i_block), len(p_block)
    do_icp = (np_ < ni * 2) and (np_ >= ni * 0.5) if np_ > 6 else False
    if not do_icp:
        return False, [0, 0, 0], 0.0, 0.0

    def stats(block):
        c = _rgb(block)
        av = np.zeros(3)
        for row in c:
            av += row
        av /= len(c)
        var = 0.0
        for row in c:
            var += (row[0] - av[0]) * (row[0] - av[0]) + (row[1] - av[1]) * (row[1] - av[1]) + (row[2] - av[2]) * (row[2] - av[2])
        return av, var / (3 * len(c))

    in_av, in_var = stats(i_block)
    out_av, out_var = stats(p_block)
    if in_var > var_threshold or out_var > var_threshold:
        do_icp = False
    off = [0, 0, 0]
    if do_icp_color_offset:
        for k in range(3):
            if abs(out_av[k] - in_av[k]) < 32:
                off[k] = int(np.int8(int(out_av[k] - in_av[k])))
    return do_icp, off, in_var, out_var


def _xyz(points):
    return np.stack([points["x"], points["y"], points["z"]], 1).astype(np.float32)


def encode_delta(i_cloud, p_cloud, res, point_res, macroblock_size=16, color_bits=8, color_coding_type=1, keep_centroid=0,
                 do_icp_color_offset=False, icp_on_original=False, write_out_cloud=True, icp_fn=icp):
    """encodePointCloudDeltaFrame (impl.hpp:787-1118), serial branch.  Returns a dict."""
    simp = p_cloud if icp_on_original else simplify(p_cloud, res, keep_centroid)
    res_mb = res * macroblock_size
    i_keys, i_lists, _, _ = tree(i_cloud, res_mb)
    p_keys, p_lists, _, _ = tree(simp, res_mb)
    i_index = {tuple(k): n for n, k in enumerate(i_keys.tolist())}
    p_stream = bytearray()
    out_parts, intra_parts = [], []
    blocks = []
    shared = converged_count = 0
    for pk, pl in zip(p_keys.tolist(), p_lists):
        cloud_out = simp[pl]
        ib = i_index.get(tuple(pk))
        rec = dict(key=tuple(pk), n_p=len(pl), shared=ib is not None, icp=False, success=False, comp=None, offsets=[0, 0, 0])
        if ib is not None:
            shared += 1
            cloud_in = i_cloud[i_lists[ib]]
            do_icp, off, _, _ = gates(cloud_in, cloud_out, do_icp_color_offset)
            rec.update(icp=do_icp, offsets=off, n_i=len(cloud_in))
            success, rt = False, None
            if do_icp:
                conv, rt, fitness = icp_fn(_xyz(cloud_in), _xyz(cloud_out))
                success = conv and fitness < point_res * 2
                rec.update(fitness=fitness, rt=rt)
            if success:
                converged_count += 1
                comp = rigid_compress(rt)
                rec.update(success=True, comp=list(comp))
                chunk = 3 * 2 + len(comp) * 2 + (3 if do_icp_color_offset else 0)
                p_stream += bytes([chunk & 0xFF])
                p_stream += np.array(pk, dtype="<i2").tobytes() + np.array(comp, dtype="<i2").tobytes()
                if do_icp_color_offset:
                    p_stream += np.array(off, dtype=np.int8).tobytes()
                if write_out_cloud:
                    mdec = rigid_decompress(comp)
                    pred = cloud_in.copy()
                    xyz = transform_points(_xyz(cloud_in), mdec)
                    pred["x"], pred["y"], pred["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
                    if do_icp_color_offset:
                        pred = _add_colour(pred, off, doubled=False)
                    out_parts.append(pred)
                blocks.append(rec)
                continue
        if write_out_cloud:
            out_parts.append(cloud_out)
        intra_parts.append(cloud_out)
        blocks.append(rec)
    intra = np.concatenate(intra_parts) if intra_parts else np.zeros(0, dtype=O.POINT_DTYPE)
    # the intra coder of the residual points is built with the constructor defaults for the arguments it does not
    # pass (impl.hpp:1089-1101 vs codec.h:108-121): createScalableStream = true, jpeg_quality = 75; a fresh coder: id 1
    params = O.make_params(octree_bits=0, color_bits=color_bits, color_coding_type=color_coding_type, keep_centroid=keep_centroid,
                           jpeg_quality=75, octree_resolution=res, point_resolution=point_res, frame_id=1, create_scalable=1)
    r = O.encode_intra(intra, params, keep=False) if len(intra) else None
    return dict(i_stream=b"" if r is None else r.bitstream, p_stream=bytes(p_stream), blocks=blocks, intra_points=intra,
                out_cloud=np.concatenate(out_parts) if out_parts else np.zeros(0, dtype=O.POINT_DTYPE),
                shared_percentage=np.float32(shared) / np.float32(len(p_keys)) if len(p_keys) else 0.0,
                convergence_percentage=np.float32(converged_count) / np.float32(shared) if shared else 0.0, simplified=simp)


def _add_colour(points, off, doubled):
    """encoder out cloud: pt.r += off (impl.hpp:901-905); decoder: p.r += p.r + off (impl.hpp:1187-1189), uint8 wrap."""
    out = points.copy()
    c = [(out["rgba"] >> 16) & 0xFF, (out["rgba"] >> 8) & 0xFF, out["rgba"] & 0xFF]
    for k in range(3):
        v = c[k].astype(np.int64)
        c[k] = ((v + v + off[k]) if doubled else (v + off[k])) & 0xFF
    out["rgba"] = (out["rgba"] & np.uint32(0xFF000000)) | (c[0].astype(np.uint32) << 16) | (c[1].astype(np.uint32) << 8) | c[2].astype(np.uint32)
    return out


def decode_delta(i_cloud, i_stream, p_stream, res, macroblock_size=16, do_icp_color_offset=False):
    """decodePointCloudDeltaFrame (impl.hpp:1120-1235)."""
    i_keys, i_lists, _, _ = tree(i_cloud, res * macroblock_size)
    i_index = {tuple(k): n for n, k in enumerate(i_keys.tolist())}
    parts = []
    pos = 0
    while pos < len(p_stream):
        chunk = p_stream[pos]; pos += 1
        if chunk == 0 or pos + chunk > len(p_stream):
            break
        key = tuple(np.frombuffer(p_stream, dtype="<i2", count=3, offset=pos).tolist()); pos += 6
        n_comp = (chunk - 6 - (3 if do_icp_color_offset else 0)) // 2
        comp = np.frombuffer(p_stream, dtype="<i2", count=n_comp, offset=pos).tolist(); pos += 2 * n_comp
        off = [0, 0, 0]
        if do_icp_color_offset:
            off = np.frombuffer(p_stream, dtype=np.int8, count=3, offset=pos).tolist(); pos += 3
        ib = i_index.get(key)
        if ib is None:
            continue
        cloud_in = i_cloud[i_lists[ib]]
        mdec = rigid_decompress(comp)
        pred = cloud_in.copy()
        xyz = transform_points(_xyz(cloud_in), mdec)
        pred["x"], pred["y"], pred["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
        if do_icp_color_offset:
            pred = _add_colour(pred, off, doubled=True)
        parts.append(pred)
    if len(i_stream):
        parts.append(O.decode_intra(i_stream).points)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=O.POINT_DTYPE)
