"""Deterministic synthetic XYZRGB frames.

Generator = SplitMix64 seeded per configuration; frame f of a sequence uses
seed + f.  All coordinates are produced as float32 and the point order is the
generation order (it matters: the reference's adaptive bounding box and the
per-voxel point lists are order dependent).  The frames are then passed through
the same group normalisation the reference applies before encoding
(normalize_pointclouds, impl.hpp:1871-1967, bb_expand_factor = 0.2).
"""
import numpy as np

POINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("rgba", "<u4"), ("pad", "<u4", (3,))]
)

_GOLDEN = np.uint64(0x9E3779B97F4A7C15)


def splitmix64(seed, n, stream=0):
    """n outputs of SplitMix64 started at `seed` (vectorised; stream offsets the counter)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64) + np.uint64(stream) * np.uint64(n)
        z = np.uint64(seed) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return z ^ (z >> np.uint64(31))


def _u01(seed, n, stream):
    return (splitmix64(seed, n, stream) >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def normalize(points, bb_expand_factor=0.2):
    """normalize_pointclouds for one cloud, float32 arithmetic as impl.hpp:1915-1946."""
    mn = np.array([points[a].min() for a in "xyz"], dtype=np.float32)
    mx = np.array([points[a].max() for a in "xyz"], dtype=np.float32)
    ext = np.abs(mx - mn)  # float
    bb_min = (mn.astype(np.float64) - bb_expand_factor * ext.astype(np.float64)).astype(np.float32)
    bb_max = (mx.astype(np.float64) + bb_expand_factor * ext.astype(np.float64)).astype(np.float32)
    dyn = bb_max - bb_min
    for i, a in enumerate("xyz"):
        points[a] = (points[a] - bb_min[i]) / dyn[i]
    return bb_min, bb_max


def _finish(xyz, seed, n, do_normalize):
    pts = np.zeros(n, dtype=POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    if do_normalize:
        normalize(pts)
    # colour: r = 255 x, g = 255 y, b = 255 z (+ U{-8..8}), clamped
    noise = (splitmix64(seed, 3 * n, 7) % np.uint64(17)).astype(np.int64).reshape(n, 3) - 8
    col = np.stack([pts["x"], pts["y"], pts["z"]], 1).astype(np.float64) * 255.0
    col = np.clip(np.floor(col).astype(np.int64) + noise, 0, 255).astype(np.uint32)
    pts["rgba"] = col[:, 2] | (col[:, 1] << 8) | (col[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


def sphere_shell(n, seed, centre=(0.5, 0.5, 0.5), radius=0.3, noise=0.002, do_normalize=True):
    """Surface-like frame (cfg1 / cfg2-surface / cfg3 fallback / cfg5)."""
    u = _u01(seed, n, 0)
    v = _u01(seed, n, 1)
    w = _u01(seed, n, 2)
    ct = 2.0 * u - 1.0
    st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
    ph = 2.0 * np.pi * v
    r = radius + (2.0 * w - 1.0) * noise
    xyz = np.stack([centre[0] + r * st * np.cos(ph), centre[1] + r * st * np.sin(ph), centre[2] + r * ct], 1)
    return _finish(xyz.astype(np.float32), seed, n, do_normalize)


def uniform_volume(n, seed, do_normalize=True):
    """Dense frame: uniform random in the unit cube (cfg2-uniform / cfg4)."""
    xyz = np.stack([_u01(seed, n, 0), _u01(seed, n, 1), _u01(seed, n, 2)], 1)
    return _finish(xyz.astype(np.float32), seed, n, do_normalize)


def voxelised_body(n, seed, grid=1024, do_normalize=True):
    """A capture-like frame in the style of the 8i voxelised full bodies (cfg3 stand-in): integer coordinates on a
    `grid`^3 lattice, one point per occupied voxel of a closed surface (an ellipsoid "torso" with a bumpy radius),
    stored in raster order of the lattice (z, y, x) -- spatially coherent, unlike the shuffled shells above --
    with a smooth colour field."""
    # sample more surface points than needed, snap to the lattice, keep the first n distinct voxels in raster order
    m = int(n * 1.6) + 1000
    u, v = _u01(seed, m, 0), _u01(seed, m, 1)
    ct = 2.0 * u - 1.0
    st = np.sqrt(np.maximum(0.0, 1.0 - ct * ct))
    ph = 2.0 * np.pi * v
    r = 1.0 + 0.08 * np.sin(5 * ph) * st + 0.05 * np.cos(7 * np.arccos(ct))
    half = grid / 2.0
    x = half + 0.23 * grid * r * st * np.cos(ph)
    y = half + 0.23 * grid * r * st * np.sin(ph)
    z = half + 0.46 * grid * r * ct
    q = np.stack([np.floor(x), np.floor(y), np.floor(z)], 1).astype(np.int64)
    q = np.clip(q, 0, grid - 1)
    code = (q[:, 2] * grid + q[:, 1]) * grid + q[:, 0]
    code = np.unique(code)            # distinct voxels, raster order
    if len(code) > n:
        keep = np.sort(np.argsort(splitmix64(seed, len(code), 3))[:n])   # thin out evenly, keep the order
        code = code[keep]
    xyz = np.stack([code % grid, (code // grid) % grid, code // (grid * grid)], 1).astype(np.float32)
    pts = np.zeros(len(xyz), dtype=POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    if do_normalize:
        normalize(pts)
    # smooth colours with a little texture
    t = xyz / grid
    col = np.stack([128 + 100 * np.sin(6.0 * t[:, 0] + 2.0 * t[:, 2]), 128 + 100 * np.cos(5.0 * t[:, 1]), 60 + 180 * t[:, 2]], 1)
    noise = (splitmix64(seed, 3 * len(xyz), 7) % np.uint64(9)).astype(np.int64).reshape(len(xyz), 3) - 4
    col = np.clip(np.floor(col).astype(np.int64) + noise, 0, 255).astype(np.uint32)
    pts["rgba"] = col[:, 2] | (col[:, 1] << 8) | (col[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


def delta_pair(n, seed, grid=256, jitter=0.15):
    """An (I frame, P frame) pair for the inter-frame path: the I frame is a voxelised body on a `grid`^3 lattice; the P
    frame is the same surface a moment later -- the whole body turned and shifted by a fraction of a voxel, the top
    third moved much further (its macroblocks cannot be predicted), every point jittered by `jitter` voxels, some points
    dropped, colours re-noised and slightly brightened.  Both frames live in the same normalised [0,1]^3 box."""
    i_cloud = voxelised_body(n, seed, grid=grid)
    m = len(i_cloud)
    xyz = np.stack([i_cloud["x"], i_cloud["y"], i_cloud["z"]], 1).astype(np.float64)
    vox = 1.0 / grid
    ang = 0.004
    rot = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    moved = (xyz - 0.5) @ rot.T + 0.5 + np.array([0.3, -0.2, 0.25]) * vox
    top = xyz[:, 2] > 0.62
    moved[top] += np.array([3.7, 2.9, 0.0]) * vox * ((xyz[top, 2:3] - 0.62) * 20.0)
    moved += (np.stack([_u01(seed, m, 11), _u01(seed, m, 12), _u01(seed, m, 13)], 1) - 0.5) * (2.0 * jitter * vox)
    keep = (splitmix64(seed, m, 14) % np.uint64(100)) >= np.uint64(7)   # 7 % of the points are not seen again
    moved = np.clip(moved, 0.001, 0.999)[keep]
    p_cloud = np.zeros(len(moved), dtype=POINT_DTYPE)
    p_cloud["x"], p_cloud["y"], p_cloud["z"] = moved[:, 0].astype(np.float32), moved[:, 1].astype(np.float32), moved[:, 2].astype(np.float32)
    p_cloud["w"] = 1.0
    col = np.stack([(i_cloud["rgba"] >> 16) & 0xFF, (i_cloud["rgba"] >> 8) & 0xFF, i_cloud["rgba"] & 0xFF], 1).astype(np.int64)[keep]
    noise = (splitmix64(seed, 3 * len(col), 15) % np.uint64(7)).astype(np.int64).reshape(len(col), 3) - 3
    col = np.clip(col + noise + np.array([5, 3, -4]), 0, 255).astype(np.uint32)
    p_cloud["rgba"] = col[:, 2] | (col[:, 1] << 8) | (col[:, 0] << 16) | np.uint32(0xFF000000)
    return i_cloud, p_cloud


CONFIGS = {
    # name: (generator, n, seed, codec settings)
    "cfg1": dict(gen="sphere", n=100_000, seed=0xC1, octree_bits=8, color_bits=8, color_coding_type=1,
                 jpeg_quality=85, keep_centroid=0),
    "cfg2": dict(gen="sphere", n=1_000_000, seed=0xC2, octree_bits=10, color_bits=8, color_coding_type=1,
                 jpeg_quality=85, keep_centroid=0),
    "cfg2u": dict(gen="uniform", n=1_000_000, seed=0xC2, octree_bits=10, color_bits=8, color_coding_type=1,
                  jpeg_quality=85, keep_centroid=0),
    "cfg3": dict(gen="sphere", n=800_000, seed=0xC3, octree_bits=10, color_bits=8, color_coding_type=1,
                 jpeg_quality=85, keep_centroid=0),
    # cfg3 with capture-like data: voxelised surface on a 1024^3 lattice in raster order (8i longdress stand-in)
    "cfg3v": dict(gen="body", n=800_000, seed=0xC3, octree_bits=10, color_bits=8, color_coding_type=1,
                  jpeg_quality=85, keep_centroid=0),
    # inter-frame family: a sphere shell whose centre moves +0.002 in x per frame, 30 frames, coded
    # as I(0), P(1|0), I(1), P(2|1), ... like the reference app with do_delta_coding=1; frames come from moving_sphere_group
    "cfg5": dict(gen="moving_sphere", n=200_000, seed=0xC5, frames=30, octree_bits=8, color_bits=8, color_coding_type=1,
                 jpeg_quality=85, keep_centroid=0, macroblock_size=16),
    "cfg4": dict(gen="uniform", n=10_000_000, seed=0xC4, octree_bits=12, color_bits=0, color_coding_type=1,
                 jpeg_quality=85, keep_centroid=0),
}


def moving_sphere_group(n, seed, frames, step=0.002, bb_expand_factor=0.2):
    """cfg5: `frames` sphere shells (same surface sample, fresh radial noise and colour noise per frame), centre moving
    +step in x per frame, normalised as ONE group like normalize_pointclouds does for a group whose first frame sets the
    box and later frames stay inside it (impl.hpp:1899-1926): every frame is mapped with the box of the union."""
    raw = [sphere_shell(n, seed, centre=(0.5 + step * f, 0.5, 0.5), noise=0.0005, do_normalize=False) for f in range(frames)]
    for f, r in enumerate(raw):   # per-frame radial jitter, deterministic
        j = ((_u01(seed + 97 * (f + 1), n, 5) - 0.5) * 0.001).astype(np.float32)
        d = np.stack([r["x"] - np.float32(0.5 + step * f), r["y"] - np.float32(0.5), r["z"] - np.float32(0.5)], 1)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r["x"] += d[:, 0] * j; r["y"] += d[:, 1] * j; r["z"] += d[:, 2] * j
    mn = np.array([min(r[a].min() for r in raw) for a in "xyz"], dtype=np.float32)
    mx = np.array([max(r[a].max() for r in raw) for a in "xyz"], dtype=np.float32)
    ext = np.abs(mx - mn)
    bb_min = (mn.astype(np.float64) - bb_expand_factor * ext.astype(np.float64)).astype(np.float32)
    bb_max = (mx.astype(np.float64) + bb_expand_factor * ext.astype(np.float64)).astype(np.float32)
    dyn = bb_max - bb_min
    for r in raw:
        for i, a in enumerate("xyz"):
            r[a] = (r[a] - bb_min[i]) / dyn[i]
    return raw


def make_frame(cfg, frame=0, n=None):
    c = CONFIGS[cfg] if isinstance(cfg, str) else cfg
    n = c["n"] if n is None else n
    seed = c["seed"] + frame
    if c["gen"] == "sphere":
        return sphere_shell(n, seed)
    if c["gen"] == "body":
        return voxelised_body(n, seed)
    return uniform_volume(n, seed)
