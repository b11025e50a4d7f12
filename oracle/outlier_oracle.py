"""CPU ORACLE (test infrastructure, not product code): remove_outliers (impl.hpp:1840-1866), i.e. pcl::RadiusOutlierRemoval
with setNegative(false): a point stays if at least `min_points` other points lie within `radius`.

PARITY STATUS: PCL is not in the reference tree ("parity unpinned"); restated from PCL 1.10's
RadiusOutlierRemoval::applyFilterIndices: the dense branch keeps a point iff the squared distance to its
(min_points + 1)-th nearest neighbour (the query itself included) is <= radius^2.  Distances in float like FLANN's L2_Simple.
"""
import numpy as np
from scipy.spatial import cKDTree


def remove_outliers(points, min_points, radius):
    if min_points <= 0:
        return points.copy()
    xyz = np.stack([points["x"], points["y"], points["z"]], 1).astype(np.float32)
    finite = np.isfinite(xyz).all(axis=1)
    tree = cKDTree(xyz[finite].astype(np.float64))
    ids = np.nonzero(finite)[0]
    keep = np.zeros(len(points), dtype=bool)
    r2 = np.float32(radius) * np.float32(radius)
    cand = tree.query_ball_point(xyz[finite].astype(np.float64), float(radius) * 1.001 + 1e-12)
    for row, (i, c) in enumerate(zip(ids, cand)):
        c = ids[np.array(c, dtype=np.int64)]
        c = c[c != i]
        d = xyz[c] - xyz[i]
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]).astype(np.float32) + (d[:, 2] * d[:, 2]).astype(np.float32)
        keep[i] = int((d2 <= r2).sum()) >= min_points
    return points[keep]
