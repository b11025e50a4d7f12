"""Quality metric: the GPU nearest-neighbour metric against the numpy restatement of
computeQualityMetric, and the restatement against brute force."""
import numpy as np
import pytest


def _cloud(pkg, xyz, rgb):
    pts = np.zeros(len(xyz), dtype=pkg.binding.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    rgb = np.asarray(rgb, dtype=np.uint32)
    pts["rgba"] = rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


def test_oracle_nearest_is_brute_force_nearest():
    from oracle import quality_oracle as Q
    rng = np.random.default_rng(3)
    a = rng.random((700, 3), dtype=np.float32)
    b = rng.random((500, 3), dtype=np.float32)
    idx, d2 = Q.nearest(a, b)
    diff = a[:, None, :] - b[None, :, :]
    full = ((diff[..., 0] * diff[..., 0]).astype(np.float32) + (diff[..., 1] * diff[..., 1]).astype(np.float32)).astype(np.float32)
    full = (full + (diff[..., 2] * diff[..., 2]).astype(np.float32)).astype(np.float32)
    assert np.array_equal(idx, full.argmin(axis=1)) and np.array_equal(d2, full.min(axis=1))


def test_oracle_metric_of_identical_clouds():
    import __graft_entry__ as G
    from oracle import quality_oracle as Q
    pkg = G.load_package()
    rng = np.random.default_rng(4)
    c = _cloud(pkg, rng.random((300, 3), dtype=np.float32), rng.integers(0, 256, (300, 3)))
    m = Q.quality_metrics(c, c)
    assert m["symm_rms"] == 0.0 and m["symm_hausdorff"] == 0.0 and m["psnr_yuv"][0] == np.inf


def _close(got, want):
    for k in ("in_point_count", "out_point_count"):
        assert got[k] == want[k]
    for k in ("left_hausdorff", "right_hausdorff", "symm_hausdorff"):
        assert got[k] == pytest.approx(want[k], rel=1e-6), k   # one float sqrt
    for k in ("left_rms", "right_rms", "symm_rms", "psnr_db"):
        assert got[k] == pytest.approx(want[k], rel=2e-6), k   # double sums in a different order, float results
    for g, w in zip(got["psnr_yuv"], want["psnr_yuv"]):
        assert g == pytest.approx(w, abs=1e-6)                 # dB; the north star's colour tolerance is 0.01 dB


@pytest.mark.gpu
@pytest.mark.parametrize("n,bits", [(2000, 5), (60_000, 7), (300_000, 9)])
def test_gpu_metric_of_encoded_then_decoded_cloud(pkg, n, bits):
    """The harness case: original (normalised) cloud against its decoded voxel cloud, cell = voxel size."""
    from oracle import quality_oracle as Q
    b = pkg.binding
    ctx = b.Context(0)
    pts = pkg.synthetic.sphere_shell(n, 0x71 + bits)
    codec_bytes, _ = ctx.encode_intra_host(pts, b.make_params(octree_bits=bits, jpeg_quality=75))
    dec, _ = ctx.decode_intra(codec_bytes)
    want = Q.quality_metrics(pts, dec)
    for hint in (2.0 ** -bits, 0.0):
        _close(ctx.quality_metrics(pts, dec, cell_hint=hint), want)
    ctx.close()


@pytest.mark.gpu
def test_gpu_metric_of_unrelated_clouds_needs_the_growing_search(pkg):
    """Two clouds far apart compared with a tiny cell: the 3x3x3 search finds nothing, the cube grows, then the scan."""
    from oracle import quality_oracle as Q
    rng = np.random.default_rng(8)
    a = _cloud(pkg, rng.random((3000, 3), dtype=np.float32) * 0.2, rng.integers(0, 256, (3000, 3)))
    c = _cloud(pkg, rng.random((2500, 3), dtype=np.float32) * 0.2 + np.float32(0.5), rng.integers(0, 256, (2500, 3)))
    ctx = pkg.binding.Context(0)
    _close(ctx.quality_metrics(a, c, cell_hint=0.01), Q.quality_metrics(a, c))
    _close(ctx.quality_metrics(a, c), Q.quality_metrics(a, c))
    ctx.close()


@pytest.mark.gpu
def test_gpu_metric_with_ties_prefers_the_lower_index(pkg):
    """Points on a lattice midway between decoded centres: every query has equally distant neighbours."""
    from oracle import quality_oracle as Q
    g = np.arange(8, dtype=np.float32)
    grid = np.stack(np.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)
    rng = np.random.default_rng(5)
    centres = _cloud(pkg, (grid + 0.5) / 8, rng.integers(0, 256, (len(grid), 3)))
    corners = _cloud(pkg, grid[(grid > 0).all(axis=1)] / 8, rng.integers(0, 256, (int((grid > 0).all(axis=1).sum()), 3)))
    ctx = pkg.binding.Context(0)
    _close(ctx.quality_metrics(corners, centres, cell_hint=1 / 8), Q.quality_metrics(corners, centres))
    ctx.close()
