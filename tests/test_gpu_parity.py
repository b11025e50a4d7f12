"""Parity tests proper: the HIP hot path (through the C ABI) against the CPU oracle.

Integer / byte / index work: bit-exact.  The only floating point in the path is fp64 key
arithmetic and fp32 voxel centres / centroids, all individually rounded: also bit-exact.
"""
import numpy as np
import pytest

import sortform

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.binding.Context(0)
    yield c
    c.close()


def cloud(pkg, xyz, rgb=None, seed=0):
    xyz = np.asarray(xyz, dtype=np.float32)
    pts = np.zeros(len(xyz), dtype=pkg.binding.POINT_DTYPE)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    if rgb is None:
        rgb = np.random.default_rng(seed).integers(0, 256, (len(xyz), 3))
    rgb = np.asarray(rgb, dtype=np.uint32)
    pts["rgba"] = rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


def run_gpu(ctx, pts, params, stride=32, rgb_offset=16, raw=None):
    dev = ctx.upload(pts if raw is None else raw)
    try:
        ctx.hotpath_launch(dev, len(pts), params, stride=stride, rgb_offset=rgb_offset)
        hot = ctx.hotpath_finish()
        stream, perf = ctx.entropy_encode(hot.raw, params)
        simplified = ctx.output_cloud()
    finally:
        ctx.free(dev)
    return hot, stream, perf, simplified


def assert_matches_oracle(pkg, oracle, ctx, pts, **kw):
    po = oracle.make_params(**kw)
    pg = pkg.binding.make_params(**kw)
    want = oracle.encode_intra(pts, po)
    hot, stream, perf, simplified = run_gpu(ctx, pts, pg)
    assert hot.depth == want.depth
    assert np.array_equal(hot.bbox, want.bbox)
    assert (hot.n_points_in, hot.n_leaves, hot.n_branches) == (want.n_points_in, want.n_leaves, want.n_branches)
    assert np.array_equal(hot.occupancy, want.occupancy)
    assert np.array_equal(hot.occupancy_histogram, np.bincount(want.occupancy, minlength=256))   # the range coder's table input
    assert np.array_equal(hot.bgr, want.bgr)
    assert np.array_equal(hot.centroid_bytes, want.centroid_bytes)
    if kw.get("color_coding_type", 1) == 1 and kw.get("color_bits", 8) > 0:
        assert (hot.image_w, hot.image_h) == (want.image_w, want.image_h)
        assert np.array_equal(hot.snake_image, want.snake_image)
    assert simplified.tobytes() == want.simplified.tobytes()
    assert stream == want.bitstream
    assert perf == want.perf
    return hot, want


# ---------------- micro cases (the reference's edge cases) ----------------

def test_appendix_f_worked_example(pkg, oracle, ctx):
    xyz = [(0.50, 0.50, 0.50), (0.60, 0.40, 0.52), (0.95, 0.10, 0.50), (0.50, 0.50, 0.51)]
    pts = cloud(pkg, xyz, rgb=[(3, 1, 0), (33, 21, 10), (63, 41, 20), (8, 2, 1)])
    hot, _ = assert_matches_oracle(pkg, oracle, ctx, pts, octree_resolution=0.25, point_resolution=0.25,
                                   color_coding_type=0)
    assert hot.occupancy.tolist() == [0x28, 0xA0, 0x08] and hot.depth == 2


def test_single_point(pkg, oracle, ctx):
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, [(0.3, 0.4, 0.5)]), octree_bits=4)


def test_two_points_one_growth(pkg, oracle, ctx):
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, [(0.3, 0.4, 0.5), (0.9, 0.4, 0.5)]), octree_bits=6,
                          color_coding_type=0)


def test_growth_every_direction(pkg, oracle, ctx):
    xyz = [(0.5, 0.5, 0.5), (0.9, 0.5, 0.5), (0.1, 0.5, 0.5), (0.5, 0.9, 0.5), (0.5, 0.1, 0.5),
           (0.5, 0.5, 0.9), (0.5, 0.5, 0.1), (0.95, 0.95, 0.95), (0.02, 0.02, 0.02)]
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_bits=5, color_coding_type=0)


def test_empty_and_all_nan_are_dropped(pkg, ctx):
    b = pkg.binding
    for pts in (np.zeros(0, dtype=b.POINT_DTYPE), cloud(pkg, [(np.nan, 0, 0), (0, np.inf, 0), (1, 1, -np.inf)])):
        with pytest.raises(b.PccError) as e:
            ctx.encode_intra_host(pts, b.make_params())
        assert e.value.code == -3  # PCC_ERR_EMPTY: the reference drops the frame (impl.hpp:206-212)


def test_nan_points_are_skipped(pkg, oracle, ctx):
    rng = np.random.default_rng(8)
    xyz = rng.uniform(0.2, 0.8, (5000, 3)).astype(np.float32)
    xyz[rng.integers(0, 5000, 700), rng.integers(0, 3, 700)] = np.nan
    xyz[rng.integers(0, 5000, 50), 0] = np.inf
    xyz[0] = np.nan  # the first points are not finite either
    xyz[1] = -np.inf
    hot, want = assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_bits=7)
    assert hot.n_points_in < 5000


def test_duplicates_and_colour_mean_truncation(pkg, oracle, ctx):
    base = np.random.default_rng(4).uniform(0.3, 0.7, (300, 3)).astype(np.float32)
    xyz = np.repeat(base, 7, axis=0)  # 7 identical points per voxel, different colours
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz, seed=1), octree_bits=8, color_coding_type=0)
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz, seed=2), octree_bits=8, color_coding_type=0, keep_centroid=1)


def test_points_on_voxel_boundaries(pkg, oracle, ctx):
    g = (np.arange(0, 64, dtype=np.float32) / 64.0)
    xyz = np.stack(np.meshgrid(g[::4], g[::4], g[::8], indexing="ij"), -1).reshape(-1, 3)
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_bits=6, color_coding_type=0)


@pytest.mark.parametrize("res", [0.01, 0.0037, 0.37, 1.0 / 3.0])
def test_non_power_of_two_resolution(pkg, oracle, ctx, res):
    """fp64 key arithmetic must round exactly like the reference (no FMA contraction, true division)."""
    rng = np.random.default_rng(int(res * 1e6))
    sc = 40.0 * res
    xyz = (rng.normal(size=(20000, 3)) * sc + rng.normal(size=3) * sc).astype(np.float32)
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_resolution=res, point_resolution=res,
                          color_coding_type=0, keep_centroid=1)


def test_sorted_input_spreads_growth_events(pkg, oracle, ctx):
    """Scan-line ordered clouds grow the box late and across many chunks."""
    rng = np.random.default_rng(12)
    xyz = rng.uniform(0.0, 1.0, (60000, 3)).astype(np.float32)
    xyz = xyz[np.lexsort((xyz[:, 2], xyz[:, 1], xyz[:, 0]))]
    hot, _ = assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_bits=8)
    assert hot.n_epochs >= 3  # several distinct points grow the box, not just the first few


def test_large_coordinates_and_deep_tree(pkg, oracle, ctx):
    rng = np.random.default_rng(13)
    xyz = (rng.uniform(-1, 1, (30000, 3)) * 100.0 + 300.0).astype(np.float32)
    hot, _ = assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_resolution=0.01, point_resolution=0.01,
                                   color_coding_type=0)
    assert hot.depth >= 15


@pytest.mark.parametrize("bits,kw", [
    (22, dict()),                                        # the first depth past the single-word codes (63 Morton bits)
    (24, dict(keep_centroid=1)),                         # point index in the second payload, colours through it
    (27, dict(color_bits=0)),                            # geometry only: no second payload at all
    (29, dict(color_coding_type=2, jpeg_quality=75)),    # depth 31: every triple of the high word in use
    (23, dict(color_coding_type=0, color_bits=6, keep_centroid=1)),
])
def test_trees_of_22_to_31_levels(pkg, oracle, ctx, bits, kw):
    """OctreePointCloud allows 32 levels; Morton codes of more than 21 levels do not fit one 64-bit sort key.  Such
    frames run the kernels' DEEP instantiations (two-word codes: the 21 low triples are the sort key, the triples above
    its payload, index or colour ride in a second payload): same bytes as the oracle's pointer octree.  The device sends
    the first deep frame back (kErrDeep), the host enqueues the deep kernels and keeps doing so until a shallow frame
    comes by -- both transitions are part of the test, as are a sorted cloud (growth events all over it), non-finite
    points and a cloud of more than one sort tile per look-back group."""
    rng = np.random.default_rng(bits)
    scale = 4096.0
    res = scale * 2.0 ** -bits
    n = 70_000 if bits == 27 else 9_000
    xyz = rng.uniform(0.1, 0.9, (n, 3)) * scale
    pts = cloud(pkg, xyz, seed=bits)
    kw = dict(octree_resolution=res, point_resolution=res, **kw)
    hot, want = assert_matches_oracle(pkg, oracle, ctx, pts, **kw)
    assert hot.depth == want.depth and 22 <= hot.depth <= 31
    if hot.depth <= 29:   # (in another order the same cloud's box may end two levels deeper, and 31 is the end: PCL's own
        # growth arithmetic -- 1 << depth -- is undefined beyond it, and the oracle restates PCL)
        srt = pts[np.argsort(pts["x"], kind="stable")].copy()
        srt["z"][::11] = np.nan
        hot, want = assert_matches_oracle(pkg, oracle, ctx, srt, **kw)                  # still deep: enqueued directly
    shallow = cloud(pkg, rng.uniform(0.2, 0.8, (5_000, 3)), seed=1)
    hot, want = assert_matches_oracle(pkg, oracle, ctx, shallow, octree_bits=9)          # deep kernels, shallow frame
    assert hot.depth <= 21
    hot, want = assert_matches_oracle(pkg, oracle, ctx, shallow, octree_bits=9)          # back on the single-word kernels
    # the decoders: the host's walk handles any depth, and so does the GPU half (node keys of two words beyond 22 levels)
    stream = oracle.encode_intra(pts, oracle.make_params(frame_id=5, **kw)).bitstream
    ref = oracle.decode_intra(stream).points
    got, info = ctx.decode_intra(stream, on_gpu=True)
    assert info["consumed"] == len(stream) and got.tobytes() == ref.tobytes()
    assert ctx.decode_times()["gpu_ms"] > 0      # (really on the GPU: the host fallback reports 0)
    host, _ = ctx.decode_intra(stream)
    assert host.tobytes() == ref.tobytes()


def test_a_large_frame_in_a_deep_tree(pkg, oracle, ctx):
    """More than 96 sort tiles AND more than 21 levels: the narrow (512 x 8) DEEP instantiations of sort pass and leaf scan, which
    the cases above (at most 70 000 points) do not reach.  (A random campaign ran 514 such frames; this is one of them kept.)"""
    rng = np.random.default_rng(2604)
    scale = 4096.0
    xyz = (rng.uniform(0.1, 0.9, (420_000, 3)) * scale).astype(np.float32)
    pts = cloud(pkg, xyz, seed=26)
    res = scale * 2.0 ** -26
    hot, want = assert_matches_oracle(pkg, oracle, ctx, pts, octree_resolution=res, point_resolution=res, color_coding_type=1)
    assert 22 <= hot.depth <= 31
    assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, rng.uniform(0.2, 0.8, (5_000, 3)), seed=1), octree_bits=9)   # and back to shallow frames


def test_a_tree_of_32_levels_is_refused(pkg, ctx):
    """Two points 2^31 voxels apart: the box would need 32 levels (PCL's growth itself stops at 31)."""
    pts = cloud(pkg, np.array([[0.0, 0.0, 0.0], [3.0e9, 1.0, 1.0]]), seed=3)
    with pytest.raises(pkg.binding.PccError):
        ctx.encode_intra_host(pts, pkg.binding.make_params(octree_resolution=1.0, point_resolution=1.0))


def test_unaligned_stride_and_colour_offset(pkg, oracle, ctx):
    """pcc_encode_* takes stride / rgb_offset: a packed 16-byte XYZ+RGBA layout must give the same frame."""
    pts = pkg.synthetic.sphere_shell(20000, 31)
    packed = np.zeros(len(pts), dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgba", "<u4")]))
    for f in ("x", "y", "z", "rgba"):
        packed[f] = pts[f]
    kw = dict(octree_bits=8, color_coding_type=1)
    want = oracle.encode_intra(pts, oracle.make_params(**kw))
    pg = pkg.binding.make_params(**kw)
    dev = ctx.upload(packed)
    ctx.hotpath_launch(dev, len(pts), pg, stride=16, rgb_offset=12)
    hot = ctx.hotpath_finish()
    stream, _ = ctx.entropy_encode(hot.raw, pg)
    ctx.free(dev)
    assert stream == want.bitstream
    # 20-byte stride: not 16-byte aligned -> scalar load path
    odd = np.zeros(len(pts), dtype=np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("q", "<f4"), ("rgba", "<u4")]))
    for f in ("x", "y", "z", "rgba"):
        odd[f] = pts[f]
    dev = ctx.upload(odd)
    ctx.hotpath_launch(dev, len(pts), pg, stride=20, rgb_offset=16)
    hot = ctx.hotpath_finish()
    stream, _ = ctx.entropy_encode(hot.raw, pg)
    ctx.free(dev)
    assert stream == want.bitstream


def test_bad_arguments(pkg, ctx):
    b = pkg.binding
    pts = pkg.synthetic.sphere_shell(100, 1)
    dev = ctx.upload(pts)
    with pytest.raises(b.PccError) as e:
        ctx.hotpath_launch(dev, 100, b.make_params(), stride=8)
    assert e.value.code == -1
    with pytest.raises(b.PccError) as e:
        ctx.hotpath_launch(dev, 100, b.make_params(octree_resolution=0.0))
    assert e.value.code == -1
    with pytest.raises(b.PccError) as e:
        ctx.hotpath_finish()
    assert e.value.code == -6
    ctx.free(dev)


# ---------------- all coding modes ----------------

@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("centroid", [0, 1])
def test_modes_bitstream_identical(pkg, oracle, ctx, mode, centroid):
    pts = pkg.synthetic.sphere_shell(30000, 0x40 + mode)
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=7, color_bits=6 if mode == 0 else 8,
                          color_coding_type=mode, keep_centroid=centroid, jpeg_quality=75)


def test_geometry_only(pkg, oracle, ctx):
    assert_matches_oracle(pkg, oracle, ctx, pkg.synthetic.uniform_volume(40000, 3), octree_bits=7, color_bits=0)


@pytest.mark.parametrize("L", [255, 256, 257, 2047, 2048, 2049, 4096 + 8 * 256])
def test_snake_image_heights(pkg, oracle, ctx, L):
    """Image height H = L/256 + 1 hits full block rows, odd partial rows and the 256 | L extra row."""
    g = np.arange(L, dtype=np.float32)
    xyz = np.stack([(g % 64) / 64.0, ((g // 64) % 64) / 64.0, (g // 4096) / 64.0], 1) + 1.0 / 256
    hot, want = assert_matches_oracle(pkg, oracle, ctx, cloud(pkg, xyz), octree_bits=6, color_coding_type=1)
    assert hot.n_leaves == L


# ---------------- BASELINE.json configurations ----------------

def test_cfg1_100k_depth8(pkg, oracle, ctx):
    pts = pkg.synthetic.make_frame("cfg1")
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=8, color_bits=8, color_coding_type=1, jpeg_quality=85)


def test_cfg2_1m_depth10_surface(pkg, oracle, ctx):
    pts = pkg.synthetic.make_frame("cfg2")
    hot, _ = assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=10, color_bits=8, color_coding_type=1,
                                   jpeg_quality=85)
    assert hot.n_points_in == 1_000_000


def test_cfg2_1m_depth10_uniform(pkg, oracle, ctx):
    pts = pkg.synthetic.make_frame("cfg2u")
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)


def test_cfg3_gop_of_8_frames(pkg, oracle, ctx):
    """GOP=8 intra-only: frame ids 1..8, every frame an independent I-frame (reduced to 100k points/frame)."""
    enc = pkg.binding.OctreePointCloudCodecV2(pkg.binding.MANUAL_CONFIGURATION, False, 2.0 ** -10, 2.0 ** -10, True, 0,
                                              True, 8, 1, False, False, False, 85, 1)
    for f in range(8):
        pts = pkg.synthetic.make_frame("cfg3", frame=f, n=100_000)
        stream = enc.encodePointCloud(pts)
        want = oracle.encode_intra(pts, oracle.make_params(octree_bits=10, jpeg_quality=85, frame_id=f + 1), keep=False)
        assert stream == want.bitstream
        assert enc.getPerformanceMetrics() == want.perf


def test_cfg3_capture_like_voxelised_frames(pkg, oracle, ctx):
    """cfg3 stand-in for the 8i sequences: integer lattice coordinates in raster order (spatially coherent input:
    neighbouring points share their high digits), smooth colours; full size and two settings."""
    pts = pkg.synthetic.make_frame("cfg3v")
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
    small = pkg.synthetic.make_frame("cfg3v", frame=1, n=150_000)
    assert_matches_oracle(pkg, oracle, ctx, small, octree_bits=9, color_bits=8, color_coding_type=1, jpeg_quality=60, keep_centroid=1)


def test_cfg4_reduced_parity_and_full_size_properties(pkg, oracle, ctx):
    # parity at 1M points with cfg4's settings (depth 12, geometry only)
    pts = pkg.synthetic.make_frame("cfg4", n=1_000_000)
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=12, color_bits=0)
    # the full 10M-point frame through size-independent properties
    pts = pkg.synthetic.make_frame("cfg4")
    pg = pkg.binding.make_params(octree_bits=12, color_bits=0)
    hot, stream, perf, simplified = run_gpu(ctx, pts, pg)
    assert hot.n_points_in == len(pts)
    pop = int(np.unpackbits(hot.occupancy).sum())
    assert pop == hot.n_branches - 1 + hot.n_leaves          # every non-root node is somebody's child bit
    assert (hot.occupancy != 0).all()                         # no branch node without children
    dec, info = ctx.decode_intra(stream)                      # own decoder rebuilds exactly L leaves ...
    assert len(dec) == hot.n_leaves and info["consumed"] == len(stream)
    for a in "xyz":                                           # ... at the encoder's voxel centres
        assert np.array_equal(dec[a], simplified[a])
    keys = np.stack([dec[a].astype(np.float64) for a in "xyz"], 1)
    res = 2.0 ** -12
    k = np.floor((keys - hot.bbox[:3]) / res).astype(np.uint64)
    code = sortform.morton(k, hot.depth)
    assert (np.diff(code.astype(np.int64)) > 0).all()        # leaves strictly ascending in Morton order
    # every input point falls in an encoded voxel (encode -> decode covers the input)
    pk = np.floor((np.stack([pts[a].astype(np.float64) for a in "xyz"], 1)[::997] - hot.bbox[:3]) / res).astype(np.uint64)
    pc = sortform.morton(pk, hot.depth)
    assert np.isin(pc, code).all()


# ---------------- the mirrored class interface ----------------

def test_codec_class_round_trip(pkg, oracle):
    b = pkg.binding
    enc = b.OctreePointCloudCodecV2(b.MANUAL_CONFIGURATION, False, 2.0 ** -9, 2.0 ** -9, True, 0, True, 8, 1, False,
                                    False, False, 85, 1)
    dec = b.OctreePointCloudCodecV2(b.MANUAL_CONFIGURATION, False, 2.0 ** -9, 2.0 ** -9, True, 0, True, 8, 1, False,
                                    False, False, 85, 1)
    pts = pkg.synthetic.sphere_shell(50_000, 0x99)
    assert enc.encodePointCloud(np.zeros(0, dtype=b.POINT_DTYPE)) == b""   # dropped, frame id not advanced
    stream = enc.encodePointCloud(pts)
    want = oracle.encode_intra(pts, oracle.make_params(octree_bits=9, jpeg_quality=85, frame_id=1))
    assert stream == want.bitstream and enc.getPerformanceMetrics() == want.perf
    out, used = dec.decodePointCloud(stream)
    ref = oracle.decode_intra(stream)
    assert used == len(stream) and out.tobytes() == ref.points.tobytes()
    assert enc.getOutputCloud().tobytes() == want.simplified.tobytes()
    # colour PSNR of decoded vs encoder-side per-voxel colours: same number as the oracle's decode (<= 0.01 dB apart)
    def psnr(a, bb):
        mse = np.mean((a.astype(np.float64) - bb) ** 2)
        return 10 * np.log10(255.0 ** 2 / mse)
    enc_bgr = want.bgr.reshape(-1, 3)
    got = np.stack([out["rgba"] & 0xFF, (out["rgba"] >> 8) & 0xFF, (out["rgba"] >> 16) & 0xFF], 1)
    exp = np.stack([ref.points["rgba"] & 0xFF, (ref.points["rgba"] >> 8) & 0xFF, (ref.points["rgba"] >> 16) & 0xFF], 1)
    assert abs(psnr(got, enc_bgr) - psnr(exp, enc_bgr)) <= 0.01


def test_pair_sort_mode_matches_packed_mode(pkg, oracle):
    """Frames whose code + index bits exceed 64 sort (u64 code, u32 index) pairs; force that path on a small frame."""
    c = pkg.binding.Context(0)
    c.set_option("force_pairs", 1)
    try:
        rng = np.random.default_rng(77)
        xyz = rng.uniform(0.1, 0.9, (70000, 3)).astype(np.float32)
        xyz[rng.integers(0, 70000, 300)] = np.nan
        xyz = np.repeat(xyz, 2, axis=0)[rng.permutation(140000)]
        for kw in (dict(octree_bits=9, color_coding_type=1), dict(octree_bits=7, color_coding_type=0, keep_centroid=1)):
            assert_matches_oracle(pkg, oracle, c, cloud(pkg, xyz), **kw)
        # the same beyond 96 sort tiles, where sort pass and leaf scan run in their 512-thread shape (two workgroups per CU)
        xyz = rng.uniform(0.1, 0.9, (210000, 3)).astype(np.float32)
        xyz[rng.integers(0, 210000, 500)] = np.nan
        xyz = np.repeat(xyz, 2, axis=0)[rng.permutation(420000)]
        assert_matches_oracle(pkg, oracle, c, cloud(pkg, xyz), octree_bits=9, color_coding_type=1, keep_centroid=1)
    finally:
        c.close()


def test_indexed_keys_match_bare_keys(pkg, oracle):
    """Without centroids nothing reads the point index of a sorted element, so the sort keys are [code | colour] or the
    code alone (8 B per key and pass); option "force_pairs" = 2 keeps the [code | index] key + colour payload sort that frames
    with centroids use, on frames that would not need it.  Both ways give the oracle's bytes."""
    rng = np.random.default_rng(78)
    xyz = rng.uniform(0.1, 0.9, (150000, 3)).astype(np.float32)
    xyz[rng.integers(0, 150000, 300)] = np.nan
    xyz = np.repeat(xyz, 2, axis=0)[rng.permutation(300000)]
    cases = (dict(octree_bits=9, color_coding_type=1), dict(octree_bits=8, color_coding_type=0), dict(octree_bits=10, color_coding_type=2),
             dict(octree_bits=9, color_bits=0), dict(octree_bits=11, color_coding_type=0, color_bits=5))
    for mode in (0, 2):
        c = pkg.binding.Context(0)
        c.set_option("force_pairs", mode)
        try:
            for kw in cases:
                assert_matches_oracle(pkg, oracle, c, cloud(pkg, xyz), **kw)
        finally:
            c.close()


def test_every_key_layout_with_growth_events_all_over_the_cloud(pkg, oracle):
    """Every key layout the sort plan of k_boxes_events can choose ([code | colour], [code | index] + colour payload, bare
    codes, LINES colours), on clouds whose growth events are spread over the chunks (earlier epochs in later chunks: the
    tiles of k_make_keys that look their epoch up), with a first chunk without a finite point, cell ranks, non-finite
    points and ragged sizes around the 2048-point chunk: the oracle's bytes."""
    b = pkg.binding
    rng = np.random.default_rng(77)
    clouds = []
    for n in (1, 2047, 2048, 2049, 6000, 70_001):
        clouds.append((cloud(pkg, rng.uniform(0.2, 0.8, (n, 3)), seed=n), dict(octree_bits=9)))
    s = pkg.synthetic.sphere_shell(150_000, 0xF5)
    clouds.append((s, dict(octree_bits=10, keep_centroid=1)))            # index in the key, colour payload
    clouds.append((s, dict(octree_bits=10, color_bits=0)))               # bare code keys
    clouds.append((s, dict(octree_bits=9, color_coding_type=2)))
    srt = s[np.argsort(s["x"], kind="stable")]                           # growth events far into the cloud
    clouds.append((srt, dict(octree_bits=10)))
    holes = s.copy(); holes["y"][::7] = np.nan; holes["x"][:3000] = np.inf   # chunk 0 without a finite point
    clouds.append((holes, dict(octree_bits=10)))
    clouds.append((pkg.synthetic.voxelised_body(120_000, 0xB0D), dict(octree_bits=10)))   # cell ranks
    c = b.Context(0)
    try:
        for pts, kw in clouds:
            assert_matches_oracle(pkg, oracle, c, pts, **kw)
    finally:
        c.close()


def test_crowded_voxels_beside_a_surface(pkg, oracle, ctx):
    """Thousands of points inside one 4 x 4 x 4-voxel cube next to an ordinary surface (more than one sort tile of equal
    leading digits: long runs of one digit in the onesweep passes, thousands of points per leaf in k_leaf_tile): the
    oracle's bytes, and an ordinary frame behind it on the same context."""
    s = pkg.synthetic.sphere_shell(450_000, 0xC0DE)   # (more than 96 sort tiles: the narrow kernel shapes)
    rng = np.random.default_rng(9)
    crowd = cloud(pkg, 0.5 + rng.uniform(0.0, 3.5 / 1024.0, (9_000, 3)), seed=4)
    pts = np.concatenate([s[:170_000], crowd, s[170_000:]])
    assert_matches_oracle(pkg, oracle, ctx, pts, octree_bits=10)
    assert_matches_oracle(pkg, oracle, ctx, s, octree_bits=10)     # ... and an ordinary frame behind it


def test_cell_ranks_save_a_sort_pass_and_change_nothing(pkg, oracle, monkeypatch):
    """A capture-like cloud (1024-voxel lattice) that straddles a high power-of-two boundary of its adaptive box varies in
    13 key bits per axis: 39 code bits, five sort passes.  The sorted code carries the rank of the 2^m-cell instead of the
    high bits (FrameState::code_low_bits): four passes, the same bytes.  Option "no_cell_ranks" sorts the full code."""
    import ctypes as C
    b = pkg.binding
    lib = b.load_library()
    lib.pcc_debug_sort_plan.argtypes = [C.c_void_p, C.POINTER(C.c_int32)]
    pts = pkg.synthetic.make_frame("cfg3v", n=200_000)
    rng = np.random.default_rng(5)
    clouds = [pts, cloud(pkg, (rng.uniform(0.0, 1.0, (60_000, 3)) * 0.23 + 0.38).astype(np.float32))]
    plans = {}
    for mode in ("0", "1"):
        c = b.Context(0)
        c.set_option("no_cell_ranks", int(mode))
        try:
            for k, p in enumerate(clouds):
                for kw in (dict(octree_bits=10, color_coding_type=1), dict(octree_bits=10, color_coding_type=0, keep_centroid=1), dict(octree_bits=11, color_bits=0)):
                    assert_matches_oracle(pkg, oracle, c, p, **kw)
                    plan = (C.c_int32 * 5)()
                    assert lib.pcc_debug_sort_plan(c.h, plan) == 0
                    plans.setdefault(mode, []).append(list(plan))
        finally:
            c.close()
    for ranked, plain in zip(plans["0"], plans["1"]):
        assert plain[1] == plain[2] and ranked[2] == plain[2]      # the plain plan sorts every varying bit
        assert ranked[0] <= plain[0] and ranked[1] <= plain[1]
    assert any(r[0] < p[0] for r, p in zip(plans["0"], plans["1"]))  # and at least the capture-like cloud saves a pass


def test_cpp_shim_example_runs(pkg):
    """The reference-style C++ caller built against the drop-in header (g++ only) runs end to end."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cwi-pcl-codec_amd", "shim", "examples", "encode_decode")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    r = subprocess.run([exe, "200000", "9"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "decoded voxels" in r.stdout
    # decodePointCloud consumes exactly one frame per call, from seekable and from forward-only streams
    assert "3 frames decoded one by one" in r.stdout


def test_cpp_pipeline_bench_runs(pkg):
    """The sequence interface of the C ABI from a C++ caller (pcc_pipeline_encode / _encode_host, no Python around it)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "cwi-pcl-codec_amd", "shim", "examples", "pipeline_bench")
    if not os.path.exists(exe):
        subprocess.run(["make", "-s", "-C", os.path.dirname(exe)], check=True)
    r = subprocess.run([exe, "60000", "48", "8", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("frames in HBM") == 3 and r.stdout.count("frames in host memory") == 2


def test_next_frame_may_be_launched_before_the_entropy_stage(pkg, oracle):
    """include/pcc_codec.h lets pcc_entropy_encode(other_ctx, hot, ...) run while the context that produced `hot`
    already works on its next frame: nothing a pcc_hot_result points to may change before the next
    pcc_hotpath_finish on that context (the occupancy histogram used to live in the pinned FrameState, which the
    next launch overwrites asynchronously)."""
    b = pkg.binding
    gpu, out_ctx = b.Context(0), b.Context(None)
    try:
        kw = dict(octree_bits=9, jpeg_quality=85)
        frames = [pkg.synthetic.sphere_shell(150_000, 0x700 + f) for f in range(2)] + [pkg.synthetic.uniform_volume(150_000, 0x7AA)]
        devs = [gpu.upload(f) for f in frames]
        want = [oracle.encode_intra(f, oracle.make_params(frame_id=k + 1, **kw)) for k, f in enumerate(frames)]
        for rounds in range(3):
            gpu.hotpath_launch(devs[0], len(frames[0]), b.make_params(frame_id=1, **kw))
            for k in range(len(frames)):
                hot = gpu.hotpath_finish(copy=False)
                nxt = (k + 1) % len(frames)
                gpu.hotpath_launch(devs[nxt], len(frames[nxt]), b.make_params(frame_id=nxt + 1, **kw))  # frame k+1 is on the GPU ...
                import time
                time.sleep(0.002)                                                                       # ... and its FrameState has landed
                import ctypes
                hist = np.frombuffer(ctypes.string_at(hot.raw.occupancy_histogram, 1024), dtype=np.uint32)
                assert np.array_equal(hist, np.bincount(want[k].occupancy, minlength=256))
                stream, perf = out_ctx.entropy_encode(hot.raw, b.make_params(frame_id=k + 1, **kw))      # ... while frame k is coded
                assert stream == want[k].bitstream
            gpu.hotpath_finish(copy=False)
    finally:
        gpu.close(); out_ctx.close()


@pytest.mark.parametrize("L", [1, 255, 256, 2047, 2048, 2049, 4095, 4096, 4097, 6143, 6144, 9000, 40000])
def test_jpeg_stage_on_gpu_equals_host_jpeg(pkg, oracle, L):
    """The JPEG stage may run on the host (0), up to the quantised coefficients on the GPU (1) or including the
    Huffman coding on the GPU (2): always the bytes of libjpeg-turbo (via the oracle)."""
    g = np.arange(L, dtype=np.float32)
    xyz = np.stack([(g % 128) / 128.0, ((g // 128) % 128) / 128.0, (g // 16384) / 128.0], 1) + 1.0 / 512
    pts = cloud(pkg, xyz, seed=L)
    for q in (85, 30):
        kw = dict(octree_bits=7, color_coding_type=1, jpeg_quality=q)
        want = oracle.encode_intra(pts, oracle.make_params(**kw), keep=False)
        for on_gpu, copy_image in ((2, 0), (1, 0), (0, 1)):
            c = pkg.binding.Context(0)
            c.set_option("jpeg_on_gpu", on_gpu)
            c.set_option("copy_image", copy_image)
            stream, perf = c.encode_intra_host(pts, pkg.binding.make_params(**kw))
            c.close()
            assert stream == want.bitstream, (L, q, on_gpu)


@pytest.mark.parametrize("L", [1, 17, 24, 2047, 2048, 2049, 2056, 4095, 4096, 4097, 6143, 6144, 10_000, 33_000])
def test_jpeg_lines_on_gpu_equal_host_jpeg(pkg, oracle, L):
    """Colour coding type 2 (jpegcc.h:244-317): strips of 2048 voxels, the last one 2048..4095 wide, each a w x 1 JPEG.
    Coded on the GPU (k_jpeg_lines: one 1-D FDCT per block, dummy luma blocks, Huffman coding) or on the host from the
    per-voxel colours: the same bytes as the oracle's, for busy and for flat colours."""
    b = pkg.binding
    rng = np.random.default_rng(L)
    side = int(np.ceil(np.sqrt(L)))
    ij = np.stack(np.meshgrid(np.arange(side), np.arange(side), indexing="ij"), -1).reshape(-1, 2)[:L]
    xyz = np.concatenate([(2 * ij + 0.5) / 256.0 + 0.1, np.full((L, 1), 0.5)], 1)  # two voxels apart: L leaves wherever the box starts
    for busy in (True, False):
        rgb = rng.integers(0, 256, (L, 3)) if busy else np.stack([ij[:, 0] % 256, ij[:, 1] % 256, (ij[:, 0] + ij[:, 1]) % 256], 1)
        pts = cloud(pkg, xyz, rgb=rgb)
        for q in (30, 85, 100):
            kw = dict(octree_resolution=1 / 256.0, point_resolution=1 / 256.0, color_coding_type=2, jpeg_quality=q)
            want = oracle.encode_intra(pts, oracle.make_params(**kw))
            assert want.n_leaves == L
            for mode in (2, 0):
                c = b.Context(0)
                c.set_option("jpeg_on_gpu", mode)
                c.set_option("copy_image", 0)
                hot, stream, perf, _ = run_gpu(c, pts, b.make_params(**kw))
                assert bool(hot.raw.jpeg_lines_dir) == (mode == 2) and bool(hot.raw.bgr) == (mode == 0)
                assert stream == want.bitstream and perf == want.perf, (L, busy, q, mode)
                c.close()


def test_jpeg_huffman_rows_that_do_not_fit_fall_back_to_coefficients(pkg, oracle):
    """White-noise colours at quality 100: an MCU row needs more bits than its record holds, the frame is then
    Huffman-coded on the host from the coefficients -- same bytes."""
    L = 20000
    g = np.arange(L, dtype=np.float32)
    xyz = np.stack([(g % 128) / 128.0, ((g // 128) % 128) / 128.0, (g // 16384) / 128.0], 1) + 1.0 / 512
    pts = cloud(pkg, xyz, seed=7)
    kw = dict(octree_bits=7, color_coding_type=1, jpeg_quality=100)
    want = oracle.encode_intra(pts, oracle.make_params(**kw), keep=False)
    c = pkg.binding.Context(0)
    c.set_option("copy_image", 0)
    dev = c.upload(pts)
    c.hotpath_launch(dev, len(pts), pkg.binding.make_params(**kw))
    hot = c.hotpath_finish(copy=False)
    assert not hot.raw.jpeg_tiles and hot.raw.jpeg_coefs   # the fallback was taken
    stream, perf = c.entropy_encode(hot.raw, pkg.binding.make_params(**kw))
    c.close()
    assert stream == want.bitstream


@pytest.mark.parametrize("color_coding_type,keep_centroid", [(1, 0), (1, 1), (0, 1), (2, 0)])
def test_short_calls_of_every_length(pkg, oracle, color_coding_type, keep_centroid):
    """Calls of 1..6 frames on two entropy threads: the batch of a coder loop follows what is left of the call (4, 3, 2, 1
    frames in one loop), and with one or two frames in a loop the colour streams are coded next to the occupancy
    streams -- with and without a centroid stream in between, for every colour coding -- the bitstreams must stay
    those of the serial loop."""
    b = pkg.binding
    sizes = [21_000, 3_000, 30_500, 8_191, 12_000, 26_000]
    frames = [pkg.synthetic.sphere_shell(n, 0xA00 + i) for i, n in enumerate(sizes)]
    kw = dict(octree_bits=8, color_coding_type=color_coding_type, jpeg_quality=75, keep_centroid=keep_centroid)
    pipe = b.Pipeline(0, workers=2)
    ctx = pipe.context(0)
    devs = [ctx.upload(f) for f in frames]
    for count in range(1, len(frames) + 1):
        want = [oracle.encode_intra(f, oracle.make_params(frame_id=5 + i, **kw), keep=False).bitstream for i, f in enumerate(frames[:count])]
        got = pipe.encode(devs[:count], sizes[:count], b.make_params(frame_id=5, **kw))
        assert [g[0] for g in got] == want, count
    for d in devs:
        ctx.free(d)
    pipe.close()


def test_pipeline_gives_the_serial_loop_bitstreams(pkg, oracle):
    """pcc_pipeline_encode: frames of different sizes in flight on several contexts, host stage for up to four
    frames at once -- the bitstreams must be those of the reference's serial frame loop; a dropped frame (all
    NaN) in the middle yields nothing and does not consume a frame id."""
    b = pkg.binding
    sizes = [30_000, 5_000, 41_000, 12_345, 30_000, 777, 20_000, 9_999, 33_333, 2, 16_000]
    frames = [pkg.synthetic.sphere_shell(n, 0x900 + i) for i, n in enumerate(sizes)]
    frames[4]["x"] = np.nan   # this frame is dropped (impl.hpp:206-212)
    kw = dict(octree_bits=8, color_coding_type=1, jpeg_quality=80)
    want, fid = [], 3
    for f in frames:
        r = oracle.encode_intra(f, oracle.make_params(frame_id=fid, **kw), keep=False)
        want.append(b"" if r is None else r.bitstream)
        fid += 0 if r is None else 1
    assert want[4] == b"" and all(len(w) for i, w in enumerate(want) if i != 4)
    pipe = b.Pipeline(0, workers=3)
    ctx = pipe.context(0)
    devs = [ctx.upload(f) for f in frames]
    for rep in range(3):   # the second call reuses every context and the bitstream arena sized by the first
        got = pipe.encode(devs, sizes, b.make_params(frame_id=3, **kw))
        assert [g[0] for g in got] == want
        if rep == 1:       # contexts and output memory prepared up front (too small on purpose: the rest overflows
            pipe.reserve(4, 9000, 50_000)   # into per-frame buffers)
    pipe.gpu_stage_only(devs, sizes, b.make_params(frame_id=3, **kw))
    st = pipe.stats()
    assert st["frames"] == 0   # nothing went through the entropy stage in the last call
    # the same sequence handed over in HOST memory (pcc_pipeline_encode_host): ordinary numpy arrays are page-locked
    # for the time of their upload, arrays from pcc_host_alloc are used as they are; one buffer twice in one call
    lib = b.load_library()
    pinned = [b.pinned_array(lib, f) if i % 2 else f for i, f in enumerate(frames)]
    for rep in range(2):
        got = pipe.encode_host(pinned, b.make_params(frame_id=3, **kw))
        assert [g[0] for g in got] == want
    twice = [frames[0], frames[2], frames[0], frames[0], frames[2]]
    got = pipe.encode_host(twice, b.make_params(frame_id=7, **kw))
    ref = [oracle.encode_intra(f, oracle.make_params(frame_id=7 + i, **kw), keep=False).bitstream for i, f in enumerate(twice)]
    assert [g[0] for g in got] == ref
    for i, a in enumerate(pinned):
        if i % 2:
            lib.pcc_host_free(a.ctypes.data)
    for d in devs:
        ctx.free(d)
    pipe.close()


def test_two_pipelines_behind_the_multi_gpu_entry_point(pkg, oracle):
    """pcc_pipeline_create_multi with the same GPU named twice (a 1-GPU box): frame f goes to pipeline f mod 2, the
    bitstreams come back in sequence order with the serial loop's frame ids -- a dropped frame in the middle included."""
    b = pkg.binding
    sizes = [20_000, 3_000, 25_000, 8_000, 100, 12_000, 30_000]
    frames = [pkg.synthetic.sphere_shell(n, 0xB00 + i) for i, n in enumerate(sizes)]
    frames[3]["y"] = np.inf   # dropped
    kw = dict(octree_bits=8, jpeg_quality=75)
    want, fid = [], 5
    for f in frames:
        r = oracle.encode_intra(f, oracle.make_params(frame_id=fid, **kw), keep=False)
        want.append(b"" if r is None else r.bitstream)
        fid += 0 if r is None else 1
    multi = b.MultiPipeline([0, 0], 2)
    try:
        for rep in range(2):
            got = multi.encode_host(frames, b.make_params(frame_id=5, **kw))
            assert [g[0] for g in got] == want
    finally:
        multi.close()


def test_pipelines_of_a_rank_pin_inside_the_ranks_share_of_the_cores():
    """One process per GPU on one host (LOCAL_RANK 1 of LOCAL_WORLD_SIZE 2): the entropy threads of every pipeline of the rank
    get core groups inside the rank's half of the allowed cores -- a second pipeline starts behind the first and WRAPS INSIDE
    that half (it used to wrap over all cores, onto the other rank's) -- and a destroyed pipeline gives its cores back, in whatever
    order the pipelines go.
    A child process: the bookkeeping is per process."""
    import os, subprocess, sys, textwrap
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = textwrap.dedent("""
        import ctypes as C, os, sys
        sys.path.insert(0, %r)
        import __graft_entry__ as G
        b = G.load_package().binding
        lib = b.load_library()
        lib.pcc_debug_pipeline_cpus.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int), C.c_int]
        def cpus(pipe):
            out = set()
            for w in range(pipe.workers):
                buf = (C.c_int * 1024)()
                n = lib.pcc_debug_pipeline_cpus(pipe.h, w, buf, 1024)
                assert n > 0
                out |= set(buf[:min(n, 1024)])
            return out
        allowed = sorted(os.sched_getaffinity(0))
        cores = []
        for c in allowed:   # one logical CPU per physical core: the lowest allowed sibling
            try:
                first = int(open("/sys/devices/system/cpu/cpu%%d/topology/thread_siblings_list" %% c).read().replace("-", ",").split(",")[0])
            except OSError:
                first = c
            if first == c or first not in allowed:
                cores.append(c)
        span = max(len(cores) // 2, 1)
        mine = set(cores[span:2 * span])
        a = b.Pipeline(0, 1)
        ca = cpus(a)
        if len(ca) == len(allowed) or span < 2:
            print("SKIP: too few cores to pin on (%%d)" %% len(cores)); sys.exit(0)
        bb = b.Pipeline(0, 1)
        cb = cpus(bb)
        assert ca <= mine and cb <= mine, (sorted(ca), sorted(cb), sorted(mine))
        if span >= 2 * len(ca):
            assert not (ca & cb), (sorted(ca), sorted(cb))
        # not in LIFO order: A goes, B stays -- the next pipeline takes A's cores, not the live B's (a bump counter that pipelines
        # decrement handed it B's)
        a.close()
        c3 = b.Pipeline(0, 1)
        assert cpus(c3) == ca, (sorted(cpus(c3)), sorted(ca))
        if span >= 2 * len(ca):
            assert not (cpus(c3) & cb)
        bb.close(); c3.close()
        c4 = b.Pipeline(0, 1)
        assert cpus(c4) == ca, (sorted(cpus(c4)), sorted(ca))     # all cores came back
        c4.close()
        print("OK", sorted(ca), sorted(cb))
    """ % root)
    for mode in (None, "cores"):   # core groups (the default), one core per thread (on a small host the only shape with room for two pipelines)
        env = dict(os.environ, LOCAL_RANK="1", LOCAL_WORLD_SIZE="2")
        for k in ("PCC_PIPELINE_PIN", "PCC_PIPELINE_PIN_SPAN", "PCC_PIPELINE_PIN_OFFSET"):
            env.pop(k, None)
        if mode:
            env["PCC_PIPELINE_PIN"] = mode
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and ("OK" in r.stdout or "SKIP" in r.stdout), r.stdout[-2000:] + r.stderr[-2000:]


def test_host_frames_through_one_context(pkg, oracle):
    """pcc_hotpath_launch_host with and without an upload lane, pageable and pinned memory, an empty cloud."""
    b = pkg.binding
    lib = b.load_library()
    c = b.Context(0)
    lane = lib.pcc_upload_lane_create(0)
    assert lane
    try:
        kw = dict(octree_bits=9, jpeg_quality=85)
        pts = pkg.synthetic.sphere_shell(120_000, 0xA11)
        want = oracle.encode_intra(pts, oracle.make_params(frame_id=1, **kw)).bitstream
        pin = b.pinned_array(lib, pts)
        for src in (pts, pin):
            for ln in (None, lane):
                c.hotpath_launch_host(src, b.make_params(frame_id=1, **kw), lane=ln)
                hot = c.hotpath_finish(copy=False)
                stream, _ = c.entropy_encode(hot.raw, b.make_params(frame_id=1, **kw))
                assert stream == want
        # option "pack_upload": x, y, z and the colour word of every point are packed to 16 bytes on the host and half the
        # bytes cross the link; same bitstream, with centroids (the kernels go back to the points through the index) and
        # for a ragged count (the vector loop's tail)
        c.set_option("pack_upload", 1)
        for m, kw2 in ((len(pts), kw), (len(pts) - 3, dict(kw, keep_centroid=1)), (5, kw), (1, dict(kw, color_coding_type=0))):
            want2 = oracle.encode_intra(pts[:m], oracle.make_params(frame_id=1, **kw2)).bitstream
            for ln in (None, lane):
                c.hotpath_launch_host(pts[:m], b.make_params(frame_id=1, **kw2), lane=ln)
                hot = c.hotpath_finish(copy=False)
                stream, _ = c.entropy_encode(hot.raw, b.make_params(frame_id=1, **kw2))
                assert stream == want2, (m, kw2)
        c.set_option("pack_upload", 0)
        c.hotpath_launch_host(np.zeros(0, dtype=b.POINT_DTYPE), b.make_params(frame_id=1, **kw), lane=lane)
        with pytest.raises(b.PccError):
            c.hotpath_finish()
        lib.pcc_host_free(pin.ctypes.data)
    finally:
        c.close()
        lib.pcc_upload_lane_destroy(lane)


def test_contexts_on_borrowed_streams(pkg, oracle):
    """pcc_use_stream: two contexts take turns on one borrowed stream (what the pipeline's GPU-stage threads do), go
    back to their own, and a context with a frame in flight refuses to change streams."""
    b = pkg.binding
    lib = b.load_library()
    ca, cb = b.Context(0), b.Context(0)
    st = lib.pcc_stream_create(0)
    assert st
    try:
        kw = dict(octree_bits=9, jpeg_quality=85)
        prm = b.make_params(frame_id=1, **kw)
        clouds = [pkg.synthetic.sphere_shell(90_000, 0xB0 + k) for k in range(2)]
        want = [oracle.encode_intra(p, oracle.make_params(frame_id=1, **kw)).bitstream for p in clouds]
        dev = [ca.upload(clouds[0]), cb.upload(clouds[1])]
        for borrowed in (st, None, st):
            for c in (ca, cb):
                assert lib.pcc_use_stream(c.h, borrowed) == 0
            ca.hotpath_launch(dev[0], len(clouds[0]), prm)
            assert lib.pcc_use_stream(ca.h, None) == -6          # PCC_ERR_STATE: a frame is in flight
            cb.hotpath_launch(dev[1], len(clouds[1]), prm)       # queued behind ca's frame on the shared stream
            for c, w in ((cb, want[1]), (ca, want[0])):
                hot = c.hotpath_finish(copy=False)
                stream, _ = c.entropy_encode(hot.raw, prm)
                assert stream == w
    finally:
        ca.close(); cb.close()
        lib.pcc_stream_destroy(st)


# ---------------- the decoder with its data-parallel half on the GPU ----------------

@pytest.mark.parametrize("kw", [
    dict(octree_bits=9, color_coding_type=1, jpeg_quality=85),
    dict(octree_bits=9, color_coding_type=1, jpeg_quality=30, keep_centroid=1),
    dict(octree_bits=8, color_coding_type=0, color_bits=6, keep_centroid=1),
    dict(octree_bits=8, color_coding_type=2, jpeg_quality=75),
    dict(octree_bits=8, color_coding_type=3),
    dict(octree_bits=10, color_bits=0),
    dict(octree_resolution=0.0037, point_resolution=0.0037, color_coding_type=1, jpeg_quality=95),
])
def test_gpu_decode_equals_the_oracles_decoder(pkg, oracle, ctx, kw):
    """pcc_decode_intra_gpu: range decoders, JPEG Huffman decoding and the level walk over the occupancy stream on the
    host, voxel keys -> points, inverse DCT, fancy chroma upsampling, colour conversion and un-snaking on the GPU.
    Bit-exact against oracle.decode_intra and against the product's host decoder, sizes from one voxel to 200 k."""
    rng = np.random.default_rng(len(str(kw)))
    for n in (1, 9, 300, 5_000, 70_000, 200_000):
        if "octree_resolution" in kw:
            pts = cloud(pkg, rng.normal(size=(n, 3)) * 0.4 + 2.0, seed=n)
        else:
            pts = pkg.synthetic.sphere_shell(n, 0xDEC0 + n) if n > 9 else cloud(pkg, rng.uniform(0.2, 0.8, (n, 3)), seed=n)
        stream = oracle.encode_intra(pts, oracle.make_params(frame_id=3, **kw)).bitstream
        ref = oracle.decode_intra(stream).points
        got, info = ctx.decode_intra(stream + b"trailing bytes of the next frame", on_gpu=True)
        host, _ = ctx.decode_intra(stream)
        assert info["consumed"] == len(stream)
        assert got.tobytes() == ref.tobytes() and host.tobytes() == ref.tobytes(), (kw, n)
    with pytest.raises(pkg.binding.PccError):
        ctx.decode_intra(stream[:len(stream) // 2], on_gpu=True)
    with pytest.raises(pkg.binding.PccError):
        ctx.decode_intra(b"no frame here", on_gpu=True)


def test_gpu_decode_accepts_the_2048_wide_strip_a_reference_encoder_writes(pkg, oracle, ctx):
    """Colour coding type 2, fewer than 2048 voxels, as the reference's own encoder writes it (one strip 2048 wide, the
    tail an over-read: jpegcc.h:256-275): pcc_decode_intra_gpu and pcc_decode_intra use the first L pixels like
    decodeJPEGLines does (jpegcc.h:319-344)."""
    for n, seed in ((40, 1), (600, 2), (1500, 3)):
        pts = pkg.synthetic.sphere_shell(n, 0x11E5 + seed)
        stream, strip, want = sortform.reference_style_lines_stream(oracle, pts, octree_bits=6, jpeg_quality=75, frame_id=2)
        ref = oracle.decode_intra(stream).points
        got, info = ctx.decode_intra(stream + b"next frame", on_gpu=True)
        host, _ = ctx.decode_intra(stream)
        assert info["consumed"] == len(stream) and len(got) == want.n_leaves
        assert got.tobytes() == ref.tobytes() and host.tobytes() == ref.tobytes()


def test_gpu_decode_of_the_headline_frame(pkg, oracle, ctx):
    """cfg2 (1 M points, 10-bit octree, JPEG snake): decoded cloud identical to the oracle's; the times of the two halves
    are reported by pcc_get_decode_times."""
    pts = pkg.synthetic.make_frame("cfg2")
    stream = oracle.encode_intra(pts, oracle.make_params(octree_bits=10, jpeg_quality=85, frame_id=1)).bitstream
    ref = oracle.decode_intra(stream).points
    got, _ = ctx.decode_intra(stream, on_gpu=True)
    assert got.tobytes() == ref.tobytes()
    t = ctx.decode_times()
    assert t["total_ms"] >= t["host_sequential_ms"] > 0 and t["gpu_ms"] > 0


# ---------------- randomised sweep ----------------

def _random_case(pkg, seed, sizes=(1, 2, 3, 17, 255, 256, 257, 1000, 4095, 4096, 4097, 9000, 30000, 70000)):
    """A random frame and codec configuration: size, shape, point order, non-finite points, duplicates, resolution
    (power of two or not), colour mode, colour bits, centroids."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice(list(sizes)))
    shape = rng.integers(0, 6)
    if shape == 0:      # volume
        xyz = rng.uniform(0.0, 1.0, (n, 3))
    elif shape == 1:    # clusters
        c = rng.uniform(0.1, 0.9, (8, 3))
        xyz = c[rng.integers(0, 8, n)] + rng.normal(size=(n, 3)) * 0.02
    elif shape == 2:    # lattice with many coincident points
        xyz = rng.integers(0, 40, (n, 3)) / 64.0
    elif shape == 3:    # a line: one long axis, deep tree for few points
        t = rng.uniform(0, 1, n)
        xyz = np.stack([t, 0.3 + 0.001 * t, 0.7 - 0.002 * t], 1)
    elif shape == 4:    # surface, large offset and scale (not normalised)
        u, v = rng.uniform(0, 1, n), rng.uniform(0, 1, n)
        xyz = np.stack([50 + 30 * u, -20 + 25 * v, 5 + 3 * np.sin(6 * u) * np.cos(5 * v)], 1)
    else:               # far outlier first: the box starts somewhere else and has to grow a lot
        xyz = rng.uniform(0.4, 0.6, (n, 3))
        xyz[0] = (7.5, -3.0, 0.01)
    xyz = xyz.astype(np.float32)
    order = rng.integers(0, 3)
    if order == 1:
        xyz = xyz[np.lexsort((xyz[:, 0], xyz[:, 1], xyz[:, 2]))]
    elif order == 2:
        xyz = xyz[::-1].copy()
    if n > 16 and rng.integers(0, 3) == 0:
        bad = rng.integers(0, n, max(1, n // 50))
        xyz[bad, rng.integers(0, 3, len(bad))] = rng.choice([np.nan, np.inf, -np.inf], len(bad))
    fin = xyz[np.isfinite(xyz).all(axis=1)]
    ext = float((fin.max(axis=0) - fin.min(axis=0)).max()) if len(fin) > 1 else 1.0
    ext = max(ext, 1e-3)
    if rng.integers(0, 2):
        res = ext / float(2 ** rng.integers(3, 11))
        res = 2.0 ** round(np.log2(res))
    else:
        res = ext / float(rng.uniform(8, 900))
    mode = int(rng.integers(0, 4))
    kw = dict(octree_resolution=float(res), point_resolution=float(res), color_coding_type=mode,
              color_bits=int(rng.choice([8, 8, 6, 4, 0])) if mode == 0 else int(rng.choice([8, 8, 0])),
              keep_centroid=int(rng.integers(0, 2)), jpeg_quality=int(rng.choice([30, 75, 85, 95])), frame_id=int(rng.integers(1, 1000)))
    return cloud(pkg, xyz, seed=seed), kw


@pytest.mark.parametrize("seed", range(48))
def test_random_sweep(pkg, oracle, ctx, seed):
    pts, kw = _random_case(pkg, seed)
    po = oracle.make_params(**kw)
    want = oracle.encode_intra(pts, po)
    if want is None:   # nothing finite: dropped on both sides
        with pytest.raises(pkg.binding.PccError):
            ctx.encode_intra_host(pts, pkg.binding.make_params(**kw))
        return
    if want.depth > 31:   # (PCL's own box growth stops there; the two-word codes hold 31 levels)
        with pytest.raises(pkg.binding.PccError):
            ctx.encode_intra_host(pts, pkg.binding.make_params(**kw))
        return
    assert_matches_oracle(pkg, oracle, ctx, pts, **kw)
    # ... and back: the decoder with its data-parallel half on the GPU gives the oracle's cloud
    ref = oracle.decode_intra(want.bitstream).points
    got, info = ctx.decode_intra(want.bitstream, on_gpu=True)
    assert info["consumed"] == len(want.bitstream) and got.tobytes() == ref.tobytes()
