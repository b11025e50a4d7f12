"""Oracle checks for the frame-level path (C1/C2/C6/C7/C8, P1-P6)."""
import struct

import numpy as np
import pytest

import sortform


def cloud(oracle, xyz, rgb=None):
    pts = np.zeros(len(xyz), dtype=oracle.POINT_DTYPE)
    xyz = np.asarray(xyz, dtype=np.float32)
    pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    pts["w"] = 1.0
    if rgb is None:
        rgb = [(10 * i % 256, (20 * i + 1) % 256, (30 * i + 3) % 256) for i in range(len(xyz))]
    rgb = np.asarray(rgb, dtype=np.uint32)
    pts["rgba"] = rgb[:, 2] | (rgb[:, 1] << 8) | (rgb[:, 0] << 16) | np.uint32(0xFF000000)
    return pts


APPENDIX_F = [(0.50, 0.50, 0.50), (0.60, 0.40, 0.52), (0.95, 0.10, 0.50), (0.50, 0.50, 0.51)]


def test_appendix_f_worked_example(oracle):
    """Hand-derived micro-case: one growth event, a shared voxel."""
    pts = cloud(oracle, APPENDIX_F, rgb=[(3, 1, 0), (33, 21, 10), (63, 41, 20), (8, 2, 1)])
    r = oracle.encode_intra(pts, oracle.make_params(octree_resolution=0.25, point_resolution=0.25, color_coding_type=0))
    eps = float(np.float32(np.finfo(np.float32).eps))
    assert r.depth == 2 and r.n_leaves == 3 and r.n_branches == 3 and r.n_points_in == 4
    assert r.bbox.tolist() == [0.25, -0.25, -0.25, 0.25 + 1.0 - eps, -0.25 + 1.0 - eps, -0.25 + 1.0 - eps]
    assert r.occupancy.tolist() == [0x28, 0xA0, 0x08]
    assert r.leaf_keys.tolist() == [[1, 2, 3], [1, 3, 3], [2, 1, 3]]  # Morton 0o35, 0o37, 0o53
    assert r.leaf_counts.tolist() == [1, 2, 1]
    # (b, g, r) per leaf; the shared voxel holds p0 and p3: truncating integer mean
    assert r.bgr.tolist() == [10, 21, 33, (0 + 1) // 2, (1 + 2) // 2, (3 + 8) // 2, 20, 41, 63]
    # simplified cloud = voxel centres, default alpha 255
    assert r.simplified["x"].tolist() == [0.625, 0.625, 0.875]
    assert r.simplified["rgba"][1] == (0 | (1 << 8) | (5 << 16) | (0xFF << 24))


def test_header_layout_140_bytes(oracle):
    pts = cloud(oracle, APPENDIX_F)
    p = oracle.make_params(octree_bits=3, color_bits=6, color_coding_type=1, jpeg_quality=85, frame_id=9,
                           keep_centroid=1, macroblock_size=8, do_icp_color_offset=1, create_scalable=1)
    r = oracle.encode_intra(pts, p)
    bs = r.bitstream
    assert bs[:28] == b"<PCL-OCT-CODECV2-COMPRESSED>" and bs[28:48] == b"<PCL-OCT-COMPRESSED>"
    frame_id, i_frame, vg, wc, count, res, cbits, pres = struct.unpack_from("<IBBBQdBd", bs, 48)
    assert (frame_id, i_frame, vg, wc, count, res, cbits, pres) == (9, 1, 1, 1, r.n_leaves, 0.125, 6, 0.125)
    assert struct.unpack_from("<6d", bs, 80) == tuple(r.bbox)
    assert struct.unpack_from("<BBBIiB", bs, 128) == (1, 0, 1, 1, 8, 1)
    (occ_len,) = struct.unpack_from("<Q", bs, 140)
    assert occ_len == r.n_branches
    # perf counters: octree bytes, centroid bytes, colour bytes
    assert r.perf[0] > 1028 and r.perf[1] > 1028 and r.perf[2] > 1028
    assert len(bs) == 140 + 8 + r.perf[0] + 4 + r.perf[1] + 8 + r.perf[2]


def test_single_point_and_empty(oracle):
    pts = cloud(oracle, [(0.3, 0.4, 0.5)])
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=4))
    res = 2.0 ** -4
    assert r.depth == 1 and r.n_leaves == 1 and r.occupancy.tolist() == [0x80]  # key (1,1,1)
    x = np.float64(np.float32(0.3))
    assert r.bbox[0] == x - res and r.bbox[3] == x + res  # first-point box, no eps
    d = oracle.decode_intra(r.bitstream)
    assert d.depth == 1 and len(d.points) == 1
    # empty cloud and all-NaN cloud are dropped (impl.hpp:206-212)
    assert oracle.encode_intra(np.zeros(0, dtype=oracle.POINT_DTYPE), oracle.make_params()) is None
    assert oracle.encode_intra(cloud(oracle, [(np.nan, 0, 0), (0, np.inf, 0)]), oracle.make_params()) is None


def test_growth_every_direction(oracle):
    """9 points forcing growth towards each side; keys of earlier points get re-rooted."""
    xyz = [(0.5, 0.5, 0.5), (0.9, 0.5, 0.5), (0.1, 0.5, 0.5), (0.5, 0.9, 0.5), (0.5, 0.1, 0.5),
           (0.5, 0.5, 0.9), (0.5, 0.5, 0.1), (0.95, 0.95, 0.95), (0.02, 0.02, 0.02)]
    pts = cloud(oracle, xyz)
    res = 2.0 ** -5
    r = oracle.encode_intra(pts, oracle.make_params(octree_resolution=res, point_resolution=res, color_coding_type=0))
    s = sortform.encode_geometry(pts, res)
    assert r.depth == s["depth"] and np.array_equal(r.bbox, s["bbox"])
    assert np.array_equal(r.occupancy, s["occupancy"])
    # every point lies in the voxel its key names
    d = oracle.decode_intra(r.bitstream)
    centres = np.stack([d.points["x"], d.points["y"], d.points["z"]], 1).astype(np.float64)
    for q in np.asarray(xyz, dtype=np.float32).astype(np.float64):
        assert (np.abs(centres - q).max(axis=1) <= res / 2 + 1e-6).any()


def test_nan_points_skipped_and_boundary_point(oracle):
    xyz = [(0.5, 0.5, 0.5), (np.nan, 0.5, 0.5), (0.5 + 2.0 ** -6, 0.5, 0.5), (0.5, np.inf, 0.2)]
    pts = cloud(oracle, xyz)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=6, color_coding_type=0))
    assert r.n_points_in == 2 and r.n_leaves == 2


@pytest.mark.parametrize("seed", range(6))
def test_pointer_octree_equals_sort_formulation(oracle, seed):
    """Cross-check the two formulations (pointer insertion vs epoch keys + sort + closed-form DFS)."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(50, 4000))
    sc = 10 ** rng.uniform(-1, 1)
    xyz = (rng.normal(size=(n, 3)) * sc + rng.normal(size=3) * sc).astype(np.float32)
    if seed % 2:
        xyz[rng.integers(0, n, n // 7)] = np.nan
    pts = cloud(oracle, xyz, rgb=rng.integers(0, 256, (n, 3)))
    res = float(rng.choice([0.01, 0.37, 2.0 ** -6])) * sc
    r = oracle.encode_intra(pts, oracle.make_params(octree_resolution=res, point_resolution=res, color_coding_type=0))
    s = sortform.encode_geometry(pts, res)
    assert r.depth == s["depth"] and np.array_equal(r.bbox, s["bbox"])
    assert np.array_equal(r.leaf_keys, s["leaf_keys"].astype(np.uint32))
    assert np.array_equal(r.leaf_counts, s["leaf_counts"])
    assert np.array_equal(r.occupancy, s["occupancy"])
    assert np.array_equal(r.bgr, s["bgr"])


def test_structural_invariants(oracle, pkg):
    pts = pkg.synthetic.sphere_shell(30_000, 0xC1)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=8))
    # popcount of all occupancy bytes = number of non-root nodes = (B - 1) + L
    pop = int(np.unpackbits(r.occupancy).sum())
    assert pop == r.n_branches - 1 + r.n_leaves
    assert r.leaf_counts.sum() == r.n_points_in == len(pts)
    # leaves are in strictly increasing Morton order
    code = sortform.morton(r.leaf_keys.astype(np.uint64), r.depth)
    assert (np.diff(code.astype(np.int64)) > 0).all()
    # snake image geometry (jpegcc.h:194-200)
    assert r.image_w == 256 and r.image_h == r.n_leaves // 256 + 1


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("centroid", [0, 1])
def test_encode_decode_round_trip(oracle, pkg, mode, centroid):
    pts = pkg.synthetic.sphere_shell(5000, 0x51 + mode)
    p = oracle.make_params(octree_bits=6, color_coding_type=mode, keep_centroid=centroid, jpeg_quality=90)
    r = oracle.encode_intra(pts, p)
    d = oracle.decode_intra(r.bitstream)
    assert d.consumed == len(r.bitstream)
    assert len(d.points) == r.n_leaves and d.depth == r.depth
    assert d.params["color_coding_type"] == mode and d.params["do_voxel_centroid"] == centroid
    if not centroid:
        # decoded geometry = voxel centres = the encoder's simplified cloud, bit for bit
        for a in "xyz":
            assert np.array_equal(d.points[a], r.simplified[a])
    else:
        res = 2.0 ** -6
        for a in "xyz":
            assert np.abs(d.points[a].astype(np.float64) - r.simplified[a]).max() <= 0.0011 + 1e-7
            assert np.abs(d.points[a].astype(np.float64) - r.simplified[a]).max() < res
    dec_bgr = np.stack([d.points["rgba"] & 0xFF, (d.points["rgba"] >> 8) & 0xFF, (d.points["rgba"] >> 16) & 0xFF], 1)
    enc_bgr = r.bgr.reshape(-1, 3)
    assert (d.points["rgba"] >> 24 == 0).all()  # decoded alpha byte is 0 (ColorCoding::decodePoints)
    if mode in (0, 3):
        assert np.array_equal(dec_bgr, enc_bgr)
    else:  # JPEG: lossy (noisy per-voxel colours), but in the right ballpark at q=90
        assert np.abs(dec_bgr.astype(int) - enc_bgr).mean() < 20.0


def test_color_bits_reduce_only_in_mode0(oracle, pkg):
    pts = pkg.synthetic.sphere_shell(2000, 3)
    r0 = oracle.encode_intra(pts, oracle.make_params(octree_bits=5, color_bits=5, color_coding_type=0))
    r8 = oracle.encode_intra(pts, oracle.make_params(octree_bits=5, color_bits=8, color_coding_type=0))
    r1 = oracle.encode_intra(pts, oracle.make_params(octree_bits=5, color_bits=5, color_coding_type=1))
    assert np.array_equal(r0.bgr, r8.bgr >> 3)
    assert np.array_equal(r1.bgr, r8.bgr)
    d = oracle.decode_intra(r0.bitstream)
    assert np.array_equal((d.points["rgba"] & 0xFF).astype(np.uint8), (r0.bgr[0::3] << 3).astype(np.uint8))


def test_geometry_only(oracle, pkg):
    pts = pkg.synthetic.uniform_volume(3000, 4)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=6, color_bits=0))
    assert r.bgr.size == 0 and r.perf[2] == 0 and r.bitstream[54] == 0
    assert (r.simplified["rgba"] == 0xFF000000).all()
    d = oracle.decode_intra(r.bitstream)
    assert (d.points["rgba"] == 0x00FFFFFF).all()  # ColorCoding::setDefaultColor


def test_normalize_matches_numpy(oracle, pkg):
    raw = pkg.synthetic.sphere_shell(4000, 77, centre=(1.5, -2.0, 0.3), radius=0.9, do_normalize=False)
    a = raw.copy()
    b = raw.copy()
    mn, mx = oracle.normalize_single(a, 0.2)
    mn2, mx2 = pkg.synthetic.normalize(b, 0.2)
    assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2)
    for ax in "xyz":
        assert np.array_equal(a[ax], b[ax])
        assert abs(a[ax].min() - 0.2 / 1.4) < 1e-6 and abs(a[ax].max() - 1.2 / 1.4) < 1e-6


def test_pin_recipe_stays_runnable(oracle):
    """oracle/pin_with_pcl.sh + pin_check.py + ref_codec_driver.cpp are the recipe that pins the PCL-inherited parts of
    the oracle the day a PCL installation exists (none here: DESIGN.md (c)).  What can be checked without PCL: the
    checker imports, its clouds encode with the oracle, the driver exports what the checker binds, and the build line
    names the reference's own sources where they lie (nothing copied)."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("pin_check", os.path.join(root, "oracle", "pin_check.py"))
    pc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pc)
    cases = pc.extra_cases()
    assert len(cases) == 4
    name, pts, kw = cases[0]
    r = oracle.encode_intra(pts, oracle.make_params(**kw))
    assert r.n_leaves > 10_000 and r.n_branches > 2 ** 16 // 8
    drv = open(os.path.join(root, "oracle", "ref_codec_driver.cpp")).read()
    for sym in ("ref_encode", "ref_stream", "ref_cloud_size", "ref_cloud", "ref_decode"):
        assert sym + "(" in drv
    assert "#include <pcl/cloud_codec_v2/impl/point_cloud_codec_v2_impl.hpp>" in drv
    sh = open(os.path.join(root, "oracle", "pin_with_pcl.sh")).read()
    assert '"$REF/jpeg_io/src/jpeg_io.cpp"' in sh and "ref_codec_driver.cpp" in sh
    assert not any(line.strip().startswith(("cmake", "make ")) for line in sh.splitlines())   # a plain compiler line, not the reference's build system
    # the dry run: one command that says whether the recipe could run here (file list, real headers or none)
    if os.path.isdir("/root/reference"):
        import subprocess
        r = subprocess.run(["bash", os.path.join(root, "oracle", "pin_with_pcl.sh"), "--dry-run"], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "MISSING" not in r.stdout
        assert "type check" in r.stdout
