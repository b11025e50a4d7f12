"""The evaluation app (cwi-pcl-codec_amd/apps/evaluate_compression, SURVEY.md 8f row 1 "harness parity"):
file loading without a GPU, and the whole encode -> decode -> quality -> CSV loop against the oracle on the GPU."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "cwi-pcl-codec_amd", "apps", "evaluate_compression", "evaluate_compression")

CSV_HEADER = ("compression setting; in point count;out point count;compressed_byte_size;compressed_byte_size_per_output_point;"
              "octree_byte_size_per_voxel;centroid_byte_size_per_voxel;color_byte_size_per_voxel;symm_rms;symm_haussdorff;"
              "psnr_db;psnr_colors_y;psnr_colors_u;psnr_colors_v;encoding_time_ms;decoding_time_ms;")  # quality_metrics_impl.hpp:266-285


def _raw_frames(pkg, n_frames=3, n=20_000):
    """Un-normalised frames in metres, like a capture: a shell whose centre drifts a little."""
    return [pkg.synthetic.sphere_shell(n + 137 * f, 0xE0 + f, centre=(1.5 + 0.01 * f, -0.4, 2.0), radius=0.8, do_normalize=False)
            for f in range(n_frames)]


def _write_ply(path, pts, binary):
    with open(path, "wb") as fh:
        fh.write(("ply\nformat %s 1.0\nelement vertex %d\nproperty float x\nproperty float y\nproperty float z\n"
                  "property uchar red\nproperty uchar green\nproperty uchar blue\nelement face 0\n"
                  "property list uchar int vertex_indices\nend_header\n" % ("binary_little_endian" if binary else "ascii", len(pts))).encode())
        r, g, b = (pts["rgba"] >> 16) & 0xFF, (pts["rgba"] >> 8) & 0xFF, pts["rgba"] & 0xFF
        if binary:
            rec = np.zeros(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("r", "u1"), ("g", "u1"), ("b", "u1")])
            rec["x"], rec["y"], rec["z"], rec["r"], rec["g"], rec["b"] = pts["x"], pts["y"], pts["z"], r, g, b
            fh.write(rec.tobytes())
        else:
            for i in range(len(pts)):
                fh.write(("%.9g %.9g %.9g %d %d %d\n" % (pts["x"][i], pts["y"][i], pts["z"][i], r[i], g[i], b[i])).encode())


def _write_pcd(path, pts, binary):
    with open(path, "wb") as fh:
        fh.write(("# .PCD v0.7\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F %s\nCOUNT 1 1 1 1\nWIDTH %d\nHEIGHT 1\n"
                  "VIEWPOINT 0 0 0 1 0 0 0\nPOINTS %d\nDATA %s\n" % ("F" if binary else "U", len(pts), len(pts), "binary" if binary else "ascii")).encode())
        if binary:
            rec = np.zeros(len(pts), dtype=[("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("rgb", "<u4")])
            rec["x"], rec["y"], rec["z"], rec["rgb"] = pts["x"], pts["y"], pts["z"], pts["rgba"] & 0xFFFFFF
            fh.write(rec.tobytes())
        else:
            for i in range(len(pts)):
                fh.write(("%.9g %.9g %.9g %d\n" % (pts["x"][i], pts["y"][i], pts["z"][i], pts["rgba"][i] & 0xFFFFFF)).encode())


def _fnv(pts):
    h = 1469598103934665603
    words = np.stack([pts["x"].view(np.uint32), pts["y"].view(np.uint32), pts["z"].view(np.uint32), pts["rgba"] & 0xFFFFFF], 1).reshape(-1)
    for w in words.tolist():
        h = ((h ^ w) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def test_app_reads_ply_and_pcd_ascii_and_binary(pkg, tmp_path):
    if not os.path.exists(APP):
        pytest.skip("app not built")
    frames = _raw_frames(pkg, 4, 1500)
    d = tmp_path / "in"
    d.mkdir()
    _write_ply(str(d / "a0.ply"), frames[0], binary=False)
    _write_ply(str(d / "a1.ply"), frames[1], binary=True)
    _write_pcd(str(d / "a2.pcd"), frames[2], binary=False)
    _write_pcd(str(d / "a3.pcd"), frames[3], binary=True)
    (d / "notes.txt").write_text("ignored")
    out = subprocess.run([APP, "--list_only", str(d)], cwd=str(tmp_path), capture_output=True, text=True, check=True).stdout
    rows = [l.split() for l in out.strip().splitlines()]
    assert [os.path.basename(r[0]) for r in rows] == ["a0.ply", "a1.ply", "a2.pcd", "a3.pcd"]
    for r, f in zip(rows, frames):
        assert int(r[1]) == len(f) and int(r[2]) == _fnv(f), r[0]


def test_app_rejects_unknown_options(tmp_path):
    if not os.path.exists(APP):
        pytest.skip("app not built")
    p = subprocess.run([APP, "--no_such_option=1", str(tmp_path)], cwd=str(tmp_path), capture_output=True, text=True)
    assert p.returncode != 0 and "Unrecognized options on command line" in p.stderr


@pytest.mark.gpu
def test_app_group_loop_matches_the_oracle(pkg, oracle, tmp_path):
    """Three frames in one group: normalisation (bounding box kept over the group), frame ids 1..3, per-frame sizes and
    byte counts equal to the oracle's bitstreams, quality columns equal to the numpy restatement, decoded .ply written."""
    from oracle import quality_oracle as Q
    frames = _raw_frames(pkg, 3, 20_000)
    d = tmp_path / "in"
    d.mkdir()
    (tmp_path / "out").mkdir()
    for i, f in enumerate(frames):
        _write_ply(str(d / ("frame_%02d.ply" % i)), f, binary=True)
    (tmp_path / "parameter_config.txt").write_text("octree_bits=7 # from the config file\njpeg_quality=85\ncolor_bits=8\ndo_quality_computation=1\n")
    subprocess.run([APP, "-b", "8", "--output_directory=out", "-i", str(d)], cwd=str(tmp_path), check=True, capture_output=True)

    # the same loop with the oracle: normalise the group, encode with ids 1.., decode, metric
    work = [f.copy() for f in frames]
    lib = pkg.binding.load_library()
    import ctypes as C
    ptrs = (C.c_void_p * 3)(*[w.ctypes.data for w in work])
    sizes = (C.c_size_t * 3)(*[len(w) for w in work])
    mn, mx = np.zeros(3, np.float32), np.zeros(3, np.float32)
    assert lib.pcc_normalize_group_boxes(ptrs, sizes, 3, 0.2, mn.ctypes.data, mx.ctypes.data, None) == 0
    lines = open(tmp_path / "intra_frame_quality.csv").read().splitlines()
    assert lines[0] == CSV_HEADER
    assert len(lines) == 4
    for i, w in enumerate(work):
        r = oracle.encode_intra(w, oracle.make_params(octree_bits=8, color_bits=8, color_coding_type=1, jpeg_quality=85, frame_id=i + 1))
        dec = oracle.decode_intra(r.bitstream).points
        m = Q.quality_metrics(w, dec)
        col = lines[i + 1].split(";")
        assert col[0] == "octree_bits=8 color_bits=8 enh._bits=0_colortype=1 centroid=0"     # the command line beats the config file
        assert int(col[1]) == len(w) and int(col[2]) == r.n_leaves and int(col[3]) == len(r.bitstream)
        assert float(col[4]) == pytest.approx(len(r.bitstream) / r.n_leaves, rel=1e-5)
        assert float(col[5]) == pytest.approx(r.perf[0] / r.n_leaves, rel=1e-5)
        assert float(col[6]) == 0.0 and float(col[7]) == pytest.approx(r.perf[2] / r.n_leaves, rel=1e-5)
        for k, name in ((8, "symm_rms"), (9, "symm_hausdorff"), (10, "psnr_db")):
            assert float(col[k]) == pytest.approx(m[name], rel=1e-5), name
        for k in range(3):
            assert float(col[11 + k]) == pytest.approx(m["psnr_yuv"][k], abs=1e-4)              # 6 significant digits in the CSV
        # decoded .ply = decoded cloud scaled back with the group's box (eval.hpp:846)
        back = dec.copy()
        assert lib.pcc_restore_scaling(back.ctypes.data, len(back), mn.ctypes.data, mx.ctypes.data) == 0
        ply = open(tmp_path / "out" / ("pointcloud_%d.ply" % i)).read().split("end_header\\n")[-1]
        got = np.loadtxt(str(tmp_path / "out" / ("pointcloud_%d.ply" % i)), skiprows=11)
        assert got.shape == (r.n_leaves, 6)
        assert np.allclose(got[:, 0], back["x"], rtol=1e-5, atol=1e-6) and np.array_equal(got[:, 3].astype(np.uint32), (back["rgba"] >> 16) & 0xFF)


@pytest.mark.gpu
def test_app_delta_coding_branch(pkg, tmp_path):
    """do_delta_coding: every frame but the last of a group also predicts its successor (eval.hpp:853-890): one line per
    predicted frame in the predictive csv, one delta_decoded_pc_<n>.ply each, sizes as the codec class reports them."""
    i_cloud, p_cloud = pkg.synthetic.delta_pair(20_000, 9, grid=256)
    d = tmp_path / "in"
    d.mkdir()
    (tmp_path / "out").mkdir()
    _write_ply(str(d / "f0.ply"), i_cloud, binary=True)
    _write_ply(str(d / "f1.ply"), p_cloud, binary=True)
    p = subprocess.run([APP, "-b", "8", "--color_bits=8", "--color_coding_type=1", "--jpeg_quality=85", "--group_size=2", "--bb_expand_factor=0.0",
                        "--do_delta_coding=1", "--do_quality_computation=1", "--output_directory=out", "-i", str(d)],
                       cwd=str(tmp_path), capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    assert " delta coding frame nr 1" in p.stdout
    lines = open(tmp_path / "predictive_quality.csv").read().splitlines()
    assert lines[0] == CSV_HEADER and len(lines) == 2
    col = lines[1].split(";")
    # the same through the class interface: I frame = the decoded (simplified) first frame
    B = pkg.binding
    codec = B.OctreePointCloudCodecV2(B.MANUAL_CONFIGURATION, False, 1 / 256, 1 / 256, True, 0, True, 8, 1, False, False, False, 85)
    codec.encodePointCloud(i_cloud)
    i_simplified = codec.getOutputCloud()
    _, i_data, p_data = codec.encodePointCloudDeltaFrame(i_simplified, p_cloud, False, False)
    assert codec.getOutputCloud().tobytes() == i_simplified.tobytes()   # the residual coder is a separate object (impl.hpp:1089)
    assert int(col[1]) == len(p_cloud)
    assert int(col[3]) == len(i_data) + len(p_data)
    dec = codec.decodePointCloudDeltaFrame(codec.getOutputCloud(), i_data, p_data)
    assert int(col[2]) == len(dec)
    got = np.loadtxt(str(tmp_path / "out" / "delta_decoded_pc_1.ply"), skiprows=11)
    assert got.shape == (len(dec), 6)


@pytest.mark.gpu
def test_app_with_a_device_list_gives_the_serial_loops_results(pkg, tmp_path):
    """--devices 0,0: the group's frames are encoded up front through pcc_pipeline_create_multi (two pipelines, here on the
    same GPU), frame f on the f mod 2-th; every CSV column but the two timing columns, and the decoded files, equal
    those of the reference's serial loop (two groups, so the frame ids restart with the second group)."""
    frames = _raw_frames(pkg, 5, 15_000)
    d = tmp_path / "in"
    d.mkdir()
    for i, f in enumerate(frames):
        _write_ply(str(d / ("frame_%02d.ply" % i)), f, binary=True)
    runs = {}
    for label, extra in (("serial", []), ("multi", ["--devices=0,0"])):
        w = tmp_path / label
        w.mkdir()
        (w / "out").mkdir()
        p = subprocess.run([APP, "-b", "8", "--jpeg_quality=85", "--group_size=3", "--do_quality_computation=1", "--output_directory=out",
                            "-i", str(d)] + extra, cwd=str(w), capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        lines = open(w / "intra_frame_quality.csv").read().splitlines()
        assert len(lines) == 6
        runs[label] = ([l.split(";")[:14] for l in lines], [open(w / "out" / ("pointcloud_%d.ply" % i), "rb").read() for i in range(5)])
    assert runs["serial"] == runs["multi"]
