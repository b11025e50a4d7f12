"""The committed vectors of tests/golden/codec_golden.npz (made by tests/golden/make_codec_golden.py) against the
oracle, the product's host code (no GPU needed) and the GPU path."""
import importlib
import os

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "codec_golden.npz")


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLDEN, allow_pickle=False)


def _kw(z, name):
    out = {}
    for k, v in z[name + "/params"]:
        f = float(v)
        out[str(k)] = f if "resolution" in str(k) else int(f)
    return out


def _points(z, name, dtype):
    return np.ascontiguousarray(z[name + "/points"]).view(dtype).reshape(-1)


def test_oracle_reproduces_the_committed_vectors(oracle, gold):
    for name in gold["names"]:
        pts = _points(gold, name, oracle.POINT_DTYPE)
        r = oracle.encode_intra(pts, oracle.make_params(**_kw(gold, name)))
        assert r.bitstream == gold[name + "/bitstream"].tobytes(), name
        assert np.array_equal(r.occupancy, gold[name + "/occupancy"]) and np.array_equal(r.bgr, gold[name + "/bgr"])
        assert [r.depth, r.n_points_in, r.n_leaves, r.n_branches] == gold[name + "/header"].tolist()
        assert np.array_equal(r.bbox, gold[name + "/bbox"])


def test_product_decoder_on_the_committed_bitstreams(pkg, gold):
    """The host decoder of the product (no GPU involved) decodes the frozen bitstreams to the frozen clouds."""
    b = pkg.binding
    host = b.Context(None)
    for name in gold["names"]:
        pts, info = host.decode_intra(gold[name + "/bitstream"].tobytes())
        want = np.ascontiguousarray(gold[name + "/decoded"]).view(b.POINT_DTYPE).reshape(-1)
        assert pts.tobytes() == want.tobytes(), name
        assert info["consumed"] == len(gold[name + "/bitstream"])
        assert info["depth"] == int(gold[name + "/header"][0])


def test_product_rigid_transform_coding_on_the_committed_vectors(gold):
    binding = importlib.import_module("cwi-pcl-codec_amd.binding")
    for m, n, comp, dec in zip(gold["rigid/matrices"], gold["rigid/lengths"], gold["rigid/comp"], gold["rigid/decoded"]):
        got = binding.host_rigid_compress(m)
        assert got == comp[:n].tolist()
        back = binding.host_rigid_decompress(got)
        assert np.array_equal(back.view(np.uint32), dec.view(np.uint32)) or np.array_equal(np.isnan(back), np.isnan(dec))
    assert set(gold["rigid/lengths"].tolist()) == {6, 10}


@pytest.mark.gpu
def test_gpu_path_reproduces_the_committed_vectors(pkg, gold):
    b = pkg.binding
    ctx = b.Context(0)
    for name in gold["names"]:
        pts = _points(gold, name, b.POINT_DTYPE)
        prm = b.make_params(**_kw(gold, name))
        dev = ctx.upload(pts)
        ctx.hotpath_launch(dev, len(pts), prm)
        hot = ctx.hotpath_finish()
        stream, _ = ctx.entropy_encode(hot.raw, prm)
        ctx.free(dev)
        assert stream == gold[name + "/bitstream"].tobytes(), name
        assert np.array_equal(hot.occupancy, gold[name + "/occupancy"])
        if len(gold[name + "/bgr"]):
            assert np.array_equal(hot.bgr, gold[name + "/bgr"])
        if len(gold[name + "/centroid"]):
            assert np.array_equal(hot.centroid_bytes, gold[name + "/centroid"])
        assert [hot.depth, hot.n_points_in, hot.n_leaves, hot.n_branches] == gold[name + "/header"].tolist()
        assert np.array_equal(hot.bbox, gold[name + "/bbox"])
    ctx.close()
