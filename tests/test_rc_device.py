"""The static range coder on the GPU (csrc/pcc_rc_device.hip) against the host coder: same bytes.  Here its default form, one
wave per stream (it has run on the chip); the opt-in form with one LANE per stream (option "rc_device_lanes": 64 streams per
wave, the coder state in vector registers, the tables in lane-private LDS columns) runs the same checks in
tests/test_zz_optional_forms.py."""
import importlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.binding.Context(0)
    yield c
    c.close()


def _streams():
    rng = np.random.default_rng(5)
    out = [b"", b"\x00", b"\x07" * 3, bytes(range(256)), rng.integers(0, 256, 4099, dtype=np.uint8).tobytes()]
    out.append((rng.integers(0, 100, 70_000) < 97).astype(np.uint8).tobytes())                 # skewed, > 65535 symbols: table rescale
    out.append(np.where(rng.integers(0, 1000, 300_000) < 995, 255, rng.integers(0, 256, 300_000)).astype(np.uint8).tobytes())  # very skewed: underflow path
    occ = (1 << rng.integers(0, 8, 200_000)).astype(np.uint8) | (1 << rng.integers(0, 8, 200_000)).astype(np.uint8)
    out.append(occ.tobytes())                                                                   # occupancy-like
    out.append(rng.integers(0, 4, 1_000_003, dtype=np.uint8).tobytes())
    return out


def test_device_range_coder_equals_host_coder(pkg, ctx):
    b = pkg.binding
    streams = _streams()
    got, ms = ctx.device_range_encode(streams)
    for s, g in zip(streams, got):
        assert g == b.host_range_encode(s), len(s)
        back, used = b.host_range_decode(g, len(s))
        assert back == s and used == len(g)
    assert ms > 0


def test_device_range_coder_many_streams_at_once(pkg, ctx):
    b = pkg.binding
    rng = np.random.default_rng(9)
    streams = [rng.integers(0, 1 + k % 7, 1000 + 37 * k, dtype=np.uint8).tobytes() for k in range(300)]
    got, _ = ctx.device_range_encode(streams)
    assert got == [b.host_range_encode(s) for s in streams]
    assert ctx.device_range_encode([])[0] == []


@pytest.mark.gpu
def test_entropy_batch_on_the_gpu_gives_the_host_stages_bitstreams(pkg, oracle):
    entropy_batch_and_pipeline(pkg, oracle, 0)


def entropy_batch_and_pipeline(pkg, oracle, lanes):
    """pcc_entropy_batch: the entropy stage of many frames with the range coders on the GPU -- every coding mode, with and
    without centroids, geometry only, a three-point frame; byte-identical to the oracle's bitstreams.  Then the frame
    pipeline in that mode (pcc_pipeline_set_option "entropy_on_gpu")."""
    import ctypes as C
    import numpy as np
    b = pkg.binding
    lib = b.load_library()
    ctx = b.Context(0)
    batch = lib.pcc_entropy_batch_create(0, 64)
    assert batch
    assert lib.pcc_entropy_batch_set_option(batch, b"rc_device_lanes", lanes) == 0   # the form of the coder this batch's flushes launch
    cases = [dict(octree_bits=8, color_coding_type=1, jpeg_quality=85), dict(octree_bits=8, color_coding_type=0, color_bits=6, keep_centroid=1),
             dict(octree_bits=7, color_coding_type=2, jpeg_quality=50), dict(octree_bits=7, color_coding_type=3, keep_centroid=1),
             dict(octree_bits=9, color_bits=0), dict(octree_bits=8, color_coding_type=1, jpeg_quality=30, keep_centroid=1)]
    want = []
    try:
        for rep in range(2):
            want.clear()
            for k, kw in enumerate(cases * 2):
                n = [30_000, 3, 7_000, 90_000][k % 4]
                pts = pkg.synthetic.sphere_shell(n, 0x600 + k)
                want.append(oracle.encode_intra(pts, oracle.make_params(frame_id=k + 1, **kw)))
                prm = b.make_params(frame_id=k + 1, **kw)
                dev = ctx.upload(pts)
                ctx.hotpath_launch(dev, n, prm)
                hot = ctx.hotpath_finish(copy=False)
                assert lib.pcc_entropy_batch_add(batch, C.byref(hot.raw), C.byref(prm)) == k
                ctx.free(dev)   # the batch holds its own copies: the context and the cloud may go
            outs = (b.Bitstream * len(want))()
            got_n = C.c_size_t()
            assert lib.pcc_entropy_batch_flush(batch, outs, len(want), C.byref(got_n)) == 0, lib.pcc_entropy_batch_last_error(batch)
            assert got_n.value == len(want) and lib.pcc_entropy_batch_size(batch) == 0
            for o, w in zip(outs, want):
                assert C.string_at(o.data, o.len) == w.bitstream and [int(x) for x in o.perf] == list(w.perf)
    finally:
        lib.pcc_entropy_batch_destroy(batch)
        ctx.close()
    # the pipeline with its entropy stage on the GPU
    sizes = [20_000, 5_000, 31_000, 12_345, 800, 26_000, 9_999, 2, 16_000]
    frames = [pkg.synthetic.sphere_shell(n, 0x680 + i) for i, n in enumerate(sizes)]
    frames[3]["z"] = np.nan   # dropped
    kw = dict(octree_bits=8, jpeg_quality=80)
    ref, fid = [], 2
    for f in frames:
        r = oracle.encode_intra(f, oracle.make_params(frame_id=fid, **kw), keep=False)
        ref.append(b"" if r is None else r.bitstream)
        fid += 0 if r is None else 1
    pipe = b.Pipeline(0, workers=2)
    try:
        pipe.set_option("entropy_on_gpu", 1)
        pipe.set_option("rc_device_lanes", lanes)
        assert pipe.get("rc_device_lanes") == lanes
        pipe.set_option("entropy_gpu_batch", 4)   # several flushes per thread and a partial one at the end
        for rep in range(2):
            got = pipe.encode_host(frames, b.make_params(frame_id=2, **kw))
            assert [g[0] for g in got] == ref
            assert pipe.last_entropy_mode() == 1
        # the batch size may change between calls: larger (the threads' batches, made for four frames, used to refuse the fifth:
        # "the batch is full", PCC_ERR_STATE -- found by random pipeline runs) and smaller
        for gb in (7, 2):
            pipe.set_option("entropy_gpu_batch", gb)
            assert [g[0] for g in pipe.encode_host(frames, b.make_params(frame_id=2, **kw))] == ref
        pipe.set_option("entropy_on_gpu", 0)
        assert [g[0] for g in pipe.encode_host(frames, b.make_params(frame_id=2, **kw))] == ref
        assert pipe.last_entropy_mode() == 0
        # 0 or 1: the -1 that meant "decide per call" in rounds 3-4 is refused, not silently taken for the GPU coder
        for bad in (-1, 2):
            with pytest.raises(b.PccError, match="PCC_ERR_ARG"):
                pipe.set_option("entropy_on_gpu", bad)
        assert [g[0] for g in pipe.encode_host(frames[:2], b.make_params(frame_id=2, **kw))] == ref[:2] and pipe.last_entropy_mode() == 0
    finally:
        pipe.close()
