"""Forms that are OFF by default until an MI355X has timed them, each still held against the oracle: this file sorts behind
every other test file on purpose -- `pytest -x` on the GPU box reaches it last, so a form that has never run on the chip cannot
stop the tests of the default forms from being run and counted.

  * round 2's layouts of k_leaf_tile (evenly spaced first probes of the parent search, block row = blockIdx) and a forced
    workgroup shape of the sort: developer switches of csrc/pcc_dev.h, so the child processes load the developer build
    (libpcc_hip_dev.so; the shipped library reads no such switch)
  * the device range coder with one LANE per stream (option "rc_device_lanes" of a context, a batch, a pipeline; the default is
    one wave per stream)
  * the pipeline's entropy threads coding sixteen frames per call instead of four (PCC_PIPELINE_BATCH=16: the host range coder's
    AVX-512 path)
(The fused front end, the XCD-aware sort tickets and the two sort experiments of rounds 3-4 left the product in round 5:
branch experiments/r04-optin-forms, built as libpcc_hip_r04x.so for the A/B of a GPU session.)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev_library():
    """the developer build of the library under test: libpcc_hip_dev.so beside libpcc_hip.so; the CPU executor's builds (tests/emu)
    read the switches themselves"""
    cur = os.environ.get("PCC_LIB")
    if cur:
        return cur
    return os.path.join(ROOT, "cwi-pcl-codec_amd", "libpcc_hip_dev.so")


@pytest.mark.parametrize("env", [{"PCC_LEAF_PROBES": "uniform"}, {"PCC_LEAF_ROWS": "linear"}, {"PCC_SORT_SHAPE": "wide"}, {"PCC_SORT_SHAPE": "narrow"}],
                         ids=lambda e: "_".join("%s=%s" % kv for kv in e.items()))
def test_developer_forms_of_the_hot_path_give_the_same_bytes(env):
    """The clouds of test_gpu_parity.py with the parent search of k_leaf_tile on evenly spaced first probes (round 2's layout;
    the default spaces them geometrically back from the tile), with its block rows in blockIdx order (round 2's; the default
    gives every XCD one contiguous range), and with the sort passes and the leaf scan forced into ONE workgroup shape for every
    frame size (the default picks 1024 x 4 up to 96 tiles, 512 x 8 beyond): child processes on the developer build, because the
    switches are read once."""
    lib = dev_library()
    if not os.path.exists(lib):
        pytest.skip("no developer build here (make -C cwi-pcl-codec_amd/csrc dev)")
    e = dict(os.environ, PCC_LIB=lib, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_gpu_parity.py"),
                        "-k", "every_key_layout or random_sweep or test_modes_bitstream or cfg1_100k or growth or nan_points or cfg3_capture or cfg2_1m_depth10_surface or crowded_voxels"],
                       cwd=ROOT, env=e, capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.fixture(scope="module")
def lanes_ctx(pkg):
    c = pkg.binding.Context(0)
    c.set_option("rc_device_lanes", 1)   # (of this context: pcc_device_range_encode through it launches the lane form)
    yield c
    c.close()


def test_lane_per_stream_range_coder_equals_host_coder(pkg, lanes_ctx):
    import test_rc_device as T
    b = pkg.binding
    rng = np.random.default_rng(11)   # every tail length behind the 16-byte groups, streams of one wave ending at different times
    streams = T._streams() + [rng.integers(0, 16, 200 + k, dtype=np.uint8).tobytes() for k in range(0, 70)]
    got, ms = lanes_ctx.device_range_encode(streams)
    for s, g in zip(streams, got):
        assert g == b.host_range_encode(s), len(s)
    assert ms > 0


def test_lane_per_stream_range_coder_many_streams_at_once(pkg, lanes_ctx):
    b = pkg.binding
    rng = np.random.default_rng(9)
    streams = [rng.integers(0, 1 + k % 7, 1000 + 37 * k, dtype=np.uint8).tobytes() for k in range(300)]
    got, _ = lanes_ctx.device_range_encode(streams)
    assert got == [b.host_range_encode(s) for s in streams]


def test_lane_per_stream_range_coder_in_the_entropy_batch_and_the_pipeline(pkg, oracle, lanes_ctx):
    import test_rc_device as T
    T.entropy_batch_and_pipeline(pkg, oracle, 1)


def test_pipeline_with_sixteen_frames_per_coder_call(pkg):
    """PCC_PIPELINE_BATCH=16: an entropy thread takes up to sixteen ready frames and codes their streams in one call (ten and
    more through the AVX-512 lanes of the host range coder where the CPU has them).  Forty frames of different sizes, one of them
    dropped, on two entropy threads: the oracle's bitstreams, from host memory and from device memory.  A child process (the
    batch size is read when the pipeline is made)."""
    import textwrap
    root = ROOT
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import __graft_entry__ as G
        from oracle import oracle as O
        pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
        sizes = [3_000 + 700 * (i %% 9) + 40 * i for i in range(40)]
        frames = [syn.sphere_shell(n, 0x900 + i) for i, n in enumerate(sizes)]
        frames[11]["x"] = np.nan   # dropped
        kw = dict(octree_bits=7, jpeg_quality=70)
        ref, fid = [], 3
        for f in frames:
            r = O.encode_intra(f, O.make_params(frame_id=fid, **kw), keep=False)
            ref.append(b"" if r is None else r.bitstream); fid += 0 if r is None else 1
        pipe = b.Pipeline(0, workers=2)
        try:
            assert pipe.n_contexts >= 2 * 16 and pipe.get("frames_per_coder_call") == 16
            for rep in range(2):
                got = pipe.encode_host(frames, b.make_params(frame_id=3, **kw))
                assert [g[0] for g in got] == ref
        finally:
            pipe.close()
        print("OK")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCC_PIPELINE_BATCH="16"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
