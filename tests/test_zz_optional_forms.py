"""The forms of the kernels that are OFF by default until an MI355X has timed them, each still held against the oracle: this
file sorts behind every other test file on purpose -- `pytest -x` on the GPU box reaches it last, so a form that has never run
on the chip cannot stop the tests of the default forms from being run and counted.

  * the hot path's optional forms (fused front end and its fallback, XCD-aware sort tickets, the two sort experiments, round
    2's probe layout): child processes, because the switches are read once per process
  * the device range coder with one LANE per stream (option "rc_device_lanes"; the default is one wave per stream)
  * the pipeline's entropy threads coding sixteen frames per call instead of four (PCC_PIPELINE_BATCH=16: the host range coder's
    AVX-512 path)
"""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [{"PCC_FUSED_KEYS": "1"}, {"PCC_FUSED_KEYS": "1", "PCC_PLAN_SPINS": "1"}, {"PCC_LEAF_PROBES": "uniform"}, {"PCC_SORT_XCD": "16"},
                                 {"PCC_SORT_XCD": "3"}, {"PCC_SORT_BARE": "1"}, {"PCC_SORT_LOCAL": "1"}, {"PCC_SORT_LOCAL": "1", "PCC_SORT_SHAPE": "wide"}])
def test_optional_forms_of_the_hot_path_give_the_same_bytes(env):
    """The clouds of test_gpu_parity.py with fused mode switched on (the default is the two-kernel form), with every wait for the plan running
    out (all chunks fall back to k_make_keys), and with the parent search of k_leaf_tile on evenly spaced first probes (round
    2's layout; the default spaces them geometrically back from the tile), and with the sort passes' tiles handed out in
    XCD-aware chunks of sixteen or three tiles (the default is one ticket counter, round 2's form), and with the payload-free sort passes enqueued first (an
    experiment: frames whose keys carry a payload are sent back once), and with the local fix-up of the low code bits in the
    leaf scan (an experiment that saves a sort pass: three passes for the headline frame; a frame with a group of equal
    high bits too long for it is sent back once; with the wide kernel shape forced, which the experiment does not exist in, it
    must quietly stay off): child processes, because the switches are read once."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider", os.path.join(root, "tests", "test_gpu_parity.py"),
                        "-k", "fused_keys_read or random_sweep or test_modes_bitstream or cfg1_100k or growth or nan_points or cfg3_capture or cfg2_1m_depth10_surface or crowded_voxels"],
                       cwd=root, env=e, capture_output=True, text=True, timeout=1800)
    # with the mode off / all chunks timing out the fused-chunk counters of the first test do not hold: it is told so
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.fixture(scope="module")
def lanes_ctx(pkg):
    c = pkg.binding.Context(0)
    c.set_option("rc_device_lanes", 1)   # process-wide
    yield c
    c.set_option("rc_device_lanes", 0)
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
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
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
            assert pipe.n_contexts >= 2 * 16
            for rep in range(2):
                got = pipe.encode_host(frames, b.make_params(frame_id=3, **kw))
                assert [g[0] for g in got] == ref
        finally:
            pipe.close()
        print("OK")
    """ % root)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCC_PIPELINE_BATCH="16"), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
