"""The forms of the kernels that are OFF by default until an MI355X has timed them, each still held against the oracle: this
file sorts behind every other test file on purpose -- `pytest -x` on the GPU box reaches it last, so a form that has never run
on the chip cannot stop the tests of the default forms from being run and counted.

  * the hot path's optional forms (fused front end and its fallback, XCD-aware sort tickets, the two sort experiments, round
    2's probe layout): child processes, because the switches are read once per process
  * the device range coder with one LANE per stream (option "rc_device_lanes"; the default is one wave per stream)
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
