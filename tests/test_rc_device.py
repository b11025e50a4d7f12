"""The static range coder on the GPU (one wave per stream, csrc/pcc_rc_device.hip) against the host coder: same bytes."""
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
