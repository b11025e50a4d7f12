"""The host stages (range coder, JPEG, rigid-transform coding, frame assembly, decoder on truncated / corrupted streams)
under AddressSanitizer + UBSan, then under ThreadSanitizer (the host decoder uses a second thread): tools/sanitize/host_fuzz.cpp.  CPU only; GPU sanitizers are not available on the pool."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_stages_under_asan_ubsan():
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    p = subprocess.run(["bash", os.path.join(ROOT, "tools", "sanitize", "run.sh")], capture_output=True, text=True, timeout=1200)
    if p.returncode != 0 and ("cannot find -lasan" in p.stderr + p.stdout or "cannot find -ltsan" in p.stderr + p.stdout):
        pytest.skip("g++ without the sanitizer runtime")
    assert p.returncode == 0, (p.stdout + p.stderr)[-4000:]
    assert p.stdout.count("frames / decoder robustness ok") == 2  # ASan + UBSan build, TSan build
