import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (oracle/liboracle.so), built on demand."""
    from oracle import oracle as O
    O.lib()
    return O


@pytest.fixture(scope="session")
def pkg():
    """The product package (cwi-pcl-codec_amd/), imported by path (hyphen in the name)."""
    import __graft_entry__ as G
    return G.load_package()
