"""cwi-pcl-codec_amd: MI355X-native intra-frame hot path of the CWI point-cloud codec.

The product is the C-ABI shared library built from csrc/ (libpcc_hip.so, declared in
include/pcc_codec.h).  This Python package only holds the ctypes binding used by the
tests and bench.py (binding.py, which also mirrors the reference class interface), and
the deterministic synthetic frame generator.  The directory name contains a hyphen, so
it is imported by path: see __graft_entry__.load_package().
"""
from . import binding, sharding, synthetic  # noqa: F401
