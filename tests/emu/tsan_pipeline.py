#!/usr/bin/env python3
"""The frame pipeline's HOST side (pcc_pipeline.cpp, pcc_api.cpp: GPU-stage threads, entropy threads, upload lane, free lists,
batches, the multi-GPU front) under ThreadSanitizer, with the kernels on the CPU executor -- TEST INFRASTRUCTURE:

    make -C tests/emu tsan
    LD_PRELOAD=$(cat tests/emu/_build_tsan/runtime.txt) TSAN_OPTIONS="halt_on_error=0 log_path=/tmp/tsan" \
        PCC_LIB=tests/emu/_build_tsan/libpcc_emu_tsan.so python tests/emu/tsan_pipeline.py

Host frames and device frames through one pipeline (four entropy threads), the entropy stage on the GPU in batches of four, two
pipelines behind the multi-GPU entry point; every bitstream against the oracle.  tests/test_emu_race.py runs it and wants no report."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as G
from oracle import oracle as O
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
sizes = [20_000, 5_000, 31_000, 12_345, 800, 26_000, 9_999, 2, 16_000, 7_000, 22_000, 3_000]
frames = [syn.sphere_shell(n, 0x680 + i) for i, n in enumerate(sizes)]
frames[3]["z"] = np.nan
kw = dict(octree_bits=8, jpeg_quality=80)
ref, fid = [], 2
for f in frames:
    r = O.encode_intra(f, O.make_params(frame_id=fid, **kw), keep=False)
    ref.append(b"" if r is None else r.bitstream); fid += 0 if r is None else 1
pipe = b.Pipeline(0, workers=4)
try:
    for rep in range(3):
        got = pipe.encode_host(frames, b.make_params(frame_id=2, **kw))
        assert [g[0] for g in got] == ref
    ctx0 = pipe.context(0)
    dev = [ctx0.upload(f) for f in frames if np.isfinite(f["z"]).any()]
    lens = [len(f) for f in frames if np.isfinite(f["z"]).any()]
    for rep in range(2):
        res = pipe.encode(dev, lens, b.make_params(frame_id=2, **kw), copy=False)
    print("stats", pipe.stats()["frames"], pipe.last_entropy_mode())
    pipe.set_option("entropy_on_gpu", 1); pipe.set_option("entropy_gpu_batch", 4)
    got = pipe.encode_host(frames, b.make_params(frame_id=2, **kw))
    assert [g[0] for g in got] == ref
finally:
    pipe.close()
m = b.MultiPipeline([0, 0], 2)
try:
    got = m.encode_host(frames, b.make_params(frame_id=2, **kw))
    assert [g[0] for g in got] == ref
finally:
    m.close()
print("pipeline under ThreadSanitizer: done")
