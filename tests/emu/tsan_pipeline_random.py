#!/usr/bin/env python3
"""One random sequence through pcc_pipeline_encode_host, for the ThreadSanitizer build of the executor (`make -C tests/emu tsan`):
5-40 frames of 2-30 000 points, two of them dropped, 1-5 entropy threads, the entropy stage on the host or on the GPU with a batch size
that changes between the two calls; every bitstream against the oracle -- TEST INFRASTRUCTURE.

    for s in $(seq 1 20); do LD_PRELOAD=$(cat tests/emu/_build_tsan/runtime.txt) TSAN_OPTIONS="halt_on_error=0 log_path=/tmp/tsan" \
        PCC_PIPELINE_BATCH=$((s % 2 ? 4 : 16)) PCC_LIB=tests/emu/_build_tsan/libpcc_emu_tsan.so python tests/emu/tsan_pipeline_random.py $s; done

(Seed 1 is the run that found the entropy_gpu_batch bug of round 4.)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as G
from oracle import oracle as O
pkg = G.load_package(); b, syn = pkg.binding, pkg.synthetic
seed = int(sys.argv[1]); rng = np.random.default_rng(seed)
nf = int(rng.integers(5, 40))
sizes = [int(rng.choice([2, 300, 2000, 6000, 15000, 30000])) for _ in range(nf)]
frames = [syn.sphere_shell(n, 0x700 + i + seed) for i, n in enumerate(sizes)]
for k in rng.integers(0, nf, 2): frames[int(k)]["y"] = np.nan
mode = int(rng.integers(0, 4))
kw = dict(octree_bits=int(rng.integers(5, 9)), jpeg_quality=70, color_coding_type=mode, color_bits=8, keep_centroid=int(rng.integers(0, 2)))
ref, fid = [], 4
for f in frames:
    r = O.encode_intra(f, O.make_params(frame_id=fid, **kw), keep=False)
    ref.append(b"" if r is None else r.bitstream); fid += 0 if r is None else 1
workers = int(rng.choice([1, 2, 3, 5]))
pipe = b.Pipeline(0, workers=workers)
try:
    for rep in range(2):
        if rng.integers(0, 3) == 0:
            pipe.set_option("entropy_on_gpu", 1); pipe.set_option("entropy_gpu_batch", int(rng.choice([1, 3, 8])))
            pipe.set_option("rc_device_lanes", int(rng.integers(0, 2)))   # (the form of the device coder the threads' batches launch)
        else: pipe.set_option("entropy_on_gpu", 0)
        got = pipe.encode_host(frames, b.make_params(frame_id=4, **kw))
        assert [g[0] for g in got] == ref, "bitstreams"
        pipe.stats()
        assert pipe.get("workers") == workers and pipe.get("last_entropy_mode") in (0, 1)
        if rep == 0 and rng.integers(0, 2):   # a second pipeline comes and goes while this one lives (the pin ranges: not in LIFO order)
            other = b.Pipeline(0, workers=1); other.encode_host(frames[:3], b.make_params(frame_id=4, **kw)); other.close()
finally:
    pipe.close()
print("ok seed", seed, "frames", nf, "workers", workers, "batch", os.environ.get("PCC_PIPELINE_BATCH", "4"))
