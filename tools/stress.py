#!/usr/bin/env python3
"""Stress run on the GPU box: thousands of frames of different sizes through the pipeline with many frames in flight,
every bitstream compared with the one a lone context produced for the same frame.  Looks for rare races in the
look-back / ticket machinery (several frames share the GPU) -- any mismatch, error or hang shows here first.
    python tools/stress.py [seconds] [gpu_threads]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2:
    os.environ["PCC_PIPELINE_GPU_THREADS"] = sys.argv[2]
import __graft_entry__ as G  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    pkg = G.load_package()
    B = pkg.binding
    sizes = [1_000_000, 333_333, 50_000, 4097, 777_777, 1_000_000, 120_000, 9]
    gens = ["cfg2", "cfg2u", "cfg3v", "cfg2", "cfg2u", "cfg3v", "cfg2", "cfg2u"]
    frames = [pkg.synthetic.make_frame(g, frame=i, n=n) for i, (g, n) in enumerate(zip(gens, sizes))]
    sizes = [len(f) for f in frames]
    prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85, frame_id=1)
    lone = B.Context(0)
    want = []
    for f in frames:
        s, _ = lone.encode_intra_host(f, prm)   # frame id 1 for all of them
        want.append(s)
    pipe = B.Pipeline(0, 16)
    devs = [pipe.context(0).upload(f) for f in frames]
    batch = 256
    seq = [devs[i % len(devs)] for i in range(batch)]
    cnt = [sizes[i % len(devs)] for i in range(batch)]
    t0 = time.time()
    done = 0
    while time.time() - t0 < seconds:
        got = pipe.encode(seq, cnt, prm)
        for i, (s, _) in enumerate(got):
            w = want[i % len(devs)]
            # the pipeline numbers the frames 1, 2, 3, ...: the u32 at offset 48 differs, nothing else
            if not (len(s) == len(w) and s[:48] == w[:48] and s[52:] == w[52:] and int.from_bytes(s[48:52], "little") == i + 1):
                print("MISMATCH at batch frame %d (kind %d) after %d frames" % (i, i % len(devs), done + i))
                sys.exit(1)
        done += batch
    print("stress ok: %d frames in %.1f s (%.0f frames/s incl. the comparisons), no mismatch" % (done, time.time() - t0, done / (time.time() - t0)))


if __name__ == "__main__":
    main()
