#!/usr/bin/env python3
"""Random inter-frame (P-frame) pairs through pcc_encode_delta / pcc_decode_delta against oracle/delta_oracle.py, on the CPU
executor: random sizes, lattice sizes, octree resolutions, macroblock sizes, jitter, colour offsets, centroids, ICP on the
original or the simplified cloud.  The oracle is given the transforms the product's ICP produced (ICP itself is "parity
unpinned"): everything else -- macroblocks of both frames, shared-block search, gates, colour offsets, chunk stream, residual
intra stream, predicted cloud, decoder output -- is compared bit for bit.

    PCC_LIB=tests/emu/_build_clang/libpcc_emu.so python tools/fuzz_delta.py --minutes 10 --first 1000"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "PCC_LIB" not in os.environ:
    raise SystemExit("set PCC_LIB to a build of the executor")
import numpy as np  # noqa: E402

import __graft_entry__ as G  # noqa: E402
from oracle import delta_oracle as D  # noqa: E402
import test_delta_gpu as T  # noqa: E402


def one(pkg, ctx, seed):
    b, syn = pkg.binding, pkg.synthetic
    r = np.random.default_rng(seed)
    n = int(r.choice([600, 3000, 12000, 40000]))
    grid = int(r.choice([64, 128, 256]))
    bits = int(np.log2(grid)) - int(r.integers(0, 2))        # the codec's voxels: the lattice or twice as coarse
    mb = int(r.choice([4, 8, 16, 32]))
    jitter = float(r.choice([0.0, 0.15, 0.4]))
    keep_centroid, colour_offset, on_original = int(r.integers(0, 2)), int(r.integers(0, 2)), bool(r.integers(0, 2))
    i_cloud, p_cloud = syn.delta_pair(n, int(r.integers(1, 1 << 30)), grid=grid, jitter=jitter)
    res = 2.0 ** -bits
    prm = b.make_params(octree_bits=bits, color_bits=8, color_coding_type=1, keep_centroid=keep_centroid, jpeg_quality=85,
                        macroblock_size=mb, do_icp_color_offset=colour_offset)
    got = ctx.encode_delta(i_cloud, p_cloud, prm, icp_on_original=on_original)
    want = D.encode_delta(i_cloud, p_cloud, res, res, macroblock_size=mb, keep_centroid=keep_centroid, do_icp_color_offset=bool(colour_offset),
                          icp_on_original=on_original, icp_fn=T._replay(got))
    blocks = got["blocks"]
    assert len(blocks) == len(want["blocks"]) == got["macro_block_count"], "macroblock count"
    for g, w in zip(blocks, want["blocks"]):
        assert tuple(g["key"][:3]) == w["key"] and g["n_p"] == w["n_p"] and (g["i_block"] >= 0) == w["shared"] and bool(g["do_icp"]) == w["icp"], "block"
        assert list(g["rgb_offsets"][:3]) == list(w["offsets"]), "colour offsets"
    assert got["p_stream"] == want["p_stream"], "chunk stream"
    assert got["i_stream"] == want["i_stream"], "residual intra stream"
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes(), "predicted cloud"
    dec = ctx.decode_delta(i_cloud, got["i_stream"], got["p_stream"], prm)
    wdec = D.decode_delta(i_cloud, got["i_stream"], got["p_stream"], res, macroblock_size=mb, do_icp_color_offset=bool(colour_offset))
    assert dec.tobytes() == wdec.tobytes(), "decoder"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--first", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    pkg = G.load_package()
    ctx = pkg.binding.Context(0)
    if a.seed is not None:
        one(pkg, ctx, a.seed)
        print(a.seed, "ok")
        return
    t_end, seed, done, bad = time.time() + 60.0 * a.minutes, a.first, 0, []
    while time.time() < t_end:
        try:
            one(pkg, ctx, seed)
        except Exception as e:  # noqa: BLE001
            bad.append(seed)
            print("FAILED seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:300]), flush=True)
            ctx.close()
            ctx = pkg.binding.Context(0)
        done += 1
        seed += 1
    print("delta: seeds %d..%d: %d pairs, %d failed %s" % (a.first, seed - 1, done, len(bad), bad[:20]), flush=True)


if __name__ == "__main__":
    main()
