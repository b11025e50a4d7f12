#!/usr/bin/env python3
"""A long random campaign of the hot path against the oracle, on the CPU executor (no GPU needed) -- what the 48 cases of
tests/test_gpu_parity.py::test_random_sweep do, for as long as one likes and with seeds nobody has looked at: random sizes,
shapes, point orders, non-finite points, resolutions (powers of two and not), colour modes, centroids; encode (every byte of
the hot path's products and the bitstream) and GPU-assisted decode.  Any seed that fails is printed and can be replayed with
`--seed N`.

    make -C tests/emu CXX=/opt/rocm/lib/llvm/bin/clang++ OUT=_build_clang OPT=-O3      # the fastest build of the executor
    PCC_LIB=tests/emu/_build_clang/libpcc_emu.so python tools/fuzz_executor.py --minutes 30 --first 1000 [--stride 8 --offset k]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
if "PCC_LIB" not in os.environ:
    raise SystemExit("set PCC_LIB to a build of the executor (this is not a measurement and not for the product library)")
import numpy as np  # noqa: E402

import __graft_entry__ as G  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_gpu_parity as T  # noqa: E402


BIG = (100_000, 196_608, 196_609, 300_000, 393_217, 450_000, 800_000)   # around and beyond 48 / 96 sort tiles: both kernel shapes, several look-back groups
SIZES = None


DEEP = False
OPTIONS = False


def one(pkg, ctx, seed):
    pts, kw = T._random_case(pkg, seed, SIZES) if SIZES else T._random_case(pkg, seed)
    if DEEP:   # the same clouds, 4096 times larger, at a resolution that gives a tree of 15 to 31 levels (two-word codes beyond 21;
        # the adaptive box may add a level or two: 32 and more are refused by the product).  Resolutions stay above 1e-6: below
        # FLT_EPSILON PCL's box arithmetic (and with it the oracle) grows the box to 60 levels and more, which nothing supports.
        pts = pts.copy()
        for ax in ("x", "y", "z"):
            pts[ax] *= np.float32(4096.0)
        xyz = np.stack([pts["x"], pts["y"], pts["z"]], 1).astype(np.float64)
        fin = xyz[np.isfinite(xyz).all(axis=1)]
        ext = max(float((fin.max(axis=0) - fin.min(axis=0)).max()) if len(fin) > 1 else 4096.0, 4.0)
        r = np.random.default_rng(seed)
        res = max(ext / 2.0 ** float(r.uniform(15.0, 30.5)), 1e-6)
        if r.integers(0, 2):
            res = 2.0 ** round(np.log2(res))
        kw = dict(kw, octree_resolution=float(res), point_resolution=float(res))
    want = O.encode_intra(pts, O.make_params(**kw))
    b = pkg.binding
    if OPTIONS:   # where the JPEG stage runs (0 host from the image, 1 coefficients on the GPU, 2 Huffman-coded rows / strips on the GPU),
        # whether the image comes back, the key layouts of the sort: any combination gives the same bytes
        r = np.random.default_rng(seed + 77)
        ctx.set_option("jpeg_on_gpu", int(r.integers(0, 3)))
        ctx.set_option("copy_image", 1)   # (assert_matches_oracle looks at the snake image)
        ctx.set_option("force_pairs", int(r.choice([0, 0, 1, 2])))
        ctx.set_option("no_cell_ranks", int(r.integers(0, 2)))
        ctx.set_option("pack_upload", int(r.integers(0, 2)))
    if want is None or want.depth > 31:
        try:
            ctx.encode_intra_host(pts, b.make_params(**kw))
        except b.PccError:
            return "refused"
        raise AssertionError("a frame the oracle drops / cannot code was accepted")
    T.assert_matches_oracle(pkg, O, ctx, pts, **kw)
    ref = O.decode_intra(want.bitstream).points
    got, info = ctx.decode_intra(want.bitstream, on_gpu=True)
    assert info["consumed"] == len(want.bitstream) and got.tobytes() == ref.tobytes(), "decoder"
    return "ok"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--first", type=int, default=1000, help="first seed (the test suite uses 0..47)")
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--offset", type=int, default=0)
    ap.add_argument("--seed", type=int, default=None, help="replay one seed")
    ap.add_argument("--verbose", action="store_true", help="print every seed before it runs (to find one that kills the process)")
    ap.add_argument("--big", action="store_true", help="frames of 100 000 to 800 000 points instead of 1 to 70 000")
    ap.add_argument("--options", action="store_true", help="a random setting of the context options (JPEG stage placement, key layouts, packed upload) per frame")
    ap.add_argument("--deep", action="store_true", help="frames of up to 9 000 points at resolutions that give trees of 15 to 31 levels")
    a = ap.parse_args()
    global SIZES, DEEP, OPTIONS
    OPTIONS = a.options
    SIZES = BIG if a.big else ((1, 2, 3, 17, 255, 256, 257, 1000, 4095, 4096, 4097, 9000) if a.deep else None)
    DEEP = a.deep
    pkg = G.load_package()
    ctx = pkg.binding.Context(0)
    if a.seed is not None:
        print(a.seed, one(pkg, ctx, a.seed))
        return
    t_end = time.time() + 60.0 * a.minutes
    seed, done, bad = a.first + a.offset, 0, []
    while time.time() < t_end:
        if a.verbose:
            print("seed", seed, flush=True)
        try:
            one(pkg, ctx, seed)
        except Exception as e:  # noqa: BLE001 -- every failure is a finding
            bad.append(seed)
            print("FAILED seed %d: %s: %s" % (seed, type(e).__name__, str(e)[:300]), flush=True)
            ctx.close()
            ctx = pkg.binding.Context(0)
        done += 1
        seed += a.stride
    print("seeds %d..%d step %d: %d cases, %d failed %s" % (a.first + a.offset, seed - a.stride, a.stride, done, len(bad), bad), flush=True)


if __name__ == "__main__":
    main()
