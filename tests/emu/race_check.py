#!/usr/bin/env python3
"""Frames through the hot path on the executor's `race` build (tests/emu/race.cpp: a happens-before checker over every load,
store, atomic and fence of the kernel sources) -- TEST INFRASTRUCTURE, not a measurement and not the product.

    make -C tests/emu race
    python tests/emu/race_check.py [case ...]        # cases: cfg1 cfg2 cfg2u deep lines centroid decode delta quality; default: cfg1

Every frame is also held against the oracle (the instrumented build must still give its bytes).  Prints one block per
distinct report -- kernel, memory space, the two accesses (kind, source line, workgroup, wave), how often -- and exits 1 if
there is any.  The developer switches of csrc/pcc_dev.h (PCC_LEAF_PROBES=uniform ...) select the forms of the kernels as everywhere else."""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LIB = os.environ.get("PCC_RACE_LIB") or os.path.join(ROOT, "tests", "emu", "_build", "libpcc_emu_race.so")
os.environ["PCC_LIB"] = LIB
os.environ["PCC_EMU_RACE"] = "1"
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import __graft_entry__ as G  # noqa: E402
from oracle import oracle as O  # noqa: E402

pkg = G.load_package()
b, syn = pkg.binding, pkg.synthetic
lib = b.load_library()
lib.pcc_emu_race_report.restype = C.c_size_t
lib.pcc_emu_race_report.argtypes = [C.c_char_p, C.c_size_t]
lib.pcc_emu_race_enabled.restype = C.c_int
assert lib.pcc_emu_race_enabled() == 1, "the checker is off"


def symbolise(pcs):
    """source lines of offsets into the library (addr2line; the build has -g)"""
    out = {}
    if not pcs:
        return out
    r = subprocess.run(["addr2line", "-e", LIB, "-f", "-C", "-i", "-a"] + [hex(p) for p in pcs], capture_output=True, text=True)
    cur = None
    for line in r.stdout.splitlines():
        if line.startswith("0x"):
            cur = int(line, 16)
            out[cur] = []
        elif cur is not None:
            out[cur].append(line.strip())
    res = {}
    for pc, lines in out.items():
        # pairs of (function, file:line); innermost first -- keep the file:line entries of the kernel sources
        locs = [l for l in lines[1::2] if ".hip" in l or "pcc_" in l]
        res[pc] = " <- ".join(os.path.basename(l.split(" ")[0]) for l in (locs or lines[1::2])[:3])
    return res


def collect(title):
    buf = C.create_string_buffer(1 << 20)
    n = lib.pcc_emu_race_report(buf, len(buf))
    rows = []
    for line in buf.value.decode().splitlines():
        head, now, then, count, alloc = [x.strip() for x in line.split("|")]
        kernel, space, offset = head.rsplit(" ", 2)
        rows.append((kernel, space, int(offset), now.split(), then.split(), int(count), alloc.split()))
    pcs = sorted({int(r[3][2], 16) for r in rows} | {int(r[4][2], 16) for r in rows})
    sym = symbolise(pcs)
    seen = (C.c_ulonglong * 4)()
    lib.pcc_emu_race_seen(seen)
    print("== %s: %d report%s   (accesses checked so far: global %d plain + %d atomic, LDS %d plain + %d atomic)" % (
        (title, n, "" if n == 1 else "s") + tuple(seen)))
    kind = {("W", "p"): "plain store", ("R", "p"): "plain load", ("W", "a"): "atomic write", ("R", "a"): "atomic load"}
    for kernel, space, offset, now, then, count, alloc in rows:
        print("  %s  [%s, byte %d%s]  x%d" % (kernel, space, offset, "" if space == "lds" else " = allocation #%s + %s of %s" % tuple(alloc), count))
        for tag, a in (("now ", now), ("then", then)):
            print("     %s %-12s workgroup %s wave %s   %s" % (tag, kind[(a[0], a[1])], a[3], a[4], sym.get(int(a[2], 16), a[2])))
    return n


def params(cfg, **over):
    kw = dict(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
              jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
    kw.update(over)
    return kw


def encode_and_compare(ctx, pts, kw, frame_id=1):
    kw = dict(kw, frame_id=frame_id)
    pg = b.make_params(**kw)
    dev = ctx.upload(pts)
    ctx.hotpath_launch(dev, len(pts), pg)
    hot = ctx.hotpath_finish()
    stream, perf = ctx.entropy_encode(hot.raw, pg)
    want = O.encode_intra(pts, O.make_params(**kw))
    assert np.array_equal(hot.occupancy, want.occupancy), "occupancy stream differs from the oracle"
    assert stream == want.bitstream, "bitstream differs from the oracle"
    return stream, hot


def run_case(name):
    ctx = b.Context(0)
    try:
        if name in syn.CONFIGS:
            cfg = syn.CONFIGS[name]
            pts = syn.make_frame(name, frame=0)
            for rep in range(2):   # the second frame runs with buffers allocated and the pass count known
                stream, hot = encode_and_compare(ctx, pts, params(cfg))
            ctx.decode_intra(stream, on_gpu=True)
        elif name == "deep":       # a tree of 26 levels: the two-word (DEEP) instantiations
            rng = np.random.default_rng(5)
            xyz = rng.uniform(0.0, 1.0, (30_000, 3))
            pts = np.zeros(len(xyz), dtype=b.POINT_DTYPE)
            pts["x"], pts["y"], pts["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
            pts["rgba"] = rng.integers(0, 1 << 24, len(xyz), dtype=np.uint32)
            kw = dict(octree_bits=26, color_bits=8, color_coding_type=1, jpeg_quality=75)
            for rep in range(2):
                encode_and_compare(ctx, pts, kw)
        elif name == "lines":      # colour mode 2: JPEG per line strip on the GPU
            pts = syn.sphere_shell(120_000, 0x11)
            encode_and_compare(ctx, pts, dict(octree_bits=9, color_bits=8, color_coding_type=2, jpeg_quality=80))
        elif name == "centroid":   # index in the key, colour payload (pair sort), centroids
            pts = syn.sphere_shell(150_000, 0xF5)
            encode_and_compare(ctx, pts, dict(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=75, keep_centroid=1))
            encode_and_compare(ctx, pts, dict(octree_bits=10, color_bits=0, color_coding_type=0, jpeg_quality=75))
        elif name == "ragged":     # small and ragged sizes, non-finite points, growth events far into the cloud
            rng = np.random.default_rng(3)
            s = syn.sphere_shell(70_001, 0xAB)
            s["y"][::7] = np.nan
            encode_and_compare(ctx, s, dict(octree_bits=9, color_bits=8, color_coding_type=1, jpeg_quality=75))
            srt = syn.sphere_shell(60_000, 0xCD)
            srt = srt[np.argsort(srt["x"], kind="stable")]
            encode_and_compare(ctx, srt, dict(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=75))
            for n in (1, 63, 2049):
                encode_and_compare(ctx, s[:n][np.isfinite(s["y"][:n])] if n > 1 else syn.sphere_shell(1, 1), dict(octree_bits=8, color_bits=8, color_coding_type=1, jpeg_quality=75))
        else:
            raise SystemExit("unknown case %r" % name)
    finally:
        ctx.close()
    return collect(name)


if __name__ == "__main__":
    cases = sys.argv[1:] or ["cfg1"]
    total = 0
    for c in cases:
        total += run_case(c)
    print("race check: %d distinct report%s over %s" % (total, "" if total == 1 else "s", " ".join(cases)))
    sys.exit(1 if total else 0)
