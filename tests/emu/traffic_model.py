#!/usr/bin/env python3
"""Per-kernel memory traffic of one frame as the executor's traffic model sees it (tests/emu/emu.cpp, "traffic model") --
a MODEL of the access pattern of the kernels as written, not a measurement:

    make -C tests/emu traffic
    python tests/emu/traffic_model.py [workload] [frame]          # e.g. cfg2; PCC_FUSED_KEYS=1 for the fused front end, PCC_LEAF_PROBES=uniform for round 2's probes

requested = bytes the lanes asked for; first-touch = 128-byte lines touched for the first time in the launch by the workgroups
of one XCD (workgroup b -> XCD b mod 8), times 128: what has to cross an XCD's L2 at least, with L2s of unlimited size that
start every launch empty.  The frame is encoded twice; the second encode is the one reported (buffers allocated, passes known)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PCC_LIB"] = os.path.join(ROOT, "tests", "emu", "_build", "libpcc_emu_traffic.so")
os.environ["PCC_EMU_TRAFFIC"] = "1"
sys.path.insert(0, ROOT)
import __graft_entry__ as G  # noqa: E402

pkg = G.load_package()
b, syn = pkg.binding, pkg.synthetic
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
frame = int(sys.argv[2]) if len(sys.argv) > 2 else 0
cfg = syn.CONFIGS[wl]
p = b.make_params(octree_bits=cfg["octree_bits"], color_bits=cfg["color_bits"], color_coding_type=cfg["color_coding_type"],
                  jpeg_quality=cfg["jpeg_quality"], keep_centroid=cfg["keep_centroid"])
lib = b.load_library()
lib.pcc_emu_traffic_report.restype = C.c_size_t
lib.pcc_emu_traffic_report.argtypes = [C.c_char_p, C.c_size_t]
ctx = b.Context(0)
pts = syn.make_frame(wl, frame=frame)
dev = ctx.upload(pts)
buf = C.create_string_buffer(1 << 16)
for rep in range(2):
    ctx.hotpath_launch(dev, len(pts), p)
    hot = ctx.hotpath_finish(copy=False)
    lib.pcc_emu_traffic_report(buf, len(buf))
rows, tot = [], [0, 0, 0, 0]
for line in buf.value.decode().splitlines():
    name, launches, rl, rs, ll, ls = line.split()
    v = [int(rl), int(rs), int(ll), int(ls)]
    rows.append((name, int(launches), v))
    tot = [a + c for a, c in zip(tot, v)]
N, L, B = len(pts), hot.n_leaves, hot.n_branches
with_color = cfg["color_bits"] > 0
alg = 32 * N + L * ((3 if with_color else 0) + 16) + B
form = "fused keys %s, probes %s" % (os.environ.get("PCC_FUSED_KEYS", "0"), os.environ.get("PCC_LEAF_PROBES", "geometric"))
print("%s frame %d: N=%d L=%d B=%d D=%d   (%s)   algorithmic bytes of the path %.1f MB" % (wl, frame, N, L, B, hot.depth, form, alg / 1e6))
print("%-18s %8s %14s %14s %16s %16s" % ("kernel", "launches", "requested read", "requested write", "first-touch read", "first-touch write"))
for name, launches, v in rows:
    print("%-18s %8d %11.1f MB %11.1f MB %13.1f MB %13.1f MB" % (name, launches, v[0] / 1e6, v[1] / 1e6, v[2] / 1e6, v[3] / 1e6))
print("%-18s %8s %11.1f MB %11.1f MB %13.1f MB %13.1f MB   first-touch total %.1f MB" % ("frame", "", tot[0] / 1e6, tot[1] / 1e6, tot[2] / 1e6, tot[3] / 1e6, (tot[2] + tot[3]) / 1e6))
if os.environ.get("TRAFFIC_JSON"):
    json.dump({"workload": wl, "form": form, "N": N, "L": L, "B": B, "kernels": {n: {"launches": k, "requested_read": v[0], "requested_write": v[1],
               "first_touch_read": v[2], "first_touch_write": v[3]} for n, k, v in rows}}, open(os.environ["TRAFFIC_JSON"], "w"), indent=1)
