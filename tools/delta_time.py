"""Timing of the inter-frame path on a capture-like pair (voxelised body, 1024^3 lattice): wall time and HIP-event time of
pcc_encode_delta / pcc_decode_delta, sizes, block statistics.  Run on the GPU box:  python tools/delta_time.py [n] [reps] [cfg5]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as G  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 800_000
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    pkg = G.load_package()
    B = pkg.binding
    ctx = B.Context(0)
    if len(sys.argv) > 3 and sys.argv[3] == "cfg5":   # moving sphere shell, 8-bit octree
        cfg = pkg.synthetic.CONFIGS["cfg5"]
        frames = pkg.synthetic.moving_sphere_group(n, cfg["seed"], 2)
        prm = B.make_params(octree_bits=cfg["octree_bits"], color_bits=8, color_coding_type=1, jpeg_quality=85)
        ctx.encode_intra_host(frames[0], prm)
        i_cloud, p_cloud = ctx.output_cloud(), frames[1]
    else:
        i_cloud, p_cloud = pkg.synthetic.delta_pair(n, 0xD1, grid=1024)
        prm = B.make_params(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
    best = None
    for r in range(reps):
        t0 = time.perf_counter()
        got = ctx.encode_delta(i_cloud, p_cloud, prm, write_out_cloud=False)
        wall = (time.perf_counter() - t0) * 1e3
        if best is None or wall < best[0]:
            best = (wall, got["gpu_ms"])
    dwall = dcall = None
    for r in range(reps):
        t0 = time.perf_counter()
        dec = ctx.decode_delta(i_cloud, got["i_stream"], got["p_stream"], prm)
        w = (time.perf_counter() - t0) * 1e3
        if dwall is None or w < dwall:
            dwall, dcall = w, ctx.last_call_ms
    intra, _ = ctx.encode_intra_host(p_cloud, prm)
    blocks = got["blocks"]
    icp = blocks[blocks["do_icp"] != 0]
    print("points I %d P %d simplified %d; macroblocks %d shared %d predicted %d; ICP iterations mean %.1f max %d" % (
        len(i_cloud), len(p_cloud), got["n_simplified"], got["macro_block_count"], got["shared_macroblock_count"], got["convergence_count"],
        icp["iterations"].mean() if len(icp) else 0, icp["iterations"].max() if len(icp) else 0))
    print("encode_delta: wall %.2f ms (best of %d), GPU events %.2f ms; decode_delta wall %.2f ms, of which pcc_decode_delta itself %.2f ms -> %d points" % (best[0], reps, best[1], dwall, dcall, len(dec)))
    print("bytes: intra part %d + predicted part %d = %d; the same frame intra coded %d" % (
        len(got["i_stream"]), len(got["p_stream"]), len(got["i_stream"]) + len(got["p_stream"]), len(intra)))


if __name__ == "__main__":
    main()
