"""The inter-frame (delta) path on the GPU against oracle/delta_oracle.py.

What is bit-exact: the simplified P cloud, the macroblocks of both frames (keys, order, membership), the size and colour
gates, the colour offsets, and -- given the transforms the GPU's ICP produced -- the chunk stream, the residual intra
stream and the predicted cloud; the decoder.  What is not: the ICP itself (PCL's IterativeClosestPoint is outside the
reference tree, "parity unpinned"): compared with the oracle's restatement to convergence accuracy.
"""
import numpy as np
import pytest

from oracle import delta_oracle as D

pytestmark = pytest.mark.gpu

RES = 1.0 / 256


@pytest.fixture(scope="module")
def ctx(pkg):
    c = pkg.binding.Context(0)
    yield c
    c.close()


@pytest.fixture(scope="module")
def pair(pkg):
    return pkg.synthetic.delta_pair(40000, 5, grid=256)


def _params(pkg, **kw):
    return pkg.binding.make_params(octree_bits=8, color_bits=8, color_coding_type=kw.pop("color_coding_type", 1),
                                   keep_centroid=kw.pop("keep_centroid", 0), jpeg_quality=85, **kw)


def _replay(got):
    """icp_fn for the oracle that hands back the GPU's result for each block that went to ICP, in order."""
    todo = [b for b in got["blocks"] if b["do_icp"]]
    it = iter(todo)

    def fn(src, tgt):
        b = next(it)
        assert (len(src), len(tgt)) == (b["n_i"], b["n_p"])
        return bool(b["converged"]), b["rt"].reshape(4, 4).copy(), 0.0
    return fn


@pytest.mark.parametrize("colour_offset,keep_centroid,on_original", [(0, 0, False), (1, 0, False), (0, 1, False), (1, 0, True)])
def test_delta_encode_matches_oracle_given_the_transforms(pkg, ctx, pair, colour_offset, keep_centroid, on_original):
    i_cloud, p_cloud = pair
    prm = _params(pkg, keep_centroid=keep_centroid, do_icp_color_offset=colour_offset)
    got = ctx.encode_delta(i_cloud, p_cloud, prm, icp_on_original=on_original)
    want = D.encode_delta(i_cloud, p_cloud, RES, RES, macroblock_size=16, keep_centroid=keep_centroid,
                          do_icp_color_offset=bool(colour_offset), icp_on_original=on_original, icp_fn=_replay(got))
    blocks = got["blocks"]
    assert len(blocks) == len(want["blocks"]) == got["macro_block_count"]
    for g, w in zip(blocks, want["blocks"]):
        assert tuple(g["key"][:3]) == w["key"]
        assert g["n_p"] == w["n_p"]
        assert (g["i_block"] >= 0) == w["shared"]
        assert bool(g["do_icp"]) == w["icp"]
        if w["shared"]:
            assert g["n_i"] == w["n_i"]
        assert list(g["rgb_offsets"][:3]) == list(w["offsets"])
    assert got["p_stream"] == want["p_stream"]
    assert got["i_stream"] == want["i_stream"]
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()
    assert got["n_intra_points"] == len(want["intra_points"])
    assert np.float32(got["shared_macroblock_percentage"]) == np.float32(want["shared_percentage"])
    assert np.float32(got["shared_macroblock_convergence_percentage"]) == np.float32(want["convergence_percentage"])
    if not on_original:
        assert got["n_simplified"] == len(want["simplified"])
    # both kinds of blocks occur in this pair
    n_icp = int(blocks["do_icp"].sum())
    assert 0 < n_icp < len(blocks)
    if colour_offset:
        assert np.abs(blocks["rgb_offsets"]).max() > 0


@pytest.mark.parametrize("waves", ["1", "4"])
def test_delta_icp_close_to_oracle_icp(pkg, ctx, pair, waves, request):
    """The GPU's per-block ICP (both kernel shapes: one wave per block, one workgroup per block) against the oracle's
    restatement of PCL's defaults on the same blocks."""
    ctx.set_option("icp_waves", int(waves))   # one ICP kernel shape for every macroblock (the context is shared: undone afterwards)
    request.addfinalizer(lambda: ctx.set_option("icp_waves", 0))
    i_cloud, p_cloud = pair
    got = ctx.encode_delta(i_cloud, p_cloud, _params(pkg))
    simp = D.simplify(p_cloud, RES)
    i_keys, i_lists, _, _ = D.tree(i_cloud, RES * 16)
    p_keys, p_lists, _, _ = D.tree(simp, RES * 16)
    agree = total = 0
    diffs = []
    for b, pl in zip(got["blocks"], p_lists):
        if not b["do_icp"]:
            continue
        src, tgt = D._xyz(i_cloud[i_lists[b["i_block"]]]), D._xyz(simp[pl])
        conv, final, fitness = D.icp(src, tgt)
        total += 1
        agree += int((conv and fitness < 2 * RES) == bool(b["converged"]))
        # compare the motions where they act: on the block's own points
        a = D.transform_points(src, final)
        g = D.transform_points(src, b["rt"].reshape(4, 4))
        diffs.append(float(np.abs(a - g).max()))
        assert abs(float(b["fitness"]) - fitness) <= 1e-6 + 0.05 * fitness
    assert total > 100
    assert agree == total
    # the targets sit on a lattice (voxel centres), so nearest-neighbour ties are common and two float implementations
    # of the same iteration can settle in slightly different fixed points: most blocks agree to a hundredth of a voxel,
    # the worst to a quarter
    diffs = np.sort(np.array(diffs))
    print("ICP difference in voxels: median %.4f, 90%% %.4f, max %.4f" % tuple(np.array([np.median(diffs), diffs[int(0.9 * len(diffs))], diffs[-1]]) / RES))
    assert np.median(diffs) < 0.02 * RES
    assert diffs[-1] < 0.25 * RES


def test_delta_decode_matches_oracle(pkg, ctx, pair):
    i_cloud, p_cloud = pair
    for colour_offset in (0, 1):
        prm = _params(pkg, do_icp_color_offset=colour_offset)
        got = ctx.encode_delta(i_cloud, p_cloud, prm)
        dec = ctx.decode_delta(i_cloud, got["i_stream"], got["p_stream"], prm)
        want = D.decode_delta(i_cloud, got["i_stream"], got["p_stream"], RES, macroblock_size=16, do_icp_color_offset=bool(colour_offset))
        assert dec.tobytes() == want.tobytes()
        assert len(dec) > 0.8 * len(p_cloud) / 2


def test_delta_round_trip_quality(pkg, ctx, pair):
    """encode -> decode: the decoded P frame is geometrically as close to the input as an intra coded one."""
    i_cloud, p_cloud = pair
    prm = _params(pkg)
    got = ctx.encode_delta(i_cloud, p_cloud, prm)
    dec = ctx.decode_delta(i_cloud, got["i_stream"], got["p_stream"], prm)
    # the part of the body that moved rigidly is reproduced to within a voxel (the top third is sheared by several
    # voxels: its blocks still pass the reference's weak fitness test, fitness < 2 * point_resolution, and are predicted badly)
    q = ctx.quality_metrics(p_cloud[p_cloud["z"] < 0.55], dec[dec["z"] < 0.6], cell_hint=RES)
    assert q["left_rms"] < 1.0 * RES
    q = ctx.quality_metrics(p_cloud, dec, cell_hint=RES)
    assert q["symm_rms"] < 5.0 * RES
    # and the two streams together are smaller than intra coding the frame
    intra, _ = ctx.encode_intra_host(p_cloud, prm)
    assert len(got["i_stream"]) + len(got["p_stream"]) < len(intra)


def test_delta_class_interface(pkg, pair):
    i_cloud, p_cloud = pair
    B = pkg.binding
    codec = B.OctreePointCloudCodecV2(B.MANUAL_CONFIGURATION, False, RES, RES, True, 0, True, 8, 1, False, False, False, 85)
    out_cloud, i_data, p_data = codec.encodePointCloudDeltaFrame(i_cloud, p_cloud)
    assert 0.5 < codec.getMacroBlockPercentage() <= 1.0
    assert 0.0 < codec.getMacroBlockConvergencePercentage() <= 1.0
    dec = codec.decodePointCloudDeltaFrame(i_cloud, i_data, p_data)
    assert len(dec) > 0 and len(out_cloud) > 0


def test_delta_disjoint_frames_are_all_intra(pkg, ctx, pair):
    """No shared macroblock: nothing is predicted, the P stream is empty, the I stream is the simplified frame intra coded."""
    i_cloud, p_cloud = pair
    far = i_cloud.copy()
    far["x"] = (far["x"] * np.float32(0.1)).astype(np.float32)   # squeezed into x < 0.09: no block in common with the P frame
    prm = _params(pkg)
    got = ctx.encode_delta(far, p_cloud, prm)
    want = D.encode_delta(far, p_cloud, RES, RES, macroblock_size=16)
    assert got["p_stream"] == b"" == want["p_stream"]
    assert got["i_stream"] == want["i_stream"]
    assert got["shared_macroblock_count"] == 0
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()


def test_delta_rejects_empty(pkg, ctx, pair):
    i_cloud, p_cloud = pair
    with pytest.raises(pkg.binding.PccError):
        ctx.encode_delta(i_cloud[:0], p_cloud, _params(pkg))


def test_cfg5_moving_sphere_sequence(pkg, oracle, ctx):
    """SURVEY.md 8(d) cfg5, reduced: I(0), P(1|0), I(1), P(2|1) like the reference app with do_delta_coding=1 -- the I
    frame of each prediction is the encoder's simplified cloud (eval.hpp:862).  Bit-exact given the transforms;
    the macroblock statistics against the oracle's own ICP (statistical parity)."""
    cfg = dict(pkg.synthetic.CONFIGS["cfg5"])
    frames = pkg.synthetic.moving_sphere_group(20_000, cfg["seed"], 3)
    res = 2.0 ** -cfg["octree_bits"]
    prm = pkg.binding.make_params(octree_bits=cfg["octree_bits"], color_bits=8, color_coding_type=1, jpeg_quality=85)
    for f in range(2):
        prm.frame_id = f + 1
        _, _ = ctx.encode_intra_host(frames[f], prm)
        i_cloud = ctx.output_cloud()
        want_i = oracle.encode_intra(frames[f], oracle.make_params(octree_bits=cfg["octree_bits"], frame_id=f + 1)).simplified
        assert i_cloud.tobytes() == want_i.tobytes()
        got = ctx.encode_delta(i_cloud, frames[f + 1], prm)
        want = D.encode_delta(i_cloud, frames[f + 1], res, res, icp_fn=_replay(got))
        assert got["p_stream"] == want["p_stream"] and got["i_stream"] == want["i_stream"]
        assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()
        own = D.encode_delta(i_cloud, frames[f + 1], res, res, write_out_cloud=False)   # the oracle's ICP
        assert abs(float(own["shared_percentage"]) - got["shared_macroblock_percentage"]) < 1e-6
        assert abs(float(own["convergence_percentage"]) - got["shared_macroblock_convergence_percentage"]) < 0.02
        assert abs(len(own["p_stream"]) - len(got["p_stream"])) <= 0.02 * len(got["p_stream"]) + 40
        assert got["shared_macroblock_percentage"] > 0.9 and got["convergence_count"] > 300


def test_cfg5_at_its_stated_size(pkg, oracle, ctx):
    """SURVEY.md 8(d) cfg5 as BASELINE.json states it: 30 frames of 200 000 points, sphere shell moving +0.002 in x
    per frame, 8-bit octree, macroblock 16, run like the reference app's loop with do_delta_coding=1 (eval.hpp:853-890):
    every frame intra coded (frame ids 1..30), every frame but the first also predicted from the encoder's simplified
    cloud of the frame before (eval.hpp:862).
      * every intra bitstream and simplified cloud: bit-exact against the oracle;
      * every P frame: chunk stream, residual intra stream, predicted cloud and the decoder's output bit-exact against
        oracle/delta_oracle.py given the transforms the GPU's ICP produced;
      * ICP (PCL's, outside the reference tree: statistical parity only): shared-block share equal, convergence share
        and predicted-part size within 2 % of the oracle's own ICP, on every fifth P frame."""
    cfg = dict(pkg.synthetic.CONFIGS["cfg5"])
    frames = pkg.synthetic.moving_sphere_group(cfg["n"], cfg["seed"], cfg["frames"])
    assert len(frames) == 30 and all(len(f) == 200_000 for f in frames)
    res = 2.0 ** -cfg["octree_bits"]
    kw = dict(octree_bits=cfg["octree_bits"], color_bits=8, color_coding_type=1, jpeg_quality=85)
    prm = pkg.binding.make_params(**kw)
    prm.macroblock_size = cfg["macroblock_size"]
    prev = None
    p_bytes = i_bytes = intra_bytes = 0
    for f, cloud in enumerate(frames):
        if prev is not None:
            got = ctx.encode_delta(prev, cloud, prm)
            want = D.encode_delta(prev, cloud, res, res, icp_fn=_replay(got))
            assert got["p_stream"] == want["p_stream"] and got["i_stream"] == want["i_stream"], f
            assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes(), f
            dec = ctx.decode_delta(prev, got["i_stream"], got["p_stream"], prm)
            ref = D.decode_delta(prev, want["i_stream"], want["p_stream"], res)
            assert dec.tobytes() == ref.tobytes(), f
            assert got["shared_macroblock_percentage"] > 0.95 and got["shared_macroblock_convergence_percentage"] > 0.85
            p_bytes += len(got["p_stream"]); i_bytes += len(got["i_stream"])
            if f % 5 == 1:
                own = D.encode_delta(prev, cloud, res, res, write_out_cloud=False)   # the oracle's ICP
                assert abs(float(own["shared_percentage"]) - got["shared_macroblock_percentage"]) < 1e-6
                assert abs(float(own["convergence_percentage"]) - got["shared_macroblock_convergence_percentage"]) < 0.02
                assert abs(len(own["p_stream"]) - len(got["p_stream"])) <= 0.02 * len(got["p_stream"]) + 40
        prm.frame_id = f + 1
        stream, _ = ctx.encode_intra_host(cloud, prm)
        r = oracle.encode_intra(cloud, oracle.make_params(frame_id=f + 1, **kw))
        assert stream == r.bitstream, f
        prev = ctx.output_cloud()
        assert prev.tobytes() == r.simplified.tobytes(), f
        intra_bytes += len(stream)
    # predictive coding pays on this sequence: a P frame is a fraction of an I frame
    assert (p_bytes + i_bytes) / 29 < 0.4 * intra_bytes / 30


@pytest.mark.parametrize("mb,on_original,waves", [(8, False, "4"), (32, False, "1"), (64, True, "1"), (128, True, "4")])
def test_delta_other_macroblock_sizes_and_big_blocks(pkg, ctx, pair, mb, on_original, waves, request):
    """Macroblock sizes other than 16; with 64-voxel blocks on the unsimplified cloud a block holds thousands of points
    (more targets than the ICP kernel stages in LDS: the HBM/L2 path)."""
    ctx.set_option("icp_waves", int(waves))   # one ICP kernel shape for every macroblock (the context is shared: undone afterwards)
    request.addfinalizer(lambda: ctx.set_option("icp_waves", 0))
    i_cloud, p_cloud = pair
    prm = _params(pkg, macroblock_size=mb)
    got = ctx.encode_delta(i_cloud, p_cloud, prm, icp_on_original=on_original)
    want = D.encode_delta(i_cloud, p_cloud, RES, RES, macroblock_size=mb, icp_on_original=on_original, icp_fn=_replay(got))
    assert [tuple(b["key"][:3]) for b in got["blocks"]] == [b["key"] for b in want["blocks"]]
    assert [bool(b["do_icp"]) for b in got["blocks"]] == [b["icp"] for b in want["blocks"]]
    assert got["p_stream"] == want["p_stream"] and got["i_stream"] == want["i_stream"]
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()
    if mb >= 64:
        assert int(got["blocks"]["n_p"].max()) > (1024 if waves == "1" else 4096)
    # a few blocks against the oracle's own ICP
    simp = p_cloud if on_original else D.simplify(p_cloud, RES)
    _, i_lists, _, _ = D.tree(i_cloud, RES * mb)
    _, p_lists, _, _ = D.tree(simp, RES * mb)
    todo = [(b, pl) for b, pl in zip(got["blocks"], p_lists) if b["do_icp"]]
    todo.sort(key=lambda t: -int(t[0]["n_p"]))
    for b, pl in todo[:3]:
        src, tgt = D._xyz(i_cloud[i_lists[b["i_block"]]]), D._xyz(simp[pl])
        conv, final, fitness = D.icp(src, tgt)
        assert (conv and fitness < 2 * RES) == bool(b["converged"])
        diff = np.abs(D.transform_points(src, final) - D.transform_points(src, b["rt"].reshape(4, 4))).max()
        assert diff < 0.5 * RES


def test_delta_many_blocks_uses_both_icp_shapes(pkg, ctx):
    """More than 2048 macroblocks: the heavy blocks go to the workgroup-per-block ICP kernel, the light ones to the
    wave-per-block kernel, on two streams.  Everything around the ICP stays bit-exact given its transforms."""
    i_cloud, p_cloud = pkg.synthetic.delta_pair(260_000, 31, grid=512)
    res = 1.0 / 512
    prm = pkg.binding.make_params(octree_bits=9, color_bits=8, color_coding_type=1, jpeg_quality=85)
    got = ctx.encode_delta(i_cloud, p_cloud, prm)
    assert got["macro_block_count"] > 2048
    work = got["blocks"]["n_i"].astype(np.int64) * got["blocks"]["n_p"] * (got["blocks"]["do_icp"] != 0)
    assert (work >= 128 * 128).any() and ((work > 0) & (work < 128 * 128)).any()      # both kinds present
    want = D.encode_delta(i_cloud, p_cloud, res, res, icp_fn=_replay(got))
    assert got["p_stream"] == want["p_stream"] and got["i_stream"] == want["i_stream"]
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()
    icp = got["blocks"][got["blocks"]["do_icp"] != 0]
    assert (icp["iterations"] > 0).all()          # every gated block was visited by one of the two kernels
    dec = ctx.decode_delta(i_cloud, got["i_stream"], got["p_stream"], prm)
    assert dec.tobytes() == D.decode_delta(i_cloud, got["i_stream"], got["p_stream"], res).tobytes()


def test_delta_clouds_outside_the_unit_cube(pkg, ctx, pair):
    """Clouds that were not normalised: the defined box [0,1]^3 of the macroblock trees and of the simplification grows
    around them (each tree on its own, as in the reference); still bit-exact given the transforms."""
    i_cloud, p_cloud = pair
    i2, p2 = i_cloud.copy(), p_cloud.copy()
    for c in (i2, p2):
        c["x"] = c["x"] * np.float32(2.5) - np.float32(0.7)
        c["y"] = c["y"] * np.float32(2.5) + np.float32(0.4)
        c["z"] = c["z"] * np.float32(2.5) - np.float32(1.2)
    res = 2.5 / 256
    prm = pkg.binding.make_params(octree_resolution=res, point_resolution=res, color_bits=8, color_coding_type=1, jpeg_quality=85)
    got = ctx.encode_delta(i2, p2, prm)
    want = D.encode_delta(i2, p2, res, res, icp_fn=_replay(got))
    assert [tuple(b["key"][:3]) for b in got["blocks"]] == [b["key"] for b in want["blocks"]]
    assert got["p_stream"] == want["p_stream"] and got["i_stream"] == want["i_stream"]
    assert got["out_cloud"].tobytes() == want["out_cloud"].tobytes()
    # (each tree grows its own box, so the keys of the two frames rarely meet: the reference app only predicts between
    # frames whose bounding boxes were aligned, eval.hpp:854)
    assert got["shared_macroblock_count"] == sum(1 for b in want["blocks"] if b["shared"])
