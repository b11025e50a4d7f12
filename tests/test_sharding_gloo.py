"""N>1 path on CPU: world_size-2 gloo run of the frame-per-rank sharding.

The GPU stage cannot run here, so each rank feeds the oracle's hot-path products of its frames to
the product's host entropy stage (pcc_entropy_encode on a host-only context).  What is covered is
what the multi-GPU path adds: the frame -> rank map, frame ids by sequence index, ordered
concatenation, and that the result is byte-identical to the sequential single-process encode.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_FRAMES = 6
DROPPED = 2   # this frame has no finite point: the reference drops it and does not advance frame_ID_ (impl.hpp:206-212)
KW = dict(octree_bits=6, color_bits=8, color_coding_type=1, jpeg_quality=80)


def _encode_frame(pkg, O, f, fid):
    from test_host_stage import _hot_from_oracle
    pts = pkg.synthetic.sphere_shell(3000, 0xC3 + f)
    if f == DROPPED:
        pts["x"] = np.nan
    r = O.encode_intra(pts, O.make_params(frame_id=fid, **KW))
    if r is None:
        return b""
    host = pkg.binding.Context(None)
    hr, keep = _hot_from_oracle(pkg, r)
    stream, _ = host.entropy_encode(hr, pkg.binding.make_params(frame_id=fid, **KW))
    assert stream == r.bitstream
    return stream


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import __graft_entry__ as G
    from oracle import oracle as O
    pkg = G.load_package()
    local = pkg.sharding.encode_shard(lambda f, fid: _encode_frame(pkg, O, f, fid), N_FRAMES, rank, world)
    assert sorted(local) == pkg.sharding.frames_for_rank(N_FRAMES, rank, world)
    whole = pkg.sharding.gather_streams(local, N_FRAMES, dist)
    dist.barrier()
    if rank == 0:
        with open(out_path, "wb") as fh:
            fh.write(whole)
    dist.destroy_process_group()


def test_frames_for_rank(pkg):
    s = pkg.sharding
    assert s.frames_for_rank(8, 0, 8) == [0] and s.frames_for_rank(8, 7, 8) == [7]
    assert s.frames_for_rank(10, 1, 4) == [1, 5, 9] and s.frames_for_rank(3, 3, 4) == []
    for world in (1, 2, 3, 8):
        got = sorted(f for r in range(world) for f in s.frames_for_rank(11, r, world))
        assert got == list(range(11))
    assert [s.frame_id(f) for f in range(3)] == [1, 2, 3]
    with pytest.raises(RuntimeError):
        s.gather_streams({0: b"a"}, 2)


def test_two_ranks_equal_sequential(pkg, oracle, tmp_path):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    out = str(tmp_path / "gop.bin")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    sharded = open(out, "rb").read()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    # the reference's serial loop: the counter only moves for frames that are not dropped
    parts, fid = [], 1
    for f in range(N_FRAMES):
        st = _encode_frame(pkg, oracle, f, fid)
        parts.append(st)
        fid += 1 if st else 0
    assert parts[DROPPED] == b"" and fid == N_FRAMES
    assert sharded == b"".join(parts)
    # the concatenated GOP decodes frame by frame, ids 1..N-1 in order
    host = pkg.binding.Context(None)
    pos = 0
    for f in range(N_FRAMES - 1):
        pts, info = host.decode_intra(sharded[pos:])
        assert info["params"]["frame_id"] == f + 1 and len(pts) > 0
        pos += info["consumed"]
    assert pos == len(sharded)


def test_multi_gpu_entry_point_refuses_missing_devices(pkg):
    """pcc_pipeline_create_multi: every named device has to exist -- nothing is silently left out, and there is no CPU
    fallback (on a box without GPUs the call fails for any list)."""
    import ctypes as C
    lib = pkg.binding.load_library()
    bad = (C.c_int32 * 2)(0, 4096)
    assert not lib.pcc_pipeline_create_multi(bad, 2, 2)
    assert not lib.pcc_pipeline_create_multi(bad, 0, 2)
    assert b"device" in lib.pcc_multi_pipeline_last_error(None)
    with pytest.raises(RuntimeError):
        pkg.binding.MultiPipeline([0, 4096], 2)
