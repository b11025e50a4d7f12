"""Frame-per-GPU sharding of a GOP.

Every frame is an independent I-frame: the reference resets tree and bounding box per frame
(impl.hpp:89-90) and forces intra coding (iFrameRate 0, eval.hpp:386).  The only thing that ties
frames together is the header counter frame_ID_ (impl.hpp:133), so frame f of a group goes to GPU
f mod N with frame_id = f + 1 assigned by sequence index, and the bitstreams are concatenated in
frame order on the host.  No collective is on the data path; torch.distributed is used only to
launch one process per GPU and, in `gather_streams`, to hand the finished bitstreams to rank 0.
"""


def frames_for_rank(n_frames, rank, world):
    """Indices of the frames rank `rank` of `world` encodes (round robin: frame f -> GPU f mod N)."""
    return list(range(rank, n_frames, world))


def frame_id(frame_index):
    """Provisional header frame_ID_ of the frame_index-th frame of a group (the encoder is re-created per group,
    eval.hpp:779, and pre-increments at impl.hpp:133): 1-based sequence index.  A frame that is dropped (empty or
    all non-finite, impl.hpp:206-212) does not advance the reference's counter, and a rank cannot know what the
    other ranks drop: `gather_streams` renumbers the finished bitstreams in sequence order."""
    return frame_index + 1


FRAME_ID_OFFSET = 48  # u32 behind the two identifiers of the frame header (28 + 20 bytes, impl.hpp:1472-1486)


def renumber(streams, first_id=1):
    """Frame ids of the reference's serial loop: consecutive over the frames that produced a bitstream."""
    out, fid = [], first_id
    for s in streams:
        if len(s) >= FRAME_ID_OFFSET + 4:
            s = s[:FRAME_ID_OFFSET] + int(fid).to_bytes(4, "little") + s[FRAME_ID_OFFSET + 4:]
            fid += 1
        out.append(s)
    return out


def encode_shard(encode_one, n_frames, rank, world):
    """Run `encode_one(frame_index, frame_id) -> bytes` for this rank's frames; {frame_index: bytes}."""
    return {f: encode_one(f, frame_id(f)) for f in frames_for_rank(n_frames, rank, world)}


def gather_streams(local, n_frames, dist=None):
    """Concatenate per-frame bitstreams in frame order on rank 0 (control plane only)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        parts = [local]
    else:
        parts = [None] * dist.get_world_size()
        dist.all_gather_object(parts, local)
    merged = {}
    for p in parts:
        merged.update(p)
    missing = [f for f in range(n_frames) if f not in merged]
    if missing:
        raise RuntimeError("frames not encoded by any rank: %r" % missing)
    return b"".join(renumber([merged[f] for f in range(n_frames)]))
