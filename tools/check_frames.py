import sys; sys.path.insert(0, '/root/repo')
import numpy as np
import __graft_entry__ as G
from oracle import oracle as O
pkg = G.load_package(); b = pkg.binding
ctx = b.Context(0)
kw = dict(octree_bits=10, color_bits=8, color_coding_type=1, jpeg_quality=85)
for f in range(int(sys.argv[1]), int(sys.argv[2])):
    pts = pkg.synthetic.make_frame("cfg2", frame=f)
    dev = ctx.upload(pts)
    ctx.hotpath_launch(dev, len(pts), b.make_params(**kw))
    hot = ctx.hotpath_finish(copy=False)
    s, perf = ctx.entropy_encode(hot.raw, b.make_params(**kw))
    want = O.encode_intra(pts, O.make_params(**kw), keep=False)
    print("frame", f, "L", hot.n_leaves, "B", hot.n_branches, "D", hot.depth, "epochs", hot.n_epochs, "match", s == want.bitstream, flush=True)
    ctx.free(dev)
