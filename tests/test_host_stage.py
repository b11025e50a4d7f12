"""Host stages of the product library (serial by design) against the oracle, and the C ABI surface.

No GPU work is launched here: the hot-path products fed to pcc_entropy_encode come from the oracle,
so what is checked is exactly the host code: header, range coder, JPEG, stream assembly, decoder.
"""
import ctypes as C
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
DEV_LIB = os.path.join(ROOT, "cwi-pcl-codec_amd", "libpcc_hip_dev.so")   # `make dev`, built by __graft_entry__.build()


def test_library_exports_every_declared_symbol(pkg):
    """Both headers of the C ABI -- pcc_codec.h, the drop-in boundary, and pcc_codec_tools.h, what measurements, tests and
    tools use on top of it -- declare exactly the binding's two lists, the library exports every one of them, and nothing
    named pcc_* besides."""
    import subprocess
    lib = pkg.binding.load_library()
    for header, names in (("pcc_codec.h", pkg.binding.BOUNDARY_EXPORTS), ("pcc_codec_tools.h", pkg.binding.TOOLS_EXPORTS)):
        text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", header)).read(), flags=re.S)
        declared = set(re.findall(r"\b(pcc_[a-z_0-9]+)\s*\(", text))
        assert declared == set(names), (header, declared ^ set(names))
        for name in declared:
            assert hasattr(lib, name), name
    assert len(pkg.binding.BOUNDARY_EXPORTS) <= 40
    import shutil
    nm = subprocess.run(["nm", "-D", "--defined-only", pkg.binding.LIB_PATH], capture_output=True, text=True) if shutil.which("nm") else None
    if nm is not None and nm.returncode == 0:
        exported = {l.split()[-1] for l in nm.stdout.splitlines() if " T pcc_" in l}
        assert exported == set(pkg.binding.EXPORTS), exported ^ set(pkg.binding.EXPORTS)
    assert b"gfx950" in lib.pcc_version()


def test_the_shipped_library_reads_only_the_pipelines_deployment_variables():
    """Developer switches go through dev_env() (csrc/pcc_dev.h: a constant nullptr unless -DPCC_DEV): what is left of getenv in
    the sources is the pipeline's deployment configuration, the nine names include/pcc_codec.h lists under "Environment"."""
    import glob
    names = set()
    for f in glob.glob(os.path.join(ROOT, "cwi-pcl-codec_amd", "csrc", "*.*")):
        if f.endswith((".cpp", ".hip", ".h")) and not f.endswith("pcc_dev.h"):
            names |= set(re.findall(r'(?<![_a-z:])getenv\("([A-Z_0-9]+)"\)', open(f).read()))
    assert names == {"PCC_PIPELINE_ENTROPY", "PCC_PIPELINE_GPU_THREADS", "PCC_PIPELINE_UPLOAD_THREADS", "PCC_PIPELINE_BATCH", "PCC_PIPELINE_PIN",
                     "PCC_PIPELINE_PIN_OFFSET", "PCC_PIPELINE_PIN_SPAN", "LOCAL_RANK", "LOCAL_WORLD_SIZE"}, names
    header = open(os.path.join(ROOT, "include", "pcc_codec.h")).read()
    for n in names:
        assert n in header, n


@pytest.mark.parametrize("tag", ["r02", "r04x"])
def test_the_libraries_of_older_commits_load_beside_head(pkg, tag):
    """tools/known_good/build.sh: r02 = the library of the last commit whose parity suite ran green on an MI355X (round 2),
    r04x = round 4's HEAD with the opt-in forms that left the product in round 5 (branch experiments/r04-optin-forms), each
    built from its commit's sources with HEAD's whole C ABI (later entry points come from a compat file) -- so that one GPU
    session can time and digest-check them beside HEAD.  Here: they load, export every declared symbol, and their host-side
    range coder and JPEG writer give HEAD's bytes.  bench.py / smoke() refuse them (not HEAD's product library)."""
    import subprocess
    path = os.path.join(ROOT, "cwi-pcl-codec_amd", "libpcc_hip_%s.so" % tag)
    if not os.path.exists(path):
        if not os.path.isdir(os.path.join(ROOT, ".git")):
            pytest.skip("no git history here: the older library cannot be rebuilt")
        assert subprocess.run(["bash", os.path.join(ROOT, "tools", "known_good", "build.sh")]).returncode == 0
    old, new = C.CDLL(path), pkg.binding.load_library()
    for name in pkg.binding.EXPORTS:
        assert hasattr(old, name), name
    rng = np.random.default_rng(2)
    data = np.minimum(rng.geometric(0.05, 50_000), 255).astype(np.uint8)
    outs = []
    for lib in (old, new):
        lib.pcc_host_range_encode.restype = C.c_size_t
        lib.pcc_host_range_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        buf = np.zeros(len(data) * 2 + 2048, dtype=np.uint8)
        n = lib.pcc_host_range_encode(data.ctypes.data, len(data), buf.ctypes.data, len(buf))
        outs.append(buf[:n].tobytes())
    assert outs[0] == outs[1] and len(outs[0]) > 1000
    e = dict(os.environ, PCC_LIB=path)
    e.pop("PCC_ALLOW_NON_PRODUCT_LIB", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)
    assert r.returncode != 0 and "not the gfx950 product library" in r.stderr


def test_struct_layouts_match_header(pkg, oracle):
    b = pkg.binding
    assert C.sizeof(b.Params) == 56 and b.POINT_DTYPE.itemsize == 32
    assert C.sizeof(b.HotResult) == 8 * 6 + 4 + 4 + 8 * 3 + 8 * 4 + 4 + 4 + 4 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 8 + 4 + 4   # ... + jpeg_lines_dir, jpeg_lines_data, jpeg_n_lines (+ padding)
    assert C.sizeof(b.Bitstream) == 8 + 8 + 24


def test_no_gpu_means_loud_failure_not_fallback(pkg):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.binding.Context(0)
    host = pkg.binding.Context(None)  # host-only context: GPU entry points must refuse
    pts = pkg.synthetic.sphere_shell(100, 1)
    with pytest.raises(pkg.binding.PccError) as e:
        host.encode_intra_host(pts, pkg.binding.make_params())
    assert e.value.code == -6


@pytest.mark.parametrize("n,kind", [(0, "u"), (1, "u"), (256, "all"), (70_000, "skew"), (20_000, "u")])
def test_range_coder_matches_oracle(pkg, oracle, n, kind):
    rng = np.random.default_rng(n + 5)
    if kind == "all":
        data = bytes(range(256))
    elif kind == "skew":
        data = bytes(np.minimum(rng.geometric(0.25, n), 255).astype(np.uint8))
    else:
        data = bytes(rng.integers(0, 256, n, dtype=np.uint8))
    enc = pkg.binding.host_range_encode(data)
    assert enc == oracle.rc_encode(data)
    dec, used = pkg.binding.host_range_decode(enc, len(data))
    assert dec == data and used == len(enc)


@pytest.mark.parametrize("lengths", [(5_000,), (70_000, 3), (0, 9_000), (12_345, 12_345, 700), (66_000, 1, 0, 40_000), (300, 200, 100, 70_001)])
def test_range_coders_sharing_a_loop_match_oracle(pkg, oracle, lengths):
    """The entropy stage codes up to four vectors in one loop (three and four: one multiply yields both the new low and
    the new range; one and two: the short dependency chain); ragged lengths walk through every hand-over between the
    loops.  Each vector must come out as the oracle codes it alone -- including a table that needs the 2^16 rescale and
    vectors long enough for the rare range-underflow branch."""
    rng = np.random.default_rng(sum(lengths) + len(lengths))
    vectors = []
    for k, n in enumerate(lengths):
        if k % 2 == 0:  # occupancy-like: one to three bits set
            bits = rng.integers(0, 8, (n, 3))
            keep = rng.random((n, 3)) < np.array([1.0, 0.45, 0.15])
            v = np.bitwise_or.reduce(np.where(keep, 1 << bits, 0), axis=1).astype(np.uint8)
        else:
            v = np.minimum(rng.geometric(0.08, n), 255).astype(np.uint8)
        vectors.append(v.tobytes())
    got = pkg.binding.host_range_encode_many(vectors)
    for v, g in zip(vectors, got):
        assert g == oracle.rc_encode(v)
        assert g == pkg.binding.host_range_encode(v)


def test_range_encode_many_refuses_bad_counts(pkg):
    with pytest.raises(pkg.binding.PccError):
        pkg.binding.host_range_encode_many([b"ab"] * 17)


@pytest.mark.parametrize("wide", ["1", "0"])
def test_five_to_sixteen_range_coders_in_one_call_match_the_oracle(pkg, oracle, wide):
    """Ten and more vectors go through the lanes of AVX-512 registers where the CPU has them (one stream per 32-bit lane, the
    same arithmetic per lane; PCC_RC_WIDE=0 or a CPU without AVX-512: scalar loops of four, one group after the other), five
    to nine through scalar loops of four.  Every count from 5 to 16, ragged lengths (the vector loop hands over to narrower
    loops as streams end, and takes its symbols four at a time), tables that need the 2^16 rescale, streams skewed enough for
    the range-underflow branch, empty and one-symbol vectors: each vector as the oracle codes it alone.  A child process,
    because the switch is read once."""
    import subprocess, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import numpy as np
        import __graft_entry__ as G
        from oracle import oracle as O
        b = G.load_package().binding
        rng = np.random.default_rng(2026)
        def vector(kind, n):
            if kind == 0:      # occupancy-like
                bits = rng.integers(0, 8, (n, 3)); keep = rng.random((n, 3)) < np.array([1.0, 0.45, 0.15])
                return np.bitwise_or.reduce(np.where(keep, 1 << bits, 0), axis=1).astype(np.uint8).tobytes()
            if kind == 1:      # very skewed: the underflow branch
                return np.where(rng.integers(0, 1000, n) < 995, 255, rng.integers(0, 256, n)).astype(np.uint8).tobytes()
            if kind == 2:
                return np.minimum(rng.geometric(0.08, n), 255).astype(np.uint8).tobytes()
            return rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        checked = 0
        for count in range(5, 17):
            for trial in range(2):
                if trial == 0:   # about equal lengths, like the frames of a sequence
                    lens = [60_000 + int(rng.integers(0, 4_000)) for _ in range(count)]
                else:            # ragged, with degenerate ones
                    lens = [int(x) for x in rng.choice([0, 1, 3, 5, 257, 1_023, 4_099, 70_001, 20_000, 33_333], count)]
                vs = [vector((k + trial) %% 4, n) for k, n in enumerate(lens)]
                got = b.host_range_encode_many(vs)
                for v, g in zip(vs, got):
                    assert g == O.rc_encode(v), (count, trial, len(v))
                    checked += 1
        print("OK", checked)
    """ % ROOT)
    if wide == "1" and not pkg.binding.load_library().pcc_debug_host_rc_wide():
        pytest.skip("no AVX-512 on this host: the vector path of the host range coder cannot be compared here")
    env = dict(os.environ, PCC_RC_WIDE=wide, PCC_LIB=DEV_LIB)   # (the switch exists in the developer build only: csrc/pcc_dev.h)
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_jpeg_matches_libjpeg_turbo_golden(pkg):
    z = np.load(os.path.join(GOLDEN, "jpeg_golden.npz"))
    n = len([k for k in z.files if k.startswith("in_")])
    for i in range(n):
        img, q = z["in_%02d" % i], int(z["q_%02d" % i])
        assert pkg.binding.host_jpeg_encode(img, q) == z["jpg_%02d" % i].tobytes(), (img.shape, q)
        assert np.array_equal(pkg.binding.host_jpeg_decode(z["jpg_%02d" % i].tobytes(), 1 << 16), z["dec_%02d" % i])


def test_jpeg_matches_oracle_on_random_images(pkg, oracle):
    rng = np.random.default_rng(99)
    for (h, w) in [(1, 256), (7, 256), (33, 256), (1, 2048), (1, 3000), (2, 9), (19, 24)]:
        for q in (5, 60, 85, 97):
            img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
            jpg = oracle.jpeg_encode(img, q)
            assert pkg.binding.host_jpeg_encode(img, q) == jpg
            assert np.array_equal(pkg.binding.host_jpeg_decode(jpg, 1 << 16), oracle.jpeg_decode(jpg))


@pytest.mark.parametrize("w,h", [(256, 1), (256, 5), (256, 8), (256, 9), (256, 17), (16, 3), (8, 7), (64, 23), (256, 100)])
def test_snake_closed_form_matches_iterator(pkg, oracle, w, h):
    assert np.array_equal(pkg.binding.host_snake_perm(w, h), oracle.snake_perm(w, h))


def _hot_from_oracle(pkg, r):
    """Build a pcc_hot_result out of the oracle's intermediate products."""
    b = pkg.binding
    hr = b.HotResult()
    for i in range(6):
        hr.bbox[i] = r.bbox[i]
    hr.depth, hr.n_points_in, hr.n_leaves, hr.n_branches = r.depth, r.n_points_in, r.n_leaves, r.n_branches
    keep = [np.ascontiguousarray(a) for a in (r.occupancy, r.bgr, r.centroid_bytes, r.snake_image)]
    hr.occupancy = keep[0].ctypes.data
    hr.bgr = keep[1].ctypes.data if keep[1].size else None
    hr.centroid = keep[2].ctypes.data if keep[2].size else None
    hr.image = keep[3].ctypes.data if keep[3].size else None
    hr.image_w, hr.image_h = r.image_w, r.image_h
    return hr, keep


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
@pytest.mark.parametrize("centroid", [0, 1])
def test_entropy_stage_and_decoder_match_oracle(pkg, oracle, mode, centroid):
    pts = pkg.synthetic.sphere_shell(6000, 0x77 + mode)
    kw = dict(octree_bits=6, color_bits=7 if mode == 0 else 8, color_coding_type=mode, keep_centroid=centroid,
              jpeg_quality=80, frame_id=3)
    r = oracle.encode_intra(pts, oracle.make_params(**kw))
    host = pkg.binding.Context(None)
    hr, keep = _hot_from_oracle(pkg, r)
    stream, perf = host.entropy_encode(hr, pkg.binding.make_params(**kw))
    assert stream == r.bitstream
    assert perf == r.perf
    dec, info = host.decode_intra(stream)
    want = oracle.decode_intra(stream)
    assert info["consumed"] == want.consumed == len(stream) and info["depth"] == want.depth
    assert np.array_equal(info["bbox"], want.bbox)
    assert dec.tobytes() == want.points.tobytes()
    assert info["params"]["frame_id"] == 3 and info["params"]["color_coding_type"] == mode


@pytest.mark.parametrize("octree_bits,mode,centroid", [(23, 1, 0), (22, 0, 1), (19, 3, 1), (9, 1, 0), (9, 3, 1)])
def test_host_decoder_on_deep_and_larger_trees(pkg, oracle, octree_bits, mode, centroid):
    """The host decoder walks branch nodes only and takes keys out of the 3-bits-per-level path while the path fits 63
    bits (depth <= 21), and falls back to the pre-order walk with a stack beyond that; without a centroid stream the walk
    runs on a second thread.  Both walks, both thread arrangements, against the oracle's decoder."""
    n = 3000 if octree_bits > 21 else 60_000
    pts = pkg.synthetic.sphere_shell(n, 0x99 + octree_bits)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=octree_bits, color_coding_type=mode, keep_centroid=centroid, jpeg_quality=70))
    assert (r.depth > 21) == (octree_bits > 21) and r.depth <= 32
    want = oracle.decode_intra(r.bitstream)
    dec, info = pkg.binding.Context(None).decode_intra(r.bitstream)
    assert info["depth"] == want.depth and info["consumed"] == want.consumed == len(r.bitstream)
    assert dec.tobytes() == want.points.tobytes()


def test_host_decoder_on_one_thread_gives_the_same_cloud(pkg, oracle, tmp_path):
    """PCC_DECODE_SERIAL=1 (read once per process, hence the child process) keeps the walk on the calling thread."""
    import subprocess, sys
    pts = pkg.synthetic.sphere_shell(20_000, 0x5e)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=8, color_coding_type=1, jpeg_quality=85))
    (tmp_path / "frame.bin").write_bytes(r.bitstream)
    want = oracle.decode_intra(r.bitstream).points.tobytes()
    code = ("import sys; sys.path.insert(0, %r); import __graft_entry__ as G; b = G.load_package().binding; "
            "pts, info = b.Context(None).decode_intra(open(%r, 'rb').read()); sys.stdout.buffer.write(pts.tobytes())"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "frame.bin")))
    for serial in ("", "1"):
        env = dict(os.environ, PCC_LIB=DEV_LIB)   # (the switch exists in the developer build only: csrc/pcc_dev.h)
        env.pop("PCC_DECODE_SERIAL", None)
        if serial:
            env["PCC_DECODE_SERIAL"] = serial
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr.decode()[-2000:]
        assert out.stdout == want


@pytest.mark.parametrize("modes", [(1, 1), (0, 2), (3, 1)])
def test_two_frames_at_once_give_the_same_bytes(pkg, oracle, modes):
    """pcc_entropy_encode2 interleaves the range-coder loops of two frames: the bytes must not change
    (different sizes, colour modes and centroid settings in one pair)."""
    hosts = [pkg.binding.Context(None), pkg.binding.Context(None)]
    hrs, prms, wants, keeps = [], [], [], []
    for i, mode in enumerate(modes):
        pts = pkg.synthetic.sphere_shell(4000 + 3000 * i, 0x51 + i)
        kw = dict(octree_bits=6 + i, color_bits=7 if mode == 0 else 8, color_coding_type=mode, keep_centroid=i,
                  jpeg_quality=75, frame_id=5 + i)
        r = oracle.encode_intra(pts, oracle.make_params(**kw))
        hr, keep = _hot_from_oracle(pkg, r)
        hrs.append(hr); prms.append(pkg.binding.make_params(**kw)); wants.append(r); keeps.append(keep)
    (sa, pa), (sb, pb) = hosts[0].entropy_encode2(hrs[0], prms[0], hosts[1], hrs[1], prms[1])
    assert sa == wants[0].bitstream and pa == wants[0].perf
    assert sb == wants[1].bitstream and pb == wants[1].perf


def test_four_frames_at_once_give_the_same_bytes(pkg, oracle):
    """pcc_entropy_encode_many with 3 and 4 frames of different sizes and settings."""
    b = pkg.binding
    hosts = [b.Context(None) for _ in range(4)]
    hrs, prms, wants, keeps = [], [], [], []
    for i, (mode, cen) in enumerate([(1, 0), (0, 1), (1, 1), (2, 0)]):
        pts = pkg.synthetic.sphere_shell(2500 + 2100 * i, 0x61 + i)
        kw = dict(octree_bits=5 + (i % 3), color_bits=6 if mode == 0 else 8, color_coding_type=mode, keep_centroid=cen,
                  jpeg_quality=60 + 10 * i, frame_id=9 + i)
        r = oracle.encode_intra(pts, oracle.make_params(**kw))
        hr, keep = _hot_from_oracle(pkg, r)
        hrs.append(hr); prms.append(b.make_params(**kw)); wants.append(r); keeps.append(keep)
    for n in (3, 4):
        got = b.Context.entropy_encode_many(hosts[:n], hrs[:n], prms[:n])
        for i in range(n):
            assert got[i][0] == wants[i].bitstream and got[i][1] == wants[i].perf, (n, i)


def test_sixteen_frames_at_once_give_the_same_bytes(pkg, oracle):
    """pcc_entropy_encode_many with 9, 12 and 16 frames (PCC_MAX_FRAMES_AT_ONCE): the occupancy streams of ten and more
    frames go through the vector coder where the CPU has AVX-512, the colour and centroid streams of the frames that have them
    through whatever their count selects -- the oracle's bitstreams and performance counters."""
    b = pkg.binding
    hosts = [b.Context(None) for _ in range(16)]
    hrs, prms, wants, keeps = [], [], [], []
    for i in range(16):
        mode, cen = [(1, 0), (0, 1), (1, 1), (2, 0), (3, 1)][i % 5]
        pts = pkg.synthetic.sphere_shell(1500 + 900 * i, 0x91 + i)
        kw = dict(octree_bits=5 + (i % 3), color_bits=0 if i == 7 else (6 if mode == 0 else 8), color_coding_type=mode, keep_centroid=cen,
                  jpeg_quality=50 + 3 * i, frame_id=20 + i)
        r = oracle.encode_intra(pts, oracle.make_params(**kw))
        hr, keep = _hot_from_oracle(pkg, r)
        hrs.append(hr); prms.append(b.make_params(**kw)); wants.append(r); keeps.append(keep)
    for n in (9, 12, 16):
        got = b.Context.entropy_encode_many(hosts[:n], hrs[:n], prms[:n])
        for i in range(n):
            assert got[i][0] == wants[i].bitstream and got[i][1] == wants[i].perf, (n, i)


def test_decoder_rejects_garbage(pkg):
    host = pkg.binding.Context(None)
    for bad in (b"", b"hello", b"<PCL-OCT-CODECV2-COMPRESSED><PCL-OCT-COMPRESSED>\x00"):
        with pytest.raises(pkg.binding.PccError) as e:
            host.decode_intra(bad)
        assert e.value.code == -5


def test_decoder_syncs_past_leading_junk(pkg, oracle):
    pts = pkg.synthetic.sphere_shell(500, 5)
    r = oracle.encode_intra(pts, oracle.make_params(octree_bits=5, color_coding_type=0))
    host = pkg.binding.Context(None)
    dec, info = host.decode_intra(b"<<PCL-junk" + r.bitstream)
    assert len(dec) == r.n_leaves and info["consumed"] == len(r.bitstream) + 10


def test_normalize_group_matches_numpy(pkg):
    b = pkg.binding
    raw = pkg.synthetic.sphere_shell(3000, 21, centre=(3.0, -1.0, 0.5), radius=1.7, do_normalize=False)
    a = raw.copy()
    ref = raw.copy()
    mn2, mx2 = pkg.synthetic.normalize(ref, 0.2)
    lib = b.load_library()
    ptrs = (C.c_void_p * 1)(a.ctypes.data)
    sizes = (C.c_size_t * 1)(len(a))
    mn = np.zeros(3, np.float32)
    mx = np.zeros(3, np.float32)
    assert lib.pcc_normalize_group_boxes(ptrs, sizes, 1, 0.2, mn.ctypes.data, mx.ctypes.data, None) == 0
    assert np.array_equal(mn, mn2) and np.array_equal(mx, mx2) and a.tobytes() == ref.tobytes()
    assert lib.pcc_restore_scaling(a.ctypes.data, len(a), mn.ctypes.data, mx.ctypes.data) == 0
    for ax in "xyz":
        assert np.abs(a[ax] - raw[ax]).max() < 1e-5


def test_decoder_accepts_streams_that_compress_a_thousandfold(pkg, oracle):
    """A static order-0 range coder spends as little as 0.0056 bits on a symbol (table total below 2^16, absent symbols count 1):
    fewer than 1424 symbols per byte.  Frames of coincident points reach hundreds -- 64 000 voxels of a lattice whose points sit
    on the voxel corners: every centroid byte equal, 192 000 symbols in 1.2 KB -- and a line along an axis does it for the
    occupancy bytes.  The decoder's guard against hostile counts (refuse before allocating) must not refuse them: round 3's
    allowed 64 symbols per byte and did (found by tools/fuzz_executor.py, seed 900562 of the large frames).  Counts beyond
    what the coded bytes can hold are still refused."""
    b = pkg.binding
    host = b.Context(None)
    import test_gpu_parity as T   # (the generator of its random sweep: this is the frame the campaign tripped over)
    lattice, lattice_kw = T._random_case(pkg, 900562, (100_000, 196_608, 196_609, 300_000, 393_217, 450_000, 800_000))   # tools/fuzz_executor.py --big
    assert len(lattice) == 450_000 and lattice_kw["keep_centroid"] == 1 and lattice_kw["color_bits"] == 0
    t = np.linspace(0.0, 1.0, 50_000)
    line = np.zeros(len(t), dtype=b.POINT_DTYPE)
    line["x"], line["y"], line["z"] = t, 0.25, 0.75
    for pts, kw in ((lattice, lattice_kw), (line, dict(octree_bits=14, color_bits=0, keep_centroid=1, frame_id=3))):
        r = oracle.encode_intra(pts, oracle.make_params(**kw))
        ref = oracle.decode_intra(r.bitstream).points
        if pts is lattice:   # three centroid bytes per voxel: far more symbols than round 3's guard (64 per byte of the stream) let through
            assert 3 * len(ref) > 64 * len(r.bitstream) + 64
        got, info = host.decode_intra(r.bitstream)
        assert info["consumed"] == len(r.bitstream) and got.tobytes() == ref.tobytes()
    # a header that asks for more symbols than its bytes can hold: refused (no allocation of 2^40 bytes)
    stream = bytearray(r.bitstream)
    at = stream.index(b"<PCL-OCT-COMPRESSED>") + len(b"<PCL-OCT-COMPRESSED>")
    import struct
    hdr = at + 4 + 1 + 1 + 1 + 8 + 8 + 1 + 8 + 48 + 1 + 1 + 1 + 4 + 4 + 1   # frame header of writeFrameHeader (impl.hpp:1472-1486), then u64 occupancy count
    n_occ, = struct.unpack_from("<Q", stream, hdr)
    assert 0 < n_occ < 10 ** 7
    struct.pack_into("<Q", stream, hdr, 1 << 40)
    with pytest.raises(b.PccError):
        host.decode_intra(bytes(stream))
    # ... and the bound is the stream's OWN table's (the widest symbol over the total gives the bits a symbol costs at least),
    # not a constant: this frame's occupancy table is nowhere near one symbol taking everything, so a count a few times the
    # true one is already more than its coded bytes can hold -- refused, although 1424 symbols per byte would allow it
    freq = struct.unpack_from("<257I", stream, hdr + 8)
    widest, total = max(freq[k + 1] - freq[k] for k in range(256)), freq[256]
    bits = -np.log2(widest / total)
    n_points, = struct.unpack_from("<Q", stream, at + 7)
    depth = info["depth"]
    bound = (8 * (r.perf[0] - 1028) + 64) / bits + 1          # symbols the coded occupancy bytes can hold by their own table
    claim = int(bound) + 100
    # more than the stream really holds, fewer than "one branch node per level and voxel", far fewer than 1424 per byte would allow
    assert n_occ <= bound and n_occ < claim <= n_points * depth and claim < 1424 * (r.perf[0] - 1028)
    struct.pack_into("<Q", stream, hdr, claim)
    with pytest.raises(b.PccError):
        host.decode_intra(bytes(stream))
    # a table in which one symbol takes everything costs nothing per symbol and could claim any count: not the encoder's, refused
    struct.pack_into("<Q", stream, hdr, n_occ)
    struct.pack_into("<257I", stream, hdr + 8, *([0] * 200 + [1000] * 57))
    with pytest.raises(b.PccError):
        host.decode_intra(bytes(stream))
    # the centroid stream holds three bytes per voxel, exactly
    stream = bytearray(r.bitstream)
    occ_coded = r.perf[0]
    cen_at = hdr + 8 + occ_coded
    n_cen, = struct.unpack_from("<I", stream, cen_at)
    assert n_cen == 3 * n_points
    struct.pack_into("<I", stream, cen_at, n_cen + 3)
    with pytest.raises(b.PccError):
        host.decode_intra(bytes(stream))
    # a voxel count that the occupancy bytes cannot open (eight voxels per branch byte at most) is refused before anything is
    # sized by it (centroids off in this copy of the header, so that the three-bytes-per-voxel check is not what catches it)
    stream = bytearray(r.bitstream)
    assert struct.unpack_from("<Q", stream, at + 7)[0] == n_points and stream[at + 4 + 1 + 1 + 1 + 8 + 8 + 1 + 8 + 48] == 1
    stream[at + 4 + 1 + 1 + 1 + 8 + 8 + 1 + 8 + 48] = 0          # do_voxel_centroid
    struct.pack_into("<Q", stream, at + 7, 8 * n_occ + 1)
    with pytest.raises(b.PccError, match="PCC_ERR_STREAM"):
        host.decode_intra(bytes(stream))
    # do_voxel_grid false (the reference's point-detail tail, impl.hpp:1728-1757: the count is points, not voxels): said, not misread
    stream = bytearray(r.bitstream)
    assert stream[at + 5] == 1
    stream[at + 5] = 0
    with pytest.raises(b.PccError, match="PCC_ERR_UNSUPPORTED"):
        host.decode_intra(bytes(stream))


def test_the_binding_warns_when_a_developer_switch_is_set_for_a_library_that_ignores_it():
    """csrc/pcc_dev.h: the shipped library reads no developer switch.  A tool that sets one without pointing PCC_LIB at the developer
    build would measure the default and report it under the switch's name: load_library() says so."""
    import subprocess
    code = ("import __graft_entry__ as G, warnings\nwarnings.simplefilter('error')\n"
            "G.load_package().binding.load_library()\nprint('loaded')")
    for extra, loads in ((dict(PCC_DECODE_TRACE="1"), False), (dict(PCC_DECODE_TRACE="1", PCC_LIB=DEV_LIB), True), ({}, True)):
        env = {k: v for k, v in os.environ.items() if not k.startswith("PCC_")}
        env.update(extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, cwd=ROOT, timeout=300)
        assert ("loaded" in r.stdout) == loads, (extra, r.stdout, r.stderr[-600:])
        if not loads:
            assert "does not read developer switches" in r.stderr and "libpcc_hip_dev.so" in r.stderr


def test_host_decoder_accepts_the_2048_wide_strip_a_reference_encoder_writes(pkg, oracle):
    """Colour coding type 2 with fewer than 2048 voxels: the reference's encoder writes ONE strip 2048 pixels wide
    (jpegcc.h:256-275, the width is never narrowed in the `num_lines == 0` branch; the pixels beyond the voxels are an
    over-read).  decodeJPEGLines (jpegcc.h:319-344) appends every decoded pixel and the voxels take the first L: the
    product's host decoder and the oracle's decoder must do the same with such a stream."""
    import sortform
    for n, seed in ((40, 1), (600, 2), (1500, 3)):
        pts = pkg.synthetic.sphere_shell(n, 0x11E5 + seed)
        stream, strip, want = sortform.reference_style_lines_stream(oracle, pts, octree_bits=6, jpeg_quality=75, frame_id=2)
        L = want.n_leaves
        ref = oracle.decode_intra(stream).points
        got, info = pkg.binding.Context(None).decode_intra(stream + b"next frame")
        assert info["consumed"] == len(stream) and len(got) == L
        assert got.tobytes() == ref.tobytes()
        own = oracle.decode_intra(want.bitstream).points       # the L x 1 strip this codec writes: same positions
        assert np.array_equal(got["x"], own["x"]) and np.array_equal(got["y"], own["y"]) and np.array_equal(got["z"], own["z"])
