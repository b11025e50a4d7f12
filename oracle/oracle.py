"""ctypes binding of the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this module.  It loads oracle/liboracle.so (built by oracle/Makefile)
and, when present, oracle/_ref/libsnake_ref.so (the reference's own
snake_grid_mapping.h compiled from /root/reference).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

POINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("rgba", "<u4"), ("pad", "<u4", (3,))]
)
assert POINT_DTYPE.itemsize == 32


class Buf(C.Structure):
    _fields_ = [("data", C.POINTER(C.c_uint8)), ("len", C.c_size_t), ("cap", C.c_size_t)]

    def bytes(self):
        return C.string_at(self.data, self.len) if self.len else b""


class Params(C.Structure):
    _fields_ = [
        ("octree_resolution", C.c_double),
        ("point_resolution", C.c_double),
        ("do_color_encoding", C.c_int),
        ("color_bit_resolution", C.c_int),
        ("color_coding_type", C.c_int),
        ("do_voxel_centroid", C.c_int),
        ("create_scalable", C.c_int),
        ("do_connectivity", C.c_int),
        ("jpeg_quality", C.c_int),
        ("macroblock_size", C.c_int),
        ("do_icp_color_offset", C.c_int),
        ("frame_id", C.c_uint32),
    ]


class Frame(C.Structure):
    _fields_ = [
        ("bbox", C.c_double * 6),
        ("depth", C.c_uint32),
        ("n_points_in", C.c_uint64),
        ("n_leaves", C.c_uint64),
        ("n_branches", C.c_uint64),
        ("leaf_keys", C.POINTER(C.c_uint32)),
        ("leaf_counts", C.POINTER(C.c_uint32)),
        ("occupancy", Buf),
        ("bgr", Buf),
        ("centroid_bytes", Buf),
        ("color_payload", Buf),
        ("snake_image", Buf),
        ("image_w", C.c_uint32),
        ("image_h", C.c_uint32),
        ("simplified", C.c_void_p),
        ("bitstream", Buf),
        ("perf", C.c_uint64 * 3),
    ]


class Cloud(C.Structure):
    _fields_ = [
        ("points", C.c_void_p),
        ("n", C.c_uint64),
        ("params", Params),
        ("bbox", C.c_double * 6),
        ("depth", C.c_uint32),
        ("consumed", C.c_size_t),
    ]


def build(opt0=False):
    """Compile the oracle (and oracle/_ref when /root/reference is present)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)


def _load(name="liboracle.so"):
    path = os.path.join(_HERE, name)
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    lib.pcco_rc_encode.restype = C.c_size_t
    lib.pcco_rc_encode.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Buf)]
    lib.pcco_rc_decode.restype = C.c_size_t
    lib.pcco_rc_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    lib.pcco_snake_perm.argtypes = [C.c_int, C.c_int, C.c_void_p]
    lib.pcco_jpeg_encode_rgb.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Buf)]
    lib.pcco_jpeg_decode_rgb.argtypes = [
        C.c_void_p, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pcco_encode_intra.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Params), C.POINTER(Frame)]
    lib.pcco_decode_intra.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(Cloud)]
    lib.pcco_normalize_single.argtypes = [C.c_void_p, C.c_size_t, C.c_double, C.c_void_p, C.c_void_p]
    lib.pcco_buf_free.argtypes = [C.POINTER(Buf)]
    lib.pcco_frame_free.argtypes = [C.POINTER(Frame)]
    lib.pcco_cloud_free.argtypes = [C.POINTER(Cloud)]
    return lib


_lib = None
_lib_o0 = None


def lib(opt0=False):
    global _lib, _lib_o0
    if opt0:
        if _lib_o0 is None:
            _lib_o0 = _load("liboracle_O0.so")
        return _lib_o0
    if _lib is None:
        _lib = _load()
    return _lib


def ref_snake_lib():
    """The reference's own snake_grid_mapping.h, compiled (None if not built)."""
    path = os.path.join(_HERE, "_ref", "libsnake_ref.so")
    if not os.path.exists(path):
        return None
    r = C.CDLL(path)
    r.ref_snake_perm.argtypes = [C.c_int, C.c_int, C.c_void_p]
    r.ref_snake_do_mapping.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    r.ref_snake_undo_mapping.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return r


def make_params(octree_bits=10, enh_bits=0, color_bits=8, color_coding_type=1, keep_centroid=0,
                jpeg_quality=85, frame_id=1, octree_resolution=None, point_resolution=None,
                create_scalable=0, macroblock_size=16, do_icp_color_offset=0):
    """App-side parameterisation, eval.hpp:377-395 (MANUAL_CONFIGURATION)."""
    p = Params()
    p.octree_resolution = octree_resolution if octree_resolution is not None else 2.0 ** (-octree_bits)
    p.point_resolution = point_resolution if point_resolution is not None else 2.0 ** (-(octree_bits + enh_bits))
    p.do_color_encoding = 1 if color_bits > 0 else 0
    p.color_bit_resolution = color_bits
    p.color_coding_type = color_coding_type
    p.do_voxel_centroid = keep_centroid
    p.create_scalable = create_scalable
    p.do_connectivity = 0
    p.jpeg_quality = jpeg_quality
    p.macroblock_size = macroblock_size
    p.do_icp_color_offset = do_icp_color_offset
    p.frame_id = frame_id
    return p


def rc_encode(data: bytes) -> bytes:
    b = Buf()
    arr = np.frombuffer(data, dtype=np.uint8)
    lib().pcco_rc_encode(arr.ctypes.data if len(arr) else None, len(arr), C.byref(b))
    out = b.bytes()
    lib().pcco_buf_free(C.byref(b))
    return out


def rc_decode(stream: bytes, n: int):
    out = np.zeros(max(n, 1), dtype=np.uint8)
    src = np.frombuffer(stream, dtype=np.uint8)
    used = lib().pcco_rc_decode(src.ctypes.data, len(src), out.ctypes.data, n)
    return out[:n].tobytes(), used


def snake_perm(w, h):
    perm = np.zeros(w * h, dtype=np.int32)
    lib().pcco_snake_perm(w, h, perm.ctypes.data)
    return perm


def jpeg_encode(rgb: np.ndarray, quality: int) -> bytes:
    h, w, _ = rgb.shape
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    b = Buf()
    rc = lib().pcco_jpeg_encode_rgb(rgb.ctypes.data, w, h, quality, C.byref(b))
    assert rc == 0
    out = b.bytes()
    lib().pcco_buf_free(C.byref(b))
    return out


def jpeg_decode(jpg: bytes) -> np.ndarray:
    src = np.frombuffer(jpg, dtype=np.uint8)
    p = C.POINTER(C.c_uint8)()
    w = C.c_int()
    h = C.c_int()
    rc = lib().pcco_jpeg_decode_rgb(src.ctypes.data, len(src), C.byref(p), C.byref(w), C.byref(h))
    if rc != 0:
        raise ValueError("oracle jpeg decode failed: %d" % rc)
    arr = np.ctypeslib.as_array(p, shape=(h.value, w.value, 3)).copy()
    C.CDLL(None).free(p)
    return arr


class EncodeResult:
    pass


def encode_intra(points: np.ndarray, params: Params, opt0=False, keep=True):
    """Run the oracle encoder. Returns None if the frame is dropped (empty)."""
    assert points.dtype == POINT_DTYPE
    points = np.ascontiguousarray(points)
    f = Frame()
    L = lib(opt0)
    rc = L.pcco_encode_intra(points.ctypes.data, len(points), C.byref(params), C.byref(f))
    if rc == 2:   # the adaptive box would need more than 32 levels: beyond what PCL's own arithmetic defines
        L.pcco_frame_free(C.byref(f))
        r = EncodeResult()
        r.depth = 33
        r.too_deep = True
        return r
    if rc != 0:
        L.pcco_frame_free(C.byref(f))
        return None
    r = EncodeResult()
    r.bbox = np.array(list(f.bbox), dtype=np.float64)
    r.depth = int(f.depth)
    r.n_points_in = int(f.n_points_in)
    r.n_leaves = int(f.n_leaves)
    r.n_branches = int(f.n_branches)
    r.perf = [int(x) for x in f.perf]
    r.bitstream = f.bitstream.bytes()
    if keep:
        n = r.n_leaves
        r.leaf_keys = np.ctypeslib.as_array(f.leaf_keys, shape=(n, 3)).copy()
        r.leaf_counts = np.ctypeslib.as_array(f.leaf_counts, shape=(n,)).copy()
        r.occupancy = np.frombuffer(f.occupancy.bytes(), dtype=np.uint8).copy()
        r.bgr = np.frombuffer(f.bgr.bytes(), dtype=np.uint8).copy()
        r.centroid_bytes = np.frombuffer(f.centroid_bytes.bytes(), dtype=np.uint8).copy()
        r.color_payload = f.color_payload.bytes()
        r.snake_image = np.frombuffer(f.snake_image.bytes(), dtype=np.uint8).copy()
        r.image_w, r.image_h = int(f.image_w), int(f.image_h)
        r.simplified = np.frombuffer(C.string_at(f.simplified, 32 * n), dtype=POINT_DTYPE).copy()
    L.pcco_frame_free(C.byref(f))
    return r


class DecodeResult:
    pass


def decode_intra(bitstream: bytes):
    src = np.frombuffer(bitstream, dtype=np.uint8)
    c = Cloud()
    rc = lib().pcco_decode_intra(src.ctypes.data, len(src), C.byref(c))
    if rc != 0:
        lib().pcco_cloud_free(C.byref(c))
        raise ValueError("oracle decode failed: %d" % rc)
    r = DecodeResult()
    r.points = np.frombuffer(C.string_at(c.points, 32 * c.n), dtype=POINT_DTYPE).copy()
    r.bbox = np.array(list(c.bbox))
    r.depth = int(c.depth)
    r.consumed = int(c.consumed)
    r.params = {k: getattr(c.params, k) for k, _ in Params._fields_}
    lib().pcco_cloud_free(C.byref(c))
    return r


def normalize_single(points: np.ndarray, bb_expand_factor=0.2):
    """normalize_pointclouds for one cloud (in place); returns (bb_min, bb_max)."""
    mn = np.zeros(3, dtype=np.float32)
    mx = np.zeros(3, dtype=np.float32)
    lib().pcco_normalize_single(points.ctypes.data, len(points), bb_expand_factor, mn.ctypes.data, mx.ctypes.data)
    return mn, mx
