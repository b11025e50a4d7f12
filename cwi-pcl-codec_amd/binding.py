"""ctypes binding of libpcc_hip.so (include/pcc_codec.h) and a host-side mirror of the
reference class interface.

`OctreePointCloudCodecV2` below has the public surface of
pcl::io::OctreePointCloudCodecV2<PointXYZRGB> (codec.h:108-227): same constructor argument
order and meaning, encodePointCloud / decodePointCloud / getPerformanceMetrics /
getOutputCloud, so the parity tests read like calls into the reference.  Everything it
does goes through the C ABI; there is no Python or CPU implementation of the hot path
here, and constructing it without a usable GPU raises.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# PCC_LIB: developer builds of the same library (e.g. libpcc_hip_ktime.so); there is no fallback either way
LIB_PATH = os.environ.get("PCC_LIB") or os.path.join(_HERE, "libpcc_hip.so")

POINT_DTYPE = np.dtype(
    [("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("w", "<f4"), ("rgba", "<u4"), ("pad", "<u4", (3,))]
)

PCC_OK = 0
PCC_NO_NUMA_NODE = -100   # pcc_pipeline_get(p, "numa_node"): the cores were not chosen by NUMA node
ERR_NAMES = {-1: "PCC_ERR_ARG", -2: "PCC_ERR_HIP", -3: "PCC_ERR_EMPTY", -4: "PCC_ERR_UNSUPPORTED",
             -5: "PCC_ERR_STREAM", -6: "PCC_ERR_STATE"}

# every symbol include/pcc_codec.h declares: the drop-in boundary (the class shim, the evaluation app, a frame loop)
BOUNDARY_EXPORTS = [
    "pcc_create", "pcc_destroy", "pcc_last_error", "pcc_version",
    "pcc_encode_intra", "pcc_encode_intra_device", "pcc_reserve", "pcc_hotpath_launch", "pcc_hotpath_finish", "pcc_entropy_encode",
    "pcc_get_output_cloud", "pcc_decode_intra", "pcc_decode_intra_gpu",
    "pcc_device_alloc", "pcc_device_free", "pcc_device_upload", "pcc_set_option",
    "pcc_pipeline_create", "pcc_pipeline_destroy", "pcc_pipeline_set_option", "pcc_pipeline_get", "pcc_pipeline_contexts", "pcc_pipeline_context",
    "pcc_pipeline_encode", "pcc_pipeline_encode_host", "pcc_pipeline_reserve", "pcc_pipeline_last_error",
    "pcc_pipeline_create_multi", "pcc_multi_pipeline_destroy", "pcc_multi_pipeline_size", "pcc_multi_pipeline_member",
    "pcc_multi_pipeline_encode_host", "pcc_multi_pipeline_encode", "pcc_multi_pipeline_last_error",
    "pcc_quality_metrics", "pcc_remove_outliers", "pcc_encode_delta", "pcc_decode_delta",
    "pcc_normalize_group_boxes", "pcc_restore_scaling",
]
# every symbol include/pcc_codec_tools.h declares: measurement, tests, tools
TOOLS_EXPORTS = [
    "pcc_create_host",
    "pcc_get_kernel_times", "pcc_get_kernel_spans", "pcc_get_kernel_span_starts", "pcc_get_host_times", "pcc_set_profiling", "pcc_get_decode_times",
    "pcc_hotpath_launch_host", "pcc_upload_lane_create", "pcc_upload_lane_destroy", "pcc_host_alloc", "pcc_host_free",
    "pcc_stream_create", "pcc_stream_destroy", "pcc_use_stream", "pcc_entropy_encode_many",
    "pcc_pipeline_gpu_stage_only", "pcc_pipeline_stats", "pcc_pipeline_cpu_times", "pcc_pipeline_kernel_times",
    "pcc_delta_blocks", "pcc_device_range_encode",
    "pcc_entropy_batch_create", "pcc_entropy_batch_destroy", "pcc_entropy_batch_size", "pcc_entropy_batch_capacity",
    "pcc_entropy_batch_add", "pcc_entropy_batch_flush", "pcc_entropy_batch_last_error", "pcc_entropy_batch_set_option",
    "pcc_host_rigid_compress", "pcc_host_rigid_decompress",
    "pcc_host_range_encode", "pcc_host_range_encode_many", "pcc_host_range_decode", "pcc_host_jpeg_encode", "pcc_host_jpeg_decode",
    "pcc_host_snake_position",
    "pcc_debug_sort_plan", "pcc_debug_pipeline_cpus", "pcc_debug_host_rc_wide",
    "pcc_debug_device_pci_bus_id", "pcc_debug_device_numa_node", "pcc_debug_numa_plan", "pcc_debug_address_node",
]
EXPORTS = BOUNDARY_EXPORTS + TOOLS_EXPORTS


# csrc/pcc_dev.h: read by `make dev` builds (libpcc_hip_dev.so) and the executor, a constant nullptr in the shipped library
DEV_SWITCHES = ("PCC_LEAF_PROBES", "PCC_LEAF_ROWS", "PCC_SORT_SHAPE", "PCC_WAIT", "PCC_WAIT_STATS", "PCC_FINISH_TRACE", "PCC_TRACE_DELTA",
                "PCC_DECODE_TRACE", "PCC_PIPELINE_TRACE", "PCC_DECODE_SERIAL", "PCC_PIPELINE_SPREAD", "PCC_PIPELINE_OWN_STREAMS", "PCC_RC_WIDE",
                "PCC_RC_DEVICE", "PCC_PACK_UPLOAD", "PCC_SYSFS_ROOT")


class PccError(RuntimeError):
    def __init__(self, code, text):
        RuntimeError.__init__(self, "%s: %s" % (ERR_NAMES.get(code, str(code)), text))
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("octree_resolution", C.c_double), ("point_resolution", C.c_double),
        ("do_color_encoding", C.c_int32), ("color_bit_resolution", C.c_int32),
        ("color_coding_type", C.c_int32), ("do_voxel_centroid", C.c_int32),
        ("create_scalable", C.c_int32), ("do_connectivity", C.c_int32),
        ("jpeg_quality", C.c_int32), ("macroblock_size", C.c_int32),
        ("do_icp_color_offset", C.c_int32), ("frame_id", C.c_uint32),
    ]


class HotResult(C.Structure):
    _fields_ = [
        ("bbox", C.c_double * 6), ("depth", C.c_uint32), ("n_epochs", C.c_uint32),
        ("n_points_in", C.c_uint64), ("n_leaves", C.c_uint64), ("n_branches", C.c_uint64),
        ("occupancy", C.c_void_p), ("bgr", C.c_void_p), ("centroid", C.c_void_p), ("image", C.c_void_p),
        ("image_w", C.c_uint32), ("image_h", C.c_uint32), ("gpu_ms", C.c_float), ("jpeg_coefs", C.c_void_p),
        ("jpeg_tiles", C.c_void_p), ("jpeg_tile_words", C.c_uint32), ("jpeg_n_tiles", C.c_uint32),
        ("occupancy_histogram", C.c_void_p),
        ("jpeg_lines_dir", C.c_void_p), ("jpeg_lines_data", C.c_void_p), ("jpeg_n_lines", C.c_uint32),
    ]


class Bitstream(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("perf", C.c_uint64 * 3)]


class Cloud(C.Structure):
    _fields_ = [("points", C.c_void_p), ("n", C.c_size_t), ("params", Params), ("bbox", C.c_double * 6),
                ("depth", C.c_uint32), ("consumed", C.c_size_t)]


class Quality(C.Structure):
    _fields_ = [("in_point_count", C.c_uint64), ("out_point_count", C.c_uint64), ("symm_rms", C.c_float),
                ("symm_hausdorff", C.c_float), ("left_hausdorff", C.c_float), ("right_hausdorff", C.c_float),
                ("left_rms", C.c_float), ("right_rms", C.c_float), ("psnr_db", C.c_double), ("psnr_yuv", C.c_double * 3),
                ("gpu_ms", C.c_float)]


class DeltaParams(C.Structure):
    _fields_ = [("codec", Params), ("icp_on_original", C.c_int32), ("write_out_cloud", C.c_int32),
                ("icp_max_iterations", C.c_int32), ("icp_var_threshold", C.c_float), ("transformation_epsilon", C.c_float)]


class DeltaResult(C.Structure):
    _fields_ = [("i_data", C.c_void_p), ("i_len", C.c_size_t), ("p_data", C.c_void_p), ("p_len", C.c_size_t),
                ("out_cloud", C.c_void_p), ("out_n", C.c_size_t), ("macro_block_count", C.c_uint32),
                ("shared_macroblock_count", C.c_uint32), ("convergence_count", C.c_uint32),
                ("shared_macroblock_percentage", C.c_float), ("shared_macroblock_convergence_percentage", C.c_float),
                ("n_intra_points", C.c_uint64), ("n_simplified", C.c_uint64), ("gpu_ms", C.c_float)]


DELTA_BLOCK_DTYPE = np.dtype([("i_block", "<i4"), ("n_p", "<u4"), ("n_i", "<u4"), ("do_icp", "<i4"), ("converged", "<i4"),
                              ("iterations", "<i4"), ("rgb_offsets", "i1", (4,)), ("key", "<u2", (4,)), ("fitness", "<f4"),
                              ("rt", "<f4", (16,))])


class KernelTimes(C.Structure):
    _fields_ = [("count", C.c_int32), ("name", C.c_char_p * 64), ("ms", C.c_float * 64)]


_lib = None

PRODUCT_VERSION = "pcc_hip 0.1 (gfx950)"


def library_identity():
    """What is loaded: {"file": basename, "version": pcc_version()}.  Goes into bench.py's line."""
    lib = load_library()
    return {"file": os.path.basename(LIB_PATH), "version": lib.pcc_version().decode()}


def require_product_library(who):
    """bench.py and smoke() measure / check the gfx950 library and nothing else: PCC_LIB can point the binding at the CPU
    executor's build of the same sources (tests) or at a developer build -- numbers from those are not the product's.
    PCC_ALLOW_NON_PRODUCT_LIB=1 (set by tests/emu/bench_on_executor.py and the A/B tools) lifts the refusal; the identity
    is reported either way."""
    ident = library_identity()
    if (ident["version"] != PRODUCT_VERSION or ident["file"] != "libpcc_hip.so") and os.environ.get("PCC_ALLOW_NON_PRODUCT_LIB") != "1":
        raise SystemExit("%s: refusing to run on %s (%s): not the gfx950 product library" % (who, ident["file"], ident["version"]))
    return ident


def load_library():
    """Load libpcc_hip.so; raises (never falls back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libpcc_hip.so is missing: run `python __graft_entry__.py` (build()) first")
    lib = C.CDLL(LIB_PATH)
    vp, sz, i32 = C.c_void_p, C.c_size_t, C.c_int
    lib.pcc_create.restype = vp
    lib.pcc_create.argtypes = [i32]
    lib.pcc_create_host.restype = vp
    lib.pcc_create_host.argtypes = []
    lib.pcc_destroy.argtypes = [vp]
    lib.pcc_destroy.restype = None
    lib.pcc_last_error.restype = C.c_char_p
    lib.pcc_last_error.argtypes = [vp]
    lib.pcc_version.restype = C.c_char_p
    lib.pcc_encode_intra.argtypes = [vp, vp, sz, sz, sz, C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_encode_intra_device.argtypes = [vp, vp, sz, sz, sz, C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_hotpath_launch.argtypes = [vp, vp, sz, sz, sz, C.POINTER(Params)]
    lib.pcc_hotpath_finish.argtypes = [vp, C.POINTER(HotResult)]
    lib.pcc_hotpath_launch_host.argtypes = [vp, vp, vp, sz, sz, sz, C.POINTER(Params)]
    lib.pcc_upload_lane_create.restype = vp
    lib.pcc_upload_lane_create.argtypes = [i32]
    lib.pcc_upload_lane_destroy.argtypes = [vp]
    lib.pcc_upload_lane_destroy.restype = None
    lib.pcc_stream_create.restype = vp
    lib.pcc_stream_create.argtypes = [i32]
    lib.pcc_stream_destroy.argtypes = [vp]
    lib.pcc_stream_destroy.restype = None
    lib.pcc_use_stream.argtypes = [vp, vp]
    lib.pcc_host_alloc.restype = vp
    lib.pcc_host_alloc.argtypes = [sz]
    lib.pcc_host_free.argtypes = [vp]
    lib.pcc_host_free.restype = None
    lib.pcc_entropy_encode.argtypes = [vp, C.POINTER(HotResult), C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_entropy_encode_many.argtypes = [i32, C.POINTER(vp), C.POINTER(C.POINTER(HotResult)), C.POINTER(C.POINTER(Params)),
                                            C.POINTER(C.POINTER(Bitstream))]
    lib.pcc_get_output_cloud.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    lib.pcc_decode_intra.argtypes = [vp, vp, sz, C.POINTER(Cloud)]
    lib.pcc_decode_intra_gpu.argtypes = [vp, vp, sz, C.POINTER(Cloud)]
    lib.pcc_get_decode_times.argtypes = [vp, C.POINTER(C.c_double)]
    lib.pcc_device_alloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.pcc_device_free.argtypes = [vp, vp]
    lib.pcc_device_upload.argtypes = [vp, vp, vp, sz]
    lib.pcc_get_kernel_spans.argtypes = [vp, C.POINTER(KernelTimes)]
    lib.pcc_get_kernel_span_starts.argtypes = [vp, C.POINTER(KernelTimes)]
    lib.pcc_get_kernel_times.argtypes = [vp, C.POINTER(KernelTimes)]
    lib.pcc_get_host_times.argtypes = [vp, C.POINTER(C.c_double)]
    lib.pcc_set_profiling.argtypes = [vp, i32]
    lib.pcc_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.pcc_pipeline_create.restype = vp
    lib.pcc_pipeline_create.argtypes = [i32, i32]
    lib.pcc_pipeline_destroy.argtypes = [vp]
    lib.pcc_pipeline_destroy.restype = None
    lib.pcc_pipeline_get.argtypes = [vp, C.c_char_p]
    lib.pcc_entropy_batch_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.pcc_debug_sort_plan.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.pcc_debug_pipeline_cpus.argtypes = [vp, i32, C.POINTER(i32), i32]
    lib.pcc_debug_host_rc_wide.argtypes = []
    lib.pcc_pipeline_contexts.argtypes = [vp]
    lib.pcc_pipeline_context.restype = vp
    lib.pcc_pipeline_context.argtypes = [vp, i32]
    lib.pcc_entropy_batch_create.restype = vp
    lib.pcc_entropy_batch_create.argtypes = [i32, sz]
    lib.pcc_entropy_batch_destroy.argtypes = [vp]
    lib.pcc_entropy_batch_destroy.restype = None
    lib.pcc_entropy_batch_size.restype = sz
    lib.pcc_entropy_batch_size.argtypes = [vp]
    lib.pcc_entropy_batch_capacity.restype = sz
    lib.pcc_entropy_batch_capacity.argtypes = [vp]
    lib.pcc_entropy_batch_add.argtypes = [vp, C.POINTER(HotResult), C.POINTER(Params)]
    lib.pcc_entropy_batch_flush.argtypes = [vp, C.POINTER(Bitstream), sz, C.POINTER(sz)]
    lib.pcc_entropy_batch_last_error.restype = C.c_char_p
    lib.pcc_entropy_batch_last_error.argtypes = [vp]
    lib.pcc_pipeline_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.pcc_debug_device_pci_bus_id.argtypes = [i32, C.c_char_p, i32]
    lib.pcc_debug_device_numa_node.argtypes = [i32, C.c_char_p]
    lib.pcc_debug_numa_plan.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), i32, C.POINTER(i32), i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), i32]
    lib.pcc_debug_address_node.argtypes = [vp]
    lib.pcc_pipeline_create_multi.restype = vp
    lib.pcc_pipeline_create_multi.argtypes = [C.POINTER(i32), i32, i32]
    lib.pcc_multi_pipeline_destroy.argtypes = [vp]
    lib.pcc_multi_pipeline_destroy.restype = None
    lib.pcc_multi_pipeline_size.argtypes = [vp]
    lib.pcc_multi_pipeline_member.restype = vp
    lib.pcc_multi_pipeline_member.argtypes = [vp, i32]
    lib.pcc_multi_pipeline_last_error.restype = C.c_char_p
    lib.pcc_multi_pipeline_last_error.argtypes = [vp]
    for fn in (lib.pcc_multi_pipeline_encode_host, lib.pcc_multi_pipeline_encode):
        fn.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, sz, C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_pipeline_encode_host.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, sz, C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_pipeline_encode.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, sz, C.POINTER(Params), C.POINTER(Bitstream)]
    lib.pcc_pipeline_reserve.argtypes = [vp, sz, sz, sz]
    lib.pcc_reserve.argtypes = [vp, sz, sz]
    lib.pcc_pipeline_gpu_stage_only.argtypes = [vp, C.POINTER(vp), C.POINTER(sz), sz, sz, sz, C.POINTER(Params)]
    lib.pcc_pipeline_stats.argtypes = [vp, C.POINTER(C.c_double)]
    lib.pcc_pipeline_cpu_times.argtypes = [vp, C.POINTER(C.c_double)]
    lib.pcc_pipeline_kernel_times.argtypes = [vp, C.POINTER(KernelTimes), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pcc_pipeline_last_error.restype = C.c_char_p
    lib.pcc_pipeline_last_error.argtypes = [vp]
    lib.pcc_quality_metrics.argtypes = [vp, vp, sz, vp, sz, C.c_double, C.POINTER(Quality)]
    lib.pcc_device_range_encode.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(C.c_float)]
    lib.pcc_remove_outliers.argtypes = [vp, vp, sz, i32, C.c_double, vp, C.POINTER(sz)]
    lib.pcc_encode_delta.argtypes = [vp, vp, sz, vp, sz, C.POINTER(DeltaParams), C.POINTER(DeltaResult)]
    lib.pcc_delta_blocks.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    lib.pcc_decode_delta.argtypes = [vp, vp, sz, vp, sz, vp, sz, C.POINTER(DeltaParams), C.POINTER(Cloud)]
    lib.pcc_host_rigid_compress.restype = sz
    lib.pcc_host_rigid_compress.argtypes = [vp, vp, sz]
    lib.pcc_host_rigid_decompress.argtypes = [vp, sz, vp]
    lib.pcc_host_range_encode.restype = sz
    lib.pcc_host_range_encode.argtypes = [vp, sz, vp, sz]
    lib.pcc_host_range_encode_many.argtypes = [i32, C.POINTER(vp), C.POINTER(sz), C.POINTER(vp), C.POINTER(sz), C.POINTER(sz)]
    lib.pcc_host_range_decode.restype = sz
    lib.pcc_host_range_decode.argtypes = [vp, sz, vp, sz]
    lib.pcc_host_jpeg_encode.restype = sz
    lib.pcc_host_jpeg_encode.argtypes = [vp, i32, i32, i32, vp, sz]
    lib.pcc_host_jpeg_decode.argtypes = [vp, sz, vp, sz, C.POINTER(i32), C.POINTER(i32)]
    lib.pcc_host_snake_position.restype = C.c_uint32
    lib.pcc_host_snake_position.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]
    lib.pcc_normalize_group_boxes.argtypes = [C.POINTER(vp), C.POINTER(sz), sz, C.c_double, vp, vp, vp]
    lib.pcc_restore_scaling.argtypes = [vp, sz, vp, vp]
    # the developer switches of csrc/pcc_dev.h are read by the developer build and the executor only: set for a library that
    # ignores them, a tool would measure something else than it says
    ignored = [k for k in DEV_SWITCHES if k in os.environ]
    version = lib.pcc_version().decode()
    if ignored and "dev build" not in version and not version.startswith("pcc_emu"):
        import warnings
        warnings.warn("%s set, but the library loaded (%s: %s) does not read developer switches: use PCC_LIB=%s"
                      % (", ".join(ignored), os.path.basename(LIB_PATH), version, os.path.join(_HERE, "libpcc_hip_dev.so")), RuntimeWarning, stacklevel=2)
    _lib = lib
    return lib


def make_params(octree_bits=10, enh_bits=0, color_bits=8, color_coding_type=1, keep_centroid=0, jpeg_quality=85,
                frame_id=1, octree_resolution=None, point_resolution=None, create_scalable=0, macroblock_size=16,
                do_icp_color_offset=0):
    """The app's parameterisation of the codec (eval.hpp:377-395)."""
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


def _bytes_at(ptr, n):
    return C.string_at(ptr, n) if (ptr and n) else b""


class HotProducts:
    """Host copies of what the GPU stage produced for one frame."""

    def __init__(self, hr: HotResult, copy=True):
        self.raw = hr
        self.bbox = np.array(list(hr.bbox), dtype=np.float64)
        self.depth = int(hr.depth)
        self.n_epochs = int(hr.n_epochs)
        self.n_points_in = int(hr.n_points_in)
        self.n_leaves = int(hr.n_leaves)
        self.n_branches = int(hr.n_branches)
        self.image_w, self.image_h = int(hr.image_w), int(hr.image_h)
        self.gpu_ms = float(hr.gpu_ms)
        if copy:
            L, B = self.n_leaves, self.n_branches
            self.occupancy = np.frombuffer(_bytes_at(hr.occupancy, B), dtype=np.uint8)
            self.bgr = np.frombuffer(_bytes_at(hr.bgr, 3 * L), dtype=np.uint8)
            self.centroid_bytes = np.frombuffer(_bytes_at(hr.centroid, 3 * L), dtype=np.uint8)
            self.snake_image = np.frombuffer(_bytes_at(hr.image, 3 * self.image_w * self.image_h), dtype=np.uint8)
            self.occupancy_histogram = np.frombuffer(_bytes_at(hr.occupancy_histogram, 1024 if hr.occupancy_histogram else 0), dtype=np.uint32)


class Context:
    """One pcc_ctx: one GPU, one stream.  `device=None` makes a host-only context."""

    def __init__(self, device=0):
        self.lib = load_library()
        self.h = self.lib.pcc_create_host() if device is None else self.lib.pcc_create(device)
        if not self.h:
            raise RuntimeError("pcc_create(%r) failed: no usable MI355X/HIP device -- the hot path has no CPU "
                               "fallback" % (device,))
        self._dev_allocs = []

    def close(self):
        if self.h:
            for p in self._dev_allocs:
                self.lib.pcc_device_free(self.h, p)
            self._dev_allocs = []
            self.lib.pcc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != PCC_OK:
            raise PccError(rc, self.lib.pcc_last_error(self.h).decode())

    # ---- device memory ----
    def upload(self, points: np.ndarray):
        points = np.ascontiguousarray(points)
        p = C.c_void_p()
        self._check(self.lib.pcc_device_alloc(self.h, max(points.nbytes, 16), C.byref(p)))
        self._dev_allocs.append(p)
        if points.nbytes:
            self._check(self.lib.pcc_device_upload(self.h, p, points.ctypes.data, points.nbytes))
        return p

    def free(self, dev_ptr):
        self._dev_allocs = [p for p in self._dev_allocs if p.value != dev_ptr.value]
        self._check(self.lib.pcc_device_free(self.h, dev_ptr))

    # ---- stages ----
    def hotpath_launch(self, dev_ptr, n, params, stride=32, rgb_offset=16):
        self._check(self.lib.pcc_hotpath_launch(self.h, dev_ptr, n, stride, rgb_offset, C.byref(params)))

    def hotpath_launch_host(self, points: np.ndarray, params, lane=None, stride=None, rgb_offset=16):
        """The frame is in host memory (a numpy array): asynchronous upload, then the kernels.  Keep `points` alive
        and untouched until hotpath_finish has returned."""
        pts = np.ascontiguousarray(points)
        stride = stride or pts.dtype.itemsize
        self._check(self.lib.pcc_hotpath_launch_host(self.h, lane, pts.ctypes.data, len(pts), stride, rgb_offset, C.byref(params)))
        self._host_frame = pts

    def hotpath_finish(self, copy=True):
        hr = HotResult()
        self._check(self.lib.pcc_hotpath_finish(self.h, C.byref(hr)))
        return HotProducts(hr, copy=copy)

    def entropy_encode(self, hot: HotResult, params, copy=True):
        bs = Bitstream()
        self._check(self.lib.pcc_entropy_encode(self.h, C.byref(hot), C.byref(params), C.byref(bs)))
        return (_bytes_at(bs.data, bs.len) if copy else bs.len), [int(x) for x in bs.perf]

    def entropy_encode2(self, hot_a: HotResult, params_a, other, hot_b: HotResult, params_b):
        """Two frames at once (the other frame's bitstream is stored in context `other`)."""
        return tuple(Context.entropy_encode_many([self, other], [hot_a, hot_b], [params_a, params_b]))

    @staticmethod
    def entropy_encode_many(ctxs, hots, params):
        """Up to four frames at once: [(bytes, perf)] in the order given (frame i's bitstream lives in ctxs[i])."""
        n = len(ctxs)
        lib = ctxs[0].lib
        outs = [Bitstream() for _ in range(n)]
        rc = lib.pcc_entropy_encode_many(
            n, (C.c_void_p * n)(*[c.h for c in ctxs]),
            (C.POINTER(HotResult) * n)(*[C.pointer(h) for h in hots]),
            (C.POINTER(Params) * n)(*[C.pointer(p) for p in params]),
            (C.POINTER(Bitstream) * n)(*[C.pointer(o) for o in outs]))
        ctxs[0]._check(rc)
        return [(_bytes_at(o.data, o.len), [int(x) for x in o.perf]) for o in outs]

    def encode_intra_host(self, points: np.ndarray, params, stride=32, rgb_offset=16):
        points = np.ascontiguousarray(points)
        bs = Bitstream()
        self._check(self.lib.pcc_encode_intra(self.h, points.ctypes.data, len(points), stride, rgb_offset,
                                              C.byref(params), C.byref(bs)))
        return _bytes_at(bs.data, bs.len), [int(x) for x in bs.perf]

    def encode_intra_device(self, dev_ptr, n, params, stride=32, rgb_offset=16, copy=True):
        bs = Bitstream()
        self._check(self.lib.pcc_encode_intra_device(self.h, dev_ptr, n, stride, rgb_offset, C.byref(params),
                                                     C.byref(bs)))
        return (_bytes_at(bs.data, bs.len) if copy else bs.len), [int(x) for x in bs.perf]

    def output_cloud(self):
        p = C.c_void_p()
        n = C.c_size_t()
        self._check(self.lib.pcc_get_output_cloud(self.h, C.byref(p), C.byref(n)))
        return np.frombuffer(_bytes_at(p, 32 * n.value), dtype=POINT_DTYPE).copy()

    def decode_intra(self, stream: bytes, on_gpu=False):
        """decodePointCloud; on_gpu: the data-parallel half runs on the GPU (pcc_decode_intra_gpu), same cloud."""
        src = np.frombuffer(stream, dtype=np.uint8)
        c = Cloud()
        fn = self.lib.pcc_decode_intra_gpu if on_gpu else self.lib.pcc_decode_intra
        self._check(fn(self.h, src.ctypes.data, len(src), C.byref(c)))
        pts = np.frombuffer(_bytes_at(c.points, 32 * c.n), dtype=POINT_DTYPE).copy()
        info = dict(bbox=np.array(list(c.bbox)), depth=int(c.depth), consumed=int(c.consumed),
                    params={k: getattr(c.params, k) for k, _ in Params._fields_})
        return pts, info

    def decode_times(self):
        """ms of the last decode_intra(on_gpu=True): sequential host stages, upload + kernels + download, whole call."""
        buf = (C.c_double * 3)()
        self._check(self.lib.pcc_get_decode_times(self.h, buf))
        return dict(host_sequential_ms=buf[0], gpu_ms=buf[1], total_ms=buf[2])

    def quality_metrics(self, cloud_a: np.ndarray, cloud_b: np.ndarray, cell_hint=0.0):
        """computeQualityMetric(original, decoded) (quality_metrics_impl.hpp:82-239) as a dict."""
        a, b2 = np.ascontiguousarray(cloud_a), np.ascontiguousarray(cloud_b)
        q = Quality()
        self._check(self.lib.pcc_quality_metrics(self.h, a.ctypes.data, len(a), b2.ctypes.data, len(b2), float(cell_hint), C.byref(q)))
        d = {k: getattr(q, k) for k, _ in Quality._fields_ if k != "psnr_yuv"}
        d["psnr_yuv"] = list(q.psnr_yuv)
        return d

    def device_range_encode(self, streams):
        """The static range coder on the GPU, one wave per stream: [bytes] -> ([bytes], kernel milliseconds)."""
        k = len(streams)
        ins = [np.frombuffer(s, dtype=np.uint8) for s in streams]
        outs = [np.zeros(1028 + len(a) + len(a) // 2 + 64, dtype=np.uint8) for a in ins]
        lens = (C.c_size_t * k)()
        ms = C.c_float()
        self._check(self.lib.pcc_device_range_encode(
            self.h, k, (C.c_void_p * k)(*[a.ctypes.data if len(a) else None for a in ins]), (C.c_size_t * k)(*[len(a) for a in ins]),
            (C.c_void_p * k)(*[o.ctypes.data for o in outs]), lens, C.byref(ms)))
        return [o[:lens[i]].tobytes() for i, o in enumerate(outs)], float(ms.value)

    def remove_outliers(self, cloud: np.ndarray, min_points: int, radius: float):
        """remove_outliers for one cloud (codec.h:216-217): the kept points, in order."""
        c = np.ascontiguousarray(cloud)
        keep = np.zeros(max(len(c), 1), dtype=np.uint8)
        n_kept = C.c_size_t()
        self._check(self.lib.pcc_remove_outliers(self.h, c.ctypes.data, len(c), int(min_points), float(radius), keep.ctypes.data, C.byref(n_kept)))
        out = c[keep[:len(c)] != 0]
        assert len(out) == n_kept.value
        return out

    def encode_delta(self, i_cloud: np.ndarray, p_cloud: np.ndarray, params, icp_on_original=False, write_out_cloud=True,
                     icp_max_iterations=0, icp_var_threshold=0.0, transformation_epsilon=0.0):
        """encodePointCloudDeltaFrame (codec.h:181-186): dict with i_stream, p_stream, out_cloud, statistics, blocks."""
        ic, pc = np.ascontiguousarray(i_cloud), np.ascontiguousarray(p_cloud)
        dp = DeltaParams()
        dp.codec = params
        dp.icp_on_original = 1 if icp_on_original else 0
        dp.write_out_cloud = 1 if write_out_cloud else 0
        dp.icp_max_iterations = icp_max_iterations
        dp.icp_var_threshold = icp_var_threshold
        dp.transformation_epsilon = transformation_epsilon
        r = DeltaResult()
        self._check(self.lib.pcc_encode_delta(self.h, ic.ctypes.data, len(ic), pc.ctypes.data, len(pc), C.byref(dp), C.byref(r)))
        bp, bn = C.c_void_p(), C.c_size_t()
        self._check(self.lib.pcc_delta_blocks(self.h, C.byref(bp), C.byref(bn)))
        out = {k: getattr(r, k) for k, _ in DeltaResult._fields_ if k not in ("i_data", "i_len", "p_data", "p_len", "out_cloud", "out_n")}
        out["i_stream"] = _bytes_at(r.i_data, r.i_len)
        out["p_stream"] = _bytes_at(r.p_data, r.p_len)
        out["out_cloud"] = np.frombuffer(_bytes_at(r.out_cloud, 32 * r.out_n), dtype=POINT_DTYPE).copy()
        out["blocks"] = np.frombuffer(_bytes_at(bp, DELTA_BLOCK_DTYPE.itemsize * bn.value), dtype=DELTA_BLOCK_DTYPE).copy()
        return out

    def decode_delta(self, i_cloud: np.ndarray, i_stream: bytes, p_stream: bytes, params):
        """decodePointCloudDeltaFrame (codec.h:188-191)."""
        ic = np.ascontiguousarray(i_cloud)
        a, b2 = np.frombuffer(i_stream, dtype=np.uint8), np.frombuffer(p_stream, dtype=np.uint8)
        dp = DeltaParams()
        dp.codec = params
        c = Cloud()
        import time
        t0 = time.perf_counter()
        self._check(self.lib.pcc_decode_delta(self.h, ic.ctypes.data, len(ic), a.ctypes.data if len(a) else None, len(a),
                                              b2.ctypes.data if len(b2) else None, len(b2), C.byref(dp), C.byref(c)))
        self.last_call_ms = (time.perf_counter() - t0) * 1e3   # the C call alone, without the copy into a numpy array below
        if not c.n:
            return np.zeros(0, dtype=POINT_DTYPE)
        return np.frombuffer((C.c_uint8 * (32 * c.n)).from_address(c.points), dtype=POINT_DTYPE).copy()

    def set_option(self, name, value):
        self._check(self.lib.pcc_set_option(self.h, name.encode(), int(value)))

    def set_profiling(self, on):
        self._check(self.lib.pcc_set_profiling(self.h, 1 if on else 0))

    def host_times(self):
        """(occupancy coder, JPEG, colour coder, whole stage) of the last entropy_encode, microseconds."""
        buf = (C.c_double * 4)()
        self._check(self.lib.pcc_get_host_times(self.h, buf))
        return tuple(buf)

    def kernel_times(self):
        kt = KernelTimes()
        self._check(self.lib.pcc_get_kernel_times(self.h, C.byref(kt)))
        return [(kt.name[i].decode(), float(kt.ms[i])) for i in range(kt.count)]

    def kernel_spans(self):
        """[(kernel, ms)] of the last profiled frame, measured on the GPU's own clock (first workgroup start to last wave end)."""
        kt = KernelTimes()
        self._check(self.lib.pcc_get_kernel_spans(self.h, C.byref(kt)))
        return [(kt.name[i].decode(), float(kt.ms[i])) for i in range(kt.count)]

    def kernel_pitches(self):
        """[(kernel, ms)]: from the start of a launch to the start of the next one of the frame (the last launch of the
        frame has no successor and is left out): what the launch cost its stream."""
        kt = KernelTimes()
        self._check(self.lib.pcc_get_kernel_span_starts(self.h, C.byref(kt)))
        st = [(kt.name[i].decode(), float(kt.ms[i])) for i in range(kt.count)]
        return [(st[i][0], st[i + 1][1] - st[i][1]) for i in range(len(st) - 1)]


def pinned_array(lib, template: np.ndarray):
    """A copy of `template` in page-locked host memory (pcc_host_alloc); free with lib.pcc_host_free(arr.ctypes.data)."""
    nbytes = template.nbytes
    ptr = lib.pcc_host_alloc(nbytes)
    if not ptr:
        raise MemoryError("pcc_host_alloc(%d)" % nbytes)
    buf = (C.c_uint8 * nbytes).from_address(ptr)
    arr = np.frombuffer(buf, dtype=template.dtype, count=len(template))
    arr[:] = template
    return arr


class _BorrowedContext(Context):
    """A pcc_ctx owned by a pipeline (never destroyed from here)."""

    def __init__(self, lib, handle):
        self.lib, self.h, self._dev_allocs = lib, handle, []

    def close(self):
        self.h = None


class Pipeline:
    """pcc_pipeline: `workers` host threads + contexts on one GPU encoding a sequence of frames."""

    def __init__(self, device=0, workers=8):
        self.lib = load_library()
        self.h = self.lib.pcc_pipeline_create(device, workers)
        if not self.h:
            raise RuntimeError("pcc_pipeline_create(%r) failed: no usable MI355X/HIP device -- the hot path has no "
                               "CPU fallback" % (device,))
        self.workers = self.get("workers")
        self.n_contexts = self.lib.pcc_pipeline_contexts(self.h)

    def close(self):
        if self.h:
            self.lib.pcc_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def last_entropy_mode(self):
        """Where the entropy stage of the last call ran: 0 host, 1 GPU (option "entropy_on_gpu")."""
        return self.get("last_entropy_mode")

    def get(self, name):
        """What the pipeline runs with: "workers", "gpu_threads", "contexts", "frames_per_coder_call", "last_entropy_mode",
        "rc_device_lanes", "entropy_gpu_batch" (pcc_pipeline_get)."""
        v = int(self.lib.pcc_pipeline_get(self.h, name.encode()))
        if name == "numa_node":
            return None if v == PCC_NO_NUMA_NODE else v      # the node whose cores the threads were given; None: not placed by node
        if v < 0:
            raise PccError(v, "pipeline value " + name)
        return v

    def set_option(self, name, value):
        rc = self.lib.pcc_pipeline_set_option(self.h, name.encode(), int(value))
        if rc != PCC_OK:
            raise PccError(rc, "pipeline option " + name)

    def context(self, index):
        """Context `index` of n_contexts (two per worker)."""
        return _BorrowedContext(self.lib, self.lib.pcc_pipeline_context(self.h, index))

    def _arrays(self, dev_frames, counts):
        k = len(dev_frames)
        fr = (C.c_void_p * k)(*[p.value if isinstance(p, C.c_void_p) else p for p in dev_frames])
        cn = (C.c_size_t * k)(*counts)
        return k, fr, cn

    def encode(self, dev_frames, counts, params, stride=32, rgb_offset=16, copy=True):
        """Frames f = 0.. get frame_id = params.frame_id + f.  Returns [(bytes or length, perf)] in frame order."""
        k, fr, cn = self._arrays(dev_frames, counts)
        out = (Bitstream * k)()
        rc = self.lib.pcc_pipeline_encode(self.h, fr, cn, k, stride, rgb_offset, C.byref(params), out)
        if rc != PCC_OK:
            raise PccError(rc, self.lib.pcc_pipeline_last_error(self.h).decode())
        return [((_bytes_at(b.data, b.len) if copy else b.len), [int(x) for x in b.perf]) for b in out]

    def encode_host(self, host_frames, params, stride=32, rgb_offset=16, copy=True):
        """Like encode(), with the frames in host memory (numpy arrays of POINT_DTYPE, pageable or pinned)."""
        frames = [np.ascontiguousarray(f) for f in host_frames]
        k = len(frames)
        fr = (C.c_void_p * k)(*[f.ctypes.data for f in frames])
        cn = (C.c_size_t * k)(*[len(f) for f in frames])
        out = (Bitstream * k)()
        rc = self.lib.pcc_pipeline_encode_host(self.h, fr, cn, k, stride, rgb_offset, C.byref(params), out)
        if rc != PCC_OK:
            raise PccError(rc, self.lib.pcc_pipeline_last_error(self.h).decode())
        return [((_bytes_at(b.data, b.len) if copy else b.len), [int(x) for x in b.perf]) for b in out]

    def reserve(self, n_frames, bytes_per_frame, max_points_per_frame=0):
        """Set aside the memory for the bitstreams of the coming calls and prepare every context of the ring (optional)."""
        rc = self.lib.pcc_pipeline_reserve(self.h, int(n_frames), int(bytes_per_frame), int(max_points_per_frame))
        if rc != PCC_OK:
            raise PccError(rc, "pipeline reserve")

    def gpu_stage_only(self, dev_frames, counts, params, stride=32, rgb_offset=16):
        k, fr, cn = self._arrays(dev_frames, counts)
        rc = self.lib.pcc_pipeline_gpu_stage_only(self.h, fr, cn, k, stride, rgb_offset, C.byref(params))
        if rc != PCC_OK:
            raise PccError(rc, self.lib.pcc_pipeline_last_error(self.h).decode())

    def kernel_times(self):
        """{kernel name: (total ms, launches)} and the number of profiled frames of the last call."""
        kt = KernelTimes()
        launches = (C.c_int32 * 64)()
        frames = C.c_int32()
        self.lib.pcc_pipeline_kernel_times(self.h, C.byref(kt), launches, C.byref(frames))
        return {kt.name[i].decode(): (float(kt.ms[i]), int(launches[i])) for i in range(kt.count)}, int(frames.value)

    def stats(self):
        buf = (C.c_double * 8)()
        self.lib.pcc_pipeline_stats(self.h, buf)
        keys = ("launch_us", "finish_us", "entropy_us", "occupancy_coder_us", "jpeg_us", "colour_coder_us",
                "host_stage_us", "frames")
        out = dict(zip(keys, buf))
        cpu = (C.c_double * 4)()
        self.lib.pcc_pipeline_cpu_times(self.h, cpu)
        out.update(launch_cpu_us=cpu[0], finish_cpu_us=cpu[1], entropy_cpu_us=cpu[2])
        return out


class MultiPipeline:
    """pcc_multi_pipeline: one pipeline per entry of `devices`, frame f on devices[f % len(devices)]."""

    def __init__(self, devices, workers_per_device=8):
        self.lib = load_library()
        arr = (C.c_int32 * len(devices))(*devices)
        self.h = self.lib.pcc_pipeline_create_multi(arr, len(devices), workers_per_device)
        if not self.h:
            raise RuntimeError("pcc_pipeline_create_multi(%r) failed: a device is missing -- there is no CPU fallback" % (list(devices),))
        self.devices = list(devices)

    def close(self):
        if self.h:
            self.lib.pcc_multi_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def numa_nodes(self):
        """The host NUMA node every member's threads were placed on (None: an even share of the cores instead)."""
        out = []
        for i in range(len(self.devices)):
            v = int(self.lib.pcc_pipeline_get(self.lib.pcc_multi_pipeline_member(self.h, i), b"numa_node"))
            out.append(None if v == PCC_NO_NUMA_NODE else v)
        return out

    def encode_host(self, host_frames, params, stride=32, rgb_offset=16, copy=True):
        frames = [np.ascontiguousarray(f) for f in host_frames]
        k = len(frames)
        fr = (C.c_void_p * k)(*[f.ctypes.data for f in frames])
        cn = (C.c_size_t * k)(*[len(f) for f in frames])
        out = (Bitstream * k)()
        rc = self.lib.pcc_multi_pipeline_encode_host(self.h, fr, cn, k, stride, rgb_offset, C.byref(params), out)
        if rc != PCC_OK:
            raise PccError(rc, self.lib.pcc_multi_pipeline_last_error(self.h).decode())
        return [((_bytes_at(b.data, b.len) if copy else b.len), [int(x) for x in b.perf]) for b in out]


MANUAL_CONFIGURATION = "MANUAL_CONFIGURATION"


class OctreePointCloudCodecV2:
    """Mirror of pcl::io::OctreePointCloudCodecV2<PointXYZRGB> (codec.h:70-368) over the C ABI.

    Constructor arguments are those of codec.h:108-121 in the same order.  Clouds are numpy arrays
    of POINT_DTYPE (the 32-byte pcl::PointXYZRGB layout); streams are bytes.
    """

    def __init__(self, compressionProfile_arg=MANUAL_CONFIGURATION, showStatistics_arg=False,
                 pointResolution_arg=0.001, octreeResolution_arg=0.01, doVoxelGridDownDownSampling_arg=False,
                 iFrameRate_arg=0, doColorEncoding_arg=True, colorBitResolution_arg=6, colorCodingType_arg=0,
                 doVoxelGridCentroid_arg=True, createScalableStream_arg=True, codeConnectivity_arg=False,
                 jpeg_quality_arg=75, num_threads=0, device=0):
        if compressionProfile_arg != MANUAL_CONFIGURATION:
            raise NotImplementedError("only MANUAL_CONFIGURATION is used by the reference app (eval.hpp:379)")
        if not doVoxelGridDownDownSampling_arg:
            raise NotImplementedError("the reference app always enables voxel-grid coding (eval.hpp:385)")
        if iFrameRate_arg != 0:
            raise NotImplementedError("iFrameRate is 0 in the reference: every frame is an I-frame (eval.hpp:386)")
        self._p = Params()
        self._p.octree_resolution = octreeResolution_arg
        self._p.point_resolution = pointResolution_arg
        self._p.do_color_encoding = 1 if doColorEncoding_arg else 0
        self._p.color_bit_resolution = colorBitResolution_arg
        self._p.color_coding_type = colorCodingType_arg
        self._p.do_voxel_centroid = 1 if doVoxelGridCentroid_arg else 0
        self._p.create_scalable = 1 if createScalableStream_arg else 0
        self._p.do_connectivity = 1 if codeConnectivity_arg else 0
        self._p.jpeg_quality = jpeg_quality_arg
        self._p.macroblock_size = 16          # codec.h:138
        self._p.do_icp_color_offset = 0       # codec.h:141
        self._frame_id = 0
        self._perf = [0, 0, 0]
        self._ctx = Context(device)
        self.last_hot = None

    def setMacroblockSize(self, size):       # codec.h:149
        self._p.macroblock_size = size

    def setDoICPColorOffset(self, doit):     # codec.h:164
        self._p.do_icp_color_offset = 1 if doit else 0

    def encodePointCloud(self, cloud_arg: np.ndarray) -> bytes:
        """codec.h:174-175.  Returns the bytes the reference would append to the ostream;
        an empty cloud yields b'' (frame dropped, impl.hpp:206-212)."""
        self._p.frame_id = self._frame_id + 1
        try:
            stream, perf = self._ctx.encode_intra_host(cloud_arg, self._p)
        except PccError as e:
            if e.code == -3:
                return b""
            raise
        self._frame_id += 1       # frame_ID_++ only happens for non-empty clouds (impl.hpp:133)
        self._perf = perf
        return stream

    def decodePointCloud(self, compressed_tree_data_in_arg: bytes):
        """codec.h:177-178.  Returns (cloud, bytes consumed)."""
        pts, info = self._ctx.decode_intra(compressed_tree_data_in_arg)
        return pts, info["consumed"]

    def encodePointCloudDeltaFrame(self, icloud_arg, pcloud_arg, icp_on_original=False, write_out_cloud=True):
        """codec.h:181-186.  Returns (out_cloud_arg, i_coded_data, p_coded_data)."""
        r = self._ctx.encode_delta(icloud_arg, pcloud_arg, self._p, icp_on_original, write_out_cloud)
        self._delta = r
        return r["out_cloud"], r["i_stream"], r["p_stream"]

    def decodePointCloudDeltaFrame(self, icloud_arg, i_coded_data: bytes, p_coded_data: bytes):
        """codec.h:188-191.  Returns cloud_out_arg."""
        return self._ctx.decode_delta(icloud_arg, i_coded_data, p_coded_data, self._p)

    def getMacroBlockPercentage(self):        # codec.h:199-202
        return self._delta["shared_macroblock_percentage"]

    def getMacroBlockConvergencePercentage(self):  # codec.h:204-207
        return self._delta["shared_macroblock_convergence_percentage"]

    def getPerformanceMetrics(self):          # codec.h:193-197
        return list(self._perf)

    def getOutputCloud(self):                 # used at eval.hpp:862
        return self._ctx.output_cloud()


# ---- host building blocks (no GPU needed) ----

def host_range_encode(data: bytes) -> bytes:
    lib = load_library()
    src = np.frombuffer(data, dtype=np.uint8)
    out = np.zeros(len(src) * 2 + 2048, dtype=np.uint8)
    n = lib.pcc_host_range_encode(src.ctypes.data if len(src) else None, len(src), out.ctypes.data, len(out))
    return out[:n].tobytes()


def host_range_encode_many(vectors) -> list:
    """Up to four vectors coded in one loop (pcc_host_range_encode_many): the bytes of each are those of host_range_encode."""
    lib = load_library()
    k = len(vectors)
    srcs = [np.frombuffer(v, dtype=np.uint8) for v in vectors]
    outs = [np.zeros(len(s) * 2 + 2048, dtype=np.uint8) for s in srcs]
    inp = (C.c_void_p * k)(*[s.ctypes.data if len(s) else None for s in srcs])
    n = (C.c_size_t * k)(*[len(s) for s in srcs])
    outp = (C.c_void_p * k)(*[o.ctypes.data for o in outs])
    cap = (C.c_size_t * k)(*[len(o) for o in outs])
    got = (C.c_size_t * k)()
    rc = lib.pcc_host_range_encode_many(k, inp, n, outp, cap, got)
    if rc != 0:
        raise PccError(rc, "pcc_host_range_encode_many")
    return [outs[i][:got[i]].tobytes() for i in range(k)]


def host_range_decode(stream: bytes, n: int):
    lib = load_library()
    src = np.frombuffer(stream, dtype=np.uint8)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    used = lib.pcc_host_range_decode(src.ctypes.data, len(src), out.ctypes.data, n)
    return out[:n].tobytes(), used


def host_jpeg_encode(rgb: np.ndarray, quality: int) -> bytes:
    lib = load_library()
    rgb = np.ascontiguousarray(rgb, dtype=np.uint8)
    h, w, _ = rgb.shape
    out = np.zeros(rgb.size * 2 + 4096, dtype=np.uint8)
    n = lib.pcc_host_jpeg_encode(rgb.ctypes.data, w, h, quality, out.ctypes.data, len(out))
    assert n > 0
    return out[:n].tobytes()


def host_jpeg_decode(jpg: bytes, max_pixels=1 << 24) -> np.ndarray:
    lib = load_library()
    src = np.frombuffer(jpg, dtype=np.uint8)
    out = np.zeros(3 * max_pixels, dtype=np.uint8)
    w, h = C.c_int(), C.c_int()
    rc = lib.pcc_host_jpeg_decode(src.ctypes.data, len(src), out.ctypes.data, len(out), C.byref(w), C.byref(h))
    if rc != PCC_OK:
        raise PccError(rc, "jpeg decode")
    return out[: 3 * w.value * h.value].reshape(h.value, w.value, 3).copy()


def host_rigid_compress(tr: np.ndarray):
    """RigidTransformCoding::compressRigidTransform: 4x4 float -> list of int16."""
    lib = load_library()
    m = np.ascontiguousarray(tr, dtype=np.float32).reshape(16)
    out = np.zeros(16, dtype=np.int16)
    n = lib.pcc_host_rigid_compress(m.ctypes.data, out.ctypes.data, len(out))
    return out[:n].tolist()


def host_rigid_decompress(comp):
    lib = load_library()
    c = np.ascontiguousarray(comp, dtype=np.int16)
    out = np.zeros(16, dtype=np.float32)
    rc = lib.pcc_host_rigid_decompress(c.ctypes.data, len(c), out.ctypes.data)
    if rc != PCC_OK:
        raise PccError(rc, "rigid transform decompress")
    return out.reshape(4, 4)


def host_snake_perm(w, h):
    lib = load_library()
    return np.array([lib.pcc_host_snake_position(i, w, h) for i in range(w * h)], dtype=np.int32)
