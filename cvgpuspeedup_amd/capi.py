"""ctypes mirror of include/cvgs_hip.h (the C-ABI of libcvgs_hip.so).

Plumbing only: the structs below are field-for-field the C structs, and the product path is the
HIP library.  There is no Python/CPU fallback: if the library is missing or a call fails, an
exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcvgs_hip.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

MAX_OPS = 12
KERNARG_PLANES = 64
MAX_MIRRORS = 7
MAX_CHAINS = 128

# status
OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_RCCL = 0, -1, -2, -3, -4, -5

# depths / types (OpenCV encoding)
CIRCULAR_MIRRORED = 1
CIRCULAR_CAPTURABLE = 2
DEPTH_8U, DEPTH_8S, DEPTH_16U, DEPTH_16S, DEPTH_32S, DEPTH_32F, DEPTH_64F, DEPTH_16F = range(8)


def make_type(depth, cn):
    return depth + ((cn - 1) << 3)


def type_depth(t):
    return t & 7


def type_cn(t):
    return ((t >> 3) & 63) + 1


# read kinds
READ_PIXEL, READ_RESIZE_LINEAR, READ_NV12, READ_NV12_RESIZE_LINEAR, READ_WARP_AFFINE, READ_WARP_PERSPECTIVE = range(6)
# aspect ratio (same values as cvGS::AspectRatio)
PRESERVE_AR, IGNORE_AR, PRESERVE_AR_RN_EVEN, PRESERVE_AR_LEFT = 0, 1, 2, 3
YUV_FULL, YUV_LIMITED = 0, 1
YUV_NV12, YUV_NV21, YUV_I420, YUV_YV12, YUV_P010 = 0, 1, 2, 3, 4
BT601, BT709, BT2020 = 0, 1, 2
READ_FLAG_TABLE_ON_DEVICE = 1
READ_FLAG_TABLE_SOURCES_VOUCHED = 2
# opcodes
(OP_NOP, OP_CAST, OP_MUL, OP_ADD, OP_SUB, OP_DIV, OP_REORDER, OP_ADD_ALPHA, OP_DROP_ALPHA, OP_GRAY,
 OP_CAST_TRUNC) = range(11)
# write kinds
(WRITE_PIXEL_2D, WRITE_PIXEL_3D, WRITE_TENSOR_SPLIT, WRITE_TENSOR_T_SPLIT, WRITE_SPLIT_2D,
 WRITE_PIXEL_2D_BATCH) = range(6)
# chain flags
CHAIN_DEFAULT, CHAIN_FORCE_GENERIC, CHAIN_NO_THREAD_FUSION = 0, 1, 2
# circular tensor
NEWEST_FIRST, OLDEST_FIRST = 0, 1
PLANES_STANDARD, PLANES_TRANSPOSED = 0, 1


class Image2D(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_int32), ("height", C.c_int32), ("step", C.c_int32),
                ("uv_offset", C.c_int32)]


class ReadDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("src_type", C.c_int32), ("batch", C.c_int32), ("used_planes", C.c_int32),
                ("src", C.c_void_p), ("dst_width", C.c_int32), ("dst_height", C.c_int32),
                ("aspect_ratio", C.c_int32), ("flags", C.c_uint32), ("background", C.c_float * 4),
                ("yuv_range", C.c_int32), ("yuv_primaries", C.c_int32), ("yuv_alpha", C.c_int32),
                ("yuv_layout", C.c_int32), ("warp_matrices", C.POINTER(C.c_float)),
                ("warp_dst_sizes", C.POINTER(C.c_int32)), ("table_src_lo", C.c_void_p), ("table_src_hi", C.c_void_p)]


class Op(C.Structure):
    _fields_ = [("opcode", C.c_int32), ("aux", C.c_int32), ("operand", C.c_float * 4), ("operand_d", C.c_double * 4)]


class WriteDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dst_type", C.c_int32), ("data", C.c_void_p), ("width", C.c_int32),
                ("height", C.c_int32), ("step", C.c_int32), ("planes", C.c_int32), ("planes2d", C.c_void_p),
                ("mirrors", C.POINTER(C.c_void_p)), ("n_mirrors", C.c_int32), ("reserved", C.c_int32)]


class ChainDesc(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("flags", C.c_uint32), ("read", ReadDesc), ("n_ops", C.c_int32),
                ("reserved", C.c_int32), ("ops", Op * MAX_OPS), ("write", WriteDesc)]


# every symbol include/cvgs_hip.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("cvgs_abi_version", C.c_int, []),
    ("cvgs_version_string", C.c_char_p, []),
    ("cvgs_last_error", C.c_char_p, []),
    ("cvgs_device_count", C.c_int, []),
    ("cvgs_execute", C.c_int, [C.POINTER(ChainDesc), C.c_void_p]),
    ("cvgs_execute_many", C.c_int, [C.POINTER(ChainDesc), C.c_int32, C.c_void_p]),
    ("cvgs_stream_release", C.c_int, [C.c_void_p]),
    ("cvgs_validate", C.c_int, [C.POINTER(ChainDesc)]),
    ("cvgs_kernel_name", C.c_int, [C.POINTER(ChainDesc), C.c_char_p, C.c_size_t]),
    ("cvgs_plane_table_bytes", C.c_size_t, [C.c_int32]),
    ("cvgs_plane_table_build", C.c_int, [C.POINTER(ReadDesc), C.c_void_p]),
    ("cvgs_plane_table_hull", C.c_int, [C.POINTER(ReadDesc), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    ("cvgs_circular_create", C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int32] * 8),
    ("cvgs_circular_create_ex", C.c_int, [C.POINTER(C.c_void_p)] + [C.c_int32] * 8 + [C.c_uint32]),
    ("cvgs_circular_update", C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p]),
    ("cvgs_circular_data", C.c_void_p, [C.c_void_p]),
    ("cvgs_circular_bytes", C.c_size_t, [C.c_void_p]),
    ("cvgs_circular_updates", C.c_int64, [C.c_void_p]),
    ("cvgs_circular_destroy", C.c_int, [C.c_void_p]),
    ("cvgs_stream_copy", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    ("cvgs_exchange_signal", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.c_void_p, C.c_void_p]),
    ("cvgs_exchange_wait", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_uint64, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]),
    ("cvgs_exchange_step", C.c_int, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_void_p, C.c_uint64, C.c_double, C.c_void_p, C.c_void_p]),
    ("cvgs_queue_create", C.c_int, [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.c_double, C.c_uint32]),
    ("cvgs_queue_submit", C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.POINTER(C.c_uint64)]),
    ("cvgs_queue_submit_many", C.c_int, [C.c_void_p, C.POINTER(C.POINTER(ChainDesc)), C.c_int32, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_submit_on", C.c_int, [C.c_void_p, C.POINTER(ChainDesc), C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_submit_many_on", C.c_int, [C.c_void_p, C.POINTER(C.POINTER(ChainDesc)), C.c_int32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_recover", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_wait", C.c_int, [C.c_void_p, C.c_uint64, C.c_double]),
    ("cvgs_queue_stream_wait", C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    ("cvgs_queue_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_stream", C.c_void_p, [C.c_void_p]),
    ("cvgs_queue_profile", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    ("cvgs_queue_destroy", C.c_int, [C.c_void_p]),
    ("cvgs_range_push", None, [C.c_char_p]),
    ("cvgs_range_pop", None, []),
]


class CvgsError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("cvgs error %d: %s" % (code, message))
        self.code = code


def build_library(force=False):
    """Compile the HIP kernels + C-ABI for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    if force:
        subprocess.run(["make", "-C", CSRC_DIR, "clean"], check=True, stdout=subprocess.DEVNULL)
    subprocess.run(["make", "-C", CSRC_DIR, "-j8"], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None


def load_library():
    """Load libcvgs_hip.so.  Fails loudly if it has not been built: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libcvgs_hip.so is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C cvgpuspeedup_amd/csrc` (the HIP extension is required, no fallback)")
    try:
        # torch bundles its own libamdhip64.so.7: importing it first makes the dynamic linker resolve our
        # DT_NEEDED libamdhip64.so.7 to the SAME runtime, so torch streams/pointers are valid in our calls.
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.cvgs_abi_version() != 6:
        raise ImportError("libcvgs_hip.so ABI version mismatch")
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise CvgsError(rc, load_library().cvgs_last_error().decode())


def new_chain():
    ch = ChainDesc()
    ch.struct_size = C.sizeof(ChainDesc)
    return ch
