"""ctypes binding of the CPU oracle (oracle/libcvgs_oracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (cvgpuspeedup_amd/) never does.  The oracle consumes the same POD chain descriptor
as the product's C-ABI, with host pointers.
"""
import ctypes as C
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(_HERE))
from cvgpuspeedup_amd import capi  # noqa: E402  (struct definitions only)

LIB_PATH = os.path.join(_HERE, "libcvgs_oracle.so")


def build_oracle():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)
    return LIB_PATH


class Geom(C.Structure):
    _fields_ = [("fx", C.c_float), ("fy", C.c_float), ("x1", C.c_int32), ("y1", C.c_int32), ("x2", C.c_int32),
                ("y2", C.c_int32)]


_lib = None


def load_oracle():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build_oracle()
    lib = C.CDLL(LIB_PATH)
    lib.oracle_execute.restype = C.c_int
    lib.oracle_execute.argtypes = [C.POINTER(capi.ChainDesc)]
    lib.oracle_resize_geometry.restype = None
    lib.oracle_resize_geometry.argtypes = [C.c_int32] * 5 + [C.POINTER(Geom)]
    lib.oracle_set_threads.argtypes = [C.c_int]
    lib.oracle_get_threads.restype = C.c_int
    lib.oracle_max_threads.restype = C.c_int
    lib.oracle_fastdiv_mismatches.restype = C.c_int64
    lib.oracle_fastdiv_mismatches.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_uint64, C.c_int32]
    lib.oracle_resize_tapped_bytes.restype = C.c_int64
    lib.oracle_resize_tapped_bytes.argtypes = [C.c_int32] * 6
    lib.oracle_circular_create.restype = C.c_int
    lib.oracle_circular_create.argtypes = [C.POINTER(C.c_void_p)] + [C.c_int32] * 7
    lib.oracle_circular_update.restype = C.c_int
    lib.oracle_circular_update.argtypes = [C.c_void_p, C.POINTER(capi.ChainDesc)]
    lib.oracle_circular_data.restype = C.c_void_p
    lib.oracle_circular_data.argtypes = [C.c_void_p]
    lib.oracle_circular_bytes.restype = C.c_size_t
    lib.oracle_circular_bytes.argtypes = [C.c_void_p]
    lib.oracle_circular_destroy.argtypes = [C.c_void_p]
    _lib = lib
    return lib


def execute(lowered):
    """Run a lowered chain (cvgpuspeedup_amd.cvgs.lower) whose pointers are HOST pointers."""
    rc = load_oracle().oracle_execute(C.byref(lowered.desc))
    if rc != 0:
        raise RuntimeError("oracle_execute failed: %d" % rc)


def execute_k1_fast(lowered, reps=1):
    """The headline chain as a plain loop nest (oracle_k1_fast_repeat): bench.py's timed CPU baseline; `reps` passes over
    the batch inside one parallel region."""
    lib = load_oracle()
    lib.oracle_k1_fast_repeat.restype = C.c_int
    lib.oracle_k1_fast_repeat.argtypes = [C.POINTER(capi.ChainDesc), C.c_int]
    rc = lib.oracle_k1_fast_repeat(C.byref(lowered.desc), int(reps))
    if rc != 0:
        raise RuntimeError("oracle_k1_fast failed: %d" % rc)


def resize_geometry(sw, sh, dw, dh, ar):
    g = Geom()
    load_oracle().oracle_resize_geometry(sw, sh, dw, dh, ar, C.byref(g))
    return g


def tapped_bytes(sw, sh, dw, dh, ar, bpp):
    return int(load_oracle().oracle_resize_tapped_bytes(sw, sh, dw, dh, ar, bpp))


class OracleCircular:
    def __init__(self, width, height, elem_type, color_planes, batch, order, cp_mode):
        self.lib = load_oracle()
        self.h = C.c_void_p(0)
        rc = self.lib.oracle_circular_create(C.byref(self.h), width, height, elem_type, color_planes, batch, order,
                                             cp_mode)
        assert rc == 0

    def update(self, lowered):
        rc = self.lib.oracle_circular_update(self.h, C.byref(lowered.desc))
        if rc != 0:
            raise RuntimeError("oracle_circular_update failed: %d" % rc)

    def array(self, dtype):
        import numpy as np
        n = self.lib.oracle_circular_bytes(self.h)
        buf = (C.c_uint8 * n).from_address(self.lib.oracle_circular_data(self.h))
        return np.frombuffer(buf, dtype=dtype).copy()

    def __del__(self):
        try:
            self.lib.oracle_circular_destroy(self.h)
        except Exception:
            pass
