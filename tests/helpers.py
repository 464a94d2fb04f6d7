"""Shared test helpers: the deterministic input generator of BASELINE.md (splitmix64, seed 0xC0FFEE),
the configs restated as concrete inputs, and exact comparison utilities."""
import numpy as np

from cvgpuspeedup_amd import cvgs

from cvgpuspeedup_amd.workloads import (K1_ALPHA, K1_DIV, K1_SUB, SEED, fixed_crops, k1_chain, random_crops,  # noqa: F401
                                        random_u8, random_u16, splitmix64_array)


def f32_list(vals):
    """cvScalar2CUDAV: double -> float narrowing, then back to python floats (exact)."""
    return [float(np.float32(v)) for v in vals]


def ulp_diff(a, b):
    """Elementwise distance in units in the last place between two float32 arrays."""
    ai = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-(2 ** 31)) - ai, ai)
    bi = np.where(bi < 0, np.int64(-(2 ** 31)) - bi, bi)
    return np.abs(ai - bi)


def assert_bit_exact(gpu, ref, what=""):
    gpu = np.ascontiguousarray(gpu)
    ref = np.ascontiguousarray(ref)
    assert gpu.shape == ref.shape and gpu.dtype == ref.dtype, (gpu.shape, ref.shape, gpu.dtype, ref.dtype)
    same = gpu.view(np.uint8) == ref.view(np.uint8)
    if not same.all():
        bad = np.argwhere(gpu != ref)
        first = tuple(bad[0]) if len(bad) else None
        extra = ""
        if gpu.dtype == np.float32:
            extra = " max_ulp=%d" % int(ulp_diff(gpu, ref).max())
        raise AssertionError("%s: %d elements differ (first at %s: gpu=%r ref=%r)%s" % (
            what, int((~same).reshape(gpu.size, -1).any(axis=1).sum()), first,
            gpu[first] if first else None, ref[first] if first else None, extra))


_testaid = None


def testaid():
    """tests/aids/libcvgs_testaid.so: cvgs_debug_occupy / cvgs_debug_poll -- test and measurement aids (a stand-in for a foreign kernel, a
    producer kernel on a stream), NOT part of the product library (round 6).  Built by `make -C tests/aids` (__graft_entry__.build())."""
    global _testaid
    if _testaid is None:
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "aids", "libcvgs_testaid.so")
        if not os.path.exists(path):
            raise ImportError("tests/aids/libcvgs_testaid.so is missing: run `make -C tests/aids` (or __graft_entry__.build())")
        import torch  # noqa: F401  (the same HIP runtime as the product library: see capi.load_library)
        lib = C.CDLL(path)
        lib.cvgs_debug_occupy.restype = C.c_int
        lib.cvgs_debug_occupy.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_double, C.c_void_p]
        lib.cvgs_debug_poll.restype = C.c_int
        lib.cvgs_debug_poll.argtypes = [C.c_void_p, C.c_double, C.c_int32, C.c_void_p]
        _testaid = lib
    return _testaid


def aid_check(rc):
    if rc != 0:
        raise RuntimeError("test aid call failed: %d" % rc)
