"""CV_16F (the half-precision hand-off option) on the CPU oracle: the reference has no half type,
so the pin is IEEE 754 itself -- numpy's float16 conversions (round to nearest even, overflow to inf, subnormals)."""
import ctypes as C

import numpy as np

from cvgpuspeedup_amd import capi, cvgs
from oracle import oracle_binding as ob


def _interesting_floats():
    allh = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    s = np.unique(allh[np.isfinite(allh)])
    mid = ((s[:-1].astype(np.float64) + s[1:].astype(np.float64)) / 2).astype(np.float32)  # exact ties
    rng = np.random.default_rng(7)
    rnd_bits = rng.integers(0, 2 ** 32, 100000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    rnd_bits = rnd_bits[np.isfinite(rnd_bits)]
    edge = np.array([65504, 65519.996, 65520, 65536, 1e10, -65520, np.inf, -np.inf, 2.0 ** -25, 2.0 ** -24, 6e-8, 0.0, -0.0],
                    np.float32)
    return np.concatenate([s, mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)),
                           rng.standard_normal(100000).astype(np.float32) * 300, rnd_bits, edge])


def test_float_to_half_chain_matches_ieee(oracle):
    """32F image -> convertTo<CV_32FC1, CV_16FC1> -> write: every half, every tie, neighbours, random bit patterns."""
    x = _interesting_floats()
    n = (x.size // 64) * 64
    src = x[:n].reshape(-1, 64, 1).copy()
    out = np.zeros(src.shape, np.float16)
    iops = [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [cvgs.GpuMat.from_array(src, cvgs.CV_32FC1)], 1),
            cvgs.convertTo(cvgs.CV_32FC1, cvgs.CV_16FC1), cvgs.write(cvgs.CV_16FC1, cvgs.GpuMat.from_array(out, cvgs.CV_16FC1))]
    ob.execute(cvgs.lower(iops))
    with np.errstate(over="ignore"):
        want = src.astype(np.float16)
    assert np.array_equal(out.view(np.uint16), want.view(np.uint16))


def test_half_source_reads_exactly(oracle):
    """16F source -> convertTo 32F: all 63488 finite halves and both infinities convert exactly."""
    bits = np.arange(65536, dtype=np.uint16)
    h = bits.view(np.float16)
    keep = ~np.isnan(h)
    src = np.resize(h[keep], (63490 // 10, 10, 1)).copy()
    out = np.zeros(src.shape, np.float32)
    iops = [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_16FC1, [cvgs.GpuMat.from_array(src, cvgs.CV_16FC1)], 1),
            cvgs.convertTo(cvgs.CV_16FC1, cvgs.CV_32FC1), cvgs.write(cvgs.CV_32FC1, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1))]
    ob.execute(cvgs.lower(iops))
    assert np.array_equal(out.view(np.uint32), src.astype(np.float32).view(np.uint32))


def test_scaled_convert_rounds_once(oracle):
    """convertTo<CV_8UC3, CV_16FC3>(alpha, beta) computes in fp32 and rounds to half once at the end."""
    from tests import helpers as H
    src = H.random_u8((33, 47, 3), 5)
    out = np.zeros((33, 47, 3), np.float16)
    iops = [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(src, cvgs.CV_8UC3)], 1),
            cvgs.convertTo(cvgs.CV_8UC3, cvgs.CV_16FC3, 1.0 / 255.0, -0.4321),
            cvgs.write(cvgs.CV_16FC3, cvgs.GpuMat.from_array(out, cvgs.CV_16FC3))]
    ob.execute(cvgs.lower(iops))
    want = (src.astype(np.float32) * np.float32(1.0 / 255.0) + np.float32(-0.4321)).astype(np.float16)
    assert np.array_equal(out.view(np.uint16), want.view(np.uint16))


def test_scalar_entry_points(oracle):
    lib = C.CDLL(ob.LIB_PATH)
    lib.oracle_float_to_half.argtypes = [C.c_float]
    lib.oracle_float_to_half.restype = C.c_uint16
    lib.oracle_half_to_float.argtypes = [C.c_uint16]
    lib.oracle_half_to_float.restype = C.c_float
    assert lib.oracle_float_to_half(1.0) == 0x3c00 and lib.oracle_float_to_half(-2.0) == 0xc000
    assert lib.oracle_float_to_half(65520.0) == 0x7c00 and lib.oracle_float_to_half(65519.0) == 0x7bff
    assert lib.oracle_half_to_float(0x0001) == 2.0 ** -24
