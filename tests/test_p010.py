"""P010 (the 10-bit decoder surface: NV12's geometry, 16-bit samples, code = sample >> 6) and the BT.2020 matrix behind the
NV12 read kinds (cvgs_read_desc.yuv_layout = CVGS_YUV_P010, yuv_primaries = CVGS_BT2020).  The reference spells the reader as
a template on the pixel format (fk::ReadYUV<PF>, tests/resize/test_fused_resize.cu:50) and instantiates NV12 only, so nothing
in the reference pins these values; the pins here are independent of the oracle's author:
  * the conversion is restated in float64 from the standards' luma weights (Kr / Kb) and the 10-bit range scaling
    (limited: (Y - 64) * 1023/876, C * 1023/896) and compared with the oracle on random surfaces;
  * FULL range on codes that are 4 x an 8-bit picture must give exactly 4 x the NV12 result (a power-of-two scaling is exact
    in binary floating point), which ties P010 to the NV12 path the Kr/Kb probes of test_independent_pins.py pin;
  * the 6 low bits of a sample do not exist: any value there leaves the result unchanged.
The GPU is compared with the oracle bit for bit below (pixel reads, resizes, crops of a surface, integer and float outputs)."""
import ctypes as C

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

STANDARDS = {capi.BT601: (0.299, 0.114), capi.BT709: (0.2126, 0.0722), capi.BT2020: (0.2627, 0.0593)}
RANGES = [capi.YUV_FULL, capi.YUV_LIMITED]


def p010_surface(w, h, seed, low_bits=False, codes=None):
    """(H*3/2, W) u16 surface: luma rows, then interleaved (U,V) rows; 10-bit codes in the high bits."""
    if codes is None:
        y = H.random_u16((h, w), seed) >> 6
        u = H.random_u16((h // 2, w // 2), seed + 1) >> 6
        v = H.random_u16((h // 2, w // 2), seed + 2) >> 6
    else:
        y, u, v = codes
    s = np.zeros((h + h // 2, w), np.uint16)
    s[:h] = y.astype(np.uint16) << 6
    s[h:, 0::2] = u.astype(np.uint16) << 6
    s[h:, 1::2] = v.astype(np.uint16) << 6
    if low_bits:
        s |= (H.random_u16(s.shape, seed + 3) & 63).astype(np.uint16)
    return s, (y, u, v)


def derived_rgb(y, u, v, range_, primaries, bits):
    """float64 restatement from Kr / Kb, codes of `bits` bits: (H, W, 3)."""
    kr, kb = STANDARDS[primaries]
    kg = 1.0 - kr - kb
    full = float((1 << bits) - 1)
    unit = float(1 << (bits - 8))
    Y = y.astype(np.float64)
    cb = np.repeat(np.repeat(u.astype(np.float64), 2, 0), 2, 1) - 128.0 * unit
    cr = np.repeat(np.repeat(v.astype(np.float64), 2, 0), 2, 1) - 128.0 * unit
    if range_ == capi.YUV_LIMITED:
        Y = (Y - 16.0 * unit) * (full / (219.0 * unit))
        cb = cb * (full / (224.0 * unit))
        cr = cr * (full / (224.0 * unit))
    r = Y + 2.0 * (1.0 - kr) * cr
    b = Y + 2.0 * (1.0 - kb) * cb
    g = Y - 2.0 * kb * (1.0 - kb) / kg * cb - 2.0 * kr * (1.0 - kr) / kg * cr
    return np.stack([r, g, b], axis=-1)


def chain(wrap, surf, w, h, dst, out, out_type, rng, prim, alpha=False, tail=()):
    m = wrap(surf)
    luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, m.data, m.step, owner=m.owner)
    return [cvgs.read_nv12(luma, dst, rng, prim, alpha, layout=capi.YUV_P010), *tail, cvgs.write(out_type, out)]


def run_oracle(oracle, surf, w, h, dst, rng, prim, alpha=False):
    ow, oh = dst or (w, h)
    cn = 4 if alpha else 3
    out = np.zeros((oh, ow, cn), np.float32)
    t = cvgs.CV_32FC4 if alpha else cvgs.CV_32FC3
    oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_16UC1), surf, w, h, dst, cvgs.GpuMat.from_array(out, t), t,
                                    rng, prim, alpha)))
    return out


@pytest.mark.parametrize("range_", RANGES)
@pytest.mark.parametrize("primaries", sorted(STANDARDS))
def test_oracle_p010_equals_the_derivation_from_kr_kb(oracle, range_, primaries):
    w, h = 64, 48
    surf, (y, u, v) = p010_surface(w, h, 70 + primaries)
    got = run_oracle(oracle, surf, w, h, None, range_, primaries, alpha=True)
    want = derived_rgb(y, u, v, range_, primaries, 10)
    # literals carry 6 decimals (5e-7 x a chroma excursion of 512 x 1023/896 = 3e-4) + a few float32 roundings at 2^11 (1.2e-4 each)
    assert np.abs(got[..., :3].astype(np.float64) - want).max() < 1.2e-3
    assert (got[..., 3] == 1023.0).all()
    # and the matrix must be THIS standard's: another standard differs by far more than the tolerance
    other = derived_rgb(y, u, v, range_, (primaries + 1) % 3, 10)
    assert np.abs(got[..., :3].astype(np.float64) - other).max() > 10.0


@pytest.mark.parametrize("range_", RANGES)
def test_oracle_bt2020_on_8_bit_surfaces(oracle, range_):
    """CVGS_BT2020 with NV12: the same derivation on 8-bit codes."""
    w, h = 64, 32
    y, u, v = H.random_u8((h, w), 5), H.random_u8((h // 2, w // 2), 6), H.random_u8((h // 2, w // 2), 7)
    s = np.zeros((h + h // 2, w), np.uint8)
    s[:h], s[h:, 0::2], s[h:, 1::2] = y, u, v
    out = np.zeros((h, w, 3), np.float32)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, s.ctypes.data, w, owner=s)
    oracle.execute(cvgs.lower([cvgs.read_nv12(luma, None, range_, capi.BT2020, False), cvgs.write(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC3))]))
    want = derived_rgb(y, u, v, range_, capi.BT2020, 8)
    assert np.abs(out.astype(np.float64) - want).max() < 3e-4


@pytest.mark.parametrize("primaries", sorted(STANDARDS))
def test_full_range_p010_is_exactly_four_times_nv12(oracle, primaries):
    w, h = 96, 64
    y, u, v = H.random_u8((h, w), 11), H.random_u8((h // 2, w // 2), 12), H.random_u8((h // 2, w // 2), 13)
    s8 = np.zeros((h + h // 2, w), np.uint8)
    s8[:h], s8[h:, 0::2], s8[h:, 1::2] = y, u, v
    s10, _ = p010_surface(w, h, 0, codes=(y.astype(np.uint16) * 4, u.astype(np.uint16) * 4, v.astype(np.uint16) * 4))
    for dst in (None, (40, 24)):
        ow, oh = dst or (w, h)
        o8 = np.zeros((oh, ow, 3), np.float32)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, s8.ctypes.data, w, owner=s8)
        oracle.execute(cvgs.lower([cvgs.read_nv12(luma, dst, capi.YUV_FULL, primaries, False), cvgs.write(cvgs.CV_32FC3, cvgs.GpuMat.from_array(o8, cvgs.CV_32FC3))]))
        o10 = run_oracle(oracle, s10, w, h, dst, capi.YUV_FULL, primaries)
        assert o8.any()
        H.assert_bit_exact(o10, o8 * np.float32(4.0), "P010 vs 4 x NV12, dst %s" % (dst,))


def test_low_bits_of_a_sample_are_ignored(oracle):
    w, h = 64, 32
    clean, codes = p010_surface(w, h, 21)
    dirty, _ = p010_surface(w, h, 21, low_bits=True, codes=codes)
    assert (clean != dirty).any()
    for dst in (None, (30, 20)):
        H.assert_bit_exact(run_oracle(oracle, dirty, w, h, dst, capi.YUV_LIMITED, capi.BT2020),
                           run_oracle(oracle, clean, w, h, dst, capi.YUV_LIMITED, capi.BT2020), "low bits")


def test_p010_validation(lib):
    w, h = 32, 16
    surf, _ = p010_surface(w, h, 3)
    out = np.zeros((h, w, 3), np.float32)
    wrap = lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_16UC1)
    mk = lambda: cvgs.lower(chain(wrap, surf, w, h, None, cvgs.GpuMat.from_array(out, cvgs.CV_32FC3), cvgs.CV_32FC3, capi.YUV_FULL, capi.BT2020))
    ch = mk()
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
    ch.desc.read.src_type = cvgs.CV_8UC1          # P010 samples are 16-bit
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = mk()
    ch.desc.read.yuv_layout = capi.YUV_NV12       # ... and NV12's are 8-bit
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = mk()
    ch.desc.read.yuv_primaries = 3
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = mk()
    ch.desc.read.yuv_range = 2
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = mk()
    ch.desc.read.yuv_layout = 5
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    ch = mk()
    C.cast(ch.desc.read.src, C.POINTER(capi.Image2D))[0].step = 2 * w + 1  # rows of 16-bit samples cannot start on odd bytes
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID


# ---- GPU vs oracle -------------------------------------------------------------------------------------------------------
def _gpu_vs_oracle(oracle, surf, w, h, dst, rng, prim, alpha, tail, out_type, np_dt, what):
    import torch
    dev = torch.device("cuda:0")
    ow, oh = dst or (w, h)
    cn = cvgs.type_cn(out_type)
    st = torch.from_numpy(surf.view(np.int16)).to(dev)
    t_dt = {np.float32: torch.float32, np.uint16: torch.int16, np.uint8: torch.uint8, np.float16: torch.float16}[np_dt]
    gt = torch.zeros((oh, ow, cn), dtype=t_dt, device=dev)
    ref = np.zeros((oh, ow, cn), np_dt)
    ops = chain(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_16UC1), surf, w, h, dst, cvgs.GpuMat.from_tensor(gt, out_type), out_type, rng, prim, alpha, tail)
    if dst is not None:  # the fused NV12-resize kernel serves P010 too (16-bit taps)
        assert cvgs.kernel_name(*ops).startswith("k4_nv12_resize"), cvgs.kernel_name(*ops)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_16UC1), surf, w, h, dst, cvgs.GpuMat.from_array(ref, out_type), out_type,
                                    rng, prim, alpha, tail)))
    assert ref.any()
    got = gt.cpu().numpy()
    H.assert_bit_exact(got.view(np_dt) if got.dtype != np_dt else got, ref, "%s via %s" % (what, cvgs.kernel_name(*ops)))


@pytest.mark.gpu
@pytest.mark.parametrize("range_", RANGES)
@pytest.mark.parametrize("primaries", sorted(STANDARDS))
@pytest.mark.parametrize("dst", [None, (213, 120), (64, 128), (1280, 720)])
def test_gpu_p010_matches_the_oracle(oracle, range_, primaries, dst):
    w, h = 640, 360
    surf, _ = p010_surface(w, h, 90, low_bits=True)
    f = cvgs.CV_32FC3
    _gpu_vs_oracle(oracle, surf, w, h, dst, range_, primaries, False, [cvgs.multiply(f, [1 / 1023.0] * 3)], f, np.float32,
                   "P010 range %d primaries %d dst %s" % (range_, primaries, dst))


@pytest.mark.gpu
@pytest.mark.parametrize("dst", [None, (100, 80)])
def test_gpu_p010_to_10_bit_rgba_image(oracle, dst):
    """-> ushort4 (saturating at 65535, not at 1023: the cast is the chain's CV_16U saturate_cast), and -> uchar3 after x 255/1023."""
    w, h = 256, 128
    surf, _ = p010_surface(w, h, 91)
    f4, u4 = cvgs.CV_32FC4, cvgs.CV_16UC4
    _gpu_vs_oracle(oracle, surf, w, h, dst, capi.YUV_LIMITED, capi.BT2020, True, [cvgs.convertTo(f4, u4)], u4, np.uint16, "ushort4")
    f3, b3 = cvgs.CV_32FC3, cvgs.CV_8UC3
    _gpu_vs_oracle(oracle, surf, w, h, dst, capi.YUV_LIMITED, capi.BT709, False, [cvgs.convertTo(f3, b3, 255.0 / 1023.0)], b3, np.uint8, "uchar3")


@pytest.mark.gpu
def test_gpu_p010_crops_of_a_surface_in_one_launch(oracle):
    import torch
    dev = torch.device("cuda:0")
    w, h = 1920, 1080
    surf, _ = p010_surface(w, h, 92)
    st = torch.from_numpy(surf.view(np.int16)).to(dev)
    rects = [(0, 0, 64, 64), (100, 200, 300, 400), (1800, 1000, 120, 80), (2, 2, 2, 2), (960, 540, 640, 360)]
    dst = (64, 128)
    f = cvgs.CV_32FC3

    def ops(wrap, out):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, m.data, m.step, owner=m.owner)
        crops = [luma.nv12_roi(*r) for r in rects]
        return [cvgs.read_nv12(crops, dst, capi.YUV_LIMITED, capi.BT2020, False, layout=capi.YUV_P010), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                cvgs.multiply(f, [1 / 1023.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                cvgs.split(f, out, dst)]

    gt = torch.zeros((len(rects), 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    ref = np.zeros((len(rects), 3 * dst[0] * dst[1]), np.float32)
    chain_gpu = ops(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_16UC1), cvgs.GpuMat.from_tensor(gt, cvgs.CV_32FC1))
    cvgs.executeOperations(torch.cuda.current_stream(), *chain_gpu)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(ops(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_16UC1), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    H.assert_bit_exact(gt.cpu().numpy(), ref, "P010 crops via %s" % cvgs.kernel_name(*chain_gpu))


@pytest.mark.gpu
@pytest.mark.parametrize("dst", [(213, 120), (800, 450)])
def test_gpu_p010_canonical_arithmetic_program(oracle, dst):
    """10-bit surface -> resize -> a chain that is not K4's compile-time program (scale, subtract, divide, add) -> planar tensor: the canonical
    arithmetic program (k_taps.hpp: K1CanonProg) on 16-bit taps, bit for bit against the oracle."""
    import torch
    dev = torch.device("cuda:0")
    w, h = 640, 360
    surf, _ = p010_surface(w, h, 95, low_bits=True)
    f = cvgs.CV_32FC3

    def build(wrap, out):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, m.data, m.step, owner=m.owner)
        return [cvgs.read_nv12(luma, dst, capi.YUV_LIMITED, capi.BT2020, False, layout=capi.YUV_P010), cvgs.multiply(f, [1 / 1023.0] * 3),
                cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]), cvgs.add(f, [0.5, 0.25, 0.125]), cvgs.split(f, out, dst)]

    st = torch.from_numpy(surf.view(np.int16)).to(dev)
    gt = torch.zeros((1, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    ref = np.zeros((1, 3 * dst[0] * dst[1]), np.float32)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_16UC1), cvgs.GpuMat.from_tensor(gt, cvgs.CV_32FC1))
    assert cvgs.kernel_name(*ops) == "k4_nv12_resize_arith", cvgs.kernel_name(*ops)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_16UC1), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    got = gt.cpu().numpy()
    H.assert_bit_exact(got, ref, "P010 canonical arithmetic program")
    gt.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(got, gt.cpu().numpy(), "canonical vs the generic kernel")
