"""GPU parity for every chain shape of the hot path, through the C-ABI:
 * every reference known-answer vector (tests/golden/reference_kats.json) directly on the GPU,
 * the same chains on NON-constant seeded inputs, bit-exact against the CPU oracle (integer and fp32 alike:
   both sides are strict IEEE, no FMA),
 * write kinds (TensorSplit / TensorTSplit / packed 3D / pitched 2D / SplitWrite), default-value planes."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests import kat_runner as K

pytestmark = pytest.mark.gpu

CASES = [c for c in K.load_cases() if "kind" not in c]


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_matches_reference_kat(case):
    out = K.run_chain_case(case, "gpu")
    K.check_chain_case(case, out)


def _random_src(shape, depth_name, seed):
    h, w, c = shape
    if depth_name in ("8U", "8S"):
        return H.random_u8(shape, seed).view(K.NP_DEPTH[depth_name])
    if depth_name in ("16U", "16S"):
        return H.random_u16(shape, seed).view(K.NP_DEPTH[depth_name])
    if depth_name == "32S":
        return (H.random_u16(shape, seed).astype(np.int32) - 30000) * 7
    if depth_name == "64F":
        return H.random_u16(shape, seed).astype(np.float64) / 48.0 - 500.0 + 1e-7 * H.random_u16(shape, seed + 1).astype(np.float64)
    if depth_name == "16F":
        return (H.random_u16(shape, seed).astype(np.float32) / 64.0 - 300.0).astype(np.float16)
    return (H.random_u16(shape, seed).astype(np.float32) / 64.0 - 300.0).astype(np.float32)


GUARD = 4096  # bytes of canary on both sides of every GPU output buffer


def _both(build, out_shape, out_dtype, flags=0):
    """build(mem_kind, wrap_in, wrap_out) -> iops; runs on the oracle and on the GPU, returns both outputs.  Every GPU
    output buffer sits between two canary bands that must come back untouched (out-of-bounds stores)."""
    import torch
    from oracle import oracle_binding as ob
    dev = torch.device("cuda:0")
    res = {}
    for backend in ("oracle", "gpu"):
        keep = []
        guards = []

        def wrap(a, cvt, guarded=False):
            if backend == "gpu":
                if guarded:
                    big = torch.full((a.nbytes + 2 * GUARD,), 0xA5, dtype=torch.uint8, device=dev)
                    t = big[GUARD:GUARD + a.nbytes].view(torch.from_numpy(a).dtype).view(a.shape)
                    t.copy_(torch.from_numpy(a))
                    guards.append(big)
                else:
                    t = torch.from_numpy(a).to(dev)
                keep.append(t)
                return cvgs.GpuMat.from_tensor(t, cvt)
            keep.append(a)
            return cvgs.GpuMat.from_array(a, cvt)

        out = np.full(out_shape, 0, out_dtype)
        outs = []

        def wrap_out(a, cvt):
            m = wrap(a, cvt, guarded=True)
            outs.append((keep[-1], a))
            return m

        iops = build(wrap, wrap_out, out)
        if backend == "gpu":
            cvgs.executeOperations(torch.cuda.current_stream(), *iops, flags=flags)
            torch.cuda.synchronize()
            res[backend] = [t.cpu().numpy() for t, _ in outs]
            for big in guards:
                assert bool((big[:GUARD] == 0xA5).all()) and bool((big[-GUARD:] == 0xA5).all()), "store outside the output buffer"
        else:
            ob.execute(cvgs.lower(iops, flags))
            res[backend] = [a for _, a in outs]
    return res["gpu"], res["oracle"]


SRC_TYPES = [("8U", 1), ("8U", 2), ("8U", 3), ("8U", 4), ("8S", 3), ("16U", 1), ("16U", 3), ("16U", 4), ("16S", 3),
             ("16S", 4), ("32S", 3), ("32F", 1), ("32F", 3), ("32F", 4)]


@pytest.mark.parametrize("depth,cn", SRC_TYPES)
def test_batch_resize_all_source_types(depth, cn):
    """K1 on every source depth/channel count the reference sweeps (and more), non-constant input."""
    src = _random_src((300, 400, cn), depth, 100 + cn)
    stype, ftype = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    crops = H.random_crops(9, 400, 300, seed=77 + cn, wmin=1, wmax=300, hmin=1, hmax=250)
    dst = (64, 128)

    def build(wrap, wrap_out, out):
        frame = wrap(src, stype)
        rd = cvgs.resize(stype, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, 9)
        ops = [rd, cvgs.multiply(ftype, [0.3] * cn), cvgs.subtract(ftype, H.K1_SUB[cn]), cvgs.divide(ftype, H.K1_DIV[cn])]
        return ops + [cvgs.split(ftype, wrap_out(out, cvgs.CV_32FC1), dst)]

    gpu, ref = _both(build, (9, cn * 64 * 128), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "batch resize %sC%d" % (depth, cn))


@pytest.mark.parametrize("depth,cn", [("8U", 3), ("8S", 4), ("16U", 2), ("16S", 3), ("32S", 3), ("32F", 4)])
def test_pointwise_chain_random(depth, cn):
    """K6 (read -> convertTo -> sub -> mul -> div -> add -> write) on random pixels, pitched input and output."""
    src = _random_src((70, 90, cn), depth, 5)
    stype, ftype = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)

    def build(wrap, wrap_out, out):
        full = wrap(src, stype)
        inp = full.roi(3, 2, 80, 60)  # pitched view
        o = wrap_out(out, ftype)
        return [cvgs.ReadIOp(capi.READ_PIXEL, stype, [inp], 1), cvgs.convertTo(stype, ftype),
                cvgs.subtract(ftype, [0.3] * cn), cvgs.multiply(ftype, H.K1_SUB[cn]), cvgs.divide(ftype, H.K1_DIV[cn]),
                cvgs.add(ftype, H.K1_DIV[cn]), cvgs.write(ftype, o)]

    gpu, ref = _both(build, (60, 80, cn), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "pointwise %sC%d" % (depth, cn))


GUARDED_DIV_CASES = {
    # name: (multiplier before the division, subtrahend, divisors) -- what the thread-fused kernels' run-time guarded division (k_common.hpp:
    # div4_guarded) must get right: dividends inside the guarded range, zeros (black pixels: +0 and -0), dividends below 2^-90 / above 2^38,
    # overflow to infinity, NaN (inf * 0, inf - inf), divisors the host must refuse (significand all ones, outside [2^-20, 2^20]), negative ones
    "plain": ([0.3, 0.3, 0.3, 0.3], [1.0, 4.0, 3.2, 0.5], [3.2, 0.6, 11.8, 33.0]),
    "zeros_plus": ([1.0, 1.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [3.2, 0.6, 11.8, 33.0]),
    "zeros_minus": ([-1.0, -1.0, -1.0, -1.0], [0.0, 0.0, 0.0, 0.0], [3.2, -0.6, 11.8, -33.0]),
    "tiny": ([1e-29, 3e-30, 1e-28, 1e-30], [0.0, 0.0, 0.0, 0.0], [3.2, 0.6, 11.8, 33.0]),
    "huge": ([1e10, 3e11, 1e9, 1e12], [0.0, 0.0, 0.0, 0.0], [3.2, 0.6, 11.8, 33.0]),
    "overflow": ([3e38, 3e38, 3e38, 3e38], [0.0, 0.0, 0.0, 0.0], [0.5, 0.6, 0.8, 0.25]),
    "inf_minus_inf": ([3e38, 3e38, 3e38, 3e38], [float("inf")] * 4, [3.2, 0.6, 11.8, 33.0]),
    "all_ones_divisor": ([0.3, 0.3, 0.3, 0.3], [1.0, 4.0, 3.2, 0.5], [float(np.float32(2.0) - np.float32(2.0 ** -23)), 0.6, 11.8, 33.0]),
    "divisor_out_of_range": ([0.3, 0.3, 0.3, 0.3], [1.0, 4.0, 3.2, 0.5], [3.2, 2.0 ** 24, 11.8, 33.0]),
    "negative_divisors": ([0.3, -0.3, 0.3, -0.3], [1.0, 4.0, 3.2, 0.5], [-3.2, 0.6, -11.8, 33.0]),
    "power_of_two_divisors": ([0.3, 0.3, 0.3, 0.3], [1.0, 4.0, 3.2, 0.5], [2.0, 0.5, 1024.0, 2.0 ** -20]),
}


@pytest.mark.parametrize("case", sorted(GUARDED_DIV_CASES))
@pytest.mark.parametrize("depth,cn,interpreted", [("8U", 3, False), ("8U", 4, True), ("8U", 1, True), ("16S", 3, True), ("16U", 2, False), ("32F", 3, True)])
def test_run_time_guarded_division(depth, cn, interpreted, case):
    """The pointwise programs divide by reciprocal + two FMA corrections when the host finds the divisors fit AND every dividend of the WAVE is
    inside [2^-90, 2^38] (else the wave divides for real): a frame whose upper part is non-zero pixels and whose lower part holds zeros, so some
    waves take each side; compile-time program (cast, mul, sub, div) and interpreted one (cast, mul, sub, div, add); vs the oracle's IEEE division
    and vs the interpreted per-pixel kernel."""
    mul, sub, div = GUARDED_DIV_CASES[case]
    w, h = 517, 23
    src = _random_src((h, w, cn), depth, 4000 + cn)
    flat = src.reshape(h, w * cn)
    if depth != "32F":
        flat[: h // 2] |= 1  # the upper rows: no zero anywhere
        flat[h // 2 + 2:, 100 * cn:300 * cn] = 0  # a black block further down
    else:
        flat[h // 2 + 2:, 100 * cn:300 * cn] = np.array([0.0, -0.0, np.inf, 1e-40], np.float32)[np.arange(200 * cn) % 4]
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)

    def build(wrap, wrap_out, out_buf):
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, [wrap(src, st)], 1)]
        if depth != "32F":
            ops.append(cvgs.convertTo(st, f))
        ops += [cvgs.multiply(f, mul[:cn]), cvgs.subtract(f, sub[:cn]), cvgs.divide(f, div[:cn])]
        if interpreted:
            ops.append(cvgs.add(f, [0.25] * cn))
        return ops + [cvgs.write(f, wrap_out(out_buf, f))]

    with np.errstate(all="ignore"):
        gpu, ref = _both(build, (h, w, cn), np.float32)
        gen, _ = _both(build, (h, w, cn), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], ref[0], "guarded division %s, %sC%d" % (case, depth, cn))
    H.assert_bit_exact(gpu[0], gen[0], "fused vs interpreted per-pixel kernel")


@pytest.mark.parametrize("dst_depth", ["8U", "8S", "16U", "16S", "32S"])
def test_saturate_cast_rounding(dst_depth):
    """fk::SaturateCast float -> integer: nearest-even, clamped, NaN -> 0; via convertTo(alpha, beta)."""
    vals = np.array([-1e10, -70000.5, -32768.5, -129.5, -128.5, -2.5, -1.5, -0.5, 0.0, 0.5, 1.5, 2.5, 10.5, 15.5, 126.5,
                     127.5, 254.5, 255.5, 256.5, 32767.5, 65534.5, 65535.5, 1e10, np.nan, np.inf, -np.inf],
                    np.float32)
    src = np.tile(vals, (4, 1))[:, :, None].copy()
    otype = cvgs.make_type(K.CV_DEPTH[dst_depth], 1)

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [wrap(src, cvgs.CV_32FC1)], 1),
                cvgs.convertTo(cvgs.CV_32FC1, otype), cvgs.write(otype, wrap_out(out, otype))]

    gpu, ref = _both(build, (4, len(vals), 1), K.NP_DEPTH[dst_depth])
    H.assert_bit_exact(gpu[0], ref[0], "saturate cast -> %s" % dst_depth)
    if dst_depth == "8U":
        assert list(ref[0][0, :, 0][[9, 10, 11, 12, 13, 17, 23]]) == [0, 2, 2, 10, 16, 255, 0]


@pytest.mark.parametrize("code,it,ot", [("RGB2BGR", "8UC3", "8UC3"), ("RGBA2BGRA", "16UC4", "16UC4"),
                                         ("RGB2GRAY", "8UC3", "8UC1"), ("BGRA2GRAY", "16UC4", "16UC1"),
                                         ("RGB2GRAY", "32FC3", "32FC1")])
def test_cvtcolor_random(code, it, ot):
    d, cn, itype = K.parse_type(it)
    od, ocn, otype = K.parse_type(ot)
    src = _random_src((50, 61, cn), d, 9)

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, itype, [wrap(src, itype)], 1), cvgs.cvtColor(K.CODES[code], itype, otype),
                cvgs.write(otype, wrap_out(out, otype))]

    gpu, ref = _both(build, (50, 61, ocn), K.NP_DEPTH[od])
    H.assert_bit_exact(gpu[0], ref[0], "cvtColor %s" % code)


def test_add_drop_alpha():
    src = _random_src((20, 33, 3), "8U", 3)

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [wrap(src, cvgs.CV_8UC3)], 1),
                cvgs.cvtColor(cvgs.COLOR_RGB2BGRA, cvgs.CV_8UC3, cvgs.CV_8UC4),
                cvgs.cvtColor(cvgs.COLOR_BGRA2RGBA, cvgs.CV_8UC4), cvgs.write(cvgs.CV_8UC4, wrap_out(out, cvgs.CV_8UC4))]

    gpu, ref = _both(build, (20, 33, 4), np.uint8)
    H.assert_bit_exact(gpu[0], ref[0], "add alpha")
    assert (gpu[0][..., 3] == 255).all() and (gpu[0][..., :3] == src).all()


@pytest.mark.parametrize("kind", ["tensor_t_split", "write3d", "split_planes_batch", "write2d_batch"])
def test_write_kinds(kind):
    """TensorTSplit (CNHW), PerThreadWrite<_3D>, SplitWrite and batched 2D writes of a resized batch."""
    src = _random_src((200, 300, 3), "8U", 17)
    crops = H.random_crops(20, 300, 200, seed=3, wmin=5, wmax=200, hmin=5, hmax=150)
    dst, n, cn = (48, 40), 20, 3
    f = cvgs.CV_32FC3

    def build(wrap, wrap_out, out):
        frame = wrap(src, cvgs.CV_8UC3)
        ops = [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n),
               cvgs.multiply(f, [0.5, 0.25, 2.0])]
        if kind == "tensor_t_split":
            o = wrap_out(out, cvgs.CV_32FC1)
            return ops + [cvgs.splitT(f, o.data, dst[0], dst[1], n, keep=o)]
        if kind == "write3d":
            return ops + [cvgs.write(f, wrap_out(out, f), dst)]
        if kind == "split_planes_batch":
            planes = [[wrap_out(np.zeros((dst[1], dst[0], 1), np.float32), cvgs.CV_32FC1) for _ in range(cn)] for _ in range(n)]
            return ops + [cvgs.split(f, planes)]
        outs = [wrap_out(np.zeros((dst[1], dst[0], 3), np.float32), f) for _ in range(n)]
        return ops + [cvgs.write_batch(f, outs)]

    shape = {"tensor_t_split": (cn * n, dst[0] * dst[1]), "write3d": (n, dst[0] * dst[1], 3)}.get(kind, (1, 1))
    gpu, ref = _both(build, shape, np.float32)
    assert len(gpu) == len(ref) and len(gpu) >= 1
    for g, r in zip(gpu, ref):
        H.assert_bit_exact(g, r, kind)
    if kind == "tensor_t_split":  # CNHW: channel-major
        nchw_gpu, _ = _both(lambda w, wo, o: build_nchw(w, wo, o, src, crops, dst, n), (n, cn * dst[0] * dst[1]), np.float32)
        a = gpu[0].reshape(cn, n, dst[1], dst[0])
        b = nchw_gpu[0].reshape(n, cn, dst[1], dst[0]).transpose(1, 0, 2, 3)
        H.assert_bit_exact(a, np.ascontiguousarray(b), "CNHW vs NCHW")


def build_nchw(wrap, wrap_out, out, src, crops, dst, n):
    frame = wrap(src, cvgs.CV_8UC3)
    return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n),
            cvgs.multiply(cvgs.CV_32FC3, [0.5, 0.25, 2.0]), cvgs.split(cvgs.CV_32FC3, wrap_out(out, cvgs.CV_32FC1), dst)]


def test_batch_pixel_read_default_value():
    """executeOperations(array<GpuMat,N>, activeBatch, defaultValue, ...) (reference :506-516)."""
    srcs = [_random_src((30, 40, 3), "8U", 50 + i) for i in range(6)]

    def build(wrap, wrap_out, out):
        mats = [wrap(s, cvgs.CV_8UC3) for s in srcs]
        rd = cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, mats, 4, None, cvgs.IGNORE_AR, [9.0, 8.0, 7.0])
        return [rd, cvgs.convertTo(cvgs.CV_8UC3, cvgs.CV_32FC3, 0.5), cvgs.write(cvgs.CV_32FC3, wrap_out(out, cvgs.CV_32FC3), (40, 30))]

    gpu, ref = _both(build, (6, 40 * 30, 3), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "activeBatch/default")
    assert (gpu[0][4:] == np.array([4.5, 4.0, 3.5], np.float32)).all()


# 701: two full 256-pixel groups (LDS-transposed packed stores) + a ragged tail; 76 / 704 / 2052: widths the C1 / C2 -> C4
# widening of k_pointwise4 divides (2052 / 4 = 513: two full groups of the widened image + a tail)
@pytest.mark.parametrize("w", [77, 701, 76, 704, 2052])
@pytest.mark.parametrize("cn", [1, 2, 3, 4])
@pytest.mark.parametrize("write_kind", ["write2d", "write3d", "split", "splitT"])
def test_thread_fused_pointwise_kernel(cn, write_kind, w):
    """The 4-pixels-per-thread kernel (u8 -> fp32): odd widths (tail lanes), pitched views, batches with default-value
    planes, packed and planar outputs -- bit-exact vs the oracle AND vs the interpreted kernel."""
    h, n = 19, (1 if write_kind == "write2d" else 5)
    srcs = [_random_src((h + 3, w + 9, cn), "8U", 900 + 10 * cn + i) for i in range(n)]
    stype, ftype = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)

    def build(wrap, wrap_out, out):
        mats = [wrap(s, stype).roi(5, 2, w, h) for s in srcs]
        used = n if n == 1 else n - 2
        rd = cvgs.ReadIOp(capi.READ_PIXEL, stype, mats, used, None, cvgs.IGNORE_AR, [9.0, 8.0, 7.0, 6.0][:cn])
        ops = [rd, cvgs.convertTo(stype, ftype, 0.5), cvgs.subtract(ftype, H.K1_SUB[cn]), cvgs.divide(ftype, H.K1_DIV[cn])]
        if write_kind == "write2d":
            return ops + [cvgs.write(ftype, wrap_out(out, ftype))]
        if write_kind == "write3d":
            return ops + [cvgs.write(ftype, wrap_out(out, ftype), (w, h))]
        o = wrap_out(out, cvgs.CV_32FC1)
        if write_kind == "split":
            return ops + [cvgs.split(ftype, o, (w, h))]
        return ops + [cvgs.splitT(ftype, o.data, w, h, n, keep=o)]

    shape = {"write2d": (h, w, cn), "write3d": (n, w * h, cn), "split": (n, cn * w * h), "splitT": (cn * n, w * h)}[write_kind]
    gpu, ref = _both(build, shape, np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "thread-fused pointwise c%d %s" % (cn, write_kind))
    gen, _ = _both(build, shape, np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], gen[0], "fused vs interpreted")


def test_thread_fused_kernel_is_selected():
    frame = np.zeros((16, 32, 3), np.uint8)
    out = np.zeros((16, 32, 3), np.float32)
    f = cvgs.CV_32FC3
    ops = [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1),
           cvgs.convertTo(cvgs.CV_8UC3, f, 1.0), cvgs.subtract(f, [1, 2, 3]), cvgs.divide(f, [1, 2, 3]),
           cvgs.write(f, cvgs.GpuMat.from_array(out, f))]
    assert cvgs.kernel_name(*ops) == "pointwise4_u8_cast_mul_sub_div"
    assert cvgs.kernel_name(*ops, flags=capi.CHAIN_NO_THREAD_FUSION).startswith("generic")


@pytest.mark.parametrize("src,cn", [("32F", 3), ("8U", 4), ("32S", 2), ("64F", 1), ("64F", 3)])
def test_double_precision_chains(src, cn):
    """CV_64F values: convertTo<I, CV_64F>(alpha, beta) then double arithmetic with non-float-representable scalars,
    written as CV_64F and (after a second cast) as CV_32F / CV_16S.  Bit-exact vs the oracle."""
    if src == "64F":
        a = (H.random_u16((40, 50, cn), 77).astype(np.float64) / 7.0 - 3000.0)
    else:
        a = _random_src((40, 50, cn), src, 77)
    stype = cvgs.make_type(K.CV_DEPTH[src], cn)
    d, f, s16 = cvgs.make_type(cvgs.CV_64F, cn), cvgs.make_type(cvgs.CV_32F, cn), cvgs.make_type(cvgs.CV_16S, cn)
    third = [1.0 / 3.0, 0.1, 2.0 / 7.0, 1e-3][:cn]

    def build_for(out_kind):
        def build(wrap, wrap_out, out):
            rd = cvgs.ReadIOp(capi.READ_PIXEL, stype, [wrap(a, stype)], 1)
            ops = [rd, cvgs.convertTo(stype, d, 0.3, 0.7), cvgs.subtract(d, third), cvgs.divide(d, [3.2, 0.6, 11.8, 33.0][:cn]),
                   cvgs.multiply(d, third)]
            if out_kind == "64F":
                return ops + [cvgs.write(d, wrap_out(out, d))]
            if out_kind == "32F":
                return ops + [cvgs.convertTo(d, f), cvgs.add(f, [0.1] * cn), cvgs.write(f, wrap_out(out, f))]
            return ops + [cvgs.convertTo(d, s16), cvgs.write(s16, wrap_out(out, s16))]
        return build

    for kind, dt in (("64F", np.float64), ("32F", np.float32), ("16S", np.int16)):
        gpu, ref = _both(build_for(kind), (40, 50, cn), dt)
        H.assert_bit_exact(gpu[0], ref[0], "64F chain %sC%d -> %s" % (src, cn, kind))


@pytest.mark.parametrize("off", [16, 3])
@pytest.mark.parametrize("nbytes", [1, 15, 4096, (8 << 20) + 7, (40 << 20) + 3, 600 << 20])
def test_stream_copy(nbytes, off):
    """cvgs_stream_copy: every size class (tail-only, unaligned tail, multi-job), aligned and unaligned pointers."""
    import torch
    lib = capi.load_library()
    g = torch.Generator(device="cuda").manual_seed(nbytes % 1000)
    src = torch.randint(0, 256, (nbytes + 32,), dtype=torch.uint8, device="cuda", generator=g)
    dst = torch.zeros(nbytes + 32, dtype=torch.uint8, device="cuda")
    capi.check(lib.cvgs_stream_copy(dst.data_ptr() + off, src.data_ptr() + off, nbytes, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    assert torch.equal(dst[off:off + nbytes], src[off:off + nbytes])
    assert not dst[:off].any() and not dst[off + nbytes:].any()


@pytest.mark.parametrize("depth,cn", [("8S", 1), ("8S", 3), ("16U", 1), ("16U", 3), ("16U", 4), ("16S", 2), ("16S", 3), ("32S", 1),
                                      ("32S", 3), ("32F", 1), ("32F", 2), ("32F", 3), ("32F", 4)])
@pytest.mark.parametrize("out", ["packed", "planar"])
@pytest.mark.parametrize("w", [523, 1036])  # 1036: a width the C1 / C2 -> C4 widening divides (259 / 518 wide as C4)
def test_thread_fused_pointwise_other_depths(depth, cn, out, w):
    """The reference sweeps its pointwise chains over every source depth (tests/batchread/test_batchread_x_write3D.cu:202-227,
    tests/read/test_read_x_write.cu:121-144): 4 pixels per thread for 8S/16U/16S/32S/32F sources too -- wide rows (full
    256-pixel groups + a ragged tail), pitched views, default-value planes; vs the oracle and vs the interpreted kernel."""
    h, n = 11, 3
    srcs = [_random_src((h + 2, w + 7, cn), depth, 1200 + 10 * cn + i) for i in range(n)]
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)

    def build(wrap, wrap_out, out_buf):
        mats = [wrap(s, st).roi(3, 1, w, h) for s in srcs]
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, mats, n - 1, None, cvgs.IGNORE_AR, [9.0, 8.0, 7.0, 6.0][:cn])]
        if depth != "32F":
            ops.append(cvgs.convertTo(st, f))
        ops += [cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn])]
        if out == "packed":
            return ops + [cvgs.write(f, wrap_out(out_buf, f), (w, h))]
        o = wrap_out(out_buf, cvgs.CV_32FC1)
        return ops + [cvgs.split(f, o, (w, h)) if cn > 1 else cvgs.write(f, o, (w, h))]

    shape = (n, w * h, cn) if out == "packed" else (n, cn * w * h)
    gpu, ref = _both(build, shape, np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "fused pointwise %sC%d %s" % (depth, cn, out))
    gen, _ = _both(build, shape, np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], gen[0], "fused vs interpreted")
    import torch
    t = torch.from_numpy(srcs[0]).cuda()
    o = torch.zeros((1, cn * w * h), dtype=torch.float32, device="cuda")
    chain = [cvgs.ReadIOp(capi.READ_PIXEL, st, [cvgs.GpuMat.from_tensor(t, st).roi(3, 1, w, h)], 1)]
    if depth != "32F":
        chain.append(cvgs.convertTo(st, f))
    chain.append(cvgs.multiply(f, [0.3] * cn))
    o_mat = cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1)
    chain.append(cvgs.split(f, o_mat, (w, h)) if cn > 1 else cvgs.write(f, o_mat, (w, h)))
    want = {"8S": "s8", "16U": "u16", "16S": "s16", "32S": "s32", "32F": "f32"}[depth]
    assert cvgs.kernel_name(*chain) == "pointwise4_%s" % want


@pytest.mark.parametrize("depth", ["8U", "8S", "16U", "16S", "32S", "32F"])
@pytest.mark.parametrize("size,n,used", [((1283, 821), 1, 1), ((1024, 1031), 1, 1), ((700, 509), 3, 2)])
def test_one_channel_whole_frames_two_rows_per_thread(depth, size, n, used):
    """Launches of >= 1 Mpixel of ONE-channel planes put two rows (y, y + 4 of an 8-row block) on a thread (k_pointwise_body.hpp: pw4_body_rows2):
    heights that leave a partial 8-row block with and without its lower rows, a ragged last pixel group, pitched views, a default-value plane;
    packed and planar targets, vs the oracle and vs the interpreted kernel."""
    w, h = size
    srcs = [_random_src((h + 2, w + 5, 1), depth, 3100 + i) for i in range(n)]
    st, f = cvgs.make_type(K.CV_DEPTH[depth], 1), cvgs.CV_32FC1

    def build(wrap, wrap_out, out_buf):
        mats = [wrap(s, st).roi(2, 1, w, h) for s in srcs]
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, mats, used, None, cvgs.IGNORE_AR, [9.0])]
        if depth != "32F":
            ops.append(cvgs.convertTo(st, f))
        return ops + [cvgs.multiply(f, [0.3]), cvgs.subtract(f, H.K1_SUB[1]), cvgs.divide(f, H.K1_DIV[1]), cvgs.write(f, wrap_out(out_buf, f), (w, h))]

    gpu, ref = _both(build, (n, w * h), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "one-channel whole frame %sC1 %dx%d x%d" % (depth, w, h, n))
    gen, _ = _both(build, (n, w * h), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], gen[0], "fused vs interpreted")


U8_CODES = [("BGR2RGB", 3, 3), ("RGBA2BGRA", 4, 4), ("BGR2BGRA", 3, 4), ("RGB2BGRA", 3, 4), ("BGRA2BGR", 4, 3), ("RGBA2BGR", 4, 3),
            ("BGR2GRAY", 3, 1), ("RGB2GRAY", 3, 1), ("BGRA2GRAY", 4, 1), ("RGBA2GRAY", 4, 1)]


@pytest.mark.parametrize("code,icn,ocn", U8_CODES)
def test_thread_fused_u8_to_u8_colour_conversions(code, icn, ocn):
    """Standalone cvGS::cvtColor on packed u8 images (reference tests/color/test_cvtColor.cu): 4 pixels per thread, the
    channel count may change; wide pitched views with a ragged tail, batch with a default-value plane."""
    w, h, n = 517, 9, 3
    srcs = [_random_src((h + 1, w + 6, icn), "8U", 1500 + i) for i in range(n)]
    it, ot = cvgs.make_type(cvgs.CV_8U, icn), cvgs.make_type(cvgs.CV_8U, ocn)
    cc = getattr(cvgs, "COLOR_" + code)

    def build(wrap, wrap_out, out):
        mats = [wrap(s, it).roi(2, 1, w, h) for s in srcs]
        return [cvgs.ReadIOp(capi.READ_PIXEL, it, mats, n - 1, None, cvgs.IGNORE_AR, [200.0, 100.0, 50.0, 25.0][:icn]),
                cvgs.cvtColor(cc, it, ot), cvgs.write(ot, wrap_out(out, ot), (w, h))]

    gpu, ref = _both(build, (n, w * h, ocn), np.uint8)
    H.assert_bit_exact(gpu[0], ref[0], "u8 cvtColor " + code)
    gen, _ = _both(build, (n, w * h, ocn), np.uint8, flags=capi.CHAIN_NO_THREAD_FUSION)
    H.assert_bit_exact(gpu[0], gen[0], "fused vs interpreted")
    import torch
    t = torch.from_numpy(srcs[0]).cuda()
    o = torch.zeros((h, w, ocn), dtype=torch.uint8, device="cuda")
    ops = [cvgs.ReadIOp(capi.READ_PIXEL, it, [cvgs.GpuMat.from_tensor(t, it).roi(2, 1, w, h)], 1), cvgs.cvtColor(cc, it, ot),
           cvgs.write(ot, cvgs.GpuMat.from_tensor(o, ot))]
    assert cvgs.kernel_name(*ops) == "pointwise4_u8_u8_interp"
    assert cvgs.kernel_name(*ops, flags=capi.CHAIN_NO_THREAD_FUSION).startswith("generic")


@pytest.mark.parametrize("cn", [1, 2, 3, 4])
def test_thread_fused_u8_brightness_contrast(cn):
    """convertTo<CV_8UCn, CV_8UCn>(alpha, beta): u8 -> float -> x alpha + beta -> saturating round back to u8, one pitched image."""
    src = _random_src((40, 301, cn), "8U", 77 + cn)
    t8 = cvgs.make_type(cvgs.CV_8U, cn)

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, t8, [wrap(src, t8)], 1), cvgs.convertTo(t8, t8, 1.7, -40.5), cvgs.write(t8, wrap_out(out, t8))]

    gpu, ref = _both(build, (40, 301, cn), np.uint8)
    H.assert_bit_exact(gpu[0], ref[0], "u8 brightness/contrast")
    assert ref[0].min() == 0 and ref[0].max() == 255  # both saturation ends are exercised


def test_colour_swap_on_the_source_type_before_the_cast_stays_thread_fused():
    """cvtColor<BGR2RGB, CV_8UC3>() BEFORE convertTo<CV_8UC3, CV_32FC3>() (a common spelling): still the 4-pixel kernel."""
    src = _random_src((33, 600, 3), "8U", 5)
    u, f = cvgs.CV_8UC3, cvgs.CV_32FC3

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, u, [wrap(src, u)], 1), cvgs.cvtColor(cvgs.COLOR_BGR2RGB, u), cvgs.convertTo(u, f),
                cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                cvgs.split(f, wrap_out(out, cvgs.CV_32FC1), (600, 33))]

    gpu, ref = _both(build, (1, 3 * 600 * 33), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "swap before cast")
    import torch
    t = torch.from_numpy(src).cuda()
    o = torch.zeros((1, 3 * 600 * 33), dtype=torch.float32, device="cuda")
    ops = [cvgs.ReadIOp(capi.READ_PIXEL, u, [cvgs.GpuMat.from_tensor(t, u)], 1), cvgs.cvtColor(cvgs.COLOR_BGR2RGB, u), cvgs.convertTo(u, f),
           cvgs.multiply(f, [1 / 255.0] * 3), cvgs.split(f, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), (600, 33))]
    assert cvgs.kernel_name(*ops) == "pointwise4_u8_interp"


def test_cvgs_execute_is_thread_safe(oracle):
    """The C-ABI claims re-entrancy (no shared mutable state except CircularTensor handles): four host threads launch
    different chains on their own streams concurrently (ctypes releases the GIL); every result must be right."""
    import ctypes as C
    import threading
    import torch
    dev = torch.device("cuda:0")
    lib = capi.load_library()
    jobs = []
    for t in range(4):
        src = H.random_u8((200, 300, 3), 600 + t)
        crops = H.random_crops(8 + t, 300, 200, seed=t, wmin=4, wmax=200, hmin=4, hmax=150)
        ref = np.zeros((len(crops), 3 * 64 * 128), np.float32)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(src, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1),
                                             swap=bool(t % 2))))
        st = torch.from_numpy(src).to(dev)
        outs = [torch.zeros((len(crops), 3 * 64 * 128), dtype=torch.float32, device=dev) for _ in range(50)]
        chains = [cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(st, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1),
                                        swap=bool(t % 2))) for o in outs]
        jobs.append((torch.cuda.Stream(), chains, outs, ref, st))
    errors = []

    def worker(stream, chains):
        for _ in range(4):
            for ch in chains:
                rc = lib.cvgs_execute(C.byref(ch.desc), stream.cuda_stream)
                if rc:
                    errors.append(lib.cvgs_last_error())
    torch.cuda.synchronize()
    threads = [threading.Thread(target=worker, args=(j[0], j[1])) for j in jobs]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    torch.cuda.synchronize()
    assert not errors, errors[:3]
    for _, _, outs, ref, _ in jobs:
        for o in outs:
            H.assert_bit_exact(o.cpu().numpy(), ref, "concurrent launches")


# ---- standalone u8 colour conversions as compile-time programs (k_cvtcolor_u8.hip) ------------------------------------
_CODES = [("BGR2RGB", cvgs.COLOR_BGR2RGB, 3, 3), ("BGRA2RGBA", cvgs.COLOR_BGRA2RGBA, 4, 4), ("BGR2BGRA", cvgs.COLOR_BGR2BGRA, 3, 4),
          ("BGR2RGBA", cvgs.COLOR_BGR2RGBA, 3, 4), ("BGRA2BGR", cvgs.COLOR_BGRA2BGR, 4, 3), ("RGBA2BGR", cvgs.COLOR_RGBA2BGR, 4, 3),
          ("BGR2GRAY", cvgs.COLOR_BGR2GRAY, 3, 1), ("RGB2GRAY", cvgs.COLOR_RGB2GRAY, 3, 1), ("BGRA2GRAY", cvgs.COLOR_BGRA2GRAY, 4, 1),
          ("RGBA2GRAY", cvgs.COLOR_RGBA2GRAY, 4, 1)]


@pytest.mark.parametrize("name,code,icn,ocn", _CODES, ids=[c[0] for c in _CODES])
@pytest.mark.parametrize("shape", [(37, 640, 1), (5, 4096, 3), (9, 16, 2), (7, 637, 1), (6, 656, 1)])
def test_u8_colour_conversions_fast_and_interpreted_agree_with_oracle(oracle, name, code, icn, ocn, shape):
    """The reference's cvtColor codes (tests/color/test_cvtColor.cu:105-123) on NON-constant u8 images: aligned widths take
    the 16-pixel-per-thread permutation / gray kernels, the ragged width (637) and the 8-byte-aligned pitch (656 * 3 with a
    crop start) stay on the interpreted kernel; all of them must give the oracle's bytes."""
    import torch
    dev = torch.device("cuda:0")
    h, w, batch = shape
    it, ot = cvgs.make_type(cvgs.CV_8U, icn), cvgs.make_type(cvgs.CV_8U, ocn)
    srcs = [H.random_u8((h, w, icn), seed=300 + b) for b in range(batch)]
    ts = [torch.from_numpy(s).to(dev) for s in srcs]

    def chain(mats, out, flags=0):
        if batch == 1:
            return [cvgs.ReadIOp(capi.READ_PIXEL, it, [mats[0]], 1), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out)]
        return [cvgs.ReadIOp(capi.READ_PIXEL, it, mats, batch), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out, (w, h))]

    if batch == 1:
        out_t = torch.zeros((h, w, ocn), dtype=torch.uint8, device=dev)
        ref = np.zeros((h, w, ocn), np.uint8)
    else:
        out_t = torch.zeros((batch, h * w, ocn), dtype=torch.uint8, device=dev)
        ref = np.zeros((batch, h * w, ocn), np.uint8)
    g_ops = chain([cvgs.GpuMat.from_tensor(t, it) for t in ts], cvgs.GpuMat.from_tensor(out_t, ot))
    name_k = cvgs.kernel_name(*g_ops)
    aligned = w % 16 == 0
    assert name_k.startswith("pointwise16_u8_" if aligned else "pointwise4_u8_u8"), (name_k, shape)
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain([cvgs.GpuMat.from_array(s, it) for s in srcs], cvgs.GpuMat.from_array(ref, ot))))
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%s %s via %s" % (name, shape, name_k))
    out_t.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops, flags=capi.CHAIN_NO_THREAD_FUSION)
    torch.cuda.synchronize()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%s %s interpreted" % (name, shape))


@pytest.mark.parametrize("name,code,icn,ocn", _CODES, ids=[c[0] for c in _CODES])
@pytest.mark.parametrize("shape", [(37, 640, 1), (3, 4096, 2), (9, 16, 2), (7, 637, 1)])
def test_u16_colour_conversions_fast_and_interpreted_agree_with_oracle(oracle, name, code, icn, ocn, shape):
    """The same codes on CV_16U images (the reference's cvtColor test sweeps CV_16UC3 / C4 next to CV_8U,
    tests/color/test_cvtColor.cu:105-123): aligned widths take the 16-pixel-per-thread kernels with 2-byte channels, the
    ragged width stays on the interpreted kernel; full-range 16-bit values (alpha = 65535, gray of 65535 stays 65535)."""
    import torch
    dev = torch.device("cuda:0")
    h, w, batch = shape
    it, ot = cvgs.make_type(cvgs.CV_16U, icn), cvgs.make_type(cvgs.CV_16U, ocn)
    srcs = [H.random_u16((h, w, icn), seed=700 + b) for b in range(batch)]
    srcs[0][0, :8] = 65535
    ts = [torch.from_numpy(s.view(np.int16)).to(dev) for s in srcs]

    def chain(mats, out):
        if batch == 1:
            return [cvgs.ReadIOp(capi.READ_PIXEL, it, [mats[0]], 1), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out)]
        return [cvgs.ReadIOp(capi.READ_PIXEL, it, mats, batch), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out, (w, h))]

    shape_o = (h, w, ocn) if batch == 1 else (batch, h * w, ocn)
    out_t = torch.zeros(shape_o, dtype=torch.int16, device=dev)
    ref = np.zeros(shape_o, np.uint16)
    g_ops = chain([cvgs.GpuMat.from_tensor(t, it) for t in ts], cvgs.GpuMat.from_tensor(out_t, ot))
    name_k = cvgs.kernel_name(*g_ops)
    assert name_k.startswith("pointwise16_u16_" if w % 16 == 0 else "generic"), (name_k, shape)
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain([cvgs.GpuMat.from_array(s, it) for s in srcs], cvgs.GpuMat.from_array(ref, ot))))
    assert ref.any()
    H.assert_bit_exact(out_t.cpu().numpy().view(np.uint16), ref, "%s %s via %s" % (name, shape, name_k))
    out_t.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops, flags=capi.CHAIN_NO_THREAD_FUSION)
    torch.cuda.synchronize()
    H.assert_bit_exact(out_t.cpu().numpy().view(np.uint16), ref, "%s %s interpreted" % (name, shape))


@pytest.mark.parametrize("name,code,icn,ocn", _CODES, ids=[c[0] for c in _CODES])
@pytest.mark.parametrize("shape", [(37, 640, 1), (3, 4096, 2), (9, 4, 2), (7, 637, 1), (5, 260, 1)])
def test_f32_colour_conversions_fast_and_interpreted_agree_with_oracle(oracle, name, code, icn, ocn, shape):
    """The same codes on CV_32F images (tests/color/test_cvtColor.cu:105-123 sweeps CV_32FC3 / C4 too): widths that are multiples
    of 4 take the four-pixels-per-thread kernels (channel permutations move bits: NaNs, infinities and -0 travel unchanged;
    gray is (R 0.299 + G 0.587) + B 0.114 without rounding), ragged widths stay on the interpreted kernel."""
    import torch
    dev = torch.device("cuda:0")
    h, w, batch = shape
    it, ot = cvgs.make_type(cvgs.CV_32F, icn), cvgs.make_type(cvgs.CV_32F, ocn)
    srcs = [(H.random_u16((h, w, icn), seed=900 + b).astype(np.float32) / 64.0 - 300.0).astype(np.float32) for b in range(batch)]
    srcs[0][0, :4, 0] = [np.inf, -np.inf, -0.0, 3.0e38]
    ts = [torch.from_numpy(s).to(dev) for s in srcs]

    def chain(mats, out):
        if batch == 1:
            return [cvgs.ReadIOp(capi.READ_PIXEL, it, [mats[0]], 1), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out)]
        return [cvgs.ReadIOp(capi.READ_PIXEL, it, mats, batch), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out, (w, h))]

    shape_o = (h, w, ocn) if batch == 1 else (batch, h * w, ocn)
    out_t = torch.zeros(shape_o, dtype=torch.float32, device=dev)
    ref = np.zeros(shape_o, np.float32)
    g_ops = chain([cvgs.GpuMat.from_tensor(t, it) for t in ts], cvgs.GpuMat.from_tensor(out_t, ot))
    name_k = cvgs.kernel_name(*g_ops)
    assert (name_k in ("pointwise4_f32_gray", "pointwise4_f32_permute")) == (w % 4 == 0), (name_k, shape)
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain([cvgs.GpuMat.from_array(s, it) for s in srcs], cvgs.GpuMat.from_array(ref, ot))))
    assert ref.any()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%s %s via %s" % (name, shape, name_k))
    out_t.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops, flags=capi.CHAIN_NO_THREAD_FUSION)
    torch.cuda.synchronize()
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%s %s interpreted" % (name, shape))


@pytest.mark.parametrize("name,code", [("RGB2GRAY", cvgs.COLOR_RGB2GRAY), ("BGR2GRAY", cvgs.COLOR_BGR2GRAY)])
def test_u8_gray_every_rgb_triple(oracle, name, code):
    """EVERY 8-bit (c0, c1, c2) triple through the compile-time-order gray kernel (whose store-side conversion does the
    rounding: k_cvtcolor_u8.hip) against the oracle's 0.299 R + 0.587 G + 0.114 B, rounded to nearest even: 2^24 pixels as one
    4096 x 4096 image."""
    import torch
    dev = torch.device("cuda:0")
    it, ot = cvgs.make_type(cvgs.CV_8U, 3), cvgs.make_type(cvgs.CV_8U, 1)
    idx = np.arange(1 << 24, dtype=np.uint32)
    src = np.stack([(idx & 255), (idx >> 8) & 255, (idx >> 16) & 255], axis=-1).astype(np.uint8).reshape(4096, 4096, 3)
    t = torch.from_numpy(src).to(dev)
    out_t = torch.zeros((4096, 4096, 1), dtype=torch.uint8, device=dev)
    ref = np.zeros((4096, 4096, 1), np.uint8)

    def chain(mat, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, it, [mat], 1), cvgs.cvtColor(code, it, ot), cvgs.write(ot, out)]

    g_ops = chain(cvgs.GpuMat.from_tensor(t, it), cvgs.GpuMat.from_tensor(out_t, ot))
    assert cvgs.kernel_name(*g_ops) == "pointwise16_u8_gray"
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(cvgs.GpuMat.from_array(src, it), cvgs.GpuMat.from_array(ref, ot))))
    assert ref.max() == 255 and ref.min() == 0
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "%s over all 2^24 triples" % name)


def test_u8_colour_conversion_of_an_unaligned_crop_stays_interpreted(oracle):
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((64, 200, 3), seed=17)
    ft = torch.from_numpy(frame).to(dev)
    out_t = torch.zeros((32, 64, 3), dtype=torch.uint8, device=dev)
    ref = np.zeros((32, 64, 3), np.uint8)

    def chain(m, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [m.roi(3, 5, 64, 32)], 1), cvgs.cvtColor(cvgs.COLOR_BGR2RGB, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, out)]

    g_ops = chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(out_t, cvgs.CV_8UC3))
    assert cvgs.kernel_name(*g_ops).startswith("pointwise4_u8_u8")  # the crop starts 9 bytes into a row: no 16-byte loads
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), cvgs.GpuMat.from_array(ref, cvgs.CV_8UC3))))
    H.assert_bit_exact(out_t.cpu().numpy(), ref, "unaligned crop")


# ---- read -> convertTo -> split(std::vector<GpuMat>): the reference's tests/read/test_read_x_split.cu -------------------
@pytest.mark.parametrize("depth", ["8U", "8S", "16U", "16S", "32S", "32F"])
@pytest.mark.parametrize("cn", [2, 3, 4])
@pytest.mark.parametrize("normalize", [False, True])
def test_read_convert_split_into_separate_planes(depth, cn, normalize):
    """cvGS::executeOperations(input, stream, convertTo<I, O>(), split<O>(std::vector<GpuMat>)) over the type pairs the
    reference sweeps (tests/read/test_read_x_split.cu:112-130; CV_32F sources go to CV_64F there, covered by the CV_64F
    tests): separate PITCHED planes, widths that are not a multiple of the 4-pixel thread tile, the thread-fused kernel
    (SplitWrite target) vs the oracle and vs the interpreted kernel."""
    w, h = 333, 41
    pad = 5  # plane pitch = (w + pad) floats
    src = _random_src((h, w, cn), depth, 900 + cn)
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)

    def build(wrap, wrap_out, out):
        o = wrap_out(out, cvgs.CV_32FC1)  # (cn * h, w + pad): plane c = rows [c*h, (c+1)*h), columns [0, w)
        planes = [cvgs.GpuMat(h, w, cvgs.CV_32FC1, o.data + c * h * o.step, o.step, owner=o) for c in range(cn)]
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, [wrap(src, st)], 1)]
        if depth != "32F":
            ops.append(cvgs.convertTo(st, f))
        if normalize:
            ops += [cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn])]
        elif depth == "32F":
            ops.append(cvgs.multiply(f, [0.5] * cn))
        return ops + [cvgs.split(f, planes)]

    fast, ref = _both(build, (cn * h, w + pad), np.float32)
    interp, _ = _both(build, (cn * h, w + pad), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    assert ref[0].any() and (ref[0][:, w:] == 0).all()
    H.assert_bit_exact(fast[0], ref[0], "%sC%d -> planes (dispatcher's kernel) vs oracle" % (depth, cn))
    H.assert_bit_exact(interp[0], ref[0], "%sC%d -> planes (interpreted kernel) vs oracle" % (depth, cn))
    ops = build(lambda a, t: cvgs.GpuMat.from_array(a, t), lambda a, t: cvgs.GpuMat.from_array(a, t), np.zeros((cn * h, w + pad), np.float32))
    assert cvgs.kernel_name(*ops).startswith("pointwise4_"), cvgs.kernel_name(*ops)


def test_fields_a_stage_does_not_use_do_not_change_the_kernel_or_the_bits(oracle):
    """A binding written against include/cvgs_hip.h fills a descriptor its own way: junk in fields the chain's stages do not use (the
    background of an IGNORE_AR resize, the YUV fields of a pixel read, the step of a dense tensor, the aux word of an arithmetic stage, the
    fourth operand of a 3-channel value, selector bits beyond the value's channels, NOP stages in between) must neither change the kernel
    the dispatcher picks nor a bit of the result."""
    import ctypes as C
    import torch
    from cvgpuspeedup_amd import capi
    dev = torch.device("cuda:0")
    frame = H.random_u8((540, 960, 3), seed=4242)
    ft = torch.from_numpy(frame).to(dev)
    crops = H.random_crops(9, 960, 540, seed=4243, wmin=8, wmax=400, hmin=8, hmax=400)
    out_a = torch.zeros((9, 3 * 64 * 128), dtype=torch.float32, device=dev)
    out_b = torch.zeros_like(out_a)
    lib = capi.load_library()
    s = torch.cuda.current_stream().cuda_stream

    def build(out):
        return cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)))

    a, b = build(out_a), build(out_b)
    d = b.desc
    for c in range(4):
        d.read.background[c] = 1e30
    d.read.yuv_range, d.read.yuv_primaries, d.read.yuv_alpha = 1, 2, 1
    d.write.step = 12345
    n = d.n_ops
    assert n == 4 and d.ops[0].opcode == capi.OP_REORDER
    d.ops[0].aux |= 3 << 6            # a selector for a fourth channel the value does not have
    for k in (1, 2, 3):
        d.ops[k].aux = 77             # arithmetic stages have no aux
        d.ops[k].operand[3] = -5.0    # nor a fourth channel
        d.ops[k].operand_d[3] = -5.0
    # NOP stages in between: REORDER, NOP, MUL, NOP, SUB, DIV
    ops = [d.ops[k] for k in range(4)]
    saved = [(o.opcode, o.aux, list(o.operand), list(o.operand_d)) for o in ops]
    order = [0, None, 1, None, 2, 3]
    for i, src in enumerate(order):
        o = d.ops[i]
        if src is None:
            o.opcode, o.aux = capi.OP_NOP, 99
        else:
            o.opcode, o.aux = saved[src][0], saved[src][1]
            for c in range(4):
                o.operand[c] = saved[src][2][c]
                o.operand_d[c] = saved[src][3][c]
    d.n_ops = len(order)
    name_a, name_b = C.create_string_buffer(128), C.create_string_buffer(128)
    capi.check(lib.cvgs_kernel_name(C.byref(a.desc), name_a, 128))
    capi.check(lib.cvgs_kernel_name(C.byref(d), name_b, 128))
    assert name_a.value == name_b.value == b"k1_u8c3_swap_mul_sub_div", (name_a.value, name_b.value)
    capi.check(lib.cvgs_execute(C.byref(a.desc), s))
    capi.check(lib.cvgs_execute(C.byref(d), s))
    torch.cuda.synchronize()
    assert bool(torch.equal(out_a.view(torch.int32), out_b.view(torch.int32)))
    assert bool(out_a.abs().sum() > 0)
