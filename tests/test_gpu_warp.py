"""GPU parity for cvGS::warp through the C-ABI: bit-exact against the CPU oracle (strict fp32 on both
sides), on the chain shapes of the reference's warp test (tests/warping/test_warping_opencv.cu) and beyond."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests import kat_runner as K
from tests import warp_cases as WC
from tests.test_gpu_chains import _both, _random_src

pytestmark = pytest.mark.gpu


def test_affine_translation_reference_chain():
    """warp<Affine, CV_8UC3>(img, [1 0 50; 0 1 100], size) -> fk::Cast<float3, uchar3> -> write<CV_8UC3>: must equal
    cv::cuda::warpAffine exactly in the reference (test_warping_opencv.cu:80-117) = the shifted image."""
    src = H.random_u8((300, 400, 3), 11)

    def build(wrap, wrap_out, out):
        return [cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, wrap(src, cvgs.CV_8UC3), [[1, 0, 50], [0, 1, 100]], (400, 300)),
                cvgs.cast(cvgs.CV_32FC3, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, wrap_out(out, cvgs.CV_8UC3))]

    gpu, ref = _both(build, (300, 400, 3), np.uint8)
    exp = np.zeros_like(src)
    exp[100:, 50:] = src[:200, :350]
    assert np.array_equal(gpu[0], exp)
    H.assert_bit_exact(gpu[0], ref[0], "affine translation")


@pytest.mark.parametrize("used", [5, 3])
def test_perspective_batch_reference_chain(used):
    """testPerspectiveBatch / testPerspectiveBatchNotAll: 5 (or 10, 3 used) warps of one image in ONE launch, each into
    its own pitched uchar3 GpuMat (PerThreadWrite<_2D> batch)."""
    n = 5 if used == 5 else 10
    src = H.random_u8((430, 470, 3), 31)
    fwd = [WC.get_perspective_transform(*WC.REF_POINT_SETS[i % 5]) for i in range(n)]

    def build(wrap, wrap_out, out):
        img = wrap(src, cvgs.CV_8UC3)
        rd = cvgs.warp(cvgs.WARP_PERSPECTIVE, cvgs.CV_8UC3, [img] * n, fwd, (470, 430), used, None)
        o = wrap_out(out, cvgs.CV_8UC3)
        outs = [cvgs.GpuMat(430, 470, cvgs.CV_8UC3, o.data + i * 430 * o.step, o.step, owner=o) for i in range(n)]
        return [rd, cvgs.cast(cvgs.CV_32FC3, cvgs.CV_8UC3), cvgs.write_batch(cvgs.CV_8UC3, outs)]

    gpu, ref = _both(build, (n * 430, 470, 3), np.uint8)
    H.assert_bit_exact(gpu[0], ref[0], "perspective batch, %d of %d used" % (used, n))
    planes = ref[0].reshape(n, 430, 470, 3)
    assert planes[0].std() > 20 and (planes[used:] == 0).all()


@pytest.mark.parametrize("depth,cn", [("8U", 1), ("8U", 4), ("16U", 3), ("16S", 2), ("32S", 3), ("32F", 3)])
@pytest.mark.parametrize("kind", [cvgs.WARP_AFFINE, cvgs.WARP_PERSPECTIVE])
def test_warp_all_source_types_into_nchw(depth, cn, kind):
    """rotation + scale (affine) / a keystone (perspective) on every source type, normalized and split into NCHW fp32."""
    src = _random_src((150, 210, cn), depth, 70 + cn)
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    a = np.deg2rad(17.0)
    aff = [[1.3 * np.cos(a), -1.3 * np.sin(a), 31.5], [1.3 * np.sin(a), 1.3 * np.cos(a), -12.25]]
    per = WC.get_perspective_transform([(10, 12), (200, 5), (3, 140), (190, 149)], [(0, 0), (96, 0), (0, 64), (96, 64)])
    n, dst = 3, (96, 64)

    def build(wrap, wrap_out, out):
        img = wrap(src, st)
        ms = [aff if kind == cvgs.WARP_AFFINE else per] * n
        if kind == cvgs.WARP_AFFINE:
            ms = [[[r[0], r[1], r[2] + 9 * i] for r in aff] for i in range(n)]
        rd = cvgs.warp(kind, st, [img] * n, ms, dst, 2, [3.0, 5.0, 7.0, 11.0][:cn])
        return [rd, cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]),
                cvgs.split(f, wrap_out(out, cvgs.CV_32FC1), dst) if cn > 1 else cvgs.write(f, wrap_out(out, cvgs.CV_32FC1), dst)]

    gpu, ref = _both(build, (n, cn * dst[0] * dst[1]), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "warp %sC%d kind %d" % (depth, cn, kind))
    assert ref[0][0].std() > 0


def test_warp_many_planes_uses_device_table():
    """More planes than travel in the kernel arguments (56): the descriptors are uploaded stream-ordered."""
    import torch
    src = H.random_u8((90, 120, 3), 8)
    n = 70
    ms = [[[1.0, 0.02 * i, 0.5 * i], [-0.01 * i, 1.0, 0.25 * i]] for i in range(n)]

    def build(wrap, wrap_out, out):
        img = wrap(src, cvgs.CV_8UC3)
        return [cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [img] * n, ms, (64, 48)),
                cvgs.split(cvgs.CV_32FC3, wrap_out(out, cvgs.CV_32FC1), (64, 48))]

    gpu, ref = _both(build, (n, 3 * 64 * 48), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "70-plane warp")
    t = torch.zeros((90, 120, 3), dtype=torch.uint8, device="cuda:0")
    o = torch.zeros((n, 3 * 64 * 48), dtype=torch.float32, device="cuda:0")
    name = cvgs.kernel_name(cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [cvgs.GpuMat.from_tensor(t, cvgs.CV_8UC3)] * n, ms, (64, 48)),
                            cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), (64, 48)))
    assert name == "warp_affine_u8c3_interp"


def test_fk_cast_truncation_on_gpu():
    vals = np.array([-300.7, -1.5, -0.5, 0.0, 0.5, 0.999, 1.5, 2.5, 254.999, 255.5, 300.2, 70000.0, -70000.0, 3e9, -3e9, np.nan,
                     np.inf, -np.inf], np.float32)
    src = np.tile(vals, (4, 1))[:, :, None].copy()
    for depth, dt in ((cvgs.CV_8U, np.uint8), (cvgs.CV_8S, np.int8), (cvgs.CV_16U, np.uint16), (cvgs.CV_16S, np.int16),
                      (cvgs.CV_32S, np.int32)):
        t = cvgs.make_type(depth, 1)

        def build(wrap, wrap_out, out):
            return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [wrap(src, cvgs.CV_32FC1)], 1), cvgs.cast(cvgs.CV_32FC1, t),
                    cvgs.write(t, wrap_out(out, t))]

        gpu, ref = _both(build, src.shape, dt)
        H.assert_bit_exact(gpu[0], ref[0], "fk::Cast -> depth %d" % depth)


def test_warp_descriptor_validation():
    import torch
    t = torch.zeros((16, 16, 3), dtype=torch.uint8, device="cuda:0")
    o = torch.zeros((1, 3 * 64), dtype=torch.float32, device="cuda:0")
    m = cvgs.GpuMat.from_tensor(t, cvgs.CV_8UC3)
    rd = cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, m, [[1, 0, 0], [0, 1, 0]], (8, 8))
    wr = cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), (8, 8))
    s = torch.cuda.current_stream()
    cvgs.executeOperations(s, rd, wr)
    rd.warp = None  # descriptor without matrices
    with pytest.raises(capi.CvgsError, match="warp_matrices"):
        cvgs.executeOperations(s, rd, wr)
    with pytest.raises(capi.CvgsError, match="plane tables"):
        rd2 = cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, m, [[1, 0, 0], [0, 1, 0]], (8, 8))
        cvgs.build_plane_table(rd2)
    torch.cuda.synchronize()


@pytest.mark.parametrize("cn", [3, 4])
@pytest.mark.parametrize("kind", [cvgs.WARP_AFFINE, cvgs.WARP_PERSPECTIVE])
@pytest.mark.parametrize("src_wh", [(470, 430), (2, 5), (1, 1)])
def test_fast_warp_kernel_agrees_with_interpreted(cn, kind, src_wh):
    """u8 C3/C4 -> [swap,] mul, sub, div -> NCHW takes the fast warp kernel (one 8-byte load per tap pair); it must give
    the interpreted kernel's bits, also on sources narrower than the load window."""
    sw, sh = src_wh
    src = H.random_u8((sh, sw, cn), 400 + cn + sw)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    n, dst = 6, (112, 96)
    rng = np.random.default_rng(cn + sw)
    if kind == cvgs.WARP_AFFINE:
        ms = [[[rng.uniform(0.2, 3) * 112 / sw, rng.uniform(-0.3, 0.3), rng.uniform(-20, 20)],
               [rng.uniform(-0.3, 0.3), rng.uniform(0.2, 3) * 96 / sh, rng.uniform(-20, 20)]] for _ in range(n)]
    else:
        ms = [WC.get_perspective_transform([(0, 0), (sw, 0.1 * sh), (0.05 * sw, sh), (sw, sh)],
                                           [(3 * i, 2), (110 - i, 5 + i), (1, 90), (105, 95 - 2 * i)]).tolist() for i in range(n)]
    code = cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA

    def build(wrap, wrap_out, out):
        img = wrap(src, u)
        return [cvgs.warp(kind, u, [img] * n, ms, dst, n - 1, [5.0, 6.0, 7.0, 8.0][:cn]), cvgs.cvtColor(code, f), cvgs.multiply(f, [0.3] * cn),
                cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]), cvgs.split(f, wrap_out(out, cvgs.CV_32FC1), dst)]

    gpu, ref = _both(build, (n, cn * dst[0] * dst[1]), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "fast warp vs oracle")
    gen, _ = _both(build, (n, cn * dst[0] * dst[1]), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], gen[0], "fast warp vs interpreted")
    import torch
    t = torch.from_numpy(src).cuda()
    o = torch.zeros((n, cn * dst[0] * dst[1]), dtype=torch.float32, device="cuda")
    ops = [cvgs.warp(kind, u, [cvgs.GpuMat.from_tensor(t, u)] * n, ms, dst), cvgs.cvtColor(code, f), cvgs.multiply(f, [0.3] * cn),
           cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]), cvgs.split(f, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), dst)]
    assert cvgs.kernel_name(*ops) == "warp_%s_u8c%d_swap_mul_sub_div" % ("affine" if kind == cvgs.WARP_AFFINE else "perspective", cn)
    assert cvgs.kernel_name(*ops, flags=capi.CHAIN_FORCE_GENERIC).endswith("_interp")


@pytest.mark.parametrize("cn,gray", [(3, False), (4, False), (3, True)])
def test_fast_warp_packed_nhwc(cn, gray):
    """N warps -> packed fp32 pixels (NHWC hand-off), also with a channel-count change (RGB2GRAY) inside the chain."""
    src = H.random_u8((200, 260, cn), 55 + cn)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    n, dst = 4, (100, 60)
    ms = [[[0.6 + 0.1 * i, 0.1, -5.0 * i], [-0.05, 0.7, 3.0 * i]] for i in range(n)]
    ocn = 1 if gray else cn
    ot = cvgs.make_type(cvgs.CV_32F, ocn)

    def build(wrap, wrap_out, out):
        img = wrap(src, u)
        ops = [cvgs.warp(cvgs.WARP_AFFINE, u, [img] * n, ms, dst, n - 1, [9.0, 8.0, 7.0, 6.0][:cn]), cvgs.multiply(f, [0.5] * cn)]
        if gray:
            ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2GRAY, f, ot))
        return ops + [cvgs.write(ot, wrap_out(out, ot), dst)]

    gpu, ref = _both(build, (n, dst[0] * dst[1], ocn), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "warp -> NHWC")
    gen, _ = _both(build, (n, dst[0] * dst[1], ocn), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gpu[0], gen[0], "fast vs interpreted")
    import torch
    t = torch.from_numpy(src).cuda()
    o = torch.zeros((n, dst[0] * dst[1], cn), dtype=torch.float32, device="cuda")
    name = cvgs.kernel_name(cvgs.warp(cvgs.WARP_AFFINE, u, [cvgs.GpuMat.from_tensor(t, u)] * n, ms, dst),
                            cvgs.write(f, cvgs.GpuMat.from_tensor(o, f), dst))
    assert name == "warp_affine_u8c%d_packed_f32" % cn


@pytest.mark.parametrize("kind", [cvgs.WARP_AFFINE, cvgs.WARP_PERSPECTIVE])
@pytest.mark.parametrize("n", [3, 12])
def test_warp_into_a_double_precision_chain(kind, n):
    """cvGS::warp followed by a detour through CV_64F (convertTo<CV_32FC3, CV_64FC3>, arithmetic with scalars that are not
    float-representable, back to CV_32F): the warp read in front of the double-precision interpreter (k_warp64), with the
    planes in the kernel arguments (3) and in an uploaded table (12)."""
    src = _random_src((120, 160, 3), "8U", 23)
    st, f, d = cvgs.CV_8UC3, cvgs.CV_32FC3, cvgs.CV_64FC3
    a = np.deg2rad(-9.0)
    aff = [[1.1 * np.cos(a), -1.1 * np.sin(a), 7.25], [1.1 * np.sin(a), 1.1 * np.cos(a), 3.5]]
    per = WC.get_perspective_transform([(8, 9), (150, 4), (2, 110), (155, 118)], [(0, 0), (80, 0), (0, 60), (80, 60)])
    dst = (80, 60)

    def build(wrap, wrap_out, out):
        img = wrap(src, st)
        ms = [[[r[0], r[1], r[2] + 2 * i] for r in aff] for i in range(n)] if kind == cvgs.WARP_AFFINE else [per] * n
        rd = cvgs.warp(kind, st, [img] * n, ms, dst, n - 1, [3.0, 5.0, 7.0])
        return [rd, cvgs.convertTo(f, d), cvgs.multiply(d, [0.1, 1.0 / 3.0, 0.7]), cvgs.subtract(d, [1e-9, 4.0, 3.2]),
                cvgs.divide(d, [3.3, 0.6, 11.8]), cvgs.convertTo(d, f), cvgs.split(f, wrap_out(out, cvgs.CV_32FC1), dst)]

    gpu, ref = _both(build, (n, 3 * dst[0] * dst[1]), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "warp -> CV_64F chain, kind %d, %d planes" % (kind, n))
    assert ref[0][0].std() > 0


@pytest.mark.parametrize("kind", [cvgs.WARP_AFFINE, cvgs.WARP_PERSPECTIVE])
@pytest.mark.parametrize("write", ["images", "planes"])
@pytest.mark.parametrize("double_chain", [False, True])
def test_batched_warp_with_one_destination_size_per_plane(oracle, kind, write, double_chain):
    """cvGS::warp<WT, I, BATCH>(inputs, matrices, std::array<cv::Size, BATCH>) (reference include/cvGPUSpeedup.cuh:381-401):
    every plane warps into its own size; the outputs are one image (or one set of channel planes) per plane.  Planes beyond
    usedPlanes carry the default value at THEIR size."""
    import torch
    dev = torch.device("cuda:0")
    src = _random_src((90, 130, 3), "8U", 31)
    st, f, d = cvgs.CV_8UC3, cvgs.CV_32FC3, cvgs.CV_64FC3
    sizes = [(70, 33), (16, 48), (97, 5), (40, 40)]
    n, used = len(sizes), 3
    a = np.deg2rad(11.0)
    aff = [[0.9 * np.cos(a), -0.9 * np.sin(a), 4.5], [0.9 * np.sin(a), 0.9 * np.cos(a), -2.0]]
    per = WC.get_perspective_transform([(5, 6), (120, 3), (2, 80), (125, 86)], [(0, 0), (60, 0), (0, 40), (60, 40)])
    ms = [[[r[0], r[1], r[2] + 3 * i] for r in aff] for i in range(n)] if kind == cvgs.WARP_AFFINE else [per] * n
    pw = [cvgs.multiply(f, [0.5, 0.25, 2.0])]
    if double_chain:
        pw = [cvgs.convertTo(f, d), cvgs.multiply(d, [1.0 / 3.0, 0.1, 0.7]), cvgs.convertTo(d, f)]

    def run(wrap, make_out):
        img = wrap(src, st)
        rd = cvgs.warp(kind, st, [img] * n, ms, sizes, used, [3.0, 5.0, 7.0])
        if write == "images":
            outs = [make_out((h, w, 3)) for (w, h) in sizes]
            wr = cvgs.write_batch(f, [o[1] for o in outs])
        else:
            outs = [[make_out((h, w)) for _ in range(3)] for (w, h) in sizes]
            wr = cvgs.split(f, [[p[1] for p in img_planes] for img_planes in outs])
        return [rd] + pw + [wr], outs

    t_src = torch.from_numpy(src).to(dev)

    def gpu_out(shape):
        t = torch.full(shape, -9.0, dtype=torch.float32, device=dev)
        return t, cvgs.GpuMat.from_tensor(t, f if len(shape) == 3 else cvgs.CV_32FC1)

    def cpu_out(shape):
        a_ = np.full(shape, -9.0, np.float32)
        return a_, cvgs.GpuMat.from_array(a_, f if len(shape) == 3 else cvgs.CV_32FC1)

    g_ops, g_outs = run(lambda s_, t_: cvgs.GpuMat.from_tensor(t_src, t_), gpu_out)
    c_ops, c_outs = run(lambda s_, t_: cvgs.GpuMat.from_array(s_, t_), cpu_out)
    cvgs.executeOperations(torch.cuda.current_stream(), *g_ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(c_ops))
    flat_g = [o[0] for o in g_outs] if write == "images" else [p[0] for img_planes in g_outs for p in img_planes]
    flat_c = [o[0] for o in c_outs] if write == "images" else [p[0] for img_planes in c_outs for p in img_planes]
    for i, (g, c) in enumerate(zip(flat_g, flat_c)):
        assert (c != -9.0).all(), "the oracle must fill every pixel of output %d" % i
        H.assert_bit_exact(g.cpu().numpy(), c, "per-plane sized warp, output %d" % i)
    # the unused plane holds the default value pushed through the chain, at its own size
    last = flat_c[-1] if write == "images" else flat_c[-3]
    assert np.unique(last.reshape(-1, last.shape[-1] if write == "images" else 1), axis=0).shape[0] == 1


def test_differently_sized_warps_need_one_image_per_plane(lib):
    import ctypes as C
    src = np.zeros((20, 30, 3), np.uint8)
    out = np.zeros((2, 3 * 16 * 16), np.float32)
    m = cvgs.GpuMat.from_array(src, cvgs.CV_8UC3)
    rd = cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [m, m], [[[1, 0, 0], [0, 1, 0]]] * 2, [(16, 16), (8, 16)])
    ch = cvgs.lower([rd, cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), (16, 16))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID and b"one destination image per plane" in lib.cvgs_last_error()
    outs = [np.zeros((16, 16, 3), np.float32), np.zeros((16, 16, 3), np.float32)]  # second image has the wrong size
    ch = cvgs.lower([rd, cvgs.write_batch(cvgs.CV_32FC3, [cvgs.GpuMat.from_array(o, cvgs.CV_32FC3) for o in outs])])
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    outs[1] = np.zeros((16, 8, 3), np.float32)
    ch = cvgs.lower([rd, cvgs.write_batch(cvgs.CV_32FC3, [cvgs.GpuMat.from_array(o, cvgs.CV_32FC3) for o in outs])])
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
