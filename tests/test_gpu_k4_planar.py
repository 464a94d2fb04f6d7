"""Planar-chroma surfaces (I420 / YV12: software decoders' yuv420p) through the fused NV12 resize kernel K4 -- the pixel-format
parameter of the reference's reader (fk::ReadYUV<PF>, tests/resize/test_fused_resize.cu:50-51,141-147) on the fast path: stretch
and letterboxed resizes (include/cvGPUSpeedup.cuh:32,218-245), default-value planes, RGB- / BGR-order normalisation into planar
fp32 / fp16 tensors, packed u8 images, alpha, every range / primaries pair, CircularTensor pushes.  Each case is compared bit for
bit with the CPU oracle and with the interpreted kernel; the same picture stored as NV12 must give the same tensor."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu

PLANAR = [capi.YUV_I420, capi.YUV_YV12]


def planar_surface(w, h, seed, layout):
    """(surface in `layout`, the same picture as NV12)."""
    y = H.random_u8((h, w), seed)
    u = H.random_u8((h // 2, w // 2), seed + 1)
    v = H.random_u8((h // 2, w // 2), seed + 2)
    s = np.zeros((h + h // 2, w), np.uint8)
    s[:h] = y
    first, second = (u, v) if layout == capi.YUV_I420 else (v, u)
    q = (h // 2) * (w // 2)
    s[h:].reshape(-1)[:q] = first.reshape(-1)
    s[h:].reshape(-1)[q:2 * q] = second.reshape(-1)
    nv = np.zeros_like(s)
    nv[:h] = y
    nv[h:, 0::2] = u
    nv[h:, 1::2] = v
    return s, nv


def luma_of(wrap, surf, w, h):
    m = wrap(surf)
    return cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)


def run_both(oracle, build, surfs, shp, np_dt, ot, want_prefix="k4_nv12_resize", rtol=None):
    """build(wrap, out) -> ops.  GPU fast path and interpreted path vs the oracle."""
    import torch
    dev = torch.device("cuda:0")
    tdt = {np.float32: torch.float32, np.uint8: torch.uint8, np.float16: torch.float16}[np_dt]
    ref = np.zeros(shp, np_dt)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), cvgs.GpuMat.from_array(ref, ot))))
    ts = {id(s): torch.from_numpy(s).to(dev) for s in surfs}
    gt = torch.zeros(shp, dtype=tdt, device=dev)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts[id(a)], cvgs.CV_8UC1), cvgs.GpuMat.from_tensor(gt, ot))
    name = cvgs.kernel_name(*ops)
    assert name.startswith(want_prefix), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    assert ref.any()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "fast path %s" % name)
    gt.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "interpreted")
    return ref, name


@pytest.mark.parametrize("layout", PLANAR)
@pytest.mark.parametrize("shape", [((640, 360), (213, 120)), ((640, 360), (64, 128)), ((1920, 1080), (1280, 720)), ((322, 198), (70, 66)),
                                   ((64, 36), (200, 150)), ((6, 4), (9, 7)), ((4, 4), (64, 3)), ((130, 2), (65, 5))])
@pytest.mark.parametrize("prog", ["bgr_norm", "rgb_norm", "plain", "u8"])
def test_planar_stretch(oracle, layout, shape, prog):
    (w, h), dst = shape
    surf, nv = planar_surface(w, h, 7000 + w + h, layout)
    f, u = cvgs.CV_32FC3, cvgs.CV_8UC3
    norm = [cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]

    def mk(lay, s):
        def build(wrap, out):
            rd = cvgs.read_nv12(luma_of(wrap, s, w, h), dst, capi.YUV_LIMITED, capi.BT709, False, layout=lay)
            if prog == "bgr_norm":
                return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f)] + norm + [cvgs.split(f, out, dst)]
            if prog == "rgb_norm":
                return [rd] + norm + [cvgs.split(f, out, dst)]
            if prog == "plain":
                return [rd, cvgs.multiply(f, [0.5, 0.25, 2.0]), cvgs.split(f, out, dst)]
            return [rd, cvgs.convertTo(f, u), cvgs.write(u, out)]
        return build

    if prog == "u8":
        shp, dt, ot = (dst[1], dst[0], 3), np.uint8, u
    else:
        shp, dt, ot = (1, 3 * dst[0] * dst[1]), np.float32, cvgs.CV_32FC1
    want = {"bgr_norm": "k4_nv12_resize_swap_mul_sub_div", "rgb_norm": "k4_nv12_resize_mul_sub_div", "plain": "k4_nv12_resize_arith",
            "u8": "k4_nv12_resize_u8c3"}[prog]
    ref, _ = run_both(oracle, mk(layout, surf), [surf], shp, dt, ot, want)
    # the same picture as NV12 (the reference's own format) gives the same output
    ref_nv = np.zeros(shp, dt)
    oracle.execute(cvgs.lower(mk(capi.YUV_NV12, nv)(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), cvgs.GpuMat.from_array(ref_nv, ot))))
    H.assert_bit_exact(ref, ref_nv, "planar == NV12 picture")


@pytest.mark.parametrize("layout", PLANAR)
@pytest.mark.parametrize("ar", [cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_RN_EVEN, cvgs.PRESERVE_AR_LEFT])
@pytest.mark.parametrize("shape", [((640, 360), (64, 64)), ((360, 640), (96, 64)), ((1920, 1080), (640, 640)), ((322, 198), (70, 70))])
@pytest.mark.parametrize("prog", ["rgb_norm", "bgr_norm", "u8_batch"])
def test_planar_letterbox_and_default_planes(oracle, layout, ar, shape, prog):
    (w, h), dst = shape
    s0, _ = planar_surface(w, h, 7100 + w, layout)
    s1, _ = planar_surface(w, h, 7200 + w, layout)
    f, u = cvgs.CV_32FC3, cvgs.CV_8UC3
    n = 3

    def build(wrap, out):
        mats = [luma_of(wrap, s0, w, h), luma_of(wrap, s1, w, h), luma_of(wrap, s0, w, h)]
        rd = cvgs.read_nv12(mats, dst, capi.YUV_LIMITED, capi.BT601, False, layout=layout)
        rd.ar = ar
        rd.background = cvgs._scalar([114.0, 100.5, 7.25])
        rd.used_planes = 2
        norm = [cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]
        if prog == "rgb_norm":
            return [rd] + norm + [cvgs.split(f, out, dst)]
        if prog == "bgr_norm":
            return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f)] + norm + [cvgs.split(f, out, dst)]
        return [rd, cvgs.convertTo(f, u), cvgs.write(u, out, dst)]

    if prog == "u8_batch":
        shp, dt, ot = (n, dst[0] * dst[1], 3), np.uint8, u
    else:
        shp, dt, ot = (n, 3 * dst[0] * dst[1]), np.float32, cvgs.CV_32FC1
    run_both(oracle, build, [s0, s1], shp, dt, ot)


@pytest.mark.parametrize("layout", PLANAR)
@pytest.mark.parametrize("range_", [capi.YUV_FULL, capi.YUV_LIMITED])
@pytest.mark.parametrize("prim", [capi.BT601, capi.BT709, capi.BT2020])
@pytest.mark.parametrize("alpha", [False, True])
def test_planar_ranges_primaries_alpha(oracle, layout, range_, prim, alpha):
    w, h, dst = 322, 198, (101, 77)
    surf, _ = planar_surface(w, h, 7300, layout)
    f = cvgs.CV_32FC4 if alpha else cvgs.CV_32FC3
    cn = 4 if alpha else 3

    def build(wrap, out):
        rd = cvgs.read_nv12(luma_of(wrap, surf, w, h), dst, range_, prim, alpha, layout=layout)
        return [rd, cvgs.multiply(f, [0.5, 0.25, 2.0, 1.5][:cn]), cvgs.split(f, out, dst)]

    run_both(oracle, build, [surf], (1, cn * dst[0] * dst[1]), np.float32, cvgs.CV_32FC1)


@pytest.mark.parametrize("layout", PLANAR)
def test_planar_fp16_tensor(oracle, layout):
    w, h, dst = 640, 360, (224, 224)
    surf, _ = planar_surface(w, h, 7400, layout)
    f = cvgs.CV_32FC3

    def build(wrap, out):
        rd = cvgs.read_nv12(luma_of(wrap, surf, w, h), dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout)
        return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]),
                cvgs.divide(f, [0.229, 0.224, 0.225]), cvgs.convertTo(f, cvgs.CV_16FC3), cvgs.split(cvgs.CV_16FC3, out, dst)]

    run_both(oracle, build, [surf], (1, 3 * dst[0] * dst[1]), np.float16, cvgs.CV_16FC1, "k4_nv12_resize_swap_mul_sub_div_f16")


@pytest.mark.parametrize("layout", PLANAR)
def test_planar_many_planes(oracle, layout):
    """More planes than the small argument block holds (frames of several software decoders in one launch)."""
    w, h, dst, n = 96, 64, (40, 24), 70
    surfs = [planar_surface(w, h, 7500 + i, layout)[0] for i in range(4)]
    f = cvgs.CV_32FC3

    def build(wrap, out):
        mats = [luma_of(wrap, surfs[i % 4], w, h) for i in range(n)]
        rd = cvgs.read_nv12(mats, dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout)
        return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]),
                cvgs.divide(f, [0.229, 0.224, 0.225]), cvgs.split(f, out, dst)]

    run_both(oracle, build, surfs, (n, 3 * dst[0] * dst[1]), np.float32, cvgs.CV_32FC1)


@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_I420, capi.YUV_YV12])
@pytest.mark.parametrize("shape", [((640, 360), (213, 120)), ((1920, 1080), (640, 360)), ((322, 198), (64, 66)), ((322, 198), (128, 5)),
                                   ((64, 36), (200, 150)), ((3840, 2160), (1920, 1080)), ((6, 4), (63, 7))])
@pytest.mark.parametrize("prog", ["cast", "swap_cast", "scale_cast", "swap_scale_add_cast"])
@pytest.mark.parametrize("batch", [False, True])
def test_u8_image_outputs(oracle, layout, shape, prog, batch):
    """Decoder surface -> packed u8 C3 image(s) (thumbnails, display surfaces): resize -> [swap / scale in float] ->
    SaturateCast -> write, the reference's resize -> convertTo<CV_32FC3, CV_8UC3> -> write (tests/resize/test_resize_write.cu)
    behind its NV12 reader: the cast is the store's conversion, full tiles leave as dword stores; values beyond 0..255 saturate."""
    (w, h), dst = shape
    if layout in PLANAR:
        surf = planar_surface(w, h, 7600 + w, layout)[0]
    else:
        surf = H.random_u8((h * 3 // 2, w), 7600 + w)
    f, u = cvgs.CV_32FC3, cvgs.CV_8UC3
    n = 3 if batch else 1

    def build(wrap, out):
        luma = luma_of(wrap, surf, w, h)
        rd = cvgs.read_nv12([luma] * n if batch else luma, dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout)
        if batch:
            rd.used_planes = 2
            rd.background = cvgs._scalar([300.0, -4.0, 17.5])
        ops = [rd]
        if prog.startswith("swap"):
            ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f))
        if prog == "scale_cast":
            ops.append(cvgs.convertTo(f, u, 1.7))  # saturates the bright pixels
        elif prog == "swap_scale_add_cast":
            ops.append(cvgs.convertTo(f, u, 0.75, -20.5))  # and the dark ones
        else:
            ops.append(cvgs.convertTo(f, u))
        return ops + [cvgs.write(u, out, dst) if batch else cvgs.write(u, out)]

    shp = (n, dst[0] * dst[1], 3) if batch else (dst[1], dst[0], 3)
    want = {"cast": "k4_nv12_resize_u8c3", "swap_cast": "k4_nv12_resize_swap_u8c3", "scale_cast": "k4_nv12_resize_interp_u8c3",
            "swap_scale_add_cast": "k4_nv12_resize_interp_u8c3"}[prog]
    ref, name = run_both(oracle, build, [surf], shp, np.uint8, u, want)
    assert name == want
    if prog == "scale_cast" and w > 6:
        assert (ref == 255).any()


@pytest.mark.parametrize("layout", PLANAR)
@pytest.mark.parametrize("n_cams,frames_per", [(4, 3), (16, 1)])
def test_execute_many_planar_surfaces(oracle, layout, n_cams, frames_per):
    """Frames of several software decoders -> one NCHW tensor per camera in ONE launch (cvgs_execute_many over K4 with planar
    chroma): bit-identical to one launch per camera and to the oracle."""
    import torch
    dev = torch.device("cuda:0")
    w, h, dst = 320, 180, (64, 128)
    f = cvgs.CV_32FC3
    chains, outs, refs, keep = [], [], [], []
    for cam in range(n_cams):
        surfs = [planar_surface(w, h, 7700 + 10 * cam + i, layout)[0] for i in range(frames_per)]
        ts = [torch.from_numpy(s).to(dev) for s in surfs]
        ot = torch.full((frames_per, 3 * dst[0] * dst[1]), -3.0, dtype=torch.float32, device=dev)
        ref = np.full((frames_per, 3 * dst[0] * dst[1]), -3.0, np.float32)

        def chain(wrap, out):
            mats = [luma_of(wrap, i, w, h) for i in range(frames_per)]
            return [cvgs.read_nv12(mats, dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                    cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]), cvgs.split(f, out, dst)]

        chains.append(chain(lambda i: cvgs.GpuMat.from_tensor(ts[i], cvgs.CV_8UC1), cvgs.GpuMat.from_tensor(ot, cvgs.CV_32FC1)))
        oracle.execute(cvgs.lower(chain(lambda i: cvgs.GpuMat.from_array(surfs[i], cvgs.CV_8UC1), cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
        outs.append(ot)
        refs.append(ref)
        keep += ts
    assert cvgs.kernel_name(*chains[0]).startswith("k4_nv12_resize")
    cvgs.executeMany(torch.cuda.current_stream(), chains)
    torch.cuda.synchronize()
    for cam in range(n_cams):
        assert refs[cam].std() > 0.1
        H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "fused planar chains, camera %d" % cam)
        outs[cam].fill_(-3.0)
    for ops in chains:
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    for cam in range(n_cams):
        H.assert_bit_exact(outs[cam].cpu().numpy(), refs[cam], "one launch per camera, camera %d" % cam)


@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_I420, capi.YUV_YV12])
@pytest.mark.parametrize("shape", [((1920, 1080), (640, 360)), ((322, 198), (70, 66)), ((64, 36), (200, 150)), ((6, 4), (63, 7))])
@pytest.mark.parametrize("spelling", ["cast", "cast_then_reorder", "reorder_then_cast", "scale"])
@pytest.mark.parametrize("cn", [3, 4])
def test_u8_image_reference_spelling(oracle, layout, shape, spelling, cn):
    """The reference's own fused chain (tests/resize/test_fused_resize.cu:141-147): Resize(fuse(ReadYUV, ConvertYUVToRGB<..., alpha, float4>))
    -> SaturateCast<float4, uchar4> -> VectorReorder<uchar4, 2, 1, 0, 3> -> write.  The reorder behind the cast is a permutation of
    bytes, so the engine moves the cast to the end and the store does it; uchar4 pixels leave as one dword per lane."""
    (w, h), dst = shape
    surf = planar_surface(w, h, 7800 + w, layout)[0] if layout in PLANAR else H.random_u8((h * 3 // 2, w), 7800 + w)
    f, u = cvgs.make_type(cvgs.DEPTH_32F, cn), cvgs.make_type(cvgs.DEPTH_8U, cn)
    swap = cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA

    def build(wrap, out):
        rd = cvgs.read_nv12(luma_of(wrap, surf, w, h), dst, capi.YUV_FULL, capi.BT709, cn == 4, layout=layout)
        mid = {"cast": [cvgs.convertTo(f, u)], "cast_then_reorder": [cvgs.convertTo(f, u), cvgs.cvtColor(swap, u)],
               "reorder_then_cast": [cvgs.cvtColor(swap, f), cvgs.convertTo(f, u)], "scale": [cvgs.convertTo(f, u, 1.4, -30.0)]}[spelling]
        return [rd] + mid + [cvgs.write(u, out)]

    tag = "u8c%d" % cn
    want = {"cast": "k4_nv12_resize_" + tag, "cast_then_reorder": "k4_nv12_resize_swap_" + tag, "reorder_then_cast": "k4_nv12_resize_swap_" + tag,
            "scale": "k4_nv12_resize_interp_" + tag}[spelling]
    ref, name = run_both(oracle, build, [surf], (dst[1], dst[0], cn), np.uint8, u, want)
    assert name == want
    if cn == 4:
        assert (ref[..., 3] == 255).all()


@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_I420])
@pytest.mark.parametrize("shape", [((1920, 1080), (640, 360)), ((322, 198), (70, 66)), ((6, 4), (63, 7))])
@pytest.mark.parametrize("cn", [3, 4])
@pytest.mark.parametrize("batch", [False, True])
def test_packed_float_image_outputs(oracle, layout, shape, cn, batch):
    """Decoder surface -> resize -> scale -> packed CV_32FC3 / C4 image(s): the pixel leaves as one vector store per lane."""
    (w, h), dst = shape
    surf = planar_surface(w, h, 7900 + w, layout)[0] if layout in PLANAR else H.random_u8((h * 3 // 2, w), 7900 + w)
    f = cvgs.make_type(cvgs.DEPTH_32F, cn)
    n = 3 if batch else 1

    def build(wrap, out):
        luma = luma_of(wrap, surf, w, h)
        rd = cvgs.read_nv12([luma] * n if batch else luma, dst, capi.YUV_LIMITED, capi.BT601, cn == 4, layout=layout)
        if batch:
            rd.used_planes = 2
            rd.background = cvgs._scalar([3.0, -4.0, 17.5, 9.0][:cn] + [0.0] * (4 - cn))
        return [rd, cvgs.multiply(f, [0.5, 0.25, 2.0, 1.5][:cn]), cvgs.write(f, out, dst) if batch else cvgs.write(f, out)]

    shp = (n, dst[0] * dst[1], cn) if batch else (dst[1], dst[0], cn)
    run_both(oracle, build, [surf], shp, np.float32, f, "k4_nv12_resize_arith")  # (multiply: the canonical arithmetic program, round 6)


# ---- round 6: the canonical arithmetic program on decoder surfaces (k_taps.hpp: K1CanonProg through launch_n12) ---------------------------------
K4_ARITH = {
    # name: (stages after the read, the kernel it must take)
    "norm_then_add": (lambda f: [cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, [1.0, 4.0, 3.2]), cvgs.divide(f, [3.2, 0.6, 11.8]),
                                 cvgs.add(f, [0.5, 0.25, 0.125])], "k4_nv12_resize_arith"),
    "sub_div": (lambda f: [cvgs.subtract(f, [127.5] * 3), cvgs.divide(f, [58.4, 57.1, 57.4])], "k4_nv12_resize_arith"),
    "div_only": (lambda f: [cvgs.divide(f, [255.0, 127.5, 2.0])], "k4_nv12_resize_arith"),
    "swap_only": (lambda f: [cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f)], "k4_nv12_resize_arith"),
    "nothing": (lambda f: [], "k4_nv12_resize_arith"),
    "zero_products": (lambda f: [cvgs.multiply(f, [0.0, -0.0, 1.0]), cvgs.divide(f, [3.2, 0.6, 11.8]), cvgs.add(f, [-0.0] * 3)], "k4_nv12_resize_arith"),
    "refused_divisor": (lambda f: [cvgs.divide(f, [2.0 ** 24, 0.6, float(np.float32(2.0) - np.float32(2.0 ** -23))])], "k4_nv12_resize_arith"),
    "three_linear_before_div": (lambda f: [cvgs.add(f, [1.0] * 3), cvgs.multiply(f, [0.5] * 3), cvgs.subtract(f, [0.25] * 3), cvgs.divide(f, [3.0, 7.0, 9.0])], "k4_nv12_resize_interp"),
}


@pytest.mark.parametrize("name", sorted(K4_ARITH))
@pytest.mark.parametrize("layout,shape", [(capi.YUV_NV12, ((640, 360), (213, 120))), (capi.YUV_NV21, ((322, 198), (70, 66))), (capi.YUV_I420, ((640, 360), (64, 128))),
                                          (capi.YUV_NV12, ((64, 36), (200, 150)))])
def test_canonical_arithmetic_programs_on_decoder_surfaces(oracle, name, layout, shape):
    """Decoder surface -> resize -> a program that is not one of K4's compile-time ones -> planar tensor: chains of the canonical arithmetic shape take the
    straight-line K1CanonProg (one fma per linear stage, the division by reciprocal under the per-wave dividend check), others the interpreted
    kernel; both against the oracle and the forced generic kernel, down- and up-scaling, limited-range BT.601 (black surfaces give zero dividends)."""
    (w, h), dst = shape
    stages, want = K4_ARITH[name]
    surf = planar_surface(w, h, 8100 + w, layout)[0] if layout in PLANAR else H.random_u8((h * 3 // 2, w), 8100 + w)
    surf[: h // 3] = 16  # a black band: Y = 16, with the limited-range matrix R = G = B = 0 for neutral chroma rows
    f = cvgs.CV_32FC3

    def build(wrap, out):
        return [cvgs.read_nv12(luma_of(wrap, surf, w, h), dst, capi.YUV_LIMITED, capi.BT601, False, layout=layout)] + stages(f) + [cvgs.split(f, out, dst)]

    with np.errstate(all="ignore"):
        ref, got = run_both(oracle, build, [surf], (1, 3 * dst[0] * dst[1]), np.float32, cvgs.CV_32FC1, want)
    assert got == want


@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_I420])
@pytest.mark.parametrize("cn", [3, 4])
def test_canonical_program_into_a_u8_image(oracle, layout, cn):
    """Decoder surface -> resize -> swap, multiply, add (brightness / contrast) -> SaturateCast -> packed u8 image: the canonical arithmetic program in front
    of the store's conversion (kernel k4_nv12_resize_arith_u8cN), against the oracle and the forced generic kernel."""
    (w, h), dst = (640, 360), (213, 120)
    surf = planar_surface(w, h, 8300 + cn, layout)[0] if layout in PLANAR else H.random_u8((h * 3 // 2, w), 8300 + cn)
    f, u = cvgs.make_type(cvgs.DEPTH_32F, cn), cvgs.make_type(cvgs.DEPTH_8U, cn)
    swap = cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA

    def build(wrap, out):
        return [cvgs.read_nv12(luma_of(wrap, surf, w, h), dst, capi.YUV_FULL, capi.BT709, cn == 4, layout=layout), cvgs.cvtColor(swap, f),
                cvgs.multiply(f, [1.25, 0.75, 1.1, 1.0][:cn]), cvgs.add(f, [-12.5, 20.0, 0.25, 0.0][:cn]), cvgs.convertTo(f, u), cvgs.write(u, out)]

    ref, name = run_both(oracle, build, [surf], (dst[1], dst[0], cn), np.uint8, u, "k4_nv12_resize_arith_u8c%d" % cn)
    assert name == "k4_nv12_resize_arith_u8c%d" % cn
