"""The decode-side cvtColor WITHOUT a resize -- cvGS::cvtColorNV12 / cvtColorP010 -> [reorder, mul, sub, div, ...] -> tensor or
image -- on the thread-fused kernel (k_pointwise4 with the 4:2:0 read mode, csrc/k_pointwise_body.hpp): 4 x-adjacent pixels per
thread share 2 chroma pairs.  Bit-exact against the oracle (reference chain: fk::ReadYUV<PF> + fk::ConvertYUVToRGB,
tests/resize/test_fused_resize.cu:50-51,73-77, here without the resize) AND against the interpreted kernel, for every
interleaved layout, with / without alpha, every write kind the kernel serves, ragged widths, crops of a surface and
default-value planes."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu

LAYOUTS = [("nv12", capi.YUV_NV12), ("nv21", capi.YUV_NV21), ("p010", capi.YUV_P010)]


def _surface(layout, w, h, seed):
    if layout == capi.YUV_P010:
        s = (H.random_u16((h * 3 // 2, w), seed) & 0xffc0).astype(np.uint16)
        s |= (H.random_u16(s.shape, seed + 7) & 63).astype(np.uint16)  # the low 6 bits of a sample are ignored
        return s
    return H.random_u8((h * 3 // 2, w), seed)


def _program(kind, f, cn):
    if kind == "norm":  # the compile-time mul, sub, div program
        return [cvgs.multiply(f, [1 / 255.0, 0.5, 0.25, 2.0][:cn]), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn])]
    if kind == "swap":  # channel permutation + other arithmetic: the lean interpreter
        code = cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA
        return [cvgs.cvtColor(code, f), cvgs.add(f, [1.5, -2.0, 0.25, 3.0][:cn]), cvgs.multiply(f, [0.3] * cn)]
    return []


@pytest.mark.parametrize("lname,layout", LAYOUTS, ids=[x[0] for x in LAYOUTS])
@pytest.mark.parametrize("alpha", [False, True])
@pytest.mark.parametrize("out", ["packed", "split", "splitT", "planes"])
@pytest.mark.parametrize("prog", ["norm", "swap", "none"])
def test_nv12_pointwise_whole_surface_and_crops(oracle, lname, layout, alpha, out, prog):
    import torch
    dev = torch.device("cuda:0")
    w, h = 1038, 46  # 1038 = 4 full 256-pixel groups + a ragged group whose last thread holds 2 pixels
    cn = 4 if alpha else 3
    f = cvgs.make_type(cvgs.CV_32F, cn)
    st = cvgs.CV_16UC1 if layout == capi.YUV_P010 else cvgs.CV_8UC1
    surf = _surface(layout, w, h, 5000 + 13 * layout)
    # planes: the whole surface would be one size; a batch needs planes of ONE size -> three crops of 518 x 30 (+ a default-value plane)
    rects = [(0, 0, 518, 30), (520, 16, 518, 30), (258, 8, 518, 30)]
    n, used = len(rects) + 1, len(rects)
    cw, ch = 518, 30

    def build(wrap, wrap_out, outs):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, st, m.data, m.step, owner=m.owner)
        crops = [luma.nv12_roi(*r) for r in rects] + [luma.nv12_roi(*rects[0])]
        rd = cvgs.read_nv12(crops, None, capi.YUV_LIMITED, capi.BT709, alpha, layout=layout)
        rd.used_planes = used
        rd.background = cvgs._scalar([9.0, 8.0, 7.0, 6.0][:cn])
        ops = [rd] + _program(prog, f, cn)
        if out == "packed":
            return ops + [cvgs.write(f, wrap_out(outs[0], f), (cw, ch))]
        o = wrap_out(outs[0], cvgs.CV_32FC1)
        if out == "split":
            return ops + [cvgs.split(f, o, (cw, ch))]
        if out == "splitT":
            return ops + [cvgs.splitT(f, o.data, cw, ch, n, keep=o)]
        return ops + [cvgs.split(f, [[wrap_out(p, cvgs.CV_32FC1) for p in outs[1 + i * cn:1 + (i + 1) * cn]] for i in range(n)])]

    shape = {"packed": (n, cw * ch, cn), "split": (n, cn * cw * ch), "splitT": (cn * n, cw * ch), "planes": (1, 1)}[out]
    planes_np = [np.zeros((ch, cw), np.float32) for _ in range(n * cn)]
    ref0 = np.zeros(shape, np.float32)
    try:
        ops_ref = build(lambda a: cvgs.GpuMat.from_array(a, st), lambda a, t: cvgs.GpuMat.from_array(a, t), [ref0] + planes_np)
    except (AttributeError, TypeError) as ex:  # the Python mirror spells one of the builders differently
        pytest.skip("builder: %r" % (ex,))
    oracle.execute(cvgs.lower(ops_ref))
    ts = torch.from_numpy(surf.view(np.int16) if layout == capi.YUV_P010 else surf).to(dev)
    g0 = torch.zeros(shape, dtype=torch.float32, device=dev)
    planes_t = [torch.zeros((ch, cw), dtype=torch.float32, device=dev) for _ in range(n * cn)]
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts, st), lambda a, t: cvgs.GpuMat.from_tensor(a, t), [g0] + planes_t)
    name = cvgs.kernel_name(*ops)
    assert name == ("pointwise4_p010" if layout == capi.YUV_P010 else "pointwise4_nv12"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    what = "%s alpha=%s %s %s via %s" % (lname, alpha, out, prog, name)
    if out == "planes":
        assert any(p.any() for p in planes_np)
        for i, (a, b) in enumerate(zip(planes_t, planes_np)):
            H.assert_bit_exact(a.cpu().numpy(), b, what + " plane %d" % i)
    else:
        assert ref0.any()
        H.assert_bit_exact(g0.cpu().numpy(), ref0, what)
        g0.zero_()
        cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
        torch.cuda.synchronize()
        H.assert_bit_exact(g0.cpu().numpy(), ref0, what + " (interpreted)")


ALL_LAYOUTS = LAYOUTS + [("i420", capi.YUV_I420), ("yv12", capi.YUV_YV12)]  # planar chroma: whole surfaces only (no crop views)


@pytest.mark.parametrize("lname,layout", ALL_LAYOUTS, ids=[x[0] for x in ALL_LAYOUTS])
@pytest.mark.parametrize("w,h", [(2, 2), (6, 4), (254, 6), (258, 2), (3840, 8)])
def test_nv12_pointwise_sizes(oracle, lname, layout, w, h):
    """Whole surfaces of awkward sizes (one thread with 2 pixels, one ragged group, exactly full groups) -> packed fp32 RGB."""
    import torch
    dev = torch.device("cuda:0")
    f = cvgs.CV_32FC3
    st = cvgs.CV_16UC1 if layout == capi.YUV_P010 else cvgs.CV_8UC1
    surf = _surface(layout, w, h, 6000 + w)

    def build(wrap, out):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, st, m.data, m.step, owner=m.owner)
        return [cvgs.read_nv12(luma, None, capi.YUV_FULL, capi.BT601, False, layout=layout), cvgs.multiply(f, [0.5, 0.25, 2.0]), cvgs.write(f, out)]

    ref = np.zeros((h, w, 3), np.float32)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, st), cvgs.GpuMat.from_array(ref, f))))
    ts = torch.from_numpy(surf.view(np.int16) if layout == capi.YUV_P010 else surf).to(dev)
    gt = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts, st), cvgs.GpuMat.from_tensor(gt, f))
    assert cvgs.kernel_name(*ops).startswith("pointwise4_")
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "%s %dx%d" % (lname, w, h))


@pytest.mark.parametrize("lname,layout", LAYOUTS, ids=[x[0] for x in LAYOUTS])
@pytest.mark.parametrize("alpha", [False, True])
@pytest.mark.parametrize("prog", ["swap", "none"])
@pytest.mark.parametrize("batched", [False, True])
def test_nv12_to_packed_u8_image(oracle, lname, layout, alpha, prog, batched):
    """NV12 / NV21 / P010 -> [BGR swap, arithmetic] -> convertTo<CV_32FCn, CV_8UCn> -> write: the plain decode-side cvtColor into a
    packed 8-bit image (one surface into a pitched image; crops of a surface + a default-value plane into a dense batch).  The
    trailing SaturateCast is the store's conversion; values overshoot [0, 255] on purpose (limited-range surfaces with random bytes)."""
    import torch
    dev = torch.device("cuda:0")
    w, h = 1038, 46
    cn = 4 if alpha else 3
    f, u8 = cvgs.make_type(cvgs.CV_32F, cn), cvgs.make_type(cvgs.CV_8U, cn)
    st = cvgs.CV_16UC1 if layout == capi.YUV_P010 else cvgs.CV_8UC1
    surf = _surface(layout, w, h, 7000 + 13 * layout)
    rects = [(0, 0, 518, 30), (520, 16, 518, 30)]
    n = len(rects) + 1 if batched else 1
    ow, oh = (518, 30) if batched else (w, h)
    scale = 255.0 / 1023.0 if layout == capi.YUV_P010 else 1.0

    def build(wrap, out):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, st, m.data, m.step, owner=m.owner)
        if batched:
            rd = cvgs.read_nv12([luma.nv12_roi(*r) for r in rects] + [luma.nv12_roi(*rects[0])], None, capi.YUV_LIMITED, capi.BT601, alpha, layout=layout)
            rd.used_planes = len(rects)
            rd.background = cvgs._scalar([300.0, -8.0, 7.5, 6.5][:cn])
        else:
            rd = cvgs.read_nv12(luma, None, capi.YUV_LIMITED, capi.BT601, alpha, layout=layout)
        ops = [rd] + _program(prog, f, cn) + [cvgs.convertTo(f, u8, scale)]
        return ops + [cvgs.write(u8, out, (ow, oh)) if batched else cvgs.write(u8, out)]

    shape = (n, ow * oh, cn) if batched else (oh, ow, cn)
    ref = np.zeros(shape, np.uint8)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, st), cvgs.GpuMat.from_array(ref, u8))))
    ts = torch.from_numpy(surf.view(np.int16) if layout == capi.YUV_P010 else surf).to(dev)
    gt = torch.zeros(shape, dtype=torch.uint8, device=dev)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts, st), cvgs.GpuMat.from_tensor(gt, u8))
    name = cvgs.kernel_name(*ops)
    assert name == ("pointwise4_p010_u8" if layout == capi.YUV_P010 else "pointwise4_nv12_u8"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    assert ref.min() == 0 and ref.max() == (255 if prog == "none" else ref.max()) and ref.max() > 100  # the clamps are exercised
    H.assert_bit_exact(gt.cpu().numpy(), ref, "%s alpha=%s %s batched=%s via %s" % (lname, alpha, prog, batched, name))
    gt.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "interpreted")


@pytest.mark.parametrize("lname,layout", LAYOUTS, ids=[x[0] for x in LAYOUTS])
@pytest.mark.parametrize("out", ["packed", "split"])
def test_nv12_to_half_precision_tensor(oracle, lname, layout, out):
    """The half-precision hand-off behind a 4:2:0 read: ... -> normalize -> convertTo<CV_32FC3, CV_16FC3> -> tensor / image."""
    import torch
    dev = torch.device("cuda:0")
    w, h = 774, 20
    f, hf = cvgs.CV_32FC3, cvgs.CV_16FC3
    st = cvgs.CV_16UC1 if layout == capi.YUV_P010 else cvgs.CV_8UC1
    surf = _surface(layout, w, h, 8000 + layout)
    full = 1023.0 if layout == capi.YUV_P010 else 255.0

    def build(wrap, out_mat_of):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, st, m.data, m.step, owner=m.owner)
        ops = [cvgs.read_nv12(luma, None, capi.YUV_LIMITED, capi.BT709, False, layout=layout), cvgs.multiply(f, [1 / full] * 3),
               cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]), cvgs.convertTo(f, hf)]
        if out == "packed":
            return ops + [cvgs.write(hf, out_mat_of(hf))]
        return ops + [cvgs.split(hf, out_mat_of(cvgs.CV_16FC1), (w, h))]

    shape = (h, w, 3) if out == "packed" else (1, 3 * w * h)
    ref = np.zeros(shape, np.float16)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, st), lambda t: cvgs.GpuMat.from_array(ref, t))))
    ts = torch.from_numpy(surf.view(np.int16) if layout == capi.YUV_P010 else surf).to(dev)
    gt = torch.zeros(shape, dtype=torch.float16, device=dev)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts, st), lambda t: cvgs.GpuMat.from_tensor(gt, t))
    name = cvgs.kernel_name(*ops)
    assert name == ("pointwise4_p010_f16" if layout == capi.YUV_P010 else "pointwise4_nv12_f16"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    assert ref.any()
    H.assert_bit_exact(gt.cpu().numpy().view(np.uint16), ref.view(np.uint16), "%s %s via %s" % (lname, out, name))
    gt.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(gt.cpu().numpy().view(np.uint16), ref.view(np.uint16), "interpreted")


@pytest.mark.parametrize("lname,layout", [("i420", capi.YUV_I420), ("yv12", capi.YUV_YV12)])
@pytest.mark.parametrize("out", ["u8", "f16", "split", "planes"])
@pytest.mark.parametrize("alpha", [False, True])
def test_planar_chroma_surfaces_on_the_thread_fused_kernel(oracle, lname, layout, out, alpha):
    """I420 / YV12 (software decoders' yuv420p): two quarter-size chroma planes behind the luma, rows of step / 2 bytes -- a batch of
    two surfaces and a default-value plane through every output form the kernel serves, pitched surfaces included."""
    import torch
    dev = torch.device("cuda:0")
    w, h, pitch = 778, 22, 800  # pitched: the chroma planes use rows of pitch / 2 bytes
    cn = 4 if alpha else 3
    f = cvgs.make_type(cvgs.CV_32F, cn)
    surfs = [H.random_u8((h * 3 // 2, pitch), 9500 + i + layout) for i in range(2)]
    n = 3
    ot = {"u8": cvgs.make_type(cvgs.CV_8U, cn), "f16": cvgs.make_type(cvgs.CV_16F, cn), "split": f, "planes": f}[out]
    np_dt = {"u8": np.uint8, "f16": np.float16}.get(out, np.float32)
    t_dt = {"u8": torch.uint8, "f16": torch.float16}.get(out, torch.float32)

    def build(wrap, wrap_out, outs):
        mats = []
        for sf in surfs + surfs[:1]:
            m = wrap(sf)
            mats.append(cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, pitch, owner=m.owner))
        rd = cvgs.read_nv12(mats, None, capi.YUV_LIMITED, capi.BT601, alpha, layout=layout)
        rd.used_planes = 2
        rd.background = cvgs._scalar([20.0, 130.5, 250.0, 7.0][:cn])
        ops = [rd, cvgs.multiply(f, [1.1, 0.9, 1.05, 0.5][:cn]), cvgs.add(f, [-3.0, 2.0, 0.5, 1.0][:cn])]
        if out in ("u8", "f16"):
            return ops + [cvgs.convertTo(f, ot), cvgs.write(ot, wrap_out(outs[0], ot), (w, h))]
        if out == "split":
            return ops + [cvgs.split(f, wrap_out(outs[0], cvgs.CV_32FC1), (w, h))]
        return ops + [cvgs.split(f, [[wrap_out(p, cvgs.CV_32FC1) for p in outs[1 + i * cn:1 + (i + 1) * cn]] for i in range(n)])]

    shape = (n, w * h, cn) if out in ("u8", "f16") else ((n, cn * w * h) if out == "split" else (1, 1))
    ref0 = np.zeros(shape, np_dt)
    planes_np = [np.zeros((h, w), np.float32) for _ in range(n * cn)]
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), lambda a, t: cvgs.GpuMat.from_array(a, t), [ref0] + planes_np)))
    ts = {id(sf): torch.from_numpy(sf).to(dev) for sf in surfs}
    g0 = torch.zeros(shape, dtype=t_dt, device=dev)
    planes_t = [torch.zeros((h, w), dtype=torch.float32, device=dev) for _ in range(n * cn)]
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts[id(a)], cvgs.CV_8UC1), lambda a, t: cvgs.GpuMat.from_tensor(a, t), [g0] + planes_t)
    name = cvgs.kernel_name(*ops)
    assert name == {"u8": "pointwise4_i420_u8", "f16": "pointwise4_i420_f16"}.get(out, "pointwise4_i420"), name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    what = "%s %s alpha=%s via %s" % (lname, out, alpha, name)
    if out == "planes":
        assert any(p.any() for p in planes_np)
        for i, (a, b) in enumerate(zip(planes_t, planes_np)):
            H.assert_bit_exact(a.cpu().numpy(), b, what + " plane %d" % i)
        return
    assert ref0.any()
    view = (lambda x: x.view(np.uint16)) if out == "f16" else (lambda x: x)
    H.assert_bit_exact(view(g0.cpu().numpy()), view(ref0), what)
    g0.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(view(g0.cpu().numpy()), view(ref0), what + " (interpreted)")
