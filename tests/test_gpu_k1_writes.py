"""The K1 kernel's other write stages (packed pixels, separate pitched planes), used by the reference's single-image
resize chains: tests/resize/test_resize_write.cu (resize -> convertTo<32F, 8U> -> write) and test_resize_x_split.cu
(resize -> multiply -> subtract -> divide -> split(vector<GpuMat>)).  Non-constant images, bit-exact vs the oracle, and
identical to the interpreted kernel."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests.test_gpu_chains import _both

pytestmark = pytest.mark.gpu


def _name(build):
    """kernel the engine picks for the chain `build` makes (on throw-away device buffers)."""
    import torch
    keep = []

    def wrap(a, cvt):
        t = torch.from_numpy(a).cuda()
        keep.append(t)
        return cvgs.GpuMat.from_tensor(t, cvt)

    return cvgs.kernel_name(*build(wrap, wrap, None))


@pytest.mark.parametrize("cn", [3, 4])
@pytest.mark.parametrize("dst", [(640, 360), (1500, 901), (37, 53), (2003, 2100)])  # the last one: 4 rows per wave, shuffled u8c3 stores
def test_resize_to_packed_u8(cn, dst):
    """K3: whole-frame resize (down, up, tiny) back to packed u8 in a PITCHED image."""
    src = H.random_u8((540, 961, cn), 300 + cn)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    pitch_w = dst[0] + 5  # the output is a view of a wider buffer -> pitched rows

    def build(wrap, wrap_out, out):
        o = wrap_out(np.zeros((dst[1], pitch_w, cn), np.uint8) if out is None else out, u)
        return [cvgs.resize(u, cvgs.INTER_LINEAR, wrap(src, u), dst), cvgs.convertTo(f, u), cvgs.write(u, o.roi(2, 0, dst[0], dst[1]))]

    gpu, ref = _both(build, (dst[1], pitch_w, cn), np.uint8)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> packed u8")
    assert not ref[0][:, :2].any() and not ref[0][:, dst[0] + 2:].any() and ref[0][:, 2:dst[0] + 2].std() > 10
    # up-scaled whole frames with dword-aligned rows (C4 here; the C3 view starts 6 bytes in) take the 4-pixels-per-lane form
    assert _name(build) in ("k1_u8c%d_packed_u8" % cn, "k1_u8c%d_packed_u8_x4" % cn)
    gen, _ = _both(build, (dst[1], pitch_w, cn), np.uint8, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gen[0], gpu[0], "interpreted kernel agrees")


@pytest.mark.parametrize("cn,half", [(3, False), (4, False), (3, True)])
def test_batch_resize_to_packed_float(cn, half):
    """N crops -> packed float pixels, dense [plane][y][x] (write<O>(GpuMat, Size)), unused planes and PRESERVE_AR padding."""
    src = H.random_u8((400, 600, cn), 310 + cn)
    crops = H.random_crops(7, 600, 400, seed=3, wmin=3, wmax=400, hmin=3, hmax=300)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    ot = cvgs.make_type(cvgs.CV_16F, cn) if half else f
    dst, n, used = (80, 48), 7, 5

    def build(wrap, wrap_out, out):
        frame = wrap(src, u)
        o = wrap_out(np.zeros((n, dst[0] * dst[1], cn), np.float16 if half else np.float32) if out is None else out, ot)
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, used, [9.0, 8.0, 7.0, 6.0][:cn], cvgs.PRESERVE_AR),
               cvgs.multiply(f, [0.25] * cn), cvgs.add(f, [1.5] * cn)]
        if half:
            ops.append(cvgs.convertTo(f, ot))
        return ops + [cvgs.write(ot, o, dst)]

    gpu, ref = _both(build, (n, dst[0] * dst[1], cn), np.float16 if half else np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "batch resize -> packed float")
    assert _name(build) == "k1_u8c%d_packed_%s_arith" % (cn, "f16" if half else "f32")  # (multiply, add: the canonical arithmetic program, round 6)


@pytest.mark.parametrize("cn,batch", [(3, 1), (4, 2), (3, 9)])
def test_resize_to_separate_planes(cn, batch):
    """K2: resize -> multiply -> subtract -> divide -> split into C pitched GpuMats per image (batch 9 x 3 = 27 planes:
    more than the 16 that travel in the kernel arguments -> uploaded table)."""
    src = H.random_u8((300, 500, cn), 320 + cn)
    crops = H.random_crops(batch, 500, 300, seed=5 + batch, wmin=8, wmax=300, hmin=8, hmax=250)
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    dst = (100, 60)
    pitch_w = dst[0] + 3

    def build(wrap, wrap_out, out):
        frame = wrap(src, u)
        o = wrap_out(np.zeros((batch * cn * dst[1], pitch_w), np.float32) if out is None else out, cvgs.CV_32FC1)
        planes = [[cvgs.GpuMat(dst[1], dst[0], cvgs.CV_32FC1, o.data + ((z * cn + c) * dst[1]) * o.step, o.step, owner=o)
                   for c in range(cn)] for z in range(batch)]
        if batch > 1:
            rd = cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, batch)
        else:
            rd = cvgs.resize(u, cvgs.INTER_LINEAR, frame.roi(*crops[0]), dst)
        return [rd, cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]),
                cvgs.split(f, planes if batch > 1 else planes[0])]

    gpu, ref = _both(build, (batch * cn * dst[1], pitch_w), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> separate planes")
    assert not ref[0][:, dst[0]:].any()
    assert _name(build) == "k1_u8c%d_planes2d_f32" % cn


def test_gray_output_of_a_resize_is_packed_single_channel():
    """a channel-count change inside the chain (RGB2GRAY after the resize) reaches the packed store with 1 channel."""
    src = H.random_u8((200, 300, 3), 9)
    f3, f1 = cvgs.CV_32FC3, cvgs.CV_32FC1

    def build(wrap, wrap_out, out):
        o = wrap_out(np.zeros((90, 120, 1), np.float32) if out is None else out, f1)
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, wrap(src, cvgs.CV_8UC3), (120, 90)), cvgs.cvtColor(cvgs.COLOR_RGB2GRAY, f3, f1),
                cvgs.write(f1, o)]

    gpu, ref = _both(build, (90, 120, 1), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> gray packed")


@pytest.mark.parametrize("depth,cn,out", [("8U", 1, "u8"), ("8U", 1, "f32"), ("8U", 2, "u8"), ("8U", 2, "f32"), ("8U", 2, "planar"),
                                          ("16U", 2, "planar"), ("16S", 2, "planar_interp"), ("16U", 1, "planar")])
def test_resize_one_and_two_channel_sources(depth, cn, out):
    """Grayscale / two-channel sources on the fast kernel (the reference's single-image resize tests sweep C1 types,
    tests/resize/test_resize_write.cu:120-123): crops from 1 pixel wide (byte-gathered windows) to wide ones, up- and
    down-scaling, against the oracle and the interpreted kernel."""
    from tests import kat_runner as K
    from tests.test_gpu_chains import _random_src
    src = _random_src((240, 330, cn), depth, 50 + cn)
    crops = [(0, 0, 1, 1), (5, 7, 2, 3), (11, 3, 3, 200), (1, 1, 7, 9), (0, 0, 330, 240), (100, 50, 200, 33), (17, 19, 64, 64)]
    n, dst = len(crops), (48, 40)
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    u = cvgs.make_type(cvgs.CV_8U, cn)

    def build(wrap, wrap_out, out_buf):
        frame = wrap(src, st)
        rd = cvgs.resize(st, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n - 1, [3.0, 9.0][:cn])
        if out == "u8":
            return [rd, cvgs.multiply(f, [0.9] * cn), cvgs.convertTo(f, u), cvgs.write(u, wrap_out(out_buf, u), dst)]
        if out == "f32":
            return [rd, cvgs.add(f, [0.25] * cn), cvgs.write(f, wrap_out(out_buf, f), dst)]
        norm = [cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn])]
        if out == "planar_interp":
            norm.append(cvgs.add(f, [1.0] * cn))
        o = wrap_out(out_buf, cvgs.CV_32FC1)
        return [rd] + norm + [cvgs.split(f, o, dst) if cn > 1 else cvgs.write(f, o, dst)]

    shape, dt = ((n, dst[0] * dst[1], cn), np.uint8) if out == "u8" else (((n, dst[0] * dst[1], cn), np.float32) if out == "f32" else
                                                                           ((n, cn * dst[0] * dst[1]), np.float32))
    gpu, ref = _both(build, shape, dt)
    H.assert_bit_exact(gpu[0], ref[0], "K1 %sC%d -> %s" % (depth, cn, out))
    gen, _ = _both(build, shape, dt, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gen[0], gpu[0], "interpreted kernel agrees")

    def named(wrap, wrap_out, _):
        return build(wrap, wrap_out, np.zeros(shape, dt))

    name = _name(named)
    if depth == "16U" and cn == 1:
        assert name.startswith("generic")  # packed == planar for one channel; 16-bit C1 stays on the interpreted kernel
    else:
        assert name.startswith("k1_%sc%d" % ({"8U": "u8", "16U": "u16", "16S": "s16"}[depth], cn)), name


# ---- the reference's single-image resize tests on 16-bit and CV_32F images (tests/resize/test_resize_write.cu:110-123) ----
@pytest.mark.parametrize("depth,cn", [("16U", 1), ("16U", 3), ("16U", 4), ("16S", 1), ("16S", 3), ("16S", 4), ("32F", 1), ("32F", 3)])
@pytest.mark.parametrize("dst", [(640, 360), (1010, 601), (37, 53)])
def test_resize_write_back_to_the_source_type(depth, cn, dst):
    """resize -> convertTo<CV_32FCn, I> -> write<I> (CV_32F: resize -> write): up, down and tiny, into a PITCHED image; the
    K1 kernel with a packed store of the source's own type, vs the oracle and vs the interpreted kernel."""
    from tests.test_gpu_chains import _random_src
    from tests import kat_runner as K
    src = _random_src((270, 481, cn), depth, 330 + cn)
    if depth == "32F":
        src = (src * 50.0).astype(np.float32)
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    np_dt = K.NP_DEPTH[depth]
    pitch_w = dst[0] + 3

    def build(wrap, wrap_out, out):
        o = wrap_out(np.zeros((dst[1], pitch_w, cn), np_dt) if out is None else out, st)
        ops = [cvgs.resize(st, cvgs.INTER_LINEAR, wrap(src, st), dst)]
        if depth != "32F":
            ops.append(cvgs.convertTo(f, st))
        return ops + [cvgs.write(st, o.roi(1, 0, dst[0], dst[1]))]

    gpu, ref = _both(build, (dst[1], pitch_w, cn), np_dt)
    H.assert_bit_exact(gpu[0], ref[0], "resize -> packed %s" % depth)
    assert not ref[0][:, :1].any() and not ref[0][:, dst[0] + 1:].any() and ref[0][:, 1:dst[0] + 1].astype(np.float64).std() > 10
    assert _name(build) == "k1_%s%sc%d_packed_%s%s" % (depth[-1].lower(), depth[:-1], cn, depth[-1].lower(), depth[:-1]), _name(build)
    gen, _ = _both(build, (dst[1], pitch_w, cn), np_dt, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gen[0], gpu[0], "interpreted kernel agrees")


@pytest.mark.parametrize("depth", ["16U", "16S"])
@pytest.mark.parametrize("cn,batch", [(3, 1), (4, 3)])
def test_resize_16_bit_sources_to_separate_planes(depth, cn, batch):
    """tests/resize/test_resize_x_split.cu on its CV_16U / CV_16S type pairs: resize -> mul -> sub -> div -> split(planes)."""
    from tests.test_gpu_chains import _random_src
    from tests import kat_runner as K
    src = _random_src((300, 500, cn), depth, 340 + cn)
    crops = H.random_crops(batch, 500, 300, seed=15 + batch, wmin=8, wmax=300, hmin=8, hmax=250)
    st, f = cvgs.make_type(K.CV_DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    dst = (64, 128)
    pitch_w = dst[0] + 2

    def build(wrap, wrap_out, out):
        frame = wrap(src, st)
        o = wrap_out(np.zeros((batch * cn * dst[1], pitch_w), np.float32) if out is None else out, cvgs.CV_32FC1)
        planes = [[cvgs.GpuMat(dst[1], dst[0], cvgs.CV_32FC1, o.data + ((z * cn + c) * dst[1]) * o.step, o.step, owner=o)
                   for c in range(cn)] for z in range(batch)]
        rd = (cvgs.resize(st, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, batch) if batch > 1 else
              cvgs.resize(st, cvgs.INTER_LINEAR, frame.roi(*crops[0]), dst))
        return [rd, cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]),
                cvgs.split(f, planes if batch > 1 else planes[0])]

    gpu, ref = _both(build, (batch * cn * dst[1], pitch_w), np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "16-bit resize -> separate planes")
    assert ref[0].any() and not ref[0][:, dst[0]:].any()
    assert _name(build) == "k1_%s16c%d_planes2d_f32" % (depth[-1].lower(), cn), _name(build)
    gen, _ = _both(build, (batch * cn * dst[1], pitch_w), np.float32, flags=capi.CHAIN_FORCE_GENERIC)
    H.assert_bit_exact(gen[0], gpu[0], "interpreted kernel agrees")
