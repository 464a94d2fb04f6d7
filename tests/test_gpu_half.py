"""GPU parity for the half-precision hand-off option (CV_16F, SURVEY.md 8(f)3: "tensor hand-off to an inference
runtime ... NCHW fp32 or fp16 output option"), bit-exact against the CPU oracle.  The reference has no half type;
the oracle's conversion is pinned against IEEE 754 (numpy) in tests/test_half_oracle.py."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests.test_gpu_chains import _both
from tests.test_gpu_circular_nv12 import _read_device

pytestmark = pytest.mark.gpu


def _k1_half_ops(cn, swap=True):
    f, h = cvgs.make_type(cvgs.CV_32F, cn), cvgs.make_type(cvgs.CV_16F, cn)
    u = cvgs.make_type(cvgs.CV_8U, cn)
    ops = []
    if swap:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA, f))
    ops += [cvgs.multiply(f, [0.3] * cn), cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]), cvgs.convertTo(f, h)]
    return u, f, h, ops


@pytest.mark.parametrize("cn,swap,transposed", [(3, True, False), (4, True, False), (3, False, True), (4, False, False)])
def test_k1_half_output(cn, swap, transposed):
    """K1 with an fp16 NCHW / CNHW tensor: same chain as the headline, one extra convertTo<CV_32F, CV_16F> before split."""
    import torch
    src = H.random_u8((500, 700, cn), 40 + cn)
    crops = H.random_crops(13, 700, 500, seed=9 + cn, wmin=1, wmax=600, hmin=1, hmax=450)
    dst, n, used = (64, 128), 13, 11
    u, f, h, ops = _k1_half_ops(cn, swap)

    def build(wrap, wrap_out, out):
        frame = wrap(src, u)
        rd = cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, used, [17.0, 99.5, 3.0, 200.0][:cn])
        o = wrap_out(out, cvgs.CV_16FC1)
        wr = cvgs.splitT(h, o.data, dst[0], dst[1], n, keep=o) if transposed else cvgs.split(h, o, dst)
        return [rd] + ops + [wr]

    gpu, ref = _both(build, (n, cn * 64 * 128), np.float16)
    H.assert_bit_exact(gpu[0], ref[0], "K1 fp16 C%d" % cn)
    assert np.isfinite(ref[0].astype(np.float32)).all() and ref[0].astype(np.float32).std() > 1.0
    # the fast path took it
    t = torch.zeros((500, 700, cn), dtype=torch.uint8, device="cuda:0")
    o = torch.zeros((n, cn * 64 * 128), dtype=torch.float16, device="cuda:0")
    frame = cvgs.GpuMat.from_tensor(t, u)
    chain = [cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n)] + ops + \
            [cvgs.split(h, cvgs.GpuMat.from_tensor(o, cvgs.CV_16FC1), dst)]
    name = cvgs.kernel_name(*chain)
    assert name == "k1_u8c%d_%s_f16" % (cn, "swap_mul_sub_div" if swap else "mul_sub_div"), name


def test_k1_half_preserve_ar_and_table():
    """PRESERVE_AR padding + more planes than fit the kernel-argument block (device descriptor table), fp16 output."""
    src = H.random_u8((300, 400, 3), 3)
    n = 70
    crops = H.random_crops(n, 400, 300, seed=5, wmin=2, wmax=200, hmin=2, hmax=280)
    dst = (48, 40)
    u, f, h, ops = _k1_half_ops(3)

    def build(wrap, wrap_out, out):
        frame = wrap(src, u)
        rd = cvgs.resize(u, cvgs.INTER_LINEAR, [frame.roi(*c) for c in crops], dst, n, [128.0, 64.0, 32.0], cvgs.PRESERVE_AR)
        return [rd] + ops + [cvgs.split(h, wrap_out(out, cvgs.CV_16FC1), dst)]

    gpu, ref = _both(build, (n, 3 * 48 * 40), np.float16)
    H.assert_bit_exact(gpu[0], ref[0], "K1 fp16 PRESERVE_AR, 70 planes")


def test_half_conversion_special_values():
    """convertTo<CV_32F, CV_16F> on every tie / overflow / subnormal class; 16F -> 32F back is exact."""
    allh = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    s = np.unique(allh[np.isfinite(allh)])
    mid = ((s[:-1].astype(np.float64) + s[1:].astype(np.float64)) / 2).astype(np.float32)
    vals = np.concatenate([s, mid, np.nextafter(mid, np.float32(np.inf)), np.nextafter(mid, np.float32(-np.inf)),
                           np.array([65519.996, 65520, 1e10, -1e10, np.inf, -np.inf, 2.0 ** -25, 0.0, -0.0], np.float32)])
    n = (vals.size // 128) * 128
    src = vals[:n].reshape(-1, 128, 1).copy()

    def build(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [wrap(src, cvgs.CV_32FC1)], 1),
                cvgs.convertTo(cvgs.CV_32FC1, cvgs.CV_16FC1), cvgs.write(cvgs.CV_16FC1, wrap_out(out, cvgs.CV_16FC1))]

    gpu, ref = _both(build, src.shape, np.float16)
    H.assert_bit_exact(gpu[0], ref[0], "fp32 -> fp16")
    half_src = ref[0].copy()

    def back(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_16FC1, [wrap(half_src, cvgs.CV_16FC1)], 1),
                cvgs.convertTo(cvgs.CV_16FC1, cvgs.CV_32FC1), cvgs.multiply(cvgs.CV_32FC1, [3.0]),
                cvgs.write(cvgs.CV_32FC1, wrap_out(out, cvgs.CV_32FC1))]

    gpu, ref = _both(back, src.shape, np.float32)
    H.assert_bit_exact(gpu[0], ref[0], "fp16 source -> fp32")


@pytest.mark.parametrize("src_depth,cn", [("8U", 3), ("16U", 4), ("16S", 1), ("32S", 2)])
def test_pointwise_to_half(src_depth, cn):
    """integer sources -> convertTo<I, CV_16F>(alpha, beta): computed in fp32, rounded once; packed and planar writes."""
    from tests import kat_runner as K
    from tests.test_gpu_chains import _random_src
    a = _random_src((45, 67, cn), src_depth, 21)
    st, ht = cvgs.make_type(K.CV_DEPTH[src_depth], cn), cvgs.make_type(cvgs.CV_16F, cn)
    if src_depth == "8U":  # wide rows too: full 256-pixel groups go through the LDS-transposed packed store
        wide = _random_src((9, 555, cn), src_depth, 22)

        def packed_wide(wrap, wrap_out, out):
            return [cvgs.ReadIOp(capi.READ_PIXEL, st, [wrap(wide, st)], 1), cvgs.convertTo(st, ht, 1.0 / 255.0, -0.25),
                    cvgs.write(ht, wrap_out(out, ht))]

        gpu, ref = _both(packed_wide, (9, 555, cn), np.float16)
        H.assert_bit_exact(gpu[0], ref[0], "pointwise -> fp16 packed, wide")

    def packed(wrap, wrap_out, out):
        return [cvgs.ReadIOp(capi.READ_PIXEL, st, [wrap(a, st)], 1), cvgs.convertTo(st, ht, 1.0 / 255.0, -0.25),
                cvgs.write(ht, wrap_out(out, ht))]

    gpu, ref = _both(packed, (45, 67, cn), np.float16)
    H.assert_bit_exact(gpu[0], ref[0], "pointwise -> fp16 packed")

    def planar(wrap, wrap_out, out):
        o = wrap_out(out, cvgs.CV_16FC1)
        return [cvgs.ReadIOp(capi.READ_PIXEL, st, [wrap(a, st)], 1), cvgs.convertTo(st, ht, 1.0 / 255.0, -0.25),
                cvgs.split_tensor(ht, o.data, 67, 45, 1, keep=o)]

    if cn > 1:
        gpu, ref = _both(planar, (1, cn * 45 * 67), np.float16)
        H.assert_bit_exact(gpu[0], ref[0], "pointwise -> fp16 planar")


def test_circular_tensor_half(oracle):
    """CircularTensor with an fp16 element type: half the bytes per update; ordering and content vs the oracle."""
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 80, 46, 4
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_16FC1, 3, B, cvgs.NewestFirst, cvgs.Standard, W, H_)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_16FC1, 3, B, cvgs.NewestFirst, cvgs.Standard)
    f, h = cvgs.CV_32FC3, cvgs.CV_16FC3
    s = torch.cuda.current_stream()
    for i in range(2 * B + 1):
        frame = H.random_u8((H_, W, 3), seed=300 + i)
        frame_t = torch.from_numpy(frame).to(dev)
        pw = [cvgs.convertTo(cvgs.CV_8UC3, f), cvgs.multiply(f, [1.0 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]),
              cvgs.divide(f, [0.229, 0.224, 0.225]), cvgs.convertTo(f, h)]
        ct.update(s, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), *pw, ct.write_split(h))
        oc.update(cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1),
                              *pw, cvgs.WriteIOp(capi.WRITE_TENSOR_SPLIT, h, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float16)
        H.assert_bit_exact(got, oc.array(np.float16), "fp16 circular update %d" % i)
    assert ct.nbytes() == B * 3 * W * H_ * 2
    ct.release()


def test_half_is_storage_only():
    """Arithmetic on CV_16F values is refused loudly; a CV_16F resize source and CV_16F next to CV_64F values are served
    since round 2 (bit-exact vs the oracle)."""
    import torch
    t = torch.zeros((8, 8, 3), dtype=torch.float16, device="cuda:0")
    o = torch.zeros((8, 8, 3), dtype=torch.float16, device="cuda:0")
    h, f, d = cvgs.CV_16FC3, cvgs.CV_32FC3, cvgs.CV_64FC3
    m, om = cvgs.GpuMat.from_tensor(t, h), cvgs.GpuMat.from_tensor(o, h)
    s = torch.cuda.current_stream()
    with pytest.raises(capi.CvgsError, match="arithmetic"):
        cvgs.executeOperations(s, cvgs.ReadIOp(capi.READ_PIXEL, h, [m], 1), cvgs.multiply(h, [2.0] * 3), cvgs.write(h, om))
    from oracle import oracle_binding as ob
    src = (H.random_u8((8, 8, 3), seed=3).astype(np.float32) / 7.0).astype(np.float16)
    t.copy_(torch.from_numpy(src))
    of = torch.zeros((5, 6, 3), dtype=torch.float32, device="cuda:0")
    cvgs.executeOperations(s, cvgs.resize(h, cvgs.INTER_LINEAR, m, (6, 5)), cvgs.write(f, cvgs.GpuMat.from_tensor(of, f)))
    torch.cuda.synchronize()
    rf = np.zeros((5, 6, 3), np.float32)
    ob.execute(cvgs.lower([cvgs.resize(h, cvgs.INTER_LINEAR, cvgs.GpuMat.from_array(src, h), (6, 5)), cvgs.write(f, cvgs.GpuMat.from_array(rf, f))]))
    H.assert_bit_exact(of.cpu().numpy(), rf, "resize of a CV_16F source")

    def chain(mm, oo):
        return [cvgs.ReadIOp(capi.READ_PIXEL, h, [mm], 1), cvgs.convertTo(h, d), cvgs.multiply(d, [1.0 / 3.0, 0.1, 7.0]),
                cvgs.add(d, [1e-3, 2.5, -4.0]), cvgs.convertTo(d, h), cvgs.write(h, oo)]

    cvgs.executeOperations(s, *chain(m, om))
    torch.cuda.synchronize()
    ref = np.zeros((8, 8, 3), np.float16)
    ob.execute(cvgs.lower(chain(cvgs.GpuMat.from_array(src, h), cvgs.GpuMat.from_array(ref, h))))
    H.assert_bit_exact(o.cpu().numpy(), ref, "half -> double -> half")
    assert ref.any()


def test_warp_and_nv12_half_tensors(oracle):
    """The fp16 hand-off on the other two fast read kinds: N warped faces and N crops of an NV12 surface -> fp16 NCHW."""
    import torch
    from tests import warp_cases as WC
    from tests.test_gpu_circular_nv12 import _nv12
    f, h = cvgs.CV_32FC3, cvgs.CV_16FC3
    # warp
    src = H.random_u8((300, 400, 3), 77)
    n, dst = 5, (112, 112)
    ms = [[[0.4 + 0.1 * i, 0.05 * i, -10.0 * i], [-0.03 * i, 0.5, 7.0]] for i in range(n)]

    def build(wrap, wrap_out, out):
        img = wrap(src, cvgs.CV_8UC3)
        return [cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [img] * n, ms, dst), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]),
                cvgs.divide(f, H.K1_DIV[3]), cvgs.convertTo(f, h), cvgs.split(h, wrap_out(out, cvgs.CV_16FC1), dst)]

    gpu, ref = _both(build, (n, 3 * dst[0] * dst[1]), np.float16)
    H.assert_bit_exact(gpu[0], ref[0], "warp -> fp16 NCHW")
    t = torch.from_numpy(src).cuda()
    o = torch.zeros((n, 3 * dst[0] * dst[1]), dtype=torch.float16, device="cuda")
    ops = [cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [cvgs.GpuMat.from_tensor(t, cvgs.CV_8UC3)] * n, ms, dst), cvgs.multiply(f, [0.3] * 3),
           cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]), cvgs.convertTo(f, h),
           cvgs.split(h, cvgs.GpuMat.from_tensor(o, cvgs.CV_16FC1), dst)]
    assert cvgs.kernel_name(*ops) == "warp_affine_u8c3_mul_sub_div_f16"
    # NV12 crops
    w, hh, d2 = 640, 360, (64, 128)
    buf = _nv12(w, hh, 99)
    rects = [(0, 0, 640, 360), (10, 20, 100, 200), (300, 100, 64, 128), (2, 2, 8, 8)]

    def chain(luma, out):
        return [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], d2, capi.YUV_FULL, capi.BT601, False), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3]), cvgs.convertTo(f, h),
                cvgs.split(h, out, d2)]

    ref = np.zeros((4, 3 * d2[0] * d2[1]), np.float16)
    m = cvgs.GpuMat.from_array(buf, cvgs.CV_8UC1)
    oracle.execute(cvgs.lower(chain(cvgs.GpuMat(hh, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner), cvgs.GpuMat.from_array(ref, cvgs.CV_16FC1))))
    tb = torch.from_numpy(buf).cuda()
    ob = torch.zeros((4, 3 * d2[0] * d2[1]), dtype=torch.float16, device="cuda")
    md = cvgs.GpuMat.from_tensor(tb, cvgs.CV_8UC1)
    ops = chain(cvgs.GpuMat(hh, w, cvgs.CV_8UC1, md.data, md.step, owner=md.owner), cvgs.GpuMat.from_tensor(ob, cvgs.CV_16FC1))
    assert cvgs.kernel_name(*ops) == "k4_nv12_resize_swap_mul_sub_div_f16"
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    H.assert_bit_exact(ob.cpu().numpy(), ref, "NV12 crops -> fp16 NCHW")
