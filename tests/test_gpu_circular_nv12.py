"""GPU parity for the CircularTensor (K9/K10) and the NV12 read-back chain (K4), through the C-ABI."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests import kat_runner as K

pytestmark = pytest.mark.gpu

CIRC = [c for c in K.load_cases() if c.get("kind") == "circular"]


def _write_iop(ct, name, ftype):
    return {"tensor_split": ct.write_split, "tensor_t_split": ct.write_splitT, "tensor_write": ct.write_packed}[name](ftype)


@pytest.mark.parametrize("case", CIRC, ids=[c["name"] for c in CIRC])
def test_circular_tensor_reference_kat(case):
    """After 100 updates with value i+1 slot z holds 100 - age(z) (reference
    tests/batchread/test_circularbatchread_x_write3D.cu:263-279,324-337,382-395,440-457)."""
    import torch
    dev = torch.device("cuda:0")
    _, icn, itype = K.parse_type(case["in_type"])
    ed, ecn, etype = K.parse_type(case["elem_type"])
    order = cvgs.NewestFirst if case["order"] == "NewestFirst" else cvgs.OldestFirst
    mode = cvgs.Transposed if case["mode"] == "Transposed" else cvgs.Standard
    W, H_, B, CP = case["width"], case["height"], case["batch"], case["color_planes"]
    ftype = cvgs.make_type(cvgs.CV_32F, icn)
    ct = cvgs.CircularTensor(itype, etype, CP, B, order, mode, W, H_)
    frame = torch.zeros((H_, W, icn), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream()
    for i in range(case["iters"]):
        frame.fill_(i + 1)
        ct.update(s, cvgs.GpuMat.from_tensor(frame, itype), cvgs.convertTo(itype, ftype), _write_iop(ct, case["write"], ftype))
    torch.cuda.synchronize()
    assert ct.updates() == case["iters"]
    n = ct.nbytes() // 4
    import ctypes as C
    host = np.empty(n, np.float32)
    out_t = torch.empty(n, dtype=torch.float32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(out_t.data_ptr()), C.c_void_p(ct.data()), C.c_size_t(n * 4), 3) == 0  # D2D
    host = out_t.cpu().numpy()
    exp = np.asarray(case["expected_slot"], np.float32)
    if case["write"] == "tensor_t_split":
        assert (host.reshape(CP, B, H_, W) == exp[None, :, None, None]).all()
    elif case["write"] == "tensor_split":
        assert (host.reshape(B, CP, H_, W) == exp[:, None, None, None]).all()
    else:
        assert (host.reshape(B, H_, W, ecn) == exp[:, None, None, None]).all()
    ct.release()


def _read_device(ptr, nbytes):
    import ctypes as C
    import torch
    t = torch.empty(nbytes, dtype=torch.uint8, device="cuda:0")
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(t.data_ptr()), C.c_void_p(ptr), C.c_size_t(nbytes), 3) == 0
    return t.cpu().numpy()


@pytest.mark.parametrize("order,mode", [(cvgs.NewestFirst, cvgs.Standard), (cvgs.OldestFirst, cvgs.Standard),
                                        (cvgs.NewestFirst, cvgs.Transposed), (cvgs.OldestFirst, cvgs.Transposed)])
def test_circular_tensor_resize_normalize_vs_oracle(oracle, order, mode):
    """cfg #4 shape in small: every update pushes a NEW random frame through resize + normalize (the
    update(stream, readIOp, ops..., write) form, reference include/cvGPUSpeedup.cuh:619-622); the whole tensor is
    compared with the oracle after every update, including the not-yet-filled slots."""
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 96, 54, 5
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, order, mode, W, H_)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, B, order, mode)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    for i in range(2 * B + 3):
        frame = H.random_u8((216, 384, 3), seed=1000 + i)
        frame_t = torch.from_numpy(frame).to(dev)
        pw = [cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]), cvgs.divide(f, H.K1_DIV[3])]
        wr_g = ct.write_splitT(f) if mode == cvgs.Transposed else ct.write_split(f)
        ct.update(s, cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), (W, H_)),
                  *pw, wr_g)
        kind = capi.WRITE_TENSOR_T_SPLIT if mode == cvgs.Transposed else capi.WRITE_TENSOR_SPLIT
        oc.update(cvgs.lower([cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), (W, H_)),
                              *pw, cvgs.WriteIOp(kind, f, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), "circular update %d" % i)
    ct.release()


def test_circular_batch_read_rotation():
    """fk::CircularBatchRead<Ascendent>: out[z] = in[(z + first) % BATCH] (reference :24-87)."""
    import torch
    dev = torch.device("cuda:0")
    case = [c for c in K.load_cases() if c["name"] == "circular_batch_read"][0]
    B, first, W, H_ = case["batch"], case["first"], case["width"], case["height"]
    planes = [torch.full((H_, W, 3), i, dtype=torch.uint8, device=dev) for i in range(B)]
    mats = [cvgs.GpuMat.from_tensor(p, cvgs.CV_8UC3) for p in planes]
    out = torch.zeros((B, W * H_, 3), dtype=torch.uint8, device=dev)
    cvgs.executeOperations(torch.cuda.current_stream(),
                           cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [mats[(z + first) % B] for z in range(B)], B),
                           cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(out, cvgs.CV_8UC3), (W, H_)))
    torch.cuda.synchronize()
    exp = np.asarray(case["expected_plane"], np.uint8)
    assert (out.cpu().numpy().reshape(B, H_, W, 3) == exp[:, None, None, None]).all()


def _nv12(w, h, seed):
    return H.random_u8((h + h // 2, w, 1), seed)


@pytest.mark.parametrize("rng,prim,alpha", [(capi.YUV_FULL, capi.BT709, True), (capi.YUV_FULL, capi.BT601, True),
                                             (capi.YUV_LIMITED, capi.BT709, False), (capi.YUV_LIMITED, capi.BT601, False)])
def test_nv12_resize_chain(oracle, rng, prim, alpha):
    """K4: Resize<LINEAR>(ReadYUV<NV12> + ConvertYUVToRGB) -> SaturateCast -> VectorReorder<2,1,0,3> -> write
    (reference tests/resize/test_fused_resize.cu:141-147) and the float/normalize/split variant of cfg #3."""
    import torch
    dev = torch.device("cuda:0")
    w, h, dst = 640, 360, (213, 120)
    buf = _nv12(w, h, 77)
    cn = 4 if alpha else 3
    f, u = cvgs.make_type(cvgs.CV_32F, cn), cvgs.make_type(cvgs.CV_8U, cn)

    def chains(wrap, out8, outf):
        luma = wrap(buf)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, luma.data, luma.step, owner=luma.owner)
        code = cvgs.COLOR_RGBA2BGRA if alpha else cvgs.COLOR_RGB2BGR
        a = [cvgs.read_nv12(luma, dst, rng, prim, alpha), cvgs.convertTo(f, u), cvgs.cvtColor(code, u), cvgs.write(u, out8)]
        b = [cvgs.read_nv12(luma, dst, rng, prim, alpha), cvgs.cvtColor(code, f), cvgs.multiply(f, [0.3] * cn),
             cvgs.subtract(f, H.K1_SUB[cn]), cvgs.divide(f, H.K1_DIV[cn]), cvgs.split(f, outf, dst)]
        return a, b

    t_buf = torch.from_numpy(buf).to(dev)
    g8 = torch.zeros((dst[1], dst[0], cn), dtype=torch.uint8, device=dev)
    gf = torch.zeros((1, cn * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    a, b = chains(lambda x: cvgs.GpuMat.from_tensor(t_buf, cvgs.CV_8UC1), cvgs.GpuMat.from_tensor(g8, u),
                     cvgs.GpuMat.from_tensor(gf, cvgs.CV_32FC1))
    s = torch.cuda.current_stream()
    cvgs.executeOperations(s, *a)
    cvgs.executeOperations(s, *b)
    torch.cuda.synchronize()
    r8 = np.zeros((dst[1], dst[0], cn), np.uint8)
    rf = np.zeros((1, cn * dst[0] * dst[1]), np.float32)
    a, b = chains(lambda x: cvgs.GpuMat.from_array(buf, cvgs.CV_8UC1), cvgs.GpuMat.from_array(r8, u),
                     cvgs.GpuMat.from_array(rf, cvgs.CV_32FC1))
    oracle.execute(cvgs.lower(a))
    oracle.execute(cvgs.lower(b))
    H.assert_bit_exact(g8.cpu().numpy(), r8, "NV12 resize -> u8")
    H.assert_bit_exact(gf.cpu().numpy(), rf, "NV12 resize -> normalize -> split")


def test_nv12_full_resolution_convert(oracle):
    """ReadYUV<NV12> -> ConvertYUVToRGB -> SaturateCast -> write at full resolution (reference :50-53)."""
    import torch
    dev = torch.device("cuda:0")
    w, h = 256, 128
    buf = _nv12(w, h, 99)
    f, u = cvgs.CV_32FC4, cvgs.CV_8UC4
    t_buf = torch.from_numpy(buf).to(dev)
    g = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
    gm = cvgs.GpuMat.from_tensor(t_buf, cvgs.CV_8UC1)
    cvgs.executeOperations(torch.cuda.current_stream(),
                           cvgs.read_nv12(cvgs.GpuMat(h, w, cvgs.CV_8UC1, gm.data, gm.step, owner=t_buf), None,
                                          capi.YUV_FULL, capi.BT601, True),
                           cvgs.convertTo(f, u), cvgs.write(u, cvgs.GpuMat.from_tensor(g, u)))
    torch.cuda.synchronize()
    r = np.zeros((h, w, 4), np.uint8)
    hm = cvgs.GpuMat.from_array(buf, cvgs.CV_8UC1)
    oracle.execute(cvgs.lower([cvgs.read_nv12(cvgs.GpuMat(h, w, cvgs.CV_8UC1, hm.data, hm.step, owner=buf), None,
                                              capi.YUV_FULL, capi.BT601, True),
                               cvgs.convertTo(f, u), cvgs.write(u, cvgs.GpuMat.from_array(r, u))]))
    H.assert_bit_exact(g.cpu().numpy(), r, "NV12 full-res")
    assert (r[..., 3] == 255).all()


@pytest.mark.parametrize("order", [cvgs.NewestFirst, cvgs.OldestFirst])
@pytest.mark.parametrize("write", ["split", "packed"])
def test_mirrored_circular_tensor_matches_default_semantics(oracle, order, write):
    """The opt-in mirrored-ring CircularTensor (cvgs_circular_create_ex): no shift traffic, data()
    moves -- but the tensor seen at data() after every update is the one the reference semantics (oracle) give."""
    import torch
    dev = torch.device("cuda:0")
    W, H_, B = 72, 40, 5
    packed = write == "packed"
    cp, elem = (1, cvgs.CV_32FC3) if packed else (3, cvgs.CV_32FC1)
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, elem, cp, B, order, cvgs.Standard, W, H_, mirrored=True)
    oc = oracle.OracleCircular(W, H_, elem, cp, B, order, cvgs.Standard)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    seen = set()
    zeros = _read_device(ct.data(), ct.nbytes())
    assert not zeros.any()
    for i in range(3 * B + 2):
        frame = H.random_u8((H_, W, 3), seed=4000 + i)
        frame_t = torch.from_numpy(frame).to(dev)
        pw = [cvgs.convertTo(cvgs.CV_8UC3, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3])]
        wr = ct.write_packed(f) if packed else ct.write_split(f)
        ct.update(s, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), *pw, wr)
        kind = capi.WRITE_PIXEL_3D if packed else capi.WRITE_TENSOR_SPLIT
        oc.update(cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1),
                              *pw, cvgs.WriteIOp(kind, f, 16, W, H_, 0, B)]))
        torch.cuda.synchronize()
        seen.add(ct.data())
        got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
        H.assert_bit_exact(got, oc.array(np.float32), "mirrored circular update %d" % i)
    assert len(seen) == B  # the window really moves: B distinct positions
    ct.release()


def test_mirrored_circular_tensor_rejects_transposed():
    with pytest.raises(capi.CvgsError, match="Standard plane order"):
        cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, 4, cvgs.NewestFirst, cvgs.Transposed, 16, 16, mirrored=True)


def test_circular_update_refuses_stream_capture():
    """The ring index is host state (as in the reference): a captured update would replay into the same slots."""
    import torch
    dev = torch.device("cuda:0")
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, 3, cvgs.NewestFirst, cvgs.Standard, 32, 16)
    frame = torch.zeros((16, 32, 3), dtype=torch.uint8, device=dev)
    f = cvgs.CV_32FC3
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        g.capture_begin()
        try:
            with pytest.raises(capi.CvgsError, match="cannot be captured"):
                ct.update(torch.cuda.current_stream(), cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3), cvgs.convertTo(cvgs.CV_8UC3, f),
                          ct.write_split(f))
            frame.add_(1)  # keep the capture non-empty
        finally:
            g.capture_end()
    torch.cuda.synchronize()
    assert ct.updates() == 0
    ct.release()


@pytest.mark.parametrize("batch", [1, 2, 30])
@pytest.mark.parametrize("mirrored", [False, True])
def test_circular_tensor_depth_extremes(oracle, batch, mirrored):
    """BATCH = 1 (nothing to shift), 2, and 30 (x3 colour planes = 87 plane copies: more than one copy launch holds)."""
    import torch
    dev = torch.device("cuda:0")
    W, H_ = 40, 24
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, batch, cvgs.OldestFirst, cvgs.Standard, W, H_, mirrored=mirrored)
    oc = oracle.OracleCircular(W, H_, cvgs.CV_32FC1, 3, batch, cvgs.OldestFirst, cvgs.Standard)
    f = cvgs.CV_32FC3
    s = torch.cuda.current_stream()
    for i in range(batch + 3):
        frame = H.random_u8((H_, W, 3), seed=9000 + i)
        frame_t = torch.from_numpy(frame).to(dev)
        pw = [cvgs.convertTo(cvgs.CV_8UC3, f), cvgs.subtract(f, [0.5] * 3)]
        ct.update(s, cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), *pw, ct.write_split(f))
        oc.update(cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3)], 1),
                              *pw, cvgs.WriteIOp(capi.WRITE_TENSOR_SPLIT, f, 16, W, H_, 0, batch)]))
    torch.cuda.synchronize()
    got = _read_device(ct.data(), ct.nbytes()).view(np.float32)
    H.assert_bit_exact(got, oc.array(np.float32), "circular depth %d mirrored %s" % (batch, mirrored))
    ct.release()


@pytest.mark.parametrize("n", [6, 50])
def test_nv12_crop_batch_to_nchw(oracle, n):
    """The decode-side version of the headline path: N crops (GpuMat.nv12_roi views: even x/y/w/h, own luma -> chroma
    offset) of ONE NV12 surface -> BGR float -> bilinear 64x128 -> normalize -> NCHW, one launch of the K4 kernel."""
    import torch
    dev = torch.device("cuda:0")
    w, h, dst = 1920, 1080, (64, 128)
    buf = _nv12(w, h, 4321)
    f = cvgs.CV_32FC3
    rng = np.random.default_rng(n)
    rects = []
    for _ in range(n):
        cw, ch = 2 * int(rng.integers(8, 250)), 2 * int(rng.integers(8, 400))
        rects.append((2 * int(rng.integers(0, (w - cw) // 2)), 2 * int(rng.integers(0, (h - ch) // 2)), cw, ch))

    def chain(luma, out):
        rd = cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], dst, capi.YUV_LIMITED, capi.BT709, False)
        return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]),
                cvgs.divide(f, H.K1_DIV[3]), cvgs.split(f, out, dst)]

    ref = np.zeros((n, 3 * dst[0] * dst[1]), np.float32)
    m = cvgs.GpuMat.from_array(buf, cvgs.CV_8UC1)
    luma_h = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
    oracle.execute(cvgs.lower(chain(luma_h, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    t = torch.from_numpy(buf).to(dev)
    o = torch.zeros((n, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    md = cvgs.GpuMat.from_tensor(t, cvgs.CV_8UC1)
    luma_d = cvgs.GpuMat(h, w, cvgs.CV_8UC1, md.data, md.step, owner=md.owner)
    ops = chain(luma_d, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1))
    assert cvgs.kernel_name(*ops) == "k4_nv12_resize_swap_mul_sub_div"
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    H.assert_bit_exact(o.cpu().numpy(), ref, "NV12 crop batch")
    assert ref.std() > 1
    # a crop view reads exactly the pixels a copy of that crop would: crop 0 as its own little surface
    x, y, cw, ch = rects[0]
    small = np.concatenate([buf[y:y + ch, x:x + cw], buf[h + y // 2:h + y // 2 + ch // 2, x:x + cw]], axis=0).copy()
    one = np.zeros((1, 3 * dst[0] * dst[1]), np.float32)
    ms = cvgs.GpuMat.from_array(small, cvgs.CV_8UC1)
    ls = cvgs.GpuMat(ch, cw, cvgs.CV_8UC1, ms.data, ms.step, owner=ms.owner)
    rd = cvgs.read_nv12(ls, dst, capi.YUV_LIMITED, capi.BT709, False)
    oracle.execute(cvgs.lower([rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, H.K1_SUB[3]),
                               cvgs.divide(f, H.K1_DIV[3]), cvgs.split(f, cvgs.GpuMat.from_array(one, cvgs.CV_32FC1), dst)]))
    H.assert_bit_exact(one[0], ref[0], "crop view == crop copy")


@pytest.mark.parametrize("ar", [cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_RN_EVEN, cvgs.PRESERVE_AR_LEFT])
@pytest.mark.parametrize("layout", [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_P010])
@pytest.mark.parametrize("shape", [((640, 360), (64, 64)), ((360, 640), (96, 64)), ((1920, 1080), (640, 640)), ((322, 198), (70, 70))])
@pytest.mark.parametrize("prog", ["rgb_norm", "bgr_norm", "u8", "u8_batch"])
def test_nv12_letterbox_resize_on_k4(oracle, ar, layout, shape, prog):
    """Aspect-ratio-preserving resizes of decoder surfaces (the letterboxed detector input: the reference's AspectRatio modes,
    include/cvGPUSpeedup.cuh:32,218-245, applied to the NV12 read-back of tests/resize/test_fused_resize.cu) on the K4 kernel:
    wide and tall sources, every mode, a default-value plane behind the used ones, RGB- and BGR-order normalisation into a planar
    tensor and a packed u8 image -- against the oracle and the interpreted kernel."""
    import torch
    dev = torch.device("cuda:0")
    (w, h), dst = shape
    st = cvgs.CV_16UC1 if layout == capi.YUV_P010 else cvgs.CV_8UC1
    if layout == capi.YUV_P010:
        surf = (H.random_u16((h * 3 // 2, w), 9100 + w) & 0xffc0).astype(np.uint16)
    else:
        surf = H.random_u8((h * 3 // 2, w), 9100 + w)
    f, u = cvgs.CV_32FC3, cvgs.CV_8UC3
    full = 1023.0 if layout == capi.YUV_P010 else 255.0
    n = 1 if prog == "u8" else 3  # "u8_batch": a dense batch of packed u8 images, the last one a default-value plane

    def build(wrap, out):
        m = wrap(surf)
        luma = cvgs.GpuMat(h, w, st, m.data, m.step, owner=m.owner)
        mats = [luma] if n == 1 else [luma, luma.nv12_roi(2, 2, w - 4, h - 4 - (h % 4)), luma]
        rd = cvgs.read_nv12(mats, dst, capi.YUV_LIMITED, capi.BT709, False, layout=layout)
        rd.ar = ar
        rd.background = cvgs._scalar([114.0, 100.5, 7.25])
        if n > 1:
            rd.used_planes = 2
        norm = [cvgs.multiply(f, [1 / full] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]
        if prog == "rgb_norm":
            return [rd] + norm + [cvgs.split(f, out, dst)]
        if prog == "bgr_norm":
            return [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f)] + norm + [cvgs.split(f, out, dst)]
        if prog == "u8_batch":
            return [rd, cvgs.convertTo(f, u, 255.0 / full), cvgs.write(u, out, dst)]
        return [rd, cvgs.convertTo(f, u, 255.0 / full), cvgs.write(u, out)]

    if prog == "u8_batch":
        shp, dt, tdt, ot = (n, dst[0] * dst[1], 3), np.uint8, torch.uint8, u
    elif prog == "u8":
        shp, dt, tdt, ot = (dst[1], dst[0], 3), np.uint8, torch.uint8, u
    else:
        shp, dt, tdt, ot = (n, 3 * dst[0] * dst[1]), np.float32, torch.float32, cvgs.CV_32FC1
    ref = np.zeros(shp, dt)
    oracle.execute(cvgs.lower(build(lambda a: cvgs.GpuMat.from_array(a, st), cvgs.GpuMat.from_array(ref, ot))))
    ts = torch.from_numpy(surf.view(np.int16) if layout == capi.YUV_P010 else surf).to(dev)
    gt = torch.zeros(shp, dtype=tdt, device=dev)
    ops = build(lambda a: cvgs.GpuMat.from_tensor(ts, st), cvgs.GpuMat.from_tensor(gt, ot))
    name = cvgs.kernel_name(*ops)
    want = {"rgb_norm": "k4_nv12_resize_mul_sub_div", "bgr_norm": "k4_nv12_resize_swap_mul_sub_div", "u8": "k4_nv12_resize_interp_u8c3",
            "u8_batch": "k4_nv12_resize_interp_u8c3"}[prog]
    assert name == want, name
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    assert ref.any()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "letterbox ar=%d layout=%d %s %s via %s" % (ar, layout, shape, prog, name))
    gt.zero_()
    cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=capi.CHAIN_FORCE_GENERIC)
    torch.cuda.synchronize()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "interpreted")


@pytest.mark.parametrize("range_", [capi.YUV_FULL, capi.YUV_LIMITED])
@pytest.mark.parametrize("primaries", [capi.BT601, capi.BT709])
@pytest.mark.parametrize("through_k4", [False, True])
def test_gpu_yuv_matrix_equals_the_derivation_from_kr_kb(range_, primaries, through_k4):
    """The product's eight coefficient sets against a float64 derivation from the standards' Kr / Kb (tests/
    test_independent_pins.py), with probe pixels that isolate each coefficient -- through the interpreted kernel (full-
    resolution NV12 read) and through the K4 fast kernel (resize to the same size: every tap weight is exactly 1)."""
    import torch
    from tests import test_independent_pins as P
    dev = torch.device("cuda:0")
    surf = P.nv12_probe_surface(P.derived_matrix(range_, primaries)[0])
    st = torch.from_numpy(surf).to(dev)
    f = cvgs.CV_32FC3
    luma = cvgs.GpuMat(2, 6, cvgs.CV_8UC1, st.data_ptr(), 6, owner=st)
    if through_k4:
        out = torch.zeros((1, 3 * 6 * 2), dtype=torch.float32, device=dev)
        ops = [cvgs.read_nv12(luma, (6, 2), range_, primaries, False), cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (6, 2))]
        assert cvgs.kernel_name(*ops).startswith("k4_nv12_resize")
        cvgs.executeOperations(torch.cuda.current_stream(), *ops)
        torch.cuda.synchronize()
        rgb = out.cpu().numpy().reshape(3, 2, 6).transpose(1, 2, 0).copy()
    else:
        out = torch.zeros((2, 6, 3), dtype=torch.float32, device=dev)
        cvgs.executeOperations(torch.cuda.current_stream(), cvgs.read_nv12(luma, None, range_, primaries, False),
                               cvgs.write(f, cvgs.GpuMat.from_tensor(out, f)))
        torch.cuda.synchronize()
        rgb = out.cpu().numpy()
    P.check_probe(rgb, range_, primaries, "gpu k4" if through_k4 else "gpu generic")
