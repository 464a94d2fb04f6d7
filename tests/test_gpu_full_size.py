"""BASELINE.json's configurations at FULL size on one MI355X, through the C-ABI:
 cfg #3  NV12 6144x3456 -> BGR float -> 1280x720 -> normalize -> split            bit-exact vs the oracle
 cfg #4  CircularTensor depth 16 of 1080p fp32 x3 (398 MB): 20 pushes             ordering + per-slot parity properties
 cfg #5  8 x 6K frames x 64 crops -> [512,3,128,64] (the single-GPU equivalent)   bit-exact vs the oracle
 cfg #1  1x1080p whole frame -> 64x128, subtract/divide (README values), split    bit-exact vs the oracle"""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from cvgpuspeedup_amd import workloads as W
from tests import helpers as H

pytestmark = pytest.mark.gpu


def test_cfg1_whole_1080p_frame(oracle):
    import torch
    dev = torch.device("cuda:0")
    frame = H.random_u8((1080, 1920, 3))
    f = cvgs.CV_32FC3

    def ops(m, outs):
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, m, (64, 128)), cvgs.subtract(f, [1, 4, 6]), cvgs.divide(f, [2, 8, 1]),
                cvgs.split(f, outs)]

    gt = [torch.zeros((128, 64), dtype=torch.float32, device=dev) for _ in range(3)]
    ft = torch.from_numpy(frame).to(dev)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3),
                                                             [cvgs.GpuMat.from_tensor(t, cvgs.CV_32FC1) for t in gt]))
    torch.cuda.synchronize()
    ref = [np.zeros((128, 64), np.float32) for _ in range(3)]
    oracle.execute(cvgs.lower(ops(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), [cvgs.GpuMat.from_array(r, cvgs.CV_32FC1) for r in ref])))
    for g, r in zip(gt, ref):
        H.assert_bit_exact(g.cpu().numpy(), r, "cfg1 plane")


def test_cfg3_nv12_6k_full(oracle):
    import torch
    dev = torch.device("cuda:0")
    w, h = W.FRAME_6K
    dst = (1280, 720)
    buf = H.random_u8((h + h // 2, w), seed=33)
    f = cvgs.CV_32FC3

    def ops(base_ptr, owner, out):
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, base_ptr, w, owner=owner)
        return [cvgs.read_nv12(luma, dst, capi.YUV_FULL, capi.BT709, False), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3]),
                cvgs.split(f, out, dst)]

    bt = torch.from_numpy(buf).to(dev)
    ot = torch.zeros((1, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    gops = ops(bt.data_ptr(), bt, cvgs.GpuMat.from_tensor(ot, cvgs.CV_32FC1))
    assert cvgs.kernel_name(*gops).startswith("k4_nv12")
    cvgs.executeOperations(torch.cuda.current_stream(), *gops)
    torch.cuda.synchronize()
    ref = np.zeros((1, 3 * dst[0] * dst[1]), np.float32)
    oracle.execute(cvgs.lower(ops(buf.ctypes.data, buf, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
    H.assert_bit_exact(ot.cpu().numpy(), ref, "cfg3 NV12 6K")


def test_cfg5_512_crops_single_gpu(oracle):
    """The tensor the 8-GPU job assembles, computed on one GPU: 8 frames x 64 crops, rank r's block at rows [64r, 64r+64)."""
    import torch
    dev = torch.device("cuda:0")
    fw, fh = W.FRAME_6K
    full_gpu = torch.zeros((512, 3 * 64 * 128), dtype=torch.float32, device=dev)
    full_ref = np.zeros((512, 3 * 64 * 128), np.float32)
    for r in range(8):
        frame = H.random_u8((fh, fw, 3), seed=W.SEED + 1000 * r)
        crops = H.random_crops(64, fw, fh, seed=W.SEED + 1000 * r + 500000)
        ft = torch.from_numpy(frame).to(dev)
        sl = full_gpu[64 * r:64 * (r + 1)]
        cvgs.executeOperations(torch.cuda.current_stream(),
                               *H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(sl, cvgs.CV_32FC1)))
        torch.cuda.synchronize()
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops,
                                             cvgs.GpuMat.from_array(full_ref[64 * r:64 * (r + 1)], cvgs.CV_32FC1))))
        del ft
    H.assert_bit_exact(full_gpu.cpu().numpy(), full_ref, "cfg5 [512,3,128,64]")


@pytest.mark.parametrize("order", [cvgs.NewestFirst, cvgs.OldestFirst])
def test_cfg4_circular_tensor_full_size(order):
    """Depth 16, 1080p fp32 x3 (398 MB tensor + 398 MB history).  After every push: slot(age a) == normalize(frame[k-a])
    for every filled slot (each slot compared against an independent NON-circular run of the same chain), unfilled
    slots are zero, and data() never moves."""
    import ctypes as C
    import torch
    dev = torch.device("cuda:0")
    Wd, Hd, B = 1920, 1080, 16
    f = cvgs.CV_32FC3
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, order, cvgs.Standard, Wd, Hd)
    base = ct.data()
    pw = [cvgs.convertTo(cvgs.CV_8UC3, f), cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3])]
    s = torch.cuda.current_stream()
    plane = Wd * Hd * 3
    view = torch.empty(B * plane, dtype=torch.float32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    singles = []
    for k in range(B + 4):
        frame = W.random_u8_torch((Hd, Wd, 3), 4000 + k, dev)
        m = cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3)
        ct.update(s, m, *pw, ct.write_split(f))
        one = torch.zeros((1, plane), dtype=torch.float32, device=dev)
        cvgs.executeOperations(s, cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [m], 1), *pw,
                               cvgs.split(f, cvgs.GpuMat.from_tensor(one, cvgs.CV_32FC1), (Wd, Hd)))
        singles.append(one)
        if k in (0, 3, B - 1, B, B + 3):
            torch.cuda.synchronize()
            assert ct.data() == base
            assert hip.hipMemcpy(C.c_void_p(view.data_ptr()), C.c_void_p(base), C.c_size_t(B * plane * 4), 3) == 0
            t = view.view(B, plane)
            for z in range(B):
                age = z if order == cvgs.NewestFirst else B - 1 - z
                src = k - age
                if src < 0:
                    assert not bool(t[z].any()), "slot %d must still be zero after %d pushes" % (z, k + 1)
                else:
                    assert torch.equal(t[z], singles[src][0]), "slot %d after push %d" % (z, k + 1)
        if len(singles) > B + 1:
            singles[len(singles) - B - 2] = None
    ct.release()


def test_cfg4_full_size_resize_push_matches_the_oracle(oracle):
    """cfg #4 as BASELINE spells it -- "push new frame with resize+normalize while shifting 15 slots" -- at FULL size:
    a 4K frame resized to 1080p and normalised into a depth-16 CircularTensor.  After 3 pushes the newest slot is compared
    bit for bit with the ORACLE's resize + normalize of that frame (2 Mpix x 3, ~2 s of CPU), the two older ones with the
    oracle's results for their frames, the 13 unfilled slots must be zero."""
    import ctypes as C
    import torch
    dev = torch.device("cuda:0")
    Wd, Hd, B = 1920, 1080, 16
    fw, fh = W.FRAME_4K
    f = cvgs.CV_32FC3
    ct = cvgs.CircularTensor(cvgs.CV_8UC3, cvgs.CV_32FC1, 3, B, cvgs.NewestFirst, cvgs.Standard, Wd, Hd)
    pw = [cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3])]
    s = torch.cuda.current_stream()
    plane = Wd * Hd * 3
    frames = [H.random_u8((fh, fw, 3), seed=7100 + k) for k in range(3)]
    for fr in frames:
        ft = torch.from_numpy(fr).to(dev)
        ct.update(s, cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), (Wd, Hd)), *pw, ct.write_split(f))
        torch.cuda.synchronize()
    view = torch.empty(B * plane, dtype=torch.float32, device=dev)
    hip = C.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(C.c_void_p(view.data_ptr()), C.c_void_p(ct.data()), C.c_size_t(B * plane * 4), 3) == 0
    got = view.view(B, plane).cpu().numpy()
    for age, fr in enumerate(reversed(frames)):
        ref = np.zeros((1, plane), np.float32)
        oracle.execute(cvgs.lower([cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, cvgs.GpuMat.from_array(fr, cvgs.CV_8UC3), (Wd, Hd)), *pw,
                                   cvgs.split(f, cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), (Wd, Hd))]))
        H.assert_bit_exact(got[age], ref[0], "cfg4 resize push, slot of age %d" % age)
    assert not got[3:].any()
    ct.release()


@pytest.mark.parametrize("flags", [0, capi.CHAIN_FORCE_GENERIC])
def test_output_tensor_larger_than_4_gib(oracle, flags):
    """45,000 crops of one 4K frame -> [45000,3,128,64] fp32 = 4.42 GB in ONE launch: plane strides and tensor offsets are
    64-bit (sized for 288 GB of HBM).  Planes at the start, across the 4 GiB boundary and at the end are compared with the
    oracle run on those crops alone (a plane depends on its crop only); on the fast kernel and on the interpreted one."""
    import torch
    dev = torch.device("cuda:0")
    n = 45000
    plane = 3 * 64 * 128
    fw, fh = W.FRAME_4K
    frame = H.random_u8((fh, fw, 3), seed=99)
    crops = H.random_crops(n, fw, fh, seed=424242)
    ft = torch.from_numpy(frame).to(dev)
    out = torch.empty((n, plane), dtype=torch.float32, device=dev)
    assert out.numel() * 4 > (1 << 32)
    out.fill_(-7.0)
    cvgs.executeOperations(torch.cuda.current_stream(), *H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)),
                           flags=flags)
    torch.cuda.synchronize()
    boundary = (1 << 32) // (plane * 4)  # the plane that straddles byte offset 2^32
    for lo in (0, boundary - 8, n - 16):
        sel = list(range(lo, lo + 16))
        ref = np.zeros((16, plane), np.float32)
        oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), [crops[i] for i in sel], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1))))
        H.assert_bit_exact(out[lo:lo + 16].cpu().numpy(), ref, "planes %d.. of a 4.4 GB tensor (flags %d)" % (lo, flags))
    # every plane was written (no -7 left anywhere): a per-plane minimum over the whole tensor, computed on the device
    assert bool((out.view(n, -1) != -7.0).any(dim=1).all())
    del out
    torch.cuda.empty_cache()
