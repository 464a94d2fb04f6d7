"""The 4:2:0 readers behind one read kind (cvgs_read_desc.yuv_layout): NV12 (the reference's, fk::ReadYUV<fk::NV12> at
tests/resize/test_fused_resize.cu:50), NV21, I420, YV12.  The same picture stored in the four layouts must convert to the
same RGB: the oracle's layouts are checked against each other here (the NV12 path itself is pinned by the Kr/Kb probes of
tests/test_independent_pins.py), the GPU against the oracle below."""
import ctypes as C

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

LAYOUTS = [capi.YUV_NV12, capi.YUV_NV21, capi.YUV_I420, capi.YUV_YV12]


def surfaces(w, h, seed):
    """One random 4:2:0 picture in the four layouts: {layout: (H*3/2, W) u8 array}."""
    y = H.random_u8((h, w), seed)
    u = H.random_u8((h // 2, w // 2), seed + 1)
    v = H.random_u8((h // 2, w // 2), seed + 2)
    out = {}
    for layout in LAYOUTS:
        s = np.zeros((h + h // 2, w), np.uint8)
        s[:h] = y
        if layout in (capi.YUV_NV12, capi.YUV_NV21):
            a, b = (u, v) if layout == capi.YUV_NV12 else (v, u)
            s[h:, 0::2] = a
            s[h:, 1::2] = b
        else:
            first, second = (u, v) if layout == capi.YUV_I420 else (v, u)
            s[h:].reshape(-1)[:(h // 2) * (w // 2)] = first.reshape(-1)
            s[h:].reshape(-1)[(h // 2) * (w // 2):] = second.reshape(-1)
        out[layout] = s
    return out


def chain(wrap, surf, w, h, layout, dst, out, rng=capi.YUV_LIMITED, prim=capi.BT601):
    m = wrap(surf)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
    f = cvgs.CV_32FC3
    wr = cvgs.write(f, out) if dst is None else cvgs.split(f, out, dst)
    return [cvgs.read_nv12(luma, dst, rng, prim, False, layout=layout), cvgs.multiply(f, [0.5, 0.25, 2.0]), wr]


@pytest.mark.parametrize("dst", [None, (50, 30)])
def test_oracle_layouts_agree(oracle, dst):
    w, h = 96, 64
    surf = surfaces(w, h, 40)
    outs = {}
    for layout in LAYOUTS:
        out = np.zeros((h, w, 3), np.float32) if dst is None else np.zeros((1, 3 * dst[0] * dst[1]), np.float32)
        ot = cvgs.CV_32FC3 if dst is None else cvgs.CV_32FC1
        oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), surf[layout], w, h, layout, dst,
                                        cvgs.GpuMat.from_array(out, ot))))
        outs[layout] = out
    assert outs[capi.YUV_NV12].any()
    for layout in LAYOUTS[1:]:
        assert (outs[layout].view(np.uint32) == outs[capi.YUV_NV12].view(np.uint32)).all(), layout
    # and a swapped layout read as NV12 must NOT agree (the test would be vacuous otherwise)
    wrong = np.zeros_like(outs[capi.YUV_NV12])
    oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), surf[capi.YUV_NV21], w, h, capi.YUV_NV12, dst,
                                    cvgs.GpuMat.from_array(wrong, cvgs.CV_32FC3 if dst is None else cvgs.CV_32FC1))))
    assert (wrong != outs[capi.YUV_NV12]).any()


def test_layout_validation(lib):
    w, h = 32, 16
    surf = surfaces(w, h, 3)[capi.YUV_I420]
    out = np.zeros((h, w, 3), np.float32)
    wrap = lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1)
    ch = cvgs.lower(chain(wrap, surf, w, h, capi.YUV_I420, None, cvgs.GpuMat.from_array(out, cvgs.CV_32FC3)))
    assert lib.cvgs_validate(C.byref(ch.desc)) == 0
    ch.desc.read.yuv_layout = 7
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_INVALID
    # a crop of a planar-chroma surface cannot name its second chroma plane: refused
    m = wrap(surf)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, m.data, m.step, owner=m.owner)
    rd = cvgs.read_nv12([luma.nv12_roi(4, 2, 16, 8)], None, layout=capi.YUV_I420, alpha=False)
    ch = cvgs.lower([rd, cvgs.write(cvgs.CV_32FC3, cvgs.GpuMat.from_array(np.zeros((8, 16, 3), np.float32), cvgs.CV_32FC3))])
    assert lib.cvgs_validate(C.byref(ch.desc)) == capi.ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("layout", LAYOUTS)
@pytest.mark.parametrize("dst", [None, (213, 120), (64, 128)])
def test_gpu_layouts_match_the_oracle(oracle, layout, dst):
    import torch
    dev = torch.device("cuda:0")
    w, h = 640, 360
    surf = surfaces(w, h, 50)[layout]
    st = torch.from_numpy(surf).to(dev)
    if dst is None:
        gt, ref, ot = torch.zeros((h, w, 3), dtype=torch.float32, device=dev), np.zeros((h, w, 3), np.float32), cvgs.CV_32FC3
    else:
        gt, ref, ot = torch.zeros((1, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev), np.zeros((1, 3 * dst[0] * dst[1]), np.float32), cvgs.CV_32FC1
    ops = chain(lambda a: cvgs.GpuMat.from_tensor(st, cvgs.CV_8UC1), surf, w, h, layout, dst, cvgs.GpuMat.from_tensor(gt, ot))
    name = cvgs.kernel_name(*ops)
    if dst is not None:
        assert name.startswith("k4_nv12_resize"), name  # planar chroma too (tests/test_gpu_k4_planar.py)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    oracle.execute(cvgs.lower(chain(lambda a: cvgs.GpuMat.from_array(a, cvgs.CV_8UC1), surf, w, h, layout, dst, cvgs.GpuMat.from_array(ref, ot))))
    H.assert_bit_exact(gt.cpu().numpy(), ref, "layout %d dst %s via %s" % (layout, dst, name))
