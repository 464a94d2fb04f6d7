"""K4 at frame size, two output pixels per lane (csrc/k_nv12_x2.hip): crops of NV12 / NV21 decoder surfaces stretched into a
planar fp32 tensor behind the two compile-time programs -- BASELINE cfg #3's chain (reference tests/resize/test_fused_resize.cu:141-147
+ the K1 tail).  Bit-exact vs the oracle and identical to the one-pixel kernel (CVGS_CHAIN_NO_THREAD_FUSION) and to the interpreted
kernel.  CVGS_K4_X2=1 makes launch_nv12 pick the kernel whenever the chain is eligible (small, ragged and odd targets included);
the size rule itself is checked on cfg #3's own shape."""
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture
def forced():
    old = os.environ.get("CVGS_K4_X2")
    os.environ["CVGS_K4_X2"] = "1"
    yield
    if old is None:
        del os.environ["CVGS_K4_X2"]
    else:
        os.environ["CVGS_K4_X2"] = old


def _ops(lumas, out, dst, range_, prim, layout, swap):
    f = cvgs.CV_32FC3
    ops = [cvgs.read_nv12(lumas, dst, range_, prim, False, layout=layout)]
    if swap:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f))
    return ops + [cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                  cvgs.split(f, out, dst)]


def _run(oracle, surf, w, h, crops, dst, range_, prim, layout, swap, want):
    import torch
    dev = torch.device("cuda:0")
    n = len(crops)
    shp = (n, 3 * dst[0] * dst[1])
    ref = np.zeros(shp, np.float32)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf.ctypes.data, w, owner=surf)
    oracle.execute(cvgs.lower(_ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, range_, prim, layout, swap)))
    st = torch.from_numpy(surf).to(dev)
    gt = torch.zeros(shp, dtype=torch.float32, device=dev)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, st.data_ptr(), w, owner=st)
    ops = _ops([luma.nv12_roi(*c) for c in crops], cvgs.GpuMat.from_tensor(gt, cvgs.CV_32FC1), dst, range_, prim, layout, swap)
    assert cvgs.kernel_name(*ops) == want, cvgs.kernel_name(*ops)
    cvgs.executeOperations(torch.cuda.current_stream(), *ops)
    torch.cuda.synchronize()
    assert ref.any()
    H.assert_bit_exact(gt.cpu().numpy(), ref, "two pixels per lane")
    for flags, what in ((capi.CHAIN_NO_THREAD_FUSION, "one-pixel kernel"), (capi.CHAIN_FORCE_GENERIC, "interpreted kernel")):
        gt.zero_()
        cvgs.executeOperations(torch.cuda.current_stream(), *ops, flags=flags)
        torch.cuda.synchronize()
        H.assert_bit_exact(gt.cpu().numpy(), ref, what)
    assert cvgs.kernel_name(*ops, flags=capi.CHAIN_NO_THREAD_FUSION).startswith("k4_nv12_resize")


# (surface w, h) -> (dst w, dst h): 4.8x down (cfg #3's scale), 2x up, odd target widths (the last lane stores one pixel), a target
# narrower than one lane pair's tile, 1 < scale < 2, a 4-pixel-wide surface, a target of one row
SHAPES = [((640, 360), (133, 75)), ((322, 198), (640, 400)), ((640, 360), (213, 120)), ((64, 36), (3, 50)), ((500, 300), (301, 201)),
          ((4, 4), (64, 3)), ((1280, 720), (640, 1)), ((130, 2), (65, 5))]


@pytest.mark.parametrize("swap", [True, False])
@pytest.mark.parametrize("layout,range_,prim", [(capi.YUV_NV12, capi.YUV_FULL, capi.BT709), (capi.YUV_NV21, capi.YUV_LIMITED, capi.BT601),
                                                (capi.YUV_NV12, capi.YUV_LIMITED, capi.BT2020)])
@pytest.mark.parametrize("shape", SHAPES)
def test_x2_matches_the_oracle(forced, oracle, shape, layout, range_, prim, swap):
    (w, h), dst = shape
    surf = H.random_u8((h + h // 2, w), 8000 + w + h)
    _run(oracle, surf, w, h, [(0, 0, w, h)], dst, range_, prim, layout, swap, "k4_nv12_x2_swap_mul_sub_div" if swap else "k4_nv12_x2_mul_sub_div")


def test_x2_crops_of_a_surface_as_one_batch(forced, oracle):
    """N crops (even x, y, w, h) of one surface -> [N, 3, h, w]: blockIdx.z = crop, each with its own luma -> chroma offset"""
    w, h = 1280, 720
    surf = H.random_u8((h + h // 2, w), 8100)
    crops = [tuple(v & ~1 for v in c) for c in H.random_crops(6, w, h, seed=81, wmin=9, wmax=600, hmin=9, hmax=400)] + [(w - 4, h - 2, 4, 2), (0, 0, w, h)]
    _run(oracle, surf, w, h, crops, (96, 54), capi.YUV_LIMITED, capi.BT709, capi.YUV_NV12, True, "k4_nv12_x2_swap_mul_sub_div")


def test_x2_extreme_samples(forced, oracle):
    """0 / 255 luma and chroma checkerboards: the ends of every conversion, zero dividends in front of the division by the uniform divisor"""
    w, h = 256, 128
    surf = np.zeros((h + h // 2, w), np.uint8)
    surf[:h:2, ::2] = 255
    surf[1:h:2, 1::2] = 255
    surf[h:, ::4] = 255
    surf[h + 1::2, 1::4] = 255
    for dst in ((128, 64), (512, 256), (100, 33)):
        _run(oracle, surf, w, h, [(0, 0, w, h)], dst, capi.YUV_FULL, capi.BT601, capi.YUV_NV12, True, "k4_nv12_x2_swap_mul_sub_div")


def test_x2_is_the_default_at_cfg3_and_leaves_the_rest(oracle):
    """no environment hook: cfg #3's own shape (6K NV12 -> 1280 x 720) takes the two-pixel kernel; a 64 x 128 crop batch, a letterboxed
    detector input and an fp16 tensor do not"""
    import torch
    assert "CVGS_K4_X2" not in os.environ
    w, h = 6144, 3456
    surf = H.random_u8((h + h // 2, w), 8200)
    _run(oracle, surf, w, h, [(0, 0, w, h)], (1280, 720), capi.YUV_FULL, capi.BT709, capi.YUV_NV12, True, "k4_nv12_x2_swap_mul_sub_div")
    dev = torch.device("cuda:0")
    st = torch.from_numpy(surf).to(dev)
    luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, st.data_ptr(), w, owner=st)
    f = cvgs.CV_32FC3
    out = torch.zeros((8, 3 * 64 * 128), dtype=torch.float32, device=dev)
    small = _ops([luma.nv12_roi(0, 0, 640, 360)] * 8, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128), capi.YUV_FULL, capi.BT709, capi.YUV_NV12, True)
    assert cvgs.kernel_name(*small) == "k4_nv12_resize_swap_mul_sub_div"
    out2 = torch.zeros((1, 3 * 1280 * 1280), dtype=torch.float32, device=dev)
    rd = cvgs.read_nv12(luma, (1280, 1280), capi.YUV_FULL, capi.BT709, False)
    rd.ar = cvgs.PRESERVE_AR
    rd.background = cvgs._scalar([114.0, 114.0, 114.0])
    box = [rd, cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.5] * 3), cvgs.divide(f, [0.25] * 3), cvgs.split(f, cvgs.GpuMat.from_tensor(out2, cvgs.CV_32FC1), (1280, 1280))]
    assert cvgs.kernel_name(*box) == "k4_nv12_resize_mul_sub_div"
