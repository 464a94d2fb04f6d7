#!/usr/bin/env python3
"""Out-of-bounds READ probe: every source image is placed so that its last byte is the last byte of a hipMalloc'ed
region whose size is a multiple of 2 MiB (so that, unless the driver happens to map another allocation right behind it,
the next page is not mapped), then the fast kernels run on crops that touch the
last row / column.  A read past the image faults the process; finishing = no over-read.  A fault aborts the interpreter, so
the GPU suite runs this file in a SUBPROCESS (tests/test_gpu_oob_probe.py: SURVEY.md 5 "GPU-side bounds checking").
Covered: K1 / the interpreted kernel (every depth and channel count), per-pixel chains, warps, K4 on NV12 / NV21 / I420 / YV12 / P010
surfaces, the u8 colour conversions, the fused multi-chain launch (cvgs_execute_many) and every kind of the descriptor queue."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402

hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
MB2 = 2 << 20


def at_end(a):
    """device copy of numpy array `a` ending exactly at the end of a 2 MiB-multiple allocation; returns (ptr, keep)"""
    size = ((a.nbytes + MB2 - 1) // MB2) * MB2
    base = C.c_void_p()
    assert hip.hipMalloc(C.byref(base), size) == 0
    ptr = base.value + size - a.nbytes
    assert hip.hipMemcpy(C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes, 1) == 0
    return ptr


def main():
    torch.cuda.init()
    s = torch.cuda.current_stream()
    rng = np.random.default_rng(0)
    n_run = 0
    for depth, dt in ((cvgs.CV_8U, np.uint8), (cvgs.CV_16U, np.uint16), (cvgs.CV_32F, np.float32)):
        for cn in (1, 2, 3, 4):
            for (w, h) in ((1, 1), (2, 3), (5, 4), (257, 9), (1023, 17)):
                a = (rng.integers(0, 200, (h, w, cn))).astype(dt)
                st, f = cvgs.make_type(depth, cn), cvgs.make_type(cvgs.CV_32F, cn)
                m = cvgs.GpuMat(h, w, st, at_end(a), w * cn * a.itemsize)
                out = torch.zeros((3, 64 * 48 * cn), dtype=torch.float32, device="cuda")
                om = cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1)
                crops = [m, m.roi(w - 1, h - 1, 1, 1), m.roi(0, h - 1, w, 1)]
                # resize (K1 / generic), full image + last pixel + last row
                cvgs.executeOperations(s, cvgs.resize(st, cvgs.INTER_LINEAR, crops, (64, 48), 3), cvgs.multiply(f, [0.5] * cn),
                                       cvgs.write(f, cvgs.GpuMat.from_tensor(out.view(3, 64 * 48, cn), f), (64, 48)))
                # per-pixel (pointwise4 / generic)
                o2 = torch.zeros((h, w, cn), dtype=torch.float32, device="cuda")
                ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, [m], 1)] + ([cvgs.convertTo(st, f)] if depth != cvgs.CV_32F else []) + \
                      [cvgs.multiply(f, [0.5] * cn), cvgs.write(f, cvgs.GpuMat.from_tensor(o2, f))]
                cvgs.executeOperations(s, *ops)
                if depth == cvgs.CV_8U and cn >= 3:  # warps (fast + interpreted)
                    for flags in (0, capi.CHAIN_FORCE_GENERIC):
                        cvgs.executeOperations(s, cvgs.warp(cvgs.WARP_AFFINE, st, [m, m], [[[1, 0, 0.4], [0, 1, 0.4]], [[0.3, 0, -1], [0, 0.3, -1]]], (64, 48)),
                                               cvgs.split(f, cvgs.GpuMat.from_tensor(out[:2], cvgs.CV_32FC1), (64, 48)), flags=flags)
                torch.cuda.synchronize()
                n_run += 1
    # NV12 surfaces (even sizes), whole + crops touching the last rows / columns
    for (w, h) in ((4, 2), (6, 4), (64, 36), (642, 362)):
        a = rng.integers(0, 255, (h + h // 2, w)).astype(np.uint8)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, at_end(a), w)
        out = torch.zeros((3, 3 * 64 * 48), dtype=torch.float32, device="cuda")
        views = [luma, luma.nv12_roi(w - 2, h - 2, 2, 2), luma.nv12_roi(0, h - 2, w, 2)]
        f = cvgs.CV_32FC3
        for flags in (0, capi.CHAIN_FORCE_GENERIC):
            cvgs.executeOperations(s, cvgs.read_nv12(views, (64, 48), capi.YUV_FULL, capi.BT709, False), cvgs.multiply(f, [0.5] * 3),
                                   cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 48)), flags=flags)
        torch.cuda.synchronize()
        n_run += 1
    # u8 colour conversions (16 pixels per thread through the LDS: k_cvtcolor_u8.hip), aligned widths, image at the very end
    for (w, h, cn, code, ocn) in ((16, 3, 3, cvgs.COLOR_BGR2RGB, 3), (1040, 5, 3, cvgs.COLOR_BGR2BGRA, 4), (2064, 2, 4, cvgs.COLOR_BGRA2BGR, 3),
                                  (1040, 4, 3, cvgs.COLOR_BGR2GRAY, 1), (4096, 1, 4, cvgs.COLOR_RGBA2GRAY, 1)):
        a = rng.integers(0, 255, (h, w, cn)).astype(np.uint8)
        it, ot = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_8U, ocn)
        m = cvgs.GpuMat(h, w, it, at_end(a), w * cn)
        o = torch.zeros((h, w, ocn), dtype=torch.uint8, device="cuda")
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, it, [m], 1), cvgs.cvtColor(code, it, ot), cvgs.write(ot, cvgs.GpuMat.from_tensor(o, ot))]
        assert cvgs.kernel_name(*ops).startswith("pointwise16_u8"), cvgs.kernel_name(*ops)
        cvgs.executeOperations(s, *ops)
        torch.cuda.synchronize()
        n_run += 1
    # NV21 / planar-chroma surfaces at the very end of an allocation
    for layout in (capi.YUV_NV21, capi.YUV_I420, capi.YUV_YV12):
        w, h = 64, 36
        a = rng.integers(0, 255, (h + h // 2, w)).astype(np.uint8)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, at_end(a), w)
        out = torch.zeros((1, 3 * 64 * 48), dtype=torch.float32, device="cuda")
        cvgs.executeOperations(s, cvgs.read_nv12(luma, (64, 48), capi.YUV_FULL, capi.BT709, False, layout=layout),
                               cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 48)))
        torch.cuda.synchronize()
        n_run += 1
    # P010 surfaces (16-bit samples: 4-byte luma and 8-byte chroma windows in K4), whole + crops touching the last rows / columns
    for (w, h) in ((4, 2), (6, 4), (64, 36), (642, 362)):
        a = rng.integers(0, 65535, (h + h // 2, w)).astype(np.uint16)
        luma = cvgs.GpuMat(h, w, cvgs.CV_16UC1, at_end(a), 2 * w)
        out = torch.zeros((3, 3 * 64 * 48), dtype=torch.float32, device="cuda")
        views = [luma, luma.nv12_roi(w - 2, h - 2, 2, 2), luma.nv12_roi(0, h - 2, w, 2)]
        if w >= 8:
            views[1] = luma.nv12_roi(w - 4, h - 2, 4, 2)  # the narrowest crop K4 serves
        f = cvgs.CV_32FC3
        for flags in (0, capi.CHAIN_FORCE_GENERIC):
            cvgs.executeOperations(s, cvgs.read_nv12(views, (64, 48), capi.YUV_LIMITED, capi.BT2020, False, layout=capi.YUV_P010), cvgs.multiply(f, [0.5] * 3),
                                   cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 48)), flags=flags)
        torch.cuda.synchronize()
        n_run += 1
    # the fused multi-chain launch (cvgs_execute_many, grid z = chain) and the descriptor queue's four kinds: K1-shaped chains whose
    # crops touch the last row / column of a frame that ends with its allocation
    f3 = cvgs.CV_32FC3

    def k1_ops(src_mat, st, crops, out, cn=3):
        ft = cvgs.make_type(cvgs.CV_32F, cn)
        return [cvgs.resize(st, cvgs.INTER_LINEAR, crops, (64, 128), len(crops)), cvgs.cvtColor(cvgs.COLOR_RGB2BGR if cn == 3 else cvgs.COLOR_RGBA2BGRA, ft),
                cvgs.multiply(ft, [0.3] * cn), cvgs.subtract(ft, [1.0, 4.0, 3.2, 0.5][:cn]), cvgs.divide(ft, [3.2, 0.6, 11.8, 33.0][:cn]),
                cvgs.split(ft, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128))]

    for depth, dt, cn in ((cvgs.CV_8U, np.uint8, 3), (cvgs.CV_8U, np.uint8, 4), (cvgs.CV_16U, np.uint16, 3), (cvgs.CV_16S, np.int16, 4)):
        w, h = 333, 77
        a = (rng.integers(0, 200, (h, w, cn))).astype(dt)
        st = cvgs.make_type(depth, cn)
        m = cvgs.GpuMat(h, w, st, at_end(a), w * cn * a.itemsize)
        crops = [m, m.roi(w - 3, h - 2, 3, 2), m.roi(0, h - 1, w, 1), m.roi(w - 9, 0, 9, h), m.roi(w - 40, h - 50, 40, 50)]
        outs = [torch.zeros((len(crops), cn * 64 * 128), dtype=torch.float32, device="cuda") for _ in range(4)]
        cvgs.executeMany(s, [k1_ops(m, st, crops, o, cn) for o in outs[:3]])
        torch.cuda.synchronize()
        q = cvgs.Queue()
        try:
            q.wait(q.submit(*k1_ops(m, st, crops, outs[3], cn)))
        finally:
            q.destroy()
        assert torch.equal(outs[0], outs[3]), "queue and fused launch disagree"
        n_run += 1
    for layout, dt, st, sb in ((capi.YUV_NV12, np.uint8, cvgs.CV_8UC1, 1), (capi.YUV_NV21, np.uint8, cvgs.CV_8UC1, 1), (capi.YUV_P010, np.uint16, cvgs.CV_16UC1, 2)):
        w, h = 642, 362
        a = rng.integers(0, 255 if sb == 1 else 65535, (h + h // 2, w)).astype(dt)
        luma = cvgs.GpuMat(h, w, st, at_end(a), w * sb)
        views = [luma.nv12_roi(0, 0, w, h), luma.nv12_roi(w - 4, h - 2, 4, 2), luma.nv12_roi(0, h - 2, w, 2), luma.nv12_roi(w - 8, 0, 8, h)]
        outs = [torch.zeros((len(views), 3 * 64 * 128), dtype=torch.float32, device="cuda") for _ in range(3)]

        def nv_ops(out):
            return [cvgs.read_nv12(views, (64, 128), capi.YUV_LIMITED, capi.BT709, False, layout=layout), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f3),
                    cvgs.multiply(f3, [0.3] * 3), cvgs.subtract(f3, [1.0, 4.0, 3.2]), cvgs.divide(f3, [3.2, 0.6, 11.8]),
                    cvgs.split(f3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128))]
        cvgs.executeMany(s, [nv_ops(o) for o in outs[:2]])
        torch.cuda.synchronize()
        q = cvgs.Queue()
        try:
            q.wait(q.submit(*nv_ops(outs[2])))
        finally:
            q.destroy()
        assert torch.equal(outs[0], outs[2]), "queue and fused launch disagree (4:2:0 surfaces)"
        n_run += 1
    print("no read past the end of any source image: %d configurations ran to completion" % n_run)


if __name__ == "__main__":
    main()
