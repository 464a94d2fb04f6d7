#!/usr/bin/env python3
"""Batch-size sweep across the descriptor thresholds (8 / 16 / 52 / 64 / 320 planes): time per EAGER call (one event pair
around a run of calls) for the batched chain shapes, printed with the kernel picked.  A jump between neighbours that is not
explained by the extra pixels is a cliff (this is how the 65-crop cliff of the headline chain was found)."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

dev = torch.device("cuda:0")
BATCHES = [1, 8, 9, 16, 17, 52, 53, 64, 65, 128, 320, 321, 640]


def timed(lib, chains, iters=200):
    s = torch.cuda.current_stream().cuda_stream
    for i in range(10):
        capi.check(lib.cvgs_execute(C.byref(chains[i % len(chains)].desc), s))
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(iters):
        capi.check(lib.cvgs_execute(C.byref(chains[i % len(chains)].desc), s))
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    lib = capi.load_library()
    torch.cuda.set_device(0)
    w, h = W.FRAME_4K
    frames = [W.random_u8_torch((h, w, 3), 100 + i, dev) for i in range(4)]
    nv12 = [W.random_u8_torch((h + h // 2, w), 200 + i, dev) for i in range(4)]
    f = cvgs.CV_32FC3
    norm = [cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3])]
    shapes = {}

    def resize_tensor(fr, n, k):
        m = cvgs.GpuMat.from_tensor(fr, cvgs.CV_8UC3)
        out = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=dev)
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [m.roi(i % 3000, i % 1900, 60, 120) for i in range(n)], (64, 128), n),
                cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), *norm, cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128))], out

    def resize_packed_u8(fr, n, k):
        m = cvgs.GpuMat.from_tensor(fr, cvgs.CV_8UC3)
        out = torch.zeros((n, 64 * 128, 3), dtype=torch.uint8, device=dev)
        return [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [m.roi(i % 3000, i % 1900, 60, 120) for i in range(n)], (64, 128), n),
                cvgs.convertTo(f, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(out, cvgs.CV_8UC3), (64, 128))], out

    def pixel_tensor(fr, n, k):
        m = cvgs.GpuMat.from_tensor(fr, cvgs.CV_8UC3)
        out = torch.zeros((n, 3 * 60 * 120), dtype=torch.float32, device=dev)
        return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [m.roi(i % 3000, i % 1900, 60, 120) for i in range(n)], n), cvgs.convertTo(cvgs.CV_8UC3, f), *norm,
                cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (60, 120))], out

    def warp_tensor(fr, n, k):
        m = cvgs.GpuMat.from_tensor(fr, cvgs.CV_8UC3)
        out = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=dev)
        ms = [[[0.5 * np.cos(0.1 * i), -0.5 * np.sin(0.1 * i), -200.0 - i], [0.5 * np.sin(0.1 * i), 0.5 * np.cos(0.1 * i), -100.0 - i]] for i in range(n)]
        return [cvgs.warp(cvgs.WARP_AFFINE, cvgs.CV_8UC3, [m] * n, ms, (64, 128)), *norm, cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128))], out

    def nv12_tensor(fr, n, k):
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, nv12[k].data_ptr(), w, owner=nv12[k])
        out = torch.zeros((n, 3 * 64 * 128), dtype=torch.float32, device=dev)
        return [cvgs.read_nv12([luma.nv12_roi(2 * (i % 1500), 2 * (i % 900), 60, 120) for i in range(n)], (64, 128), capi.YUV_LIMITED, capi.BT709, False),
                cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), *norm, cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128))], out

    shapes = {"resize -> tensor (K1)": resize_tensor, "resize -> packed u8": resize_packed_u8, "per-pixel -> tensor": pixel_tensor,
              "warp -> tensor": warp_tensor, "nv12 crops -> tensor (K4)": nv12_tensor}
    for name, make in shapes.items():
        row = {}
        for n in BATCHES:
            chains, keep, ops = [], [], None
            for k in range(4):
                ops, out = make(frames[k], n, k)
                chains.append(cvgs.lower(ops))
                keep.append(out)
            t = timed(lib, chains)
            row[n] = (round(t, 2), cvgs.kernel_name(*ops))
        print(json.dumps({"chain": name, "us_by_batch": {str(n): row[n][0] for n in BATCHES},
                          "kernels": sorted(set(v[1] for v in row.values()))}), flush=True)


if __name__ == "__main__":
    main()
