#!/usr/bin/env python3
"""Whole-surface colour conversion of a 4K decoder surface WITHOUT a resize (the decode-side cvtColor: cvGS::cvtColorNV12 ->
convertTo -> write / split): NV12 -> packed u8 BGR, -> packed fp32 BGR, -> normalized planar fp32 (NCHW at full resolution).
Prints the kernel the dispatcher picks, microseconds per eager call (8 surfaces and 8 outputs in rotation, 0.4 .. 0.9 GB per case), the
algorithmic bytes and the fraction of 8 TB/s."""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

dev = torch.device("cuda:0")
lib = capi.load_library()
w, h = W.FRAME_4K
N = 8  # 8 surfaces + 8 outputs in rotation: 0.4 .. 0.9 GB per case, twice the Infinity Cache and more


def run(name, mode, iters=60):
    chains, keep = [], []
    f3 = cvgs.CV_32FC3
    for i in range(N):
        surf = W.random_u8_torch((h * 3 // 2, w), 4100 + i, dev)
        luma = cvgs.GpuMat.from_tensor(surf[:h], cvgs.CV_8UC1)
        rd = cvgs.read_nv12(luma, None, capi.YUV_LIMITED, capi.BT709, alpha=False)
        if mode == "bgr_ref":  # the same output from an already-converted packed BGR frame (k_pointwise4's u8c3 path): the yardstick
            src = W.random_u8_torch((h, w, 3), 4100 + i, dev)
            out = torch.zeros((1, 3 * h * w), dtype=torch.float32, device=dev)
            ops = [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_8UC3, [cvgs.GpuMat.from_tensor(src, cvgs.CV_8UC3)], 1), cvgs.convertTo(cvgs.CV_8UC3, f3),
                   cvgs.multiply(f3, [1 / 255.0] * 3), cvgs.subtract(f3, [0.485, 0.456, 0.406]), cvgs.divide(f3, [0.229, 0.224, 0.225]),
                   cvgs.split(f3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (w, h))]
            chains.append(cvgs.lower(ops))
            keep += [src, out]
            out_bytes = 12
            continue
        if mode == "u8":
            out = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
            ops = [rd, cvgs.convertTo(f3, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(out, cvgs.CV_8UC3))]
            out_bytes = 3
        elif mode == "f32":
            out = torch.zeros((h, w, 3), dtype=torch.float32, device=dev)
            ops = [rd, cvgs.write(f3, cvgs.GpuMat.from_tensor(out, f3))]
            out_bytes = 12
        else:
            out = torch.zeros((1, 3 * h * w), dtype=torch.float32, device=dev)
            ops = [rd, cvgs.multiply(f3, [1 / 255.0] * 3), cvgs.subtract(f3, [0.485, 0.456, 0.406]), cvgs.divide(f3, [0.229, 0.224, 0.225]),
                   cvgs.split(f3, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (w, h))]
            out_bytes = 12
        chains.append(cvgs.lower(ops))
        keep += [surf, out]
    s = torch.cuda.current_stream().cuda_stream
    st = {"i": 0}

    def launch():
        capi.check(lib.cvgs_execute(C.byref(chains[st["i"] % N].desc), s))
        st["i"] += 1

    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) * 1e-3 / iters
    alg = w * h * (3 if mode == "bgr_ref" else 1.5) + w * h * out_bytes
    res = {"case": name, "kernel": cvgs.kernel_name(*ops), "us": round(t * 1e6, 2), "GB_per_s": round(alg / t / 1e9, 1),
           "frac_of_8TBs": round(alg / t / 8e12, 4)}
    del chains, keep
    torch.cuda.empty_cache()
    return res


def run_all(iters=60):
    return [run("4K NV12 -> BGR u8 packed", "u8", iters), run("4K NV12 -> BGR fp32 packed", "f32", iters),
            run("4K NV12 -> normalize -> NCHW fp32", "nchw", iters),
            run("yardstick: 4K packed BGR u8 -> normalize -> NCHW fp32", "bgr_ref", iters)]


if __name__ == "__main__":
    for r in run_all():
        print(json.dumps(r))
