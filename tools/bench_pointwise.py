#!/usr/bin/env python3
"""Pointwise chains on whole 4K frames (K5/K6 of the reference: read -> convertTo -> arithmetic -> write / split), per
source type: which kernel runs and what fraction of the HBM roofline it reaches."""
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

TORCH = {"8U": torch.uint8, "16U": torch.int16, "16S": torch.int16, "32F": torch.float32, "32S": torch.int32}
DEPTH = {"8U": cvgs.CV_8U, "16U": cvgs.CV_16U, "16S": cvgs.CV_16S, "32F": cvgs.CV_32F, "32S": cvgs.CV_32S}


def events_time(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def case(dev, depth, cn, out, iters=50, convert_only=False):
    w, h = W.FRAME_4K
    st, f = cvgs.make_type(DEPTH[depth], cn), cvgs.make_type(cvgs.CV_32F, cn)
    esz = torch.empty(0, dtype=TORCH[depth]).element_size()
    nb = max(3, (600 << 20) // (w * h * cn * (esz + 4)))
    chains, keep, ops = [], [], None
    for i in range(nb):
        src = torch.randint(0, 100, (h, w, cn), device=dev, dtype=torch.int32).to(TORCH[depth])
        m = cvgs.GpuMat.from_tensor(src, st)
        ops = [cvgs.ReadIOp(capi.READ_PIXEL, st, [m], 1)]
        if depth != "32F":
            ops.append(cvgs.convertTo(st, f))
        if not convert_only:
            ops += [cvgs.multiply(f, [W.K1_ALPHA] * cn), cvgs.subtract(f, W.K1_SUB[cn]), cvgs.divide(f, W.K1_DIV[cn])]
        if out == "planes":  # cvGS::split(std::vector<GpuMat>): separate pitched planes, the reference's tests/read/test_read_x_split.cu
            o = [torch.zeros((h, w), dtype=torch.float32, device=dev) for _ in range(cn)]
            ops.append(cvgs.split(f, [cvgs.GpuMat.from_tensor(p, cvgs.CV_32FC1) for p in o]))
        elif out == "planar":
            o = torch.zeros((1, cn * w * h), dtype=torch.float32, device=dev)
            ops.append(cvgs.split(f, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), (w, h)) if cn > 1 else
                       cvgs.write(f, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), (w, h)))
        else:
            o = torch.zeros((h, w, cn), dtype=torch.float32, device=dev)
            ops.append(cvgs.write(f, cvgs.GpuMat.from_tensor(o, f)))
        keep += [src, o]
        chains.append(cvgs.lower(ops))
    lib = capi.load_library()
    s = torch.cuda.current_stream().cuda_stream
    st_ = {"i": 0}

    def launch():
        ch = chains[st_["i"] % len(chains)]
        st_["i"] += 1
        capi.check(lib.cvgs_execute(C.byref(ch.desc), s))

    t = events_time(launch, iters)
    alg = w * h * cn * (esz + 4)
    return {"case": "4K %sC%d -> fp32 %s (%s)" % (depth, cn, out, "convertTo only" if convert_only else "normalize"), "kernel": cvgs.kernel_name(*ops), "us": round(t * 1e6, 2),
            "GB_per_s": round(alg / t / 1e9, 1), "frac_of_8TBs": round(alg / t / 1e9 / 8000.0, 4)}


if __name__ == "__main__":
    dev = torch.device("cuda:0")
    for depth, cn in (("8U", 3), ("8U", 4), ("8U", 1), ("16U", 3), ("32F", 3), ("32F", 4), ("16S", 1)):
        for out in ("planar", "packed"):
            print(json.dumps(case(dev, depth, cn, out)))
    # the reference's single-image tests: read -> convertTo -> split(vector<GpuMat>) (tests/read/test_read_x_split.cu:58-60)
    for depth, cn in (("8U", 3), ("8U", 4), ("8U", 2), ("16U", 3), ("16S", 4)):
        print(json.dumps(case(dev, depth, cn, "planes", convert_only=True)))
        print(json.dumps(case(dev, depth, cn, "planes")))
