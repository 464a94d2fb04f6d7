#!/usr/bin/env python3
"""Whole-frame resize chains (K2 / K3 of the reference's tests: tests/resize/test_resize_x_split.cu, test_resize_write.cu):
resize<CV_8UC3, INTER_LINEAR>(frame, size) -> [convertTo<32F, 8U>] -> write / split.  Prints one JSON line per case."""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

PEAK = 8000.0


def events_time(fn, iters, warm=10):
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


def case(dev, src_wh, dst_wh, out, iters):
    sw, sh = src_wh
    dw, dh = dst_wh
    n_buf = max(2, (512 << 20) // (sw * sh * 3 + dw * dh * 12) + 1)
    n_buf = min(n_buf, 24)
    frames = [W.random_u8_torch((sh, sw, 3), 700 + i, dev) for i in range(n_buf)]
    f = cvgs.CV_32FC3
    chains, keep, ops = [], [], None
    for fr in frames:
        m = cvgs.GpuMat.from_tensor(fr, cvgs.CV_8UC3)
        rd = cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, m, (dw, dh))
        if out == "u8_packed":
            o = torch.zeros((dh, dw, 3), dtype=torch.uint8, device=dev)
            ops = [rd, cvgs.convertTo(f, cvgs.CV_8UC3), cvgs.write(cvgs.CV_8UC3, cvgs.GpuMat.from_tensor(o, cvgs.CV_8UC3))]
            wbytes = dw * dh * 3
        elif out == "f32_packed":
            o = torch.zeros((dh, dw, 3), dtype=torch.float32, device=dev)
            ops = [rd, cvgs.write(f, cvgs.GpuMat.from_tensor(o, f))]
            wbytes = dw * dh * 12
        else:  # three separate pitched planes (K2), normalized
            o = torch.zeros((3, dh, dw), dtype=torch.float32, device=dev)
            planes = [cvgs.GpuMat.from_tensor(o[c], cvgs.CV_32FC1) for c in range(3)]
            ops = [rd, cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3]), cvgs.split(f, planes)]
            wbytes = dw * dh * 12
        keep.append(o)
        chains.append(cvgs.lower(ops))
    lib = capi.load_library()
    s = torch.cuda.current_stream().cuda_stream
    st = {"i": 0}

    def launch():
        ch = chains[st["i"] % len(chains)]
        st["i"] += 1
        capi.check(lib.cvgs_execute(C.byref(ch.desc), s))

    t = events_time(launch, iters)
    rbytes = W.tapped_bytes(sw, sh, dw, dh, 3)
    alg = rbytes + wbytes
    return {"case": "%dx%d -> %dx%d %s" % (sw, sh, dw, dh, out), "kernel": cvgs.kernel_name(*ops), "us": round(t * 1e6, 2),
            "algorithmic_bytes": alg, "GB_per_s": round(alg / t / 1e9, 1), "frac_of_8TBs": round(alg / t / 1e9 / PEAK, 4),
            "out_Mpix_per_s": round(dw * dh / t / 1e6, 1), "src_Mpix_per_s": round(sw * sh / t / 1e6, 1)}


def run_all(dev, iters=100):
    res = []
    for src, dst in ((W.FRAME_4K, (1920, 1080)), (W.FRAME_1080P, (3840, 2160)), (W.FRAME_6K, (1280, 720)), (W.FRAME_1080P, (64, 128))):
        for out in ("u8_packed", "f32_packed", "f32_planes"):
            res.append(case(dev, src, dst, out, iters))
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    a = ap.parse_args()
    for r in run_all(torch.device("cuda:0"), a.iters):
        print(json.dumps(r))
