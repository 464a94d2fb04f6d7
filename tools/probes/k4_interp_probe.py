#!/usr/bin/env python3
"""Probe (GPU box): a K4 tick (16 NV12 4K surfaces x 50 crops -> 16 NCHW tensors, one cvgs_execute_many launch, eager, host descriptors) with its compile-time
program (cvtColor, multiply, subtract, divide) against the same chain with one more stage (-> add), which runs K4's interpreted program.  us per tick."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402
from bench_more import events_time  # noqa: E402


def run(dev, extra, cams=16, n=50, sets=8, iters=100):
    w, h = W.FRAME_4K
    dst = W.DST
    f = cvgs.CV_32FC3
    lib = capi.load_library()
    s = torch.cuda.current_stream()
    lowered, keep, ops = [], [], None
    for i in range(cams * sets):
        buf = W.random_u8_torch((h + h // 2, w), 1800 + i, dev)
        out = torch.zeros((n, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
        luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, buf.data_ptr(), w, owner=buf)
        rects = [(x & ~1, y & ~1, max(4, cw & ~1), max(2, ch & ~1)) for (x, y, cw, ch) in W.random_crops(n, w, h, seed=W.SEED + 1900 + i)]
        ops = [cvgs.read_nv12([luma.nv12_roi(*r) for r in rects], dst, capi.YUV_LIMITED, capi.BT709, False), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
               cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3])] + extra(f)
        ops.append(cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), dst))
        keep += [buf, out]
        lowered.append(cvgs.lower(ops))
    arrs = [cvgs.pack_chains(lowered[k * cams:(k + 1) * cams]) for k in range(sets)]
    state = {"i": 0}

    def many():
        state["i"] += 1
        capi.check(lib.cvgs_execute_many(arrs[state["i"] % sets], cams, s.cuda_stream))
    t = events_time(many, iters)
    return {"kernel": cvgs.kernel_name(*ops), "us_per_tick": round(t * 1e6, 2)}


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    out = {"compile-time program": run(dev, lambda f: []), "+ add (interpreted)": run(dev, lambda f: [cvgs.add(f, [0.5, 0.25, 0.125])])}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
