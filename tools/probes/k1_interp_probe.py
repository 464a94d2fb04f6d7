#!/usr/bin/env python3
"""Probe (GPU box): what a K1 tick costs when its program is NOT one of the compile-time ones -- the headline's chain (resize -> cvtColor(RGB2BGR) -> multiply ->
subtract -> divide -> split, a compile-time program) against the same chain with one more stage (-> add), which runs K1's interpreted program.  Ticks of 16
frames x 50 variable crops (cfg #2b), 32 frames in rotation, graph-replayed; us per tick."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402


def build(dev, extra, n_frames=32, tick=16):
    f = cvgs.CV_32FC3
    sets, keep = [], []
    for t in range(n_frames // tick):
        low = []
        for m in range(tick):
            k = t * tick + m
            frame = W.random_u8_torch((2160, 3840, 3), 1000 + k, dev)
            crops = W.random_crops(50, 3840, 2160, seed=500000 + k)
            out = torch.zeros((50, 3 * 128 * 64), dtype=torch.float32, device=dev)
            src = cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3)
            ops = [cvgs.resize(cvgs.CV_8UC3, cvgs.INTER_LINEAR, [src.roi(*c) for c in crops], (64, 128), 50), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                   cvgs.multiply(f, [0.3] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3])]
            ops += extra(f)
            ops.append(cvgs.split(f, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1), (64, 128)))
            low.append(cvgs.lower(ops))
            keep.append((frame, out))
        sets.append((low, cvgs.pack_chains(low)))
    return sets, keep, cvgs.kernel_name(*ops)


def timed(lib, sets, tick=16, reps=8):
    side = torch.cuda.Stream()
    for low, arr in sets:
        capi.check(lib.cvgs_execute_many(arr, tick, torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            for low, arr in sets:
                capi.check(lib.cvgs_execute_many(arr, tick, torch.cuda.current_stream().cuda_stream))
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3 / (reps * len(sets)))
    return sorted(ts)[2]


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = capi.load_library()
    out = {}
    for name, extra in (("compile-time program (mul, sub, div)", lambda f: []), ("+ add (interpreted)", lambda f: [cvgs.add(f, [0.5, 0.25, 0.125])]),
                        ("+ add, multiply (interpreted)", lambda f: [cvgs.add(f, [0.5, 0.25, 0.125]), cvgs.multiply(f, [2.0, 2.0, 2.0])])):
        sets, keep, kname = build(dev, extra)
        out[name] = {"kernel": kname, "us_per_tick": round(timed(lib, sets), 2)}
        del sets, keep
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
