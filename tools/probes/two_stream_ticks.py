#!/usr/bin/env python3
"""Ticks of 16 frames on ONE stream against the same ticks alternating over TWO (or more) streams, graph-replayed: does overlapping the ~3.9 us
each launch costs beyond its frames (launch floor, pipeline fill, drain; profiles/r06_a_tick_ablation_m16 vs _m64: t = 3.9 + 2.22 x frames)
with the NEXT tick's body buy the difference?  Throughput only: per-kernel durations stretch when kernels overlap, so the roofline line keeps
the one-stream clock.  usage (GPU box): python tools/probes/two_stream_ticks.py [--m 16] [--frames 96] [--streams 1,2,3,4]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--frames", type=int, default=96)
    p.add_argument("--streams", default="1,2,3,4")
    p.add_argument("--rounds", type=int, default=3)
    a = p.parse_args()
    import numpy as np
    import torch
    import bench as B
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    main_s = torch.cuda.Stream()
    torch.cuda.set_stream(main_s)
    M = a.m
    nf = ((a.frames + M - 1) // M) * M
    wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M)
    ticks = 2 * (nf // M) * 2  # launches per replay (a whole number of rotations, even)
    out = {"m": M, "frames": nf, "launches_per_replay": ticks, "us_per_frame": {}}
    graphs = {}
    for ns in [int(v) for v in a.streams.split(",")]:
        side = [torch.cuda.Stream() for _ in range(ns)]
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            cap = torch.cuda.current_stream()
            for s in side:
                s.wait_stream(cap)
            for i in range(ticks):
                wl.launch(i, side[i % ns].cuda_stream)  # tick i on stream i mod ns: each stream strictly ordered, the streams independent
            for s in side:
                cap.wait_stream(s)
        graphs[ns] = (g, side)
    res = {ns: [] for ns in graphs}
    for g, _ in graphs.values():
        g.replay()
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for ns, (g, _) in graphs.items():
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            reps = 40
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            res[ns].append(e0.elapsed_time(e1) * 1e3 / (reps * ticks * M))
    alg = wl.algorithmic_bytes() / M
    for ns, v in res.items():
        t = float(np.median(v))
        out["us_per_frame"][str(ns)] = {"us": round(t, 4), "frac": round(alg / t / 1e6 / 8.0, 4), "min": round(min(v), 4), "max": round(max(v), 4)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
