#!/usr/bin/env python3
"""Ticks of M frames (50 crops each, cfg #2b) through cvgs_execute_many (ONE launch per tick, grid z = chain) on a rotation whose read-touched
set is >= 2 x the Infinity Cache:
  graph   device plane tables, the K ticks captured into HIP graphs and replayed: device time per tick (HIP events around replays)
  eager   host descriptors (a fresh table every call), a one-wave producer kernel on the stream in front of every tick, ONE stream, host wall
          clock incl. the final synchronise + the host's enqueue time per call (what a serving loop pays)
  host    the same call on 16 x 1-crop chains (a kernel of a few microseconds): what the HOST needs per call
usage: bench_tick.py [--m 16] [--frames 96]"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(a):
    import numpy as np
    import torch
    import bench as B
    from cvgpuspeedup_amd import capi, cvgs
    from tests import helpers as H  # noqa: E402
    from cvgpuspeedup_amd import workloads as W
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = capi.load_library()
    M = a.m
    nf = ((a.frames + M - 1) // M) * M
    out = {"env": {k: os.environ[k] for k in os.environ if k.startswith("CVGS_")}, "m": M, "frames": nf}
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    # ---- graph-replayed, device tables ----
    wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M)
    alg = wl.algorithmic_bytes()  # per launch (M frames)
    m = B.measure(wl, max(16, 256 // M), 4, target_s=0.15, min_replays=20, est_step_s=2.5e-6 * M)
    out["graph"] = {"us_per_tick": round(m["step_s"] * 1e6, 3), "us_per_frame": round(m["step_s"] * 1e6 / M, 4), "frac": round(alg / m["step_s"] / 1e9 / 8000.0, 4),
                    "p10_us": round(m["p10_s"] * 1e6, 3), "p90_us": round(m["p90_s"] * 1e6, 3)}
    # ---- eager, the SAME device tables (what eager launches cost by themselves: the difference to the next leg is the host-descriptor path --
    # table build, the kernel reading its descriptors from pinned host memory) ----
    n_ticks_d = max(64, 2048 // M)
    for i in range(n_ticks_d):
        wl.launch(i, s)
    side.synchronize()
    walls = []
    for _ in range(5):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n_ticks_d):
            wl.launch(i, s)
        side.synchronize()
        walls.append((time.perf_counter() - t0) / n_ticks_d)
    out["eager_device_tables"] = {"us_per_tick_wall": round(float(np.median(walls)) * 1e6, 3), "us_per_frame_wall": round(float(np.median(walls)) * 1e6 / M, 4)}
    del wl
    torch.cuda.empty_cache()
    # ---- eager, host descriptors, producer on the stream ----
    wlh = B.Workload(dev, nf, 50, 0, 1, False)
    packs = [cvgs.pack_chains(wlh.chains[g * M:(g + 1) * M]) for g in range(nf // M)]
    n_ticks = max(64, 2048 // M)

    def loop(producer):
        for i in range(n_ticks):
            if producer:
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, s)
            rc = lib.cvgs_execute_many(packs[i % len(packs)], M, s)
            if rc:
                capi.check(rc)
    for producer in (True, False):
        loop(producer)
        side.synchronize()
        walls, hosts, evs = [], [], []
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            e0.record()
            loop(producer)
            e1.record()
            t1 = time.perf_counter()
            side.synchronize()
            t2 = time.perf_counter()
            walls.append((t2 - t0) / n_ticks)
            hosts.append((t1 - t0) / n_ticks)
            evs.append(e0.elapsed_time(e1) * 1e-3 / n_ticks)
        w = float(np.median(walls))
        out["eager_producer" if producer else "eager"] = {"us_per_tick_wall": round(w * 1e6, 3), "us_per_frame_wall": round(w * 1e6 / M, 4),
                                                        "host_enqueue_us_per_tick": round(float(np.median(hosts)) * 1e6, 3), "us_per_tick_events": round(float(np.median(evs)) * 1e6, 3),
                                                        "frac": round(alg / w / 1e9 / 8000.0, 4)}
    ok = True
    for g in range(nf // M):  # every tensor against one plain launch of its chain
        for i in range(g * M, (g + 1) * M):
            got = wlh.outs[i].clone()
            wlh.outs[i].zero_()
            wlh.launch(i, s)
            torch.cuda.synchronize()
            ok = ok and bool(torch.equal(got.view(torch.int32), wlh.outs[i].view(torch.int32)))
    out["bit_identical_to_cvgs_execute"] = ok
    # ---- the host's share: the same call on M one-crop chains (the kernel is a few microseconds) ----
    wl1 = B.Workload(dev, M * 4, 1, 0, 1, False)
    p1 = [cvgs.pack_chains(wl1.chains[g * M:(g + 1) * M]) for g in range(4)]
    for i in range(64):
        capi.check(lib.cvgs_execute_many(p1[i % 4], M, s))
    side.synchronize()
    t0 = time.perf_counter()
    for i in range(512):
        capi.check(lib.cvgs_execute_many(p1[i % 4], M, s))
    t1 = time.perf_counter()
    side.synchronize()
    out["host_us_per_call_1_crop_chains"] = round((t1 - t0) / 512 * 1e6, 3)
    # ... and on M x 50-crop chains with an 8 x 8 target (the full lowering / table work of the headline's call, a kernel of a few microseconds)
    small, keep = [], []
    for f in range(M * 4):
        crops = W.random_crops(50, W.FRAME_4K[0], W.FRAME_4K[1], seed=W.SEED + 77 + f)
        o = torch.zeros((50, 3 * 8 * 8), dtype=torch.float32, device=dev)
        keep.append(o)
        small.append(cvgs.lower(W.k1_chain(cvgs.GpuMat.from_tensor(wlh.frames[f % len(wlh.frames)], cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), dst=(8, 8))))
    p8 = [cvgs.pack_chains(small[g * M:(g + 1) * M]) for g in range(4)]
    for i in range(64):
        capi.check(lib.cvgs_execute_many(p8[i % 4], M, s))
    side.synchronize()
    t0 = time.perf_counter()
    for i in range(512):
        capi.check(lib.cvgs_execute_many(p8[i % 4], M, s))
    t1 = time.perf_counter()
    side.synchronize()
    t2 = time.perf_counter()
    out["host_us_per_call_50_crop_chains_8x8_target"] = round((t1 - t0) / 512 * 1e6, 3)
    out["wall_us_per_call_50_crop_chains_8x8_target"] = round((t2 - t0) / 512 * 1e6, 3)
    print(json.dumps(out), flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--frames", type=int, default=96)
    a = p.parse_args()
    return run(a)


if __name__ == "__main__":
    main()
