#!/usr/bin/env python3
"""Ticks of M frames (50 crops each, cfg #2b) through cvgs_execute_many -- the tick kernel (csrc/k_tick.hip) or, with CVGS_TICK=0, the grid
kernel it replaced -- on a rotation whose read-touched set is >= 2 x the Infinity Cache:
  graph   device plane tables, the K ticks captured into HIP graphs and replayed: device time per tick (HIP events around replays)
  eager   host descriptors (a fresh table every call), a one-wave producer kernel on the stream in front of every tick, ONE stream, host wall
          clock incl. the final synchronise + the host's enqueue time per call (what a serving loop pays)
The environment knobs are read once per process: --sweep runs this file once per setting in a subprocess.
usage: bench_tick.py [--m 16] [--frames 96] [--sweep]"""
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
    from cvgpuspeedup_amd import workloads as W
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = capi.load_library()
    M = a.m
    nf = ((a.frames + M - 1) // M) * M
    out = {"env": {k: os.environ[k] for k in os.environ if k.startswith("CVGS_TICK")}, "m": M, "frames": nf}
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    # ---- graph-replayed, device tables ----
    wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M)
    alg = wl.algorithmic_bytes()  # per launch (M frames)
    before = lib.cvgs_debug_tick_launches()
    m = B.measure(wl, max(16, 256 // M), 4, target_s=0.15, min_replays=20, est_step_s=2.5e-6 * M)
    out["graph"] = {"us_per_tick": round(m["step_s"] * 1e6, 3), "us_per_frame": round(m["step_s"] * 1e6 / M, 4), "frac": round(alg / m["step_s"] / 1e9 / 8000.0, 4),
                    "p10_us": round(m["p10_s"] * 1e6, 3), "p90_us": round(m["p90_s"] * 1e6, 3), "tick_kernel_launches_captured": lib.cvgs_debug_tick_launches() - before}
    del wl
    torch.cuda.empty_cache()
    # ---- eager, host descriptors, producer on the stream ----
    wlh = B.Workload(dev, nf, 50, 0, 1, False)
    packs = [cvgs.pack_chains(wlh.chains[g * M:(g + 1) * M]) for g in range(nf // M)]
    n_ticks = max(64, 2048 // M)

    def loop(producer):
        for i in range(n_ticks):
            if producer:
                lib.cvgs_debug_occupy(1, 64, 0, 0.0, s)
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
    out["tick_kernel_launches"] = lib.cvgs_debug_tick_launches()
    print(json.dumps(out), flush=True)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--frames", type=int, default=96)
    p.add_argument("--sweep", action="store_true")
    a = p.parse_args()
    if not a.sweep:
        return run(a)
    settings = [{"CVGS_TICK": "0"}, {}]
    for rows in (4, 8, 16, 32):
        for wgs in (2, 3, 4):
            settings.append({"CVGS_TICK_ROWS": str(rows), "CVGS_TICK_WGS_PER_CU": str(wgs)})
    settings += [{"CVGS_TICK_ST": "0"}, {"CVGS_TICK_ST": "0", "CVGS_TICK_ROWS": "8", "CVGS_TICK_WGS_PER_CU": "4"}]
    for st in settings:
        env = dict(os.environ)
        env.update(st)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--m", str(a.m), "--frames", str(a.frames)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else json.dumps({"env": st, "error": r.stderr[-400:]}), flush=True)


if __name__ == "__main__":
    main()
