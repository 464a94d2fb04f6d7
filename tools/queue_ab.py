#!/usr/bin/env python3
"""A/B of the device-side descriptor queue (cvgs_queue_*) on the headline workload (cfg2b: 50 variable crops of a 4K frame ->
[50,3,128,64] fp32, one submit per frame, 20 resident frames in rotation), next to one cvgs_execute launch per frame.

Per variant: R replays of { submit_many(N batches); wait(last) } timed with the host's wall clock (the whole path: host
lowering + ring write + PCIe poll + server), median us per batch; the tensors of the last pass are compared bit for bit with
what cvgs_execute wrote for the same frames.  Variants = store flavour x tap-load flavour x worker workgroups:
  st 2 = sc1 16-byte transposed stores (the product), 1 = sc1 dword stores, 0 = nt dword stores WITHOUT a safe publish (upper bound)
  ld 1 = sc1 tap loads (the product), 0 = plain cached loads (upper bound; may serve stale source lines)
usage: queue_ab.py [--batches 260] [--replays 40] [--variants st,ld,G;...]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402


def build(dev, n_frames, crops):
    fw, fh = W.FRAME_4K
    frames, outs, chains = [], [], []
    plane = 3 * W.DST[0] * W.DST[1]
    for f in range(n_frames):
        seed = W.SEED + f
        frame = W.random_u8_torch((fh, fw, 3), seed, dev)
        cr = W.random_crops(crops, fw, fh, seed=seed + 500000)
        out = torch.zeros((crops, plane), dtype=torch.float32, device=dev)
        ops = W.k1_chain(cvgs.GpuMat.from_tensor(frame, cvgs.CV_8UC3), cr, cvgs.GpuMat.from_tensor(out, cvgs.CV_32FC1))
        frames.append(frame)
        outs.append(out)
        chains.append(cvgs.lower(ops))
    return frames, outs, chains


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=260)
    ap.add_argument("--replays", type=int, default=40)
    ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--crops", type=int, default=50)
    ap.add_argument("--depth", type=int, default=64)
    ap.add_argument("--retire-between", action="store_true", help="sleep 2 ms between replays (the server retires; each replay pays a launch)")
    ap.add_argument("--sustained", action="store_true",
                    help="PMC runs in the regime bench.py times: the replays are pipelined one ahead (a 128-deep ring stays full), nothing else is "
                         "submitted -- ONE server call serves (replays + 1) x batches batches; prints the count to divide the call's counters by")
    ap.add_argument("--variants", default="2,1,0;1,1,0;0,1,0;2,0,0;0,0,0;2,1,1024;2,1,512;2,1,256")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = capi.load_library()
    frames, outs, chains = build(dev, a.frames, a.crops)
    s = torch.cuda.current_stream().cuda_stream
    # reference results + the launch-per-frame rate (eager from python is host bound; bench.py has the graph-replayed figure)
    for c in chains:
        capi.check(lib.cvgs_execute(C_byref(c), s))
    torch.cuda.synchronize()
    want = [o.clone() for o in outs]
    n = a.batches
    order = [chains[i % len(chains)] for i in range(n)]
    ptrs = cvgs.Queue.chain_pointers(order)
    for var in a.variants.split(";"):
        st, ld, G = [int(x) for x in var.split(",")]
        flags = ((st + 1) << 8) | ((ld + 1) << 12) | (G << 16)
        for o in outs:
            o.zero_()
        torch.cuda.synchronize()
        try:
            q = cvgs.Queue(depth=a.depth, idle_us=300.0, flags=flags)
        except capi.CvgsError as ex:
            print(json.dumps({"variant": var, "error": str(ex)}), flush=True)
            continue
        if a.sustained:
            try:
                q.wait(q.submit_many(ptrs, n), 30.0)  # warm-up: its own (short) server call once the server has retired
                time.sleep(0.01)
                base = q.stats()
                t0 = time.perf_counter()
                prev = q.submit_many(ptrs, n)
                for r in range(a.replays):
                    cur = q.submit_many(ptrs, n)
                    q.wait(prev, 30.0)
                    prev = cur
                q.wait(prev, 30.0)
                dt = time.perf_counter() - t0
                time.sleep(0.01)
                st_ = q.stats()
                ok = all(torch.equal(o, w) for o, w in zip(outs, want))
                print(json.dumps({"variant": {"st": st, "ld": ld, "G": st_["workgroups"]}, "sustained": True, "batches_in_the_timed_server_calls": st_["submitted"] - base["submitted"],
                                  "server_calls": st_["server_launches"] - base["server_launches"], "warmup_batches_in_their_own_call": n, "us_per_batch": round(dt / ((a.replays + 1) * n) * 1e6, 3),
                                  "bit_identical_to_cvgs_execute": bool(ok), "error": st_["error"]}), flush=True)
            finally:
                q.destroy()
            continue
        try:
            times = []
            for r in range(a.replays + 3):
                t0 = time.perf_counter()
                last = q.submit_many(ptrs, n)
                t1 = time.perf_counter()
                q.wait(last, 5.0)
                t2 = time.perf_counter()
                if r >= 3:
                    times.append(((t2 - t0) / n * 1e6, (t1 - t0) / n * 1e6))
                if a.retire_between:
                    time.sleep(0.002)  # let the server retire: every replay is one server launch (cold clocks included)
            time.sleep(0.002)
            prof = q.profile()
            prof["server"] = {k: q.stats()[k] for k in ("janitor_rounds", "server_ticks_100MHz", "host_writes_device_memory")}
            # single-batch latency: submit one, wait, server idle in between
            lat = []
            for r in range(30):
                time.sleep(0.001)
                t0 = time.perf_counter()
                q.wait(q.submit_lowered(chains[r % len(chains)]), 5.0)
                lat.append((time.perf_counter() - t0) * 1e6)
            warm = []  # the server is alive: submit, wait, submit ... back to back
            for r in range(60):
                t0 = time.perf_counter()
                q.wait(q.submit_lowered(chains[r % len(chains)]), 5.0)
                warm.append((time.perf_counter() - t0) * 1e6)
            ok = all(torch.equal(o, w) for o, w in zip(outs, want))
            st_ = q.stats()
            tot = np.array([t[0] for t in times])
            sub = np.array([t[1] for t in times])
            print(json.dumps({"variant": {"st": st, "ld": ld, "G": st_["workgroups"]}, "us_per_batch_median": round(float(np.median(tot)), 3),
                              "us_per_batch_p10": round(float(np.percentile(tot, 10)), 3), "us_per_batch_p90": round(float(np.percentile(tot, 90)), 3),
                              "host_submit_us_per_batch": round(float(np.median(sub)), 3), "single_batch_latency_us_median": round(float(np.median(lat)), 2), "warm_latency_us_median": round(float(np.median(warm[10:])), 2),
                              "bit_identical_to_cvgs_execute": bool(ok), "server_launches": st_["server_launches"], "error": st_["error"],
                              "Gpix_per_s": round(a.crops * 8192 / float(np.median(tot)) / 1e3, 1), "profile_last_replay": prof}), flush=True)
        finally:
            q.destroy()


def C_byref(lowered):
    import ctypes
    return ctypes.byref(lowered.desc)


if __name__ == "__main__":
    main()
