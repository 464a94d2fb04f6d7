#!/usr/bin/env python3
"""Three measurements bench.py carries in its compact line beside the headline, all on the headline's workload
(50 variable crops of a 4K frame -> [50,3,128,64] fp32, 96 resident frames in rotation: touched set 4.6 x the Infinity Cache):

  stream_ordered()   the reference's contract -- executeOperations(stream, ...) ordered behind a PRODUCER kernel on the stream and in
                     front of whatever follows (include/cvGPUSpeedup.cuh:464-473) -- on the queue (cvgs_queue_submit_on /
                     cvgs_queue_submit_many_on), against the same loop with one cvgs_execute launch per step.
  latency_by_depth() p50 / p99 of one batch's submit -> complete time with 1, 2 and 8 batches in flight (host tickets), and of the
                     hybrid policy's choice for a lone batch (the direct launch).  Reference figure: 18 us for the whole step
                     (README.md:147).
  coexistence()      the use case the reference sells -- pre-processing in front of a network ON THE SAME GPU (README.md:145-155): a
                     stand-in consumer (bf16 GEMM loop on a second stream, one tile per CU and four tiles per CU) runs while the
                     headline is served (a) by the queue's resident server, (b) by graph-replayed launches: pre-processing us per
                     batch, the consumer's slow-down, and whether the server's watchdog ever fires.
"""
import ctypes as C
import os
import sys

if __name__ == "__main__":
    # a live server slows kernel dispatch on the hardware queues that share its command-processor pipe (tools/probes/server_vs_streams.py):
    # with <= 3 queues the caller's and the server's never share one.  Read by the runtime at initialisation; an explicit setting wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402

from tests import helpers as H  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402

HBM = 8000.0


def _pct(a, q):
    a = np.sort(np.asarray(a))
    return float(a[min(len(a) - 1, int(len(a) * q))])


# ---- 1. stream-ordered submission ------------------------------------------------------------------------------------------------
def stream_ordered(wl, steps=1920, reps=5):
    lib = capi.load_library()
    alg = wl.algorithmic_bytes()
    nch = len(wl.chains)
    t = C.c_uint64()
    out = {"runtime": {"GPU_MAX_HW_QUEUES": os.environ.get("GPU_MAX_HW_QUEUES", "unset (4)")},
           "producer": "a one-wave kernel on the stream in front of every submit (stand-in for the decoder / the kernel that writes the frame)",
           "clock": "host wall clock over %d batches incl. the final stream synchronise, median of %d" % (steps, reps)}
    q = cvgs.Queue(depth=128, idle_us=2000.0)

    def timed(fn, streams, n):
        fn()
        for s in streams:
            s.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            for s in streams:
                s.synchronize()
            ts.append((time.perf_counter() - t0) / n)
        return float(np.median(ts)) * 1e6

    try:
        # (a) the loop without a queue: producer, one cvgs_execute, per step, one stream
        s1 = [torch.cuda.Stream()]
        h1 = s1[0].cuda_stream

        def launches():
            for i in range(steps):
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h1)
                rc = lib.cvgs_execute(C.byref(wl.chains[i % nch].desc), h1)
                if rc:
                    capi.check(rc)
        us = timed(launches, s1, steps)
        out["one_launch_per_step"] = {"us": round(us, 3), "frac": round(alg / us / 1e3 / HBM, 4)}

        # (b) a lone strictly ordered stream on the queue: the hybrid policy's choice (the direct launch) and the server
        for name, flags in (("lone_stream_hybrid", cvgs.Queue.HYBRID), ("lone_stream_on_the_server", 0)):
            direct = [0]

            def lone():
                direct[0] = 0
                for i in range(steps // 4):
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h1)
                    rc = lib.cvgs_queue_submit_on(q.handle, C.byref(wl.chains[i % nch].desc), h1, flags, C.byref(t))
                    if rc:
                        capi.check(rc)
                    direct[0] += t.value == cvgs.Queue.TICKET_DIRECT
            us = timed(lone, s1, steps // 4)
            out[name] = {"us": round(us, 3), "frac": round(alg / us / 1e3 / HBM, 4), "direct_launches": direct[0], "of": steps // 4}

        # (c) ticks: G frames behind ONE gate (cvgs_queue_submit_many_on), ticks alternating over two streams (two camera groups)
        s2 = [torch.cuda.Stream(), torch.cuda.Stream()]
        for G in (4, 16):
            groups = [cvgs.Queue.chain_pointers([wl.chains[(g * G + j) % nch] for j in range(G)]) for g in range(nch)]

            def ticks():
                for i in range(steps // G):
                    h = s2[i & 1].cuda_stream
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
                    rc = lib.cvgs_queue_submit_many_on(q.handle, groups[i % len(groups)], G, h, 0, C.byref(t))
                    if rc:
                        capi.check(rc)
            us = timed(ticks, s2, steps // G * G)
            out["ticks_of_%d_on_2_streams" % G] = {"us": round(us, 3), "frac": round(alg / us / 1e3 / HBM, 4)}
        # (d) the same ticks on ONE stream with the completion wait deferred (cvGS::attachQueue(stream, queue, deferWait) + cvGS::fence): the
        #     consumer of tick k is ordered behind it two ticks later -- a pipeline of depth 2, no second stream, nothing the runtime's
        #     stream -> hardware-queue mapping can serialise
        for G in (8, 16):
            groups = [cvgs.Queue.chain_pointers([wl.chains[(g * G + j) % nch] for j in range(G)]) for g in range(nch)]

            def ticks_deferred():
                pend = []
                for i in range(steps // G):
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h1)
                    rc = lib.cvgs_queue_submit_many_on(q.handle, groups[i % len(groups)], G, h1, cvgs.Queue.DEFER_WAIT, C.byref(t))
                    if rc:
                        capi.check(rc)
                    pend.append(t.value)
                    if len(pend) > 2:
                        lib.cvgs_queue_stream_wait(q.handle, pend.pop(0), h1)
                for tk in pend:
                    lib.cvgs_queue_stream_wait(q.handle, tk, h1)
            us = timed(ticks_deferred, s1, steps // G * G)
            out["ticks_of_%d_on_1_stream_wait_deferred_2_ticks" % G] = {"us": round(us, 3), "frac": round(alg / us / 1e3 / HBM, 4)}
        st = q.stats()
        out["queue_error"] = st["error"]
    finally:
        q.destroy()
    torch.cuda.synchronize()
    # (e) the same strict ticks with NO queue: ONE cvgs_execute_many launch per tick on a plain stream (ChainBatch::execute on a stream that
    #     is not attached; what the hybrid policy gives strictly ordered groups on attached streams) -- nothing resident between ticks
    for G, S in ((16, 1), (16, 2)):
        packs = [cvgs.pack_chains([wl.chains[(g * G + j) % nch] for j in range(G)]) for g in range(nch)]
        ss = s1 if S == 1 else s2

        def many():
            for i in range(steps // G):
                h = ss[i % S].cuda_stream
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
                rc = lib.cvgs_execute_many(packs[i % len(packs)], G, h)
                if rc:
                    capi.check(rc)
        us = timed(many, ss, steps // G * G)
        out["ticks_of_%d_one_launch_%d_stream%s" % (G, S, "s" if S > 1 else "")] = {"us": round(us, 3), "frac": round(alg / us / 1e3 / HBM, 4)}
    # every resident frame's tensor against ONE cvgs_execute launch of the same chain
    s = torch.cuda.current_stream().cuda_stream
    ok = True
    for i in range(nch):
        got = wl.outs[i].clone()
        wl.outs[i].zero_()
        wl.launch(i, s)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(got.view(torch.int32), wl.outs[i].view(torch.int32)))
    out["bit_identical_to_cvgs_execute"] = ok
    return out


def stream_ordered_compact(r):
    c = {}
    for k, short in (("ticks_of_16_on_1_stream_wait_deferred_2_ticks", "tick16_deferred"), ("ticks_of_8_on_1_stream_wait_deferred_2_ticks", "tick8_deferred"),
                     ("ticks_of_16_on_2_streams", "tick16x2"), ("ticks_of_4_on_2_streams", "tick4x2"), ("lone_stream_hybrid", "lone_hybrid"),
                     ("lone_stream_on_the_server", "lone_server"), ("one_launch_per_step", "launch"),
                     ("ticks_of_16_one_launch_1_stream", "tick16_many"), ("ticks_of_16_one_launch_2_streams", "tick16_manyx2")):
        if k in r:
            c[short + "_us"] = r[k]["us"]
    best = [r[k]["frac"] for k in ("ticks_of_16_on_1_stream_wait_deferred_2_ticks", "ticks_of_16_on_2_streams", "ticks_of_16_one_launch_2_streams",
                                   "ticks_of_16_one_launch_1_stream") if k in r]
    if best:
        c["frac"] = max(best)  # the best of the 16-frame tick regimes (all: producer on the stream, no host synchronisation)
    c["ok"] = bool(r.get("bit_identical_to_cvgs_execute")) and not r.get("queue_error")
    return c


# ---- 2. latency by queue depth ----------------------------------------------------------------------------------------------------
def latency_by_depth(wl, n=1500):
    lib = capi.load_library()
    nch = len(wl.chains)
    out = {}
    q = cvgs.Queue(depth=128, idle_us=5000.0)
    t = C.c_uint64()
    try:
        q.wait(q.submit_lowered(wl.chains[0]))
        for depth in (1, 2, 8):
            pending, lat = [], []
            for i in range(n + depth):
                if len(pending) == depth:
                    tk, t0 = pending.pop(0)
                    q.wait(tk)
                    lat.append((time.perf_counter() - t0) * 1e6)
                if i < n:
                    t0 = time.perf_counter()
                    pending.append((q.submit_lowered(wl.chains[i % nch]), t0))
            lat = lat[50:]
            out["queue_depth_%d" % depth] = {"p50_us": round(_pct(lat, 0.5), 2), "p99_us": round(_pct(lat, 0.99), 2)}
        # the hybrid policy's choice for a lone batch: one launch on the stream, timed launch -> stream synchronise
        s = torch.cuda.Stream()
        lat = []
        for i in range(600):
            t0 = time.perf_counter()
            capi.check(lib.cvgs_queue_submit_on(q.handle, C.byref(wl.chains[i % nch].desc), s.cuda_stream, cvgs.Queue.HYBRID, C.byref(t)))
            s.synchronize()
            lat.append((time.perf_counter() - t0) * 1e6)
        out["lone_batch_hybrid_direct_launch"] = {"p50_us": round(_pct(lat[50:], 0.5), 2), "p99_us": round(_pct(lat[50:], 0.99), 2),
                                                  "direct": bool(t.value == cvgs.Queue.TICKET_DIRECT)}
        out["queue_error"] = q.stats()["error"]
    finally:
        q.destroy()
    return out


def latency_compact(r):
    return {k.replace("queue_depth_", "d").replace("lone_batch_hybrid_direct_launch", "lone"): [v["p50_us"], v["p99_us"]] for k, v in r.items() if isinstance(v, dict)}


# ---- 3. coexistence with a consumer ---------------------------------------------------------------------------------------------------
def _gemm_setup(dev, n):
    a = torch.randn((n, 8192), dtype=torch.bfloat16, device=dev)
    b = torch.randn((8192, n), dtype=torch.bfloat16, device=dev)
    c = torch.empty((n, n), dtype=torch.bfloat16, device=dev)
    return a, b, c


def _gemm_rate(a, b, c, stream, seconds):
    """TFLOP/s of back-to-back GEMMs on `stream` for about `seconds` (enqueued up front, timed by events)."""
    n = c.shape[0]
    flop = 2.0 * n * n * 8192
    with torch.cuda.stream(stream):
        for _ in range(3):
            torch.matmul(a, b, out=c)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        torch.matmul(a, b, out=c)
        e1.record()
    stream.synchronize()
    one = e0.elapsed_time(e1) * 1e-3
    count = max(8, int(seconds / one))
    return flop, one, count


def coexistence(dev, wl, bench_mod, seconds=2.0, soak_seconds=0.0, queue_flags=0):
    """Returns per consumer size: the consumer's rate alone and beside each submission path, the pre-processing rate alone and beside it."""
    out = {"consumer": "torch.matmul bf16 [n,8192]x[8192,n] back to back on a second stream; n = 4096 (256 output tiles: about one per CU) and "
                       "n = 8192 (1024 tiles: the whole chip several times over)"}
    gs = torch.cuda.Stream()

    # (everything that allocates -- the queue, the captured graph -- exists BEFORE a consumer is enqueued: an allocation synchronises
    #  the device, i.e. waits for seconds of queued GEMMs)
    q = cvgs.Queue(depth=128, idle_us=2000.0, flags=queue_flags)
    out["server_workgroups"] = q.stats()["workgroups"]
    nch = len(wl.chains)
    nq = 256
    ptrs = cvgs.Queue.chain_pointers([wl.chains[i % nch] for i in range(nq)])
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        graph = bench_mod.capture(wl, 256)
        graph.replay()
    side.synchronize()
    q.wait(q.submit_many(ptrs, nq))

    def prep_queue(duration):
        launches0 = q.stats()["server_launches"]
        t0 = time.perf_counter()
        batches = 0
        prev = q.submit_many(ptrs, nq)
        while time.perf_counter() - t0 < duration:
            cur = q.submit_many(ptrs, nq)
            q.wait(prev, timeout_s=30.0)
            prev = cur
            batches += nq
        q.wait(prev, timeout_s=30.0)
        batches += nq
        dt = time.perf_counter() - t0
        st = q.stats()
        return dt / batches * 1e6, st["error"], st["server_launches"] - launches0

    def prep_graph(duration):
        with torch.cuda.stream(side):
            t0 = time.perf_counter()
            launches = 0
            while time.perf_counter() - t0 < duration:
                for _ in range(8):
                    graph.replay()
                side.synchronize()
                launches += 8 * 256
            dt = time.perf_counter() - t0
        return dt / launches * 1e6, 0, 0

    lib = capi.load_library()
    packs = [cvgs.pack_chains([wl.chains[(g * 16 + j) % nch] for j in range(16)]) for g in range(nch)]

    def prep_many(duration):
        """ticks of 16 frames as ONE cvgs_execute_many launch each, eager, 8 ticks between stream synchronisations (no queue, nothing resident)"""
        h = side.cuda_stream
        t0 = time.perf_counter()
        ticks = 0
        while time.perf_counter() - t0 < duration:
            for k in range(8):
                capi.check(lib.cvgs_execute_many(packs[(ticks + k) % len(packs)], 16, h))
            side.synchronize()
            ticks += 8
        dt = time.perf_counter() - t0
        return dt / (ticks * 16) * 1e6, 0, 0

    def prep_paced(duration, period_us=50.0):
        """one batch every 50 us (20,000 batches/s: what 16 cameras at 1250 fps would ask for -- far above any real pipeline, far below the
        server's 450,000/s): the server is alive and mostly idle; returns the batch latency p50 (submit -> host sees it complete)"""
        lat = []
        t0 = time.perf_counter()
        nxt = t0
        i = 0
        while True:
            now = time.perf_counter()
            if now - t0 >= duration:
                break
            if now < nxt:
                continue
            nxt += period_us * 1e-6
            s0 = time.perf_counter()
            q.wait(q.submit_lowered(wl.chains[i % nch]), timeout_s=30.0)
            lat.append((time.perf_counter() - s0) * 1e6)
            i += 1
        st = q.stats()
        return _pct(lat, 0.5), st["error"], len(lat)

    prep_many(0.1)
    alone = {"queue": prep_queue(min(seconds, 1.0))[0], "graph_launches": prep_graph(min(seconds, 1.0))[0], "ticks_of_16_one_launch": prep_many(min(seconds, 1.0))[0]}
    out["paced_latency_alone_p50_us"] = round(prep_paced(0.5)[0], 2)
    out["preprocessing_alone_us_per_batch"] = {k: round(v, 3) for k, v in alone.items()}
    for n in (4096, 8192):
        a, b, c = _gemm_setup(dev, n)
        flop, one, count = _gemm_rate(a, b, c, gs, seconds + 0.5)
        row = {"consumer_alone_TFLOPs": round(flop / one / 1e12, 1), "consumer_ms_per_gemm": round(one * 1e3, 3)}
        for name, fn in (("queue", prep_queue), ("graph_launches", prep_graph), ("ticks_of_16_one_launch", prep_many)):
            torch.cuda.synchronize()
            with torch.cuda.stream(gs):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(count):
                    torch.matmul(a, b, out=c)
                e1.record()
            us, err, launches = fn(seconds)  # runs while the GEMMs execute (they were enqueued for `seconds` + 0.5 s of device time)
            gs.synchronize()
            gemm_with = e0.elapsed_time(e1) * 1e-3 / count
            row[name] = {"preprocessing_us_per_batch": round(us, 3), "preprocessing_slowdown": round(us / alone[name], 2),
                         "consumer_TFLOPs": round(flop / gemm_with / 1e12, 1), "consumer_slowdown": round(gemm_with / one, 3),
                         "watchdog_error": err}
        torch.cuda.synchronize()
        with torch.cuda.stream(gs):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(count):
                torch.matmul(a, b, out=c)
            e1.record()
        p50, err, nb = prep_paced(seconds)
        gs.synchronize()
        gemm_with = e0.elapsed_time(e1) * 1e-3 / count
        row["queue_paced_20k_per_s"] = {"batch_latency_p50_us": round(p50, 2), "batches": nb, "consumer_slowdown": round(gemm_with / one, 3), "watchdog_error": err}
        out["gemm_%d" % n] = row
        del a, b, c
        torch.cuda.empty_cache()
    try:
        _soak(out, dev, gs, prep_queue, soak_seconds)
    finally:
        q.destroy()
    return out


def _soak(out, dev, gs, prep_queue, soak_seconds):
    if soak_seconds > 0:
        a, b, c = _gemm_setup(dev, 8192)
        flop, one, count = _gemm_rate(a, b, c, gs, soak_seconds + 1.0)
        with torch.cuda.stream(gs):
            for _ in range(count):
                torch.matmul(a, b, out=c)
        us, err, launches = prep_queue(soak_seconds)
        gs.synchronize()
        out["soak"] = {"seconds": soak_seconds, "consumer": "gemm_8192 throughout", "preprocessing_us_per_batch": round(us, 3), "watchdog_error": err,
                       "server_launches": launches}


def coexistence_compact(r):
    c = {}
    for n in (4096, 8192):
        g = r.get("gemm_%d" % n)
        if g:
            c["gemm%d" % n] = {"queue_us": g["queue"]["preprocessing_us_per_batch"], "graph_us": g["graph_launches"]["preprocessing_us_per_batch"],
                               "tick16_us": g.get("ticks_of_16_one_launch", {}).get("preprocessing_us_per_batch"),
                               "consumer_slowdown": [g["queue"]["consumer_slowdown"], g["graph_launches"]["consumer_slowdown"]] +
                                                    ([g["ticks_of_16_one_launch"]["consumer_slowdown"]] if "ticks_of_16_one_launch" in g else []),
                               "paced_consumer_slowdown": g.get("queue_paced_20k_per_s", {}).get("consumer_slowdown"),
                               "paced_latency_us": g.get("queue_paced_20k_per_s", {}).get("batch_latency_p50_us"),
                               "watchdog": g["queue"]["watchdog_error"]}
    if "soak" in r:
        c["soak_s"] = r["soak"]["seconds"]
        c["soak_watchdog"] = r["soak"]["watchdog_error"]
    return c


if __name__ == "__main__":
    import argparse

    import bench as B
    p = argparse.ArgumentParser()
    p.add_argument("--soak", type=float, default=0.0)
    p.add_argument("--g-sweep", action="store_true", help="coexistence only, for several server sizes")
    p.add_argument("--json", action="store_true", help="ONE JSON object with the three blocks on the last stdout line (bench.py runs this in a fresh process)")
    p.add_argument("--seconds", type=float, default=2.0, help="coexistence: seconds per leg")
    p.add_argument("--only-stream", action="store_true", help="--json: the stream-ordered block only (tools/perf_gate.py)")
    a = p.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    # rotation sized from TOUCHED bytes like bench.py's headline (96 frames: read-touched set 2.8 x the Infinity Cache)
    wl = B.Workload(dev, max(W.rotation_units(W.k1_touched_per_frame(50, W.FRAME_4K)[0]), 96), 50, 0, 1, False)
    import json
    if a.g_sweep:
        for g in (127, 255, 383, 511, 767):
            r = coexistence(dev, wl, B, seconds=1.0, queue_flags=g << 16)
            print("G %4d | alone: queue %.3f us/batch, paced latency p50 %.1f us |" % (r["server_workgroups"], r["preprocessing_alone_us_per_batch"]["queue"], r["paced_latency_alone_p50_us"]),
                  " | ".join("gemm%d: queue %.3f us (consumer x%.2f), paced lat %.1f us (consumer x%.2f), graph launches %.2f us (consumer x%.2f), ticks of 16 in one launch %.2f us (consumer x%.2f)" % (
                      n, r["gemm_%d" % n]["queue"]["preprocessing_us_per_batch"], r["gemm_%d" % n]["queue"]["consumer_slowdown"],
                      r["gemm_%d" % n]["queue_paced_20k_per_s"]["batch_latency_p50_us"], r["gemm_%d" % n]["queue_paced_20k_per_s"]["consumer_slowdown"],
                      r["gemm_%d" % n]["graph_launches"]["preprocessing_us_per_batch"], r["gemm_%d" % n]["graph_launches"]["consumer_slowdown"],
                      r["gemm_%d" % n]["ticks_of_16_one_launch"]["preprocessing_us_per_batch"], r["gemm_%d" % n]["ticks_of_16_one_launch"]["consumer_slowdown"]) for n in (4096, 8192)), flush=True)
        sys.exit(0)
    if a.json:
        r = {"stream_ordered": stream_ordered(wl)}
        if not a.only_stream:
            r["queue_latency_by_depth"] = latency_by_depth(wl)
            r["coexistence"] = coexistence(dev, wl, B, seconds=a.seconds, soak_seconds=a.soak)
        print(json.dumps(r), flush=True)
        sys.exit(0)
    print(json.dumps({"stream_ordered": stream_ordered(wl)}, indent=1), flush=True)
    print(json.dumps({"queue_latency_by_depth": latency_by_depth(wl)}, indent=1), flush=True)
    print(json.dumps({"coexistence": coexistence(dev, wl, B, soak_seconds=a.soak)}, indent=1), flush=True)
