#!/usr/bin/env python3
"""What the stream-ordered submit (cvgs_queue_submit_on) delivers per 50-crop batch of the headline workload, by regime:
  strict  S streams   every submit holds its stream until its batch is complete (the reference's contract); S batches can overlap
  defer   1 stream    the gate orders each batch behind the stream, the completion wait is enqueued `lag` submits later
  launch  S streams   one cvgs_execute per step on the same streams (the drop-in path without a queue), eager
`--producer`: a one-wave kernel in front of every submit on its stream (a stand-in for the decoder / the kernel that writes the frame).
`--threads T`: T host threads, each driving its share of the streams (a multi-camera host).  Wall clock per step, end to end."""
import argparse
import ctypes as C
import os
import sys

if __name__ == "__main__":
    # a live server slows kernel dispatch on the hardware queues that share its command-processor pipe (tools/probes/server_vs_streams.py):
    # with <= 3 queues the caller's and the server's never share one.  Read by the runtime at initialisation; an explicit setting wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tests import helpers as H  # noqa: E402


def run(wl, q, lib, mode, n_streams, steps, producer, lag, threads, hybrid=False):
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    handles = [s.cuda_stream for s in streams]
    nch = len(wl.chains)
    flags = (cvgs.Queue.DEFER_WAIT if mode == "defer" else 0) | (cvgs.Queue.HYBRID if hybrid else 0)
    direct = [0]

    def drive(tid):
        t = C.c_uint64()
        mine = [k for k in range(n_streams) if k % threads == tid]
        pend = []
        for i in range(steps):
            for k in mine:
                s = handles[k]
                ch = wl.chains[(i * n_streams + k) % nch]
                if producer:
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, s)
                if mode == "launch":
                    rc = lib.cvgs_execute(C.byref(ch.desc), s)
                else:
                    rc = lib.cvgs_queue_submit_on(q.handle, C.byref(ch.desc), s, flags, C.byref(t))
                    if t.value == cvgs.Queue.TICKET_DIRECT:
                        direct[0] += 1
                    elif mode == "defer":
                        pend.append(t.value)
                        if len(pend) > lag:
                            lib.cvgs_queue_stream_wait(q.handle, pend.pop(0), s)
                if rc:
                    return
        for tk in pend:
            lib.cvgs_queue_stream_wait(q.handle, tk, handles[mine[0]])

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ths = [threading.Thread(target=drive, args=(tid,)) for tid in range(threads)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        for s in streams:
            s.synchronize()
        return (time.perf_counter() - t0) / (steps * n_streams)

    once()
    ts = sorted(once() for _ in range(5))
    return ts[len(ts) // 2] * 1e6, direct[0]


def run_ticks_deferred(wl, q, lib, group, ticks, producer, lag=2):
    """ONE stream, `group` frames per tick behind one gate, DEFER_WAIT: the tick's consumer is ordered by cvgs_queue_stream_wait `lag`
    ticks later (what cvGS::attachQueue(stream, queue, deferWait=true) + cvGS::fence(stream) spell)"""
    s = torch.cuda.Stream()
    h = s.cuda_stream
    nch = len(wl.chains)
    groups = [cvgs.Queue.chain_pointers([wl.chains[(g * group + j) % nch] for j in range(group)]) for g in range(max(1, nch // group) + 1)]
    t = C.c_uint64()

    def once():
        torch.cuda.synchronize()
        pend = []
        t0 = time.perf_counter()
        for i in range(ticks):
            if producer:
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
            capi.check(lib.cvgs_queue_submit_many_on(q.handle, groups[i % len(groups)], group, h, cvgs.Queue.DEFER_WAIT, C.byref(t)))
            pend.append(t.value)
            if len(pend) > lag:
                lib.cvgs_queue_stream_wait(q.handle, pend.pop(0), h)
        for tk in pend:
            lib.cvgs_queue_stream_wait(q.handle, tk, h)
        s.synchronize()
        return (time.perf_counter() - t0) / (ticks * group)

    once()
    ts = sorted(once() for _ in range(5))
    return ts[len(ts) // 2] * 1e6


def run_ticks(wl, q, lib, group, n_streams, ticks, producer):
    """`group` frames per tick behind ONE gate (cvgs_queue_submit_many_on), strict; ticks alternate over n_streams streams."""
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    nch = len(wl.chains)
    groups = [cvgs.Queue.chain_pointers([wl.chains[(g * group + j) % nch] for j in range(group)]) for g in range(max(1, nch // group) + 1)]
    t = C.c_uint64()

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(ticks):
            s = streams[i % n_streams].cuda_stream
            if producer:
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, s)
            capi.check(lib.cvgs_queue_submit_many_on(q.handle, groups[i % len(groups)], group, s, 0, C.byref(t)))
        for st in streams:
            st.synchronize()
        return (time.perf_counter() - t0) / (ticks * group)

    once()
    ts = sorted(once() for _ in range(5))
    return ts[len(ts) // 2] * 1e6


def run_ticks_many(wl, lib, group, n_streams, ticks, producer):
    """`group` frames per tick as ONE cvgs_execute_many launch on a plain stream (no queue, no server, strictly stream-ordered); ticks
    alternate over n_streams streams."""
    streams = [torch.cuda.Stream() for _ in range(n_streams)]
    nch = len(wl.chains)
    groups = [cvgs.pack_chains([wl.chains[(g * group + j) % nch] for j in range(group)]) for g in range(max(1, nch // group) + 1)]

    def once():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(ticks):
            s = streams[i % n_streams].cuda_stream
            if producer:
                H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, s)
            capi.check(lib.cvgs_execute_many(groups[i % len(groups)], group, s))
        for st in streams:
            st.synchronize()
        return (time.perf_counter() - t0) / (ticks * group)

    once()
    ts = sorted(once() for _ in range(5))
    return ts[len(ts) // 2] * 1e6


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--steps", type=int, default=400)
    p.add_argument("--producer", action="store_true")
    p.add_argument("--depth", type=int, default=128, help="ring slots")
    p.add_argument("--quick", action="store_true", help="the diagnostic subset only")
    p.add_argument("--only-many", action="store_true", help="the cvgs_execute_many ticks only (no queue is created)")
    p.add_argument("--frames", type=int, default=20, help="resident frames (a tick's chains must be distinct frames: >= the largest tick)")
    a = p.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    wl = B.Workload(dev, a.frames, 50, 0, 1, False)
    lib = capi.load_library()
    alg = wl.algorithmic_bytes()
    if a.only_many:
        many_ticks(a, wl, lib, alg)
        return
    q = cvgs.Queue(depth=a.depth, idle_us=2000.0)
    rows = []
    try:
        full = (("launch", 1, 1, 0, False), ("launch", 8, 1, 0, False), ("launch", 8, 4, 0, False),
                                     ("strict", 1, 1, 0, False), ("strict", 1, 1, 0, True), ("strict", 4, 1, 0, False), ("strict", 8, 1, 0, False),
                                     ("strict", 16, 1, 0, False), ("strict", 16, 2, 0, False), ("strict", 16, 4, 0, False), ("strict", 32, 4, 0, False),
                                     ("strict", 32, 8, 0, False), ("defer", 1, 1, 4, False), ("defer", 1, 1, 16, False), ("defer", 1, 1, 64, False),
                                     ("defer", 4, 4, 32, False))
        quick = (("strict", 1, 1, 0, False), ("strict", 4, 1, 0, False), ("strict", 16, 1, 0, False), ("defer", 1, 1, 4, False), ("defer", 1, 1, 16, False))
        for mode, S, T, lag, hyb in (quick if a.quick else full):
            us, nd = run(wl, q, lib, mode, S, max(20, a.steps // S), a.producer, lag, T, hyb)
            err = q.stats()["error"]
            if err:
                print("  !! queue error word %d in this configuration (its figure is void); recovering: %d batches lost" % (err, q.recover()), flush=True)
            rows.append((mode, S, T, lag, hyb, us, nd))
            print("%-7s streams %2d host threads %d lag %2d hybrid %d : %7.3f us per 50-crop batch  frac %.3f  (direct launches: %d)" % (
                mode, S, T, lag, hyb, us, alg / (us * 1e-6) / 8e12, nd), flush=True)
        for group, S in (((4, 1), (4, 2), (8, 2), (16, 2)) if a.quick else ((4, 1), (4, 2), (8, 1), (8, 2), (16, 1), (16, 2), (16, 4), (32, 2), (64, 2))):
            us = run_ticks(wl, q, lib, group, S, max(10, a.steps // group), a.producer)
            print("tick    %2d frames behind one gate, %d stream(s), 1 host thread    : %7.3f us per 50-crop batch  frac %.3f" % (group, S, us, alg / (us * 1e-6) / 8e12), flush=True)
            err = q.stats()["error"]
            if err:
                print("  !! queue error word %d; recovering: %d batches lost" % (err, q.recover()), flush=True)
        for group, lag in ((8, 2), (16, 1), (16, 2), (16, 4), (32, 2)):
            us = run_ticks_deferred(wl, q, lib, group, max(10, a.steps // group), a.producer, lag)
            print("tick    %2d frames behind one gate, ONE stream, deferred wait trailing %d ticks : %7.3f us per 50-crop batch  frac %.3f" % (group, lag, us, alg / (us * 1e-6) / 8e12), flush=True)
        st = q.stats()
        print("queue error word:", st["error"])
        ok = B.queue_outputs_match_execute(wl)
        print("every frame bit-identical to cvgs_execute:", ok)
    finally:
        q.destroy()
    torch.cuda.synchronize()
    many_ticks(a, wl, lib, alg)


def many_ticks(a, wl, lib, alg):
    # the same ticks with no queue at all: one cvgs_execute_many launch per tick (what ChainBatch::execute does on a stream that is not attached)
    for group, S in ((4, 1), (8, 1), (8, 2), (16, 1), (16, 2), (16, 4), (32, 1), (32, 2), (64, 1), (64, 2)):
        if group > len(wl.chains):
            continue  # (chains of one launch must be independent: distinct frames)
        us = run_ticks_many(wl, lib, group, S, max(10, a.steps // group), a.producer)
        print("tick    %2d frames in ONE cvgs_execute_many launch, %d plain stream(s), no queue : %7.3f us per 50-crop batch  frac %.3f" % (group, S, us, alg / (us * 1e-6) / 8e12), flush=True)


if __name__ == "__main__":
    main()
