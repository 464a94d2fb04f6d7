#!/usr/bin/env python3
"""Does a live server grid slow kernel dispatch on OTHER streams?  For each of 10 fresh streams: 300 empty one-wave kernels back to back,
us per kernel (a) with no server, (b) with the queue's server alive and idle, (c) alive and busy (a host thread feeds it)."""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tests import helpers as H  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = B.Workload(dev, 20, 50, 0, 1, False)
lib = capi.load_library()
FIRST = "--queue-first" in sys.argv
N = 24 if "--many" in sys.argv else 10
q0 = cvgs.Queue(depth=128, idle_us=500000.0) if FIRST else None
streams = [torch.cuda.Stream() for _ in range(N)]


def rate(s, n=300):
    h = s.cuda_stream
    for _ in range(20):
        H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
    s.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
    s.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("no server      :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
q = q0 or cvgs.Queue(depth=128, idle_us=500000.0)
q.wait(q.submit_lowered(wl.chains[0]))
print("server idle    :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
stop = [False]
ptrs = cvgs.Queue.chain_pointers([wl.chains[i % 20] for i in range(256)])


def feed():
    while not stop[0]:
        q.wait(q.submit_many(ptrs, 256), 30.0)


th = threading.Thread(target=feed)
th.start()
time.sleep(0.05)
print("server busy    :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
stop[0] = True
th.join()
q.destroy()
torch.cuda.synchronize()
print("server gone    :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
