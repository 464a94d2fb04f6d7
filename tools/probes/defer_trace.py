#!/usr/bin/env python3
"""One stream, deferred waits trailing by `lag` batches (CVGS_QUEUE_DEBUG=2): gate-kernel starts and wait-kernel spans per ticket.
Prints the spacing of consecutive gate kernels, the wait kernels' durations, and the time from a gate's opening to the end of the wait
that covers it."""
import ctypes as C
import os
import sys
import time

os.environ["CVGS_QUEUE_DEBUG"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as B  # noqa: E402
from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tests import helpers as H  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
wl = B.Workload(dev, 20, 50, 0, 1, False)
lib = capi.load_library()
q = cvgs.Queue(depth=128, idle_us=2000.0)
out = (C.c_uint64 * 16)()
lib.cvgs_queue_profile(q.handle, out)
tr = np.ctypeslib.as_array((C.c_uint64 * 16384).from_address(int(out[15])))
gate, wait = tr[:8192].reshape(4096, 2), tr[8192:].reshape(4096, 2)
t = C.c_uint64()
for lag in (4, 16):
    for producer in (1, 0):
        s = torch.cuda.Stream()
        h = s.cuda_stream
        for rep in range(2):
            torch.cuda.synchronize()
            first = q.stats()["submitted"]
            pend = []
            t0 = time.perf_counter()
            for i in range(600):
                if producer:
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, h)
                capi.check(lib.cvgs_queue_submit_on(q.handle, C.byref(wl.chains[i % 20].desc), h, cvgs.Queue.DEFER_WAIT, C.byref(t)))
                pend.append(t.value)
                if len(pend) > lag:
                    lib.cvgs_queue_stream_wait(q.handle, pend.pop(0), h)
            for tk in pend:
                lib.cvgs_queue_stream_wait(q.handle, tk, h)
            s.synchronize()
            wall = (time.perf_counter() - t0) / 600 * 1e6
        tk = np.arange(first + 8, first + 592)
        g = gate[tk & 4095, 0].astype(np.int64)
        w0, w1 = wait[tk & 4095, 0].astype(np.int64), wait[tk & 4095, 1].astype(np.int64)
        ok = w0 > 0
        sp = np.diff(g) / 100.0
        wd = (w1 - w0)[ok] / 100.0
        print("lag %2d producer %d: wall %6.2f us/batch | gate-to-gate med %6.1f p90 %6.1f max %7.1f | wait kernels launched for %d of %d tickets, duration med %6.1f p90 %6.1f max %7.1f | gate open -> its wait's end med %6.1f" % (
            lag, producer, wall, np.median(sp), np.percentile(sp, 90), sp.max(), int(ok.sum()), len(tk), np.median(wd) if ok.any() else 0, np.percentile(wd, 90) if ok.any() else 0,
            wd.max() if ok.any() else 0, np.median((w1 - g)[ok]) / 100.0 if ok.any() else 0), flush=True)
q.destroy()
