#!/usr/bin/env python3
"""Where a stream-ordered batch spends its time (CVGS_QUEUE_DEBUG=2): per ticket the gate kernel's start and the moment it saw
the batch complete, 100 MHz ticks.  latency = complete - start (gate store -> workers -> rows -> flag -> seen by the gate kernel);
gap = next gate kernel of the SAME stream's start - this one's completion (kernel exit -> producer -> next gate kernel start)."""
import ctypes as C
import os
import sys

if __name__ == "__main__":
    # a live server slows kernel dispatch on the hardware queues that share its command-processor pipe (tools/probes/server_vs_streams.py):
    # with <= 3 queues the caller's and the server's never share one.  Read by the runtime at initialisation; an explicit setting wins.
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "3")
import time

os.environ["CVGS_QUEUE_DEBUG"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as B  # noqa: E402
from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    wl = B.Workload(dev, 20, 50, 0, 1, False)
    lib = capi.load_library()
    q = cvgs.Queue(depth=128, idle_us=2000.0)
    out = (C.c_uint64 * 16)()
    lib.cvgs_queue_profile(q.handle, out)
    trace = np.ctypeslib.as_array((C.c_uint64 * 8192).from_address(int(out[15]))).reshape(4096, 2)
    t = C.c_uint64()
    for n_streams, group, producer in ((1, 1, True), (4, 1, True), (4, 1, False), (16, 1, True), (1, 4, True), (2, 4, True), (2, 16, True)):
        streams = [torch.cuda.Stream() for _ in range(n_streams)]
        ptrs = [cvgs.Queue.chain_pointers([wl.chains[(g * group + j) % 20] for j in range(group)]) for g in range(20)]
        owner = {}
        for rep in range(2):
            torch.cuda.synchronize()
            first = q.stats()["submitted"]
            w0 = time.perf_counter()
            for i in range(600 // group):
                s = streams[i % n_streams].cuda_stream
                if producer:
                    H.testaid().cvgs_debug_occupy(1, 64, 0, 0.0, s)
                capi.check(lib.cvgs_queue_submit_many_on(q.handle, ptrs[i % 20], group, s, 0, C.byref(t)))
                owner[t.value] = i % n_streams
            for st in streams:
                st.synchronize()
            wall = (time.perf_counter() - w0) / 600 * 1e6
        last = q.stats()["submitted"]
        tk = np.arange(max(first, last - 4000), last)
        st_, dn = trace[tk & 4095, 0].astype(np.int64), trace[tk & 4095, 1].astype(np.int64)
        lat = (dn - st_) / 100.0
        # per stream: consecutive gate kernels (the LAST ticket of each group identifies the launch)
        gaps = []
        prev = {}
        for k in sorted(owner):
            if k < tk[0]:
                continue
            o = owner[k]
            if o in prev:
                gaps.append((int(trace[k & 4095, 0]) - prev[o]) / 100.0)
            prev[o] = int(trace[k & 4095, 1])
        span = (dn.max() - st_.min()) / 100.0 / len(tk)
        print("streams %2d group %2d producer %d: wall %6.2f us/batch | device span %6.2f us/batch | batch latency med %6.1f p90 %6.1f us | "
              "same-stream gap (complete -> next gate start) med %6.1f p90 %6.1f us" % (
                  n_streams, group, producer, wall, span, np.median(lat), np.percentile(lat, 90), np.median(gaps), np.percentile(gaps, 90)), flush=True)
    q.destroy()


if __name__ == "__main__":
    main()
