#!/usr/bin/env python3
"""Is the dispatch slow-down beside a live server (server_vs_streams.py) a property of the server or of ANY long-running kernel?  One
workgroup that sleeps for 400 ms on its own stream; meanwhile 300 empty kernels on each of 16 other streams, us per kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from cvgpuspeedup_amd import capi  # noqa: E402

from tests import helpers as H  # noqa: E402

torch.cuda.set_device(0)
lib = capi.load_library()
hog = torch.cuda.Stream(priority=-1) if "--high" in sys.argv else torch.cuda.Stream()
streams = [torch.cuda.Stream() for _ in range(16)]


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


print("nothing running        :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
blocks = 767 if "--grid" in sys.argv else 1
if "--poll-uncached" in sys.argv:
    grid = 768 if "--grid" in sys.argv else 1
    H.aid_check(H.testaid().cvgs_debug_poll(None, 400000.0, 1 | (grid << 8 if grid > 1 else 0), hog.cuda_stream))
    time.sleep(0.01)
    print("%d workgroup(s) polling an UNCACHED device word:" % grid, " ".join("%5.1f" % rate(s) for s in streams), flush=True)
elif "--poll-host" in sys.argv or "--poll-device" in sys.argv:
    word = torch.zeros(16, dtype=torch.int64).pin_memory() if "--poll-host" in sys.argv else torch.zeros(16, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    H.aid_check(H.testaid().cvgs_debug_poll(word.data_ptr(), 400000.0, 1, hog.cuda_stream))
    time.sleep(0.01)
    print("one wave polling %s memory:" % ("pinned HOST" if "--poll-host" in sys.argv else "device"), " ".join("%5.1f" % rate(s) for s in streams), flush=True)
else:
    H.testaid().cvgs_debug_occupy(blocks, 256, 0, 400000.0, hog.cuda_stream)
    time.sleep(0.01)
    print("%4d-block sleeper alive :" % blocks, " ".join("%5.1f" % rate(s) for s in streams), flush=True)
hog.synchronize()
print("sleeper gone           :", " ".join("%5.1f" % rate(s) for s in streams), flush=True)
