#!/usr/bin/env python3
"""Write-only and read-only HBM ceilings beside the copy ceiling (what a store-dominated kernel -- the mirrored CircularTensor push writes
8 bytes for every byte it reads -- can hope for): torch fill / sum / copy over 1 GiB, events, best of 5."""
import torch

dev = torch.device("cuda:0")
n = 1 << 28  # floats: 1 GiB
a = torch.empty(n, dtype=torch.float32, device=dev)
b = torch.empty(n, dtype=torch.float32, device=dev)


def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e-3)
    return best


t = timed(lambda: a.fill_(1.5))
print("fill  (write only) : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: a.zero_())
print("zero  (write only) : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: torch.sum(a))
print("sum   (read only)  : %.2f TB/s" % (4 * n / t / 1e12))
t = timed(lambda: b.copy_(a))
print("copy  (read+write) : %.2f TB/s of bytes moved" % (8 * n / t / 1e12))
