#!/usr/bin/env python3
"""Whole-frame resize -> normalize -> NCHW (one "crop" = the frame, a frame-sized target) on the descriptor queue against one launch per frame:
4K u8c3 -> 1920x1080 and -> 1280x720 fp32 planar, 12 resident frames in rotation (HBM-sourced)."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tests import helpers as H  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    lib = capi.load_library()
    n_frames = 12
    frames = [torch.randint(0, 256, (2160, 3840, 3), dtype=torch.uint8, device=dev) for _ in range(n_frames)]
    for dst in ((1920, 1080), (1280, 720), (640, 640)):
        outs = [torch.zeros((1, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev) for _ in range(n_frames)]
        low = [cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(f, cvgs.CV_8UC3), [(0, 0, 3840, 2160)], cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), dst, 3))
               for f, o in zip(frames, outs)]
        alg = 3840 * 2160 * 3 * min(1.0, 4.0 * dst[0] * dst[1] / (3840 * 2160)) + dst[0] * dst[1] * 12
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for lc in low:
                capi.check(lib.cvgs_execute(C.byref(lc.desc), s.cuda_stream))
            s.synchronize()
            refs = [o.clone() for o in outs]
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for k in range(48):
                    capi.check(lib.cvgs_execute(C.byref(low[k % n_frames].desc), torch.cuda.current_stream().cuda_stream))
            g.replay()
            s.synchronize()
            t0 = time.perf_counter()
            for _ in range(20):
                g.replay()
            s.synchronize()
            us_launch = (time.perf_counter() - t0) / (20 * 48) * 1e6
        q = cvgs.Queue(depth=64, idle_us=2000.0)
        try:
            ptrs = cvgs.Queue.chain_pointers([low[k % n_frames] for k in range(48)])
            for o in outs:
                o.zero_()
            torch.cuda.synchronize()
            q.wait(q.submit_many(ptrs, 48))
            ok = all(bool(torch.equal(o.view(torch.int32), r.view(torch.int32))) for o, r in zip(outs, refs))
            prev = q.submit_many(ptrs, 48)
            t0 = time.perf_counter()
            for _ in range(20):
                cur = q.submit_many(ptrs, 48)
                q.wait(prev)
                prev = cur
            q.wait(prev)
            us_queue = (time.perf_counter() - t0) / (21 * 48) * 1e6
            err = q.stats()["error"]
        finally:
            q.destroy()
        print("4K u8c3 -> %4dx%-4d fp32 planar normalized: one launch per frame %.2f us (%.2f of 8 TB/s) | one queue submit per frame %.2f us (%.2f) | bit-identical %s, queue error %d"
              % (dst[0], dst[1], us_launch, alg / us_launch / 8e6, us_queue, alg / us_queue / 8e6, ok, err), flush=True)


if __name__ == "__main__":
    main()
