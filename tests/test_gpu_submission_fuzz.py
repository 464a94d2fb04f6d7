"""Randomised mix of the engine's SUBMISSION paths for one chain shape (K1's hot shape: crops -> resize -> swap -> mul/sub/div -> NCHW):
every path must leave exactly the bits of one plain cvgs_execute launch (which tests/test_gpu_k1.py and the fuzz hold to the oracle):
  host tickets (cvgs_queue_submit + wait), stream-ordered single submits (strict / HYBRID / DEFER_WAIT + stream_wait), groups behind one
  gate (cvgs_queue_submit_many_on: strict on the server via MIN_GROUP, strict through the policy's one-launch route, deferred), and
  cvgs_execute_many on a plain stream -- issued in BURSTS over three streams with disjoint cameras, a producer copy on the stream rewriting
  every frame before its submit, nothing synchronised inside a burst.  CVGS_FUZZ_SUBMIT_N=500 for a long hunt."""
import ctypes as C
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu

DST, CN = (64, 128), 3
PLANE = CN * DST[0] * DST[1]


class Cam:
    def __init__(self, torch, dev, rng, k):
        w, h = int(rng.integers(96, 961)), int(rng.integers(96, 541))
        self.n = int(rng.choice([1, 2, 5, 12, 30, 50, 74, 75, 90]))
        self.crops = H.random_crops(self.n, w, h, wmin=4, wmax=min(300, w), hmin=4, hmax=min(300, h), seed=int(rng.integers(1 << 30)))
        self.pics = [torch.from_numpy(H.random_u8((h, w, 3), seed=int(rng.integers(1 << 30)))).to(dev) for _ in range(2)]
        self.frame = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
        self.out = torch.zeros((self.n, PLANE), dtype=torch.float32, device=dev)
        self.low = cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(self.frame, cvgs.CV_8UC3), self.crops, cvgs.GpuMat.from_tensor(self.out, cvgs.CV_32FC1), DST, CN))
        self.refs = []


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVGS_FUZZ_SUBMIT_N", "12"))))
def test_every_submission_path_leaves_the_bits_of_one_launch(seed):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0xABC000 + seed)
    lib = capi.load_library()
    cams = [Cam(torch, dev, rng, k) for k in range(int(rng.integers(6, 15)))]
    s0 = torch.cuda.current_stream()
    for c in cams:  # the reference bits: one plain launch per picture
        for p in c.pics:
            c.frame.copy_(p)
            capi.check(lib.cvgs_execute(C.byref(c.low.desc), s0.cuda_stream))
            c.refs.append(c.out.clone().view(torch.int32))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(3)]
    q = cvgs.Queue(depth=int(rng.choice([8, 32, 128])), idle_us=float(rng.choice([50.0, 2000.0])))
    D, Hy = cvgs.Queue.DEFER_WAIT, cvgs.Queue.HYBRID
    try:
        for burst in range(int(rng.integers(4, 9))):
            pic = int(rng.integers(2))
            order = list(rng.permutation(len(cams)))
            ops, host_jobs = [], []
            while order:
                kind = str(rng.choice(["host", "on", "on_hybrid", "on_defer", "many_server", "many_policy", "many_defer", "execute_many"]))
                take = 1 if kind in ("host", "on", "on_hybrid", "on_defer") else int(rng.integers(2, 7))
                group = [cams[i] for i in order[:take]]
                order = order[take:]
                if max(c.n for c in group) > 74 and kind in ("on", "on_defer"):
                    kind = "on_hybrid"  # a stream-ordered batch holds at most 74 planes: without HYBRID such a call is refused (by contract)
                ops.append((kind, group, streams[int(rng.integers(3))]))
            torch.cuda.synchronize()
            pend = []  # (queue ticket, stream) of deferred submits
            for kind, group, st in ops:
                with torch.cuda.stream(st):
                    for c in group:  # the producer on the stream: poison the tensor, rewrite the frame
                        c.out.fill_(-3.0)
                        c.frame.copy_(c.pics[pic], non_blocking=True)
                if kind == "host":  # not stream-ordered: the producer must be complete when the submit is made
                    st.synchronize()
                    host_jobs.append(q.submit_lowered(group[0].low))
                elif kind in ("on", "on_hybrid", "on_defer"):
                    t = q.submit_lowered_on(st, group[0].low, {"on": 0, "on_hybrid": Hy, "on_defer": D}[kind])
                    if kind == "on_defer":
                        pend.append((t, st))
                elif kind == "execute_many":
                    capi.check(lib.cvgs_execute_many(cvgs.pack_chains([c.low for c in group]), len(group), st.cuda_stream))
                else:
                    big = max(c.n for c in group) > 74  # (HYBRID: chains the server does not take are launched on the stream)
                    flags = {"many_server": Hy | cvgs.Queue.MIN_GROUP(2), "many_policy": Hy, "many_defer": D | ((Hy | cvgs.Queue.MIN_GROUP(2)) if big else 0)}[kind]
                    t = q.submit_many_on(st, cvgs.Queue.chain_pointers([c.low for c in group]), len(group), flags)
                    if kind == "many_defer" and t != cvgs.Queue.TICKET_DIRECT:
                        pend.append((t, st))
            for t, st in pend:
                q.stream_wait(t, st)
            for t in host_jobs:
                q.wait(t, timeout_s=20.0)
            for st in streams:
                st.synchronize()
            assert q.stats()["error"] == 0
            for kind, group, _ in ops:
                for c in group:
                    assert bool(torch.equal(c.out.view(torch.int32), c.refs[pic])), "seed %d burst %d: %s, %d crops" % (seed, burst, kind, c.n)
    finally:
        q.destroy()
