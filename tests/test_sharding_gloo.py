"""The N > 1 path on CPU: world_size 2 over gloo.  Each rank computes the crops of ITS shard (the CPU oracle stands
in for the kernel here -- this test is about the sharding + all-gather layout, not the arithmetic) into its slice of
the full tensor, the in-place all-gather assembles it, and every rank must hold exactly the single-process result."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from cvgpuspeedup_amd import cvgs, sharding
from tests import helpers as H


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _reference(frame, crops):
    from oracle import oracle_binding as ob
    out = np.zeros((len(crops), 3 * 64 * 128), np.float32)
    if crops:
        ob.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops,
                                         cvgs.GpuMat.from_array(out, cvgs.CV_32FC1))))
    return out


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        frame = H.random_u8((240, 320, 3), seed=5)
        crops = H.random_crops(n_items, 320, 240, seed=6, wmin=4, wmax=150, hmin=4, hmax=120)
        lo, hi = sharding.shard_bounds(n_items, world, rank)
        full = torch.zeros((n_items, 3 * 64 * 128), dtype=torch.float32)
        full[lo:hi] = torch.from_numpy(_reference(frame, crops[lo:hi]))
        sharding.gather_shards(full, n_items, dist)
        expect = _reference(frame, crops)
        q.put((rank, bool((full.numpy().view(np.uint32) == expect.view(np.uint32)).all())))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n_items", [8, 7])
def test_sharded_k1_allgather_world2(n_items):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_items, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_shard_bounds_cover_everything():
    for n in (0, 1, 7, 50, 64, 513):
        for world in (1, 2, 3, 8):
            spans = [sharding.shard_bounds(n, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in spans) - min(b - a for a, b in spans) <= 1
