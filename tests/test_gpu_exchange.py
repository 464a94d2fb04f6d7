"""The exchange step of the sharded batched-crop path (BASELINE cfg #5, SURVEY.md 8e option 2) as a REAL world-2 run on the
one GPU this box has: two processes on cuda:0, gloo for the IPC handle exchange, each rank's K1 launch writing its rows into
the peer's tensor through the IPC mapping, arrival flags on the device (no collective per step), fp32 and fp16 tensors, eager
and graph-replayed -- every rank's copy of every step's tensor bit-exact against the oracle.  (VERDICT r2 #3.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("half", [False, True])
def test_two_processes_exchange_through_ipc_mappings_and_device_flags(half):
    port = 29700 + (os.getpid() % 200) + (50 if half else 0)
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_PORT=str(port), EXCHANGE_HALF="1" if half else "0",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "exchange_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("exchange worker timed out")
        outs.append(out)
    for rank, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and "exchange worker rank %d OK" % rank in out, out[-3000:]


def test_a_lost_peer_costs_one_timeout_not_one_per_step():
    """the watchdog: a flag that never arrives is reported after `timeout_ms` (err[0] = 1, err[1] = the peer's index) instead of
    hanging the stream, and every later wait behind the same error words returns at once -- a 256-step replayed graph behind a
    dead peer ends after ONE timeout (bench_dist.py then keeps the all-gather leg)."""
    import ctypes as C
    import time

    import torch

    from cvgpuspeedup_amd import capi
    lib = capi.load_library()
    dev = torch.device("cuda:0")
    flags = torch.zeros(32, dtype=torch.int64, device=dev)       # "the peers' words in my block": nobody ever writes word 16
    peer = torch.zeros(32, dtype=torch.int64, device=dev)        # "my word in the peer's block"
    err = torch.zeros(2, dtype=torch.int64, device=dev)
    counter = torch.zeros(1, dtype=torch.int64, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    own = (C.c_void_p * 2)(flags.data_ptr(), flags.data_ptr() + 128)
    sig = (C.c_void_p * 2)(peer.data_ptr(), peer.data_ptr() + 128)
    flags[0] = 1000                                               # peer 0 is far ahead, peer 1 (word 16) is dead
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.check(lib.cvgs_exchange_wait(own, 2, 5, None, 0, 100.0, err.data_ptr(), s))
    torch.cuda.synchronize()
    first = time.perf_counter() - t0
    assert err.tolist() == [1, 1] and first >= 0.09, (err.tolist(), first)
    t0 = time.perf_counter()
    for _ in range(50):                                           # 50 more steps behind the dead peer: signal goes out, the wait fails fast
        capi.check(lib.cvgs_exchange_step(sig, own, 2, counter.data_ptr(), 0, 100.0, err.data_ptr(), s))
    capi.check(lib.cvgs_exchange_wait(own, 2, 7, None, 0, 100.0, err.data_ptr(), s))
    torch.cuda.synchronize()
    rest = time.perf_counter() - t0
    assert rest < 0.09, "later waits must not wait again: %.3f s" % rest
    assert int(counter.item()) == 50 and int(peer[0].item()) == 50 and int(peer[16].item()) == 50  # the signals still went out
    err.zero_()                                                   # cleared error words: waits wait again (and pass when the flag is there)
    flags[16] = 1000
    torch.cuda.synchronize()
    capi.check(lib.cvgs_exchange_wait(own, 2, 7, None, 0, 100.0, err.data_ptr(), s))
    torch.cuda.synchronize()
    assert err.tolist() == [0, 0]
