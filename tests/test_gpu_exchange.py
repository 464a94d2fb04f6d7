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
