"""Two PROCESSES, each with its own descriptor queue, on ONE GPU ("nothing tested across two processes on one GPU").
Every workgroup of a server must be resident -- each worker holds a task number -- so the servers of several processes have to FIT
the chip together: 4 workgroups of the pixel worker per CU at most (103 VGPRs, 4 waves per SIMD).  Inside one process the library
arbitrates (one server per device at a time); across processes nothing can, so the processes size their servers: the flags' bits
16..27 / CVGS_QUEUE_G give the worker workgroups.  Here: 383 each (1.5 per CU, 3 per CU together) -- both resident at once, 30 bursts of
40 batches each, every tensor bit-exact, no watchdog."""
import os
import subprocess
import sys
import tempfile

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_processes_two_queues_one_gpu():
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(2):
            env = dict(os.environ, QP_DIR=d, QP_RANK=str(r), QP_G="383", QP_ROUNDS="30")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "queue_process_worker.py")], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=300)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
            outs.append(o)
        for r, (p, o) in enumerate(zip(procs, outs)):
            assert p.returncode == 0, "rank %d:\n%s" % (r, o[-3000:])
            line = [ln for ln in o.splitlines() if ln.startswith("RESULT")][-1]
            assert "workgroups 383 mismatches 0 error 0" in line, line
