"""`python bench.py --gpus 2` end to end on ONE GPU (CVGS_BENCH_WORLD_ON_ONE_GPU=1: both ranks on cuda:0, gloo for the collectives):
the self-spawn, both compute legs, the all-gather leg, the IPC mappings with K1's mirror stores into the PEER'S tensor, the device-side
arrival flags, the link probe and the assembly of rank 0's compact line -- exactly the code `--gpus 8` runs on a node,
minus RCCL.  Only the line's structure and the bit-exactness verdicts are asserted: the timings of two ranks on one GPU mean nothing."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus_2_on_one_gpu():
    env = dict(os.environ, CVGS_BENCH_WORLD_ON_ONE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "8", "--warmup", "2", "--frames", "3", "--no-extra"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines  # ONE line on stdout, nothing behind it
    assert len(lines[0]) < 4096
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["value"] > 0 and j["config"]["workload"].startswith("cfg5: 2 x")
    assert j["roofline"]["per_gpu_frac"] > 0 and j["roofline"]["per_gpu_frac_on_the_queue"] > 0
    assert j["n1_same_workload"]["Mpix_per_s"] > 0 and j["n1_same_workload"]["on_the_queue_Mpix_per_s"] > 0
    legs = j["legs"]
    assert legs["compute_only_us"] > 0 and legs["allgather_us"] > 0
    assert legs["p2p_write_us"] is not None and legs["p2p_write_us"] > 0, j   # the P2P leg matched the all-gather bit for bit on both ranks
    assert j["xgmi_probe"].get("GB_per_s_per_link_one_direction_min", 0) > 0, j["xgmi_probe"]
    assert legs["link_floor_us"] is not None


def test_bench_force_dist_gathers_through_the_c_abi():
    """`python bench.py --force-dist` (ONE rank, backend nccl = RCCL): the all-gather leg goes through libcvgs_rccl.so -- cvgs_comm_init_rank
    with the unique id torch.distributed only carries, cvgs_allgather_inplace on a second stream behind an event of the step's K1 (round 6;
    VERDICT r5 next #4) -- and the line says so; the P2P leg must reproduce the gathered tensor bit for bit."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "CVGS_BENCH_WORLD_ON_ONE_GPU"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--steps", "8", "--warmup", "2", "--frames", "3", "--no-extra"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4096, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 1 and j["rccl_ranks_seen"] == 1 and j["legs"]["allgather_us"] > 0
    extra = json.load(open(os.path.join(ROOT, "bench_extra.json")))
    assert "libcvgs_rccl.so" in extra["extra"]["allgather"]["via"], extra["extra"]["allgather"]
    assert j["legs"]["p2p_write_us"] is not None
