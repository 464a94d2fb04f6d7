"""One of the two processes of tests/test_gpu_queue_two_processes.py: its own descriptor queue on cuda:0, `ROUNDS` bursts of batches,
every tensor compared with the oracle.  The two processes start their bursts together (a file barrier), so that two SERVER GRIDS --
one per process -- are resident on the chip at the same time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from cvgpuspeedup_amd import cvgs  # noqa: E402
from oracle import oracle_binding  # noqa: E402
from tests import helpers as H  # noqa: E402


def barrier(tag, me, n=2, timeout=120.0):
    d = os.environ["QP_DIR"]
    open(os.path.join(d, "%s.%d" % (tag, me)), "w").close()
    t0 = time.time()
    while sum(os.path.exists(os.path.join(d, "%s.%d" % (tag, r))) for r in range(n)) < n:
        if time.time() - t0 > timeout:
            raise SystemExit("barrier %s timed out" % tag)
        time.sleep(0.001)


def main():
    me = int(os.environ["QP_RANK"])
    g = int(os.environ.get("QP_G", "0"))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    oracle_binding.load_oracle()
    dst, cn, n = (64, 128), 3, 24
    frames = [H.random_u8((720, 1280, 3), seed=4000 + 10 * me + f) for f in range(4)]
    crops = [H.random_crops(n, 1280, 720, wmax=300, hmax=400, seed=4100 + 10 * me + f) for f in range(4)]
    refs, chains, outs, keep = [], [], [], []
    for f in range(4):
        ref = np.zeros((n, cn * dst[0] * dst[1]), np.float32)
        oracle_binding.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frames[f], cvgs.CV_8UC3), crops[f], cvgs.GpuMat.from_array(ref, cvgs.CV_32FC1), dst, cn)))
        refs.append(ref)
        ft = torch.from_numpy(frames[f]).to(dev)
        ot = torch.zeros((n, cn * dst[0] * dst[1]), dtype=torch.float32, device=dev)
        keep.append(ft)
        outs.append(ot)
        chains.append(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.CV_8UC3), crops[f], cvgs.GpuMat.from_tensor(ot, cvgs.CV_32FC1), dst, cn)))
    torch.cuda.synchronize()
    q = cvgs.Queue(depth=64, idle_us=20000.0, flags=(g & 0xfff) << 16)  # 20 ms: the two servers stay resident between the bursts
    bad = 0
    try:
        barrier("ready", me)
        for rnd in range(int(os.environ.get("QP_ROUNDS", "30"))):
            for o in outs:
                o.zero_()
            torch.cuda.current_stream().synchronize()
            last = None
            for k in range(40):
                last = q.submit_lowered(chains[k % 4])
            q.wait(last, timeout_s=20.0)
            for f in range(4):
                got = outs[f].cpu().numpy()
                bad += int((got.view(np.uint32) != refs[f].view(np.uint32)).sum())
        st = q.stats()
        print("RESULT rank %d workgroups %d mismatches %d error %d launches %d" % (me, st["workgroups"], bad, st["error"], st["server_launches"]), flush=True)
        barrier("done", me)
    finally:
        q.destroy()
    sys.exit(0 if bad == 0 and st["error"] == 0 else 1)


if __name__ == "__main__":
    main()
