"""One queue, thousands of retire / relaunch cycles: submits separated by pauses around the idle time (idle_us = 20)."""
import sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch
from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
dev = torch.device("cuda:0")
frame_t = torch.from_numpy(H.random_u8((720, 1280, 3), seed=61)).to(dev)
crops = H.random_crops(12, 1280, 720, wmax=300, hmax=400, seed=63)
out_a = torch.zeros((12, 3 * 64 * 128), dtype=torch.float32, device=dev)
la = cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out_a, cvgs.CV_32FC1), (64, 128), 3))
torch.cuda.synchronize()
rng = np.random.default_rng(1)
fails = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
    q = cvgs.Queue(idle_us=20.0)
    try:
        t = None
        for i in range(1500):
            t = q.submit_lowered(la)
            r = rng.uniform()
            if r < 0.5:
                time.sleep(float(rng.uniform(0, 80e-6)))
            if i % 64 == 63:
                q.wait(t)
        q.wait(t)
    except Exception as ex:
        fails += 1
        print("rep", rep, "FAILED", repr(ex), q.stats(), flush=True)
    st = q.stats()
    q.destroy()
print("fails", fails, "launches in the last rep", st["server_launches"])
