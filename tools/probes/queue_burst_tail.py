"""Bursts of B 50-crop batches (submit_many, then wait for the last ticket): microseconds per burst.  Large tasks (64 / 128 rows once 8 / 32
batches are in flight) raise the sustained rate but lengthen the tail of a burst; CVGS_QUEUE_DEEP_ROWS=16 pins the old task size."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
from cvgpuspeedup_amd import cvgs
import queue_ab
dev = torch.device("cuda:0")
frames, outs, chains = queue_ab.build(dev, 20, 50)
torch.cuda.synchronize()
q = cvgs.Queue(depth=128, idle_us=2000.0)
res = {}
for B in (2, 4, 8, 16, 32, 64, 128, 512):
    ptrs = cvgs.Queue.chain_pointers([chains[i % 20] for i in range(B)])
    ts = []
    for r in range(40):
        time.sleep(0.0002)
        t0 = time.perf_counter()
        q.wait(q.submit_many(ptrs, B))
        ts.append((time.perf_counter() - t0) * 1e6)
    res[B] = round(float(np.median(ts[5:])), 1)
print(json.dumps({"deep_rows": os.environ.get("CVGS_QUEUE_DEEP_ROWS", "default"), "us_per_burst": res, "us_per_batch": {k: round(v / k, 2) for k, v in res.items()}}))
q.destroy()
