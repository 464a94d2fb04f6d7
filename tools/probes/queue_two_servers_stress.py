"""Two queues on one device (a pixel queue and an NV12 queue), each fed by its own host thread, short idle times: server switches
every few submits.  Every run failed before the round-3 protocol fixes; prints the stalled queue's statistics."""
import sys, threading, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import numpy as np, torch
from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
dev = torch.device("cuda:0")
frame_t = torch.from_numpy(H.random_u8((720, 1280, 3), seed=61)).to(dev)
crops = H.random_crops(12, 1280, 720, wmax=300, hmax=400, seed=63)
out_a = torch.zeros((12, 3 * 64 * 128), dtype=torch.float32, device=dev)
out_b = torch.zeros((12, 3 * 64 * 128), dtype=torch.float32, device=dev)
la = cvgs.lower(H.k1_chain(cvgs.GpuMat.from_tensor(frame_t, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_tensor(out_a, cvgs.CV_32FC1), (64, 128), 3))
w, h = 1280, 720
surf_t = torch.from_numpy(H.random_u8((h + h // 2, w), seed=62)).to(dev)
luma = cvgs.GpuMat(h, w, cvgs.CV_8UC1, surf_t.data_ptr(), w, owner=surf_t)
ecrops = [tuple(v & ~1 for v in c) for c in H.random_crops(6, w, h, seed=64, wmin=9, wmax=w // 2, hmin=9, hmax=h // 2)]
f = cvgs.CV_32FC3
lb = cvgs.lower([cvgs.read_nv12([luma.nv12_roi(*c) for c in ecrops], (64, 128), capi.YUV_FULL, capi.BT709, False), cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f),
                 cvgs.multiply(f, [1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225]),
                 cvgs.split(f, cvgs.GpuMat.from_tensor(out_b, cvgs.CV_32FC1), (64, 128))])
torch.cuda.synchronize()
fails = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    qa, qb = cvgs.Queue(idle_us=30.0), cvgs.Queue(idle_us=30.0)
    errors = []
    def feed(q, lowered, n, name):
        try:
            t = None
            for i in range(n):
                t = q.submit_lowered(lowered)
                if i % 16 == 15:
                    q.wait(t)
            q.wait(t)
        except Exception as ex:
            errors.append((name, repr(ex), q.stats()))
    th = [threading.Thread(target=feed, args=(qa, la, 600, "A")), threading.Thread(target=feed, args=(qb, lb, 600, "B"))]
    t0 = time.time()
    for t in th: t.start()
    for t in th: t.join()
    if errors:
        fails += 1
        print("rep", rep, "FAILED", errors, "A", qa.stats(), "B", qb.stats(), flush=True)
    qa.destroy(); qb.destroy()
print("fails", fails)
