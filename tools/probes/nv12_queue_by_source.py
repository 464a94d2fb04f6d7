"""cfg #3 through the NV12 queue with smaller sources (same 1280 x 720 target): how much of the frame time is the source traffic."""
import sys, os, json
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench_more
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
sizes = ((6144, 3456), (2560, 1440), (1280, 720)) if len(sys.argv) < 2 else ((int(sys.argv[1]), int(sys.argv[2])),)
for wh in sizes:
    bench_more.W.FRAME_6K = wh
    r = bench_more.cfg3(dev, 30, queue=True)
    print(wh, r["us_per_launch"], flush=True)
    torch.cuda.empty_cache()
