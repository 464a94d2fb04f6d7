"""cfg #3 per frame for ticks of 1 / 2 / 4 / 8 surfaces per launch (one chain, batch = tick) and the queue: where the fixed cost per launch ends."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import bench_more as M
dev = torch.device("cuda:0")
torch.cuda.set_device(0)
torch.cuda.set_stream(torch.cuda.Stream())
for n in (1, 2, 4, 8):
    r = M.cfg3(dev, 100, per_launch=n)
    print("tick of", n, ":", r["us_per_launch"], "us per frame", r["frac_of_8TBs"], r["kernel"])
r = M.cfg3(dev, 100, queue=True)
print("queue      :", r["us_per_launch"], "us per frame", r["frac_of_8TBs"])
