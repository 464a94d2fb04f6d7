"""Probe (GPU box): does the ORDER of a frame's crops inside the tick move the headline's time?  The tick's workgroups are dispatched chain by chain,
crop by crop; the drain of the launch is made of the last chain's last crops.  Same crops, same bytes: as drawn / largest source first / smallest first
(per frame) / by y (frame locality).  If ordering mattered the library could sort host-described ticks itself (the destination plane would travel in the
descriptor); it does not."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench as B  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    drawn = W.random_crops
    orders = {
        "as drawn": lambda c: c,
        "largest source first": lambda c: sorted(c, key=lambda r: -r[2] * r[3]),
        "smallest source first": lambda c: sorted(c, key=lambda r: r[2] * r[3]),
        "by top row": lambda c: sorted(c, key=lambda r: (r[1], r[0])),
    }
    wls = {}
    for name, f in orders.items():
        W.random_crops = lambda *a, _f=f, **k: _f(drawn(*a, **k))
        wls[name] = B.Workload(dev, 96, 50, 0, 1, True, per_launch=16)
    W.random_crops = drawn
    rows = {k: [] for k in orders}
    for r in range(4):
        for name, wl in wls.items():
            m = B.measure(wl, 16, 4, target_s=0.12, min_replays=20, est_step_s=2.5e-6 * 16, exact_steps=True)
            rows[name].append(m["step_s"] * 1e6)
    out = {k: {"us_per_tick_median": round(sorted(v)[len(v) // 2], 3), "min": round(min(v), 3), "max": round(max(v), 3)} for k, v in rows.items()}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
