#!/usr/bin/env python3
"""Would XCD-banded work lists pay?  (round 6 experiment)

The headline's eight L2s fetch 175 MB per 16-frame tick where the union of the tapped 128-byte lines is 137 MB (profiles/r06_a_request_census.txt):
crops of one frame overlap, the workgroups that tap the same source rows land on different XCDs (XCD = linear workgroup id mod 8 = row tile
mod 8), and the second L2's fetch is served by the Infinity Cache -- at ~0.08 us per MB against ~0.17 for HBM (profiles/r06_b_tick_ablation_4k / _8k).
This probe gives every chain a WORK LIST in which slot s belongs to XCD s mod 8 and every XCD's items tap one band of source rows (band =
source row // ROWS mod 8), so that overlapping crops share an L2; the kernel variant (build/ablate/libcvgs_xcdwl.so: -DCVGS_K1_ABLATE=32) looks its
(crop, row tile) up there.  Results are bit-identical by construction (same items, another order).
usage (GPU box): python tools/probes/xcd_worklist_probe.py [--m 16] [--frames 96] [--band-rows 64]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def build_lists(crops_per_frame, band_rows, tiles=32, rows_per_tile=4, dst_h=128):
    import numpy as np
    per_frame, maxc = [], 0
    for crops in crops_per_frame:
        bins = [[] for _ in range(8)]
        for c, (x, y, w, h) in enumerate(crops):
            fy = h / float(dst_h)
            for rt in range(tiles):
                centre = y + (rt * rows_per_tile + rows_per_tile / 2.0) * fy
                bins[int(centre // band_rows) % 8].append((c << 8) | rt)
        per_frame.append(bins)
        maxc = max(maxc, max(len(b) for b in bins))
    slots = 8 * maxc
    out = np.full((len(crops_per_frame), slots), 0xffffffff, dtype=np.uint32)
    counts = []
    for f, bins in enumerate(per_frame):
        counts.append([len(b) for b in bins])
        for k, b in enumerate(bins):
            out[f, k:k + 8 * len(b):8] = np.array(b, dtype=np.uint32)
    return out, slots, np.array(counts)


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--m", type=int, default=16)
    p.add_argument("--frames", type=int, default=96)
    p.add_argument("--band-rows", default="64")
    p.add_argument("--rounds", type=int, default=3)
    a = p.parse_args()
    import numpy as np
    import torch
    import bench as B
    sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
    import tick_ablation as TA
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    M = a.m
    nf = ((a.frames + M - 1) // M) * M
    wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M)
    installed = wl.lib
    var = TA.load_variant(os.path.join(ROOT, "build", "ablate", "libcvgs_xcdwl.so"))
    var.cvgs_probe_set_worklist.argtypes = [C.c_void_p, C.c_uint32]
    var.cvgs_probe_set_worklist.restype = None
    # reference tensors
    for g in range(nf // M):
        wl.launch(g, s)
    torch.cuda.synchronize()
    ref = [o.clone() for o in wl.outs]
    launches = max(16, 256 // M)
    out = {"m": M, "frames": nf, "rows": {}}

    class Banded:
        def __init__(self, lists, slots):
            self.lists, self.slots, self.lib, self.per_launch, self.n = lists, slots, var, M, 50

        def launch(self, i, stream):
            g = i % (nf // M)
            var.cvgs_probe_set_worklist(self.lists[g * M].data_ptr(), self.slots)
            rc = var.cvgs_execute_many(wl.groups[g], M, stream)
            assert rc == 0, rc

    def timed(obj):
        m = B.measure(obj, launches, 4, target_s=0.12, min_replays=20, est_step_s=2.5e-6 * M, exact_steps=True)
        return m["step_s"] * 1e6

    variants = {"installed": wl}
    for br in [int(v) for v in a.band_rows.split(",")]:
        lists_np, slots, counts = build_lists(wl.crops, br)
        lists = torch.from_numpy(lists_np.view(np.int32)).to(dev)
        b = Banded(lists, slots)
        for o in wl.outs:
            o.zero_()
        for g in range(nf // M):
            b.launch(g, s)
        torch.cuda.synchronize()
        same = all(bool(torch.equal(o.view(torch.int32), r.view(torch.int32))) for o, r in zip(wl.outs, ref))
        tick_counts = counts.reshape(nf // M, M, 8).sum(axis=1)  # per tick, per XCD
        out["rows"]["banded_%d" % br] = {"slots_per_chain": int(slots), "padding": round(slots / 1600.0, 3), "bit_identical": same,
                                         "xcd_load_max_over_mean_per_tick": round(float((tick_counts.max(axis=1) / tick_counts.mean(axis=1)).mean()), 3)}
        variants["banded_%d" % br] = b
    times = {k: [] for k in variants}
    for _ in range(a.rounds):
        for k, obj in variants.items():
            times[k].append(timed(obj))
    for k, v in times.items():
        out["rows"].setdefault(k, {})["us_per_launch"] = round(float(np.median(v)), 3)
        out["rows"][k]["us_per_frame"] = round(float(np.median(v)) / M, 4)
    var.cvgs_probe_set_worklist(None, 0)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
