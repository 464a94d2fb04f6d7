#!/usr/bin/env python3
"""A/B of tools/bench_reference_tests.py rows between the installed library and variants from build/ablate/ (tools/probes/build_ablate.sh), round-robin.
usage (GPU box): python tools/probes/reference_chains_ab.py --variants pw_direct,pw_plain --rows read_x_write:8U:3,read_x_write:8U:4,read_x_split:8U:3"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--variants", default="")
    p.add_argument("--rows", default="read_x_write:8U:3,read_x_write:8U:4,read_x_write:8U:1,read_x_split:8U:3")
    p.add_argument("--rounds", type=int, default=3)
    a = p.parse_args()
    import torch
    import bench_reference_tests as BR
    from tick_ablation import load_variant
    from cvgpuspeedup_amd import capi
    torch.cuda.set_device(0)
    libs = {"installed": capi.load_library()}
    for v in [v for v in a.variants.split(",") if v]:
        libs[v] = load_variant(os.path.join(ROOT, "build", "ablate", "libcvgs_%s.so" % v))
    BR.VERBOSE = False
    res = {}
    for r in range(a.rounds):
        for name, lib in libs.items():
            BR.lib = lib
            del BR.ROWS[:]
            for row in a.rows.split(","):
                fn, depth, cn = row.split(":")
                getattr(BR, fn)(depth, int(cn))
            for row in BR.ROWS:
                res.setdefault(row["test"], {}).setdefault(name, []).append(row["us"])
    for test, by in res.items():
        print(test)
        for name, v in by.items():
            v = sorted(v)
            print("    %-14s %8.2f us  (%.2f-%.2f)" % (name, v[len(v) // 2], v[0], v[-1]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
