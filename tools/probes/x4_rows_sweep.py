"""Rows-per-wave sweep of the packed multi-pixel resize kernel (k1_packed_x4) at the reference's resize_write size (4K -> 3870 x 2260, every type
it sweeps) and at 1080p -> 4K: one process per setting (CVGS_K1_X4_ROWS is read once).  usage: python tools/probes/x4_rows_sweep.py [rows ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r"""
import sys
sys.path.insert(0, %r)
sys.path.insert(0, %r + "/tools")
import bench_reference_tests as B
import torch
B.VERBOSE = False
B.lib = B.capi.load_library()
torch.cuda.set_device(0)
for depth, cn in (("8U", 1), ("8U", 3), ("8U", 4), ("16U", 3), ("16S", 1), ("32F", 1)):
    B.resize_write(depth, cn, (3870, 2260))
for r in B.ROWS:
    print("ROW", r["test"], r["us"], r["frac_of_8TBs"])
""" % (ROOT, ROOT)

rows = sys.argv[1:] or ["default", "1", "2", "3", "4", "6", "8"]
table = {}
for r in rows:
    env = dict(os.environ)
    env.pop("CVGS_K1_X4_ROWS", None)
    if r != "default":
        env["CVGS_K1_X4_ROWS"] = r
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    for ln in p.stdout.splitlines():
        if ln.startswith("ROW "):
            name, us, frac = ln[4:].rsplit(" ", 2)
            table.setdefault(name, {})[r] = float(us)
    if p.returncode:
        print("rows", r, "failed:", p.stderr[-500:])
for name, t in table.items():
    print(name.ljust(48), "  ".join("%s: %6.2f" % (k, v) for k, v in t.items()))
