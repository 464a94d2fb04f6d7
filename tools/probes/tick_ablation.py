#!/usr/bin/env python3
"""What bounds the headline tick?  The production kernel's own memory skeletons on the headline's exact launch (VERDICT r5 "next" #1).

The headline = ticks of M = 16 frames x 50 crops (cfg #2b), ONE cvgs_execute_many launch each, device plane tables, a rotation of 96 frames
(read-touched set >= 2 x the Infinity Cache), graph-replayed -- bench.py's protocol (bench.measure) and bench.py's Workload, so grid, tables,
frames, tensors and tap addresses ARE the headline's.  Only the library behind the call changes: tools/probes/build_ablate.sh compiles
k_k1_c3.hip again with -DCVGS_K1_ABLATE / -DCVGS_K1_STORE into build/ablate/libcvgs_<name>.so (never into the product library):

  full        the product kernel, rebuilt by the same recipe (control: must read what the installed library reads)
  ldst        tap loads + stores, no arithmetic (the dwords of the windows are stored as they are)   -> the kernel's MEMORY SKELETON
  ld          tap loads only (stores behind a test that never holds)
  st          stores only (windows synthesised from the lane id, product arithmetic)
  desc        every wave ends behind its batch of scalar loads (launch + dispatch + descriptor fetch)
  zfast       product arithmetic, the grid's linear workgroup index re-read chain-fastest (consecutive workgroups = different frames)
  zfast_ldst  the skeleton in that order
  plain / sc1 / sys   product kernel with plain / agent-scope (sc1) / system-scope write-through stores instead of non-temporal ones

Variants are measured round-robin (ABAB...), R rounds, the median over rounds is reported (A-then-B orderings drift by ~3 % on this pool).
usage (GPU box): python tools/probes/tick_ablation.py [--m 16] [--frames 96] [--rounds 4] [--variants full,ldst,...]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

BITWISE = ("full", "zfast", "plain", "sc1", "sys")  # variants whose output must equal the product's bit for bit


def load_variant(path):
    from cvgpuspeedup_amd import capi
    lib = C.CDLL(path)
    for name, restype, argtypes in capi.SYMBOLS:
        if name in ("cvgs_execute", "cvgs_execute_many", "cvgs_abi_version"):
            fn = getattr(lib, name)
            fn.restype = restype
            fn.argtypes = argtypes
    return lib


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--m", type=int, default=16, help="frames per launch (1 = one cvgs_execute per frame)")
    p.add_argument("--frames", type=int, default=96)
    p.add_argument("--rounds", type=int, default=4)
    p.add_argument("--variants", default="full,ldst,ld,st,desc,zfast,zfast_ldst,plain,sc1,sys")
    p.add_argument("--out", default=None)
    a = p.parse_args()
    import numpy as np
    import torch
    import bench as B
    from cvgpuspeedup_amd import capi
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    M = a.m
    nf = ((a.frames + M - 1) // M) * M
    wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M)
    installed = wl.lib
    names = [v for v in a.variants.split(",") if v]
    libs = {"installed": installed}
    for v in names:
        path = os.path.join(ROOT, "build", "ablate", "libcvgs_%s.so" % v)
        if not os.path.exists(path):
            print("missing %s: run tools/probes/build_ablate.sh in the container first" % path, file=sys.stderr)
            return 2
        libs[v] = load_variant(path)
    order = ["installed"] + names
    launches = max(16, 256 // M)

    def run_all(lib):
        wl.lib = lib
        for g in range(max(1, nf // M)):
            wl.launch(g if M > 1 else g, s)
        if M == 1:
            for g in range(nf):
                wl.launch(g, s)
        torch.cuda.synchronize()

    # reference tensors from the installed library
    run_all(installed)
    ref = [o.clone() for o in wl.outs]
    bit = {}
    for v in names:
        if v in BITWISE:
            for o in wl.outs:
                o.zero_()
            run_all(libs[v])
            bit[v] = all(bool(torch.equal(o.view(torch.int32), r.view(torch.int32))) for o, r in zip(wl.outs, ref))
    copy = B.copy_ceiling(dev)
    times = {v: [] for v in order}
    for r in range(a.rounds):
        for v in order:
            wl.lib = libs[v]
            m = B.measure(wl, launches, 4, target_s=0.12, min_replays=20, est_step_s=2.5e-6 * M, exact_steps=True)
            times[v].append(m["step_s"] * 1e6)
            print("round %d %-12s %8.3f us per launch" % (r, v, times[v][-1]), file=sys.stderr, flush=True)
    wl.lib = installed
    alg = wl.algorithmic_bytes()
    sect = wl.sector_bound_bytes()
    rows = {}
    for v in order:
        t = float(np.median(times[v]))
        rows[v] = {"us_per_launch": round(t, 3), "us_per_frame": round(t / M, 4), "min_us": round(min(times[v]), 3), "max_us": round(max(times[v]), 3),
                   "frac_alg_of_8TBs": round(alg / t / 1e6 / 8000.0, 4), "sector_TBs": round(sect / t / 1e6, 3)}
        if v in bit:
            rows[v]["bit_identical_to_installed"] = bit[v]
    out = {"m": M, "frames": nf, "rounds": a.rounds, "launches_per_replay": launches, "algorithmic_bytes_per_launch": alg, "sector_floor_bytes_per_launch": sect,
           "copy_ceiling_TBs": copy, "rows": rows}
    full = rows.get("full", rows["installed"])["us_per_launch"]
    if "ldst" in rows:
        out["full_over_skeleton"] = round(full / rows["ldst"]["us_per_launch"], 4)
    text = ["# tick ablation: M = %d frames x 50 crops per launch, %d-frame rotation, %d rounds round-robin, median (min-max) us per launch" % (M, nf, a.rounds),
            "# algorithmic bytes per launch %.0f, 64-B sector floor %.0f, copy ceiling of this run %s TB/s" % (alg, sect, json.dumps(copy)),
            "%-12s %10s %10s %18s %10s %12s %s" % ("variant", "us/launch", "us/frame", "min-max", "frac(alg)", "sector TB/s", "bits")]
    for v in order:
        r = rows[v]
        text.append("%-12s %10.3f %10.4f %8.3f-%-9.3f %10.4f %12.3f %s" % (v, r["us_per_launch"], r["us_per_frame"], r["min_us"], r["max_us"], r["frac_alg_of_8TBs"], r["sector_TBs"],
                                                                          r.get("bit_identical_to_installed", "")))
    if "full_over_skeleton" in out:
        text.append("# product kernel / its own memory skeleton (ldst) = %.4f" % out["full_over_skeleton"])
    print("\n".join(text))
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(text) + "\n" + json.dumps(out) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
