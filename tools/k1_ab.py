#!/usr/bin/env python3
"""Within-process interleaved A/B of K1 kernel variants (cvgpuspeedup_amd/csrc/k_k1_exp.hip).

The experimental variants are NOT part of the product library: they live in cvgpuspeedup_amd/lib/libcvgs_exp.so (the
product objects + k_k1_exp.hip behind one extra entry point, cvgs_exp_execute), which only this tool loads.

  python tools/k1_ab.py --crops 50 --variants 0,1,2,8 --rounds 7 [--table]

variant 0 = the production dispatch; others = experimental ids.  Reports the median / min HIP-event time per launch
over the rounds, GB/s on algorithmic bytes, and checks that every FULL variant reproduces variant 0 bit for bit."""
import argparse
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import ctypes as C  # noqa: E402

import bench  # noqa: E402
from cvgpuspeedup_amd import capi  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402


def load_exp():
    path = os.path.join(ROOT, "cvgpuspeedup_amd", "lib", "libcvgs_exp.so")
    if not os.path.exists(path):  # built on demand (`make exp`): not part of the product build
        import subprocess
        subprocess.run(["make", "-C", os.path.join(ROOT, "cvgpuspeedup_amd", "csrc"), "-j8", "exp"], check=True)
    lib = C.CDLL(path)
    lib.cvgs_exp_execute.restype = C.c_int
    lib.cvgs_exp_execute.argtypes = [C.POINTER(capi.ChainDesc), C.c_int32, C.c_void_p]
    lib.cvgs_exp_name.restype = C.c_char_p
    lib.cvgs_exp_name.argtypes = [C.c_int32]
    lib.cvgs_last_error.restype = C.c_char_p
    return lib


class ExpWorkload(bench.Workload):
    """The headline workload launched through an experimental variant (variant 0 = the product dispatch)."""

    def __init__(self, exp, variant, *a, **kw):
        super().__init__(*a, **kw)
        self.exp, self.variant = exp, variant
        if variant:
            self.kernel = (exp.cvgs_exp_name(variant) or b"?").decode()

    def launch(self, i, stream):
        if not self.variant:
            return super().launch(i, stream)
        ch = self.chains[i % len(self.chains)]
        rc = self.exp.cvgs_exp_execute(C.byref(ch.desc), self.variant, stream)
        if rc:
            raise RuntimeError("cvgs_exp_execute: %d %s" % (rc, self.exp.cvgs_last_error().decode()))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--crops", type=int, default=50)
    p.add_argument("--variants", default="0,1,2,3")
    p.add_argument("--rounds", type=int, default=7)
    p.add_argument("--steps", type=int, default=1024)
    p.add_argument("--frames", type=int, default=0)
    p.add_argument("--table", action="store_true")
    p.add_argument("--eager", action="store_true")
    a = p.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    variants = [int(v) for v in a.variants.split(",")]
    per_frame = W.FRAME_4K[0] * W.FRAME_4K[1] * 3 + a.crops * 3 * 64 * 128 * 4
    nf = a.frames or max(4, min(24, (2 * bench.INFINITY_CACHE) // per_frame + 1))
    steps = max(16, a.steps * 50 // a.crops) if a.crops > 50 else a.steps
    torch.cuda.set_stream(torch.cuda.Stream())
    wls, base = {}, None
    exp = load_exp()
    for v in variants:
        wl = ExpWorkload(exp, v, dev, nf, a.crops, 0, 1, a.table, share=base)
        base = base or wl
        wls[v] = wl
    alg = base.algorithmic_bytes()
    # correctness: each variant's frame-0 output vs variant 0
    ref = None
    for v in variants:
        wls[v].outs[0].zero_()
        wls[v].launch(0, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        o = wls[v].outs[0].cpu().numpy()
        if ref is None:
            ref = o
        same = bool((o.view(np.uint32) == ref.view(np.uint32)).all())
        print("variant %2d %-18s bit-identical to variant %d: %s" % (v, wls[v].kernel, variants[0], same))
    times = {v: [] for v in variants}
    for r in range(a.rounds + 1):
        for v in variants:
            m = bench.measure(wls[v], min(steps, 256), 8, eager=a.eager, target_s=0.05, min_replays=10,
                              est_step_s=5e-6 * max(1.0, a.crops / 100.0))
            if r > 0:
                times[v].append(m["step_s"] * 1e6)
    print("crops/launch %d, %d frames, %d steps/round, %d rounds, alg bytes/launch %.0f, %s" % (
        a.crops, nf, steps, a.rounds, alg, "table" if a.table else "kernarg"))
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        print("variant %2d %-18s median %8.3f us  min %8.3f us   %7.1f GB/s  frac %.4f" % (
            v, wls[v].kernel, med, mn, alg / med / 1e3, alg / med / 1e3 / 8000.0))


if __name__ == "__main__":
    main()
