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

--workload cfg3 does the same for k4_nv12_x2 (BASELINE cfg #3, NV12 6K -> 1280x720, --m surfaces per launch), variants k4_full / k4_ldst / k4_ld / k4_st /
k4_math (no loads, no stores: the arithmetic alone) / k4_empty (the wave ends behind its first bounds test) from -DCVGS_K4_ABLATE builds of k_nv12_x2.hip.
--frame 8k draws the headline's 50 crops from 7680x4320 frames, where crops of the same sizes barely overlap (what the eight L2s fetch = what HBM delivers).

Variants are measured round-robin (ABAB...), R rounds, the median over rounds is reported (A-then-B orderings drift by ~3 % on this pool).
usage (GPU box): python tools/probes/tick_ablation.py [--m 16] [--frames 96] [--rounds 4] [--variants full,ldst,...]"""
import argparse
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

ABLATED = ("ldst", "ld", "st", "math", "empty", "desc")  # name tokens of variants that do NOT compute the product's values; every other variant must match it bit for bit


def bitwise(v):
    return not any(t in ABLATED for t in v.split("_"))


class Cfg3Workload:
    """BASELINE cfg #3 as tools/bench_more.py: cfg3() builds it: NV12 6144x3456 surfaces -> BGR float -> 1280x720 -> normalize -> split, one
    cvgs_execute per launch (`cams` surfaces per chain: 1 = one launch per frame), a rotation sized from the touched sectors."""

    def __init__(self, dev, cams=1):
        import ctypes as C
        import torch
        from cvgpuspeedup_amd import capi, cvgs
        from cvgpuspeedup_amd import workloads as W
        w, h = W.FRAME_6K
        dst = (1280, 720)
        nbuf = W.rotation_units(W.nv12_sector_read_bytes(w, h, dst[0], dst[1], 1), minimum=6)
        nbuf = (nbuf // cams) * cams
        self.bufs = [W.random_u8_torch((h + h // 2, w), 500 + i, dev) for i in range(nbuf)]
        self.outs = [torch.zeros((cams, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev) for _ in range(nbuf // cams)]
        f = cvgs.CV_32FC3
        self.chains = []
        for k, o in enumerate(self.outs):
            lumas = [cvgs.GpuMat(h, w, cvgs.CV_8UC1, b.data_ptr(), w, owner=b) for b in self.bufs[k * cams:(k + 1) * cams]]
            rd = cvgs.read_nv12(lumas[0] if cams == 1 else lumas, dst, capi.YUV_FULL, capi.BT709, False)
            ops = [rd, cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f), cvgs.multiply(f, [W.K1_ALPHA] * 3), cvgs.subtract(f, W.K1_SUB[3]), cvgs.divide(f, W.K1_DIV[3]),
                   cvgs.split(f, cvgs.GpuMat.from_tensor(o, cvgs.CV_32FC1), dst)]
            self.chains.append(cvgs.lower(ops))
        self.kernel = cvgs.kernel_name(*ops)
        self.lib = capi.load_library()
        self.per_launch = 1
        self.n = cams
        self.groups = []
        self._C = C
        self._check = capi.check
        write = dst[0] * dst[1] * 3 * 4
        self._alg = cams * (write + dst[0] * dst[1] * 4 + dst[0] * dst[1] * 2 * 4)
        self._sector = cams * (W.nv12_sector_read_bytes(w, h, dst[0], dst[1], 1) + write)

    def launch(self, i, stream):
        rc = self.lib.cvgs_execute(self._C.byref(self.chains[i % len(self.chains)].desc), stream)
        if rc:
            self._check(rc)

    def algorithmic_bytes(self):
        return float(self._alg)

    def sector_bound_bytes(self):
        return float(self._sector)


class ResizeWriteWorkload:
    """The reference's tests/resize/test_resize_write.cu chain at its own size (tools/bench_reference_tests.py: resize_write): a 4K image of
    `cn` u8 channels -> resize 3870 x 2260 -> convertTo back to u8 -> write (packed), one cvgs_execute per launch, 12 image pairs in rotation."""

    def __init__(self, dev, cn=3, dst=(3870, 2260), pairs=12):
        import ctypes as C
        import torch
        from cvgpuspeedup_amd import capi, cvgs
        from cvgpuspeedup_amd import workloads as W
        fw, fh = W.FRAME_4K
        st, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
        self.chains, self.outs, self.keep = [], [], []
        for i in range(pairs):
            src = W.random_u8_torch((fh, fw, cn), 7000 + i, dev)
            out = torch.zeros((dst[1], dst[0], cn), dtype=torch.uint8, device=dev)
            ops = [cvgs.resize(st, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(src, st), dst), cvgs.convertTo(f, st), cvgs.write(st, cvgs.GpuMat.from_tensor(out, st))]
            self.chains.append(cvgs.lower(ops))
            self.outs.append(out)
            self.keep.append(src)
        self.kernel = cvgs.kernel_name(*ops)
        self.lib = capi.load_library()
        self.per_launch, self.n, self.groups = 1, 1, []
        self._C, self._check = C, capi.check
        self._alg = (fw * fh + dst[0] * dst[1]) * cn

    def launch(self, i, stream):
        rc = self.lib.cvgs_execute(self._C.byref(self.chains[i % len(self.chains)].desc), stream)
        if rc:
            self._check(rc)

    def algorithmic_bytes(self):
        return float(self._alg)

    def sector_bound_bytes(self):
        return float(self._alg)


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
    p.add_argument("--variants", default=None)
    p.add_argument("--workload", default="k1", choices=["k1", "cfg3", "resize_write"],
                   help="k1: the headline ticks (cfg #2b); cfg3: NV12 6K -> 1280x720, --m = cameras per launch; resize_write: 4K u8 (--m channels) -> 3870x2260 -> u8 (k1_packed_x4)")
    p.add_argument("--frame", default="4k", choices=["4k", "8k"], help="k1 only: frame size the 50 crops are drawn from (8k: the same crop sizes barely overlap)")
    p.add_argument("--fixed", action="store_true", help="k1 only: cfg #2a's crops (60 x 120 at (i, i), the reference's own test layout) instead of cfg #2b's")
    p.add_argument("--out", default=None)
    a = p.parse_args()
    import numpy as np
    import torch
    import bench as B
    from cvgpuspeedup_amd import capi
    from cvgpuspeedup_amd import workloads as W
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    torch.cuda.set_stream(side)
    s = side.cuda_stream
    M = a.m
    if a.variants is None:
        a.variants = "full,ldst,ld,st,desc,zfast,zfast_ldst,plain,sc1,sys" if a.workload == "k1" else "k4_full,k4_ldst,k4_ld,k4_st,k4_math,k4_empty"
    if a.workload == "resize_write":
        wl = ResizeWriteWorkload(dev, cn=M if M in (1, 2, 3, 4) else 3)
        nf, M = len(wl.chains), 1
        if a.variants is None or a.variants.startswith("k4_"):
            a.variants = "x4_full,x4_ldst,x4_ld,x4_st,x4_math"
    elif a.workload == "cfg3":
        wl = Cfg3Workload(dev, cams=M)
        nf, M = len(wl.chains), 1  # one cvgs_execute per launch
    else:
        nf = ((a.frames + M - 1) // M) * M
        wl = B.Workload(dev, nf, 50, 0, 1, True, per_launch=M, frame_wh=(7680, 4320) if a.frame == "8k" else W.FRAME_4K, fixed=a.fixed)
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
    if a.workload in ("cfg3", "resize_write"):
        launches = 2 * nf

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
        if bitwise(v):
            for o in wl.outs:
                o.zero_()
            run_all(libs[v])
            bit[v] = all(bool(torch.equal(o.reshape(-1).view(torch.uint8), r.reshape(-1).view(torch.uint8))) for o, r in zip(wl.outs, ref))
    copy = B.copy_ceiling(dev)
    times = {v: [] for v in order}
    for r in range(a.rounds):
        for v in order:
            wl.lib = libs[v]
            m = B.measure(wl, launches, 4, target_s=0.12, min_replays=20, est_step_s=(8e-6 * a.m if a.workload == "cfg3" else (2e-5 if a.workload == "resize_write" else 2.5e-6 * M)), exact_steps=True)
            times[v].append(m["step_s"] * 1e6)
            print("round %d %-12s %8.3f us per launch" % (r, v, times[v][-1]), file=sys.stderr, flush=True)
    wl.lib = installed
    alg = wl.algorithmic_bytes()
    sect = wl.sector_bound_bytes()
    rows = {}
    for v in order:
        t = float(np.median(times[v]))
        rows[v] = {"us_per_launch": round(t, 3), "us_per_frame": round(t / M, 4), "min_us": round(min(times[v]), 3), "max_us": round(max(times[v]), 3),
                   "frac_alg_of_8TBs": round(alg / t / 1e6 / 8.0, 4), "sector_TBs": round(sect / t / 1e6, 3)}
        if v in bit:
            rows[v]["bit_identical_to_installed"] = bit[v]
    out = {"m": M, "frames": nf, "rounds": a.rounds, "launches_per_replay": launches, "algorithmic_bytes_per_launch": alg, "sector_floor_bytes_per_launch": sect,
           "copy_ceiling_TBs": copy, "rows": rows}
    full = rows.get("full", rows.get("k4_full", rows.get("x4_full", rows["installed"])))["us_per_launch"]
    skel = next((k for k in ("ldst", "k4_ldst", "k4_rows2_ldst", "x4_ldst") if k in rows), None)
    if skel:
        out["full_over_skeleton"] = round(full / rows[skel]["us_per_launch"], 4)
    what = ("resize_write 4K u8c%d -> 3870x2260 u8, %d image pairs in rotation" % (a.m, nf) if a.workload == "resize_write" else
            "cfg #3 (NV12 6K -> 1280x720 normalized NCHW), %d surface(s) per launch, %d launches in rotation" % (a.m, nf) if a.workload == "cfg3" else
            "M = %d frames (%s) x 50 %s crops per launch, %d-frame rotation" % (M, a.frame, "fixed 60x120 (cfg #2a)" if a.fixed else "variable (cfg #2b)", nf))
    text = ["# ablation of %s [%s]: %d rounds round-robin, median (min-max) us per launch" % (wl.kernel, what, a.rounds),
            "# algorithmic bytes per launch %.0f, 64-B sector floor %.0f, copy ceiling of this run %s TB/s" % (alg, sect, json.dumps(copy)),
            "%-12s %10s %10s %18s %10s %12s %s" % ("variant", "us/launch", "us/frame", "min-max", "frac(alg)", "sector TB/s", "bits")]
    for v in order:
        r = rows[v]
        text.append("%-12s %10.3f %10.4f %8.3f-%-9.3f %10.4f %12.3f %s" % (v, r["us_per_launch"], r["us_per_frame"], r["min_us"], r["max_us"], r["frac_alg_of_8TBs"], r["sector_TBs"],
                                                                          r.get("bit_identical_to_installed", "")))
    if "full_over_skeleton" in out:
        text.append("# product kernel / its own memory skeleton (%s) = %.4f" % (skel, out["full_over_skeleton"]))
    print("\n".join(text))
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            f.write("\n".join(text) + "\n" + json.dumps(out) + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
