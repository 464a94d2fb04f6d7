"""Up-scaling sweep for K1's packed-u8 whole-frame kernels (k_k1_x4.hip against k1_resize_split): src -> dst packed u8c3 / u8c4,
HIP-event time per launch.  The environment hook CVGS_K1_X4 (0 = never, 1 = whenever eligible) picks the kernel; other rows-per-wave shapes: tools/probes/build_ablate.sh (-DCVGS_X4_ROWS)."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cvgpuspeedup_amd import capi, cvgs  # noqa: E402
from tools.bench_resize import events_time  # noqa: E402

CASES = [((1920, 1080), (3840, 2160)), ((960, 540), (1920, 1080)), ((1280, 720), (2560, 1440)), ((1280, 720), (3840, 2160)),
         ((1920, 1080), (2560, 1440)), ((3840, 2160), (1920, 1080)), ((640, 360), (1280, 720)), ((2560, 1440), (3840, 2160)),
         ((3840, 2160), (3870, 2260))]  # the last one: the reference's own resize_write size (tests/resize/test_resize_write.cu:55-56)


def case(dev, cn, src, dst, iters, nbuf=0):
    lib = capi.load_library()
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    n_buf = min(24, max(2, (512 << 20) // ((src[0] * src[1] + dst[0] * dst[1]) * cn) + 1))  # cycle through > 512 MB: no cache-resident re-reads
    n_buf = nbuf or n_buf
    frames = [torch.randint(0, 256, (src[1], src[0], cn), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
    outs = [torch.empty((dst[1], dst[0], cn), dtype=torch.uint8, device=dev) for _ in range(n_buf)]
    chains = []
    for fr, o in zip(frames, outs):
        ops = [cvgs.resize(u, cvgs.INTER_LINEAR, cvgs.GpuMat.from_tensor(fr, u), dst), cvgs.convertTo(f, u),
               cvgs.write(u, cvgs.GpuMat.from_tensor(o, u))]
        chains.append((cvgs.lower(ops), ops))
    s = torch.cuda.current_stream().cuda_stream
    st = {"i": 0}

    def launch():
        ch = chains[st["i"] % len(chains)][0]
        st["i"] += 1
        capi.check(lib.cvgs_execute(C.byref(ch.desc), s))

    t = events_time(launch, iters)
    alg = dst[0] * dst[1] * cn + min(src[0] * src[1], 4 * dst[0] * dst[1]) * cn
    return {"case": "%dx%d -> %dx%d u8c%d" % (src[0], src[1], dst[0], dst[1], cn), "kernel": cvgs.kernel_name(*chains[0][1]),
            "us": round(t * 1e6, 2), "GB_per_s": round(alg / t / 1e9, 1), "frac_of_8TBs": round(alg / t / 8e12, 3)}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cn", type=int, nargs="*", default=[3])
    ap.add_argument("--nbuf", type=int, default=0, help="buffers cycled through (default: enough for > 512 MB)")
    ap.add_argument("--only", type=int, default=-1, help="index into CASES")
    a = ap.parse_args()
    for cn in a.cn:
        for src, dst in (CASES if a.only < 0 else CASES[a.only:a.only + 1]):
            print(json.dumps(case(torch.device("cuda:0"), cn, src, dst, a.iters, a.nbuf)))
