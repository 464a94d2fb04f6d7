#!/usr/bin/env python3
"""Regression fixtures for NON-constant inputs ("golden fixtures the build commits").

These are outputs of THIS repository's CPU oracle on seeded synthetic inputs (splitmix64, seed 0xC0FFEE family) --
NOT reference-derived vectors: the reference cannot be run here (see make_reference_kats.py) and its own tests never
use non-constant images.  They pin the oracle against drift (compiler, platform, later edits) and let the GPU tests
check large outputs through a checksum of checksums without re-running the oracle.

Stored per case: an xxhash64 of every output image (crop) and three full 64x128x3 images (first, middle, last).
Run:  python tests/golden/make_seeded_fixtures.py
"""
import json
import os
import sys

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cvgpuspeedup_amd import cvgs  # noqa: E402
from cvgpuspeedup_amd import workloads as W  # noqa: E402
from oracle import oracle_binding as ob  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def k1_case(name, frame_wh, n, seed, ar=cvgs.IGNORE_AR, fixed=False, cn=3, half=False):
    fw, fh = frame_wh
    frame = W.random_u8((fh, fw, cn), seed)
    crops = W.fixed_crops(n) if fixed else W.random_crops(n, fw, fh, seed=seed + 1)
    out = np.zeros((n, cn * 64 * 128), np.float16 if half else np.float32)
    bg = [128.0] * cn if ar != cvgs.IGNORE_AR else None
    ob.execute(cvgs.lower(W.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), crops,
                                     cvgs.GpuMat.from_array(out, cvgs.CV_16FC1 if half else cvgs.CV_32FC1), cn=cn, ar=ar,
                                     background=bg, half=half)))
    hashes = [xxhash.xxh64(out[i].tobytes()).hexdigest() for i in range(n)]
    picks = sorted({0, n // 2, n - 1})
    np.save(os.path.join(HERE, name + "_images.npy"), out[picks])
    return {"name": name, "frame": [fw, fh], "channels": cn, "crops": n, "seed": seed, "fixed": fixed, "ar": ar, "half": half,
            "image_hashes": hashes, "stored_images": picks,
            "all": xxhash.xxh64("".join(hashes).encode()).hexdigest()}


def warp_case(name, seed, perspective):
    """cvGS::warp on a seeded image: the reference test's five point sets (perspective) / five rotations (affine),
    one launch, fp32 NCHW output; one hash per plane."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import warp_cases as WC
    src = W.random_u8((430, 470, 3), seed)
    n, dst = 5, (300, 300)
    if perspective:
        ms = [WC.get_perspective_transform(*WC.REF_POINT_SETS[i]).tolist() for i in range(n)]
    else:
        ms = [[[np.cos(0.2 * i) * 0.8, -np.sin(0.2 * i) * 0.8, 20.5 * i], [np.sin(0.2 * i) * 0.8, np.cos(0.2 * i) * 0.8, 3.25 * i]]
              for i in range(n)]
    out = np.zeros((n, 3 * dst[0] * dst[1]), np.float32)
    img = cvgs.GpuMat.from_array(src, cvgs.CV_8UC3)
    rd = cvgs.warp(cvgs.WARP_PERSPECTIVE if perspective else cvgs.WARP_AFFINE, cvgs.CV_8UC3, [img] * n, ms, dst)
    ob.execute(cvgs.lower([rd, cvgs.split(cvgs.CV_32FC3, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), dst)]))
    hashes = [xxhash.xxh64(out[i].tobytes()).hexdigest() for i in range(n)]
    return {"name": name, "seed": seed, "perspective": perspective, "matrices": ms, "dst": list(dst), "image_hashes": hashes}


cases = [
    k1_case("k1_cfg2a_fixed", W.FRAME_4K, 50, W.SEED, fixed=True),
    k1_case("k1_cfg2b_variable", W.FRAME_4K, 50, W.SEED + 7),
    k1_case("k1_cfg2b_preserve_ar", W.FRAME_4K, 50, W.SEED + 9, ar=cvgs.PRESERVE_AR),
    k1_case("k1_cfg5_rank0", W.FRAME_6K, 64, W.SEED + 11),
    k1_case("k1_u8c4", W.FRAME_1080P, 32, W.SEED + 13, cn=4),
    k1_case("k1_cfg2b_half", W.FRAME_4K, 50, W.SEED + 7, half=True),
]
warps = [warp_case("warp_affine", W.SEED + 21, False), warp_case("warp_perspective", W.SEED + 22, True)]
with open(os.path.join(HERE, "seeded_fixtures.json"), "w") as f:
    json.dump({"comment": "oracle outputs on seeded inputs; see make_seeded_fixtures.py", "cases": cases, "warp_cases": warps},
              f, indent=1)
print("wrote", len(cases), "+", len(warps), "cases")
