#!/usr/bin/env python3
"""Compare raw reference tensors (tests/golden/reference_outputs/<case>.f32, produced by reference_goldens.cu on an NVIDIA box
with the reference built) with this repository's seeded fixtures (tests/golden/seeded_fixtures.json: per-image xxhash64 of the
oracle's output) and with the oracle itself.  Prints, per case: bit-identical images, the largest ULP / absolute difference."""
import json
import os
import sys

import numpy as np
import xxhash

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REF_DIR = os.path.join(GOLD, "reference_outputs")
CASES = ("k1_cfg2a_fixed", "k1_cfg2b_variable", "k1_cfg2b_preserve_ar")
# round 6: the aspect-ratio extent probe (50 crop sizes on which rounding and truncating the fitted extent differ) and the NV12 read-back
EXTRA_CASES = ("k1_ar_extent_probe", "nv12_8k_to_1080p_bt709_full", "nv12_8k_to_1080p_bt601_full")
AR_PROBE_SEED, NV12_SEEDS = 0xC0FFF9, {"nv12_8k_to_1080p_bt709_full": 0xC0FFFB, "nv12_8k_to_1080p_bt601_full": 0xC0FFFD}


def ar_probe_crops():
    fx = json.load(open(os.path.join(GOLD, "ar_extent_differences.json")))
    return [(8 * i, 4 * i, r[0], r[1]) for i, r in enumerate(fx["probe_crops_64x128"])], fx["probe_crops_64x128"]


def oracle_ar_probe():
    from cvgpuspeedup_amd import cvgs
    from oracle import oracle_binding
    from tests import helpers as H
    crops, _ = ar_probe_crops()
    frame = H.random_u8((2160, 3840, 3), AR_PROBE_SEED)
    out = np.zeros((50, 3 * 64 * 128), np.float32)
    oracle_binding.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1),
                                                  cn=3, ar=cvgs.PRESERVE_AR, background=[128.0] * 3)))
    return out


def window_seen(img):
    """(width, height) of the window a [3,128,64] PRESERVE_AR image shows: the columns / rows that are not the background's chain value"""
    p = img.reshape(3, 128, 64)
    bg = p[:, 0, 0] if not np.array_equal(p[:, 0, 0], p[:, 64, 32]) else None  # a corner is background unless the window fills the target
    if bg is None:
        return 64, 128
    inside = (p != bg[:, None, None]).any(axis=0)
    return int(inside.any(axis=0).sum()), int(inside.any(axis=1).sum())


def compare_ar_probe():
    """-> which rule the reference's tensor follows, image by image: 'round' (the oracle's), 'trunc' (the reference test's OpenCV side)"""
    path = os.path.join(REF_DIR, "k1_ar_extent_probe.f32")
    if not os.path.exists(path):
        return None
    ref = np.fromfile(path, dtype="<f4").reshape(50, 3 * 64 * 128)
    _, rows = ar_probe_crops()
    tally = {"round": 0, "trunc": 0, "neither": 0}
    for i, (sw, sh, rw, rh, tw, th) in enumerate(rows):
        seen = window_seen(ref[i])
        tally["round" if seen == (rw, rh) else ("trunc" if seen == (tw, th) else "neither")] += 1
    ours = oracle_ar_probe()
    tally["max_abs_vs_oracle"] = float(np.abs(ref.astype(np.float64) - ours.astype(np.float64)).max())
    return tally


def oracle_nv12(name):
    """the chain of the reference's tests/resize/test_fused_resize.cu:141-147 on the seeded 7680 x 4320 surface -> 1920 x 1080 x 4 u8"""
    from cvgpuspeedup_amd import capi, cvgs
    from oracle import oracle_binding
    from tests import helpers as H
    W_, H_ = 7680, 4320
    surf = H.random_u8((H_ + H_ // 2, W_), NV12_SEEDS[name])
    out = np.zeros((1080, 1920, 4), np.uint8)
    luma = cvgs.GpuMat(H_, W_, cvgs.CV_8UC1, surf.ctypes.data, W_, owner=surf)
    prim = capi.BT709 if "bt709" in name else capi.BT601
    f4 = cvgs.CV_32FC4
    ops = [cvgs.read_nv12(luma, (1920, 1080), capi.YUV_FULL, prim, True), cvgs.convertTo(f4, cvgs.CV_8UC4),
           cvgs.cvtColor(cvgs.COLOR_RGBA2BGRA, cvgs.CV_8UC4), cvgs.write(cvgs.CV_8UC4, cvgs.GpuMat.from_array(out, cvgs.CV_8UC4))]
    oracle_binding.execute(cvgs.lower(ops))
    return out


def compare_nv12(name):
    path = os.path.join(REF_DIR, name + ".u8")
    if not os.path.exists(path):
        return None
    ref = np.fromfile(path, dtype=np.uint8).reshape(1080, 1920, 4)
    ours = oracle_nv12(name)
    d = np.abs(ref.astype(np.int32) - ours.astype(np.int32))
    return {"pixels": 1080 * 1920, "identical_pixels": int((d.max(axis=2) == 0).sum()), "max_grey_levels": int(d.max())}


def oracle_output(case):
    from cvgpuspeedup_amd import cvgs
    from oracle import oracle_binding
    from tests import helpers as H
    oracle_binding.load_oracle()
    fw, fh = case["frame"]
    frame = H.random_u8((fh, fw, 3), case["seed"])
    crops = H.fixed_crops(case["crops"]) if case["fixed"] else H.random_crops(case["crops"], fw, fh, seed=case["seed"] + 1)
    bg = [128.0] * 3 if case["ar"] != cvgs.IGNORE_AR else None
    out = np.zeros((case["crops"], 3 * 64 * 128), np.float32)
    oracle_binding.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1),
                                                  cn=3, ar=case["ar"], background=bg)))
    return out


def compare(name):
    """-> dict(images, bit_identical, max_ulp, max_abs) or None when the reference tensor is not there"""
    path = os.path.join(REF_DIR, name + ".f32")
    if not os.path.exists(path):
        return None
    case = next(c for c in json.load(open(os.path.join(GOLD, "seeded_fixtures.json")))["cases"] if c["name"] == name)
    ref = np.fromfile(path, dtype="<f4").reshape(case["crops"], 3 * 64 * 128)
    same = sum(xxhash.xxh64(ref[i].tobytes()).hexdigest() == h for i, h in enumerate(case["image_hashes"]))
    from tests import helpers as H
    ours = oracle_output(case)
    return {"images": case["crops"], "bit_identical": int(same), "max_ulp": int(H.ulp_diff(ref, ours).max()),
            "max_abs": float(np.abs(ref.astype(np.float64) - ours.astype(np.float64)).max())}


if __name__ == "__main__":
    found = False
    for n in CASES:
        r = compare(n)
        if r is None:
            print("%-24s no reference tensor at tests/golden/reference_outputs/%s.f32" % (n, n))
            continue
        found = True
        print("%-24s %d / %d images bit-identical; max difference %d ULP, %.3g absolute" % (n, r["bit_identical"], r["images"], r["max_ulp"], r["max_abs"]))
    r = compare_ar_probe()
    if r is None:
        print("%-24s no reference tensor" % "k1_ar_extent_probe")
    else:
        found = True
        print("k1_ar_extent_probe       windows that follow ROUND (the oracle) %d, TRUNCATE %d, neither %d of 50; max |difference| to the oracle %.3g" % (
            r["round"], r["trunc"], r["neither"], r["max_abs_vs_oracle"]))
    for n in NV12_SEEDS:
        r = compare_nv12(n)
        if r is None:
            print("%-24s no reference image" % n)
        else:
            found = True
            print("%-24s %d / %d pixels identical; max difference %d grey levels (SaturateCast of values a few ULP apart: <= 1 expected)" % (
                n, r["identical_pixels"], r["pixels"], r["max_grey_levels"]))
    sys.exit(0 if found else 2)
