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
    sys.exit(0 if found else 2)
