"""Seeded NON-constant regression fixtures (tests/golden/seeded_fixtures.json + *_images.npy): the CPU oracle must
reproduce them bit for bit (CPU), and so must the HIP path (GPU) -- a checksum of checksums over every output image."""
import json
import os

import numpy as np
import pytest
import xxhash

from cvgpuspeedup_amd import cvgs
from tests import helpers as H

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_FIX = json.load(open(os.path.join(HERE, "seeded_fixtures.json")))
CASES, WARPS = _FIX["cases"], _FIX["warp_cases"]


def _inputs(case):
    fw, fh = case["frame"]
    cn = case["channels"]
    frame = H.random_u8((fh, fw, cn), case["seed"])
    crops = H.fixed_crops(case["crops"]) if case["fixed"] else H.random_crops(case["crops"], fw, fh, seed=case["seed"] + 1)
    bg = [128.0] * cn if case["ar"] != cvgs.IGNORE_AR else None
    return frame, crops, cn, bg


def _check(case, out):
    hashes = [xxhash.xxh64(out[i].tobytes()).hexdigest() for i in range(case["crops"])]
    bad = [i for i, (a, b) in enumerate(zip(hashes, case["image_hashes"])) if a != b]
    assert not bad, "%s: images %s differ from the fixture" % (case["name"], bad[:8])
    assert xxhash.xxh64("".join(hashes).encode()).hexdigest() == case["all"]
    stored = np.load(os.path.join(HERE, case["name"] + "_images.npy"))
    H.assert_bit_exact(out[case["stored_images"]], stored, case["name"] + " stored images")


@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_oracle_reproduces_seeded_fixture(case, oracle):
    frame, crops, cn, bg = _inputs(case)
    half = case.get("half", False)
    out = np.zeros((case["crops"], cn * 64 * 128), np.float16 if half else np.float32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.make_type(cvgs.CV_8U, cn)), crops,
                                         cvgs.GpuMat.from_array(out, cvgs.CV_16FC1 if half else cvgs.CV_32FC1), cn=cn,
                                         ar=case["ar"], background=bg, half=half)))
    _check(case, out)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_gpu_reproduces_seeded_fixture(case):
    import torch
    frame, crops, cn, bg = _inputs(case)
    dev = torch.device("cuda:0")
    ft = torch.from_numpy(frame).to(dev)
    half = case.get("half", False)
    ot = torch.zeros((case["crops"], cn * 64 * 128), dtype=torch.float16 if half else torch.float32, device=dev)
    cvgs.executeOperations(torch.cuda.current_stream(),
                           *H.k1_chain(cvgs.GpuMat.from_tensor(ft, cvgs.make_type(cvgs.CV_8U, cn)), crops,
                                       cvgs.GpuMat.from_tensor(ot, cvgs.CV_16FC1 if half else cvgs.CV_32FC1), cn=cn,
                                       ar=case["ar"], background=bg, half=half))
    torch.cuda.synchronize()
    _check(case, ot.cpu().numpy())


def _warp_chain(case, img, out_mat):
    kind = cvgs.WARP_PERSPECTIVE if case["perspective"] else cvgs.WARP_AFFINE
    n, dst = len(case["matrices"]), tuple(case["dst"])
    return [cvgs.warp(kind, cvgs.CV_8UC3, [img] * n, case["matrices"], dst), cvgs.split(cvgs.CV_32FC3, out_mat, dst)]


@pytest.mark.parametrize("case", WARPS, ids=[c["name"] for c in WARPS])
def test_oracle_reproduces_warp_fixture(case, oracle):
    src = H.random_u8((430, 470, 3), case["seed"])
    n, dst = len(case["matrices"]), case["dst"]
    out = np.zeros((n, 3 * dst[0] * dst[1]), np.float32)
    oracle.execute(cvgs.lower(_warp_chain(case, cvgs.GpuMat.from_array(src, cvgs.CV_8UC3), cvgs.GpuMat.from_array(out, cvgs.CV_32FC1))))
    assert [xxhash.xxh64(out[i].tobytes()).hexdigest() for i in range(n)] == case["image_hashes"]


@pytest.mark.gpu
@pytest.mark.parametrize("case", WARPS, ids=[c["name"] for c in WARPS])
def test_gpu_reproduces_warp_fixture(case):
    import torch
    dev = torch.device("cuda:0")
    src = torch.from_numpy(H.random_u8((430, 470, 3), case["seed"])).to(dev)
    n, dst = len(case["matrices"]), case["dst"]
    ot = torch.zeros((n, 3 * dst[0] * dst[1]), dtype=torch.float32, device=dev)
    cvgs.executeOperations(torch.cuda.current_stream(),
                           *_warp_chain(case, cvgs.GpuMat.from_tensor(src, cvgs.CV_8UC3), cvgs.GpuMat.from_tensor(ot, cvgs.CV_32FC1)))
    torch.cuda.synchronize()
    out = ot.cpu().numpy()
    assert [xxhash.xxh64(out[i].tobytes()).hexdigest() for i in range(n)] == case["image_hashes"]
