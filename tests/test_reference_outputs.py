"""The geometry pin, closable by ONE external run (tools/reference_goldens/README.md): if someone has run
tools/reference_goldens/reference_goldens.cu on an NVIDIA box with the reference built and dropped the raw tensors into
tests/golden/reference_outputs/, the oracle must agree with them within the reference's own tolerance (1e-4 absolute,
tests/testsCommon.cuh:36-61 -- nvcc contracts multiply-adds, so a few ULP are expected; a coordinate-map disagreement is whole
grey levels).  Without those files the test is skipped: the pin stays open and says so."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools", "reference_goldens"))
import compare as RG  # noqa: E402


@pytest.mark.parametrize("name", RG.CASES)
def test_oracle_agrees_with_the_reference_on_seeded_non_constant_frames(name):
    r = RG.compare(name)
    if r is None:
        pytest.skip("no reference tensor for %s (tools/reference_goldens/README.md): bilinear geometry on non-constant images stays pinned by the spec only" % name)
    assert r["max_abs"] <= 1e-4, r


def test_aspect_ratio_extent_probe_if_a_reference_run_exists():
    """tools/reference_goldens: k1_ar_extent_probe -- 50 crop sizes on which rounding and truncating the fitted extent give different windows
    (tests/golden/ar_extent_differences.json).  With a reference tensor present every window must follow the oracle's rule (round); without
    one the choice stays spec-based and the test says so."""
    r = RG.compare_ar_probe()
    if r is None:
        pytest.skip("no reference tensor for k1_ar_extent_probe: the rounding of the aspect-ratio extent (oracle: round; reference test's OpenCV side: truncate) stays unpinned")
    assert r["round"] == 50 and r["max_abs_vs_oracle"] <= 1e-4, r


@pytest.mark.parametrize("name", sorted(RG.NV12_SEEDS))
def test_nv12_read_back_if_a_reference_run_exists(name):
    r = RG.compare_nv12(name)
    if r is None:
        pytest.skip("no reference image for %s: NV12 coefficients and chroma siting stay pinned by the derivation from Kr / Kb only" % name)
    assert r["max_grey_levels"] <= 1, r


def test_the_extra_cases_run_on_the_oracle():
    """the oracle side of the round-6 cases is runnable and self-consistent (what compare.py would hold the reference against)"""
    out = RG.oracle_ar_probe()
    _, rows = RG.ar_probe_crops()
    for i, (sw, sh, rw, rh, tw, th) in enumerate(rows):
        assert RG.window_seen(out[i]) == (rw, rh), (i, sw, sh)  # the oracle's tensor shows the ROUND window
    img = RG.oracle_nv12("nv12_8k_to_1080p_bt709_full")
    assert img.shape == (1080, 1920, 4) and (img[..., 3] == 255).all() and img[..., :3].std() > 10  # RGBA-reordered, opaque, non-constant
