"""cvGS::warp on the CPU oracle.  Pinned by the reference: the affine translation case must equal
cv::cuda::warpAffine(INTER_LINEAR, constant 0 border) exactly (tests/warping/test_warping_opencv.cu:80-117), which for an
integer translation is the shifted image.  Everything else about fk::Warping is unpinned there (its perspective
comparisons print EXPECTED_FAIL); the oracle is cross-checked against an independent float64 restatement instead."""
import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from oracle import oracle_binding as ob
from tests import helpers as H
from tests import warp_cases as WC


def _run(src, stype, kind, transforms, dsize, used=None, default=None, tail=None, out_dtype=np.float32):
    single = not isinstance(src, list)
    mats = cvgs.GpuMat.from_array(src, stype) if single else [cvgs.GpuMat.from_array(s, stype) for s in src]
    n = 1 if single else len(src)
    cn = cvgs.type_cn(stype)
    f = cvgs.make_type(cvgs.CV_32F, cn)
    rd = cvgs.warp(kind, stype, mats, transforms, dsize, used, default)
    out = np.zeros((n, dsize[1], dsize[0], cn), out_dtype)
    ops = [rd] + (tail or [])
    otype = ops[-1].out_type if tail else f
    ops.append(cvgs.write(otype, cvgs.GpuMat.from_array(out.reshape(n, -1, cn), otype), dsize))
    ob.execute(cvgs.lower(ops))
    return out, rd


def test_reference_affine_translation_kat(oracle):
    """tx = 50, ty = 100 -> fk::Cast<float3, uchar3> -> write: the image shifted, zeros elsewhere."""
    src = H.random_u8((300, 400, 3), 11)
    out, _ = _run(src, cvgs.CV_8UC3, cvgs.WARP_AFFINE, [[1, 0, 50], [0, 1, 100]], (400, 300),
                  tail=[cvgs.cast(cvgs.CV_32FC3, cvgs.CV_8UC3)], out_dtype=np.uint8)
    exp = np.zeros_like(src)
    exp[100:, 50:] = src[:200, :350]
    assert np.array_equal(out[0], exp)


def test_identity_and_scale(oracle):
    src = H.random_u8((50, 70, 1), 3)
    out, _ = _run(src, cvgs.CV_8UC1, cvgs.WARP_PERSPECTIVE, np.eye(3), (70, 50))
    assert np.array_equal(out[0], src.astype(np.float32))
    # x2 upscale: even pixels are source pixels, odd ones the mean of two neighbours (last column clamps)
    out, _ = _run(src, cvgs.CV_8UC1, cvgs.WARP_AFFINE, [[2, 0, 0], [0, 2, 0]], (140, 100))
    s = src.astype(np.float32)[..., 0]
    assert np.array_equal(out[0][::2, ::2, 0], s)
    assert np.array_equal(out[0][::2, 1:-1:2, 0], (s[:, :-1] + s[:, 1:]) / 2)
    assert np.array_equal(out[0][::2, -1, 0], s[:, -1])


@pytest.mark.parametrize("idx", range(5))
def test_perspective_vs_float64_restatement(oracle, idx):
    """The reference test's five point sets on a random image: oracle (fp32 coordinates) vs float64 restatement."""
    src = H.random_u8((430, 470, 3), 20 + idx)
    fwd = WC.get_perspective_transform(*WC.REF_POINT_SETS[idx])
    out, rd = _run(src, cvgs.CV_8UC3, cvgs.WARP_PERSPECTIVE, fwd, (470, 430))
    inv = np.asarray(rd.warp[:9])
    ref, inside, sx, sy = WC.warp_f64(src, inv, (470, 430), True)
    # fp32 coordinate rounding moves taps by ~1e-4 px: away from the source border and from integer coordinates
    # (where floor() may flip) the two agree to a fraction of a grey level
    frac = np.minimum(np.abs(sx - np.round(sx)), np.abs(sy - np.round(sy)))
    safe = (sx > 0.01) & (sx < 469.99) & (sy > 0.01) & (sy < 429.99) & (frac > 0.01)
    assert safe.mean() > 0.05
    assert np.abs(out[0] - ref)[safe].max() < 0.25
    far_out = (sx < -0.01) | (sx > 470.01) | (sy < -0.01) | (sy > 430.01)
    assert (out[0][far_out] == 0).all() and far_out.any()
    # the inverse really is the inverse of the forward matrix
    assert np.allclose(np.asarray(inv, np.float64).reshape(3, 3) @ fwd / (np.asarray(inv).reshape(3, 3) @ fwd)[2, 2], np.eye(3), atol=1e-4)


def test_batch_with_unused_planes_gets_default_value(oracle):
    """warp(inputs, matrices, dstSize, usedPlanes, defaultValue): planes >= usedPlanes carry the default value through
    the chain (reference include/cvGPUSpeedup.cuh:411-434; tests/warping/test_warping_opencv.cu:191-260)."""
    src = [H.random_u8((60, 80, 3), 40 + i) for i in range(5)]
    ms = [WC.get_perspective_transform([(5, 5), (70, 8), (3, 50), (75, 55)], [(0, 0), (80, 0), (0, 60), (80, 60)])] * 5
    f = cvgs.CV_32FC3
    out, _ = _run(src, cvgs.CV_8UC3, cvgs.WARP_PERSPECTIVE, ms, (80, 60), used=2, default=[7.0, 8.0, 9.0],
                  tail=[cvgs.multiply(f, [2.0] * 3)])
    assert (out[2:] == np.array([14.0, 16.0, 18.0], np.float32)).all()
    assert out[0].std() > 10 and not np.array_equal(out[0], out[1])


def test_fk_cast_truncates(oracle):
    vals = np.array([-300.7, -1.5, -0.5, 0.0, 0.5, 0.999, 1.5, 2.5, 254.999, 255.5, 300.2, 70000.0, np.nan], np.float32)
    src = np.tile(vals, (2, 1))[:, :, None].copy()
    for depth, dt, lo, hi in ((cvgs.CV_8U, np.uint8, 0, 255), (cvgs.CV_16S, np.int16, -32768, 32767), (cvgs.CV_32S, np.int32, None, None)):
        out = np.zeros(src.shape, dt)
        t = cvgs.make_type(depth, 1)
        ob.execute(cvgs.lower([cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [cvgs.GpuMat.from_array(src, cvgs.CV_32FC1)], 1),
                               cvgs.cast(cvgs.CV_32FC1, t), cvgs.write(t, cvgs.GpuMat.from_array(out, t))]))
        exp = np.trunc(np.nan_to_num(vals.astype(np.float64), nan=0.0))
        if lo is not None:
            exp = np.clip(exp, lo, hi)
        assert np.array_equal(out[0, :, 0].astype(np.float64), exp), depth


def test_host_side_inversions_match_numpy():
    rng = np.random.default_rng(5)
    for _ in range(50):
        m = rng.standard_normal((2, 3)) * [1, 1, 50]
        inv = np.asarray(cvgs.invert_affine(m))
        full = np.vstack([m, [0, 0, 1]])
        assert np.allclose(np.vstack([inv, [0, 0, 1]]), np.linalg.inv(full), rtol=1e-9, atol=1e-9)
        p = rng.standard_normal((3, 3)) + 2 * np.eye(3)
        assert np.allclose(np.asarray(cvgs.invert_3x3(p)), np.linalg.inv(p), rtol=1e-8, atol=1e-10)
    assert cvgs.invert_3x3(np.zeros((3, 3))) == [[0.0] * 3] * 3
