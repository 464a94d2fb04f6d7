"""How much can FMA contraction move the results?  The reference is built with nvcc's default -fmad=true, so its
kernels (and OpenCV-CUDA's) may fuse a*b+c where this engine and its oracle never do (strict IEEE,
-ffp-contract=off).  This test evaluates the K1 chain with the most aggressive contraction a compiler could choose
(every multiply-add of the bilinear sum fused, and x*alpha-sub fused), emulating fma(a,b,c) exactly through float64,
and checks the result stays far inside the reference's own tolerance (1e-4 absolute, tests/testsCommon.cuh:36-61):
the strict result is a valid reference result whatever contraction the reference's compiler picked."""
import numpy as np

from cvgpuspeedup_amd import cvgs
from tests import helpers as H

f32 = np.float32


def fma(a, b, c):
    """fused multiply-add of float32 arrays: the product of two fp32 numbers is exact in fp64; one rounding at the end
    (double rounding through fp64 can differ from a true fma only in astronomically rare ties)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def test_fma_contraction_stays_within_reference_tolerance(oracle):
    frame = H.random_u8((1080, 1920, 3), seed=123)
    crops = H.random_crops(24, 1920, 1080, seed=124)
    strict = np.zeros((24, 3 * 64 * 128), f32)
    oracle.execute(cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops, cvgs.GpuMat.from_array(strict, cvgs.CV_32FC1))))
    strict = strict.reshape(24, 3, 128, 64)
    alpha, sub, div = f32(H.K1_ALPHA), np.array(H.K1_SUB[3], f32), np.array(H.K1_DIV[3], f32)
    worst, worst_ulp, differing = 0.0, 0, 0
    for i, (cx, cy, w, h) in enumerate(crops):
        g = oracle.resize_geometry(w, h, 64, 128, cvgs.IGNORE_AR)
        sx = (np.arange(64, dtype=f32) * f32(g.fx)).astype(f32)
        sy = (np.arange(128, dtype=f32) * f32(g.fy)).astype(f32)
        x1 = np.floor(sx).astype(int); y1 = np.floor(sy).astype(int)
        x2r = np.minimum(x1 + 1, w - 1); y2r = np.minimum(y1 + 1, h - 1)
        wxa = ((x1 + 1).astype(f32) - sx).astype(f32); wxb = (sx - x1.astype(f32)).astype(f32)
        wya = ((y1 + 1).astype(f32) - sy).astype(f32); wyb = (sy - y1.astype(f32)).astype(f32)
        crop = frame[cy:cy + h, cx:cx + w].astype(f32)
        w00 = (wya[:, None] * wxa[None, :]).astype(f32); w10 = (wya[:, None] * wxb[None, :]).astype(f32)
        w01 = (wyb[:, None] * wxa[None, :]).astype(f32); w11 = (wyb[:, None] * wxb[None, :]).astype(f32)
        for c in range(3):
            src_c = 2 - c  # RGB2BGR
            p = crop[:, :, src_c]
            acc = (p[y1][:, x1] * w00).astype(f32)
            acc = fma(p[y1][:, x2r], w10, acc)
            acc = fma(p[y2r][:, x1], w01, acc)
            acc = fma(p[y2r][:, x2r], w11, acc)
            v = fma(acc, np.full_like(acc, alpha), np.full_like(acc, -sub[c]))  # x*alpha - sub fused
            v = (v / div[c]).astype(f32)
            d = np.abs(v.astype(np.float64) - strict[i, c].astype(np.float64))
            worst = max(worst, float(d.max()))
            worst_ulp = max(worst_ulp, int(H.ulp_diff(v, strict[i, c]).max()))
            differing += int((v != strict[i, c]).sum())
    print("max |strict - fully contracted| = %.3g (%d ULP max), %d of %d elements differ" % (worst, worst_ulp, differing, strict.size))
    assert differing > 0, "the contracted evaluation is expected to differ somewhere"
    assert worst <= 1e-4
