"""Independent cross-checks of the oracle's bilinear arithmetic (the part no reference KAT pins):
(1) against torch's CPU grid_sample kernel evaluated at the SAME source coordinates (different code, same maths);
(2) against a float64 numpy evaluation of the published OpenCV-CUDA resize_linear formula.
Both within 1e-4 absolute -- the reference's own float tolerance (tests/testsCommon.cuh:36-61).  No GPU needed."""
import numpy as np
import pytest
import torch

from cvgpuspeedup_amd import cvgs
from tests import helpers as H


def oracle_resize(oracle, img, dst, ar=cvgs.IGNORE_AR, bg=None):
    cn = img.shape[2]
    out = np.zeros((1, cn * dst[0] * dst[1]), np.float32)
    src_t, f_t = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    rd = cvgs.resize(src_t, cvgs.INTER_LINEAR, [cvgs.GpuMat.from_array(img, src_t)], dst, 1, bg, ar)
    oracle.execute(cvgs.lower([rd, cvgs.split(f_t, cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), dst)]))
    return out.reshape(cn, dst[1], dst[0])


def formula_f64(img, dst):
    """out(x,y) = sum of 4 taps, src = dst * float32(1/(dst/src)), floor, +1 clamped, weights from unclamped x2/y2."""
    h, w, cn = img.shape
    fx = float(np.float32(1.0 / (dst[0] / w)))
    fy = float(np.float32(1.0 / (dst[1] / h)))
    xs = np.arange(dst[0], dtype=np.float32) * np.float32(fx)
    ys = np.arange(dst[1], dtype=np.float32) * np.float32(fy)
    x1 = np.floor(xs).astype(int); y1 = np.floor(ys).astype(int)
    x2r = np.minimum(x1 + 1, w - 1); y2r = np.minimum(y1 + 1, h - 1)
    wx2 = (xs.astype(np.float64) - x1); wx1 = 1.0 - wx2
    wy2 = (ys.astype(np.float64) - y1); wy1 = 1.0 - wy2
    f = img.astype(np.float64)
    out = (f[y1][:, x1] * (wy1[:, None] * wx1[None, :])[..., None] + f[y1][:, x2r] * (wy1[:, None] * wx2[None, :])[..., None] +
           f[y2r][:, x1] * (wy2[:, None] * wx1[None, :])[..., None] + f[y2r][:, x2r] * (wy2[:, None] * wx2[None, :])[..., None])
    return out.transpose(2, 0, 1), (xs, ys)


@pytest.mark.parametrize("shape,dst", [((120, 60), (64, 128)), ((97, 211), (64, 128)), ((300, 500), (64, 128)),
                                        ((16, 16), (100, 37)), ((1080, 1920), (64, 128))])
def test_bilinear_vs_formula_and_grid_sample(oracle, shape, dst):
    img = H.random_u8((shape[0], shape[1], 3), seed=shape[0] * 7 + dst[0])
    got = oracle_resize(oracle, img, dst)
    ref64, (xs, ys) = formula_f64(img, dst)
    assert np.abs(got - ref64).max() <= 1e-4 * 2.55  # u8 range 255: 1e-4 relative-to-unit scale
    # torch: sample at the same source coordinates; align_corners=True maps [-1,1] onto pixel centres 0..W-1,
    # padding_mode='border' reproduces the clamped +1 tap.
    h, w = shape
    gx = (torch.from_numpy(xs.astype(np.float64)) / max(w - 1, 1)) * 2 - 1
    gy = (torch.from_numpy(ys.astype(np.float64)) / max(h - 1, 1)) * 2 - 1
    grid = torch.stack(torch.meshgrid(gy, gx, indexing="ij")[::-1], dim=-1)[None]
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    samp = torch.nn.functional.grid_sample(t, grid, mode="bilinear", padding_mode="border", align_corners=True)[0].numpy()
    assert np.abs(got - samp).max() <= 1e-3


def test_constant_image_resizes_to_constant(oracle):
    """The property the reference's tests rely on (SURVEY.md 8c note): for 60x120 -> 64x128 the weights are exact."""
    img = np.empty((120, 60, 3), np.uint8)
    img[...] = (5, 37, 128)
    got = oracle_resize(oracle, img, (64, 128))
    assert (got == np.array([5, 37, 128], np.float32)[:, None, None]).all()


def test_preserve_ar_window(oracle):
    img = H.random_u8((120, 30, 3), seed=5)
    got = oracle_resize(oracle, img, (64, 128), cvgs.PRESERVE_AR, [128.0, 64.0, 32.0])
    assert (got[:, :, :16] == np.array([128, 64, 32], np.float32)[:, None, None]).all()
    assert (got[:, :, 48:] == np.array([128, 64, 32], np.float32)[:, None, None]).all()
    inner = oracle_resize(oracle, img, (32, 128))
    assert (got[:, :, 16:48] == inner).all()


def test_tap_census_of_the_bench_matches_the_oracle(oracle):
    """bench.py prices roofline.achieved on cvgpuspeedup_amd.workloads.tapped_bytes (numpy); the oracle holds an
    independent C census of the same SURVEY.md 8d quantity -- they must agree on every crop shape."""
    from cvgpuspeedup_amd import workloads as W
    rng = np.random.default_rng(3)
    shapes = [(60, 120), (1, 1), (2, 3), (64, 128), (128, 256), (129, 257), (512, 1024), (31, 1000), (3840, 2160)]
    shapes += [(int(w), int(h)) for w, h in zip(rng.integers(1, 600, 200), rng.integers(1, 1100, 200))]
    for w, h in shapes:
        assert W.tapped_bytes(w, h, 64, 128, 3) == oracle.tapped_bytes(w, h, 64, 128, cvgs.IGNORE_AR, 3), (w, h)
    assert W.tapped_bytes(60, 120, 64, 128, 3) == 21600  # upscaling taps every source pixel
    crops = W.fixed_crops(50)
    assert W.k1_algorithmic_bytes(crops, desc_bytes=0) == 5995200  # SURVEY.md 8d, cfg #2 fixed variant
    crops = W.random_crops(50, 3840, 2160, seed=W.SEED + 500000)
    assert W.k1_algorithmic_bytes(crops) == W.k1_algorithmic_bytes(crops, tapped_bytes_fn=oracle.tapped_bytes)


@pytest.mark.parametrize("swap", [True, False])
def test_k1_loop_nest_equals_the_interpreter(oracle, swap):
    """oracle_k1_fast (the plain loop nest bench.py times as the CPU baseline) == oracle_execute (the checker), bit for bit,
    on variable crops incl. 1-pixel ones; chains it does not cover are refused."""
    frame = H.random_u8((300, 500, 3), 8)
    crops = H.random_crops(17, 500, 300, seed=4, wmin=1, wmax=400, hmin=1, hmax=280)
    a = np.zeros((17, 3 * 64 * 128), np.float32)
    b = np.zeros_like(a)
    mk = lambda out, **kw: cvgs.lower(H.k1_chain(cvgs.GpuMat.from_array(frame, cvgs.CV_8UC3), crops,  # noqa: E731
                                                 cvgs.GpuMat.from_array(out, cvgs.CV_32FC1), swap=swap, **kw))
    oracle.execute(mk(a))
    oracle.execute_k1_fast(mk(b), 3)
    H.assert_bit_exact(b, a, "loop nest vs interpreter")
    with pytest.raises(RuntimeError):
        oracle.execute_k1_fast(mk(b, ar=cvgs.PRESERVE_AR, background=[1.0, 2.0, 3.0]))
