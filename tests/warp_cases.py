"""Warp test inputs shared by the CPU and GPU tests: the point sets of the reference's warp test
(tests/warping/test_warping_opencv.cu:47-48,140-153) and an independent float64 restatement of the warp."""
import numpy as np

# (src points, dst points) of the reference's perspective tests
REF_POINT_SETS = [
    ([(56, 65), (368, 52), (28, 387), (389, 390)], [(0, 0), (300, 0), (0, 300), (300, 300)]),
    ([(50, 50), (400, 50), (50, 400), (400, 400)], [(0, 0), (300, 0), (0, 300), (300, 300)]),
    ([(30, 30), (350, 30), (30, 350), (350, 350)], [(0, 0), (250, 0), (0, 250), (250, 250)]),
    ([(70, 70), (370, 70), (70, 370), (370, 370)], [(0, 0), (280, 0), (0, 280), (280, 280)]),
    ([(20, 20), (320, 20), (20, 320), (320, 320)], [(0, 0), (200, 0), (0, 200), (200, 200)]),
]


def get_perspective_transform(src, dst):
    """cv::getPerspectiveTransform: the 3x3 H with H*(x,y,1) ~ (u,v,1) for the four point pairs (h22 = 1)."""
    a = np.zeros((8, 8))
    b = np.zeros(8)
    for i, ((x, y), (u, v)) in enumerate(zip(src, dst)):
        a[i] = [x, y, 1, 0, 0, 0, -x * u, -y * u]
        a[i + 4] = [0, 0, 0, x, y, 1, -x * v, -y * v]
        b[i], b[i + 4] = u, v
    h = np.linalg.solve(a, b)
    return np.append(h, 1.0).reshape(3, 3)


def warp_f64(img, inv, dsize, perspective):
    """Independent restatement in float64 (no shared code with the oracle): zero outside, bilinear inside with the
    right/bottom taps clamped to the last column/row."""
    h, w = img.shape[:2]
    dw, dh = dsize
    inv = np.asarray(inv, np.float32).astype(np.float64).reshape(3, 3)
    ys, xs = np.mgrid[0:dh, 0:dw].astype(np.float64)
    sx = inv[0, 0] * xs + inv[0, 1] * ys + inv[0, 2]
    sy = inv[1, 0] * xs + inv[1, 1] * ys + inv[1, 2]
    if perspective:
        den = inv[2, 0] * xs + inv[2, 1] * ys + inv[2, 2]
        sx, sy = sx / den, sy / den
    inside = (sx >= 0) & (sx < w) & (sy >= 0) & (sy < h)
    sxc, syc = np.where(inside, sx, 0.0), np.where(inside, sy, 0.0)
    x1, y1 = np.floor(sxc).astype(np.int64), np.floor(syc).astype(np.int64)
    x2r, y2r = np.minimum(x1 + 1, w - 1), np.minimum(y1 + 1, h - 1)
    ax, ay = (sxc - x1)[..., None], (syc - y1)[..., None]
    im = img.astype(np.float64).reshape(h, w, -1)
    out = (im[y1, x1] * (1 - ax) * (1 - ay) + im[y1, x2r] * ax * (1 - ay) + im[y2r, x1] * (1 - ax) * ay + im[y2r, x2r] * ax * ay)
    return np.where(inside[..., None], out, 0.0), inside, sx, sy
