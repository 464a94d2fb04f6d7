"""Shared test helpers: the deterministic input generator of BASELINE.md (splitmix64, seed 0xC0FFEE),
the configs restated as concrete inputs, and exact comparison utilities."""
import numpy as np

from cvgpuspeedup_amd import cvgs

SEED = 0xC0FFEE
MASK = (1 << 64) - 1


def splitmix64_array(seed, n):
    """n successive splitmix64 outputs (vectorised; identical bytes on any box)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def random_u8(shape, seed=SEED):
    n = int(np.prod(shape))
    words = splitmix64_array(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n].reshape(shape).copy()


def random_u16(shape, seed=SEED):
    n = int(np.prod(shape))
    words = splitmix64_array(seed, (n + 3) // 4)
    return words.view(np.uint16)[:n].reshape(shape).copy()


def random_crops(n, frame_w, frame_h, seed=SEED + 1, wmin=32, wmax=512, hmin=64, hmax=1024):
    """BASELINE.md cfg #2(b): w~U[32,512], h~U[64,1024], position uniform inside the frame."""
    r = splitmix64_array(seed, 4 * n).astype(np.uint64)
    crops = []
    for i in range(n):
        w = int(wmin + r[4 * i] % np.uint64(wmax - wmin + 1))
        h = int(hmin + r[4 * i + 1] % np.uint64(hmax - hmin + 1))
        w, h = min(w, frame_w), min(h, frame_h)
        x = int(r[4 * i + 2] % np.uint64(frame_w - w + 1))
        y = int(r[4 * i + 3] % np.uint64(frame_h - h + 1))
        crops.append((x, y, w, h))
    return crops


def fixed_crops(n, w=60, h=120):
    """reference tests/batchresize/test_batchresize_x_split3D.cu:254-263: 60x120 at (i,i)."""
    return [(i, i, w, h) for i in range(n)]


# the per-channel tables of the reference's K1 tests (test_batchresize_x_split3D.cu:56-67,241-252)
K1_ALPHA = 0.3
K1_SUB = {1: [1.0], 2: [1.0, 4.0], 3: [1.0, 4.0, 3.2], 4: [1.0, 4.0, 3.2, 0.5]}
K1_DIV = {1: [3.2], 2: [3.2, 0.6], 3: [3.2, 0.6, 11.8], 4: [3.2, 0.6, 11.8, 33.0]}


def f32_list(vals):
    """cvScalar2CUDAV: double -> float narrowing, then back to python floats (exact)."""
    return [float(np.float32(v)) for v in vals]


def k1_chain(src_mat, crops, out_mat, dst=(64, 128), cn=3, used=None, ar=cvgs.IGNORE_AR, background=None,
             swap=True, src_depth=cvgs.CV_8U, table=None):
    """The K1 chain exactly as the reference test spells it (test_batchresize_x_split3D.cu:311-319)."""
    src_type = cvgs.make_type(src_depth, cn)
    f_type = cvgs.make_type(cvgs.CV_32F, cn)
    mats = [src_mat.roi(*c) for c in crops]
    rd = cvgs.resize(src_type, cvgs.INTER_LINEAR, mats, dst, len(crops) if used is None else used, background, ar)
    if table is not None:
        rd.table = table
    ops = [rd]
    if swap and cn == 3:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f_type))
    elif swap and cn == 4:
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGBA2BGRA, f_type))
    ops += [cvgs.multiply(f_type, [K1_ALPHA] * cn), cvgs.subtract(f_type, K1_SUB[cn]), cvgs.divide(f_type, K1_DIV[cn]),
            cvgs.split(f_type, out_mat, dst)]
    return ops


def ulp_diff(a, b):
    """Elementwise distance in units in the last place between two float32 arrays."""
    ai = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.ascontiguousarray(b, np.float32).view(np.int32).astype(np.int64)
    ai = np.where(ai < 0, np.int64(-(2 ** 31)) - ai, ai)
    bi = np.where(bi < 0, np.int64(-(2 ** 31)) - bi, bi)
    return np.abs(ai - bi)


def assert_bit_exact(gpu, ref, what=""):
    gpu = np.ascontiguousarray(gpu)
    ref = np.ascontiguousarray(ref)
    assert gpu.shape == ref.shape and gpu.dtype == ref.dtype, (gpu.shape, ref.shape, gpu.dtype, ref.dtype)
    same = gpu.view(np.uint8) == ref.view(np.uint8)
    if not same.all():
        bad = np.argwhere(gpu != ref)
        first = tuple(bad[0]) if len(bad) else None
        extra = ""
        if gpu.dtype == np.float32:
            extra = " max_ulp=%d" % int(ulp_diff(gpu, ref).max())
        raise AssertionError("%s: %d elements differ (first at %s: gpu=%r ref=%r)%s" % (
            what, int((~same).reshape(gpu.size, -1).any(axis=1).sum()), first,
            gpu[first] if first else None, ref[first] if first else None, extra))
