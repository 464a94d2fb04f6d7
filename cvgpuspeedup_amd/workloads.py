"""Synthetic workloads of BASELINE.md section 2, restated as concrete inputs.

Deterministic generator: splitmix64, seed 0xC0FFEE (identical bytes on any box, numpy on the host or
torch on the device).  No pixel arithmetic of the hot path happens here."""
import numpy as np

from . import cvgs

SEED = 0xC0FFEE

FRAME_1080P = (1920, 1080)
FRAME_4K = (3840, 2160)
FRAME_6K = (6144, 3456)
DST = (64, 128)  # crop target, width x height (reference tests/batchresize/test_batchresize_x_split3D.cu:265)
LLC_BYTES = 256 << 20  # MI355X Infinity Cache (MI355X_MICROARCH.md)

# the per-channel tables of the reference's K1 tests (test_batchresize_x_split3D.cu:56-67,241-252)
K1_ALPHA = 0.3
K1_SUB = {1: [1.0], 2: [1.0, 4.0], 3: [1.0, 4.0, 3.2], 4: [1.0, 4.0, 3.2, 0.5]}
K1_DIV = {1: [3.2], 2: [3.2, 0.6], 3: [3.2, 0.6, 11.8], 4: [3.2, 0.6, 11.8, 33.0]}


def splitmix64_array(seed, n):
    """n successive splitmix64 outputs as uint64 (vectorised numpy)."""
    with np.errstate(over="ignore"):
        idx = np.arange(1, n + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def _to_i64(v):
    v &= 0xFFFFFFFFFFFFFFFF
    return v - (1 << 64) if v >= (1 << 63) else v


def splitmix64_torch(seed, n, device):
    """The same stream generated on `device` with wrapping int64 arithmetic (bit-identical to numpy)."""
    import torch
    idx = torch.arange(1, n + 1, dtype=torch.int64, device=device)
    z = idx * _to_i64(0x9E3779B97F4A7C15) + _to_i64(seed)

    def lsr(v, k):  # logical shift right on int64
        return (v >> k) & ((1 << (64 - k)) - 1)

    z = (z ^ lsr(z, 30)) * _to_i64(0xBF58476D1CE4E5B9)
    z = (z ^ lsr(z, 27)) * _to_i64(0x94D049BB133111EB)
    z = z ^ lsr(z, 31)
    return z


def random_u8(shape, seed=SEED):
    n = int(np.prod(shape))
    words = splitmix64_array(seed, (n + 7) // 8)
    return words.view(np.uint8)[:n].reshape(shape).copy()


def random_u8_torch(shape, seed, device):
    n = 1
    for s in shape:
        n *= int(s)
    words = splitmix64_torch(seed, (n + 7) // 8, device)
    return words.view(__import__("torch").uint8)[:n].reshape(shape).contiguous()


def random_u16(shape, seed=SEED):
    n = int(np.prod(shape))
    words = splitmix64_array(seed, (n + 3) // 4)
    return words.view(np.uint16)[:n].reshape(shape).copy()


def random_crops(n, frame_w, frame_h, seed=SEED + 1, wmin=32, wmax=512, hmin=64, hmax=1024):
    """BASELINE.md cfg #2(b): w~U[32,512], h~U[64,1024], position uniform inside the frame -> (x,y,w,h)."""
    r = splitmix64_array(seed, 4 * n)
    crops = []
    for i in range(n):
        w = int(wmin + int(r[4 * i]) % (wmax - wmin + 1))
        h = int(hmin + int(r[4 * i + 1]) % (hmax - hmin + 1))
        w, h = min(w, frame_w), min(h, frame_h)
        x = int(r[4 * i + 2]) % (frame_w - w + 1)
        y = int(r[4 * i + 3]) % (frame_h - h + 1)
        crops.append((x, y, w, h))
    return crops


def fixed_crops(n, w=60, h=120):
    """reference tests/batchresize/test_batchresize_x_split3D.cu:254-263: 60x120 at (i,i)."""
    return [(i, i, w, h) for i in range(n)]


def k1_chain(src_mat, crops, out_mat, dst=DST, cn=3, used=None, ar=cvgs.IGNORE_AR, background=None, swap=True,
             src_depth=cvgs.CV_8U, table=None, half=False):
    """The K1 chain exactly as the reference test spells it (test_batchresize_x_split3D.cu:311-319):
    resize -> cvtColor(RGB2BGR) -> multiply(0.3) -> subtract -> divide -> split(tensor)."""
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
    ops += [cvgs.multiply(f_type, [K1_ALPHA] * cn), cvgs.subtract(f_type, K1_SUB[cn]), cvgs.divide(f_type, K1_DIV[cn])]
    if half:  # the half-precision hand-off option: one extra convertTo, fp16 NCHW tensor
        h_type = cvgs.make_type(cvgs.CV_16F, cn)
        ops += [cvgs.convertTo(f_type, h_type), cvgs.split(h_type, out_mat, dst)]
    else:
        ops.append(cvgs.split(f_type, out_mat, dst))
    return ops


def _distinct_taps(src, n_out):
    """Distinct source indices a stretch resize (IGNORE_AR) taps along one axis: for every output index d both
    floor(d*f) and min(floor(d*f)+1, src-1), f = float32(1 / (n_out / src)) -- the kernel's own coordinate rule."""
    f = np.float32(1.0 / (float(n_out) / float(src)))
    a = np.floor(np.arange(n_out, dtype=np.float32) * f).astype(np.int64)
    b = np.minimum(a + 1, src - 1)
    return int(np.unique(np.concatenate([a, b])).size)


def tapped_bytes(src_w, src_h, dst_w, dst_h, bytes_per_pixel):
    """SURVEY.md 8d tap census of one K1 crop: taps are separable, so distinct pixels = ux * uy."""
    return _distinct_taps(src_w, dst_w) * _distinct_taps(src_h, dst_h) * bytes_per_pixel


def k1_algorithmic_bytes(crops, dst=DST, cn=3, src_elem=1, tapped_bytes_fn=None, desc_bytes=48, out_elem=4):
    """per crop, write = cn*out_elem*dstW*dstH, read = distinct tapped source pixels * bytes per
    pixel, plus the per-crop descriptor.  `tapped_bytes_fn(sw, sh, dw, dh, ar, bpp)` overrides the tap census
    (tests pass the oracle's to cross-check this module's)."""
    write = cn * out_elem * dst[0] * dst[1]
    total = 0
    for (_, _, w, h) in crops:
        if tapped_bytes_fn is not None:
            read = tapped_bytes_fn(w, h, dst[0], dst[1], cvgs.IGNORE_AR, cn * src_elem)
        else:
            read = tapped_bytes(w, h, dst[0], dst[1], cn * src_elem)
        total += write + read + desc_bytes
    return total


def _tap_indices(src, n_out):
    """Sorted distinct source indices tapped along one axis (same rule as _distinct_taps)."""
    f = np.float32(1.0 / (float(n_out) / float(src)))
    a = np.floor(np.arange(n_out, dtype=np.float32) * f).astype(np.int64)
    b = np.minimum(a + 1, src - 1)
    return np.unique(np.concatenate([a, b]))


def nv12_sector_read_bytes(src_w, src_h, dst_w, dst_h, sample_bytes=1, sector=64):
    """The sector-granular READ bound of one whole-surface K4 launch (NV12: sample_bytes 1, P010: 2): distinct
    `sector`-byte pieces of the luma plane and of the interleaved chroma plane that hold a tapped sample.  Rows are taken as
    sector-aligned (step = src_w * sample_bytes, a multiple of 64 for decoder surfaces).  Every luma row tapped shares one
    column pattern, so the count is rows x sectors-per-row for each plane."""
    cols = _tap_indices(src_w, dst_w)
    rows = _tap_indices(src_h, dst_h)
    luma_secs = np.unique(np.concatenate([(cols * sample_bytes) // sector, (cols * sample_bytes + sample_bytes - 1) // sector]))
    pair = (cols // 2) * 2 * sample_bytes  # byte offset of the (U,V) pair of a luma column
    chroma_secs = np.unique(np.concatenate([pair // sector, (pair + 2 * sample_bytes - 1) // sector]))
    chroma_rows = np.unique(rows // 2)
    return int(len(rows) * len(luma_secs) + len(chroma_rows) * len(chroma_secs)) * sector


def k1_sector_read_bytes(crops, frame_w, frame_h, dst=DST, px_bytes=3, sector=64, step=None):
    """The sector-granular READ bound of one K1 launch: bytes of the distinct `sector`-byte aligned pieces of the frame
    that hold at least one tapped pixel byte (union over the launch's crops -- overlapping crops share sectors).  HBM
    and the caches move whole sectors, so no kernel can read less than this; the gap between it and the algorithmic
    (distinct tapped bytes) figure is physics (sparse 3-byte taps at up to 8x downscale), the gap between it and the
    measured FETCH_SIZE is the kernel's.  The frame is taken as sector-aligned with dense rows (step = w * px_bytes)."""
    step = step or frame_w * px_bytes
    mark = np.zeros((frame_h, (step + sector - 1) // sector), dtype=bool)
    for (x0, y0, w, h) in crops:
        cols = _tap_indices(w, dst[0]) + x0
        rows = _tap_indices(h, dst[1]) + y0
        first = (cols * px_bytes) // sector
        last = (cols * px_bytes + px_bytes - 1) // sector
        secs = np.unique(np.concatenate([first, last]))
        mark[np.ix_(rows, secs)] = True
    return int(mark.sum()) * sector


def nv12_crops_sector_read_bytes(crops, surf_w, surf_h, dst=DST, sample_bytes=1, sector=64):
    """Sector-granular READ bound of one K4 launch over crops of ONE NV12 / P010 surface (the decode-side cfg #2b): distinct sectors of
    the luma plane and of the interleaved chroma plane holding a tapped sample, union over the crops (rows dense: step = w * sample_bytes)."""
    step = surf_w * sample_bytes
    ns = (step + sector - 1) // sector
    luma = np.zeros((surf_h, ns), dtype=bool)
    chroma = np.zeros((surf_h // 2 + 1, ns), dtype=bool)
    for (x0, y0, w, h) in crops:
        cols = _tap_indices(w, dst[0]) + x0
        rows = _tap_indices(h, dst[1]) + y0
        lb = cols * sample_bytes
        luma[np.ix_(rows, np.unique(np.concatenate([lb // sector, (lb + sample_bytes - 1) // sector])))] = True
        pb = (cols // 2) * 2 * sample_bytes
        chroma[np.ix_(np.unique(rows // 2), np.unique(np.concatenate([pb // sector, (pb + 2 * sample_bytes - 1) // sector])))] = True
    return int(luma.sum() + chroma.sum()) * sector


def rotation_units(read_touched_bytes_per_unit, factor=2.0, minimum=8, requested=0):
    """How many distinct frames / surfaces a benchmark must rotate through so that NONE of what a launch reads can still be in
    the 256 MiB Infinity Cache on its next turn: the rule is on the bytes a launch actually TOUCHES on the read side (the distinct
    64-byte sectors that hold a tapped byte), not on whole-frame sizes -- the READ-touched set of the rotation alone is >= `factor`
    x the cache, so the rule holds even if streaming (nt) stores never allocate there ("working sets that defeat the
    256 MB Infinity Cache"; VERDICT r4 "What's weak" #2: 20 whole 4K frames are 596 MB but only 257 MB of touched sectors)."""
    if requested:
        return int(requested)
    return int(max(minimum, -(-int(factor * LLC_BYTES) // max(1, int(read_touched_bytes_per_unit)))))


def k1_touched_per_frame(n_crops, frame_wh, rank=0, out_elem=4, cn=3, sample=8):
    """(read-touched, written) bytes of ONE K1 launch over `n_crops` cfg #2b crops of one frame: mean over `sample` of the crop
    lists bench.py's Workload generates (seed rule below), the sector census of k1_sector_read_bytes."""
    fw, fh = frame_wh
    rd = float(np.mean([k1_sector_read_bytes(random_crops(n_crops, fw, fh, seed=SEED + 1000 * rank + f + 500000), fw, fh, px_bytes=cn)
                        for f in range(sample)]))
    return rd, float(n_crops * cn * out_elem * DST[0] * DST[1])


def residency(units, read_touched_per_unit, written_per_unit, whole_unit_bytes=None):
    """The block bench.py prints beside a roofline figure: how large the rotation's touched set is against the Infinity Cache."""
    r = {"frames": int(units), "touched_MB": round(units * (read_touched_per_unit + written_per_unit) / 1e6, 1),
         "read_touched_MB": round(units * read_touched_per_unit / 1e6, 1), "llc_MB": round(LLC_BYTES / 1e6, 1)}
    if whole_unit_bytes:
        r["whole_frames_MB"] = round(units * whole_unit_bytes / 1e6, 1)
    return r
