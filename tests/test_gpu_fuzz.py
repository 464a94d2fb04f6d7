"""Seeded differential fuzzing of the whole C-ABI: random VALID chains (read kind x source type x batch / unused planes x
geometry x pointwise program x write kind), each run on the CPU oracle and on the GPU and compared bit for bit.  The
dispatcher picks whatever kernel matches (fast or interpreted); half of the cases are also forced onto the interpreted
kernels.  Catches disagreements between kernels on chain shapes no hand-written test spells."""
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H
from tests import kat_runner as K
from tests.test_gpu_chains import _both, _random_src

pytestmark = pytest.mark.gpu

NP = {cvgs.CV_64F: np.float64, cvgs.CV_8U: np.uint8, cvgs.CV_8S: np.int8, cvgs.CV_16U: np.uint16, cvgs.CV_16S: np.int16, cvgs.CV_32S: np.int32,
      cvgs.CV_32F: np.float32, cvgs.CV_16F: np.float16}
NAME = {cvgs.CV_8U: "8U", cvgs.CV_8S: "8S", cvgs.CV_16U: "16U", cvgs.CV_16S: "16S", cvgs.CV_32S: "32S", cvgs.CV_32F: "32F", cvgs.CV_64F: "64F",
        cvgs.CV_16F: "16F"}


def _program(rng, depth, cn):
    """random type-correct pointwise stages; returns (iops, final_depth, final_cn)"""
    ops = []
    T = lambda d, c: cvgs.make_type(d, c)  # noqa: E731
    for _ in range(int(rng.integers(0, 6))):
        if sum(len(o.ops) for o in ops) > 7:  # leave room: a stage lowers to at most 4 ops, the chain holds 12
            break
        choice = rng.integers(0, 7)
        if choice == 6 and depth in (cvgs.CV_32F, cvgs.CV_8U, cvgs.CV_16S, cvgs.CV_32S):  # a detour through double precision
            ops.append(cvgs.convertTo(T(depth, cn), T(cvgs.CV_64F, cn), 1.0 / 3.0, 0.1))
            ops.append(cvgs.divide(T(cvgs.CV_64F, cn), [float(v) for v in rng.uniform(0.3, 3.0, cn)]))
            depth = cvgs.CV_64F
            if rng.integers(0, 3):
                ops.append(cvgs.convertTo(T(cvgs.CV_64F, cn), T(cvgs.CV_32F, cn)))
                depth = cvgs.CV_32F
            continue
        if depth == cvgs.CV_64F:
            nd = [cvgs.CV_32F, cvgs.CV_16S, cvgs.CV_8U][int(rng.integers(0, 3))]
            ops.append(cvgs.convertTo(T(depth, cn), T(nd, cn)))
            depth = nd
            continue
        if choice == 0 and depth != cvgs.CV_32F:
            ops.append(cvgs.convertTo(T(depth, cn), T(cvgs.CV_32F, cn)))
            depth = cvgs.CV_32F
        elif choice in (1, 2) and depth == cvgs.CV_32F:
            fn = [cvgs.multiply, cvgs.add, cvgs.subtract, cvgs.divide][int(rng.integers(0, 4))]
            vals = [float(np.float32(v)) for v in rng.uniform(0.3, 3.0, cn)]
            ops.append(fn(T(depth, cn), vals))
        elif choice in (1, 2) and depth in (cvgs.CV_8U, cvgs.CV_8S, cvgs.CV_16U, cvgs.CV_16S, cvgs.CV_32S):
            # arithmetic on an integer-typed value (cvGS::multiply<CV_8UC3> ...): integer operands incl. zero divisors, negative
            # and fractional scalars (truncated by the facade's conversion), results that saturate
            fn = [cvgs.multiply, cvgs.add, cvgs.subtract, cvgs.divide][int(rng.integers(0, 4))]
            pool = [0.0, 1.0, -1.0, 2.0, 3.0, -3.0, 7.5, -2.25, 100.0, 300.0, -40000.0, 70000.0, 3e9]
            ops.append(fn(T(depth, cn), [pool[int(rng.integers(0, len(pool)))] for _ in range(cn)]))
        elif choice == 3 and depth in (cvgs.CV_8U, cvgs.CV_16U, cvgs.CV_32F) and cn in (3, 4):
            codes = {3: [("BGR2RGB", 3), ("BGR2BGRA", 4), ("RGB2BGRA", 4), ("BGR2GRAY", 1), ("RGB2GRAY", 1)],
                     4: [("RGBA2BGRA", 4), ("BGRA2BGR", 3), ("RGBA2BGR", 3), ("BGRA2GRAY", 1), ("RGBA2GRAY", 1)]}[cn]
            code, ocn = codes[int(rng.integers(0, len(codes)))]
            ops.append(cvgs.cvtColor(getattr(cvgs, "COLOR_" + code), T(depth, cn), T(depth, ocn)))
            cn = ocn
        elif choice == 4 and depth == cvgs.CV_32F:
            nd = [cvgs.CV_8U, cvgs.CV_8S, cvgs.CV_16U, cvgs.CV_16S, cvgs.CV_32S, cvgs.CV_16F][int(rng.integers(0, 6))]
            if rng.integers(0, 2) and nd != cvgs.CV_16F:
                ops.append(cvgs.convertTo(T(depth, cn), T(nd, cn), float(np.float32(rng.uniform(0.2, 2.0))), float(np.float32(rng.uniform(-20, 20)))))
            else:
                ops.append(cvgs.convertTo(T(depth, cn), T(nd, cn)))
            depth = nd
        elif choice == 5 and depth == cvgs.CV_32F:
            nd = [cvgs.CV_8U, cvgs.CV_16S, cvgs.CV_32S][int(rng.integers(0, 3))]
            ops.append(cvgs.cast(T(depth, cn), T(nd, cn)))  # fk::Cast: truncation
            depth = nd
        if sum(len(o.ops) for o in ops) > 9:
            break
    return ops, depth, cn


def _case(seed, big=False, tiers=False):
    rng = np.random.default_rng(seed)
    k = 8 if big else 1  # big: whole-frame sizes (4 rows per wave, full 256-pixel groups, shuffled / LDS-transposed stores)
    kind = ["pixel", "resize", "resize", "warp", "nv12"][int(rng.integers(0, 5))]
    n = int(rng.integers(1, 4 if big else 6))
    if tiers:  # batch sizes around the descriptor thresholds (kernel arguments 52 / 64 / 320 planes, 16 destination planes, tables beyond)
        n = [16, 17, 52, 53, 64, 65, 130, 320, 321][int(rng.integers(0, 9))]
    used = n if rng.integers(0, 3) else int(rng.integers(0, n + 1))
    yuv_layout = int(rng.integers(0, 5))  # NV12 / NV21 / I420 / YV12 / P010
    if kind == "nv12":
        sdepth, scn = (cvgs.CV_16U if yuv_layout == capi.YUV_P010 else cvgs.CV_8U), 1
        sw, sh = 2 * int(rng.integers(2, 60 * k)), 2 * int(rng.integers(2, 40 * k))
        if tiers:
            sw, sh = 2 * int(rng.integers(2, 20)), 2 * int(rng.integers(2, 12))
        # P010: random 16-bit samples, i.e. 10-bit codes with garbage in the 6 low bits (which must be ignored)
        srcs = [(H.random_u16 if sdepth == cvgs.CV_16U else H.random_u8)((sh + sh // 2, sw, 1), seed * 10 + i) for i in range(n)]
        used = n
    else:
        # one case in five draws a CV_64F / CV_16F source (per-pixel and resize reads; CV_16F also warps)
        pool = [cvgs.CV_8U, cvgs.CV_8U, cvgs.CV_8S, cvgs.CV_16U, cvgs.CV_16S, cvgs.CV_32S, cvgs.CV_32F, cvgs.CV_16F, cvgs.CV_16F if kind == "warp" else cvgs.CV_64F]
        sdepth = pool[int(rng.integers(0, 7))] if rng.integers(0, 5) else pool[int(rng.integers(7, 9))]
        scn = int(rng.integers(1, 5))
        sw, sh = int(rng.integers(1, 300 * k)), int(rng.integers(1, 60 * k))
        if tiers:
            sw, sh = int(rng.integers(1, 40)), int(rng.integers(1, 24))
        srcs = [_random_src((sh, sw, scn), NAME[sdepth], seed * 10 + i) for i in range(n)]
    st = cvgs.make_type(sdepth, scn)
    dw, dh = (sw, sh) if kind == "pixel" else (int(rng.integers(1, 200 * k)), int(rng.integers(1, 150 * k)))
    if tiers and kind != "pixel":
        dw, dh = int(rng.integers(1, 70)), int(rng.integers(1, 20))
    alpha = bool(rng.integers(0, 2))
    ar = [cvgs.IGNORE_AR, cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_LEFT, cvgs.PRESERVE_AR_RN_EVEN][int(rng.integers(0, 4))] if kind == "resize" else cvgs.IGNORE_AR
    bg = [float(v) for v in rng.integers(0, 100, 4)]
    if kind == "pixel":
        d0, c0 = sdepth, scn
    elif kind == "nv12":
        d0, c0 = cvgs.CV_32F, (4 if alpha else 3)
    else:
        d0, c0 = cvgs.CV_32F, scn
    prog, fd, fc = _program(rng, d0, c0)
    ft = cvgs.make_type(fd, fc)
    nv12_resize = bool(rng.integers(0, 2))
    crop_views = None
    if kind == "nv12" and rng.integers(0, 2):
        crop_views = []
        for _ in range(n):
            cw, ch = 2 * int(rng.integers(1, sw // 2 + 1)), 2 * int(rng.integers(1, sh // 2 + 1))
            crop_views.append((2 * int(rng.integers(0, (sw - cw) // 2 + 1)), 2 * int(rng.integers(0, (sh - ch) // 2 + 1)), cw, ch))
    if kind == "nv12" and not nv12_resize:
        dw, dh = sw, sh
    wkinds = ["write3d", "write2d_batch"]
    if n == 1:
        wkinds.append("write2d")
    if fc >= 2:
        wkinds += ["split", "splitT", "split2d"]
    wk = wkinds[int(rng.integers(0, len(wkinds)))]
    pitch_pad = int(rng.integers(0, 4))
    mats_persp = None
    if kind == "warp":
        persp = bool(rng.integers(0, 2))
        mats_persp = []
        for _ in range(n):
            m = np.eye(3)
            m[:2, :2] += rng.uniform(-0.4, 0.4, (2, 2))
            m[:2, :2] *= rng.uniform(0.3, 2.0) * max(dw, 1) / max(sw, 1)
            m[:2, 2] = rng.uniform(-10, 10, 2)
            if persp:
                m[2, :2] = rng.uniform(-0.002, 0.002, 2)
            mats_persp.append(m if persp else m[:2])

    # batched warps into one image / set of planes per plane may give every plane its own (smaller) destination size
    warp_sizes = None
    if kind == "warp" and wk in ("write2d_batch", "split2d") and n > 1 and rng.integers(0, 2):
        warp_sizes = [(int(rng.integers(1, dw + 1)), int(rng.integers(1, dh + 1))) for _ in range(n)]
    esz = np.dtype(NP[fd]).itemsize
    if wk in ("write3d",):
        shape = (n, dw * dh, fc)
    elif wk == "write2d":
        shape = (dh, dw + pitch_pad, fc)
    elif wk == "write2d_batch":
        shape = (n * dh, dw + pitch_pad, fc)
    elif wk == "split":
        shape = (n, fc * dw * dh)
    elif wk == "splitT":
        shape = (fc * n, dw * dh)
    else:
        shape = (n * fc * dh, dw + pitch_pad)

    def build(wrap, wrap_out, out):
        mats = [wrap(s, st) for s in srcs]
        if kind == "pixel":
            rd = cvgs.ReadIOp(capi.READ_PIXEL, st, mats, used, None, cvgs.IGNORE_AR, bg[:scn])
        elif kind == "resize":
            rd = cvgs.resize(st, cvgs.INTER_LINEAR, mats, (dw, dh), used, bg[:scn], ar)
        elif kind == "warp":
            rd = cvgs.warp(cvgs.WARP_PERSPECTIVE if mats_persp[0].shape[0] == 3 else cvgs.WARP_AFFINE, st, mats,
                           [m.tolist() for m in mats_persp], warp_sizes or (dw, dh), max(used, 1) if used == 0 else used, bg[:scn])
        else:
            lumas = [cvgs.GpuMat(sh, sw, st, m.data, m.step, owner=m.owner) for m in mats]
            layout = yuv_layout
            if nv12_resize and n > 1 and crop_views:  # N crop views (own luma -> chroma offsets) of ONE decoder surface
                lumas = [lumas[0].nv12_roi(*c) for c in crop_views]
                layout = yuv_layout if yuv_layout == capi.YUV_P010 else yuv_layout & 1  # crops exist for the interleaved layouts only
            rd = cvgs.read_nv12(lumas if n > 1 else lumas[0], (dw, dh) if nv12_resize else None,
                                int(rng_choice[0]), int(rng_choice[1]), alpha, layout=layout)
        if use_table and kind in ("pixel", "resize") and used == n and type(mats[0].owner).__module__.startswith("torch"):
            # GPU side only: the crop list as a resident device plane table instead of kernel-argument descriptors
            import torch
            tab = torch.frombuffer(bytearray(cvgs.build_plane_table(rd)), dtype=torch.uint8).cuda()
            rd.table, rd.table_keep = tab.data_ptr(), tab
            rd.dsize = (dw, dh)  # a table carries no extent: the descriptor must
        o_t = cvgs.make_type(fd, 1) if wk in ("split", "splitT", "split2d") else ft
        o = wrap_out(out, o_t)
        if wk == "write3d":
            wr = cvgs.write(ft, o, (dw, dh))
        elif wk == "write2d":
            wr = cvgs.write(ft, o.roi(0, 0, dw, dh))
        elif wk == "write2d_batch":
            zs = warp_sizes or [(dw, dh)] * n  # smaller planes sit in the top-left corner of their dh x dw slot
            wr = cvgs.write_batch(ft, [cvgs.GpuMat(zs[i][1], zs[i][0], ft, o.data + i * dh * o.step, o.step, owner=o) for i in range(n)])
        elif wk == "split":
            wr = cvgs.split(ft, o, (dw, dh))
        elif wk == "splitT":
            wr = cvgs.splitT(ft, o.data, dw, dh, n, keep=o)
        else:
            zs = warp_sizes or [(dw, dh)] * n
            planes = [[cvgs.GpuMat(zs[z][1], zs[z][0], o_t, o.data + ((z * fc + c) * dh) * o.step, o.step, owner=o) for c in range(fc)] for z in range(n)]
            wr = cvgs.split(ft, planes if n > 1 else planes[0])
        return [rd] + prog + [wr]

    rng_choice = (rng.integers(0, 2), rng.integers(0, 3))  # range; BT.601 / BT.709 / BT.2020
    use_table = bool(rng.integers(0, 3) == 0)
    return build, shape, NP[fd], "%s n=%d used=%d %sC%d %dx%d->%dx%d ops=%d %s out=%s" % (
        kind, n, used, NAME[sdepth], scn, sw, sh, dw, dh, sum(len(o.ops) for o in prog), wk, np.dtype(NP[fd]).name)


def _run(seed, big, flags, tiers=False):
    build, shape, dt, what = _case(seed, big=big, tiers=tiers)
    # every chain the generator can spell is served (round 2 closed the last two refusals: warps and fp16 next to CV_64F
    # values); a refusal is a failure
    gpu, ref = _both(build, shape, dt, flags=flags)
    g, r = gpu[0], ref[0]
    if dt in (np.float32, np.float16, np.float64):  # NaN payloads may differ; everything else must be the same bits
        gn, rn = np.isnan(g), np.isnan(r)
        assert np.array_equal(gn, rn), what
        g, r = np.where(gn, 0, g), np.where(rn, 0, r)
    H.assert_bit_exact(g, r, what)


_BASE = int(os.environ.get("CVGS_FUZZ_BASE", "0"))  # a long hunt in several runs: CVGS_FUZZ_BASE=1000000, 2000000, ... with CVGS_FUZZ_N=200000 each


@pytest.mark.parametrize("seed", range(_BASE, _BASE + int(os.environ.get("CVGS_FUZZ_N", "600"))))  # CVGS_FUZZ_N=20000 for a long hunt
def test_random_chain_matches_oracle(seed):
    _run(seed, False, [0, capi.CHAIN_FORCE_GENERIC, 0, capi.CHAIN_NO_THREAD_FUSION][seed % 4])


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVGS_FUZZ_BIG_N", "40"))))
def test_random_chain_whole_frame_sizes(seed):
    _run(500_000 + seed, True, 0)


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVGS_FUZZ_TIERS_N", "120"))))
def test_random_chain_around_the_descriptor_thresholds(seed):
    """the same generator with 16 ... 321 planes per chain (tiny images): kernel-argument descriptors (4 KB / 16 KB blocks),
    inline / table destination planes, pinned tables read in place."""
    _run(900_000 + seed, False, [0, capi.CHAIN_FORCE_GENERIC, 0][seed % 3], tiers=True)


def test_integer_values_have_no_negative_zero():
    """Found by the fuzzer (seed 91): float -> integer type -> float must give +0 for values in (-1, 0), for both the
    saturating cast (round to nearest even) and fk::Cast (truncation): integers have no -0."""
    vals = np.array([-0.9, -0.5, -0.3, -0.0, 0.0, 0.3, -1.0, -1.4], np.float32)
    src = np.tile(vals, (2, 8))[:, :, None].copy()
    for mk in (lambda a, b: cvgs.convertTo(a, b), lambda a, b: cvgs.cast(a, b)):
        for depth in (cvgs.CV_8S, cvgs.CV_16S, cvgs.CV_32S, cvgs.CV_8U):
            t = cvgs.make_type(depth, 1)

            def build(wrap, wrap_out, out):
                return [cvgs.ReadIOp(capi.READ_PIXEL, cvgs.CV_32FC1, [wrap(src, cvgs.CV_32FC1)], 1), mk(cvgs.CV_32FC1, t),
                        cvgs.convertTo(t, cvgs.CV_32FC1), cvgs.write(cvgs.CV_32FC1, wrap_out(out, cvgs.CV_32FC1))]

            for flags in (0, capi.CHAIN_FORCE_GENERIC):
                gpu, ref = _both(build, src.shape, np.float32, flags=flags)
                H.assert_bit_exact(gpu[0], ref[0], "no -0 through depth %d" % depth)
                assert not np.signbit(ref[0][ref[0] == 0]).any()


@pytest.mark.parametrize("seed", range(int(os.environ.get("CVGS_FUZZ_CIRCULAR_N", "60"))))
def test_random_circular_tensor_sequence(oracle, seed):
    """CircularTensor fuzz: order x plane mode x mirrored ring x depth x element type x write kind x push chain (per-pixel,
    resize, fp16 / u8 / fp32 elements); the whole tensor is compared with the oracle's after EVERY update."""
    import torch
    from tests.test_gpu_circular_nv12 import _read_device
    rng = np.random.default_rng(10_000 + seed)
    dev = torch.device("cuda:0")
    W, H_ = int(rng.integers(1, 300)), int(rng.integers(1, 40))
    B = int(rng.integers(1, 7))
    order = [cvgs.NewestFirst, cvgs.OldestFirst][int(rng.integers(0, 2))]
    cn = int(rng.integers(1, 5))
    packed = bool(rng.integers(0, 2)) or cn == 1
    mirrored = bool(rng.integers(0, 2))
    transposed = (not packed) and (not mirrored) and bool(rng.integers(0, 2))
    mode = cvgs.Transposed if transposed else cvgs.Standard
    edepth = [cvgs.CV_32F, cvgs.CV_32F, cvgs.CV_16F, cvgs.CV_8U, cvgs.CV_64F][int(rng.integers(0, 5))]
    resize_push = bool(rng.integers(0, 2))
    u, f = cvgs.make_type(cvgs.CV_8U, cn), cvgs.make_type(cvgs.CV_32F, cn)
    et = cvgs.make_type(edepth, cn)
    elem_type = et if packed else cvgs.make_type(edepth, 1)
    cp = 1 if packed else cn
    ct = cvgs.CircularTensor(u, elem_type, cp, B, order, mode, W, H_, mirrored=mirrored)
    oc = oracle.OracleCircular(W, H_, elem_type, cp, B, order, mode)
    s = torch.cuda.current_stream()
    for i in range(B + 2):
        sw, sh = (int(rng.integers(1, 200)), int(rng.integers(1, 60))) if resize_push else (W, H_)
        frame = H.random_u8((sh, sw, cn), seed=70_000 + seed * 100 + i)
        frame_t = torch.from_numpy(frame).to(dev)
        pw = [] if resize_push else [cvgs.convertTo(u, f)]
        pw += [cvgs.multiply(f, [0.5] * cn), cvgs.add(f, [3.25] * cn)]
        if edepth != cvgs.CV_32F:
            pw.append(cvgs.convertTo(f, et))

        def chain(mat, wr):
            rd = cvgs.resize(u, cvgs.INTER_LINEAR, mat, (W, H_)) if resize_push else cvgs.ReadIOp(capi.READ_PIXEL, u, [mat], 1)
            return [rd] + pw + [wr]

        wr_g = ct.write_packed(et) if packed else (ct.write_splitT(et) if transposed else ct.write_split(et))
        ct.update(s, *chain(cvgs.GpuMat.from_tensor(frame_t, u), wr_g))
        kind = capi.WRITE_PIXEL_3D if packed else (capi.WRITE_TENSOR_T_SPLIT if transposed else capi.WRITE_TENSOR_SPLIT)
        oc.update(cvgs.lower(chain(cvgs.GpuMat.from_array(frame, u), cvgs.WriteIOp(kind, et, 16, W, H_, 0, B))))
        torch.cuda.synchronize()
        got = _read_device(ct.data(), ct.nbytes())
        want = oc.array(np.uint8)
        what = "seed %d update %d: %dx%d B=%d cn=%d %s %s %s elem=%d %s" % (seed, i, W, H_, B, cn, "packed" if packed else "planar",
                                                                      "T" if transposed else "S", "mirrored" if mirrored else "ring",
                                                                      edepth, "resize" if resize_push else "pixel")
        assert np.array_equal(got, want), what
    ct.release()
