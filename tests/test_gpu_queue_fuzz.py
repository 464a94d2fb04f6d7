"""Random batches through the device-side descriptor queue at random queue depths.  The task size a batch is cut into follows the
number of batches in flight when it is submitted (4 / 16 / 64 / 128 rows per task, csrc/k_queue.hip: queue_submit_slot), and the
workers draw tasks from ticket counters: every tested batch is submitted behind 0 ... 120 filler batches (one submit_many call), so
that all four task sizes meet ragged target sizes (rows not a multiple of the task, one-row and one-column targets), every source
kind (8U / 16U / 16S C3 / C4 crops, NV12 / NV21 surface crops, P010 surface crops), aspect-ratio windows, default planes, fp16 tensors and batches larger
than a ring slot.  Each result must equal cvgs_execute's bit for bit (180,000 fuzzed chains pin THAT to the oracle, test_gpu_fuzz.py),
every fourth one is checked against the oracle directly.  CVGS_QUEUE_FUZZ_N sets the number of trials per kind (default 40)."""
import os

import numpy as np
import pytest

from cvgpuspeedup_amd import capi, cvgs
from tests import helpers as H

pytestmark = pytest.mark.gpu
N = int(os.environ.get("CVGS_QUEUE_FUZZ_N", "40"))
FW, FH = 1920, 1080


def _pixel_ops(src_mat, crops, out_mat, dst, cn, sd, kw):
    return H.k1_chain(src_mat, crops, out_mat, dst, cn, src_depth=sd, **kw)


def _nv12_ops(lumas, out_mat, dst, kw):
    f = cvgs.CV_32FC3
    rd = cvgs.read_nv12(lumas, dst, kw["range_"], kw["prim"], False, layout=kw["layout"])
    if "ar" in kw:
        rd.ar = kw["ar"]
        rd.background = cvgs._scalar(kw["background"])
    if "used" in kw:
        rd.used_planes = kw["used"]
        rd.background = cvgs._scalar(kw["background"])
    ops = [rd]
    if kw.get("swap", True):
        ops.append(cvgs.cvtColor(cvgs.COLOR_RGB2BGR, f))
    ops += [cvgs.multiply(f, [1 / 1023.0 if kw["layout"] == capi.YUV_P010 else 1 / 255.0] * 3), cvgs.subtract(f, [0.485, 0.456, 0.406]), cvgs.divide(f, [0.229, 0.224, 0.225])]
    if kw.get("half"):
        ops += [cvgs.convertTo(f, cvgs.CV_16FC3), cvgs.split(cvgs.CV_16FC3, out_mat, dst)]
    else:
        ops.append(cvgs.split(f, out_mat, dst))
    return ops


@pytest.mark.parametrize("kind", ["8U", "16U", "16S", "NV12", "P010"])
def test_queue_fuzz(oracle, kind):
    import torch
    dev = torch.device("cuda:0")
    rng = np.random.default_rng({"8U": 11, "16U": 12, "16S": 13, "NV12": 14, "P010": 15}[kind])
    sd = {"8U": cvgs.CV_8U, "16U": cvgs.CV_16U, "16S": cvgs.CV_16S, "NV12": cvgs.CV_8U, "P010": cvgs.CV_16U}[kind]
    yuv = kind in ("NV12", "P010")
    ytype, ystep = (cvgs.CV_16UC1, 2 * FW) if kind == "P010" else (cvgs.CV_8UC1, FW)
    frames = {}
    for cn in ((3, 4) if not yuv else (1,)):
        if kind == "NV12":
            a = H.random_u8((FH + FH // 2, FW), seed=500)
        elif kind == "P010":  # 10-bit codes in the high bits, dirty low bits
            a = H.random_u16((FH + FH // 2, FW), seed=499).view(np.uint16)
        elif kind == "8U":
            a = H.random_u8((FH, FW, cn), seed=501 + cn)
        else:
            a = H.random_u16((FH, FW, cn), seed=503 + cn).view(np.uint16 if kind == "16U" else np.int16)
        frames[cn] = (a, torch.from_numpy(a.view(np.int16) if kind in ("16U", "16S", "P010") else a).to(dev))
    # the filler: a 40-crop batch of the same kind into a scratch tensor
    q = cvgs.Queue(depth=int(rng.choice([16, 64, 128])))
    try:
        fcn = 3 if not yuv else 1
        fill_out = torch.zeros((40, 3 * 64 * 128), dtype=torch.float32, device=dev)
        if yuv:
            luma = cvgs.GpuMat(FH, FW, ytype, frames[1][1].data_ptr(), ystep, owner=frames[1][1])
            fcrops = [tuple(v & ~1 for v in c) for c in H.random_crops(40, FW, FH, seed=77, wmin=9, wmax=500, hmin=9, hmax=500)]
            filler = cvgs.lower(_nv12_ops([luma.nv12_roi(*c) for c in fcrops], cvgs.GpuMat.from_tensor(fill_out, cvgs.CV_32FC1), (64, 128),
                                          dict(range_=capi.YUV_FULL, prim=capi.BT709, layout=capi.YUV_P010 if kind == "P010" else capi.YUV_NV12)))
        else:
            filler = cvgs.lower(_pixel_ops(cvgs.GpuMat.from_tensor(frames[3][1], cvgs.make_type(sd, 3)), H.random_crops(40, FW, FH, wmax=500, hmax=500, seed=77),
                                           cvgs.GpuMat.from_tensor(fill_out, cvgs.CV_32FC1), (64, 128), 3, sd, {}))
        for trial in range(N):
            cn = int(rng.choice([3, 4])) if not yuv else 3
            n = int(rng.choice([1, 2, 7, 33, 50, 80, 150])) if trial % 5 else int(rng.integers(1, 100))
            dw = int(rng.choice([1, 3, 63, 64, 65, 100, 160, 256])) if rng.uniform() < 0.5 else int(rng.integers(1, 200))
            dh = int(rng.choice([1, 4, 5, 16, 17, 63, 64, 65, 127, 128, 129, 200, 257])) if rng.uniform() < 0.6 else int(rng.integers(1, 300))
            while n * cn * dw * dh > 6_000_000:
                n = max(1, n // 2)
            kw = {}
            mode = rng.integers(0, 4)
            bgv = [float(v) for v in rng.uniform(0.5, 200.0, 4)]
            if mode == 1:
                kw.update(ar=[cvgs.PRESERVE_AR, cvgs.PRESERVE_AR_RN_EVEN, cvgs.PRESERVE_AR_LEFT][int(rng.integers(0, 3))], background=bgv[:3] if yuv else bgv)
            elif mode == 2:
                kw.update(used=int(rng.integers(0, n + 1)), background=bgv[:3] if yuv else bgv)
            if rng.uniform() < 0.3:
                kw["swap"] = False
            half = rng.uniform() < 0.25
            if half:
                kw["half"] = True
            t_out = cvgs.CV_16FC1 if half else cvgs.CV_32FC1
            tdt = torch.float16 if half else torch.float32
            vcn = cn
            if yuv:
                crops = [tuple(max(2, v & ~1) if i >= 2 else v & ~1 for i, v in enumerate(c)) for c in H.random_crops(n, FW, FH, seed=9000 + trial, wmin=5, wmax=700, hmin=3, hmax=700)]
                crops = [(x, y, max(4, w), h) for x, y, w, h in crops]
                crops = [(min(x, FW - w) & ~1, min(y, FH - h) & ~1, w, h) for x, y, w, h in crops]
                kw.update(range_=int(rng.choice([capi.YUV_FULL, capi.YUV_LIMITED])), prim=int(rng.choice([capi.BT601, capi.BT709, capi.BT2020])),
                          layout=capi.YUV_P010 if kind == "P010" else int(rng.choice([capi.YUV_NV12, capi.YUV_NV21])))
                luma = cvgs.GpuMat(FH, FW, ytype, frames[1][1].data_ptr(), ystep, owner=frames[1][1])
                build = lambda out_mat, host=False: _nv12_ops([(cvgs.GpuMat(FH, FW, ytype, frames[1][0].ctypes.data, ystep, owner=frames[1][0]) if host else luma).nv12_roi(*c) for c in crops], out_mat, (dw, dh), kw)
            else:
                crops = H.random_crops(n, FW, FH, seed=9000 + trial, wmin=3, wmax=700, hmin=1, hmax=700)
                pkw = {k: v for k, v in kw.items()}
                build = lambda out_mat, host=False: _pixel_ops(cvgs.GpuMat.from_array(frames[cn][0], cvgs.make_type(sd, cn)) if host else
                                                               cvgs.GpuMat.from_tensor(frames[cn][1], cvgs.make_type(sd, cn)), crops, out_mat, (dw, dh), cn, sd, pkw)
            got_t = torch.full((n, vcn * dw * dh), -9.0, dtype=tdt, device=dev)
            want_t = torch.full((n, vcn * dw * dh), -9.0, dtype=tdt, device=dev)
            ops_q = build(cvgs.GpuMat.from_tensor(got_t, t_out))
            cvgs.executeOperations(torch.cuda.current_stream(), *build(cvgs.GpuMat.from_tensor(want_t, t_out)))
            torch.cuda.synchronize()
            lowered = cvgs.lower(ops_q)
            fill = int(rng.choice([0, 0, 4, 30, 120]))
            chains = [filler] * fill + [lowered]
            q.wait(q.submit_many(cvgs.Queue.chain_pointers(chains), len(chains)))
            view = np.uint16 if half else np.uint32
            what = "%s trial %d: %d crops cn %d -> %dx%d %s behind %d fillers" % (kind, trial, n, cn, dw, dh, {k: v for k, v in kw.items() if k != "background"}, fill)
            H.assert_bit_exact(got_t.cpu().numpy().view(view), want_t.cpu().numpy().view(view), what + " (vs cvgs_execute)")
            if trial % 4 == 0:
                ref = np.full((n, vcn * dw * dh), -9.0, dtype=np.float16 if half else np.float32)
                oracle.execute(cvgs.lower(build(cvgs.GpuMat.from_array(ref, t_out), host=True)))
                H.assert_bit_exact(got_t.cpu().numpy().view(view), ref.view(view), what + " (vs the oracle)")
        assert q.stats()["error"] == 0
    finally:
        q.destroy()
